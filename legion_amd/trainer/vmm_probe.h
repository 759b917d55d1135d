// vmm_probe.h -- which calling convention hipMemImportFromShareableHandle of the HIP runtime in THIS process has for a POSIX file
// descriptor.  Shared by the trainer end (ipc_service.cpp, which runs on the runtime bundled with torch), by the library
// (storage.hip: a clique member mapping another process's lane arena) and by tools/micro/vmm_convention_probe.cpp (built against
// /opt/rocm): tests/test_gpu_boundary.py runs both runtimes.
#pragma once
#include <atomic>
#include <cstdint>
#include <mutex>
#include <fcntl.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>

// hipMemImportFromShareableHandle(handle, osHandle, PosixFileDescriptor): ROCm 7.2's runtime takes the descriptor BY VALUE in
// osHandle (as CUDA does), the runtime bundled with torch 2.10+rocm7.0 -- the one this module runs on inside a trainer --
// takes a POINTER to it and dereferences what it is given: the value convention on that runtime is a segmentation fault, not an
// error code.  Round 4 picked by hipRuntimeGetVersion() >= 70200000, i.e. guessed for every runtime it had not seen.
//
// vmm_import_fd() finds out on the first descriptor it is given -- one that came from the exporting process, so that nothing is
// ever imported into the process that exported it -- and cannot crash: it passes a POINTER first, and the descriptor's copy sits at
// an address whose low 32 bits are no descriptor of this process.  A runtime that wants the value reads those bits as a descriptor
// number, finds none and returns an error.  The value is tried afterwards only when ALL of this holds (a pointer attempt
// also fails for honest reasons -- out of memory, no access to the exporter's GPU, a chunk the server already released -- and on a
// pointer runtime the value convention is a segmentation fault, not an error code):
//   * the convention is still unknown,
//   * the runtime reports a version from which the value convention is known (hipRuntimeGetVersion() >= 7.2: an independent prior;
//     the runtime bundled with torch reports 7.0 and is never handed a value),
//   * the pointer attempt did not fail for lack of memory,
//   * fcntl() says the descriptor is open.
// Otherwise the error goes back to the caller, which falls back to the pipe slots / the direct arrangement.
// One import at a time (the cell is shared): a mutex; g_fd_convention: -2 not known yet, 0 pointer, 1 value.
static std::atomic<int> g_fd_convention{-2};
static std::mutex g_fd_import_mu;

static int* vmm_safe_cell()
{
    // an address whose low 32 bits are no descriptor number (>= 2^24 or negative as an int): a static array first, then the heap
    // (32 MB of address space, untouched, kept: wider than the 16 MB window that fails the test)
    static int cells[4096];
    static int* chosen = nullptr;
    if (chosen != nullptr) return chosen;
    auto usable = [](const int* p) {
        const int32_t low = (int32_t)(uint32_t)(uintptr_t)p;
        return low < 0 || low >= (1 << 24);
    };
    for (int i = 0; i < 4096 && chosen == nullptr; i++)
        if (usable(&cells[i])) chosen = &cells[i];
    if (chosen == nullptr) {
        int* on_heap = new int[1 << 23];
        for (int i = 0; i < (1 << 23) && chosen == nullptr; i += 1 << 20)
            if (usable(&on_heap[i])) chosen = &on_heap[i];
    }
    return chosen;
}

static bool vmm_runtime_known_to_take_the_value()
{
    int v = 0;
    if (hipRuntimeGetVersion(&v) != hipSuccess) { (void)hipGetLastError(); return false; }
    return v >= 70200000;
}

static hipError_t vmm_import_fd(hipMemGenericAllocationHandle_t* h, int fd)
{
    std::lock_guard<std::mutex> lock(g_fd_import_mu);
    if (g_fd_convention.load() == 1) return hipMemImportFromShareableHandle(h, (void*)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor);
    int* cell = vmm_safe_cell();
    if (cell == nullptr) return hipErrorNotSupported;          // (cannot happen: see vmm_safe_cell)
    *cell = fd;
    hipError_t e = hipMemImportFromShareableHandle(h, (void*)cell, hipMemHandleTypePosixFileDescriptor);
    if (e == hipSuccess) { g_fd_convention.store(0); return e; }
    if (g_fd_convention.load() == 0) return e;                 // a runtime known to take the pointer refused this descriptor
    (void)hipGetLastError();
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) return e;      // an honest failure says nothing about the convention
    if (!vmm_runtime_known_to_take_the_value()) return e;      // never a value blind: the caller falls back
    if (fcntl(fd, F_GETFD) == -1) return e;                    // not an open descriptor: nothing learned
    e = hipMemImportFromShareableHandle(h, (void*)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor);
    if (e == hipSuccess) g_fd_convention.store(1);
    return e;
}

// For the test tools only (tools/micro/vmm_convention_probe.cpp, ipc_service.vmm_fd_convention() in a process of its own): the
// convention found on a chunk this process exports itself.  0 pointer, 1 value, -1 neither.  Product paths never call this -- they
// learn the convention from the first descriptor they receive (above) and import nothing of their own.
static int vmm_fd_convention()
{
    if (g_fd_convention.load() >= 0) return g_fd_convention.load();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) { (void)hipGetLastError(); return -1; }
    hipMemGenericAllocationHandle_t own;
    if (hipMemCreate(&own, gran, &prop, 0) != hipSuccess) { (void)hipGetLastError(); return -1; }
    int fd = -1;
    if (hipMemExportToShareableHandle(&fd, own, hipMemHandleTypePosixFileDescriptor, 0) == hipSuccess && fd >= 0) {
        hipMemGenericAllocationHandle_t h;
        if (vmm_import_fd(&h, fd) == hipSuccess) (void)hipMemRelease(h);
        else (void)hipGetLastError();
        close(fd);
    } else {
        (void)hipGetLastError();
    }
    (void)hipMemRelease(own);
    return g_fd_convention.load() >= 0 ? g_fd_convention.load() : -1;
}
