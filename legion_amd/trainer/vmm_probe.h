// vmm_probe.h -- which calling convention hipMemImportFromShareableHandle of the HIP runtime in THIS process has for a POSIX file
// descriptor.  Shared by the trainer end (ipc_service.cpp, which runs on the runtime bundled with torch) and by
// tools/micro/vmm_convention_probe.cpp (built against /opt/rocm): tests/test_gpu_boundary.py runs both.
#pragma once
#include <cstdint>
#include <unistd.h>

#include <hip/hip_runtime_api.h>

// hipMemImportFromShareableHandle(handle, osHandle, PosixFileDescriptor): ROCm 7.2's runtime takes the descriptor BY VALUE in
// osHandle (as CUDA does), the runtime bundled with torch 2.10+rocm7.0 -- the one this module runs on inside a trainer --
// takes a POINTER to it and dereferences what it is given: the value convention on that runtime is a segmentation fault, not an
// error code.  Round 4 picked by hipRuntimeGetVersion() >= 70200000, i.e. guessed for every runtime it had not seen.  This
// probe cannot crash: it exports a chunk of its own and imports it by POINTER first -- a runtime that wants the value reads the
// pointer's low 32 bits as a descriptor number, finds none (the probe makes sure of that) and returns an error -- and tries the
// value only when the pointer was refused for a descriptor known to be good.  0 pointer, 1 value, -1 neither (no views).
static int g_fd_convention = -2;
static int vmm_fd_convention()
{
    if (g_fd_convention != -2) return g_fd_convention;
    g_fd_convention = -1;
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
        // where the descriptor's copy sits: an address whose low 32 bits are no descriptor of this process (>= 2^24 or negative).
        // A static array, then the stack, then the heap: one of them lies outside the 0.4 % of the address space that fails the test.
        static int cells[4096];
        int on_stack[2] = {-1, -1};
        int* on_heap = new int[1 << 23];          // (32 MB of address space, untouched: wider than the 16 MB window that fails the test)
        auto usable = [](const int* p) {
            const int32_t low = (int32_t)(uint32_t)(uintptr_t)p;
            return low < 0 || low >= (1 << 24);
        };
        int* cell = nullptr;
        for (int i = 0; i < 4096 && cell == nullptr; i++)
            if (usable(&cells[i])) cell = &cells[i];
        if (cell == nullptr && usable(&on_stack[0])) cell = &on_stack[0];
        for (int i = 0; i < (1 << 23) && cell == nullptr; i += 1 << 20)
            if (usable(&on_heap[i])) cell = &on_heap[i];
        hipMemGenericAllocationHandle_t h;
        bool pointer_refused = false;
        if (cell != nullptr) {
            *cell = fd;
            if (hipMemImportFromShareableHandle(&h, (void*)cell, hipMemHandleTypePosixFileDescriptor) == hipSuccess) {
                (void)hipMemRelease(h);
                g_fd_convention = 0;
            } else {
                (void)hipGetLastError();
                pointer_refused = true;
            }
        }
        delete[] on_heap;
        if (pointer_refused) {            // the pointer to a good descriptor was refused: this runtime reads osHandle as the value
                                          // (never tried blind: on a runtime that wants the pointer the value is a segmentation fault)
            if (hipMemImportFromShareableHandle(&h, (void*)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor) == hipSuccess) {
                (void)hipMemRelease(h);
                g_fd_convention = 1;
            } else {
                (void)hipGetLastError();
            }
        }
        close(fd);
    } else {
        (void)hipGetLastError();
    }
    (void)hipMemRelease(own);
    return g_fd_convention;
}

