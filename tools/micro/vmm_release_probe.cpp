// When does a chunk that another process imported and mapped go back to the GPU?  (trainer end: finalize() must give the server's
// lane arena back)  A "server" process creates n exportable chunks and hands them over as file descriptors;
// a "trainer" process imports + maps them, the server exits (its own references die), and the trainer tears its mapping down in
// one of several orders, printing the GPU's free memory at every step:
//   order 0: hipMemRelease(import handle) right after hipMemMap; at the end hipMemUnmap + hipMemAddressFree
//   order 1: handles kept; at the end hipMemUnmap, hipMemRelease, hipMemAddressFree
//   order 2: as 1 but hipMemRelease BEFORE hipMemUnmap
//     hipcc -O2 tools/micro/vmm_release_probe.cpp -o /tmp/vmm_release_probe && /tmp/vmm_release_probe <order> [lives]
//   (LD_LIBRARY_PATH=<torch>/lib runs the same binary on the HIP runtime bundled with torch)
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../legion_amd/trainer/vmm_probe.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("[%d] %s -> %s\n", getpid(), #x, hipGetErrorString(e)); fflush(stdout); _exit(2); } } while (0)
static void send_fd(int sock, int fd)
{
    char dummy = 'x'; iovec io = {&dummy, 1};
    char ctl[CMSG_SPACE(sizeof(int))] = {};
    msghdr msg = {}; msg.msg_iov = &io; msg.msg_iovlen = 1; msg.msg_control = ctl; msg.msg_controllen = sizeof(ctl);
    cmsghdr* c = CMSG_FIRSTHDR(&msg); c->cmsg_level = SOL_SOCKET; c->cmsg_type = SCM_RIGHTS; c->cmsg_len = CMSG_LEN(sizeof(int));
    memcpy(CMSG_DATA(c), &fd, sizeof(int));
    if (sendmsg(sock, &msg, 0) < 0) { perror("sendmsg"); _exit(3); }
}
static int recv_fd(int sock)
{
    char dummy; iovec io = {&dummy, 1};
    char ctl[CMSG_SPACE(sizeof(int))] = {};
    msghdr msg = {}; msg.msg_iov = &io; msg.msg_iovlen = 1; msg.msg_control = ctl; msg.msg_controllen = sizeof(ctl);
    if (recvmsg(sock, &msg, 0) <= 0) { perror("recvmsg"); _exit(3); }
    int fd = -1; memcpy(&fd, CMSG_DATA(CMSG_FIRSTHDR(&msg)), sizeof(int));
    return fd;
}
static long long free_mib() { size_t f = 0, t = 0; CK(hipMemGetInfo(&f, &t)); return (long long)(f >> 20); }
// what THIS process holds in VRAM according to the driver: the amdgpu fdinfo of its drm / kfd descriptors ("drm-memory-vram: N KiB",
// one entry per drm client id), in MiB.  hipMemGetInfo answers from the runtime's own book-keeping and does not move when another
// process's chunks are mapped here.
#include <dirent.h>
#include <set>
static long long vram_used_mib()
{
    long long sum = 0;
    std::set<long long> seen;
    DIR* d = opendir("/proc/self/fdinfo");
    if (!d) return -1;
    while (dirent* e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        char path[512]; snprintf(path, sizeof(path), "/proc/self/fdinfo/%s", e->d_name);
        FILE* f = fopen(path, "r"); if (!f) continue;
        char line[256]; long long id = -1, vram = 0; bool has = false;
        while (fgets(line, sizeof(line), f)) {
            long long v;
            if (sscanf(line, "drm-client-id: %lld", &v) == 1) id = v;
            if (sscanf(line, "drm-memory-vram: %lld", &v) == 1) { vram = v; has = true; }
        }
        fclose(f);
        if (has && seen.insert(id).second) sum += vram >> 10;
    }
    closedir(d);
    return sum;
}

int main(int argc, char** argv)
{
    const int order = argc > 1 ? atoi(argv[1]) : 0, lives = argc > 2 ? atoi(argv[2]) : 3;
    const size_t g = 128ull << 20; const int n = 3;
    // every "server life" is a child forked BEFORE this process touches the GPU; the trainer (this process) outlives them
    std::vector<int> socks(lives);
    std::vector<pid_t> pids(lives);
    for (int l = 0; l < lives; l++) {
        int sv[2]; socketpair(AF_UNIX, SOCK_STREAM, 0, sv);
        pid_t pid = fork();
        if (pid == 0) {                                   // server life l: waits for "go", creates + exports, waits for "done", exits
            close(sv[0]);
            char go; if (read(sv[1], &go, 1) != 1) _exit(4);
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
            for (int i = 0; i < n; i++) {
                hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, g, &prop, 0));
                int fd = -1; CK(hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0));
                send_fd(sv[1], fd); close(fd);
            }
            char done; if (read(sv[1], &done, 1) != 1) _exit(4);
            _exit(0);
        }
        close(sv[1]); socks[l] = sv[0]; pids[l] = pid;
    }
    CK(hipSetDevice(0));
    int rt = 0; (void)hipRuntimeGetVersion(&rt);
    const int conv = vmm_fd_convention();
    printf("runtime %d, fd convention %d, order %d\n", rt, conv, order);
    if (conv < 0) return 2;
    hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    const long long base = free_mib(), vram0 = vram_used_mib();
    printf("baseline free %lld MiB\n", base);
    for (int l = 0; l < lives; l++) {
        char go = 'g'; if (write(socks[l], &go, 1) != 1) return 3;
        void* p = nullptr; CK(hipMemAddressReserve(&p, n * g, 0, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> hs(n);
        for (int i = 0; i < n; i++) {
            int fd = recv_fd(socks[l]);
            CK(hipMemImportFromShareableHandle(&hs[i], conv == 1 ? (void*)(uintptr_t)fd : (void*)&fd, hipMemHandleTypePosixFileDescriptor));
            CK(hipMemMap((char*)p + i * g, g, 0, hs[i], 0));
            if (order == 0) CK(hipMemRelease(hs[i]));
            close(fd);
        }
        CK(hipMemSetAccess(p, n * g, &acc, 1));
        CK(hipMemset(p, 7, n * g)); CK(hipDeviceSynchronize());
        const long long attached = free_mib(), held_attached = vram_used_mib();
        char done = 'd'; if (write(socks[l], &done, 1) != 1) return 3;
        int st = 0; waitpid(pids[l], &st, 0);
        const long long server_gone = free_mib(), held_gone = vram_used_mib();
        if (order == 2) for (int i = 0; i < n; i++) CK(hipMemRelease(hs[i]));
        for (int i = 0; i < n; i++) CK(hipMemUnmap((char*)p + i * g, g));
        const long long unmapped = free_mib(), held_unmapped = vram_used_mib();
        if (order == 1) for (int i = 0; i < n; i++) CK(hipMemRelease(hs[i]));
        const long long released = free_mib();
        CK(hipMemAddressFree(p, n * g));
        printf("life %d: attached %lld, server gone %lld, unmapped %lld, released %lld, range freed %lld (baseline - now = %lld MiB); driver (this process's fdinfo): VRAM held now %lld MiB (at start %lld)\n", l,
               attached, server_gone, unmapped, released, free_mib(), base - free_mib(), vram_used_mib(), vram0);
        printf("        held by this process: attached %lld, server gone %lld, unmapped %lld MiB\n", held_attached, held_gone, held_unmapped);
    }
    return 0;
}
