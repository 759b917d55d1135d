// Can a chunk made with hipMemCreate be handed to another process as a file descriptor?  (the server's lane arena in shuffled chunks needs it)
//   hipcc --offload-arch=gfx950 -o /tmp/vmm_ipc_probe tools/micro/vmm_ipc_probe.hip && /tmp/vmm_ipc_probe
#include <hip/hip_runtime.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("[%d] %s -> %s\n", getpid(), #x, hipGetErrorString(e)); _exit(2); } } while (0)
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
    if (recvmsg(sock, &msg, 0) < 0) { perror("recvmsg"); _exit(3); }
    int fd = -1; memcpy(&fd, CMSG_DATA(CMSG_FIRSTHDR(&msg)), sizeof(int));
    return fd;
}
int main()
{
    const size_t g = 64ull << 20; const int n = 4;
    int sv[2]; socketpair(AF_UNIX, SOCK_STREAM, 0, sv);
    pid_t pid = fork();                                   // (before either process touches the GPU)
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    if (pid == 0) {                                        // the "trainer": receives n fds, maps them in order, checks the pattern
        close(sv[0]);
        void* p = nullptr; CK(hipMemAddressReserve(&p, n * g, 0, nullptr, 0));
        for (int i = 0; i < n; i++) {
            int fd = recv_fd(sv[1]);
            hipMemGenericAllocationHandle_t h;
            CK(hipMemImportFromShareableHandle(&h, (void*)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor));
            CK(hipMemMap((char*)p + i * g, g, 0, h, 0));
            close(fd);
        }
        CK(hipMemSetAccess(p, n * g, &acc, 1));
        std::vector<unsigned char> host(n);
        for (int i = 0; i < n; i++) CK(hipMemcpy(&host[i], (char*)p + i * g + 12345, 1, hipMemcpyDeviceToHost));
        int ok = 1; for (int i = 0; i < n; i++) ok &= host[i] == (unsigned char)(17 + i);
        printf("child: pattern %s (%d %d %d %d)\n", ok ? "OK" : "WRONG", host[0], host[1], host[2], host[3]);
        char done = ok ? 'k' : 'w'; write(sv[1], &done, 1);
        _exit(ok ? 0 : 1);
    }
    close(sv[1]);
    void* p = nullptr; CK(hipMemAddressReserve(&p, n * g, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    for (int i = 0; i < n; i++) { CK(hipMemCreate(&h[i], g, &prop, 0)); CK(hipMemMap((char*)p + i * g, g, 0, h[i], 0)); }
    CK(hipMemSetAccess(p, n * g, &acc, 1));
    for (int i = 0; i < n; i++) CK(hipMemset((char*)p + i * g, 17 + i, g));
    CK(hipDeviceSynchronize());
    for (int i = 0; i < n; i++) {
        int fd = -1;
        CK(hipMemExportToShareableHandle(&fd, h[i], hipMemHandleTypePosixFileDescriptor, 0));
        send_fd(sv[0], fd);
        close(fd);
    }
    char done = 0; read(sv[0], &done, 1);
    int st = 0; waitpid(pid, &st, 0);
    printf("parent: child said '%c', exit status %d\n", done, WEXITSTATUS(st));
    return done == 'k' ? 0 : 1;
}
