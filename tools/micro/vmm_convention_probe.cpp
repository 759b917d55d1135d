// Prints how the HIP runtime this program is linked against takes a file-descriptor handle in hipMemImportFromShareableHandle:
//   hipcc -O2 tools/micro/vmm_convention_probe.cpp -o /tmp/vmm_convention_probe && /tmp/vmm_convention_probe
// -> "convention <0 pointer | 1 value | -1 none> runtime <hipRuntimeGetVersion>".  The trainer end (legion_amd/trainer/ipc_service.cpp)
// runs the same probe on the runtime bundled with torch: ipc_service.vmm_fd_convention().
#include <cstdio>

#include "../../legion_amd/trainer/vmm_probe.h"

int main()
{
    int v = 0;
    (void)hipRuntimeGetVersion(&v);
    if (hipSetDevice(0) != hipSuccess) { printf("no device\n"); return 2; }
    printf("convention %d runtime %d\n", vmm_fd_convention(), v);
    return 0;
}
