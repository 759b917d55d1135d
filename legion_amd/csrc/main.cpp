// main.cpp -- the `sampling_server` binary.  Reference: sampling_server/src/main.cu:5-16
// (argv = <gpu_number> <cache_agg_mode>, fan-out hard-coded {25,10}).  This build accepts the
// fan-out as optional extra arguments (the reference's pybind Run(fanout, ...) signature,
// sampling_server/sampling_server.cpp:7): sampling_server <gpu_number> <cache_agg_mode> [f1 f2 ...] [--disk]
// --disk = Run()'s in_memory_mode 0: meta_config carries fifteen fields and the caches are the hybrid CPU-cache / GPU-cache tier
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/legion_hip.h"

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::printf("usage: %s <gpu_number> <cache_agg_mode> [fanout ...] [--disk]\n", argv[0]);
        return 2;
    }
    std::vector<int32_t> fanout;
    int32_t in_memory_mode = 1;
    for (int i = 3; i < argc; i++) {
        if (std::strcmp(argv[i], "--disk") == 0) in_memory_mode = 0;
        else fanout.push_back(std::atoi(argv[i]));
    }
    if (fanout.empty()) { fanout.push_back(25); fanout.push_back(10); }
    return legion_run(fanout.data(), (int32_t)fanout.size(), std::atoi(argv[1]), in_memory_mode, (int)std::atof(argv[2]));
}
