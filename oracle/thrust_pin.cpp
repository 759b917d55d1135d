// thrust_pin.cpp -- generates tests/golden/rng_thrust.json from rocThrust's OWN host code.
//
// TEST INFRASTRUCTURE ONLY.  The reference's random draw and hotness ordering live in a
// third-party dependency that is not vendored under /root/reference: Thrust (no version pinned;
// README.md:24 says CUDA 11.7 => Thrust 1.15.x).  This image ships rocThrust 2.8.5
// (/opt/rocm/include/thrust), whose random/ and sort code is the same algorithm.  This program
// runs the exact expressions of the reference call sites on the host:
//   SS/engine/operator_impl.cu:235-238   minstd_rand engine; engine.discard(idx);
//                                        uniform_int_distribution<> dist(0, col_size-1); dist(engine)
//   SS/cache/cache.cu:415                sort_by_key(keys, keys+N, order, greater<unsigned long long>())
// and prints the known answers the oracle (oracle/legion_oracle.c) and the HIP path are pinned to.
//
// Build + run: oracle/Makefile target `golden` (hipcc: this rocThrust's host paths include rocPRIM headers that need it);
// target `san` builds the same program with ASan + UBSan on the host half (tests/test_sanitizers_cpu.py compares its output
// with the committed golden file).
#include <thrust/random/linear_congruential_engine.h>
#include <thrust/random/uniform_int_distribution.h>
#include <thrust/sort.h>
#include <thrust/functional.h>
#include <thrust/execution_policy.h>

#include <cstdint>
#include <cstdio>
#include <vector>

static int32_t thrust_draw(int32_t idx, int32_t col_size)
{
    thrust::minstd_rand engine;
    engine.discard(idx);
    thrust::uniform_int_distribution<> dist(0, col_size - 1);
    return dist(engine);
}

int main()
{
    // (idx, deg) grid: every slot-index regime the configs reach (< 6M) and beyond,
    // degrees from 1 to RMAT hub sizes, plus boundary-heavy combinations.
    std::vector<int32_t> idxs = {0, 1, 2, 3, 24, 25, 63, 64, 65, 1023, 1024, 4095, 4096, 16384,
                                 25599, 25600, 255999, 256000, 1999999, 2000000, 5999999, 6000000,
                                 8388607, 8388608, 16777215, 16777216, 100000000, 2147483646};
    std::vector<int32_t> degs = {1, 2, 3, 5, 7, 10, 15, 25, 26, 100, 1000, 4097, 65536, 123457,
                                 1000003, 16777216, 2147483647};
    std::printf("{\n \"source\": \"rocThrust host code, see oracle/thrust_pin.cpp\",\n");
    std::printf(" \"minstd_min\": %u, \"minstd_max\": %u,\n", (unsigned)thrust::minstd_rand::min,
                (unsigned)thrust::minstd_rand::max);
    std::printf(" \"draws\": [\n");
    bool first = true;
    for (int32_t idx : idxs)
        for (int32_t deg : degs) {
            std::printf("%s  [%d, %d, %d]", first ? "" : ",\n", idx, deg, thrust_draw(idx, deg));
            first = false;
        }
    // a dense block: idx 0..2047 at deg 25 and deg 10 (the two fan-outs' typical regime)
    for (int32_t idx = 0; idx < 2048; idx++)
        for (int32_t deg : {10, 25, 37}) {
            std::printf(",\n  [%d, %d, %d]", idx, deg, thrust_draw(idx, deg));
        }
    std::printf("\n ],\n");

    // raw engine outputs after discard(n): pins lgo_minstd_pow / the device power tables
    std::printf(" \"engine\": [\n");
    first = true;
    for (unsigned long long n : {0ull, 1ull, 2ull, 10ull, 2047ull, 2048ull, 4194303ull, 4194304ull,
                                 4294967295ull, 4294967296ull}) {
        thrust::minstd_rand e;
        e.discard(n);
        std::printf("%s  [%llu, %u]", first ? "" : ",\n", n, (unsigned)e());
        first = false;
    }
    std::printf("\n ],\n");

    // stable descending sort_by_key with many ties (most vertices tie at hotness 0)
    std::vector<unsigned long long> keys = {0, 5, 0, 7, 5, 0, 7, 1, 0, 5, 18446744073709551615ull, 0, 1, 7, 0, 3};
    std::vector<int32_t> order(keys.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (int32_t)i;
    std::vector<unsigned long long> in = keys;
    thrust::sort_by_key(thrust::host, keys.begin(), keys.end(), order.begin(),
                        thrust::greater<unsigned long long>());
    std::printf(" \"sort_in\": [");
    for (size_t i = 0; i < in.size(); i++) std::printf("%s%llu", i ? ", " : "", in[i]);
    std::printf("],\n \"sort_keys\": [");
    for (size_t i = 0; i < keys.size(); i++) std::printf("%s%llu", i ? ", " : "", keys[i]);
    std::printf("],\n \"sort_order\": [");
    for (size_t i = 0; i < order.size(); i++) std::printf("%s%d", i ? ", " : "", order[i]);
    std::printf("]\n}\n");
    return 0;
}
