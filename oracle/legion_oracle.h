/*
 * legion_oracle.h -- CPU restatement of the RC4ML/Legion sampling + feature-cache hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and there only
 * as the checker / the reported CPU baseline.  The product path (legion_amd/, liblegion_hip.so)
 * never links, imports or falls back to it.
 *
 * PARITY STATUS.  The reference ships no tests, golden vectors or known-answer fixtures for this
 * path and its CUDA sources cannot be built in this image (nvcc, libcu++
 * and a GPU are required), so the restatement as a whole is "parity unpinned" against the
 * reference.  What IS pinned:
 *   - the topology cache's fill (lgo_fill_up's cached CSR) against the REFERENCE'S OWN KERNELS: the one file of the reference that
 *     compiles here, sampling_server/src/storage/graph_storage_impl.cuh (GetNeighborCount, TopoFillUp), is built by hipcc from where
 *     it lies into oracle/_ref/ref_graph_cache (oracle/Makefile target `ref`, oracle/ref_graph_cache_driver.hip) and run on the GPU by
 *     tests/test_gpu_ref_graph_cache.py;
 *   - the random draw (thrust::minstd_rand + discard + uniform_int_distribution) against
 *     rocThrust's own host implementation (oracle/thrust_pin.cpp -> tests/golden/rng_thrust.json);
 *   - the stable descending hotness order against rocThrust's host sort_by_key semantics
 *     (stable_sort_by_key with greater<>), same generator;
 *   - the launcher's meta_config line and clique -> cache_agg_mode arithmetic against the
 *     reference's own legion_server.py imported in the build container
 *     (tests/golden/gen_launcher_golden.py -> tests/golden/launcher.json).
 *
 * Every function cites the reference lines it restates; paths are relative to /root/reference,
 * SS = sampling_server/src.
 *
 * Canonical order: the reference compacts a hop's edges and new nodes through
 * shared-memory and global atomicAdd, so their order is a race.  The oracle (and the HIP path)
 * use slot-index-ascending order, which is one legal outcome of that race: edges are appended in
 * ascending slot index, and a node first touched in a hop is owned by the lowest slot that samples
 * it and is appended in ascending order of that slot.
 */
#ifndef LEGION_ORACLE_H
#define LEGION_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SS/include/system_config.cuh:47-57 */
#define LGO_INTERBATCH_CON 2
#define LGO_INTRABATCH_CON 3
#define LGO_MAX_DEVICE 8
#define LGO_TRAINMODE 0
#define LGO_VALIDMODE 1
#define LGO_TESTMODE 2
#define LGO_CACHEMISS_FLAG (-2)

/* ---- the draw: SS/engine/operator_impl.cu:235-238 (thrust::minstd_rand, discard(idx),
 *      uniform_int_distribution<>(0, deg-1)); Thrust algorithm restated from rocThrust 2.8.5
 *      /opt/rocm/include/thrust/random/detail/{uniform_int,uniform_real}_distribution.inl ---- */
uint32_t lgo_minstd_pow(uint64_t n);            /* 48271^n mod (2^31-1), n >= 0 */
int32_t  lgo_draw(int32_t idx, int32_t deg);    /* the neighbour index picked by slot idx */

/* ---- per-GPU working set: SS/engine/memorypool.cuh:20-221 restated as plain arrays ---- */
typedef struct lgo_pool {
    int32_t   total_num_nodes;
    int32_t   num_ids;              /* SS/engine/server.cu:187-199 */
    int32_t*  sampled_ids;          /* [num_ids]  */
    int32_t*  labels;               /* [batch]    */
    int32_t*  agg_src_ids;          /* [num_ids] global id of sampled neighbour */
    int32_t*  agg_dst_ids;          /* [num_ids] global id of the node sampled for */
    int32_t*  agg_src_off;          /* [num_ids] local positions (DGL COO src) */
    int32_t*  agg_dst_off;          /* [num_ids] local positions (DGL COO dst) */
    int32_t*  cache_search_buffer;  /* [num_ids] */
    int8_t*   tmp_part_ind;         /* [num_ids] */
    int32_t*  tmp_part_off;         /* [num_ids] */
    uint32_t* accessed_map;         /* [N/32+1] */
    int32_t*  position_map;         /* [N] */
    int32_t   node_counter[16];
    int32_t   edge_counter[16];
    float*    float_features;       /* [feature_rows * D] */
    int64_t   feature_rows;
    int32_t   feature_dim;
} lgo_pool;

lgo_pool* lgo_pool_create(int32_t total_num_nodes, int32_t num_ids, int32_t batch_size,
                          int64_t feature_rows, int32_t feature_dim);
void      lgo_pool_destroy(lgo_pool* p);

/* ---- topology as the sampler sees it: SS/storage/graph_storage.cu:12-73,76-111.
 *      P+1 slots; slot P = the full ("host") CSR, slot d = GPU d's cached CSR. ---- */
typedef struct lgo_graph {
    int32_t         partition_count;               /* P */
    const int64_t*  csr_node_index[LGO_MAX_DEVICE + 1];
    const int32_t*  csr_dst_node_ids[LGO_MAX_DEVICE + 1];
} lgo_graph;

/* ---- operators ---- */
/* SS/engine/operator_impl.cu:27-55 + :151-165 (memset bitmap & counters, kernel, counter_update(0)) */
void lgo_batch_generate(lgo_pool* p, const int32_t* all_ids, const int32_t* all_labels,
                        int32_t total_cap, int32_t batch_size, int32_t counter, int32_t hop_num);
/* SS/engine/operator_impl.cu:57-89 */
void lgo_counter_update(int32_t* node_counter, int32_t* edge_counter, int32_t op_id, int32_t size,
                        int32_t hop_num);
/* SS/engine/operator_impl.cu:175-281 (serve) / :301-397 (presample: always slot P, counts
 * edge_access_time[src]); canonical slot order.  part_ind/part_off are the FindTopo outputs. */
void lgo_random_sample(lgo_pool* p, const lgo_graph* g, int32_t op_id, int32_t count,
                       const int8_t* part_ind, const int32_t* part_off,
                       int is_presc, uint64_t* edge_access_time);
/* SS/engine/operator_impl.cu:283-296 */
void lgo_construct_graph(lgo_pool* p);
/* SS/engine/operator_impl.cu:542-548 */
void lgo_clear_pos_map(lgo_pool* p);
/* SS/cache/cache_impl.cuh:190-198 */
void lgo_hotness_measure(const lgo_pool* p, uint64_t* node_access_time);

/* ---- unified cache (one NVLink/xGMI clique): SS/cache/cache.cu ---- */
typedef struct lgo_cache {
    int32_t   total_num_nodes;
    int32_t   feature_dim;
    int32_t   Kg;                 /* GPUs in the clique */
    int32_t   Ki;                 /* clique index */
    int32_t   node_capacity;      /* rows per GPU */
    int32_t   edge_capacity;      /* vertices per GPU */
    int32_t*  QF;                 /* [N] feature hotness order */
    int32_t*  QT;                 /* [N] topology hotness order */
    uint64_t* AF;                 /* [N] sorted node hotness */
    uint64_t* AT;                 /* [N] sorted edge hotness */
    /* id -> value maps (the BGHT contract, SS/include/hashmap/bcht.hpp:105-165), direct mapped */
    int32_t*  node_map;           /* [N] global slot g or -2 */
    int8_t*   edge_index_map;     /* [N] owner device or -2 */
    int32_t*  edge_offset_map;    /* [N] row in owner's CSR or -2 */
    float*    feat_cache[LGO_MAX_DEVICE];      /* [node_capacity*D] per in-clique GPU */
    int64_t*  topo_indptr[LGO_MAX_DEVICE];     /* [edge_capacity+1] */
    int32_t*  topo_col[LGO_MAX_DEVICE];
    /* hybrid CPU-cache / GPU-cache tier of ONE GPU (lgo_hybrid_init; SS/cache/cache.cu:614-670): the GPU cache is
     * feat_cache[0] with gpu_cache_capacity rows */
    int32_t   hybrid;                          /* 0: the clique cache above, 1: the hybrid tier */
    int32_t   cpu_cache_capacity, gpu_cache_capacity;
    float*    cpu_cache;                       /* [cpu_cache_capacity*D] (the reference's mapped pinned cpu_float_features_) */
} lgo_cache;

lgo_cache* lgo_cache_create(int32_t total_num_nodes, int32_t feature_dim, int32_t Kg, int32_t Ki);
void       lgo_cache_destroy(lgo_cache* c);

/* SS/cache/cache.cu:360-443: sum the Kg per-GPU counters, iota, stable descending sort. */
void lgo_candidate_selection(lgo_cache* c, const uint64_t* const* node_access, const uint64_t* const* edge_access);
/* SS/cache/cache.cu:445-551.  counters = the two PCIe transaction counts (zeros in v2);
 * max_id_num[j] = PreSCCacheController::MaxIdNum of in-clique GPU j.  Returns alpha index. */
int32_t lgo_cost_model(lgo_cache* c, int64_t cache_memory, const int64_t* csr_index,
                       const uint64_t counters[2], const int32_t* max_id_num, int32_t train_step,
                       float* out_trans_total /* may be NULL */);
/* SS/cache/cache.cu:553-611 + :71-136 + cache_impl.cuh:89-109,183-188 + graph_storage.cu:76-111 */
void lgo_fill_up(lgo_cache* c, const float* host_features, const int64_t* csr_index,
                 const int32_t* csr_dst);
/* The hybrid CPU-cache / GPU-cache tier (Legion-SSD / Helios, unreleased: SS/engine/server.cu:112 is commented out).
 * SS/cache/cache.cu:614-670 (HybridInit: THIS GPU's own hotness order, no clique sum), :138-153 (HybridInsert) with
 * SS/cache/cache_impl.cuh:113-123 (HybridInitPair): rank t < gpu_cap -> value cpu_cap + t, gpu_cap <= t < gpu_cap + cpu_cap ->
 * value t - gpu_cap; the topology maps stay empty (:642 "edge cache disabled now").  The reference allocates both caches and
 * fills neither (:616 allocates, nothing writes; :656 FeatFillUp commented out); here, as FillUp does for its stripes, GPU-cache
 * row r holds the features of QF[r] and CPU-cache row r those of QF[gpu_cap + r].  Ranks at or beyond N are skipped. */
void lgo_hybrid_init(lgo_cache* c, const uint64_t* node_access, const float* host_features,
                     int32_t cpu_cache_capacity, int32_t gpu_cache_capacity);
/* SS/cache/cache.cu:217-225 */
void lgo_find_topo(const lgo_cache* c, const int32_t* input_ids, int8_t* part_ind,
                   int32_t* part_off, int32_t batch_size);
/* SS/cache/cache.cu:180-215 (node_counter[(op%3)*2], [(op%3)*2+1]) */
void lgo_find_feat(const lgo_cache* c, lgo_pool* p, int32_t op_id);
/* SS/cache/cache_impl.cuh:239-272 driven by SS/engine/operator_impl.cu:502-519; with c->hybrid the single-GPU kernel
 * feat_cache_lookup (SS/cache/cache_impl.cuh:202-235): a miss row is NOT written by that kernel (the unreleased SSD reader's job,
 * SS/engine/operator_impl.cu:522-539); host_features != NULL stands in for that reader: the row comes from the full table */
void lgo_feature_cache_lookup(const lgo_cache* c, lgo_pool* p, const float* host_features,
                              int32_t op_id);

/* ---- whole batch, exactly the op order of GPURunner::RunOnce / RunPreSc
 *      (SS/engine/server.cu:285-332): returns total valid edges ---- */
int64_t lgo_run_batch(lgo_pool* p, const lgo_graph* g, const lgo_cache* c /* NULL in presc */,
                      const float* host_features,
                      const int32_t* all_ids, const int32_t* all_labels, int32_t total_cap,
                      int32_t batch_size, int32_t counter, const int32_t* fanout, int32_t hop_num,
                      int32_t mode, int is_presc, uint64_t* node_access_time,
                      uint64_t* edge_access_time);

/* ---- step arithmetic: SS/engine/ipc_service.cu:60-132,213-253 ---- */
typedef struct lgo_steps {
    int32_t train_step, valid_step, test_step, epoch, raw_batch_size;
    int32_t train_bs[LGO_MAX_DEVICE], valid_bs[LGO_MAX_DEVICE], test_bs[LGO_MAX_DEVICE];
} lgo_steps;
void    lgo_coordinate(lgo_steps* s, int32_t partition_count, const int32_t* train_num,
                       const int32_t* valid_num, const int32_t* test_num, int32_t raw_batch_size,
                       int32_t epoch);
int32_t lgo_max_step(const lgo_steps* s);
int32_t lgo_current_mode(const lgo_steps* s, int32_t global_batch_id);
int32_t lgo_local_batch_id(const lgo_steps* s, int32_t global_batch_id);
int32_t lgo_current_batchsize(const lgo_steps* s, int32_t dev_id, int32_t mode);

/* ---- bounded multi-thread CPU baseline for bench.py (pthread over independent batches) ---- */
int64_t lgo_bench_batches(const lgo_graph* g, int32_t total_num_nodes, const int32_t* all_ids,
                          int32_t total_cap, int32_t batch_size, const int32_t* fanout,
                          int32_t hop_num, int32_t first_batch, int32_t num_batches,
                          int32_t threads, const float* features, int32_t feature_dim,
                          double* seconds_out, int64_t* nodes_out);

#ifdef __cplusplus
}
#endif
#endif
