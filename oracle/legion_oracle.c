/*
 * legion_oracle.c -- CPU restatement of the RC4ML/Legion sampling + feature-cache hot path.
 * TEST INFRASTRUCTURE ONLY; see legion_oracle.h for the scope, the parity status ("parity
 * unpinned" against the reference, which has no tests; RNG / sort / launcher pinned as listed
 * there) and the canonical slot order.  Plain C11, no dependencies besides libc + pthread.
 */
#define _GNU_SOURCE
#include "legion_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------
 * The draw.  thrust::minstd_rand = linear_congruential_engine<uint32, 48271, 0, 2147483647>,
 * default seed 1 (rocThrust linear_congruential_engine.h:147-158).  engine.discard(idx) leaves
 * the state at 48271^idx, the distribution's single urng() call returns 48271^(idx+1) mod m.
 * uniform_int_distribution<int>(0, deg-1) maps it through uniform_real_distribution<double>:
 *   r = double(x - min) / (1.0 + double(max - min));  result = int(r * ((deg-1)+1.0 - 0.0) + 0.0)
 * (uniform_int_distribution.inl:66-86, uniform_real_distribution.inl:66-84); min=1, max=m-1.
 * Call site: SS/engine/operator_impl.cu:235-238.
 * ------------------------------------------------------------------------------------------ */
#define LGO_M 2147483647ull
#define LGO_A 48271ull

uint32_t lgo_minstd_pow(uint64_t n)
{
    uint64_t result = 1, base = LGO_A;
    while (n) {
        if (n & 1) result = (result * base) % LGO_M;
        base = (base * base) % LGO_M;
        n >>= 1;
    }
    return (uint32_t)result;
}

int32_t lgo_draw(int32_t idx, int32_t deg)
{
    uint32_t x = lgo_minstd_pow((uint64_t)(uint32_t)idx + 1ull);
    double r = (double)(uint32_t)(x - 1u);
    r /= (1.0 + (double)(uint32_t)(2147483646u - 1u));
    double scaled = r * (((double)(deg - 1) + 1.0) - 0.0);
    return (int32_t)(scaled + 0.0);
}

/* ------------------------------------------------------------------------------------------ */
static void* xcalloc(size_t n, size_t sz)
{
    void* p = calloc(n ? n : 1, sz);
    if (!p) { fprintf(stderr, "legion_oracle: out of memory (%zu x %zu)\n", n, sz); abort(); }
    return p;
}

lgo_pool* lgo_pool_create(int32_t total_num_nodes, int32_t num_ids, int32_t batch_size,
                          int64_t feature_rows, int32_t feature_dim)
{
    lgo_pool* p = (lgo_pool*)xcalloc(1, sizeof(*p));
    p->total_num_nodes = total_num_nodes;
    p->num_ids = num_ids;
    p->sampled_ids = (int32_t*)xcalloc(num_ids, 4);
    p->labels = (int32_t*)xcalloc(batch_size, 4);
    p->agg_src_ids = (int32_t*)xcalloc(num_ids, 4);
    p->agg_dst_ids = (int32_t*)xcalloc(num_ids, 4);
    p->agg_src_off = (int32_t*)xcalloc(num_ids, 4);
    p->agg_dst_off = (int32_t*)xcalloc(num_ids, 4);
    p->cache_search_buffer = (int32_t*)xcalloc(num_ids, 4);
    p->tmp_part_ind = (int8_t*)xcalloc(num_ids, 1);
    p->tmp_part_off = (int32_t*)xcalloc(num_ids, 4);
    p->accessed_map = (uint32_t*)xcalloc((size_t)(total_num_nodes / 32) + 1, 4);
    p->position_map = (int32_t*)xcalloc(total_num_nodes, 4);
    p->feature_rows = feature_rows;
    p->feature_dim = feature_dim;
    p->float_features = (float*)xcalloc((size_t)feature_rows * (size_t)(feature_dim > 0 ? feature_dim : 1), 4);
    return p;
}

void lgo_pool_destroy(lgo_pool* p)
{
    if (!p) return;
    free(p->sampled_ids); free(p->labels); free(p->agg_src_ids); free(p->agg_dst_ids);
    free(p->agg_src_off); free(p->agg_dst_off); free(p->cache_search_buffer);
    free(p->tmp_part_ind); free(p->tmp_part_off); free(p->accessed_map); free(p->position_map);
    free(p->float_features); free(p);
}

/* SS/engine/operator_impl.cu:57-89, INTRABATCH_CON = 3 */
void lgo_counter_update(int32_t* nc, int32_t* ec, int32_t op_id, int32_t size, int32_t hop_num)
{
    const int C = LGO_INTRABATCH_CON;
    if (op_id == 0) {
        nc[0] = 0;
        nc[1] = size;
        nc[C * 3 + (op_id / C)] = nc[0] + nc[1];
        nc[C * 3 - 1] = hop_num;
    } else if (op_id > 0 && op_id % C == 0) {
        nc[0] = nc[0] + nc[1];
        nc[1] = nc[C * 2];
        nc[C * 2] = 0;
        nc[C * 2 + 1] = nc[0] + nc[1];
        ec[0] = ec[0] + ec[1];
        ec[1] = ec[2];
        ec[2] = 0;
        nc[C * 3 + (op_id / C)] = nc[0] + nc[1];
        ec[C * 3 + (op_id / C)] = ec[0] + ec[1];
    } else if (op_id % C > 0) {
        nc[(op_id % C) * 2] = nc[0];
        nc[(op_id % C) * 2 + 1] = nc[1];
    }
}

/* SS/engine/operator_impl.cu:92-172.  The kernel (:27-55) is launched with `size` in its
 * batch_size parameter (:162), so a clamped last batch indexes all_ids at size*counter+idx --
 * restated as written. */
void lgo_batch_generate(lgo_pool* p, const int32_t* all_ids, const int32_t* all_labels,
                        int32_t total_cap, int32_t batch_size, int32_t counter, int32_t hop_num)
{
    memset(p->accessed_map, 0, ((size_t)(p->total_num_nodes / 32) + 1) * 4);   /* :151 */
    memset(p->node_counter, 0, sizeof(p->node_counter));                        /* :155 */
    memset(p->edge_counter, 0, sizeof(p->edge_counter));                        /* :156 */
    int32_t size = ((int64_t)batch_size * (counter + 1) >= total_cap)
                       ? (total_cap - batch_size * counter) : batch_size;      /* :159 */
    for (int32_t idx = 0; idx < size; idx++) {                                  /* :38-54 */
        int64_t at = (int64_t)size * counter + idx;
        if (at >= total_cap) {
            p->sampled_ids[idx] = -1;
            p->labels[idx] = -1;
        } else {
            int32_t src_id = all_ids[at % total_cap];
            p->sampled_ids[idx] = src_id;
            p->accessed_map[src_id / 32] |= (1u << (src_id % 32));
            p->position_map[src_id] = idx;
            p->labels[idx] = all_labels[at % total_cap];
        }
    }
    lgo_counter_update(p->node_counter, p->edge_counter, 0, size, hop_num);     /* :165 */
}

/* SS/engine/operator_impl.cu:175-281 and :301-397, canonical slot order. */
void lgo_random_sample(lgo_pool* p, const lgo_graph* g, int32_t op_id, int32_t count,
                       const int8_t* part_ind, const int32_t* part_off,
                       int is_presc, uint64_t* edge_access_time)
{
    int32_t* nc = p->node_counter;
    int32_t* ec = p->edge_counter;
    const int32_t* input_ids = NULL;
    int32_t batch_size = 0;
    if (op_id == LGO_INTRABATCH_CON) {          /* :201-203 */
        input_ids = p->sampled_ids;
        batch_size = nc[1];
    } else if (op_id > LGO_INTRABATCH_CON) {    /* :204-207 */
        input_ids = p->agg_src_ids + ec[0];
        batch_size = ec[1];
    }
    const int32_t P = g->partition_count;
    const int32_t node_base = nc[0] + nc[1];     /* :268 */
    const int32_t edge_base = ec[0] + ec[1];     /* :275 */
    int32_t n_new = 0, n_edge = 0;
    const int32_t total = batch_size * count;    /* int32 as in :208 */
    for (int32_t idx = 0; idx < total; idx++) {
        int32_t src = input_ids[idx / count];
        if (src < 0) continue;                   /* :218 */
        int32_t k = idx % count;
        int32_t part_id = is_presc ? -1 : (int32_t)part_ind[idx / count];
        int64_t start; int32_t col_size; const int32_t* col;
        if (part_id < 0) {                       /* :224-226 / :347-348 */
            start = g->csr_node_index[P][src];
            col_size = (int32_t)(g->csr_node_index[P][src + 1] - start);
            col = g->csr_dst_node_ids[P];
        } else {                                 /* :227-230 */
            int32_t off = part_off[idx / count];
            start = g->csr_node_index[part_id][off];
            col_size = (int32_t)(g->csr_node_index[part_id][off + 1] - start);
            col = g->csr_dst_node_ids[part_id];
        }
        if (k >= col_size) continue;             /* :232-233 */
        int32_t pick = lgo_draw(idx, col_size);  /* :235-238 */
        int32_t dst = col[start + (int64_t)pick];/* :239-243 */
        if (dst < 0) continue;                   /* :244 */
        if (is_presc && edge_access_time) edge_access_time[src] += 1;   /* :358 */
        uint32_t bit = 1u << (dst % 32);
        uint32_t old = p->accessed_map[dst / 32];
        p->accessed_map[dst / 32] = old | bit;   /* :248 */
        if (!(old & bit)) {                      /* :251-254, :267-272 */
            p->sampled_ids[node_base + n_new] = dst;
            p->position_map[dst] = node_base + n_new;
            n_new++;
        }
        p->agg_src_ids[edge_base + n_edge] = dst;   /* :255-257, :274-278 */
        p->agg_dst_ids[edge_base + n_edge] = src;
        n_edge++;
    }
    nc[LGO_INTRABATCH_CON * 2] += n_new;         /* :263 */
    ec[2] += n_edge;                             /* :264 */
}

/* SS/engine/operator_impl.cu:283-296 */
void lgo_construct_graph(lgo_pool* p)
{
    int32_t edge_num = p->edge_counter[2];
    int32_t edge_off = p->edge_counter[0] + p->edge_counter[1];
    for (int32_t i = 0; i < edge_num; i++) {
        p->agg_src_off[edge_off + i] = p->position_map[p->agg_src_ids[edge_off + i]];
        p->agg_dst_off[edge_off + i] = p->position_map[p->agg_dst_ids[edge_off + i]];
    }
}

/* SS/engine/operator_impl.cu:542-548 */
void lgo_clear_pos_map(lgo_pool* p)
{
    int32_t n = p->node_counter[LGO_INTRABATCH_CON * 2 + 1];
    for (int32_t i = 0; i < n; i++) p->position_map[p->sampled_ids[i]] = 0;
}

/* SS/cache/cache_impl.cuh:190-198 */
void lgo_hotness_measure(const lgo_pool* p, uint64_t* node_access_time)
{
    int32_t n = p->node_counter[LGO_INTRABATCH_CON * 2 + 1];
    for (int32_t i = 0; i < n; i++) {
        int32_t cid = p->sampled_ids[i];
        if (cid >= 0) node_access_time[cid] += 1;
    }
}

/* ------------------------------------------------------------------------------------------ */
lgo_cache* lgo_cache_create(int32_t total_num_nodes, int32_t feature_dim, int32_t Kg, int32_t Ki)
{
    lgo_cache* c = (lgo_cache*)xcalloc(1, sizeof(*c));
    c->total_num_nodes = total_num_nodes;
    c->feature_dim = feature_dim;
    c->Kg = Kg;
    c->Ki = Ki;
    c->QF = (int32_t*)xcalloc(total_num_nodes, 4);
    c->QT = (int32_t*)xcalloc(total_num_nodes, 4);
    c->AF = (uint64_t*)xcalloc(total_num_nodes, 8);
    c->AT = (uint64_t*)xcalloc(total_num_nodes, 8);
    c->node_map = (int32_t*)xcalloc(total_num_nodes, 4);
    c->edge_index_map = (int8_t*)xcalloc(total_num_nodes, 1);
    c->edge_offset_map = (int32_t*)xcalloc(total_num_nodes, 4);
    return c;
}

void lgo_cache_destroy(lgo_cache* c)
{
    if (!c) return;
    free(c->QF); free(c->QT); free(c->AF); free(c->AT);
    free(c->node_map); free(c->edge_index_map); free(c->edge_offset_map);
    for (int i = 0; i < LGO_MAX_DEVICE; i++) {
        free(c->feat_cache[i]); free(c->topo_indptr[i]); free(c->topo_col[i]);
    }
    free(c->cpu_cache);
    free(c);
}

typedef struct { uint64_t key; int32_t id; } lgo_kv;

static int lgo_kv_cmp(const void* a, const void* b)
{
    const lgo_kv* x = (const lgo_kv*)a; const lgo_kv* y = (const lgo_kv*)b;
    if (x->key != y->key) return x->key > y->key ? -1 : 1;   /* thrust::greater: descending */
    return (x->id > y->id) - (x->id < y->id);                 /* stable over iota => ascending id */
}

static void lgo_sorted_order(const uint64_t* const* access, int32_t Kg, int32_t N, int32_t* order,
                             uint64_t* agg_sorted)
{
    lgo_kv* kv = (lgo_kv*)xcalloc(N, sizeof(lgo_kv));
    for (int32_t i = 0; i < N; i++) {
        uint64_t s = 0;
        for (int32_t j = 0; j < Kg; j++) s += access[j][i];   /* aggregate_access cache_impl.cuh:72-76 */
        kv[i].key = s;
        kv[i].id = i;                                         /* init_cache_order :79-83 */
    }
    qsort(kv, N, sizeof(lgo_kv), lgo_kv_cmp);                 /* sort_by_key cache.cu:415,435 */
    for (int32_t i = 0; i < N; i++) { order[i] = kv[i].id; agg_sorted[i] = kv[i].key; }
    free(kv);
}

/* SS/cache/cache.cu:360-443 (one clique) */
void lgo_candidate_selection(lgo_cache* c, const uint64_t* const* node_access,
                             const uint64_t* const* edge_access)
{
    lgo_sorted_order(node_access, c->Kg, c->total_num_nodes, c->QF, c->AF);
    lgo_sorted_order(edge_access, c->Kg, c->total_num_nodes, c->QT, c->AT);
}

static uint64_t prefix_at(const uint64_t* prefix, int64_t i) { return i < 0 ? 0 : prefix[i]; }

/* SS/cache/cache.cu:445-551.  Arithmetic types follow the reference statement by statement:
 * std::vector<float> tables, `x * 1.0 / y * z` in double then narrowed to float, integer
 * capacities stored through float.  The reference reads prefix[-1] when a candidate count is 0
 * (:529,:533, undefined); the oracle and the product read 0 there. */
int32_t lgo_cost_model(lgo_cache* c, int64_t cache_memory, const int64_t* csr_index,
                       const uint64_t counters[2], const int32_t* max_id_num, int32_t train_step,
                       float* out_trans_total)
{
    const int32_t N = c->total_num_nodes;
    const int32_t D = c->feature_dim;
    const int32_t Kg = c->Kg;
    const int max_payload_size = 64;                                            /* CLS */
    int64_t memory_step = (int64_t)((double)(cache_memory * Kg) * 0.01);        /* :458 */
    uint64_t total_trans_of_topo = counters[0] + counters[1];                   /* :459 */
    uint64_t total_trans_of_feat = 0;
    for (int j = 0; j < Kg; j++)                                                /* :461-463 */
        total_trans_of_feat += (uint64_t)((int64_t)(((int64_t)max_id_num[j] * train_step) * D) * sizeof(float))
                               / (uint64_t)max_payload_size;

    uint64_t* node_prefix = (uint64_t*)xcalloc(N, 8);
    uint64_t* edge_prefix = (uint64_t*)xcalloc(N, 8);
    uint64_t* edge_mem_prefix = (uint64_t*)xcalloc(N, 8);
    uint64_t a = 0, b = 0, m = 0;
    for (int32_t i = 0; i < N; i++) {                                           /* :471-472, :496-500 */
        a += c->AF[i]; node_prefix[i] = a;
        b += c->AT[i]; edge_prefix[i] = b;
        int32_t id = c->QT[i];
        int64_t neighbor_count = csr_index[id + 1] - csr_index[id];
        m += (uint64_t)(sizeof(int64_t) + sizeof(int32_t) * neighbor_count);    /* GetEdgeMem */
        edge_mem_prefix[i] = m;
    }

    int64_t total_mem = cache_memory * Kg;                                      /* :481 */
    int64_t steps = (total_mem - 1) / memory_step + 1;                          /* :482 */
    float* trans_of_topo = (float*)xcalloc(steps + 1, 4);
    float* trans_of_feat = (float*)xcalloc(steps + 1, 4);
    float* cap_of_topo = (float*)xcalloc(steps + 1, 4);
    float* cap_of_feat = (float*)xcalloc(steps + 1, 4);
    float* trans_of_total = (float*)xcalloc(steps + 1, 4);
    int64_t current_steps = 0;
    int32_t node_num_topo = 0, node_num_feat = 0;
    for (int64_t current_mem = 0; current_mem < total_mem; current_mem += memory_step) {   /* :507 */
        if ((uint64_t)current_mem > (uint64_t)N * D * sizeof(float))            /* :508 */
            node_num_feat = N;
        else
            node_num_feat = (int32_t)((uint64_t)(current_steps + 1) *
                                      ((uint64_t)memory_step / (D * sizeof(float))));     /* :511 */
        if ((uint64_t)current_mem > edge_mem_prefix[N - 1]) {                   /* :513 */
            node_num_topo = N;
        } else {                                                                /* :516 lower_bound */
            int64_t lo = 0, hi = N;
            while (lo < hi) {
                int64_t mid = lo + (hi - lo) / 2;
                if (edge_mem_prefix[mid] < (uint64_t)current_mem) lo = mid + 1; else hi = mid;
            }
            node_num_topo = (int32_t)lo;
        }
        if (node_num_topo < N) {                                                /* :528-531 */
            trans_of_topo[current_steps] = (float)((double)total_trans_of_topo * 1.0 /
                (double)edge_prefix[N - 1] * (double)prefix_at(edge_prefix, (int64_t)node_num_topo - 1));
            cap_of_topo[current_steps] = (float)(node_num_topo / Kg);
        }
        if (node_num_feat < N) {                                                /* :532-535 */
            trans_of_feat[current_steps] = (float)((double)total_trans_of_feat * 1.0 /
                (double)node_prefix[N - 1] * (double)prefix_at(node_prefix, (int64_t)node_num_feat - 1));
            cap_of_feat[current_steps] = (float)(node_num_feat / Kg);
        }
        current_steps++;
    }
    for (int64_t s = 1; s < steps; s++)                                         /* :539-544 */
        trans_of_total[s] = trans_of_topo[s] + trans_of_feat[steps - 1 - s];
    int64_t max_sidx = 0;                                                       /* :545 max_element: first max */
    for (int64_t s = 1; s < steps + 1; s++)
        if (trans_of_total[s] > trans_of_total[max_sidx]) max_sidx = s;
    c->node_capacity = (int32_t)(cap_of_feat[steps - 1 - max_sidx] + 1);        /* :547 */
    c->edge_capacity = (int32_t)(cap_of_topo[max_sidx] + 1);                    /* :548 */
    if (out_trans_total) *out_trans_total = trans_of_total[max_sidx];
    free(node_prefix); free(edge_prefix); free(edge_mem_prefix);
    free(trans_of_topo); free(trans_of_feat); free(cap_of_topo); free(cap_of_feat);
    free(trans_of_total);
    return (int32_t)max_sidx;
}

/* SS/cache/cache.cu:553-611.  The reference indexes QF/QT up to capacity*Kg-1, which can pass
 * N-1 by up to Kg-1 entries (undefined); entries at or beyond N are skipped here. */
void lgo_fill_up(lgo_cache* c, const float* host_features, const int64_t* csr_index,
                 const int32_t* csr_dst)
{
    const int32_t N = c->total_num_nodes, D = c->feature_dim, Kg = c->Kg;
    const int64_t ncap = c->node_capacity, ecap = c->edge_capacity;
    for (int32_t i = 0; i < N; i++) {                      /* sentinel: cache.cu:77-86 */
        c->node_map[i] = LGO_CACHEMISS_FLAG;
        c->edge_index_map[i] = (int8_t)LGO_CACHEMISS_FLAG;
        c->edge_offset_map[i] = LGO_CACHEMISS_FLAG;
    }
    for (int64_t t = 0; t < ncap * Kg && t < N; t++)       /* InitPair cache_impl.cuh:104-109 */
        c->node_map[c->QF[t]] = (int32_t)((t % Kg) * ncap + t / Kg);
    for (int64_t t = 0; t < ecap * Kg && t < N; t++) {     /* InitIndexPair/InitOffsetPair :89-101 */
        c->edge_index_map[c->QT[t]] = (int8_t)(t % Kg + c->Ki * Kg);
        c->edge_offset_map[c->QT[t]] = (int32_t)(t / Kg);
    }
    for (int32_t j = 0; j < Kg; j++) {
        free(c->feat_cache[j]); free(c->topo_indptr[j]); free(c->topo_col[j]);
        c->feat_cache[j] = (float*)xcalloc((size_t)ncap * (size_t)(D > 0 ? D : 1), 4);
        if (host_features) {
            for (int64_t r = 0; r < ncap; r++) {               /* FeatFillUp cache_impl.cuh:183-188 */
                int64_t t = r * Kg + j;
                if (t >= N) continue;
                memcpy(c->feat_cache[j] + r * D, host_features + (int64_t)c->QF[t] * D, (size_t)D * 4);
            }
        }
        /* GraphCache graph_storage.cu:81-104, kernels graph_storage_impl.cuh:33-53 */
        c->topo_indptr[j] = (int64_t*)xcalloc((size_t)ecap + 1, 8);
        for (int64_t r = 0; r < ecap; r++) {
            int64_t t = r * Kg + j;
            int64_t cnt = 0;
            if (t < N) { int32_t id = c->QT[t]; cnt = csr_index[id + 1] - csr_index[id]; }
            c->topo_indptr[j][r + 1] = c->topo_indptr[j][r] + cnt;
        }
        c->topo_col[j] = (int32_t*)xcalloc((size_t)c->topo_indptr[j][ecap], 4);
        for (int64_t r = 0; r < ecap; r++) {
            int64_t t = r * Kg + j;
            if (t >= N) continue;
            int32_t id = c->QT[t];
            int64_t cnt = csr_index[id + 1] - csr_index[id];
            memcpy(c->topo_col[j] + c->topo_indptr[j][r], csr_dst + csr_index[id], (size_t)cnt * 4);
        }
    }
}

/* SS/cache/cache.cu:614-670 (HybridInit) + :138-153 (HybridInsert) + cache_impl.cuh:113-123 (HybridInitPair) */
void lgo_hybrid_init(lgo_cache* c, const uint64_t* node_access, const float* host_features,
                     int32_t cpu_cache_capacity, int32_t gpu_cache_capacity)
{
    const int32_t N = c->total_num_nodes, D = c->feature_dim;
    const int64_t cpu_cap = cpu_cache_capacity, gpu_cap = gpu_cache_capacity;
    const uint64_t* one[1] = {node_access};
    c->Kg = 1;
    c->hybrid = 1;
    c->cpu_cache_capacity = cpu_cache_capacity;
    c->gpu_cache_capacity = gpu_cache_capacity;
    lgo_sorted_order(one, 1, N, c->QF, c->AF);             /* :630-635: this GPU's counters alone, iota, sort_by_key greater */
    c->node_capacity = (int32_t)(gpu_cap + cpu_cap);       /* :642 InitializeMap(gpu + cpu, 100) */
    c->edge_capacity = 0;                                  /* nothing is inserted into the two topology maps */
    for (int32_t i = 0; i < N; i++) {
        c->node_map[i] = LGO_CACHEMISS_FLAG;
        c->edge_index_map[i] = (int8_t)LGO_CACHEMISS_FLAG;
        c->edge_offset_map[i] = LGO_CACHEMISS_FLAG;
    }
    for (int64_t t = 0; t < cpu_cap + gpu_cap && t < N; t++)          /* HybridInitPair cache_impl.cuh:113-123 */
        c->node_map[c->QF[t]] = (int32_t)(t < gpu_cap ? cpu_cap + t : t - gpu_cap);
    free(c->feat_cache[0]); free(c->cpu_cache);
    c->feat_cache[0] = (float*)xcalloc((size_t)(gpu_cap > 0 ? gpu_cap : 1) * (size_t)(D > 0 ? D : 1), 4);   /* :653-657 */
    c->cpu_cache = (float*)xcalloc((size_t)(cpu_cap > 0 ? cpu_cap : 1) * (size_t)(D > 0 ? D : 1), 4);        /* :616 */
    if (host_features) {
        for (int64_t r = 0; r < gpu_cap && r < N; r++)
            memcpy(c->feat_cache[0] + r * D, host_features + (int64_t)c->QF[r] * D, (size_t)D * 4);
        for (int64_t r = 0; r < cpu_cap && gpu_cap + r < N; r++)
            memcpy(c->cpu_cache + r * D, host_features + (int64_t)c->QF[gpu_cap + r] * D, (size_t)D * 4);
    }
}

/* SS/cache/cache.cu:217-225: two bcht::find calls, sentinel -2 (as char for the index map) */
void lgo_find_topo(const lgo_cache* c, const int32_t* input_ids, int8_t* part_ind,
                   int32_t* part_off, int32_t batch_size)
{
    for (int32_t i = 0; i < batch_size; i++) {
        int32_t id = input_ids[i];
        part_ind[i] = id >= 0 ? c->edge_index_map[id] : (int8_t)LGO_CACHEMISS_FLAG;
        part_off[i] = id >= 0 ? c->edge_offset_map[id] : LGO_CACHEMISS_FLAG;
    }
}

/* SS/cache/cache.cu:180-215; c == NULL: nothing is cached, every find() returns the sentinel */
void lgo_find_feat(const lgo_cache* c, lgo_pool* p, int32_t op_id)
{
    int32_t node_off = p->node_counter[(op_id % LGO_INTRABATCH_CON) * 2];
    int32_t batch_size = p->node_counter[(op_id % LGO_INTRABATCH_CON) * 2 + 1];
    for (int32_t i = 0; i < batch_size; i++) {
        int32_t id = p->sampled_ids[node_off + i];
        p->cache_search_buffer[i] = (c && id >= 0) ? c->node_map[id] : LGO_CACHEMISS_FLAG;
    }
}

/* SS/engine/operator_impl.cu:502-519 -> SS/cache/cache.cu:726-748 -> cache_impl.cuh:239-272.
 * c == NULL restates serving before FillUp: no map, every row is a miss (cache_index = -2). */
void lgo_feature_cache_lookup(const lgo_cache* c, lgo_pool* p, const float* host_features,
                              int32_t op_id)
{
    lgo_counter_update(p->node_counter, p->edge_counter, op_id, 0, 0);          /* :515 */
    const int32_t D = c ? c->feature_dim : p->feature_dim;
    const int32_t cap = c ? c->node_capacity : 1;
    const int32_t N = p->total_num_nodes;
    int32_t node_off = p->node_counter[(op_id % LGO_INTRABATCH_CON) * 2];
    int32_t batch_size = p->node_counter[(op_id % LGO_INTRABATCH_CON) * 2 + 1];
    if (c && c->hybrid) {                                                       /* feat_cache_lookup, cache_impl.cuh:202-235 */
        const int32_t cpu_cap = c->cpu_cache_capacity, gpu_cap = c->gpu_cache_capacity;
        if (D <= 0) return;                                                     /* :216 */
        for (int32_t r = 0; r < batch_size; r++) {
            int32_t gidx = p->cache_search_buffer[r];
            float* dst = p->float_features + ((int64_t)node_off + r) * D;
            if (gidx < cpu_cap && gidx >= 0) {                                  /* :224-226 cache in cpu */
                int32_t fidx = gidx % cpu_cap;
                memcpy(dst, c->cpu_cache + (int64_t)fidx * D, (size_t)D * 4);
            } else if (gidx >= cpu_cap) {                                       /* :227-231 cache in gpu */
                int32_t fidx = (gidx - cpu_cap) % gpu_cap;
                memcpy(dst, c->feat_cache[0] + (int64_t)fidx * D, (size_t)D * 4);
            } else if (host_features != NULL) {                                 /* the kernel writes nothing; the storage tier's reader */
                int32_t fidx = p->sampled_ids[node_off + r];
                if (fidx >= 0) memcpy(dst, host_features + (int64_t)(fidx % N) * D, (size_t)D * 4);
            }
        }
        return;
    }
    if (D <= 0 || host_features == NULL) return;                                /* :256 */
    for (int32_t r = 0; r < batch_size; r++) {
        int32_t gidx = p->cache_search_buffer[r];
        float* dst = p->float_features + ((int64_t)node_off + r) * D;
        if (gidx < 0) {                                                         /* :262-266 */
            int32_t fidx = p->sampled_ids[node_off + r];
            if (fidx >= 0) memcpy(dst, host_features + (int64_t)(fidx % N) * D, (size_t)D * 4);
        } else {                                                                /* :267-269 */
            int32_t didx = gidx / cap, fidx = gidx % cap;
            memcpy(dst, c->feat_cache[didx] + (int64_t)fidx * D, (size_t)D * 4);
        }
    }
}

/* Op order of GPURunner::RunOnce (SS/engine/server.cu:302-332) / RunPreSc (:285-300). */
int64_t lgo_run_batch(lgo_pool* p, const lgo_graph* g, const lgo_cache* c,
                      const float* host_features,
                      const int32_t* all_ids, const int32_t* all_labels, int32_t total_cap,
                      int32_t batch_size, int32_t counter, const int32_t* fanout, int32_t hop_num,
                      int32_t mode, int is_presc, uint64_t* node_access_time,
                      uint64_t* edge_access_time)
{
    const int use_cache = (!is_presc && c != NULL);
    const int serve = !is_presc;
    lgo_batch_generate(p, all_ids, all_labels, total_cap, batch_size, counter, hop_num);  /* op 0 */
    if (serve) {
        lgo_find_feat(c, p, 0);
        lgo_feature_cache_lookup(c, p, host_features, 1);                                 /* op 1 */
    }
    for (int32_t h = 0; h < hop_num; h++) {
        int32_t op = LGO_INTRABATCH_CON * (h + 1);                                        /* op 3h+3 */
        int32_t frontier = (op == LGO_INTRABATCH_CON) ? p->node_counter[1] : p->edge_counter[1];
        const int32_t* input_ids = (op == LGO_INTRABATCH_CON) ? p->sampled_ids
                                                              : p->agg_src_ids + p->edge_counter[0];
        if (use_cache) {
            lgo_find_topo(c, input_ids, p->tmp_part_ind, p->tmp_part_off, frontier > 0 ? frontier : 0);
            lgo_random_sample(p, g, op, fanout[h], p->tmp_part_ind, p->tmp_part_off, 0, NULL);
        } else {
            /* no cache: every row comes from slot P, as in pre_sample */
            lgo_random_sample(p, g, op, fanout[h], NULL, NULL, 1, is_presc ? edge_access_time : NULL);
        }
        lgo_construct_graph(p);
        lgo_counter_update(p->node_counter, p->edge_counter, op, 0, 0);
        if (serve) {
            lgo_find_feat(c, p, op);
            lgo_feature_cache_lookup(c, p, host_features, op + 1);                        /* op 3h+4 */
        }
    }
    int64_t edges = p->edge_counter[LGO_INTRABATCH_CON * 3 + hop_num];
    /* IOComplete operator_impl.cu:551-580: ClearPosMap then CacheProfiling */
    if (mode == LGO_TRAINMODE) {
        if (is_presc && node_access_time) lgo_hotness_measure(p, node_access_time);
        lgo_clear_pos_map(p);
    }
    return edges;
}

/* ------------------------------------------------------------------------------------------
 * Step arithmetic: SS/engine/ipc_service.cu:60-132 (Coordinate), :130-132, :213-253.
 * ------------------------------------------------------------------------------------------ */
void lgo_coordinate(lgo_steps* s, int32_t partition_count, const int32_t* train_num,
                    const int32_t* valid_num, const int32_t* test_num, int32_t raw_batch_size,
                    int32_t epoch)
{
    memset(s, 0, sizeof(*s));
    s->epoch = epoch;
    s->raw_batch_size = raw_batch_size;
    int32_t min_train = 1000000000;
    for (int i = 0; i < partition_count; i++) if (train_num[i] < min_train) min_train = train_num[i];
    s->train_step = (min_train - 1) / raw_batch_size;
    for (int i = 0; i < partition_count; i++) s->train_bs[i] = raw_batch_size;
    int32_t max_valid = 0, max_test = 0;
    for (int i = 0; i < partition_count; i++) if (valid_num[i] > max_valid) max_valid = valid_num[i];
    s->valid_step = (max_valid - 1) / 512 + 1;
    for (int i = 0; i < partition_count; i++) s->valid_bs[i] = (valid_num[i] - 1) / s->valid_step + 1;
    for (int i = 0; i < partition_count; i++) if (test_num[i] > max_test) max_test = test_num[i];
    s->test_step = (max_test - 1) / 512 + 1;
    for (int i = 0; i < partition_count; i++) s->test_bs[i] = (test_num[i] - 1) / s->test_step + 1;
}

int32_t lgo_max_step(const lgo_steps* s)
{
    return ((s->train_step + s->valid_step) * s->epoch) + s->test_step;
}

int32_t lgo_current_mode(const lgo_steps* s, int32_t gb)
{
    if (gb < (s->train_step + s->valid_step) * s->epoch) {
        int32_t e = gb % (s->train_step + s->valid_step);
        return e < s->train_step ? LGO_TRAINMODE : LGO_VALIDMODE;
    }
    return LGO_TESTMODE;
}

int32_t lgo_local_batch_id(const lgo_steps* s, int32_t gb)
{
    if (gb < (s->train_step + s->valid_step) * s->epoch) {
        int32_t e = gb % (s->train_step + s->valid_step);
        return e < s->train_step ? e : e - s->train_step;
    }
    return (gb - ((s->train_step + s->valid_step) * s->epoch)) % s->test_step;
}

int32_t lgo_current_batchsize(const lgo_steps* s, int32_t dev_id, int32_t mode)
{
    if (mode == LGO_TRAINMODE) return s->train_bs[dev_id];
    if (mode == LGO_VALIDMODE) return s->valid_bs[dev_id];
    return s->test_bs[dev_id];
}

/* ------------------------------------------------------------------------------------------
 * CPU baseline for bench.py: T threads, each with a private pool, walk disjoint batches.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const lgo_graph* g; int32_t N; const int32_t* all_ids; int32_t total_cap; int32_t batch_size;
    const int32_t* fanout; int32_t hop_num; int32_t first, last, stride; int32_t num_ids;
    const float* features; int32_t D; int64_t edges, nodes;
} lgo_bench_arg;

static void* lgo_bench_thread(void* arg_)
{
    lgo_bench_arg* a = (lgo_bench_arg*)arg_;
    int64_t rows = a->features ? a->num_ids : 1;
    lgo_pool* p = lgo_pool_create(a->N, a->num_ids, a->batch_size, rows, a->D);
    int32_t* zero_labels = (int32_t*)xcalloc(a->total_cap, 4);
    for (int32_t b = a->first; b < a->last; b += a->stride) {
        /* serve mode without a cache: topology from slot P, every feature row a miss (full table) */
        a->edges += lgo_run_batch(p, a->g, NULL, a->features, a->all_ids, zero_labels, a->total_cap,
                                  a->batch_size, b, a->fanout, a->hop_num, LGO_VALIDMODE, 0, NULL, NULL);
        a->nodes += p->node_counter[LGO_INTRABATCH_CON * 3 + a->hop_num];
        /* valid mode leaves position_map dirty exactly as the reference does; harmless */
    }
    free(zero_labels);
    lgo_pool_destroy(p);
    return NULL;
}

int64_t lgo_bench_batches(const lgo_graph* g, int32_t total_num_nodes, const int32_t* all_ids,
                          int32_t total_cap, int32_t batch_size, const int32_t* fanout,
                          int32_t hop_num, int32_t first_batch, int32_t num_batches,
                          int32_t threads, const float* features, int32_t feature_dim,
                          double* seconds_out, int64_t* nodes_out)
{
    if (threads < 1) threads = 1;
    int64_t num_ids = batch_size, per = batch_size;
    for (int h = 0; h < hop_num; h++) { per *= fanout[h]; num_ids += per; }
    pthread_t* th = (pthread_t*)xcalloc(threads, sizeof(pthread_t));
    lgo_bench_arg* args = (lgo_bench_arg*)xcalloc(threads, sizeof(lgo_bench_arg));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        args[t] = (lgo_bench_arg){g, total_num_nodes, all_ids, total_cap, batch_size, fanout, hop_num,
                                  first_batch + t, first_batch + num_batches, threads,
                                  (int32_t)num_ids, features, feature_dim, 0, 0};
        pthread_create(&th[t], NULL, lgo_bench_thread, &args[t]);
    }
    int64_t edges = 0, nodes = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); edges += args[t].edges; nodes += args[t].nodes; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds_out) *seconds_out = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (nodes_out) *nodes_out = nodes;
    free(th); free(args);
    return edges;
}
