/* dgl_semantics.c -- CPU baseline with DGL NeighborSampler SEMANTICS (test infrastructure; never shipped).
 *
 * BASELINE.json configs[0] / north_star name "DGL's CPU NeighborSampler timed on the same box's host cores" as the reported
 * CPU baseline.  DGL is a third-party dependency of the reference's training scripts (training_backend/legion_graphsage.py
 * imports dgl; no version is pinned, the README says DGL 0.9) and is NOT installed in this image (`import dgl` fails, SURVEY
 * A.10), so its sampler cannot be run here.  This file restates the published semantics of
 *     dgl.dataloading.NeighborSampler(fanouts) = per layer  sample_neighbors(g, frontier, fanout, replace=False)
 *                                                + to_block(frontier_graph, dst_nodes=frontier)
 * exactly as SURVEY.md A.9 lists the differences from Legion's sampler:
 *   - WITHOUT replacement: a vertex with deg <= fan-out contributes all its neighbours, otherwise `fan-out` DISTINCT
 *     adjacency positions (Legion draws min(f, deg) times WITH replacement, operator_impl.cu:228-242);
 *   - the next hop expands the DE-DUPLICATED frontier (to_block's unique src nodes, destination nodes first, then new source
 *     nodes in order of first appearance); Legion expands the duplicated edge frontier (operator_impl.cu:244-281);
 *   - one block per hop (Legion's blocks are cumulative prefixes, ipc_cuda_kernel.cu:221-232).
 * It is a THROUGHPUT reference in the reference's own words, never a parity oracle: its batches differ from
 * Legion's by construction, its RNG is its own (xorshift64*, one stream per batch), and bench.py labels it
 * "dgl-semantics port".  tests/test_oracle_dgl_semantics.py checks the properties above on small graphs.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    uint64_t s;
} rng_t;
static inline uint64_t rng_next(rng_t* r)
{
    uint64_t x = r->s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    r->s = x;
    return x * 0x2545F4914F6CDD1DULL;
}
static inline uint32_t rng_below(rng_t* r, uint32_t n) { return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32); }

/* id -> position map of one block (to_block's IdHashMap): open addressing, power-of-two capacity, stamp = validity */
typedef struct {
    int32_t* key; int32_t* val; uint32_t* stamp; uint32_t mask; uint32_t cur;
} idmap_t;
static void idmap_init(idmap_t* m, int64_t cap_hint)
{
    uint32_t cap = 1024;
    while ((int64_t)cap < cap_hint * 2) cap <<= 1;
    m->key = (int32_t*)malloc((size_t)cap * 4);
    m->val = (int32_t*)malloc((size_t)cap * 4);
    m->stamp = (uint32_t*)calloc(cap, 4);
    m->mask = cap - 1;
    m->cur = 0;
}
static void idmap_free(idmap_t* m) { free(m->key); free(m->val); free(m->stamp); }
static inline void idmap_clear(idmap_t* m)
{
    if (++m->cur == 0) { memset(m->stamp, 0, ((size_t)m->mask + 1) * 4); m->cur = 1; }
}
/* returns the position of id, inserting it with position *count (and bumping *count) when new */
static inline int32_t idmap_get_or_add(idmap_t* m, int32_t id, int32_t* count, int32_t* order)
{
    uint32_t h = ((uint32_t)id * 2654435761u) & m->mask;
    for (;;) {
        if (m->stamp[h] != m->cur) {
            m->stamp[h] = m->cur; m->key[h] = id; m->val[h] = *count;
            order[*count] = id;
            return (*count)++;
        }
        if (m->key[h] == id) return m->val[h];
        h = (h + 1) & m->mask;
    }
}

typedef struct {
    idmap_t map;
    int32_t* frontier;      /* unique nodes of the current block: dst nodes first, then new src nodes */
    int32_t* next;          /* the next block's node list under construction */
    int32_t* src_pos;       /* per emitted edge: position of the sampled neighbour in `next` */
    int32_t* dst_pos;       /* per emitted edge: position of the node it was sampled for (= its index in `frontier`) */
    int64_t cap_nodes, cap_edges;
} dgl_scratch;

static void scratch_init(dgl_scratch* s, int32_t batch_size, const int32_t* fanout, int32_t hops)
{
    /* a block's src nodes include its dst nodes, so hop h samples around EVERY node found so far (the seeds again in hop 2):
       edges_h <= nodes_h * f_h, nodes_{h+1} <= nodes_h + edges_h */
    int64_t nodes = batch_size, edges = 0;
    for (int h = 0; h < hops; h++) { const int64_t e = nodes * fanout[h]; if (e > edges) edges = e; nodes += e; }
    s->cap_nodes = nodes; s->cap_edges = edges > 0 ? edges : 1;
    idmap_init(&s->map, nodes);
    s->frontier = (int32_t*)malloc((size_t)nodes * 4);
    s->next = (int32_t*)malloc((size_t)nodes * 4);
    s->src_pos = (int32_t*)malloc((size_t)s->cap_edges * 4);
    s->dst_pos = (int32_t*)malloc((size_t)s->cap_edges * 4);
}
static void scratch_free(dgl_scratch* s)
{
    idmap_free(&s->map); free(s->frontier); free(s->next); free(s->src_pos); free(s->dst_pos);
}

/* One mini-batch.  hop h samples around the (unique) frontier of hop h-1; fanout[0] is the hop next to the seeds (the
 * order Legion's --fanout uses; DGL lists the same sampler as NeighborSampler(reversed(fanout))).  Optional outputs (may be
 * NULL): per-hop edge counts, per-hop frontier sizes (hop_nodes[0] = seeds, hop_nodes[h+1] = nodes after hop h), the edges
 * of every hop as GLOBAL ids concatenated (src = sampled neighbour, dst = the node it was sampled for) and the final node
 * list.  Returns the number of edges emitted. */
static int64_t dgl_sample_one(dgl_scratch* s, const int64_t* indptr, const int32_t* col, const int32_t* seeds, int32_t n_seeds,
                              const int32_t* fanout, int32_t hops, rng_t* rng, int64_t* hop_edges, int32_t* hop_nodes,
                              int32_t* out_src, int32_t* out_dst, int32_t* out_nodes)
{
    int32_t n_front = 0;
    idmap_clear(&s->map);
    for (int32_t i = 0; i < n_seeds; i++) (void)idmap_get_or_add(&s->map, seeds[i], &n_front, s->frontier);   /* unique seeds */
    if (hop_nodes) hop_nodes[0] = n_front;
    int64_t total = 0;
    for (int h = 0; h < hops; h++) {
        const int32_t f = fanout[h];
        /* to_block: the block's src nodes start with its dst nodes (the frontier), new ones are appended */
        idmap_clear(&s->map);
        int32_t n_next = 0;
        for (int32_t i = 0; i < n_front; i++) (void)idmap_get_or_add(&s->map, s->frontier[i], &n_next, s->next);
        int64_t e = 0;
        for (int32_t i = 0; i < n_front; i++) {
            const int64_t lo = indptr[s->frontier[i]];
            const int64_t deg = indptr[s->frontier[i] + 1] - lo;
            if (deg <= f) {                                   /* all neighbours */
                for (int64_t k = 0; k < deg; k++) {
                    const int32_t nb = col[lo + k];
                    if (out_src) { out_src[total + e] = nb; out_dst[total + e] = s->frontier[i]; }
                    s->src_pos[e] = idmap_get_or_add(&s->map, nb, &n_next, s->next);
                    s->dst_pos[e] = i;
                    e++;
                }
            } else {                                          /* f distinct adjacency positions: Floyd's algorithm */
                uint32_t picked[64];
                int32_t np = 0;
                for (int64_t j = deg - f; j < deg; j++) {
                    uint32_t t = rng_below(rng, (uint32_t)(j + 1));
                    int dup = 0;
                    for (int32_t q = 0; q < np; q++) if (picked[q] == t) { dup = 1; break; }
                    if (dup) t = (uint32_t)j;
                    picked[np++] = t;
                    const int32_t nb = col[lo + t];
                    if (out_src) { out_src[total + e] = nb; out_dst[total + e] = s->frontier[i]; }
                    s->src_pos[e] = idmap_get_or_add(&s->map, nb, &n_next, s->next);
                    s->dst_pos[e] = i;
                    e++;
                }
            }
        }
        if (hop_edges) hop_edges[h] = e;
        total += e;
        int32_t* t = s->frontier; s->frontier = s->next; s->next = t;       /* the block's src nodes are the next frontier */
        n_front = n_next;
        if (hop_nodes) hop_nodes[h + 1] = n_front;
    }
    if (out_nodes) memcpy(out_nodes, s->frontier, (size_t)n_front * 4);
    return total;
}

/* test entry: one batch with explicit outputs (arrays sized by the caller: edges <= sum_h B f1..fh, nodes <= B(1+f1+...)) */
int64_t lgo_dgl_sample_batch(const int64_t* indptr, const int32_t* col, const int32_t* seeds, int32_t n_seeds,
                             const int32_t* fanout, int32_t hops, uint64_t rng_seed, int64_t* hop_edges, int32_t* hop_nodes,
                             int32_t* out_src, int32_t* out_dst, int32_t* out_nodes)
{
    if (hops < 1 || hops > 8) return -1;
    for (int h = 0; h < hops; h++) if (fanout[h] < 1 || fanout[h] > 64) return -1;
    dgl_scratch s;
    scratch_init(&s, n_seeds, fanout, hops);
    rng_t r = {rng_seed * 0x9E3779B97F4A7C15ULL + 0x632BE59BD9B4E019ULL};
    const int64_t e = dgl_sample_one(&s, indptr, col, seeds, n_seeds, fanout, hops, &r, hop_edges, hop_nodes, out_src, out_dst, out_nodes);
    scratch_free(&s);
    return e;
}

typedef struct {
    const int64_t* indptr; const int32_t* col; const int32_t* all_ids; int32_t total_cap, batch_size;
    const int32_t* fanout; int32_t hops, first, last, stride; int64_t edges, nodes;
} dgl_arg;

static void* dgl_thread(void* a_)
{
    dgl_arg* a = (dgl_arg*)a_;
    dgl_scratch s;
    scratch_init(&s, a->batch_size, a->fanout, a->hops);
    int32_t hop_nodes[16];
    for (int32_t b = a->first; b < a->last; b += a->stride) {
        int64_t off = (int64_t)b * a->batch_size;
        int32_t n = a->batch_size;
        if (off + n > a->total_cap) n = (int32_t)(a->total_cap - off);
        if (n <= 0) break;
        rng_t r = {((uint64_t)b + 1) * 0x9E3779B97F4A7C15ULL};
        a->edges += dgl_sample_one(&s, a->indptr, a->col, a->all_ids + off, n, a->fanout, a->hops, &r, NULL, hop_nodes, NULL, NULL, NULL);
        a->nodes += hop_nodes[a->hops];
    }
    scratch_free(&s);
    return NULL;
}

/* bounded multi-thread baseline for bench.py: `threads` workers walk disjoint batches (DGL's DataLoader workers do the
 * same), same seed batches as the GPU path; returns the edges emitted, fills the wall time and the input nodes produced */
int64_t lgo_dgl_bench_batches(const int64_t* indptr, const int32_t* col, const int32_t* all_ids, int32_t total_cap,
                              int32_t batch_size, const int32_t* fanout, int32_t hops, int32_t first_batch, int32_t num_batches,
                              int32_t threads, double* seconds_out, int64_t* nodes_out)
{
    if (threads < 1) threads = 1;
    if (hops < 1 || hops > 8) return -1;
    for (int h = 0; h < hops; h++) if (fanout[h] < 1 || fanout[h] > 64) return -1;
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    dgl_arg* args = (dgl_arg*)calloc((size_t)threads, sizeof(dgl_arg));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        args[t] = (dgl_arg){indptr, col, all_ids, total_cap, batch_size, fanout, hops, first_batch + t,
                            first_batch + num_batches, threads, 0, 0};
        pthread_create(&th[t], NULL, dgl_thread, &args[t]);
    }
    int64_t edges = 0, nodes = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); edges += args[t].edges; nodes += args[t].nodes; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds_out) *seconds_out = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (nodes_out) *nodes_out = nodes;
    free(th); free(args);
    return edges;
}
