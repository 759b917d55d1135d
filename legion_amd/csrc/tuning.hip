// tuning.hip -- the one place where the LEGION_* tuning environment is parsed (include/legion_hip.h section 6).
//
// The reference has compile-time constants only (SS/include/system_config.cuh); this build has run-time switches for
// HOW the path runs (bucket and tile counts, stream layout of the Runner, where tables live), never WHAT it computes --
// every combination is parity-tested against the oracle.  Switches whose measured alternative lost were removed in round 5
// (DESIGN_HISTORY.md keeps the measurements).
// The library keeps one process-wide LegionTuning; launch paths read it through lg::tuning() and never call getenv.
#include "legion_core.h"

#include <cstring>
#include <initializer_list>
#include <mutex>
#include <string>
#include <utility>

namespace {
LegionTuning g_tuning;
bool g_tuning_valid = false;     // parsed at least once
bool g_tuning_pinned = false;    // set programmatically: creation-time refreshes keep it
std::mutex g_tuning_mu;

int env_int(const char* name, int dflt)
{
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

void parse_env(LegionTuning& t)
{
    memset(&t, 0, sizeof(t));
    t.lds_known_cap = env_int("LEGION_LDS_KNOWN_CAP", 0);
    t.lds_claim_cap = env_int("LEGION_LDS_CLAIM_CAP", 0);
    t.arena_scatter_mb = env_int("LEGION_ARENA_SCATTER_MB", 2);
    t.lds_part_wg = env_int("LEGION_LDS_PART_WG", 8192);
    t.lds_small_buckets = env_int("LEGION_LDS_SMALL_BUCKETS", 0);
    t.sample_max_wg = env_int("LEGION_SAMPLE_MAX_WG", 4096);
    t.gather_rows_per_wg = env_int("LEGION_GATHER_ROWS", 0);
    t.col_slots = env_int("LEGION_COL_SLOTS", -1);
    t.weave_priority = env_int("LEGION_WEAVE_PRIORITY", -1);
    t.runner_graph = env_int("LEGION_RUNNER_GRAPH", 1);
    t.runner_lanes = env_int("LEGION_RUNNER_LANES", 0);
    t.runner_ho_stream = env_int("LEGION_RUNNER_HO_STREAM", 2);
    t.runner_spin_us = env_int("LEGION_RUNNER_SPIN_US", -1);
    t.runner_overflow = env_int("LEGION_RUNNER_OVERFLOW", 1);
    t.runner_stats = getenv("LEGION_RUNNER_STATS") != nullptr ? 1 : 0;
    auto word = [](const char* name, std::initializer_list<std::pair<const char*, int>> words, int dflt) {
        const char* e = getenv(name);
        if (!e || !*e) return dflt;
        std::string all;
        for (const auto& w : words) {
            if (strcmp(e, w.first) == 0) return w.second;
            all += (all.empty() ? "" : "|") + std::string(w.first);
        }
        printf("legion_hip: %s=%s is not one of %s\n", name, e, all.c_str());
        exit(EXIT_FAILURE);
    };
    t.runner_handover = word("LEGION_RUNNER_HANDOVER", {{"auto", 0}, {"gather", 1}}, 0);
    t.runner_slots = env_int("LEGION_RUNNER_SLOTS", 3);
    t.peer_gather = word("LEGION_PEER_GATHER", {{"direct", 0}, {"bulk", 1}}, 0);
    t.hotness_reduce = word("LEGION_HOTNESS_REDUCE", {{"auto", -1}, {"p2p", 0}, {"rccl", 1}}, -1);
    t.markers = env_int("LEGION_MARKERS", 1);
    t.table_placement = 0;
    if (const char* e = getenv("LEGION_TABLE_PLACEMENT")) t.table_placement = strcmp(e, "pinned") == 0 ? 1 : 0;
    t.shm_mirror = getenv("LEGION_NO_SHM_MIRROR") != nullptr ? 0 : 1;
    t.link_counters = 0;
    if (const char* e = getenv("LEGION_LINK_COUNTERS")) {
        unsigned long long a = 0, b = 0;
        if (strcmp(e, "measured") == 0) t.link_counters = 1;
        else if (strcmp(e, "smi") == 0) t.link_counters = 2;
        else if (sscanf(e, "%llu,%llu", &a, &b) == 2) {
            t.link_counters = 3;
            t.link_counter_values[0] = a;
            t.link_counter_values[1] = b;
        } else if (strcmp(e, "v2") != 0 && *e) {
            printf("legion_hip: LEGION_LINK_COUNTERS=%s is not one of v2|measured|smi|<a>,<b>\n", e);
            exit(EXIT_FAILURE);
        }
    }
}
}  // namespace

namespace lg {
// A SNAPSHOT by value: the process-wide copy is replaced as a whole under the lock (parsed into a local first -- parse_env may
// exit() on a bad value and must not do so holding the lock), so a launch path running beside another thread's refresh sees
// either the old values or the new ones, never a half-written struct (a transient sample_max_wg = 0 once baked gx = 1 into a
// captured hipGraph for the life of its pipeline).
LegionTuning tuning()
{
    {
        std::lock_guard<std::mutex> lk(g_tuning_mu);
        if (g_tuning_valid) return g_tuning;
    }
    LegionTuning t;
    parse_env(t);
    std::lock_guard<std::mutex> lk(g_tuning_mu);
    if (!g_tuning_valid) {
        g_tuning = t;
        g_tuning_valid = true;
    }
    return g_tuning;
}

// called where objects are created (pool, pipeline, server): the environment as it is NOW decides, unless a host program
// installed its own values
void tuning_refresh()
{
    {
        std::lock_guard<std::mutex> lk(g_tuning_mu);
        if (g_tuning_pinned) return;
    }
    LegionTuning t;
    parse_env(t);
    std::lock_guard<std::mutex> lk(g_tuning_mu);
    if (g_tuning_pinned) return;
    g_tuning = t;
    g_tuning_valid = true;
}
}  // namespace lg

extern "C" void legion_tuning_from_env(void)
{
    LegionTuning t;
    parse_env(t);
    std::lock_guard<std::mutex> lk(g_tuning_mu);
    g_tuning = t;
    g_tuning_valid = true;
    g_tuning_pinned = false;
}

extern "C" void legion_tuning_get(LegionTuning* out)
{
    if (out) *out = lg::tuning();
}

extern "C" void legion_tuning_set(const LegionTuning* in)
{
    if (!in) return;
    std::lock_guard<std::mutex> lk(g_tuning_mu);
    g_tuning = *in;
    g_tuning_valid = true;
    g_tuning_pinned = true;
}
