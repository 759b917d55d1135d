// markers.hip -- roctx ranges around ops and launch groups (tracing row: the reference has none).
//
// `rocprofv3 --kernel-trace --marker-trace` then shows, beside every kernel, which op (eager launches) or which launch group
// (hipGraph replay: the ranges bracket the host-side graph launches of a group's phases) it belongs to; tools/trace_group.py
// folds such a trace into DESIGN.md's "what hides under what" timeline.  The marker library (librocprofiler-sdk-roctx) is
// opened on first use: without it, or with LegionTuning.markers = 0, a range is one predictable branch.
#include "legion_core.h"

#include <dlfcn.h>

#include <atomic>
#include <cstdarg>

namespace {
typedef int (*push_fn)(const char*);
typedef int (*pop_fn)(void);
push_fn g_push = nullptr;
pop_fn g_pop = nullptr;
std::atomic<int> g_state{0};     // 0 not tried, 1 available, 2 unavailable / switched off

bool markers_on()
{
    int st = g_state.load(std::memory_order_acquire);
    if (st == 0) {
        st = 2;
        if (lg::tuning().markers != 0) {
            void* h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) {
                g_push = (push_fn)dlsym(h, "roctxRangePushA");
                g_pop = (pop_fn)dlsym(h, "roctxRangePop");
                if (g_push && g_pop) st = 1;
            }
        }
        g_state.store(st, std::memory_order_release);
    }
    return st == 1;
}
}  // namespace

namespace lg {
Range::Range(const char* fmt, ...) : on_(markers_on())
{
    if (!on_) return;
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_push(buf);
}
Range::~Range() { if (on_) g_pop(); }
}  // namespace lg
