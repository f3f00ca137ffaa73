// api.cpp — error plumbing shared by every entry point of libsgnrast.so (host only).
#include <stdarg.h>
#include <stdio.h>

#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "sgn_rast.h"

static thread_local char g_err[512] = "";

void sgn_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" __attribute__((visibility("default"))) const char *sgn_last_error(void) { return g_err; }
extern "C" __attribute__((visibility("default"))) int sgn_version(void) { return 100; }

// ---------------------------------------------------------------- fork / join events
// Two untimed events per (thread, device) for entry points that fan work out to a caller-provided auxiliary stream
// (sgn_raster_bwd).  Created on first use and kept: a resource cache like the error string above, not configuration.
int sgn_fork_events(hipEvent_t *fork, hipEvent_t *join) {
    struct Pair { int dev; hipEvent_t a, b; };
    static thread_local std::vector<Pair> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    for (auto &p : cache)
        if (p.dev == dev) { *fork = p.a; *join = p.b; return 0; }
    Pair p;
    p.dev = dev;
    if (hipEventCreateWithFlags(&p.a, hipEventDisableTiming) != hipSuccess) return 2;
    if (hipEventCreateWithFlags(&p.b, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(p.a); return 3; }
    cache.push_back(p);
    *fork = p.a; *join = p.b;
    return 0;
}

// ---------------------------------------------------------------- kernel timing (bench/profiles)
// Opt-in: when enabled every timed launch is bracketed by hipEventRecord on the SAME stream the
// kernel is launched on; sgn_timing_get() synchronises the recorded events and sums them.
namespace {
struct Span { hipEvent_t a, b; int slot; };
std::mutex g_mu;
std::vector<Span> g_spans;
int g_timing = 0;
}  // namespace

int sgn_timing_enabled() { return g_timing; }

void sgn_timing_begin(int slot, void *stream) {
    if (!g_timing) return;
    Span s;
    s.slot = slot;
    if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return;
    (void)hipEventRecord(s.a, (hipStream_t)stream);
    std::lock_guard<std::mutex> lk(g_mu);
    g_spans.push_back(s);
}

void sgn_timing_end(int slot, void *stream) {
    if (!g_timing) return;
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = g_spans.size(); i-- > 0;)
        if (g_spans[i].slot == slot) { (void)hipEventRecord(g_spans[i].b, (hipStream_t)stream); return; }
}

extern "C" __attribute__((visibility("default"))) void sgn_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &s : g_spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    g_spans.clear();
    g_timing = on ? 1 : 0;
}

extern "C" __attribute__((visibility("default"))) int sgn_timing_get(int slot, int *count, float *total_ms) {
    std::lock_guard<std::mutex> lk(g_mu);
    int c = 0;
    float t = 0.f;
    for (auto &s : g_spans) {
        if (s.slot != slot) continue;
        if (hipEventSynchronize(s.b) != hipSuccess) return 1;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) return 2;
        t += ms;
        ++c;
    }
    if (count) *count = c;
    if (total_ms) *total_ms = t;
    return 0;
}
