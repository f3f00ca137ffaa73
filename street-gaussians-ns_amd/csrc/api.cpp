// api.cpp — error plumbing shared by every entry point of libsgnrast.so (host only).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime.h>

#include <mutex>
#include <utility>
#include <vector>

#include "sgn_rast.h"

#define SGN_GRAPH_KEY_WORDS 12      // as in sgn_common.h (this file is built without the device headers' helpers)

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

// ---------------------------------------------------------------- launch chains replayed as HIP graphs
// (sgn_common.h: sgn_graph_find / sgn_graph_capture_begin / _end.)  A resource cache per (thread, device): up to 8
// instantiated graphs, least recently used out; one capture stream.  No configuration lives here: the key is the
// chain's own arguments.
namespace {
struct GraphEntry { int dev; uint64_t key[SGN_GRAPH_KEY_WORDS]; hipGraphExec_t exec; uint64_t used; };
struct GraphState { std::vector<GraphEntry> entries; std::vector<std::pair<int, hipStream_t>> streams; uint64_t tick = 0; };
thread_local GraphState g_graphs;
int graphs_on() {
    static const int on = [] { const char *e = getenv("SGN_HIP_GRAPHS"); return (e && e[0] == '1') ? 1 : 0; }();
    return on;
}
}  // namespace

int sgn_timing_enabled();

hipGraphExec_t sgn_graph_find(const uint64_t *key) {
    if (!graphs_on() || sgn_timing_enabled()) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (auto &e : g_graphs.entries)
        if (e.dev == dev && memcmp(e.key, key, sizeof(e.key)) == 0) { e.used = ++g_graphs.tick; return e.exec; }
    return nullptr;
}

hipStream_t sgn_graph_capture_begin() {
    if (!graphs_on() || sgn_timing_enabled()) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    hipStream_t cs = nullptr;
    for (auto &p : g_graphs.streams)
        if (p.first == dev) cs = p.second;
    if (cs == nullptr) {
        if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) return nullptr;
        g_graphs.streams.emplace_back(dev, cs);
    }
    if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return cs;
}

hipGraphExec_t sgn_graph_capture_end(const uint64_t *key) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    hipStream_t cs = nullptr;
    for (auto &p : g_graphs.streams)
        if (p.first == dev) cs = p.second;
    if (cs == nullptr) return nullptr;
    hipGraph_t graph = nullptr;
    if (hipStreamEndCapture(cs, &graph) != hipSuccess || graph == nullptr) { (void)hipGetLastError(); return nullptr; }
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess || exec == nullptr) { (void)hipGetLastError(); return nullptr; }
    if (g_graphs.entries.size() >= 8) {
        size_t lru = 0;
        for (size_t i = 1; i < g_graphs.entries.size(); ++i)
            if (g_graphs.entries[i].used < g_graphs.entries[lru].used) lru = i;
        (void)hipGraphExecDestroy(g_graphs.entries[lru].exec);
        g_graphs.entries.erase(g_graphs.entries.begin() + (long)lru);
    }
    GraphEntry ge;
    ge.dev = dev;
    memcpy(ge.key, key, sizeof(ge.key));
    ge.exec = exec;
    ge.used = ++g_graphs.tick;
    g_graphs.entries.push_back(ge);
    return exec;
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
