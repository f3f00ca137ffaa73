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

// ---------------------------------------------------------------- one C-ABI call per autograd node (round 5)
// sgn_rasterize_fwd_all: the whole forward of `rasterize_gaussians` for the plain case (the full scene, gather mode) — the
// sequence the Python host otherwise drives call by call (sgn_rast/ops.py `_RasterizeGaussians.forward`): first half of
// the binning, the asynchronous read-back of the intersection count, the per-Gaussian rows, the SPECULATIVE second half
// of the binning (emission + tile sort + bins sized by the caller's capacity, the true count read on the device), the
// wait for the count — the path's one host sync, as upstream's `.item()` — the launch order and the forward kernels.
// Host only: it calls the library's own entry points on the caller's stream and carves its temporaries from ONE arena.
namespace {
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
int sync_event(hipEvent_t *ev) {          // one untimed event per (thread, device), like the fork / join pair above
    struct One { int dev; hipEvent_t e; };
    static thread_local std::vector<One> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    for (auto &p : cache)
        if (p.dev == dev) { *ev = p.e; return 0; }
    One o;
    o.dev = dev;
    if (hipEventCreateWithFlags(&o.e, hipEventDisableTiming) != hipSuccess) return 2;
    cache.push_back(o);
    *ev = o.e;
    return 0;
}
}  // namespace

// sgn_project_fwd_all: `project_gaussians` as ONE call — upstream's quats assertion as a device pass whose flag travels to
// the host while the projection and (optionally) the depth ranking of the coming binning are already queued; the host
// waits for the flag last (the "eager" check of sgn_rast/ops.py, which otherwise takes three calls).
extern "C" __attribute__((visibility("default")))
int sgn_project_fwd_all(int n, const float *means3d, const float *scales, float glob_scale, const float *quats,
                        const float *viewmat12, float fx, float fy, float cx, float cy, int img_h, int img_w,
                        int block_width, float clip_thresh, float *cov3d, float *xys, float *depths, int32_t *radii,
                        float *conics, float *compensation, int32_t *num_tiles_hit, int check_quats, float quat_tol,
                        int32_t *flag_dev, int32_t *flag_pinned, int32_t *gid_by_rank, void *rank_ws,
                        size_t rank_ws_bytes, int sort_rank_mode, int32_t *quats_bad_host, sgn_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t ev = nullptr;
    int32_t pageable = 0;
    int32_t *dst = flag_pinned ? flag_pinned : &pageable;
    if (check_quats) {
        if (!flag_dev || !quats_bad_host) { sgn_set_error("sgn_project_fwd_all: check_quats needs flag_dev and quats_bad_host"); return -1; }
        int rc = sgn_check_unit_quats(n, quats, quat_tol, flag_dev, stream);
        if (rc) return rc;
        hipError_t e = hipMemcpyAsync(dst, flag_dev, sizeof(int32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && sync_event(&ev) != 0) e = hipErrorUnknown;
        if (e == hipSuccess) e = hipEventRecord(ev, s);
        if (e != hipSuccess) { sgn_set_error("sgn_project_fwd_all: flag read-back: %s", hipGetErrorString(e)); return (int)e; }
    }
    int rc = sgn_project_fwd(n, means3d, scales, glob_scale, quats, viewmat12, fx, fy, cx, cy, img_h, img_w, block_width,
                             clip_thresh, cov3d, xys, depths, radii, conics, compensation, num_tiles_hit, stream);
    if (rc) return rc;
    if (gid_by_rank != nullptr && n > 0) {
        rc = sgn_depth_rank(n, depths, radii, gid_by_rank, rank_ws, rank_ws_bytes, sort_rank_mode, stream);
        if (rc) return rc;
    }
    if (check_quats) {
        const hipError_t e = hipEventSynchronize(ev);      // the projection (and the ranking) are queued: wait now
        if (e != hipSuccess) { sgn_set_error("sgn_project_fwd_all: %s", hipGetErrorString(e)); return (int)e; }
        *quats_bad_host = *dst;
    }
    return 0;
}

extern "C" __attribute__((visibility("default")))
size_t sgn_rasterize_arena_bytes(int n, int64_t isect_capacity) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    return 2 * al256(nn * 4) + al256(nn * 32) + al256(sgn_bin_prepare_workspace_bytes(n)) +
           al256(sgn_bin_intersect_workspace_bytes(isect_capacity)) + 256;
}

extern "C" __attribute__((visibility("default")))
int sgn_rasterize_fwd_all(int n, const float *xys, const float *depths, const int32_t *radii, const float *conics,
                          const float *colors, const float *opacities, int opacity_is_logit, int cull, int img_h,
                          int img_w, int block_width, const float *background3, const int32_t *gid_by_rank_ready,
                          int quadrant_masks, float *out_img, float *final_Ts, int32_t *final_idx,
                          int32_t *gaussian_ids_sorted, int64_t isect_capacity, int32_t *tile_bins,
                          int32_t *tile_order, int32_t *tile_stats, void *rows, size_t rows_bytes,
                          void *order_scratch, size_t order_scratch_bytes, void *arena, size_t arena_bytes,
                          int32_t *count_pinned, const int32_t *extra_dev, int32_t *extra_pinned,
                          int64_t *n_isect_host, int sort_rank_mode, const sgn_raster_opts *opts,
                          sgn_stream_t stream) {
    if (n < 1 || !xys || !depths || !radii || !colors || !opacities || !background3 || !out_img || !final_Ts ||
        !final_idx || !gaussian_ids_sorted || !tile_bins || !tile_order || !tile_stats || !rows || !arena ||
        !n_isect_host || isect_capacity < 1 || block_width < 2 || block_width > 16 || img_h < 1 || img_w < 1) {
        sgn_set_error("sgn_rasterize_fwd_all: argument check failed");
        return -1;
    }
    if (arena_bytes < sgn_rasterize_arena_bytes(n, isect_capacity)) {
        sgn_set_error("sgn_rasterize_fwd_all: arena too small");
        return -2;
    }
    sgn_raster_opts o;
    sgn_raster_default_opts(&o);
    if (opts) o = *opts;
    if (!o.gather) {
        sgn_set_error("sgn_rasterize_fwd_all: gather mode only (opts->gather = 1)");
        return -3;
    }
    hipStream_t s = (hipStream_t)stream;
    const int tiles_x = (img_w + block_width - 1) / block_width, tiles_y = (img_h + block_width - 1) / block_width;
    const int n_tiles = tiles_x * tiles_y;
    char *p = (char *)arena;
    int32_t *cum_r = (int32_t *)p; p += al256((size_t)n * 4);
    int32_t *gid_own = (int32_t *)p; p += al256((size_t)n * 4);
    float *bin_recs = (float *)p; p += al256((size_t)n * 32);
    void *ws1 = p; const size_t ws1_bytes = sgn_bin_prepare_workspace_bytes(n); p += al256(ws1_bytes);
    void *ws2 = p; const size_t ws2_bytes = sgn_bin_intersect_workspace_bytes(isect_capacity);
    int32_t *gid = gid_by_rank_ready ? const_cast<int32_t *>(gid_by_rank_ready) : gid_own;
    const int do_cull = (cull && conics && opacities) ? 1 : 0;
    int rc = sgn_bin_prepare(n, xys, depths, radii, do_cull ? conics : nullptr, do_cull ? opacities : nullptr,
                             opacity_is_logit, do_cull, tiles_x, tiles_y, block_width, cum_r, gid,
                             gid_by_rank_ready ? 1 : 0, bin_recs, ws1, ws1_bytes, sort_rank_mode, stream);
    if (rc) return rc;
    // the count starts its way to the host now; everything queued below runs while it travels
    int32_t pageable = -1;
    int32_t *dst = count_pinned ? count_pinned : &pageable;
    hipError_t e = hipMemcpyAsync(dst, cum_r + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && extra_dev && extra_pinned)
        e = hipMemcpyAsync(extra_pinned, extra_dev, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    hipEvent_t ev = nullptr;
    if (e == hipSuccess && sync_event(&ev) != 0) e = hipErrorUnknown;
    if (e == hipSuccess) e = hipEventRecord(ev, s);
    if (e != hipSuccess) { sgn_set_error("sgn_rasterize_fwd_all: count read-back: %s", hipGetErrorString(e)); return (int)e; }
    rc = sgn_raster_build_rows(n, xys, conics, colors, opacities, opacity_is_logit, 0, n, 0, rows, rows_bytes, nullptr,
                               stream);
    if (rc) return rc;
    rc = sgn_bin_intersect(n, isect_capacity, bin_recs, cum_r, gid, tiles_x, tiles_y, block_width, gaussian_ids_sorted,
                           tile_bins, quadrant_masks, ws2, ws2_bytes, cum_r + (n - 1), sort_rank_mode, stream);
    if (rc) return rc;
    e = hipEventSynchronize(ev);                       // the path's one host sync (upstream: `.item()` on the count)
    if (e != hipSuccess) { sgn_set_error("sgn_rasterize_fwd_all: %s", hipGetErrorString(e)); return (int)e; }
    const int64_t count = (int64_t)*dst;
    *n_isect_host = count;
    if (count > isect_capacity) return SGN_E_CAPACITY;   // the list did not fit: call again with more room
    if (count < 1) return 0;                             // nothing visible: the caller writes the background image
    rc = sgn_tile_order(n_tiles, tile_bins, nullptr, (o.waves_fwd == 2) ? (o.adapt_fwd > 0 ? o.adapt_fwd : 1024) : 0, 0,
                        tile_order, order_scratch, order_scratch_bytes, stream);
    if (rc) return rc;
    sgn_raster_opts oo = o;
    oo.ids_qmask = quadrant_masks ? 1 : 0;
    return sgn_raster_fwd(img_h, img_w, block_width, n, count, gaussian_ids_sorted, tile_bins, xys, conics, colors,
                          opacities, opacity_is_logit, 0, n, 0, background3, out_img, final_Ts, final_idx, rows,
                          rows_bytes, 1, tile_order, tile_stats, nullptr, nullptr, nullptr, &oo, stream);
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
