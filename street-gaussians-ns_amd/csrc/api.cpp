// api.cpp — error plumbing shared by every entry point of libsgnrast.so (host only).
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <hip/hip_runtime.h>

#include <mutex>
#include <utility>
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

// ---------------------------------------------------------------- one C-ABI call per autograd node (round 5)
// sgn_rasterize_fwd_all: the whole forward of `rasterize_gaussians` for the plain case (the full scene, gather mode) — the
// sequence the Python host otherwise drives call by call (sgn_rast/ops.py `_RasterizeGaussians.forward`): first half of
// the binning, the asynchronous read-back of the intersection count, the per-Gaussian rows, the SPECULATIVE second half
// of the binning (emission + tile sort + bins sized by the caller's capacity, the true count read on the device), the
// wait for the count — the path's one host sync, as upstream's `.item()` — the launch order and the forward kernels.
// Host only: it calls the library's own entry points on the caller's stream and carves its temporaries from ONE arena.
namespace {
// host time spent BLOCKED in the one-call entries' event waits (profiles: is a composite call's host time its launches or
// its wait for the device?); read and reset by sgn_timing_host_wait_us
thread_local double g_wait_us = 0.0;
thread_local long g_wait_n = 0;
inline hipError_t timed_event_sync(hipEvent_t ev) {
    timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    const hipError_t e = hipEventSynchronize(ev);
    clock_gettime(CLOCK_MONOTONIC, &b);
    g_wait_us += (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3;
    ++g_wait_n;
    return e;
}
// Round 6: a word of MAPPED pinned memory that a kernel stores its result into is POLLED by the host instead of being
// waited for through an event: an event record is a barrier packet on the stream (a ~6 us bubble on the device timeline,
// profiles/r05q_timeline_eager.md) and a host call, and its completion reaches the host later than the store itself.
// `ready()` is evaluated on volatile reads; after SPIN_FALLBACK_US without the word the stream is drained once (the cold
// path: a fault on the device, a platform where mapped writes are not visible before completion) and the caller checks
// again.  Time spent here counts as blocked time (sgn_timing_host_wait_us), like the event waits.
constexpr double SPIN_FALLBACK_US = 2.0e5;
#ifdef SGN_AB_EVENT_WAITS            // A/B builds only (profiles/scripts/build_variant.sh api eventwaits -DSGN_AB_EVENT_WAITS)
constexpr bool POLL_WAITS = false;   // round 5's form: an event behind the producing kernel, hipEventSynchronize
#else
constexpr bool POLL_WAITS = true;
#endif
template <class Ready>
inline hipError_t spin_wait(Ready ready, hipStream_t s) {
    timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    hipError_t e = hipSuccess;
    for (unsigned it = 1;; ++it) {
        if (ready()) break;
        __builtin_ia32_pause();
        if ((it & 1023u) == 0) {
            (void)hipStreamQuery(s);          // (a runtime that batches submissions flushes on a query; a no-op otherwise)
            (void)hipGetLastError();          // hipErrorNotReady is the expected answer: not an error of the next launch
            clock_gettime(CLOCK_MONOTONIC, &b);
            if ((b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3 > SPIN_FALLBACK_US) {
                e = hipStreamSynchronize(s);
                break;
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    clock_gettime(CLOCK_MONOTONIC, &b);
    g_wait_us += (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3;
    ++g_wait_n;
    return e;
}
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
// The device's view of a word of pinned host memory (nullptr: not mapped — the caller falls back to a copy command).
// Kernels store flags / counts there directly, so that a read-back needs only an event behind the kernel, not a copy.
int32_t *mapped(int32_t *pinned) {
    if (pinned == nullptr) return nullptr;
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, pinned, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return (int32_t *)d;
}
}  // namespace

// sgn_project_fwd_all: `project_gaussians` as ONE call — upstream's quats assertion rides the projection kernel (the
// quaternion is in its registers anyway; a failing row stamps the flag), the flag travels to the host while the depth
// ranking of the coming binning is already queued, and the host waits for it last (the "eager" check of sgn_rast/ops.py,
// which otherwise takes three calls and two more launches).  `flag_stamp` > 0: the caller guarantees that *flag_dev holds
// no value >= flag_stamp (a word zeroed ONCE and a call counter): nothing is cleared; <= 0: the call clears the flag.
// flag_pinned is TWO words (round 6): [0] receives the stamp from a failing row, [1] the stamp from the
// kernel's first lane — "this launch's stores have landed where the host looks" — both as system-scope atomic stores;
// a wait that does not find [1] == stamp drains the device once and, if the word is still missing, FAILS (a mapped write
// that is not visible at event completion must not read as "all quaternions passed").
int sgn_project_fwd_checked(int n, const float *means3d, const float *scales, float glob_scale, const float *quats,
                            const float *viewmat12, float fx, float fy, float cx, float cy, int img_h, int img_w,
                            int block_width, float clip_thresh, float *cov3d, float *xys, float *depths, int32_t *radii,
                            float *conics, float *compensation, int32_t *num_tiles_hit, int32_t *quat_flag,
                            float quat_tol, int32_t quat_stamp, int32_t *quat_ok, int semantics,
                            sgn_stream_t stream);       // project.hip

int sgn_publish_words(const int32_t *src_dev, int n, int32_t *dst_mapped, int32_t *flag_mapped, int32_t flag_value,
                      sgn_stream_t stream);     // project.hip

namespace {
int check_event(hipEvent_t *ev) {         // the quats check's OWN event per (thread, device): a deferred wait
    struct One { int dev; hipEvent_t e; };   // (check_quats = 2) stays correct whatever the library records in between
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
// after the check's event: has the launch's "ok" stamp landed, and did a row fail?
int read_quat_flag(const int32_t *flag2, int32_t stamp, int32_t *bad, const char *who) {
    const volatile int32_t *f = flag2;
    if (f[1] != stamp) {
        (void)hipDeviceSynchronize();                          // cold path: never seen on this platform
        if (f[1] != stamp) {
            sgn_set_error("%s: the quats check's stamp never reached host memory (mapped pinned write not visible "
                          "at event completion): refusing to report 'normalized'", who);
            return -8;
        }
    }
    *bad = (f[0] == stamp) ? 1 : 0;
    return 0;
}
}  // namespace

extern "C" __attribute__((visibility("default")))
int sgn_project_fwd_all(int n, const float *means3d, const float *scales, float glob_scale, const float *quats,
                        const float *viewmat12, float fx, float fy, float cx, float cy, int img_h, int img_w,
                        int block_width, float clip_thresh, float *cov3d, float *xys, float *depths, int32_t *radii,
                        float *conics, float *compensation, int32_t *num_tiles_hit, int check_quats, float quat_tol,
                        int32_t *flag_dev, int32_t flag_stamp, int32_t *flag_pinned, int32_t *gid_by_rank,
                        void *rank_ws, size_t rank_ws_bytes, int sort_rank_mode, int32_t *quats_bad_host,
                        int semantics, sgn_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t ev = nullptr;
    int32_t pageable[2] = {0, 0};
    int32_t *dst = flag_pinned ? flag_pinned : pageable;
    int32_t stamp = flag_stamp > 0 ? flag_stamp : 1;
    if (check_quats) {
        const int mode = check_quats;
        if (mode != 1 && mode != 2) { sgn_set_error("sgn_project_fwd_all: check_quats must be 0, 1 or 2"); return -3; }
        if ((!flag_dev && !(flag_stamp > 0 && flag_pinned)) || (mode == 1 && !quats_bad_host) || (mode == 2 && !flag_pinned)) { sgn_set_error("sgn_project_fwd_all: check_quats needs flag_dev and quats_bad_host (1) / flag_pinned (2)"); return -1; }
        if (flag_stamp <= 0) {
            if (!flag_dev) { sgn_set_error("sgn_project_fwd_all: flag_stamp <= 0 needs flag_dev"); return -1; }
            const hipError_t e = hipMemsetAsync(flag_dev, 0, sizeof(int32_t), s);
            if (e != hipSuccess) { sgn_set_error("sgn_project_fwd_all: flag clear: %s", hipGetErrorString(e)); return (int)e; }
        }
    }
    // flag_pinned mapped into the device's address space and stamped (flag_stamp > 0: stale slots never equal the stamp):
    // a failing row stores straight into it and the read-back needs no copy command behind the kernel
    int32_t *direct = (check_quats && flag_stamp > 0 && n > 0) ? mapped(flag_pinned) : nullptr;
    int rc = sgn_project_fwd_checked(n, means3d, scales, glob_scale, quats, viewmat12, fx, fy, cx, cy, img_h, img_w,
                                     block_width, clip_thresh, cov3d, xys, depths, radii, conics, compensation,
                                     num_tiles_hit, check_quats ? (direct ? direct : flag_dev) : nullptr, quat_tol,
                                     stamp, direct ? direct + 1 : nullptr, semantics, stream);
    if (rc) return rc;
    // flag_pinned[2] (round 6): "the projection kernel is COMPLETE" — a one-wave kernel behind it stores the stamp there
    // and the host polls that word; without a mapped slot (or for n = 0) an event behind a copy command does it, and
    // the word is set to -stamp so that the wait knows which of the two to use
    bool polled = false;
    if (check_quats) {
        hipError_t e = hipSuccess;
        if (direct && POLL_WAITS) {
            ((volatile int32_t *)flag_pinned)[2] = 0;
            rc = sgn_publish_words(nullptr, 0, nullptr, direct + 2, stamp, stream);
            if (rc) return rc;
            polled = true;
        } else {
            if (!direct) {                  // the copy command is the transport: its completion is the landing
                if (n > 0) e = hipMemcpyAsync(dst, flag_dev, sizeof(int32_t), hipMemcpyDeviceToHost, s);
                else dst[0] = 0;
                dst[1] = stamp;
            }
            if (flag_pinned) flag_pinned[2] = -stamp;
            if (e == hipSuccess && check_event(&ev) != 0) e = hipErrorUnknown;
            if (e == hipSuccess) e = hipEventRecord(ev, s);
        }
        if (e != hipSuccess) { sgn_set_error("sgn_project_fwd_all: flag read-back: %s", hipGetErrorString(e)); return (int)e; }
    }
    if (gid_by_rank != nullptr && n > 0) {
        rc = sgn_depth_rank(n, depths, radii, gid_by_rank, rank_ws, rank_ws_bytes, sort_rank_mode, stream);
        if (rc) return rc;
    }
    if (check_quats == 1) {
        // the ranking is queued behind the projection: wait now
        const hipError_t e = polled ? spin_wait([&] { return ((volatile int32_t *)flag_pinned)[2] == stamp; }, s)
                                    : timed_event_sync(ev);
        if (e != hipSuccess) { sgn_set_error("sgn_project_fwd_all: %s", hipGetErrorString(e)); return (int)e; }
        int32_t bad = 0;
        rc = read_quat_flag(dst, stamp, &bad, "sgn_project_fwd_all");
        if (rc) return rc;
        *quats_bad_host = (n > 0 && bad) ? 1 : 0;
    }
    return 0;
}

// The wait of a sgn_project_fwd_all(check_quats = 2) call, made by the same thread on the same device: the caller does
// its own host-side bookkeeping for the projection's outputs first, while the flag is still on its way (the device is
// busy behind it; what the host does before this wait is off the critical path).  The check has its own event, so other
// calls of this library may come in between.
extern "C" __attribute__((visibility("default")))
int sgn_project_check_wait(const int32_t *flag_pinned, int32_t flag_stamp, int32_t *quats_bad_host,
                           sgn_stream_t stream) {
    if (!flag_pinned || !quats_bad_host) { sgn_set_error("sgn_project_check_wait: NULL argument"); return -1; }
    const int32_t stamp = flag_stamp > 0 ? flag_stamp : 1;
    hipError_t e = hipSuccess;
    if (((const volatile int32_t *)flag_pinned)[2] == -stamp) {     // the call went the copy + event way
        hipEvent_t ev = nullptr;
        if (check_event(&ev) != 0) { sgn_set_error("sgn_project_check_wait: no event"); return -2; }
        e = timed_event_sync(ev);
    } else {                                                        // polled: the one-wave kernel behind the projection
        e = spin_wait([&] { return ((const volatile int32_t *)flag_pinned)[2] == stamp; }, (hipStream_t)stream);
    }
    if (e != hipSuccess) { sgn_set_error("sgn_project_check_wait: %s", hipGetErrorString(e)); return (int)e; }
    return read_quat_flag(flag_pinned, stamp, quats_bad_host, "sgn_project_check_wait");
}

int sgn_bin_prepare_total(int n, const float *xys, const float *depths, const int32_t *radii,
                          const float *conics, const float *opacities, int opacity_is_logit, int cull,
                          int tiles_x, int tiles_y, int block_width, int32_t *cum_by_rank,
                          int32_t *gid_by_rank, int rank_ready, float *bin_records, void *ws, size_t ws_bytes,
                          int sort_rank_mode, int32_t *total_host, const int32_t *extra_dev, int32_t *extra_host,
                          int semantics, sgn_stream_t stream);   // binning.hip
int sgn_bin_intersect_zero(int n, int64_t n_isect, const float *bin_records, const int32_t *cum_by_rank,
                           const int32_t *gid_by_rank, int tiles_x, int tiles_y, int block_width,
                           int32_t *gaussian_ids_sorted, int32_t *tile_bins, int quadrant_masks, void *ws,
                           size_t ws_bytes, const int32_t *n_isect_dev, int sort_rank_mode, int also_zero_words,
                           sgn_stream_t stream);                                          // binning.hip
int sgn_raster_fwd_precleared(int img_h, int img_w, int block_width, int n, int64_t n_isect,
                              const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
                              const float *conics, const float *colors, const float *opacities,
                              int opacity_is_logit, const float *background3, float *out_img, float *final_Ts,
                              int32_t *final_idx, void *recs_ws, size_t recs_ws_bytes, const int32_t *tile_order,
                              int32_t *tile_kmax, const float *depths, float *out_depth, const sgn_raster_opts *opts,
                              sgn_stream_t stream);   // raster.hip

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
                          int quadrant_masks, float *out_img, float *final_Ts, int32_t *final_idx, float *out_depth,
                          int32_t *gaussian_ids_sorted, int64_t isect_capacity, int32_t *tile_bins,
                          int32_t *tile_order, int32_t *tile_stats, void *rows, size_t rows_bytes,
                          void *order_scratch, size_t order_scratch_bytes, void *arena, size_t arena_bytes,
                          int32_t *count_pinned, const int32_t *extra_dev, int32_t *extra_pinned,
                          int64_t *n_isect_host, int sort_rank_mode, int semantics, const sgn_raster_opts *opts,
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
    // the count goes straight from the scan into count_pinned where that is mapped into the device's address space (no
    // copy command, no event, no bubble behind it: the host POLLS the word), by a copy + event otherwise; everything
    // queued below runs while it travels.  `extra` rides the same way (stored by the same thread, ahead of the count).
    int32_t *direct = mapped(count_pinned);
    int32_t *extra_direct = (direct && extra_dev && extra_pinned) ? mapped(extra_pinned) : nullptr;
    const bool polled = POLL_WAITS && direct != nullptr && (!(extra_dev && extra_pinned) || extra_direct != nullptr);
    if (!polled) extra_direct = nullptr;
    if (direct) *(volatile int32_t *)count_pinned = -1;            // poison: "not landed yet"
    int rc = sgn_bin_prepare_total(n, xys, depths, radii, do_cull ? conics : nullptr, do_cull ? opacities : nullptr,
                                   opacity_is_logit, do_cull, tiles_x, tiles_y, block_width, cum_r, gid,
                                   gid_by_rank_ready ? 1 : 0, bin_recs, ws1, ws1_bytes, sort_rank_mode, direct,
                                   extra_direct ? extra_dev : nullptr, extra_direct, semantics, stream);
    if (rc) return rc;
    int32_t pageable = -1;
    int32_t *dst = count_pinned ? count_pinned : &pageable;
    hipError_t e = hipSuccess;
    hipEvent_t ev = nullptr;
    if (!polled) {
        if (!direct) e = hipMemcpyAsync(dst, cum_r + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && extra_dev && extra_pinned && !extra_direct)
            e = hipMemcpyAsync(extra_pinned, extra_dev, sizeof(int32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && sync_event(&ev) != 0) e = hipErrorUnknown;
        if (e == hipSuccess) e = hipEventRecord(ev, s);
        if (e != hipSuccess) { sgn_set_error("sgn_rasterize_fwd_all: count read-back: %s", hipGetErrorString(e)); return (int)e; }
    }
    rc = sgn_raster_build_rows(n, xys, conics, colors, opacities, opacity_is_logit, 0, n, 0, rows, rows_bytes, nullptr,
                               stream);
    if (rc) return rc;
    // tile_stats laid out right behind tile_bins (ONE buffer of the caller's): the emission clears both, and neither the
    // bins nor the statistics cost a clear launch of their own
    const bool stats_behind_bins = tile_stats == tile_bins + 2 * (size_t)n_tiles;
    rc = sgn_bin_intersect_zero(n, isect_capacity, bin_recs, cum_r, gid, tiles_x, tiles_y, block_width,
                                gaussian_ids_sorted, tile_bins, quadrant_masks, ws2, ws2_bytes, cum_r + (n - 1),
                                sort_rank_mode, stats_behind_bins ? 2 * n_tiles : 0, stream);
    if (rc) return rc;
    // the path's one host sync (upstream: `.item()` on the count)
    if (polled) e = spin_wait([&] { return *(volatile int32_t *)dst != -1; }, s);
    else e = timed_event_sync(ev);
    if (e != hipSuccess) { sgn_set_error("sgn_rasterize_fwd_all: %s", hipGetErrorString(e)); return (int)e; }
    if (direct && *(volatile int32_t *)dst == -1) {    // never seen; the plain copy is the safety net
        e = hipMemcpy(dst, cum_r + (n - 1), sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e == hipSuccess && extra_dev && extra_pinned) e = hipMemcpy(extra_pinned, extra_dev, sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { sgn_set_error("sgn_rasterize_fwd_all: count copy: %s", hipGetErrorString(e)); return (int)e; }
    }
    const int64_t count = (int64_t)*(volatile int32_t *)dst;
    *n_isect_host = count;
    if (count > isect_capacity) return SGN_E_CAPACITY;   // the list did not fit: call again with more room
    if (count < 1) return 0;                             // nothing visible: the caller writes the background image
    rc = sgn_tile_order(n_tiles, tile_bins, nullptr, (block_width == 16) ? (o.adapt_fwd > 0 ? o.adapt_fwd : 1024) : 0, 0,
                        tile_order, order_scratch, order_scratch_bytes, stream);
    if (rc) return rc;
    sgn_raster_opts oo = o;
    oo.ids_qmask = quadrant_masks ? 1 : 0;
    // out_depth: the DEPTH CHANNEL rides the colour pass (sgn_raster_fwd's depths / out_depth; the scene graph's and the
    // reference's depth pass is then answered from it, sgn_depth_reuse)
    const float *dch = out_depth ? depths : nullptr;
    if (stats_behind_bins)
        return sgn_raster_fwd_precleared(img_h, img_w, block_width, n, count, gaussian_ids_sorted, tile_bins, xys, conics,
                                         colors, opacities, opacity_is_logit, background3, out_img, final_Ts, final_idx,
                                         rows, rows_bytes, tile_order, tile_stats, dch, out_depth, &oo, stream);
    return sgn_raster_fwd(img_h, img_w, block_width, n, count, gaussian_ids_sorted, tile_bins, xys, conics, colors,
                          opacities, opacity_is_logit, 0, n, 0, background3, out_img, final_Ts, final_idx, rows,
                          rows_bytes, 1, tile_order, tile_stats, dch, out_depth, nullptr, &oo, stream);
}

// ---------------------------------------------------------------- a sub-model pass over a CACHED list, one call (round 6)
// sgn_rasterize_window_all: the forward of a `rasterize_gaussians` call whose tensors are a ROW WINDOW of a scene that was
// binned a moment ago (the scene graph's objects-only / background-only accumulation passes,
// sgn_splatfacto_scene_graph.py:364-366: torch.cat copies of per-model slices of the main projection): the device-side
// comparison that proves it (sgn_rows_match over whichever tensor pairs are given, at up to 4 candidate offsets), the
// read-back of its verdict — this path's one host sync, in place of the intersection count's read-back of the binning it
// saves —, the window's rows, its own compacted sub-list (sgn_list_window) when asked for, the launch order and the
// forward kernels, on `stream`, temporaries from one arena.  *matched_lo_host = the matching offset, or -1: nothing matched,
// nothing was rasterized, the caller bins the tensors as a scene of their own.
size_t sgn_list_window_workspace_bytes(int n_tiles);

extern "C" __attribute__((visibility("default")))
size_t sgn_rasterize_window_arena_bytes(int n_tiles) {
    return al256(sgn_list_window_workspace_bytes(n_tiles)) + 256 + 256;
}

extern "C" __attribute__((visibility("default")))
int sgn_rasterize_window_all(int n_win, int n_full, int n_cand, const int32_t *cand_lo_host, const float *xys_w,
                             const float *depths_w, const int32_t *radii_w, const int32_t *num_tiles_hit_w,
                             const float *conics_w, const float *colors_w, const float *opacities_w,
                             int opacity_is_logit, const float *xys, const float *depths, const int32_t *radii,
                             const int32_t *num_tiles_hit, const float *conics, const float *opacities,
                             int64_t n_isect, const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                             int ids_qmask, int img_h, int img_w, int block_width, const float *background3,
                             int sub_list, const int32_t *tile_order_ready, float *out_img, float *final_Ts,
                             int32_t *final_idx, int32_t *ids_out, int32_t *tile_bins_out, int32_t *tile_order,
                             int32_t *tile_stats, void *rows, size_t rows_bytes, void *order_scratch,
                             size_t order_scratch_bytes, void *arena, size_t arena_bytes, int32_t *verdict_pinned,
                             int *matched_lo_host, const sgn_raster_opts *opts, sgn_stream_t stream) {
    if (n_win < 1 || n_full < n_win || n_cand < 1 || n_cand > 4 || !cand_lo_host || !xys_w || !conics_w || !colors_w ||
        !opacities_w || !gaussian_ids_sorted || !tile_bins || !background3 || !out_img || !final_Ts || !final_idx ||
        !tile_stats || !rows || !arena || !matched_lo_host || n_isect < 1 || block_width < 2 || block_width > 16 ||
        img_h < 1 || img_w < 1 || (sub_list && (!ids_out || !tile_bins_out)) || (!tile_order_ready && !tile_order)) {
        sgn_set_error("sgn_rasterize_window_all: argument check failed");
        return -1;
    }
    const int tiles_x = (img_w + block_width - 1) / block_width, tiles_y = (img_h + block_width - 1) / block_width;
    const int n_tiles = tiles_x * tiles_y;
    if (arena_bytes < sgn_rasterize_window_arena_bytes(n_tiles)) {
        sgn_set_error("sgn_rasterize_window_all: arena too small");
        return -2;
    }
    sgn_raster_opts o;
    sgn_raster_default_opts(&o);
    if (opts) o = *opts;
    hipStream_t s = (hipStream_t)stream;
    char *p = (char *)arena;
    int32_t *mismatch = (int32_t *)p; p += 256;
    void *lw_ws = p;
    const size_t lw_bytes = sgn_list_window_workspace_bytes(n_tiles);
    *matched_lo_host = -1;
    int lo = -1;
    const bool compare = xys || depths || radii || num_tiles_hit || conics || opacities;
    if (compare) {
        int rc = sgn_rows_match(n_win, n_full, n_cand, cand_lo_host, xys ? xys_w : nullptr, depths ? depths_w : nullptr,
                                radii ? radii_w : nullptr, num_tiles_hit ? num_tiles_hit_w : nullptr,
                                conics ? conics_w : nullptr, opacities ? opacities_w : nullptr, xys, depths, radii,
                                num_tiles_hit, conics, opacities, mismatch, stream);
        if (rc) return rc;
        int32_t pageable[8] = {1, 1, 1, 1, 0, 0, 0, 0};
        int32_t *dst = verdict_pinned ? verdict_pinned : pageable;
        int32_t *vd = POLL_WAITS ? mapped(verdict_pinned) : nullptr;      // EIGHT words: [0, 4) the verdicts, [7] "they have landed" (polled)
        hipError_t e = hipSuccess;
        if (vd) {
            ((volatile int32_t *)dst)[7] = 0;
            rc = sgn_publish_words(mismatch, n_cand, vd, vd + 7, 1, stream);
            if (rc) return rc;
            e = spin_wait([&] { return ((volatile int32_t *)dst)[7] == 1; }, s);
            if (e == hipSuccess && ((volatile int32_t *)dst)[7] != 1)      // the cold path drained the stream: plain copy
                e = hipMemcpy(dst, mismatch, sizeof(int32_t) * n_cand, hipMemcpyDeviceToHost);
        } else {
            e = hipMemcpyAsync(dst, mismatch, sizeof(int32_t) * n_cand, hipMemcpyDeviceToHost, s);
            hipEvent_t ev = nullptr;
            if (e == hipSuccess && sync_event(&ev) != 0) e = hipErrorUnknown;
            if (e == hipSuccess) e = hipEventRecord(ev, s);
            if (e == hipSuccess) e = timed_event_sync(ev);
        }
        if (e != hipSuccess) { sgn_set_error("sgn_rasterize_window_all: verdict read-back: %s", hipGetErrorString(e)); return (int)e; }
        for (int c = 0; c < n_cand && lo < 0; ++c)
            if (((volatile int32_t *)dst)[c] == 0) lo = cand_lo_host[c];
        if (lo < 0) return 0;                   // not a window of that scene: the caller re-bins
    } else {
        lo = cand_lo_host[0];                   // the caller has settled every tensor another way
        if (lo < 0 || lo + n_win > n_full) { sgn_set_error("sgn_rasterize_window_all: offset out of range"); return -4; }
    }
    *matched_lo_host = lo;
    const int id_lo = lo, id_hi = lo + n_win;
    int rc = sgn_raster_build_rows(n_full, xys_w, conics_w, colors_w, opacities_w, opacity_is_logit, id_lo, id_hi, 1,
                                   rows, rows_bytes, nullptr, stream);
    if (rc) return rc;
    const int32_t *ids = gaussian_ids_sorted, *bins = tile_bins;
    const int32_t *order = tile_order_ready;
    if (sub_list) {      // a SMALL window walks its own entries only (same relative order: same image and gradients)
        rc = sgn_list_window(n_tiles, gaussian_ids_sorted, tile_bins, id_lo, id_hi, ids_qmask, ids_out, tile_bins_out,
                             lw_ws, lw_bytes, stream);
        if (rc) return rc;
        ids = ids_out; bins = tile_bins_out;
        order = nullptr;                        // another list: another order
    }
    if (!order) {
        rc = sgn_tile_order(n_tiles, bins, nullptr, (block_width == 16) ? (o.adapt_fwd > 0 ? o.adapt_fwd : 1024) : 0, 0,
                            tile_order, order_scratch, order_scratch_bytes, stream);
        if (rc) return rc;
        order = tile_order;
    }
    sgn_raster_opts oo = o;
    oo.ids_qmask = ids_qmask ? 1 : 0;
    return sgn_raster_fwd(img_h, img_w, block_width, n_full, n_isect, ids, bins, xys_w, conics_w, colors_w, opacities_w,
                          opacity_is_logit, id_lo, id_hi, 1, background3, out_img, final_Ts, final_idx, rows, rows_bytes,
                          1, order, tile_stats, nullptr, nullptr, nullptr, &oo, stream);
}

// ---------------------------------------------------------------- the backward of a rasterize node, one call (round 6)
// sgn_rasterize_bwd_all = sgn_tile_order (the reverse walks' launch order, from the forward's tile statistics) +
// sgn_raster_bwd / sgn_raster_bwd_part.  tile_order [tiles + 2] stays the caller's (its last word is the walked / listed
// statistic the host's quadrant-mask policy reads back later).  first / last as in sgn_raster_bwd_part; first = last = 1 is
// the plain sgn_raster_bwd (window allowed).
extern "C" __attribute__((visibility("default")))
int sgn_rasterize_bwd_all(int img_h, int img_w, int block_width, int n, int64_t n_isect,
                          const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const int32_t *tile_stats,
                          int stats_have_pairs, const float *xys, const float *conics, const float *colors,
                          const float *opacities, int opacity_is_logit, int id_lo, int id_hi, int window,
                          const float *background3, const float *final_Ts, const int32_t *final_idx,
                          const float *v_out_img, const float *v_out_alpha, float alpha_clamp_bwd, float *v_xy,
                          float *v_conic, float *v_colors, float *v_opacity, void *recs_ws, size_t recs_ws_bytes,
                          int recs_packed, void *grad_ws, size_t grad_ws_bytes, int32_t *tile_order,
                          void *order_scratch, size_t order_scratch_bytes, int small_q16,
                          const float *colors_pre_clamp, const sgn_raster_opts *opts, sgn_stream_t stream,
                          sgn_stream_t aux_stream, int first, int last) {
    sgn_raster_opts o;
    sgn_raster_default_opts(&o);
    if (opts) o = *opts;
    const int tiles_x = (img_w + block_width - 1) / block_width, tiles_y = (img_h + block_width - 1) / block_width;
    const int32_t *order = nullptr;
    if (tile_order != nullptr && n_isect > 0 && n > 0) {
        if (!tile_bins || !tile_stats) { sgn_set_error("sgn_rasterize_bwd_all: tile_order needs tile_bins and tile_stats"); return -1; }
        int rc = sgn_tile_order(tiles_x * tiles_y, tile_bins, tile_stats, o.adapt_bwd > 0 ? o.adapt_bwd : 256,
                                stats_have_pairs ? small_q16 : 0, tile_order, order_scratch, order_scratch_bytes, stream);
        if (rc) return rc;
        order = tile_order;
    }
    if (first && last)
        return sgn_raster_bwd(img_h, img_w, block_width, n, n_isect, gaussian_ids_sorted, tile_bins, xys, conics, colors,
                              opacities, opacity_is_logit, id_lo, id_hi, window, background3, final_Ts, final_idx,
                              v_out_img, v_out_alpha, alpha_clamp_bwd, v_xy, v_conic, v_colors, v_opacity, recs_ws,
                              recs_ws_bytes, recs_packed, grad_ws, grad_ws_bytes, order, colors_pre_clamp, opts, stream,
                              aux_stream);
    return sgn_raster_bwd_part(img_h, img_w, block_width, n, n_isect, gaussian_ids_sorted, tile_bins, xys, conics, colors,
                               opacities, opacity_is_logit, id_lo, id_hi, window, background3, final_Ts, final_idx,
                               v_out_img, v_out_alpha, alpha_clamp_bwd, v_xy, v_conic, v_colors, v_opacity, recs_ws,
                               recs_ws_bytes, recs_packed, grad_ws, grad_ws_bytes, order, colors_pre_clamp, opts, stream,
                               aux_stream, first, last);
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

// host microseconds the calling thread has spent blocked in the event waits of the one-call entries (sgn_project_fwd_all /
// sgn_project_check_wait, sgn_rasterize_fwd_all, sgn_rasterize_window_all) since the last reset, and how many waits
extern "C" __attribute__((visibility("default"))) double sgn_timing_host_wait_us(int reset, int64_t *n_waits) {
    const double t = g_wait_us;
    if (n_waits) *n_waits = g_wait_n;
    if (reset) { g_wait_us = 0.0; g_wait_n = 0; }
    return t;
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
