// binning.hip — tile binning around the sort: int32 inclusive scan, intersection emission,
// tile bin edges.  gfx950.
//
// Replaces torch.cumsum in gsplat/utils.py:compute_cumulative_intersects and gsplat 0.1.x
// forward.cu:map_gaussian_to_intersects / get_tile_bin_edges, reached from the
// reference through rasterize_gaussians at sgn_splatfacto.py:954-967, :982-994.
// Integer work: results are bit-exact with the oracle by construction.
#include "sgn_common.h"

#include <algorithm>

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_ITEMS;  // 2048 items per workgroup

__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(v, d, 64);
        if (lane >= d) v += u;
    }
    return v;
}

// inclusive scan across a 256-thread block of one value per thread; returns inclusive prefix and
// the block total through `total`.
__device__ __forceinline__ int block_incl_scan(int v, int *lds4, int &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = wave_incl_scan(v);
    if (lane == 63) lds4[wave] = w;
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = lds4[k];
        if (k < wave) off += t;
    }
    total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    __syncthreads();
    return w + off;
}

// `idx` (optional, with `gathered`): the first pass reads in[idx[i]] and leaves the gathered values in `gathered`,
// which the final pass then reads linearly — the fused binning scans the kept-tile counts in depth-rank order without
// a gather launch of its own (gathering in BOTH passes was measured: 10.3 us each against 4.6 / 5.6 linear)
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(int n, const int32_t *__restrict__ in,
                                                                   const int32_t *__restrict__ idx,
                                                                   int32_t *__restrict__ gathered,
                                                                   int32_t *__restrict__ partial) {
    __shared__ int lds4[4];
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_ITEMS;
    int j[SCAN_ITEMS], v[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) j[k] = (base + k < n) ? (idx ? idx[base + k] : base + k) : -1;
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (j[k] >= 0) ? in[j[k]] : 0;
        s += v[k];
    }
    if (gathered) {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k)
            if (j[k] >= 0) gathered[base + k] = v[k];
    }
    int total;
    block_incl_scan(s, lds4, total);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// single workgroup: in-place exclusive scan of the per-chunk totals; SCAN_ITEMS consecutive totals per thread, all
// requested before the first use (up to 2048 chunks = 4 M items in one round trip and one block scan)
__global__ __launch_bounds__(SCAN_THREADS) void scan_partials_kernel(int nb, int32_t *__restrict__ partial) {
    __shared__ int lds4[4];
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += SCAN_CHUNK) {
        const int i0 = b0 + threadIdx.x * SCAN_ITEMS;
        int v[SCAN_ITEMS], sum = 0;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) v[k] = (i0 + k < nb) ? partial[i0 + k] : 0;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) sum += v[k];
        int total;
        int run = carry + block_incl_scan(sum, lds4, total) - sum;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            if (i0 + k < nb) partial[i0 + k] = run;
            run += v[k];
        }
        carry += total;
    }
}

// SCANNED = true: `partial` holds the exclusive scan of the chunk totals (scan_partials_kernel ran).  SCANNED = false
// (r03: one launch less per binning): `partial` holds the raw chunk totals and every workgroup sums the ones before
// its own itself — at most a few hundred 4-byte values per workgroup (489 chunks at 1 M items), all requested with
// the workgroup's own items, against a launch of its own (~5 us + the dependency bubble) for one workgroup's scan.
template <bool SCANNED>
__global__ __launch_bounds__(SCAN_THREADS) void scan_final_kernel(int n, const int32_t *__restrict__ in,
                                                                  const int32_t *__restrict__ partial,
                                                                  int32_t *__restrict__ out,
                                                                  int32_t *__restrict__ total_host,
                                                                  const int32_t *__restrict__ extra_dev = nullptr,
                                                                  int32_t *__restrict__ extra_host = nullptr) {
    __shared__ int lds4[4];
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
    int before = 0;
    if constexpr (!SCANNED) {
        for (int j = threadIdx.x; j < (int)blockIdx.x; j += SCAN_THREADS) before += partial[j];
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    int total;
    int chunk_base;
    if constexpr (SCANNED) {
        chunk_base = partial[blockIdx.x];
    } else {
        block_incl_scan(before, lds4, chunk_base);          // total over the block = sum of the earlier chunks
    }
    const int inc = block_incl_scan(s, lds4, total);
    int run = chunk_base + inc - s;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        run += v[k];
        if (base + k < n) out[base + k] = run;
        // the grand total straight into (mapped, pinned) HOST memory: the caller's read-back of out[n-1] then needs no
        // copy command behind this kernel (a ~4 us blit and a ~6 us bubble on the stream) and — round 6 — no event
        // either: the host polls the word.  `extra`: one more device word the caller wants on the host with the count
        // (stored FIRST: posted writes keep their order, so a host that sees the count sees the extra word)
        if (total_host != nullptr && base + k == n - 1) {
            if (extra_host != nullptr) {
                __hip_atomic_store(extra_host, *extra_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __threadfence_system();
            }
            __hip_atomic_store(total_host, run, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
        }
    }
}

// One lane per Gaussian; bboxes larger than BIG tiles are emitted by the whole wave so one huge
// splat does not serialise 63 idle lanes.
constexpr int MAP_BIG = 32;

__global__ __launch_bounds__(256) void map_isect_kernel(int n, const float *__restrict__ xys,
                                                        const float *__restrict__ depths,
                                                        const int32_t *__restrict__ radii,
                                                        const int32_t *__restrict__ cum, int tiles_x,
                                                        int tiles_y, int block,
                                                        int64_t *__restrict__ keys,
                                                        int32_t *__restrict__ vals, int sem) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int mnx = 0, mny = 0, mxx = 0, mxy = 0, cur = 0;
    int64_t depth_id = 0;
    bool live = false;
    if (i < n) {
        const int r = radii[i];
        if (r > 0) {
            live = true;
            sgn_tile_bbox(xys[2 * i], xys[2 * i + 1], (float)r, tiles_x, tiles_y, block, mnx, mny, mxx, mxy, sem);
            cur = (i == 0) ? 0 : cum[i - 1];
            depth_id = (int64_t)__float_as_int(depths[i]);  // sign-extends like upstream; depth > 0
        }
    }
    const int w = mxx - mnx, h = mxy - mny;
    const int area = live ? w * h : 0;
    if (area > 0 && area <= MAP_BIG) {
        for (int ty = mny; ty < mxy; ++ty)
            for (int tx = mnx; tx < mxx; ++tx) {
                keys[cur] = ((int64_t)(ty * tiles_x + tx) << 32) | depth_id;
                vals[cur] = i;
                ++cur;
            }
    }
    unsigned long long big = __ballot(area > MAP_BIG);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const int bw = __shfl(w, src, 64), bmnx = __shfl(mnx, src, 64), bmny = __shfl(mny, src, 64);
        const int barea = __shfl(area, src, 64), bcur = __shfl(cur, src, 64), bi = __shfl(i, src, 64);
        const int dlo = __shfl((int)(depth_id & 0xffffffffll), src, 64);
        const int64_t bdepth = (int64_t)dlo;
        for (int t = lane; t < barea; t += 64) {
            const int ty = bmny + t / bw, tx = bmnx + t % bw;
            keys[bcur + t] = ((int64_t)(ty * tiles_x + tx) << 32) | bdepth;
            vals[bcur + t] = bi;
        }
    }
}

// (tile ids outside [0, n_tiles) — keys that are not this image's — are ignored, never written through)
__global__ __launch_bounds__(256) void tile_bins_kernel(int64_t n_isect, const int64_t *__restrict__ keys,
                                                        int32_t *__restrict__ bins, int n_tiles) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_isect) return;
    const int32_t cur = (int32_t)(keys[idx] >> 32);
    const bool cur_ok = cur >= 0 && cur < n_tiles;
    if (idx == 0 && cur_ok) bins[2 * cur] = 0;
    if (idx == n_isect - 1 && cur_ok) bins[2 * cur + 1] = (int32_t)n_isect;
    if (idx == 0) return;
    const int32_t prev = (int32_t)(keys[idx - 1] >> 32);
    if (prev != cur) {
        if (prev >= 0 && prev < n_tiles) bins[2 * prev + 1] = (int32_t)idx;
        if (cur_ok) bins[2 * cur] = (int32_t)idx;
    }
}

// ------------------------------------------------------------------ fused (rank-order) binning path
// Upstream's order — stable sort by (tile, depth bits), ties in emission (= Gaussian id) order — is an
// LSD sort whose low digits can be had for free: rank the N Gaussians by depth once (stable, ties by
// id), EMIT the intersections in rank order, and the list only needs a stable sort by the tile id
// (14 bits at 1920x1280/16: two 8-bit passes over (u32 tile, i32 gaussian id) pairs, 40 B/intersection
// instead of six passes x 32 B on 64-bit (tile|depth, id) pairs).  gaussian_ids_sorted / tile_bins come
// out bit-identical to the upstream-shaped path (tests/test_gpu_parity.py, tests/test_gpu_e2e.py).
// Exact tile culling (fused path only).  Upstream bins a Gaussian into every tile of the square that bounds
// its 3-sigma CIRCLE; a tile can only receive colour from it if some pixel centre has
// alpha = min(0.999, o * exp(-sigma)) >= 1/255, i.e. sigma <= s := ln(255 o).  The ellipse {sigma <= s} is
// convex, so in every tile ROW the tiles it can touch form one interval: for the row's pixel-centre band
// Y in [y0, y1] the ellipse's x-extent is X_right(Y) = (-bY + sqrt(2as - DY^2))/a maximised (concave) and
// X_left(Y) minimised (convex) over the band, each at its clamped unconstrained optimum; a tile is kept iff
// its pixel-centre span [x0, x0+B-1] meets [x_min, x_max].  O(rows) per Gaussian instead of O(tiles).
// Dropped (tile, Gaussian) pairs have no valid pixel in forward or backward, so images and gradients are
// unchanged (tests), while the intersection count roughly halves on the benchmark scene (17.2 M -> 8.3 M).
struct Cull {
    const float *conics;  // [n,3]
    const float *opac;    // [n]
    int opac_is_logit;
    int enable;
};

struct Ellipse {       // per-Gaussian constants of the row-interval test
    float gx, gy;      // centre (pixels)
    float a, b;        // conic a, b
    float inv_a;       // 1/a
    float two_as;      // 2 a s
    float D;           // a c - b^2
    float y_ext;       // sqrt(2 a s / D): |Y| extent of the ellipse
    float y_at_xmax;   // Y where X is maximal  (= -(b/c) X_max); X minimal at -y_at_xmax
    int valid;         // 0: nothing can be hit; 1: use the test; 2: degenerate conic -> keep everything
};

// 32-byte per-Gaussian bin record, written in id order by bin_count_kernel and gathered ONCE (one sector) by the
// rank-order emission instead of five separate random gathers (xys, radii, conics x3, opacity).
struct BinRec {
    float gx, gy;      // projected centre
    float a, b, c;     // conic
    float s;           // ln(255 o) + margin; < 0: never visible; +inf: culling off (keep every bbox tile)
    int rad;           // radius in pixels (0: culled by projection), at most BINREC_RAD_MAX; bit 30: the tile boxes of
                       // this binning follow SGN_SEM_BBOX_ADD_AFTER_CAST (the emission recomputes the box from the record)
    int cnt;           // kept tiles
};
constexpr int BINREC_RAD_MAX = (1 << 30) - 1;   // (a larger radius covers every tile of any image anyway)
static_assert(sizeof(BinRec) == SGN_BIN_RECORD_FLOATS * 4, "bin record size");

// The culling test is conservative by construction (margins: 0.01 in sigma, 1e-3 px in x), so it runs on the
// 1-ulp hardware approximations (v_rcp_f32, v_sqrt_f32, v_log_f32, v_exp_f32) instead of the ~10-20-instruction
// correctly-rounded sequences; count and emission share the code, so they always agree with each other.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

__device__ __forceinline__ float cull_threshold(const Cull &cu, int gid) {
    if (!cu.enable) return __builtin_inff();
    float o = cu.opac[gid];
    if (cu.opac_is_logit) o = fast_rcp(1.f + __expf(-o));
    // margin 0.01 in sigma (1 % in alpha) >> fp32 error of any evaluation order of the quadratic form and of the
    // hardware log / exp / rcp / sqrt (1-2 ulp) used throughout this test
    return (o * 255.f > 0.f) ? __logf(255.f * o) + 0.01f : -1.f;
}

__device__ __forceinline__ Ellipse make_ellipse(float gx, float gy, float a, float b, float c, float s) {
    Ellipse E;
    E.gx = gx; E.gy = gy;
    const float D = a * c - b * b;
    const float inv_D = fast_rcp(D);
    E.a = a; E.b = b; E.inv_a = fast_rcp(a); E.two_as = 2.f * a * s; E.D = D;
    E.valid = !(s >= 0.f) ? 0 : ((s < 3.0e38f && a > 0.f && c > 0.f && D > 0.f) ? 1 : 2);
    E.y_ext = fast_sqrt(fmaxf(E.two_as * inv_D, 0.f)) * 1.000001f;
    const float x_max = fast_sqrt(fmaxf(2.f * c * s * inv_D, 0.f));
    E.y_at_xmax = -(b * fast_rcp(c)) * x_max;
    return E;
}

// kept tile interval [lo, hi] (inclusive, may be empty: hi < lo) of tile row `ty`, clamped to [mnx, mxx)
__device__ __forceinline__ void row_interval(const Ellipse &E, int ty, int block, int mnx, int mxx, int &lo,
                                             int &hi) {
    lo = mnx; hi = mxx - 1;
    if (E.valid == 2) return;
    if (E.valid == 0) { hi = lo - 1; return; }
    const float y0 = (float)(ty * block) + 0.5f - E.gy, y1 = y0 + (float)(block - 1);
    const float ya = fmaxf(y0, -E.y_ext), yb = fminf(y1, E.y_ext);   // band clipped to the ellipse's Y extent
    if (ya > yb) { hi = lo - 1; return; }
    const float yr = fminf(fmaxf(E.y_at_xmax, ya), yb);               // maximiser of X_right on the band
    const float yl = fminf(fmaxf(-E.y_at_xmax, ya), yb);              // minimiser of X_left on the band
    const float x_max = (-E.b * yr + fast_sqrt(fmaxf(E.two_as - E.D * yr * yr, 0.f))) * E.inv_a + 1e-3f;
    const float x_min = (-E.b * yl - fast_sqrt(fmaxf(E.two_as - E.D * yl * yl, 0.f))) * E.inv_a - 1e-3f;
    // tile tx spans pixel centres [tx*B + 0.5, tx*B + B - 0.5]; keep it iff that span meets [x_min, x_max] + gx
    const float fb = (float)block, inv_fb = fast_rcp(fb);
    const int t_hi = sgn_f2i(floorf((x_max + E.gx - 0.5f) * inv_fb));
    const int t_lo = sgn_f2i(ceilf((x_min + E.gx + 0.5f - fb) * inv_fb));
    lo = max(lo, t_lo);
    hi = min(hi, t_hi);
}

// Quadrant masks (r03).  The raster kernels skip the 8x8 quadrants of a tile an entry cannot touch; they used to decide
// that per entry from the axis-aligned box of the ellipse (two subtractions, two compares, a ballot — and 10 % of the
// quadrants it lets through have no valid pixel, because the ellipse is not its box).  The emission already holds the
// exact convex test, so it evaluates it once more per tile row for the two 8-pixel half bands and hands each
// (tile, Gaussian) pair its four bits in the top of the id word (ids < 2^28): bit q = quadrant (x half = q & 1,
// y half = q >> 1), the numbering of raster.hip's slot_pixel.  Same margins as the tile test; a pair whose ellipse
// slips between the pixel centres of every quadrant gets 0 and is skipped by the kernels.
// x-extent [x_min, x_max] (absolute pixels) of the ellipse over the band Y in [y0, y1] (relative to its centre)
__device__ __forceinline__ bool band_extent(const Ellipse &E, float y0, float y1, float &x_min, float &x_max) {
    const float ya = fmaxf(y0, -E.y_ext), yb = fminf(y1, E.y_ext);
    if (ya > yb) return false;
    const float yr = fminf(fmaxf(E.y_at_xmax, ya), yb);
    const float yl = fminf(fmaxf(-E.y_at_xmax, ya), yb);
    x_max = (-E.b * yr + fast_sqrt(fmaxf(E.two_as - E.D * yr * yr, 0.f))) * E.inv_a + 1e-3f + E.gx;
    x_min = (-E.b * yl - fast_sqrt(fmaxf(E.two_as - E.D * yl * yl, 0.f))) * E.inv_a - 1e-3f + E.gx;
    return true;
}

// Output sinks of the tile expansion below.
struct NoOut {
    __device__ __forceinline__ void operator()(int, uint32_t, int32_t) const {}
};
// TK = tile-key type: uint16_t whenever the image has at most 65536 tiles (1920x1280/16 = 9600), which takes a
// quarter of the bytes off the emission and off every pass of the tile sort; uint32_t otherwise.
template <typename TK>
struct GlobalOut {
    TK *__restrict__ tkeys;
    int32_t *__restrict__ tvals;
    __device__ __forceinline__ void operator()(int pos, uint32_t key, int32_t val) const {
        tkeys[pos] = (TK)key;
        tvals[pos] = val;
    }
};
template <typename TK>
struct BoundedOut {       // a speculative launch whose buffers turn out too small: never write past `cap` slots
    TK *__restrict__ tkeys;
    int32_t *__restrict__ tvals;
    int cap;
    __device__ __forceinline__ void operator()(int pos, uint32_t key, int32_t val) const {
        if (pos < cap) {
            tkeys[pos] = (TK)key;
            tvals[pos] = val;
        }
    }
};
template <typename TK>
struct LdsOut {          // wave-local staging: positions relative to the wave's first output slot
    TK *keys;
    int32_t *vals;
    int base;
    __device__ __forceinline__ void operator()(int pos, uint32_t key, int32_t val) const {
        keys[pos - base] = (TK)key;
        vals[pos - base] = val;
    }
};

// wave64 inclusive scan on the DPP network (no LDS crossbar): Hillis-Steele inside each 16-lane row, then lane 15
// of rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3.  All 64 lanes must be active.
__device__ __forceinline__ int wave_incl_scan_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// LDS traffic between lanes of ONE wave: the hardware retires a wave's LDS operations in order, the fence only
// keeps the compiler from moving them across
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Per-wave scratch of the flattened row list: the Gaussians' row_interval constants, looked up by the lanes that
// own their tile rows.
struct FlatScratch {
    float4 pa[64];    // gx, gy, b, 1/a
    float4 pb[64];    // 2as, D, y_ext, y_at_xmax
    int4 pc[64];      // mnx, mxx, mny, valid
    int gid[64];      // gaussian id (emission)
    int rowend[64];   // inclusive prefix of the row counts (a Gaussian without rows repeats its predecessor)
    int acc[64];      // kept tiles per Gaussian (count)
};

// Kept tiles of the 64 Gaussians of a wave, one Gaussian per lane on entry: row-major over each bbox, one kept
// interval per tile row.  EMIT = false: returns the lane's number of kept tiles; EMIT = true: writes the wave's
// (tile, gaussian id) pairs from output slot `base` on (the Gaussians' runs follow each other in lane order, so
// slots come from a running scan and agree with the counts by construction: same code).
// Heights vary widely inside a wave (1-3 rows typically, 7-20 for one Gaussian in ten, the whole image for a few),
// so the wave does not loop per lane: all tile rows are laid end to end into one list and handed out 64 at a time,
// lane <-> (Gaussian, row) by binary search in the prefix of the row counts; every lane evaluates one row_interval
// per round whatever the mix.  (Round 1 looped over rows per lane and took the tall boxes one at a time with the
// whole wave: 80.7 / 39.4 us emission / count on the benchmark scene against 41.0 / 25.4 us for this form,
// profiles/experiments/r01_tall_quad_notes.md, profiles/r02k_*.)
// Must be called by all 64 lanes of the wave.
template <bool EMIT, class Out>
__device__ __forceinline__ int tiles_of(bool live, const Ellipse &E, int mnx, int mny, int mxx, int mxy, int gid,
                                        int base, int tiles_x, int block, FlatScratch &fs, const Out &out,
                                        int qmask = 0) {
    const int lane = threadIdx.x & 63;
    const int h = (live && mxx > mnx) ? mxy - mny : 0;
    const int rend = wave_incl_scan_dpp(h);
    const int n_rows = __builtin_amdgcn_readlane(rend, 63);
    if (n_rows == 0) return 0;
    fs.pa[lane] = make_float4(E.gx, E.gy, E.b, E.inv_a);
    fs.pb[lane] = make_float4(E.two_as, E.D, E.y_ext, E.y_at_xmax);
    fs.pc[lane] = make_int4(mnx, mxx, mny, E.valid);
    fs.rowend[lane] = rend;
    if (EMIT) fs.gid[lane] = gid; else fs.acc[lane] = 0;
    wave_lds_fence();
    int pos0 = base;                                         // output slot of this round's first tile
    for (int r0 = 0; r0 < n_rows; r0 += 64) {                // wave-uniform trip count (n_rows is an SGPR)
        const int item = r0 + lane;
        const bool valid = item < n_rows;
        int g = 0;                                           // first Gaussian whose rowend exceeds item
#pragma unroll
        for (int step = 32; step; step >>= 1)
            if (fs.rowend[g + step - 1] <= item) g += step;
        const int first = g ? fs.rowend[g - 1] : 0;          // list index of that Gaussian's first row
        const float4 pa = fs.pa[g], pb = fs.pb[g];
        const int4 pc = fs.pc[g];
        Ellipse B;
        B.gx = pa.x; B.gy = pa.y; B.a = 0.f; B.b = pa.z; B.inv_a = pa.w;
        B.two_as = pb.x; B.D = pb.y; B.y_ext = pb.z; B.y_at_xmax = pb.w; B.valid = pc.w;
        const int ty = pc.z + item - first;
        int lo = 0, hi = -1;
        if (valid) row_interval(B, ty, block, pc.x, pc.y, lo, hi);
        const int c = max(hi - lo + 1, 0);
        if (EMIT) {
            const int incl = wave_incl_scan_dpp(c);
            const int val = fs.gid[g];
            int pos = pos0 + incl - c;
            if (qmask) {                                     // wave-uniform; 16x16 tiles only (the host checks)
                // the two half bands' x-extents as inclusive ranges of HALF COLUMNS (h = 2 tx + x half; pixel centres
                // 8h + 0.5 .. 8h + 7.5): h is touched iff x_max >= 8h + 0.5 and x_min <= 8h + 7.5
                int ht_lo = 1, ht_hi = 0, hb_lo = 1, hb_hi = 0;     // empty
                if (c > 0 && B.valid == 1) {
                    const float yt = (float)(ty * 16) + 0.5f - B.gy;
                    float x0, x1;
                    if (band_extent(B, yt, yt + 7.f, x0, x1)) {
                        ht_lo = sgn_f2i(ceilf(fmaxf(x0 - 7.5f, -1.0e6f) * 0.125f));
                        ht_hi = sgn_f2i(floorf(fminf(x1 - 0.5f, 1.0e6f) * 0.125f));
                    }
                    if (band_extent(B, yt + 8.f, yt + 15.f, x0, x1)) {
                        hb_lo = sgn_f2i(ceilf(fmaxf(x0 - 7.5f, -1.0e6f) * 0.125f));
                        hb_hi = sgn_f2i(floorf(fminf(x1 - 0.5f, 1.0e6f) * 0.125f));
                    }
                }
                if (B.valid == 2) { ht_lo = hb_lo = -(1 << 29); ht_hi = hb_hi = 1 << 29; }   // degenerate conic: all
                // bit i of a run word: half column `base + i`; 32 half columns = 16 tiles per word
                auto run = [](int h_lo, int h_hi, int base) __attribute__((always_inline)) -> unsigned {
                    const int a = max(h_lo - base, 0), b = min(h_hi - base, 31);
                    return a > b ? 0u : (((2u << (b - a)) - 1u) << a);
                };
                unsigned run_t = run(ht_lo, ht_hi, 2 * lo), run_b = run(hb_lo, hb_hi, 2 * lo);
                int i = 0;
                for (int tx = lo; tx <= hi; ++tx, ++pos, i += 2) {
                    if (i == 32) {                           // a row wider than 16 tiles: next word
                        run_t = run(ht_lo, ht_hi, 2 * tx); run_b = run(hb_lo, hb_hi, 2 * tx);
                        i = 0;
                    }
                    const unsigned m = ((run_t >> i) & 3u) | (((run_b >> i) & 3u) << 2);
                    out(pos, (uint32_t)(ty * tiles_x + tx), (int32_t)((uint32_t)val | (m << 28)));
                }
            } else {
                for (int tx = lo; tx <= hi; ++tx, ++pos) out(pos, (uint32_t)(ty * tiles_x + tx), val);
            }
            pos0 += __builtin_amdgcn_readlane(incl, 63);
        } else if (c > 0) {
            atomicAdd(&fs.acc[g], c);
        }
    }
    if (EMIT) return 0;
    wave_lds_fence();
    return h > 0 ? fs.acc[lane] : 0;
}

// lane = Gaussian id (coalesced reads): depth sort key, bin record and its kept-tile count in one pass
__global__ __launch_bounds__(256) void bin_count_kernel(int n, const float *__restrict__ xys,
                                                        const float *__restrict__ depths,
                                                        const int32_t *__restrict__ radii, Cull cull, int tiles_x,
                                                        int tiles_y, int block, uint32_t *__restrict__ dkeys,
                                                        int32_t *__restrict__ dvals, BinRec *__restrict__ recs,
                                                        int32_t *__restrict__ cnt_gid, int write_keys, int sem) {
    __shared__ FlatScratch scratch[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    int mnx = 0, mny = 0, mxx = 0, mxy = 0;
    BinRec R;
    R.gx = 0.f; R.gy = 0.f; R.a = 1.f; R.b = 0.f; R.c = 1.f; R.s = -1.f; R.rad = 0; R.cnt = 0;
    bool live = false;
    if (i < n) {
        R.rad = min(radii[i], BINREC_RAD_MAX);
        if (R.rad > 0) {
            live = true;
            R.gx = xys[2 * i]; R.gy = xys[2 * i + 1];
            sgn_tile_bbox(R.gx, R.gy, (float)R.rad, tiles_x, tiles_y, block, mnx, mny, mxx, mxy, sem);
            R.s = cull_threshold(cull, i);
            if (cull.enable) { R.a = cull.conics[3 * i]; R.b = cull.conics[3 * i + 1]; R.c = cull.conics[3 * i + 2]; }
        }
    }
    const Ellipse E = make_ellipse(R.gx, R.gy, R.a, R.b, R.c, R.s);
    R.cnt = tiles_of<false>(live, E, mnx, mny, mxx, mxy, i, 0, tiles_x, block, scratch[threadIdx.x >> 6], NoOut{});
    if (i < n) {
        if (write_keys) {           // 0: the depth ranking was started earlier (sgn_depth_rank)
            dkeys[i] = R.rad > 0 ? (uint32_t)__float_as_int(depths[i]) : 0xFFFFFFFFu;  // culled splats sort last
            dvals[i] = i;
        }
        cnt_gid[i] = R.cnt;             // dense copy: the rank-order gather below then works on 4 B/Gaussian
        float4 *o = reinterpret_cast<float4 *>(recs + i);
        o[0] = make_float4(R.gx, R.gy, R.a, R.b);
        const int radw = R.rad > 0 ? (R.rad | ((sem & SGN_SEM_BBOX_ADD_AFTER_CAST) ? (1 << 30) : 0)) : R.rad;
        o[1] = make_float4(R.c, R.s, __int_as_float(radw), __int_as_float(R.cnt));
    }
}

// lane = depth rank r: writes the (tile, gaussian id) pairs of Gaussian gid_by_rank[r] starting at cum_r[r-1].
// One wave per workgroup.  The 64 Gaussians of a wave own one contiguous output range; when it fits EMIT_CAP
// entries the pairs are staged in LDS and written out with full-width coalesced stores (per-lane sequential
// dword stores cost one memory request each: 16.6 M requests per view on the benchmark scene).
constexpr int EMIT_CAP = 1024;
template <typename TK>
__global__ __launch_bounds__(64) void bin_emit_kernel(int n, const int32_t *__restrict__ gid_by_rank,
                                                      const int32_t *__restrict__ cum_r,
                                                      const BinRec *__restrict__ recs, int tiles_x, int tiles_y,
                                                      int block, TK *__restrict__ tkeys,
                                                      int32_t *__restrict__ tvals, int cap,
                                                      int32_t *__restrict__ zero_buf, int zero_n, int qmask) {
    __shared__ TK lk[EMIT_CAP];
    __shared__ int32_t lv[EMIT_CAP];
    __shared__ FlatScratch scratch;
    const int lane = threadIdx.x;
    // tile_bins must read 0 for tiles without entries and tile_bins32_kernel (two launches later on this stream)
    // only writes the tiles that have some: cleared here instead of by a memset launch of its own (~5 us each)
    for (int j = blockIdx.x * 64 + lane; j < zero_n; j += gridDim.x * 64) zero_buf[j] = 0;
    const int r0 = blockIdx.x * 64, r = r0 + lane;
    int mnx = 0, mny = 0, mxx = 0, mxy = 0, gid = 0;
    float gx = 0.f, gy = 0.f, a = 1.f, b = 0.f, c = 1.f, s = -1.f;
    bool live = false;
    const int base = (r0 == 0) ? 0 : cum_r[r0 - 1];
    const int total = cum_r[min(r0 + 63, n - 1)] - base;       // wave-uniform
    if (total == 0) return;
    if (r < n) {
        gid = gid_by_rank[r];
        const float4 *q = reinterpret_cast<const float4 *>(recs + gid);
        const float4 q0 = q[0], q1 = q[1];
        const int radw = __float_as_int(q1.z);
        const int rad = radw > 0 ? (radw & BINREC_RAD_MAX) : radw;
        if (rad > 0 && __float_as_int(q1.w) > 0) {
            live = true;
            gx = q0.x; gy = q0.y; a = q0.z; b = q0.w; c = q1.x; s = q1.y;
            sgn_tile_bbox(gx, gy, (float)rad, tiles_x, tiles_y, block, mnx, mny, mxx, mxy,
                          (radw >> 30) & 1 ? SGN_SEM_BBOX_ADD_AFTER_CAST : 0);
        }
    }
    const Ellipse E = make_ellipse(gx, gy, a, b, c, s);
    if (base + total > cap) {                                 // wave-uniform; only a mis-sized speculative launch
        tiles_of<true>(live, E, mnx, mny, mxx, mxy, gid, base, tiles_x, block, scratch, BoundedOut<TK>{tkeys, tvals, cap}, qmask);
    } else if (total <= EMIT_CAP) {
        tiles_of<true>(live, E, mnx, mny, mxx, mxy, gid, base, tiles_x, block, scratch, LdsOut<TK>{lk, lv, base}, qmask);
        __syncthreads();                                      // single-wave workgroup: a fence, no s_barrier
        for (int j = lane; j < total; j += 64) {
            tkeys[base + j] = lk[j];
            tvals[base + j] = lv[j];
        }
    } else {
        tiles_of<true>(live, E, mnx, mny, mxx, mxy, gid, base, tiles_x, block, scratch, GlobalOut<TK>{tkeys, tvals}, qmask);
    }
}

// ---- launch-order helpers (tile_order_kernel further down)
__device__ __forceinline__ int len_bucket(int len) {
    if (len <= 0) return 0;
    const int e = 31 - __clz(len);
    const int half = e > 0 ? (len >> (e - 1)) & 1 : 0;
    return min(63, 1 + 2 * e + half);
}
// length of tile t for the ordering: its depth-list length, or (forward statistics given) its reverse-walk length.
// A tile of SMALL splats - fewer than small_q16 / 16 evaluated (entry, quadrant) pairs per walked entry, i.e. most
// Gaussians touch a single 8x8 quadrant - is promoted to the long class whatever its length: four waves per tile then
// cost no extra gradient reductions and quadruple the parallelism (street scene: 0.56 vs 0.68 ms).
__device__ __forceinline__ int order_len(int t, const int2 *__restrict__ bins, const int32_t *__restrict__ stats,
                                         int long_thresh, int small_q16) {
    const int2 r = bins[t];
    const int len = r.y - r.x;
    if (stats == nullptr || len <= 0) return len;
    const int walk = min(len, max(0, stats[2 * t] - r.x + 1));
    if (small_q16 > 0 && long_thresh > 0 && walk >= 32 && walk < long_thresh && stats[2 * t + 1] * 16 < walk * small_q16)
        return long_thresh;
    return walk;
}
// wave-aggregated LDS counter: lanes with the same bucket are served by ONE atomic (all tiles of a uniform scene fall
// in one length class: 64 lanes hammering one LDS address serialise, measured 65 us per pass before this)
__device__ __forceinline__ int bucket_slot(int *counters, int bucket, bool active) {
    int slot = 0;
    unsigned long long todo = __ballot(active);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const int b = __shfl(bucket, src, 64);
        const unsigned long long same = __ballot(active && bucket == b);
        int base = 0;
        if (lane == src) base = atomicAdd(&counters[b], __popcll(same));
        base = __shfl(base, src, 64);
        if (active && bucket == b) slot = base + __popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return slot;
}

// Counting sort of the tiles into half-octave length classes, longest class first, WITHOUT LDS atomics: every thread
// holds the classes of its ROUNDS tiles (tile = round * blockDim + thread) in registers; per class present in the wave
// one ballot per round counts / ranks its tiles (popcount, v_mbcnt), the waves' counts meet in LDS once.  (The r02 form
// did one wave-aggregated LDS atomic per round and class: ~80 dependent LDS round trips per wave in a 256-thread
// workgroup; this form is ALU-only inside the rounds: 15 -> ~5 us per launch for 9600 tiles, two launches per step.)  order[0..n_tiles) = the permutation,
// order[n_tiles] = number of tiles whose class reaches long_thresh's, order[n_tiles + 1] = 0 (the backward's order
// puts its walk statistic there afterwards, tile_order_kernel).  Tiles beyond
// ROUNDS * blockDim are not covered: callers fall back for larger grids.
template <int ROUNDS>
__device__ __forceinline__ void order_by_class(const signed char (&bucket)[ROUNDS], int n_tiles, int long_thresh,
                                               int32_t *__restrict__ order, int (*wave_cnt)[64], int *start) {
    const int nthr = blockDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = nthr >> 6;
    unsigned long long present = 0ull;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i)
        if (bucket[i] >= 0) present |= 1ull << bucket[i];
    uint32_t plo = (uint32_t)present, phi = (uint32_t)(present >> 32);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        plo |= (uint32_t)__shfl_xor((int)plo, d, 64);
        phi |= (uint32_t)__shfl_xor((int)phi, d, 64);
    }
    present = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)phi) << 32) |
              (uint32_t)__builtin_amdgcn_readfirstlane((int)plo);
    wave_cnt[wave][lane] = 0;
    for (unsigned long long todo = present; todo; todo &= todo - 1) {        // wave-uniform
        const int b = __ffsll((long long)todo) - 1;
        int c = 0;
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i)
            if (i * nthr < n_tiles) c += __popcll(__ballot(bucket[i] == b));
        if (lane == 0) wave_cnt[wave][b] = c;
    }
    __syncthreads();
    if (wave == 0) {           // start[b] = tiles in classes above b (longest class first)
        int tot = 0;
        for (int w = 0; w < nw; ++w) tot += wave_cnt[w][lane];
        int above = tot;       // inclusive suffix sum over lanes >= mine, then exclusive
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int u = __shfl_down(above, d, 64);
            if (lane + d < 64) above += u;
        }
        start[lane] = above - tot;
        const int b_long = long_thresh > 0 ? len_bucket(long_thresh) : 64;
        if (lane == min(b_long, 63)) order[n_tiles] = b_long < 64 ? above : 0;
        if (lane == 0) order[n_tiles + 1] = 0;
    }
    __syncthreads();
    for (unsigned long long todo = present; todo; todo &= todo - 1) {
        const int b = __ffsll((long long)todo) - 1;
        int base = start[b];
        for (int w = 0; w < wave; ++w) base += wave_cnt[w][b];
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            if (i * nthr >= n_tiles) continue;
            const bool mine = bucket[i] == b;
            const unsigned long long m = __ballot(mine);
            if (mine) order[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
                          i * nthr + threadIdx.x;
            base += __popcll(m);
        }
    }
}

template <typename TK>
__device__ __forceinline__ void tile_bins32_body(int64_t n_isect, const TK *__restrict__ tkeys,
                                                 int32_t *__restrict__ bins, int64_t i0) {
    constexpr int KPT = 16 / (int)sizeof(TK);
    if (i0 >= n_isect) return;
    TK k[KPT];
    if (i0 + KPT <= n_isect) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(tkeys + i0);
        __builtin_memcpy(k, &raw, 16);
    } else {
#pragma unroll
        for (int j = 0; j < KPT; ++j) k[j] = (i0 + j < n_isect) ? tkeys[i0 + j] : (TK)0;
    }
    int32_t prev = (i0 > 0) ? (int32_t)tkeys[i0 - 1] : -1;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const int64_t idx = i0 + j;
        if (idx >= n_isect) break;
        const int32_t cur = (int32_t)k[j];
        if (idx == 0) bins[2 * cur] = 0;
        else if (prev != cur) {
            bins[2 * prev + 1] = (int32_t)idx;
            bins[2 * cur] = (int32_t)idx;
        }
        if (idx == n_isect - 1) bins[2 * cur + 1] = (int32_t)n_isect;
        prev = cur;
    }
}

// sorted tile ids -> tile_bins.  A thread takes the 16 bytes of keys at 16 * idx (8 sixteen-bit or 4 thirty-two-bit
// ids: one vector load) plus the key before them and writes the boundaries it sees (one key per thread with two
// scalar loads each took 12.9 us for the 8.3 M pairs of the benchmark scene).  `tkeys` is 16-byte aligned
// (workspace carve-outs are 256-byte aligned).
template <typename TK>
__global__ __launch_bounds__(256) void tile_bins32_kernel(int64_t n_isect, const TK *__restrict__ tkeys,
                                                          int32_t *__restrict__ bins,
                                                          const int32_t *__restrict__ n_dev) {
    constexpr int KPT = 16 / (int)sizeof(TK);
    if (n_dev) n_isect = min(n_isect, (int64_t)max(*n_dev, 0));   // speculative launch: n_isect is the capacity
    tile_bins32_body<TK>(n_isect, tkeys, bins, ((int64_t)blockIdx.x * 256 + threadIdx.x) * KPT);
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int bit_length(uint32_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

}  // namespace

SGN_EXPORT size_t sgn_scan_workspace_bytes(int n) {
    return (size_t)(sgn_cdiv(n > 0 ? n : 1, SCAN_CHUNK) + 1) * sizeof(int32_t);
}

// out[i] = sum_{r <= i} in[idx ? idx[r] : r];  with idx, `gathered` (n ints) receives in[idx[r]]
static int scan_launch(int n, const int32_t *in, const int32_t *idx, int32_t *gathered, int32_t *out, void *ws,
                       hipStream_t s, int32_t *total_host = nullptr, const int32_t *extra_dev = nullptr,
                       int32_t *extra_host = nullptr) {
    const int nb = sgn_cdiv(n, SCAN_CHUNK);
    int32_t *partial = (int32_t *)ws;
    sgn_timing_begin(SGN_T_SCAN, s);
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(SCAN_THREADS), 0, s, n, in, idx, idx ? gathered : nullptr,
                       partial);
    if (nb <= 4096) {       // every workgroup adds up the chunk totals before its own (<= 16 KB of reads each)
        hipLaunchKernelGGL(scan_final_kernel<false>, dim3(nb), dim3(SCAN_THREADS), 0, s, n, idx ? gathered : in, partial,
                           out, total_host, extra_dev, extra_host);
    } else {
        hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, nb, partial);
        hipLaunchKernelGGL(scan_final_kernel<true>, dim3(nb), dim3(SCAN_THREADS), 0, s, n, idx ? gathered : in, partial,
                           out, total_host, extra_dev, extra_host);
    }
    sgn_timing_end(SGN_T_SCAN, s);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_scan_i32(int n, const int32_t *in, int32_t *out, void *ws, size_t ws_bytes,
                            sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0, -1);
    if (n == 0) return 0;
    SGN_ARG_CHECK(in && out && ws, -2);
    SGN_ARG_CHECK(ws_bytes >= sgn_scan_workspace_bytes(n), -3);
    return scan_launch(n, in, nullptr, nullptr, out, ws, (hipStream_t)stream);
}

SGN_EXPORT int sgn_map_isect(int n, const float *xys, const float *depths, const int32_t *radii,
                             const int32_t *cum_tiles_hit, int tiles_x, int tiles_y, int block_width,
                             int64_t *isect_keys, int32_t *isect_vals, int semantics, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0, -1);
    SGN_ARG_CHECK(block_width >= 2 && block_width <= 16 && tiles_x > 0 && tiles_y > 0, -2);
    if (n == 0) return 0;
    SGN_ARG_CHECK(xys && depths && radii && cum_tiles_hit, -3);
    sgn_timing_begin(SGN_T_MAP, stream);
    hipLaunchKernelGGL(map_isect_kernel, dim3(sgn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, xys,
                       depths, radii, cum_tiles_hit, tiles_x, tiles_y, block_width, isect_keys, isect_vals, semantics);
    sgn_timing_end(SGN_T_MAP, stream);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_tile_bins(int64_t n_isect, const int64_t *keys_sorted, int n_tiles, int32_t *tile_bins,
                             sgn_stream_t stream) {
    SGN_ARG_CHECK(n_isect >= 0 && n_tiles > 0, -1);
    SGN_ARG_CHECK(tile_bins != nullptr, -2);
    hipStream_t s = (hipStream_t)stream;
    SGN_HIP_CHECK(hipMemsetAsync(tile_bins, 0, (size_t)n_tiles * 2 * sizeof(int32_t), s));
    if (n_isect == 0) return 0;
    SGN_ARG_CHECK(keys_sorted != nullptr, -3);
    sgn_timing_begin(SGN_T_BINS, s);
    hipLaunchKernelGGL(tile_bins_kernel, dim3(sgn_cdiv(n_isect, 256)), dim3(256), 0, s, n_isect, keys_sorted,
                       tile_bins, n_tiles);
    sgn_timing_end(SGN_T_BINS, s);
    SGN_LAUNCH_CHECK();
    return 0;
}

// ---- fused path entry points (declared in sgn_rast.h) ------------------------------------------
size_t sgn_sort_pairs32_ws_bytes(int64_t n);
void sgn_sort_pairs32_launch(uint32_t n, int end_bit, const uint32_t *kin, const int32_t *vin, uint32_t *kout,
                             int32_t *vout, void *ws, hipStream_t s, const int32_t *n_dev, int rank_mode);
void sgn_sort_pairs16_launch(uint32_t n, int end_bit, const uint16_t *kin, const int32_t *vin, uint16_t *kout,
                             int32_t *vout, void *ws, hipStream_t s, const int32_t *n_dev, int rank_mode);
void sgn_sort_depth_rank_launch(uint32_t n, const float *depths, const int32_t *radii, uint32_t *kout, int32_t *vout,
                                void *ws, hipStream_t s, int rank_mode);

// The depth ranking needs depths and radii only, i.e. it can be queued the moment the projection is: the caller's
// colour evaluation and activations then run BEHIND ~80 us of device work instead of in front of an idle GPU (the
// drop-in path loses its lead over the device at upstream's eager argument check / the reference's `radii.sum() == 0`).
SGN_EXPORT size_t sgn_depth_rank_workspace_bytes(int n) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    return 3 * al256(nn * 4) + sgn_sort_pairs32_ws_bytes(n);
}

SGN_EXPORT int sgn_depth_rank(int n, const float *depths, const int32_t *radii, int32_t *gid_by_rank, void *ws,
                              size_t ws_bytes, int sort_rank_mode, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0, -1);
    if (n == 0) return 0;
    SGN_ARG_CHECK(depths && radii && gid_by_rank && ws, -2);
    SGN_ARG_CHECK(ws_bytes >= sgn_depth_rank_workspace_bytes(n), -3);
    hipStream_t s = (hipStream_t)stream;
    char *p = (char *)ws;
    uint32_t *dkeys = (uint32_t *)p; p += al256((size_t)n * 4);
    int32_t *dvals = (int32_t *)p;   p += al256((size_t)n * 4);
    uint32_t *dkeys_sorted = (uint32_t *)p; p += al256((size_t)n * 4);
    // 12 launches: four passes of histogram / scan / scatter; the first pass reads its keys from (depths, radii) itself
    // (round 5: no key kernel).  (Rounds 4-5 could replay the chain as one HIP graph, opt-in: no gain on this ROCm,
    // profiles/r04ab_hip_graphs_ab.log; removed in round 6.)
    (void)dkeys; (void)dvals;
    sgn_timing_begin(SGN_T_SORT, s);
    sgn_sort_depth_rank_launch((uint32_t)n, depths, radii, dkeys_sorted, gid_by_rank, p, s, sort_rank_mode);
    sgn_timing_end(SGN_T_SORT, s);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT size_t sgn_bin_prepare_workspace_bytes(int n) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    return al256(sgn_scan_workspace_bytes(n)) + 5 * al256(nn * 4) + sgn_sort_pairs32_ws_bytes(n);
}

static Cull make_cull(const float *conics, const float *opac, int opac_is_logit, int cull) {
    Cull c;
    c.conics = conics; c.opac = opac; c.opac_is_logit = opac_is_logit;
    c.enable = (cull && conics && opac) ? 1 : 0;
    return c;
}

// `total_host`: device-visible pointer into pinned host memory (or nullptr) that receives cum_by_rank[n-1] from the scan
// itself (api.cpp sgn_rasterize_fwd_all: the count's read-back without a copy command)
int sgn_bin_prepare_total(int n, const float *xys, const float *depths, const int32_t *radii,
                          const float *conics, const float *opacities, int opacity_is_logit, int cull,
                          int tiles_x, int tiles_y, int block_width, int32_t *cum_by_rank,
                          int32_t *gid_by_rank, int rank_ready, float *bin_records, void *ws, size_t ws_bytes,
                          int sort_rank_mode, int32_t *total_host, const int32_t *extra_dev, int32_t *extra_host,
                          int semantics, sgn_stream_t stream);

SGN_EXPORT int sgn_bin_prepare(int n, const float *xys, const float *depths, const int32_t *radii,
                               const float *conics, const float *opacities, int opacity_is_logit, int cull,
                               int tiles_x, int tiles_y, int block_width, int32_t *cum_by_rank,
                               int32_t *gid_by_rank, int rank_ready, float *bin_records, void *ws, size_t ws_bytes,
                               int sort_rank_mode, int semantics, sgn_stream_t stream) {
    return sgn_bin_prepare_total(n, xys, depths, radii, conics, opacities, opacity_is_logit, cull, tiles_x, tiles_y,
                                 block_width, cum_by_rank, gid_by_rank, rank_ready, bin_records, ws, ws_bytes,
                                 sort_rank_mode, nullptr, nullptr, nullptr, semantics, stream);
}

int sgn_bin_prepare_total(int n, const float *xys, const float *depths, const int32_t *radii,
                          const float *conics, const float *opacities, int opacity_is_logit, int cull,
                          int tiles_x, int tiles_y, int block_width, int32_t *cum_by_rank,
                          int32_t *gid_by_rank, int rank_ready, float *bin_records, void *ws, size_t ws_bytes,
                          int sort_rank_mode, int32_t *total_host, const int32_t *extra_dev, int32_t *extra_host,
                          int semantics, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0, -1);
    SGN_ARG_CHECK(block_width >= 2 && block_width <= 16 && tiles_x > 0 && tiles_y > 0, -2);
    if (n == 0) return 0;
    SGN_ARG_CHECK(xys && depths && radii && cum_by_rank && gid_by_rank && bin_records && ws, -3);
    SGN_ARG_CHECK(ws_bytes >= sgn_bin_prepare_workspace_bytes(n), -4);
    hipStream_t s = (hipStream_t)stream;
    char *p = (char *)ws;
    void *scan_ws = p; p += al256(sgn_scan_workspace_bytes(n));
    uint32_t *dkeys = (uint32_t *)p; p += al256((size_t)n * 4);
    int32_t *dvals = (int32_t *)p;   p += al256((size_t)n * 4);
    uint32_t *dkeys_sorted = (uint32_t *)p; p += al256((size_t)n * 4);
    int32_t *cnt_gid = (int32_t *)p; p += al256((size_t)n * 4);
    int32_t *cnt_r = (int32_t *)p;   p += al256((size_t)n * 4);
    void *sort_ws = p;
    const Cull c = make_cull(conics, opacities, opacity_is_logit, cull);
    BinRec *recs = reinterpret_cast<BinRec *>(bin_records);
    sgn_timing_begin(SGN_T_MAP, s);
    hipLaunchKernelGGL(bin_count_kernel, dim3(sgn_cdiv(n, 256)), dim3(256), 0, s, n, xys, depths, radii, c, tiles_x,
                       tiles_y, block_width, dkeys, dvals, recs, cnt_gid, rank_ready ? 0 : 1, semantics);
    sgn_timing_end(SGN_T_MAP, s);
    if (!rank_ready) {       // rank_ready: gid_by_rank already holds sgn_depth_rank's result for these depths / radii
        sgn_timing_begin(SGN_T_SORT, s);
        sgn_sort_pairs32_launch((uint32_t)n, 32, dkeys, dvals, dkeys_sorted, gid_by_rank, sort_ws, s, nullptr, sort_rank_mode);
        sgn_timing_end(SGN_T_SORT, s);
    }
    // cum_by_rank[r] = sum of the kept-tile counts of ranks <= r: the scan's first pass gathers cnt_gid[gid_by_rank[r]]
    return scan_launch(n, cnt_gid, gid_by_rank, cnt_r, cum_by_rank, scan_ws, s, total_host,
                       (total_host && extra_dev && extra_host) ? extra_dev : nullptr,
                       (total_host && extra_dev && extra_host) ? extra_host : nullptr);
}

SGN_EXPORT size_t sgn_bin_intersect_workspace_bytes(int64_t n_isect) {
    const size_t ni = (size_t)(n_isect > 0 ? n_isect : 1);
    return 3 * al256(ni * 4) + sgn_sort_pairs32_ws_bytes(n_isect);
}

// `also_zero_words`: int32 words BEHIND tile_bins' 2 * n_tiles that the emission clears as well (the composite forward
// puts the raster kernels' tile statistics there: one more clear that needs no launch of its own)
int sgn_bin_intersect_zero(int n, int64_t n_isect, const float *bin_records, const int32_t *cum_by_rank,
                           const int32_t *gid_by_rank, int tiles_x, int tiles_y, int block_width,
                           int32_t *gaussian_ids_sorted, int32_t *tile_bins, int quadrant_masks, void *ws,
                           size_t ws_bytes, const int32_t *n_isect_dev, int sort_rank_mode, int also_zero_words,
                           sgn_stream_t stream);

SGN_EXPORT int sgn_bin_intersect(int n, int64_t n_isect, const float *bin_records, const int32_t *cum_by_rank,
                                 const int32_t *gid_by_rank, int tiles_x, int tiles_y, int block_width,
                                 int32_t *gaussian_ids_sorted, int32_t *tile_bins, int quadrant_masks, void *ws,
                                 size_t ws_bytes, const int32_t *n_isect_dev, int sort_rank_mode, sgn_stream_t stream) {
    return sgn_bin_intersect_zero(n, n_isect, bin_records, cum_by_rank, gid_by_rank, tiles_x, tiles_y, block_width,
                                  gaussian_ids_sorted, tile_bins, quadrant_masks, ws, ws_bytes, n_isect_dev,
                                  sort_rank_mode, 0, stream);
}

int sgn_bin_intersect_zero(int n, int64_t n_isect, const float *bin_records, const int32_t *cum_by_rank,
                           const int32_t *gid_by_rank, int tiles_x, int tiles_y, int block_width,
                           int32_t *gaussian_ids_sorted, int32_t *tile_bins, int quadrant_masks, void *ws,
                           size_t ws_bytes, const int32_t *n_isect_dev, int sort_rank_mode, int also_zero_words,
                           sgn_stream_t stream) {
    SGN_ARG_CHECK(also_zero_words >= 0, -7);
    SGN_ARG_CHECK(n >= 0 && n_isect >= 0 && n_isect < ((int64_t)1 << 31), -1);
    SGN_ARG_CHECK(block_width >= 2 && block_width <= 16 && tiles_x > 0 && tiles_y > 0, -2);
    SGN_ARG_CHECK(tile_bins != nullptr, -3);
    SGN_ARG_CHECK(!quadrant_masks || (block_width == 16 && n < SGN_QMASK_MAX_IDS), -6);
    hipStream_t s = (hipStream_t)stream;
    const int n_tiles = tiles_x * tiles_y;
    if (n_isect == 0 || n == 0) {
        SGN_HIP_CHECK(hipMemsetAsync(tile_bins, 0, ((size_t)n_tiles * 2 + also_zero_words) * sizeof(int32_t), s));
        return 0;
    }
    SGN_ARG_CHECK(bin_records && cum_by_rank && gid_by_rank && gaussian_ids_sorted && ws, -4);
    SGN_ARG_CHECK(ws_bytes >= sgn_bin_intersect_workspace_bytes(n_isect), -5);
    const int tile_bits = bit_length((uint32_t)(n_tiles - 1)) > 0 ? bit_length((uint32_t)(n_tiles - 1)) : 1;
    char *p = (char *)ws;
    uint32_t *tkeys = (uint32_t *)p;        p += al256((size_t)n_isect * 4);
    int32_t *tvals = (int32_t *)p;          p += al256((size_t)n_isect * 4);
    uint32_t *tkeys_sorted = (uint32_t *)p; p += al256((size_t)n_isect * 4);
    void *sort_ws = p;
    const BinRec *recs = reinterpret_cast<const BinRec *>(bin_records);
    if (n_tiles <= 65536) {       // 16-bit tile keys (the buffers keep their 4-byte-per-key size)
        uint16_t *k16 = (uint16_t *)tkeys, *k16s = (uint16_t *)tkeys_sorted;
        sgn_timing_begin(SGN_T_MAP, s);
        hipLaunchKernelGGL(bin_emit_kernel<uint16_t>, dim3(sgn_cdiv(n, 64)), dim3(64), 0, s, n, gid_by_rank, cum_by_rank,
                           recs, tiles_x, tiles_y, block_width, k16, tvals, (int)n_isect, tile_bins, 2 * n_tiles + also_zero_words,
                           quadrant_masks);
        sgn_timing_end(SGN_T_MAP, s);
        sgn_timing_begin(SGN_T_SORT, s);
        sgn_sort_pairs16_launch((uint32_t)n_isect, tile_bits, k16, tvals, k16s, gaussian_ids_sorted, sort_ws, s,
                                n_isect_dev, sort_rank_mode);
        sgn_timing_end(SGN_T_SORT, s);
        sgn_timing_begin(SGN_T_BINS, s);
        hipLaunchKernelGGL(tile_bins32_kernel<uint16_t>, dim3(sgn_cdiv(n_isect, 256 * 8)), dim3(256), 0, s, n_isect, k16s,
                           tile_bins, n_isect_dev);
        sgn_timing_end(SGN_T_BINS, s);
    } else {
        sgn_timing_begin(SGN_T_MAP, s);
        hipLaunchKernelGGL(bin_emit_kernel<uint32_t>, dim3(sgn_cdiv(n, 64)), dim3(64), 0, s, n, gid_by_rank, cum_by_rank,
                           recs, tiles_x, tiles_y, block_width, tkeys, tvals, (int)n_isect, tile_bins, 2 * n_tiles + also_zero_words,
                           quadrant_masks);
        sgn_timing_end(SGN_T_MAP, s);
        sgn_timing_begin(SGN_T_SORT, s);
        sgn_sort_pairs32_launch((uint32_t)n_isect, tile_bits, tkeys, tvals, tkeys_sorted, gaussian_ids_sorted, sort_ws,
                                s, n_isect_dev, sort_rank_mode);
        sgn_timing_end(SGN_T_SORT, s);
        sgn_timing_begin(SGN_T_BINS, s);
        hipLaunchKernelGGL(tile_bins32_kernel<uint32_t>, dim3(sgn_cdiv(n_isect, 256 * 4)), dim3(256), 0, s, n_isect,
                           tkeys_sorted, tile_bins, n_isect_dev);
        sgn_timing_end(SGN_T_BINS, s);
    }
    SGN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------ window recognition
// The scene graph renders its sub-model passes (sgn_splatfacto_scene_graph.py:364-366) from torch.cat COPIES of the
// per-model slices of the main projection (:270-276): new tensors, same bytes.  To reuse the depth list binned for the
// whole scene the host must know that a call's geometry EQUALS rows [lo, lo + n_win) of the geometry that list was
// built from; this kernel compares them bit for bit (integer compare: NaNs match themselves) for up to 4 candidate
// offsets in one pass and ORs a mismatch flag per candidate.
namespace {
struct MatchArgs {
    const uint32_t *w[6];   // window tensors: xys[2], depths[1], radii[1], num_tiles_hit[1], conics[3], opac[1]
    const uint32_t *f[6];   // the same six of the full scene
    int width[6];
    int lo[4];
    int n_cand;
};
// flat, coalesced: thread j compares WORD j of every tensor (j < n_win * width) against word lo * width + j of the
// full tensor, for each candidate offset
__global__ __launch_bounds__(256) void rows_match_kernel(int n_win, MatchArgs a, int32_t *__restrict__ mismatch) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    // candidates already known to mismatch need no further work (and no further atomics: a window that is NOT at a
    // candidate offset differs almost everywhere, and 40 k waves hammering one address cost more than the compare)
    unsigned dead = 0u;
    for (int c = 0; c < a.n_cand; ++c)
        if (__atomic_load_n(mismatch + c, __ATOMIC_RELAXED) != 0) dead |= 1u << c;
    if (dead == (1u << a.n_cand) - 1u) return;
    unsigned bad = 0u;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        if (a.w[t] == nullptr) continue;
        const int wd = a.width[t];
        if (j >= (size_t)n_win * wd) continue;
        const uint32_t v = a.w[t][j];
        for (int c = 0; c < a.n_cand; ++c)
            if (!((dead >> c) & 1u) && a.f[t][(size_t)a.lo[c] * wd + j] != v) bad |= 1u << c;
    }
    for (int c = 0; c < a.n_cand; ++c) {
        const bool any = __ballot((bad >> c) & 1u) != 0ull;
        if (any && (threadIdx.x & 63) == 0) atomicOr(mismatch + c, 1);
    }
}
}  // namespace

SGN_EXPORT int sgn_rows_match(int n_win, int n_full, int n_cand, const int32_t *cand_lo_host, const float *xys_w,
                              const float *depths_w, const int32_t *radii_w, const int32_t *num_tiles_hit_w,
                              const float *conics_w, const float *opacities_w, const float *xys, const float *depths,
                              const int32_t *radii, const int32_t *num_tiles_hit, const float *conics,
                              const float *opacities, int32_t *mismatch, sgn_stream_t stream) {
    SGN_ARG_CHECK(n_win > 0 && n_full >= n_win && n_cand >= 1 && n_cand <= 4 && cand_lo_host && mismatch, -1);
    // any subset of the six tensors may be compared (a caller that has settled some of them another way passes NULL for
    // both sides of those), but each pair must be given or omitted together, and at least one must be given
    SGN_ARG_CHECK((xys_w == nullptr) == (xys == nullptr) && (depths_w == nullptr) == (depths == nullptr) &&
                      (radii_w == nullptr) == (radii == nullptr) &&
                      (num_tiles_hit_w == nullptr) == (num_tiles_hit == nullptr), -2);
    SGN_ARG_CHECK((conics_w == nullptr) == (conics == nullptr) && (opacities_w == nullptr) == (opacities == nullptr), -3);
    SGN_ARG_CHECK(xys_w || depths_w || radii_w || num_tiles_hit_w || conics_w || opacities_w, -5);
    MatchArgs a;
    const void *w[6] = {xys_w, depths_w, radii_w, num_tiles_hit_w, conics_w, opacities_w};
    const void *f[6] = {xys, depths, radii, num_tiles_hit, conics, opacities};
    const int width[6] = {2, 1, 1, 1, 3, 1};
    for (int t = 0; t < 6; ++t) { a.w[t] = (const uint32_t *)w[t]; a.f[t] = (const uint32_t *)f[t]; a.width[t] = width[t]; }
    a.n_cand = n_cand;
    for (int c = 0; c < 4; ++c) {
        a.lo[c] = c < n_cand ? cand_lo_host[c] : 0;
        SGN_ARG_CHECK(a.lo[c] >= 0 && a.lo[c] + n_win <= n_full, -4);
    }
    hipStream_t s = (hipStream_t)stream;
    SGN_HIP_CHECK(hipMemsetAsync(mismatch, 0, sizeof(int32_t) * n_cand, s));
    int max_width = 1;
    for (int t = 0; t < 6; ++t)
        if (w[t] != nullptr && width[t] > max_width) max_width = width[t];
    hipLaunchKernelGGL(rows_match_kernel, dim3(sgn_cdiv((int64_t)n_win * max_width, 256)), dim3(256), 0, s, n_win, a,
                       mismatch);
    SGN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------- tile order
// Launch order of the raster workgroups: tiles with the longest depth lists first (longest-processing-time-first).
// The hardware starts workgroups in index order; on skewed content the few very long lists otherwise start whenever
// their tile index comes up and the kernel ends with a tail of lone waves.  A counting sort into half-octave length
// classes (64 buckets) is all the ordering this needs; one workgroup, ~5 us for 9600 tiles.
namespace {
constexpr int ORDER_PER_THREAD = 16;   // tiles per thread kept in registers (one workgroup covers 16384 tiles)
__global__ __launch_bounds__(1024) void tile_order_kernel(int n_tiles, const int2 *__restrict__ bins,
                                                          const int32_t *__restrict__ kmax, int long_thresh,
                                                          int small_q16, int32_t *__restrict__ order) {
    __shared__ int wave_cnt[16][64], start[64];
    // every load of the thread is issued before the first use: as a plain loop this kernel was a chain of ~20 dependent
    // global round trips (16 us for 9600 tiles)
    signed char bucket[ORDER_PER_THREAD];
    __shared__ unsigned long long sums[2];      // backward order only: list entries WALKED by the forward / LISTED
    if (threadIdx.x < 2) sums[threadIdx.x] = 0ull;
    long long walked = 0, listed = 0;
#pragma unroll
    for (int i = 0; i < ORDER_PER_THREAD; ++i) {
        const int t = i * 1024 + threadIdx.x;
        bucket[i] = (signed char)(t < n_tiles ? len_bucket(order_len(t, bins, kmax, long_thresh, small_q16)) : -1);
        if (kmax != nullptr && t < n_tiles) {
            const int2 r = bins[t];
            const int len = max(r.y - r.x, 0);
            listed += len;
            walked += len > 0 ? min(len, max(0, kmax[2 * t] - r.x + 1)) : 0;
        }
    }
    order_by_class<ORDER_PER_THREAD>(bucket, min(n_tiles, ORDER_PER_THREAD * 1024), long_thresh, order, wave_cnt, start);
    if (kmax != nullptr) {
        // order[n_tiles + 1] = 1000 * walked / listed: how much of the lists the forward got through before its tiles
        // saturated.  The host reads it with its next read-back and decides from it whether quadrant masks (paid per
        // LISTED entry by the emission, earned per WALKED entry by the raster kernels) are worth computing.
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            walked += __shfl_xor(walked, d, 64);
            listed += __shfl_xor(listed, d, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&sums[0], (unsigned long long)walked);
            atomicAdd(&sums[1], (unsigned long long)listed);
        }
        __syncthreads();
        if (threadIdx.x == 0) order[n_tiles + 1] = sums[1] ? (int)((1000ull * sums[0]) / sums[1]) : 0;
    }
}

// images with more than 16384 tiles: the r02 form (wave-aggregated LDS counters, any tile count)
__global__ __launch_bounds__(1024) void tile_order_big_kernel(int n_tiles, const int2 *__restrict__ bins,
                                                              const int32_t *__restrict__ kmax, int long_thresh,
                                                              int small_q16, int32_t *__restrict__ order) {
    __shared__ int hist[64], start[64];
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int t0 = 0; t0 < n_tiles; t0 += 1024) {
        const int t = t0 + threadIdx.x;
        bucket_slot(hist, t < n_tiles ? len_bucket(order_len(t, bins, kmax, long_thresh, small_q16)) : 0, t < n_tiles);
    }
    __syncthreads();
    if (threadIdx.x == 0) {                      // longest class first
        int run = 0, n_long = 0;
        const int b_long = long_thresh > 0 ? len_bucket(long_thresh) : 64;
        for (int b = 63; b >= 0; --b) {
            start[b] = run;
            run += hist[b];
            if (b >= b_long) n_long = run;       // classes >= the threshold's class form a prefix of the order
        }
        order[n_tiles] = n_long;
        order[n_tiles + 1] = 0;
    }
    __syncthreads();
    for (int t0 = 0; t0 < n_tiles; t0 += 1024) {
        const int t = t0 + threadIdx.x;
        const bool act = t < n_tiles;
        const int slot = bucket_slot(start, act ? len_bucket(order_len(t, bins, kmax, long_thresh, small_q16)) : 0, act);
        if (act) order[slot] = t;
    }
}
}  // namespace

// Multi-workgroup form (round 4).  The single-workgroup kernels above are one CU doing ~500 ballots back to back while
// 255 CUs wait: 10-30 us per launch, two to four launches per step, on the critical path in front of each raster
// kernel.  Here every workgroup classes 256 tiles (one wave-aggregated LDS atomic per class and wave), reserves a range
// per class with ONE global atomic per class present, and leaves a packed (class, rank) word per tile; the LAST
// workgroup to finish (arrival counter) turns the 64 class totals into starts and scatters the permutation.
// Cross-workgroup data follows MI355X_MICROARCH.md "inter-workgroup visibility": records and totals are written with
// agent-scope (write-through) stores / atomics and drained before the arrival, and read back with agent-scope loads.
// `scratch` ([0,64) class totals, [64] arrivals, [66,70) two 64-bit sums, [72, 72 + n_tiles) records) must be zero on
// first use; the last workgroup leaves it zero again.  Ranks inside a class follow arrival order (launch order only —
// results never depend on it).
namespace {
constexpr int ORDER_SCRATCH_HEAD = 72;
typedef __attribute__((address_space(1))) int32_t g_i32;
__global__ __launch_bounds__(256) void tile_order_mb_kernel(int n_tiles, const int2 *__restrict__ bins,
                                                            const int32_t *__restrict__ kmax, int long_thresh,
                                                            int small_q16, int32_t *__restrict__ order,
                                                            int32_t *__restrict__ scratch) {
    __shared__ int lhist[64], lbase[64], start[64];
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63;
    const int t = blockIdx.x * 256 + tid;
    const bool act = t < n_tiles;
    if (tid < 64) lhist[tid] = 0;
    __syncthreads();
    const int b = act ? len_bucket(order_len(t, bins, kmax, long_thresh, small_q16)) : 0;
    const int lrank = bucket_slot(lhist, b, act);
    long long walked = 0, listed = 0;
    if (kmax != nullptr && act) {
        const int2 r = bins[t];
        const int len = max(r.y - r.x, 0);
        listed = len;
        walked = len > 0 ? min(len, max(0, kmax[2 * t] - r.x + 1)) : 0;
    }
    __syncthreads();
    if (tid < 64) lbase[tid] = lhist[tid] > 0 ? atomicAdd(scratch + tid, lhist[tid]) : 0;
    if (kmax != nullptr) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            walked += __shfl_xor(walked, d, 64);
            listed += __shfl_xor(listed, d, 64);
        }
        if (lane == 0 && listed > 0) {
            atomicAdd((unsigned long long *)(scratch + 66), (unsigned long long)walked);
            atomicAdd((unsigned long long *)(scratch + 68), (unsigned long long)listed);
        }
    }
    __syncthreads();
    if (act)
        __hip_atomic_store((g_i32 *)(scratch + ORDER_SCRATCH_HEAD + t), (int)(((unsigned)b << 26) | (unsigned)(lbase[b] + lrank)),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains before the arrival
    __syncthreads();
    if (tid == 0) is_last = atomicAdd(scratch + 64, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    // ---- the last workgroup: class totals -> starts (longest class first), then the scatter
    if (tid < 64) {
        const int tot = __hip_atomic_load((g_i32 *)(scratch + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int above = tot;           // inclusive suffix sum over classes >= mine, then exclusive
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int u = __shfl_down(above, d, 64);
            if (lane + d < 64) above += u;
        }
        start[tid] = above - tot;
        const int b_long = long_thresh > 0 ? len_bucket(long_thresh) : 64;
        if (tid == min(b_long, 63)) order[n_tiles] = b_long < 64 ? above : 0;
        if (tid == 0) {
            int stat = 0;
            if (kmax != nullptr) {
                const unsigned long long w = __hip_atomic_load((unsigned long long *)(scratch + 66), __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long l = __hip_atomic_load((unsigned long long *)(scratch + 68), __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT);
                stat = l ? (int)((1000ull * w) / l) : 0;
            }
            order[n_tiles + 1] = stat;
        }
    }
    __syncthreads();
    // sixteen records per thread requested before the first is used: the loop is a chain of L2 round trips otherwise
    for (int i0 = 0; i0 < n_tiles; i0 += 256 * 16) {
        int rec[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = i0 + j * 256 + tid;
            rec[j] = i < n_tiles ? __hip_atomic_load((g_i32 *)(scratch + ORDER_SCRATCH_HEAD + i), __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT) : 0;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = i0 + j * 256 + tid;
            // (the position is < n_tiles whenever the scratch head was zero on entry, which every completed launch
            // guarantees; an aborted launch or a foreign buffer must not turn into an out-of-bounds store)
            const int at = start[(unsigned)rec[j] >> 26] + (rec[j] & ((1 << 26) - 1));
            if (i < n_tiles && (unsigned)at < (unsigned)n_tiles) order[at] = i;
        }
    }
    __syncthreads();
    if (tid < ORDER_SCRATCH_HEAD) scratch[tid] = 0;       // ready for the next launch (the records are overwritten)
}
}  // namespace

SGN_EXPORT size_t sgn_tile_order_scratch_bytes(int n_tiles) {
    return sizeof(int32_t) * (size_t)(ORDER_SCRATCH_HEAD + (n_tiles > 0 ? n_tiles : 0));
}

SGN_EXPORT int sgn_tile_order(int n_tiles, const int32_t *tile_bins, const int32_t *tile_stats, int long_thresh,
                              int small_q16, int32_t *order, void *scratch, size_t scratch_bytes, sgn_stream_t stream) {
    SGN_ARG_CHECK(n_tiles > 0 && tile_bins && order, -1);
    if (scratch != nullptr && n_tiles < (1 << 26)) {
        SGN_ARG_CHECK(scratch_bytes >= sgn_tile_order_scratch_bytes(n_tiles), -2);
        hipLaunchKernelGGL(tile_order_mb_kernel, dim3(sgn_cdiv(n_tiles, 256)), dim3(256), 0, (hipStream_t)stream, n_tiles,
                           (const int2 *)tile_bins, tile_stats, long_thresh, small_q16, order, (int32_t *)scratch);
        SGN_LAUNCH_CHECK();
        return 0;
    }
    if (n_tiles <= ORDER_PER_THREAD * 1024)
        hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n_tiles,
                           (const int2 *)tile_bins, tile_stats, long_thresh, small_q16, order);
    else
        hipLaunchKernelGGL(tile_order_big_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n_tiles,
                           (const int2 *)tile_bins, tile_stats, long_thresh, small_q16, order);
    SGN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------- list window
// Sub-list of a binned scene for an id window.  The scene graph renders its objects-only and background-only
// accumulation passes (sgn_splatfacto_scene_graph.py:364-366) from subsets of the Gaussians of the main pass; the host
// serves them from the main pass's depth list (id range / window recognition) instead of ranking, emitting and sorting
// again.  Walking the FULL list with the other Gaussians made inert is fine for the background (90 % of the entries are
// its own and the tiles saturate as early as in the main pass) and ruinous for the objects: a tenth of the entries are
// live, nothing saturates, so every tile walks its whole list — 0.2 ms forward and 0.3 ms backward for a pass that
// draws 100 k Gaussians.  This keeps, in order, the entries whose id lies in [id_lo, id_hi): one wave per tile counts,
// one workgroup scans the tile counts, one wave per tile writes — two reads of the list, no sort, bit-identical
// relative order (so image and gradients equal those of the reference's own re-binned pass).
namespace {
__device__ __forceinline__ bool in_window(int raw, int idmask, int lo, int hi) {
    const int id = raw & idmask;
    return id >= lo && id < hi;
}

__global__ __launch_bounds__(256) void list_window_count_kernel(int n_tiles, const int32_t *__restrict__ ids,
                                                                const int2 *__restrict__ bins, int lo, int hi,
                                                                int idmask, int32_t *__restrict__ counts) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= n_tiles) return;
    const int2 r = bins[t];
    int c = 0;
    for (int k = r.x + lane; k < r.y; k += 64) c += in_window(ids[k], idmask, lo, hi) ? 1 : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if (lane == 0) counts[t] = c;
}

// exclusive scan of the tile counts in place, total behind them; one workgroup, any n
__global__ __launch_bounds__(1024) void list_window_scan_kernel(int n_tiles, int32_t *__restrict__ counts) {
    __shared__ int wave_sum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n_tiles; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n_tiles ? counts[i] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int u = __shfl_up(inc, d, 64);
            if (lane >= d) inc += u;
        }
        if (lane == 63) wave_sum[wave] = inc;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w) before += wave_sum[w];
        if (i < n_tiles) counts[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[n_tiles] = carry;
}

__global__ __launch_bounds__(256) void list_window_write_kernel(int n_tiles, const int32_t *__restrict__ ids,
                                                                const int2 *__restrict__ bins, int lo, int hi,
                                                                int idmask, const int32_t *__restrict__ starts,
                                                                int32_t *__restrict__ ids_out, int2 *__restrict__ bins_out) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= n_tiles) return;
    const int2 r = bins[t];
    const int first = starts[t];
    int at = first;
    for (int k0 = r.x; k0 < r.y; k0 += 64) {
        const int k = k0 + lane;
        const int raw = k < r.y ? ids[k] : 0;
        const bool keep = k < r.y && in_window(raw, idmask, lo, hi);
        const unsigned long long m = __ballot(keep);
        if (keep) ids_out[at + __popcll(m & ((1ull << lane) - 1ull))] = raw;
        at += __popcll(m);
    }
    if (lane == 0) bins_out[t] = make_int2(first, at);
}
}  // namespace

SGN_EXPORT size_t sgn_list_window_workspace_bytes(int n_tiles) { return sizeof(int32_t) * ((size_t)(n_tiles > 0 ? n_tiles : 0) + 1); }

SGN_EXPORT int sgn_list_window(int n_tiles, const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, int id_lo,
                               int id_hi, int ids_qmask, int32_t *ids_out, int32_t *tile_bins_out, void *ws,
                               size_t ws_bytes, sgn_stream_t stream) {
    SGN_ARG_CHECK(n_tiles > 0 && gaussian_ids_sorted && tile_bins && ids_out && tile_bins_out && ws, -1);
    SGN_ARG_CHECK(id_lo >= 0 && id_hi >= id_lo, -2);
    SGN_ARG_CHECK(ws_bytes >= sgn_list_window_workspace_bytes(n_tiles), -3);
    hipStream_t s = (hipStream_t)stream;
    int32_t *counts = (int32_t *)ws;
    const int idmask = ids_qmask ? (SGN_QMASK_MAX_IDS - 1) : -1;
    hipLaunchKernelGGL(list_window_count_kernel, dim3(sgn_cdiv(n_tiles, 4)), dim3(256), 0, s, n_tiles, gaussian_ids_sorted,
                       (const int2 *)tile_bins, id_lo, id_hi, idmask, counts);
    hipLaunchKernelGGL(list_window_scan_kernel, dim3(1), dim3(1024), 0, s, n_tiles, counts);
    hipLaunchKernelGGL(list_window_write_kernel, dim3(sgn_cdiv(n_tiles, 4)), dim3(256), 0, s, n_tiles, gaussian_ids_sorted,
                       (const int2 *)tile_bins, id_lo, id_hi, idmask, counts, ids_out, (int2 *)tile_bins_out);
    SGN_LAUNCH_CHECK();
    return 0;
}
