// raster.hip — per-tile front-to-back alpha compositing (forward) and the reverse per-pixel walk
// with per-Gaussian gradient reduction (backward), gfx950.
//
// Replaces gsplat 0.1.x forward.cu:rasterize_forward and backward.cu:rasterize_backward_kernel
// (3-channel path), reached from the reference at
// sgn_splatfacto.py:954-967 (rgb+alpha) and :982-994 (depth).
//
// CDNA4 design (not the upstream 256-thread-tile / shared-memory-batch shape):
//   * ONE wave64 owns ONE 16x16 tile.  Each lane carries 4 pixels (one per 8x8 quadrant), so a
//     workgroup is a single wave: no LDS staging, no __syncthreads, no cross-wave early-exit vote.
//   * Per-Gaussian operands live in 48-byte rows (build_grec).  They are wave-uniform in the hot loop, so the
//     wave chases gaussian_ids_sorted[k] -> row with scalar loads (s_load_dword, s_load_dwordx8 + x4) into
//     SGPRs one entry ahead and the VALU instructions take them as scalar operands: no LDS traffic, no VGPRs,
//     and only the (tile, Gaussian) pairs actually walked are ever fetched.  Lists long enough to leave a
//     lone wave latency-bound go through 64-entry batches staged in wave-private LDS instead.  (Rounds 1-5 also
//     carried a "stream" form — rows copied into depth order first — for A/B runs; it lost everywhere and was
//     removed in round 6, like the forced one- / four-wave shapes, the XCD swizzle and the MFMA reduction.)
//   * Per-quadrant wave-uniform skips (`__ballot`) give the early termination and the
//     "nobody in this 8x8 block is touched" shortcut for free in the scalar branch unit.
//   * Backward: the 4 pixels of a lane are accumulated in registers, so ONE wave reduction per
//     (tile, Gaussian) replaces upstream's 8 warp reductions — a transposed reduction on gfx950's
//     v_permlane32_swap / v_permlane16_swap (two values folded per swap) — and nine lanes then issue a
//     single global_atomic_add_f32 into a packed 48-byte per-Gaussian gradient row (one cache line),
//     unpacked into v_xy / v_conic / v_colors / v_opacity afterwards.
//
// Arithmetic contract of the hot loop (shared with oracle/c/sgn_oracle.c; this TU is built with
// -ffp-contract=off and spells every fma):
//   sigma = fma(b*dx, dy, fma(hc*dy, dy, (ha*dx)*dx))   with ha = a/2, hc = c/2  (exact scaling)
//   alpha = min(0.999, opac * exp(-sigma));  skip if sigma < 0 or alpha < 1/255
//   nT = T*(1-alpha); stop (not composited) if nT <= 1e-4;  C = fma(color, alpha*T, C)
#include "sgn_common.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

struct __attribute__((aligned(16))) Rec {
    float x, y, opac, ha;   // ha = 0.5 * conic.x
    float b, hc, r, g;      // b = conic.y, hc = 0.5 * conic.z
    float bl;               // blue
    int gid;                // Gaussian id (backward scatter target)
    float ex, ey;           // half-extents of the bbox of {alpha >= 1/255} (+margin); < 0: never visible
};
static_assert(sizeof(Rec) == SGN_RECORD_FLOATS * sizeof(float), "record size");

__device__ __forceinline__ float exp_portable(float x) {
    // same recipe as oracle/c/sgn_oracle.c exp_portable (written independently; bit-identical)
    float t = x * 1.44269504088896341f;
    t = fminf(fmaxf(t, -125.0f), 126.0f);
    const float n = __builtin_rintf(t);
    const float f = t - n;
    float p = 1.53533063e-4f;
    p = fmaf(p, f, 1.33988744e-3f);
    p = fmaf(p, f, 9.61843736e-3f);
    p = fmaf(p, f, 5.55035681e-2f);
    p = fmaf(p, f, 2.40226488e-1f);
    p = fmaf(p, f, 6.93147182e-1f);
    p = fmaf(p, f, 1.0f);
    return ldexpf(p, (int)n);
}

template <bool EXACT>
__device__ __forceinline__ float sgn_exp(float x) {
    if constexpr (EXACT) return exp_portable(x);
    else return __expf(x);
}

// Packing is two streaming kernels: (1) per-Gaussian AoS rows (coalesced, N x 48 B), (2) a pure
// 16-byte-granule gather of those rows into depth order: three lanes move one record, so the
// stores of a wave are one contiguous 1 KiB span and each gathered row is one or two cache lines
// (instead of four separate SoA gathers per intersection).
__global__ __launch_bounds__(256) void build_grec_kernel(int n, const float *__restrict__ xys,
                                                         const float *__restrict__ conics,
                                                         const float *__restrict__ colors,
                                                         const float *__restrict__ opac, int opac_is_logit,
                                                         int id_lo, int id_hi, int window,
                                                         float4 *__restrict__ grec,
                                                         const int32_t *__restrict__ skip_flag) {
    if (skip_flag != nullptr && *skip_flag == 0) return;   // this pass's result is reused (sgn_depth_reuse): no rows
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
    // window != 0: the four input arrays hold rows [id_lo, id_hi) only (a sub-model's own tensors), row g - id_lo
    const int off = window ? id_lo : 0;
    xys -= 2 * (ptrdiff_t)off; conics -= 3 * (ptrdiff_t)off; colors -= 3 * (ptrdiff_t)off; opac -= off;
    if (g < id_lo || g >= id_hi) {
        // sub-model pass over a shared depth list (scene graph): Gaussians outside [id_lo, id_hi) become inert
        // rows — zero opacity (alpha < 1/255 on every pixel) and a bbox no quadrant test can pass
        grec[3 * g + 0] = make_float4(0.f, 0.f, 0.f, 0.f);
        grec[3 * g + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        grec[3 * g + 2] = make_float4(0.f, __int_as_float(g), -3.0e38f, -3.0e38f);
        return;
    }
    const float x = xys[2 * g], y = xys[2 * g + 1];
    const float a = conics[3 * g], b = conics[3 * g + 1], c = conics[3 * g + 2];
    const float r = colors[3 * g], gg = colors[3 * g + 1], bl = colors[3 * g + 2];
    float o = opac[g];
    if (opac_is_logit) o = 1.f / (1.f + expf(-o));  // fused torch.sigmoid (sgn_splatfacto.py:949)
    // bbox of the region where alpha = o*exp(-sigma) can reach 1/255: sigma <= s = ln(255 o) (+1 % margin);
    // x half-extent sqrt(2 s c / D), y half-extent sqrt(2 s a / D), D = ac - b^2.  Used by the raster kernels
    // to skip 8x8 quadrants the Gaussian cannot touch (no valid pixel there => results unchanged).
    const float s = (o * 255.f > 0.f) ? logf(255.f * o) + 0.01f : -1.f;
    const float D = a * c - b * b;
    float ex = -1.f, ey = -1.f;
    if (s >= 0.f) {
        const bool proper = a > 0.f && c > 0.f && D > 0.f;
        ex = proper ? sqrtf(2.f * s * c / D) + 1e-3f : 3.0e38f;
        ey = proper ? sqrtf(2.f * s * a / D) + 1e-3f : 3.0e38f;
    }
    grec[3 * g + 0] = make_float4(x, y, o, 0.5f * a);
    grec[3 * g + 1] = make_float4(b, 0.5f * c, r, gg);
    grec[3 * g + 2] = make_float4(bl, __int_as_float(g), ex, ey);
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}

// pixel of (slot q, lane) inside tile (tx,ty).  block 16: four 8x8 quadrants; otherwise linear.
__device__ __forceinline__ void slot_pixel(int q, int lane, int B, int &ox, int &oy, bool &in_tile) {
    if (B == 16) {
        ox = (q & 1) * 8 + (lane & 7);
        oy = (q >> 1) * 8 + (lane >> 3);
        in_tile = true;
    } else {
        const int p = q * 64 + lane;
        ox = p % B;
        oy = p / B;
        in_tile = p < B * B;
    }
}

// Which of the tile's four 8x8 quadrants can this Gaussian touch?  Lanes 0..3 each test one quadrant
// (|centre distance| <= half-extent + 3.5 px, the half-span of the quadrant's pixel centres), the ballot
// turns the answers into a wave-uniform 4-bit mask that the per-quadrant branches test in the scalar unit.
__device__ __forceinline__ unsigned quadrant_mask(const Rec &g, float qcx, float qcy, bool enable) {
    if (!enable) return 0xFu;
    const bool hit = fabsf(g.x - qcx) <= g.ex + 3.5f && fabsf(g.y - qcy) <= g.ey + 3.5f;
    return (unsigned)(__ballot(hit) & 0xFull);
}

// Quadrant masks handed in with the list (sgn_raster_opts.ids_qmask, include/sgn_rast.h: sgn_bin_intersect with
// quadrant_masks): bits 28-31 of an id word say which quadrants the entry can touch — the exact convex test, done once
// by the emission — and the kernels neither run the box test above per entry nor evaluate the ~10 % of quadrants the
// box lets through although the ellipse misses them.  A row build_grec_kernel made inert (window passes: ex < 0) keeps
// answering "none".
constexpr int QM_SHIFT = SGN_QMASK_ID_BITS;
__device__ __forceinline__ int qm_idmask(int use_qm) { return use_qm ? (SGN_QMASK_MAX_IDS - 1) : -1; }
__device__ __forceinline__ unsigned qm_bits(int raw_id, float ex) {
    return __float_as_int(ex) < 0 ? 0u : ((unsigned)raw_id >> QM_SHIFT);
}

// Batched path: every lane holds ONE row of the batch and tests all four quadrants for it (lane-parallel over
// 64 entries instead of once per entry), so entries that cannot touch this wave's pixels are never visited.
__device__ __forceinline__ unsigned row_quadrants(float gx, float gy, float ex, float ey, int tile_x0, int tile_y0,
                                                  bool enable) {
    if (!enable) return 0xFu;
    unsigned m = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float cx = (float)(tile_x0 + (q & 1) * 8) + 4.0f, cy = (float)(tile_y0 + (q >> 1) * 8) + 4.0f;
        if (fabsf(gx - cx) <= ex + 3.5f && fabsf(gy - cy) <= ey + 3.5f) m |= 1u << q;
    }
    return m;
}

// `recs` is the per-Gaussian row table and `ids` the sorted id list; the wave chases ids[k] -> row with two dependent
// scalar loads, the id two records ahead and the row one record ahead, so only the (tile, Gaussian) pairs that are
// actually walked before the tile terminates are ever fetched (12 % of them on the benchmark scene).
// QPW = quadrants per wave: 4 -> one wave owns the whole 16x16 tile (fewest scalar fetches and, in the
// backward, one reduction per (tile, Gaussian)); 1 -> four waves per tile, one 8x8 quadrant each (4x
// shorter critical path per tile: for scenes whose longest depth lists dominate the kernel's tail).
// ADAPT (backward only; with QPW = 4, grid = 4 waves per tile): a tile whose reverse walk is shorter than `adapt_thresh`
// is done by its wave 0 alone (the other three exit at once); a longer one is split, one quadrant per wave.  (The forward
// runs the packed two-waves-per-tile kernel on 16x16 tiles and QPW = 1 elsewhere and for its longest lists.)
// DEPTH (r03): a FOURTH accumulated channel, D = sum depth_g * vis — the image the reference gets from a whole second
// rasterization of `depths.repeat(1, 3)` (sgn_splatfacto.py:982-994) — for one more fma per evaluated pair.  The
// per-Gaussian depth is not part of the 48-byte row: it comes from `depths[id]` with one more scalar load next to the
// row's (scalar-chase path) or rides in a 64-float side array of the LDS stage (batched path); same operation order
// as the colour channels, so in exact-exp mode D is bit-equal to channel 0 of the two-pass result.
// GROUPS (r04): TWO MORE accumulations ride on the walk — the alpha images of the passes that would render only the
// Gaussians with id < split (the "head" group) and only those with id >= split (the "tail" group) of the same list:
// the scene graph's background-only and objects-only accumulation passes (sgn_splatfacto_scene_graph.py:364-366), which
// the reference pays two more rasterizations for.  Each entry belongs to exactly one group (its id: wave-uniform), and
// its alpha — the expensive part — is the one the main pass evaluates anyway; per entry the group costs one more
// transmittance recursion (nT = T (1 - a), the 1e-4 stop, the last index), with the arithmetic and the bookkeeping of
// the single pass, so T / final index per group are BIT-EQUAL to the separate pass's.  A group's final index is recorded
// in the index space its backward will walk: positions of the shared list, or — when the group has its own compacted
// list (sgn_list_window: the small group) — positions of THAT list (its tile's first position + the number of the
// group's entries met before).  The shared walk goes on while the main pass or a group WITHOUT its own list is alive;
// a group with its own list that is still alive then (a few objects in front of a saturated background never finish)
// continues on its own list from where the shared walk left it — walking the shared list to its end for it would
// visit ten times the entries the separate passes do (measured: 394 instead of 549 images/s on the scene graph).
// (Few pointers on purpose: the scalar-chase walk keeps two 48-byte rows in SGPRs; with ten group pointers live as
// well the GROUPS kernels spilled 33 SGPRs, part of them to scratch memory.)
struct FwdGroups {
    int split;                  // ids >= split: tail group
    int own;                    // the group that has its own compacted list: -1 none, 0 head, 1 tail
    const int2 *own_bins;       // its bins and ids (own >= 0)
    const int32_t *own_ids;
    float *state;               // [4][H*W]: T_head, T_tail, then (int32) final index head, tail
    int32_t *kmax;              // [2][n_tiles*2] deepest composited position per tile, head then tail (zero-filled by the caller)
};

template <bool EXACT, int QPW, bool DEPTH, bool GROUPS = false>
__device__ __forceinline__ void raster_fwd_tile(int tile, int wv, int W, int H, int B, int tiles_x,
                                                const int2 *__restrict__ bins, const Rec *__restrict__ recs,
                                                const int32_t *__restrict__ ids, const float *__restrict__ bg,
                                                float *__restrict__ out_img, float *__restrict__ final_T,
                                                int32_t *__restrict__ final_idx, int batch_thresh,
                                                int32_t *__restrict__ tile_kmax, float4 (*stage)[64 * 3],
                                                const float *__restrict__ depths = nullptr,
                                                float *__restrict__ out_depth = nullptr,
                                                float (*stage_d)[64] = nullptr, int use_qm = 0,
                                                const FwdGroups *Gp = nullptr) {
    const int lane = threadIdx.x;
    const int2 range = bins[tile];
    const bool qm_on = use_qm != 0 && B == 16;      // wave-uniform
    const int idmask = qm_idmask(use_qm);
    const int q0 = wv * QPW;                   // first quadrant (pixel slot) of this wave
    constexpr int qlo = 0, qhi = QPW;          // active slots of this wave
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];  // wave-uniform -> scalar loads

    // Per-pixel state; "this pixel is finished (or outside the image)" is carried in the SIGN of T
    // (T > 0 <=> still compositing), so liveness costs one v_cmp and no mask bookkeeping in VGPRs.
    float px[QPW], py[QPW], T[QPW], C0[QPW], C1[QPW], C2[QPW], Dq[DEPTH ? QPW : 1];
    int last[QPW], pix[QPW];
    bool inside[QPW];
#pragma unroll
    for (int q = 0; q < QPW; ++q) {
        int ox, oy;
        bool in_tile;
        slot_pixel(q0 + q, lane, B, ox, oy, in_tile);
        const int j = tx * B + ox, i = ty * B + oy;
        inside[q] = in_tile && j < W && i < H && q >= qlo && q < qhi;
        pix[q] = i * W + j;
        px[q] = (float)j + 0.5f;
        py[q] = (float)i + 0.5f;
        T[q] = inside[q] ? 1.f : -1.f;
        C0[q] = 0.f; C1[q] = 0.f; C2[q] = 0.f;
        if constexpr (DEPTH) Dq[q] = 0.f;
        last[q] = 0;
    }
    // GROUPS: per-pixel state of the two group passes (same liveness convention), entries of each group met so far
    float TH[GROUPS ? QPW : 1], TT[GROUPS ? QPW : 1];
    int lastH[GROUPS ? QPW : 1], lastT[GROUPS ? QPW : 1];
    int cO = 0, baseO = 0, split = 0, own = -1;   // own: the group with its own list; cO: its entries met so far
    int rO = -1;                // where the shared walk stopped, in entries of the own group (-1: at the end of the list)
    if constexpr (GROUPS) {
#pragma unroll
        for (int q = 0; q < QPW; ++q) { TH[q] = TT[q] = T[q]; lastH[q] = lastT[q] = 0; }
        split = Gp->split;
        own = Gp->own;
        if (own >= 0) baseO = Gp->own_bins[tile].x;
    }

    // centre of quadrant (lane & 3)'s pixel centres, for the quadrant-reject test (16x16 tiles only)
    const bool qtest = (B == 16);
    const float qcx = (float)(tx * 16 + (lane & 1) * 8) + 4.0f, qcy = (float)(ty * 16 + ((lane >> 1) & 1) * 8) + 4.0f;

    // one depth-list entry for this wave's pixels; returns false once every pixel of the wave is finished
    // (GROUPS: `tail` says which group the entry belongs to, `kg` is its position in that group's index space)
    int n_eval = 0;   // (entry, quadrant) pairs this wave evaluated (wave-uniform): the backward's size-of-splat hint
    auto entry = [&](const Rec &cur, int k, unsigned qm, float dep, bool tail, int kg, bool own_walk) __attribute__((always_inline)) -> bool {
        unsigned long long live[QPW], liveg[QPW], any_live = 0ull;
#pragma unroll
        for (int q = 0; q < QPW; ++q) {
            live[q] = __ballot(T[q] > 0.f);
            liveg[q] = 0ull;
            if constexpr (GROUPS) {
                liveg[q] = __ballot((tail ? TT[q] : TH[q]) > 0.f);
                // the shared walk is kept alive by the main pass and by the groups that have no list of their own; a
                // group's walk of its own list by that group alone (the main pass takes no part in it)
                if (own_walk) { live[q] = 0ull; any_live |= liveg[q]; }
                else any_live |= live[q] | __ballot((own != 0 && TH[q] > 0.f) || (own != 1 && TT[q] > 0.f));
            } else {
                any_live |= live[q];
            }
        }
        if (any_live == 0ull) return false;
#pragma unroll
        for (int q = 0; q < QPW; ++q) {
            if ((live[q] | liveg[q]) == 0ull || !((qm >> (q0 + q)) & 1u)) continue;  // wave-uniform
            if (live[q] != 0ull) ++n_eval;
            const float dx = cur.x - px[q], dy = cur.y - py[q];
            float s = (cur.ha * dx) * dx;
            s = fmaf(cur.hc * dy, dy, s);
            const float sigma = fmaf(cur.b * dx, dy, s);
            const float alpha = fminf(0.999f, cur.opac * sgn_exp<EXACT>(-sigma));
            const bool valid = !(GROUPS && own_walk) && T[q] > 0.f && sigma >= 0.f && alpha >= (1.f / 255.f);
            // branch-free update: lanes that skip or stop add vis = 0 (fma(c, 0, C) == C exactly)
            const float nT = T[q] * (1.f - alpha);
            const bool stop = valid && nT <= 1e-4f;
            const bool acc = valid && !stop;
            const float vis = acc ? alpha * T[q] : 0.f;
            C0[q] = fmaf(cur.r, vis, C0[q]);
            C1[q] = fmaf(cur.g, vis, C1[q]);
            C2[q] = fmaf(cur.bl, vis, C2[q]);
            if constexpr (DEPTH) Dq[q] = fmaf(dep, vis, Dq[q]);
            last[q] = acc ? k : last[q];
            const float Tk = acc ? nT : T[q];
            T[q] = stop ? -Tk : Tk;  // terminating Gaussian is NOT composited; T keeps its last value
            if constexpr (GROUPS) {   // the entry's own group: the same recursion on the group's transmittance
                const float Tg = tail ? TT[q] : TH[q];
                const bool validg = Tg > 0.f && sigma >= 0.f && alpha >= (1.f / 255.f);
                const float nTg = Tg * (1.f - alpha);
                const bool stopg = validg && nTg <= 1e-4f;
                const bool accg = validg && !stopg;
                const float Tkg = accg ? nTg : Tg;
                const float Tng = stopg ? -Tkg : Tkg;
                if (tail) { TT[q] = Tng; lastT[q] = accg ? kg : lastT[q]; }
                else { TH[q] = Tng; lastH[q] = accg ? kg : lastH[q]; }
            }
        }
        return true;
    };
    // GROUPS: group and group-space position of list position k holding raw id word `raw` (wave-uniform)
    auto group_of = [&](int raw, int k, bool &tail, int &kg) __attribute__((always_inline)) {
        tail = false; kg = k;
        if constexpr (GROUPS) {
            tail = (raw & idmask) >= split;
            if (own == (int)tail) { kg = baseO + cO; ++cO; }
        }
    };

    const int L = range.y - range.x;
    if (L > 0 && L < batch_thresh) {
        // short list: chase ids -> rows with scalar loads, one entry ahead (operands arrive in SGPRs)
        int idc = ids[range.x];                         // raw id word (mask bits included)
        Rec cur = recs[idc & idmask];
        float dcur = 0.f;
        if constexpr (DEPTH) dcur = depths[idc & idmask];
        int idn = ids[min(range.x + 1, range.y - 1)];
        for (int k = range.x; k < range.y; ++k) {
            const Rec nxt = recs[idn & idmask];  // scalar prefetch of the next record
            float dnxt = 0.f;
            if constexpr (DEPTH) dnxt = depths[idn & idmask];
            const int idnn = idn;
            idn = ids[min(k + 2, range.y - 1)];
            const unsigned qm = qm_on ? qm_bits(idc, cur.ex) : quadrant_mask(cur, qcx, qcy, qtest);
            bool tail; int kg;
            group_of(idc, k, tail, kg);
            if (!entry(cur, k, qm, dcur, tail, kg, false)) {
                if constexpr (GROUPS) rO = (own == (int)tail) ? cO - 1 : cO;   // this entry is still to come
                break;
            }
            cur = nxt;
            dcur = dnxt;
            idc = idnn;
        }
    } else if (L > 0) {
        // long list: the one-entry scalar look-ahead leaves a lone wave latency-bound (a dependent id -> row
        // load pair per entry), so stage 64-entry batches through wave-private LDS instead: lane l gathers
        // row ids[k0+l] with vector loads a whole batch ahead, entries are then read back with broadcast
        // ds_read_b128 (no barrier: the workgroup is this one wave and its LDS ops retire in order).
        const int nb = (L + 63) >> 6;
        float rd = 0.f;      // DEPTH: the depth of this lane's row of the batch in flight
        int rid = 0;         // raw id word of this lane's row (quadrant bits on top)
        auto fetch = [&](int bidx, float4 &r0, float4 &r1, float4 &r2) __attribute__((always_inline)) {
            const int k = range.x + (bidx << 6) + lane;
            if (k < range.y) {
                rid = ids[k];
                const int id = rid & idmask;
                const float4 *p = reinterpret_cast<const float4 *>(recs + id);
                r0 = p[0]; r1 = p[1]; r2 = p[2];
                if constexpr (DEPTH) rd = depths[id];
            }
        };
        auto row_mask = [&](const float4 &r0, const float4 &r2) __attribute__((always_inline)) -> unsigned {
            return qm_on ? qm_bits(rid, r2.z) : row_quadrants(r0.x, r0.y, r2.z, r2.w, tx * 16, ty * 16, qtest);
        };
        // this wave's quadrants as a bit mask (all four, or the single one of a split tile)
        unsigned mine_q = 0u;
#pragma unroll
        for (int q = 0; q < QPW; ++q)
            if (q >= qlo && q < qhi) mine_q |= 1u << (q0 + q);
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
        fetch(0, r0, r1, r2);
        stage[0][lane * 3 + 0] = r0; stage[0][lane * 3 + 1] = r1; stage[0][lane * 3 + 2] = r2;
        if constexpr (DEPTH) stage_d[0][lane] = rd;
        unsigned qrow = row_mask(r0, r2) & mine_q;
        bool trow = GROUPS && (rid & idmask) >= split;    // GROUPS: this lane's row belongs to the tail group
        bool go = true;
        for (int bi = 0; bi < nb && go; ++bi) {
            const int cnt = min(64, L - (bi << 6));
            unsigned long long todo = __ballot(lane < cnt && qrow != 0u);  // entries that can touch my pixels
            const unsigned long long tmask = GROUPS ? __ballot(lane < cnt && trow) : 0ull;   // the batch's tail entries
            const unsigned long long omask =       // ... and the entries of the group that has its own list
                !GROUPS || own < 0 ? 0ull : (own == 1 ? tmask : (__ballot(lane < cnt) & ~tmask));
            const unsigned qcur = qrow;
            if (bi + 1 < nb) fetch(bi + 1, r0, r1, r2);   // next batch: in flight while this one is composited
            const float4 *sb = stage[bi & 1];
            while (todo) {
                const int j = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                Rec cur;
                const float4 a0 = sb[j * 3 + 0], a1 = sb[j * 3 + 1], a2 = sb[j * 3 + 2];
                cur.x = a0.x; cur.y = a0.y; cur.opac = a0.z; cur.ha = a0.w;
                cur.b = a1.x; cur.hc = a1.y; cur.r = a1.z; cur.g = a1.w;
                cur.bl = a2.x; cur.gid = __float_as_int(a2.y); cur.ex = a2.z; cur.ey = a2.w;
                const unsigned qm = (unsigned)__builtin_amdgcn_readlane((int)qcur, j);
                float dep = 0.f;
                if constexpr (DEPTH) dep = stage_d[bi & 1][j];
                const int k = range.x + (bi << 6) + j;
                bool tail = false; int kg = k;
                if constexpr (GROUPS) {     // entries of the batch before j that belong to the own-list group
                    tail = (tmask >> j) & 1ull;
                    if ((omask >> j) & 1ull) kg = baseO + cO + __popcll(omask & ((1ull << j) - 1ull));
                }
                if (!entry(cur, k, qm, dep, tail, kg, false)) {
                    if constexpr (GROUPS) rO = cO + __popcll(omask & ((1ull << j) - 1ull));
                    go = false;
                    break;
                }
            }
            if constexpr (GROUPS) cO += __popcll(omask);
            if (go && bi + 1 < nb) {                      // first use of the prefetched registers
                float4 *sn = stage[(bi + 1) & 1];
                sn[lane * 3 + 0] = r0; sn[lane * 3 + 1] = r1; sn[lane * 3 + 2] = r2;
                if constexpr (DEPTH) stage_d[(bi + 1) & 1][lane] = rd;
                qrow = row_mask(r0, r2) & mine_q;
                trow = GROUPS && (rid & idmask) >= split;
            }
        }
    }
    if constexpr (GROUPS) {
        // a group with its own list that outlived the shared walk: the rest of ITS list (scalar chase, one entry ahead)
        auto own_rest = [&](bool tail, const int2 *gbins, const int32_t *gids, int from) __attribute__((always_inline)) {
            const int end = gbins[tile].y;
            if (from >= end) return;
            int idc = gids[from];
            Rec cur = recs[idc & idmask];
            int idn = gids[min(from + 1, end - 1)];
            for (int p = from; p < end; ++p) {
                const Rec nxt = recs[idn & idmask];
                const int idnn = idn;
                idn = gids[min(p + 2, end - 1)];
                const unsigned qm = qm_on ? qm_bits(idc, cur.ex) : quadrant_mask(cur, qcx, qcy, qtest);
                if (!entry(cur, 0, qm, 0.f, tail, p, true)) break;
                cur = nxt;
                idc = idnn;
            }
        };
        if (rO < 0) rO = cO;
        if (own >= 0) own_rest(own == 1, Gp->own_bins, Gp->own_ids, baseO + rO);
        const int HW = W * H, n_tiles2 = 2 * tiles_x * ((H + B - 1) / B);
        float *gT = Gp->state;
        int32_t *gI = reinterpret_cast<int32_t *>(Gp->state + 2 * (size_t)HW);
        int lh = 0, lt = 0;
#pragma unroll
        for (int q = 0; q < QPW; ++q) {
            lh = max(lh, inside[q] ? lastH[q] : 0);
            lt = max(lt, inside[q] ? lastT[q] : 0);
            if (inside[q]) {
                gT[pix[q]] = fabsf(TH[q]); gI[pix[q]] = lastH[q];
                gT[HW + pix[q]] = fabsf(TT[q]); gI[HW + pix[q]] = lastT[q];
            }
        }
        lh = wave_max_i(lh); lt = wave_max_i(lt);
        if (lane == 0 && lh > 0) atomicMax(Gp->kmax + 2 * tile, lh);
        if (lane == 0 && lt > 0) atomicMax(Gp->kmax + n_tiles2 + 2 * tile, lt);
    }
    if (tile_kmax) {   // [2t]: deepest list position any pixel of the tile composited (the backward's walk starts
                       // there); [2t + 1]: (entry, quadrant) pairs evaluated: pairs / walk ~ 1 means small splats
        int lm = 0;
#pragma unroll
        for (int q = 0; q < QPW; ++q) lm = max(lm, inside[q] ? last[q] : 0);
        lm = wave_max_i(lm);
        if (lane == 0 && lm > 0) atomicMax(tile_kmax + 2 * tile, lm);
        if (lane == 0 && n_eval > 0) atomicAdd(tile_kmax + 2 * tile + 1, n_eval);
    }
#pragma unroll
    for (int q = 0; q < QPW; ++q) {
        if (inside[q]) {
            const float Tq = fabsf(T[q]);
            final_T[pix[q]] = Tq;
            final_idx[pix[q]] = last[q];
            out_img[3 * pix[q] + 0] = fmaf(Tq, bg0, C0[q]);
            out_img[3 * pix[q] + 1] = fmaf(Tq, bg1, C1[q]);
            out_img[3 * pix[q] + 2] = fmaf(Tq, bg2, C2[q]);
            if constexpr (DEPTH) out_depth[pix[q]] = Dq[q];
        }
    }
}

template <bool EXACT, int QPW, bool DEPTH>
__global__ __launch_bounds__(64) void raster_fwd_kernel(int W, int H, int B, int tiles_x,
                                                        const int2 *__restrict__ bins,
                                                        const Rec *__restrict__ recs,
                                                        const int32_t *__restrict__ ids,
                                                        const float *__restrict__ bg, float *__restrict__ out_img,
                                                        float *__restrict__ final_T,
                                                        int32_t *__restrict__ final_idx,
                                                        int batch_thresh, const int32_t *__restrict__ tile_order,
                                                        int32_t *__restrict__ tile_kmax,
                                                        const float *__restrict__ depths,
                                                        float *__restrict__ out_depth,
                                                        const int32_t *__restrict__ skip_flag, int use_qm) {
    if (skip_flag != nullptr && *skip_flag == 0) return;   // the caller already holds this pass's result (sgn_depth_reuse)
    constexpr int WPT = 4 / QPW;               // waves launched per tile
    int tile = (int)(blockIdx.x / WPT);
    if (tile_order) tile = tile_order[tile];   // longest depth lists first (sgn_tile_order): no long tile starts late
    const int wv = (int)(blockIdx.x % WPT);
    __shared__ float4 stage[2][64 * 3];         // 64-entry batches of the long-list path (wave-private)
    __shared__ float stage_d[DEPTH ? 2 : 1][64];
    raster_fwd_tile<EXACT, QPW, DEPTH>(tile, wv, W, H, B, tiles_x, bins, recs, ids, bg, out_img, final_T, final_idx,
                                       batch_thresh, tile_kmax, stage, depths, out_depth, stage_d, use_qm);
}

// ---------------------------------------------------------------- forward, packed-FP32 form (16x16 tiles)
// Two waves per tile, each owns a 16x8 half: lane l carries the pixel at (l & 7, l >> 3) of the half's LEFT 8x8
// quadrant and the one 8 columns to the right (slot 0 / slot 1) as a register pair, and every entry that can
// touch the half is evaluated for both with v_pk_add/mul/fma_f32 (operands of the Gaussian stay wave-uniform:
// SGPRs on the scalar-chase path, broadcast LDS reads on the batched one, selected through op_sel).
// What this buys on gfx950 is NOT cheaper arithmetic: a packed FP32 instruction costs the cycles of two single
// ones here (profiles/microbench/valu_rates.hip: v_pk_fma_f32 5.0 cycles per wave-instruction, v_fma_f32 2.5-2.9).
// It halves what is paid per ENTRY and per WAVE rather than per pixel — liveness ballots, list / loop control in
// the scalar unit, the broadcast LDS reads (or the scalar row fetch), the index bookkeeping — because one wave
// now serves 128 pixels: SQ_INSTS_VALU 1.00e8 -> 0.80e8, SQ_INSTS_SALU 7.7e7 -> 4.4e7, LDS instructions
// 1.01e7 -> 0.52e7 per launch, 183 -> 157 us on the benchmark scene (profiles/r02k_pmc_sq*.md).
// (Tried and dropped, profiles/experiments/r02_packed_forward_notes.md: a three-way form that sends entries
// touching only one of the two quadrants down a one-slot path — the compiler joins the three paths with ~12
// register copies per entry, 194 us; and the same pairing in the backward, whose cost is the arithmetic itself.)
// The first tile_order[n_tiles] tiles of the launch order — lists of at least adapt_fwd entries, sgn_tile_order —
// are given four waves, one quadrant each (raster_fwd_tile<QPW = 1>): on street scenes most of the work sits in
// a few thousand-entry lists, and two waves per such tile leave the SIMDs short of waves (547 vs 294 us).
//
// Same arithmetic, operation by operation, as raster_fwd_tile (file header); only the bookkeeping around it is
// rearranged so that the packed part needs no "valid" mask:
//   a' = (T > 0 && sigma >= 0 && alpha >= 1/255) ? alpha : 0        (a' = 0: nT = T*1 = T, vis = 0*T = 0 exactly)
//   nT = T * (1 - a'),  stop <=> bits(nT) <= bits(1e-4)  as UNSIGNED (finished pixels carry T < 0: huge unsigned;
//                                                          a live T is always > 1e-4, so a' = 0 never stops)
//   vis = stop ? 0 : a' * T,   T = stop ? -T : nT,   last = (a' > 0 && !stop) ? k : last
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat2(float v) { return v2f{v, v}; }
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

template <bool EXACT, bool DEPTH, bool GROUPS>
__device__ __forceinline__ void raster_fwd_pk_body(int tile, int wv, bool is_long, int W, int H, int tiles_x, int n_tiles_,
                                                   const int2 *__restrict__ bins, const Rec *__restrict__ recs,
                                                   const int32_t *__restrict__ ids, const float *__restrict__ bg,
                                                   float *__restrict__ out_img, float *__restrict__ final_T,
                                                   int32_t *__restrict__ final_idx, int batch_thresh,
                                                   int32_t *__restrict__ tile_kmax, const float *__restrict__ depths,
                                                   float *__restrict__ out_depth, int use_qm, const FwdGroups &G,
                                                   float4 (*stage)[64 * 3], float (*stage_d)[64]) {
    const int2 range = bins[tile];
    const int L = range.y - range.x;
    if (is_long) {
        raster_fwd_tile<EXACT, 1, DEPTH, GROUPS>(tile, wv, W, H, 16, tiles_x, bins, recs, ids, bg, out_img, final_T,
                                                 final_idx, batch_thresh, tile_kmax, stage, depths, out_depth, stage_d,
                                                 use_qm, &G);
        return;
    }
    const bool qm_on = use_qm != 0;            // wave-uniform
    const int idmask = qm_idmask(use_qm);
    const unsigned shift_q = 2u * wv;                    // this wave's quadrants: bits shift_q, shift_q + 1
    const int lane = threadIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

    const int j0 = tx * 16 + (lane & 7), i0 = ty * 16 + wv * 8 + (lane >> 3);
    const bool in0 = j0 < W && i0 < H, in1 = j0 + 8 < W && i0 < H;
    const int pix = i0 * W + j0;
    const v2f px = {(float)j0 + 0.5f, (float)j0 + 8.5f};
    const float py = (float)i0 + 0.5f;
    v2f T = {in0 ? 1.f : -1.f, in1 ? 1.f : -1.f};
    v2f C0 = {0.f, 0.f}, C1 = C0, C2 = C0, Dp = C0;
    int last0 = 0, last1 = 0;
    const float qcx = (float)(tx * 16 + (lane & 1) * 8) + 4.0f, qcy = (float)(ty * 16 + ((lane >> 1) & 1) * 8) + 4.0f;
    // GROUPS (see FwdGroups): the two group passes' transmittances / last positions, group entries met so far
    v2f TH = T, TT = T;
    int lastH0 = 0, lastH1 = 0, lastT0 = 0, lastT1 = 0;
    const int own = GROUPS ? G.own : -1;   // the group with its own list (-1: none)
    int cO = 0, baseO = 0;      // its entries met so far, its tile's first position in that list
    int rO = -1;                // where the shared walk stopped, in entries of the own group (-1: at the end of the list)
    if constexpr (GROUPS) {
        if (own >= 0) baseO = G.own_bins[tile].x;
    }

    int n_eval = 0;
    // m: which of this wave's two quadrants the entry can touch (bit 0 = slot 0, bit 1 = slot 1), never 0
    // (GROUPS: `tail` = the entry's group, `kg` = its position in that group's index space)
    auto entry = [&](const Rec &cur, int k, unsigned m, float dep, bool tail, int kg, bool own_walk) __attribute__((always_inline)) -> bool {
        // (own_walk: a group's walk of the rest of its OWN list — the main pass takes no part in it)
        const bool l0 = !(GROUPS && own_walk) && T.x > 0.f, l1 = !(GROUPS && own_walk) && T.y > 0.f;
        const unsigned long long b0 = __ballot(l0), b1 = __ballot(l1);
        bool g0 = false, g1 = false;
        if constexpr (GROUPS) {
            const v2f Tg = tail ? TT : TH;
            g0 = Tg.x > 0.f; g1 = Tg.y > 0.f;
            const unsigned long long bg = __ballot(g0 || g1);
            if (own_walk) {
                if (bg == 0ull) return false;
            } else {
                // the shared walk is kept alive by the main pass and by the groups that have no list of their own
                if ((b0 | b1 | __ballot((own != 0 && (TH.x > 0.f || TH.y > 0.f)) ||
                                        (own != 1 && (TT.x > 0.f || TT.y > 0.f)))) == 0ull) return false;
                if ((b0 | b1 | bg) == 0ull) return true;          // neither the main pass nor this entry's group
            }
        } else {
            if ((b0 | b1) == 0ull) return false;
        }
        n_eval += __popc(m & ((b0 ? 1u : 0u) | (b1 ? 2u : 0u)));
        {
            const v2f dx = splat2(cur.x) - px;
            const float dy = cur.y - py;
            v2f s = (splat2(cur.ha) * dx) * dx;
            s = fma2(splat2(cur.hc * dy), splat2(dy), s);
            const v2f sigma = fma2(splat2(cur.b) * dx, splat2(dy), s);
            v2f e;
            if constexpr (EXACT) {
                e = v2f{exp_portable(-sigma.x), exp_portable(-sigma.y)};
            } else {
                const v2f t = sigma * splat2(-1.44269504088896341f);   // __expf(-sigma) = v_exp_f32(-sigma * log2 e)
                e = v2f{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
            }
            const v2f ao = splat2(cur.opac) * e;
            const float a0 = fminf(0.999f, ao.x), a1 = fminf(0.999f, ao.y);
            const bool ok0 = l0 && sigma.x >= 0.f && a0 >= (1.f / 255.f);
            const bool ok1 = l1 && sigma.y >= 0.f && a1 >= (1.f / 255.f);
            const v2f a = {ok0 ? a0 : 0.f, ok1 ? a1 : 0.f};
            const v2f nT = T * (splat2(1.f) - a);
            const v2f w = a * T;
            const bool stop0 = __float_as_uint(nT.x) <= 0x38D1B717u;    // 1e-4f; see the header of this kernel
            const bool stop1 = __float_as_uint(nT.y) <= 0x38D1B717u;
            const v2f vis = {stop0 ? 0.f : w.x, stop1 ? 0.f : w.y};
            C0 = fma2(splat2(cur.r), vis, C0);
            C1 = fma2(splat2(cur.g), vis, C1);
            C2 = fma2(splat2(cur.bl), vis, C2);
            if constexpr (DEPTH) Dp = fma2(splat2(dep), vis, Dp);
            last0 = (ok0 && !stop0) ? k : last0;
            last1 = (ok1 && !stop1) ? k : last1;
            T = v2f{stop0 ? -T.x : nT.x, stop1 ? -T.y : nT.y};
            if constexpr (GROUPS) {   // the entry's own group: the same recursion on the group's transmittance
                const v2f Tg = tail ? TT : TH;
                const bool okg0 = g0 && sigma.x >= 0.f && a0 >= (1.f / 255.f);
                const bool okg1 = g1 && sigma.y >= 0.f && a1 >= (1.f / 255.f);
                const v2f ag = {okg0 ? a0 : 0.f, okg1 ? a1 : 0.f};
                const v2f nTg = Tg * (splat2(1.f) - ag);
                const bool sg0 = __float_as_uint(nTg.x) <= 0x38D1B717u;
                const bool sg1 = __float_as_uint(nTg.y) <= 0x38D1B717u;
                const v2f Tn = {sg0 ? -Tg.x : nTg.x, sg1 ? -Tg.y : nTg.y};
                if (tail) {
                    TT = Tn;
                    lastT0 = (okg0 && !sg0) ? kg : lastT0;
                    lastT1 = (okg1 && !sg1) ? kg : lastT1;
                } else {
                    TH = Tn;
                    lastH0 = (okg0 && !sg0) ? kg : lastH0;
                    lastH1 = (okg1 && !sg1) ? kg : lastH1;
                }
            }
        }
        return true;
    };

    if (L > 0 && L < batch_thresh) {
        int idc = ids[range.x];                         // raw id word (quadrant bits on top)
        Rec cur = recs[idc & idmask];
        float dcur = 0.f;
        if constexpr (DEPTH) dcur = depths[idc & idmask];
        int idn = ids[min(range.x + 1, range.y - 1)];
        for (int k = range.x; k < range.y; ++k) {
            const Rec nxt = recs[idn & idmask];
            float dnxt = 0.f;
            if constexpr (DEPTH) dnxt = depths[idn & idmask];
            const int idnn = idn;
            idn = ids[min(k + 2, range.y - 1)];
            const unsigned m = ((qm_on ? qm_bits(idc, cur.ex) : quadrant_mask(cur, qcx, qcy, true)) >> shift_q) & 3u;
            bool tail = false; int kg = k;
            if constexpr (GROUPS) {
                tail = (idc & idmask) >= G.split;
                if (own == (int)tail) { kg = baseO + cO; ++cO; }
            }
            if (m != 0u && !entry(cur, k, m, dcur, tail, kg, false)) {
                if constexpr (GROUPS) rO = (own == (int)tail) ? cO - 1 : cO;   // this entry is still to come
                break;
            }
            cur = nxt;
            dcur = dnxt;
            idc = idnn;
        }
    } else if (L > 0) {
        const int nb = (L + 63) >> 6;
        float rd = 0.f;
        int rid = 0;
        auto fetch = [&](int bidx, float4 &r0, float4 &r1, float4 &r2) __attribute__((always_inline)) {
            const int k = range.x + (bidx << 6) + lane;
            if (k < range.y) {
                rid = ids[k];
                const int id = rid & idmask;
                const float4 *p = reinterpret_cast<const float4 *>(recs + id);
                r0 = p[0]; r1 = p[1]; r2 = p[2];
                if constexpr (DEPTH) rd = depths[id];
            }
        };
        auto row_mask = [&](const float4 &r0, const float4 &r2) __attribute__((always_inline)) -> unsigned {
            return qm_on ? qm_bits(rid, r2.z) : row_quadrants(r0.x, r0.y, r2.z, r2.w, tx * 16, ty * 16, true);
        };
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
        fetch(0, r0, r1, r2);
        stage[0][lane * 3 + 0] = r0; stage[0][lane * 3 + 1] = r1; stage[0][lane * 3 + 2] = r2;
        if constexpr (DEPTH) stage_d[0][lane] = rd;
        unsigned qrow = (row_mask(r0, r2) >> shift_q) & 3u;
        bool trow = GROUPS && (rid & idmask) >= G.split;      // GROUPS: this lane's row belongs to the tail group
        bool go = true;
        for (int bi = 0; bi < nb && go; ++bi) {
            const int cnt = min(64, L - (bi << 6));
            unsigned long long todo = __ballot(lane < cnt && qrow != 0u);
            const unsigned long long tmask = GROUPS ? __ballot(lane < cnt && trow) : 0ull;
            const unsigned long long omask =       // the batch's entries of the group that has its own list
                !GROUPS || own < 0 ? 0ull : (own == 1 ? tmask : (__ballot(lane < cnt) & ~tmask));
            const unsigned qcur = qrow;
            if (bi + 1 < nb) fetch(bi + 1, r0, r1, r2);
            const float4 *sb = stage[bi & 1];
            while (todo) {
                const int j = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                Rec cur;
                const float4 a0 = sb[j * 3 + 0], a1 = sb[j * 3 + 1], a2 = sb[j * 3 + 2];
                cur.x = a0.x; cur.y = a0.y; cur.opac = a0.z; cur.ha = a0.w;
                cur.b = a1.x; cur.hc = a1.y; cur.r = a1.z; cur.g = a1.w;
                cur.bl = a2.x; cur.gid = __float_as_int(a2.y); cur.ex = a2.z; cur.ey = a2.w;
                const unsigned m = (unsigned)__builtin_amdgcn_readlane((int)qcur, j);
                float dep = 0.f;
                if constexpr (DEPTH) dep = stage_d[bi & 1][j];
                const int k = range.x + (bi << 6) + j;
                bool tail = false; int kg = k;
                if constexpr (GROUPS) {
                    tail = (tmask >> j) & 1ull;
                    if ((omask >> j) & 1ull) kg = baseO + cO + __popcll(omask & ((1ull << j) - 1ull));
                }
                if (!entry(cur, k, m, dep, tail, kg, false)) {
                    if constexpr (GROUPS) rO = cO + __popcll(omask & ((1ull << j) - 1ull));
                    go = false;
                    break;
                }
            }
            if constexpr (GROUPS) cO += __popcll(omask);
            if (go && bi + 1 < nb) {
                float4 *sn = stage[(bi + 1) & 1];
                sn[lane * 3 + 0] = r0; sn[lane * 3 + 1] = r1; sn[lane * 3 + 2] = r2;
                if constexpr (DEPTH) stage_d[(bi + 1) & 1][lane] = rd;
                qrow = (row_mask(r0, r2) >> shift_q) & 3u;
                trow = GROUPS && (rid & idmask) >= G.split;
            }
        }
    }
    if constexpr (GROUPS) {
        // a group with its own list that outlived the shared walk: the rest of ITS list (scalar chase, one entry ahead)
        auto own_rest = [&](bool tail, const int2 *gbins, const int32_t *gids, int from) __attribute__((always_inline)) {
            const int end = gbins[tile].y;
            if (from >= end) return;
            int idc = gids[from];
            Rec cur = recs[idc & idmask];
            int idn = gids[min(from + 1, end - 1)];
            for (int p = from; p < end; ++p) {
                const Rec nxt = recs[idn & idmask];
                const int idnn = idn;
                idn = gids[min(p + 2, end - 1)];
                const unsigned m = ((qm_on ? qm_bits(idc, cur.ex) : quadrant_mask(cur, qcx, qcy, true)) >> shift_q) & 3u;
                if (m != 0u && !entry(cur, 0, m, 0.f, tail, p, true)) break;
                cur = nxt;
                idc = idnn;
            }
        };
        if (rO < 0) rO = cO;
        if (own >= 0) own_rest(own == 1, G.own_bins, G.own_ids, baseO + rO);
        int lh = wave_max_i(max(in0 ? lastH0 : 0, in1 ? lastH1 : 0));
        int lt = wave_max_i(max(in0 ? lastT0 : 0, in1 ? lastT1 : 0));
        if (lane == 0 && lh > 0) atomicMax(G.kmax + 2 * tile, lh);
        if (lane == 0 && lt > 0) atomicMax(G.kmax + 2 * n_tiles_ + 2 * tile, lt);
        const int HW = W * H;
        float *gT = G.state;
        int32_t *gI = reinterpret_cast<int32_t *>(G.state + 2 * (size_t)HW);
        if (in0) {
            gT[pix] = fabsf(TH.x); gI[pix] = lastH0;
            gT[HW + pix] = fabsf(TT.x); gI[HW + pix] = lastT0;
        }
        if (in1) {
            gT[pix + 8] = fabsf(TH.y); gI[pix + 8] = lastH1;
            gT[HW + pix + 8] = fabsf(TT.y); gI[HW + pix + 8] = lastT1;
        }
    }
    if (tile_kmax) {
        int lm = max(in0 ? last0 : 0, in1 ? last1 : 0);
        lm = wave_max_i(lm);
        if (lane == 0 && lm > 0) atomicMax(tile_kmax + 2 * tile, lm);
        if (lane == 0 && n_eval > 0) atomicAdd(tile_kmax + 2 * tile + 1, n_eval);
    }
    if (in0) {
        const float Tq = fabsf(T.x);
        final_T[pix] = Tq;
        final_idx[pix] = last0;
        out_img[3 * pix + 0] = fmaf(Tq, bg0, C0.x);
        out_img[3 * pix + 1] = fmaf(Tq, bg1, C1.x);
        out_img[3 * pix + 2] = fmaf(Tq, bg2, C2.x);
        if constexpr (DEPTH) out_depth[pix] = Dp.x;
    }
    if (in1) {
        const float Tq = fabsf(T.y);
        final_T[pix + 8] = Tq;
        final_idx[pix + 8] = last1;
        out_img[3 * pix + 24] = fmaf(Tq, bg0, C0.y);
        out_img[3 * pix + 25] = fmaf(Tq, bg1, C1.y);
        out_img[3 * pix + 26] = fmaf(Tq, bg2, C2.y);
        if constexpr (DEPTH) out_depth[pix + 8] = Dp.y;
    }
}

template <bool EXACT, bool DEPTH, bool GROUPS = false>
__global__ __launch_bounds__(64) void raster_fwd_pk_kernel(int W, int H, int tiles_x, int n_tiles_,
                                                           const int2 *__restrict__ bins,
                                                           const Rec *__restrict__ recs,
                                                           const int32_t *__restrict__ ids,
                                                           const float *__restrict__ bg, float *__restrict__ out_img,
                                                           float *__restrict__ final_T,
                                                           int32_t *__restrict__ final_idx,
                                                           int batch_thresh, const int32_t *__restrict__ tile_order,
                                                           int32_t *__restrict__ tile_kmax,
                                                           const float *__restrict__ depths,
                                                           float *__restrict__ out_depth,
                                                           const int32_t *__restrict__ skip_flag, int use_qm,
                                                           const FwdGroups G) {
    if (skip_flag != nullptr && *skip_flag == 0) return;   // the caller already holds this pass's result (sgn_depth_reuse)
    __shared__ float4 stage[2][64 * 3];
    __shared__ float stage_d[DEPTH ? 2 : 1][64];
    // The first n_long tiles of the launch order (sgn_tile_order with long_thresh: the longest lists) get four waves
    // each, the others two.  Blocks come in groups of eight tiles (8 x 4, then 8 x 2 blocks): block b runs on XCD
    // b % 8, so a tile's waves share an XCD (and its L2) and are dispatched together, longest tiles first.  No block
    // inside the two regions is empty: interleaving waves that exit at once with working ones left half of the SIMDs
    // idle (the dispatcher places consecutive workgroups round-robin), 267 vs 159 us on the benchmark scene.
    const int n_long = tile_order ? min(max(tile_order[n_tiles_], 0), n_tiles_) : 0;
    const int b_long = ((n_long + 7) >> 3) << 5;
    const int b = (int)blockIdx.x;
    int t_idx, wv;
    const bool is_long = b < b_long;
    if (is_long) {
        t_idx = (b >> 5) * 8 + (b & 7);
        wv = (b >> 3) & 3;
        if (t_idx >= n_long) return;
    } else {
        const int b2 = b - b_long;
        t_idx = n_long + (b2 >> 4) * 8 + (b2 & 7);
        wv = (b2 >> 3) & 1;
        if (t_idx >= n_tiles_) return;
    }
    int tile = t_idx;
    if (tile_order) tile = tile_order[tile];
    if constexpr (GROUPS) {
        // A tile that holds NO entry of the group with its own list (most tiles: the objects cover a part of the image)
        // needs no group arithmetic at all: the other group's pass IS the main pass there — same list, same arithmetic —
        // and the own-list group's pass is empty.  Such tiles run the plain walk and copy its per-pixel state.
        bool mixed = true;
        if (G.own >= 0) {
            const int2 ob = G.own_bins[tile];
            mixed = ob.y > ob.x;
        }
        if (mixed) {
            raster_fwd_pk_body<EXACT, DEPTH, true>(tile, wv, is_long, W, H, tiles_x, n_tiles_, bins, recs, ids, bg,
                                                           out_img, final_T, final_idx, batch_thresh, tile_kmax, depths,
                                                           out_depth, use_qm, G, stage, stage_d);
            return;
        }
        raster_fwd_pk_body<EXACT, DEPTH, false>(tile, wv, is_long, W, H, tiles_x, n_tiles_, bins, recs, ids, bg,
                                                        out_img, final_T, final_idx, batch_thresh, tile_kmax, depths,
                                                        out_depth, use_qm, G, stage, stage_d);
        const int lane = threadIdx.x, tx = tile % tiles_x, ty = tile / tiles_x, HW = W * H;
        const int other = 1 - G.own;
        float *gT = G.state;
        int32_t *gI = reinterpret_cast<int32_t *>(G.state + 2 * (size_t)HW);
        int lm = 0;
        auto copy_px = [&](int j, int i) __attribute__((always_inline)) {
            if (j < W && i < H) {
                const int pix = i * W + j;
                const int im = final_idx[pix];          // written by this very lane a moment ago
                gT[other * HW + pix] = final_T[pix]; gI[other * HW + pix] = im;
                gT[G.own * HW + pix] = 1.f; gI[G.own * HW + pix] = 0;
                lm = max(lm, im);
            }
        };
        if (is_long) {                                  // four waves, one 8x8 quadrant each
            copy_px(tx * 16 + (wv & 1) * 8 + (lane & 7), ty * 16 + (wv >> 1) * 8 + (lane >> 3));
        } else {                                        // two waves, a 16x8 half each
            copy_px(tx * 16 + (lane & 7), ty * 16 + wv * 8 + (lane >> 3));
            copy_px(tx * 16 + 8 + (lane & 7), ty * 16 + wv * 8 + (lane >> 3));
        }
        lm = wave_max_i(lm);
        if (lane == 0 && lm > 0) atomicMax(G.kmax + other * 2 * n_tiles_ + 2 * tile, lm);
    } else {
        raster_fwd_pk_body<EXACT, DEPTH, false>(tile, wv, is_long, W, H, tiles_x, n_tiles_, bins, recs, ids, bg,
                                                        out_img, final_T, final_idx, batch_thresh, tile_kmax, depths,
                                                        out_depth, use_qm, G, stage, stage_d);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Transposed wave reduction of the nine per-Gaussian partial gradients (REDUCE == 1).  gfx950's
// v_permlane32_swap / v_permlane16_swap exchange half-waves / odd-even rows BETWEEN two registers, so one swap +
// one add folds two values at once: after the 32- and 16-lane stages three registers hold, per 16-lane row,
// t0 = [v0 v2 v1 v3], t1 = [v4 v6 v5 v7], t2 = [v8 0 0 0]; four DPP adds finish each row.  28 cross-lane
// ops instead of 54 shuffles.
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum_dpp(float v) {   // every lane ends with its 16-lane row total
    v += dpp_get<0xB1>(v);    // quad_perm:[1,0,3,2]
    v += dpp_get<0x4E>(v);    // quad_perm:[2,3,0,1]
    v += dpp_get<0x141>(v);   // row_half_mirror
    v += dpp_get<0x140>(v);   // row_mirror
    return v;
}
__device__ __forceinline__ float fold32(float a, float b) {   // lanes 0-31: a[l]+a[l+32]; lanes 32-63: same of b
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float fold16(float a, float b) {   // rows: [a.r0+a.r1, b.r0+b.r1, a.r2+a.r3, b.r2+b.r3]
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// grad_ws row layout (12 floats / Gaussian): 0,1 m_x, m_y | 2,3,4 s_xx, s_xy, s_yy (raw moments; the conic is applied by
// unpack_grads_kernel) | 5,6,7 v_rgb | 8 v_opacity
template <bool EXACT, int REDUCE, int QPW, bool ADAPT>
__device__ __forceinline__ void raster_bwd_tile(int tile, int wv, int W, int H, int B, int tiles_x,
                                                const int2 *__restrict__ bins, const Rec *__restrict__ recs,
                                                const int32_t *__restrict__ ids, const float *__restrict__ bg,
                                                const float *__restrict__ final_T,
                                                const int32_t *__restrict__ final_idx,
                                                const float *__restrict__ v_out,
                                                const float *__restrict__ v_out_alpha, float alpha_clamp,
                                                float *__restrict__ grad_ws, int dbg, int adapt_thresh,
                                                int batch_thresh, int use_qm) {
    static_assert(!ADAPT || QPW == 4, "adaptive splitting starts from the 4-quadrant wave");
    const bool qm_on = use_qm != 0 && B == 16;      // wave-uniform
    const int idmask = qm_idmask(use_qm);
    const int q0 = ADAPT ? 0 : wv * QPW;
    const int lane = threadIdx.x;
    const int2 range = bins[tile];
    if (range.x >= range.y) return;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

    // Per-pixel state.  Algebra: with dotc = sum_c color_c * v_out_c (per pixel x Gaussian)
    // and bv = sum_c buffer_c * v_out_c (maintained incrementally: bv += fac * dotc), upstream's
    //   v_alpha = sum_c (color_c*T - buffer_c*ra) * v_out_c + T_final*ra*v_out_alpha - T_final*ra*sum_c bg_c*v_out_c
    // becomes  v_alpha = ra * (T_before * dotc - bv + c0),  c0 = T_final * (v_out_alpha - sum_c bg_c*v_out_c),
    // i.e. 3 VALU ops instead of ~26, one running scalar instead of a 3-channel buffer.
    float px[QPW], py[QPW], T[QPW], c0[QPW], bv[QPW], vo0[QPW], vo1[QPW], vo2[QPW];
    int kfin[QPW], pixi[QPW];
    int kmax_l = -1;
#pragma unroll
    for (int q = 0; q < QPW; ++q) {          // phase 1: where does each pixel's walk start?
        int ox, oy;
        bool in_tile;
        slot_pixel(q0 + q, lane, B, ox, oy, in_tile);
        const int j = tx * B + ox, i = ty * B + oy;
        const bool inside = in_tile && j < W && i < H;
        pixi[q] = inside ? i * W + j : -1;
        px[q] = (float)j + 0.5f;
        py[q] = (float)i + 0.5f;
        kfin[q] = inside ? final_idx[pixi[q]] : -1;  // -1: this slot never participates
        kmax_l = max(kmax_l, kfin[q]);
    }
    int kmax = __builtin_amdgcn_readfirstlane(wave_max_i(kmax_l));
    kmax = min(kmax, range.y - 1);
    if (kmax < range.x) return;
    if constexpr (ADAPT) {                   // split tiles whose reverse walk is long, one quadrant per wave
        const bool split = (kmax - range.x + 1) >= adapt_thresh;
        if (!split && wv != 0) return;
        if (split) {
            int km = -1;
#pragma unroll
            for (int q = 0; q < QPW; ++q) {
                if (q != wv) { kfin[q] = -1; pixi[q] = -1; } else km = kfin[q];
            }
            kmax = min(__builtin_amdgcn_readfirstlane(wave_max_i(km)), range.y - 1);
            if (kmax < range.x) return;
        }
    }
#pragma unroll
    for (int q = 0; q < QPW; ++q) {          // phase 2: per-pixel state of the active slots
        const bool inside = pixi[q] >= 0;
        const int pix = inside ? pixi[q] : 0;
        const float Tf = inside ? final_T[pix] : 1.f;
        T[q] = Tf;
        // A pixel nothing was composited onto (its transmittance stayed EXACTLY 1: the first composited entry makes it
        // <= 254/255) takes no part in the reverse walk; upstream never reads its incoming gradients.  The body below is
        // branch-free — a masked lane contributes `0 * v_out` — so a non-finite value there would reach the sums as
        // 0 x NaN.  Such values are ordinary: `depth / alpha` under a `where(alpha > eps, ...)` (sgn_splatfacto.py:995)
        // has gradient 0 / 0 on every uncovered pixel.  Their gradients are therefore not loaded at all (round 5).
        const bool live = inside && Tf < 1.f;
        // v_out == NULL: the image took no part in the loss (an accumulation-only pass): zeros, and no 29 MB read
        vo0[q] = (live && v_out != nullptr) ? v_out[3 * pix] : 0.f;
        vo1[q] = (live && v_out != nullptr) ? v_out[3 * pix + 1] : 0.f;
        vo2[q] = (live && v_out != nullptr) ? v_out[3 * pix + 2] : 0.f;
        const float voa = live ? v_out_alpha[pix] : 0.f;
        c0[q] = Tf * (voa - fmaf(bg0, vo0[q], fmaf(bg1, vo1[q], bg2 * vo2[q])));
        bv[q] = 0.f;
    }

    const bool qtest = (B == 16);
    const float qcx = (float)(tx * 16 + (lane & 1) * 8) + 4.0f, qcy = (float)(ty * 16 + ((lane >> 1) & 1) * 8) + 4.0f;
    bool kfin_active[QPW];   // wave-uniform: does any pixel of this slot take part in the reverse walk?
#pragma unroll
    for (int q = 0; q < QPW; ++q) kfin_active[q] = __ballot(kfin[q] >= range.x) != 0ull;

    // one depth-list entry (reverse walk) for this wave's pixels
    auto entry = [&](const Rec &cur, int k, unsigned qm) __attribute__((always_inline)) {
        // Spatial terms as raw MOMENTS (r03): m = sum vs (dx, dy), s = sum vs (dx^2, dx dy, dy^2).  The conic that turns
        // them into v_xy = (a m_x + b m_y, b m_x + c m_y) and v_conic = (s_xx / 2, s_xy, s_yy / 2) is a per-GAUSSIAN
        // constant, so it is applied once per Gaussian in unpack_grads_kernel instead of once per (pixel, entry) here:
        // 2 mul + 2 add + 3 fma in place of 4 mul + 7 fma.
        float m_x = 0.f, m_y = 0.f, s_xx = 0.f, s_xy = 0.f, s_yy = 0.f;
        float g_r = 0.f, g_g = 0.f, g_b = 0.f, g_o = 0.f;
        bool any = false;
#pragma unroll
        for (int q = 0; q < QPW; ++q) {
            if (!((qm >> (q0 + q)) & 1u) || __ballot(k <= kfin[q]) == 0ull) continue;  // wave-uniform
            const float dx = cur.x - px[q], dy = cur.y - py[q];
            const float t1 = cur.ha * dx, t2 = cur.hc * dy, t3 = cur.b * dx;
            const float sigma = fmaf(t3, dy, fmaf(t2, dy, t1 * dx));
            const float vis = sgn_exp<EXACT>(-sigma);
            const float av = cur.opac * vis;
            const float alpha = fminf(alpha_clamp, av);
            const bool valid = (k <= kfin[q]) && sigma >= 0.f && alpha >= (1.f / 255.f);
            // branch-free: everything is evaluated, invalid lanes are masked out by three selects.  (ONE select — an
            // invalid lane continuing with vis = 0, so that alpha = 0, ra = 1, fac = 0, vs = 0 — was measured in r03:
            // 3 % SLOWER on all three workloads; the select then sits in front of the rcp on the dependency chain.)
            const float ra = __builtin_amdgcn_rcpf(1.f - alpha);   // v_rcp_f32 (1 ulp); oracle divides exactly
            const float Tn = T[q] * ra;
            const float fac = alpha * Tn;
            const float dotc = fmaf(cur.r, vo0[q], fmaf(cur.g, vo1[q], cur.bl * vo2[q]));
            const float v_alpha = ra * fmaf(T[q], dotc, c0[q] - bv[q]);
            const float vm = valid ? v_alpha : 0.f;
            const float facm = valid ? fac : 0.f;
            bv[q] = fmaf(facm, dotc, bv[q]);
            T[q] = valid ? Tn : T[q];
            const float vs = -av * vm;                              // v_sigma (upstream: not zeroed by the clamp)
            const float p = vs * dx, qq = vs * dy;
            m_x += p;
            m_y += qq;
            s_xx = fmaf(p, dx, s_xx);
            s_xy = fmaf(p, dy, s_xy);
            s_yy = fmaf(qq, dy, s_yy);
            g_r = fmaf(facm, vo0[q], g_r);
            g_g = fmaf(facm, vo1[q], g_g);
            g_b = fmaf(facm, vo2[q], g_b);
            g_o = fmaf(vis, vm, g_o);
            any = any || valid;
        }
        if (__ballot(any) != 0ull) {  // wave-uniform
            if constexpr (REDUCE == 1) {
                float t0 = fold16(fold32(m_x, m_y), fold32(s_xx, s_xy));     // rows: m_x, s_xx, m_y, s_xy
                float t1 = fold16(fold32(s_yy, g_r), fold32(g_g, g_b));      // rows: s_yy, g, r, b
                float t2 = fold16(fold32(g_o, 0.f), 0.f);                    // rows: o, 0, 0, 0
                t0 = row_sum_dpp(t0); t1 = row_sum_dpp(t1); t2 = row_sum_dpp(t2);
                const int c = lane & 15, row = lane >> 4;
                const int rowmap = ((row & 1) << 1) | (row >> 1);            // rows 0,1,2,3 hold values 0,2,1,3
                const float mine = (c == 0) ? t0 : (c == 1) ? t1 : t2;
                if ((c < 2 || lane == 2) && !(dbg & 1))
                    unsafeAtomicAdd(grad_ws + (size_t)cur.gid * SGN_RECORD_FLOATS + (c * 4 + rowmap), mine);
            } else {
                if (!(dbg & 2)) {          // dbg bit1: ablation only, skip the wave reduction (results are wrong)
                    m_x = wave_sum(m_x); m_y = wave_sum(m_y);
                    s_xx = wave_sum(s_xx); s_xy = wave_sum(s_xy); s_yy = wave_sum(s_yy);
                    g_r = wave_sum(g_r); g_g = wave_sum(g_g); g_b = wave_sum(g_b);
                    g_o = wave_sum(g_o);
                }
                float mine = m_x;
                mine = (lane == 1) ? m_y : mine;
                mine = (lane == 2) ? s_xx : mine;
                mine = (lane == 3) ? s_xy : mine;
                mine = (lane == 4) ? s_yy : mine;
                mine = (lane == 5) ? g_r : mine;
                mine = (lane == 6) ? g_g : mine;
                mine = (lane == 7) ? g_b : mine;
                mine = (lane == 8) ? g_o : mine;
                if (lane < 9 && !(dbg & 1))  // dbg bit0: ablation, no atomics
                    unsafeAtomicAdd(grad_ws + (size_t)cur.gid * SGN_RECORD_FLOATS + lane, mine);
            }
        }
    };

    const int L = kmax - range.x + 1;   // entries of the reverse walk
    if (L < batch_thresh) {
        int idc = ids[kmax];                         // raw id word (quadrant bits on top)
        Rec cur = recs[idc & idmask];
        int idn = ids[max(kmax - 1, range.x)];
        for (int k = kmax; k >= range.x; --k) {
            const Rec nxt = recs[idn & idmask];
            const int idnn = idn;
            idn = ids[max(k - 2, range.x)];
            entry(cur, k, qm_on ? qm_bits(idc, cur.ex) : quadrant_mask(cur, qcx, qcy, qtest));
            cur = nxt;
            idc = idnn;
        }
    } else {
        // long walk: 64-entry batches staged through wave-private LDS (see the forward kernel)
        __shared__ float4 stage[2][64 * 3];
        const int nb = (L + 63) >> 6;
        int rid = 0;         // raw id word of this lane's row
        auto fetch = [&](int bidx, float4 &r0, float4 &r1, float4 &r2) __attribute__((always_inline)) {
            const int k = kmax - (bidx << 6) - lane;
            if (k >= range.x) {
                rid = ids[k];
                const float4 *p = reinterpret_cast<const float4 *>(recs + (rid & idmask));
                r0 = p[0]; r1 = p[1]; r2 = p[2];
            }
        };
        auto row_mask = [&](const float4 &r0, const float4 &r2) __attribute__((always_inline)) -> unsigned {
            return qm_on ? qm_bits(rid, r2.z) : row_quadrants(r0.x, r0.y, r2.z, r2.w, tx * 16, ty * 16, qtest);
        };
        unsigned mine_q = 0u;
#pragma unroll
        for (int q = 0; q < QPW; ++q)
            if (kfin_active[q]) mine_q |= 1u << (q0 + q);
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
        fetch(0, r0, r1, r2);
        stage[0][lane * 3 + 0] = r0; stage[0][lane * 3 + 1] = r1; stage[0][lane * 3 + 2] = r2;
        unsigned qrow = row_mask(r0, r2) & mine_q;
        for (int bi = 0; bi < nb; ++bi) {
            const int cnt = min(64, L - (bi << 6));
            unsigned long long todo = __ballot(lane < cnt && qrow != 0u);
            const unsigned qcur = qrow;
            if (bi + 1 < nb) fetch(bi + 1, r0, r1, r2);
            const float4 *sb = stage[bi & 1];
            while (todo) {
                const int j = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                Rec cur;
                const float4 a0 = sb[j * 3 + 0], a1 = sb[j * 3 + 1], a2 = sb[j * 3 + 2];
                cur.x = a0.x; cur.y = a0.y; cur.opac = a0.z; cur.ha = a0.w;
                cur.b = a1.x; cur.hc = a1.y; cur.r = a1.z; cur.g = a1.w;
                cur.bl = a2.x; cur.gid = __float_as_int(a2.y); cur.ex = a2.z; cur.ey = a2.w;
                entry(cur, kmax - (bi << 6) - j, (unsigned)__builtin_amdgcn_readlane((int)qcur, j));
            }
            if (bi + 1 < nb) {
                float4 *sn = stage[(bi + 1) & 1];
                sn[lane * 3 + 0] = r0; sn[lane * 3 + 1] = r1; sn[lane * 3 + 2] = r2;
                qrow = row_mask(r0, r2) & mine_q;
            }
        }
    }
}

// Launch shapes of the backward.
//   MODE 0: one workgroup per (tile, wave), the in-kernel adaptive split (QPW = 4, ADAPT): callers without a launch
//           order, tile sizes other than 16.
//   MODE 1: "short" half of the two-kernel adaptive scheme: one wave per tile (QPW = 4); block b takes tile
//           order[b] and leaves the first n_long = order[n_tiles] entries (the long walks) to MODE 2.  (Its own
//           kernel, raster_bwd_short_kernel below: it carries an occupancy attribute the other shapes must not.)
//   MODE 2: "long" half: four lean waves per tile (QPW = 1, half the registers of the QPW = 4 body) over the n_long
//           longest walks, longest first.
// Why two kernels: splitting a long walk over four waves INSIDE the QPW = 4 kernel keeps that kernel's register
// budget and its per-slot control flow (street scene: 0.93 ms); the QPW = 1 body on the same tiles takes 0.59 ms,
// but costs 1.6x on tiles whose walks are short (benchmark scene: 0.52 vs 0.33 ms), where one wave per tile means one
// gradient reduction per (tile, Gaussian) instead of up to four.
template <bool EXACT, int REDUCE, int QPW, bool ADAPT, int MODE>
__global__ __launch_bounds__(64) void raster_bwd_kernel(int W, int H, int B, int tiles_x, int n_tiles,
                                                        const int2 *__restrict__ bins,
                                                        const Rec *__restrict__ recs,
                                                        const int32_t *__restrict__ ids,
                                                        const float *__restrict__ bg,
                                                        const float *__restrict__ final_T,
                                                        const int32_t *__restrict__ final_idx,
                                                        const float *__restrict__ v_out,
                                                        const float *__restrict__ v_out_alpha,
                                                        float alpha_clamp, float *__restrict__ grad_ws, int dbg,
                                                        int adapt_thresh, int batch_thresh,
                                                        const int32_t *__restrict__ tile_order, int use_qm) {
    if constexpr (MODE == 0) {
        static_assert(QPW == 4 && ADAPT, "MODE 0 is the in-kernel adaptive split");
        // wave-major numbering (blocks [0, n_tiles) are wave 0 of every tile): the waves that do the work of un-split
        // tiles are spread over all 8 XCDs (block b runs on XCD b % 8), not on every 4th block
        int tile = (int)(blockIdx.x % n_tiles);
        if (tile_order) tile = tile_order[tile];
        const int wv = (int)(blockIdx.x / n_tiles);
        raster_bwd_tile<EXACT, REDUCE, QPW, ADAPT>(tile, wv, W, H, B, tiles_x, bins, recs, ids, bg, final_T,
                                                           final_idx, v_out, v_out_alpha, alpha_clamp, grad_ws, dbg,
                                                           adapt_thresh, batch_thresh, use_qm);
    } else {
        static_assert(MODE == 2, "MODE 1 lives in raster_bwd_short_kernel");
        static_assert(QPW == 1 && !ADAPT, "long tiles: four lean waves per tile");
        // one workgroup per (tile, quadrant) of the n_long longest walks, longest first; the hardware's own dispatch
        // balances them.  (Persistent waves were tried: a static stride left the wave with the longest item of every
        // round last, 0.75 vs 0.57 ms on the street scene; a shared atomic cursor serialised on its one address,
        // +90 us even when there was nothing to do.)  Workgroups beyond the long prefix exit at once.
        if ((int)blockIdx.x >= 4 * tile_order[n_tiles]) return;
        // These waves are the critical path of the backward on skewed content (one 3000-entry walk = 0.5 ms) and share
        // their SIMDs with the short-walk kernel's waves: give them the issue priority (street scene: 516 -> 537
        // images/s; two quadrants per wave here, i.e. half the reductions, made it 435: it IS the critical path).
        __builtin_amdgcn_s_setprio(3);
        raster_bwd_tile<EXACT, REDUCE, 1, false>(tile_order[blockIdx.x >> 2], blockIdx.x & 3, W, H, B, tiles_x,
                                                         bins, recs, ids, bg, final_T, final_idx, v_out, v_out_alpha,
                                                         alpha_clamp, grad_ws, dbg, adapt_thresh, batch_thresh, use_qm);
    }
}

// The "short" half of the two-kernel scheme (MODE 1 above) as its own kernel, held at FOUR waves per SIMD.  Built
// without the SLP vectoriser (see the Makefile) its body needs 94 VGPRs and would run five; on the street scene the
// extra waves take issue slots from the concurrently running long-walk kernel, which is the critical path there
// (long-walk kernel 613 -> 768 us, step 1.97 -> 2.24 ms, profiles/experiments/r02_packed_forward_notes.md).
template <bool EXACT, int REDUCE>
// (re-measured in r03 with the moment-form body, profiles/scripts/r03o.sh: 5 or 6 waves change nothing on the benchmark
// scene — 0.276-0.279 ms — and cost the street scene 8 %: 490 vs 535-540 images/s)
#ifndef SGN_BWD_SHORT_WAVES_MAX
#define SGN_BWD_SHORT_WAVES_MAX 4
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, SGN_BWD_SHORT_WAVES_MAX))) void raster_bwd_short_kernel(
    int W, int H, int B, int tiles_x, int n_tiles, const int2 *__restrict__ bins, const Rec *__restrict__ recs,
    const int32_t *__restrict__ ids, const float *__restrict__ bg, const float *__restrict__ final_T,
    const int32_t *__restrict__ final_idx, const float *__restrict__ v_out, const float *__restrict__ v_out_alpha,
    float alpha_clamp, float *__restrict__ grad_ws, int dbg, int adapt_thresh, int batch_thresh,
    const int32_t *__restrict__ tile_order, int use_qm) {
    const int n_long = tile_order[n_tiles];
    if ((int)blockIdx.x < n_long) return;
    raster_bwd_tile<EXACT, REDUCE, 4, false>(tile_order[blockIdx.x], 0, W, H, B, tiles_x, bins, recs, ids, bg,
                                                     final_T, final_idx, v_out, v_out_alpha, alpha_clamp, grad_ws, dbg,
                                                     adapt_thresh, batch_thresh, use_qm);
}

// rows [row0, row0 + n) of the packed gradient workspace -> the n rows of the four output arrays
__global__ __launch_bounds__(256) void unpack_grads_kernel(int n, int row0, const float *__restrict__ ws,
                                                           const float *__restrict__ conics,
                                                           const float *__restrict__ opac, int opac_is_logit,
                                                           const float *__restrict__ colors_pre,
                                                           float *__restrict__ v_xy, float *__restrict__ v_conic,
                                                           float *__restrict__ v_colors,
                                                           float *__restrict__ v_opac) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 a = reinterpret_cast<const float4 *>(ws)[3 * (size_t)(row0 + i) + 0];
    const float4 b = reinterpret_cast<const float4 *>(ws)[3 * (size_t)(row0 + i) + 1];
    const float4 c = reinterpret_cast<const float4 *>(ws)[3 * (size_t)(row0 + i) + 2];
    // moments -> gradients with this Gaussian's conic: sigma = (A dx^2 + C dy^2) / 2 + B dx dy, so
    //   v_xy = (A m_x + B m_y, B m_x + C m_y),  v_conic = (s_xx / 2, s_xy, s_yy / 2)  ([:,1] is the TRUE dL/dB)
    const float A = conics[3 * i], B = conics[3 * i + 1], C = conics[3 * i + 2];
    v_xy[2 * i] = fmaf(A, a.x, B * a.y);
    v_xy[2 * i + 1] = fmaf(B, a.x, C * a.y);
    v_conic[3 * i] = 0.5f * a.z; v_conic[3 * i + 1] = a.w; v_conic[3 * i + 2] = 0.5f * b.x;
    float v0 = b.y, v1 = b.z, v2 = b.w;
    if (colors_pre != nullptr) {   // the colours were clamp(pre, min = 0): gradient w.r.t. `pre` (torch: grad * (pre >= 0))
        v0 = colors_pre[3 * i] >= 0.f ? v0 : 0.f;
        v1 = colors_pre[3 * i + 1] >= 0.f ? v1 : 0.f;
        v2 = colors_pre[3 * i + 2] >= 0.f ? v2 : 0.f;
    }
    v_colors[3 * i] = v0; v_colors[3 * i + 1] = v1; v_colors[3 * i + 2] = v2;
    float vo = c.x;
    if (opac_is_logit == 1) {         // opac holds logits: chain through the fused sigmoid
        const float sg = 1.f / (1.f + expf(-opac[i]));
        vo = vo * sg * (1.f - sg);
    } else if (opac_is_logit == 2) {  // opac holds sigmoid OUTPUTS, the gradient w.r.t. their logits is wanted
        const float sg = opac[i];
        vo = vo * sg * (1.f - sg);
    }
    v_opac[i] = vo;
}

// Kernel-selection options arrive with every call (include/sgn_rast.h: sgn_raster_opts); NULL = defaults.
sgn_raster_opts resolve_opts(const sgn_raster_opts *o) {
    sgn_raster_opts r;
    sgn_raster_default_opts(&r);
    if (o) {
        r = *o;
        r.exact_exp = r.exact_exp ? 1 : 0;
        r.reduce_mode = r.reduce_mode ? 1 : 0;
        r.ids_qmask = r.ids_qmask ? 1 : 0;
        sgn_raster_opts d;
        sgn_raster_default_opts(&d);
        if (r.adapt_fwd <= 0) r.adapt_fwd = d.adapt_fwd;
        if (r.adapt_bwd <= 0) r.adapt_bwd = d.adapt_bwd;
        if (r.batch_fwd <= 0) r.batch_fwd = d.batch_fwd;
        if (r.batch_bwd <= 0) r.batch_bwd = d.batch_bwd;
    }
    return r;
}

}  // namespace

SGN_EXPORT void sgn_raster_default_opts(sgn_raster_opts *out) {
    if (!out) return;
    out->exact_exp = 0;        // hardware v_exp_f32
    out->reduce_mode = 1;      // transposed permlane-swap reduction
    out->adapt_fwd = 1024;     // forward: lists with >= this many entries get four waves (half-octave classes)
    out->adapt_bwd = 256;      // backward: reverse walks of >= this many entries go to the four-waves-per-tile kernel
    out->batch_fwd = 256;      // forward: lists with >= this many entries go through the LDS-batched path
    out->batch_bwd = 128;      // backward: same for reverse walks
    out->debug_flags = 0;
    out->ids_qmask = 0;        // the list carries no quadrant masks unless its builder says so
}

SGN_EXPORT size_t sgn_raster_workspace_bytes(int n, int64_t n_isect, const sgn_raster_opts *opts) {
    (void)n_isect; (void)opts;       // one 48-byte row per Gaussian (rounds 1-5 added a depth-ordered stream in "stream" mode)
    return (size_t)(n > 0 ? n : 1) * sizeof(Rec);
}

SGN_EXPORT size_t sgn_raster_bwd_workspace_bytes(int n) {
    return (size_t)(n > 0 ? n : 1) * SGN_RECORD_FLOATS * sizeof(float);
}

// builds the per-Gaussian rows
static int pack_records(int n, const float *xys, const float *conics, const float *colors, const float *opac,
                        int opac_is_logit, int id_lo, int id_hi, int window, void *recs, hipStream_t s) {
    sgn_timing_begin(SGN_T_PACK, s);
    hipLaunchKernelGGL(build_grec_kernel, dim3(sgn_cdiv(n, 256)), dim3(256), 0, s, n, xys, conics, colors, opac,
                       opac_is_logit, id_lo, id_hi, window, (float4 *)recs, (const int32_t *)nullptr);
    sgn_timing_end(SGN_T_PACK, s);
    return 0;
}

// The per-Gaussian rows do not depend on the intersection list: a caller may build them while it waits for the
// intersection count (pass rows_built = 1 to sgn_raster_fwd afterwards; gather mode only — in stream mode the rows are
// re-packed into depth order and sgn_raster_fwd builds them itself).
SGN_EXPORT int sgn_raster_build_rows(int n, const float *xys, const float *conics, const float *colors,
                                     const float *opacities, int opacity_is_logit, int id_lo, int id_hi, int window,
                                     void *recs_ws, size_t recs_ws_bytes, const int32_t *skip_flag,
                                     sgn_stream_t stream) {
    SGN_ARG_CHECK(!window || (0 <= id_lo && id_lo <= id_hi && id_hi <= n), -4);
    SGN_ARG_CHECK(n >= 0, -1);
    if (n == 0) return 0;
    SGN_ARG_CHECK(xys && conics && colors && opacities && recs_ws, -2);
    SGN_ARG_CHECK(recs_ws_bytes >= (size_t)n * sizeof(Rec), -3);
    hipStream_t s = (hipStream_t)stream;
    sgn_timing_begin(SGN_T_PACK, s);
    hipLaunchKernelGGL(build_grec_kernel, dim3(sgn_cdiv(n, 256)), dim3(256), 0, s, n, xys, conics, colors, opacities,
                       opacity_is_logit, id_lo, id_hi, window, (float4 *)recs_ws, skip_flag);
    sgn_timing_end(SGN_T_PACK, s);
    SGN_LAUNCH_CHECK();
    return 0;
}

static int raster_fwd_impl(int img_h, int img_w, int block_width, int n, int64_t n_isect,
                           const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
                           const float *conics, const float *colors, const float *opacities,
                           int opacity_is_logit, int id_lo, int id_hi, int window, const float *background3,
                           float *out_img, float *final_Ts, int32_t *final_idx, void *recs_ws, size_t recs_ws_bytes,
                           int rows_built, const int32_t *tile_order, int32_t *tile_kmax,
                           const float *depths, float *out_depth, const int32_t *skip_flag,
                           const sgn_raster_opts *opts, sgn_stream_t stream, const FwdGroups *groups,
                           int kmax_cleared = 0) {
    const sgn_raster_opts o = resolve_opts(opts);
    // group accumulations ride on the packed two-waves-per-tile forward (16x16 tiles, the whole scene)
    SGN_ARG_CHECK(groups == nullptr || (block_width == 16 && !window && !skip_flag), -12);
    FwdGroups Gv = {};
    if (groups) Gv = *groups;
    SGN_ARG_CHECK((depths == nullptr) == (out_depth == nullptr), -8);
    SGN_ARG_CHECK(skip_flag == nullptr || rows_built, -9);                 // a skipped pass builds no rows of its own
    SGN_ARG_CHECK(!(window && depths), -10);                               // the depth channel is a whole-scene pass
    SGN_ARG_CHECK(img_h > 0 && img_w > 0, -1);
    SGN_ARG_CHECK(block_width >= 2 && block_width <= 16, -2);
    SGN_ARG_CHECK(n_isect >= 0 && n_isect < ((int64_t)1 << 31), -3);
    SGN_ARG_CHECK(tile_bins && background3 && out_img && final_Ts && final_idx, -4);
    SGN_ARG_CHECK(n_isect == 0 || (gaussian_ids_sorted && xys && conics && colors && opacities && recs_ws), -5);
    SGN_ARG_CHECK(n >= 0 && recs_ws_bytes >= sgn_raster_workspace_bytes(n, n_isect, &o), -6);
    SGN_ARG_CHECK(!window || (0 <= id_lo && id_lo <= id_hi && id_hi <= n), -7);
    SGN_ARG_CHECK(!o.ids_qmask || (block_width == 16 && n < SGN_QMASK_MAX_IDS), -11);
    hipStream_t s = (hipStream_t)stream;
    if (n_isect > 0 && !rows_built)
        pack_records(n, xys, conics, colors, opacities, opacity_is_logit, id_lo, id_hi, window, recs_ws, s);
    const int tiles_x = (img_w + block_width - 1) / block_width, tiles_y = (img_h + block_width - 1) / block_width;
    const Rec *rows = (const Rec *)recs_ws;
    if (tile_kmax && !kmax_cleared) SGN_HIP_CHECK(hipMemsetAsync(tile_kmax, 0, sizeof(int32_t) * 2 * tiles_x * tiles_y, s));
    if (groups) SGN_HIP_CHECK(hipMemsetAsync(Gv.kmax, 0, sizeof(int32_t) * 4 * tiles_x * tiles_y, s));
    sgn_timing_begin(SGN_T_RASTER_FWD, s);
    // Two launch shapes: 16x16 tiles run the PACKED kernel (two waves per tile, two pixels per lane; the longest lists of
    // the launch order get four waves); every other tile size four waves per tile, one quadrant each.
#define SGN_LAUNCH_FWD(EX, DE)                                                                                       \
    hipLaunchKernelGGL((raster_fwd_kernel<EX, 1, DE>), dim3(tiles_x * tiles_y * 4), dim3(64), 0, s, img_w, img_h,    \
                       block_width, tiles_x, (const int2 *)tile_bins, rows, gaussian_ids_sorted, background3, out_img, \
                       final_Ts, final_idx, o.batch_fwd, tile_order, tile_kmax, depths, out_depth, skip_flag, o.ids_qmask)
#define SGN_LAUNCH_FWD_PKG(EX, DE, GR)                                                                               \
    hipLaunchKernelGGL((raster_fwd_pk_kernel<EX, DE, GR>), dim3(((tiles_x * tiles_y + 7) / 8) * 32 + 32), dim3(64), 0, s, \
                       img_w, img_h, tiles_x, tiles_x * tiles_y, (const int2 *)tile_bins, rows, gaussian_ids_sorted, \
                       background3, out_img, final_Ts, final_idx, o.batch_fwd, tile_order, tile_kmax, depths, out_depth, \
                       skip_flag, o.ids_qmask, Gv)
#define SGN_LAUNCH_FWD3(EX, DE)                                                         \
    do {                                                                                \
        if (block_width == 16) {                                                        \
            if (groups) SGN_LAUNCH_FWD_PKG(EX, DE, true); else SGN_LAUNCH_FWD_PKG(EX, DE, false); \
        } else SGN_LAUNCH_FWD(EX, DE);                                                  \
    } while (0)
#define SGN_LAUNCH_FWD2(EX) do { if (depths) SGN_LAUNCH_FWD3(EX, true); else SGN_LAUNCH_FWD3(EX, false); } while (0)
    if (o.exact_exp) SGN_LAUNCH_FWD2(true); else SGN_LAUNCH_FWD2(false);
#undef SGN_LAUNCH_FWD2
#undef SGN_LAUNCH_FWD3
#undef SGN_LAUNCH_FWD_PKG
#undef SGN_LAUNCH_FWD
    sgn_timing_end(SGN_T_RASTER_FWD, s);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_raster_fwd(int img_h, int img_w, int block_width, int n, int64_t n_isect,
                              const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
                              const float *conics, const float *colors, const float *opacities,
                              int opacity_is_logit, int id_lo, int id_hi, int window, const float *background3,
                              float *out_img, float *final_Ts, int32_t *final_idx, void *recs_ws, size_t recs_ws_bytes,
                              int rows_built, const int32_t *tile_order, int32_t *tile_kmax,
                              const float *depths, float *out_depth, const int32_t *skip_flag,
                              const sgn_raster_opts *opts, sgn_stream_t stream) {
    return raster_fwd_impl(img_h, img_w, block_width, n, n_isect, gaussian_ids_sorted, tile_bins, xys, conics, colors,
                           opacities, opacity_is_logit, id_lo, id_hi, window, background3, out_img, final_Ts, final_idx,
                           recs_ws, recs_ws_bytes, rows_built, tile_order, tile_kmax, depths, out_depth, skip_flag, opts,
                           stream, nullptr);
}

// sgn_raster_fwd for a caller inside the library whose earlier launch on the same stream already cleared tile_kmax
// (api.cpp sgn_rasterize_fwd_all: the emission clears the statistics together with the bins)
int sgn_raster_fwd_precleared(int img_h, int img_w, int block_width, int n, int64_t n_isect,
                              const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
                              const float *conics, const float *colors, const float *opacities,
                              int opacity_is_logit, const float *background3, float *out_img, float *final_Ts,
                              int32_t *final_idx, void *recs_ws, size_t recs_ws_bytes, const int32_t *tile_order,
                              int32_t *tile_kmax, const float *depths, float *out_depth, const sgn_raster_opts *opts,
                              sgn_stream_t stream) {
    return raster_fwd_impl(img_h, img_w, block_width, n, n_isect, gaussian_ids_sorted, tile_bins, xys, conics, colors,
                           opacities, opacity_is_logit, 0, n, 0, background3, out_img, final_Ts, final_idx, recs_ws,
                           recs_ws_bytes, 1, tile_order, tile_kmax, depths, out_depth, nullptr, opts, stream, nullptr, 1);
}

// The forward with the two GROUP accumulations riding on it (FwdGroups above): besides everything sgn_raster_fwd
// writes, the final transmittance / final index / per-tile walk depth of the pass that would render only ids < split
// ("head") and of the pass that would render only ids >= split ("tail") over the same list — bit-equal to what two more
// sgn_raster_fwd calls with id ranges [0, split) / [split, n) produce (tests/test_gpu_groups.py).  `own_group` (0 head,
// 1 tail, -1 none) + `own_ids` / `own_bins`: ONE group may come with its own compacted list (sgn_list_window) — the one
// its backward will walk: its final indices are then positions of that list, and it finishes its walk there.
// group_state [4][H*W]: T_head, T_tail, (int32) idx_head, idx_tail; group_stats [2][tiles*2] like tile_stats.
// Packed two-waves-per-tile forward (block_width 16), whole
// scene (no id window), rows already built or built here; -12 otherwise.
SGN_EXPORT int sgn_raster_fwd_groups(int img_h, int img_w, int n, int64_t n_isect, const int32_t *gaussian_ids_sorted,
                                     const int32_t *tile_bins, const float *xys, const float *conics,
                                     const float *colors, const float *opacities, int opacity_is_logit,
                                     const float *background3, float *out_img, float *final_Ts, int32_t *final_idx,
                                     void *recs_ws, size_t recs_ws_bytes, int rows_built, const int32_t *tile_order,
                                     int32_t *tile_kmax, const float *depths, float *out_depth, int split,
                                     int own_group, const int32_t *own_ids, const int32_t *own_bins,
                                     float *group_state, int32_t *group_stats, const sgn_raster_opts *opts,
                                     sgn_stream_t stream) {
    SGN_ARG_CHECK(group_state && group_stats, -13);
    SGN_ARG_CHECK(split >= 0 && split <= n, -14);
    SGN_ARG_CHECK(own_group >= -1 && own_group <= 1 && ((own_group >= 0) == (own_ids != nullptr)) &&
                      ((own_group >= 0) == (own_bins != nullptr)), -15);
    FwdGroups g;
    g.split = split;
    g.own = own_group;
    g.own_bins = (const int2 *)own_bins;
    g.own_ids = own_ids;
    g.state = group_state;
    g.kmax = group_stats;
    return raster_fwd_impl(img_h, img_w, 16, n, n_isect, gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities,
                           opacity_is_logit, 0, n, 0, background3, out_img, final_Ts, final_idx, recs_ws, recs_ws_bytes,
                           rows_built, tile_order, tile_kmax, depths, out_depth, nullptr, opts, stream, &g);
}

// ------------------------------------------------------------------ reuse of the depth channel (r03)
// The reference renders depth with a SECOND full rasterization of the same geometry whose colours are the depths
// themselves (sgn_splatfacto.py:982-994: `depths[:, None].repeat(1, 3)`).  When the first pass accumulated the depth
// channel, that second call is answered from it — if, and only if, its colours really are its depths, which only the
// device can tell without a host sync:
//   sgn_colors_match_depths  flag = 0 iff colors[i, c] == depths[i] bit for bit for every row and channel (else 1)
//   (the caller then queues the ordinary forward with skip_flag = flag: its kernels return at once when flag == 0)
//   sgn_depth_reuse          iff flag == 0: out_img[p, c] = fma(T[p], background[c], D[p]) — exactly what the
//                            rasterization would have produced (same fma) — and final_Ts / final_idx are copied
//                            from the first pass (the backward of the second node needs its own)
// Everything is queued unconditionally; nothing waits for the flag on the host.
__global__ __launch_bounds__(256) void colors_match_depths_kernel(int n, const uint32_t *__restrict__ colors,
                                                                  const uint32_t *__restrict__ depths,
                                                                  int32_t *__restrict__ flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool bad = false;
    if (i < n) {
        const uint32_t d = depths[i];
        bad = colors[3 * i] != d || colors[3 * i + 1] != d || colors[3 * i + 2] != d;
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__global__ __launch_bounds__(256) void depth_reuse_kernel(int n_pix, const int32_t *__restrict__ flag,
                                                          const float *__restrict__ D, const float *__restrict__ T1,
                                                          const int32_t *__restrict__ idx1,
                                                          const float *__restrict__ bg, float *__restrict__ out_img,
                                                          float *__restrict__ final_T, int32_t *__restrict__ final_idx,
                                                          int n_stats, const int32_t *__restrict__ stats_first,
                                                          int32_t *__restrict__ stats_out) {
    if (flag != nullptr && *flag != 0) return;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < n_stats) stats_out[p] = stats_first[p];      // the tile statistics of the pass (same geometry: same walks)
    if (p >= n_pix) return;
    const float t = T1[p], d = D[p];
    out_img[3 * p + 0] = fmaf(t, bg[0], d);
    out_img[3 * p + 1] = fmaf(t, bg[1], d);
    out_img[3 * p + 2] = fmaf(t, bg[2], d);
    if (final_T != nullptr) {          // NULL: the caller's node shares the first pass's buffers (host-proven reuse)
        final_T[p] = t;
        final_idx[p] = idx1[p];
    }
}

SGN_EXPORT int sgn_colors_match_depths(int n, const float *colors, const float *depths, int32_t *flag,
                                       sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0 && flag != nullptr, -1);
    hipStream_t s = (hipStream_t)stream;
    SGN_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int32_t), s));
    if (n == 0) return 0;
    SGN_ARG_CHECK(colors && depths, -2);
    hipLaunchKernelGGL(colors_match_depths_kernel, dim3(sgn_cdiv(n, 256)), dim3(256), 0, s, n, (const uint32_t *)colors,
                       (const uint32_t *)depths, flag);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_depth_reuse(int img_h, int img_w, const int32_t *flag, const float *depth_channel,
                               const float *final_Ts_first, const int32_t *final_idx_first, const float *background3,
                               float *out_img, float *final_Ts, int32_t *final_idx, int n_stats,
                               const int32_t *tile_stats_first, int32_t *tile_stats, sgn_stream_t stream) {
    SGN_ARG_CHECK(img_h > 0 && img_w > 0, -1);
    SGN_ARG_CHECK(n_stats == 0 || (n_stats <= img_h * img_w && tile_stats_first && tile_stats), -3);
    SGN_ARG_CHECK(depth_channel && final_Ts_first && final_idx_first && background3 && out_img, -2);
    SGN_ARG_CHECK((final_Ts == nullptr) == (final_idx == nullptr), -4);
    const int n_pix = img_h * img_w;
    sgn_timing_begin(SGN_T_RASTER_FWD, stream);
    hipLaunchKernelGGL(depth_reuse_kernel, dim3(sgn_cdiv(n_pix, 256)), dim3(256), 0, (hipStream_t)stream, n_pix, flag,
                       depth_channel, final_Ts_first, final_idx_first, background3, out_img, final_Ts, final_idx, n_stats,
                       tile_stats_first, tile_stats);
    sgn_timing_end(SGN_T_RASTER_FWD, stream);
    SGN_LAUNCH_CHECK();
    return 0;
}

static int raster_bwd_impl(int img_h, int img_w, int block_width, int n, int64_t n_isect,
                              const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
                              const float *conics, const float *colors, const float *opacities,
                              int opacity_is_logit, int id_lo, int id_hi, int window, const float *background3,
                              const float *final_Ts, const int32_t *final_idx,
                              const float *v_out_img, const float *v_out_alpha, float alpha_clamp_bwd,
                              float *v_xy, float *v_conic, float *v_colors, float *v_opacity, void *recs_ws,
                              size_t recs_ws_bytes, int recs_packed, void *grad_ws, size_t grad_ws_bytes,
                              const int32_t *tile_order, const float *colors_pre_clamp,
                              const sgn_raster_opts *opts, sgn_stream_t stream, sgn_stream_t aux_stream,
                              int keep_ws, int defer_unpack) {
    const sgn_raster_opts o = resolve_opts(opts);
    SGN_ARG_CHECK(opacity_is_logit >= 0 && opacity_is_logit <= 2, -11);
    const int logit_rows = opacity_is_logit == 1;     // 2: the values ARE probabilities, only the gradient is converted
    SGN_ARG_CHECK(!window || (0 <= id_lo && id_lo <= id_hi && id_hi <= n), -9);
    SGN_ARG_CHECK(img_h > 0 && img_w > 0 && n >= 0, -1);
    SGN_ARG_CHECK(block_width >= 2 && block_width <= 16, -2);
    SGN_ARG_CHECK(n_isect >= 0 && n_isect < ((int64_t)1 << 31), -3);
    if (n == 0) return 0;
    SGN_ARG_CHECK(v_xy && v_conic && v_colors && v_opacity && grad_ws && opacities && conics, -4);
    SGN_ARG_CHECK(grad_ws_bytes >= sgn_raster_bwd_workspace_bytes(n), -5);
    SGN_ARG_CHECK(alpha_clamp_bwd > 0.f && alpha_clamp_bwd < 1.f, -6);
    SGN_ARG_CHECK(!o.ids_qmask || (block_width == 16 && n < SGN_QMASK_MAX_IDS), -10);
    hipStream_t s = (hipStream_t)stream;
    if (!keep_ws) SGN_HIP_CHECK(hipMemsetAsync(grad_ws, 0, (size_t)n * SGN_RECORD_FLOATS * sizeof(float), s));
    if (n_isect > 0) {
        SGN_ARG_CHECK(gaussian_ids_sorted && tile_bins && xys && conics && colors && opacities && background3 &&
                          final_Ts && final_idx && v_out_alpha && recs_ws, -7);     // (v_out_img may be NULL: zeros)
        SGN_ARG_CHECK(recs_ws_bytes >= sgn_raster_workspace_bytes(n, n_isect, &o), -8);
        if (!recs_packed)
            pack_records(n, xys, conics, colors, opacities, logit_rows, id_lo, id_hi, window, recs_ws, s);
        const int tiles_x = (img_w + block_width - 1) / block_width, tiles_y = (img_h + block_width - 1) / block_width;
        const Rec *rows = (const Rec *)recs_ws;
        const int n_tiles = tiles_x * tiles_y;
        const int long_grid = n_tiles * 4;   // the long-walk kernel: workgroups beyond 4 * n_long exit at once
        // fork: the long-walk kernel goes to the auxiliary stream (if any) behind everything queued so far
        const bool two_kernel = tile_order != nullptr && block_width == 16;
        hipStream_t s2 = s;
        hipEvent_t ev_fork = nullptr, ev_join = nullptr;
        if (two_kernel && aux_stream != nullptr && (hipStream_t)aux_stream != s && sgn_fork_events(&ev_fork, &ev_join) == 0) {
            s2 = (hipStream_t)aux_stream;
            SGN_HIP_CHECK(hipEventRecord(ev_fork, s));
            SGN_HIP_CHECK(hipStreamWaitEvent(s2, ev_fork, 0));
        }
        sgn_timing_begin(SGN_T_RASTER_BWD, s);
#define SGN_BWD_ARGS                                                                                             \
    img_w, img_h, block_width, tiles_x, n_tiles, (const int2 *)tile_bins, rows, gaussian_ids_sorted, background3, \
        final_Ts, final_idx, v_out_img, v_out_alpha, alpha_clamp_bwd, (float *)grad_ws, o.debug_flags, o.adapt_bwd, \
        o.batch_bwd, tile_order, o.ids_qmask
#define SGN_LAUNCH_BWD(EX, RM)                                                                                   \
    do {                                                                                                         \
        if (!two_kernel)     /* no launch order / tiles other than 16x16: the in-kernel adaptive split */        \
            hipLaunchKernelGGL((raster_bwd_kernel<EX, RM, 4, true, 0>), dim3(n_tiles * 4), dim3(64), 0, s, SGN_BWD_ARGS); \
        else {   /* two-kernel adaptive scheme: order[0..n_long) = long walks, the rest short; the two halves */ \
                 /* touch disjoint tiles and run CONCURRENTLY when the caller lends a second stream           */ \
            hipLaunchKernelGGL((raster_bwd_kernel<EX, RM, 1, false, 2>), dim3(long_grid), dim3(64), 0, s2, SGN_BWD_ARGS); \
            hipLaunchKernelGGL((raster_bwd_short_kernel<EX, RM>), dim3(n_tiles), dim3(64), 0, s, SGN_BWD_ARGS);   \
        }                                                                                                        \
    } while (0)
        if (o.exact_exp) {
            if (o.reduce_mode) SGN_LAUNCH_BWD(true, 1); else SGN_LAUNCH_BWD(true, 0);
        } else {
            if (o.reduce_mode) SGN_LAUNCH_BWD(false, 1); else SGN_LAUNCH_BWD(false, 0);
        }
#undef SGN_LAUNCH_BWD
#undef SGN_BWD_ARGS
        if (s2 != s) {   // join: the unpack (and everything after) waits for the long-walk kernel
            SGN_HIP_CHECK(hipEventRecord(ev_join, s2));
            SGN_HIP_CHECK(hipStreamWaitEvent(s, ev_join, 0));
        }
        sgn_timing_end(SGN_T_RASTER_BWD, s);
    }
    sgn_timing_begin(SGN_T_UNPACK, s);
    if (!defer_unpack) {   // window: the outputs (and `opacities`) hold rows [id_lo, id_hi) only
        const int n_out = window ? id_hi - id_lo : n, row0 = window ? id_lo : 0;
        if (n_out > 0)
            hipLaunchKernelGGL(unpack_grads_kernel, dim3(sgn_cdiv(n_out, 256)), dim3(256), 0, s, n_out, row0,
                               (const float *)grad_ws, conics, opacities, opacity_is_logit, colors_pre_clamp, v_xy,
                               v_conic, v_colors, v_opacity);
    }
    sgn_timing_end(SGN_T_UNPACK, s);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_raster_bwd(int img_h, int img_w, int block_width, int n, int64_t n_isect,
                              const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
                              const float *conics, const float *colors, const float *opacities,
                              int opacity_is_logit, int id_lo, int id_hi, int window, const float *background3,
                              const float *final_Ts, const int32_t *final_idx,
                              const float *v_out_img, const float *v_out_alpha, float alpha_clamp_bwd,
                              float *v_xy, float *v_conic, float *v_colors, float *v_opacity, void *recs_ws,
                              size_t recs_ws_bytes, int recs_packed, void *grad_ws, size_t grad_ws_bytes,
                              const int32_t *tile_order, const float *colors_pre_clamp,
                              const sgn_raster_opts *opts, sgn_stream_t stream, sgn_stream_t aux_stream) {
    return raster_bwd_impl(img_h, img_w, block_width, n, n_isect, gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities, opacity_is_logit, id_lo, id_hi, window, background3, final_Ts, final_idx, v_out_img, v_out_alpha, alpha_clamp_bwd, v_xy, v_conic, v_colors, v_opacity, recs_ws, recs_ws_bytes, recs_packed, grad_ws, grad_ws_bytes, tile_order, colors_pre_clamp, opts, stream, aux_stream, 0, 0);
}

// One of SEVERAL reverse walks whose gradients belong to the same tensors (the main pass of sgn_raster_fwd_groups and the
// group accumulations that reached the loss): all of them accumulate into ONE packed gradient workspace and the last one
// unpacks — instead of a 48 MB clear, an unpack and four tensor additions per extra walk.  `first` != 0 clears the
// workspace, `last` != 0 unpacks it into v_xy / v_conic / v_colors / v_opacity (only then are those written; every
// walk of the sequence must be a whole-tensor pass — window = 0 — with the same opacities / conics / opacity_is_logit).
SGN_EXPORT int sgn_raster_bwd_part(int img_h, int img_w, int block_width, int n, int64_t n_isect,
                              const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
                              const float *conics, const float *colors, const float *opacities,
                              int opacity_is_logit, int id_lo, int id_hi, int window, const float *background3,
                              const float *final_Ts, const int32_t *final_idx,
                              const float *v_out_img, const float *v_out_alpha, float alpha_clamp_bwd,
                              float *v_xy, float *v_conic, float *v_colors, float *v_opacity, void *recs_ws,
                              size_t recs_ws_bytes, int recs_packed, void *grad_ws, size_t grad_ws_bytes,
                              const int32_t *tile_order, const float *colors_pre_clamp,
                              const sgn_raster_opts *opts, sgn_stream_t stream, sgn_stream_t aux_stream,
                                   int first, int last) {
    SGN_ARG_CHECK(!window, -13);
    return raster_bwd_impl(img_h, img_w, block_width, n, n_isect, gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities, opacity_is_logit, id_lo, id_hi, window, background3, final_Ts, final_idx, v_out_img, v_out_alpha, alpha_clamp_bwd, v_xy, v_conic, v_colors, v_opacity, recs_ws, recs_ws_bytes, recs_packed, grad_ws, grad_ws_bytes, tile_order, colors_pre_clamp, opts, stream, aux_stream, first ? 0 : 1, last ? 0 : 1);
}
