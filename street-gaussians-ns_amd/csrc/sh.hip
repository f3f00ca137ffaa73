// sh.hip — view-dependent colour from real spherical harmonics (forward + backward), gfx950.
//
// Replaces gsplat 0.1.x sh.cuh compute_sh_forward_kernel / compute_sh_backward_kernel
// (method="fast", Sloan recurrences), reached from the reference at
// sgn_splatfacto.py:939 and sgn_splatfacto_scene_graph.py:285.
//
// Pure streaming, HBM-bound: 12 B dir + 12*K B coeffs -> 12 B colour per Gaussian (216 B at K=16).
// Layout trick for 64-wide waves: the [n,K,3] coefficient block of 64 consecutive Gaussians is one
// contiguous 64*K*12-byte span, so each wave copies its span to LDS with full-width
// global_load_dwordx4 (lane-contiguous, every byte of every cache line used once) and then each
// lane walks its own K*3 row from LDS; the row stride (3K+1 dwords) is odd, so the 64 lanes hit
// distinct banks.  The backward writes v_coeffs the same way in reverse (row -> LDS -> coalesced
// dwordx4 stores).
#include "sgn_common.h"

namespace {

__device__ __forceinline__ int sh_bases(float dx, float dy, float dz, int deg, float *b) {
    b[0] = 0.2820947917738781f;
    if (deg < 1) return 1;
    const float inorm = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx * inorm, y = dy * inorm, z = dz * inorm;
    const float fTmp0A = 0.48860251190292f;
    b[1] = -fTmp0A * y; b[2] = fTmp0A * z; b[3] = -fTmp0A * x;
    if (deg < 2) return 4;
    const float z2 = z * z;
    const float fTmp0B = -1.092548430592079f * z;
    const float fTmp1A = 0.5462742152960395f;
    const float fC1 = x * x - y * y;
    const float fS1 = 2.f * x * y;
    b[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
    b[7] = fTmp0B * x; b[5] = fTmp0B * y; b[8] = fTmp1A * fC1; b[4] = fTmp1A * fS1;
    if (deg < 3) return 9;
    const float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
    const float fTmp1B = 1.445305721320277f * z;
    const float fTmp2A = -0.5900435899266435f;
    const float fC2 = x * fC1 - y * fS1;
    const float fS2 = x * fS1 + y * fC1;
    b[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
    b[13] = fTmp0C * x; b[11] = fTmp0C * y; b[14] = fTmp1B * fC1; b[10] = fTmp1B * fS1;
    b[15] = fTmp2A * fC2; b[9] = fTmp2A * fS2;
    if (deg < 4) return 16;
    const float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    const float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f;
    const float fTmp2B = -1.770130769779931f * z;
    const float fTmp3A = 0.6258357354491763f;
    const float fC3 = x * fC2 - y * fS2;
    const float fS3 = x * fS2 + y * fC2;
    b[20] = 1.984313483298443f * z * b[12] + -1.006230589874905f * b[6];
    b[21] = fTmp0D * x; b[19] = fTmp0D * y; b[22] = fTmp1C * fC1; b[18] = fTmp1C * fS1;
    b[23] = fTmp2B * fC2; b[17] = fTmp2B * fS2; b[24] = fTmp3A * fC3; b[16] = fTmp3A * fS3;
    return 25;
}

// Copy this wave's `cnt` contiguous coefficient rows (KC floats each, 16-byte aligned span) into its LDS slab with
// row stride LS.  ALL global loads are issued before the first LDS write: written as a plain `for (t...) { load;
// store }` loop the compiler emits load -> s_waitcnt vmcnt(0) -> ds_write per iteration, i.e. 12 serialised HBM
// round trips per wave (sh_fwd: 68 us -> see DESIGN.md).
template <int KC, int LS>
__device__ __forceinline__ void stage_rows(float *__restrict__ my, const float *__restrict__ src, int cnt, int lane) {
    constexpr int NIT = (16 * KC + 63) / 64;          // float4 loads per lane for a full wave of 64 rows
    const int total = cnt * KC;
    if (reinterpret_cast<uintptr_t>(src) & 15) {      // a view with an odd storage offset: plain dword copy
        for (int e = lane; e < total; e += 64) {
            const int r = e / KC, c = e - r * KC;
            my[r * LS + c] = src[e];
        }
        return;
    }
    const int n4 = total >> 2;
    const float4 *src4 = reinterpret_cast<const float4 *>(src);
    float4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int t = lane + 64 * it;
        v[it] = (t < n4) ? src4[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int t = lane + 64 * it;
        if (t < n4) {
            const float f[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * t + j, r = e / KC, c = e - r * KC;
                my[r * LS + c] = f[j];
            }
        }
    }
    const int e = (n4 << 2) + lane;                   // up to three trailing floats of a partial wave
    if (lane < (total & 3)) {
        const int r = e / KC, c = e - r * KC;
        my[r * LS + c] = src[e];
    }
}

// LDS traffic between lanes of ONE wave (private slab): the hardware retires a wave's LDS operations in order, the
// fence only keeps the compiler from moving them across
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// stage_rows split in two, so that the loads of the NEXT span can be in flight while this one is evaluated
template <int KC>
__device__ __forceinline__ void load_span(float4 (&v)[(16 * KC + 63) / 64], const float *__restrict__ src, int cnt,
                                          int lane) {
    constexpr int NIT = (16 * KC + 63) / 64;
    const int n4 = (cnt * KC) >> 2;
    const float4 *src4 = reinterpret_cast<const float4 *>(src);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int t = lane + 64 * it;
        v[it] = (t < n4) ? src4[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int KC, int LS>
__device__ __forceinline__ void store_span(float *__restrict__ my, const float4 (&v)[(16 * KC + 63) / 64],
                                           const float *__restrict__ src, int cnt, int lane) {
    constexpr int NIT = (16 * KC + 63) / 64;
    const int total = cnt * KC, n4 = total >> 2;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int t = lane + 64 * it;
        if (t < n4) {
            const float f[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * t + j, r = e / KC, c = e - r * KC;
                my[r * LS + c] = f[j];
            }
        }
    }
    const int e = (n4 << 2) + lane;                   // up to three trailing floats of a partial wave
    if (lane < (total & 3)) {
        const int r = e / KC, c = e - r * KC;
        my[r * LS + c] = src[e];
    }
}

// One wave (64 lanes) per span of 64 Gaussians; a block is WAVES waves with private LDS slabs and each wave walks
// spans in a grid-stride loop with the NEXT span's twelve float4 loads already in flight while it evaluates the
// current one (one span per wave and no prefetch reached 4.1 TB/s where a plain streaming read on this GPU reaches
// 6.5, profiles/microbench/hbm_rates.hip: with 12.5 KB of LDS per wave only 12 waves fit a CU, and each spent most
// of its life NOT waiting for memory).
// KC = K*3 dwords per row; LDS row stride KC+1 (odd) => conflict-free per-lane row walks.
template <int K, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sh_fwd_kernel(int n, int deg, const float *__restrict__ dirs,
                                                            const float *__restrict__ coeffs,
                                                            float *__restrict__ colors) {
    constexpr int KC = K * 3, LS = KC + 1, NIT = (16 * KC + 63) / 64;
    __shared__ float lds[WAVES][64 * LS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_spans = (n + 63) >> 6, stride = gridDim.x * WAVES;
    float *my = lds[wave];
    const bool aligned = (reinterpret_cast<uintptr_t>(coeffs) & 15) == 0;   // 64 rows * KC floats keep the alignment
    int span = blockIdx.x * WAVES + wave;
    float4 v[NIT];
    float d0 = 0.f, d1 = 0.f, d2 = 1.f;               // this lane's view direction of the span in flight
    auto fetch = [&](int sp) __attribute__((always_inline)) {
        const int g = sp * 64, c = min(64, n - g);
        if (aligned) load_span<KC>(v, coeffs + (size_t)g * KC, c, lane);
        if (lane < c) { d0 = dirs[3 * (g + lane)]; d1 = dirs[3 * (g + lane) + 1]; d2 = dirs[3 * (g + lane) + 2]; }
    };
    if (span < n_spans) fetch(span);
    for (; span < n_spans; span += stride) {
        const int g0 = span * 64, cnt = min(64, n - g0);
        const float *src = coeffs + (size_t)g0 * KC;
        if (aligned) store_span<KC, LS>(my, v, src, cnt, lane);
        else stage_rows<KC, LS>(my, src, cnt, lane);                        // a view with an odd storage offset
        const float x = d0, y = d1, z = d2;
        const int next = span + stride;
        if (next < n_spans) fetch(next);
        wave_lds_fence();
        if (lane < cnt) {
            const int i = g0 + lane;
            float b[25];
            const int nb = sh_bases(x, y, z, deg, b);
            const float *row = my + lane * LS;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (k < nb) {
                    a0 += b[k] * row[3 * k];
                    a1 += b[k] * row[3 * k + 1];
                    a2 += b[k] * row[3 * k + 2];
                }
            }
            colors[3 * i] = a0; colors[3 * i + 1] = a1; colors[3 * i + 2] = a2;
        }
        wave_lds_fence();                              // this span's row reads stay ahead of the next span's writes
    }
}

template <int K, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sh_bwd_kernel(int n, int deg, const float *__restrict__ dirs,
                                                            const float *__restrict__ v_colors,
                                                            float *__restrict__ v_coeffs) {
    constexpr int KC = K * 3, LS = KC + 1;
    __shared__ float lds[WAVES][64 * LS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g0 = (blockIdx.x * WAVES + wave) * 64;
    const int cnt = max(0, min(64, n - g0));
    float *my = lds[wave];
    if (lane < cnt) {
        const int i = g0 + lane;
        float b[25];
        const int nb = sh_bases(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], deg, b);
        const float v0 = v_colors[3 * i], v1 = v_colors[3 * i + 1], v2 = v_colors[3 * i + 2];
        float *row = my + lane * LS;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float bk = (k < nb) ? b[k] : 0.f;
            row[3 * k] = bk * v0; row[3 * k + 1] = bk * v1; row[3 * k + 2] = bk * v2;
        }
    }
    __syncthreads();
    float *dst = v_coeffs + (size_t)g0 * KC;
    const int total = cnt * KC;
    if constexpr (KC % 4 == 0) {
        float4 *dst4 = reinterpret_cast<float4 *>(dst);
        for (int t = lane; t < total / 4; t += 64) {
            const int e = t * 4, r = e / KC, c = e - r * KC;
            const float *s = my + r * LS + c;
            dst4[t] = make_float4(s[0], s[1], s[2], s[3]);
        }
    } else {
        for (int e = lane; e < total; e += 64) {
            const int r = e / KC, c = e - r * KC;
            dst[e] = my[r * LS + c];
        }
    }
}

__device__ __forceinline__ void fused_viewdir(int i, const float *__restrict__ means,
                                              const float *__restrict__ cam_pos,
                                              const int32_t *__restrict__ object_ids,
                                              const float *__restrict__ poses, float &dx, float &dy, float &dz) {
    float m0 = means[3 * i], m1 = means[3 * i + 1], m2 = means[3 * i + 2];
    if (poses != nullptr && object_ids != nullptr) {  // local -> world (scene_graph.py:414)
        const float *P = poses + 16 * (size_t)object_ids[i];
        const float w0 = P[0] * m0 + P[1] * m1 + P[2] * m2 + P[9];
        const float w1 = P[3] * m0 + P[4] * m1 + P[5] * m2 + P[10];
        const float w2 = P[6] * m0 + P[7] * m1 + P[8] * m2 + P[11];
        m0 = w0; m1 = w1; m2 = w2;
    }
    dx = m0 - cam_pos[0]; dy = m1 - cam_pos[1]; dz = m2 - cam_pos[2];
}

// ---- fused SH front end (extension) ------------------------------------------------
// colour = clamp( SH(deg, means - cam_pos, [dc_eff | rest]) + 0.5, min 0 )  with
// dc_eff = sum_f features_dc[:, f, :] * idft[object, f]   (Fourier "4D-SH" DC term,
// sgn_splatfacto_scene_graph.py:239-247; F = 1 and idft = [1] for the background model).
// Replaces, per call: view-dir subtract + norm + divide, the Fourier sum, torch.cat((dc, rest)) (192 MB at
// 1 M Gaussians), the SH op, +0.5 and clamp — and the matching backward kernels — by one pass.
template <int K, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sh_fwd_fused_kernel(
    int n, int deg, const float *__restrict__ means, const float *__restrict__ cam_pos,
    const float *__restrict__ dc, int F, const float *__restrict__ rest, const int32_t *__restrict__ object_ids,
    const float *__restrict__ idft, const float *__restrict__ poses, int post, float *__restrict__ colors) {
    constexpr int KC = (K - 1) * 3, LS = (KC | 1);
    __shared__ float lds[WAVES][64 * (KC > 0 ? LS : 1)];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g0 = (blockIdx.x * WAVES + wave) * 64;
    const int cnt = max(0, min(64, n - g0));
    float *my = lds[wave];
    if constexpr (KC > 0) stage_rows<KC, LS>(my, rest + (size_t)g0 * KC, cnt, lane);
    __syncthreads();
    if (lane >= cnt) return;
    const int i = g0 + lane;
    float b[25];
    float vx, vy, vz;
    fused_viewdir(i, means, cam_pos, object_ids, poses, vx, vy, vz);
    const int nb = sh_bases(vx, vy, vz, deg, b);
    const float *w = idft + (size_t)(object_ids ? object_ids[i] : 0) * F;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    for (int f = 0; f < F; ++f) {
        const float wf = w[f];
        d0 += dc[((size_t)i * F + f) * 3 + 0] * wf;
        d1 += dc[((size_t)i * F + f) * 3 + 1] * wf;
        d2 += dc[((size_t)i * F + f) * 3 + 2] * wf;
    }
    float a0 = b[0] * d0, a1 = b[0] * d1, a2 = b[0] * d2;
    if constexpr (KC > 0) {
        const float *row = my + lane * LS;
#pragma unroll
        for (int k = 1; k < K; ++k) {
            if (k < nb) {
                a0 += b[k] * row[3 * (k - 1)];
                a1 += b[k] * row[3 * (k - 1) + 1];
                a2 += b[k] * row[3 * (k - 1) + 2];
            }
        }
    }
    if (post) { a0 = fmaxf(a0 + 0.5f, 0.f); a1 = fmaxf(a1 + 0.5f, 0.f); a2 = fmaxf(a2 + 0.5f, 0.f); }
    colors[3 * i] = a0; colors[3 * i + 1] = a1; colors[3 * i + 2] = a2;
}

template <int K, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sh_bwd_fused_kernel(
    int n, int deg, const float *__restrict__ means, const float *__restrict__ cam_pos, int F,
    const int32_t *__restrict__ object_ids, const float *__restrict__ idft, const float *__restrict__ poses,
    int post, const float *__restrict__ colors, const float *__restrict__ v_colors, float *__restrict__ v_dc,
    float *__restrict__ v_rest) {
    constexpr int KC = (K - 1) * 3, LS = (KC | 1);
    __shared__ float lds[WAVES][64 * (KC > 0 ? LS : 1)];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g0 = (blockIdx.x * WAVES + wave) * 64;
    const int cnt = max(0, min(64, n - g0));
    float *my = lds[wave];
    if (lane < cnt) {
        const int i = g0 + lane;
        float b[25];
        float vx, vy, vz;
        fused_viewdir(i, means, cam_pos, object_ids, poses, vx, vy, vz);
        const int nb = sh_bases(vx, vy, vz, deg, b);
        float v0 = v_colors[3 * i], v1 = v_colors[3 * i + 1], v2 = v_colors[3 * i + 2];
        if (post) {  // clamp(x + 0.5, min=0): gradient passes where the output is positive
            v0 = colors[3 * i] > 0.f ? v0 : 0.f;
            v1 = colors[3 * i + 1] > 0.f ? v1 : 0.f;
            v2 = colors[3 * i + 2] > 0.f ? v2 : 0.f;
        }
        const float *w = idft + (size_t)(object_ids ? object_ids[i] : 0) * F;
        for (int f = 0; f < F; ++f) {
            const float wf = w[f] * b[0];
            v_dc[((size_t)i * F + f) * 3 + 0] = wf * v0;
            v_dc[((size_t)i * F + f) * 3 + 1] = wf * v1;
            v_dc[((size_t)i * F + f) * 3 + 2] = wf * v2;
        }
        if constexpr (KC > 0) {
            float *row = my + lane * LS;
#pragma unroll
            for (int k = 1; k < K; ++k) {
                const float bk = (k < nb) ? b[k] : 0.f;
                row[3 * (k - 1)] = bk * v0; row[3 * (k - 1) + 1] = bk * v1; row[3 * (k - 1) + 2] = bk * v2;
            }
        }
    }
    __syncthreads();
    if constexpr (KC > 0) {
        float *dst = v_rest + (size_t)g0 * KC;
        const int total = cnt * KC;
        for (int e = lane; e < total; e += 64) {
            const int r = e / KC, c = e - r * KC;
            dst[e] = my[r * LS + c];
        }
    }
}

// ---- fused SH over UN-CONCATENATED sub-models (round 4) ---------------------------------------------------------------
// The scene graph keeps one `features_dc` / `features_rest` parameter per sub-model (background F = 1, objects
// F = fourier_features_dim); the fused front end above still wanted ONE [N,F,3] / [N,K-1,3] tensor, i.e. the caller's
// `torch.cat` (180 MB read + 180 MB write per step at 1 M Gaussians for features_rest, a zero-padded copy for
// features_dc) and autograd's split of the gradient on the way back.  Here the kernel takes the sub-models as they are:
// a by-value table of up to 32 parts (rows, first row in the aggregated order, pointers, Fourier dimension), the grid is
// the concatenation of the parts' own spans of 64 rows (a wave never straddles two parts), the part index IS the object
// index (pose row / idft row), and the backward writes each part's gradients into that part's own tensors.
constexpr int SH_MAX_PARTS = 32;
struct ShParts {
    int n_parts;
    int span0[SH_MAX_PARTS + 1];        // first 64-row span of part p in the grid; span0[n_parts] = total spans
    int row0[SH_MAX_PARTS];             // first row of part p in the aggregated (means / colours) order
    int rows[SH_MAX_PARTS];
    int F[SH_MAX_PARTS];
    const float *dc[SH_MAX_PARTS];      // [rows, F, 3]
    const float *rest[SH_MAX_PARTS];    // [rows, K-1, 3]
    float *v_dc[SH_MAX_PARTS];
    float *v_rest[SH_MAX_PARTS];
};

__device__ __forceinline__ int sh_part_of(const ShParts &P, int span) {
    int p = 0;
    while (p + 1 < P.n_parts && span >= P.span0[p + 1]) ++p;
    return p;
}

template <int K, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sh_fwd_parts_kernel(
    ShParts P, int deg, const float *__restrict__ means, const float *__restrict__ cam_pos,
    const float *__restrict__ idft, int idft_stride, const float *__restrict__ poses, int post,
    float *__restrict__ colors) {
    constexpr int KC = (K - 1) * 3, LS = (KC | 1);
    __shared__ float lds[WAVES][64 * (KC > 0 ? LS : 1)];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int span = blockIdx.x * WAVES + wave;
    const bool live = span < P.span0[P.n_parts];
    const int p = live ? sh_part_of(P, span) : 0;
    const int r0 = live ? (span - P.span0[p]) * 64 : 0;           // first row of the span INSIDE its part
    const int cnt = live ? max(0, min(64, P.rows[p] - r0)) : 0;
    float *my = lds[wave];
    if constexpr (KC > 0) stage_rows<KC, LS>(my, P.rest[p] + (size_t)r0 * KC, cnt, lane);
    __syncthreads();
    if (lane >= cnt) return;
    const int r = r0 + lane, i = P.row0[p] + r;
    float b[25];
    float m0 = means[3 * i], m1 = means[3 * i + 1], m2 = means[3 * i + 2];
    if (poses != nullptr) {                                      // local -> world (scene_graph.py:414); row p
        const float *Q = poses + 16 * (size_t)p;
        const float w0 = Q[0] * m0 + Q[1] * m1 + Q[2] * m2 + Q[9];
        const float w1 = Q[3] * m0 + Q[4] * m1 + Q[5] * m2 + Q[10];
        const float w2 = Q[6] * m0 + Q[7] * m1 + Q[8] * m2 + Q[11];
        m0 = w0; m1 = w1; m2 = w2;
    }
    const int nb = sh_bases(m0 - cam_pos[0], m1 - cam_pos[1], m2 - cam_pos[2], deg, b);
    const int F = P.F[p];
    const float *w = idft + (size_t)p * idft_stride;
    const float *dc = P.dc[p] + (size_t)r * F * 3;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    for (int f = 0; f < F; ++f) {
        const float wf = w[f];
        d0 += dc[3 * f] * wf; d1 += dc[3 * f + 1] * wf; d2 += dc[3 * f + 2] * wf;
    }
    float a0 = b[0] * d0, a1 = b[0] * d1, a2 = b[0] * d2;
    if constexpr (KC > 0) {
        const float *row = my + lane * LS;
#pragma unroll
        for (int k = 1; k < K; ++k) {
            if (k < nb) {
                a0 += b[k] * row[3 * (k - 1)];
                a1 += b[k] * row[3 * (k - 1) + 1];
                a2 += b[k] * row[3 * (k - 1) + 2];
            }
        }
    }
    if (post) { a0 = fmaxf(a0 + 0.5f, 0.f); a1 = fmaxf(a1 + 0.5f, 0.f); a2 = fmaxf(a2 + 0.5f, 0.f); }
    colors[3 * i] = a0; colors[3 * i + 1] = a1; colors[3 * i + 2] = a2;
}

template <int K, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sh_bwd_parts_kernel(
    ShParts P, int deg, const float *__restrict__ means, const float *__restrict__ cam_pos,
    const float *__restrict__ idft, int idft_stride, const float *__restrict__ poses, int post,
    const float *__restrict__ colors, const float *__restrict__ v_colors) {
    constexpr int KC = (K - 1) * 3, LS = (KC | 1);
    __shared__ float lds[WAVES][64 * (KC > 0 ? LS : 1)];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int span = blockIdx.x * WAVES + wave;
    const bool live = span < P.span0[P.n_parts];
    const int p = live ? sh_part_of(P, span) : 0;
    const int r0 = live ? (span - P.span0[p]) * 64 : 0;
    const int cnt = live ? max(0, min(64, P.rows[p] - r0)) : 0;
    float *my = lds[wave];
    if (lane < cnt) {
        const int r = r0 + lane, i = P.row0[p] + r;
        float b[25];
        float m0 = means[3 * i], m1 = means[3 * i + 1], m2 = means[3 * i + 2];
        if (poses != nullptr) {
            const float *Q = poses + 16 * (size_t)p;
            const float w0 = Q[0] * m0 + Q[1] * m1 + Q[2] * m2 + Q[9];
            const float w1 = Q[3] * m0 + Q[4] * m1 + Q[5] * m2 + Q[10];
            const float w2 = Q[6] * m0 + Q[7] * m1 + Q[8] * m2 + Q[11];
            m0 = w0; m1 = w1; m2 = w2;
        }
        const int nb = sh_bases(m0 - cam_pos[0], m1 - cam_pos[1], m2 - cam_pos[2], deg, b);
        float v0 = v_colors[3 * i], v1 = v_colors[3 * i + 1], v2 = v_colors[3 * i + 2];
        if (post) {  // clamp(x + 0.5, min=0): gradient passes where the output is positive
            v0 = colors[3 * i] > 0.f ? v0 : 0.f;
            v1 = colors[3 * i + 1] > 0.f ? v1 : 0.f;
            v2 = colors[3 * i + 2] > 0.f ? v2 : 0.f;
        }
        const int F = P.F[p];
        const float *w = idft + (size_t)p * idft_stride;
        float *vd = P.v_dc[p] + (size_t)r * F * 3;
        for (int f = 0; f < F; ++f) {
            const float wf = w[f] * b[0];
            vd[3 * f] = wf * v0; vd[3 * f + 1] = wf * v1; vd[3 * f + 2] = wf * v2;
        }
        if constexpr (KC > 0) {
            float *row = my + lane * LS;
#pragma unroll
            for (int k = 1; k < K; ++k) {
                const float bk = (k < nb) ? b[k] : 0.f;
                row[3 * (k - 1)] = bk * v0; row[3 * (k - 1) + 1] = bk * v1; row[3 * (k - 1) + 2] = bk * v2;
            }
        }
    }
    __syncthreads();
    if constexpr (KC > 0) {
        if (cnt > 0) {
            float *dst = P.v_rest[p] + (size_t)r0 * KC;
            const int total = cnt * KC;
            for (int e = lane; e < total; e += 64) {
                const int r = e / KC, c = e - r * KC;
                dst[e] = my[r * LS + c];
            }
        }
    }
}

// ---- multi-view SH backward for data-parallel training ---------------------------------
// v_coeffs[n,k,c] = sum_r basis_k(dir_{r,n}) * v_colors[r,n,c]: the SH gradient of ONE view is a rank-1 outer
// product (K bases x 3 channels) of two things that are tiny on the wire — the 3-float colour gradient and the
// view direction (or just the camera position).  Instead of all-reducing the dense [N,K,3] gradient
// (192 B/Gaussian, 2*(R-1)/R of it over xGMI), ranks all-gather v_colors (+ viewdirs or cam_pos) and every rank
// rebuilds the summed gradient locally with this kernel: 4x (cam_pos) or 2x (viewdirs) less traffic on the
// point-to-point links, and the extra compute is hidden behind the 192 B/Gaussian store this kernel does anyway.
template <int K, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sh_bwd_multi_kernel(
    int n, int deg, int n_views, const float *__restrict__ dirs_all /*[R,n,3] or null*/,
    const float *__restrict__ means /*[n,3] or null*/, const float *__restrict__ cam_pos /*[R,3] or null*/,
    const int32_t *__restrict__ object_ids, const float *__restrict__ poses,
    const float *__restrict__ v_colors_all /*[R,n,3]*/, float scale, float *__restrict__ v_coeffs,
    float *__restrict__ v_dc /*null: v_coeffs is [n,K,3]; else v_dc [n,3] takes band 0, v_coeffs [n,K-1,3] the rest*/) {
    constexpr int KC = K * 3, LS = (KC | 1);
    __shared__ float lds[WAVES][64 * LS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g0 = (blockIdx.x * WAVES + wave) * 64;
    const int cnt = max(0, min(64, n - g0));
    float *my = lds[wave];
    if (lane < cnt) {
        const int i = g0 + lane;
        float acc[KC];
#pragma unroll
        for (int e = 0; e < KC; ++e) acc[e] = 0.f;
        for (int r = 0; r < n_views; ++r) {
            float dx, dy, dz;
            if (dirs_all != nullptr) {
                const float *d = dirs_all + ((size_t)r * n + i) * 3;
                dx = d[0]; dy = d[1]; dz = d[2];
            } else {
                fused_viewdir(i, means, cam_pos + 3 * r, object_ids, poses, dx, dy, dz);
            }
            float b[25];
            const int nb = sh_bases(dx, dy, dz, deg, b);
            const float *v = v_colors_all + ((size_t)r * n + i) * 3;
            const float v0 = v[0], v1 = v[1], v2 = v[2];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float bk = (k < nb) ? b[k] : 0.f;
                acc[3 * k] += bk * v0; acc[3 * k + 1] += bk * v1; acc[3 * k + 2] += bk * v2;
            }
        }
        float *row = my + lane * LS;
#pragma unroll
        for (int e = 0; e < KC; ++e) row[e] = acc[e] * scale;
    }
    __syncthreads();
    const int total = cnt * KC;
    if (v_dc == nullptr) {
        float *dst = v_coeffs + (size_t)g0 * KC;
        for (int e = lane; e < total; e += 64) {
            const int r = e / KC, c = e - r * KC;
            dst[e] = my[r * LS + c];
        }
    } else {                       // the two leaves of the reference (features_dc, features_rest) get their own tensor
        for (int e = lane; e < cnt * 3; e += 64) v_dc[(size_t)g0 * 3 + e] = my[(e / 3) * LS + (e % 3)];
        if constexpr (KC > 3) {
            constexpr int RC = KC - 3;
            float *dst = v_coeffs + (size_t)g0 * RC;
            for (int e = lane; e < cnt * RC; e += 64) {
                const int r = e / RC, c = e - r * RC;
                dst[e] = my[r * LS + 3 + c];
            }
        }
    }
}

template <int K>
int launch_fwd(int n, int deg, const float *dirs, const float *coeffs, float *colors, hipStream_t s) {
    constexpr int WAVES = (K > 16) ? 2 : 4;  // keep static LDS under 64 KiB
    sgn_timing_begin(SGN_T_SH_FWD, s);
    // grid-stride over spans: enough workgroups to fill the LDS-limited residency (160 KB / slab bytes per CU) twice
    const int full = sgn_cdiv(n, WAVES * 64);
    const int grid = full < 256 * 6 ? full : 256 * 6;
    hipLaunchKernelGGL((sh_fwd_kernel<K, WAVES>), dim3(grid), dim3(WAVES * 64), 0, s, n, deg, dirs, coeffs, colors);
    sgn_timing_end(SGN_T_SH_FWD, s);
    return 0;
}
template <int K>
int launch_bwd(int n, int deg, const float *dirs, const float *v_colors, float *v_coeffs, hipStream_t s) {
    constexpr int WAVES = (K > 16) ? 2 : 4;
    sgn_timing_begin(SGN_T_SH_BWD, s);
    hipLaunchKernelGGL((sh_bwd_kernel<K, WAVES>), dim3(sgn_cdiv(n, WAVES * 64)), dim3(WAVES * 64), 0, s, n,
                       deg, dirs, v_colors, v_coeffs);
    sgn_timing_end(SGN_T_SH_BWD, s);
    return 0;
}

}  // namespace

SGN_EXPORT int sgn_sh_fwd(int n, int k, int degree, const float *viewdirs, const float *coeffs,
                          float *colors, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0, -1);
    SGN_ARG_CHECK(degree >= 0 && degree <= 4, -2);
    SGN_ARG_CHECK(k == 1 || k == 4 || k == 9 || k == 16 || k == 25, -3);
    SGN_ARG_CHECK((degree + 1) * (degree + 1) <= k, -4);
    if (n == 0) return 0;
    SGN_ARG_CHECK(viewdirs && coeffs && colors, -5);
    hipStream_t s = (hipStream_t)stream;
    switch (k) {
        case 1: launch_fwd<1>(n, degree, viewdirs, coeffs, colors, s); break;
        case 4: launch_fwd<4>(n, degree, viewdirs, coeffs, colors, s); break;
        case 9: launch_fwd<9>(n, degree, viewdirs, coeffs, colors, s); break;
        case 16: launch_fwd<16>(n, degree, viewdirs, coeffs, colors, s); break;
        default: launch_fwd<25>(n, degree, viewdirs, coeffs, colors, s); break;
    }
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_sh_bwd(int n, int k, int degree, const float *viewdirs, const float *v_colors,
                          float *v_coeffs, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0, -1);
    SGN_ARG_CHECK(degree >= 0 && degree <= 4, -2);
    SGN_ARG_CHECK(k == 1 || k == 4 || k == 9 || k == 16 || k == 25, -3);
    SGN_ARG_CHECK((degree + 1) * (degree + 1) <= k, -4);
    if (n == 0) return 0;
    SGN_ARG_CHECK(viewdirs && v_colors && v_coeffs, -5);
    hipStream_t s = (hipStream_t)stream;
    switch (k) {
        case 1: launch_bwd<1>(n, degree, viewdirs, v_colors, v_coeffs, s); break;
        case 4: launch_bwd<4>(n, degree, viewdirs, v_colors, v_coeffs, s); break;
        case 9: launch_bwd<9>(n, degree, viewdirs, v_colors, v_coeffs, s); break;
        case 16: launch_bwd<16>(n, degree, viewdirs, v_colors, v_coeffs, s); break;
        default: launch_bwd<25>(n, degree, viewdirs, v_colors, v_coeffs, s); break;
    }
    SGN_LAUNCH_CHECK();
    return 0;
}

template <int K>
static void launch_fwd_fused(int n, int deg, const float *means, const float *cam_pos, const float *dc, int F,
                             const float *rest, const int32_t *oid, const float *idft, const float *poses, int post,
                             float *colors, hipStream_t s) {
    constexpr int WAVES = (K > 16) ? 2 : 4;
    hipLaunchKernelGGL((sh_fwd_fused_kernel<K, WAVES>), dim3(sgn_cdiv(n, WAVES * 64)), dim3(WAVES * 64), 0, s, n, deg,
                       means, cam_pos, dc, F, rest, oid, idft, poses, post, colors);
}
template <int K>
static void launch_bwd_fused(int n, int deg, const float *means, const float *cam_pos, int F, const int32_t *oid,
                             const float *idft, const float *poses, int post, const float *colors, const float *v_colors, float *v_dc,
                             float *v_rest, hipStream_t s) {
    constexpr int WAVES = (K > 16) ? 2 : 4;
    hipLaunchKernelGGL((sh_bwd_fused_kernel<K, WAVES>), dim3(sgn_cdiv(n, WAVES * 64)), dim3(WAVES * 64), 0, s, n, deg,
                       means, cam_pos, F, oid, idft, poses, post, colors, v_colors, v_dc, v_rest);
}

SGN_EXPORT int sgn_sh_fwd_fused(int n, int k, int degree, const float *means, const float *cam_pos3,
                                const float *features_dc, int n_fourier, const float *features_rest,
                                const int32_t *object_ids, const float *idft, const float *poses, int post_half_clamp,
                                float *colors, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0, -1);
    SGN_ARG_CHECK(degree >= 0 && degree <= 4, -2);
    SGN_ARG_CHECK(k == 1 || k == 4 || k == 9 || k == 16 || k == 25, -3);
    SGN_ARG_CHECK((degree + 1) * (degree + 1) <= k, -4);
    SGN_ARG_CHECK(n_fourier >= 1 && n_fourier <= 16, -5);
    if (n == 0) return 0;
    SGN_ARG_CHECK(means && cam_pos3 && features_dc && idft && colors && (k == 1 || features_rest), -6);
    hipStream_t s = (hipStream_t)stream;
    sgn_timing_begin(SGN_T_SH_FWD, s);
    switch (k) {
        case 1: launch_fwd_fused<1>(n, degree, means, cam_pos3, features_dc, n_fourier, features_rest, object_ids, idft, poses, post_half_clamp, colors, s); break;
        case 4: launch_fwd_fused<4>(n, degree, means, cam_pos3, features_dc, n_fourier, features_rest, object_ids, idft, poses, post_half_clamp, colors, s); break;
        case 9: launch_fwd_fused<9>(n, degree, means, cam_pos3, features_dc, n_fourier, features_rest, object_ids, idft, poses, post_half_clamp, colors, s); break;
        case 16: launch_fwd_fused<16>(n, degree, means, cam_pos3, features_dc, n_fourier, features_rest, object_ids, idft, poses, post_half_clamp, colors, s); break;
        default: launch_fwd_fused<25>(n, degree, means, cam_pos3, features_dc, n_fourier, features_rest, object_ids, idft, poses, post_half_clamp, colors, s); break;
    }
    sgn_timing_end(SGN_T_SH_FWD, s);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_sh_bwd_fused(int n, int k, int degree, const float *means, const float *cam_pos3, int n_fourier,
                                const int32_t *object_ids, const float *idft, const float *poses, int post_half_clamp,
                                const float *colors, const float *v_colors, float *v_features_dc,
                                float *v_features_rest, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0, -1);
    SGN_ARG_CHECK(degree >= 0 && degree <= 4, -2);
    SGN_ARG_CHECK(k == 1 || k == 4 || k == 9 || k == 16 || k == 25, -3);
    SGN_ARG_CHECK((degree + 1) * (degree + 1) <= k, -4);
    SGN_ARG_CHECK(n_fourier >= 1 && n_fourier <= 16, -5);
    if (n == 0) return 0;
    SGN_ARG_CHECK(means && cam_pos3 && idft && colors && v_colors && v_features_dc && (k == 1 || v_features_rest), -6);
    hipStream_t s = (hipStream_t)stream;
    sgn_timing_begin(SGN_T_SH_BWD, s);
    switch (k) {
        case 1: launch_bwd_fused<1>(n, degree, means, cam_pos3, n_fourier, object_ids, idft, poses, post_half_clamp, colors, v_colors, v_features_dc, v_features_rest, s); break;
        case 4: launch_bwd_fused<4>(n, degree, means, cam_pos3, n_fourier, object_ids, idft, poses, post_half_clamp, colors, v_colors, v_features_dc, v_features_rest, s); break;
        case 9: launch_bwd_fused<9>(n, degree, means, cam_pos3, n_fourier, object_ids, idft, poses, post_half_clamp, colors, v_colors, v_features_dc, v_features_rest, s); break;
        case 16: launch_bwd_fused<16>(n, degree, means, cam_pos3, n_fourier, object_ids, idft, poses, post_half_clamp, colors, v_colors, v_features_dc, v_features_rest, s); break;
        default: launch_bwd_fused<25>(n, degree, means, cam_pos3, n_fourier, object_ids, idft, poses, post_half_clamp, colors, v_colors, v_features_dc, v_features_rest, s); break;
    }
    sgn_timing_end(SGN_T_SH_BWD, s);
    SGN_LAUNCH_CHECK();
    return 0;
}

template <int K>
static void launch_bwd_multi(int n, int deg, int R, const float *dirs_all, const float *means, const float *cam_pos,
                             const int32_t *oid, const float *poses, const float *v_all, float scale, float *out,
                             float *out_dc, hipStream_t s) {
    constexpr int WAVES = (K > 16) ? 2 : 4;
    hipLaunchKernelGGL((sh_bwd_multi_kernel<K, WAVES>), dim3(sgn_cdiv(n, WAVES * 64)), dim3(WAVES * 64), 0, s, n, deg, R,
                       dirs_all, means, cam_pos, oid, poses, v_all, scale, out, out_dc);
}

SGN_EXPORT int sgn_sh_bwd_multi(int n, int k, int degree, int n_views, const float *viewdirs_all, const float *means,
                                const float *cam_pos_all, const int32_t *object_ids, const float *poses,
                                const float *v_colors_all, float scale, float *v_coeffs, float *v_dc,
                                sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0 && n_views >= 1, -1);
    SGN_ARG_CHECK(degree >= 0 && degree <= 4, -2);
    SGN_ARG_CHECK(k == 1 || k == 4 || k == 9 || k == 16 || k == 25, -3);
    SGN_ARG_CHECK((degree + 1) * (degree + 1) <= k, -4);
    if (n == 0) return 0;
    SGN_ARG_CHECK(v_colors_all && (v_coeffs || (v_dc && k == 1)), -5);
    SGN_ARG_CHECK((viewdirs_all != nullptr) != (means != nullptr && cam_pos_all != nullptr), -6);
    hipStream_t s = (hipStream_t)stream;
    sgn_timing_begin(SGN_T_SH_BWD, s);
    switch (k) {
        case 1: launch_bwd_multi<1>(n, degree, n_views, viewdirs_all, means, cam_pos_all, object_ids, poses, v_colors_all, scale, v_coeffs, v_dc, s); break;
        case 4: launch_bwd_multi<4>(n, degree, n_views, viewdirs_all, means, cam_pos_all, object_ids, poses, v_colors_all, scale, v_coeffs, v_dc, s); break;
        case 9: launch_bwd_multi<9>(n, degree, n_views, viewdirs_all, means, cam_pos_all, object_ids, poses, v_colors_all, scale, v_coeffs, v_dc, s); break;
        case 16: launch_bwd_multi<16>(n, degree, n_views, viewdirs_all, means, cam_pos_all, object_ids, poses, v_colors_all, scale, v_coeffs, v_dc, s); break;
        default: launch_bwd_multi<25>(n, degree, n_views, viewdirs_all, means, cam_pos_all, object_ids, poses, v_colors_all, scale, v_coeffs, v_dc, s); break;
    }
    sgn_timing_end(SGN_T_SH_BWD, s);
    SGN_LAUNCH_CHECK();
    return 0;
}

// ---- Fourier DC fan-out (drop-in graph proofs over the scene graph's aggregation) -------------------------------------
// The scene graph builds an object's effective DC term as `sum(features_dc * idft[..., None], dim=1, keepdim=True)`
// (sgn_splatfacto_scene_graph.py:239-247) and concatenates those per-object terms (:356).  When the SH node's backward
// holds the gradient of the concatenated effective term, v_dc_eff [N,3], each Fourier leaf's gradient is
// v_features_dc[r, f, :] = v_dc_eff[row0 + r, :] * idft[f]: one launch for all the objects instead of three autograd
// nodes and two kernels per object.
namespace {
constexpr int FOURIER_MAX_PARTS = 32;
struct FourierParts {
    int n_parts;
    int cum[FOURIER_MAX_PARTS + 1];   // cumulative row counts of the listed parts
    int row0[FOURIER_MAX_PARTS];      // first row of the part in v_dc_eff
    int F[FOURIER_MAX_PARTS];
    const float *w[FOURIER_MAX_PARTS];
    float *out[FOURIER_MAX_PARTS];
};

__global__ __launch_bounds__(256) void fourier_dc_bwd_kernel(FourierParts P, const float *__restrict__ v_dc_eff) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= P.cum[P.n_parts]) return;
    int p = 0;
    while (p + 1 < P.n_parts && t >= P.cum[p + 1]) ++p;
    const int r = t - P.cum[p];
    const size_t row = (size_t)P.row0[p] + r;
    const float v0 = v_dc_eff[3 * row], v1 = v_dc_eff[3 * row + 1], v2 = v_dc_eff[3 * row + 2];
    const int F = P.F[p];
    float *o = P.out[p] + (size_t)r * F * 3;
    const float *w = P.w[p];
    for (int f = 0; f < F; ++f) {
        const float wf = w[f];
        o[3 * f] = v0 * wf; o[3 * f + 1] = v1 * wf; o[3 * f + 2] = v2 * wf;
    }
}
}  // namespace

SGN_EXPORT int sgn_fourier_dc_bwd(int n_parts, const int32_t *row0_host, const int32_t *rows_host,
                                  const int32_t *n_fourier_host, const float *const *weights_host,
                                  float *const *out_host, const float *v_dc_eff, sgn_stream_t stream) {
    SGN_ARG_CHECK(n_parts >= 0, -1);
    if (n_parts == 0) return 0;
    SGN_ARG_CHECK(row0_host && rows_host && n_fourier_host && weights_host && out_host && v_dc_eff, -2);
    for (int base = 0; base < n_parts; base += FOURIER_MAX_PARTS) {
        FourierParts P;
        P.n_parts = n_parts - base < FOURIER_MAX_PARTS ? n_parts - base : FOURIER_MAX_PARTS;
        P.cum[0] = 0;
        for (int i = 0; i < P.n_parts; ++i) {
            SGN_ARG_CHECK(rows_host[base + i] >= 0 && row0_host[base + i] >= 0, -3);
            SGN_ARG_CHECK(n_fourier_host[base + i] >= 1 && n_fourier_host[base + i] <= 64, -4);
            SGN_ARG_CHECK(weights_host[base + i] && out_host[base + i], -5);
            P.cum[i + 1] = P.cum[i] + rows_host[base + i];
            P.row0[i] = row0_host[base + i];
            P.F[i] = n_fourier_host[base + i];
            P.w[i] = weights_host[base + i];
            P.out[i] = out_host[base + i];
        }
        if (P.cum[P.n_parts] == 0) continue;
        hipLaunchKernelGGL(fourier_dc_bwd_kernel, dim3(sgn_cdiv(P.cum[P.n_parts], 256)), dim3(256), 0,
                           (hipStream_t)stream, P, v_dc_eff);
    }
    SGN_LAUNCH_CHECK();
    return 0;
}

// ---- fused SH over un-concatenated sub-models: entry points
namespace {
static int fill_parts(ShParts &P, int n_parts, const int32_t *rows_host, const int32_t *n_fourier_host,
                      const float *const *dc_host, const float *const *rest_host, float *const *v_dc_host,
                      float *const *v_rest_host, int k, bool backward) {
    if (n_parts < 1 || n_parts > SH_MAX_PARTS || !rows_host || !n_fourier_host) return -1;
    if (backward ? !v_dc_host : !dc_host) return -1;
    P.n_parts = n_parts;
    int span = 0, row = 0;
    for (int p = 0; p < n_parts; ++p) {
        if (rows_host[p] < 0 || n_fourier_host[p] < 1 || n_fourier_host[p] > 16) return -2;
        P.span0[p] = span; P.row0[p] = row; P.rows[p] = rows_host[p]; P.F[p] = n_fourier_host[p];
        P.dc[p] = backward ? nullptr : dc_host[p];
        P.rest[p] = (backward || k == 1) ? nullptr : rest_host[p];
        P.v_dc[p] = backward ? v_dc_host[p] : nullptr;
        P.v_rest[p] = (backward && k > 1) ? v_rest_host[p] : nullptr;
        if (rows_host[p] > 0) {
            if (!backward && (!dc_host[p] || (k > 1 && !rest_host[p]))) return -3;
            if (backward && (!v_dc_host[p] || (k > 1 && !v_rest_host[p]))) return -3;
        }
        span += (rows_host[p] + 63) / 64;
        row += rows_host[p];
    }
    P.span0[n_parts] = span;
    for (int p = n_parts; p < SH_MAX_PARTS; ++p) { P.row0[p] = 0; P.rows[p] = 0; P.F[p] = 1; P.dc[p] = P.rest[p] = nullptr; P.v_dc[p] = P.v_rest[p] = nullptr; }
    for (int p = n_parts + 1; p <= SH_MAX_PARTS; ++p) P.span0[p] = span;
    return span;
}
template <int K>
static void launch_fwd_parts(const ShParts &P, int spans, int deg, const float *means, const float *cam, const float *idft,
                             int stride, const float *poses, int post, float *colors, hipStream_t s) {
    constexpr int WAVES = (K > 16) ? 2 : 4;
    hipLaunchKernelGGL((sh_fwd_parts_kernel<K, WAVES>), dim3(sgn_cdiv(spans, WAVES)), dim3(WAVES * 64), 0, s, P, deg, means,
                       cam, idft, stride, poses, post, colors);
}
template <int K>
static void launch_bwd_parts(const ShParts &P, int spans, int deg, const float *means, const float *cam, const float *idft,
                             int stride, const float *poses, int post, const float *colors, const float *v_colors,
                             hipStream_t s) {
    constexpr int WAVES = (K > 16) ? 2 : 4;
    hipLaunchKernelGGL((sh_bwd_parts_kernel<K, WAVES>), dim3(sgn_cdiv(spans, WAVES)), dim3(WAVES * 64), 0, s, P, deg, means,
                       cam, idft, stride, poses, post, colors, v_colors);
}
}  // namespace

SGN_EXPORT int sgn_sh_fwd_parts(int n_parts, const int32_t *rows_host, const int32_t *n_fourier_host,
                                const float *const *features_dc_host, const float *const *features_rest_host, int k,
                                int degree, const float *means, const float *cam_pos3, const float *idft,
                                int idft_stride, const float *poses, int post_half_clamp, float *colors,
                                sgn_stream_t stream) {
    SGN_ARG_CHECK(degree >= 0 && degree <= 4, -2);
    SGN_ARG_CHECK(k == 1 || k == 4 || k == 9 || k == 16 || k == 25, -3);
    SGN_ARG_CHECK((degree + 1) * (degree + 1) <= k, -4);
    SGN_ARG_CHECK(means && cam_pos3 && idft && colors && idft_stride >= 1 && (k == 1 || features_rest_host), -6);
    ShParts P;
    const int spans = fill_parts(P, n_parts, rows_host, n_fourier_host, features_dc_host, features_rest_host, nullptr, nullptr,
                                 k, false);
    SGN_ARG_CHECK(spans >= 0, -7);
    if (spans == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    sgn_timing_begin(SGN_T_SH_FWD, s);
    switch (k) {
        case 1: launch_fwd_parts<1>(P, spans, degree, means, cam_pos3, idft, idft_stride, poses, post_half_clamp, colors, s); break;
        case 4: launch_fwd_parts<4>(P, spans, degree, means, cam_pos3, idft, idft_stride, poses, post_half_clamp, colors, s); break;
        case 9: launch_fwd_parts<9>(P, spans, degree, means, cam_pos3, idft, idft_stride, poses, post_half_clamp, colors, s); break;
        case 16: launch_fwd_parts<16>(P, spans, degree, means, cam_pos3, idft, idft_stride, poses, post_half_clamp, colors, s); break;
        default: launch_fwd_parts<25>(P, spans, degree, means, cam_pos3, idft, idft_stride, poses, post_half_clamp, colors, s); break;
    }
    sgn_timing_end(SGN_T_SH_FWD, s);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_sh_bwd_parts(int n_parts, const int32_t *rows_host, const int32_t *n_fourier_host, int k, int degree,
                                const float *means, const float *cam_pos3, const float *idft, int idft_stride,
                                const float *poses, int post_half_clamp, const float *colors, const float *v_colors,
                                float *const *v_features_dc_host, float *const *v_features_rest_host,
                                sgn_stream_t stream) {
    SGN_ARG_CHECK(degree >= 0 && degree <= 4, -2);
    SGN_ARG_CHECK(k == 1 || k == 4 || k == 9 || k == 16 || k == 25, -3);
    SGN_ARG_CHECK((degree + 1) * (degree + 1) <= k, -4);
    SGN_ARG_CHECK(means && cam_pos3 && idft && colors && v_colors && v_features_dc_host && idft_stride >= 1 &&
                      (k == 1 || v_features_rest_host), -6);
    ShParts P;
    const int spans = fill_parts(P, n_parts, rows_host, n_fourier_host, nullptr, nullptr, v_features_dc_host,
                                 v_features_rest_host, k, true);
    SGN_ARG_CHECK(spans >= 0, -7);
    if (spans == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    sgn_timing_begin(SGN_T_SH_BWD, s);
    switch (k) {
        case 1: launch_bwd_parts<1>(P, spans, degree, means, cam_pos3, idft, idft_stride, poses, post_half_clamp, colors, v_colors, s); break;
        case 4: launch_bwd_parts<4>(P, spans, degree, means, cam_pos3, idft, idft_stride, poses, post_half_clamp, colors, v_colors, s); break;
        case 9: launch_bwd_parts<9>(P, spans, degree, means, cam_pos3, idft, idft_stride, poses, post_half_clamp, colors, v_colors, s); break;
        case 16: launch_bwd_parts<16>(P, spans, degree, means, cam_pos3, idft, idft_stride, poses, post_half_clamp, colors, v_colors, s); break;
        default: launch_bwd_parts<25>(P, spans, degree, means, cam_pos3, idft, idft_stride, poses, post_half_clamp, colors, v_colors, s); break;
    }
    sgn_timing_end(SGN_T_SH_BWD, s);
    SGN_LAUNCH_CHECK();
    return 0;
}
