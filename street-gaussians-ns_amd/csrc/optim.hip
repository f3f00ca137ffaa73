// optim.hip — multi-tensor Adam step for the per-Gaussian parameter groups, gfx950.
//
// SURVEY.md §8f row 3.  The reference trains with one torch.optim.Adam per parameter group (nerfstudio
// AdamOptimizerConfig, eps 1e-15: street_gaussians_ns/sgn_config.py:71-108 — xyz, features_dc, features_rest,
// opacity, scaling, rotation, + sky_sphere), i.e. six to sixty small optimisers per step on the scene graph.  Here ONE
// launch updates every tensor: a by-value table of (param, grad, exp_avg, exp_avg_sq, n, hyper-parameters) rows, a
// workgroup -> (tensor, chunk) map, float4 accesses.  Pure HBM stream: 28 B per element (4 reads + 3 writes).
//
// Arithmetic follows torch.optim.Adam's single-tensor path operation by operation (amsgrad = False, weight_decay = 0,
// maximize = False):   exp_avg.lerp_(grad, 1 - beta1);  exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2);
//   denom = exp_avg_sq.sqrt() / sqrt(1 - beta2^t) + eps;  param.addcdiv_(exp_avg, denom, value = -lr / (1 - beta1^t))
// (bias corrections are computed on the host in double, like torch's Python scalars).  torch is installed in the
// test container, so this one is pinned against the real thing (tests/test_gpu_optim.py, tests/test_optim_host.py).
#include "sgn_common.h"

namespace {

constexpr int ADAM_MAX_TENSORS = 24;
constexpr int ADAM_CHUNK = 256 * 4 * 4;     // elements per workgroup: 256 threads x 4 float4

struct AdamTable {
    float *p[ADAM_MAX_TENSORS];
    const float *g[ADAM_MAX_TENSORS];
    float *m[ADAM_MAX_TENSORS];
    float *v[ADAM_MAX_TENSORS];
    long long n[ADAM_MAX_TENSORS];
    float step_size[ADAM_MAX_TENSORS];      // lr / (1 - beta1^t)
    float inv_bc2_sqrt[ADAM_MAX_TENSORS];   // 1 / sqrt(1 - beta2^t)   (torch divides; see kernel)
    float bc2_sqrt[ADAM_MAX_TENSORS];
    float w1[ADAM_MAX_TENSORS];             // 1 - beta1, rounded from double like torch's Python scalar
    float beta2[ADAM_MAX_TENSORS], w2[ADAM_MAX_TENSORS], eps[ADAM_MAX_TENSORS];
    int blk_start[ADAM_MAX_TENSORS + 1];    // first workgroup of each tensor
    int count;
};

__device__ __forceinline__ void adam_elem(float &p, float g, float &m, float &v, float w1, float b2, float w2,
                                          float bc2_sqrt, float eps, float step_size) {
    m = fmaf(w1, g - m, m);                       // lerp_(grad, 1 - beta1): m + w (g - m), weight < 0.5 form
    v = fmaf(w2 * g, g, v * b2);                  // mul_(beta2).addcmul_(g, g, value = 1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);              // addcdiv_(m, denom, value = -step_size)
}

__global__ __launch_bounds__(256) void adam_kernel(AdamTable T) {
    // which tensor does this workgroup belong to?  (<= 24 rows: linear search in SGPRs)
    int t = 0;
    while (t + 1 < T.count && (int)blockIdx.x >= T.blk_start[t + 1]) ++t;
    const long long n = T.n[t];
    const long long base = (long long)(blockIdx.x - T.blk_start[t]) * ADAM_CHUNK;
    float *__restrict__ p = T.p[t];
    const float *__restrict__ g = T.g[t];
    float *__restrict__ m = T.m[t];
    float *__restrict__ v = T.v[t];
    const float w1 = T.w1[t], b2 = T.beta2[t], w2 = T.w2[t];
    const float bc2 = T.bc2_sqrt[t], eps = T.eps[t], ss = T.step_size[t];
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const long long i = base + ((long long)it * 256 + threadIdx.x) * 4;
        if (i >= n) break;
        if (vec && i + 4 <= n) {
            float4 P = *reinterpret_cast<float4 *>(p + i), M = *reinterpret_cast<float4 *>(m + i);
            float4 V = *reinterpret_cast<float4 *>(v + i);
            const float4 G = *reinterpret_cast<const float4 *>(g + i);
            adam_elem(P.x, G.x, M.x, V.x, w1, b2, w2, bc2, eps, ss);
            adam_elem(P.y, G.y, M.y, V.y, w1, b2, w2, bc2, eps, ss);
            adam_elem(P.z, G.z, M.z, V.z, w1, b2, w2, bc2, eps, ss);
            adam_elem(P.w, G.w, M.w, V.w, w1, b2, w2, bc2, eps, ss);
            *reinterpret_cast<float4 *>(p + i) = P;
            *reinterpret_cast<float4 *>(m + i) = M;
            *reinterpret_cast<float4 *>(v + i) = V;
        } else {
            for (long long j = i; j < n && j < i + 4; ++j) {
                float P = p[j], M = m[j], V = v[j];
                adam_elem(P, g[j], M, V, w1, b2, w2, bc2, eps, ss);
                p[j] = P; m[j] = M; v[j] = V;
            }
        }
    }
}

}  // namespace

// One Adam step over `count` tensors.  All array arguments are HOST arrays of length `count` (hyper-parameters in
// double, as torch keeps them); the pointers inside
// params/grads/exp_avgs/exp_avg_sqs are DEVICE pointers to contiguous fp32 tensors of numel[i] elements.
// steps[i] is the step number AFTER the increment (torch: state['step'] += 1 first), >= 1.
SGN_EXPORT int sgn_adam_step(int count, float *const *params, const float *const *grads, float *const *exp_avgs,
                             float *const *exp_avg_sqs, const int64_t *numel, const double *lr, const double *beta1,
                             const double *beta2, const double *eps, const int64_t *steps, sgn_stream_t stream) {
    SGN_ARG_CHECK(count >= 0, -1);
    if (count == 0) return 0;
    SGN_ARG_CHECK(params && grads && exp_avgs && exp_avg_sqs && numel && lr && beta1 && beta2 && eps && steps, -2);
    hipStream_t s = (hipStream_t)stream;
    sgn_timing_begin(SGN_T_ADAM, (void *)s);
    // `i` is consumed ACROSS launches: empty tensors are skipped without taking a table slot, so a launch may
    // consume more than ADAM_MAX_TENSORS input rows — restarting the next launch at first + ADAM_MAX_TENSORS stepped
    // some rows twice (advisor finding, round 1).
    int i = 0;
    while (i < count) {
        AdamTable T;
        T.count = 0;
        int blocks = 0;
        for (; i < count && T.count < ADAM_MAX_TENSORS; ++i) {
            SGN_ARG_CHECK(numel[i] >= 0 && steps[i] >= 1, -3);
            if (numel[i] == 0) continue;
            SGN_ARG_CHECK(params[i] && grads[i] && exp_avgs[i] && exp_avg_sqs[i], -4);
            const int k = T.count++;
            T.p[k] = params[i]; T.g[k] = grads[i]; T.m[k] = exp_avgs[i]; T.v[k] = exp_avg_sqs[i];
            T.n[k] = numel[i];
            const double bc1 = 1.0 - pow(beta1[i], (double)steps[i]);
            const double bc2 = 1.0 - pow(beta2[i], (double)steps[i]);
            T.step_size[k] = (float)(lr[i] / bc1);
            T.bc2_sqrt[k] = (float)sqrt(bc2);
            T.inv_bc2_sqrt[k] = (float)(1.0 / sqrt(bc2));
            // torch hands Python doubles to the elementwise ops, which round them to fp32 once: 1 - 0.999 in double
            // -> 0.001000000047f, not 1.f - 0.999f = 0.00099998713f (a 1.3e-5 relative difference in exp_avg_sq)
            T.w1[k] = (float)(1.0 - beta1[i]); T.beta2[k] = (float)beta2[i]; T.w2[k] = (float)(1.0 - beta2[i]);
            T.eps[k] = (float)eps[i];
            T.blk_start[k] = blocks;
            blocks += (int)sgn_cdiv(numel[i], ADAM_CHUNK);
        }
        T.blk_start[T.count] = blocks;
        if (blocks > 0) hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, T);
    }
    sgn_timing_end(SGN_T_ADAM, (void *)s);
    SGN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Per-step densification statistics (SplatfactoModel.after_train, street_gaussians_ns/sgn_splatfacto.py:513-541):
//   grads = xys.grad.norm(dim=-1); visible = radii > 0
//   first call : xys_grad_norm = grads (every Gaussian), vis_counts = 1, max_2Dsize = 0 then the max below
//   afterwards : vis_counts[visible] += 1; xys_grad_norm[visible] += grads[visible]
//   always     : max_2Dsize[visible] = max(max_2Dsize[visible], radii[visible] / max(H, W))
// The reference does this with boolean-mask indexing (nonzero + index_put: several kernels and a host sync per
// line); here it is one coalesced pass with no sync.
namespace {
__global__ __launch_bounds__(256) void densify_stats_kernel(int n, const float *__restrict__ xys_grad,
                                                            const int32_t *__restrict__ radii, float max_dim,
                                                            int first, float *__restrict__ grad_norm,
                                                            float *__restrict__ vis_counts,
                                                            float *__restrict__ max_2dsize) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gx = xys_grad[2 * i], gy = xys_grad[2 * i + 1];
    const float g = sqrtf(gx * gx + gy * gy);
    const int rad = radii[i];
    const bool vis = rad > 0;
    if (first) {
        grad_norm[i] = g;
        vis_counts[i] = 1.f;
        max_2dsize[i] = vis ? fmaxf(0.f, (float)rad / max_dim) : 0.f;
    } else if (vis) {
        vis_counts[i] = vis_counts[i] + 1.f;
        grad_norm[i] = g + grad_norm[i];
        max_2dsize[i] = fmaxf(max_2dsize[i], (float)rad / max_dim);
    }
}
}  // namespace

SGN_EXPORT int sgn_densify_stats(int n, const float *xys_grad, const int32_t *radii, float max_dim, int first,
                                 float *xys_grad_norm, float *vis_counts, float *max_2dsize, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0 && max_dim > 0.f, -1);
    if (n == 0) return 0;
    SGN_ARG_CHECK(xys_grad && radii && xys_grad_norm && vis_counts && max_2dsize, -2);
    hipLaunchKernelGGL(densify_stats_kernel, dim3(sgn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, xys_grad,
                       radii, max_dim, first, xys_grad_norm, vis_counts, max_2dsize);
    SGN_LAUNCH_CHECK();
    return 0;
}
