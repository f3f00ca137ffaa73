// loss.hip — fused photometric loss of the reference's training step: L1 + SSIM (forward + gradient), gfx950.
//
// SURVEY.md §8f row 3.  Replaces, at 1920x1280x3, `torch.abs(gt - rgb).mean()` plus
// `pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3)(gt, rgb)` — ten depthwise conv2d launches
// forward and their autograd graph backward — reached at street_gaussians_ns/sgn_splatfacto.py:1084-1087.
// pytorch_msssim is a PyPI dependency that is not vendored by the reference (PARITY UNPINNED); the definition
// restated here and in oracle/torch_oracle.py:ssim: 11-tap Gaussian window (sigma 1.5), separable, NO padding
// (valid region (H-10) x (W-10)), K = (0.01, 0.03), data_range 1, mean over space then channels.
//
// One workgroup per 16x16 pixel tile, HWC images as the rasterizer produces them (no NCHW permute copy):
//   forward : 26x26 input patch -> LDS, horizontal then vertical 11-tap pass for the five moments
//             (x, y, x^2, y^2, xy), SSIM per pixel, block-reduced sums (one atomic per block), and the three
//             partial derivatives dS/d mu_x, dS/d E[x^2], dS/d E[xy] stored for the backward;
//   backward: 26x26 patch of those three maps -> LDS, same separable filter (transposed valid correlation:
//             zero outside the valid domain), grad = G*A + 2x G*B + y G*C, fused with the L1 sign term.
// HBM-bound by design: forward reads 2 images and writes 3 maps, backward reads 3 maps + 2 images and writes 1.
#include "sgn_common.h"

namespace {

constexpr int WIN = 11, HALO = WIN - 1, TS = 16, PS = TS + HALO;   // tile 16, patch 26

struct Win { float g[WIN]; };

__device__ __forceinline__ float block_sum(float v, float *lds4) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds4[wave] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

constexpr int ROWF = PS * 3;          // floats per patch row with the three channels interleaved (HWC)
constexpr int ROWS_ = ROWF + 3;       // LDS row stride (odd: 3q + c walks distinct banks, rows shifted)

// pred/gt [H,W,3]; per-workgroup partial sums of |gt - pred| (all pixels) and of ssim_map (valid region);
// dmaps [3 maps][3 channels][H-10][W-10]: dS/d mu_pred, dS/d E[pred^2], dS/d E[pred*gt] (NULL: no backward wanted)
__global__ __launch_bounds__(256) void l1_ssim_fwd_kernel(int H, int W, Win win, float C1, float C2, float cmax,
                                                          const float *__restrict__ pred,
                                                          const float *__restrict__ gt,
                                                          float *__restrict__ partials,
                                                          float *__restrict__ dmaps) {
    __shared__ float px[PS][ROWS_], py[PS][ROWS_];            // HWC patch, all three channels: contiguous row loads
    __shared__ float hm[5][PS][TS + 1];
    __shared__ float lds4[4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
    const int Ho = H - HALO, Wo = W - HALO;
    const int ox = x0 + tx, oy = y0 + ty;                     // pixel (L1) = output position (SSIM)
    const bool in_img = ox < W && oy < H, in_out = ox < Wo && oy < Ho;
    const int rowlen = W * 3;
    {   // all patch loads are issued before the first LDS write (a load -> store loop is compiled to one HBM round
        // trip per iteration: s_waitcnt vmcnt(0) after every global_load)
        constexpr int NL = (PS * ROWF + 255) / 256;
        float va[NL], vb[NL];
#pragma unroll
        for (int it = 0; it < NL; ++it) {
            const int e = tid + it * 256;
            const int r = e / ROWF, j = e - r * ROWF;
            const int iy = y0 + r, gx = x0 * 3 + j;
            const bool ok = e < PS * ROWF && iy < H && gx < rowlen;
            const size_t idx = (size_t)iy * rowlen + gx;
            va[it] = ok ? fminf(pred[idx], cmax) : 0.f;      // fused torch.clamp(rgb, max=1) (sgn_splatfacto.py:969)
            vb[it] = ok ? gt[idx] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NL; ++it) {
            const int e = tid + it * 256;
            if (e < PS * ROWF) {
                const int r = e / ROWF, j = e - r * ROWF;
                px[r][j] = va[it];
                py[r][j] = vb[it];
            }
        }
    }
    __syncthreads();
    float l1 = 0.f, ss = 0.f;
    if (in_img) {
#pragma unroll
        for (int c = 0; c < 3; ++c) l1 += fabsf(py[ty][tx * 3 + c] - px[ty][tx * 3 + c]);
    }
    const size_t plane = (size_t)Ho * Wo;
    for (int c = 0; c < 3; ++c) {
        if (c) __syncthreads();                               // hm of the previous channel fully consumed
        for (int e = tid; e < PS * TS; e += 256) {            // horizontal pass: 26 rows x 16 output columns
            const int r = e >> 4, q = e & 15;
            float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const float a = px[r][(q + k) * 3 + c], b = py[r][(q + k) * 3 + c], g = win.g[k];
                const float ga = g * a, gb = g * b;
                sx += ga; sy += gb;
                sxx = fmaf(ga, a, sxx); syy = fmaf(gb, b, syy); sxy = fmaf(ga, b, sxy);
            }
            hm[0][r][q] = sx; hm[1][r][q] = sy; hm[2][r][q] = sxx; hm[3][r][q] = syy; hm[4][r][q] = sxy;
        }
        __syncthreads();
        if (in_out) {
            float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const float g = win.g[k];
                mu1 = fmaf(g, hm[0][ty + k][tx], mu1);
                mu2 = fmaf(g, hm[1][ty + k][tx], mu2);
                e11 = fmaf(g, hm[2][ty + k][tx], e11);
                e22 = fmaf(g, hm[3][ty + k][tx], e22);
                e12 = fmaf(g, hm[4][ty + k][tx], e12);
            }
            const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
            const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2;
            const float B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s1 + s2 + C2;
            const float iB1 = 1.f / B1, iB2 = 1.f / B2;
            const float S = A1 * A2 * iB1 * iB2;
            ss += S;
            if (dmaps) {
                // partials w.r.t. the filtered moments of `pred` (mu1, E[x^2], E[xy]); sigma's depend on mu1 too
                const float dS_ds1 = -S * iB2;                  // via sigma1^2
                const float dS_ds12 = 2.f * A1 * iB1 * iB2;     // via sigma12
                const float dS_dmu1 = 2.f * mu2 * A2 * iB1 * iB2 - 2.f * mu1 * S * iB1 - 2.f * mu1 * dS_ds1 -
                                      mu2 * dS_ds12;
                const size_t o = (size_t)c * plane + (size_t)oy * Wo + ox;   // planar: row stores are contiguous
                dmaps[o] = dS_dmu1;
                dmaps[3 * plane + o] = dS_ds1;
                dmaps[6 * plane + o] = dS_ds12;
            }
        }
    }
    const float bl1 = block_sum(l1, lds4);
    const float bss = block_sum(ss, lds4);
    if (tid == 0) {   // one slot per workgroup: 9600 same-address float atomics cost more than the whole kernel
        const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partials[2 * b] = bl1;
        partials[2 * b + 1] = bss;
    }
}

// out3 = [Ll1 = mean |gt - pred|, ssim mean, (1 - lambda) Ll1 + lambda (1 - ssim)]: the means and the reference's
// weighted sum (sgn_splatfacto.py:1086-1087) come out of the reduction itself instead of a dozen scalar torch kernels
__global__ __launch_bounds__(256) void l1_ssim_reduce_kernel(int nblk, const float *__restrict__ partials,
                                                             float inv_n_l1, float inv_n_ss, float lambda,
                                                             float *__restrict__ out3) {
    __shared__ float lds4[4];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) { a += partials[2 * i]; b += partials[2 * i + 1]; }
    a = block_sum(a, lds4);
    b = block_sum(b, lds4);
    if (threadIdx.x == 0) {
        const float l1 = a * inv_n_l1, ss = b * inv_n_ss;
        out3[0] = l1;
        out3[1] = ss;
        out3[2] = (1.f - lambda) * l1 + lambda * (1.f - ss);
    }
}

// v_pred [H,W,3] = gl1 * sign(pred - gt) / (3HW) + gss / (3 Ho Wo) * (G*A + 2 pred G*B + gt G*C)
// gscale (device, 2 floats): upstream gradients of the two means (d loss / d Ll1, d loss / d ssim)
__global__ __launch_bounds__(256) void l1_ssim_bwd_kernel(int H, int W, Win win, float cmax,
                                                          const float *__restrict__ pred,
                                                          const float *__restrict__ gt,
                                                          const float *__restrict__ dmaps,
                                                          const float *__restrict__ gscale,
                                                          float *__restrict__ v_pred) {
    __shared__ float pm[3][PS][PS + 1];
    __shared__ float hm[3][PS][TS + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
    const int Ho = H - HALO, Wo = W - HALO;
    const int qx = x0 + tx, qy = y0 + ty;
    const bool in_img = qx < W && qy < H;
    const float w_l1 = gscale[0] / (3.f * (float)H * (float)W);
    const float w_ss = gscale[1] / (3.f * (float)Ho * (float)Wo);
    const size_t plane = (size_t)Ho * Wo;
    const size_t pix = ((size_t)qy * W + qx) * 3;
    float out[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < 3; ++c) {
        __syncthreads();
        // outputs p that see pixel q: p in [q - 10, q]; patch row r <-> p_y = y0 - 10 + r
        {
            constexpr int NL = (PS * PS + 255) / 256;
            float v0[NL], v1[NL], v2[NL];
#pragma unroll
            for (int it = 0; it < NL; ++it) {
                const int e = tid + it * 256;
                const int r = e / PS, q = e - r * PS;
                const int py_ = y0 - HALO + r, px_ = x0 - HALO + q;
                const bool ok = e < PS * PS && py_ >= 0 && px_ >= 0 && py_ < Ho && px_ < Wo;
                const size_t o = (size_t)c * plane + (size_t)py_ * Wo + px_;
                v0[it] = ok ? dmaps[o] : 0.f;
                v1[it] = ok ? dmaps[3 * plane + o] : 0.f;
                v2[it] = ok ? dmaps[6 * plane + o] : 0.f;
            }
#pragma unroll
            for (int it = 0; it < NL; ++it) {
                const int e = tid + it * 256;
                if (e < PS * PS) {
                    const int r = e / PS, q = e - r * PS;
                    pm[0][r][q] = v0[it]; pm[1][r][q] = v1[it]; pm[2][r][q] = v2[it];
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < PS * TS; e += 256) {
            const int r = e >> 4, q = e & 15;
            float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {      // p_x = q_x - 10 + k  has weight g[10 - k] (window tap q - p)
                const float g = win.g[WIN - 1 - k];
                a = fmaf(g, pm[0][r][q + k], a);
                b = fmaf(g, pm[1][r][q + k], b);
                d = fmaf(g, pm[2][r][q + k], d);
            }
            hm[0][r][q] = a; hm[1][r][q] = b; hm[2][r][q] = d;
        }
        __syncthreads();
        if (in_img) {
            float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const float g = win.g[WIN - 1 - k];
                a = fmaf(g, hm[0][ty + k][tx], a);
                b = fmaf(g, hm[1][ty + k][tx], b);
                d = fmaf(g, hm[2][ty + k][tx], d);
            }
            const float raw = pred[pix + c], y = gt[pix + c];
            const float x = fminf(raw, cmax);
            const float df = x - y;
            const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            const float gval = fmaf(w_l1, sgn, w_ss * (a + 2.f * x * b + y * d));
            out[c] = (raw <= cmax) ? gval : 0.f;          // clamp passes the gradient where raw <= max
        }
    }
    if (in_img) {
        v_pred[pix] = out[0]; v_pred[pix + 1] = out[1]; v_pred[pix + 2] = out[2];
    }
}

Win make_window(float sigma) {
    Win w;
    float fs = 0.f;
    for (int k = 0; k < WIN; ++k) {
        // same fp32 recipe as pytorch_msssim._fspecial_gauss_1d (fp32 exp, fp32 normalisation)
        const float c = (float)(k - WIN / 2);
        w.g[k] = expf(-(c * c) / (2.f * sigma * sigma));
        fs += w.g[k];
    }
    for (int k = 0; k < WIN; ++k) w.g[k] /= fs;
    return w;
}

}  // namespace

static size_t partial_bytes(int h, int w) {
    return (((size_t)sgn_cdiv(w, TS) * sgn_cdiv(h, TS) * 2 * sizeof(float)) + 255) & ~(size_t)255;
}

// workspace = [per-workgroup partial sums][3 maps x 3 channels x (h-10) x (w-10) floats]
SGN_EXPORT size_t sgn_l1_ssim_workspace_bytes(int h, int w, int with_grad) {
    if (h <= HALO || w <= HALO) return 256;
    return partial_bytes(h, w) + (with_grad ? (size_t)9 * (h - HALO) * (w - HALO) * sizeof(float) : 0);
}

SGN_EXPORT int sgn_l1_ssim_fwd(int h, int w, const float *pred, const float *gt, float data_range, float clamp_max,
                               float ssim_lambda, float *out3 /*device: [Ll1, ssim, (1-l) Ll1 + l (1-ssim)]*/,
                               int with_grad, void *ws, size_t ws_bytes, sgn_stream_t stream) {
    SGN_ARG_CHECK(h > HALO && w > HALO, -1);      // pytorch_msssim asserts the image is larger than the window
    SGN_ARG_CHECK(pred && gt && out3 && ws, -2);
    SGN_ARG_CHECK(ws_bytes >= sgn_l1_ssim_workspace_bytes(h, w, with_grad), -3);
    hipStream_t s = (hipStream_t)stream;
    float *partials = (float *)ws;
    float *dmaps = with_grad ? (float *)((char *)ws + partial_bytes(h, w)) : nullptr;
    const Win win = make_window(1.5f);
    const float C1 = (0.01f * data_range) * (0.01f * data_range), C2 = (0.03f * data_range) * (0.03f * data_range);
    const dim3 grid(sgn_cdiv(w, TS), sgn_cdiv(h, TS));
    sgn_timing_begin(SGN_T_LOSS_FWD, (void *)s);
    hipLaunchKernelGGL(l1_ssim_fwd_kernel, grid, dim3(256), 0, s, h, w, win, C1, C2, clamp_max, pred, gt, partials,
                       dmaps);
    hipLaunchKernelGGL(l1_ssim_reduce_kernel, dim3(1), dim3(256), 0, s, (int)(grid.x * grid.y), partials,
                       1.f / (3.f * (float)h * (float)w), 1.f / (3.f * (float)(h - HALO) * (float)(w - HALO)),
                       ssim_lambda, out3);
    sgn_timing_end(SGN_T_LOSS_FWD, (void *)s);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_l1_ssim_bwd(int h, int w, const float *pred, const float *gt, float clamp_max, const void *ws,
                               const float *gscale2 /*device: [d loss/d Ll1, d loss/d ssim]*/, float *v_pred,
                               sgn_stream_t stream) {
    SGN_ARG_CHECK(h > HALO && w > HALO, -1);
    SGN_ARG_CHECK(pred && gt && ws && gscale2 && v_pred, -2);
    hipStream_t s = (hipStream_t)stream;
    const float *dmaps = (const float *)((const char *)ws + partial_bytes(h, w));
    const Win win = make_window(1.5f);
    sgn_timing_begin(SGN_T_LOSS_BWD, (void *)s);
    hipLaunchKernelGGL(l1_ssim_bwd_kernel, dim3(sgn_cdiv(w, TS), sgn_cdiv(h, TS)), dim3(256), 0, s, h, w, win,
                       clamp_max, pred, gt, dmaps, gscale2, v_pred);
    sgn_timing_end(SGN_T_LOSS_BWD, (void *)s);
    SGN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Accumulation regularisers of the reference's loss dictionary, both means over the H*W pixels
// of an [H,W,1] accumulation image the rasterizer produced:
//   losses["sky_accumulation"]        = mult * (sky_mask * accumulation).mean(),  sky_mask = (gt_semantic == SKY)
//                                       street_gaussians_ns/sgn_splatfacto.py:1090-1093 (gt_semantic int64 [H,W,1])
//   losses["object_acc_entropy_loss"] = mult * -(o log o + (1 - o) log(1 - o)).mean(),  o = clamp(object_acc, 1e-5, 1 - 1e-5)
//                                       street_gaussians_ns/sgn_splatfacto_scene_graph.py:386-389
// In torch these are ~10 elementwise launches forward and as many backward over 2.46 M pixels; here one streaming pass
// each way computes both (either may be absent), per-workgroup partial sums, means formed in the reduction.
namespace {

constexpr int ACC_THREADS = 256, ACC_MAX_BLOCKS = 2048;

__device__ __forceinline__ bool sem_is(const void *sem, int sem_bytes, size_t i, long long value) {
    if (sem_bytes == 8) return ((const long long *)sem)[i] == value;
    if (sem_bytes == 4) return (long long)((const int *)sem)[i] == value;
    return (long long)((const unsigned char *)sem)[i] == value;
}

__device__ __forceinline__ float acc_clamp_lo() { return 1e-5f; }
__device__ __forceinline__ float acc_clamp_hi() { return (float)(1.0 - 1e-5); }   // torch rounds the Python scalar to fp32

__global__ __launch_bounds__(ACC_THREADS) void acc_losses_fwd_kernel(long long n, const float *__restrict__ acc,
                                                                     const void *__restrict__ sem, int sem_bytes,
                                                                     long long sky_value,
                                                                     const float *__restrict__ obj,
                                                                     float *__restrict__ partials) {
    __shared__ float lds4[4];
    float sky = 0.f, ent = 0.f;
    const long long stride = (long long)gridDim.x * ACC_THREADS;
    for (long long i = (long long)blockIdx.x * ACC_THREADS + threadIdx.x; i < n; i += stride) {
        if (acc) {
            const float a = acc[i];
            sky += sem_is(sem, sem_bytes, (size_t)i, sky_value) ? a : 0.f;
        }
        if (obj) {
            const float o = fminf(fmaxf(obj[i], acc_clamp_lo()), acc_clamp_hi());
            const float q = 1.f - o;
            ent -= o * logf(o) + q * logf(q);
        }
    }
    const float bs = block_sum(sky, lds4);
    const float be = block_sum(ent, lds4);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = bs;
        partials[2 * blockIdx.x + 1] = be;
    }
}

__global__ __launch_bounds__(ACC_THREADS) void acc_losses_reduce_kernel(int nblk, const float *__restrict__ partials,
                                                                        double inv_n, float *__restrict__ out2) {
    __shared__ double lds[ACC_THREADS];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblk; i += ACC_THREADS) { a += (double)partials[2 * i]; b += (double)partials[2 * i + 1]; }
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        lds[threadIdx.x] = pass ? b : a;
        __syncthreads();
        for (int d = ACC_THREADS / 2; d >= 1; d >>= 1) {
            if ((int)threadIdx.x < d) lds[threadIdx.x] += lds[threadIdx.x + d];
            __syncthreads();
        }
        if (threadIdx.x == 0) out2[pass] = (float)(lds[0] * inv_n);
    }
}

// v_acc[i] = g0 / n * [sem == sky];  v_obj[i] = g1 / n * (log(1 - o) - log(o)) inside the clamp, 0 outside
// (torch.clamp passes the gradient where min <= x <= max)
__global__ __launch_bounds__(ACC_THREADS) void acc_losses_bwd_kernel(long long n, const void *__restrict__ sem,
                                                                     int sem_bytes, long long sky_value,
                                                                     const float *__restrict__ obj,
                                                                     const float *__restrict__ gscale, float n_f,
                                                                     float *__restrict__ v_acc,
                                                                     float *__restrict__ v_obj) {
    const float g0 = gscale[0] / n_f, g1 = gscale[1] / n_f;      // mean's backward: grad / numel, a true division
    const long long stride = (long long)gridDim.x * ACC_THREADS;
    for (long long i = (long long)blockIdx.x * ACC_THREADS + threadIdx.x; i < n; i += stride) {
        if (v_acc) v_acc[i] = sem_is(sem, sem_bytes, (size_t)i, sky_value) ? g0 : 0.f;
        if (v_obj) {
            const float raw = obj[i];
            const bool inside = raw >= acc_clamp_lo() && raw <= acc_clamp_hi();
            const float o = fminf(fmaxf(raw, acc_clamp_lo()), acc_clamp_hi());
            v_obj[i] = inside ? g1 * (logf(1.f - o) - logf(o)) : 0.f;
        }
    }
}

inline int acc_blocks(int64_t n) {
    const int b = sgn_cdiv(n, ACC_THREADS);
    return b < 1 ? 1 : (b > ACC_MAX_BLOCKS ? ACC_MAX_BLOCKS : b);
}

}  // namespace

SGN_EXPORT size_t sgn_acc_losses_workspace_bytes(int64_t n_pixels) {
    return (((size_t)acc_blocks(n_pixels) * 2 * sizeof(float)) + 255) & ~(size_t)255;
}

SGN_EXPORT int sgn_acc_losses_fwd(int64_t n_pixels, const float *accumulation, const void *semantic, int sem_bytes,
                                  int64_t sky_value, const float *object_acc,
                                  float *out2 /*device: [mean(sky_mask * acc), mean entropy]*/, void *ws, size_t ws_bytes,
                                  sgn_stream_t stream) {
    SGN_ARG_CHECK(n_pixels > 0, -1);
    SGN_ARG_CHECK(out2 && ws && (accumulation || object_acc), -2);
    SGN_ARG_CHECK(!accumulation || (semantic && (sem_bytes == 1 || sem_bytes == 4 || sem_bytes == 8)), -3);
    SGN_ARG_CHECK(ws_bytes >= sgn_acc_losses_workspace_bytes(n_pixels), -4);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = acc_blocks(n_pixels);
    sgn_timing_begin(SGN_T_LOSS_FWD, (void *)s);
    hipLaunchKernelGGL(acc_losses_fwd_kernel, dim3(nblk), dim3(ACC_THREADS), 0, s, (long long)n_pixels, accumulation,
                       semantic, sem_bytes, (long long)sky_value, object_acc, (float *)ws);
    hipLaunchKernelGGL(acc_losses_reduce_kernel, dim3(1), dim3(ACC_THREADS), 0, s, nblk, (const float *)ws,
                       1.0 / (double)n_pixels, out2);
    sgn_timing_end(SGN_T_LOSS_FWD, (void *)s);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_acc_losses_bwd(int64_t n_pixels, const void *semantic, int sem_bytes, int64_t sky_value,
                                  const float *object_acc,
                                  const float *gscale2 /*device: [d loss/d sky mean, d loss/d entropy mean]*/,
                                  float *v_accumulation, float *v_object_acc, sgn_stream_t stream) {
    SGN_ARG_CHECK(n_pixels > 0, -1);
    SGN_ARG_CHECK(gscale2 && (v_accumulation || v_object_acc), -2);
    SGN_ARG_CHECK(!v_accumulation || (semantic && (sem_bytes == 1 || sem_bytes == 4 || sem_bytes == 8)), -3);
    SGN_ARG_CHECK(!v_object_acc || object_acc, -4);
    hipStream_t s = (hipStream_t)stream;
    sgn_timing_begin(SGN_T_LOSS_BWD, (void *)s);
    hipLaunchKernelGGL(acc_losses_bwd_kernel, dim3(acc_blocks(n_pixels)), dim3(ACC_THREADS), 0, s, (long long)n_pixels,
                       semantic, sem_bytes, (long long)sky_value, object_acc, gscale2, (float)n_pixels,
                       v_accumulation, v_object_acc);
    sgn_timing_end(SGN_T_LOSS_BWD, (void *)s);
    SGN_LAUNCH_CHECK();
    return 0;
}
