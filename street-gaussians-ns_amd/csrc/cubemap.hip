// cubemap.hip — sky cube-map lookup with bilinear filtering (forward + texture gradient), gfx950.
//
// replaces nvdiffrast `dr.texture(base[None], dirs, filter_mode='linear',
// boundary_mode='cube')`, the only other CUDA-only native dependency on the reference's training path
// (EnvLight, street_gaussians_ns/sgn_splatfacto.py:109-150, called at :876).  nvdiffrast is a third-party
// package whose submodule directory is EMPTY in /root/reference (dependencies/nvdiffrast), so this restates its
// published behaviour (PARITY UNPINNED, see oracle/torch_oracle.py:cube_texture):
//   * faces in OpenGL order +x,-x,+y,-y,+z,-z; face = major axis (z wins ties over y over x, as upstream's
//     indexCubeMap); (s,t) follow the GL cube-map table; u,v = s/(2|m|)+1/2, clamped to [0,1];
//   * bilinear taps at texel centres (u*R-0.5); a tap that leaves the face is taken from the adjacent face
//     (seamless edges: the tap's texel centre is re-projected through the cube), a tap that leaves through a
//     corner is dropped and the other three weights are renormalised;
//   * non-finite directions give 0.
// One lane per direction; HBM/L2-bound gather of 4 x C floats.
#include "sgn_common.h"

namespace {

// direction -> (face, u, v); returns -1 for invalid input.  Mirrors nvdiffrast's indexCubeMap.
__device__ __forceinline__ int cube_face_uv(float x, float y, float z, float &u, float &v) {
    const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    int idx;
    float c, sx = x, sy = y;
    if (az > fmaxf(ax, ay)) { idx = 4; c = z; }
    else if (ay > ax)       { idx = 2; c = y; sy = z; }
    else                    { idx = 0; c = x; sx = z; }
    if (c < 0.f) idx += 1;
    const float m = 0.5f / fabsf(c);
    const float m0 = (idx == 0 || idx == 5) ? -m : m;   // sign table of the GL cube-map s coordinate
    const float m1 = (idx != 2) ? -m : m;               // and of t
    u = sx * m0 + 0.5f;
    v = sy * m1 + 0.5f;
    if (!(fabsf(u) < 3.0e38f) || !(fabsf(v) < 3.0e38f)) return -1;
    u = fminf(fmaxf(u, 0.f), 1.f);
    v = fminf(fmaxf(v, 0.f), 1.f);
    return idx;
}

// (face, u, v) on the (possibly extended) face plane -> 3D direction; inverse of cube_face_uv
__device__ __forceinline__ void cube_dir(int face, float u, float v, float &x, float &y, float &z) {
    const float s = 2.f * u - 1.f, t = 2.f * v - 1.f;
    switch (face) {
        case 0: x = 1.f;  y = -t; z = -s; break;
        case 1: x = -1.f; y = -t; z = s;  break;
        case 2: x = s;  y = 1.f;  z = t;  break;
        case 3: x = s;  y = -1.f; z = -t; break;
        case 4: x = s;  y = -t; z = 1.f;  break;
        default: x = -s; y = -t; z = -1.f; break;
    }
}

struct Taps {
    int off[4];     // texel offsets (face*R*R + iy*R + ix), -1 = dropped
    float w[4];
};

__device__ __forceinline__ Taps cube_taps(float dx, float dy, float dz, int R) {
    Taps T;
#pragma unroll
    for (int k = 0; k < 4; ++k) { T.off[k] = -1; T.w[k] = 0.f; }
    float u, v;
    const int face = cube_face_uv(dx, dy, dz, u, v);
    if (face < 0) return T;
    const float fu = u * (float)R - 0.5f, fv = v * (float)R - 0.5f;
    const float flu = floorf(fu), flv = floorf(fv);
    const int iu0 = (int)flu, iv0 = (int)flv;
    const float au = fu - flu, av = fv - flv;
    float wsum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int iu = iu0 + (k & 1), iv = iv0 + (k >> 1);
        const float w = ((k & 1) ? au : 1.f - au) * ((k >> 1) ? av : 1.f - av);
        const bool ou = iu < 0 || iu >= R, ov = iv < 0 || iv >= R;
        if (ou && ov) continue;                       // left through a corner: dropped, renormalised below
        int f = face, ix = iu, iy = iv;
        if (ou || ov) {                               // left through an edge: re-project the texel centre
            float x, y, z, u2, v2;
            cube_dir(face, ((float)iu + 0.5f) / (float)R, ((float)iv + 0.5f) / (float)R, x, y, z);
            f = cube_face_uv(x, y, z, u2, v2);
            ix = min(max((int)floorf(u2 * (float)R), 0), R - 1);
            iy = min(max((int)floorf(v2 * (float)R), 0), R - 1);
        }
        T.off[k] = (f * R + iy) * R + ix;
        T.w[k] = w;
        wsum += w;
    }
    if (wsum > 0.f && wsum < 1.f) {
        const float inv = 1.f / wsum;
#pragma unroll
        for (int k = 0; k < 4; ++k) T.w[k] *= inv;
    }
    return T;
}

__global__ __launch_bounds__(256) void cube_fwd_kernel(int64_t n, int R, int C, const float *__restrict__ tex,
                                                       const float *__restrict__ dirs, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Taps T = cube_taps(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], R);
    for (int c = 0; c < C; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (T.off[k] >= 0) acc = fmaf(T.w[k], tex[(size_t)T.off[k] * C + c], acc);
        out[i * C + c] = acc;
    }
}

__global__ __launch_bounds__(256) void cube_bwd_kernel(int64_t n, int R, int C, const float *__restrict__ dirs,
                                                       const float *__restrict__ v_out, float *__restrict__ v_tex) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Taps T = cube_taps(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], R);
    for (int c = 0; c < C; ++c) {
        const float g = v_out[i * C + c];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (T.off[k] >= 0) atomicAdd(v_tex + (size_t)T.off[k] * C + c, T.w[k] * g);
    }
}

// Fused EnvLight: pixel -> camera ray -> world -> GL axes -> cube lookup, no [H,W,3] direction tensor in HBM.
// Follows EnvLight.get_world_directions/forward (sgn_splatfacto.py:117-150) operation by operation:
// d = normalize(((u-cx+ju)/fx, (v-cy+jv)/fy, 1)); w = c2w[:3,:3] d; l = to_opengl w = (w.x, w.z, -w.y).
struct SkyCam {
    int h, w;
    float fx, fy, cx, cy;
    const float *c2w;   // device, row-major, row stride `ld` (>= 3)
    int ld;
    const float *jitter;   // device [2,h,w] (u then v offsets in [0,1)) or null = 0.5 (eval)
};

__device__ __forceinline__ void sky_dir(const SkyCam &c, int64_t i, float &lx, float &ly, float &lz) {
    const int py = (int)(i / c.w), px = (int)(i - (int64_t)py * c.w);
    const float ju = c.jitter ? c.jitter[i] : 0.5f;
    const float jv = c.jitter ? c.jitter[(int64_t)c.h * c.w + i] : 0.5f;
    const float dx = ((float)px - c.cx + ju) / c.fx, dy = ((float)py - c.cy + jv) / c.fy;
    const float nrm = fmaxf(sqrtf(fmaf(dx, dx, fmaf(dy, dy, 1.f))), 1e-12f);
    const float x = dx / nrm, y = dy / nrm, z = 1.f / nrm;
    const float *R = c.c2w;
    const int ld = c.ld;
    const float wx = fmaf(R[2], z, fmaf(R[1], y, R[0] * x));
    const float wy = fmaf(R[ld + 2], z, fmaf(R[ld + 1], y, R[ld] * x));
    const float wz = fmaf(R[2 * ld + 2], z, fmaf(R[2 * ld + 1], y, R[2 * ld] * x));
    lx = wx; ly = wz; lz = -wy;
}

__global__ __launch_bounds__(256) void sky_fwd_kernel(SkyCam cam, int R, int C, const float *__restrict__ tex,
                                                      float *__restrict__ out) {
    const int64_t n = (int64_t)cam.h * cam.w;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float lx, ly, lz;
    sky_dir(cam, i, lx, ly, lz);
    const Taps T = cube_taps(lx, ly, lz, R);
    for (int c = 0; c < C; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (T.off[k] >= 0) acc = fmaf(T.w[k], tex[(size_t)T.off[k] * C + c], acc);
        out[i * C + c] = acc;
    }
}

__global__ __launch_bounds__(256) void sky_bwd_kernel(SkyCam cam, int R, int C, const float *__restrict__ v_out,
                                                      float *__restrict__ v_tex) {
    const int64_t n = (int64_t)cam.h * cam.w;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float lx, ly, lz;
    sky_dir(cam, i, lx, ly, lz);
    const Taps T = cube_taps(lx, ly, lz, R);
    for (int c = 0; c < C; ++c) {
        const float g = v_out[i * C + c];
        if (g == 0.f) continue;                       // pixels fully covered by Gaussians contribute nothing
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (T.off[k] >= 0) atomicAdd(v_tex + (size_t)T.off[k] * C + c, T.w[k] * g);
    }
}

// Fused sky + compositing (sgn_splatfacto.py:969-972): out = min(rgb,1)*alpha + sky*(1-alpha), C == 3.
__global__ __launch_bounds__(256) void sky_blend_fwd_kernel(SkyCam cam, int R, const float *__restrict__ tex,
                                                            const float *__restrict__ rgb,
                                                            const float *__restrict__ alpha,
                                                            float *__restrict__ out, float *__restrict__ sky_out) {
    const int64_t n = (int64_t)cam.h * cam.w;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float lx, ly, lz;
    sky_dir(cam, i, lx, ly, lz);
    const Taps T = cube_taps(lx, ly, lz, R);
    const float a = alpha[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (T.off[k] >= 0) acc = fmaf(T.w[k], tex[(size_t)T.off[k] * 3 + c], acc);
        if (sky_out) sky_out[i * 3 + c] = acc;
        out[i * 3 + c] = fminf(rgb[i * 3 + c], 1.f) * a + acc * (1.f - a);
    }
}

__global__ __launch_bounds__(256) void sky_blend_bwd_kernel(SkyCam cam, int R, const float *__restrict__ tex,
                                                            const float *__restrict__ rgb,
                                                            const float *__restrict__ alpha,
                                                            const float *__restrict__ v_out,
                                                            float *__restrict__ v_rgb, float *__restrict__ v_alpha,
                                                            float *__restrict__ v_tex) {
    const int64_t n = (int64_t)cam.h * cam.w;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float lx, ly, lz;
    sky_dir(cam, i, lx, ly, lz);
    const Taps T = cube_taps(lx, ly, lz, R);
    const float a = alpha[i];
    float va = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float sky = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (T.off[k] >= 0) sky = fmaf(T.w[k], tex[(size_t)T.off[k] * 3 + c], sky);
        const float g = v_out[i * 3 + c], r = rgb[i * 3 + c];
        v_rgb[i * 3 + c] = (r <= 1.f) ? g * a : 0.f;          // torch.clamp(max=1) passes the gradient at r == 1
        va = fmaf(g, fminf(r, 1.f) - sky, va);
        const float gs = g * (1.f - a);
        if (gs != 0.f) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (T.off[k] >= 0) atomicAdd(v_tex + (size_t)T.off[k] * 3 + c, T.w[k] * gs);
        }
    }
    v_alpha[i] = va;
}

}  // namespace

SGN_EXPORT int sgn_sky_blend_fwd(int h, int w, float fx, float fy, float cx, float cy, const float *c2w, int c2w_ld,
                                 const float *jitter, int resolution, const float *tex, const float *rgb,
                                 const float *alpha, float *out, float *sky_out, sgn_stream_t stream) {
    SGN_ARG_CHECK(h >= 0 && w >= 0 && resolution > 0 && c2w_ld >= 3, -1);
    if ((int64_t)h * w == 0) return 0;
    SGN_ARG_CHECK(c2w && tex && rgb && alpha && out, -2);
    SkyCam cam{h, w, fx, fy, cx, cy, c2w, c2w_ld, jitter};
    sgn_timing_begin(SGN_T_SKY_FWD, (void *)stream);
    hipLaunchKernelGGL(sky_blend_fwd_kernel, dim3(sgn_cdiv((int64_t)h * w, 256)), dim3(256), 0, (hipStream_t)stream,
                       cam, resolution, tex, rgb, alpha, out, sky_out);
    sgn_timing_end(SGN_T_SKY_FWD, (void *)stream);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_sky_blend_bwd(int h, int w, float fx, float fy, float cx, float cy, const float *c2w, int c2w_ld,
                                 const float *jitter, int resolution, const float *tex, const float *rgb,
                                 const float *alpha, const float *v_out, float *v_rgb, float *v_alpha, float *v_tex,
                                 sgn_stream_t stream) {
    SGN_ARG_CHECK(h >= 0 && w >= 0 && resolution > 0 && c2w_ld >= 3, -1);
    SGN_ARG_CHECK(v_tex != nullptr, -2);
    hipStream_t s = (hipStream_t)stream;
    SGN_HIP_CHECK(hipMemsetAsync(v_tex, 0, (size_t)6 * resolution * resolution * 3 * sizeof(float), s));
    if ((int64_t)h * w == 0) return 0;
    SGN_ARG_CHECK(c2w && tex && rgb && alpha && v_out && v_rgb && v_alpha, -3);
    SkyCam cam{h, w, fx, fy, cx, cy, c2w, c2w_ld, jitter};
    sgn_timing_begin(SGN_T_SKY_BWD, (void *)s);
    hipLaunchKernelGGL(sky_blend_bwd_kernel, dim3(sgn_cdiv((int64_t)h * w, 256)), dim3(256), 0, s, cam, resolution,
                       tex, rgb, alpha, v_out, v_rgb, v_alpha, v_tex);
    sgn_timing_end(SGN_T_SKY_BWD, (void *)s);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_sky_fwd(int h, int w, float fx, float fy, float cx, float cy, const float *c2w, int c2w_ld,
                           const float *jitter, int resolution, int channels, const float *tex, float *out,
                           sgn_stream_t stream) {
    SGN_ARG_CHECK(h >= 0 && w >= 0 && resolution > 0 && channels > 0 && c2w_ld >= 3, -1);
    if ((int64_t)h * w == 0) return 0;
    SGN_ARG_CHECK(c2w && tex && out, -2);
    SkyCam cam{h, w, fx, fy, cx, cy, c2w, c2w_ld, jitter};
    sgn_timing_begin(SGN_T_SKY_FWD, (void *)stream);
    hipLaunchKernelGGL(sky_fwd_kernel, dim3(sgn_cdiv((int64_t)h * w, 256)), dim3(256), 0, (hipStream_t)stream, cam,
                       resolution, channels, tex, out);
    sgn_timing_end(SGN_T_SKY_FWD, (void *)stream);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_sky_bwd(int h, int w, float fx, float fy, float cx, float cy, const float *c2w, int c2w_ld,
                           const float *jitter, int resolution, int channels, const float *v_out, float *v_tex,
                           sgn_stream_t stream) {
    SGN_ARG_CHECK(h >= 0 && w >= 0 && resolution > 0 && channels > 0 && c2w_ld >= 3, -1);
    SGN_ARG_CHECK(v_tex != nullptr, -2);
    hipStream_t s = (hipStream_t)stream;
    SGN_HIP_CHECK(hipMemsetAsync(v_tex, 0, (size_t)6 * resolution * resolution * channels * sizeof(float), s));
    if ((int64_t)h * w == 0) return 0;
    SGN_ARG_CHECK(c2w && v_out, -3);
    SkyCam cam{h, w, fx, fy, cx, cy, c2w, c2w_ld, jitter};
    sgn_timing_begin(SGN_T_SKY_BWD, (void *)s);
    hipLaunchKernelGGL(sky_bwd_kernel, dim3(sgn_cdiv((int64_t)h * w, 256)), dim3(256), 0, s, cam, resolution,
                       channels, v_out, v_tex);
    sgn_timing_end(SGN_T_SKY_BWD, (void *)s);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_cube_texture_fwd(int64_t n, int resolution, int channels, const float *tex,
                                    const float *dirs, float *out, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0 && resolution > 0 && channels > 0, -1);
    if (n == 0) return 0;
    SGN_ARG_CHECK(tex && dirs && out, -2);
    sgn_timing_begin(SGN_T_SKY_FWD, (void *)stream);
    hipLaunchKernelGGL(cube_fwd_kernel, dim3(sgn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, resolution,
                       channels, tex, dirs, out);
    sgn_timing_end(SGN_T_SKY_FWD, (void *)stream);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_cube_texture_bwd(int64_t n, int resolution, int channels, const float *dirs,
                                    const float *v_out, float *v_tex, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0 && resolution > 0 && channels > 0, -1);
    SGN_ARG_CHECK(v_tex != nullptr, -2);
    hipStream_t s = (hipStream_t)stream;
    SGN_HIP_CHECK(hipMemsetAsync(v_tex, 0, (size_t)6 * resolution * resolution * channels * sizeof(float), s));
    if (n == 0) return 0;
    SGN_ARG_CHECK(dirs && v_out, -3);
    sgn_timing_begin(SGN_T_SKY_BWD, (void *)s);
    hipLaunchKernelGGL(cube_bwd_kernel, dim3(sgn_cdiv(n, 256)), dim3(256), 0, s, n, resolution, channels, dirs, v_out,
                       v_tex);
    sgn_timing_end(SGN_T_SKY_BWD, (void *)s);
    SGN_LAUNCH_CHECK();
    return 0;
}
