// Internal helpers shared by the HIP translation units of libsgnrast.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "sgn_rast.h"

#define SGN_EXPORT extern "C" __attribute__((visibility("default")))

void sgn_set_error(const char *fmt, ...);
int sgn_timing_enabled();
void sgn_timing_begin(int slot, void *stream);
void sgn_timing_end(int slot, void *stream);
int sgn_fork_events(hipEvent_t *fork, hipEvent_t *join);   // api.cpp: cached per (thread, device)

#define SGN_ARG_CHECK(cond, code)                                              \
    do {                                                                       \
        if (!(cond)) {                                                         \
            sgn_set_error("%s: argument check failed: %s", __func__, #cond);   \
            return (code);                                                     \
        }                                                                      \
    } while (0)

#define SGN_HIP_CHECK(expr)                                                    \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) {                                                \
            sgn_set_error("%s: %s -> %s", __func__, #expr, hipGetErrorString(e_)); \
            return (int)e_;                                                    \
        }                                                                      \
    } while (0)

#define SGN_LAUNCH_CHECK()                                                     \
    do {                                                                       \
        hipError_t e_ = hipGetLastError();                                     \
        if (e_ != hipSuccess) {                                                \
            sgn_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return (int)e_;                                                    \
        }                                                                      \
    } while (0)

// project.hip: sgn_project_fwd with upstream's unit-quaternion assertion riding the kernel (quat_flag nullptr: none); a
// failing row stores quat_stamp into *quat_flag, and — quat_ok != nullptr — the kernel's first lane stores it into
// *quat_ok ("this launch's stores land where the host looks"); both system-scope
int sgn_project_fwd_checked(int n, const float *means3d, const float *scales, float glob_scale, const float *quats,
                            const float *viewmat12, float fx, float fy, float cx, float cy, int img_h, int img_w,
                            int block_width, float clip_thresh, float *cov3d, float *xys, float *depths, int32_t *radii,
                            float *conics, float *compensation, int32_t *num_tiles_hit, int32_t *quat_flag,
                            float quat_tol, int32_t quat_stamp, int32_t *quat_ok, int semantics, sgn_stream_t stream);

static inline int sgn_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// float -> int with v_cvt_i32_f32 semantics (saturating, NaN -> 0); the C oracle spells the
// same rule out (oracle/c/sgn_oracle.c f2i).
__device__ __forceinline__ int sgn_f2i(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

// gsplat helpers.cuh get_tile_bbox / get_bbox: inclusive min, exclusive max.
// `semantics` & SGN_SEM_BBOX_ADD_AFTER_CAST: the max side is (int)(c + r) + 1 (gsplat/_torch_impl.py) instead of the
// default (int)(c + r + 1) (gsplat helpers.cuh); include/sgn_rast.h "upstream-variant semantics".
__device__ __forceinline__ int sgn_bbox_max(float v, int semantics) {
    if (semantics & SGN_SEM_BBOX_ADD_AFTER_CAST) {
        const int t = sgn_f2i(v);
        return t == 2147483647 ? t : t + 1;
    }
    return sgn_f2i(v + 1.0f);
}
__device__ __forceinline__ void sgn_tile_bbox(float cx, float cy, float radius, int tiles_x,
                                              int tiles_y, int block, int &mnx, int &mny, int &mxx,
                                              int &mxy, int semantics = 0) {
    const float fb = (float)block;
    const float tcx = cx / fb, tcy = cy / fb, tr = radius / fb;
    mnx = min(max(0, sgn_f2i(tcx - tr)), tiles_x);
    mxx = min(max(0, sgn_bbox_max(tcx + tr, semantics)), tiles_x);
    mny = min(max(0, sgn_f2i(tcy - tr)), tiles_y);
    mxy = min(max(0, sgn_bbox_max(tcy + tr, semantics)), tiles_y);
}
