// project.hip — EWA projection of 3D Gaussians (forward + backward), gfx950.
//
// Replaces gsplat 0.1.x forward.cu:project_gaussians_forward_kernel and
// backward.cu:project_gaussians_backward_kernel (third-party, un-vendored; semantics in
// SURVEY.md A.1 / A.5), reached from the reference at sgn_splatfacto.py:860-873.
//
// HBM-bound streaming kernels: one lane per Gaussian, AoS rows are consecutive across lanes so
// every global_load_dwordx{2,3,4} of a wave covers one contiguous span (fwd: 40 B in / 60 B out
// per Gaussian; bwd: 112 B in / 76 B out).  This TU is compiled with -ffp-contract=off and every
// expression is written in the operation order of the arithmetic contract, so
// xys / depths / radii / num_tiles_hit — and therefore the 64-bit sort keys — are bit-identical
// to the CPU oracle.
#include "sgn_common.h"

namespace {

struct Cam {
    const float *V;  // device pointer to the 3x4 row-major view matrix (wave-uniform -> s_load)
    float fx, fy, cx, cy;
    float lim_x, lim_y;
    int tiles_x, tiles_y, block;
    float clip, glob_scale;
    int sem;         // SGN_SEM_* bits (include/sgn_rast.h: upstream-variant semantics; 0 = the decided defaults)
};

__device__ __forceinline__ void quat_to_R(float w, float x, float y, float z, float R[3][3]) {
    R[0][0] = 1.f - 2.f * (y * y + z * z);
    R[0][1] = 2.f * (x * y - w * z);
    R[0][2] = 2.f * (x * z + w * y);
    R[1][0] = 2.f * (x * y + w * z);
    R[1][1] = 1.f - 2.f * (x * x + z * z);
    R[1][2] = 2.f * (y * z - w * x);
    R[2][0] = 2.f * (x * z - w * y);
    R[2][1] = 2.f * (y * z + w * x);
    R[2][2] = 1.f - 2.f * (x * x + y * y);
}

struct QuatCheck {
    int32_t *flag;   // nullptr: no check
    float tol;
    int32_t stamp;   // value a failing row stores
    int32_t *ok;     // nullptr, or: the kernel's first lane stores the stamp here (the launch's stores are visible)
};

// Optional fused front end (SURVEY.md §8 a8 / north star): the scene graph's per-object rigid transform
// (sgn_splatfacto_scene_graph.py:404-417: means_w = R m + t, q_w = q_o2w (x) q), the quaternion
// normalisation (sgn_splatfacto.py:864) and exp(log-scale) (:857) evaluated in registers on load,
// instead of ~15 elementwise torch kernels and three N-sized concatenations per step.
struct Fuse {
    const int32_t *object_ids;  // [n] row of `poses` per Gaussian, or nullptr (no rigid transform)
    const float *poses;         // [n_objects,16]: R row-major (9), t (3), q_o2w wxyz (4)
};

struct Loaded {
    float p[3], q[4], s[3];
    float qw[4], inv_norm;  // un-normalised (possibly rotated) quaternion and 1/|qw| (fused only)
    float es[3];            // exp(log_scale) (fused only)
};

// MODE 0: activated inputs as upstream hands them.  MODE 1: the fused front end (log-scales, raw quaternions, optional
// pose).  MODE 2 (backward of the drop-in graph proofs, sgn_project_bwd_act): ACTIVATED scales as the caller computed
// them (exp of something: d/d log-scale = v_scale * scale, no second exp) and the UN-normalised quaternions X the
// caller divided by their norm (sgn_splatfacto.py:864); world-frame means, no pose.
template <int MODE>
__device__ __forceinline__ Loaded load_gaussian(int i, const float *__restrict__ means,
                                                const float *__restrict__ scales,
                                                const float *__restrict__ quats, float glob_scale, Fuse f) {
    Loaded L;
    const float m0 = means[3 * i], m1 = means[3 * i + 1], m2 = means[3 * i + 2];
    const float q0 = quats[4 * i], q1 = quats[4 * i + 1], q2 = quats[4 * i + 2], q3 = quats[4 * i + 3];
    if constexpr (MODE == 0) {
        L.p[0] = m0; L.p[1] = m1; L.p[2] = m2;
        L.q[0] = q0; L.q[1] = q1; L.q[2] = q2; L.q[3] = q3;
#pragma unroll
        for (int c = 0; c < 3; ++c) L.s[c] = glob_scale * scales[3 * i + c];
    } else {
        float rw = q0, rx = q1, ry = q2, rz = q3;
        L.p[0] = m0; L.p[1] = m1; L.p[2] = m2;
        if (f.object_ids != nullptr) {
            const float *P = f.poses + 16 * (size_t)f.object_ids[i];
            L.p[0] = P[0] * m0 + P[1] * m1 + P[2] * m2 + P[9];
            L.p[1] = P[3] * m0 + P[4] * m1 + P[5] * m2 + P[10];
            L.p[2] = P[6] * m0 + P[7] * m1 + P[8] * m2 + P[11];
            const float aw = P[12], ax = P[13], ay = P[14], az = P[15];
            rw = aw * q0 - ax * q1 - ay * q2 - az * q3;
            rx = aw * q1 + ax * q0 + ay * q3 - az * q2;
            ry = aw * q2 - ax * q3 + ay * q0 + az * q1;
            rz = aw * q3 + ax * q2 - ay * q1 + az * q0;
        }
        L.qw[0] = rw; L.qw[1] = rx; L.qw[2] = ry; L.qw[3] = rz;
        L.inv_norm = 1.f / sqrtf(rw * rw + rx * rx + ry * ry + rz * rz);
        L.q[0] = rw * L.inv_norm; L.q[1] = rx * L.inv_norm; L.q[2] = ry * L.inv_norm; L.q[3] = rz * L.inv_norm;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            L.es[c] = MODE == 2 ? scales[3 * i + c] : expf(scales[3 * i + c]);
            L.s[c] = glob_scale * L.es[c];
        }
    }
    return L;
}

template <int FUSED>
__global__ __launch_bounds__(256) void project_fwd_kernel(
    int n, const float *__restrict__ means, const float *__restrict__ scales,
    const float *__restrict__ quats, Cam cam, Fuse fuse, float *__restrict__ cov3d, float *__restrict__ xys,
    float *__restrict__ depths, int32_t *__restrict__ radii, float *__restrict__ conics,
    float *__restrict__ comp, int32_t *__restrict__ num_tiles_hit, QuatCheck qc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float V[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) V[k] = cam.V[k];
    const Loaded LG = load_gaussian<FUSED>(i, means, scales, quats, cam.glob_scale, fuse);
    if constexpr (FUSED == 0) {
        // upstream's `quats must be normalized` assertion riding the projection (sgn_project_fwd_all): the quaternion is
        // in registers anyway.  Same one-sided test as check_unit_quats_kernel below; a failing row STAMPS the flag
        // (every writer stores the same value: benign race), so a flag that was zero once needs no clear per call.
        if (qc.flag != nullptr) {
            const float nrm = sqrtf(LG.q[0] * LG.q[0] + LG.q[1] * LG.q[1] + LG.q[2] * LG.q[2] + LG.q[3] * LG.q[3]);
            // system scope: the flag may live in mapped pinned HOST memory (sgn_project_fwd_all), where the host reads it
            // behind an event without a copy command (a plain store is not guaranteed to have landed)
            if (!(nrm - 1.f < qc.tol)) {
                __hip_atomic_store(qc.flag, qc.stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __threadfence_system();
            }
            if (qc.ok != nullptr && i == 0) {
                __hip_atomic_store(qc.ok, qc.stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __threadfence_system();
            }
        }
    }
    const float p0 = LG.p[0], p1 = LG.p[1], p2 = LG.p[2];
    const float pvx = V[0] * p0 + V[1] * p1 + V[2] * p2 + V[3];
    const float pvy = V[4] * p0 + V[5] * p1 + V[6] * p2 + V[7];
    const float pvz = V[8] * p0 + V[9] * p1 + V[10] * p2 + V[11];

    float o_cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float o_con[3] = {0.f, 0.f, 0.f};
    float o_x = 0.f, o_y = 0.f, o_d = 0.f, o_comp = 0.f;
    int o_r = 0, o_n = 0;

    if (pvz > cam.clip) {
        float R[3][3], M[3][3], S[3][3];
        quat_to_R(LG.q[0], LG.q[1], LG.q[2], LG.q[3], R);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float s = LG.s[c];
#pragma unroll
            for (int r = 0; r < 3; ++r) M[r][c] = R[r][c] * s;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                S[a][b] = M[a][0] * M[b][0] + M[a][1] * M[b][1] + M[a][2] * M[b][2];
        o_cov[0] = S[0][0]; o_cov[1] = S[0][1]; o_cov[2] = S[0][2];
        o_cov[3] = S[1][1]; o_cov[4] = S[1][2]; o_cov[5] = S[2][2];

        const float tz = pvz;
        const float tx = tz * fminf(cam.lim_x, fmaxf(-cam.lim_x, pvx / tz));
        const float ty = tz * fminf(cam.lim_y, fmaxf(-cam.lim_y, pvy / tz));
        const float rz = 1.f / tz, rz2 = rz * rz;
        const float J00 = cam.fx * rz, J02 = -cam.fx * tx * rz2;
        const float J11 = cam.fy * rz, J12 = -cam.fy * ty * rz2;
        float T[2][3], U[2][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            T[0][j] = J00 * V[j] + J02 * V[8 + j];
            T[1][j] = J11 * V[4 + j] + J12 * V[8 + j];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                U[a][j] = T[a][0] * S[0][j] + T[a][1] * S[1][j] + T[a][2] * S[2][j];
        const float c00 = U[0][0] * T[0][0] + U[0][1] * T[0][1] + U[0][2] * T[0][2];
        const float c01 = U[0][0] * T[1][0] + U[0][1] * T[1][1] + U[0][2] * T[1][2];
        const float c11 = U[1][0] * T[1][0] + U[1][1] * T[1][1] + U[1][2] * T[1][2];
        const float det0 = c00 * c11 - c01 * c01;
        const float a = c00 + 0.3f, b = c01, c = c11 + 0.3f;
        const float det = a * c - b * b;
        const float compensation = sqrtf(fmaxf(0.f, det0 / det));
        if (det != 0.f) {
            const float inv_det = 1.f / det;
            o_con[0] = c * inv_det; o_con[1] = -b * inv_det; o_con[2] = a * inv_det;
            const float mid = 0.5f * (a + c);
            const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
            const float v1 = mid + sq, v2 = mid - sq;
            const float radius = ceilf(3.f * sqrtf(fmaxf(v1, v2)));
            const float rw = 1.f / (pvz + 1e-6f);
            const float ux = pvx * rw * cam.fx + cam.cx, uy = pvy * rw * cam.fy + cam.cy;
            int mnx, mny, mxx, mxy;
            sgn_tile_bbox(ux, uy, radius, cam.tiles_x, cam.tiles_y, cam.block, mnx, mny, mxx, mxy, cam.sem);
            const int area = (mxx - mnx) * (mxy - mny);
            if (area > 0) {
                o_n = area; o_d = pvz; o_r = sgn_f2i(radius); o_x = ux; o_y = uy;
                o_comp = compensation;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) cov3d[6 * i + k] = o_cov[k];
    xys[2 * i] = o_x; xys[2 * i + 1] = o_y;
    depths[i] = o_d;
    radii[i] = o_r;
#pragma unroll
    for (int k = 0; k < 3; ++k) conics[3 * i + k] = o_con[k];
    comp[i] = o_comp;
    num_tiles_hit[i] = o_n;
}

template <int FUSED>
__global__ __launch_bounds__(256) void project_bwd_kernel(
    int n, const float *__restrict__ means, const float *__restrict__ scales,
    const float *__restrict__ quats, Cam cam, Fuse fuse, const float *__restrict__ cov3d,
    const int32_t *__restrict__ radii, const float *__restrict__ conics,
    const float *__restrict__ comp, const float *__restrict__ v_xy,
    const float *__restrict__ v_depth, const float *__restrict__ v_conic,
    const float *__restrict__ v_comp, float *__restrict__ v_cov2d, float *__restrict__ v_cov3d,
    float *__restrict__ v_mean, float *__restrict__ v_scale, float *__restrict__ v_quat) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float o_vm[3] = {0.f, 0.f, 0.f}, o_vs[3] = {0.f, 0.f, 0.f}, o_vq[4] = {0.f, 0.f, 0.f, 0.f};
    float o_vc2[3] = {0.f, 0.f, 0.f}, o_vc3[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (radii[i] > 0) {
        float V[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) V[k] = cam.V[k];
        const float fx = cam.fx, fy = cam.fy;
        const Loaded LG = load_gaussian<FUSED>(i, means, scales, quats, cam.glob_scale, fuse);
        const float p0 = LG.p[0], p1 = LG.p[1], p2 = LG.p[2];
        const float pvx = V[0] * p0 + V[1] * p1 + V[2] * p2 + V[3];
        const float pvy = V[4] * p0 + V[5] * p1 + V[6] * p2 + V[7];
        const float pvz = V[8] * p0 + V[9] * p1 + V[10] * p2 + V[11];
        const float rw = 1.f / (pvz + 1e-6f);
        const float vpx = fx * v_xy[2 * i], vpy = fy * v_xy[2 * i + 1];
        const float vv0 = vpx * rw, vv1 = vpy * rw, vv2 = -(vpx * pvx + vpy * pvy) * rw * rw;
        float vm[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) vm[j] = V[j] * vv0 + V[4 + j] * vv1 + V[8 + j] * vv2;
        const float vz = v_depth ? v_depth[i] : 0.f;      // NULL: depths took no part in the loss
        vm[0] += V[8] * vz; vm[1] += V[9] * vz; vm[2] += V[10] * vz;

        const float X00 = conics[3 * i], X01 = conics[3 * i + 1], X11 = conics[3 * i + 2];
        // v_conic[:,1] is the true dL/d(conic.y); the symmetric matrix gradient carries half of it per slot (exact)
        const float g0 = v_conic[3 * i], g1 = 0.5f * v_conic[3 * i + 1], g2 = v_conic[3 * i + 2];
        const float A00 = X00 * g0 + X01 * g1, A01 = X00 * g1 + X01 * g2;
        const float A10 = X01 * g0 + X11 * g1, A11 = X01 * g1 + X11 * g2;
        const float S00 = -(A00 * X00 + A01 * X01), S01 = -(A00 * X01 + A01 * X11);
        const float S10 = -(A10 * X00 + A11 * X01), S11 = -(A10 * X01 + A11 * X11);
        float vc2[3] = {S00, S01 + S10, S11};
        if (v_comp != nullptr) {
            const float cmp = comp[i];
            const float inv_det = X00 * X11 - X01 * X01;
            const float om = 1.f - cmp * cmp;
            const float vsq = v_comp[i] * 0.5f / (cmp + 1e-6f);
            vc2[0] += vsq * (om * X00 - 0.3f * inv_det);
            vc2[1] += 2.f * vsq * (om * X01);
            vc2[2] += vsq * (om * X11 - 0.3f * inv_det);
        }
        o_vc2[0] = vc2[0]; o_vc2[1] = vc2[1]; o_vc2[2] = vc2[2];

        const float rz = 1.f / pvz, rz2 = rz * rz, rz3 = rz2 * rz;
        // default: upstream CUDA's project_cov3d_ewa_vjp — the Jacobian of the UN-clamped view-space point.  Variant
        // (SGN_SEM_EWA_VJP_CLAMPED): differentiate through the forward's clamp tx = tz * clamp(pvx / tz, +-lim): where
        // the clamp is active d tx / d pvx = 0 and d tx / d tz = tx / tz (what autograd through _torch_impl gives)
        float ex = pvx, ey = pvy;
        bool clx = false, cly = false;
        if (cam.sem & SGN_SEM_EWA_VJP_CLAMPED) {
            const float qx = pvx / pvz, qy = pvy / pvz;
            clx = !(qx <= cam.lim_x && qx >= -cam.lim_x);
            cly = !(qy <= cam.lim_y && qy >= -cam.lim_y);
            ex = pvz * fminf(cam.lim_x, fmaxf(-cam.lim_x, qx));
            ey = pvz * fminf(cam.lim_y, fmaxf(-cam.lim_y, qy));
        }
        const float J00 = fx * rz, J02 = -fx * ex * rz2, J11 = fy * rz, J12 = -fy * ey * rz2;
        float T[2][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            T[0][j] = J00 * V[j] + J02 * V[8 + j];
            T[1][j] = J11 * V[4 + j] + J12 * V[8 + j];
        }
        const float c0 = cov3d[6 * i], c1 = cov3d[6 * i + 1], c2 = cov3d[6 * i + 2];
        const float c3 = cov3d[6 * i + 3], c4 = cov3d[6 * i + 4], c5 = cov3d[6 * i + 5];
        const float S[3][3] = {{c0, c1, c2}, {c1, c3, c4}, {c2, c4, c5}};
        const float G[2][2] = {{vc2[0], 0.5f * vc2[1]}, {0.5f * vc2[1], vc2[2]}};
        float GT[2][3];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < 3; ++j) GT[a][j] = G[a][0] * T[0][j] + G[a][1] * T[1][j];
        float vS[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) vS[a][b] = T[0][a] * GT[0][b] + T[1][a] * GT[1][b];
        o_vc3[0] = vS[0][0]; o_vc3[1] = vS[0][1] + vS[1][0]; o_vc3[2] = vS[0][2] + vS[2][0];
        o_vc3[3] = vS[1][1]; o_vc3[4] = vS[1][2] + vS[2][1]; o_vc3[5] = vS[2][2];
        float vT[2][3];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                vT[a][j] = 2.f * (GT[a][0] * S[0][j] + GT[a][1] * S[1][j] + GT[a][2] * S[2][j]);
        const float vJ00 = vT[0][0] * V[0] + vT[0][1] * V[1] + vT[0][2] * V[2];
        const float vJ02 = vT[0][0] * V[8] + vT[0][1] * V[9] + vT[0][2] * V[10];
        const float vJ11 = vT[1][0] * V[4] + vT[1][1] * V[5] + vT[1][2] * V[6];
        const float vJ12 = vT[1][0] * V[8] + vT[1][1] * V[9] + vT[1][2] * V[10];
        const float vt0 = clx ? 0.f : -fx * rz2 * vJ02, vt1 = cly ? 0.f : -fy * rz2 * vJ12;
        // d J02 / d tz = 2 fx tx / tz^3 with tx free, fx tx / tz^3 with tx = tz * lim (clamped)
        const float vt2 = -fx * rz2 * vJ00 + (clx ? 1.f : 2.f) * fx * ex * rz3 * vJ02 - fy * rz2 * vJ11 +
                          (cly ? 1.f : 2.f) * fy * ey * rz3 * vJ12;
#pragma unroll
        for (int j = 0; j < 3; ++j) vm[j] += V[j] * vt0 + V[4 + j] * vt1 + V[8 + j] * vt2;
        o_vm[0] = vm[0]; o_vm[1] = vm[1]; o_vm[2] = vm[2];

        const float vV[3][3] = {{o_vc3[0], 0.5f * o_vc3[1], 0.5f * o_vc3[2]},
                                {0.5f * o_vc3[1], o_vc3[3], 0.5f * o_vc3[4]},
                                {0.5f * o_vc3[2], 0.5f * o_vc3[4], o_vc3[5]}};
        const float w = LG.q[0], x = LG.q[1], y = LG.q[2], z = LG.q[3];
        float R[3][3], M[3][3], sc[3];
        quat_to_R(w, x, y, z, R);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            sc[c] = LG.s[c];
#pragma unroll
            for (int r = 0; r < 3; ++r) M[r][c] = R[r][c] * sc[c];
        }
        float vM[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                vM[a][c] = 2.f * (vV[a][0] * M[0][c] + vV[a][1] * M[1][c] + vV[a][2] * M[2][c]);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            o_vs[c] = (R[0][c] * vM[0][c] + R[1][c] * vM[1][c] + R[2][c] * vM[2][c]) * cam.glob_scale;
        float vR[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) vR[a][c] = vM[a][c] * sc[c];
        o_vq[0] = 2.f * (x * (vR[2][1] - vR[1][2]) + y * (vR[0][2] - vR[2][0]) + z * (vR[1][0] - vR[0][1]));
        o_vq[1] = 2.f * (-2.f * x * (vR[1][1] + vR[2][2]) + y * (vR[1][0] + vR[0][1]) +
                         z * (vR[2][0] + vR[0][2]) + w * (vR[2][1] - vR[1][2]));
        o_vq[2] = 2.f * (x * (vR[1][0] + vR[0][1]) - 2.f * y * (vR[0][0] + vR[2][2]) +
                         z * (vR[2][1] + vR[1][2]) + w * (vR[0][2] - vR[2][0]));
        o_vq[3] = 2.f * (x * (vR[2][0] + vR[0][2]) + y * (vR[2][1] + vR[1][2]) -
                         2.f * z * (vR[0][0] + vR[1][1]) + w * (vR[1][0] - vR[0][1]));
        if constexpr (FUSED != 0) {
            // chain rules of the fused front end: exp, normalisation, Hamilton product, rigid transform
#pragma unroll
            for (int c = 0; c < 3; ++c) o_vs[c] = o_vs[c] * LG.es[c];            // d/d log_scale
            const float dq = LG.q[0] * o_vq[0] + LG.q[1] * o_vq[1] + LG.q[2] * o_vq[2] + LG.q[3] * o_vq[3];
            float g[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = (o_vq[k] - LG.q[k] * dq) * LG.inv_norm;   // d/d (un-normalised q_w)
            if (fuse.object_ids != nullptr) {
                const float *P = fuse.poses + 16 * (size_t)fuse.object_ids[i];
                const float aw = P[12], ax = P[13], ay = P[14], az = P[15];
                o_vq[0] = aw * g[0] + ax * g[1] + ay * g[2] + az * g[3];
                o_vq[1] = -ax * g[0] + aw * g[1] + az * g[2] - ay * g[3];
                o_vq[2] = -ay * g[0] - az * g[1] + aw * g[2] + ax * g[3];
                o_vq[3] = -az * g[0] + ay * g[1] - ax * g[2] + aw * g[3];
                const float v0 = o_vm[0], v1 = o_vm[1], v2 = o_vm[2];           // v_means_local = R^T v_means_w
                o_vm[0] = P[0] * v0 + P[3] * v1 + P[6] * v2;
                o_vm[1] = P[1] * v0 + P[4] * v1 + P[7] * v2;
                o_vm[2] = P[2] * v0 + P[5] * v1 + P[8] * v2;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) o_vq[k] = g[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { v_mean[3 * i + k] = o_vm[k]; v_scale[3 * i + k] = o_vs[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) v_quat[4 * i + k] = o_vq[k];
    if (v_cov2d != nullptr)
#pragma unroll
        for (int k = 0; k < 3; ++k) v_cov2d[3 * i + k] = o_vc2[k];
    if (v_cov3d != nullptr)
#pragma unroll
        for (int k = 0; k < 6; ++k) v_cov3d[6 * i + k] = o_vc3[k];
}

Cam make_cam(const float *V, float fx, float fy, float cx, float cy, int h, int w, int block,
             float clip, float glob_scale, int semantics = 0) {
    Cam c;
    c.V = V;
    c.fx = fx; c.fy = fy; c.cx = cx; c.cy = cy;
    const float tan_fovx = 0.5f * (float)w / fx, tan_fovy = 0.5f * (float)h / fy;
    c.lim_x = 1.3f * tan_fovx; c.lim_y = 1.3f * tan_fovy;
    c.block = block;
    c.tiles_x = (w + block - 1) / block; c.tiles_y = (h + block - 1) / block;
    c.clip = clip; c.glob_scale = glob_scale;
    c.sem = semantics;
    return c;
}

// the backward's camera: the image size only matters for the clamped EWA vjp (its +-1.3 tan(fov/2) limits)
Cam bwd_cam(const float *V, float fx, float fy, float glob_scale, int semantics, int img_h, int img_w) {
    const bool cl = (semantics & SGN_SEM_EWA_VJP_CLAMPED) != 0;
    return make_cam(V, fx, fy, 0.f, 0.f, cl ? img_h : 16, cl ? img_w : 16, 16, 0.f, glob_scale, semantics);
}

}  // namespace

// viewmat12 is a DEVICE pointer (the reference hands a device tensor, viewmat.squeeze()[:3,:],
// sgn_splatfacto.py:865): the kernels read it through wave-uniform scalar loads, so no host sync.
int sgn_project_fwd_checked(int n, const float *means3d, const float *scales, float glob_scale,
                            const float *quats, const float *viewmat12, float fx, float fy,
                            float cx, float cy, int img_h, int img_w, int block_width,
                            float clip_thresh, float *cov3d, float *xys, float *depths,
                            int32_t *radii, float *conics, float *compensation,
                            int32_t *num_tiles_hit, int32_t *quat_flag, float quat_tol, int32_t quat_stamp,
                            int32_t *quat_ok, int semantics, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0, -1);
    SGN_ARG_CHECK(block_width >= 2 && block_width <= 16, -2);
    SGN_ARG_CHECK(img_h > 0 && img_w > 0, -3);
    if (n == 0) return 0;
    SGN_ARG_CHECK(means3d && scales && quats && viewmat12 && cov3d && xys && depths && radii &&
                      conics && compensation && num_tiles_hit, -4);
    const Cam cam = make_cam(viewmat12, fx, fy, cx, cy, img_h, img_w, block_width, clip_thresh, glob_scale, semantics);
    sgn_timing_begin(SGN_T_PROJECT_FWD, stream);
    hipLaunchKernelGGL(project_fwd_kernel<0>, dim3(sgn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n,
                       means3d, scales, quats, cam, Fuse{nullptr, nullptr}, cov3d, xys, depths, radii, conics,
                       compensation, num_tiles_hit, QuatCheck{quat_flag, quat_tol, quat_stamp, quat_ok});
    sgn_timing_end(SGN_T_PROJECT_FWD, stream);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_project_fwd(int n, const float *means3d, const float *scales, float glob_scale,
                               const float *quats, const float *viewmat12, float fx, float fy,
                               float cx, float cy, int img_h, int img_w, int block_width,
                               float clip_thresh, float *cov3d, float *xys, float *depths,
                               int32_t *radii, float *conics, float *compensation,
                               int32_t *num_tiles_hit, int semantics, sgn_stream_t stream) {
    return sgn_project_fwd_checked(n, means3d, scales, glob_scale, quats, viewmat12, fx, fy, cx, cy, img_h, img_w,
                                   block_width, clip_thresh, cov3d, xys, depths, radii, conics, compensation,
                                   num_tiles_hit, nullptr, 0.f, 0, nullptr, semantics, stream);
}

SGN_EXPORT int sgn_project_bwd(int n, const float *means3d, const float *scales, float glob_scale,
                               const float *quats, const float *viewmat12, float fx, float fy,
                               const float *cov3d, const int32_t *radii, const float *conics,
                               const float *compensation, const float *v_xy, const float *v_depth,
                               const float *v_conic, const float *v_compensation, float *v_cov2d,
                               float *v_cov3d, float *v_mean3d, float *v_scale, float *v_quat,
                               int semantics, int img_h, int img_w, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0, -1);
    if (n == 0) return 0;
    SGN_ARG_CHECK(means3d && scales && quats && viewmat12 && cov3d && radii && conics && v_xy &&
                      v_conic && v_mean3d && v_scale && v_quat, -4);
    SGN_ARG_CHECK(v_compensation == nullptr || compensation != nullptr, -5);
    SGN_ARG_CHECK(!(semantics & SGN_SEM_EWA_VJP_CLAMPED) || (img_h > 0 && img_w > 0), -7);
    const Cam cam = bwd_cam(viewmat12, fx, fy, glob_scale, semantics, img_h, img_w);
    sgn_timing_begin(SGN_T_PROJECT_BWD, stream);
    hipLaunchKernelGGL(project_bwd_kernel<0>, dim3(sgn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n,
                       means3d, scales, quats, cam, Fuse{nullptr, nullptr}, cov3d, radii, conics, compensation, v_xy,
                       v_depth, v_conic, v_compensation, v_cov2d, v_cov3d, v_mean3d, v_scale, v_quat);
    sgn_timing_end(SGN_T_PROJECT_BWD, stream);
    SGN_LAUNCH_CHECK();
    return 0;
}

// ---- fused front end (extension beyond gsplat's surface) --------------------------
SGN_EXPORT int sgn_project_fwd_fused(int n, const float *means_local, const float *log_scales, float glob_scale,
                                     const float *quats_raw, const int32_t *object_ids, const float *poses,
                                     const float *viewmat12, float fx, float fy, float cx, float cy, int img_h,
                                     int img_w, int block_width, float clip_thresh, float *cov3d, float *xys,
                                     float *depths, int32_t *radii, float *conics, float *compensation,
                                     int32_t *num_tiles_hit, int semantics, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0, -1);
    SGN_ARG_CHECK(block_width >= 2 && block_width <= 16, -2);
    SGN_ARG_CHECK(img_h > 0 && img_w > 0, -3);
    if (n == 0) return 0;
    SGN_ARG_CHECK(means_local && log_scales && quats_raw && viewmat12 && cov3d && xys && depths && radii && conics &&
                      compensation && num_tiles_hit, -4);
    SGN_ARG_CHECK((object_ids == nullptr) == (poses == nullptr), -5);
    const Cam cam = make_cam(viewmat12, fx, fy, cx, cy, img_h, img_w, block_width, clip_thresh, glob_scale, semantics);
    sgn_timing_begin(SGN_T_PROJECT_FWD, stream);
    hipLaunchKernelGGL(project_fwd_kernel<1>, dim3(sgn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n,
                       means_local, log_scales, quats_raw, cam, Fuse{object_ids, poses}, cov3d, xys, depths, radii,
                       conics, compensation, num_tiles_hit, QuatCheck{nullptr, 0.f, 0, nullptr});
    sgn_timing_end(SGN_T_PROJECT_FWD, stream);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_project_bwd_fused(int n, const float *means_local, const float *log_scales, float glob_scale,
                                     const float *quats_raw, const int32_t *object_ids, const float *poses,
                                     const float *viewmat12, float fx, float fy, const float *cov3d,
                                     const int32_t *radii, const float *conics, const float *compensation,
                                     const float *v_xy, const float *v_depth, const float *v_conic,
                                     const float *v_compensation, float *v_means_local, float *v_log_scales,
                                     float *v_quats_raw, int semantics, int img_h, int img_w, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0, -1);
    if (n == 0) return 0;
    SGN_ARG_CHECK(means_local && log_scales && quats_raw && viewmat12 && cov3d && radii && conics && v_xy &&
                      v_conic && v_means_local && v_log_scales && v_quats_raw, -4);
    SGN_ARG_CHECK(v_compensation == nullptr || compensation != nullptr, -5);
    SGN_ARG_CHECK((object_ids == nullptr) == (poses == nullptr), -6);
    SGN_ARG_CHECK(!(semantics & SGN_SEM_EWA_VJP_CLAMPED) || (img_h > 0 && img_w > 0), -7);
    const Cam cam = bwd_cam(viewmat12, fx, fy, glob_scale, semantics, img_h, img_w);
    sgn_timing_begin(SGN_T_PROJECT_BWD, stream);
    hipLaunchKernelGGL(project_bwd_kernel<1>, dim3(sgn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n,
                       means_local, log_scales, quats_raw, cam, Fuse{object_ids, poses}, cov3d, radii, conics,
                       compensation, v_xy, v_depth, v_conic, v_compensation, nullptr, nullptr, v_means_local,
                       v_log_scales, v_quats_raw);
    sgn_timing_end(SGN_T_PROJECT_BWD, stream);
    SGN_LAUNCH_CHECK();
    return 0;
}

// Backward of `project_gaussians(means, scales, g, X / |X|)` (the drop-in call, sgn_splatfacto.py:857-873) taken
// one step further back than sgn_project_bwd: gradients w.r.t. the means, the LOGARITHM of the scales (v_scale * scale)
// and the UN-normalised quaternions X.  What the graph proofs of the drop-in operators call when the autograd graph
// shows `scales = exp(...)` and `quats = X / X.norm(dim=-1, keepdim=True)` (sgn_rast/proofs.py).
SGN_EXPORT int sgn_project_bwd_act(int n, const float *means3d, const float *scales_activated, float glob_scale,
                                   const float *quats_unnormalised, const float *viewmat12, float fx, float fy,
                                   const float *cov3d, const int32_t *radii, const float *conics,
                                   const float *compensation, const float *v_xy, const float *v_depth,
                                   const float *v_conic, const float *v_compensation, float *v_mean3d,
                                   float *v_log_scales, float *v_quats_unnormalised, int semantics, int img_h,
                                   int img_w, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0, -1);
    if (n == 0) return 0;
    SGN_ARG_CHECK(means3d && scales_activated && quats_unnormalised && viewmat12 && cov3d && radii && conics && v_xy &&
                      v_conic && v_mean3d && v_log_scales && v_quats_unnormalised, -4);
    SGN_ARG_CHECK(v_compensation == nullptr || compensation != nullptr, -5);
    SGN_ARG_CHECK(!(semantics & SGN_SEM_EWA_VJP_CLAMPED) || (img_h > 0 && img_w > 0), -7);
    const Cam cam = bwd_cam(viewmat12, fx, fy, glob_scale, semantics, img_h, img_w);
    sgn_timing_begin(SGN_T_PROJECT_BWD, stream);
    hipLaunchKernelGGL(project_bwd_kernel<2>, dim3(sgn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n,
                       means3d, scales_activated, quats_unnormalised, cam, Fuse{nullptr, nullptr}, cov3d, radii, conics,
                       compensation, v_xy, v_depth, v_conic, v_compensation, nullptr, nullptr, v_mean3d,
                       v_log_scales, v_quats_unnormalised);
    sgn_timing_end(SGN_T_PROJECT_BWD, stream);
    SGN_LAUNCH_CHECK();
    return 0;
}

// gsplat's `assert (quats.norm(dim=-1) - 1 < 1e-6).all(), "quats must be normalized"` (project_gaussians.py) as ONE
// pass that raises a device flag, so the host can look at it at its next existing sync point instead of stalling
// the queue for the assertion alone (4 torch kernels + a host sync per call otherwise).
namespace {
__global__ __launch_bounds__(256) void check_unit_quats_kernel(int n, const float *__restrict__ quats, float tol,
                                                               int32_t *__restrict__ flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 q = reinterpret_cast<const float4 *>(quats)[i];
    const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    if (!(nrm - 1.f < tol)) *flag = 1;      // same one-sided test as upstream; NaN fails it too.  Benign race.
}
}  // namespace

// Round 6 — "this point of the stream has been reached, and here are a few words": ONE wave copies up to 64 device words
// into mapped pinned host memory and then stores `flag_value` into a flag word there (system scope, after a fence: a host
// that sees the flag sees the words, and — stream order — everything queued before this launch is complete).  The host
// POLLS the flag (api.cpp spin_wait): no copy command, no event record (a ~6 us barrier bubble on the stream each).
namespace {
__global__ __launch_bounds__(64) void publish_words_kernel(const int32_t *__restrict__ src, int n,
                                                           int32_t *__restrict__ dst_mapped,
                                                           int32_t *__restrict__ flag_mapped, int32_t flag_value) {
    const int i = threadIdx.x;
    if (i < n) __hip_atomic_store(dst_mapped + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (i == 0) {
        __hip_atomic_store(flag_mapped, flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
    }
}
}  // namespace

int sgn_publish_words(const int32_t *src_dev, int n, int32_t *dst_mapped, int32_t *flag_mapped, int32_t flag_value,
                      sgn_stream_t stream) {
    SGN_ARG_CHECK(flag_mapped != nullptr && n >= 0 && n <= 64 && (n == 0 || (src_dev && dst_mapped)), -1);
    hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, src_dev, n, dst_mapped,
                       flag_mapped, flag_value);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_check_unit_quats(int n, const float *quats, float tol, int32_t *flag, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0 && flag != nullptr, -1);
    hipStream_t s = (hipStream_t)stream;
    SGN_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int32_t), s));
    if (n == 0) return 0;
    SGN_ARG_CHECK(quats != nullptr && (reinterpret_cast<uintptr_t>(quats) & 15) == 0, -2);
    hipLaunchKernelGGL(check_unit_quats_kernel, dim3(sgn_cdiv(n, 256)), dim3(256), 0, s, n, quats, tol, flag);
    SGN_LAUNCH_CHECK();
    return 0;
}
