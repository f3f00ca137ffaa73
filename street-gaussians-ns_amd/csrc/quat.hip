// quat.hip — fused Hamilton product for the scene graph's object->world rotation (gfx950).
//
// Replaces pytorch3d.transforms.quaternion_multiply as the reference uses it in object2world_gs
// (street_gaussians_ns/sgn_splatfacto_scene_graph.py:416: `quat_w = quaternion_multiply(quat_o2w, quats)`, one
// constant quaternion against the [N,4] quaternions of an object model).  pytorch3d's version is 16 multiplies, 12
// adds, a stack and a `where` as separate torch kernels — ~30 launches forward and ~60 backward per visible object,
// which left the drop-in scene-graph step launch-bound (profiles/r02c_sg_dropin_gaps.md: GPU busy 43 %).
// One pass each way here; same arithmetic order as pytorch3d's quaternion_raw_multiply, real part first, result
// standardised to a non-negative real part.
#include "sgn_common.h"

namespace {

struct Quat { float w, x, y, z; };

// a (x) b with pytorch3d's operation order (left-to-right sums of four products, no FMA: -ffp-contract=off)
__device__ __forceinline__ Quat qmul(const Quat a, const Quat b) {
    Quat o;
    o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    o.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    o.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return o;
}

// a_stride = 0: one quaternion for all rows (host passes it by value); 4: one per row
__global__ __launch_bounds__(256) void quat_mul_fwd_kernel(int n, Quat a0, const float4 *__restrict__ a_rows,
                                                           const float4 *__restrict__ b, float4 *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Quat a = a0;
    if (a_rows) { const float4 t = a_rows[i]; a = {t.x, t.y, t.z, t.w}; }
    const float4 bb = b[i];
    Quat o = qmul(a, {bb.x, bb.y, bb.z, bb.w});
    if (o.w < 0.f) { o.w = -o.w; o.x = -o.x; o.y = -o.y; o.z = -o.z; }   // standardize_quaternion
    out[i] = make_float4(o.w, o.x, o.y, o.z);
}

// o = s * L(a) b  =>  v_b = s * L(a)^T v_o = s * conj(a) (x) v_o ;  v_a = s * v_o (x) conj(b)  (when requested)
__global__ __launch_bounds__(256) void quat_mul_bwd_kernel(int n, Quat a0, const float4 *__restrict__ a_rows,
                                                           const float4 *__restrict__ b,
                                                           const float4 *__restrict__ v_out,
                                                           float4 *__restrict__ v_b, float4 *__restrict__ v_a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Quat a = a0;
    if (a_rows) { const float4 t = a_rows[i]; a = {t.x, t.y, t.z, t.w}; }
    const float4 bb = b[i], vv = v_out[i];
    const Quat bq = {bb.x, bb.y, bb.z, bb.w};
    const float s = qmul(a, bq).w < 0.f ? -1.f : 1.f;
    const Quat v = {s * vv.x, s * vv.y, s * vv.z, s * vv.w};
    if (v_b) {
        const Quat g = qmul({a.w, -a.x, -a.y, -a.z}, v);
        v_b[i] = make_float4(g.w, g.x, g.y, g.z);
    }
    if (v_a) {
        const Quat g = qmul(v, {bq.w, -bq.x, -bq.y, -bq.z});
        v_a[i] = make_float4(g.w, g.x, g.y, g.z);
    }
}

}  // namespace

SGN_EXPORT int sgn_quat_mul_fwd(int n, const float *a_host4, const float *a_rows, const float *b, float *out,
                                sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0, -1);
    if (n == 0) return 0;
    SGN_ARG_CHECK((a_host4 != nullptr) != (a_rows != nullptr) && b && out, -2);
    Quat a0 = {1.f, 0.f, 0.f, 0.f};
    if (a_host4) a0 = {a_host4[0], a_host4[1], a_host4[2], a_host4[3]};
    hipLaunchKernelGGL(quat_mul_fwd_kernel, dim3(sgn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, a0,
                       (const float4 *)a_rows, (const float4 *)b, (float4 *)out);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_quat_mul_bwd(int n, const float *a_host4, const float *a_rows, const float *b, const float *v_out,
                                float *v_b, float *v_a_rows, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0, -1);
    if (n == 0) return 0;
    SGN_ARG_CHECK((a_host4 != nullptr) != (a_rows != nullptr) && b && v_out && (v_b || v_a_rows), -2);
    SGN_ARG_CHECK(v_a_rows == nullptr || a_rows != nullptr, -3);   // per-row a gradient only for per-row a
    Quat a0 = {1.f, 0.f, 0.f, 0.f};
    if (a_host4) a0 = {a_host4[0], a_host4[1], a_host4[2], a_host4[3]};
    hipLaunchKernelGGL(quat_mul_bwd_kernel, dim3(sgn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, a0,
                       (const float4 *)a_rows, (const float4 *)b, (const float4 *)v_out, (float4 *)v_b,
                       (float4 *)v_a_rows);
    SGN_LAUNCH_CHECK();
    return 0;
}
