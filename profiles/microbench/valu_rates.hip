// valu_rates.hip — wave64 VALU issue rates on gfx950, to interpret SQ_ACTIVE_INST_VALU of the raster kernels.
//   hipcc --offload-arch=gfx950 -O3 profiles/microbench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
// Each test: every wave runs ITERS x 64 independent instructions of one kind (8 accumulators x 8);
// grid = 1024 SIMDs x W waves of 64 threads.  Reported: wave-instructions per ns per SIMD and the implied
// cycles per wave-instruction at the measured wall time (clock read from hipDeviceProp, nominal).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>

#define REP8(X) X X X X X X X X
#define BODY(INSTR)                                                                                         \
    for (int i = 0; i < iters; ++i) {                                                                       \
        REP8(asm volatile(INSTR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                          : "v"(b), "v"(c));)                                                                \
    }

#define KERNEL(NAME, INSTR)                                                                \
    __global__ __launch_bounds__(64) void NAME(float *out, int iters) {                     \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
              a7 = a0 + 7;                                                                  \
        float b = 1.0001f, c = 0.5f;                                                        \
        BODY(INSTR)                                                                         \
        out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;         \
    }

#define I8(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" \
               op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define I8_3(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" \
                 op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n"
#define I8_1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" \
                 op " %6, %6\n" op " %7, %7\n"

KERNEL(k_mul, I8("v_mul_f32"))
KERNEL(k_add, I8("v_add_f32"))
KERNEL(k_sub, I8("v_sub_f32"))
KERNEL(k_min, I8("v_min_f32"))
KERNEL(k_fma, I8_3("v_fma_f32"))
KERNEL(k_fmac, I8("v_fmac_f32"))
KERNEL(k_exp, I8_1("v_exp_f32"))
KERNEL(k_rcp, I8_1("v_rcp_f32"))
KERNEL(k_mov, I8_1("v_mov_b32"))
KERNEL(k_cndmask, I8("v_cndmask_b32"))
KERNEL(k_dpp, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
              "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
              "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
              "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
              "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
              "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
              "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
              "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_cmp, "v_cmp_le_f32 vcc, %0, %8\nv_cmp_le_f32 vcc, %1, %8\nv_cmp_le_f32 vcc, %2, %8\nv_cmp_le_f32 vcc, %3, %8\n"
              "v_cmp_le_f32 vcc, %4, %8\nv_cmp_le_f32 vcc, %5, %8\nv_cmp_le_f32 vcc, %6, %8\nv_cmp_le_f32 vcc, %7, %8\n")

// packed: two floats per lane per instruction
__global__ __launch_bounds__(64) void k_pkfma(float *out, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,
       a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\n"
                          "v_pk_fma_f32 %3, %3, %8, %9\nv_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\n"
                          "v_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                          : "v"(b), "v"(c));)
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 64 + threadIdx.x] = s.x + s.y;
}
__global__ __launch_bounds__(64) void k_pkmul(float *out, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,
       a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 b = {1.0001f, 0.9999f};
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\nv_pk_mul_f32 %1, %1, %8\nv_pk_mul_f32 %2, %2, %8\n"
                          "v_pk_mul_f32 %3, %3, %8\nv_pk_mul_f32 %4, %4, %8\nv_pk_mul_f32 %5, %5, %8\n"
                          "v_pk_mul_f32 %6, %6, %8\nv_pk_mul_f32 %7, %7, %8\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                          : "v"(b));)
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 64 + threadIdx.x] = s.x + s.y;
}

// v_permlane32_swap: the cross-lane exchange of the backward's transposed reduction (8 per walked entry)
__global__ __launch_bounds__(64) void k_permswap(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_permlane32_swap_b32 %0, %1\nv_permlane32_swap_b32 %2, %3\nv_permlane32_swap_b32 %4, %5\n"
                          "v_permlane32_swap_b32 %6, %7\nv_permlane32_swap_b32 %1, %2\nv_permlane32_swap_b32 %3, %4\n"
                          "v_permlane32_swap_b32 %5, %6\nv_permlane32_swap_b32 %7, %0\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// the raster backward's per-(entry, quadrant) mix, as 16 independent instructions per group: 9 fma/mul/add-class,
// 2 v_cmp, 3 v_cndmask / v_min, 1 v_exp, 1 v_rcp  (profiles/scripts/valu_mix.py gives the static mix of the kernel)
__global__ __launch_bounds__(64) void k_mix(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\nv_mul_f32 %1, %1, %8\nv_fma_f32 %2, %2, %8, %9\nv_add_f32 %3, %3, %9\n"
                          "v_cmp_le_f32 vcc, %0, %8\nv_cndmask_b32 %4, %4, %8, vcc\nv_fma_f32 %5, %5, %8, %9\n"
                          "v_exp_f32 %6, %6\nv_fma_f32 %7, %7, %8, %9\nv_mul_f32 %0, %0, %8\n"
                          "v_cmp_ge_f32 vcc, %1, %9\nv_cndmask_b32 %2, %2, %8, vcc\nv_min_f32 %3, %3, %8\n"
                          "v_rcp_f32 %4, %4\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                          : "v"(b), "v"(c) : "vcc");)
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <typename F>
void run(const char *name, F kern, float *out, int waves_per_simd) {
    const int iters = 4000, grid = 1024 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)grid * iters * 64.0;          // wave-instructions
    const double per_simd_per_ns = instr / 1024.0 / (ms * 1e6);
    printf("%-10s waves/SIMD=%d  %.3f ms  %.3f wave-instr/ns/SIMD  => %.2f cycles/instr @2.4GHz\n", name,
           waves_per_simd, ms, per_simd_per_ns, 2.4 / per_simd_per_ns);
}

// --calib: ONE long launch (~0.3 ms) of each class at four waves per SIMD (the occupancy of the backward's short-walk
// kernel), for a rocprofv3 --pmc pass: each launch is 100 % VALU-issue-bound by construction, so what the SQ counters
// read on it IS their saturation value — the normalisation bench.py's roofline.valu block uses.
template <typename F>
void calib(const char *name, F kern, float *out, int iters, int per_iter) {
    const int grid = 1024 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("calib %-10s %.3f ms  wave-instructions %.4e\n", name, ms, (double)grid * iters * per_iter);
}

int main(int argc, char **argv) {
    float *out;
    hipMalloc(&out, 1024 * 8 * 64 * sizeof(float));
    if (argc > 1 && std::string(argv[1]) == "--calib") {
        calib("v_fma", k_fma, out, 1500, 64); calib("v_mul", k_mul, out, 1500, 64);
        calib("v_cmp", k_cmp, out, 900, 64); calib("v_cndmask", k_cndmask, out, 900, 64);
        calib("v_min", k_min, out, 900, 64); calib("v_exp", k_exp, out, 500, 64); calib("v_rcp", k_rcp, out, 500, 64);
        calib("v_add_dpp", k_dpp, out, 900, 64); calib("v_mov", k_mov, out, 1500, 64);
        calib("permswap", k_permswap, out, 900, 64); calib("mix", k_mix, out, 300, 128);
        return 0;
    }
    for (int w : {1, 2, 4}) {
        run("v_mul", k_mul, out, w); run("v_add", k_add, out, w); run("v_sub", k_sub, out, w);
        run("v_min", k_min, out, w); run("v_fma", k_fma, out, w); run("v_fmac", k_fmac, out, w);
        run("v_pk_fma", k_pkfma, out, w); run("v_pk_mul", k_pkmul, out, w);
        run("v_exp", k_exp, out, w); run("v_rcp", k_rcp, out, w); run("v_mov", k_mov, out, w);
        run("v_cndmask", k_cndmask, out, w); run("v_cmp", k_cmp, out, w); run("v_add_dpp", k_dpp, out, w);
    }
    return 0;
}
