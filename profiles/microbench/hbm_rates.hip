// hbm_rates.hip — what a plain streaming kernel reaches on this GPU, to read the SH / projection numbers against
// (their "roofline" column in DESIGN.md is the 8 TB/s datasheet peak).
//   hipcc --offload-arch=gfx950 -O3 profiles/microbench/hbm_rates.hip -o /tmp/hbm_rates && /tmp/hbm_rates
// read: every thread sums 8 float4 of a 192 MiB buffer (grid-stride, all loads issued before use), one store per thread;
// write: float4 fill; copy: float4 load + store.  20 timed launches each, hipEvent timing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ src, float *__restrict__ out, size_t n4) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    float acc = 0.f;
    for (size_t i = tid; i < n4; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (i + k * stride < n4) ? src[i + k * stride] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    out[tid] = acc;
}
__global__ __launch_bounds__(256) void k_write(float4 *__restrict__ dst, size_t n4) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (size_t i = tid; i < n4; i += stride) dst[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (size_t i = tid; i < n4; i += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (i + k * stride < n4) ? src[i + k * stride] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + k * stride < n4) dst[i + k * stride] = v[k];
    }
}

int main() {
    const size_t bytes = (size_t)(getenv("HBM_MB") ? atoi(getenv("HBM_MB")) : 192) << 20, n4 = bytes / 16;
    float4 *a, *b; float *out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, (size_t)65536 * 256 * 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 32768}) {
        for (int which = 0; which < 3; ++which) {
            auto launch = [&]() {
                if (which == 0) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, out, n4);
                if (which == 1) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n4);
                if (which == 2) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n4);
            };
            for (int i = 0; i < 3; ++i) launch();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 20; ++i) launch();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
            const double moved = (which == 2 ? 2.0 : 1.0) * bytes;
            printf("%-5s grid %6d: %7.1f us  %6.2f TB/s\n", which == 0 ? "read" : which == 1 ? "write" : "copy", grid,
                   ms * 1e3, moved / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
