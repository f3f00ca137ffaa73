// lookback_probe.hip — does a decoupled look-back ("one-sweep") prefix beat hist + scan for the sizes of THIS binning?
//   hipcc --offload-arch=gfx950 -O3 profiles/microbench/lookback_probe.hip -o /tmp/lookback_probe && /tmp/lookback_probe
//
// VERDICT r03 #3 asked for one-sweep sort passes and for the claim "the look-back degenerates when every tile is
// resident" to be MEASURED.  A radix pass needs, per tile and digit, the number of keys with that digit in all EARLIER
// tiles.  The library computes it with two kernels (rs_hist: per-tile digit histogram -> table; rs_scan: one workgroup
// per digit scans its row); one-sweep computes it inside the scatter kernel: a tile publishes its 256 digit counts as
// "aggregate" words, then every digit's thread walks back over the predecessors' words, adding aggregates until it
// meets an "inclusive" word, and publishes its own inclusive prefix.  The sizes here: the depth rank is 1 M keys in
// 1024-key tiles = 977 tiles; the tile sort 8.3 M keys in 4096-key tiles = 2032 tiles; 256 digits.  Every tile of either
// grid is resident at once on 256 CUs, so nothing has "already finished" when a tile starts looking back.
//
// Measured here (keys = hashed ints, resident in HBM; times = min of 20 launches, HIP events):
//   A  hist kernel + scan kernel            (the two launches one-sweep would remove; the library's own shapes)
//   B  ONE kernel: tile histogram + publish + look-back (thread per digit, serial walk) + publish inclusive
//   C  the same with a wave-parallel walk (64 predecessors per step, as CUB's one-sweep does)
// and every variant's exclusive prefixes are checked against the host.  Cross-workgroup words follow
// MI355X_MICROARCH.md "inter-workgroup visibility": agent-scope atomic stores / loads (write-through, L1-bypassing),
// the data IS the flag (2 flag bits + 30 count bits in one word), ticketed tile ids (an atomic counter), bounded spins.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(1))) uint32_t g_u32;
constexpr int NB = 256;                 // digits
constexpr uint32_t FLAG_AGG = 1u << 30, FLAG_INC = 2u << 30, VAL_MASK = (1u << 30) - 1u;

__device__ __forceinline__ uint32_t ld(const uint32_t *p) {
    return __hip_atomic_load((g_u32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st(uint32_t *p, uint32_t v) {
    __hip_atomic_store((g_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int IPT>
__device__ __forceinline__ void tile_hist(const uint32_t *__restrict__ keys, uint32_t n, int tile, uint32_t *hist) {
    for (int d = threadIdx.x; d < NB; d += 256) hist[d] = 0;
    __syncthreads();
    const uint32_t base = (uint32_t)tile * 256u * IPT;
    uint32_t k[IPT];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const uint32_t i = base + r * 256 + threadIdx.x;
        k[r] = i < n ? keys[i] : 0xffffffffu;
    }
#pragma unroll
    for (int r = 0; r < IPT; ++r)
        if (base + r * 256 + threadIdx.x < n) atomicAdd(&hist[k[r] & 255u], 1u);
    __syncthreads();
}

// ---- A: the two-kernel form
template <int IPT>
__global__ __launch_bounds__(256) void hist_kernel(const uint32_t *keys, uint32_t n, uint32_t nblk, uint32_t *table) {
    __shared__ uint32_t hist[NB];
    tile_hist<IPT>(keys, n, blockIdx.x, hist);
    table[(size_t)threadIdx.x * nblk + blockIdx.x] = hist[threadIdx.x];
}
__global__ __launch_bounds__(256) void scan_kernel(uint32_t nblk, uint32_t *table) {   // one workgroup per digit
    __shared__ uint32_t part[4];
    uint32_t *row = table + (size_t)blockIdx.x * nblk;
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nblk; b0 += 256 * 8) {
        uint32_t v[8], sum = 0;
        const uint32_t i0 = b0 + threadIdx.x * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = i0 + j < nblk ? row[i0 + j] : 0u; sum += v[j]; }
        uint32_t inc = sum;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d, 64); if (lane >= d) inc += u; }
        if (lane == 63) part[wave] = inc;
        __syncthreads();
        uint32_t off = 0, tot = 0;
        for (int w = 0; w < 4; ++w) { if (w < wave) off += part[w]; tot += part[w]; }
        uint32_t run = carry + off + inc - sum;
#pragma unroll
        for (int j = 0; j < 8; ++j) { if (i0 + j < nblk) row[i0 + j] = run; run += v[j]; }
        carry += tot;
        __syncthreads();
    }
}

// ---- B: one kernel, decoupled look-back, one thread per digit walks back serially.  status[tile][digit]
template <int IPT>
__global__ __launch_bounds__(256) void lookback_kernel(const uint32_t *keys, uint32_t n, uint32_t nblk, uint32_t *status,
                                                       uint32_t *ticket, uint32_t *excl_out, uint32_t *timeouts) {
    __shared__ uint32_t hist[NB];
    __shared__ int tile_s;
    if (threadIdx.x == 0) tile_s = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int tile = tile_s;
    tile_hist<IPT>(keys, n, tile, hist);
    uint32_t *mine = status + (size_t)tile * NB;
    const int d = threadIdx.x;
    const uint32_t cnt = hist[d];
    st(mine + d, (tile == 0 ? FLAG_INC : FLAG_AGG) | cnt);
    uint32_t excl = 0;
    for (int t = tile - 1; t >= 0;) {
        uint32_t s = ld(status + (size_t)t * NB + d);
        unsigned spins = 0;
        while ((s >> 30) == 0u) {
            if (++spins > (1u << 22)) { atomicAdd(timeouts, 1u); return; }
            __builtin_amdgcn_s_sleep(1);
            s = ld(status + (size_t)t * NB + d);
        }
        excl += s & VAL_MASK;
        if ((s >> 30) == 2u) break;
        --t;
    }
    if (tile > 0) st(mine + d, FLAG_INC | (excl + cnt));
    excl_out[(size_t)d * nblk + tile] = excl;
}

// ---- C: digits handled 64 per wave, each digit's look-back done by the whole wave (64 predecessors per step)
template <int IPT>
__global__ __launch_bounds__(256) void lookback_wave_kernel(const uint32_t *keys, uint32_t n, uint32_t nblk,
                                                            uint32_t *status, uint32_t *ticket, uint32_t *excl_out,
                                                            uint32_t *timeouts) {
    __shared__ uint32_t hist[NB];
    __shared__ uint32_t excl_s[NB];
    __shared__ int tile_s;
    if (threadIdx.x == 0) tile_s = (int)atomicAdd(ticket, 1u);
    __syncthreads();
    const int tile = tile_s;
    tile_hist<IPT>(keys, n, tile, hist);
    uint32_t *mine = status + (size_t)tile * NB;
    st(mine + threadIdx.x, (tile == 0 ? FLAG_INC : FLAG_AGG) | hist[threadIdx.x]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = 0; j < 64; ++j) {
        const int d = wave * 64 + j;
        uint32_t excl = 0;
        int hi = tile - 1;                                   // nearest predecessor not yet accounted for
        bool done = hi < 0;
        while (!done) {
            const int t = hi - lane;
            uint32_t s = t >= 0 ? ld(status + (size_t)t * NB + d) : FLAG_INC;      // below tile 0: inclusive zero
            unsigned spins = 0;
            while (__ballot((s >> 30) == 0u) != 0ull) {     // somebody not published yet: re-read (wave-uniform loop)
                if (++spins > (1u << 22)) { if (lane == 0) atomicAdd(timeouts, 1u); return; }
                __builtin_amdgcn_s_sleep(1);
                if ((s >> 30) == 0u) s = ld(status + (size_t)t * NB + d);
            }
            const unsigned long long inc = __ballot((s >> 30) == 2u);
            const int first_inc = inc ? __ffsll((long long)inc) - 1 : 64;          // lane of the nearest inclusive word
            uint32_t v = lane <= first_inc ? (s & VAL_MASK) : 0u;
#pragma unroll
            for (int k = 32; k >= 1; k >>= 1) v += __shfl_xor(v, k, 64);
            excl += v;
            if (inc) done = true; else hi -= 64;
        }
        if (lane == 0) {
            excl_s[d] = excl;
            if (tile > 0) st(mine + d, FLAG_INC | (excl + hist[d]));
        }
    }
    __syncthreads();
    excl_out[(size_t)threadIdx.x * nblk + tile] = excl_s[threadIdx.x];
}

template <int IPT>
static int run(const char *name, uint32_t n) {
    const uint32_t nblk = (n + 256 * IPT - 1) / (256 * IPT);
    std::vector<uint32_t> hk(n);
    for (uint32_t i = 0; i < n; ++i) hk[i] = (i * 2654435761u) >> 11;
    uint32_t *keys, *table, *status, *ticket, *excl, *tmo;
    CHECK(hipMalloc(&keys, n * 4)); CHECK(hipMalloc(&table, (size_t)NB * nblk * 4)); CHECK(hipMalloc(&status, (size_t)NB * nblk * 4));
    CHECK(hipMalloc(&ticket, 4)); CHECK(hipMalloc(&excl, (size_t)NB * nblk * 4)); CHECK(hipMalloc(&tmo, 4));
    CHECK(hipMemcpy(keys, hk.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(tmo, 0, 4));
    // host reference
    std::vector<uint32_t> ref((size_t)NB * nblk, 0), run_(NB, 0);
    for (uint32_t t = 0; t < nblk; ++t) {
        for (int d = 0; d < NB; ++d) ref[(size_t)d * nblk + t] = run_[d];
        const uint32_t lo = t * 256 * IPT, hi = std::min(n, lo + 256 * IPT);
        for (uint32_t i = lo; i < hi; ++i) run_[hk[i] & 255u]++;
    }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto check = [&](uint32_t *dev, const char *what) {
        std::vector<uint32_t> got((size_t)NB * nblk);
        if (hipMemcpy(got.data(), dev, got.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
        size_t bad = 0;
        for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
        uint32_t t = 0; (void)hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost);
        if (bad || t) printf("  %s: %zu WRONG prefixes, %u timeouts\n", what, bad, t);
        return bad == 0 && t == 0;
    };
    float best[3] = {1e9f, 1e9f, 1e9f};
    bool ok[3] = {true, true, true};
    for (int it = 0; it < 20; ++it) {
        float ms;
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(hist_kernel<IPT>, dim3(nblk), dim3(256), 0, 0, keys, n, nblk, table);
        hipLaunchKernelGGL(scan_kernel, dim3(NB), dim3(256), 0, 0, nblk, table);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        best[0] = std::min(best[0], ms);
        if (it == 0) ok[0] = check(table, "hist + scan");
        CHECK(hipMemsetAsync(status, 0, (size_t)NB * nblk * 4)); CHECK(hipMemsetAsync(ticket, 0, 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(lookback_kernel<IPT>, dim3(nblk), dim3(256), 0, 0, keys, n, nblk, status, ticket, excl, tmo);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        best[1] = std::min(best[1], ms);
        if (it == 0) ok[1] = check(excl, "look-back, thread per digit");
        CHECK(hipMemsetAsync(status, 0, (size_t)NB * nblk * 4)); CHECK(hipMemsetAsync(ticket, 0, 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(lookback_wave_kernel<IPT>, dim3(nblk), dim3(256), 0, 0, keys, n, nblk, status, ticket, excl, tmo);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        best[2] = std::min(best[2], ms);
        if (it == 0) ok[2] = check(excl, "look-back, wave per digit");
    }
    printf("%-44s n = %8u, %4u tiles of %4d keys: hist + scan %6.1f us%s | look-back (thread per digit) %7.1f us%s | "
           "look-back (wave per digit) %7.1f us%s\n", name, n, nblk, 256 * IPT, best[0] * 1e3f, ok[0] ? "" : " (WRONG)",
           best[1] * 1e3f, ok[1] ? "" : " (WRONG)", best[2] * 1e3f, ok[2] ? "" : " (WRONG)");
    (void)hipFree(keys); (void)hipFree(table); (void)hipFree(status); (void)hipFree(ticket); (void)hipFree(excl); (void)hipFree(tmo);
    return 0;
}

int main() {
    if (run<4>("depth rank (metric): 1 M keys", 1000000u)) return 1;
    if (run<4>("depth rank (C4): 2 M keys", 2000000u)) return 1;
    if (run<16>("tile sort (metric): 8.3 M pairs", 8321119u)) return 1;
    if (run<16>("tile sort (street-like): 16 M pairs", 16000000u)) return 1;
    return 0;
}
