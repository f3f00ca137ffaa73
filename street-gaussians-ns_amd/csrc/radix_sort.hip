// radix_sort.hip — on-device stable LSD radix sorts, gfx950.
//
// Replaces `torch.sort(isect_ids)` + `torch.gather(gaussian_ids)` (CUB radix sort) inside
// gsplat/utils.py:bin_and_sort_gaussians, reached from the reference through
// rasterize_gaussians (sgn_splatfacto.py:954-967, :982-994).  Stable => equal keys keep their input
// order exactly like upstream, so gaussian_ids_sorted is bit-exact.
//
// One template, two instantiations:
//   <u64, int32 payload, 8-bit digits>  sgn_sort_pairs    upstream-shaped (tile<<32|depth, gaussian id)
//   <u32, int32 payload, 8-bit digits>  sgn_sort_pairs32  fused path: depth ranking of the N Gaussians
//                                                         (32 bits) and the tile sort of the I pairs (14 bits)
// Three kernels per pass —
//   rs_hist    per-tile (4096 keys) digit histogram in LDS            -> table[digit][tile]
//   rs_scan    one workgroup per digit: exclusive scan of its row     -> table (in place), total[digit]
//   rs_scatter per tile: stable ranking inside a wave on per-wave LDS counters (wave64 ballot "match" ranking, or —
//              once sgn_sort_selftest has proved it on this device — one returning LDS atomic per key), cross-wave prefix,
//              then the tile is reordered in LDS so the global stores of one digit are consecutive lanes ->
//              consecutive addresses.
// HBM traffic per key per pass: sizeof(key) (hist) + 2 * (sizeof(key) + payload).
#include "sgn_common.h"

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = 4;
// Ranking inside a wave (the sort must be stable).  Both forms are compiled (template parameter ATOMIC):
//   false: the ballot-match ranking — one ballot per digit bit, DOCUMENTED ISA semantics only.  This is what runs
//          until a device has passed the probe below.
//   true : ONE returning LDS atomic per key on the digit's per-wave counter (8x fewer instructions in the ranking
//          section, -7 us per binning).  Stable iff lanes of one ds_add_rtn_u32 that hit the same address are served in
//          ascending lane order — how the gfx950 LDS resolves same-address lanes, observed rather than documented.
// The bit-exact depth order of every tile list depends on it, so the DEFAULT is the documented form and the choice is an
// ARGUMENT of every sorting entry point (`sort_rank_mode`: 0 ballot, 1 atomic) — the library keeps no state about it
// (round 5; rounds 2-4 kept a process-wide switch that a load-time probe flipped).  sgn_sort_selftest() queues
// adversarial same-digit / lane-interleaved probes sorted with BOTH rankings and counts differing output pairs on the
// device; the host side (sgn_rast/_lib.py) runs it only when asked for the atomic form (SGN_SORT_RANK=atomic), UNDER
// LOAD (a second instance and a GEMM on other streams, >= 1000 sorts), per device, and passes 1 only where it counted
// zero.  tests/test_gpu_sort_stability.py stresses both forms.
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "radix_sort.hip is written for gfx950 (MI355X) only"
#endif
// keys per thread: 16 (4096-key tiles) for large inputs; 4 (1024-key tiles) below RS_SMALL_N keys, where 4096-key
// tiles would leave most of the 256 CUs idle (1 M keys = 245 tiles) and every pass latency-bound
#ifndef SGN_RS_IPT_LARGE
#define SGN_RS_IPT_LARGE 16
#endif
#ifndef SGN_RANK_BITS
#define SGN_RANK_BITS 8
#endif
#ifndef SGN_RANK_IPT
#define SGN_RANK_IPT 0
#endif
constexpr int RS_IPT_LARGE = SGN_RS_IPT_LARGE, RS_IPT_SMALL = 4;
constexpr uint32_t RS_SMALL_N = 3u << 19;   // (r05h: at 2 M keys the 4096-key tiles win by 15 us, at 1 M they lose by 11)
inline int rs_pick_ipt(int64_t n) { return n < (int64_t)RS_SMALL_N ? RS_IPT_SMALL : RS_IPT_LARGE; }

template <typename K>
__device__ __forceinline__ unsigned digit_of(K key, int shift, unsigned mask) {
    return (unsigned)(key >> shift) & mask;
}

// FROM_DEPTH (round 5): the FIRST pass of the depth ranking reads its keys straight from the projection's outputs — key i
// = bits of depths[i] for a visible Gaussian (radii[i] > 0; positive floats order like their bit patterns), all ones for
// a culled one, payload i — instead of from a key / index pair a separate kernel wrote (depth_keys_kernel: one launch,
// 8 MB written and 12 MB read again per million Gaussians).  `keys` then points at depths, `vals_in` at radii.
__device__ __forceinline__ uint32_t depth_key(const void *depths, const void *radii, uint32_t i) {
    return reinterpret_cast<const int32_t *>(radii)[i] > 0 ? reinterpret_cast<const uint32_t *>(depths)[i] : 0xFFFFFFFFu;
}

template <typename K, int BITS, int RS_IPT, bool FROM_DEPTH = false>
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(uint32_t n, const K *__restrict__ keys, int shift,
                                                             unsigned dmask, uint32_t nblk,
                                                             uint32_t *__restrict__ table,
                                                             const int32_t *__restrict__ n_dev,
                                                             const int32_t *__restrict__ radii = nullptr) {
    constexpr int NB = 1 << BITS;
    if (n_dev) n = min(n, (uint32_t)max(*n_dev, 0));   // speculative launch: n is the capacity, *n_dev the count
    constexpr int RS_TILE = RS_THREADS * RS_IPT;
    __shared__ uint32_t hist[NB];
    for (int d = threadIdx.x; d < NB; d += RS_THREADS) hist[d] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE;
    if constexpr (FROM_DEPTH) {
        static_assert(sizeof(K) == 4, "depth keys are 32-bit");
        uint32_t kd[RS_IPT];
#pragma unroll
        for (int k = 0; k < RS_IPT; ++k) {                  // all loads first (two per key, coalesced)
            const uint32_t i = base + k * RS_THREADS + threadIdx.x;
            kd[k] = (i < n) ? depth_key(keys, radii, i) : 0u;
        }
#pragma unroll
        for (int k = 0; k < RS_IPT; ++k) {
            const uint32_t i = base + k * RS_THREADS + threadIdx.x;
            if (i < n) atomicAdd(&hist[digit_of<uint32_t>(kd[k], shift, dmask)], 1u);
        }
        __syncthreads();
        for (int d = threadIdx.x; d < NB; d += RS_THREADS) table[(size_t)d * nblk + blockIdx.x] = hist[d];
        return;
    }
    // Which key a thread counts does not matter for a histogram, so a thread takes 16-byte vectors of consecutive keys
    // (8 sixteen-bit tile ids per load instead of eight 2-byte loads: the 16-bit histogram of the 8.3 M-pair tile sort
    // took 11.2 us for 16.6 MB).  All loads are issued before the first LDS atomic.
    constexpr int VEC = 16 / (int)sizeof(K);
    if constexpr (RS_IPT % VEC == 0) {
        if ((reinterpret_cast<uintptr_t>(keys) & 15u) == 0) {       // workgroup-uniform
            constexpr int NV = RS_IPT / VEC;
            K kv[NV][VEC];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const uint32_t i = base + (v * RS_THREADS + threadIdx.x) * VEC;
                if (i + VEC <= n) {
                    const uint4 raw = *reinterpret_cast<const uint4 *>(keys + i);
                    __builtin_memcpy(kv[v], &raw, 16);
                } else {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) kv[v][j] = (i + j < n) ? keys[i + j] : (K)0;
                }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const uint32_t i = base + (v * RS_THREADS + threadIdx.x) * VEC;
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if (i + j < n) atomicAdd(&hist[digit_of<K>(kv[v][j], shift, dmask)], 1u);
            }
            __syncthreads();
            for (int d = threadIdx.x; d < NB; d += RS_THREADS) table[(size_t)d * nblk + blockIdx.x] = hist[d];
            return;
        }
    }
    K kk[RS_IPT];                       // unaligned input / short tiles: one key per load, all loads first
#pragma unroll
    for (int k = 0; k < RS_IPT; ++k) {
        const uint32_t i = base + k * RS_THREADS + threadIdx.x;
        kk[k] = (i < n) ? keys[i] : (K)0;
    }
#pragma unroll
    for (int k = 0; k < RS_IPT; ++k) {
        const uint32_t i = base + k * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&hist[digit_of<K>(kk[k], shift, dmask)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < NB; d += RS_THREADS) table[(size_t)d * nblk + blockIdx.x] = hist[d];
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t u = __shfl_up(v, d, 64);
        if (lane >= d) v += u;
    }
    return v;
}

// exclusive scan of one value per thread over a 256-thread block
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t *lds4, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t w = wave_incl_scan_u32(v);
    if (lane == 63) lds4[wave] = w;
    __syncthreads();
    uint32_t off = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t t = lds4[k];
        if (k < wave) off += t;
    }
    total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    __syncthreads();
    return w + off - v;
}

// grid = number of digits (one workgroup per digit): table[d][0..nblk) -> exclusive prefix,
// totals[d] = row sum.  Each thread owns RS_SCAN_ITEMS consecutive entries and requests all of them before the
// first use, so a row of up to 256 * RS_SCAN_ITEMS tiles (8 M keys in 4096-key tiles) is ONE global round trip and
// ONE block scan; the kernel is pure latency (a few KB per workgroup), and as a 256-entry-per-iteration loop it paid
// four to eight dependent round trips and twice as many barriers per launch, six launches per binning.
constexpr int RS_SCAN_ITEMS = 8;
__global__ __launch_bounds__(RS_THREADS) void rs_scan_kernel(uint32_t nblk, uint32_t *__restrict__ table,
                                                             uint32_t *__restrict__ totals) {
    __shared__ uint32_t lds4[4];
    uint32_t *row = table + (size_t)blockIdx.x * nblk;
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nblk; b0 += RS_THREADS * RS_SCAN_ITEMS) {
        const uint32_t i0 = b0 + threadIdx.x * RS_SCAN_ITEMS;
        uint32_t v[RS_SCAN_ITEMS], sum = 0;
#pragma unroll
        for (int k = 0; k < RS_SCAN_ITEMS; ++k) v[k] = (i0 + k < nblk) ? row[i0 + k] : 0u;
#pragma unroll
        for (int k = 0; k < RS_SCAN_ITEMS; ++k) sum += v[k];
        uint32_t total;
        uint32_t run = carry + block_excl_scan_u32(sum, lds4, total);
#pragma unroll
        for (int k = 0; k < RS_SCAN_ITEMS; ++k) {
            if (i0 + k < nblk) row[i0 + k] = run;
            run += v[k];
        }
        carry += total;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

template <typename K, bool HAS_VAL, int BITS, int RS_IPT, bool ATOMIC, bool FROM_DEPTH = false>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(
    uint32_t n, const K *__restrict__ keys_in, const int32_t *__restrict__ vals_in, K *__restrict__ keys_out,
    int32_t *__restrict__ vals_out, int shift, unsigned dmask, uint32_t nblk, const uint32_t *__restrict__ table,
    const uint32_t *__restrict__ totals, const int32_t *__restrict__ n_dev) {
    constexpr int NB = 1 << BITS;
    if (n_dev) n = min(n, (uint32_t)max(*n_dev, 0));
    if (blockIdx.x * (RS_THREADS * RS_IPT) >= n) return;   // workgroup-uniform
    constexpr int DPT = NB / RS_THREADS;  // digits owned per thread in the prefix phase (1 or 2)
    constexpr int RS_TILE = RS_THREADS * RS_IPT;   // keys per workgroup
    constexpr int RS_WAVE_ITEMS = 64 * RS_IPT;     // keys per wave
    static_assert(NB % RS_THREADS == 0, "digit count must be a multiple of the block size");
    __shared__ K skeys[RS_TILE];
    __shared__ int32_t svals[HAS_VAL ? RS_TILE : 1];
    __shared__ uint32_t wcnt[RS_WAVES][NB];
    __shared__ uint32_t dstart[NB];
    __shared__ uint32_t gbase[NB];
    __shared__ uint32_t lds4[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tile_base = blockIdx.x * RS_TILE;
    const uint32_t tile_cnt = min((uint32_t)RS_TILE, n - tile_base);   // > 0: checked above
    K key[RS_IPT];
    int32_t val[RS_IPT];
    uint32_t rank[RS_IPT];
    // the wave's counters through an explicit LDS pointer: as a generic `volatile uint32_t *` every access was compiled
    // to flat_load / flat_store with sc0 sc1 and a full s_waitcnt vmcnt(0) behind it (two per key)
    typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
    volatile lds_u32_t *mycnt = (volatile lds_u32_t *)(&wcnt[wave][0]);   // ballot form only
    // every key / value of the tile is requested before the first ranking step: the volatile LDS counters below
    // pin program order, so loads left inside the ranking loop are waited for one HBM round trip at a time
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
        const uint32_t li = wave * RS_WAVE_ITEMS + r * 64 + lane;  // index inside the tile
        const bool valid = li < tile_cnt;
        if constexpr (FROM_DEPTH) {           // keys_in = depths, vals_in = radii (see depth_key)
            key[r] = valid ? (K)depth_key(keys_in, vals_in, tile_base + li) : (K)~(K)0;
            if constexpr (HAS_VAL) val[r] = (int32_t)(tile_base + li);
        } else {
            key[r] = valid ? keys_in[tile_base + li] : (K)~(K)0;
            if constexpr (HAS_VAL) val[r] = valid ? vals_in[tile_base + li] : 0;
        }
    }
    // the digit totals and this tile's column of the scanned table do not depend on the ranking: requested here, with
    // the keys, instead of after the ranking barrier (a strided global read per thread: ~2 us on the critical path of
    // every workgroup, and the 1 M-key depth passes are ONE round of workgroups, i.e. pure workgroup latency)
    uint32_t gtot[DPT], tcol[DPT];
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        const int d = tid * DPT + j;
        gtot[j] = totals[d];
        tcol[j] = table[(size_t)d * nblk + blockIdx.x];
    }
    // counters cleared while the loads are in flight
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w)
#pragma unroll
        for (int j = 0; j < DPT; ++j) wcnt[w][tid * DPT + j] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
        const uint32_t li = wave * RS_WAVE_ITEMS + r * 64 + lane;
        const bool valid = li < tile_cnt;
        const unsigned d = digit_of<K>(key[r], shift, dmask);
        if constexpr (ATOMIC) {
        // one returning LDS atomic per key (see the note on ATOMIC above): the value it returns is the key's rank among
        // the wave's keys of the same digit seen so far — earlier keys (r) by program order, same-r keys by lane order
        uint32_t rk = 0;
        if (valid) rk = __hip_atomic_fetch_add((lds_u32_t *)&wcnt[wave][d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");     // keep the atomics of successive keys in program order
        rank[r] = rk;
        } else {
        // match mask m = lanes of this wave holding the same digit: per bit, keep the lanes that voted like me.  Spelled
        // on 32-bit halves with nb = all-ones where my bit is set, m &= ~(vote ^ nb) is one v_bitop3 per half (the
        // `bit ? vote : ~vote` form on a 64-bit value took 13 instructions per bit); invalid lanes are masked once, by
        // the initial value, and passes narrower than BITS stop early (wave-uniform).
        const unsigned long long vmask = __ballot(valid);
        uint32_t m_lo = (uint32_t)vmask, m_hi = (uint32_t)(vmask >> 32);
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            if (!((dmask >> b) & 1u)) break;
            const uint32_t nb = (uint32_t)(((int32_t)(d << (31 - b))) >> 31);   // v_bfe_i32: all ones where bit b is set
            const unsigned long long vote = __builtin_amdgcn_ballot_w64(nb != 0u);
            m_lo &= ~((uint32_t)vote ^ nb);
            m_hi &= ~((uint32_t)(vote >> 32) ^ nb);
        }
        // lanes of my group below me (v_mbcnt): my rank inside the group; the group's first lane has none
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
        // lanes of one match group read the wave's running counter, the first lane of the group bumps it.
        // LDS ops of one wave retire in order, `volatile` keeps program order.
        uint32_t prev = 0;
        if (valid) {
            prev = mycnt[d];
            if (below == 0u) mycnt[d] = prev + (uint32_t)(__popc(m_lo) + __popc(m_hi));
        }
        rank[r] = prev + below;
        }
    }
    __syncthreads();

    // thread t owns digits [t*DPT, t*DPT+DPT): per-wave counts -> exclusive per-wave offsets + totals
    uint32_t tot[DPT], tsum = 0, gsum = 0;
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        const int d = tid * DPT + j;
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            const uint32_t c = wcnt[w][d];
            wcnt[w][d] = t;
            t += c;
        }
        tot[j] = t;
        tsum += t;
        gsum += gtot[j];
    }
    uint32_t dummy;
    uint32_t dst = block_excl_scan_u32(tsum, lds4, dummy);  // start of this thread's first digit in the tile
    uint32_t gst = block_excl_scan_u32(gsum, lds4, dummy);  // global start of this thread's first digit
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        const int d = tid * DPT + j;
        dstart[d] = dst;
        gbase[d] = gst + tcol[j] - dst;  // global pos = gbase[d] + local pos
        dst += tot[j];
        gst += gtot[j];
    }
    __syncthreads();

#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
        const uint32_t li = wave * RS_WAVE_ITEMS + r * 64 + lane;
        if (li < tile_cnt) {
            const unsigned d = digit_of<K>(key[r], shift, dmask);
            const uint32_t lp = dstart[d] + wcnt[wave][d] + rank[r];
            skeys[lp] = key[r];
            if constexpr (HAS_VAL) svals[lp] = val[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RS_IPT; ++k) {
        const uint32_t lp = k * RS_THREADS + tid;
        if (lp < tile_cnt) {
            const K kk = skeys[lp];
            const uint32_t pos = gbase[digit_of<K>(kk, shift, dmask)] + lp;
            keys_out[pos] = kk;
            if constexpr (HAS_VAL) vals_out[pos] = svals[lp];
        }
    }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename K, bool HAS_VAL, int BITS>
size_t sort_ws_bytes(int64_t n) {
    if (n <= 0) return 256;
    const size_t nblk = (size_t)sgn_cdiv(n, RS_THREADS * rs_pick_ipt(n));
    return align256((size_t)n * sizeof(K)) + (HAS_VAL ? align256((size_t)n * 4) : 0) +
           align256(((size_t)1 << BITS) * nblk * 4) + align256(((size_t)1 << BITS) * 4);
}

// ping-pong LSD passes; the last pass lands in keys_out / vals_out
template <typename K, bool HAS_VAL, int BITS, int RS_IPT, bool ATOMIC>
void sort_launch_ipt(uint32_t n, int begin_bit, int end_bit, const K *keys_in, const int32_t *vals_in, K *keys_out,
                     int32_t *vals_out, void *ws, hipStream_t s, const int32_t *n_dev, bool from_depth = false) {
    constexpr int NB = 1 << BITS;
    const uint32_t nblk = (uint32_t)sgn_cdiv(n, RS_THREADS * RS_IPT);
    char *p = (char *)ws;
    K *alt_keys = (K *)p; p += align256((size_t)n * sizeof(K));
    int32_t *alt_vals = nullptr;
    if (HAS_VAL) { alt_vals = (int32_t *)p; p += align256((size_t)n * 4); }
    uint32_t *table = (uint32_t *)p; p += align256((size_t)NB * nblk * 4);
    uint32_t *totals = (uint32_t *)p;
    const int npass = (end_bit - begin_bit + BITS - 1) / BITS;
    // balanced digits (14 tile bits -> 7 + 7, not 8 + 6): fewer buckets per pass = longer contiguous store runs
    const int bpp = (end_bit - begin_bit + npass - 1) / npass;
    const K *src_k = keys_in;
    const int32_t *src_v = vals_in;
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = begin_bit + bpp * pass;
        const int bits = (end_bit - shift) < bpp ? (end_bit - shift) : bpp;
        const unsigned dmask = (1u << bits) - 1u;
        const bool to_out = ((npass - 1 - pass) % 2) == 0;
        K *dst_k = to_out ? keys_out : alt_keys;
        int32_t *dst_v = to_out ? vals_out : alt_vals;
        bool depth_pass = false;
        if constexpr (sizeof(K) == 4 && HAS_VAL) depth_pass = from_depth && pass == 0;
        if (depth_pass) {
            if constexpr (sizeof(K) == 4 && HAS_VAL) {   // keys_in = depths, vals_in = radii
                hipLaunchKernelGGL((rs_hist_kernel<K, BITS, RS_IPT, true>), dim3(nblk), dim3(RS_THREADS), 0, s, n, src_k, shift,
                                   dmask, nblk, table, n_dev, src_v);
                hipLaunchKernelGGL(rs_scan_kernel, dim3(NB), dim3(RS_THREADS), 0, s, nblk, table, totals);
                hipLaunchKernelGGL((rs_scatter_kernel<K, HAS_VAL, BITS, RS_IPT, ATOMIC, true>), dim3(nblk), dim3(RS_THREADS), 0,
                                   s, n, src_k, src_v, dst_k, dst_v, shift, dmask, nblk, table, totals, n_dev);
            }
        } else {
        hipLaunchKernelGGL((rs_hist_kernel<K, BITS, RS_IPT>), dim3(nblk), dim3(RS_THREADS), 0, s, n, src_k, shift, dmask,
                           nblk, table, n_dev, (const int32_t *)nullptr);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(NB), dim3(RS_THREADS), 0, s, nblk, table, totals);
        hipLaunchKernelGGL((rs_scatter_kernel<K, HAS_VAL, BITS, RS_IPT, ATOMIC>), dim3(nblk), dim3(RS_THREADS), 0, s, n, src_k, src_v,
                           dst_k, dst_v, shift, dmask, nblk, table, totals, n_dev);
        }
        src_k = dst_k;
        src_v = dst_v;
    }
}

// rank_mode 0: ballot-match ranking (documented semantics), 1: returning-atomic ranking (the caller proved it on its
// device with sgn_sort_selftest, or forces it for an A/B)
template <typename K, bool HAS_VAL, int BITS>
void sort_launch(uint32_t n, int begin_bit, int end_bit, const K *keys_in, const int32_t *vals_in, K *keys_out,
                 int32_t *vals_out, void *ws, hipStream_t s, const int32_t *n_dev, int rank_mode, int force_ipt = 0,
                 bool from_depth = false) {
    // n_dev != nullptr: n is the CAPACITY the launch is sized for, the element count is read on the device
    const bool atomic = rank_mode != 0;
    const int ipt = force_ipt ? force_ipt : rs_pick_ipt(n);
#define SGN_RS_GO(IPT, AT) \
    sort_launch_ipt<K, HAS_VAL, BITS, IPT, AT>(n, begin_bit, end_bit, keys_in, vals_in, keys_out, vals_out, ws, s, n_dev, \
                                               from_depth)
    if (ipt == RS_IPT_SMALL) { if (atomic) SGN_RS_GO(RS_IPT_SMALL, true); else SGN_RS_GO(RS_IPT_SMALL, false); }
    else                     { if (atomic) SGN_RS_GO(RS_IPT_LARGE, true); else SGN_RS_GO(RS_IPT_LARGE, false); }
#undef SGN_RS_GO
}

// ---- probe of the returning-atomic ranking (sgn_sort_selftest)
// Pattern p of the probe: p = 0 every key equal (all 64 lanes of every wave on ONE counter), 1..4 lane l holds key
// l % (p + 1) (interleaved lanes), 5 runs of 37 equal keys, 6 a multiplicative hash (256 digits), 7 hash % 3.
__global__ __launch_bounds__(256) void rs_probe_fill_kernel(uint32_t n, int pattern, uint32_t *__restrict__ keys,
                                                            int32_t *__restrict__ vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = (i * 2654435761u) >> 13;
    uint32_t k;
    switch (pattern) {
        case 0: k = 5u; break;
        case 1: case 2: case 3: case 4: k = i % (uint32_t)(pattern + 1); break;
        case 5: k = (i / 37u) % 5u; break;
        case 6: k = h & 255u; break;
        default: k = h % 3u; break;
    }
    keys[i] = k;
    vals[i] = (int32_t)i;
}
__global__ __launch_bounds__(256) void rs_probe_compare_kernel(uint32_t n, const uint32_t *__restrict__ ka,
                                                               const int32_t *__restrict__ va,
                                                               const uint32_t *__restrict__ kb,
                                                               const int32_t *__restrict__ vb,
                                                               int32_t *__restrict__ mismatches) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // kb / vb come from the ballot ranking (documented semantics): it must itself be sorted and stable, and the
    // atomic ranking must reproduce it pair for pair
    bool bad = ka[i] != kb[i] || va[i] != vb[i];
    if (i + 1 < n) bad = bad || kb[i] > kb[i + 1] || (kb[i] == kb[i + 1] && vb[i] >= vb[i + 1]);
    if (bad) atomicAdd(mismatches, 1);
}

}  // namespace

// internal (fused binning path, binning.hip)
size_t sgn_sort_pairs32_ws_bytes(int64_t n) {
    const size_t a = sort_ws_bytes<uint32_t, true, 8>(n), b = sort_ws_bytes<uint32_t, true, SGN_RANK_BITS>(n);
    return a > b ? a : b;
}
void sgn_sort_pairs32_launch(uint32_t n, int end_bit, const uint32_t *kin, const int32_t *vin, uint32_t *kout,
                             int32_t *vout, void *ws, hipStream_t s, const int32_t *n_dev, int rank_mode) {
    sort_launch<uint32_t, true, 8>(n, 0, end_bit, kin, vin, kout, vout, ws, s, n_dev, rank_mode);
}

// the depth ranking: keys read from (depths, radii) by the first pass, payload = index (see FROM_DEPTH).
// SGN_RANK_BITS / SGN_RANK_IPT: A/B builds of round 5 (profiles/scripts/r05h.sh) — 11-bit digits make it three passes
// instead of four at the price of a 2048-column table and two-key store runs; measured, the shipped form stays 8 bits.
void sgn_sort_depth_rank_launch(uint32_t n, const float *depths, const int32_t *radii, uint32_t *kout, int32_t *vout,
                                void *ws, hipStream_t s, int rank_mode) {
    sort_launch<uint32_t, true, SGN_RANK_BITS>(n, 0, 32, (const uint32_t *)depths, radii, kout, vout, ws, s, nullptr,
                                               rank_mode, SGN_RANK_IPT, true);
}

// 16-bit keys (tile ids of images with <= 65536 tiles); same workspace layout and size as the 32-bit entry
void sgn_sort_pairs16_launch(uint32_t n, int end_bit, const uint16_t *kin, const int32_t *vin, uint16_t *kout,
                             int32_t *vout, void *ws, hipStream_t s, const int32_t *n_dev, int rank_mode) {
    sort_launch<uint16_t, true, 8>(n, 0, end_bit, kin, vin, kout, vout, ws, s, n_dev, rank_mode);
}

constexpr uint32_t RS_PROBE_N = 1u << 16;   // 64 K pairs per probe: 64 tiles of 1024 keys, 16 tiles of 4096
SGN_EXPORT size_t sgn_sort_selftest_workspace_bytes(void) {
    return 6 * align256((size_t)RS_PROBE_N * 4) + 256 + sort_ws_bytes<uint32_t, true, 8>(RS_PROBE_N);
}

// QUEUES `rounds` x (eight adversarial 64 K-pair probes, one 8-bit pass each, both tile sizes) sorted with the ballot
// ranking and with the returning-atomic ranking on `stream`, every output pair compared on the device; differing pairs
// are ADDED to *mismatches (device int32; the caller zeroes it, synchronises and reads it).  Asynchronous and
// stateless, so several instances (own workspaces) can run on several streams at once: the host's "under load" probe.
SGN_EXPORT int sgn_sort_selftest(void *ws, size_t ws_bytes, int rounds, int32_t *mismatches, sgn_stream_t stream) {
    SGN_ARG_CHECK(ws != nullptr && ws_bytes >= sgn_sort_selftest_workspace_bytes() && rounds >= 1 && mismatches, -1);
    hipStream_t s = (hipStream_t)stream;
    char *p = (char *)ws;
    auto take = [&](size_t b) { char *q = p; p += align256(b); return q; };
    uint32_t *kin = (uint32_t *)take((size_t)RS_PROBE_N * 4);
    int32_t *vin = (int32_t *)take((size_t)RS_PROBE_N * 4);
    uint32_t *ka = (uint32_t *)take((size_t)RS_PROBE_N * 4);
    int32_t *va = (int32_t *)take((size_t)RS_PROBE_N * 4);
    uint32_t *kb = (uint32_t *)take((size_t)RS_PROBE_N * 4);
    int32_t *vb = (int32_t *)take((size_t)RS_PROBE_N * 4);
    (void)take(256);
    void *sws = (void *)p;
    const uint32_t n = RS_PROBE_N - 37;       // a ragged last tile
    for (int r = 0; r < rounds; ++r)
        for (int pattern = 0; pattern < 8; ++pattern) {
            hipLaunchKernelGGL(rs_probe_fill_kernel, dim3(sgn_cdiv(n, 256)), dim3(256), 0, s, n, pattern, kin, vin);
            for (int ipt : {RS_IPT_SMALL, RS_IPT_LARGE}) {
                sort_launch<uint32_t, true, 8>(n, 0, 8, kin, vin, kb, vb, sws, s, nullptr, /*ballot*/ 0, ipt);
                sort_launch<uint32_t, true, 8>(n, 0, 8, kin, vin, ka, va, sws, s, nullptr, /*atomic*/ 1, ipt);
                hipLaunchKernelGGL(rs_probe_compare_kernel, dim3(sgn_cdiv(n, 256)), dim3(256), 0, s, n, ka, va, kb, vb,
                                   mismatches);
            }
        }
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT size_t sgn_sort_workspace_bytes(int64_t n_isect) { return sort_ws_bytes<uint64_t, true, 8>(n_isect); }

SGN_EXPORT int sgn_sort_pairs(int64_t n_isect, int begin_bit, int end_bit, const int64_t *keys_in,
                              const int32_t *vals_in, int64_t *keys_out, int32_t *vals_out, void *ws,
                              size_t ws_bytes, int sort_rank_mode, sgn_stream_t stream) {
    SGN_ARG_CHECK(n_isect >= 0 && n_isect < ((int64_t)1 << 31), -1);
    SGN_ARG_CHECK(begin_bit >= 0 && end_bit <= 64 && begin_bit < end_bit, -2);
    if (n_isect == 0) return 0;
    SGN_ARG_CHECK(keys_in && vals_in && keys_out && vals_out && ws, -3);
    SGN_ARG_CHECK(ws_bytes >= sgn_sort_workspace_bytes(n_isect), -4);
    SGN_ARG_CHECK((const void *)keys_in != (const void *)keys_out && vals_in != vals_out, -5);
    hipStream_t s = (hipStream_t)stream;
    sgn_timing_begin(SGN_T_SORT, s);
    sort_launch<uint64_t, true, 8>((uint32_t)n_isect, begin_bit, end_bit, (const uint64_t *)keys_in, vals_in,
                                   (uint64_t *)keys_out, vals_out, ws, s, nullptr, sort_rank_mode);
    sgn_timing_end(SGN_T_SORT, s);
    SGN_LAUNCH_CHECK();
    return 0;
}
