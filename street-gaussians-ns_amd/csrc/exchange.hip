// exchange.hip — device side of the data-parallel ROW exchange (sgn_rast/dp.py), gfx950.
//
// No upstream counterpart: the reference is single-GPU (scripts/shells/train.sh:6; SURVEY.md §8e makes view-parallel
// training with one gradient exchange per step part of this path).  One view's backward leaves most gradient rows exactly
// zero, so ranks exchange only the rows a view can touch:
//   sgn_mark_walked   which Gaussians a forward pass walked (a superset of the rows its backward can touch), as a list
//   sgn_rows_pack     the message: [header | id + the listed rows of every small per-Gaussian gradient + colour gradient]
//   sgn_rows_scatter  one rank's message added into the dense sums (called in rank order: same bits on every replica)
#include "sgn_common.h"

// ------------------------------------------------------------------------------------------------- walked ids
// Which Gaussians can receive a gradient from this view?  Only those a tile walked before it saturated: entries
// [bins[t].x, kmax[t]] of the tile's depth list (kmax: the deepest list position any pixel of the tile composited, left
// behind by the forward).  On content whose tiles saturate that is under 1 % of the scene (profiles/r04_touched_fraction),
// which is what the data-parallel row exchange sends instead of dense gradients (sgn_rast/dp.py).  This kernel turns the
// forward's (ids, bins, kmax) into the LIST of distinct walked ids — known right after the forward, so its length
// reaches the host (and the other ranks) long before the backward ends and the exchange needs no host sync of its own.
// stamps[id] holds the epoch of the last step that listed the id (no clearing pass); list order is arbitrary.
namespace {
__global__ __launch_bounds__(256) void mark_walked_kernel(int n_tiles, const int32_t *__restrict__ ids,
                                                          const int2 *__restrict__ bins,
                                                          const int32_t *__restrict__ kmax, int idmask, int epoch,
                                                          int32_t *__restrict__ stamps, int32_t *__restrict__ list,
                                                          int32_t *__restrict__ count) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= n_tiles) return;
    const int2 r = bins[t];
    const int last = min(kmax[2 * t], r.y - 1);
    for (int k = r.x + lane; k <= last; k += 64) {
        const int id = ids[k] & idmask;
        // a plain look first: a walked Gaussian is walked by ~100 tiles, and all but the first of them find it stamped
        // (a stale look only sends the lane to the exchange below, which decides)
        if (stamps[id] != epoch && atomicExch(stamps + id, epoch) != epoch) list[atomicAdd(count, 1)] = id;
    }
}
}  // namespace

SGN_EXPORT int sgn_mark_walked(int n_tiles, const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                               const int32_t *tile_stats, int ids_qmask, int epoch, int32_t *stamps, int32_t *list,
                               int32_t *count, sgn_stream_t stream) {
    SGN_ARG_CHECK(n_tiles > 0 && gaussian_ids_sorted && tile_bins && tile_stats && stamps && list && count, -1);
    SGN_ARG_CHECK(epoch != 0, -2);
    hipStream_t s = (hipStream_t)stream;
    SGN_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(int32_t), s));
    const int idmask = ids_qmask ? (SGN_QMASK_MAX_IDS - 1) : -1;
    hipLaunchKernelGGL(mark_walked_kernel, dim3(sgn_cdiv(n_tiles, 4)), dim3(256), 0, s, n_tiles, gaussian_ids_sorted,
                       (const int2 *)tile_bins, tile_stats, idmask, epoch, stamps, list, count);
    SGN_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------------- row exchange
// The two data movements of the data-parallel row exchange (sgn_rast/dp.py).  pack: rows `list[0..count)` of up to
// eight per-Gaussian float tensors (widths w_j) side by side behind the id, one message row each, under a header row
// that carries three floats (the rank's camera position).  scatter: the message rows of ONE rank added into the dense
// per-tensor sums (ids are unique within a rank's message: no collisions; ranks are scattered one launch after the
// other, so every replica adds in the same order), the trailing `tail` words of each row copied into a per-rank side
// table (the colour gradient the SH rebuild reads).
namespace {
constexpr int ROWS_MAX_TENSORS = 8;
struct RowTensors {
    int n_tensors;
    int width[ROWS_MAX_TENSORS];
    const float *src[ROWS_MAX_TENSORS];
    float *dst[ROWS_MAX_TENSORS];
};

__global__ __launch_bounds__(256) void rows_pack_kernel(int count, const int32_t *__restrict__ list, RowTensors T,
                                                        int row_words, const float *__restrict__ header3,
                                                        float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) {
        out[0] = header3[0]; out[1] = header3[1]; out[2] = header3[2];
        for (int k = 3; k < row_words; ++k) out[k] = 0.f;
    }
    if (i >= count) return;
    const int id = list[i];
    float *o = out + (size_t)(1 + i) * row_words;
    o[0] = __int_as_float(id);
    int at = 1;
    for (int j = 0; j < T.n_tensors; ++j) {
        const int w = T.width[j];
        // a tensor without a gradient this step arrives as NULL (its rows are zeros): test BEFORE offsetting
        const float *base = T.src[j];
        if (base != nullptr) {
            const float *s = base + (size_t)id * w;
            for (int k = 0; k < w; ++k) o[at + k] = s[k];
        } else {
            for (int k = 0; k < w; ++k) o[at + k] = 0.f;
        }
        at += w;
    }
}

// Contract check of the row exchange: rows with a non-zero gradient word whose id the forward did NOT list (stamp !=
// epoch).  The exchange sends listed rows only, so such a row — a regulariser on per-Gaussian parameters, anything the
// caller adds to the loss besides the rendered images — would be dropped; sgn_rast/dp.py runs this on the first steps
// and falls back to the dense sequence for good when it counts one.
__global__ __launch_bounds__(256) void rows_outside_kernel(int n, RowTensors T, const int32_t *__restrict__ stamps,
                                                           int epoch, int32_t *__restrict__ count) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    bool hit = false;
    if (id < n && stamps[id] != epoch) {
        for (int j = 0; j < T.n_tensors; ++j) {
            const float *base = T.src[j];
            if (base == nullptr) continue;
            const int w = T.width[j];
            const float *s = base + (size_t)id * w;
            for (int k = 0; k < w; ++k) hit |= s[k] != 0.f;
        }
    }
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && m != 0ull) atomicAdd(count, __popcll(m));
}

__global__ __launch_bounds__(256) void rows_scatter_kernel(int count, const float *__restrict__ rows, RowTensors T,
                                                           int row_words, float scale, int tail,
                                                           float *__restrict__ tail_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float *r = rows + (size_t)(1 + i) * row_words;
    const int id = __float_as_int(r[0]);
    int at = 1;
    for (int j = 0; j < T.n_tensors; ++j) {
        const int w = T.width[j];
        float *d = T.dst[j] + (size_t)id * w;
        for (int k = 0; k < w; ++k) d[k] += r[at + k] * scale;
        at += w;
    }
    if (tail_out != nullptr)
        for (int k = 0; k < tail; ++k) tail_out[(size_t)id * tail + k] = r[at + k];
}
}  // namespace

SGN_EXPORT int sgn_rows_pack(int count, const int32_t *list, int n_tensors, const float *const *srcs_host,
                             const int32_t *widths_host, const float *header3, float *out, int row_words,
                             sgn_stream_t stream) {
    SGN_ARG_CHECK(count >= 0 && n_tensors >= 1 && n_tensors <= ROWS_MAX_TENSORS && srcs_host && widths_host && header3 &&
                      out, -1);
    SGN_ARG_CHECK(count == 0 || list != nullptr, -2);
    RowTensors T;
    T.n_tensors = n_tensors;
    int words = 1;
    for (int j = 0; j < n_tensors; ++j) {
        SGN_ARG_CHECK(widths_host[j] >= 1 && widths_host[j] <= 64, -3);
        T.width[j] = widths_host[j]; T.src[j] = srcs_host[j]; T.dst[j] = nullptr;
        words += widths_host[j];
    }
    SGN_ARG_CHECK(words == row_words && row_words >= 3, -4);
    hipLaunchKernelGGL(rows_pack_kernel, dim3(sgn_cdiv(count > 0 ? count : 1, 256)), dim3(256), 0, (hipStream_t)stream,
                       count, list, T, row_words, header3, out);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_rows_outside(int n, int n_tensors, const float *const *srcs_host, const int32_t *widths_host,
                                const int32_t *stamps, int epoch, int32_t *count, sgn_stream_t stream) {
    SGN_ARG_CHECK(n >= 0 && n_tensors >= 1 && n_tensors <= ROWS_MAX_TENSORS && srcs_host && widths_host && stamps &&
                      count && epoch != 0, -1);
    RowTensors T;
    T.n_tensors = n_tensors;
    for (int j = 0; j < n_tensors; ++j) {
        SGN_ARG_CHECK(widths_host[j] >= 1 && widths_host[j] <= 64, -3);
        T.width[j] = widths_host[j]; T.src[j] = srcs_host[j]; T.dst[j] = nullptr;
    }
    hipStream_t s = (hipStream_t)stream;
    SGN_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(int32_t), s));
    if (n == 0) return 0;
    hipLaunchKernelGGL(rows_outside_kernel, dim3(sgn_cdiv(n, 256)), dim3(256), 0, s, n, T, stamps, epoch, count);
    SGN_LAUNCH_CHECK();
    return 0;
}

SGN_EXPORT int sgn_rows_scatter(int count, const float *rows, int row_words, int n_tensors, float *const *dsts_host,
                                const int32_t *widths_host, float scale, int tail_words, float *tail_out,
                                sgn_stream_t stream) {
    SGN_ARG_CHECK(count >= 0 && n_tensors >= 1 && n_tensors <= ROWS_MAX_TENSORS && dsts_host && widths_host, -1);
    if (count == 0) return 0;
    SGN_ARG_CHECK(rows != nullptr && tail_words >= 0, -2);
    RowTensors T;
    T.n_tensors = n_tensors;
    int words = 1 + tail_words;
    for (int j = 0; j < n_tensors; ++j) {
        SGN_ARG_CHECK(widths_host[j] >= 1 && widths_host[j] <= 64 && dsts_host[j], -3);
        T.width[j] = widths_host[j]; T.dst[j] = dsts_host[j]; T.src[j] = nullptr;
        words += widths_host[j];
    }
    SGN_ARG_CHECK(words == row_words, -4);
    hipLaunchKernelGGL(rows_scatter_kernel, dim3(sgn_cdiv(count, 256)), dim3(256), 0, (hipStream_t)stream, count, rows, T,
                       row_words, scale, tail_words, tail_out);
    SGN_LAUNCH_CHECK();
    return 0;
}
