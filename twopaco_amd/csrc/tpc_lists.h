// tpc_lists.h -- set-bit lists of filter slices: the wire format of the combined multi-GPU exchange (tpc_combine.hip) and the
// device code that ORs such lists into a slice held in LDS (k_slice_combine, k_apply_lookup, k_apply_lookup6).
//
// A filter slice of 2^slice_bits bits (at most 2^20) is cut into WINDOWS of 2^16 bits.  The set bits of one window travel as
// ascending 16-bit offsets, every window's list starting on a 16-byte unit (8 entries); a directory entry per (slice, window) holds
//     unit << 24 | n      unit = first 16-byte unit of the list inside its block of the payload, n = number of entries (<= 65536)
// 2 bytes per distinct set bit on the wire instead of the 4 bytes per INSERT ADDRESS (duplicates included) and 8 per query probe of
// the entry-routing exchange (tpc_shard_*).  A block is what one rank produced for one destination (tpc_combine_export: the slices
// of the level-1 buckets that destination owns, in the order [local bucket][b2]; tpc_combine_merge: the slices this rank owns).
#pragma once
#include "tpc_internal.h"  // TpcListSrc, TPC_LIST_WINDOW_BITS
#include "tpc_bins.h"

#ifdef __HIPCC__
// ORs the lists of the sources s_first, s_first + s_stride, ... (s_count of them) for directory key `key` into `slice` (LDS, zeroed or loaded by the caller, all THREADS
// threads of the workgroup call this; no barrier inside).  The THREADS / n_windows threads of a window stream that window's lists,
// 16 bytes = 8 entries per lane and load; the directory entries and first loads of up to four sources are issued before any is used.
template <int THREADS>
__device__ __forceinline__ void tpc_lists_or(const TpcListSrc &ls, uint32_t s_first, uint32_t s_count, uint32_t s_stride, uint64_t key, uint32_t *slice, int slice_bits)
{
    const uint32_t n_win = tpc_list_windows(slice_bits);
    const uint32_t tpw = (uint32_t)THREADS / n_win;  // threads per window: a multiple of 64 (THREADS = 1024, n_win <= 16)
    const uint32_t w = threadIdx.x / tpw, tl = threadIdx.x % tpw;
    const uint32_t wbase = w << (TPC_LIST_WINDOW_BITS - 5);
    constexpr int G = 4;
    for (uint32_t i0 = 0; i0 < s_count; i0 += G) {
        uint64_t d[G];
#pragma unroll
        for (int u = 0; u < G; u++) d[u] = i0 + u < s_count ? ls.dir[(uint64_t)(s_first + (i0 + u) * s_stride) * ls.dir_stride + key * n_win + w] : 0ull;
        const uint4 *src[G];
        uint4 x[G];
        uint32_t n[G];
#pragma unroll
        for (int u = 0; u < G; u++) {
            n[u] = (uint32_t)d[u] & 0xFFFFFFu;
            src[u] = reinterpret_cast<const uint4 *>(ls.payload) + (n[u] ? ls.base[s_first + (i0 + u) * s_stride] + (d[u] >> 24) : 0ull);
            x[u] = make_uint4(0, 0, 0, 0);
            if (tl * 8u < n[u]) x[u] = src[u][tl];
        }
#pragma unroll
        for (int u = 0; u < G; u++) {
            for (uint32_t i = tl; i * 8u < n[u]; i += tpw) {
                const uint4 q = i == tl ? x[u] : src[u][i];
                const uint32_t e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t v = (e[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                    if (i * 8u + (uint32_t)j < n[u]) atomicOr(&slice[wbase + (v >> 5)], 1u << (v & 31u));
                }
            }
        }
    }
}

// The sources a slice reads and its directory key (TpcListSrc::n_owner): (permuted) level-1 bucket b1, second-level index b2, index of
// the slice in the calling grid.
template <int THREADS>
__device__ __forceinline__ void tpc_lists_apply(const TpcListSrc &ls, uint32_t b1, uint32_t b2, int log_nb2, uint32_t grid_index, uint32_t *slice, int slice_bits)
{
    if (ls.n_owner) tpc_lists_or<THREADS>(ls, b1 & (ls.n_owner - 1u), ls.n_src / ls.n_owner, ls.n_owner, ((uint64_t)(b1 / ls.n_owner) << log_nb2) | b2, slice, slice_bits);
    else tpc_lists_or<THREADS>(ls, 0u, ls.n_src, 1u, (uint64_t)grid_index, slice, slice_bits);
}
#endif
