// tpc_lists.h -- set-bit lists of filter slices: the wire format of the combined multi-GPU exchange (tpc_combine.hip) and the
// device code that ORs such lists into a slice held in LDS (k_slice_combine, k_apply_lookup, k_apply_lookup6).
//
// A filter slice of 2^slice_bits bits (at most 2^20) is cut into WINDOWS of 2^16 bits.  The set bits of one window travel as
// 16-bit offsets (in no particular order), every window's list starting on a 16-byte unit (8 entries); a directory entry per (slice, window) holds
//     unit << 24 | n      unit = first 16-byte unit of the list inside its block of the payload, n = number of entries (<= 65536)
// 2 bytes per distinct set bit on the wire instead of the 4 bytes per INSERT ADDRESS (duplicates included) and 8 per query probe of
// the entry-routing exchange (tpc_shard_*).  A block is what one rank produced for one destination (tpc_combine_export: the slices
// of the level-1 buckets that destination owns, in the order [local bucket][b2]; tpc_combine_merge: the slices this rank owns).
#pragma once
#include "tpc_internal.h"  // TpcListSrc, TPC_LIST_WINDOW_BITS
#include "tpc_bins.h"

#ifdef __HIPCC__
// ORs the lists of a slice into the slice held in LDS.  begin() issues the directory loads and the first payload load of up to four
// sources -- before the caller zeroes the slice, so that the two dependent round trips run under that -- finish() (after the barrier
// behind the zeroing; all THREADS threads call both, no barrier inside) ORs them in and streams what is left: longer lists, further
// sources.  The THREADS / n_windows threads of a window stream that window's lists, 16 bytes = 8 entries per lane and load.
template <int THREADS>
struct TpcListReader {
    static constexpr int G = 4;
    const uint4 *src[G];
    uint4 x[G];
    uint32_t n[G];
    uint32_t s_first, s_count, s_stride, tpw, tl, wbase, n_win, w;
    uint64_t key;
    __device__ __forceinline__ void load_group(const TpcListSrc &ls, uint32_t i0)
    {
        uint64_t d[G];
#pragma unroll
        for (int u = 0; u < G; u++) d[u] = i0 + u < s_count ? ls.dir[(uint64_t)(s_first + (i0 + u) * s_stride) * ls.dir_stride + key * n_win + w] : 0ull;
#pragma unroll
        for (int u = 0; u < G; u++) {
            n[u] = (uint32_t)d[u] & 0xFFFFFFu;
            src[u] = reinterpret_cast<const uint4 *>(ls.payload) + (n[u] ? ls.base[s_first + (i0 + u) * s_stride] + (d[u] >> 24) : 0ull);
            x[u] = make_uint4(0, 0, 0, 0);
            if (tl * 8u < n[u]) x[u] = src[u][tl];
        }
    }
    __device__ __forceinline__ void use_group(uint32_t *slice)
    {
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
    // (permuted) level-1 bucket b1, second-level index b2 of the slice, its index in the calling grid (TpcListSrc::n_owner says which
    // sources list it and under which key)
    // win0, wins: the workgroup holds the windows [win0, win0 + wins) of the slice (all of them by default), slice[] starting at window win0
    __device__ __forceinline__ void begin(const TpcListSrc &ls, uint32_t b1, uint32_t b2, int log_nb2, uint32_t grid_index, int slice_bits, uint32_t win0 = 0, uint32_t wins = 0)
    {
        n_win = tpc_list_windows(slice_bits);
        if (wins == 0) wins = n_win;
        tpw = (uint32_t)THREADS / wins;  // threads per window: a multiple of 64 (THREADS / wins = 1024 / n_win >= 64)
        const uint32_t wl = threadIdx.x / tpw;
        w = win0 + wl; tl = threadIdx.x % tpw;
        wbase = wl << (TPC_LIST_WINDOW_BITS - 5);
        if (ls.n_owner) { s_first = b1 & (ls.n_owner - 1u); s_count = ls.n_src / ls.n_owner; s_stride = ls.n_owner; key = ((uint64_t)(b1 / ls.n_owner) << log_nb2) | b2; }
        else { s_first = 0; s_count = ls.n_src; s_stride = 1; key = grid_index; }
        if (s_count) load_group(ls, 0);
    }
    __device__ __forceinline__ void finish(const TpcListSrc &ls, uint32_t *slice)
    {
        for (uint32_t i0 = 0; i0 < s_count; i0 += G) {
            if (i0) load_group(ls, i0);
            use_group(slice);
        }
    }
};
#endif
