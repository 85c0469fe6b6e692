// tpc_bins2.h -- LDS write-combining bins, third generation: the same rings and flush-per-round protocol as
// tpc_bins.h:Bins, rebuilt around what the SQ counters showed (profiles/r03a_sq.csv): the binning kernels were not
// waiting for LDS or HBM, their SIMDs were 60-77 % busy ISSUING VALU instructions -- ~660 per wave and round in
// k_q_hash, half of them in the flush (a 1024-thread scan, per-lane region arithmetic in 64-bit, four barriers).
//
// What changed against Bins:
//   * bookkeeping is done by the bins' own threads only (NB / 64 waves; the other waves go straight to the next
//     barrier), the exclusive scan of the group counts is a wave-level ballot / mbcnt over bit planes, and each
//     64-bin segment has its own slice of the item list, so there is no cross-wave scan and one barrier fewer;
//   * an item carries everything a copy lane needs -- LDS offset of the ring group and the 128-byte unit index of its
//     destination in the global buffer -- computed ONCE per group by the bin's thread: a copy lane does one 8-byte LDS
//     read, one 16-byte LDS read, one 64-bit multiply-add and one 16-byte store;
//   * waves are assigned to item segments statically (no division of a linear index);
//   * the per-bin limit (head + CAP) is stored instead of the head: the ring-full test of a push is one compare;
//   * the LDS layout is fixed (offsets are compile-time constants that fold into the ds instructions' offset fields).
// Semantics are unchanged: only whole GROUP-entry groups (one aligned 128-byte line) leave the workgroup, the < GROUP
// leftovers stay in the ring, the final flush pads the last group of every bin with the all-ones sentinel, entries that
// find their ring or their region full are handed to lost().
#pragma once
#include "tpc_bins.h"

template <class T, int THREADS>
struct Bins2 {
    static constexpr int LOG_T = sizeof(T) == 4 ? 2 : 3;
    static constexpr int GROUP = PT_LINE / (int)sizeof(T);
    static constexpr int LOG_GROUP = sizeof(T) == 4 ? 5 : 4;
    static constexpr int LOG_ENTRIES = 17 - LOG_T;               // entries in PT_BIN_BYTES = 128 KiB of rings
    static constexpr int MAX_ITEMS = PT_BIN_BYTES / PT_LINE;     // ring groups in all = 1024
    static constexpr int NB_MAX = 512;
    static constexpr int WAVES = THREADS / 64;
    static constexpr T SENT = (T)~(T)0;
    static constexpr uint32_t OFF_ITEMS = PT_BIN_BYTES, OFF_TAIL = OFF_ITEMS + MAX_ITEMS * 8, OFF_LIMIT = OFF_TAIL + NB_MAX * 4,
                              OFF_SEG = OFF_LIMIT + NB_MAX * 4, OFF_END = OFF_SEG + 64;
    static_assert(PT_BIN_BYTES == 131072 && PT_LINE == 128, "layout constants");

    unsigned char *base;   // LDS: rings at 0, then items, tail, limit, per-segment item counts
    unsigned char *gbase;  // global buffer the regions live in (+ this lane's 16-byte column of a line)
    uint32_t NB, LOG_NB, LOG_CAP, CAP;
    uint32_t log_wps;      // copy-out: waves per item segment
    uint32_t log_seg;      // items per segment
    uint32_t my_unit0, my_cap;  // bin threads: first 128-byte unit of the bin's region in the global buffer, its capacity in entries

    static size_t lds_bytes(int) { return OFF_END; }

    __device__ __forceinline__ uint32_t *tail() const { return reinterpret_cast<uint32_t *>(base + OFF_TAIL); }
    __device__ __forceinline__ uint32_t *limit() const { return reinterpret_cast<uint32_t *>(base + OFF_LIMIT); }
    __device__ __forceinline__ uint32_t *segcnt() const { return reinterpret_cast<uint32_t *>(base + OFF_SEG); }
    __device__ __forceinline__ uint2 *items() const { return reinterpret_cast<uint2 *>(base + OFF_ITEMS); }

    __device__ __forceinline__ unsigned char *carve(unsigned char *p, int log_nb)
    {
        base = p;
        LOG_NB = (uint32_t)log_nb;
        NB = 1u << log_nb;
        LOG_CAP = (uint32_t)(LOG_ENTRIES - log_nb);
        CAP = 1u << LOG_CAP;
        const uint32_t log_nseg = log_nb > 6 ? (uint32_t)log_nb - 6u : 0u;  // 64 bins (one wave of bookkeeping threads) per segment
        constexpr uint32_t LOG_WAVES = THREADS == 1024 ? 4 : THREADS == 512 ? 3 : 2;
        log_wps = LOG_WAVES - log_nseg;
        log_seg = 10u - log_nseg;
        my_unit0 = 0; my_cap = 0;
        return p + OFF_END;
    }

    // global: the buffer every region lives in.  unit0 / cap: for thread b < NB the first 128-byte unit (offset from
    // `global` in GROUP-entry units) and the capacity in entries of bin b's private region; ignored for other threads.
    __device__ __forceinline__ void init(T *global, uint32_t unit0, uint32_t cap)
    {
        gbase = reinterpret_cast<unsigned char *>(global) + (threadIdx.x & 7u) * 16u;
        my_unit0 = unit0; my_cap = cap;
        for (uint32_t b = threadIdx.x; b < (uint32_t)NB_MAX; b += THREADS) { tail()[b] = 0; limit()[b] = CAP; }
        if (threadIdx.x < 16) segcnt()[threadIdx.x] = 0;
    }

    // N entries per lane at once: all ring slots are claimed (N independent LDS atomics in flight) before any entry is
    // stored.  lost(b, val) receives the entries that found their ring full.
    template <int N, class Lost>
    __device__ __forceinline__ void push_batch(const uint32_t (&b)[N], const T (&val)[N], const bool (&ok)[N], Lost lost)
    {
        uint32_t slot[N], lim[N];
#pragma unroll
        for (int i = 0; i < N; i++) {
            slot[i] = 0; lim[i] = 0;
            if (ok[i]) {
                slot[i] = atomicAdd(&tail()[b[i]], 1u);
                lim[i] = limit()[b[i]];
            }
        }
#pragma unroll
        for (int i = 0; i < N; i++) {
            if (ok[i]) {
                if (slot[i] < lim[i]) *reinterpret_cast<T *>(base + (((b[i] << LOG_CAP) | (slot[i] & (CAP - 1u))) << LOG_T)) = val[i];
                else lost(b[i], val[i]);
            }
        }
    }

    template <bool FINAL, class Lost>
    __device__ __forceinline__ void flush(Lost lost)
    {
        pt_barrier_lds();  // every push of the round is in its ring
        const uint32_t tid = threadIdx.x;
        if (tid < NB) {  // whole waves when NB >= 64
            const uint32_t lim = limit()[tid], h = lim - CAP, t = tail()[tid];
            const uint32_t n = min(t - h, CAP);  // entries beyond CAP were handed to lost() by push_batch
            const uint32_t f = FINAL ? ((n + GROUP - 1u) & ~(uint32_t)(GROUP - 1)) : (n & ~(uint32_t)(GROUP - 1));
            const uint32_t ng = f >> LOG_GROUP;
            // exclusive scan of ng over the wave's 64 bins, one ballot per bit plane (ng <= CAP / GROUP)
            uint32_t excl = 0, total = 0;
            const uint32_t planes = LOG_CAP - LOG_GROUP + 1u;
            for (uint32_t k = 0; k < planes; k++) {
                const unsigned long long m = __ballot((ng >> k) & 1u);
                excl += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) << k;
                total += (uint32_t)__popcll(m) << k;
            }
            const uint32_t seg = tid >> 6;
            uint2 *it = items() + (seg << log_seg) + excl;
            for (uint32_t g = 0; g < ng; g++) {
                const uint32_t pos = h + g * GROUP;  // position of the group in the bin's region
                const uint32_t valid = FINAL ? min(n - g * GROUP, (uint32_t)GROUP) : (uint32_t)GROUP;
                const uint32_t over = pos + GROUP > my_cap ? 0x80000000u : 0u;  // the region is full: the group's entries go to lost()
                const uint32_t off16 = (((tid << LOG_CAP) | (pos & (CAP - 1u))) << LOG_T) >> 4;
                it[g] = make_uint2(off16 | (tid << 13) | (valid << 22) | over, my_unit0 + (pos >> LOG_GROUP));
            }
            limit()[tid] = h + f + CAP;
            tail()[tid] = h + (FINAL ? f : n);
            if ((tid & 63u) == 0) segcnt()[seg] = total;
        }
        pt_barrier_lds();
        // copy-out: 8 lanes per 128-byte group, 16 bytes per lane; wave w serves segment w >> log_wps
        constexpr int EPL = 16 / (int)sizeof(T);
        const uint32_t wave = tid >> 6, lane = tid & 63u, l = lane & 7u;
        const uint32_t seg = wave >> log_wps, sub = wave & ((1u << log_wps) - 1u);
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)segcnt()[seg]);
        const uint2 *its = items() + (seg << log_seg);
        for (uint32_t i = sub * 8u + (lane >> 3); i < cnt; i += 8u << log_wps) {
            const uint2 it = its[i];
            union { uint4 q; T e[EPL]; } u;
            u.q = *reinterpret_cast<const uint4 *>(base + ((it.x & 0x1FFFu) << 4) + l * 16u);
            unsigned char *dst = gbase + (uint64_t)it.y * (uint64_t)PT_LINE;
            if ((it.x >> 22) == (uint32_t)GROUP) {
                *reinterpret_cast<uint4 *>(dst) = u.q;
            } else {  // a padded last group (final flush) or a full region
                const uint32_t valid = (it.x >> 22) & 63u, b = (it.x >> 13) & 511u;
                const bool over = (it.x >> 31) != 0;
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    if (l * EPL + e >= valid) u.e[e] = SENT;
                    else if (over) lost(b, u.e[e]);
                }
                if (!over) *reinterpret_cast<uint4 *>(dst) = u.q;
            }
        }
        pt_barrier_lds();  // the groups have been read: their ring space may be claimed again
    }

    // idx(b): position of bin b's count in `out` (call after the final flush, by all threads)
    template <class Idx>
    __device__ __forceinline__ void store_counts(uint32_t *out, Idx idx)
    {
        if (threadIdx.x < NB) out[idx(threadIdx.x)] = min(limit()[threadIdx.x] - CAP, my_cap);
    }
};
