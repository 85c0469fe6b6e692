// tpc_bins3.h -- LDS write-combining bins whose flush is done by every wave on its own.
//
// Why.  The SQ counters of the binning kernels (profiles/r03a_sq.csv) show SIMDs that are 60-77 % busy issuing VALU
// instructions and waves that spend the rest parked at barriers: a flush of tpc_bins.h:Bins is four workgroup barriers
// around phases in which all 16 waves do the same thing (push -> scan -> item list -> copy), so hashing arithmetic, LDS
// atomics and the region stores never overlap, and ~330 of the ~660 VALU instructions per wave and round are bookkeeping.
//
// Here the 1024 ring groups (128 KiB of rings / 128-byte groups) are owned one per LANE: lane l of wave w owns ring group
// w * 64 + l for the whole kernel, hence a fixed bin (ring groups of a bin are consecutive) and a fixed LDS address.  A flush:
//   B1  barrier: every push of the round is in its ring
//       each lane reads the tail of its bin (a snapshot: no push is in flight)
//   B2  barrier: nobody pushes before every snapshot is taken
//   then, per wave and with no further workgroup synchronisation:
//       every lane decides from (head, snapshot) whether its ring group is complete; ballot + mbcnt compact the complete
//       groups into the wave's own item list; 8 lanes per group copy them out (16 bytes per lane); the bin's limit
//       (head + CAP) moves forward once the ring reads have landed; the wave goes on to hash and push the next round.
// Between B2 and the next B1 the waves drift apart: one still copies while another hashes and a third pushes, so the
// three resources overlap.  A push that finds its ring full while the ring's owner has not finished its copy yet waits
// for it (the owner never waits for anybody between B2 and B1, so this cannot deadlock); a ring that is full although its
// owner is done is genuinely full (address skew) and the entry goes to lost(), as before.
// Semantics are those of Bins: only whole GROUP-entry groups leave the workgroup, leftovers stay in the ring, the final
// flush pads every bin's last group with the all-ones sentinel, full regions hand their entries to lost().
#pragma once
#include "tpc_bins.h"
#include <type_traits>
#include <utility>

// A `lost` handler may offer reserve(n) -> first of n consecutive places and put(bin, value, place): the flush then hands a whole ring
// group that found its region full to the overflow list with ONE global atomic instead of one per entry (round 5: a slice that holds
// the k-mers of a repeat family fills its region; 1.9 M + 3.2 M entries per step of the 62-genome text with repeat families went
// through a single cursor, ~12 ns each).  Plain callables (bin, value) keep working.
template <class L, class = void> struct pt_bulk : std::false_type {};
template <class L> struct pt_bulk<L, std::void_t<decltype(std::declval<L &>().reserve(0u))>> : std::true_type {};

template <class T, int THREADS = 1024, int DEBUG = 0>
struct Bins3 {
    static_assert(THREADS == 1024, "one lane per ring group");
    static constexpr int LOG_T = sizeof(T) == 4 ? 2 : 3;
    static constexpr int GROUP = PT_LINE / (int)sizeof(T);
    static constexpr int LOG_GROUP = sizeof(T) == 4 ? 5 : 4;
    static constexpr int LOG_ENTRIES = 17 - LOG_T;  // entries in PT_BIN_BYTES = 128 KiB of rings
    static constexpr int NB_MAX = 512;
    static constexpr int WAVES = THREADS / 64;
    static constexpr T SENT = (T)~(T)0;
    // The control words sit BELOW the rings: with the workgroup's dynamic LDS starting at address 0 their offsets (and the rings'
    // base, 4224) fit the 16-bit immediate of the DS instructions, so a push forms no address beyond `bin << 2` (round 3 had
    // the rings first: every tail / limit access paid a v_add of a 131072+ base that no immediate can hold).
    // (Round 5: {tail, limit} of a bin are ONE 64-bit word, tail in the low half, claimed with a single returning ds_add_u64 -- a push
    //  no longer reads the limit separately; profiles/r02_lds_bench.txt priced the pair at 8.17 + 10.5 lane-clocks against 5.35.  Whoever
    //  changes a half while pushes may be in flight must do it with a 64-bit atomic too: the claim writes both halves back.)
    static constexpr uint32_t OFF_CTL = 0, OFF_DONE = OFF_CTL + NB_MAX * 8, OFF_CNTDOWN = OFF_DONE + 64, OFF_RINGS = OFF_CNTDOWN + 64,
                              OFF_END = OFF_RINGS + PT_BIN_BYTES;
    static_assert(PT_BIN_BYTES == 131072 && PT_LINE == 128 && OFF_RINGS % PT_LINE == 0, "layout constants");

    unsigned char *base;   // LDS: tail, limit, done, cntdown, then the rings
    unsigned char *gbase;  // the global buffer all regions live in + this lane's 16-byte column of a line
    uint32_t LOG_NB, LOG_CAP, CAP, LOG_GPB;  // GPB = ring groups per bin
    uint32_t my_bin, my_slot;     // the bin and the ring-group slot inside it that this lane owns
    uint32_t my_unit0, my_cap;    // first 128-byte unit of that bin's region in the global buffer; its capacity in entries
    uint32_t flushes;             // flushes done so far (uniform)
#ifdef TPC_BINS3_DEBUG
    unsigned long long *dbg = nullptr;  // [0] ring-full entries lost, [1] region-full entries lost, [2] retry-loop iterations, [3] entries that waited and were stored
#endif

    static size_t lds_bytes(int) { return OFF_END; }

    __device__ __forceinline__ unsigned long long *ctl() const { return reinterpret_cast<unsigned long long *>(base + OFF_CTL); }
    __device__ __forceinline__ uint32_t *done() const { return reinterpret_cast<uint32_t *>(base + OFF_DONE); }
    __device__ __forceinline__ uint32_t *cntdown() const { return reinterpret_cast<uint32_t *>(base + OFF_CNTDOWN); }  // bins that span several waves: owners still copying
    __device__ __forceinline__ bool multi() const { return LOG_GPB > 6u; }  // a bin's ring groups span more than one wave

    __device__ __forceinline__ unsigned char *carve(unsigned char *p, int log_nb)
    {
        base = p;
        LOG_NB = (uint32_t)log_nb;
        LOG_CAP = (uint32_t)(LOG_ENTRIES - log_nb);
        CAP = 1u << LOG_CAP;
        LOG_GPB = LOG_CAP - LOG_GROUP;
        my_bin = threadIdx.x >> LOG_GPB;
        my_slot = threadIdx.x & ((1u << LOG_GPB) - 1u);
        my_unit0 = 0; my_cap = 0; flushes = 0;
        return p + OFF_END;
    }

    // global: the buffer every region lives in.  region(b) -> uint2{first 128-byte unit of bin b's private region (offset from
    // `global` in GROUP-entry units), capacity in entries}.
    template <class Region>
    __device__ __forceinline__ void init(T *global, Region region)
    {
        gbase = reinterpret_cast<unsigned char *>(global) + (threadIdx.x & 7u) * 16u;
        const uint2 r = region(my_bin);
        my_unit0 = r.x; my_cap = r.y;
        for (uint32_t b = threadIdx.x; b < (uint32_t)NB_MAX; b += THREADS) ctl()[b] = (unsigned long long)CAP << 32;  // tail 0, limit CAP
        if (threadIdx.x < 16) done()[threadIdx.x] = 0;
    }

    // N entries per lane at once: all ring slots are claimed (N independent LDS atomics in flight) before any entry is
    // stored.  lost(b, val) receives the entries that found their ring genuinely full.
    template <int N, class Lost>
    __device__ __forceinline__ void push_batch(const uint32_t (&b)[N], const T (&val)[N], const bool (&ok)[N], Lost lost)
    {
        uint32_t slot[N], lim[N];
#pragma unroll
        for (int i = 0; i < N; i++) {
            slot[i] = 0; lim[i] = 0;
            if (ok[i]) {
                const unsigned long long c = atomicAdd(&ctl()[b[i]], 1ull);
                slot[i] = (uint32_t)c; lim[i] = (uint32_t)(c >> 32);
            }
        }
        uint32_t pend = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            if (ok[i]) {
                if (slot[i] < lim[i]) *reinterpret_cast<T *>(base + OFF_RINGS + (((b[i] << LOG_CAP) | (slot[i] & (CAP - 1u))) << LOG_T)) = val[i];
                else pend |= 1u << i;
            }
        }
        // A ring that looked full: its owner may still be copying the previous round out (it releases the space right after
        // its ring reads and never waits for anybody before that, so waiting here cannot deadlock).  Full although the owner
        // is done with this flush: genuinely full (address skew), the entry goes to lost().  One loop for all N entries.
        while (__ballot(pend != 0u) != 0ull) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                if ((pend >> i) & 1u) {
                    // (the empty asm keeps this rare path's address arithmetic inside the branch: the compiler otherwise hoists it,
                    //  ~13 instructions per entry, in front of the loop, where every push pays for it)
                    uint32_t bi = b[i], si = slot[i];
                    T vi = val[i];
                    asm volatile("" : "+v"(bi), "+v"(si), "+v"(vi));
                    // the wave(s) that own the bin's ring groups: done with this flush?  (read BEFORE the limit: a finished owner's limit is final)
                    const uint32_t w0 = (bi << LOG_GPB) >> 6, nw = multi() ? 1u << (LOG_GPB - 6u) : 1u;
                    bool owners_done = true;
                    for (uint32_t w = w0; w < w0 + nw; w++) owners_done = owners_done && __hip_atomic_load(&done()[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == flushes;
                    const uint32_t l2 = __hip_atomic_load(reinterpret_cast<uint32_t *>(&ctl()[bi]) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (si < l2) {
                        *reinterpret_cast<T *>(base + OFF_RINGS + (((bi << LOG_CAP) | (si & (CAP - 1u))) << LOG_T)) = vi;
                        pend &= ~(1u << i);
#ifdef TPC_BINS3_DEBUG
                        if (dbg) atomicAdd(dbg + 3, 1ull);
#endif
                    } else if (owners_done) {
                        lost(bi, vi);
                        pend &= ~(1u << i);
#ifdef TPC_BINS3_DEBUG
                        if (dbg) atomicAdd(dbg + 0, 1ull);
#endif
                    }
                }
            }
#ifdef TPC_BINS3_DEBUG
            if (dbg && pend) atomicAdd(dbg + 2, 1ull);
#endif
            if (pend) __builtin_amdgcn_s_sleep(2);
        }
    }

    template <bool FINAL, class Lost>
    __device__ __forceinline__ void flush(Lost lost)
    {
        pt_barrier_lds();  // B1: every push of the round is in its ring
        const unsigned long long c0 = ctl()[my_bin];
        uint32_t t = (uint32_t)c0;
        const uint32_t lim = (uint32_t)(c0 >> 32);
        const uint32_t h = lim - CAP;
        if (t - h > CAP) {  // the ring overflowed: the slot numbers beyond CAP were handed to lost() by push_batch
            t = h + CAP;
            if (my_slot == 0) *reinterpret_cast<uint32_t *>(&ctl()[my_bin]) = t;  // (no push is in flight between the two barriers)
        }
        if (multi() && my_slot == 0) cntdown()[my_bin] = 1u << (LOG_GPB - 6u);  // owner waves that have to finish their copy before the ring space is released
        pt_barrier_lds();  // B2: no push of the next round before every snapshot is taken
        const uint32_t n = t - h;
        const uint32_t nfull = FINAL ? (n + GROUP - 1u) >> LOG_GROUP : n >> LOG_GROUP;
        if (FINAL && my_slot == 0) *reinterpret_cast<uint32_t *>(&ctl()[my_bin]) = h + (nfull << LOG_GROUP);  // the padded tail (nothing is pushed any more)
        const uint32_t gmask = (1u << LOG_GPB) - 1u;
        const uint32_t rel = (my_slot - (h >> LOG_GROUP)) & gmask;  // my ring group is the rel-th group after the head
        const bool ready = rel < nfull;
        const uint32_t pos = h + (rel << LOG_GROUP);  // position of the group in the bin's region
        const unsigned long long m = __ballot(ready);
        const uint32_t cnt = (uint32_t)__popcll(m);
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, l = lane & 7u;
        // The complete groups of the wave, compacted: lane r < cnt receives the descriptor of the r-th complete ring group.  A
        // full permutation (the other lanes take the ranks cnt, cnt + 1, ...) through the LDS crossbar (ds_permute), no LDS
        // memory: the 8 KiB of per-wave item lists this replaced are what lets the hash kernels keep their seed tables in LDS.
        uint32_t dx, dy;
        {
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));  // complete groups in lower lanes
            const uint32_t rank = ready ? below : cnt + (lane - below);
            const uint32_t valid = FINAL ? min(n - (rel << LOG_GROUP), (uint32_t)GROUP) : (uint32_t)GROUP;
            const uint32_t over = pos + GROUP > my_cap ? 0x80000000u : 0u;  // the region is full: the group's entries go to lost()
            dx = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)(lane | (valid << 8) | over));
            dy = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)(my_unit0 + (pos >> LOG_GROUP)));
        }
        // copy-out: 8 lanes per 128-byte group, 16 bytes per lane
        constexpr int EPL = 16 / (int)sizeof(T);
        const unsigned char *ring = base + OFF_RINGS + (wave << 13) + l * 16u;  // this wave's 64 ring groups
        for (uint32_t i0 = 0; i0 < cnt; i0 += 8u) {  // uniform trip count: the crossbar reads need every SOURCE lane active
            const uint32_t i = i0 + (lane >> 3);
            uint2 it;
            it.x = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i << 2), (int)dx);
            it.y = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i << 2), (int)dy);
            if (DEBUG == 1) it.y &= 1023u;
            if (i >= cnt) continue;
            union { uint4 q; T e[EPL]; } u;
            u.q = *reinterpret_cast<const uint4 *>(ring + ((it.x & 63u) << 7));
            unsigned char *dst = gbase + (uint64_t)it.y * (uint64_t)PT_LINE;
            if ((it.x >> 8) == (uint32_t)GROUP) {
                *reinterpret_cast<uint4 *>(dst) = u.q;
            } else {  // a padded last group (final flush) or a full region
                uint32_t x = it.x;
                asm volatile("" : "+v"(x));  // keep the decoding below out of the common path
                const uint32_t valid = (x >> 8) & 63u, owner_lane = x & 63u;
                const bool over = (x >> 31) != 0;
                const uint32_t b = ((wave << 6) + owner_lane) >> LOG_GPB;
                unsigned long long place = 0;
                if constexpr (pt_bulk<Lost>::value) {
                    if (over) {  // (the 8 lanes of a group take this branch together)
                        if (l == 0) place = lost.reserve(valid);
                        place = __shfl(place, (int)(lane & ~7u), 64);
                    }
                }
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    if (l * EPL + e >= valid) u.e[e] = SENT;
                    else if (over) {
                        if constexpr (pt_bulk<Lost>::value) lost.put(b, u.e[e], place + l * EPL + e);
                        else lost(b, u.e[e]);
#ifdef TPC_BINS3_DEBUG
                        if (dbg) atomicAdd(dbg + 1, 1ull);
#endif
                    }
                }
                if (!over) *reinterpret_cast<uint4 *>(dst) = u.q;
            }
        }
        // release the ring space: only after this wave's ring reads have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        flushes++;
        if (!multi()) {
            if (my_slot == 0 && nfull) atomicAdd(&ctl()[my_bin], (unsigned long long)(nfull << LOG_GROUP) << 32);
        } else if (lane == 0) {  // the wave's 64 ring groups belong to one bin: the last owner wave to finish releases the space
            if (atomicSub(&cntdown()[my_bin], 1u) == 1u && nfull) atomicAdd(&ctl()[my_bin], (unsigned long long)(nfull << LOG_GROUP) << 32);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&done()[wave], flushes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    // idx(b): position of bin b's count in `out` (call after the final flush, by all threads)
    template <class Idx>
    __device__ __forceinline__ void store_counts(uint32_t *out, Idx idx)
    {
        pt_barrier_lds();
        if (my_slot == 0) {
            const uint32_t head = (uint32_t)(ctl()[my_bin] >> 32) - CAP;
            out[idx(my_bin)] = min(head, my_cap);
        }
    }
};

