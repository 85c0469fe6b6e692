// tpc_binsp.h -- LDS write-combining bins for entries that are not a power of two bytes wide (round 5: the "entry diet").
//
// Why.  Every address of the partitioned passes crosses HBM four times (written at level 1, read + written at level 2, read where
// it is applied), and the level-2 / lookup kernels already stream at what HBM gives their access pattern: only fewer bytes per entry
// make them faster.  A level-2 query entry needs 48 bits (slice offset, edge, the low bits of the position: tpc_qpartition.hip), a
// level-2 insert entry 20; they travelled as 64 and 32.  Entries of 6 or 3 bytes do not tile a 128-byte line, and rings whose groups
// are not a power of two were the reason this was sized twice and not built.  The way out is a BLOCKED line: 128 bytes = 4 blocks of
// 32 bytes, and a block holds whole entries as two small arrays,
//
//     PFmt6  block = 5 entries of 48 bits:   5 x uint32 (low halves) | 5 x uint16 (high parts) | 2 bytes unused      20 per line
//     PFmt3  block = 10 entries of 24 bits: 10 x uint16              | 10 x uint8              | 2 bytes unused      40 per line
//
// so that (1) a ring group in LDS IS the image of the line -- the copy-out stays ds_read_b128 -> global_store_dwordx4, 16 bytes per
// lane, exactly as in Bins3; (2) a push is two naturally aligned LDS stores (b32 + b16, or b16 + b8) -- what one ds_write_b64 costs the
// banks; (3) a reader lane takes one block with two 16-byte loads and holds 5 (10) complete entries: no lane idles, nothing crosses
// lanes, and entry number 5 L + i sits in lane L.  6.4 instead of 8 bytes per query entry, 3.2 instead of 4 per insert entry.
// (Round 5 first built the fully planar line -- 21 x uint32 then 21 x uint16 -- read with a dword and a ushort load per entry-lane:
// the bytes fell by 24 %, but two partial-line loads per 63 entries left a quarter of the bytes in flight and the fused lookup went
// from 5.6 to 9.5 ms; profiles/r05a_*.  The block form gives up 1 entry per line for full-width loads.)
//
// Structure = Bins3 (tpc_bins3.h): lane l of wave w owns ring group 64 w + l for the whole kernel; a flush is two LDS-only barriers
// (every push is in its ring / every snapshot is taken) after which each wave copies its complete groups out on its own.  What is
// different:
//   * {tail, limit} of a bin are ONE 64-bit word claimed with a single returning ds_add_u64 (tail in the low half): a push no longer
//     reads the limit separately.
//   * CAP = groups per bin x GROUP is not a power of two.  Slot numbers are kept below 2 CAP -- the flush, between its two barriers
//     where no push is in flight, subtracts CAP from tail and limit whenever the head is about to pass CAP -- so "slot mod CAP" is
//     one compare-and-subtract (v_sub, v_min_u32) and "/ GROUP" one 24-bit multiply and a shift.
//   * counts are EXACT (the last line of a region is partly garbage and the readers stop at the count): no sentinel entries, all 48 /
//     24 bits of an entry are the caller's.
//   * bins must not span waves (at least 16 bins): the callers fall back to the power-of-two formats below that.
#pragma once
#include "tpc_bins3.h"

struct PFmt6 {
    using T = uint64_t;  // low 48 bits used
    static constexpr int GROUP = 20, PER_BLOCK = 5;
    static constexpr uint32_t DIV_SH = 16;  // r / 20 == (r * 3277) >> 16 for r < 2600
    __device__ __forceinline__ static uint32_t blk(uint32_t e) { return (e * 13u) >> 6; }  // e / 5, e < 20
    __device__ __forceinline__ static void store(unsigned char *grp, uint32_t e, T v)
    {
        const uint32_t b = blk(e);
        *reinterpret_cast<uint32_t *>(grp + 4u * e + 12u * b) = (uint32_t)v;
        *reinterpret_cast<uint16_t *>(grp + 20u + 2u * e + 22u * b) = (uint16_t)(v >> 32);
    }
    template <class P>
    __device__ __forceinline__ static T load(P grp, uint32_t e)
    {
        const uint32_t b = blk(e);
        return (T) * reinterpret_cast<const uint32_t *>(grp + 4u * e + 12u * b) | ((T) * reinterpret_cast<const uint16_t *>(grp + 20u + 2u * e + 22u * b) << 32);
    }
    // entry i (0..4) of the block a lane holds as two uint4
    __device__ __forceinline__ static T get(const uint4 &a, const uint4 &b, int i)
    {
        const uint32_t lo = i == 0 ? a.x : i == 1 ? a.y : i == 2 ? a.z : i == 3 ? a.w : b.x;
        const uint32_t hw = i < 2 ? b.y : i < 4 ? b.z : b.w;
        return (T)lo | ((T)((i & 1) ? hw >> 16 : hw & 0xFFFFu) << 32);
    }
};

struct PFmt3 {
    using T = uint32_t;  // low 24 bits used
    static constexpr int GROUP = 40, PER_BLOCK = 10;
    static constexpr uint32_t DIV_SH = 17;  // r / 40 == (r * 3277) >> 17 for r < 2600
    __device__ __forceinline__ static uint32_t blk(uint32_t e) { return (e * 13u) >> 7; }  // e / 10, e < 40
    __device__ __forceinline__ static void store(unsigned char *grp, uint32_t e, T v)
    {
        const uint32_t b = blk(e);
        *reinterpret_cast<uint16_t *>(grp + 2u * e + 12u * b) = (uint16_t)v;
        *reinterpret_cast<uint8_t *>(grp + 20u + e + 22u * b) = (uint8_t)(v >> 16);
    }
    template <class P>
    __device__ __forceinline__ static T load(P grp, uint32_t e)
    {
        const uint32_t b = blk(e);
        return (T) * reinterpret_cast<const uint16_t *>(grp + 2u * e + 12u * b) | ((T) * reinterpret_cast<const uint8_t *>(grp + 20u + e + 22u * b) << 16);
    }
    // entry i (0..9) of the block a lane holds as two uint4
    __device__ __forceinline__ static T get(const uint4 &a, const uint4 &b, int i)
    {
        const uint32_t lw = i < 2 ? a.x : i < 4 ? a.y : i < 6 ? a.z : i < 8 ? a.w : b.x;
        const uint32_t hb = i < 4 ? b.y : i < 8 ? b.z : b.w;
        return (T)((i & 1) ? lw >> 16 : lw & 0xFFFFu) | (((hb >> (8 * (i & 3))) & 0xFFu) << 16);
    }
};

// entries -> 128-byte lines (host and device)
template <class F> __host__ __device__ constexpr uint64_t pl_lines(uint64_t entries) { return (entries + F::GROUP - 1) / F::GROUP; }

// Streams a region of blocked lines through f(entry, index in the region): one 32-byte block per lane and step (two 16-byte loads),
// unpredicated, two batches of UNR steps in flight (PtStream's scheme).  n (entries, exact) must be uniform over the workgroup.
template <class F, int THREADS, int UNR>
struct PlStream {
    static constexpr uint32_t STEP = THREADS;  // blocks per step of the workgroup
    const uint4 *src4;
    uint32_t n, nb;  // entries, blocks
    uint4 a0[UNR], a1[UNR], b0[UNR], b1[UNR];
    __device__ __forceinline__ void ld(uint4 (&d0)[UNR], uint4 (&d1)[UNR], uint32_t i0)
    {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const uint32_t blk = min(i0 + u * STEP + threadIdx.x, nb - 1u);
            d0[u] = src4[2u * blk];
            d1[u] = src4[2u * blk + 1u];
        }
    }
    template <class Fn>
    __device__ __forceinline__ void use(const uint4 (&d0)[UNR], const uint4 (&d1)[UNR], uint32_t i0, Fn &f)
    {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const uint32_t blk = i0 + u * STEP + threadIdx.x;
            const uint32_t idx0 = blk * (uint32_t)F::PER_BLOCK;
            if (idx0 + (uint32_t)F::PER_BLOCK <= n) {  // a whole block (all but the region's last one): no test per entry
#pragma unroll
                for (int i = 0; i < F::PER_BLOCK; i++) f(F::get(d0[u], d1[u], i), idx0 + (uint32_t)i);
            } else if (blk < nb) {
#pragma unroll
                for (int i = 0; i < F::PER_BLOCK; i++)
                    if (idx0 + (uint32_t)i < n) f(F::get(d0[u], d1[u], i), idx0 + (uint32_t)i);
            }
        }
    }
    __device__ __forceinline__ void begin(const void *__restrict__ s, uint32_t count)
    {
        src4 = reinterpret_cast<const uint4 *>(s);
        n = count;
        nb = (count + F::PER_BLOCK - 1) / F::PER_BLOCK;
        if (nb == 0) return;
        ld(a0, a1, 0);
        if (UNR * STEP < nb) ld(b0, b1, UNR * STEP);
    }
    template <class Fn, class G2>
    __device__ __forceinline__ void finish_with(Fn f, G2 between)
    {
        if (nb == 0) return;
        constexpr uint32_t B = UNR * STEP;
        for (uint32_t i0 = 0;; i0 += 2 * B) {
            use(a0, a1, i0, f);
            if (i0 + B >= nb) break;
            if (i0 + 2 * B < nb) ld(a0, a1, i0 + 2 * B);
            use(b0, b1, i0 + B, f);
            if (i0 + 2 * B >= nb) break;
            if (i0 + 3 * B < nb) ld(b0, b1, i0 + 3 * B);
            between();
        }
    }
    template <class Fn>
    __device__ __forceinline__ void finish(Fn f) { finish_with(f, []() {}); }
};

template <class F, int THREADS = 1024>
struct BinsP {
    static_assert(THREADS == 1024, "one lane per ring group");
    using T = typename F::T;
    static constexpr int GROUP = F::GROUP;
    static constexpr int NB_MAX = 512, NB_MIN = 16;
    static constexpr int WAVES = THREADS / 64;
    static constexpr uint32_t OFF_CTL = 0, OFF_DONE = NB_MAX * 8, OFF_RINGS = OFF_DONE + 128, OFF_END = OFF_RINGS + PT_BIN_BYTES;
    static_assert(OFF_RINGS % PT_LINE == 0, "ring groups are line images");

    unsigned char *base;   // LDS: {tail, limit} per bin, done per wave, then 1024 ring groups of 128 bytes
    unsigned char *gbase;  // the global buffer all regions live in + this lane's 16-byte column of a line
    uint32_t LOG_GPB, GPB_MASK, CAP;
    uint32_t my_bin, my_slot;     // the bin and the ring group inside it that this lane owns
    uint32_t my_unit0;            // first 128-byte line of that bin's region in the global buffer
    uint32_t my_cap_groups;       // its capacity in lines
    uint32_t my_groups;           // lines the bin has flushed so far (every lane of the bin counts the same)
    uint32_t my_exact;            // entries of the region after the final flush
    uint32_t flushes;             // flushes done so far (uniform)

    static size_t lds_bytes(int) { return OFF_END; }
    __host__ __device__ static constexpr uint32_t cap_for(int log_nb) { return (1024u >> log_nb) * (uint32_t)GROUP; }

    __device__ __forceinline__ unsigned long long *ctl() const { return reinterpret_cast<unsigned long long *>(base + OFF_CTL); }
    __device__ __forceinline__ uint32_t *done() const { return reinterpret_cast<uint32_t *>(base + OFF_DONE); }
    __device__ __forceinline__ static uint32_t div_group(uint32_t r) { return __umul24(r, 3277u) >> F::DIV_SH; }

    __device__ __forceinline__ unsigned char *carve(unsigned char *p, int log_nb)
    {
        base = p;
        LOG_GPB = 10u - (uint32_t)log_nb;  // <= 6: a bin's ring groups sit in one wave
        GPB_MASK = (1u << LOG_GPB) - 1u;
        CAP = (uint32_t)GROUP << LOG_GPB;
        my_bin = threadIdx.x >> LOG_GPB;
        my_slot = threadIdx.x & GPB_MASK;
        my_unit0 = 0; my_cap_groups = 0; my_groups = 0; my_exact = 0; flushes = 0;
        return p + OFF_END;
    }

    // global: the buffer every region lives in.  region(b) -> uint2{first 128-byte line of bin b's private region (offset from
    // `global`), capacity in lines}.  Call once per kernel (all threads, before the first barrier-separated push).
    template <class Region>
    __device__ __forceinline__ void init(void *global, Region region)
    {
        gbase = reinterpret_cast<unsigned char *>(global) + (threadIdx.x & 7u) * 16u;
        for (uint32_t b = threadIdx.x; b < (uint32_t)NB_MAX; b += THREADS) ctl()[b] = (unsigned long long)CAP << 32;  // tail 0, limit CAP
        if (threadIdx.x < 16) done()[threadIdx.x] = 0;
        const uint2 r = region(my_bin);
        my_unit0 = r.x; my_cap_groups = r.y; my_groups = 0;
    }

    __device__ __forceinline__ void store_slot(uint32_t b, uint32_t slot, T v) const
    {
        const uint32_t r = min(slot, slot - CAP);  // slot < 2 CAP: slot mod CAP
        const uint32_t g = div_group(r);
        F::store(base + OFF_RINGS + (((b << LOG_GPB) + g) << 7), r - g * (uint32_t)GROUP, v);
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
                if (slot[i] < lim[i]) store_slot(b[i], slot[i], val[i]);
                else pend |= 1u << i;
            }
        }
        // A ring that looked full: its owner wave may still be copying the previous round out (it releases the space right after its
        // ring reads and waits for nobody before that: no deadlock).  Full although the owner is done with this flush: genuinely full
        // (address skew), the entry goes to lost().
        while (__ballot(pend != 0u) != 0ull) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                if ((pend >> i) & 1u) {
                    uint32_t bi = b[i], si = slot[i];
                    T vi = val[i];
                    asm volatile("" : "+v"(bi), "+v"(si), "+v"(vi));  // keep this rare path's arithmetic inside the branch
                    const bool owner_done = __hip_atomic_load(&done()[(bi << LOG_GPB) >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == flushes;  // read BEFORE the limit
                    const uint32_t l2 = __hip_atomic_load(reinterpret_cast<uint32_t *>(&ctl()[bi]) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (si < l2) { store_slot(bi, si, vi); pend &= ~(1u << i); }
                    else if (owner_done) { lost(bi, vi); pend &= ~(1u << i); }
                }
            }
            if (pend) __builtin_amdgcn_s_sleep(2);
        }
    }

    // lost_at(b, val, idx): an entry whose region is full, with the index it would have had there.
    // snap(b, n): optional hook called by the bin's first owner lane between the barriers with the number of entries the bin's region
    // has received so far (flushed lines + what waits in the ring) -- the group boundaries of the query's level 2
    template <bool FINAL, bool WANT_IDX, class LostAt, class Snap>
    __device__ __forceinline__ void flush_with(LostAt lost_at, Snap snap)
    {
        pt_barrier_lds();  // B1: every push of the round is in its ring
        const unsigned long long c = ctl()[my_bin];
        uint32_t t = (uint32_t)c;
        const uint32_t lim = (uint32_t)(c >> 32);
        const uint32_t h = lim - CAP;  // < CAP
        if (t - h > CAP) t = h + CAP;  // the ring overflowed: the slot numbers beyond CAP were handed to lost() by push_batch
        const uint32_t n = t - h;
        const uint32_t nfull = FINAL ? div_group(n + (uint32_t)GROUP - 1u) : div_group(n);
        const uint32_t adv = nfull * (uint32_t)GROUP;
        const uint32_t norm = h + adv >= CAP ? CAP : 0u;  // the head passes CAP with this flush: renumber now, while no push is in flight
        if (my_slot == 0) {
            ctl()[my_bin] = (unsigned long long)((FINAL ? h + adv : t) - norm) | ((unsigned long long)(lim - norm) << 32);
            snap(my_bin, my_groups * (uint32_t)GROUP + n);
        }
        if (FINAL) my_exact = my_groups * (uint32_t)GROUP + n;
        pt_barrier_lds();  // B2: no push of the next round before every snapshot is taken
        const uint32_t hg = div_group(h);
        const uint32_t rel = (my_slot - hg) & GPB_MASK;  // my ring group is the rel-th group after the head
        const bool ready = rel < nfull;
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, l = lane & 7u;
        unsigned char *ring0 = base + OFF_RINGS + (wave << 13);  // this wave's 64 ring groups
        const uint32_t valid = FINAL && ready ? min(n - rel * (uint32_t)GROUP, (uint32_t)GROUP) : (uint32_t)GROUP;
        const unsigned long long m = __ballot(ready);
        const uint32_t cnt = (uint32_t)__popcll(m);
        uint32_t dx, dy, dz = 0;
        {
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const uint32_t rank = ready ? below : cnt + (lane - below);
            const uint32_t over = my_groups + rel >= my_cap_groups ? 0x80000000u : 0u;  // the region is full: the group's entries go to lost()
            dx = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)(lane | (valid << 8) | over));
            dy = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)(my_unit0 + my_groups + rel));
            if constexpr (WANT_IDX) dz = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)(my_groups + rel));  // the line's number inside its region
        }
        // copy-out: 8 lanes per 128-byte group, 16 bytes per lane (the unused tail of a region's last line travels as it is)
        for (uint32_t i0 = 0; i0 < cnt; i0 += 8u) {  // uniform trip count: the crossbar reads need every SOURCE lane active
            const uint32_t i = i0 + (lane >> 3);
            uint2 it;
            it.x = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i << 2), (int)dx);
            it.y = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i << 2), (int)dy);
            uint32_t rline = 0;
            if constexpr (WANT_IDX) rline = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i << 2), (int)dz);
            if (i >= cnt) continue;
            const unsigned char *grp = ring0 + ((it.x & 63u) << 7);
            if ((int)it.x >= 0) {
                *reinterpret_cast<uint4 *>(gbase + (uint64_t)it.y * (uint64_t)PT_LINE) = *reinterpret_cast<const uint4 *>(grp + l * 16u);
            } else {  // a full region: its entries go to lost()
                uint32_t x = it.x;
                asm volatile("" : "+v"(x));
                const uint32_t nv = (x >> 8) & 63u, owner_lane = x & 63u;
                const uint32_t b = ((wave << 6) + owner_lane) >> LOG_GPB;
                if constexpr (pt_bulk<LostAt>::value) {  // one reservation per group (tpc_bins3.h:pt_bulk; the 8 lanes of a group are here together)
                    unsigned long long place = 0;
                    if (l == 0) place = lost_at.reserve(nv);
                    place = __shfl(place, (int)(lane & ~7u), 64);
                    for (uint32_t e = l; e < nv; e += 8u) lost_at.put(b, F::load(grp, e), place + e);
                } else
                for (uint32_t e = l; e < nv; e += 8u) lost_at(b, F::load(grp, e), rline * (uint32_t)GROUP + e);
            }
        }
        // release the ring space: only after this wave's ring reads have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        flushes++;
        if (my_slot == 0 && adv) atomicAdd(&ctl()[my_bin], (unsigned long long)adv << 32);
        my_groups += nfull;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&done()[wave], flushes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    template <bool FINAL, class Lost>
    __device__ __forceinline__ void flush(Lost lost)
    {
        if constexpr (pt_bulk<Lost>::value) flush_with<FINAL, false>(lost, [](uint32_t, uint32_t) {});
        else flush_with<FINAL, false>([&](uint32_t b, T v, uint32_t) { lost(b, v); }, [](uint32_t, uint32_t) {});
    }

    // idx(b): position of bin b's EXACT entry count in `out`; call after the final flush, by all threads
    template <class Idx>
    __device__ __forceinline__ void store_counts(uint32_t *out, Idx idx)
    {
        if (my_slot == 0) out[idx(my_bin)] = min(my_exact, my_cap_groups * (uint32_t)GROUP);
    }
};
