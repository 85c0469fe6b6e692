// tpc_rbins.h -- barrier-free LDS write-combining bins (second generation of tpc_bins.h:Bins).
//
// Bins (tpc_bins.h) collects a round of entries, then the whole workgroup stops for a flush: four
// barriers, a scan and a copy per round, and a round is one or two positions per thread because 128 KiB
// of rings hold only 64 (uint64) or 128 (uint32) entries per bin.  The hash kernels spent most of their
// time waiting at those barriers (profiles/r02_lds_bench.txt: the LDS itself sustains 1.5-1.9 T pushes/s,
// the kernels reached 0.25 T/s).
//
// Here a bin is a ring of NG groups of GROUP entries (GROUP entries = one aligned 128-byte line of the
// bin's private region).  A lane
//   1. claims slot s = tail[bin]++          (the entry's final position in the region is s),
//   2. waits until ring group (s / GROUP) % NG has been drained generation (s / GROUP) / NG times,
//   3. stores the entry and counts itself in state[bin][ring group];
//   4. the lane whose count completes the group hands it to its own wave, which copies the line out
//      (GROUP lanes per group, 64 / GROUP groups per pass) and bumps the group's generation.
// Nothing waits on a particular wave: whoever writes the last entry of a group drains it, so the only
// stall is back-pressure when a ring is full -- the oldest incomplete group of a bin always has writers
// that are free to proceed (their slot's previous occupant is older still), hence no deadlock.
// push_batch must be called by whole waves (wave-uniform control flow); lanes without an entry pass
// ok = false.  No workgroup barrier is needed until flush_final.
//
// Where it is used: k_q_hash / k_q_split<.., RB = true> (tpc_qpartition.hip) for levels of 512 bins, where a ring of the
// barrier bins holds only 32 uint64 entries.  At 256 bins the barrier bins are as fast or faster (tools/bins_bench.hip).
#pragma once
#include "tpc_bins.h"

template <class T, int THREADS>
struct RBins {
    static constexpr int GROUP = PT_LINE / (int)sizeof(T);
    static constexpr int LOG_GROUP = sizeof(T) == 4 ? 5 : 4;
    static constexpr int GPW = 64 / GROUP;   // groups a wave copies per pass
    static constexpr int WAVES = THREADS / 64;
    static constexpr int LIST = 128;         // drain descriptors per wave
    static constexpr int LOG_LINES = 10;     // PT_BIN_BYTES / PT_LINE ring groups in all
    static constexpr T SENT = (T)~(T)0;
    int NB, LOG_NG;
    uint32_t *tail;    // [NB] entries ever claimed by the bin
    uint32_t *state;   // [NB << LOG_NG] (generation << 8) | entries stored in the ring group
    T *data;           // [NB << LOG_NG][GROUP]
    uint32_t *wlist;   // [WAVES][LIST] drain descriptors: region group << 10 | ring group index

    static size_t lds_bytes(int log_nb) { return (size_t)PT_BIN_BYTES + ((size_t)4 << log_nb) + ((size_t)4 << LOG_LINES) + (size_t)WAVES * LIST * 4 + 64; }

    __device__ __forceinline__ unsigned char *carve(unsigned char *p, int log_nb)
    {
        NB = 1 << log_nb;
        LOG_NG = LOG_LINES - log_nb;
        data = reinterpret_cast<T *>(p);
        state = reinterpret_cast<uint32_t *>(p + PT_BIN_BYTES);
        tail = state + (1 << LOG_LINES);
        wlist = tail + NB;
        return reinterpret_cast<unsigned char *>(wlist + WAVES * LIST);
    }

    __device__ __forceinline__ void init()
    {
        for (int i = threadIdx.x; i < (1 << LOG_LINES); i += THREADS) state[i] = 0;
        for (int i = threadIdx.x; i < NB; i += THREADS) tail[i] = 0;
    }

    template <class Reg>
    __device__ __forceinline__ void copy_out(uint32_t total, Reg reg)
    {
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t *wl = wlist + (threadIdx.x >> 6) * LIST;
        const uint32_t l = lane & (GROUP - 1);
        for (uint32_t p = 0; p < total; p += GPW) {
            const uint32_t gi = p + (lane >> LOG_GROUP);
            uint32_t sidx = 0;
            if (gi < total) {
                const uint32_t d = wl[gi];
                sidx = d & 1023u;
                const T v = data[(sidx << LOG_GROUP) + l];
                const PtRegion<T> rg = reg(sidx >> LOG_NG);
                rg.base[(uint64_t)(d >> 10) * GROUP + l] = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the ring group has been read: hand it to the next generation
            if (gi < total && l == 0) __hip_atomic_fetch_add(&state[sidx], 256u - GROUP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }

    // reg(b): the workgroup's private region of bin b (cap a multiple of GROUP); lost(b, val): entries beyond it
    template <int N, class Reg, class Lost>
    __device__ __forceinline__ void push_batch(const uint32_t (&b)[N], const T (&val)[N], const bool (&ok)[N], Reg reg, Lost lost)
    {
        uint32_t *wl = wlist + (threadIdx.x >> 6) * LIST;
        uint32_t slot[N];
        uint32_t pend = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            slot[i] = 0;
            if (ok[i]) { slot[i] = atomicAdd(&tail[b[i]], 1u); pend |= 1u << i; }
        }
#pragma unroll
        for (int i = 0; i < N; i++)
            if (((pend >> i) & 1u) && slot[i] >= reg(b[i]).cap) { lost(b[i], val[i]); pend &= ~(1u << i); }
        for (;;) {
            uint32_t st[N], wr = 0, done = 0;
#pragma unroll
            for (int i = 0; i < N; i++) {
                st[i] = 0;
                if ((pend >> i) & 1u)
                    st[i] = __hip_atomic_load(&state[(b[i] << LOG_NG) | ((slot[i] >> LOG_GROUP) & ((1u << LOG_NG) - 1u))], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int i = 0; i < N; i++) {
                const uint32_t g = slot[i] >> LOG_GROUP;
                if (((pend >> i) & 1u) && (st[i] >> 8) == (g >> LOG_NG)) {
                    data[((((b[i] << LOG_NG) | (g & ((1u << LOG_NG) - 1u)))) << LOG_GROUP) + (slot[i] & (GROUP - 1))] = val[i];
                    wr |= 1u << i;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // entries are in the ring before they are counted
#pragma unroll
            for (int i = 0; i < N; i++) {
                if ((wr >> i) & 1u) {
                    const uint32_t g = slot[i] >> LOG_GROUP;
                    const uint32_t old = __hip_atomic_fetch_add(&state[(b[i] << LOG_NG) | (g & ((1u << LOG_NG) - 1u))], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if ((old & 255u) == (uint32_t)GROUP - 1u) done |= 1u << i;
                }
            }
            pend &= ~wr;
            // completed groups of this wave -> descriptor list -> copied out GPW groups at a time
            uint32_t total = 0;
#pragma unroll
            for (int i = 0; i < N; i++) {
                const unsigned long long m = __ballot((done >> i) & 1u);
                if (m == 0) continue;
                const uint32_t c = (uint32_t)__popcll(m);
                if (total + c > (uint32_t)LIST) { copy_out(total, reg); total = 0; }
                if ((done >> i) & 1u) {
                    const uint32_t g = slot[i] >> LOG_GROUP;
                    const uint32_t at = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    wl[at] = (g << 10) | (b[i] << LOG_NG) | (g & ((1u << LOG_NG) - 1u));
                }
                total += c;
            }
            if (total) copy_out(total, reg);
            if (__ballot(pend != 0) == 0) break;
            __builtin_amdgcn_s_sleep(2);  // a ring is full: its older groups are still being written by other waves
        }
    }

    // interface parity with Bins: the rings drain themselves, only the last call does anything
    template <class Reg, class Lost>
    __device__ __forceinline__ void flush(bool final, Reg reg, Lost)
    {
        if (!final) return;
        __syncthreads();
        const uint32_t l = threadIdx.x & (GROUP - 1);
        for (uint32_t bb = threadIdx.x >> LOG_GROUP; bb < (uint32_t)NB; bb += THREADS / GROUP) {
            const uint32_t t = tail[bb], part = t & (GROUP - 1);
            if (!part) continue;
            const uint32_t g = t >> LOG_GROUP;
            const PtRegion<T> rg = reg(bb);
            if ((uint64_t)g * GROUP >= rg.cap) continue;  // these entries went to lost()
            const T v = l < part ? data[(((bb << LOG_NG) | (g & ((1u << LOG_NG) - 1u))) << LOG_GROUP) + l] : SENT;
            rg.base[(uint64_t)g * GROUP + l] = v;
        }
        __syncthreads();
    }

    template <class Reg, class Idx>
    __device__ __forceinline__ void store_counts(uint32_t *out, Reg reg, Idx idx)
    {
        for (int b = threadIdx.x; b < NB; b += THREADS)
            out[idx((uint32_t)b)] = (uint32_t)min((uint64_t)((tail[b] + GROUP - 1u) & ~(uint32_t)(GROUP - 1)), reg((uint32_t)b).cap);
    }
};
