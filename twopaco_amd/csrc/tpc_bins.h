// tpc_bins.h -- LDS write-combining bins shared by the partitioned insert (uint32 entries) and the
// partitioned query (uint64 entries).  See tpc_partition.hip for the scheme.
#pragma once
#include "tpc_device.h"

constexpr int PT_THREADS = 512;          // levels 1 and 2
constexpr int PT_BIN_BYTES = 131072;     // LDS bin storage
constexpr int PT_LINE = 128;             // flush granule: one aligned 128-byte line
constexpr int PT_APPLY_THREADS = 1024;

// Slice-index permutation.  Canonical edge addresses of hash function 0 are min(H, H') of two hashes
// (reference vertexrollinghash.h:170-184), so their density over [0, 2^L) is 2(1-x): binning by the
// plain top bits would hand bucket 0 twice the average load.  The slice index s = a >> slice_bits is
// therefore multiplied by an odd constant mod 2^F (a bijection) before it is split into the two
// bucket levels; the workgroup that owns permuted slice s' touches filter slice s = s' * inv.
struct PtPerm {
    int slice_bits, F;
    uint32_t mult, inv;
    __host__ __device__ __forceinline__ uint64_t fwd(uint64_t a) const
    {   // F < 32 for every accepted geometry (three levels of at most 9 bits): one 32-bit multiply
        const uint64_t smask = ((uint64_t)1 << slice_bits) - 1;
        const uint32_t sp = ((uint32_t)(a >> slice_bits) * mult) & (uint32_t)(((uint64_t)1 << F) - 1);
        return ((uint64_t)sp << slice_bits) | (a & smask);
    }
    __host__ __device__ __forceinline__ uint64_t back(uint64_t a) const
    {
        const uint64_t smask = ((uint64_t)1 << slice_bits) - 1, fmask = ((uint64_t)1 << F) - 1;
        return ((((a >> slice_bits) * inv) & fmask) << slice_bits) | (a & smask);
    }
    __host__ __device__ __forceinline__ uint32_t slice_of(uint32_t permuted) const { return (uint32_t)(((uint64_t)permuted * inv) & (((uint64_t)1 << F) - 1)); }
};

// Filter sharding over `world` ranks by level-1 bucket: rank r owns the buckets b1 with b1 % world == r
// (balanced, because buckets are balanced by the permutation).  world == 1: the whole filter, natural layout.
struct PtShard {
    uint32_t rank, world;  // world is a power of two
    __host__ __device__ __forceinline__ uint32_t log_world() const { return 31u - (uint32_t)__builtin_clz(world); }
};

// Level-1 region index of (workgroup w, bucket b1) in the buffer the hash kernels write: destination
// major, so that the regions of the buckets owned by rank o form one contiguous block (block o of an
// equal-split all_to_all).  After the exchange the block received from rank src sits at index src, i.e.
// region (src, w) of local bucket bl = b1 / world is pt_r1_recv(...).
__host__ __device__ __forceinline__ uint64_t pt_r1_send(PtShard sh, uint32_t nb1, uint32_t nwg1, uint32_t w, uint32_t b1)
{
    const uint32_t lw = sh.log_world();
    return ((uint64_t)(b1 & (sh.world - 1)) * (nb1 >> lw) + (b1 >> lw)) * nwg1 + w;
}
__host__ __device__ __forceinline__ uint64_t pt_r1_recv(PtShard sh, uint32_t nb1, uint32_t nwg1, uint32_t src, uint32_t w, uint32_t bl)
{
    return ((uint64_t)src * (nb1 >> sh.log_world()) + bl) * nwg1 + w;
}

// Word-array position of a PERMUTED address in this rank's filter storage: the natural position when the
// filter is whole (world == 1), else the compact shard [local bucket][b2][slice]; mine = false when the
// address belongs to another rank.
__host__ __device__ __forceinline__ uint64_t pt_local_addr(const struct PtPerm &perm, PtShard sh, int log_nb2, uint64_t a_perm, bool &mine);

// Densest level-1 bucket of the function-0 addresses relative to the mean: they are the smaller of two uniform values, density
// 2 (1 - x) over the slices, and bucket b collects the slices slice_of((b << (F - b1)) | j) -- evenly mixed when there are many
// slices per bucket, visibly not when there are few (F = 6: 8 slices per bucket).  Tight level-1 regions are sized for this bucket.
inline double pt_bucket_peak(const PtPerm &pm, int F, int b1)
{
    const double S = (double)((uint64_t)1 << F);
    double peak = 0;
    for (uint32_t b = 0; b < (1u << b1); b++) {
        double d = 0;
        for (uint32_t j = 0; j < (1u << (F - b1)); j++) d += 2.0 * (1.0 - ((double)pm.slice_of((b << (F - b1)) | j) + 0.5) / S) / S;
        peak = d > peak ? d : peak;
    }
    return peak * (double)(1u << b1);
}

inline PtPerm pt_make_perm(int slice_bits, int F)
{
    PtPerm p;
    p.slice_bits = slice_bits; p.F = F;
    p.mult = 0x9E3779B1u;
    uint32_t x = p.mult;  // Newton iteration for the inverse mod 2^32 (then truncated to F bits by the masks)
    for (int i = 0; i < 5; i++) x *= 2u - p.mult * x;
    p.inv = x;
    return p;
}

__host__ __device__ __forceinline__ uint64_t pt_local_addr(const PtPerm &perm, PtShard sh, int log_nb2, uint64_t a_perm, bool &mine)
{
    if (sh.world == 1) { mine = true; return perm.back(a_perm); }
    const uint64_t smask = ((uint64_t)1 << perm.slice_bits) - 1;
    const uint32_t sp = (uint32_t)(a_perm >> perm.slice_bits);
    const uint32_t b1 = sp >> log_nb2;
    mine = (b1 & (sh.world - 1)) == sh.rank;
    const uint32_t ls = ((b1 >> sh.log_world()) << log_nb2) | (sp & ((1u << log_nb2) - 1u));
    return ((uint64_t)ls << perm.slice_bits) | (a_perm & smask);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global stores (s_waitcnt vmcnt(0)
// before s_barrier): after a flush that is 48 KB of region lines per workgroup, ~2 us of HBM write latency per round and
// almost half of a binning kernel's time (tools/bins_bench.hip phase profile).  Nothing in these kernels reads back what
// a flush stored, so the barriers around the rings wait for the LDS operations alone.
__device__ __forceinline__ void pt_barrier_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// exclusive scan over a THREADS-thread workgroup
template <int THREADS>
__device__ __forceinline__ uint32_t pt_block_excl_scan(uint32_t v, uint32_t *s_w, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_w[wv] = inc;
    pt_barrier_lds();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < THREADS / 64; i++) { const uint32_t x = s_w[i]; if (i < wv) base += x; tot += x; }
    total = tot;
    return base + inc - v;
}

// Host side of the round schedule below: LDS words for a split workgroup that streams up to ceil(nvw / wpb) regions of at
// most cap1 entries in rounds of `step` entries.  The whole schedule when it fits beside lds_base bytes, else a segment.
extern uint32_t tpc_test_sched_cap;  // tpc_partition.hip; > 0 (option "test_sched_cap"): rounds per schedule segment, to exercise the segment path on small inputs
inline void pt_schedule_dims(uint32_t nvw, uint32_t wpb, uint64_t cap1, uint32_t step, size_t lds_base, uint32_t &nreg_cap, uint32_t &sched_cap, size_t &lds)
{
    constexpr size_t LDS_MAX = 160 * 1024 - 256;  // gfx950: 160 KB per workgroup
    nreg_cap = (nvw + wpb - 1) / wpb;
    const uint64_t need = (uint64_t)nreg_cap * ((cap1 + step - 1) / step) + 1;
    const size_t fixed = lds_base + (size_t)nreg_cap * 4;
    uint64_t room = LDS_MAX > fixed + 1024 ? (LDS_MAX - fixed) / 4 : 256;
    if (tpc_test_sched_cap) room = tpc_test_sched_cap;
    sched_cap = (uint32_t)(need < room ? need : room);
    lds = fixed + (size_t)sched_cap * 4;
}

// Round schedule of a level-2 (or level-3) workgroup.  The workgroup streams `nreg` source regions with count(t) entries
// each, `step` entries per round; a round never spans two regions.  The schedule lists, for every round, the region and
// the chunk inside it: sched[r] = t | chunk << 16.  Built once with a block scan, it turns the round loop into a
// counted loop over uniform (scalar) values -- a cursor that walked the regions by loading their counts on the way kept
// the loop's control flow in vector registers, and the compiler then waited with vmcnt(0) on every prefetch.
// Rounds [skip, skip + cap) are stored (segments, for inputs whose schedule outgrows the LDS left beside the rings);
// returns the total number of rounds.  s_cnt[nreg] receives the counts.  Ends with a barrier.
template <int THREADS, class Count>
__device__ __forceinline__ uint32_t pt_build_schedule(uint32_t nreg, uint32_t step, uint32_t skip, uint32_t cap, uint32_t *s_cnt, uint32_t *s_sched,
                                                      uint32_t *s_w, Count count)
{
    uint32_t running = 0;
    for (uint32_t t0 = 0; t0 < nreg; t0 += THREADS) {
        const uint32_t t = t0 + threadIdx.x;
        uint32_t n = 0;
        if (t < nreg) { n = skip ? s_cnt[t] : count(t); s_cnt[t] = n; }
        const uint32_t rounds = (n + step - 1) / step;
        uint32_t total;
        const uint32_t off = running + pt_block_excl_scan<THREADS>(rounds, s_w, total);
        for (uint32_t c = 0; c < rounds; c++) {
            const uint32_t r = off + c;
            if (r >= skip && r - skip < cap) s_sched[r - skip] = t | (c << 16);
        }
        running += total;
        pt_barrier_lds();  // s_w is reused by the next pass; the schedule is complete after the last one
    }
    return running;
}

template <class T>
struct PtRegion { T *base; uint64_t cap; };

// Streams a region written by Bins (16-byte aligned, its count a multiple of the flush group, padded with the all-ones
// sentinel) through f(entry): 16 bytes per lane and load, unpredicated, two batches of UNR loads in flight per lane.  n must
// be uniform over the workgroup.  (One batch at a time with predicated loads left ~32 KB per CU in flight, short of the
// bandwidth-delay product; the compiler also has to wait with vmcnt(0) after predicated loads.)
template <int THREADS, int UNR, class T>
struct PtStream {
    static constexpr int EPL = 16 / (int)sizeof(T);
    static constexpr T SENT = (T)~(T)0;
    static constexpr uint32_t STEP = UNR * THREADS;
    const uint4 *src4;
    uint32_t n, n4;
    uint4 a[UNR], b[UNR];
    __device__ __forceinline__ void ld(uint4 (&d)[UNR], uint32_t i0)
    {
#pragma unroll
        for (int u = 0; u < UNR; u++) d[u] = src4[min(i0 + u * THREADS + threadIdx.x, n4 - 1u)];
    }
    template <class F>
    __device__ __forceinline__ void use(const uint4 (&d)[UNR], uint32_t i0, F &f)
    {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const uint32_t idx = i0 + u * THREADS + threadIdx.x;
            if (idx < n4) {
                union { uint4 q; T e[EPL]; } x;
                x.q = d[u];
#pragma unroll
                for (int e = 0; e < EPL; e++) if (x.e[e] != SENT && idx * EPL + e < n) f(x.e[e]);
            }
        }
    }
    // issues the loads of the first two batches; other work may follow before finish() consumes them
    __device__ __forceinline__ void begin(const T *__restrict__ src, uint32_t count)
    {
        src4 = reinterpret_cast<const uint4 *>(src);
        n = count;
        n4 = (count + EPL - 1) / EPL;
        if (n4 == 0) return;
        ld(a, 0);
        if (STEP < n4) ld(b, STEP);
    }
    template <class F>
    __device__ __forceinline__ void finish(F f)
    {
        if (n4 == 0) return;
        for (uint32_t i0 = 0;; i0 += 2 * STEP) {
            use(a, i0, f);
            if (i0 + STEP >= n4) break;
            if (i0 + 2 * STEP < n4) ld(a, i0 + 2 * STEP);
            use(b, i0 + STEP, f);
            if (i0 + 2 * STEP >= n4) break;
            if (i0 + 3 * STEP < n4) ld(b, i0 + 3 * STEP);
        }
    }
    // the same with between() after every pair of batches (every branch is uniform: whole-workgroup work such as a flush may go there)
    template <class F, class G>
    __device__ __forceinline__ void finish_with(F f, G between)
    {
        if (n4 == 0) return;
        for (uint32_t i0 = 0;; i0 += 2 * STEP) {
            use(a, i0, f);
            if (i0 + STEP >= n4) break;
            if (i0 + 2 * STEP < n4) ld(a, i0 + 2 * STEP);
            use(b, i0 + STEP, f);
            if (i0 + 2 * STEP >= n4) break;
            if (i0 + 3 * STEP < n4) ld(b, i0 + 3 * STEP);
            between();
        }
    }
};

template <int THREADS, int UNR, class T, class F, class G>
__device__ __forceinline__ void pt_stream_region_with(const T *__restrict__ src, uint32_t n, F f, G between)
{   // pt_stream_region with between() run by the whole workgroup after every 2 x STEP x EPL entries (uniform: n is)
    PtStream<THREADS, UNR, T> st;
    st.begin(src, n);
    st.finish_with(f, between);
}

template <int THREADS, int UNR, class T, class F>
__device__ __forceinline__ void pt_stream_region(const T *__restrict__ src, uint32_t n, F f)
{
    PtStream<THREADS, UNR, T> st;
    st.begin(src, n);
    st.finish(f);
}


// LDS bins: each bin is a ring of CAP entries.  Only whole groups of GROUP entries leave the
// workgroup, so every global write is a full aligned 128-byte line; the < GROUP leftovers simply stay
// in the ring.  A flush is two short data-parallel phases: one bookkeeping thread per bin builds the
// list of 128-byte groups (scan of the group counts), then GROUP lanes copy each group.
template <class T, int THREADS = PT_THREADS, int BIN_BYTES = PT_BIN_BYTES, int LINE = PT_LINE>
struct Bins {
    static constexpr int ENTRIES = BIN_BYTES / (int)sizeof(T);
    static constexpr int GROUP = LINE / (int)sizeof(T);
    static constexpr int LOG_GROUP = (sizeof(T) == 4 ? 5 : 4) - (LINE == 64 ? 1 : 0);
    static constexpr int LOG_ENTRIES = (BIN_BYTES == 131072 ? 17 : 16) - (sizeof(T) == 4 ? 2 : 3);
    static constexpr int MAX_ITEMS = ENTRIES / GROUP;
    static constexpr T SENT = (T)~(T)0;
    int NB, CAP, LOG_CAP;  // bins (<= PT_THREADS: one bookkeeping thread per bin), entries per bin
    uint32_t *tail;   // [NB + 1] entries ever pushed into the bin; [NB] = dummy bin of invalid lanes
    uint32_t *head;   // [NB + 1] entries already written to the bin's private region (multiple of GROUP)
    T *data;          // [NB * CAP] rings
    uint2 *items;     // [MAX_ITEMS] x = bin | ring index << 10 | valid << 24, y = position in the region
    uint32_t *scan;   // [32]
#ifdef TPC_PROFILE_PHASES
    unsigned long long prof[6] = {0, 0, 0, 0, 0, 0};  // thread 0: ticks in push / bookkeeping / copy, rounds
    unsigned long long t_mark = 0;
    __device__ __forceinline__ void tick(int slot) { const unsigned long long t = wall_clock64(); if (t_mark) prof[slot] += t - t_mark; t_mark = t; }
    __device__ __forceinline__ void dump(unsigned long long *out) { if (threadIdx.x == 0) for (int i = 0; i < 6; i++) atomicAdd(&out[i], prof[i]); }
#else
    __device__ __forceinline__ void tick(int) {}
    __device__ __forceinline__ void dump(unsigned long long *) {}
#endif

    static size_t lds_bytes(int log_nb) { return (size_t)BIN_BYTES + ((size_t)8 << log_nb) + 16 + (size_t)MAX_ITEMS * 8 + 128 + 64; }

    // carve: data first (16-byte aligned), then the bookkeeping arrays; returns the first free byte
    __device__ __forceinline__ unsigned char *carve(unsigned char *p, int log_nb)
    {
        NB = 1 << log_nb;
        LOG_CAP = LOG_ENTRIES - log_nb;
        CAP = 1 << LOG_CAP;
        data = reinterpret_cast<T *>(p);
        items = reinterpret_cast<uint2 *>(p + BIN_BYTES);
        tail = reinterpret_cast<uint32_t *>(items + MAX_ITEMS);
        head = tail + NB + 2;
        scan = head + NB + 2;
        return reinterpret_cast<unsigned char *>(scan + 32);
    }

    __device__ __forceinline__ void init()
    {
        for (int b = threadIdx.x; b <= NB; b += THREADS) { tail[b] = 0; head[b] = 0; }
    }

    // N entries per lane at once: all ring slots are claimed (N independent LDS atomics in flight)
    // before any entry is stored.  ok[i] == false lanes are masked off.  lost(b, val) receives
    // the entries that found their ring full.
    template <int N, class Lost>
    __device__ __forceinline__ void push_batch(const uint32_t (&b)[N], const T (&val)[N], const bool (&ok)[N], Lost lost)
    {
        uint32_t slot[N], hd[N];
#pragma unroll
        for (int i = 0; i < N; i++) {
            slot[i] = 0; hd[i] = 0;
            if (ok[i]) {  // predicated, not a dummy bin: a shared dummy counter is a 64-way same-address LDS atomic
                slot[i] = atomicAdd(&tail[b[i]], 1u);
                hd[i] = head[b[i]];
            }
        }
#pragma unroll
        for (int i = 0; i < N; i++) {
            if (ok[i]) {
                if (slot[i] - hd[i] < (uint32_t)CAP) data[(b[i] << LOG_CAP) + (slot[i] & (uint32_t)(CAP - 1))] = val[i];
                else lost(b[i], val[i]);
            }
        }
    }

    // reg(b): this workgroup's private output area of bin b as PtRegion{base, cap}.
    // lost(b, val): called for an entry that does not fit its region (goes to an overflow list).
    template <class Reg, class Lost>
    __device__ __forceinline__ void flush(bool final, Reg reg, Lost lost)
    {
        pt_barrier_lds();
        tick(0);  // everything since the previous flush: loads + hashing + pushes
        const uint32_t tid = threadIdx.x;
        uint32_t n = 0, f = 0, h = 0;
        if (tid < (uint32_t)NB) {
            h = head[tid];
            n = min(tail[tid] - h, (uint32_t)CAP);  // entries beyond CAP were handed to lost() by push_batch
            f = final ? ((n + GROUP - 1u) & ~(uint32_t)(GROUP - 1)) : (n & ~(uint32_t)(GROUP - 1));
        }
        uint32_t total;  // (one same-address LDS atomic per bin instead of this scan serialises: 3x slower bookkeeping)
        const uint32_t off = pt_block_excl_scan<THREADS>(f >> LOG_GROUP, scan, total);
        for (uint32_t g = 0; g < (f >> LOG_GROUP); g++) {
            const uint32_t left = n - g * GROUP;
            items[off + g] = make_uint2(tid | (((h + g * GROUP) & (uint32_t)(CAP - 1)) << 10) | (min(left, (uint32_t)GROUP) << 24), h + g * GROUP);
        }
        if (tid < (uint32_t)NB) { head[tid] = h + f; tail[tid] = h + (final ? f : n); }
        pt_barrier_lds();
        tick(1);
        // copy: 8 lanes per 128-byte group, 16 bytes per lane (ds_read_b128 -> global_store_dwordx4).  Narrow stores are
        // issue bound: with one entry per lane this phase was 47 % of a binning kernel (tools/bins_bench.hip).
        constexpr int EPL = 16 / (int)sizeof(T);   // entries per lane
        constexpr int LPG = GROUP / EPL;           // lanes per group = 8
        const uint32_t l = tid & (LPG - 1);
        // four groups per lane in flight: all item reads, then all ring reads, then all stores (a dependent chain per group,
        // taken one group at a time, left the LDS latency of every step exposed)
        constexpr int UNR = 4;
        for (uint32_t w0 = tid / LPG; w0 < total; w0 += UNR * (THREADS / LPG)) {
            uint2 it[UNR];
            union { uint4 q; T e[EPL]; } u[UNR];
#pragma unroll
            for (int k = 0; k < UNR; k++) {
                const uint32_t w = w0 + k * (THREADS / LPG);
                it[k] = w < total ? items[w] : make_uint2(0u, 0u);
            }
#pragma unroll
            for (int k = 0; k < UNR; k++) {
                const uint32_t b = it[k].x & 1023u, idx0 = (it[k].x >> 10) & 16383u;
                u[k].q = *reinterpret_cast<const uint4 *>(&data[(b << LOG_CAP) + idx0 + l * EPL]);
            }
#pragma unroll
            for (int k = 0; k < UNR; k++) {
                const uint32_t w = w0 + k * (THREADS / LPG);
                if (w >= total) continue;
                const uint32_t b = it[k].x & 1023u, valid = it[k].x >> 24;
                if (valid < (uint32_t)GROUP) {  // the final flush pads a bin's last group
#pragma unroll
                    for (int e = 0; e < EPL; e++) if (l * EPL + e >= valid) u[k].e[e] = SENT;
                }
                const uint64_t pos = (uint64_t)it[k].y + l * EPL;
                const PtRegion<T> rg = reg(b);
                if (pos < rg.cap) *reinterpret_cast<uint4 *>(rg.base + pos) = u[k].q;
                else {
#pragma unroll
                    for (int e = 0; e < EPL; e++) if (u[k].e[e] != SENT) lost(b, u[k].e[e]);
                }
            }
        }
        pt_barrier_lds();
        tick(2);
#ifdef TPC_PROFILE_PHASES
        prof[4]++;
#endif
    }

    // idx(b): position of bin b's count in `out`
    template <class Reg, class Idx>
    __device__ __forceinline__ void store_counts(uint32_t *out, Reg reg, Idx idx)
    {
        for (int b = threadIdx.x; b < NB; b += THREADS) out[idx((uint32_t)b)] = (uint32_t)min((uint64_t)head[b], reg((uint32_t)b).cap);
    }
};
