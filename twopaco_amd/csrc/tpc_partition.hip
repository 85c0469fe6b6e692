// tpc_partition.hip -- first-pass insert behind an LDS write-combining stage.
//
// Why.  A scattered atomicOr into a multi-GiB filter is not HBM-byte bound on MI355X but
// transaction bound: device-scope atomics execute at the memory side at ~27 G/s whatever the
// footprint (17.8 G/s at 8 GiB; profiles/r01_microbench.txt), so the direct kernel (k_insert, q
// atomics per k-mer) tops out at ~3.6 G k-mers/s.  LDS atomics run at ~855 G/s.  This path
// therefore routes every Bloom address to the workgroup that owns its 2^SLICE-bit slice of the
// filter and sets the bit in LDS:
//
//   level 1  k_part_hash   rolling hash of the packed text (same tpc_insert_step as k_insert) ->
//                          each address is binned by its top B1 bits into LDS bins; full 128-byte
//                          groups of 32 entries are flushed to the workgroup's private region
//                          of bucket b1 (coalesced, aligned, no global atomics)
//   level 2  k_part_split  streams bucket b1, bins by the next B2 bits, flushes the same way
//   level 3  k_part_apply  one workgroup per slice: slice in LDS (zeroed, or loaded when the
//                          filter already holds bits), ds_or every entry, one coalesced write-out
//   level 4  k_part_ovf    entries that did not fit a region (adversarial skew) are applied with
//                          plain atomicOr; if even that list overflows the host re-runs the
//                          direct kernel -- OR is idempotent, so parity never depends on luck.
//
// The filter contents are identical to the direct path's (bitwise OR is order independent).
// HBM traffic per address: 4 B written + 4 B read per level, plus one sequential pass over the
// filter -- versus one 64-byte read-for-ownership and write-back per address for the atomics.
#include "tpc_device.h"
#include "tpc_insert_step.h"
#include "tpc_internal.h"
#include <algorithm>
#include <cmath>

namespace {

constexpr int PT_THREADS = 512;          // levels 1 and 2
constexpr int PT_BIN_ENTRIES = 32768;    // LDS bin storage: 128 KiB of uint32 entries
constexpr uint32_t PT_SENT = 0xFFFFFFFFu;
constexpr int PT_APPLY_THREADS = 1024;

struct Overflow {
    uint64_t *list;
    unsigned long long *cursor;  // [0] entries appended, [1] set when the list itself overflowed
    uint64_t cap;
    __device__ __forceinline__ void push(uint64_t a) const
    {
        const unsigned long long o = atomicAdd(cursor, 1ull);
        if (o < cap) list[o] = a; else cursor[1] = 1ull;
    }
};

// exclusive scan over the PT_THREADS-thread workgroup
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_w, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < PT_THREADS / 64; i++) { const uint32_t x = s_w[i]; if (i < wv) base += x; tot += x; }
    total = tot;
    return base + inc - v;
}

// LDS bins with carry: only whole groups of 32 entries leave the workgroup, so every global
// write is a full aligned 128-byte line; the <32 leftovers stay for the next round.  A flush is
// three short data-parallel phases (bookkeeping + scan of the group counts, one half-wave per
// 128-byte group, one half-wave per bin for the carry), not a serial walk over the bins.
template <int LOG_NB>
struct Bins {
    static constexpr int NB = 1 << LOG_NB;
    static constexpr int CAP = PT_BIN_ENTRIES >> LOG_NB;
    static constexpr int MAX_ITEMS = PT_BIN_ENTRIES / 32;
    static_assert(NB <= PT_THREADS, "one bookkeeping thread per bin");
    uint32_t *cnt;    // [NB] entries currently in the bin (may exceed CAP: the excess overflowed)
    uint32_t *cur;    // [NB] entries already written to the private region of the bin
    uint32_t *meta;   // [NB] n | f << 16 of the flush in progress
    uint32_t *data;   // [NB * CAP]
    uint32_t *items;  // [MAX_ITEMS] bin | group << 16
    uint32_t *scan;   // [8]

    __device__ __forceinline__ void init()
    {
        for (int b = threadIdx.x; b < NB; b += PT_THREADS) { cnt[b] = 0; cur[b] = 0; }
    }

    __device__ __forceinline__ bool push(uint32_t b, uint32_t val)
    {
        const uint32_t slot = atomicAdd(&cnt[b], 1u);
        if (slot < (uint32_t)CAP) { data[b * CAP + slot] = val; return true; }
        return false;
    }

    // region: this workgroup's private output, NB consecutive areas of `cap` entries.
    // addr_of(b, val): full filter address of an entry, for the ones diverted to the overflow list.
    template <class AddrOf>
    __device__ __forceinline__ void flush(bool final, uint32_t *region, uint64_t cap, const Overflow &ovf, AddrOf addr_of)
    {
        __syncthreads();
        const uint32_t tid = threadIdx.x;
        uint32_t n = 0, f = 0;
        if (tid < (uint32_t)NB) {
            n = min(cnt[tid], (uint32_t)CAP);
            f = final ? ((n + 31u) & ~31u) : (n & ~31u);
            meta[tid] = n | (f << 16);
        }
        uint32_t total;
        const uint32_t off = block_excl_scan(f >> 5, scan, total);
        for (uint32_t g = 0; g < (f >> 5); g++) items[off + g] = tid | (g << 16);
        __syncthreads();
        const uint32_t l = tid & 31u;
        for (uint32_t w = tid >> 5; w < total; w += PT_THREADS / 32) {
            const uint32_t it = items[w];
            const uint32_t b = it & 0xFFFFu, idx = (it >> 16) * 32u + l;
            const uint32_t nb = meta[b] & 0xFFFFu;
            const uint32_t val = idx < nb ? data[b * CAP + idx] : PT_SENT;
            const uint64_t pos = (uint64_t)cur[b] + idx;
            if (pos < cap) region[(uint64_t)b * cap + pos] = val;
            else if (val != PT_SENT) ovf.push(addr_of(b, val));
        }
        __syncthreads();
        for (uint32_t b = tid >> 5; b < (uint32_t)NB; b += PT_THREADS / 32) {
            const uint32_t m = meta[b];
            const uint32_t nb = m & 0xFFFFu, fb = m >> 16;
            if (fb == 0) continue;  // nothing left the bin
            const uint32_t carry = nb - min(nb, fb);  // < 32
            uint32_t tmp = 0;
            if (l < carry) tmp = data[b * CAP + fb + l];
            if (l < carry) data[b * CAP + l] = tmp;
            if (l == 0) { cur[b] += fb; cnt[b] = carry; }
        }
        __syncthreads();
    }

    __device__ __forceinline__ void store_counts(uint32_t *out, uint64_t cap)
    {
        for (int b = threadIdx.x; b < NB; b += PT_THREADS) out[b] = (uint32_t)min((uint64_t)cur[b], cap);
    }
};

// ------------------------------------------------------------------------------------------ level 1
template <int LOG_NB>
struct HashEmit {
    Bins<LOG_NB> *bins;
    const Overflow *ovf;
    int shift;          // L - B1
    uint32_t remmask;   // 2^(L-B1) - 1
    __device__ __forceinline__ void operator()(uint64_t a)
    {
        if (!bins->push((uint32_t)(a >> shift), (uint32_t)a & remmask)) ovf->push(a);
    }
};

template <int Q, bool GATED, int LOG_NB>
__global__ void __launch_bounds__(PT_THREADS)
k_part_hash(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases,
            const uint32_t *__restrict__ nmask, uint64_t n_text, uint64_t n_tiles, int pos_per_round, uint64_t lo, uint64_t hi,
            uint32_t *buf1, uint32_t *cnt1, uint64_t cap1, Overflow ovf, unsigned long long *n_kmers)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NB = 1 << LOG_NB;
    constexpr int TW = PT_THREADS + 1 + TPC_XW_MAX;
    uint32_t *s_data = reinterpret_cast<uint32_t *>(smem);
    uint64_t *s_b = reinterpret_cast<uint64_t *>(s_data + PT_BIN_ENTRIES);
    uint64_t *s_h = s_b + TW;
    uint64_t *s_hk = s_h + Q * 5;
    uint32_t *s_n = reinterpret_cast<uint32_t *>(s_hk + Q * 5);
    uint32_t *s_cnt = s_n + TW;
    uint32_t *s_cur = s_cnt + NB;
    uint32_t *s_meta = s_cur + NB;
    uint32_t *s_items = s_meta + NB;
    uint32_t *s_w = s_items + Bins<LOG_NB>::MAX_ITEMS;  // 8 words

    Bins<LOG_NB> bins{s_cnt, s_cur, s_meta, s_data, s_items, s_w};
    bins.init();
    const int tid = threadIdx.x;
    if (tid < Q * 5) { s_h[tid] = tab[tid]; s_hk[tid] = tab[TPC_TAB_HK + tid]; }
    const int shift = P.L - LOG_NB;
    HashEmit<LOG_NB> emit{&bins, &ovf, shift, (uint32_t)((1ull << shift) - 1ull)};
    uint32_t *region = buf1 + (uint64_t)blockIdx.x * NB * cap1;
    auto addr_of = [shift](uint32_t b, uint32_t val) { return ((uint64_t)b << shift) | val; };
    const int xw = (P.k + 1) / 32 + 2;
    unsigned hashed = 0;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();  // previous tile's staging is no longer read
        const uint64_t wfirst = tile * PT_THREADS;
        const uint64_t wbase = wfirst - 1;
        for (int i = tid; i < PT_THREADS + 1 + xw; i += PT_THREADS) {
            const int64_t w = (int64_t)wfirst - 1 + i;
            s_b[i] = w >= 0 ? bases[w] : 0ull;
            s_n[i] = w >= 0 ? nmask[w] : 0xFFFFFFFFu;
        }
        __syncthreads();
        const uint64_t g0 = (wfirst + tid) * TPC_RUN;
        const bool active = g0 < n_text;
        TpcRoll<Q> r;
        if (active) tpc_roll_init<Q>(r, P, s_h, s_b, s_n, g0, wbase);
        for (int s0 = 0; s0 < TPC_RUN; s0 += pos_per_round) {
            if (active)
                for (int s = s0; s < s0 + pos_per_round; s++)
                    hashed += tpc_insert_step<Q, GATED>(r, P, s_h, s_hk, s_b, s_n, g0 + s, wbase, lo, hi, emit);
            bins.flush(false, region, cap1, ovf, addr_of);
        }
    }
    bins.flush(true, region, cap1, ovf, addr_of);
    bins.store_counts(cnt1 + (uint64_t)blockIdx.x * NB, cap1);
    if (n_kmers) {
        for (int off = 32; off > 0; off >>= 1) hashed += __shfl_down(hashed, off, 64);
        if ((tid & 63) == 0) s_w[tid >> 6] = hashed;
        __syncthreads();
        if (tid == 0) {
            unsigned t = 0;
            for (int i = 0; i < PT_THREADS / 64; i++) t += s_w[i];
            if (t) atomicAdd(n_kmers, (unsigned long long)t);
        }
    }
}

// ------------------------------------------------------------------------------------------ level 2
template <int LOG_NB1, int LOG_NB2>
__global__ void __launch_bounds__(PT_THREADS)
k_part_split(int L, int slice_bits, uint32_t nwg1, uint32_t wpb, const uint32_t *__restrict__ buf1, const uint32_t *__restrict__ cnt1,
             uint64_t cap1, uint32_t *buf2, uint32_t *cnt2, uint64_t cap2, Overflow ovf)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NB1 = 1 << LOG_NB1, NB2 = 1 << LOG_NB2;
    constexpr int LOADS = 16;
    uint32_t *s_data = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_cnt = s_data + PT_BIN_ENTRIES;
    uint32_t *s_cur = s_cnt + NB2;
    uint32_t *s_meta = s_cur + NB2;
    uint32_t *s_items = s_meta + NB2;
    uint32_t *s_w = s_items + Bins<LOG_NB2>::MAX_ITEMS;
    Bins<LOG_NB2> bins{s_cnt, s_cur, s_meta, s_data, s_items, s_w};
    bins.init();
    const uint32_t b1 = blockIdx.x / wpb, j = blockIdx.x % wpb;
    const uint32_t slice_mask = (1u << slice_bits) - 1u;
    const int shift1 = L - LOG_NB1;
    uint32_t *region = buf2 + (uint64_t)blockIdx.x * NB2 * cap2;
    auto addr_of = [=](uint32_t b2, uint32_t val) { return ((uint64_t)b1 << shift1) | ((uint64_t)b2 << slice_bits) | val; };
    __syncthreads();
    // rounds of LOADS x PT_THREADS entries over the regions (w, b1), w = j, j + wpb, ...; the next
    // round's loads are issued before the current round is binned and flushed
    uint32_t w = j, base = 0;
    uint32_t n = w < nwg1 ? cnt1[(uint64_t)w * NB1 + b1] : 0;
    while (w < nwg1 && n == 0) { w += wpb; n = w < nwg1 ? cnt1[(uint64_t)w * NB1 + b1] : 0; }
    uint32_t v[LOADS], vn[LOADS];
    auto load = [&](uint32_t (&dst)[LOADS], uint32_t ww, uint32_t bb, uint32_t nn) {
        const uint32_t *src = buf1 + ((uint64_t)ww * NB1 + b1) * cap1;
#pragma unroll
        for (int i = 0; i < LOADS; i++) {
            const uint32_t idx = bb + i * PT_THREADS + threadIdx.x;
            dst[i] = idx < nn ? src[idx] : PT_SENT;
        }
    };
    if (w < nwg1) load(v, w, base, n);
    while (w < nwg1) {
        // advance to the next round and prefetch it
        uint32_t w2 = w, base2 = base + LOADS * PT_THREADS, n2 = n;
        if (base2 >= n2) {
            base2 = 0;
            do { w2 += wpb; n2 = w2 < nwg1 ? cnt1[(uint64_t)w2 * NB1 + b1] : 0; } while (w2 < nwg1 && n2 == 0);
        }
        if (w2 < nwg1) load(vn, w2, base2, n2);
#pragma unroll
        for (int i = 0; i < LOADS; i++) {
            if (v[i] != PT_SENT) {
                const uint32_t b2 = v[i] >> slice_bits, val = v[i] & slice_mask;
                if (!bins.push(b2, val)) ovf.push(addr_of(b2, val));
            }
        }
        bins.flush(false, region, cap2, ovf, addr_of);
#pragma unroll
        for (int i = 0; i < LOADS; i++) v[i] = vn[i];
        w = w2; base = base2; n = n2;
    }
    bins.flush(true, region, cap2, ovf, addr_of);
    bins.store_counts(cnt2 + (uint64_t)blockIdx.x * NB2, cap2);
}

// ------------------------------------------------------------------------------------------ level 3
// One workgroup per 2^slice_bits-bit slice of the filter.
__global__ void __launch_bounds__(PT_APPLY_THREADS)
k_part_apply(int slice_bits, int log_nb2, uint32_t wpb, const uint32_t *__restrict__ buf2, const uint32_t *__restrict__ cnt2,
             uint64_t cap2, uint32_t *__restrict__ filter, int fresh)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *slice = reinterpret_cast<uint32_t *>(smem);
    const uint32_t words = 1u << (slice_bits - 5);
    const uint32_t nb2 = 1u << log_nb2;
    const uint32_t b1 = blockIdx.x >> log_nb2, b2 = blockIdx.x & (nb2 - 1);
    uint32_t *out = filter + (uint64_t)blockIdx.x * words;
    const bool wide = (words & 3u) == 0;  // 16-byte accesses whenever the slice allows
    if (fresh) {
        if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += PT_APPLY_THREADS) reinterpret_cast<uint4 *>(slice)[i] = make_uint4(0, 0, 0, 0);
        else for (uint32_t i = threadIdx.x; i < words; i += PT_APPLY_THREADS) slice[i] = 0;
    } else {
        if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += PT_APPLY_THREADS) reinterpret_cast<uint4 *>(slice)[i] = reinterpret_cast<const uint4 *>(out)[i];
        else for (uint32_t i = threadIdx.x; i < words; i += PT_APPLY_THREADS) slice[i] = out[i];
    }
    __syncthreads();
    for (uint32_t j = 0; j < wpb; j++) {
        const uint64_t r = ((uint64_t)b1 * wpb + j) * nb2 + b2;
        const uint32_t *src = buf2 + r * cap2;
        const uint32_t n = cnt2[r];
        for (uint32_t i0 = 0; i0 < n; i0 += 8 * PT_APPLY_THREADS) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t i = i0 + u * PT_APPLY_THREADS + threadIdx.x;
                v[u] = i < n ? src[i] : PT_SENT;
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (v[u] != PT_SENT) atomicOr(&slice[v[u] >> 5], 1u << (v[u] & 31u));
        }
    }
    __syncthreads();
    if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += PT_APPLY_THREADS) reinterpret_cast<uint4 *>(out)[i] = reinterpret_cast<const uint4 *>(slice)[i];
    else for (uint32_t i = threadIdx.x; i < words; i += PT_APPLY_THREADS) out[i] = slice[i];
}

// ------------------------------------------------------------------------------------------ level 4
__global__ void k_part_ovf(const uint64_t *__restrict__ list, const unsigned long long *cursor, uint64_t cap, uint32_t *filter)
{
    const uint64_t n = min((uint64_t)cursor[0], cap);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t a = list[i];
        atomicOr(&filter[a >> 5], 1u << ((uint32_t)a & 31u));
    }
}

template <int Q, int LOG_NB>
void launch_hash(const TpcLaunch &a, const TpcPartPlan &pl, bool gated, uint64_t lo, uint64_t hi, unsigned long long *n_kmers, size_t lds)
{
    Overflow ovf{pl.ovf, pl.ovf_cur, pl.ovf_cap};
    if (gated) {
        (void)hipFuncSetAttribute((const void *)k_part_hash<Q, true, LOG_NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_part_hash<Q, true, LOG_NB>), dim3(pl.nwg1), dim3(PT_THREADS), lds, a.stream, a.P, a.tab, a.bases, a.nmask, a.n_text,
                           pl.n_tiles, pl.pos_per_round, lo, hi, pl.buf1, pl.cnt1, pl.cap1, ovf, n_kmers);
    } else {
        (void)hipFuncSetAttribute((const void *)k_part_hash<Q, false, LOG_NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_part_hash<Q, false, LOG_NB>), dim3(pl.nwg1), dim3(PT_THREADS), lds, a.stream, a.P, a.tab, a.bases, a.nmask, a.n_text,
                           pl.n_tiles, pl.pos_per_round, lo, hi, pl.buf1, pl.cnt1, pl.cap1, ovf, n_kmers);
    }
}

template <int Q>
int launch_hash_q(const TpcLaunch &a, const TpcPartPlan &pl, bool gated, uint64_t lo, uint64_t hi, unsigned long long *n_kmers)
{
    const int nb = 1 << pl.b1;
    const size_t lds = (size_t)PT_BIN_ENTRIES * 4 + (size_t)(PT_THREADS + 1 + TPC_XW_MAX) * 12 + (size_t)Q * 5 * 16 + (size_t)nb * 12 + (size_t)PT_BIN_ENTRIES / 32 * 4 + 64;
    switch (pl.b1) {
    case 1: launch_hash<Q, 1>(a, pl, gated, lo, hi, n_kmers, lds); break;
    case 2: launch_hash<Q, 2>(a, pl, gated, lo, hi, n_kmers, lds); break;
    case 3: launch_hash<Q, 3>(a, pl, gated, lo, hi, n_kmers, lds); break;
    case 4: launch_hash<Q, 4>(a, pl, gated, lo, hi, n_kmers, lds); break;
    case 5: launch_hash<Q, 5>(a, pl, gated, lo, hi, n_kmers, lds); break;
    case 6: launch_hash<Q, 6>(a, pl, gated, lo, hi, n_kmers, lds); break;
    case 7: launch_hash<Q, 7>(a, pl, gated, lo, hi, n_kmers, lds); break;
    case 8: launch_hash<Q, 8>(a, pl, gated, lo, hi, n_kmers, lds); break;
    case 9: launch_hash<Q, 9>(a, pl, gated, lo, hi, n_kmers, lds); break;
    default: return -1;
    }
    return 0;
}

template <int LOG_NB1>
int launch_split_1(const TpcLaunch &a, const TpcPartPlan &pl)
{
    Overflow ovf{pl.ovf, pl.ovf_cur, pl.ovf_cap};
    const int nb2 = 1 << pl.b2;
    const size_t lds = (size_t)PT_BIN_ENTRIES * 4 + (size_t)nb2 * 12 + (size_t)PT_BIN_ENTRIES / 32 * 4 + 64;
    const dim3 grid((unsigned)((1u << pl.b1) * pl.wpb));
#define TPC_SPLIT(B2)                                                                                                              \
    case B2:                                                                                                                       \
        (void)hipFuncSetAttribute((const void *)k_part_split<LOG_NB1, B2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
        hipLaunchKernelGGL((k_part_split<LOG_NB1, B2>), grid, dim3(PT_THREADS), lds, a.stream, a.P.L, pl.slice_bits, pl.nwg1, pl.wpb, pl.buf1, \
                           pl.cnt1, pl.cap1, pl.buf2, pl.cnt2, pl.cap2, ovf);                                                      \
        break;
    switch (pl.b2) {
        TPC_SPLIT(1) TPC_SPLIT(2) TPC_SPLIT(3) TPC_SPLIT(4) TPC_SPLIT(5) TPC_SPLIT(6) TPC_SPLIT(7) TPC_SPLIT(8) TPC_SPLIT(9)
    default: return -1;
    }
#undef TPC_SPLIT
    return 0;
}

}  // namespace

// Partition geometry for a filter of 2^L bits: slices of 2^slice_bits bits, fan-out split over two
// levels.  Returns false when the partitioned path does not apply (tiny filters: direct kernel).
bool tpc_part_plan(int L, int q, int slice_bits, uint64_t n_text, TpcPartPlan &pl)
{
    const int F = L - slice_bits;
    if (F < 2 || slice_bits < 6 || slice_bits > 20) return false;
    pl.slice_bits = slice_bits;
    pl.b1 = (F + 1) / 2;
    pl.b2 = F / 2;
    if (pl.b1 > 9 || L - pl.b1 > 31) return false;  // entries are remainders below the 0xFFFFFFFF sentinel
    pl.n_tiles = (n_text / TPC_RUN + PT_THREADS) / PT_THREADS;
    pl.nwg1 = (uint32_t)std::min<uint64_t>(256, pl.n_tiles);
    pl.wpb = 4;
    // positions per thread per round: keep a round's entries near a third of the bin storage
    const int cap = PT_BIN_ENTRIES >> pl.b1;
    int budget = (1 << pl.b1) * (cap - 32) * 5 / 8;  // entries per round
    int ppr = budget / (PT_THREADS * q);
    pl.pos_per_round = ppr >= 8 ? 8 : ppr >= 4 ? 4 : ppr >= 2 ? 2 : 1;
    const double a_max = (double)q * (double)n_text * 1.02 + 4096;
    const double avg1 = a_max / ((double)pl.nwg1 * (1 << pl.b1));
    pl.cap1 = ((uint64_t)(avg1 * 1.5 + 8 * std::sqrt(avg1) + 128) + 31) & ~31ull;
    const double avg2 = a_max / ((double)(1 << pl.b1) * pl.wpb * (1 << pl.b2));
    pl.cap2 = ((uint64_t)(avg2 * 1.5 + 8 * std::sqrt(avg2) + 128) + 31) & ~31ull;
    pl.ovf_cap = (uint64_t)(a_max / 16) + 65536;
    return true;
}

size_t tpc_part_buf1_bytes(const TpcPartPlan &pl) { return (size_t)pl.nwg1 * (1u << pl.b1) * pl.cap1 * 4; }
size_t tpc_part_cnt1_bytes(const TpcPartPlan &pl) { return (size_t)pl.nwg1 * (1u << pl.b1) * 4; }
size_t tpc_part_buf2_bytes(const TpcPartPlan &pl) { return ((size_t)(1u << pl.b1) * pl.wpb * (1u << pl.b2)) * pl.cap2 * 4; }
size_t tpc_part_cnt2_bytes(const TpcPartPlan &pl) { return ((size_t)(1u << pl.b1) * pl.wpb * (1u << pl.b2)) * 4; }

int tpc_launch_insert_partitioned(const TpcLaunch &a, const TpcPartPlan &pl, uint64_t lo, uint64_t hi, bool gated, bool fresh,
                                  unsigned long long *n_kmers)
{
    int rc = -1;
    switch (a.P.q) {
    case 1: rc = launch_hash_q<1>(a, pl, gated, lo, hi, n_kmers); break;
    case 2: rc = launch_hash_q<2>(a, pl, gated, lo, hi, n_kmers); break;
    case 3: rc = launch_hash_q<3>(a, pl, gated, lo, hi, n_kmers); break;
    case 4: rc = launch_hash_q<4>(a, pl, gated, lo, hi, n_kmers); break;
    case 5: rc = launch_hash_q<5>(a, pl, gated, lo, hi, n_kmers); break;
    case 6: rc = launch_hash_q<6>(a, pl, gated, lo, hi, n_kmers); break;
    case 7: rc = launch_hash_q<7>(a, pl, gated, lo, hi, n_kmers); break;
    case 8: rc = launch_hash_q<8>(a, pl, gated, lo, hi, n_kmers); break;
    }
    if (rc) return rc;
    switch (pl.b1) {
    case 1: rc = launch_split_1<1>(a, pl); break;
    case 2: rc = launch_split_1<2>(a, pl); break;
    case 3: rc = launch_split_1<3>(a, pl); break;
    case 4: rc = launch_split_1<4>(a, pl); break;
    case 5: rc = launch_split_1<5>(a, pl); break;
    case 6: rc = launch_split_1<6>(a, pl); break;
    case 7: rc = launch_split_1<7>(a, pl); break;
    case 8: rc = launch_split_1<8>(a, pl); break;
    case 9: rc = launch_split_1<9>(a, pl); break;
    default: rc = -1;
    }
    if (rc) return rc;
    const size_t lds = (size_t)4 << (pl.slice_bits - 5);
    (void)hipFuncSetAttribute((const void *)k_part_apply, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_part_apply, dim3(1u << (pl.b1 + pl.b2)), dim3(PT_APPLY_THREADS), lds, a.stream, pl.slice_bits, pl.b2, pl.wpb, pl.buf2,
                       pl.cnt2, pl.cap2, a.filter, fresh ? 1 : 0);
    hipLaunchKernelGGL(k_part_ovf, dim3(1024), dim3(256), 0, a.stream, pl.ovf, pl.ovf_cur, pl.ovf_cap, a.filter);
    return 0;
}
