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
//   level 3  k_part_apply  a workgroup takes one slice at a time (long-lived: tpc_slice_grid): slice in LDS (zeroed, or loaded when the
//                          filter already holds bits), ds_or every entry, one coalesced write-out
//   level 4  k_part_ovf    entries that did not fit a region (adversarial skew) are applied with
//                          plain atomicOr; if even that list overflows the host re-runs the
//                          direct kernel -- OR is idempotent, so parity never depends on luck.
//
// The filter contents are identical to the direct path's (bitwise OR is order independent).
// HBM traffic per address: 4 B written + 4 B read per level, plus one sequential pass over the
// filter -- versus one 64-byte read-for-ownership and write-back per address for the atomics.
#include "tpc_bins3.h"
#include "tpc_binsp.h"
#include "tpc_lean.h"
#include "tpc_insert_step.h"
#include "tpc_internal.h"
#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <cmath>

// Two code objects from this file (as tpc_pass2.hip does): part 0 (tpc_partition.o) holds everything a run with the reference's
// default of q = 5 hash functions launches; part 1 (-DTPC_PARTITION_PART=1 -> tpc_partition_q.o) only the level-1 hash kernels of
// the other 15 values of q.  The runtime loads a code object when one of its kernels is first used: 240 kernel instantiations that
// a q = 5 run never launches cost it ~15 ms of start-up (`code object partition` 23.8 -> ~5 ms of the CLI's 253 ms).
#ifndef TPC_PARTITION_PART
#define TPC_PARTITION_PART 0
#endif

#if TPC_PARTITION_PART == 0
uint32_t tpc_test_sched_cap = 0;  // see tpc_bins.h:pt_schedule_dims
int tpc_test_tight_pinch = 0;    // option "test_tight_pinch" (tests): tight level-1 regions at this percentage of their expected fill and a 64-entry overflow list -- a tight plan that must fail
int tpc_test_insert_p3 = 0;       // option "insert_entry_fmt" = 3: 24-bit level-2 insert entries (tpc_part_plan_sharded)
#endif
int tpc_launch_insert_part_hash_other_q(const TpcLaunch &a, const TpcPartPlan &pl, uint64_t lo, uint64_t hi, bool gated, unsigned long long *n_kmers);  // part 1

namespace {

[[maybe_unused]] constexpr uint32_t PT_SENT = 0xFFFFFFFFu;

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

// ------------------------------------------------------------------------------------------ level 1
struct HashEmit {
    Bins3<uint32_t, 1024> *bins;  // k_part_hash's bins (PH_THREADS)
    const Overflow *ovf;
    int shift;          // L - B1
    uint32_t remmask;   // 2^(L-B1) - 1
    PtPerm perm;
    template <int Q>
    __device__ __forceinline__ void edge(const uint64_t (&a0)[Q])
    {
        uint32_t b[Q], val[Q];
        bool ok[Q];
#pragma unroll
        for (int i = 0; i < Q; i++) {
            const uint64_t a = perm.fwd(a0[i]);  // everything downstream works on permuted addresses
            b[i] = (uint32_t)(a >> shift);
            val[i] = (uint32_t)a & remmask;
            ok[i] = true;
        }
        const int sh = shift;
        const Overflow *o = ovf;
        bins->template push_batch<Q>(b, val, ok, [sh, o](uint32_t bb, uint32_t v) { o->push(((uint64_t)bb << sh) | v); });
    }
};

constexpr int PH_THREADS = 1024;  // two 512-word tiles per workgroup round: 16 waves hide the LDS latencies that 8 left exposed
template <int Q, bool GATED, bool SHARDED>
__global__ void __launch_bounds__(PH_THREADS)
k_part_hash(int LOG_NB, TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases,
            const uint32_t *__restrict__ nmask, uint64_t n_text, uint64_t tile0, uint64_t n_tiles, int pos_per_round, uint64_t lo, uint64_t hi,
            uint32_t *buf1, uint32_t *cnt1, uint64_t cap1, Overflow ovf, PtPerm perm, PtShard sh, unsigned long long *n_kmers)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NB = 1 << LOG_NB;
    constexpr int TW = PT_THREADS + 1 + TPC_XW_MAX;
    Bins3<uint32_t, PH_THREADS> bins;
    uint64_t *s_b2 = reinterpret_cast<uint64_t *>(bins.carve(smem, LOG_NB));  // [2][TW]
#ifdef TPC_BINS3_DEBUG
    bins.dbg = ovf.cursor + 8;
#endif
    uint64_t *s_h = s_b2 + 2 * TW;
    uint64_t *s_hk = s_h + Q * 5;
    uint32_t *s_n2 = reinterpret_cast<uint32_t *>(s_hk + Q * 5);              // [2][TW]
    uint32_t *s_w = s_n2 + 2 * TW;  // 16 words
    const int tid = threadIdx.x, half = tid >> 9, lt = tid & (PT_THREADS - 1);
    const uint64_t *s_b = s_b2 + half * TW;
    const uint32_t *s_n = s_n2 + half * TW;
    if (tid < Q * 5) { s_h[tid] = tab[tid]; s_hk[tid] = tab[TPC_TAB_HK + tid]; }
    const int shift = P.L - LOG_NB;
    HashEmit emit{&bins, &ovf, shift, (uint32_t)((1ull << shift) - 1ull), perm};
    const uint32_t wg = blockIdx.x, nwg = gridDim.x;
    // one rank: a workgroup's regions are contiguous ([w][b1]); sharded: destination-major (pt_r1_send)
    auto ridx = [sh, NB, wg, nwg](uint32_t b) { return SHARDED ? pt_r1_send(sh, (uint32_t)NB, nwg, wg, b) : (uint64_t)wg * NB + b; };
    bins.init(buf1, [cap1, ridx](uint32_t b) { return make_uint2((uint32_t)((ridx(b) * cap1) >> 5), (uint32_t)cap1); });  // 32 entries = one 128-byte unit
    auto lost = [shift, ovf](uint32_t b, uint32_t val) { ovf.push(((uint64_t)b << shift) | val); };
    const int xw = (P.k + 1) / 32 + 2;
    unsigned hashed = 0;
    for (uint64_t pair = tile0 + 2 * (uint64_t)blockIdx.x; pair < tile0 + n_tiles; pair += 2 * (uint64_t)gridDim.x) {
        __syncthreads();  // previous tiles' staging is no longer read
        const uint64_t tile = pair + half;
        const bool have = tile < tile0 + n_tiles;
        const uint64_t wfirst = tile * PT_THREADS;
        const uint64_t wbase = wfirst - 1;
        if (have)
            for (int i = lt; i < PT_THREADS + 1 + xw; i += PT_THREADS) {
                const int64_t w = (int64_t)wfirst - 1 + i;
                s_b2[half * TW + i] = w >= 0 ? bases[w] : 0ull;
                s_n2[half * TW + i] = w >= 0 ? nmask[w] : 0xFFFFFFFFu;
            }
        __syncthreads();
        const uint64_t g0 = (wfirst + lt) * TPC_RUN;
        const bool active = have && g0 < n_text;
        TpcRoll<Q> r;
        if (active) tpc_roll_init<Q>(r, P, s_h, s_b, s_n, g0, wbase);
        for (int s0 = 0; s0 < TPC_RUN; s0 += pos_per_round) {
            if (active)
                for (int s = s0; s < min(s0 + pos_per_round, TPC_RUN); s++)  // any round length: the last round of a run may be short
                    hashed += tpc_insert_step<Q, GATED>(r, P, s_h, s_hk, s_b, s_n, g0 + s, wbase, lo, hi, emit);
            bins.template flush<false>(lost);
        }
    }
    bins.template flush<true>(lost);
    bins.store_counts(cnt1, ridx);
    if (n_kmers) {
        for (int off = 32; off > 0; off >>= 1) hashed += __shfl_down(hashed, off, 64);
        if ((tid & 63) == 0) s_w[tid >> 6] = hashed;
        __syncthreads();
        if (tid == 0) {
            unsigned t = 0;
            for (int i = 0; i < PH_THREADS / 64; i++) t += s_w[i];
            if (t) atomicAdd(n_kmers, (unsigned long long)t);
        }
    }
}

// ------------------------------------------------------------------------------------------ level 1, instruction-lean
// k_part_hash rebuilt around the VALU instruction count (the SQ counters put it at 76 % VALU issue; tpc_lean.h): the 32 first
// and 32 next characters of a thread's run sit in registers; every function's roll reads {h[c], hk[rc c]} and {hk[c], h[rc c]}
// as two 16-byte LDS pairs; rotations and the address split run on 32-bit halves; the run is seeded from a table of
// pre-rotated letter hashes when that fits LDS (H(w) = XOR_t rotl(h[w_t], k-1-t), H'(w) = XOR_t rotl(h[rc w_t], t)).
// Same addresses, same regions and overflow entries as k_part_hash (FilterFillerWorker, reference VE.h:1035-1083).
__device__ __forceinline__ uint64_t p_rotl_n(uint64_t x, int L, int r)
{   // fastleftshiftn (cyclichash.h:42-44)
    if (r == 0) return x;
    return ((x & ((1ull << (L - r)) - 1ull)) << r) | (x >> (L - r));
}

__device__ __forceinline__ uint64_t p_spread2(uint32_t m)
{   // every mask bit doubled: bit i -> bits 2i, 2i+1
    uint64_t x = m;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x | (x << 1);
}

__device__ __forceinline__ uint64_t p_lo(const uint4 &e) { return ((uint64_t)e.y << 32) | e.x; }
__device__ __forceinline__ uint64_t p_hi(const uint4 &e) { return ((uint64_t)e.w << 32) | e.z; }

template <int Q, bool GATED, bool SHARDED, bool LHI>
__global__ void __launch_bounds__(PH_THREADS)
k_part_hash2(int LOG_NB, TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases,
             const uint32_t *__restrict__ nmask, uint64_t n_text, uint64_t tile0, uint64_t n_tiles, int pos_per_round, uint64_t lo, uint64_t hi,
             uint32_t *buf1, uint32_t *cnt1, uint64_t cap1, Overflow ovf, PtPerm perm, PtShard sh, unsigned long long *n_kmers, int seed_rows,
             const uint32_t *__restrict__ skip32)
{   // skip32 (tpc_qpartition.hip:k_periodic_build's per_i, or nullptr): positions whose out-edge repeats the one of the position 1 .. 63
    // before them insert nothing.
    // LHI: L > 32.  Every L-bit value lives in two separate 32-bit registers (LeanV, tpc_lean.h: round 4); for L <= 32 the high
    // halves do not exist.
    using V = LeanV<LHI>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NB = 1 << LOG_NB;
    constexpr int TW = PT_THREADS + 1 + TPC_XW_MAX;
    Bins3<uint32_t, PH_THREADS> bins;
    uint64_t *s_b2 = reinterpret_cast<uint64_t *>(bins.carve(smem, LOG_NB));  // [2][TW] bases, N positions cleared to code 0
    uint4 *s_roll = reinterpret_cast<uint4 *>(s_b2 + 2 * TW);                // [2][5][Q]: as next character {h_i[c], hk_i[rc c]}, as first {hk_i[c], h_i[rc c]}
    uint4 *s_seed = s_roll + 10 * Q;                                         // [seed_rows][5][Q]: {rotl(h_i[c], k-1-t), rotl(h_i[rc c], t)}
    uint32_t *s_n2 = reinterpret_cast<uint32_t *>(s_seed + (size_t)seed_rows * 5 * Q);  // [2][TW]
    uint32_t *s_w = s_n2 + 2 * TW;  // 16 words
    const int tid = threadIdx.x, half = tid >> 9, lt = tid & (PT_THREADS - 1);
    const int k = P.k, L = P.L;
    const uint64_t *s_b = s_b2 + half * TW;
    const uint32_t *s_n = s_n2 + half * TW;
    for (int i = tid; i < 5 * Q; i += PH_THREADS) {
        const int c = i / Q, f = i % Q, rc = c == 4 ? 4 : 3 - c;
        const uint64_t h = tab[f * 5 + c], hk = tab[TPC_TAB_HK + f * 5 + c], hr = tab[f * 5 + rc], hkr = tab[TPC_TAB_HK + f * 5 + rc];
        s_roll[i] = make_uint4((uint32_t)h, (uint32_t)(h >> 32), (uint32_t)hkr, (uint32_t)(hkr >> 32));
        s_roll[5 * Q + i] = make_uint4((uint32_t)hk, (uint32_t)(hk >> 32), (uint32_t)hr, (uint32_t)(hr >> 32));
    }
    for (int i = tid; i < seed_rows * 5 * Q; i += PH_THREADS) {
        const int t = i / (5 * Q), c = (i / Q) % 5, f = i % Q, rc = c == 4 ? 4 : 3 - c;
        const uint64_t a = p_rotl_n(tab[f * 5 + c], L, (k - 1 - t) % L), b = p_rotl_n(tab[f * 5 + rc], L, t % L);
        s_seed[i] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    }
    const int shift = L - LOG_NB;
    const uint32_t wg = blockIdx.x, nwg = gridDim.x;
    // one rank: a workgroup's regions are contiguous ([w][b1]); sharded: destination-major (pt_r1_send)
    auto ridx = [sh, NB, wg, nwg](uint32_t b) { return SHARDED ? pt_r1_send(sh, (uint32_t)NB, nwg, wg, b) : (uint64_t)wg * NB + b; };
    bins.init(buf1, [cap1, ridx](uint32_t b) { return make_uint2((uint32_t)((ridx(b) * cap1) >> 5), (uint32_t)cap1); });  // 32 entries = one 128-byte unit
    auto lost = [shift, ovf](uint32_t b, uint32_t val) { ovf.push(((uint64_t)b << shift) | val); };
    LeanRotH<LHI> R;
    R.set(L);
    LeanSplit S;
    S.set(perm, LOG_NB);
    const int xw = (k + 1) / 32 + 2;
    const uint32_t p0 = 32u + (uint32_t)lt * 32u;  // my first position relative to the first staged word (the one before the tile)
    int since_flush = 0;
    // the q Bloom addresses of one edge -> bins
    auto emit_edge = [&](const V (&a)[Q]) {
        constexpr int H = Q > 8 ? (Q + 1) / 2 : Q;  // more than 8 functions: two batches (claims in flight and live registers stay bounded)
#pragma unroll
        for (int h0 = 0; h0 < Q; h0 += H) {
            uint32_t b[H], val[H];
            bool ok[H];
#pragma unroll
            for (int i = 0; i < H; i++) {
                ok[i] = h0 + i < Q;
                if (h0 + i < Q) lean_split_h<LHI>(S, a[h0 + i], b[i], val[i]); else { b[i] = 0; val[i] = 0; }
            }
            bins.template push_batch<H>(b, val, ok, lost);
        }
    };
    for (uint64_t pair = tile0 + 2 * (uint64_t)blockIdx.x; pair < tile0 + n_tiles; pair += 2 * (uint64_t)gridDim.x) {
        __syncthreads();  // previous tiles' staging is no longer read
        const uint64_t tile = pair + half;
        const bool have = tile < tile0 + n_tiles;
        const uint64_t wfirst = tile * PT_THREADS;
        const uint32_t sk = have && skip32 ? skip32[wfirst + (uint64_t)lt] : 0u;  // (issued with the staging loads: its round trip hides behind theirs)
        for (int i = lt; i < PT_THREADS + 1 + xw; i += PT_THREADS) {
            const int64_t w = (int64_t)wfirst - 1 + i;
            uint64_t b = have && w >= 0 ? bases[w] : 0ull;
            const uint32_t m = have && w >= 0 ? nmask[w] : 0xFFFFFFFFu;  // no tile: all N, nothing is emitted
            if (m) b &= ~p_spread2(m);  // an N has code 0 in the staged word: code = c | isN << 2 below
            s_b2[half * TW + i] = b;
            s_n2[half * TW + i] = m;
        }
        __syncthreads();
        // The text is padded with N to whole tiles (tpc_seq_upload): runs past its end find no vertex and need no guard.
        const uint64_t cw64 = s_b[lt + 1];
        const uint32_t cw_lo = (uint32_t)cw64, cw_hi = (uint32_t)(cw64 >> 32);
        const uint32_t nw = s_n[lt + 1];
        const uint32_t cx_lo = lean_chars16(s_b, p0 + (uint32_t)k), cx_hi = lean_chars16(s_b, p0 + (uint32_t)k + 16u);
        const uint32_t nx = lean_nbits32(s_n, p0 + (uint32_t)k);
        uint32_t cp = lean_char(s_b, s_n, p0 - 1u);
        int ncnt = 0;  // N characters inside the window (k - definiteCount, VE.h:1033)
        for (int t = 0; t < k; t += 32) {
            uint32_t bits = lean_nbits32(s_n, p0 + (uint32_t)t);
            if (k - t < 32) bits &= (1u << (k - t)) - 1u;
            ncnt += __popc(bits);
        }
        V pos[Q], neg[Q];  // VertexRollingHash ctor (vertexrollinghash.h:79-102)
#pragma unroll
        for (int i = 0; i < Q; i++) { pos[i] = lv_make<LHI>(0u, 0u); neg[i] = lv_make<LHI>(0u, 0u); }
        if (seed_rows) {
            for (int t0 = 0; t0 < k; t0 += 16) {
                uint32_t ch = lean_chars16(s_b, p0 + (uint32_t)t0), nb = lean_nbits32(s_n, p0 + (uint32_t)t0);
                const int m = min(16, k - t0);
                for (int j = 0; j < m; j++) {
                    const uint32_t c = (ch & 3u) | ((nb & 1u) << 2);
                    ch >>= 2; nb >>= 1;
                    const uint4 *row = s_seed + ((t0 + j) * 5 + (int)c) * Q;
#pragma unroll
                    for (int i = 0; i < Q; i++) { const uint4 e = row[i]; pos[i] = lv_xor<LHI>(pos[i], e.x, e.y); neg[i] = lv_xor<LHI>(neg[i], e.z, e.w); }
                }
            }
        } else {
            for (int t = 0; t < k; t++) {
                const uint32_t c = lean_char(s_b, s_n, p0 + (uint32_t)t), cr = lean_char(s_b, s_n, p0 + (uint32_t)(k - 1 - t));
                const uint4 *rowp = s_roll + c * Q, *rown = s_roll + (5 + cr) * Q;
#pragma unroll
                for (int i = 0; i < Q; i++) {
                    const uint4 ep = rowp[i], en = rown[i];
                    pos[i] = lv_xor<LHI>(R.rotl1(pos[i]), ep.x, ep.y);
                    neg[i] = lv_xor<LHI>(R.rotl1(neg[i]), en.z, en.w);
                }
            }
        }
        // (Round 4 also built an N-free fast path here -- a second copy of the loop for waves whose 64 runs hold no N: no 5-letter
        //  codes, no window N count, no dummy-edge test, VE.h:1040-1058 being the rare case -- and measured it: 2.53 -> 2.80 ms.  The
        //  ~12 instructions it saves per position cost 30 more spilled registers around the two loop bodies; profiles/r04f_*.)
#pragma unroll 1
        for (int s = 0; s < TPC_RUN; s++) {  // not unrolled: one copy of the push and flush code
            // first character of the window and the character after it, as code | isN << 2 (two bit-field extracts and a v_lshl_or each)
            uint32_t cwh = cw_lo, cxh = cx_lo;
            if (s >= 16) { cwh = cw_hi; cxh = cx_hi; }  // (uniform)
            const uint32_t cf = __builtin_amdgcn_ubfe(cwh, 2u * ((uint32_t)s & 15u), 2u) | (__builtin_amdgcn_ubfe(nw, (uint32_t)s, 1u) << 2);
            const uint32_t cn = __builtin_amdgcn_ubfe(cxh, 2u * ((uint32_t)s & 15u), 2u) | (__builtin_amdgcn_ubfe(nx, (uint32_t)s, 1u) << 2);
            const uint4 *rowN = s_roll + cn * Q, *rowF = s_roll + (5 + cf) * Q;
            // function 0 decides the strand and the round gate; hash_extend / hash_prepend of the outgoing edge
            // (cyclichash.h:112-121) are the intermediates of update / reverse_update (cyclichash.h:86-102)
            const uint4 eN0 = rowN[0], eF0 = rowF[0];
            const V ep0 = lv_xor<LHI>(R.rotl1(pos[0]), eN0.x, eN0.y), en0 = lv_xor<LHI>(neg[0], eN0.z, eN0.w);
            const V np0 = lv_xor<LHI>(ep0, eF0.x, eF0.y), nn0 = R.rotr1(lv_xor<LHI>(en0, eF0.z, eF0.w));
            const bool vertex = ncnt == 0;
            bool go = vertex;
            if (GATED && go) {  // VE.h:1063-1073
                const uint64_t first = lv_u64<LHI>(lv_min<LHI>(pos[0], neg[0])), second = lv_u64<LHI>(lv_min<LHI>(np0, nn0));
                go = (first >= lo && first <= hi) || (second >= lo && second <= hi);
            }
            const bool main_edge = go && cn < 4u && !__builtin_amdgcn_ubfe(sk, (uint32_t)s, 1u);
            if (go && (cn | cp) >= 4u) {  // rare: the dummy edges beside an N (VE.h:1048-1058), from the window's hashes before they roll
                const bool out_side = cn >= 4u, in_side = cp >= 4u;
#pragma unroll 1
                for (int d = 0; d < 4; d++) {
                    const int c = (d & 1) ? 3 : 0;
                    if (d < 2 ? !out_side : !in_side) continue;
                    uint64_t p[Q], n[Q];
                    if (d < 2) {  // out-edge v + c
                        const uint4 *r = s_roll + c * Q;
#pragma unroll
                        for (int i = 0; i < Q; i++) { const uint4 e = r[i]; p[i] = lv_u64<LHI>(lv_xor<LHI>(R.rotl1(pos[i]), e.x, e.y)); n[i] = lv_u64<LHI>(lv_xor<LHI>(neg[i], e.z, e.w)); }
                    } else {      // in-edge c + v
                        const uint4 *r = s_roll + (5 + c) * Q;
#pragma unroll
                        for (int i = 0; i < Q; i++) { const uint4 e = r[i]; p[i] = lv_u64<LHI>(lv_xor<LHI>(pos[i], e.x, e.y)); n[i] = lv_u64<LHI>(lv_xor<LHI>(R.rotl1(neg[i]), e.z, e.w)); }
                    }
                    const bool ngd = tpc_pick_neg<Q>(p, n);
                    V a[Q];
#pragma unroll
                    for (int i = 0; i < Q; i++) a[i] = lv_from64<LHI>(ngd ? n[i] : p[i]);
                    emit_edge(a);
                }
            }
            // canonical strand of the out-edge (DetermineStrandExtend, vertexrollinghash.h:170-184)
            bool ng = lv_lt<LHI>(en0, ep0);
            if (main_edge && lv_eq<LHI>(ep0, en0)) {
                ng = false;
                bool decided = false;
#pragma unroll
                for (int i = 1; i < Q; i++) {  // (fully unrolled: a dynamic index would put pos[] / neg[] into scratch memory)
                    const uint4 e = rowN[i];
                    const V pp = lv_xor<LHI>(R.rotl1(pos[i]), e.x, e.y), nn = lv_xor<LHI>(neg[i], e.z, e.w);
                    if (!decided && !lv_eq<LHI>(pp, nn)) { ng = lv_lt<LHI>(nn, pp); decided = true; }
                }
            }
            V a[Q];
            a[0] = lv_sel<LHI>(ng, en0, ep0);
            pos[0] = np0;
            neg[0] = nn0;
#pragma unroll
            for (int i = 1; i < Q; i++) {
                const uint4 eN = rowN[i], eF = rowF[i];
                const V ep = lv_xor<LHI>(R.rotl1(pos[i]), eN.x, eN.y), en = lv_xor<LHI>(neg[i], eN.z, eN.w);
                a[i] = lv_sel<LHI>(ng, en, ep);
                pos[i] = lv_xor<LHI>(ep, eF.x, eF.y);
                neg[i] = R.rotr1(lv_xor<LHI>(en, eF.z, eF.w));
            }
            if (main_edge) emit_edge(a);
            ncnt += (int)(cn >> 2) - (int)(cf >> 2);
            cp = cf;
            if (++since_flush == pos_per_round) { bins.template flush<false>(lost); since_flush = 0; }
        }
    }
    bins.template flush<true>(lost);
    bins.store_counts(cnt1, ridx);
    (void)n_kmers;  // (the vertex count is a property of the text: k_count_kmers, launched beside this kernel when a count is asked for --
                    //  it cost this kernel a register and an instruction per position, used or not)
}

// vertex k-mers (windows of k definite characters) starting in the words [w0, w1): what FilterFillerWorker counts as it goes (VE.h:1033)
__global__ void __launch_bounds__(256) k_count_kmers(const uint32_t *__restrict__ nmask, int k, uint64_t w0, uint64_t w1, unsigned long long *n_kmers)
{
    __shared__ uint32_t s_w[4];
    const uint64_t w = w0 + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t n = 0;
    if (w < w1) {
        const uint64_t first = w << 5;
        int run = 0;
        for (uint64_t j = first; j < first + 31 + (uint64_t)k; j++) {
            run = (nmask[j >> 5] >> (j & 31u)) & 1u ? 0 : run + 1;
            if (j + 1 >= first + (uint64_t)k && run >= k) n++;
        }
    }
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (t) atomicAdd(n_kmers, (unsigned long long)t);
    }
}

#if TPC_PARTITION_PART == 0
// ------------------------------------------------------------------------------------------ level 2
constexpr int PS_THREADS = 1024;  // 16 waves hide the LDS atomic round trips better than 8
// P3: the output regions are blocked lines of 40 x 24-bit entries (tpc_binsp.h:PFmt3, round 5) instead of 32-bit entries -- the
// last level's entries are slice offsets of at most 20 bits; 3.05 bytes each instead of 4 on this kernel's writes and the apply's reads.
// cap2 is then a multiple of 40 and the buffer is addressed in 128-byte lines.
template <bool SHARDED, bool P3>
__global__ void __launch_bounds__(PS_THREADS)
k_part_split(int LOG_NB1, int LOG_NB2, int L, int slice_bits, uint32_t nwg1, uint32_t wpb, const uint32_t *__restrict__ buf1, const uint32_t *__restrict__ cnt1,
             uint64_t cap1, uint32_t *buf2, uint32_t *cnt2, uint64_t cap2, Overflow ovf, PtShard sh, uint32_t prev_wpb, int log_prev_nb2, int loads,
             uint32_t nreg_cap, uint32_t sched_cap, const uint64_t *__restrict__ off1, const uint32_t *__restrict__ own1, const uint32_t *__restrict__ owncnt1)
{   // own1 / owncnt1 (sharded, optional): the block of source rank == this rank is read from the send buffers it was hashed into   // prev_wpb == 0: the input is level 1's output, regions [workgroup][bucket].  prev_wpb > 0: the input is the output of
    // another k_part_split (three-level geometry): this bucket is (b1, b2) of that level, its regions are [b1][j][b2].
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t NB1 = 1u << LOG_NB1, NB2 = 1u << LOG_NB2;
    constexpr int LOADS = 16;  // upper bound; `loads` of them are used
    typename std::conditional<P3, BinsP<PFmt3, PS_THREADS>, Bins3<uint32_t, PS_THREADS>>::type bins;
    unsigned char *s_free = bins.carve(smem, LOG_NB2);
    const uint32_t bl = blockIdx.x / wpb, j = blockIdx.x % wpb;  // local bucket, share of its source regions
    // global bucket: a sharded rank numbers its buckets compactly (bl = b1 / world); at the third level bl = (local b1, b2)
    const uint32_t b1 = prev_wpb ? (sh.world > 1 ? ((((bl >> log_prev_nb2) * sh.world + sh.rank) << log_prev_nb2) | (bl & ((1u << log_prev_nb2) - 1u))) : bl)
                                 : (SHARDED ? bl * sh.world + sh.rank : bl);
    const uint32_t nvw = prev_wpb ? prev_wpb : SHARDED ? nwg1 * sh.world : nwg1;  // source regions: (source rank, workgroup)
    auto r1 = [=](uint32_t vw) {
        if (prev_wpb) return ((((uint64_t)(bl >> log_prev_nb2) * prev_wpb) + vw) << log_prev_nb2) + (bl & ((1u << log_prev_nb2) - 1u));
        return SHARDED ? pt_r1_recv(sh, NB1, nwg1, vw / nwg1, vw % nwg1, bl) : (uint64_t)vw * NB1 + bl;
    };
    auto mine = [=](uint32_t vw) { return SHARDED && own1 && !prev_wpb && vw / nwg1 == sh.rank; };
    const uint32_t slice_mask = (1u << slice_bits) - 1u;
    const int shift1 = L - LOG_NB1;
    auto addr_of = [=](uint32_t b2, uint32_t val) { return ((uint64_t)b1 << shift1) | ((uint64_t)b2 << slice_bits) | val; };
    // entries that found no room: one at a time from a push, a whole ring group with one reservation from a flush (tpc_bins3.h:pt_bulk)
    struct Lost2 {
        Overflow ovf; uint32_t b1; int shift1, slice_bits;
        __device__ __forceinline__ uint64_t addr(uint32_t b2, uint32_t val) const { return ((uint64_t)b1 << shift1) | ((uint64_t)b2 << slice_bits) | val; }
        __device__ __forceinline__ void operator()(uint32_t b2, uint32_t val) const { ovf.push(addr(b2, val)); }
        __device__ __forceinline__ unsigned long long reserve(uint32_t n) const { return atomicAdd(ovf.cursor, (unsigned long long)n); }
        __device__ __forceinline__ void put(uint32_t b2, uint32_t val, unsigned long long at) const { if (at < ovf.cap) ovf.list[at] = addr(b2, val); else ovf.cursor[1] = 1ull; }
    };
    const Lost2 lost{ovf, b1, shift1, slice_bits};
    (void)addr_of;
    {
        if constexpr (P3) {
            const uint32_t cl = (uint32_t)(cap2 / PFmt3::GROUP);  // lines per region
            bins.init(buf2, [cl, NB2](uint32_t b) { return make_uint2((uint32_t)(((uint64_t)blockIdx.x * NB2 + b) * cl), cl); });
        } else {
            const uint64_t first = (uint64_t)blockIdx.x * NB2 * cap2;  // this workgroup's regions in buf2, entries
            bins.init(buf2, [first, cap2](uint32_t b) { return make_uint2((uint32_t)((first + (uint64_t)b * cap2) >> 5), (uint32_t)cap2); });
        }
    }
    __syncthreads();
    // rounds of `loads` x PS_THREADS entries over the regions (w, b1), w = j, j + wpb, ..., taken from the round schedule;
    // the next round's loads are issued before the current round is binned and flushed.  The loads are unpredicated (lanes
    // past the end read entry 0 and are masked when the round is consumed) and the two buffers alternate by name.
    const uint32_t nreg = j < nvw ? (nvw - j + wpb - 1) / wpb : 0;
    const uint32_t step = (uint32_t)loads * PS_THREADS;
    uint32_t *s_scan = reinterpret_cast<uint32_t *>(s_free);  // [32] scratch of the schedule's block scan
    uint32_t *s_cnt = s_scan + 32;                            // [nreg_cap]
    uint32_t *s_sched = s_cnt + nreg_cap;                    // [sched_cap]
    struct Round { uint32_t t, base, n; };  // region (j + t * wpb), first entry of the round, entries in the region; all scalar
    uint32_t va[LOADS], vb[LOADS];
    for (uint32_t skip = 0;; skip += sched_cap) {
        const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)pt_build_schedule<PS_THREADS>(
            nreg, step, skip, sched_cap, s_cnt, s_sched, s_scan, [&](uint32_t t) { const uint32_t vw = j + t * wpb; return (mine(vw) ? owncnt1 : cnt1)[r1(vw)]; }));
        const uint32_t n_seg = min(total - min(total, skip), sched_cap);
        auto round_at = [&](uint32_t r) {
            const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_sched[min(r, n_seg - 1u)]);
            Round x{e & 0xFFFFu, (e >> 16) * step, 0u};
            if (r < n_seg) x.n = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cnt[x.t]);
            return x;
        };
        auto valid = [&](const Round &x, int i) { return i < loads && x.base + i * PS_THREADS + threadIdx.x < x.n; };
        auto load = [&](uint32_t (&dst)[LOADS], const Round &x) {
            const uint32_t vw = j + x.t * wpb;
            const uint64_t ri = r1(vw);
            const uint32_t *src = (mine(vw) ? own1 : buf1) + (off1 ? off1[ri] : ri * cap1);  // off1: the regions arrived packed (compacted exchange)
#pragma unroll
            for (int i = 0; i < LOADS; i++) dst[i] = src[valid(x, i) ? x.base + i * PS_THREADS + threadIdx.x : 0u];
        };
        if (n_seg) {
            Round x0 = round_at(0);
            load(va, x0);
            uint32_t r = 0;
            auto round = [&](uint32_t (&cur)[LOADS], uint32_t (&pre)[LOADS]) {
                const Round x1 = round_at(r + 1);
                load(pre, x1);
                // two batches of 8: the claims of a batch (and what a full ring's retry loop keeps live) fit the register file
#pragma unroll
                for (int h = 0; h < LOADS; h += LOADS / 2) {
                    uint32_t bb[LOADS / 2], val[LOADS / 2];
                    bool ok[LOADS / 2];
#pragma unroll
                    for (int i = 0; i < LOADS / 2; i++) {
                        // An entry equal to its left neighbour's (the lane before holds the entry before in the region) is dropped: OR is
                        // idempotent, and a low-complexity tract -- poly-A: the same five addresses for hundreds of positions in a row --
                        // otherwise arrives as a burst of identical entries that fills its bin's ring inside the round (the 62-genome text
                        // with such tracts: 1.9 M entries per step took the overflow list's way, k_part_split 2.4 -> 4.3 ms).  One DPP
                        // move and a compare; lane 0 of a row of 16 has no neighbour and keeps its entry.
                        const uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp((int)~cur[h + i], (int)cur[h + i], 0x111 /* row_shr:1 */, 0xF, 0xF, false);
                        ok[i] = valid(x0, h + i) && cur[h + i] != PT_SENT && left != cur[h + i];
                        bb[i] = cur[h + i] >> slice_bits; val[i] = cur[h + i] & slice_mask;
                    }
                    bins.template push_batch<LOADS / 2>(bb, val, ok, lost);
                }
                bins.template flush<false>(lost);
                x0 = x1; r++;
            };
            while (true) {
                if (r >= n_seg) break;
                round(va, vb);
                if (r >= n_seg) break;
                round(vb, va);
            }
        }
        if (total <= skip + sched_cap) break;
        pt_barrier_lds();  // every wave is done with this segment of the schedule
    }
    bins.template flush<true>(lost);
    bins.store_counts(cnt2 + (uint64_t)blockIdx.x * NB2, [](uint32_t b) { return b; });
}

// ------------------------------------------------------------------------------------------ level 3
// One 2^slice_bits-bit slice of the filter at a time per workgroup (every gridDim.x-th slice: tpc_internal.h:tpc_slice_grid).
template <bool P3>  // the regions are blocked lines of 40 x 24-bit entries (see k_part_split)
__global__ void __launch_bounds__(PT_APPLY_THREADS)
k_part_apply(int slice_bits, int log_nb2, uint32_t wpb, const uint32_t *__restrict__ buf2, const uint32_t *__restrict__ cnt2,
             uint64_t cap2, uint32_t *__restrict__ filter, int fresh, PtPerm perm, PtShard sh, uint32_t n_slices)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *slice = reinterpret_cast<uint32_t *>(smem);
    const uint32_t words = 1u << (slice_bits - 5);
    const uint32_t nb2 = 1u << log_nb2;
    // a long-lived workgroup takes every gridDim.x-th slice (tpc_internal.h:tpc_slice_grid)
    for (uint32_t sl = blockIdx.x; sl < n_slices; sl += gridDim.x) {
        const uint32_t b1 = sl >> log_nb2, b2 = sl & (nb2 - 1);  // local bucket, sub-bucket
        // whole filter: natural slice position of permuted slice blockIdx; shard: compact [local bucket][b2]
        uint32_t *out = filter + (uint64_t)(sh.world == 1 ? perm.slice_of(sl) : sl) * words;
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
            const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt2[r]);
            if constexpr (P3) {
                PlStream<PFmt3, PT_APPLY_THREADS, 1> is;
                is.begin(reinterpret_cast<const unsigned char *>(buf2) + r * (cap2 / PFmt3::GROUP) * PT_LINE, n);
                is.finish([slice](uint32_t v, uint32_t) { atomicOr(&slice[v >> 5], 1u << (v & 31u)); });
            } else
            pt_stream_region<PT_APPLY_THREADS, 2>(buf2 + r * cap2, n, [slice](uint32_t v) { atomicOr(&slice[v >> 5], 1u << (v & 31u)); });
        }
        __syncthreads();
        if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += PT_APPLY_THREADS) reinterpret_cast<uint4 *>(out)[i] = reinterpret_cast<const uint4 *>(slice)[i];
        else for (uint32_t i = threadIdx.x; i < words; i += PT_APPLY_THREADS) out[i] = slice[i];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ level 4
__global__ void k_part_ovf(const uint64_t *__restrict__ list, const unsigned long long *cursor, uint64_t cap, uint32_t *filter, PtPerm perm,
                           PtShard sh, int log_nb2)
{
    const uint64_t n = min((uint64_t)cursor[0], cap);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        bool mine;
        const uint64_t a = pt_local_addr(perm, sh, log_nb2, list[i], mine);  // entries of other ranks are routed by the host layer
        if (mine) atomicOr(&filter[a >> 5], 1u << ((uint32_t)a & 31u));
    }
}

#endif  // part 0

template <int Q>
int launch_hash_q(const TpcLaunch &a, const TpcPartPlan &pl, bool gated, uint64_t lo, uint64_t hi, unsigned long long *n_kmers)
{
    Overflow ovf{pl.ovf, pl.ovf_cur, pl.ovf_cap};
    const PtPerm perm{pl.slice_bits, pl.b1 + pl.b2 + pl.b3, pl.perm_mult, pl.perm_inv};
    const PtShard sh{pl.rank, pl.world};
    if (perm.F <= 24 && !TpcEnv::get().no_lean) {  // the instruction-lean kernel (a 24-bit slice index)
        const size_t fixed = Bins3<uint32_t, PH_THREADS>::lds_bytes(pl.b1) + (size_t)(PT_THREADS + 1 + TPC_XW_MAX) * 24 + (size_t)Q * 10 * 16 + 64 + 64;
        const size_t seed = (size_t)a.P.k * 5 * Q * 16;  // the seed table, when it fits beside the rings (160 KB per workgroup)
        const int seed_rows = fixed + seed <= (size_t)160 * 1024 - 256 ? a.P.k : 0;
        const size_t lds = fixed + (seed_rows ? seed : 0);
#define TPC_HASH2_GO(G, S, H)                                                                                                               \
    do {                                                                                                                                    \
        (void)hipFuncSetAttribute((const void *)k_part_hash2<Q, G, S, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        hipLaunchKernelGGL((k_part_hash2<Q, G, S, H>), dim3(pl.nwg1), dim3(PH_THREADS), lds, a.stream, pl.b1, a.P, a.tab, a.bases, a.nmask, a.n_text, \
                           pl.tile0, pl.n_tiles, pl.pos_per_round, lo, hi, pl.buf1, pl.cnt1, pl.cap1, ovf, perm, sh, n_kmers, seed_rows,   \
                           a.per_i);  /* (global word index: the sharded variants skip too) */                                    \
    } while (0)
#define TPC_HASH2_GS(H)                                                                                                                     \
    do {                                                                                                                                    \
        if (pl.world > 1) { if (gated) TPC_HASH2_GO(true, true, H); else TPC_HASH2_GO(false, true, H); }                                    \
        else { if (gated) TPC_HASH2_GO(true, false, H); else TPC_HASH2_GO(false, false, H); }                                               \
    } while (0)
        if (a.P.L > 32) TPC_HASH2_GS(true); else TPC_HASH2_GS(false);  // L-bit values on 32-bit halves: no high halves at all for L <= 32
        if (n_kmers && pl.n_tiles) {
            const uint64_t w0 = pl.tile0 * PT_THREADS, w1 = (pl.tile0 + pl.n_tiles) * PT_THREADS;
            hipLaunchKernelGGL(k_count_kmers, dim3((unsigned)((w1 - w0 + 255) / 256)), dim3(256), 0, a.stream, a.nmask, a.P.k, w0, w1, n_kmers);
        }
#undef TPC_HASH2_GS
#undef TPC_HASH2_GO
        return 0;
    }
    if constexpr (Q > 8) return -1;  // 9..16 functions: instantiated for the lean kernel only (the caller falls back to the direct kernel)
    else {
    const size_t lds = Bins3<uint32_t, PH_THREADS>::lds_bytes(pl.b1) + (size_t)(PT_THREADS + 1 + TPC_XW_MAX) * 24 + (size_t)Q * 5 * 16 + 64;
#define TPC_HASH_GO(G, S)                                                                                                                   \
    do {                                                                                                                                    \
        (void)hipFuncSetAttribute((const void *)k_part_hash<Q, G, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
        hipLaunchKernelGGL((k_part_hash<Q, G, S>), dim3(pl.nwg1), dim3(PH_THREADS), lds, a.stream, pl.b1, a.P, a.tab, a.bases, a.nmask, a.n_text, \
                           pl.tile0, pl.n_tiles, pl.pos_per_round, lo, hi, pl.buf1, pl.cnt1, pl.cap1, ovf, perm, sh, n_kmers);            \
    } while (0)
    if (pl.world > 1) { if (gated) TPC_HASH_GO(true, true); else TPC_HASH_GO(false, true); }
    else { if (gated) TPC_HASH_GO(true, false); else TPC_HASH_GO(false, false); }
#undef TPC_HASH_GO
    return 0;
    }
}

#if TPC_PARTITION_PART == 0
// entries per thread and round of k_part_split for a level with 2^bits bins: 5/8 of the ring storage, less the leftovers (< one line) per bin
int split_loads(int bits, bool p3 = false)
{
    if (bits == 9 && !p3 && TpcEnv::get().split_loads9 > 0) return std::min(14, TpcEnv::get().split_loads9);
    const int group = p3 ? PFmt3::GROUP : 32;
    const int cap = p3 ? (int)BinsP<PFmt3, PS_THREADS>::cap_for(bits) : (PT_BIN_BYTES / 4) >> bits;
    // (512 bins: rings of 64 entries -- 10 loads put 20 entries into a bin per round against 33 free slots at worst, and 1.5 rings per round
    //  overflow into the wait path; 8 measured 0.18 ms better on the 62-genome text at f = 38, 6 the same, 12 far worse: TPC_SPLIT_LOADS9)
    const int most = p3 ? 16 : bits >= 9 ? 8 : 14;
    return std::max(1, std::min(most, (1 << bits) * std::max(cap - group, 4) * 5 / 8 / PS_THREADS));
}

int launch_split(const TpcLaunch &a, const TpcPartPlan &pl)
{
    Overflow ovf{pl.ovf, pl.ovf_cur, pl.ovf_cap};
    const PtShard sh{pl.rank, pl.world};
    const bool p3 = pl.fmt2 == 3;  // (two levels, one rank: tpc_part_plan_sharded)
    const size_t lds_base = (p3 ? BinsP<PFmt3, PS_THREADS>::lds_bytes(pl.b2) : Bins3<uint32_t, PS_THREADS>::lds_bytes(std::max(pl.b2, pl.b3))) + 128;
    const dim3 grid((unsigned)(((1u << pl.b1) / pl.world) * pl.wpb));  // local buckets only
    const int low_bits = pl.slice_bits + pl.b3;  // address bits below this level's bin index
    uint32_t nreg_cap, sched_cap;
    size_t lds;
    const int loads2 = split_loads(pl.b2, p3);
    pt_schedule_dims(pl.nwg1 * pl.world, pl.wpb, pl.cap1, (uint32_t)loads2 * PS_THREADS, lds_base, nreg_cap, sched_cap, lds);
    if (pl.world > 1) {
        (void)hipFuncSetAttribute((const void *)k_part_split<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_part_split<true, false>), grid, dim3(PS_THREADS), lds, a.stream, pl.b1, pl.b2, a.P.L, low_bits, pl.nwg1, pl.wpb, pl.rbuf1, pl.rcnt1,
                           pl.cap1, pl.buf2, pl.cnt2, pl.cap2, ovf, sh, 0u, 0, loads2, nreg_cap, sched_cap, pl.roff1, pl.rown1, pl.rowncnt1);
    } else if (p3) {
        (void)hipFuncSetAttribute((const void *)k_part_split<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_part_split<false, true>), grid, dim3(PS_THREADS), lds, a.stream, pl.b1, pl.b2, a.P.L, low_bits, pl.nwg1, pl.wpb, pl.rbuf1, pl.rcnt1,
                           pl.cap1, pl.buf2, pl.cnt2, pl.cap2, ovf, sh, 0u, 0, loads2, nreg_cap, sched_cap, pl.roff1, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
    } else {
        (void)hipFuncSetAttribute((const void *)k_part_split<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_part_split<false, false>), grid, dim3(PS_THREADS), lds, a.stream, pl.b1, pl.b2, a.P.L, low_bits, pl.nwg1, pl.wpb, pl.rbuf1, pl.rcnt1,
                           pl.cap1, pl.buf2, pl.cnt2, pl.cap2, ovf, sh, 0u, 0, loads2, nreg_cap, sched_cap, pl.roff1, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
    }
    if (pl.b3) {  // third level: bucket (b1, b2), input = the regions written above
        const int loads3 = split_loads(pl.b3);
        pt_schedule_dims(pl.wpb, pl.wpb3, pl.cap2, (uint32_t)loads3 * PS_THREADS, lds_base, nreg_cap, sched_cap, lds);
        (void)hipFuncSetAttribute((const void *)k_part_split<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_part_split<false, false>), dim3((unsigned)(((1u << (pl.b1 + pl.b2)) / pl.world) * pl.wpb3)), dim3(PS_THREADS), lds, a.stream, pl.b1 + pl.b2, pl.b3, a.P.L,
                           pl.slice_bits, 0u, pl.wpb3, pl.buf2, pl.cnt2, pl.cap2, pl.buf3, pl.cnt3, pl.cap3, ovf, sh, pl.wpb, pl.b2, loads3, nreg_cap, sched_cap, (const uint64_t *)nullptr,
                           (const uint32_t *)nullptr, (const uint32_t *)nullptr);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------ compacted exchange
// off[i] = sum of cnt[0..i) (entries), off[n] = total: one workgroup, every wave owns a contiguous segment (coalesced reads).
__global__ void __launch_bounds__(1024) k_region_offsets(const uint32_t *__restrict__ cnt, uint32_t n, uint64_t *__restrict__ off)
{
    __shared__ unsigned long long s_w[16];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t seg = ((n + 15u) / 16u + 63u) & ~63u;
    const uint64_t i0 = (uint64_t)wv * seg, i1 = min((uint64_t)n, i0 + seg);
    unsigned long long sum = 0;
    for (uint64_t i = i0 + lane; i < i1; i += 64) sum += cnt[i];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) s_w[wv] = sum;
    __syncthreads();
    unsigned long long run = 0, tot = 0;
    for (uint32_t i = 0; i < 16; i++) { const unsigned long long x = s_w[i]; if (i < wv) run += x; tot += x; }
    for (uint64_t i = i0; i < i1; i += 64) {  // uniform per wave
        const bool in = i + lane < i1;
        const unsigned long long v = in ? cnt[i + lane] : 0;
        unsigned long long inc = v;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(inc, o, 64);
            if ((int)lane >= o) inc += t;
        }
        if (in) off[i + lane] = run + inc - v;
        run += __shfl(inc, 63, 64);
    }
    if (threadIdx.x == 0) off[n] = tot;
}

// The insert's overflow entries (permuted full addresses) in slice order for the fused apply + lookup kernel: count per slice,
// offsets (k_region_offsets), scatter.  The order inside a slice does not matter (the bits are OR-ed).
// (a sharded filter: the slice number is the LOCAL one, [local bucket][b2], and the entries of other ranks -- the list is the all-gathered
//  one -- are left out: pt_local_addr)
__device__ __forceinline__ bool ovf_slice(uint64_t a, int slice_bits, PtShard sh, int log_nb2, uint32_t &s)
{
    const uint32_t sp = (uint32_t)(a >> slice_bits);
    if (sh.world == 1) { s = sp; return true; }
    const uint32_t b1 = sp >> log_nb2;
    s = ((b1 >> sh.log_world()) << log_nb2) | (sp & ((1u << log_nb2) - 1u));
    return (b1 & (sh.world - 1)) == sh.rank;
}

__global__ void __launch_bounds__(256) k_ovf_count(const uint64_t *__restrict__ list, uint64_t n, int slice_bits, PtShard sh, int log_nb2, uint32_t *cnt)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    // neighbours in the list mostly share their slice (a ring group that found its region full is appended as one run): a run inside
    // the wave costs ONE atomic -- the entries of a hot slice otherwise queue up on its counter (2.5 ms for 1.9 M entries)
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += stride) {
        const uint64_t i = i0 + (threadIdx.x & 63u);
        uint32_t s = 0xFFFFFFFFu;
        const bool mine = i < n && ovf_slice(list[i], slice_bits, sh, log_nb2, s);
        if (!mine) s = 0xFFFFFFFFu;
        const uint32_t prev = __shfl_up(s, 1, 64);
        const bool head = (threadIdx.x & 63u) == 0u || prev != s;
        const unsigned long long hm = __ballot(head);
        if (mine && head) {
            const uint32_t lane = threadIdx.x & 63u;
            const unsigned long long above = lane == 63u ? 0ull : (hm >> (lane + 1u));
            const uint32_t len = above ? (uint32_t)__ffsll((long long)above) : 64u - lane;
            atomicAdd(&cnt[s], len);
        }
    }
}

__global__ void __launch_bounds__(256) k_ovf_scatter(const uint64_t *__restrict__ list, uint64_t n, int slice_bits, PtShard sh, int log_nb2, const uint64_t *__restrict__ off,
                                                     uint32_t *cursor, uint64_t *__restrict__ sorted)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += stride) {  // runs of one slice: one atomic (k_ovf_count)
        const uint64_t i = i0 + (threadIdx.x & 63u);
        const uint32_t lane = threadIdx.x & 63u;
        const uint64_t a = i < n ? list[i] : 0ull;
        uint32_t s = 0xFFFFFFFFu;
        const bool mine = i < n && ovf_slice(a, slice_bits, sh, log_nb2, s);
        if (!mine) s = 0xFFFFFFFFu;
        const uint32_t prev = __shfl_up(s, 1, 64);
        const bool head = lane == 0u || prev != s;
        const unsigned long long hm = __ballot(head);
        const unsigned long long below = hm & ((2ull << lane) - 1ull);       // heads at or below me: the highest is my run's
        const uint32_t hl = 63u - (uint32_t)__clzll((long long)below);
        uint32_t base = 0;
        if (mine && head) {
            const unsigned long long above = lane == 63u ? 0ull : (hm >> (lane + 1u));
            const uint32_t len = above ? (uint32_t)__ffsll((long long)above) : 64u - lane;
            base = atomicAdd(&cursor[s], len);
        }
        base = __shfl(base, (int)hl, 64);
        if (mine) sorted[off[s] + base + (lane - hl)] = a;
    }
}

// One workgroup per region: the used prefix (a whole number of 16-byte units) moves to its packed position.
__global__ void __launch_bounds__(256) k_region_pack(const uint4 *__restrict__ regions, uint64_t cap16, uint32_t per16, const uint32_t *__restrict__ cnt,
                                                     const uint64_t *__restrict__ off, uint4 *__restrict__ packed)
{   // per16: entries per 16 bytes
    const uint32_t r = blockIdx.x;
    const uint32_t n16 = (cnt[r] + per16 - 1) / per16;
    const uint4 *src = regions + (uint64_t)r * cap16;
    uint4 *dst = packed + off[r] / per16;
    for (uint32_t i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
}

#endif  // part 0
}  // namespace

#if TPC_PARTITION_PART == 0
// Partition geometry for a filter of 2^L bits: slices of 2^slice_bits bits, fan-out split over two
// levels.  Returns false when the partitioned path does not apply (tiny filters: direct kernel).
bool tpc_part_plan(int L, int q, int slice_bits, uint64_t n_tiles, double frac, TpcPartPlan &pl, int levels)
{
    return tpc_part_plan_sharded(L, q, slice_bits, n_tiles, frac, 0, 1, pl, levels, false, true);
}

// n_tiles: the tiles THIS rank hashes; the level-2 regions are sized for the entries of all ranks
// tight: level-1 regions sized at the expected fill + 6 sigma instead of 1.3 x + 8 sigma -- the regions of a sharded pass travel
// whole (equal-block all_to_all: no packing pass), so slack is wire bytes; what does not fit goes the overflow list's way
static bool part_plan_compute(int L, int q, int slice_bits, uint64_t n_tiles, double frac, uint32_t rank, uint32_t world, TpcPartPlan &pl, int levels, bool tight, bool packed);

bool tpc_part_plan_sharded(int L, int q, int slice_bits, uint64_t n_tiles, double frac, uint32_t rank, uint32_t world, TpcPartPlan &pl, int levels, bool tight, bool packed)
{   // the last plan per thread is kept (tpc_qpart_plan_sharded says why); the caller fills in the device pointers afterwards
    struct Key { int L, q, slice_bits, levels, p3; uint64_t n_tiles; double frac; uint32_t rank, world; bool tight, packed; };
    static thread_local Key last{};
    static thread_local TpcPartPlan last_pl;
    static thread_local bool have = false, last_ok = false;
    const Key k{L, q, slice_bits, levels, (int)tpc_test_insert_p3 + (tpc_test_tight_pinch << 8), n_tiles, frac, rank, world, tight, packed};
    if (have && k.L == last.L && k.q == last.q && k.slice_bits == last.slice_bits && k.levels == last.levels && k.p3 == last.p3 && k.n_tiles == last.n_tiles &&
        k.frac == last.frac && k.rank == last.rank && k.world == last.world && k.tight == last.tight && k.packed == last.packed) {
        if (last_ok) pl = last_pl;
        return last_ok;
    }
    last_ok = part_plan_compute(L, q, slice_bits, n_tiles, frac, rank, world, pl, levels, tight, packed);
    last = k; have = true;
    if (last_ok) last_pl = pl;
    return last_ok;
}

static bool part_plan_compute(int L, int q, int slice_bits, uint64_t n_tiles, double frac, uint32_t rank, uint32_t world, TpcPartPlan &pl, int levels, bool tight, bool packed)
{
    pl = TpcPartPlan();
    pl.rank = rank; pl.world = world;
    const uint64_t n_text = n_tiles * PT_THREADS * TPC_RUN;  // positions of this batch of 512-word tiles
    const int F = L - slice_bits;
    if (F < 2 || slice_bits < 6 || slice_bits > 20) return false;
    pl.slice_bits = slice_bits;
    // fan-out 2^F over two levels of <= 9 bits, or three when F > 18 (L > 38 at the default slice size) or asked for;
    // level-1 entries are the L - b1 low address bits in a uint32 below the 0xFFFFFFFF sentinel
    const bool three = levels == 3 || (levels == 0 && F > 18);
    if (three) {
        if (F < 3) return false;
        pl.b1 = std::max((F + 2) / 3, L - 31);
        pl.b2 = (F - pl.b1 + 1) / 2;
        pl.b3 = F - pl.b1 - pl.b2;
        if (pl.b1 > 9 || pl.b2 < 1 || pl.b3 < 1 || pl.b2 > 9) return false;
    } else {
        pl.b1 = (F + 1) / 2;
        pl.b2 = F / 2;
        pl.b3 = 0;
    }
    if (pl.b1 > 9 || L - pl.b1 > 31) return false;  // entries are remainders below the 0xFFFFFFFF sentinel
    if (world == 0 || (world & (world - 1)) || world > (1u << pl.b1)) return false;  // ranks own whole buckets
    pl.n_tiles = n_tiles;
    pl.tile0 = 0;
    pl.nwg1 = (uint32_t)std::min<uint64_t>(256, (pl.n_tiles + 1) / 2);  // k_part_hash takes two tiles per workgroup round
    // level-2 workgroups per bucket: one per bucket is fastest once the buckets alone fill the chip (measured 1 / 2 / 4 / 8 on M2:
    // 45.9 / 46.3 / 47.9 / 51.3 ms per step); fewer local buckets (small filters, sharded filters) are split further
    pl.wpb = std::max<uint32_t>(1, std::min<uint32_t>(8, (256u * world) >> pl.b1));
    // positions per thread per round: keep a round's entries near a third of the bin storage
    const int cap = (PT_BIN_BYTES / 4) >> pl.b1;
    int budget = (1 << pl.b1) * (cap - 32) * 5 / 8;  // entries per round
    // frac: expected share of positions that emit (a gated round only inserts edges touching its
    // vertex-hash range), so a round can cover more positions before the rings fill
    int ppr = (int)(budget / (1024 * q * std::max(frac, 1.0 / 64)));  // k_part_hash: 1024 threads (two tiles) x pos_per_round
    pl.pos_per_round = std::max(1, std::min(32, ppr));
    if (TpcEnv::get().ppr_insert) pl.pos_per_round = std::max(1, std::min(32, TpcEnv::get().ppr_insert));  // measurements: positions per ring round
    const double a_max = (double)q * (double)n_text * 1.02 + 4096;
    // a workgroup takes ceil(pairs / nwg1) tile pairs: with few tiles per workgroup the busiest one holds well over the mean
    const uint64_t pairs = (pl.n_tiles + 1) / 2, pairs_wg = (pairs + pl.nwg1 - 1) / pl.nwg1;
    const double share1 = std::min(1.0, (double)(2 * pairs_wg) / (double)std::max<uint64_t>(pl.n_tiles, 1));
    // a gated round (frac < 1: only edges touching the round's vertex-hash range are inserted) fills that share of every region: sized for it,
    // a multi-round pass over a huge filter takes half the batches -- and every batch after the first sweeps the whole filter
    const double a_exp = TpcEnv::get().gated_full ? a_max : a_max * std::min(1.0, std::max(frac, 1.0 / 64));
    const double avg1 = a_exp * share1 / (double)(1 << pl.b1);
    const PtPerm pm = pt_make_perm(slice_bits, F);
    pl.perm_mult = pm.mult; pl.perm_inv = pm.inv;
    // tight: the densest bucket's expectation (one of the q addresses of an edge is a function-0 address: pt_bucket_peak), the gate's share
    const double avg1t = avg1 * (1.0 + (pt_bucket_peak(pm, F, pl.b1) - 1.0) / q);
    pl.cap1 = ((uint64_t)(tight ? avg1t + 6 * std::sqrt(avg1t) + 128 : avg1 * 1.3 + 8 * std::sqrt(avg1) + 128) + 31) & ~31ull;
    // A gated round is not uniform over the slices: the vertices of a hash range put the function-0 addresses of their edges into the XOR
    // image of that range (tpc_qpart_plan_sharded: the query's hot slices take ~2.5 x their share).  One of the q addresses of an edge is a
    // function-0 address, so an insert slice of such a round can hold 1 + 1.5 / q times its share; with one or two functions the regions
    // are sized for all the entries, as the query's are.  (sized for the share alone, hot slices of a q <= 2 round overflow
    // the 1.5 x slack; a sharded filter has no direct-kernel fallback behind its overflow list.)
    const bool gated_plan = frac < 1.0;
    const double a_l2 = !gated_plan ? a_exp : q <= 2 ? a_max : std::min(a_max, a_exp * (1.0 + 1.5 / q));
    // a bucket's nwg1 x world level-1 regions are dealt to its wpb level-2 workgroups whole: with fewer regions than workgroups (a few
    // tiles per rank: small inputs, many ranks) some workgroups take one region each and the others none, so a level-2 region holds
    // up to ceil(regions / wpb) / regions of the bucket's entries of its slice, not 1 / wpb
    const uint64_t nvw = (uint64_t)pl.nwg1 * world;
    const double deal = (double)((nvw + pl.wpb - 1) / pl.wpb) / (double)nvw;
    const double avg2 = a_l2 * world * deal / ((double)(1 << pl.b1) * (1 << pl.b2));
    pl.cap2 = ((uint64_t)(avg2 * 1.5 + 8 * std::sqrt(avg2) + 128) + 31) & ~31ull;
    // the last level's entries as planar 24-bit lines (tpc_binsp.h): one rank, two levels, slice offsets of at most 20 bits, bins that
    // do not span waves.  TPC_ENTRY_FMT=legacy (read once per process) keeps the 32-bit entries for A/B measurements.
    // Measured on the 62-genome workload (profiles/r05d_*): the split kernel pays for the two narrow LDS stores per entry what the
    // apply saves in bytes (k_part_split 2.40 -> 2.88 ms, the fused lookup 5.9 -> 5.7), so the format is OFF unless asked for
    // (TPC_ENTRY_FMT=all, read once per process; the tests force it through option "insert_entry_fmt").
    static const bool want_p3 = [] { const char *e = getenv("TPC_ENTRY_FMT"); return e && e[0] == 'a'; }();
    const bool legacy_fmt = !(want_p3 || tpc_test_insert_p3);
    pl.fmt2 = (packed && !legacy_fmt && world == 1 && !three && slice_bits <= 20 && pl.b2 >= 4 && pl.b2 <= 9) ? 3 : 0;
    if (pl.fmt2 == 3) pl.cap2 = (pl.cap2 + PFmt3::GROUP - 1) / PFmt3::GROUP * PFmt3::GROUP;
    pl.wpb3 = 1;
    const double avg3 = a_l2 * world / ((double)(1ull << F) * pl.wpb3);
    pl.cap3 = pl.b3 ? ((uint64_t)(avg3 * 1.5 + 8 * std::sqrt(avg3) + 128) + 31) & ~31ull : 0;
    pl.ovf_cap = (uint64_t)(a_max / 16) + 65536;
    if (tight && tpc_test_tight_pinch > 0) {  // (tests: dist.py's one re-plan with shard_tight_regions = 0)
        pl.cap1 = std::max<uint64_t>(32, ((uint64_t)(avg1t * tpc_test_tight_pinch / 100.0) + 31) & ~31ull);
        pl.ovf_cap = 64;
    }
    return true;
}

size_t tpc_part_buf1_bytes(const TpcPartPlan &pl) { return (size_t)pl.nwg1 * (1u << pl.b1) * pl.cap1 * 4; }
size_t tpc_part_cnt1_bytes(const TpcPartPlan &pl) { return (size_t)pl.nwg1 * (1u << pl.b1) * 4; }
size_t tpc_part_buf2_bytes(const TpcPartPlan &pl)
{
    const size_t regions = (size_t)((1u << pl.b1) / pl.world) * pl.wpb * (1u << pl.b2);
    return pl.fmt2 == 3 ? regions * (pl.cap2 / PFmt3::GROUP) * PT_LINE : regions * pl.cap2 * 4;
}
size_t tpc_part_cnt2_bytes(const TpcPartPlan &pl) { return ((size_t)((1u << pl.b1) / pl.world) * pl.wpb * (1u << pl.b2)) * 4; }
size_t tpc_part_buf3_bytes(const TpcPartPlan &pl) { return pl.b3 ? (((size_t)pl.wpb3 << (pl.b1 + pl.b2 + pl.b3)) / pl.world) * pl.cap3 * 4 : 0; }
size_t tpc_part_cnt3_bytes(const TpcPartPlan &pl) { return pl.b3 ? (((size_t)pl.wpb3 << (pl.b1 + pl.b2 + pl.b3)) / pl.world) * 4 : 0; }

int tpc_launch_insert_part_hash(const TpcLaunch &a, const TpcPartPlan &pl, uint64_t lo, uint64_t hi, bool gated, unsigned long long *n_kmers)
{
    if (a.P.q == 5) return launch_hash_q<5>(a, pl, gated, lo, hi, n_kmers);
#ifdef TPC_DEV_Q5  // development builds: one instantiation
    return -1;
#else
    return tpc_launch_insert_part_hash_other_q(a, pl, lo, hi, gated, n_kmers);  // its own code object (part 1)
#endif
}

int tpc_launch_insert_part_split(const TpcLaunch &a, const TpcPartPlan &pl) { return launch_split(a, pl); }

int tpc_launch_insert_part_apply_only(const TpcLaunch &a, const TpcPartPlan &pl, bool fresh)
{
    const size_t lds = (size_t)4 << (pl.slice_bits - 5);
    const PtPerm perm{pl.slice_bits, pl.b1 + pl.b2 + pl.b3, pl.perm_mult, pl.perm_inv};
    const PtShard sh{pl.rank, pl.world};
    if (pl.fmt2 == 3) {  // planar 24-bit regions (two levels, one rank)
        (void)hipFuncSetAttribute((const void *)k_part_apply<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_part_apply<true>, dim3(tpc_slice_grid((1u << (pl.b1 + pl.b2)) / pl.world)), dim3(PT_APPLY_THREADS), lds, a.stream, pl.slice_bits, pl.b2, pl.wpb,
                           pl.buf2, pl.cnt2, pl.cap2, a.filter, fresh ? 1 : 0, perm, sh, (uint32_t)((1u << (pl.b1 + pl.b2)) / pl.world));
    } else {
        (void)hipFuncSetAttribute((const void *)k_part_apply<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (pl.b3)  // regions of the third level: [(b1, b2)][j][b3]
            hipLaunchKernelGGL(k_part_apply<false>, dim3(tpc_slice_grid((1u << (pl.b1 + pl.b2 + pl.b3)) / pl.world)), dim3(PT_APPLY_THREADS), lds, a.stream, pl.slice_bits, pl.b3, pl.wpb3, pl.buf3,
                               pl.cnt3, pl.cap3, a.filter, fresh ? 1 : 0, perm, sh, (uint32_t)((1u << (pl.b1 + pl.b2 + pl.b3)) / pl.world));
        else
            hipLaunchKernelGGL(k_part_apply<false>, dim3(tpc_slice_grid((1u << (pl.b1 + pl.b2)) / pl.world)), dim3(PT_APPLY_THREADS), lds, a.stream, pl.slice_bits, pl.b2, pl.wpb,
                               pl.buf2, pl.cnt2, pl.cap2, a.filter, fresh ? 1 : 0, perm, sh, (uint32_t)((1u << (pl.b1 + pl.b2)) / pl.world));
    }
    hipLaunchKernelGGL(k_part_ovf, dim3(1024), dim3(256), 0, a.stream, pl.ovf, pl.ovf_cur, pl.ovf_cap, a.filter, perm, sh, pl.b2 + pl.b3);
    return 0;
}

int tpc_launch_insert_part_apply(const TpcLaunch &a, const TpcPartPlan &pl, bool fresh)
{
    int rc;
    if ((rc = launch_split(a, pl))) return rc;
    return tpc_launch_insert_part_apply_only(a, pl, fresh);
}

int tpc_launch_insert_partitioned(const TpcLaunch &a, const TpcPartPlan &pl0, uint64_t lo, uint64_t hi, bool gated, bool fresh,
                                  unsigned long long *n_kmers)
{
    TpcPartPlan pl = pl0;
    pl.rbuf1 = pl.buf1;  // one rank: the split kernel reads what the hash kernel wrote
    pl.rcnt1 = pl.cnt1;
    int rc = tpc_launch_insert_part_hash(a, pl, lo, hi, gated, n_kmers);
    if (rc) return rc;
    return tpc_launch_insert_part_apply(a, pl, fresh);
}

// tpc_preload: the first use of any kernel of this translation unit makes the runtime load its code object
__global__ void k_warm_partition() {}
int tpc_warm_partition() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_warm_partition)) == hipSuccess ? 0 : -1; }

int tpc_launch_region_offsets(const TpcLaunch &a, const uint32_t *cnt, uint32_t n_regions, uint64_t *off)
{
    hipLaunchKernelGGL(k_region_offsets, dim3(1), dim3(1024), 0, a.stream, cnt, n_regions, off);
    return 0;
}

int tpc_launch_region_pack(const TpcLaunch &a, const void *regions, uint64_t cap_entries, uint32_t entry_bytes, const uint32_t *cnt, const uint64_t *off,
                           uint32_t n_regions, void *packed)
{
    if ((entry_bytes != 4 && entry_bytes != 8) || (cap_entries * entry_bytes) % 16) return -1;
    if (n_regions)
        hipLaunchKernelGGL(k_region_pack, dim3(n_regions), dim3(256), 0, a.stream, (const uint4 *)regions, cap_entries * entry_bytes / 16, 16u / entry_bytes, cnt, off,
                           (uint4 *)packed);
    return 0;
}

// list[0..n) -> sorted[0..n) grouped by slice (address >> slice_bits); off[s] .. off[s + 1] = the entries of slice s.
// cnt and cursor: n_slices words each; off: n_slices + 1.
// the same exclusive scan by many workgroups (one took 59 us for the 65536 slice counts of the 62-genome workload's 125 overflow entries, in
// the middle of every step): workgroup g first adds up everything in front of its segment -- a few hundred KB out of the L2 at worst --,
// then scans its own 1024 counts
__global__ void __launch_bounds__(256) k_region_offsets_wide(const uint32_t *__restrict__ cnt, uint32_t n, uint64_t *__restrict__ off)
{
    __shared__ unsigned long long s_w[4];
    __shared__ unsigned long long s_base;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t i0 = blockIdx.x * 1024u, i1 = min(n, i0 + 1024u);
    unsigned long long sum = 0;
    for (uint32_t i = threadIdx.x * 4u; i + 3u < i0; i += 1024u) { const uint4 v = *reinterpret_cast<const uint4 *>(cnt + i); sum += (unsigned long long)v.x + v.y + v.z + v.w; }  // (i0 is a multiple of 1024)
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) s_w[wv] = sum;
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    // my four counts, a wave scan of the per-thread sums, the waves' totals through LDS
    const uint32_t j = i0 + threadIdx.x * 4u;
    uint32_t v[4];
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = j + e < i1 ? cnt[j + e] : 0u;
    const unsigned long long mine = (unsigned long long)v[0] + v[1] + v[2] + v[3];
    unsigned long long inc = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(inc, o, 64);
        if ((int)lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    unsigned long long run = s_base + inc - mine;
    for (uint32_t w = 0; w < wv; w++) run += s_w[w];
#pragma unroll
    for (int e = 0; e < 4; e++) { if (j + e < i1) off[j + e] = run; run += v[e]; }
    if (i1 == n && threadIdx.x == 255) off[n] = s_base + s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

int tpc_launch_ovf_by_slice(const TpcLaunch &a, const uint64_t *list, uint64_t n, int slice_bits, uint32_t n_slices, uint32_t *cnt, uint32_t *cursor,
                            uint64_t *off, uint64_t *sorted, uint32_t rank, uint32_t world, int log_nb2)
{   // world > 1: n_slices local slices, entries of other ranks skipped (off[n_slices] = this rank's entries)
    const PtShard sh{rank, world};
    if (hipMemsetAsync(cnt, 0, (size_t)n_slices * sizeof(uint32_t), a.stream) != hipSuccess ||
        hipMemsetAsync(cursor, 0, (size_t)n_slices * sizeof(uint32_t), a.stream) != hipSuccess) return -1;
    const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
    if (n) hipLaunchKernelGGL(k_ovf_count, dim3(grid), dim3(256), 0, a.stream, list, n, slice_bits, sh, log_nb2, cnt);
    hipLaunchKernelGGL(k_region_offsets_wide, dim3((n_slices + 1023u) / 1024u), dim3(256), 0, a.stream, cnt, n_slices, off);
    if (n) hipLaunchKernelGGL(k_ovf_scatter, dim3(grid), dim3(256), 0, a.stream, list, n, slice_bits, sh, log_nb2, off, cursor, sorted);
    return 0;
}
#endif  // part 0

#if TPC_PARTITION_PART == 1
int tpc_launch_insert_part_hash_other_q(const TpcLaunch &a, const TpcPartPlan &pl, uint64_t lo, uint64_t hi, bool gated, unsigned long long *n_kmers)
{
    switch (a.P.q) {
    case 1: return launch_hash_q<1>(a, pl, gated, lo, hi, n_kmers);
    case 2: return launch_hash_q<2>(a, pl, gated, lo, hi, n_kmers);
    case 3: return launch_hash_q<3>(a, pl, gated, lo, hi, n_kmers);
    case 4: return launch_hash_q<4>(a, pl, gated, lo, hi, n_kmers);
    case 6: return launch_hash_q<6>(a, pl, gated, lo, hi, n_kmers);
    case 7: return launch_hash_q<7>(a, pl, gated, lo, hi, n_kmers);
    case 8: return launch_hash_q<8>(a, pl, gated, lo, hi, n_kmers);
    case 9: return launch_hash_q<9>(a, pl, gated, lo, hi, n_kmers);
    case 10: return launch_hash_q<10>(a, pl, gated, lo, hi, n_kmers);
    case 11: return launch_hash_q<11>(a, pl, gated, lo, hi, n_kmers);
    case 12: return launch_hash_q<12>(a, pl, gated, lo, hi, n_kmers);
    case 13: return launch_hash_q<13>(a, pl, gated, lo, hi, n_kmers);
    case 14: return launch_hash_q<14>(a, pl, gated, lo, hi, n_kmers);
    case 15: return launch_hash_q<15>(a, pl, gated, lo, hi, n_kmers);
    case 16: return launch_hash_q<16>(a, pl, gated, lo, hi, n_kmers);
    }
    return -1;
}
#endif  // part 1
