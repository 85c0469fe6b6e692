// tpc_qpartition.hip -- first-pass query (CandidateCheckingWorker, reference
// src/graphconstructor/vertexenumerator.h:586-704) through the same LDS write-combining bins as
// the partitioned insert.
//
// Why.  The direct kernel (tpc_pass1.hip:k_query) issues ~6 scattered 4-byte loads per k-mer into a
// multi-GiB filter; scattered loads top out at ~50 G/s on MI355X (profiles/r01_microbench.txt), a
// 38 ms floor on the 62-genome workload.  Here the *first* Bloom probe of every unknown edge
// travels to the workgroup that holds its filter slice in LDS:
//
//   A  k_q_hash    rolling hash; for every N-free vertex in the round's range: prev or next is N ->
//                  mark directly (VE.h:640-641); otherwise each unknown edge (c != prev / c != next)
//                  gets its canonical strand and first address a0 -> entry {a0 remainder, edge,
//                  position} binned by the top B1 bits of a0 (uint64 entries, 16 per 128-B line)
//   B  k_q_split   second-level binning by the next B2 bits
//   C  k_q_lookup  a workgroup takes one slice at a time (long-lived: tpc_slice_grid): slice of the filter loaded into LDS once, every entry
//                  tests its bit there; survivors (first probe hit: Bloom false positives ~ fill
//                  rate, plus true second edges) are appended to 64 survivor sub-lists
//   D  k_q_verify  one thread per survivor: recomputes the vertex hashes from the text, probes
//                  functions 1..q-1 directly, and ORs the mark bit.
//
// mark(g) <=> some unknown edge is present: with prev and next definite the known in- and out-edge
// count 1 each, so "inCount > 1 || outCount > 1" (VE.h:656) holds exactly when one more edge passes
// all q probes.  The mask is bit-identical to k_query's (tests/test_gpu_parity.py).
#ifndef TPC_QPARTITION_PART
#define TPC_QPARTITION_PART 0  // csrc/Makefile compiles this file twice: 1 -> tpc_qpartition_x.o, the rarely used kernels (verification for q != 5, the sharded path's routing)
#endif
#include "tpc_rbins.h"
#include "tpc_bins3.h"
#include "tpc_binsp.h"
#include "tpc_lean.h"
#include "tpc_internal.h"
#include "tpc_lists.h"
#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <cmath>

#if TPC_QPARTITION_PART == 0
int tpc_test_q6_pb2 = 0;  // option "test_q6_pb2" (tests)
#endif

namespace {

constexpr int QE_E_SHIFT = 31;   // entry: [0,31) address remainder, [31,34) edge, [34,64) position
constexpr uint64_t QE_REM_MASK = (1ull << QE_E_SHIFT) - 1ull;
constexpr int QS_LISTS = 64;     // survivor sub-lists (independent cursors: same-address atomics serialise)

struct QOverflow {  // entries that did not fit a region: {full address, survivor id} pairs
    uint64_t *list;
    unsigned long long *cursor;  // [0] pairs appended, [1] overflow flag (-> host falls back to k_query)
    uint64_t cap;
    __device__ __forceinline__ void push(uint64_t addr, uint64_t sid, int site = 0) const
    {
#ifdef TPC_PROFILE_PHASES
        atomicAdd(cursor + 24 + site, 1ull);
#endif
        // (one atomic per wave instead of per entry was measured here when gated rounds lost entries by the million: the appends were
        //  not what took the time -- the full rings were, see pl.loads -- and the ballot in this rarely taken path cost the split
        //  kernels 0.2 ms of registers and code on the ungated workload)
        const unsigned long long o = atomicAdd(cursor, 1ull);
        if (o < cap) { list[2 * o] = addr; list[2 * o + 1] = sid; } else cursor[1] = 1ull;
    }
};

__device__ __forceinline__ bool within(uint64_t v, uint64_t lo, uint64_t hi) { return v >= lo && v <= hi; }

// ------------------------------------------------------------------------------------------ A
// Only hash function 0 is rolled here: the first probe of an edge is its function-0 address, the smaller of the
// two strand values (DetermineStrandExtend / Prepend, vertexrollinghash.h:170-200).  When the two values tie the
// reference lets functions 1..q-1 pick the strand, but the function-0 address is the same either way, so nothing
// more is needed at this level; k_q_verify recomputes all q functions, ties included, for the few survivors.
constexpr int QH_THREADS = 1024;  // two threads per packed word: 16 positions each
constexpr int QH_RUN = 16;

__device__ __forceinline__ uint64_t q_rotl_n(uint64_t x, int L, int r)
{   // fastleftshiftn (cyclichash.h:42-44 applied r times)
    if (r == 0) return x;
    return ((x & ((1ull << (L - r)) - 1ull)) << r) | (x >> (L - r));
}

#if TPC_QPARTITION_PART == 0
// RB: barrier-free rings (tpc_rbins.h) instead of the flush-per-round bins.  At 512 bins a ring of the latter holds 32 uint64
// entries, rounds have to be split over thread subsets and the kernel runs at half the rate of the 256-bin geometry; the
// barrier-free rings do not care (tools/bins_bench.hip: 5.1 against 28 ms for 1.86 G entries).  Their pushes must be made by
// whole waves, so this variant computes the candidate edges of every lane and masks the entries instead of branching.
template <bool GATED, bool SHARDED, bool RB>
__global__ void __launch_bounds__(QH_THREADS)
k_q_hash(int LOG_NB, TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases,
         const uint32_t *__restrict__ nmask, uint64_t n_text, uint64_t tile0, uint64_t n_tiles, int pos_per_round, int sub_rounds,
         uint64_t lo, uint64_t hi, uint64_t *buf1, uint32_t *cnt1, uint64_t cap1, QOverflow ovf, PtPerm perm, PtShard sh,
         uint64_t gbase, uint32_t *__restrict__ rmask)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NB = 1 << LOG_NB;
    constexpr int TW = PT_THREADS + 1 + TPC_XW_MAX;  // a tile is still 512 packed words
    typename std::conditional<RB, RBins<uint64_t, QH_THREADS>, Bins3<uint64_t, QH_THREADS>>::type bins;
    uint64_t *s_b = reinterpret_cast<uint64_t *>(bins.carve(smem, LOG_NB));
    uint64_t *s_h = s_b + TW;
    uint64_t *s_hk = s_h + 5;
    uint32_t *s_n = reinterpret_cast<uint32_t *>(s_hk + 5);
    const int tid = threadIdx.x;
    if (tid < 5) { s_h[tid] = tab[tid]; s_hk[tid] = tab[TPC_TAB_HK + tid]; }  // function 0, letters A C G T N
    const int shift = P.L - LOG_NB;
    const uint32_t wg = blockIdx.x, nwg = gridDim.x;
    // one rank: a workgroup's regions are contiguous ([w][b1]); sharded: destination-major (pt_r1_send)
    auto ridx = [sh, NB, wg, nwg](uint32_t b) { return SHARDED ? pt_r1_send(sh, (uint32_t)NB, nwg, wg, b) : (uint64_t)wg * NB + b; };
    auto reg = [buf1, cap1, ridx](uint32_t b) { return PtRegion<uint64_t>{buf1 + ridx(b) * cap1, cap1}; };
    auto lost = [shift, ovf](uint32_t b, uint64_t val) { ovf.push(((uint64_t)b << shift) | (val & QE_REM_MASK), val >> QE_E_SHIFT, 1); };
    if constexpr (RB) bins.init();
    else bins.init(buf1, [cap1, ridx](uint32_t b) { return make_uint2((uint32_t)((ridx(b) * cap1) >> 4), (uint32_t)cap1); });  // 16 entries = one 128-byte unit
    const int xw = (P.k + 1) / 32 + 2;
    uint16_t *rmask16 = reinterpret_cast<uint16_t *>(rmask);
    // function 0's table entries of the four letters, as scalars (uniform loads): the eight candidate edges of a
    // position use them with constant letters, and LDS reads could not be hoisted over the ring traffic
    uint64_t h0[4], hk0[4];
#pragma unroll
    for (int c = 0; c < 4; c++) { h0[c] = tab[c]; hk0[c] = tab[TPC_TAB_HK + c]; }
    for (uint64_t tile = tile0 + blockIdx.x; tile < tile0 + n_tiles; tile += gridDim.x) {
        __syncthreads();
        const uint64_t wfirst = tile * PT_THREADS;
        const uint64_t wbase = wfirst - 1;
        for (int i = tid; i < PT_THREADS + 1 + xw; i += QH_THREADS) {
            const int64_t w = (int64_t)wfirst - 1 + i;
            s_b[i] = w >= 0 ? bases[w] : 0ull;
            s_n[i] = w >= 0 ? nmask[w] : 0xFFFFFFFFu;
        }
        __syncthreads();
        const uint64_t g0 = wfirst * TPC_RUN + (uint64_t)tid * QH_RUN;
        const bool active = g0 < n_text;
        TpcVHash<1> v;  // function 0 only
        v.pos[0] = 0; v.neg[0] = 0;
        int ncnt = 0, c_prev = TPC_CODE_N, c_first = TPC_CODE_N;
        uint32_t word = 0;
        if (active) {
            tpc_vhash_init<1>(v, P, s_h, s_b, s_n, g0, wbase);
            for (int t = 0; t < P.k; t++) ncnt += tpc_tile_char(s_b, s_n, g0 + t, wbase) == TPC_CODE_N;
            c_prev = tpc_tile_char(s_b, s_n, g0 - 1, wbase);
            c_first = tpc_tile_char(s_b, s_n, g0, wbase);
        }
        // a round = pos_per_round positions per thread, or (many small bins) one position for every
        // sub_rounds-th thread, so that a round never outgrows the rings
        if constexpr (RB) {
            for (int s = 0; s < QH_RUN; s++) {  // wave-uniform: every lane reaches every push
                const uint64_t g = g0 + s;
                const int c_next = tpc_tile_char(s_b, s_n, g + P.k, wbase);
                const int c_first_nx = tpc_tile_char(s_b, s_n, g + 1, wbase);
                const uint64_t r1p = tpc_rotl1(v.pos[0], P.L, P.lmask);
                const uint64_t r1n = tpc_rotl1(v.neg[0], P.L, P.lmask);
                bool check = active && ncnt == 0;
                if (GATED) check = check && within(tpc_min(v.pos[0], v.neg[0]), lo, hi);  // VE.h:638
                const bool nadj = c_prev == TPC_CODE_N || c_next == TPC_CODE_N;
                if (check && nadj) word |= 1u << s;  // VE.h:640-641: an N neighbour counts 2
                const bool probe = check && !nadj;
                const uint64_t sid_g = ((g - gbase) | (SHARDED ? (uint64_t)sh.rank << (30u - sh.log_world()) : 0ull)) << 3;
                uint32_t eb[8];
                uint64_t ev[8];
                bool eok[8];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    {   // in-edge c + v (DetermineStrandPrepend, vertexrollinghash.h:186-200)
                        const uint64_t a0 = perm.fwd(tpc_min(hk0[c] ^ v.pos[0], r1n ^ h0[3 - c]));
                        eb[c] = (uint32_t)(a0 >> shift);
                        ev[c] = (a0 & (((uint64_t)1 << shift) - 1)) | ((sid_g | (uint64_t)c) << QE_E_SHIFT);
                        eok[c] = probe && c != c_prev;
                    }
                    {   // out-edge v + c (DetermineStrandExtend, vertexrollinghash.h:170-184)
                        const uint64_t a0 = perm.fwd(tpc_min(r1p ^ h0[c], v.neg[0] ^ hk0[3 - c]));
                        eb[4 + c] = (uint32_t)(a0 >> shift);
                        ev[4 + c] = (a0 & (((uint64_t)1 << shift) - 1)) | ((sid_g | (uint64_t)(4 + c)) << QE_E_SHIFT);
                        eok[4 + c] = probe && c != c_next;
                    }
                }
                bins.template push_batch<8>(eb, ev, eok, reg, lost);
                // roll: VertexRollingHash::Update (vertexrollinghash.h:104-113), function 0
                v.pos[0] = r1p ^ s_h[c_next] ^ s_hk[c_first];
                v.neg[0] = tpc_rotr1(v.neg[0] ^ s_hk[tpc_rc(c_next)] ^ s_h[tpc_rc(c_first)], P.L);
                ncnt += (c_next == TPC_CODE_N) - (c_first == TPC_CODE_N);
                c_prev = c_first;
                c_first = c_first_nx;
            }
        } else
        for (int s0 = 0; s0 < QH_RUN * sub_rounds; s0 += pos_per_round) {
            if (active && (sub_rounds == 1 || (tid % sub_rounds) == (s0 % sub_rounds))) {
                for (int s = s0 / sub_rounds; s < s0 / sub_rounds + pos_per_round; s++) {
                    const uint64_t g = g0 + s;
                    const int c_next = tpc_tile_char(s_b, s_n, g + P.k, wbase);
                    const int c_first_nx = tpc_tile_char(s_b, s_n, g + 1, wbase);
                    const uint64_t r1p = tpc_rotl1(v.pos[0], P.L, P.lmask);
                    const uint64_t r1n = tpc_rotl1(v.neg[0], P.L, P.lmask);
                    bool check = ncnt == 0;
                    if (GATED) check = check && within(tpc_min(v.pos[0], v.neg[0]), lo, hi);  // VE.h:638
                    if (check) {
                        if (c_prev == TPC_CODE_N || c_next == TPC_CODE_N) {
                            word |= 1u << s;  // VE.h:640-641: an N neighbour counts 2
                        } else {
                            // position relative to the batch; sharded: the source rank rides in the top log2(world) bits of the 30
                            const uint64_t sid_g = ((g - gbase) | (SHARDED ? (uint64_t)sh.rank << (30u - sh.log_world()) : 0ull)) << 3;
                            uint32_t eb[8];
                            uint64_t ev[8];
                            bool eok[8];
#pragma unroll
                            for (int c = 0; c < 4; c++) {
                                {   // in-edge c + v (DetermineStrandPrepend, vertexrollinghash.h:186-200)
                                    const uint64_t a0 = perm.fwd(tpc_min(hk0[c] ^ v.pos[0], r1n ^ h0[3 - c]));  // permuted address from here on
                                    eb[c] = (uint32_t)(a0 >> shift);
                                    ev[c] = (a0 & (((uint64_t)1 << shift) - 1)) | ((sid_g | (uint64_t)c) << QE_E_SHIFT);
                                    eok[c] = c != c_prev;
                                }
                                {   // out-edge v + c (DetermineStrandExtend, vertexrollinghash.h:170-184)
                                    const uint64_t a0 = perm.fwd(tpc_min(r1p ^ h0[c], v.neg[0] ^ hk0[3 - c]));
                                    eb[4 + c] = (uint32_t)(a0 >> shift);
                                    ev[4 + c] = (a0 & (((uint64_t)1 << shift) - 1)) | ((sid_g | (uint64_t)(4 + c)) << QE_E_SHIFT);
                                    eok[4 + c] = c != c_next;
                                }
                            }
                            bins.template push_batch<8>(eb, ev, eok, lost);
                        }
                    }
                    // roll: VertexRollingHash::Update (vertexrollinghash.h:104-113), function 0
                    v.pos[0] = r1p ^ s_h[c_next] ^ s_hk[c_first];
                    v.neg[0] = tpc_rotr1(v.neg[0] ^ s_hk[tpc_rc(c_next)] ^ s_h[tpc_rc(c_first)], P.L);
                    ncnt += (c_next == TPC_CODE_N) - (c_first == TPC_CODE_N);
                    c_prev = c_first;
                    c_first = c_first_nx;
                }
            }
            bins.template flush<false>(lost);
        }
        rmask16[(wfirst * 2) + tid] = (uint16_t)word;  // N-neighbour marks (16 positions per thread); k_q_verify ORs the rest
    }
    if constexpr (RB) { bins.flush(true, reg, lost); bins.store_counts(cnt1, reg, ridx); }
    else { bins.template flush<true>(lost); bins.store_counts(cnt1, ridx); }
}

// ------------------------------------------------------------------------------------------ A, instruction-lean
// The same level-1 pass as k_q_hash<.., RB = false>, rebuilt around the VALU instruction count (tpc_lean.h): the 16 first and
// 16 next characters of a thread's run sit in two registers, rotations and the address split work on 32-bit halves, the run
// is seeded from a table of pre-rotated letter hashes (one 16-byte LDS read per character: H(w) = XOR_t rotl(h[w_t], k-1-t),
// H'(w) = XOR_t rotl(h[rc w_t], t); cyclichash.h:106-109 folded), the roll reads {h[c], hk[rc c]} / {hk[c], h[rc c]} as
// 16-byte pairs, and the bins are Bins3.  Used whenever the geometry allows (launch_qhash); entries, masks and overflow
// entries are those of k_q_hash (order inside a region differs: nothing downstream depends on it).
constexpr int QT_MAXK = 64;  // seed table: k x 5 letters x 16 bytes

// every mask bit doubled: bit i -> bits 2i, 2i+1
__device__ __forceinline__ uint64_t q_spread2(uint32_t m)
{
    uint64_t x = m;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x | (x << 1);
}

template <bool GATED, bool SHARDED, bool HALF, bool LHI>
__global__ void __launch_bounds__(QH_THREADS)
k_q_hash2(int LOG_NB, TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases,
          const uint32_t *__restrict__ nmask, uint64_t n_text, uint64_t tile0, uint64_t n_tiles, int pos_per_round,
          uint64_t lo, uint64_t hi, uint64_t *buf1, uint32_t *cnt1, uint64_t cap1, QOverflow ovf, PtPerm perm, PtShard sh,
          uint64_t gbase, uint32_t *__restrict__ rmask, uint32_t tiles_per_wg, const uint16_t *__restrict__ skip16)
{   // LHI: L > 32; L-bit values on two separate 32-bit registers (LeanV, tpc_lean.h: round 4)
    // skip16 (k_periodic_build's per_qs, or nullptr): positions whose k + 2 characters repeat those of the position 1 .. 63 before
    // them send no probes -- k_periodic_copy gives them that position's verdict after the verification
    // tiles_per_wg > 0: workgroup w takes the tiles [w T, (w + 1) T) of the batch instead of w, w + nwg, ...: its regions then hold
    // ascending positions, which the 6-byte level-2 entries rely on (k_q_split<.., P6>)
    using V = LeanV<LHI>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NB = 1 << LOG_NB;
    constexpr int TW = PT_THREADS + 1 + TPC_XW_MAX;  // a tile is 512 packed words
    Bins3<uint64_t, QH_THREADS> bins;
    uint64_t *s_b = reinterpret_cast<uint64_t *>(bins.carve(smem, LOG_NB));  // [TW] bases, N positions cleared to code 0
    uint64_t *s_roll = s_b + TW;                   // [2][5][2]: as next character {h[c], hk[rc c]}, as first character {hk[c], h[rc c]}
    uint64_t *s_seed = s_roll + 20;                // [k][5][2] (k <= QT_MAXK): {rotl(h[c], k-1-t), rotl(h[rc c], t)}
    uint32_t *s_n = reinterpret_cast<uint32_t *>(s_seed + QT_MAXK * 10);  // [TW]
    const uint32_t tid = threadIdx.x;
    const int k = P.k, L = P.L;
    if (tid < 5) {  // function 0, letters A C G T N
        const uint32_t c = tid, rc = c == 4 ? 4u : 3u - c;
        s_roll[c * 2] = tab[c]; s_roll[c * 2 + 1] = tab[TPC_TAB_HK + rc];
        s_roll[10 + c * 2] = tab[TPC_TAB_HK + c]; s_roll[10 + c * 2 + 1] = tab[rc];
    }
    if (k <= QT_MAXK)
        for (uint32_t i = tid; i < 5u * (uint32_t)k; i += QH_THREADS) {
            const uint32_t t = i / 5u, c = i % 5u, rc = c == 4 ? 4u : 3u - c;
            s_seed[2 * i] = q_rotl_n(tab[c], L, (k - 1 - (int)t) % L);
            s_seed[2 * i + 1] = q_rotl_n(tab[rc], L, (int)t % L);
        }
    const int shift = L - LOG_NB;
    const uint32_t wg = blockIdx.x, nwg = gridDim.x;
    auto ridx = [sh, NB, wg, nwg](uint32_t b) { return SHARDED ? pt_r1_send(sh, (uint32_t)NB, nwg, wg, b) : (uint64_t)wg * NB + b; };
    auto lost = [shift, ovf](uint32_t b, uint64_t val) { ovf.push(((uint64_t)b << shift) | (val & QE_REM_MASK), val >> QE_E_SHIFT, 1); };
    bins.init(buf1, [cap1, ridx](uint32_t b) { return make_uint2((uint32_t)((ridx(b) * cap1) >> 4), (uint32_t)cap1); });  // 16 entries = one 128-byte unit
    LeanRotH<LHI> R;
    R.set(L);
    LeanSplit S;
    S.set(perm, LOG_NB);
    const int xw = (k + 1) / 32 + 2;
    uint16_t *rmask16 = reinterpret_cast<uint16_t *>(rmask);
    // function 0's table entries of the four letters as scalars: the eight candidate edges use them with constant letters
    uint32_t h0l[4], h0h[4], hk0l[4], hk0h[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint64_t h = tab[c], hk = tab[TPC_TAB_HK + c];
        h0l[c] = (uint32_t)h; h0h[c] = (uint32_t)(h >> 32); hk0l[c] = (uint32_t)hk; hk0h[c] = (uint32_t)(hk >> 32);
    }
    const uint32_t ppr_mask = (uint32_t)pos_per_round - 1u;  // a power of two <= 16
    constexpr bool half_rounds = HALF;                       // 512 bins: see the push below
    const uint32_t p0 = 32u + tid * (uint32_t)QH_RUN;        // my first position, relative to the first staged word (the one before the tile)
    const uint64_t t_first = tile0 + (tiles_per_wg ? (uint64_t)blockIdx.x * tiles_per_wg : (uint64_t)blockIdx.x);
    const uint64_t t_end = tiles_per_wg ? min(tile0 + n_tiles, t_first + tiles_per_wg) : tile0 + n_tiles;
    const uint64_t t_step = tiles_per_wg ? 1u : gridDim.x;
    for (uint64_t tile = t_first; tile < t_end; tile += t_step) {
        const uint64_t wfirst = tile * PT_THREADS;
        __syncthreads();
        const uint32_t sk = skip16 ? (uint32_t)skip16[(wfirst * 2) + tid] : 0u;  // (issued with the staging loads below: its round trip hides behind theirs)
        for (int i = (int)tid; i < PT_THREADS + 1 + xw; i += QH_THREADS) {
            const int64_t w = (int64_t)wfirst - 1 + i;
            uint64_t b = w >= 0 ? bases[w] : 0ull;
            const uint32_t m = w >= 0 ? nmask[w] : 0xFFFFFFFFu;
            if (m) b &= ~q_spread2(m);  // an N has code 0 in the staged word: code = c | isN << 2 below
            s_b[i] = b;
            s_n[i] = m;
        }
        __syncthreads();
        // The text is padded with N to whole tiles (tpc_seq_upload), so runs past its end find no vertex and need no guard.
        const uint32_t cw = lean_chars16(s_b, p0), nw = lean_nbits32(s_n, p0);                            // first characters of my 16 windows
        const uint32_t cx = lean_chars16(s_b, p0 + (uint32_t)k), nx = lean_nbits32(s_n, p0 + (uint32_t)k);  // the characters after them
        uint32_t cp = lean_char(s_b, s_n, p0 - 1u);  // character before the window
        int ncnt = 0;                                // N characters inside the window
        for (int t = 0; t < k; t += 32) {
            uint32_t bits = lean_nbits32(s_n, p0 + (uint32_t)t);
            if (k - t < 32) bits &= (1u << (k - t)) - 1u;
            ncnt += __popc(bits);
        }
        V pos = lv_make<LHI>(0u, 0u), neg = lv_make<LHI>(0u, 0u);  // VertexRollingHash ctor (vertexrollinghash.h:79-102), function 0
        if (k <= QT_MAXK) {
            for (int t0 = 0; t0 < k; t0 += 16) {
                uint32_t ch = lean_chars16(s_b, p0 + (uint32_t)t0), nb = lean_nbits32(s_n, p0 + (uint32_t)t0);
                const int m = min(16, k - t0);
                const uint4 *row = reinterpret_cast<const uint4 *>(s_seed) + t0 * 5;
                for (int j = 0; j < m; j++) {
                    const uint32_t c = (ch & 3u) | ((nb & 1u) << 2);
                    ch >>= 2; nb >>= 1;
                    const uint4 e = row[j * 5 + c];
                    pos = lv_xor<LHI>(pos, e.x, e.y);
                    neg = lv_xor<LHI>(neg, e.z, e.w);
                }
            }
        } else {
            for (int t = 0; t < k; t++) {
                const uint32_t c = lean_char(s_b, s_n, p0 + (uint32_t)t), cr = lean_char(s_b, s_n, p0 + (uint32_t)(k - 1 - t));
                pos = lv_xor<LHI>(R.rotl1(pos), lv_from64<LHI>(s_roll[c * 2]));
                neg = lv_xor<LHI>(R.rotl1(neg), lv_from64<LHI>(s_roll[10 + cr * 2 + 1]));
            }
        }
        const uint32_t sid0 = ((uint32_t)(wfirst * TPC_RUN + (uint64_t)tid * QH_RUN - gbase) | (SHARDED ? sh.rank << (30u - sh.log_world()) : 0u)) << 2;
        uint32_t word = 0;
#pragma unroll 1
        for (int s = 0; s < QH_RUN; s++) {  // not unrolled: one copy of the push and flush code (the loop body is ~600 instructions)
            const uint32_t cf = __builtin_amdgcn_ubfe(cw, 2u * (uint32_t)s, 2u) | (__builtin_amdgcn_ubfe(nw, (uint32_t)s, 1u) << 2);  // first character of the window
            const uint32_t cn = __builtin_amdgcn_ubfe(cx, 2u * (uint32_t)s, 2u) | (__builtin_amdgcn_ubfe(nx, (uint32_t)s, 1u) << 2);  // the character after it
            const V r1p = R.rotl1(pos), r1n = R.rotl1(neg);
            bool check = ncnt == 0;
            if (GATED) check = check && within(lv_u64<LHI>(lv_min<LHI>(pos, neg)), lo, hi);  // VE.h:638
            const bool nadj = (cp | cn) >= 4u;
            if (check && nadj) word |= 1u << s;  // VE.h:640-641: an N neighbour counts 2
            uint32_t eb[8];
            uint64_t ev[8];
            bool eok[8];
            bool probing = false;
            if (check && !nadj && !__builtin_amdgcn_ubfe(sk, (uint32_t)s, 1u)) {
                const uint32_t hi_s = sid0 + (uint32_t)(s << 2);  // (survivor id >> 1) without the edge: position << 2
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    uint32_t rem;
                    // in-edge c + v (DetermineStrandPrepend, vertexrollinghash.h:186-200): survivor id edge = c
                    lean_split_h<LHI>(S, lv_min<LHI>(lv_xor<LHI>(pos, hk0l[c], hk0h[c]), lv_xor<LHI>(r1n, h0l[3 - c], h0h[3 - c])), eb[c], rem);
                    ev[c] = ((uint64_t)(hi_s | (uint32_t)(c >> 1)) << 32) | (rem | ((uint32_t)(c & 1) << 31));
                    eok[c] = (uint32_t)c != cp;
                    // out-edge v + c (DetermineStrandExtend, vertexrollinghash.h:170-184): edge = 4 + c
                    lean_split_h<LHI>(S, lv_min<LHI>(lv_xor<LHI>(r1p, h0l[c], h0h[c]), lv_xor<LHI>(neg, hk0l[3 - c], hk0h[3 - c])), eb[4 + c], rem);
                    ev[4 + c] = ((uint64_t)(hi_s | (uint32_t)((4 + c) >> 1)) << 32) | (rem | ((uint32_t)(c & 1) << 31));
                    eok[4 + c] = (uint32_t)c != cn;
                }
                if constexpr (!half_rounds) bins.template push_batch<8>(eb, ev, eok, lost);
                else {  // 512 bins: a ring holds 32 entries, so the in-edges and the out-edges of a position go in two rounds
                    uint32_t b4[4];
                    uint64_t v4[4];
                    bool o4[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) { b4[c] = eb[c]; v4[c] = ev[c]; o4[c] = eok[c]; }
                    bins.template push_batch<4>(b4, v4, o4, lost);
                }
                probing = true;
            }
            if constexpr (half_rounds) {  // (uniform: every lane reaches both flushes of the position)
                bins.template flush<false>(lost);
                if (probing) {
                    uint32_t b4[4];
                    uint64_t v4[4];
                    bool o4[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) { b4[c] = eb[4 + c]; v4[c] = ev[4 + c]; o4[c] = eok[4 + c]; }
                    bins.template push_batch<4>(b4, v4, o4, lost);
                }
            }
            if (s + 1 < QH_RUN) {  // roll: VertexRollingHash::Update (vertexrollinghash.h:104-113), function 0
                const uint4 en = reinterpret_cast<const uint4 *>(s_roll)[cn], ef = reinterpret_cast<const uint4 *>(s_roll)[5 + cf];
                pos = lv_xor<LHI>(lv_xor<LHI>(r1p, en.x, en.y), ef.x, ef.y);
                neg = R.rotr1(lv_xor<LHI>(lv_xor<LHI>(neg, en.z, en.w), ef.z, ef.w));
                ncnt += (int)(cn >> 2) - (int)(cf >> 2);
                cp = cf;
            }
            if ((((uint32_t)s + 1u) & ppr_mask) == 0u) bins.template flush<false>(lost);
        }
        rmask16[(wfirst * 2) + tid] = (uint16_t)word;  // N-neighbour marks (16 positions per thread); k_q_verify ORs the rest
    }
    bins.template flush<true>(lost);
    bins.store_counts(cnt1, ridx);
}

// ------------------------------------------------------------------------------------------ B
constexpr int QS_THREADS = 1024;  // split: 16 waves hide the LDS atomic round trips better than 8
// P6 (round 5, one rank, last level, flush-per-round bins): the OUTPUT regions are blocked lines of 20 x 48-bit entries (tpc_binsp.h:
// PFmt6; off2 counts 128-byte lines, cnt2 is exact) instead of 8-byte entries:
//     { slice offset S = slice_bits | edge 3 | low PB2 = 44 - S bits of the position | parity of the position's group }
// The group -- the position's bits above PB2 -- is implicit in where the entry lies.  Level 1 hashed the text in contiguous blocks of
// tiles_per_wg tiles per workgroup (k_q_hash2), this kernel streams the regions (w, b1) in ascending w and each region in the order
// it was written, so the positions of the stream ascend region by region and a region spans at most two groups (the plan makes
// sure of tiles_per_wg tiles <= 2^PB2 positions).  Before the first round of every region that may reach a group g not seen so far
// the flush records how many entries each output region holds: bnd[g].  An entry at index i of its region then lies in zone
// z = #{g >= 1 : bnd[g] <= i}, which holds entries of groups z - 1 and z only -- the parity bit says which (q6_sid).
template <bool SHARDED, bool RB, bool P6 = false>
__global__ void __launch_bounds__(QS_THREADS)
k_q_split(int LOG_NB1, int LOG_NB2, int L, int slice_bits, int loads, uint32_t nwg1, uint32_t wpb, const uint64_t *__restrict__ buf1,
          const uint32_t *__restrict__ cnt1, uint64_t cap1, uint64_t *buf2, uint32_t *cnt2, const uint64_t *__restrict__ off2, QOverflow ovf,
          PtShard sh, uint32_t prev_wpb, int log_prev_nb2, uint32_t nreg_cap, uint32_t sched_cap, const uint64_t *__restrict__ off1,
          const uint64_t *__restrict__ own1, const uint32_t *__restrict__ owncnt1, uint32_t *__restrict__ bnd = nullptr, uint32_t n_groups = 0,
          uint32_t tiles_per_wg = 0, uint32_t batch_tiles = 0, uint32_t pb2 = 0)
{   // own1 / owncnt1 (sharded, optional): the block of source rank == this rank is read from the send buffers it was hashed into
    // prev_wpb > 0 (three-level geometry): this bucket is (b1, b2) of an earlier k_q_split whose regions [b1][j][b2] are the input
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t NB1 = 1u << LOG_NB1, NB2 = 1u << LOG_NB2;
    constexpr int LOADS = P6 ? 5 : 4;  // (6-byte entries: a ring holds 80 instead of 64)
    constexpr uint64_t SENT = ~0ull;
    static_assert(!P6 || (!SHARDED && !RB), "the 6-byte output is the one-rank, flush-per-round form");
    typename std::conditional<P6, BinsP<PFmt6, QS_THREADS>,
                              typename std::conditional<RB, RBins<uint64_t, QS_THREADS>, Bins3<uint64_t, QS_THREADS>>::type>::type bins;  // RB: see k_q_hash
    uint64_t *s_off = reinterpret_cast<uint64_t *>(bins.carve(smem, LOG_NB2));  // [NB2 + 1] region offsets of this workgroup
    if constexpr (RB) bins.init();
    else {
        const uint64_t *o2 = off2 + (uint64_t)blockIdx.x * NB2;
        if constexpr (P6) bins.init(buf2, [o2](uint32_t b) { const uint64_t o = o2[b]; return make_uint2((uint32_t)o, (uint32_t)(o2[b + 1] - o)); });  // lines
        else bins.init(buf2, [o2](uint32_t b) { const uint64_t o = o2[b]; return make_uint2((uint32_t)(o >> 4), (uint32_t)(o2[b + 1] - o)); });
    }
    for (uint32_t i = threadIdx.x; i <= NB2; i += QS_THREADS) s_off[i] = off2[(uint64_t)blockIdx.x * NB2 + i];
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
    const int shift1 = L - LOG_NB1;
    const uint64_t rem_mask = ((uint64_t)1 << shift1) - 1;
    // level-2 regions are sized per filter slice (function-0 addresses are denser in low slices): off2
    auto reg = [buf2, s_off](uint32_t b) { const uint64_t o = s_off[b]; return PtRegion<uint64_t>{buf2 + o, s_off[b + 1] - o}; };
    // P6: entry layout and the group bookkeeping (all uniform)
    const uint32_t S6 = (uint32_t)slice_bits, PB2 = pb2;  // <= min(44 - S6, 30): a level-1 entry holds a 30-bit position
    uint32_t next_g = 1;             // groups below this one have their boundary
    uint32_t *my_bnd = P6 ? bnd + (uint64_t)blockIdx.x * NB2 * n_groups * 2u : nullptr;  // per region: n_groups zone starts, then n_groups zone ends
    // highest group a region of level-1 workgroup w can hold (its tiles: [w T, (w + 1) T) of the batch)
    auto gmax = [=](uint32_t w) { return (uint32_t)((min((uint64_t)(w + 1u) * tiles_per_wg, (uint64_t)batch_tiles) * (uint64_t)(PT_THREADS * TPC_RUN) - 1ull) >> PB2); };
    // (P6) An entry that found no room -- its ring full at the push, or its region full at a flush -> the overflow list's {full
    // permuted address, survivor id = edge | position << 3}.  Every snapshot empties the rings (below), so whatever sits in a ring
    // belongs to the zone at hand, next_g - 1, whose entries are of that group or the one below: the parity bit says which.
    // (both handlers: one entry at a time from a push, a whole ring group with ONE reservation from a flush -- tpc_bins3.h:pt_bulk)
    struct Lost6 {
        QOverflow ovf; const uint32_t *next_g; uint32_t b1, S6, PB2; int shift1;
        __device__ __forceinline__ void pair(uint32_t b2, uint64_t val, uint64_t &addr, uint64_t &sid) const
        {
            const uint32_t z = *next_g - 1u;
            const uint64_t plow = (val >> (S6 + 3u)) & ((1ull << PB2) - 1ull);
            const uint64_t g = (z & 1u) == ((uint32_t)(val >> 47) & 1u) ? z : z - 1u;
            addr = ((uint64_t)b1 << shift1) | ((uint64_t)b2 << S6) | (val & ((1ull << S6) - 1ull));
            sid = ((val >> S6) & 7ull) | (((g << PB2) | plow) << 3);
        }
        __device__ __forceinline__ void operator()(uint32_t b2, uint64_t val) const { uint64_t a, i; pair(b2, val, a, i); ovf.push(a, i, 3); }
        __device__ __forceinline__ unsigned long long reserve(uint32_t n) const { return atomicAdd(ovf.cursor, (unsigned long long)n); }
        __device__ __forceinline__ void put(uint32_t b2, uint64_t val, unsigned long long at) const
        {
            uint64_t a, i;
            pair(b2, val, a, i);
            if (at < ovf.cap) { ovf.list[2 * at] = a; ovf.list[2 * at + 1] = i; } else ovf.cursor[1] = 1ull;
        }
    };
    const Lost6 lost_p6{ovf, &next_g, b1, S6, PB2, shift1};
    struct Lost8 {
        QOverflow ovf; uint32_t b1; int shift1; uint64_t rem_mask;
        __device__ __forceinline__ void operator()(uint32_t, uint64_t val) const { ovf.push(((uint64_t)b1 << shift1) | (val & rem_mask), val >> QE_E_SHIFT, 3); }
        __device__ __forceinline__ unsigned long long reserve(uint32_t n) const { return atomicAdd(ovf.cursor, (unsigned long long)n); }
        __device__ __forceinline__ void put(uint32_t, uint64_t val, unsigned long long at) const
        {
            if (at < ovf.cap) { ovf.list[2 * at] = ((uint64_t)b1 << shift1) | (val & rem_mask); ovf.list[2 * at + 1] = val >> QE_E_SHIFT; } else ovf.cursor[1] = 1ull;
        }
    };
    const Lost8 lost{ovf, b1, shift1, rem_mask};
    // (P6) the snapshot: a FINAL-type flush -- every bin's last line goes out partly filled, the next entry starts a new line -- that
    // records, for the groups [lo, hi) it opens, where their zone starts (bnd, a whole number of lines) and, for the zone it closes,
    // where its entries end (vend: what follows up to the next line is garbage).  No entry ever waits in a ring across a boundary.
    auto snapshot = [&](uint32_t lo, uint32_t hi) {
      if constexpr (P6) {
        const uint32_t ng = n_groups;
        uint32_t *mb0 = my_bnd;  // per region: n_groups zone starts, then n_groups zone ends
        bins.template flush_with<true, false>(lost_p6,
            [lo, hi, ng, mb0](uint32_t b, uint32_t n) {
                const uint32_t padded = (n + (uint32_t)PFmt6::GROUP - 1u) / (uint32_t)PFmt6::GROUP * (uint32_t)PFmt6::GROUP;
                uint32_t *mb = mb0 + (uint64_t)b * 2u * ng, *mv = mb + ng;
                mv[lo - 1u] = n;
                for (uint32_t g = lo; g < hi; g++) { mb[g] = padded; mv[g] = padded; }  // (the last of them is the zone now open: its end follows)
            });
        next_g = hi;
      }
    };
    __syncthreads();
    // rounds of `loads` x QS_THREADS entries over the source regions (j, j + wpb, ...), taken from the round schedule; the
    // loads of the next TWO rounds are in flight while a round is binned and flushed (one round ahead left ~32 KB per CU
    // outstanding, short of what HBM latency needs).  Every round issues exactly LOADS unpredicated loads (lanes past the
    // end read entry 0 and are masked when the round is consumed -- a select right after the load would make its result
    // needed at once) and the three buffers rotate by name, not by copies.
    const uint32_t nreg = j < nvw ? (nvw - j + wpb - 1) / wpb : 0;
    const uint32_t step = (uint32_t)loads * QS_THREADS;
    uint32_t *s_scan = reinterpret_cast<uint32_t *>(s_off + NB2 + 1);  // [32] scratch of the schedule's block scan
    uint32_t *s_cnt = s_scan + 32;                                     // [nreg_cap]
    uint32_t *s_sched = s_cnt + nreg_cap;                             // [sched_cap]
    struct Round { uint32_t t, base, n; };  // region (j + t * wpb), first entry of the round, entries in the region; all scalar
    uint64_t va[LOADS], vb[LOADS], vc[LOADS];
    for (uint32_t skip = 0;; skip += sched_cap) {
        const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)pt_build_schedule<QS_THREADS>(
            nreg, step, skip, sched_cap, s_cnt, s_sched, s_scan, [&](uint32_t t) { const uint32_t vw = j + t * wpb; return (mine(vw) ? owncnt1 : cnt1)[r1(vw)]; }));
        const uint32_t n_seg = min(total - min(total, skip), sched_cap);
        auto round_at = [&](uint32_t r) {
            const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_sched[min(r, n_seg - 1u)]);
            Round x{e & 0xFFFFu, (e >> 16) * step, 0u};
            if (r < n_seg) x.n = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cnt[x.t]);
            return x;
        };
        auto valid = [&](const Round &x, int i) { return i < loads && x.base + i * QS_THREADS + threadIdx.x < x.n; };
        auto load = [&](uint64_t (&dst)[LOADS], const Round &x) {
            const uint32_t vw = j + x.t * wpb;
            const uint64_t ri = r1(vw);
            const uint64_t *src = (mine(vw) ? own1 : buf1) + (off1 ? off1[ri] : ri * cap1);  // off1: the regions arrived packed (compacted exchange)
#pragma unroll
            for (int i = 0; i < LOADS; i++) dst[i] = src[valid(x, i) ? x.base + i * QS_THREADS + threadIdx.x : 0u];
        };
        if (n_seg) {
            Round x0 = round_at(0), x1 = round_at(1);
            load(va, x0);
            load(vb, x1);
            uint32_t r = 0;
            if constexpr (P6) {  // a segment that opens with the first round of a region which may hold groups without a boundary yet
                if (x0.base == 0u) {  // (the very first region among them: boundary 0)
                    const uint32_t gm = min(gmax(j + x0.t * wpb), n_groups - 1u);
                    if (gm >= next_g) snapshot(next_g, gm + 1u);
                }
            }
            auto round = [&](uint64_t (&cur)[LOADS], uint64_t (&pre)[LOADS]) {
                const Round x2 = round_at(r + 2);
                load(pre, x2);
                uint32_t bb[LOADS];
                bool ok[LOADS];
#pragma unroll
                for (int i = 0; i < LOADS; i++) { ok[i] = valid(x0, i) && cur[i] != SENT; bb[i] = (uint32_t)((cur[i] & rem_mask) >> slice_bits); }
                if constexpr (P6) {
                    uint64_t v6[LOADS];
#pragma unroll
                    for (int i = 0; i < LOADS; i++) {
                        const uint32_t pos = (uint32_t)(cur[i] >> 34);  // 30 bits, batch relative
                        const uint64_t id = (uint64_t)((uint32_t)(cur[i] >> QE_E_SHIFT) & 7u) | ((uint64_t)(pos & (uint32_t)((1ull << PB2) - 1ull)) << 3);  // edge | low position bits
                        v6[i] = (id << S6) | ((uint32_t)cur[i] & ((1u << S6) - 1u)) | ((uint64_t)((uint32_t)((uint64_t)pos >> PB2) & 1u) << 47);
                    }
                    bins.template push_batch<LOADS>(bb, v6, ok, lost_p6);
                    bins.template flush<false>(lost_p6);

                    // is the NEXT round the first of a region that may hold groups without a boundary yet?  Then they start here.
                    if (r + 1 < n_seg && x1.base == 0u) {
                        const uint32_t gm = min(gmax(j + x1.t * wpb), n_groups - 1u);
                        if (gm >= next_g) snapshot(next_g, gm + 1u);
                    }
                } else if constexpr (RB) bins.template push_batch<LOADS>(bb, cur, ok, reg, lost);
                else {
                    bins.template push_batch<LOADS>(bb, cur, ok, lost); bins.template flush<false>(lost);
                }
                x0 = x1; x1 = x2; r++;
            };
            while (true) {
                if (r >= n_seg) break;
                round(va, vc);
                if (r >= n_seg) break;
                round(vb, va);
                if (r >= n_seg) break;
                round(vc, vb);
            }
        }
        if (total <= skip + sched_cap) break;
        pt_barrier_lds();  // every wave is done with this segment of the schedule
    }
    if constexpr (P6) {  // the end of the last zone; groups the stream never reached get empty zones behind it
        snapshot(next_g, n_groups);
        bins.store_counts(cnt2 + (uint64_t)blockIdx.x * NB2, [](uint32_t b) { return b; });
    } else if constexpr (RB) { bins.flush(true, reg, lost); bins.store_counts(cnt2 + (uint64_t)blockIdx.x * NB2, reg, [](uint32_t b) { return b; }); }
    else { bins.template flush<true>(lost); bins.store_counts(cnt2 + (uint64_t)blockIdx.x * NB2, [](uint32_t b) { return b; }); }
}

// ------------------------------------------------------------------------------------------ C
// One slice at a time per workgroup.  Survivors (entry >> 31 = edge | position << 3) are staged in LDS and
// appended to sub-list (blockIdx % QS_LISTS) with one global atomic per flush.
constexpr int QL_STAGE = 3072;
constexpr int QL_BUCKETS = 1024;  // = PT_APPLY_THREADS: one counting-sort bucket per thread
constexpr uint32_t QL_LONG_REGION = 65536;  // entries: regions beyond this flush the staged survivors as they go (see k_q_lookup)
constexpr size_t QL_LDS = (size_t)QL_STAGE * 8 + (size_t)QL_BUCKETS * 4 + 128 + 64;  // staged ids (their bucket in the spare high bits) + histogram + scan scratch + control

// Survivors of a slice's first probe, staged in LDS and appended to the workgroup's survivor sub-list GROUPED BY ADDRESS (round 4).
// All occurrences of an edge -- the same (k+1)-mer at its position in every genome that has it -- probe the same address, reach
// the same slice and survive together; k_q_verify then probes functions 1..q-1 of the SAME q-1 addresses once per occurrence
// (the 62-genome workload: 54 M true second edges among 58 M survivors, ~11 occurrences each).  The lookup appended them in
// arrival order, i.e. spread over the flush; a counting sort by the high bits of the slice offset (one LDS atomic per survivor)
// puts equal addresses next to each other, so the 64 lanes of a verifying wave ask for a handful of distinct filter words instead
// of 4 x 64.  Nothing downstream depends on the order of a sub-list.  A staged entry is the 33-bit survivor id with its bucket
// in bits 40..49 (no second array: the 32 KB beside a 128 KB slice hold 3072 entries, as before the grouping).
// (Round 4 also built the counterpart for batches whose survivors are mostly Bloom false positives -- every survivor to the sub-list of
//  its POSITION, the verifying workgroups walking the sub-lists one per XCD at a time so that the scattered text reads stay in that
//  XCD's L2 -- and measured it on 7 x 160 Mbp at f = 34 (5.5 % fill, 0.33 survivors per position): k_q_verify2 7.5 -> 6.6 ms per
//  batch, the query 70.0 -> 68.3 ms; not kept for 2 %.)
// (Round 5: a staged value may be RAW -- the 6-byte path stages {entry bits, index in the region} and turns them into a survivor id
//  only when the staging area is flushed, with every lane busy: done where the bit is tested, the search over the region's group
//  boundaries ran in a branch that 86 % of the wave-instructions entered for 3 % of the lanes.  `res` below: raw -> id, or ~0 for a
//  padding entry, which the verification kernels skip.)
struct SurvIdentity { __device__ __forceinline__ uint64_t operator()(uint64_t raw) const { return raw; } };
struct SurvStage {
    static constexpr int ID_BITS = 54;
    uint64_t *sid;     // [QL_STAGE]
    uint32_t *hist;    // [QL_BUCKETS]
    uint32_t *scan;    // [32]
    uint32_t *ctl;     // [0] staged count, [2..3] flush base
    uint64_t *my_list, *surv0;  // the sub-list at hand, sub-list 0
    unsigned long long *surv_cur;
    uint64_t surv_cap;
    int list, shift;   // bucket = slice offset >> shift
    int group;         // 0: append in arrival order (no sort: a batch whose survivors are mostly Bloom false positives has no equal addresses to bring together)
    __device__ __forceinline__ unsigned char *carve(unsigned char *p, int slice_bits)
    {
        sid = reinterpret_cast<uint64_t *>(p);
        hist = reinterpret_cast<uint32_t *>(sid + QL_STAGE);
        scan = hist + QL_BUCKETS;
        ctl = scan + 32;
        shift = slice_bits > 10 ? slice_bits - 10 : 0;
        return reinterpret_cast<unsigned char *>(ctl + 16);
    }
    template <class Res = SurvIdentity>
    __device__ __forceinline__ void push(uint64_t id, uint32_t a, Res res = Res())
    {
        const uint32_t slot = atomicAdd(&ctl[0], 1u);
        if (slot < (uint32_t)QL_STAGE) sid[slot] = id | ((uint64_t)min(a >> shift, (uint32_t)QL_BUCKETS - 1u) << ID_BITS);
        else {  // staging full (dense hits): straight to the sub-list
            const unsigned long long o = atomicAdd(&surv_cur[list], 1ull);
            if (o < surv_cap) my_list[o] = res(id); else surv_cur[QS_LISTS] = 1ull;
        }
    }
    template <class Res = SurvIdentity>
    __device__ __forceinline__ void flush(Res res = Res())  // all PT_APPLY_THREADS threads
    {
        constexpr int PER = QL_STAGE / PT_APPLY_THREADS;
        static_assert(QL_BUCKETS == PT_APPLY_THREADS && QL_STAGE % PT_APPLY_THREADS == 0, "one bucket per thread, whole entries per thread");
        __syncthreads();
        const uint32_t m = min(ctl[0], (uint32_t)QL_STAGE);
        if (m && !group) {  // (uniform) arrival order
            if (threadIdx.x == 0) {
                const unsigned long long base = atomicAdd(&surv_cur[list], (unsigned long long)m);
                ctl[2] = (uint32_t)base; ctl[3] = (uint32_t)(base >> 32);
            }
            __syncthreads();
            const uint64_t base = (uint64_t)ctl[2] | ((uint64_t)ctl[3] << 32);
            for (uint32_t i = threadIdx.x; i < m; i += PT_APPLY_THREADS) {
                if (base + i < surv_cap) my_list[base + i] = res(sid[i] & ((1ull << ID_BITS) - 1ull));
                else surv_cur[QS_LISTS] = 1ull;
            }
            __syncthreads();
            if (threadIdx.x == 0) ctl[0] = 0;
        } else if (m) {  // (uniform)
            hist[threadIdx.x] = 0;
            __syncthreads();
            uint64_t e[PER];
            uint32_t rank[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const uint32_t i = threadIdx.x + u * PT_APPLY_THREADS;
                e[u] = 0; rank[u] = 0;
                if (i < m) { e[u] = sid[i]; rank[u] = atomicAdd(&hist[(uint32_t)(e[u] >> ID_BITS)], 1u); }
            }
            __syncthreads();
            uint32_t total;
            const uint32_t off = pt_block_excl_scan<PT_APPLY_THREADS>(hist[threadIdx.x], scan, total);
            __syncthreads();
            hist[threadIdx.x] = off;
            if (threadIdx.x == 0) {
                const unsigned long long base = atomicAdd(&surv_cur[list], (unsigned long long)m);
                ctl[2] = (uint32_t)base; ctl[3] = (uint32_t)(base >> 32);
            }
            __syncthreads();
            const uint64_t base = (uint64_t)ctl[2] | ((uint64_t)ctl[3] << 32);
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const uint32_t i = threadIdx.x + u * PT_APPLY_THREADS;
                if (i < m) {
                    const uint64_t at = base + hist[(uint32_t)(e[u] >> ID_BITS)] + rank[u];
                    if (at < surv_cap) my_list[at] = res(e[u] & ((1ull << ID_BITS) - 1ull));
                    else surv_cur[QS_LISTS] = 1ull;  // sub-list overflow -> host falls back
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) ctl[0] = 0;
        }
        __syncthreads();
        // the next flush appends to another sub-list: a slice with hot addresses (a repeat family's k-mers: thousands of survivors) would
        // otherwise put all of them into ONE of the 64 lists (the 62-genome text with repeat families: the fullest list held 27 M of 257 M)
        if (m) { list = (list + 17) & (QS_LISTS - 1); my_list = surv0 + (uint64_t)list * surv_cap; }
    }
    // after a region: flush once the staging area is more than half full
    template <class Res = SurvIdentity>
    __device__ __forceinline__ void maybe_flush(Res res = Res())
    {
        __syncthreads();
        const uint32_t staged = min(ctl[0], (uint32_t)QL_STAGE);
        __syncthreads();  // everyone has read the count before anyone stages more
        if (staged > QL_STAGE / 2) flush(res);
    }
};

__global__ void __launch_bounds__(PT_APPLY_THREADS)
k_q_lookup(int slice_bits, int log_nb2, uint32_t wpb, const uint64_t *__restrict__ buf2, const uint32_t *__restrict__ cnt2,
           const uint64_t *__restrict__ off2, const uint32_t *__restrict__ filter, uint64_t *surv, unsigned long long *surv_cur, uint64_t surv_cap, PtPerm perm, PtShard sh, int group, uint32_t n_slices)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t words = 1u << (slice_bits - 5);
    uint32_t *slice = reinterpret_cast<uint32_t *>(smem);
    SurvStage st;
    st.carve(reinterpret_cast<unsigned char *>(slice + ((words + 3u) & ~3u)), slice_bits);
    st.group = group;
    const uint32_t nb2 = 1u << log_nb2;
    // a long-lived workgroup takes every gridDim.x-th slice (tpc_internal.h:tpc_slice_grid)
    for (uint32_t sl = blockIdx.x; sl < n_slices; sl += gridDim.x) {
        const uint32_t b1 = sl >> log_nb2, b2 = sl & (nb2 - 1);
        // whole filter: natural position of permuted slice blockIdx; shard: compact [local bucket][b2]
        const uint32_t *src_slice = filter + (uint64_t)(sh.world == 1 ? perm.slice_of(sl) : sl) * words;
        if ((words & 3u) == 0) for (uint32_t i = threadIdx.x; i < words / 4; i += PT_APPLY_THREADS) reinterpret_cast<uint4 *>(slice)[i] = reinterpret_cast<const uint4 *>(src_slice)[i];
        else for (uint32_t i = threadIdx.x; i < words; i += PT_APPLY_THREADS) slice[i] = src_slice[i];
        if (threadIdx.x == 0) st.ctl[0] = 0;
        __syncthreads();
        const uint32_t slice_mask = (1u << slice_bits) - 1u;
        st.list = sl % QS_LISTS;
        st.my_list = surv + (uint64_t)st.list * surv_cap; st.surv0 = surv;
        st.surv_cur = surv_cur; st.surv_cap = surv_cap;
        for (uint32_t j = 0; j < wpb; j++) {
            const uint64_t r = ((uint64_t)b1 * wpb + j) * nb2 + b2;
            const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt2[r]);
            auto probe = [&](uint64_t v) {
                const uint32_t a = (uint32_t)v & slice_mask;
                if ((slice[a >> 5] >> (a & 31u)) & 1u) st.push(v >> QE_E_SHIFT, a);
            };
            // A long region (a small filter under a large batch: 200 K entries per slice at f = 34 with 0.56 G positions) would fill the
            // 3072-entry staging area many times over before its end, and every survivor beyond it costs a same-address global atomic
            // (75 ms per lookup there instead of 8): such a region flushes the staging area every 8192 entries.  (Not the short ones: two
            // barriers per 8192 entries would be ~5 % of a 28 K-entry slice of the 62-genome workload.)
            if (n > QL_LONG_REGION) pt_stream_region_with<PT_APPLY_THREADS, 2>(buf2 + off2[r], n, probe, [&]() { st.maybe_flush(); });
            else pt_stream_region<PT_APPLY_THREADS, 2>(buf2 + off2[r], n, probe);
            st.maybe_flush();
        }
        st.flush();
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ C'
// (Round 4 also measured a persistent form of this kernel -- one workgroup per CU taking slices from a counter and issuing the next
//  slice's insert loads before it sorts and appends the survivors of the one at hand, so that no CU waits out a launch, a region
//  count and a first round trip per 450 KB slice: 11.16 ms for k_q_split + this kernel against 11.0 as one short workgroup per
//  slice.  The hardware already overlaps the end of one workgroup with the start of the next; the kernel runs at what HBM gives
//  a two-reads-to-one-write mix.  Round 6 measured the PLAIN form of the same idea on the 6-byte kernel -- 512 workgroups, each taking
//  every 512th slice, nothing prefetched across slices -- and that one pays: 5.83 -> 5.48 ms there, 36.9 -> 35.4 ms per step at f = 38
//  through this kernel (profiles/r06_lookup_grid_ab.txt); asking for the next slice's counts a slice ahead made it slower again.
//  All one-slice-at-a-time kernels are launched that way now: tpc_internal.h:tpc_slice_grid.)
// Fused k_part_apply + k_q_lookup (tpc_partition.hip: deferred apply).  When the insert and the query of a round both
// fit one tile batch, the workgroup that builds a filter slice in LDS from the insert's level-2 entries writes it out
// AND tests the query's entries of that slice on the spot: the 2^L / 8 bytes of the filter are not read back.
__global__ void __launch_bounds__(PT_APPLY_THREADS)
k_apply_lookup(int slice_bits, int log_nb2, uint32_t iwpb, const uint32_t *__restrict__ ibuf2, const uint32_t *__restrict__ icnt2, uint64_t icap2, int fresh,
               const uint64_t *__restrict__ iovf, const uint64_t *__restrict__ iovf_off, uint32_t qwpb, const uint64_t *__restrict__ qbuf2, const uint32_t *__restrict__ qcnt2, const uint64_t *__restrict__ qoff2,
               uint32_t *__restrict__ filter, uint64_t *surv, unsigned long long *surv_cur, uint64_t surv_cap, PtPerm perm, int group, PtShard sh, TpcListSrc ls, uint32_t n_slices)
{   // ls (ls.n_src > 0, sh.world == 1: the combined multi-GPU exchange, tpc_lists.h): set-bit lists of the slice, from this and the other ranks' inserts
    // sh.world > 1 (round 5): the owned slices of a sharded filter, compact layout [local bucket][b2] as k_part_apply / k_q_lookup write and read it
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t words = 1u << (slice_bits - 5);
    uint32_t *slice = reinterpret_cast<uint32_t *>(smem);
    SurvStage st;
    st.carve(reinterpret_cast<unsigned char *>(slice + ((words + 3u) & ~3u)), slice_bits);
    st.group = group;
    uint32_t *s_ctl = st.ctl;
    const uint32_t nb2 = 1u << log_nb2;
    // a long-lived workgroup takes every gridDim.x-th slice (tpc_internal.h:tpc_slice_grid)
    for (uint32_t sl = blockIdx.x; sl < n_slices; sl += gridDim.x) {
        const uint32_t b1 = sl >> log_nb2, b2 = sl & (nb2 - 1);
        uint32_t *out = filter + (uint64_t)(sh.world == 1 ? perm.slice_of(sl) : sl) * words;
        const bool wide = (words & 3u) == 0;
        TpcListReader<PT_APPLY_THREADS> lists;  // (their first loads go out before the slice is zeroed)
        if (ls.n_src) lists.begin(ls, b1, b2, log_nb2, sl, slice_bits);
        // ---- apply (k_part_apply)
        if (fresh) {
            if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += PT_APPLY_THREADS) reinterpret_cast<uint4 *>(slice)[i] = make_uint4(0, 0, 0, 0);
            else for (uint32_t i = threadIdx.x; i < words; i += PT_APPLY_THREADS) slice[i] = 0;
        } else {
            if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += PT_APPLY_THREADS) reinterpret_cast<uint4 *>(slice)[i] = reinterpret_cast<const uint4 *>(out)[i];
            else for (uint32_t i = threadIdx.x; i < words; i += PT_APPLY_THREADS) slice[i] = out[i];
        }
        if (threadIdx.x == 0) s_ctl[0] = 0;
        __syncthreads();
        for (uint32_t j = 0; j < iwpb; j++) {
            const uint64_t r = ((uint64_t)b1 * iwpb + j) * nb2 + b2;
            const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)icnt2[r]);
            pt_stream_region<PT_APPLY_THREADS, 2>(ibuf2 + r * icap2, n, [slice](uint32_t v) { atomicOr(&slice[v >> 5], 1u << (v & 31u)); });
        }
        // the insert's overflow entries (permuted addresses that found a ring or region full), grouped by slice beforehand
        if (iovf_off) {
            const uint64_t o0 = iovf_off[sl], o1 = iovf_off[sl + 1];
            for (uint64_t i = o0 + threadIdx.x; i < o1; i += PT_APPLY_THREADS) {
                const uint64_t a = iovf[i];
                atomicOr(&slice[((uint32_t)a & ((1u << slice_bits) - 1u)) >> 5], 1u << ((uint32_t)a & 31u));
            }
        }
        if (ls.n_src) lists.finish(ls, slice);
        __syncthreads();
        // the first query region's loads go out before the slice's stores: the 128 KB write-out then drains under them
        PtStream<PT_APPLY_THREADS, 2, uint64_t> q0;
        {
            const uint64_t r0 = ((uint64_t)b1 * qwpb) * nb2 + b2;
            q0.begin(qbuf2 + qoff2[r0], (uint32_t)__builtin_amdgcn_readfirstlane((int)qcnt2[r0]));
        }
        if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += PT_APPLY_THREADS) reinterpret_cast<uint4 *>(out)[i] = reinterpret_cast<const uint4 *>(slice)[i];
        else for (uint32_t i = threadIdx.x; i < words; i += PT_APPLY_THREADS) out[i] = slice[i];
        // ---- lookup (k_q_lookup) against the slice still in LDS
        const uint32_t slice_mask = (1u << slice_bits) - 1u;
        st.list = sl % QS_LISTS;
        st.my_list = surv + (uint64_t)st.list * surv_cap; st.surv0 = surv;
        st.surv_cur = surv_cur; st.surv_cap = surv_cap;
        for (uint32_t j = 0; j < qwpb; j++) {
            const uint64_t r = ((uint64_t)b1 * qwpb + j) * nb2 + b2;
            auto probe = [&](uint64_t v) {
                const uint32_t a = (uint32_t)v & slice_mask;
                if ((slice[a >> 5] >> (a & 31u)) & 1u) st.push(v >> QE_E_SHIFT, a);
            };
            if (j == 0) {
                if (q0.n > QL_LONG_REGION) q0.finish_with(probe, [&]() { st.maybe_flush(); });  // (uniform; see k_q_lookup)
                else q0.finish(probe);
            } else {
                const uint32_t nq = (uint32_t)__builtin_amdgcn_readfirstlane((int)qcnt2[r]);
                if (nq > QL_LONG_REGION) pt_stream_region_with<PT_APPLY_THREADS, 2>(qbuf2 + qoff2[r], nq, probe, [&]() { st.maybe_flush(); });
                else pt_stream_region<PT_APPLY_THREADS, 2>(qbuf2 + qoff2[r], nq, probe);
            }
            st.maybe_flush();
        }
        st.flush();
        __syncthreads();
    }
}

// Region-overflow entries: first probe straight from the filter; hits join sub-list 0.
__global__ void k_q_ovf(const uint64_t *__restrict__ list, const unsigned long long *cursor, uint64_t cap, const uint32_t *__restrict__ filter,
                        uint64_t *surv, unsigned long long *surv_cur, uint64_t surv_cap, PtPerm perm, PtShard sh, int log_nb2)
{
    const uint64_t n = min((uint64_t)cursor[0], cap);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        bool mine;
        const uint64_t a = pt_local_addr(perm, sh, log_nb2, list[2 * i], mine);  // entries of other ranks are routed by the host layer
        const bool hit = mine && ((filter[a >> 5] >> ((uint32_t)a & 31u)) & 1u);
        const unsigned long long m = __ballot(hit);  // one append per wave (every lane of a wave runs the same number of rounds but the last)
        if (hit) {
            const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__ffsll((long long)m) - 1u;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(&surv_cur[0], (unsigned long long)__popcll(m));
            base = __shfl(base, (int)leader, 64);
            const unsigned long long o = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
            if (o < surv_cap) surv[o] = list[2 * i + 1]; else surv_cur[QS_LISTS] = 1ull;
        }
    }
}

#endif  // part 0 only
// ------------------------------------------------------------------------------------------ D
template <int Q>
__global__ void __launch_bounds__(256)
k_q_verify(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases, const uint32_t *__restrict__ filter,
           const uint64_t *__restrict__ surv, const unsigned long long *__restrict__ surv_cur, uint64_t surv_cap, uint64_t gbase, uint32_t *rmask)
{
    __shared__ uint64_t s_h[Q * 5], s_hk[Q * 5];
    if (threadIdx.x < Q * 5) { s_h[threadIdx.x] = tab[threadIdx.x]; s_hk[threadIdx.x] = tab[TPC_TAB_HK + threadIdx.x]; }
    __syncthreads();
    const int list = blockIdx.y;
    const uint64_t n = min((uint64_t)surv_cur[list], surv_cap);
    const uint64_t *my = surv + (uint64_t)list * surv_cap;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
        const uint64_t sid = my[idx];
        if (sid == ~0ull) continue;  // a padding entry of the 6-byte path (tpc_qpart6.h)
        const int e = (int)(sid & 7);
        const uint64_t g = gbase + (sid >> 3);
        // (no "is the position marked already?" test: that scattered read cost more than the work it saved, 3.73 -> 3.40 ms.
        //  Ablation on the 62-genome workload: 2.3 ms remain without probes and marks -- the scattered text reads -- the q - 1
        //  probes add 1.1 and the mark atomics 0.4; hashing four bases per step from byte tables, or a cache of edge verdicts
        //  keyed by the exact (k+1)-mer, change nothing: the kernel waits for scattered accesses, not for arithmetic.  Bucketing
        //  the survivors by position first (one radix pass) makes the text reads local but piles the mark atomics of the waves
        //  in flight onto the same words: 5.4 ms.)
        const int c = e & 3;
        // vertex hash of the window at g for functions LO .. HI-1 on one strand (VertexRollingHash ctor, vertexrollinghash.h:79-102;
        // the negative strand reads the window backwards and complemented)
        auto roll = [&](auto lo_tag, auto hi_tag, bool neg_strand, uint64_t (&h)[Q]) {
            constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
#pragma unroll
            for (int i = LO; i < HI; i++) h[i] = 0;
            if (!neg_strand) {
                for (int t0 = 0; t0 < P.k; t0 += 32) {
                    uint64_t w = tpc_text_word(bases, g + t0);
                    const int m = min(32, P.k - t0);
                    for (int t = 0; t < m; t++) {
                        const int ch = (int)(w & 3);
                        w >>= 2;
#pragma unroll
                        for (int i = LO; i < HI; i++) h[i] = tpc_rotl1(h[i], P.L, P.lmask) ^ s_h[i * 5 + ch];
                    }
                }
            } else {
                for (int t1 = P.k; t1 > 0; t1 -= 32) {  // reverse complement: last base first
                    const int m = min(32, t1);
                    const uint64_t w = tpc_text_word(bases, g + t1 - m);
                    for (int t = m - 1; t >= 0; t--) {
                        const int ch = 3 - (int)((w >> (2 * t)) & 3);
#pragma unroll
                        for (int i = LO; i < HI; i++) h[i] = tpc_rotl1(h[i], P.L, P.lmask) ^ s_h[i * 5 + ch];
                    }
                }
            }
        };
        // hash of the edge (in-edge c + v for e < 4, out-edge v + c otherwise) from the vertex hash of its strand
        auto edge = [&](int i, bool neg_strand, uint64_t vh) -> uint64_t {
            if (!neg_strand) return e < 4 ? (s_hk[i * 5 + c] ^ vh) : (tpc_rotl1(vh, P.L, P.lmask) ^ s_h[i * 5 + c]);
            return e < 4 ? (tpc_rotl1(vh, P.L, P.lmask) ^ s_h[i * 5 + 3 - c]) : (vh ^ s_hk[i * 5 + 3 - c]);
        };
        auto probe = [&](uint64_t a) -> bool { return (filter[a >> 5] >> ((uint32_t)a & 31u)) & 1u; };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, (Q > 2 ? 2 : Q)>;
        using IQ = std::integral_constant<int, Q>;
        // Function 0 on both strands decides the strand unless its two values tie (vertexrollinghash.h:170-200); then function 1
        // of that strand and its probe alone -- a Bloom false positive of function 0 (most survivors of a well-filled filter)
        // ends there -- and functions 2..Q-1 only for what passed: 3 rolls instead of 2Q for the common survivor.
        bool present = true;  // function 0 passed in k_q_lookup
        uint64_t hp[Q], hn[Q];
        roll(I0(), I1(), false, hp);
        roll(I0(), I1(), true, hn);
        const uint64_t p0 = edge(0, false, hp[0]), n0 = edge(0, true, hn[0]);
        if (p0 != n0) {
            const bool ng = n0 < p0;
            if (Q > 1) {
                roll(I1(), I2(), ng, hp);
                present = probe(edge(1, ng, hp[1 % Q]));
            }
            if (present && Q > 2) {
                roll(I2(), IQ(), ng, hp);
                uint64_t addr[Q];
                uint32_t wv[Q];
#pragma unroll
                for (int i = 2; i < Q; i++) { addr[i] = edge(i, ng, hp[i]); wv[i] = filter[addr[i] >> 5]; }  // independent loads
#pragma unroll
                for (int i = 2; i < Q; i++) present = present && ((wv[i] >> ((uint32_t)addr[i] & 31u)) & 1u);
            }
        } else {  // a function-0 tie: all 2Q hashes, the first function whose two values differ decides
            roll(I1(), IQ(), false, hp);
            roll(I1(), IQ(), true, hn);
            uint64_t p[Q], nn[Q];
#pragma unroll
            for (int i = 0; i < Q; i++) { p[i] = edge(i, false, hp[i]); nn[i] = edge(i, true, hn[i]); }
            const bool ng = tpc_pick_neg<Q>(p, nn);
#pragma unroll
            for (int i = 1; i < Q; i++) present = present && probe(ng ? nn[i] : p[i]);
        }
        // (Round 4 tried marking with plain byte stores -- one byte per position of the batch, folded into the mask by a small kernel
        //  afterwards -- instead of these 54 M device-scope atomics: 3.14 -> 3.27 ms + 0.06 for the fold, profiles/r04a_*.  A scattered
        //  partial-line store costs the memory system what the atomic does.)
        if (present) atomicOr(&rmask[g >> 5], 1u << ((uint32_t)g & 31u));
    }
}

// ------------------------------------------------------------------------------------------ D, table form
// k_q_verify for k <= 31: the edge is a (k+1)-mer that fits one packed word, and its 2Q hashes are evaluated in closed form
// from a table of pre-rotated letter hashes -- H_i(E) = XOR_t rotl(h_i[E_t], k - t), H_i(rc E) = XOR_t rotl(h_i[rc E_t], t)
// (the fold of cyclichash.h:106-109 written out) -- one 16-byte LDS read per letter and function for both strands, no
// rotations in the loop: ~600 instead of ~1600 vector instructions per survivor (the kernel was 74 % VALU issue,
// profiles/r03a_sq.csv).  Same verdicts: canonical strand by the first function whose two values differ (tpc_pick_neg).
// LAZY (the form in use; TPC_VERIFY_LAZY=0 selects the other one for measurements): function 0 on both strands -- it decides
// the strand -- then function 1 of that strand and its probe ALONE, and only a survivor that passes goes on to functions
// 2..Q-1.  On a well-filled filter most survivors are Bloom false positives of function 0 and end at that first probe: a
// quarter of the scattered loads and under half of the instructions of the all-at-once form (full configs[3], f = 38, 0.4
// survivors per position: query 1798 -> 1654 ms).  Where the survivors are mostly true second edges (M2: 54 of 58 M pass every
// probe) the two forms measure the same (27.7 ms per step either way).
template <int Q, bool LAZY>
__global__ void __launch_bounds__(256)
k_q_verify2(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases, const uint32_t *__restrict__ filter,
            const uint64_t *__restrict__ surv, const unsigned long long *__restrict__ surv_cur, uint64_t surv_cap, uint64_t gbase, uint32_t *rmask)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *s_t = reinterpret_cast<uint4 *>(smem);  // [k + 1][4][Q]: {rotl(h_i[c], k - t), rotl(h_i[3 - c], t)}
    const int k = P.k, L = P.L;
    for (int i = threadIdx.x; i < (k + 1) * 4 * Q; i += 256) {
        const int t = i / (4 * Q), c = (i / Q) & 3, f = i % Q;
        const uint64_t a = q_rotl_n(tab[f * 5 + c], L, (k - t) % L), b = q_rotl_n(tab[f * 5 + 3 - c], L, t % L);
        s_t[i] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    }
    __syncthreads();
    const int list = blockIdx.y;
    const uint64_t n = min((uint64_t)surv_cur[list], surv_cap);
    const uint64_t *my = surv + (uint64_t)list * surv_cap;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t wmask = (1ull << (2 * k)) - 1ull;  // k <= 31
    // software pipeline: the survivor id and the text word of the NEXT survivor are loaded before this one is hashed (the
    // kernel waits for scattered accesses, not for arithmetic: twice the loads in flight per lane)
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // (~0 = a padding entry of the 6-byte path, tpc_qpart6.h: skipped; its text load reads position 0)
    uint64_t sid_n = idx < n ? my[idx] : 0;
    uint64_t w_n = idx < n ? tpc_text_word_x2(bases, gbase + (sid_n == ~0ull ? 0ull : sid_n >> 3)) : 0;
    for (; idx < n; idx += stride) {
        const uint64_t sid = sid_n;
        const uint64_t w = w_n & wmask;
        if (idx + stride < n) {
            sid_n = my[idx + stride];
            w_n = tpc_text_word_x2(bases, gbase + (sid_n == ~0ull ? 0ull : sid_n >> 3));
        }
        if (sid == ~0ull) continue;
        const int e = (int)(sid & 7), c = e & 3;
        const uint64_t g = gbase + (sid >> 3);
        // the edge's k + 1 letters, first letter in the low bits: in-edge c + v, out-edge v + c
        uint64_t E = e < 4 ? ((w << 2) | (uint64_t)c) : (w | ((uint64_t)c << (2 * k)));
        bool present = true;  // function 0 passed in k_q_lookup
        if constexpr (!LAZY || Q < 3) {
            uint64_t p[Q], nn[Q];
#pragma unroll
            for (int i = 0; i < Q; i++) { p[i] = 0; nn[i] = 0; }
            const uint4 *row = s_t;
            for (int t = 0; t <= k; t++) {
                const uint4 *r = row + ((uint32_t)E & 3u) * Q;
                E >>= 2;
                row += 4 * Q;
#pragma unroll
                for (int i = 0; i < Q; i++) {
                    const uint4 x = r[i];
                    p[i] ^= ((uint64_t)x.y << 32) | x.x;
                    nn[i] ^= ((uint64_t)x.w << 32) | x.z;
                }
            }
            const bool ng = tpc_pick_neg<Q>(p, nn);  // DetermineStrandExtend / Prepend (vertexrollinghash.h:170-200)
            uint32_t wv[Q];  // the other probes are independent loads
            uint64_t addr[Q];
#pragma unroll
            for (int i = 1; i < Q; i++) { addr[i] = ng ? nn[i] : p[i]; wv[i] = filter[addr[i] >> 5]; }
#pragma unroll
            for (int i = 1; i < Q; i++) present = present && ((wv[i] >> ((uint32_t)addr[i] & 31u)) & 1u);
        } else {
            // function 0, both strands
            uint64_t p0 = 0, n0 = 0;
            {
                uint64_t e2 = E;
                const uint4 *row = s_t;
                for (int t = 0; t <= k; t++) {
                    const uint4 x = row[((uint32_t)e2 & 3u) * Q];
                    e2 >>= 2;
                    row += 4 * Q;
                    p0 ^= ((uint64_t)x.y << 32) | x.x;
                    n0 ^= ((uint64_t)x.w << 32) | x.z;
                }
            }
            bool ng = n0 < p0;
            if (p0 == n0) {  // a strand tie on function 0 (palindromic edge or a collision): the later functions decide
                uint64_t p[Q], nn[Q];
#pragma unroll
                for (int i = 0; i < Q; i++) { p[i] = 0; nn[i] = 0; }
                uint64_t e2 = E;
                const uint4 *row = s_t;
                for (int t = 0; t <= k; t++) {
                    const uint4 *r = row + ((uint32_t)e2 & 3u) * Q;
                    e2 >>= 2;
                    row += 4 * Q;
#pragma unroll
                    for (int i = 0; i < Q; i++) {
                        const uint4 x = r[i];
                        p[i] ^= ((uint64_t)x.y << 32) | x.x;
                        nn[i] ^= ((uint64_t)x.w << 32) | x.z;
                    }
                }
                ng = tpc_pick_neg<Q>(p, nn);
            }
            // function 1 of the canonical strand alone
            const uint2 *half = reinterpret_cast<const uint2 *>(s_t) + (ng ? 1 : 0);  // .xy = positive, .zw = negative strand
            uint64_t a1 = 0;
            {
                uint64_t e2 = E;
                const uint2 *row = half + 2;  // function 1
                for (int t = 0; t <= k; t++) {
                    const uint2 x = row[((uint32_t)e2 & 3u) * (2 * Q)];
                    e2 >>= 2;
                    row += 8 * Q;
                    a1 ^= ((uint64_t)x.y << 32) | x.x;
                }
            }
            present = (filter[a1 >> 5] >> ((uint32_t)a1 & 31u)) & 1u;
            if (present) {
                uint64_t a[Q];
#pragma unroll
                for (int i = 2; i < Q; i++) a[i] = 0;
                uint64_t e2 = E;
                const uint2 *row = half;
                for (int t = 0; t <= k; t++) {
                    const uint2 *r = row + ((uint32_t)e2 & 3u) * (2 * Q);
                    e2 >>= 2;
                    row += 8 * Q;
#pragma unroll
                    for (int i = 2; i < Q; i++) {
                        const uint2 x = r[2 * i];
                        a[i] ^= ((uint64_t)x.y << 32) | x.x;
                    }
                }
                uint32_t wv[Q];
#pragma unroll
                for (int i = 2; i < Q; i++) wv[i] = filter[a[i] >> 5];
#pragma unroll
                for (int i = 2; i < Q; i++) present = present && ((wv[i] >> ((uint32_t)a[i] & 31u)) & 1u);
            }
        }
        // (Round 4 tried marking with plain byte stores -- one byte per position of the batch, folded into the mask by a small kernel
        //  afterwards -- instead of these 54 M device-scope atomics: 3.14 -> 3.27 ms + 0.06 for the fold, profiles/r04a_*.  A scattered
        //  partial-line store costs the memory system what the atomic does.)
        if (present) atomicOr(&rmask[g >> 5], 1u << ((uint32_t)g & 31u));
    }
}

#if TPC_QPARTITION_PART == 0
#include "tpc_qpart6.h"
#endif

// ------------------------------------------------------------------------------------------ sharded verification
constexpr int RT_CHUNK = 4096, RT_MAXW = 64;  // owner routing: items per workgroup round, most ranks
constexpr int V_OWNER_SHIFT = TPC_V_OWNER_SHIFT;  // tagged probe addresses: owner rank above the shard-local bit address (< 2^41)
// With the filter sharded by bit address the q-1 remaining probes of a survivor live on other ranks:
// k_v_addrs gives, for hash functions fn .. fn+fn_count-1, the (owner rank, address inside the owner's
// shard) of every survivor's edge; the host layer exchanges them, k_v_probe answers on the owner, the
// survivors with a miss are dropped, and k_v_mark sets the marks of those that passed all q functions.
// (The driver probes function 1 alone first -- it rejects all but a fill-rate share of the Bloom false
// positives -- and functions 2..q-1 together for what is left, which is mostly true second edges.)
template <int Q>
__global__ void __launch_bounds__(256)
k_v_addrs(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases, const uint64_t *__restrict__ sid_list, uint64_t n,
          uint64_t gbase, PtPerm perm, PtShard sh, int log_nb2, int fn, int fn_count, uint64_t *__restrict__ addr_out, int32_t *__restrict__ owner_out,
          unsigned long long *owner_counts)
{   // owner_out == nullptr: the owner rides in bits 56.. of the address (V_OWNER_SHIFT) and owner_counts[o] += probes for owner o
    __shared__ uint64_t s_h[Q * 5], s_hk[Q * 5];
    __shared__ uint32_t s_own[RT_MAXW];
    if (threadIdx.x < RT_MAXW) s_own[threadIdx.x] = 0;
    if (threadIdx.x < Q * 5) { s_h[threadIdx.x] = tab[threadIdx.x]; s_hk[threadIdx.x] = tab[TPC_TAB_HK + threadIdx.x]; }
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
        const uint64_t sid = sid_list[idx];
        const int e = (int)(sid & 7);
        const uint64_t g = gbase + ((sid >> 3) & ((1ull << (30u - sh.log_world())) - 1ull));  // the source-rank bits are this rank's own
        uint64_t pos[Q], neg[Q];
#pragma unroll
        for (int i = 0; i < Q; i++) { pos[i] = 0; neg[i] = 0; }
        for (int t0 = 0; t0 < P.k; t0 += 32) {
            uint64_t w = tpc_text_word(bases, g + t0);
            const int m = min(32, P.k - t0);
            for (int t = 0; t < m; t++) {
                const int c = (int)(w & 3);
                w >>= 2;
#pragma unroll
                for (int i = 0; i < Q; i++) pos[i] = tpc_rotl1(pos[i], P.L, P.lmask) ^ s_h[i * 5 + c];
            }
        }
        for (int t1 = P.k; t1 > 0; t1 -= 32) {
            const int m = min(32, t1);
            const uint64_t w = tpc_text_word(bases, g + t1 - m);
            for (int t = m - 1; t >= 0; t--) {
                const int c = 3 - (int)((w >> (2 * t)) & 3);
#pragma unroll
                for (int i = 0; i < Q; i++) neg[i] = tpc_rotl1(neg[i], P.L, P.lmask) ^ s_h[i * 5 + c];
            }
        }
        const int c = e & 3;
        uint64_t p[Q], nn[Q];
#pragma unroll
        for (int i = 0; i < Q; i++) {
            if (e < 4) { p[i] = s_hk[i * 5 + c] ^ pos[i]; nn[i] = tpc_rotl1(neg[i], P.L, P.lmask) ^ s_h[i * 5 + 3 - c]; }
            else { p[i] = tpc_rotl1(pos[i], P.L, P.lmask) ^ s_h[i * 5 + c]; nn[i] = neg[i] ^ s_hk[i * 5 + 3 - c]; }
        }
        const bool ng = tpc_pick_neg<Q>(p, nn);
#pragma unroll
        for (int i = 0; i < Q; i++) {
            if (i < fn || i >= fn + fn_count) continue;
            const uint64_t ap = perm.fwd(ng ? nn[i] : p[i]);
            bool mine;
            const uint64_t la = pt_local_addr(perm, sh, log_nb2, ap, mine);
            const uint32_t own = ((uint32_t)(ap >> perm.slice_bits) >> log_nb2) & (sh.world - 1);
            if (owner_out) {
                addr_out[idx * fn_count + (i - fn)] = la;
                owner_out[idx * fn_count + (i - fn)] = (int32_t)own;
            } else {
                addr_out[idx * fn_count + (i - fn)] = la | ((uint64_t)own << V_OWNER_SHIFT);
                if (owner_counts) atomicAdd(&s_own[own & (RT_MAXW - 1)], 1u);
            }
        }
    }
    if (owner_counts) {
        __syncthreads();
        if (threadIdx.x < RT_MAXW && s_own[threadIdx.x]) atomicAdd(&owner_counts[threadIdx.x], (unsigned long long)s_own[threadIdx.x]);
    }
}

// k_v_addrs for k <= 31, table form (as k_q_verify2): the edge's 2Q hashes in closed form from a table of pre-rotated letter hashes --
// one 16-byte LDS read per letter and function for both strands instead of two rotations and two lookups: ~750 instead of
// ~2500 vector instructions per survivor.  Same addresses and owners.
template <int Q>
__global__ void __launch_bounds__(256)
k_v_addrs2(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases, const uint64_t *__restrict__ sid_list, uint64_t n,
           uint64_t gbase, PtPerm perm, PtShard sh, int log_nb2, int fn, int fn_count, uint64_t *__restrict__ addr_out, int32_t *__restrict__ owner_out,
           unsigned long long *owner_counts)
{   // owner_out == nullptr: tagged addresses and owner counts, as k_v_addrs
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *s_t = reinterpret_cast<uint4 *>(smem);  // [k + 1][4][Q]: {rotl(h_i[c], k - t), rotl(h_i[3 - c], t)}
    __shared__ uint32_t s_own[RT_MAXW];
    if (threadIdx.x < RT_MAXW) s_own[threadIdx.x] = 0;
    const int k = P.k, L = P.L;
    for (int i = threadIdx.x; i < (k + 1) * 4 * Q; i += 256) {
        const int t = i / (4 * Q), c = (i / Q) & 3, f = i % Q;
        const uint64_t a = q_rotl_n(tab[f * 5 + c], L, (k - t) % L), b = q_rotl_n(tab[f * 5 + 3 - c], L, t % L);
        s_t[i] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    }
    __syncthreads();
    const uint64_t wmask = (1ull << (2 * k)) - 1ull;  // k <= 31
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
        const uint64_t sid = sid_list[idx];
        const int e = (int)(sid & 7), c = e & 3;
        const uint64_t g = gbase + ((sid >> 3) & ((1ull << (30u - sh.log_world())) - 1ull));  // the source-rank bits are this rank's own
        const uint64_t w = tpc_text_word_x2(bases, g) & wmask;
        uint64_t E = e < 4 ? ((w << 2) | (uint64_t)c) : (w | ((uint64_t)c << (2 * k)));  // in-edge c + v, out-edge v + c: first letter in the low bits
        uint64_t p[Q], nn[Q];
#pragma unroll
        for (int i = 0; i < Q; i++) { p[i] = 0; nn[i] = 0; }
        const uint4 *row = s_t;
        for (int t = 0; t <= k; t++) {
            const uint4 *r = row + ((uint32_t)E & 3u) * Q;
            E >>= 2;
            row += 4 * Q;
#pragma unroll
            for (int i = 0; i < Q; i++) {
                const uint4 x = r[i];
                p[i] ^= ((uint64_t)x.y << 32) | x.x;
                nn[i] ^= ((uint64_t)x.w << 32) | x.z;
            }
        }
        const bool ng = tpc_pick_neg<Q>(p, nn);
#pragma unroll
        for (int i = 0; i < Q; i++) {
            if (i < fn || i >= fn + fn_count) continue;
            const uint64_t ap = perm.fwd(ng ? nn[i] : p[i]);
            bool mine;
            const uint64_t la = pt_local_addr(perm, sh, log_nb2, ap, mine);
            const uint32_t own = ((uint32_t)(ap >> perm.slice_bits) >> log_nb2) & (sh.world - 1);
            if (owner_out) {
                addr_out[idx * fn_count + (i - fn)] = la;
                owner_out[idx * fn_count + (i - fn)] = (int32_t)own;
            } else {
                addr_out[idx * fn_count + (i - fn)] = la | ((uint64_t)own << V_OWNER_SHIFT);
                if (owner_counts) atomicAdd(&s_own[own & (RT_MAXW - 1)], 1u);
            }
        }
    }
    if (owner_counts) {
        __syncthreads();
        if (threadIdx.x < RT_MAXW && s_own[threadIdx.x]) atomicAdd(&owner_counts[threadIdx.x], (unsigned long long)s_own[threadIdx.x]);
    }
}

#if TPC_QPARTITION_PART == 1
// the 64 survivor sub-lists as one contiguous list
__global__ void k_surv_gather(const uint64_t *__restrict__ surv, const unsigned long long *__restrict__ surv_cur, uint64_t surv_cap, uint64_t *__restrict__ out)
{
    const int list = blockIdx.y;
    uint64_t off = 0;
    for (int i = 0; i < list; i++) off += min((uint64_t)surv_cur[i], surv_cap);
    const uint64_t n = min((uint64_t)surv_cur[list], surv_cap);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[off + i] = surv[(uint64_t)list * surv_cap + i];
}

__global__ void k_v_probe(const uint32_t *__restrict__ filter, const uint64_t *__restrict__ addr, uint64_t n, uint8_t *__restrict__ hit)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t a = addr[i];
        hit[i] = (uint8_t)((filter[a >> 5] >> ((uint32_t)a & 31u)) & 1u);
    }
}

__global__ void k_v_mark(const uint64_t *__restrict__ sid_list, uint64_t n, uint64_t gbase, uint64_t posmask, uint32_t *rmask)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t g = gbase + ((sid_list[i] >> 3) & posmask);
        atomicOr(&rmask[g >> 5], 1u << ((uint32_t)g & 31u));
    }
}

// ---- owner routing of the survivor probes (the host layer only moves the buffers) -------------------------
// perm[i] = slot of item i in the owner-major send order.  Each workgroup ranks a chunk of items in LDS (one LDS
// atomic per item) and reserves its share of every owner's range with one global atomic per owner and chunk.
__global__ void __launch_bounds__(256)
k_route_count(const int32_t *__restrict__ owner, uint64_t n, unsigned long long *counts)
{
    __shared__ uint32_t h[RT_MAXW];
    if (threadIdx.x < RT_MAXW) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) atomicAdd(&h[owner[i] & (RT_MAXW - 1)], 1u);
    __syncthreads();
    if (threadIdx.x < RT_MAXW && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

__global__ void __launch_bounds__(256)
k_route_scatter(const int32_t *__restrict__ owner, uint64_t n, unsigned long long *cursor, uint32_t *__restrict__ perm)
{
    __shared__ uint32_t h[RT_MAXW];
    __shared__ unsigned long long base[RT_MAXW];
    for (uint64_t c0 = (uint64_t)blockIdx.x * RT_CHUNK; c0 < n; c0 += (uint64_t)gridDim.x * RT_CHUNK) {
        if (threadIdx.x < RT_MAXW) h[threadIdx.x] = 0;
        __syncthreads();
        uint32_t rank[RT_CHUNK / 256];
        int own[RT_CHUNK / 256];
#pragma unroll
        for (int u = 0; u < RT_CHUNK / 256; u++) {
            const uint64_t i = c0 + (uint64_t)u * 256 + threadIdx.x;
            own[u] = i < n ? (owner[i] & (RT_MAXW - 1)) : -1;
            rank[u] = own[u] >= 0 ? atomicAdd(&h[own[u]], 1u) : 0u;
        }
        __syncthreads();
        if (threadIdx.x < RT_MAXW && h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], (unsigned long long)h[threadIdx.x]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RT_CHUNK / 256; u++) {
            const uint64_t i = c0 + (uint64_t)u * 256 + threadIdx.x;
            if (own[u] >= 0) perm[i] = (uint32_t)(base[own[u]] + rank[u]);
        }
        __syncthreads();
    }
}

__global__ void k_permute64(const uint64_t *__restrict__ src, const uint32_t *__restrict__ perm, uint64_t n, uint64_t *__restrict__ dst)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[perm[i]] = src[i];
}

__global__ void k_permute_rows(const uint64_t *__restrict__ src, const uint32_t *__restrict__ perm, uint64_t n, int row_words, uint64_t *__restrict__ dst)
{   // one thread per word: dst row perm[i] = src row i
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, total = n * (uint64_t)row_words;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) {
        const uint64_t i = j / (uint64_t)row_words, w = j - i * (uint64_t)row_words;
        dst[(uint64_t)perm[i] * (uint64_t)row_words + w] = src[j];
    }
}

// all fn_count answers of survivor i are 1.  The answers arrive in SEND order: hit[perm[i * fn_count + t]]; perm == nullptr: the send
// order was the natural one (one rank).  Every load is issued before any is tested.
__device__ __forceinline__ bool v_all_hit(const uint8_t *__restrict__ hit, const uint32_t *__restrict__ perm, uint64_t i, int fn_count)
{
    const uint64_t b = i * (uint64_t)fn_count;
    uint32_t acc = 1;
    if (!perm) {
        if (fn_count == 4) return *reinterpret_cast<const uint32_t *>(hit + b) == 0x01010101u;
        for (int t = 0; t < fn_count; t++) acc &= hit[b + t];
        return acc != 0;
    }
    constexpr int U = 4;
    for (int t0 = 0; t0 < fn_count; t0 += U) {
        uint32_t slot[U], h[U];
#pragma unroll
        for (int u = 0; u < U; u++) slot[u] = t0 + u < fn_count ? perm[b + t0 + u] : 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < U; u++) h[u] = slot[u] != 0xFFFFFFFFu ? hit[slot[u]] : 1u;
#pragma unroll
        for (int u = 0; u < U; u++) acc &= h[u];
    }
    return acc != 0;
}

// survivors whose fn_count answers are all 1; order of the output is not significant.  Compaction per wave: a ballot, one global
// atomic per wave and round (no workgroup barrier: the waves run independently).
__global__ void __launch_bounds__(256)
k_select(const uint64_t *__restrict__ sid, uint64_t n, int fn_count, const uint8_t *__restrict__ hit, const uint32_t *__restrict__ perm,
         uint64_t *__restrict__ out, unsigned long long *n_out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += stride) {
        const uint64_t i = i0 + lane;
        const bool keep = i < n && v_all_hit(hit, perm, i, fn_count);
        const uint64_t id = keep ? sid[i] : 0ull;
        const unsigned long long m = __ballot(keep);
        if (m == 0) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(n_out, (unsigned long long)__popcll(m));
        base = __shfl(base, 0, 64);
        if (keep) out[base + __popcll(m & ((1ull << lane) - 1ull))] = id;
    }
}

// last round of a batch: select and mark in one pass (no list of the kept ids); *n_marked += survivors that passed
__global__ void __launch_bounds__(256)
k_v_finish(const uint64_t *__restrict__ sid, uint64_t n, int fn_count, const uint8_t *__restrict__ hit, const uint32_t *__restrict__ perm,
           uint64_t gbase, uint64_t posmask, uint32_t *rmask, unsigned long long *n_marked)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!v_all_hit(hit, perm, i, fn_count)) continue;
        const uint64_t g = gbase + ((sid[i] >> 3) & posmask);
        atomicOr(&rmask[g >> 5], 1u << ((uint32_t)g & 31u));
        mine++;
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_marked, (unsigned long long)mine);
}

// owner routing of tagged 64-bit items in one go: the owner is ((v >> shift) & omask); counts[o] (exclusive cursors set by the caller)
// give every owner its range of dst; dst[slot] = v & keep, perm[i] = slot (perm may be null).  The ranking is k_route_scatter's.
__global__ void __launch_bounds__(256)
k_route_scatter64(const uint64_t *__restrict__ v, uint64_t n, int shift, uint32_t omask, uint64_t keep, unsigned long long *cursor,
                  uint32_t *__restrict__ perm, uint64_t *__restrict__ dst)
{
    __shared__ uint32_t h[RT_MAXW];
    __shared__ unsigned long long base[RT_MAXW];
    for (uint64_t c0 = (uint64_t)blockIdx.x * RT_CHUNK; c0 < n; c0 += (uint64_t)gridDim.x * RT_CHUNK) {
        if (threadIdx.x < RT_MAXW) h[threadIdx.x] = 0;
        __syncthreads();
        uint32_t rank[RT_CHUNK / 256];
        int own[RT_CHUNK / 256];
        uint64_t val[RT_CHUNK / 256];
#pragma unroll
        for (int u = 0; u < RT_CHUNK / 256; u++) {
            const uint64_t i = c0 + (uint64_t)u * 256 + threadIdx.x;
            val[u] = i < n ? v[i] : 0ull;
            own[u] = i < n ? (int)((uint32_t)(val[u] >> shift) & omask & (RT_MAXW - 1)) : -1;
        }
#pragma unroll
        for (int u = 0; u < RT_CHUNK / 256; u++) rank[u] = own[u] >= 0 ? atomicAdd(&h[own[u]], 1u) : 0u;
        __syncthreads();
        if (threadIdx.x < RT_MAXW && h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], (unsigned long long)h[threadIdx.x]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RT_CHUNK / 256; u++) {
            const uint64_t i = c0 + (uint64_t)u * 256 + threadIdx.x;
            if (own[u] >= 0) {
                const uint64_t slot = base[own[u]] + rank[u];
                dst[slot] = val[u] & keep;
                if (perm) perm[i] = (uint32_t)slot;
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
k_route_count64(const uint64_t *__restrict__ v, uint64_t n, int shift, uint32_t omask, unsigned long long *counts)
{
    __shared__ uint32_t h[RT_MAXW];
    if (threadIdx.x < RT_MAXW) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) atomicAdd(&h[(uint32_t)(v[i] >> shift) & omask & (RT_MAXW - 1)], 1u);
    __syncthreads();
    if (threadIdx.x < RT_MAXW && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

#endif  // part 1 only
#if TPC_QPARTITION_PART == 0
// 512 bins per level (f = 37, 38 at the default slice size): the barrier-free rings (k_q_hash / k_q_split<.., RB = true>)
inline bool q_use_rbins(int log_nb) { return log_nb >= 9; }

void launch_qhash(const TpcLaunch &a, const TpcQPlan &pl, bool gated, uint64_t lo, uint64_t hi, uint32_t *rmask)
{
    QOverflow ovf{pl.ovf, pl.ovf_cur, pl.ovf_cap};
    const PtPerm perm{pl.slice_bits, pl.b1 + pl.b2 + pl.b3, pl.perm_mult, pl.perm_inv};
    const PtShard sh{pl.rank, pl.world};
    const bool rb = q_use_rbins(pl.b1);
    // the instruction-lean kernel: flush-per-round bins (two rounds per position at 512 bins), a 24-bit slice index
    const bool lean = pl.b1 <= 9 && pl.sub_rounds <= 2 && perm.F <= 24 && !TpcEnv::get().no_lean && !(rb && TpcEnv::get().rb_hash);
    if (lean) {
        const size_t lds = Bins3<uint64_t, QH_THREADS>::lds_bytes(pl.b1) + (size_t)(PT_THREADS + 1 + TPC_XW_MAX) * 12 + 160 + (size_t)QT_MAXK * 80 + 64;
#define TPC_QHASH2_GO(G, S, H, X)                                                                                                           \
    do {                                                                                                                                    \
        (void)hipFuncSetAttribute((const void *)k_q_hash2<G, S, H, X>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        hipLaunchKernelGGL((k_q_hash2<G, S, H, X>), dim3(pl.nwg1), dim3(QH_THREADS), lds, a.stream, pl.b1, a.P, a.tab, a.bases, a.nmask, a.n_text, \
                           pl.tile0, pl.n_tiles, pl.pos_per_round, lo, hi, pl.buf1, pl.cnt1, pl.cap1, ovf, perm, sh,                          \
                           pl.tile0_global * (uint64_t)(PT_THREADS * TPC_RUN), rmask, pl.fmt == 6 ? pl.tiles_per_wg : 0u,                  \
                           reinterpret_cast<const uint16_t *>(a.per_qs));  /* (global word index: the sharded variants skip too) */          \
    } while (0)
#define TPC_QHASH2_GS(H, X)                                                                                                                 \
    do {                                                                                                                                    \
        if (pl.world > 1) { if (gated) TPC_QHASH2_GO(true, true, H, X); else TPC_QHASH2_GO(false, true, H, X); }                            \
        else { if (gated) TPC_QHASH2_GO(true, false, H, X); else TPC_QHASH2_GO(false, false, H, X); }                                       \
    } while (0)
        if (a.P.L > 32) { if (pl.b1 >= 9) TPC_QHASH2_GS(true, true); else TPC_QHASH2_GS(false, true); }
        else { if (pl.b1 >= 9) TPC_QHASH2_GS(true, false); else TPC_QHASH2_GS(false, false); }
#undef TPC_QHASH2_GS
#undef TPC_QHASH2_GO
        return;
    }
    const size_t lds = (rb ? RBins<uint64_t, QH_THREADS>::lds_bytes(pl.b1) : Bins3<uint64_t, QH_THREADS>::lds_bytes(pl.b1)) +
                       (size_t)(PT_THREADS + 1 + TPC_XW_MAX) * 12 + (size_t)5 * 16 + 64;
#define TPC_QHASH_GO(G, S, R)                                                                                                               \
    do {                                                                                                                                    \
        (void)hipFuncSetAttribute((const void *)k_q_hash<G, S, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
        hipLaunchKernelGGL((k_q_hash<G, S, R>), dim3(pl.nwg1), dim3(QH_THREADS), lds, a.stream, pl.b1, a.P, a.tab, a.bases, a.nmask, a.n_text,  \
                           pl.tile0, pl.n_tiles, pl.pos_per_round, pl.sub_rounds, lo, hi, pl.buf1, pl.cnt1, pl.cap1, ovf, perm, sh,              \
                           pl.tile0_global * (uint64_t)(PT_THREADS * TPC_RUN), rmask);                                                      \
    } while (0)
#define TPC_QHASH_GS(R)                                                                                                                     \
    do {                                                                                                                                    \
        if (pl.world > 1) { if (gated) TPC_QHASH_GO(true, true, R); else TPC_QHASH_GO(false, true, R); }                                    \
        else { if (gated) TPC_QHASH_GO(true, false, R); else TPC_QHASH_GO(false, false, R); }                                               \
    } while (0)
    if (rb) TPC_QHASH_GS(true); else TPC_QHASH_GS(false);
#undef TPC_QHASH_GS
#undef TPC_QHASH_GO
}

// ------------------------------------------------------------------------------------------ periodic windows
// A homopolymer, a dinucleotide tract, a telomere sends the SAME six probes (and the same q insert addresses) from hundreds of positions in a
// row: bursts of identical entries for one ring of one level-2 workgroup, in every such tract of the input -- m2r's tracts, 0.13 % of its
// text, cost the two split kernels 2.4 ms (DESIGN_HISTORY.md, round 5; four attempts to make the overflow path cheap enough failed).  They
// are removed at the source instead.  Both passes are functions of a window of the text: the first-pass verdict of the vertex at i of the
// k + 2 characters T[i - 1 .. i + k] (VE.h:633-674), the insert of its out-edge of T[i .. i + k] (VE.h:1035-1092).  When such a window
// equals the one p positions earlier (p = 1 .. 63: homopolymers, every microsatellite unit, the telomere hexamer, minisatellite units up to 63 bp), character for character and all of them
// definite, the position repeats that one's work: its insert is dropped (per_i; OR is idempotent), and its probes are dropped (per_qs) and
// its mark copied from position i - p once the verification is done (k_periodic_copy; p in six bit planes).  The masks depend on the text
// and k alone: built once per upload -- a first launch without outputs only says whether there is anything to skip at all.
// Thread = one word of 32 positions.  c_p(j) = length of the run of j' <= j with T[j'] == T[j' - p], both definite; window ending at j = i + k.
constexpr int PER_MAXP = TPC_PER_MAXP;      // periods 1 .. 63 (round 6; 6 until then: minisatellite units of 7 .. 60 bp cost the 62-genome text + 12 %, profiles/r06_m2r2.txt)
constexpr int PER_PLANES = TPC_PER_PLANES;  // the distance p of a copying position in six bit planes
constexpr int PER_CBITS = 10;               // bit-sliced run counters: runs up to 1023 >= k + 2 for every supported k (<= 603)
// All 63 periods at once, bit-sliced over a 64-bit word (bit p - 1 = period p): H0 / H1 / HN hold the two code bits and the N flag of the
// 63 characters before j, E = the periods whose character p back equals T[j] (both definite), C[b] = bit b of every period's run counter
// (incremented where E, cleared where not, saturating), and "run >= k + 2" is a bit-sliced comparison: ~130 word operations per
// character for all periods, where a counter per period took 5 per period.
// A cheap NECESSARY condition for a word of 32 positions to hold any periodic one (round 6: the detector below costs ~16 K word operations
// per 32 positions and nearly every word of a genome fails this test in ~2 K).  A position i is flagged only if text[t] == text[t - p] for
// all t in [i, i + k] and some p <= 63 (k + 1 definite characters, none of text[i - p .. i + k] an N).  Tile the line with blocks of
// B = min(16, (k + 2) / 2) characters from s0: every stretch of k + 1 >= 2B - 1 characters that starts at or after s0 contains a whole block,
// so a flagged position implies a block [a, a + B) with text[a .. a + B) == text[a - p .. a + B - p).  The blocks tested: those inside
// [first, first + 31 + k], the union of the word's windows -- or, when k >= 31 + 2B - 1, inside [first + 31, first + k], which every one of
// them contains.  Anything doubtful (an N or the end of what this context holds within 63 characters, k < 18) passes: the detector decides.
__device__ __forceinline__ bool periodic_may_flag(const uint64_t *__restrict__ bases, const uint32_t *__restrict__ nmask, int k, int64_t first, int64_t lo, int64_t hi)
{
    const int B = min(16, (k + 2) / 2);
    if (B < 10) return true;
    int64_t s0 = first, s1 = first + 31 + k;
    if (k >= 31 + 2 * B - 1) { s0 = first + 31; s1 = first + k; }
    const uint32_t M = B == 16 ? 0xFFFFFFFFu : (1u << (2 * B)) - 1u;
    for (int64_t a = s0; a + B - 1 <= s1; a += B) {
        if (a - 63 < lo || a + B > hi) return true;
        const int64_t wl = (a - 63) >> 5;
        const uint32_t off = (uint32_t)((a - 63) & 31), sh = 2u * off;
        const bool four = off + 63u + (uint32_t)B > 96u;  // the block's last character lies in a fourth word (never read beyond it: a context may hold a window of the text)
        const uint64_t b0 = bases[wl], b1 = bases[wl + 1], b2 = bases[wl + 2], b3 = four ? bases[wl + 3] : 0ull;
        const uint64_t nlo = (uint64_t)nmask[wl] | ((uint64_t)nmask[wl + 1] << 32), nhi = (uint64_t)nmask[wl + 2] | ((uint64_t)(four ? nmask[wl + 3] : 0u) << 32);
        // characters a - 63 .. a + B - 1 as bits 0 .. 2 (63 + B) - 1 of R2:R1:R0; their N flags as bits 0 .. 62 + B of n1:n0
        uint64_t R0 = sh ? (b0 >> sh) | (b1 << (64u - sh)) : b0, R1 = sh ? (b1 >> sh) | (b2 << (64u - sh)) : b1, R2 = sh ? (b2 >> sh) | (b3 << (64u - sh)) : b2;
        const uint64_t n0 = off ? (nlo >> off) | (nhi << (64u - off)) : nlo, n1 = nhi >> off;
        if (n0 | (n1 & ((1ull << (B - 1)) - 1ull))) return true;
        const uint32_t X = (uint32_t)((R1 >> 62) | (R2 << 2)) & M;  // the block itself: characters 63 .. 62 + B
        for (int p = PER_MAXP; p >= 1; p--) {                         // R0's low bits: the block p characters back
            if (((uint32_t)R0 & M) == X) return true;
            R0 = (R0 >> 2) | (R1 << 62); R1 = (R1 >> 2) | (R2 << 62); R2 >>= 2;
        }
    }
    return false;
}

__global__ void __launch_bounds__(256) k_periodic_build(const uint64_t *__restrict__ bases, const uint32_t *__restrict__ nmask, uint64_t n_text, int k, uint32_t *__restrict__ qs,
                                                        uint32_t *__restrict__ qd, uint64_t stride, uint32_t *__restrict__ ins, uint64_t w_begin, uint64_t n_words, uint64_t pos_lo,
                                                        uint64_t pos_hi, uint32_t *any)
{   // words [w_begin, n_words); characters outside [pos_lo, pos_hi) -- a context that holds only its window of the text -- count as N
    // qs == nullptr: detection only.  any[0] / any[1]: set when some position copies its verdict / drops its insert
    // qd: the distance p of a copying position in PER_PLANES bit planes (qd + b * stride)
    const uint64_t w = w_begin + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= n_words) return;
    uint32_t oqs = 0, oin = 0, od[PER_PLANES];
#pragma unroll
    for (int b = 0; b < PER_PLANES; b++) od[b] = 0;
    const int64_t first = (int64_t)(w << 5);
    if ((uint64_t)first < n_text && periodic_may_flag(bases, nmask, k, first, (int64_t)pos_lo, (int64_t)(pos_hi < n_text ? pos_hi : n_text))) {
        auto ch_at = [&](int64_t j) { return j >= (int64_t)pos_lo && (uint64_t)j < pos_hi && (uint64_t)j < n_text ? tpc_text_char(bases, nmask, (uint64_t)j) : 4; };
        const int64_t j0 = first - 2;  // (run lengths only matter up to k + 2: starting k + 2 characters before the first window's end is exact)
        const uint64_t pmask = (1ull << PER_MAXP) - 1ull;
        uint64_t H0 = 0, H1 = 0, HN = 0, C[PER_CBITS];
#pragma unroll
        for (int b = 0; b < PER_CBITS; b++) C[b] = 0;
        for (int p = PER_MAXP; p >= 1; p--) {  // bit p - 1 = the character p before j0
            const int ch = ch_at(j0 - p);
            H0 = (H0 << 1) | (uint64_t)(ch & 1); H1 = (H1 << 1) | (uint64_t)((ch >> 1) & 1); HN = (HN << 1) | (uint64_t)(ch >= 4);
        }
        // (the loop above shifted the NEAREST character in last: bit 0 = one back ... bit 62 = 63 back)
        const uint32_t T2 = (uint32_t)(k + 2), T1 = (uint32_t)(k + 1);
        for (int64_t j = j0; j <= first + 31 + k; j++) {
            const int ch = ch_at(j);
            const uint64_t c0 = (uint64_t)(ch & 1), c1 = (uint64_t)((ch >> 1) & 1), n = (uint64_t)(ch >= 4);
            const uint64_t E = n ? 0ull : (~(H0 ^ (0ull - c0)) & ~(H1 ^ (0ull - c1)) & ~HN & pmask);
            uint64_t sat = ~0ull;
#pragma unroll
            for (int b = 0; b < PER_CBITS; b++) sat &= C[b];
            uint64_t carry = E & ~sat;  // (a saturated counter stays where it is while its run goes on)
#pragma unroll
            for (int b = 0; b < PER_CBITS; b++) { const uint64_t t = C[b] & carry; C[b] = ((C[b] ^ carry) | (C[b] & sat)) & E; carry = t; }
            H0 = ((H0 << 1) | c0) & pmask; H1 = ((H1 << 1) | c1) & pmask; HN = ((HN << 1) | n) & pmask;
            const int64_t i = j - k;
            if (i < first) continue;
            uint64_t G2 = ~0ull, G1 = ~0ull;  // periods whose run reaches k + 2 / k + 1
#pragma unroll
            for (int b = 0; b < PER_CBITS; b++) {
                G2 = ((T2 >> b) & 1u) ? (C[b] & G2) : (C[b] | G2);
                G1 = ((T1 >> b) & 1u) ? (C[b] & G1) : (C[b] | G1);
            }
            G2 &= pmask; G1 &= pmask;
            const uint32_t bit = 1u << (uint32_t)(i - first);
            if (G1) oin |= bit;
            // the position it copies from lies in the same 512-word tile, and the first PER_MAXP positions of a tile always probe: a run of
            // copying positions never crosses a tile (batches and ranks are made of tiles) and k_periodic_copy's walks end there
            if (G2 && ((uint32_t)i & (uint32_t)(PT_THREADS * TPC_RUN - 1)) >= (uint32_t)PER_MAXP) {
                const uint32_t d = (uint32_t)__ffsll((long long)G2);  // (the smallest period wins)
                oqs |= bit;
#pragma unroll
                for (int b = 0; b < PER_PLANES; b++) if ((d >> b) & 1u) od[b] |= bit;
            }
        }
    }
    if (qs) {
        qs[w] = oqs; ins[w] = oin;
#pragma unroll
        for (int b = 0; b < PER_PLANES; b++) qd[w + (uint64_t)b * stride] = od[b];
    }
    if (oqs) any[0] = 1u;
    if (oin) any[1] = 1u;
}

// mark(i) = mark(i - p(i)) at every copying position.  A segment = a maximal stretch of positions in which no PER_MAXP consecutive ones
// probe for themselves; it starts at a copying position whose PER_MAXP predecessors all probed (their marks are final) and is walked by one
// thread, so every position's source -- inside the segment or one of those predecessors -- is known when it is needed.
__global__ void __launch_bounds__(256) k_periodic_copy(uint32_t *__restrict__ rmask, const uint32_t *__restrict__ qs, const uint32_t *__restrict__ qd, uint64_t stride, uint64_t n_words)
{
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= n_words) return;
    const uint32_t S = qs[w];
    if (S == 0u) return;
    // the copying positions among the 64 before this word (bit t = position 32 w - 64 + t)
    const uint64_t prev = (w ? (uint64_t)qs[w - 1] << 32 : 0ull) | (w > 1 ? (uint64_t)qs[w - 2] : 0ull);
    uint32_t starts = 0;
    for (uint32_t m = S; m; m &= m - 1u) {
        const uint32_t b = (uint32_t)__ffs((int)m) - 1u;
        // any copying position among the PER_MAXP = 63 before 32 w + b: those of this word below b, those of `prev` from bit b + 1 up
        const bool before = (S & ((1u << b) - 1u)) != 0u || (prev >> (b + 1u)) != 0ull;
        if (!before) starts |= 1u << b;
    }
    while (starts) {
        const uint32_t b = (uint32_t)__ffs((int)starts) - 1u;
        starts &= starts - 1u;
        uint64_t i = (w << 5) + b;  // (>= PER_MAXP: the first positions of a tile never copy)
        uint64_t mh = 0;            // bit t - 1 = mark(i - t)
        for (int t = 1; t <= PER_MAXP; t++) mh |= (uint64_t)((rmask[(i - t) >> 5] >> ((i - t) & 31u)) & 1u) << (t - 1);
        // the walk keeps the words of the 32 positions at hand in registers (a 500-position tract is 16 word fetches, not 2500 loads one
        // after the other: the kernel lasts as long as its longest walk)
        uint64_t cw = i >> 5;
        uint32_t sw = S, dpl[PER_PLANES], rw = rmask[cw], add = 0;
#pragma unroll
        for (int q = 0; q < PER_PLANES; q++) dpl[q] = qd[cw + (uint64_t)q * stride];
        for (int zeros = 0;; i++) {
            if ((i >> 5) != cw) {
                if (add) atomicOr(&rmask[cw], add);
                cw = i >> 5;
                add = 0;
                if (cw >= n_words) break;
                sw = qs[cw]; rw = rmask[cw];
#pragma unroll
                for (int q = 0; q < PER_PLANES; q++) dpl[q] = qd[cw + (uint64_t)q * stride];
            }
            const uint32_t bi = (uint32_t)i & 31u;
            uint32_t m;
            if ((sw >> bi) & 1u) {
                zeros = 0;
                uint32_t d = 0;
#pragma unroll
                for (int q = 0; q < PER_PLANES; q++) d |= ((dpl[q] >> bi) & 1u) << q;
                m = (uint32_t)(mh >> (d - 1u)) & 1u;
                add |= m << bi;
            } else {
                if (++zeros == PER_MAXP) break;
                m = (rw >> bi) & 1u;  // (a probing position: its mark is final, and no walk ever sets it)
            }
            mh = ((mh << 1) | (uint64_t)m) & ((1ull << PER_MAXP) - 1ull);
        }
        if (add) atomicOr(&rmask[cw], add);
    }
}

// one level of k_q_split: log_nb1 bits already binned, log_nb2 bits binned here; nvw source regions per bucket
void launch_qsplit(const TpcLaunch &a, bool sharded, int log_nb1, int log_nb2, int low_bits, int loads, uint32_t nwg1, uint32_t wpb, uint32_t nvw,
                   const uint64_t *buf1, const uint32_t *cnt1, uint64_t cap1, uint64_t *buf2, uint32_t *cnt2, const uint64_t *off2, QOverflow ovf, PtShard sh,
                   uint32_t prev_wpb, int log_prev_nb2, const uint64_t *off1, unsigned grid, const uint64_t *own1 = nullptr, const uint32_t *owncnt1 = nullptr)
{
    const bool rb = q_use_rbins(log_nb2);
    const size_t lds_base = (rb ? RBins<uint64_t, QS_THREADS>::lds_bytes(log_nb2) : Bins3<uint64_t, QS_THREADS>::lds_bytes(log_nb2)) + ((size_t)8 << log_nb2) + 64 + 128;
    if (rb) loads = 4;  // the barrier-free rings have no round to outgrow
    uint32_t nreg_cap, sched_cap;
    size_t lds;
    pt_schedule_dims(nvw, wpb, cap1, (uint32_t)loads * QS_THREADS, lds_base, nreg_cap, sched_cap, lds);
#define TPC_QSPLIT_GO(S, R)                                                                                                                  \
    do {                                                                                                                                    \
        (void)hipFuncSetAttribute((const void *)k_q_split<S, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
        hipLaunchKernelGGL((k_q_split<S, R>), dim3(grid), dim3(QS_THREADS), lds, a.stream, log_nb1, log_nb2, a.P.L, low_bits, loads, nwg1, wpb, buf1, cnt1, cap1,   \
                           buf2, cnt2, off2, ovf, sh, prev_wpb, log_prev_nb2, nreg_cap, sched_cap, off1, own1, owncnt1);                     \
    } while (0)
    if (sharded) { if (rb) TPC_QSPLIT_GO(true, true); else TPC_QSPLIT_GO(true, false); }
    else { if (rb) TPC_QSPLIT_GO(false, true); else TPC_QSPLIT_GO(false, false); }
#undef TPC_QSPLIT_GO
}

// the level-2 binning of a fmt-6 plan: 8-byte entries in, blocked 48-bit lines + group boundaries out (k_q_split<false, false, true>)
void launch_qsplit6(const TpcLaunch &a, const TpcQPlan &pl, QOverflow ovf)
{
    const PtShard sh{0, 1};
    const size_t lds_base = BinsP<PFmt6, QS_THREADS>::lds_bytes(pl.b2) + ((size_t)8 << pl.b2) + 64 + 128;
    uint32_t nreg_cap, sched_cap;
    size_t lds;
    pt_schedule_dims(pl.nwg1, pl.wpb, pl.cap1, (uint32_t)pl.loads * QS_THREADS, lds_base, nreg_cap, sched_cap, lds);
    (void)hipFuncSetAttribute((const void *)k_q_split<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_q_split<false, false, true>), dim3((1u << pl.b1) * pl.wpb), dim3(QS_THREADS), lds, a.stream, pl.b1, pl.b2, a.P.L, pl.slice_bits, pl.loads, pl.nwg1, pl.wpb,
                       pl.rbuf1, pl.rcnt1, pl.cap1, pl.buf2, pl.cnt2, pl.off2, ovf, sh, 0u, 0, nreg_cap, sched_cap, pl.roff1, (const uint64_t *)nullptr, (const uint32_t *)nullptr,
                       pl.bnd, pl.n_groups, pl.tiles_per_wg, (uint32_t)pl.n_tiles, pl.pb2);
}

#endif  // part 0 only
template <int Q>
void launch_qverify(const TpcLaunch &a, const TpcQPlan &pl, uint32_t *rmask)
{
    const uint64_t gbase = pl.tile0_global * (uint64_t)(PT_THREADS * TPC_RUN);
    const size_t table = (size_t)(a.P.k + 1) * 4 * Q * 16;  // k_q_verify2's letter table
    if (a.P.k <= 31 && table <= 48 * 1024 && !TpcEnv::get().no_lean) {
        const bool lazy = !TpcEnv::get().verify_eager;  // (TPC_VERIFY_LAZY=0, measurements: all Q - 1 probes at once)
        if (lazy) {
            (void)hipFuncSetAttribute((const void *)k_q_verify2<Q, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)table);
            hipLaunchKernelGGL((k_q_verify2<Q, true>), dim3(256, QS_LISTS), dim3(256), table, a.stream, a.P, a.tab, a.bases, a.filter, pl.surv, pl.surv_cur, pl.surv_cap, gbase,
                               rmask);
        } else {
            (void)hipFuncSetAttribute((const void *)k_q_verify2<Q, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)table);
            hipLaunchKernelGGL((k_q_verify2<Q, false>), dim3(256, QS_LISTS), dim3(256), table, a.stream, a.P, a.tab, a.bases, a.filter, pl.surv, pl.surv_cur, pl.surv_cap, gbase,
                               rmask);
        }
        return;
    }
    hipLaunchKernelGGL((k_q_verify<Q>), dim3(256, QS_LISTS), dim3(256), 0, a.stream, a.P, a.tab, a.bases, a.filter, pl.surv, pl.surv_cur, pl.surv_cap, gbase, rmask);
}

}  // namespace

#if TPC_QPARTITION_PART == 0
bool tpc_qpart_plan(int L, int slice_bits, uint64_t n_tiles, double frac, TpcQPlan &pl, int levels)
{
    return tpc_qpart_plan_sharded(L, slice_bits, n_tiles, frac, 0, 1, pl, levels, false, true);
}

// n_tiles: the tiles THIS rank hashes; the level-2 regions cover the slices this rank owns and are sized
// for the entries of all ranks
static bool qpart_plan_compute(int L, int slice_bits, uint64_t n_tiles, double frac, uint32_t rank, uint32_t world, TpcQPlan &pl, int levels, bool tight, bool packed);

bool tpc_qpart_plan_sharded(int L, int slice_bits, uint64_t n_tiles, double frac, uint32_t rank, uint32_t world, TpcQPlan &pl, int levels, bool tight, bool packed)
{
    // A plan costs ~0.5 ms of host time (the densest-bucket search and the per-slice region table: 65536 entries with a square root
    // each) and every pass of every round asks for the same one while the GPU sits idle behind the previous call's synchronisation:
    // the last plan per thread is kept (the device pointers of a plan are filled in by the caller afterwards and are not part of it).
    struct Key { int L, slice_bits, levels, q6; uint64_t n_tiles; double frac; uint32_t rank, world; bool tight, packed; };
    static thread_local Key last{};
    static thread_local TpcQPlan last_pl;
    static thread_local bool have = false, last_ok = false;
    const Key k{L, slice_bits, levels, tpc_test_q6_pb2 + (tpc_test_tight_pinch << 8), n_tiles, frac, rank, world, tight, packed};
    if (have && k.L == last.L && k.slice_bits == last.slice_bits && k.levels == last.levels && k.q6 == last.q6 && k.n_tiles == last.n_tiles && k.frac == last.frac &&
        k.rank == last.rank && k.world == last.world && k.tight == last.tight && k.packed == last.packed) {
        if (last_ok) pl = last_pl;
        return last_ok;
    }
    last_ok = qpart_plan_compute(L, slice_bits, n_tiles, frac, rank, world, pl, levels, tight, packed);
    last = k; have = true;
    if (last_ok) last_pl = pl;
    return last_ok;
}

static bool qpart_plan_compute(int L, int slice_bits, uint64_t n_tiles, double frac, uint32_t rank, uint32_t world, TpcQPlan &pl, int levels, bool tight, bool packed)
{
    pl = TpcQPlan();
    pl.rank = rank; pl.world = world;
    const uint64_t n_text = n_tiles * PT_THREADS * TPC_RUN;  // positions of this batch of 512-word tiles
    const int F = L - slice_bits;
    if (F < 2 || slice_bits < 6 || slice_bits > 20) return false;
    if (n_text > (1ull << 30)) return false;  // an entry holds a 30-bit position relative to the batch
    pl.slice_bits = slice_bits;
    const bool three = levels == 3 || (levels == 0 && F > 18);  // as in tpc_part_plan_sharded
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
    if (pl.b1 > 9 || L - pl.b1 > 31) return false;
    if (world == 0 || (world & (world - 1)) || world > (1u << pl.b1)) return false;  // ranks own whole buckets
    pl.n_tiles = n_tiles;
    pl.tile0 = 0;
    pl.nwg1 = (uint32_t)std::min<uint64_t>(256, pl.n_tiles);
    // level-2 workgroups per bucket: one per bucket is fastest once the buckets alone fill the chip (measured 1 / 2 / 4 / 8 on M2:
    // 45.9 / 46.3 / 47.9 / 51.3 ms per step); fewer local buckets (small filters, sharded filters) are split further
    pl.wpb = std::max<uint32_t>(1, std::min<uint32_t>(8, (256u * world) >> pl.b1));
    const int cap = (PT_BIN_BYTES / 8) >> pl.b1;
    const int budget = std::max(1, (1 << pl.b1) * (cap - 16) * 5 / 8);  // entries per round
    // k_q_hash runs 1024 threads x 16 positions; frac = expected share of vertices inside the round's range
    const int ppr = (int)(budget / (1024 * 6 * std::max(frac, 1.0 / 64)));
    pl.pos_per_round = ppr >= 16 ? 16 : ppr >= 8 ? 8 : ppr >= 4 ? 4 : ppr >= 2 ? 2 : 1;
    pl.sub_rounds = ppr >= 1 ? 1 : std::min(16, (1024 * 6 + budget - 1) / budget);
    // k_q_split entries per thread per round for a level with 2^bits bins (the last level's bins see a 2x skew)
    auto loads_for = [](int bits) {
        const int c = (PT_BIN_BYTES / 8) >> bits;
        return std::max(1, std::min(4, std::max(1, (1 << bits) * (c - 16) * 5 / 8) * 9 / 8 / QS_THREADS / 2));
    };
    pl.loads = loads_for(pl.b3 ? pl.b2 : pl.b1);
    // A gated round is not uniform over the slices: the in-edge c + v of a vertex has the forward hash H(v) ^ const (cyclichash.h: hash_prepend),
    // so the vertices of a hash range put their in-edges into the image of that range -- an eighth of the slices take ~2.5 x their
    // share, the rings of those bins fill inside a round (waves wait for the owner's copy, then the entries are lost to the overflow
    // list: 4.9 M of 232 M and an 11 ms k_q_split for one range of eight on the 62-genome workload).  Half a round's entries keeps the
    // hot rings below their capacity.
    // (a range of half the hash space -- two rounds, two ranks: frac 0.58 -- is mild: its hot half takes 1.5 x; from a third down it is not)
    const bool gated_round = frac < 0.45;
    if (gated_round) pl.loads = std::max(1, pl.loads / 2);
    if (TpcEnv::get().gated_loads && gated_round) pl.loads = std::max(1, std::min(4, TpcEnv::get().gated_loads));  // measurements
    const double a_max = 6.0 * (double)n_text * 1.02 + 4096;
    // a workgroup takes ceil(n_tiles / nwg1) tiles: with few tiles per workgroup the busiest one holds well over the mean
    const uint64_t tiles_wg = (pl.n_tiles + pl.nwg1 - 1) / pl.nwg1;
    const double share1 = std::min(1.0, (double)tiles_wg / (double)std::max<uint64_t>(pl.n_tiles, 1));
    const double a_exp = TpcEnv::get().gated_full ? a_max : a_max * std::min(1.0, std::max(frac, 1.0 / 64));  // a gated round: the share of the vertices inside its range (as tpc_part_plan_sharded)
    const double avg1 = a_exp * share1 / (double)(1 << pl.b1);
    const PtPerm pm = pt_make_perm(slice_bits, F);
    pl.perm_mult = pm.mult; pl.perm_inv = pm.inv;
    const double avg1t = avg1 * pt_bucket_peak(pm, F, pl.b1);  // tight, as in tpc_part_plan_sharded: every query address is a function-0 address
    pl.cap1 = ((uint64_t)(tight ? avg1t + 6 * std::sqrt(avg1t) + 128 : avg1 * 1.3 + 8 * std::sqrt(avg1) + 128) + 15) & ~15ull;
    pl.ovf_cap = (uint64_t)(a_max / 32) + 65536;
    if (tight && tpc_test_tight_pinch > 0) {  // (tests, as in tpc_part_plan_sharded: a tight plan that must fail)
        pl.cap1 = std::max<uint64_t>(16, ((uint64_t)(avg1t * tpc_test_tight_pinch / 100.0) + 15) & ~15ull);
        pl.ovf_cap = 64;
    }
    pl.surv_cap = (uint64_t)((double)n_text * 1.0 / QS_LISTS) + 65536;  // per sub-list; beyond it the direct kernel takes over (repeat-rich input: 0.83 survivors per position)
    // Region sizes of the LAST level: a region is one filter slice, and every query address is a function-0
    // address whose density over the slices falls linearly from 2x to 0 (tpc_bins.h).  With three levels the
    // middle regions each collect 2^b3 slices spread over the whole filter by the permutation: uniform.
    const double S = (double)(1ull << F);
    // log_mid: bits of the bucket index that sit below the rank-interleaved level-1 bits (0 for the two-level table, b2 for the third level)
    auto slice_table = [&](std::vector<uint64_t> &off, uint64_t nreg, int log_last, uint32_t wpb_last, double avg, int log_mid) {
        off.resize(nreg + 1);
        uint64_t o = 0;
        for (uint64_t r = 0; r < nreg; r++) {
            const uint32_t bl = (uint32_t)(r / ((uint64_t)wpb_last << log_last)), bb = (uint32_t)(r & ((1u << log_last) - 1));
            const uint32_t gb = world > 1 ? ((((bl >> log_mid) * world + rank) << log_mid) | (bl & ((1u << log_mid) - 1u))) : bl;  // global bucket
            const uint32_t s = pm.slice_of((gb << log_last) | bb);  // the filter slice behind this region
            const double d = avg * 2.0 * (1.0 - ((double)s + 0.5) / S) + avg * 0.02;
            off[r] = o;
            o += ((uint64_t)(d * 1.25 + 8 * std::sqrt(d) + 96) + 15) & ~15ull;
        }
        off[nreg] = o;
        return o;
    };
    const uint64_t nreg2 = (uint64_t)((1u << pl.b1) / world) * pl.wpb * (1u << pl.b2);  // local regions
    // ... but only at level 1, whose buckets mix the slices evenly: the slice regions of a gated round are sized for ALL the
    // entries, because the round's hot slices (see pl.loads above) take ~2.5 x their share of what is there (sized for the share alone,
    // the regions of one range of eight lost 25 M of 232 M entries to the overflow list)
    // entries of all ranks over all regions -- dealt to the wpb level-2 workgroups of a bucket as whole level-1 regions, so with fewer
    // regions than workgroups a region holds up to ceil(regions / wpb) / regions of its slice's entries (as tpc_part_plan_sharded)
    const uint64_t nvw = (uint64_t)pl.nwg1 * world;
    const double deal = (double)((nvw + pl.wpb - 1) / pl.wpb) / (double)nvw * (double)pl.wpb;
    const double avg2 = a_max * world * deal / ((double)nreg2 * world);
    pl.wpb3 = 1;
    pl.loads3 = pl.loads;
    if (pl.b3) {
        pl.cap2 = ((uint64_t)(avg2 * 1.3 + 8 * std::sqrt(avg2) + 128) + 15) & ~15ull;
        pl.off2_host.resize(nreg2 + 1);
        for (uint64_t r = 0; r <= nreg2; r++) pl.off2_host[r] = r * pl.cap2;
        pl.buf2_entries = nreg2 * pl.cap2;
        const uint64_t nreg3 = ((uint64_t)pl.wpb3 << (pl.b1 + pl.b2 + pl.b3)) / world;  // local regions
        pl.buf3_entries = slice_table(pl.off3_host, nreg3, pl.b3, pl.wpb3, a_max / (double)nreg3, pl.b2);
        pl.loads3 = loads_for(pl.b3);
        if (gated_round) pl.loads3 = std::max(1, pl.loads3 / 2);
    } else {
        pl.cap2 = 0;
        pl.buf2_entries = slice_table(pl.off2_host, nreg2, pl.b2, pl.wpb, avg2, 0);
        pl.off3_host.clear();
        pl.buf3_entries = 0;
    }
    // ---- 48-bit level-2 entries in blocked lines (tpc_qpart6.h): one rank, two levels, flush-per-round bins at level 2 that do not
    // span waves.  TPC_ENTRY_FMT=legacy (read once per process) keeps the 8-byte entries for A/B measurements.
    pl.fmt = 0;
    static const bool legacy_fmt = [] { const char *e = getenv("TPC_ENTRY_FMT"); return e && (e[0] == 'l' || e[0] == 'i'); }();  // "legacy", or "insert": only the insert's new format
    int PB2 = std::min(44 - slice_bits, 30);  // position bits below the group (a level-1 entry holds a 30-bit position)
    if (tpc_test_q6_pb2 >= 14 && tpc_test_q6_pb2 < PB2) PB2 = tpc_test_q6_pb2;  // tests: many groups on a small input
    const uint64_t tpw = (n_tiles + pl.nwg1 - 1) / pl.nwg1;  // level 1: contiguous tiles per workgroup
    const uint64_t groups = ((n_tiles * (uint64_t)(PT_THREADS * TPC_RUN)) + (1ull << PB2) - 1) >> PB2;
    const bool no_lean = TpcEnv::get().no_lean;  // (measurements: the generic hash kernel takes its tiles interleaved)
    // (512 bins at level 2 -- f = 37, 38 -- stay on the 8-byte barrier-free rings: 40-entry rings allow rounds of 3072 entries only, and
    //  the 62-genome text at f = 38 measured 40.6 ms per step against 39.6; TPC_P6_MAXB2=9 lifts the gate for measurements)
    static const int max_b2 = [] { const char *e = getenv("TPC_P6_MAXB2"); return e ? atoi(e) : 8; }();
    if (packed && !legacy_fmt && !no_lean && world == 1 && !three && pl.b2 >= 4 && pl.b2 <= max_b2 && pl.b1 <= 9 && pl.sub_rounds <= 2 && F <= 24 && PB2 >= 14 &&
        tpw * (uint64_t)(PT_THREADS * TPC_RUN) <= (1ull << PB2) && groups >= 1 && groups <= 64 && [&] {
            // Q6Res::raw keeps a survivor's index in its level-2 region above 4 + PB2 bits of a 54-bit staged id (SurvStage::ID_BITS): the
            // largest region, in padded lines x 20 entries, must stay below 2^(50 - PB2) or the index would spill into the bucket bits
            uint64_t most = 0;
            for (uint64_t r = 0; r < nreg2; r++) most = std::max(most, pl.off2_host[r + 1] - pl.off2_host[r]);
            return ((most + PFmt6::GROUP - 1) / PFmt6::GROUP + 1) * PFmt6::GROUP < (1ull << (50 - PB2));
        }()) {
        pl.fmt = 6;
        pl.tiles_per_wg = (uint32_t)tpw;
        pl.n_groups = (uint32_t)groups;
        pl.pb2 = (uint32_t)PB2;
        // ring rounds of the level-2 bins: CAP = 20 entries x groups per bin; a round = loads x 1024 entries; the last level's bins see a 2x skew
        const int cap_s = (int)BinsP<PFmt6, QS_THREADS>::cap_for(pl.b2);
        static const int loads_cap = [] { const char *e = getenv("TPC_P6_LOADS"); return e ? std::max(1, std::min(5, atoi(e))) : 5; }();  // (measurements)
        pl.loads = std::max(1, std::min(loads_cap, std::max(1, (1 << pl.b2) * (cap_s - PFmt6::GROUP) * 5 / 8) * 9 / 8 / QS_THREADS / 2));
        if (gated_round) pl.loads = std::max(1, pl.loads / 2);
        // level-2 regions in lines
        uint64_t o = 0;
        for (uint64_t r = 0; r < nreg2; r++) {
            const uint64_t e = pl.off2_host[r + 1] - pl.off2_host[r];  // entries the 8-byte path gives this slice
            pl.off2_host[r] = o;
            o += (e + PFmt6::GROUP - 1) / PFmt6::GROUP;
        }
        pl.off2_host[nreg2] = o;
        pl.buf2_entries = o;  // lines
    }
    return true;
}

size_t tpc_qpart_bytes(const TpcQPlan &pl, int which)
{
    if (pl.fmt == 6) {
        switch (which) {
        case 2: return (size_t)pl.buf2_entries * PT_LINE;
        case 18: return ((size_t)(1u << pl.b1) * pl.wpb * (1u << pl.b2)) * pl.n_groups * 8;  // zone starts and zone ends
        }
    }
    switch (which) {
    case 0: return (size_t)pl.nwg1 * (1u << pl.b1) * pl.cap1 * 8;
    case 1: return (size_t)pl.nwg1 * (1u << pl.b1) * 4;
    case 2: return (size_t)pl.buf2_entries * 8;
    case 3: return ((size_t)((1u << pl.b1) / pl.world) * pl.wpb * (1u << pl.b2)) * 4;
    case 4: return pl.ovf_cap * 16;
    case 5: return 32 * sizeof(unsigned long long);
    case 6: return (size_t)QS_LISTS * pl.surv_cap * 8;
    case 7: return TPC_SURV_CUR_WORDS * sizeof(unsigned long long);  // 64 cursors, the overflow flag
    case 8: return pl.off2_host.size() * 8;
    case 9: return (size_t)pl.buf3_entries * 8;
    case 10: return pl.b3 ? (((size_t)pl.wpb3 << (pl.b1 + pl.b2 + pl.b3)) / pl.world) * 4 : 0;
    case 11: return pl.off3_host.size() * 8;
    }
    return 0;
}

int tpc_launch_query_part_hash(const TpcLaunch &a, const TpcQPlan &pl, uint32_t *rmask, uint64_t lo, uint64_t hi, bool gated)
{
    if (a.P.q < 1 || a.P.q > TPC_KERNEL_MAXQ) return -1;  // k_q_verify is instantiated for 1..16 functions
    launch_qhash(a, pl, gated, lo, hi, rmask);
    return 0;
}

int tpc_launch_query_part_split(const TpcLaunch &a, const TpcQPlan &pl)
{   // levels 2 (and 3) of the query's binning alone: the part of the lookup launches below that does not need the filter
    const PtShard sh{pl.rank, pl.world};
    QOverflow ovf{pl.ovf, pl.ovf_cur, pl.ovf_cap};
    if (pl.fmt == 6) { launch_qsplit6(a, pl, ovf); return 0; }
    const int low_bits = pl.slice_bits + pl.b3;
    launch_qsplit(a, pl.world > 1, pl.b1, pl.b2, low_bits, pl.loads, pl.nwg1, pl.wpb, pl.nwg1 * pl.world, pl.rbuf1, pl.rcnt1, pl.cap1, pl.buf2, pl.cnt2, pl.off2,
                  ovf, sh, 0u, 0, pl.roff1, (unsigned)(((1u << pl.b1) / pl.world) * pl.wpb), pl.rown1, pl.rowncnt1);
    if (pl.b3)
        launch_qsplit(a, false, pl.b1 + pl.b2, pl.b3, pl.slice_bits, pl.loads3, 0u, pl.wpb3, pl.wpb, pl.buf2, pl.cnt2, pl.cap2, pl.buf3, pl.cnt3, pl.off3,
                      ovf, sh, pl.wpb, pl.b2, nullptr, (unsigned)(((1u << (pl.b1 + pl.b2)) / pl.world) * pl.wpb3));
    return 0;
}

int tpc_launch_query_part_lookup(const TpcLaunch &a, const TpcQPlan &pl)
{
    const PtPerm perm{pl.slice_bits, pl.b1 + pl.b2 + pl.b3, pl.perm_mult, pl.perm_inv};
    const PtShard sh{pl.rank, pl.world};
    if (pl.fmt == 6) {
        if (!pl.presplit) launch_qsplit6(a, pl, QOverflow{pl.ovf, pl.ovf_cur, pl.ovf_cap});
        const size_t words = (size_t)1 << (pl.slice_bits - 5);
        const size_t lds = ((words + 3) & ~(size_t)3) * 4 + QL6_LDS;
        (void)hipFuncSetAttribute((const void *)k_q_lookup6, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_q_lookup6, dim3(tpc_slice_grid(1u << (pl.b1 + pl.b2))), dim3(PT_APPLY_THREADS), lds, a.stream, pl.slice_bits, pl.b2, pl.wpb, (const unsigned char *)pl.buf2,
                           pl.cnt2, pl.off2, pl.bnd, pl.n_groups, pl.pb2, a.filter, pl.surv, pl.surv_cur, pl.surv_cap, perm, pl.group_survivors ? 1 : 0, (uint32_t)(1u << (pl.b1 + pl.b2)));
        hipLaunchKernelGGL(k_q_ovf, dim3(1024), dim3(256), 0, a.stream, pl.ovf, pl.ovf_cur, pl.ovf_cap, a.filter, pl.surv, pl.surv_cur, pl.surv_cap, perm, sh, pl.b2);
        return 0;
    }
    if (!pl.presplit) {
        QOverflow ovf{pl.ovf, pl.ovf_cur, pl.ovf_cap};
        const int low_bits = pl.slice_bits + pl.b3;  // address bits below this level's bin index
        launch_qsplit(a, pl.world > 1, pl.b1, pl.b2, low_bits, pl.loads, pl.nwg1, pl.wpb, pl.nwg1 * pl.world, pl.rbuf1, pl.rcnt1, pl.cap1, pl.buf2, pl.cnt2, pl.off2,
                      ovf, sh, 0u, 0, pl.roff1, (unsigned)(((1u << pl.b1) / pl.world) * pl.wpb), pl.rown1, pl.rowncnt1);
        if (pl.b3)  // third level: bucket (b1, b2); the middle regions are uniform (cap2 entries each)
            launch_qsplit(a, false, pl.b1 + pl.b2, pl.b3, pl.slice_bits, pl.loads3, 0u, pl.wpb3, pl.wpb, pl.buf2, pl.cnt2, pl.cap2, pl.buf3, pl.cnt3, pl.off3,
                          ovf, sh, pl.wpb, pl.b2, nullptr, (unsigned)(((1u << (pl.b1 + pl.b2)) / pl.world) * pl.wpb3));
    }
    {
        const size_t words = (size_t)1 << (pl.slice_bits - 5);
        const size_t lds = ((words + 3) & ~(size_t)3) * 4 + QL_LDS;
        (void)hipFuncSetAttribute((const void *)k_q_lookup, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (pl.b3)
            hipLaunchKernelGGL(k_q_lookup, dim3(tpc_slice_grid((1u << (pl.b1 + pl.b2 + pl.b3)) / pl.world)), dim3(PT_APPLY_THREADS), lds, a.stream, pl.slice_bits, pl.b3, pl.wpb3, pl.buf3,
                               pl.cnt3, pl.off3, a.filter, pl.surv, pl.surv_cur, pl.surv_cap, perm, sh, pl.group_survivors ? 1 : 0, (uint32_t)((1u << (pl.b1 + pl.b2 + pl.b3)) / pl.world));
        else
            hipLaunchKernelGGL(k_q_lookup, dim3(tpc_slice_grid((1u << (pl.b1 + pl.b2)) / pl.world)), dim3(PT_APPLY_THREADS), lds, a.stream, pl.slice_bits, pl.b2, pl.wpb,
                               pl.buf2, pl.cnt2, pl.off2, a.filter, pl.surv, pl.surv_cur, pl.surv_cap, perm, sh, pl.group_survivors ? 1 : 0, (uint32_t)((1u << (pl.b1 + pl.b2)) / pl.world));
    }
    hipLaunchKernelGGL(k_q_ovf, dim3(1024), dim3(256), 0, a.stream, pl.ovf, pl.ovf_cur, pl.ovf_cap, a.filter, pl.surv, pl.surv_cur, pl.surv_cap, perm,
                       sh, pl.b2 + pl.b3);
    return 0;
}

// Fused tail of the query when the insert's apply was deferred (see tpc_capi.hip:flush_pending_apply; tpc_ctx.h): level-2 binning of
// the query, then k_apply_lookup over the insert's and the query's level-2 regions, then the overflow probes.
int tpc_launch_query_part_fused_lookup(const TpcLaunch &a, const TpcQPlan &pl, const TpcPartPlan &ipl, bool fresh, const uint64_t *iovf, const uint64_t *iovf_off, const TpcListSrc *lists)
{   // lists: set-bit lists to OR into every slice as well (the combined exchange; ipl.wpb may then be 0: no regions of its own)
    const TpcListSrc ls = lists ? *lists : TpcListSrc();
    if (ls.n_src && pl.world != 1) return -1;
    if (pl.b3 || ipl.b3 || pl.world != ipl.world || pl.rank != ipl.rank || pl.slice_bits != ipl.slice_bits || pl.b1 != ipl.b1 || pl.b2 != ipl.b2) return -1;
    if (pl.world > 1 && (pl.fmt == 6 || ipl.fmt2 == 3)) return -1;  // (the sharded passes keep the power-of-two entries)
    const PtPerm perm{pl.slice_bits, pl.b1 + pl.b2, pl.perm_mult, pl.perm_inv};
    const PtShard sh{pl.rank, pl.world};
    QOverflow ovf{pl.ovf, pl.ovf_cur, pl.ovf_cap};
    if (pl.fmt == 6) {
        if (!pl.presplit) launch_qsplit6(a, pl, ovf);
        const size_t words = (size_t)1 << (pl.slice_bits - 5);
        const size_t lds = ((words + 3) & ~(size_t)3) * 4 + QL6_LDS;
        const uint32_t al6_slices = 1u << (pl.b1 + pl.b2), al6_grid = tpc_slice_grid(al6_slices);  // long-lived workgroups, each taking every al6_grid-th slice
#define TPC_AL6_GO(I3)                                                                                                                                  \
    do {                                                                                                                                                \
        (void)hipFuncSetAttribute((const void *)k_apply_lookup6<I3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                             \
        hipLaunchKernelGGL(k_apply_lookup6<I3>, dim3(I3 ? al6_slices : al6_grid), dim3(PT_APPLY_THREADS), lds, a.stream, pl.slice_bits, pl.b2, ipl.wpb, (const unsigned char *)ipl.buf2, \
                           ipl.cnt2, (uint64_t)(I3 ? ipl.cap2 / PFmt3::GROUP : ipl.cap2 / 32), fresh ? 1 : 0, iovf, iovf_off, pl.wpb, (const unsigned char *)pl.buf2, pl.cnt2, \
                           pl.off2, pl.bnd, pl.n_groups, pl.pb2, a.filter, pl.surv, pl.surv_cur, pl.surv_cap, perm, pl.group_survivors ? 1 : 0, ls, al6_slices);   \
    } while (0)
        if (a.ev_lookup0) (void)hipEventRecord(a.ev_lookup0, a.stream);
        if (ls.n_src && (ipl.wpb != 0 || iovf_off)) return -1;  // (lists come alone: tpc_combine_import leaves no regions of this rank's own)
        if (ls.n_src) {  // lists alone (the combined exchange): its own instantiation, loads asked for up front; one short workgroup per slice
            // (as long-lived workgroups a rank's query took 4.14 instead of 3.51 ms at eight ranks: its ~12 us per slice are round trips,
            // and the end of one workgroup overlaps the start of the next only when they are separate workgroups)
            (void)hipFuncSetAttribute((const void *)k_apply_lookup6<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((k_apply_lookup6<false, true>), dim3(al6_slices), dim3(PT_APPLY_THREADS), lds, a.stream, pl.slice_bits, pl.b2, 0u, (const unsigned char *)nullptr,
                               (const uint32_t *)nullptr, (uint64_t)0, fresh ? 1 : 0, (const uint64_t *)nullptr, (const uint64_t *)nullptr, pl.wpb, (const unsigned char *)pl.buf2, pl.cnt2,
                               pl.off2, pl.bnd, pl.n_groups, pl.pb2, a.filter, pl.surv, pl.surv_cur, pl.surv_cap, perm, pl.group_survivors ? 1 : 0, ls, al6_slices);
        } else if (ipl.fmt2 == 3) TPC_AL6_GO(true); else TPC_AL6_GO(false);
        if (a.ev_lookup1) (void)hipEventRecord(a.ev_lookup1, a.stream);
#undef TPC_AL6_GO
        hipLaunchKernelGGL(k_q_ovf, dim3(1024), dim3(256), 0, a.stream, pl.ovf, pl.ovf_cur, pl.ovf_cap, a.filter, pl.surv, pl.surv_cur, pl.surv_cap, perm, sh, pl.b2);
        return 0;
    }
    if (ipl.fmt2 == 3) return -1;  // the 8-byte lookup reads 32-bit insert entries (the caller plans both passes with the same format switch)
    if (!pl.presplit) {
        launch_qsplit(a, pl.world > 1, pl.b1, pl.b2, pl.slice_bits, pl.loads, pl.nwg1, pl.wpb, pl.nwg1 * pl.world, pl.rbuf1, pl.rcnt1, pl.cap1, pl.buf2, pl.cnt2, pl.off2, ovf, sh, 0u, 0,
                      pl.roff1, ((1u << pl.b1) / pl.world) * pl.wpb, pl.rown1, pl.rowncnt1);
    }
    {
        const size_t words = (size_t)1 << (pl.slice_bits - 5);
        const size_t lds = ((words + 3) & ~(size_t)3) * 4 + QL_LDS;
        (void)hipFuncSetAttribute((const void *)k_apply_lookup, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (a.ev_lookup0) (void)hipEventRecord(a.ev_lookup0, a.stream);
        hipLaunchKernelGGL(k_apply_lookup, dim3(ls.n_src ? (1u << (pl.b1 + pl.b2)) / pl.world : tpc_slice_grid((1u << (pl.b1 + pl.b2)) / pl.world)), dim3(PT_APPLY_THREADS), lds, a.stream, pl.slice_bits, pl.b2, ipl.wpb, ipl.buf2, ipl.cnt2,
                           ipl.cap2, fresh ? 1 : 0, iovf, iovf_off, pl.wpb, pl.buf2, pl.cnt2, pl.off2, a.filter, pl.surv, pl.surv_cur, pl.surv_cap, perm, pl.group_survivors ? 1 : 0, sh, ls, (uint32_t)((1u << (pl.b1 + pl.b2)) / pl.world));
        if (a.ev_lookup1) (void)hipEventRecord(a.ev_lookup1, a.stream);
    }
    hipLaunchKernelGGL(k_q_ovf, dim3(1024), dim3(256), 0, a.stream, pl.ovf, pl.ovf_cur, pl.ovf_cap, a.filter, pl.surv, pl.surv_cur, pl.surv_cap, perm, sh, pl.b2);
    return 0;
}

#endif  // part 0 only
// The verification kernels are instantiated per number of hash functions: q = 5, the default every run of the CLI without -q uses, lives in
// this code object; the other fifteen in the second one (tpc_qpartition_x.o), which the runtime loads only when such a run happens.
int tpc_launch_query_verify_other(const TpcLaunch &a, const TpcQPlan &pl, uint32_t *rmask);
#if TPC_QPARTITION_PART == 0
int tpc_launch_query_verify(const TpcLaunch &a, const TpcQPlan &pl, uint32_t *rmask)
{
    if (a.P.q == 5) { launch_qverify<5>(a, pl, rmask); return 0; }
    return tpc_launch_query_verify_other(a, pl, rmask);
}

// Launches A-D on the stream (one rank, whole filter).  The caller zeroes ovf_cur / surv_cur first and
// reads both flags back afterwards: surv_cur[QS_LISTS] != 0 or ovf_cur[1] != 0 means a list overflowed and
// the mask is incomplete (re-run the direct kernel).
int tpc_launch_query_partitioned(const TpcLaunch &a, const TpcQPlan &pl0, uint32_t *rmask, uint64_t lo, uint64_t hi, bool gated)
{
    TpcQPlan pl = pl0;
    pl.rbuf1 = pl.buf1;
    pl.rcnt1 = pl.cnt1;
    int rc = tpc_launch_query_part_hash(a, pl, rmask, lo, hi, gated);
    if (rc) return rc;
    if ((rc = tpc_launch_query_part_lookup(a, pl))) return rc;
    return tpc_launch_query_verify(a, pl, rmask);
}
#else
int tpc_launch_query_verify_other(const TpcLaunch &a, const TpcQPlan &pl, uint32_t *rmask)
{
    switch (a.P.q) {
    case 1: launch_qverify<1>(a, pl, rmask); break;
    case 2: launch_qverify<2>(a, pl, rmask); break;
    case 3: launch_qverify<3>(a, pl, rmask); break;
    case 4: launch_qverify<4>(a, pl, rmask); break;
    case 6: launch_qverify<6>(a, pl, rmask); break;
    case 7: launch_qverify<7>(a, pl, rmask); break;
    case 8: launch_qverify<8>(a, pl, rmask); break;
    case 9: launch_qverify<9>(a, pl, rmask); break;
    case 10: launch_qverify<10>(a, pl, rmask); break;
    case 11: launch_qverify<11>(a, pl, rmask); break;
    case 12: launch_qverify<12>(a, pl, rmask); break;
    case 13: launch_qverify<13>(a, pl, rmask); break;
    case 14: launch_qverify<14>(a, pl, rmask); break;
    case 15: launch_qverify<15>(a, pl, rmask); break;
    case 16: launch_qverify<16>(a, pl, rmask); break;
    default: return -1;
    }
    return 0;
}
#endif
#if TPC_QPARTITION_PART == 1
int tpc_launch_verify_addrs(const TpcLaunch &a, const TpcQPlan &pl, int fn, int fn_count, const uint64_t *sid, uint64_t n, uint64_t *addr_out,
                            int32_t *owner_out, unsigned long long *owner_counts)
{   // owner_out == nullptr: tagged addresses (owner << V_OWNER_SHIFT) and, when owner_counts is given, the probes per owner
    if (fn < 0 || fn_count < 1 || fn + fn_count > a.P.q) return -1;
    if (n == 0) return 0;
    const PtPerm perm{pl.slice_bits, pl.b1 + pl.b2 + pl.b3, pl.perm_mult, pl.perm_inv};
    const PtShard sh{pl.rank, pl.world};
    const uint64_t gbase = pl.tile0_global * (uint64_t)(PT_THREADS * TPC_RUN);
    const dim3 grid((unsigned)std::min<uint64_t>((n + 255) / 256, 4096));
    const size_t table = (size_t)(a.P.k + 1) * 4 * a.P.q * 16;  // k_v_addrs2's letter table
    const bool lean = a.P.k <= 31 && table <= 48 * 1024 && !TpcEnv::get().no_lean;
#define CALL(Q_)                                                                                                                                         \
    do {                                                                                                                                                 \
        if (lean) {                                                                                                                                      \
            (void)hipFuncSetAttribute((const void *)k_v_addrs2<Q_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)table);                             \
            hipLaunchKernelGGL((k_v_addrs2<Q_>), grid, dim3(256), table, a.stream, a.P, a.tab, a.bases, sid, n, gbase, perm, sh, pl.b2 + pl.b3, fn, fn_count, addr_out, owner_out, owner_counts); \
        } else                                                                                                                                           \
            hipLaunchKernelGGL((k_v_addrs<Q_>), grid, dim3(256), 0, a.stream, a.P, a.tab, a.bases, sid, n, gbase, perm, sh, pl.b2 + pl.b3, fn, fn_count, addr_out, owner_out, owner_counts); \
    } while (0)
    switch (a.P.q) {
    case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; case 5: CALL(5); break;
    case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break;
    case 9: CALL(9); break; case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break;
    case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break; case 16: CALL(16); break;
    default: return -1;
    }
#undef CALL
    return 0;
}

int tpc_launch_shard_probe(const TpcLaunch &a, const uint64_t *addr, uint64_t n, uint8_t *hit)
{
    if (n) hipLaunchKernelGGL(k_v_probe, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 8192)), dim3(256), 0, a.stream, a.filter, addr, n, hit);
    return 0;
}

int tpc_launch_shard_mark(const TpcLaunch &a, const TpcQPlan &pl, const uint64_t *sid, uint64_t n, uint32_t *rmask)
{
    uint32_t lw = 0;
    while ((1u << lw) < pl.world) ++lw;
    if (n) hipLaunchKernelGGL(k_v_mark, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, a.stream, sid, n,
                              pl.tile0_global * (uint64_t)(PT_THREADS * TPC_RUN), (1ull << (30u - lw)) - 1ull, rmask);
    return 0;
}

__global__ void k_survivor_sources(const uint64_t *__restrict__ sid, uint64_t n, uint32_t shift, int32_t *__restrict__ src)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) src[i] = (int32_t)((sid[i] >> 3) >> shift);
}

int tpc_launch_survivor_sources(hipStream_t s, const uint64_t *sid, uint64_t n, uint32_t world, int32_t *src)
{   // the rank that hashed the survivor's position: the top log2(world) bits of its 30-bit position field (k_q_hash<SHARDED>)
    uint32_t lw = 0;
    while ((1u << lw) < world) ++lw;
    if (n) hipLaunchKernelGGL(k_survivor_sources, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, sid, n, 30u - lw, src);
    return 0;
}

int tpc_launch_route(hipStream_t s, const int32_t *owner, uint64_t n, unsigned long long *counts, unsigned long long *cursor, uint32_t *perm, int phase)
{   // phase 0: counts[64] += items per owner; phase 1: perm from the cursors (exclusive prefix of the counts, set by the caller)
    if (n == 0) return 0;
    if (phase == 0) hipLaunchKernelGGL(k_route_count, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 2048)), dim3(256), 0, s, owner, n, counts);
    else hipLaunchKernelGGL(k_route_scatter, dim3((unsigned)std::min<uint64_t>((n + RT_CHUNK - 1) / RT_CHUNK, 4096)), dim3(256), 0, s, owner, n, cursor, perm);
    return 0;
}

int tpc_launch_permute64(hipStream_t s, const uint64_t *src, const uint32_t *perm, uint64_t n, uint64_t *dst)
{
    if (n) hipLaunchKernelGGL(k_permute64, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 8192)), dim3(256), 0, s, src, perm, n, dst);
    return 0;
}

int tpc_launch_permute_rows(hipStream_t s, const uint64_t *src, const uint32_t *perm, uint64_t n, int row_words, uint64_t *dst)
{
    if (n) hipLaunchKernelGGL(k_permute_rows, dim3((unsigned)std::min<uint64_t>((n * row_words + 255) / 256, 8192)), dim3(256), 0, s, src, perm, n, row_words, dst);
    return 0;
}

int tpc_launch_select(hipStream_t s, const uint64_t *sid, uint64_t n, int fn_count, const uint8_t *hit, const uint32_t *perm, uint64_t *out, unsigned long long *n_out)
{
    if (n) hipLaunchKernelGGL(k_select, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 8192)), dim3(256), 0, s, sid, n, fn_count, hit, perm, out, n_out);
    return 0;
}

int tpc_launch_finish(hipStream_t s, const TpcQPlan &pl, const uint64_t *sid, uint64_t n, int fn_count, const uint8_t *hit, const uint32_t *perm, uint32_t *rmask,
                      unsigned long long *n_marked)
{
    uint32_t lw = 0;
    while ((1u << lw) < pl.world) ++lw;
    if (n) hipLaunchKernelGGL(k_v_finish, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 8192)), dim3(256), 0, s, sid, n, fn_count, hit, perm,
                              pl.tile0_global * (uint64_t)(PT_THREADS * TPC_RUN), (1ull << (30u - lw)) - 1ull, rmask, n_marked);
    return 0;
}

int tpc_launch_route64(hipStream_t s, const uint64_t *v, uint64_t n, int shift, uint32_t omask, uint64_t keep, unsigned long long *counts, unsigned long long *cursor,
                       uint32_t *perm, uint64_t *dst, int phase)
{   // phase 0: counts[o] += items of owner o; phase 1: scatter by the cursors
    if (n == 0) return 0;
    if (phase == 0) hipLaunchKernelGGL(k_route_count64, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 2048)), dim3(256), 0, s, v, n, shift, omask, counts);
    else hipLaunchKernelGGL(k_route_scatter64, dim3((unsigned)std::min<uint64_t>((n + RT_CHUNK - 1) / RT_CHUNK, 4096)), dim3(256), 0, s, v, n, shift, omask, keep, cursor, perm, dst);
    return 0;
}

int tpc_launch_surv_gather(const TpcLaunch &a, const TpcQPlan &pl, uint64_t *out)
{
    hipLaunchKernelGGL(k_surv_gather, dim3(64, QS_LISTS), dim3(256), 0, a.stream, pl.surv, pl.surv_cur, pl.surv_cap, out);
    return 0;
}

#endif  // part 1 only
#if TPC_QPARTITION_PART == 0
// tpc_preload: the first use of any kernel of this translation unit makes the runtime load its code object
__global__ void k_warm_qpartition() {}
int tpc_warm_qpartition() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_warm_qpartition)) == hipSuccess ? 0 : -1; }

int tpc_launch_periodic_build(const TpcLaunch &a, uint32_t *qs, uint32_t *qd, uint64_t stride, uint32_t *ins, uint64_t w_begin, uint64_t w_end, uint64_t pos_lo, uint64_t pos_hi,
                              uint32_t *any)
{   // qs == nullptr: detection only (any[0], any[1])
    if (w_end > w_begin)
        hipLaunchKernelGGL(k_periodic_build, dim3((unsigned)((w_end - w_begin + 255) / 256)), dim3(256), 0, a.stream, a.bases, a.nmask, a.n_text, a.P.k, qs, qd, stride, ins, w_begin,
                           w_end, pos_lo, pos_hi, any);
    return 0;
}

int tpc_launch_periodic_copy(hipStream_t stream, uint32_t *rmask, const uint32_t *qs, const uint32_t *qd, uint64_t stride, uint64_t n_words)
{
    if (n_words) hipLaunchKernelGGL(k_periodic_copy, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, stream, rmask, qs, qd, stride, n_words);
    return 0;
}

#endif  // part 0 only