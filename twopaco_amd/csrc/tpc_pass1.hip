// tpc_pass1.hip -- first pass on gfx950: rolling canonical k-mer hash over the 2-bit packed
// text, Bloom insert (atomicOr on HBM words), Bloom query (candidate junction mask) and the
// multi-round split histogram.
//
// Work decomposition (all three kernels): one workgroup = 256 threads = 8192 consecutive
// vertex positions; thread t owns the 32 positions of packed word (tile*256 + t), seeds its
// 2q hashes from the k bases of its first window and then rolls.  The tile's packed words
// (+ halo) are staged once through LDS with coalesced loads; the q x 5 character tables live
// in LDS too.  Integer/bit work only -- no MFMA.
//
// Reference (paths relative to /root/reference/src, VE.h = graphconstructor/vertexenumerator.h):
//   FilterFillerWorker         VE.h:995-1105    -> k_insert
//   CandidateCheckingWorker    VE.h:586-704     -> k_query
//   InitialFilterFillerWorker  VE.h:503-583     -> k_split
//   VertexRollingHash          graphconstructor/vertexrollinghash.h:54-252
#include "tpc_device.h"
#include "tpc_insert_step.h"
#include "tpc_internal.h"

namespace {

__device__ __forceinline__ bool within(uint64_t v, uint64_t lo, uint64_t hi) { return v >= lo && v <= hi; }  // VE.h:473-476

__device__ __forceinline__ void set_bit(uint32_t *filter, uint64_t a)
{   // ConcurrentBitVector::SetBitConcurrently, concurrentbitvector.cpp:31-37 (fetch_or, result unused)
    atomicOr(&filter[a >> 5], 1u << ((uint32_t)a & 31u));
}

template <bool TEST>
__device__ __forceinline__ void insert_bit(uint32_t *filter, uint64_t a)
{
    if (TEST) {  // "if(!GetBit) SetBitConcurrently", VE.h:1086-1092
        const uint32_t w = filter[a >> 5];
        if ((w >> ((uint32_t)a & 31u)) & 1u) return;
    }
    set_bit(filter, a);
}

__device__ __forceinline__ bool get_bit(const uint32_t *filter, uint64_t a)
{   // ConcurrentBitVector::GetBit, concurrentbitvector.cpp:39-45
    return (filter[a >> 5] >> ((uint32_t)a & 31u)) & 1u;
}

template <bool TEST>
struct DirectEmit {
    uint32_t *filter;
    template <int Q>
    __device__ __forceinline__ void edge(const uint64_t (&a)[Q])
    {
#pragma unroll
        for (int i = 0; i < Q; i++) insert_bit<TEST>(filter, a[i]);
    }
};

// one atomic per workgroup (same-address device atomics serialise at ~12 ns each)
__device__ __forceinline__ void block_add(unsigned long long *dst, unsigned v, uint32_t *s_w)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (t) atomicAdd(dst, (unsigned long long)t);
    }
}

// ------------------------------------------------------------------------------------------
// First-pass insert.
template <int Q, bool GATED, bool TEST>
__global__ void __launch_bounds__(TPC_TILE_THREADS)
k_insert(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases,
         const uint32_t *__restrict__ nmask, uint64_t n_text, uint32_t *__restrict__ filter, uint64_t lo, uint64_t hi,
         unsigned long long *n_kmers)
{
    __shared__ uint64_t s_h[Q * 5], s_hk[Q * 5];
    __shared__ uint64_t s_b[TPC_TILE_WORDS];
    __shared__ uint32_t s_n[TPC_TILE_WORDS];
    const int tid = threadIdx.x;
    const uint64_t wfirst = (uint64_t)blockIdx.x * TPC_TILE_THREADS;
    const uint64_t wbase = wfirst - 1;
    if (tid < Q * 5) { s_h[tid] = tab[tid]; s_hk[tid] = tab[TPC_TAB_HK + tid]; }
    tpc_stage_tile(s_b, s_n, bases, nmask, wfirst, (P.k + 1) / 32 + 2);
    __syncthreads();

    const uint64_t g0 = (wfirst + tid) * TPC_RUN;
    unsigned hashed = 0;
    if (g0 < n_text) {
        TpcRoll<Q> r;
        tpc_roll_init<Q>(r, P, s_h, s_b, s_n, g0, wbase);
        DirectEmit<TEST> emit{filter};
        for (int s = 0; s < TPC_RUN; s++)
            hashed += tpc_insert_step<Q, GATED>(r, P, s_h, s_hk, s_b, s_n, g0 + s, wbase, lo, hi, emit);
    }
    __shared__ uint32_t s_w[4];
    if (n_kmers) block_add(n_kmers, hashed, s_w);
}

// ------------------------------------------------------------------------------------------
// First-pass query.  mark(g) <=> (#in-edges present >= 2) || (#out-edges present >= 2) with an N
// neighbour counting 2 -- the early exits of VE.h:640-660 do not change that predicate.  The first
// Bloom probe of every unknown edge is issued before any is consumed (8 independent loads in
// flight per lane); the remaining q-1 probes run only for edges whose first bit is set.
template <int Q, bool OUT>
struct EdgeEval {
    // p_i / n_i of edge (c + v) or (v + c) for hash function i
    __device__ __forceinline__ static uint64_t P_(const TpcVHash<Q> &v, const uint64_t *r1, const uint64_t *s_h, const uint64_t *s_hk, int i, int c)
    {
        return OUT ? (r1[i] ^ s_h[i * 5 + c]) : (s_hk[i * 5 + c] ^ v.pos[i]);
    }
    __device__ __forceinline__ static uint64_t N_(const TpcVHash<Q> &v, const uint64_t *r1, const uint64_t *s_h, const uint64_t *s_hk, int i, int c)
    {
        return OUT ? (v.neg[i] ^ s_hk[i * 5 + 3 - c]) : (r1[i] ^ s_h[i * 5 + 3 - c]);
    }
};

template <int Q, bool GATED>
__global__ void __launch_bounds__(TPC_TILE_THREADS)
k_query(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases,
        const uint32_t *__restrict__ nmask, uint64_t n_text, const uint32_t *__restrict__ filter,
        uint32_t *__restrict__ rmask, uint64_t lo, uint64_t hi, unsigned long long *n_marks)
{
    __shared__ uint64_t s_h[Q * 5], s_hk[Q * 5];
    __shared__ uint64_t s_b[TPC_TILE_WORDS];
    __shared__ uint32_t s_n[TPC_TILE_WORDS];
    const int tid = threadIdx.x;
    const uint64_t wfirst = (uint64_t)blockIdx.x * TPC_TILE_THREADS;
    const uint64_t wbase = wfirst - 1;
    if (tid < Q * 5) { s_h[tid] = tab[tid]; s_hk[tid] = tab[TPC_TAB_HK + tid]; }
    tpc_stage_tile(s_b, s_n, bases, nmask, wfirst, (P.k + 1) / 32 + 2);
    __syncthreads();

    const uint64_t g0 = (wfirst + tid) * TPC_RUN;
    uint32_t word = 0;
    if (g0 < n_text) {
        TpcVHash<Q> v;
        tpc_vhash_init<Q>(v, P, s_h, s_b, s_n, g0, wbase);
        int ncnt = 0;
        for (int t = 0; t < P.k; t++) ncnt += tpc_tile_char(s_b, s_n, g0 + t, wbase) == TPC_CODE_N;
        int c_prev = tpc_tile_char(s_b, s_n, g0 - 1, wbase);
        int c_first = tpc_tile_char(s_b, s_n, g0, wbase);
        for (int s = 0; s < TPC_RUN; s++) {
            const uint64_t g = g0 + s;
            const int c_next = tpc_tile_char(s_b, s_n, g + P.k, wbase);
            const int c_first_nx = tpc_tile_char(s_b, s_n, g + 1, wbase);
            uint64_t r1p[Q], r1n[Q];
#pragma unroll
            for (int i = 0; i < Q; i++) {
                r1p[i] = tpc_rotl1(v.pos[i], P.L, P.lmask);
                r1n[i] = tpc_rotl1(v.neg[i], P.L, P.lmask);
            }
            bool check = ncnt == 0;
            if (GATED) check = check && within(tpc_min(v.pos[0], v.neg[0]), lo, hi);  // VE.h:638
            if (check) {
                bool mark = (c_prev == TPC_CODE_N) || (c_next == TPC_CODE_N);  // VE.h:640-641
                if (!mark) {
                    // phase 1: strand + first address + first probe of all 8 edges
                    uint64_t a0[8];
                    bool ng[8];
                    uint32_t w0[8];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        {   // in-edge c + v
                            bool neg = false, decided = false;
#pragma unroll
                            for (int i = 0; i < Q; i++) {
                                const uint64_t p = EdgeEval<Q, false>::P_(v, r1n, s_h, s_hk, i, c);
                                const uint64_t n = EdgeEval<Q, false>::N_(v, r1n, s_h, s_hk, i, c);
                                if (!decided && p != n) { neg = n < p; decided = true; }
                            }
                            ng[c] = neg;
                            a0[c] = neg ? EdgeEval<Q, false>::N_(v, r1n, s_h, s_hk, 0, c) : EdgeEval<Q, false>::P_(v, r1n, s_h, s_hk, 0, c);
                        }
                        {   // out-edge v + c
                            bool neg = false, decided = false;
#pragma unroll
                            for (int i = 0; i < Q; i++) {
                                const uint64_t p = EdgeEval<Q, true>::P_(v, r1p, s_h, s_hk, i, c);
                                const uint64_t n = EdgeEval<Q, true>::N_(v, r1p, s_h, s_hk, i, c);
                                if (!decided && p != n) { neg = n < p; decided = true; }
                            }
                            ng[4 + c] = neg;
                            a0[4 + c] = neg ? EdgeEval<Q, true>::N_(v, r1p, s_h, s_hk, 0, c) : EdgeEval<Q, true>::P_(v, r1p, s_h, s_hk, 0, c);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const bool known = (e < 4) ? (e == c_prev) : (e - 4 == c_next);
                        w0[e] = known ? 0xFFFFFFFFu : filter[a0[e] >> 5];
                    }
                    // phase 2: finish the probe chain of edges that survived the first bit
                    int in = 0, out = 0;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const int c = e & 3;
                        const bool known = (e < 4) ? (c == c_prev) : (c == c_next);
                        bool present = (w0[e] >> ((uint32_t)a0[e] & 31u)) & 1u;
                        if (present && !known) {
                            for (int i = 1; i < Q && present; i++) {
                                uint64_t a;
                                if (e < 4) a = ng[e] ? EdgeEval<Q, false>::N_(v, r1n, s_h, s_hk, i, c) : EdgeEval<Q, false>::P_(v, r1n, s_h, s_hk, i, c);
                                else a = ng[e] ? EdgeEval<Q, true>::N_(v, r1p, s_h, s_hk, i, c) : EdgeEval<Q, true>::P_(v, r1p, s_h, s_hk, i, c);
                                present = get_bit(filter, a);
                            }
                        }
                        if (e < 4) in += present; else out += present;
                    }
                    mark = in > 1 || out > 1;  // VE.h:656
                }
                if (mark) word |= 1u << s;
            }
            // roll: VertexRollingHash::Update (vertexrollinghash.h:104-113)
#pragma unroll
            for (int i = 0; i < Q; i++) {
                v.pos[i] = r1p[i] ^ s_h[i * 5 + c_next] ^ s_hk[i * 5 + c_first];
                v.neg[i] = tpc_rotr1(v.neg[i] ^ s_hk[i * 5 + tpc_rc(c_next)] ^ s_h[i * 5 + tpc_rc(c_first)], P.L);
            }
            ncnt += (c_next == TPC_CODE_N) - (c_first == TPC_CODE_N);
            c_prev = c_first;
            c_first = c_first_nx;
        }
    }
    rmask[wfirst + tid] = word;
    __shared__ uint32_t s_w[4];
    block_add(n_marks, (unsigned)__popc(word), s_w);
}

// ------------------------------------------------------------------------------------------
// Split pass histogram (rounds > 1).  Every (k+1)-mer of 'N'+record+'N' (no N gate) is inserted
// into the scratch filter; the FIRST-SEEN occurrence of an edge bumps the bins of its two endpoint
// vertex hashes (VE.h:538-571: "wasSet" is false when some bit of the edge was still unset).
// The reference evaluates "first seen" in the arrival order of its worker threads (text order at -t 1).
// Here an edge is counted once, by ballot over its own bits, whatever the arrival order:
//   phase i (one launch each, i = 0 .. phases-1): every occurrence that has not won yet does one
//   atomicOr on bit a_i of its edge; the occurrence that flips it wins: it is counted, sets the edge's
//   other q-1 bits and leaves the game.  All occurrences of one edge share their q addresses, so exactly
//   one of them can flip a_i; an edge whose a_0 was taken by a DIFFERENT edge gets its chance on a_1 in
//   the next launch (by then every winner's bits are visible), and so on.
// An edge therefore counts once unless its first `phases` bits are all covered by other edges -- the
// sequential rule is "unless all q bits are covered by earlier edges".  phases = min(q, 3) while the filter
// is sparse (the two rules then differ by (fill/q)^3 of the edges), q when it is crowded.  With a scratch
// filter large enough that no such coverage happens both give the number of distinct edges, bin for bin
// (tests/test_gpu_parity.py::test_split_histogram_exact_when_collision_free).
// One order dependence survives even then, in the reference as here: for k + 1 > L the cyclic polynomial
// hash rotates characters L positions apart by the same amount (cyclichash.h:29-35), so two DIFFERENT
// (k+1)-mers with the same letters in every rotation class (typically windows over the edge of an N run)
// share all q addresses; only one of them is ever "first seen", and which one (text order at -t 1, arrival
// order otherwise) decides whose endpoint bins are bumped.  The total and all other bins are unaffected.
template <int Q>
__global__ void __launch_bounds__(TPC_TILE_THREADS)
k_split(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases,
        const uint32_t *__restrict__ nmask, const uint32_t *todo_in, uint32_t *todo_out, int phase, uint64_t n_text,
        uint32_t *__restrict__ filter, uint32_t *__restrict__ bins, uint64_t bin_size)
{
    __shared__ uint64_t s_h[Q * 5], s_hk[Q * 5];
    __shared__ uint64_t s_b[TPC_TILE_WORDS];
    __shared__ uint32_t s_n[TPC_TILE_WORDS];
    const int tid = threadIdx.x;
    const uint64_t wfirst = (uint64_t)blockIdx.x * TPC_TILE_THREADS;
    const uint64_t wbase = wfirst - 1;
    if (tid < Q * 5) { s_h[tid] = tab[tid]; s_hk[tid] = tab[TPC_TAB_HK + tid]; }
    tpc_stage_tile(s_b, s_n, bases, nmask, wfirst, (P.k + 1) / 32 + 2);
    __syncthreads();
    const uint64_t g0 = (wfirst + tid) * TPC_RUN;
    if (g0 >= n_text) return;
    uint32_t em = todo_in[wfirst + tid];  // bit s: an edge occurrence that has not won yet starts at g0+s
    if (em == 0) { if (todo_out != todo_in) todo_out[wfirst + tid] = 0; return; }
    TpcVHash<Q> v;
    tpc_vhash_init<Q>(v, P, s_h, s_b, s_n, g0, wbase);
    int c_first = tpc_tile_char(s_b, s_n, g0, wbase);
    for (int s = 0; s < TPC_RUN; s++) {
        const uint64_t g = g0 + s;
        const int c_next = tpc_tile_char(s_b, s_n, g + P.k, wbase);
        const int c_first_nx = tpc_tile_char(s_b, s_n, g + 1, wbase);
        uint64_t ep[Q], en[Q], npos[Q], nneg[Q];
#pragma unroll
        for (int i = 0; i < Q; i++) {
            ep[i] = tpc_rotl1(v.pos[i], P.L, P.lmask) ^ s_h[i * 5 + c_next];
            en[i] = v.neg[i] ^ s_hk[i * 5 + tpc_rc(c_next)];
            npos[i] = ep[i] ^ s_hk[i * 5 + c_first];
            nneg[i] = tpc_rotr1(en[i] ^ s_h[i * 5 + tpc_rc(c_first)], P.L);
        }
        if ((em >> s) & 1u) {
            const bool neg = tpc_pick_neg<Q>(ep, en);
            uint64_t a[Q];
#pragma unroll
            for (int i = 0; i < Q; i++) a[i] = neg ? en[i] : ep[i];
            uint64_t mine = a[0];
#pragma unroll
            for (int i = 1; i < Q; i++) if (i == phase) mine = a[i];
            const uint32_t bit = 1u << ((uint32_t)mine & 31u);
            const uint32_t old = atomicOr(&filter[mine >> 5], bit);
            if (!(old & bit)) {  // this occurrence is the edge's first-seen one
#pragma unroll
                for (int i = 0; i < Q; i++)
                    if (i != phase) atomicOr(&filter[a[i] >> 5], 1u << ((uint32_t)a[i] & 31u));
                const uint64_t b0 = tpc_min(v.pos[0], v.neg[0]) / bin_size;
                const uint64_t b1 = tpc_min(npos[0], nneg[0]) / bin_size;
                // MAX_COUNTER saturation (common.cpp:6): counts beyond 2^31-1 are not reachable per bin here
                if (bins[b0] < 0x7FFFFFFFu) atomicAdd(&bins[b0], 1u);
                if (bins[b1] < 0x7FFFFFFFu) atomicAdd(&bins[b1], 1u);
                em &= ~(1u << s);
            }
        }
#pragma unroll
        for (int i = 0; i < Q; i++) { v.pos[i] = npos[i]; v.neg[i] = nneg[i]; }
        c_first = c_first_nx;
    }
    todo_out[wfirst + tid] = em;
}

// Vertex hashes of a range of windows (parity tap).
__global__ void k_hash_dump(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases,
                            const uint32_t *__restrict__ nmask, uint64_t g0, uint64_t n, uint64_t *out)
{
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const uint64_t g = g0 + idx;
    for (int i = 0; i < P.q; i++) {
        uint64_t p = 0, ng = 0;
        for (int t = 0; t < P.k; t++) {
            p = tpc_rotl1(p, P.L, P.lmask) ^ tab[i * 5 + tpc_text_char(bases, nmask, g + t)];
            ng = tpc_rotl1(ng, P.L, P.lmask) ^ tab[i * 5 + tpc_rc(tpc_text_char(bases, nmask, g + P.k - 1 - t))];
        }
        out[idx * 2 * P.q + 2 * i] = p;
        out[idx * 2 * P.q + 2 * i + 1] = ng;
    }
}

template <int Q>
void launch_insert_q(const TpcLaunch &a, uint64_t lo, uint64_t hi, bool gated, bool test, unsigned long long *n_kmers)
{
    dim3 grid((unsigned)a.n_tiles), block(TPC_TILE_THREADS);
#define TPC_GO(G, T) hipLaunchKernelGGL((k_insert<Q, G, T>), grid, block, 0, a.stream, a.P, a.tab, a.bases, a.nmask, a.n_text, a.filter, lo, hi, n_kmers)
    if (gated) { if (test) TPC_GO(true, true); else TPC_GO(true, false); }
    else { if (test) TPC_GO(false, true); else TPC_GO(false, false); }
#undef TPC_GO
}

template <int Q>
void launch_query_q(const TpcLaunch &a, uint32_t *rmask, uint64_t lo, uint64_t hi, bool gated, unsigned long long *n_marks)
{
    dim3 grid((unsigned)a.n_tiles), block(TPC_TILE_THREADS);
    if (gated) hipLaunchKernelGGL((k_query<Q, true>), grid, block, 0, a.stream, a.P, a.tab, a.bases, a.nmask, a.n_text, a.filter, rmask, lo, hi, n_marks);
    else hipLaunchKernelGGL((k_query<Q, false>), grid, block, 0, a.stream, a.P, a.tab, a.bases, a.nmask, a.n_text, a.filter, rmask, lo, hi, n_marks);
}

template <int Q>
void launch_split_q(const TpcLaunch &a, uint32_t *todo, uint32_t *bins, uint64_t bin_size)
{
    dim3 grid((unsigned)a.n_tiles), block(TPC_TILE_THREADS);
    const bool crowded = (double)Q * (double)a.n_text > (double)(a.P.lmask >> 3);  // more than 1/8 of the bits could be set
    const int phases = (Q < 3 || crowded) ? Q : 3;  // see k_split
    for (int phase = 0; phase < phases; phase++)
        hipLaunchKernelGGL((k_split<Q>), grid, block, 0, a.stream, a.P, a.tab, a.bases, a.nmask, todo, todo, phase, a.n_text, a.filter, bins, bin_size);
}

}  // namespace

#define TPC_DISPATCH_Q(q, CALL)                \
    switch (q) {                               \
    case 1: CALL(1); break;                    \
    case 2: CALL(2); break;                    \
    case 3: CALL(3); break;                    \
    case 4: CALL(4); break;                    \
    case 5: CALL(5); break;                    \
    case 6: CALL(6); break;                    \
    case 7: CALL(7); break;                    \
    case 8: CALL(8); break;                    \
    case 9: CALL(9); break;                    \
    case 10: CALL(10); break;                  \
    case 11: CALL(11); break;                  \
    case 12: CALL(12); break;                  \
    case 13: CALL(13); break;                  \
    case 14: CALL(14); break;                  \
    case 15: CALL(15); break;                  \
    case 16: CALL(16); break;                  \
    default: return -1;                        \
    }

int tpc_test_force_anyq = 0;  // option "test_force_anyq" (tests only, process-wide): the closed-form kernels of tpc_pass1_anyq.hip for every q

int tpc_launch_insert(const TpcLaunch &a, uint64_t lo, uint64_t hi, bool gated, bool test, unsigned long long *n_kmers)
{
    if (a.P.q > TPC_KERNEL_MAXQ || tpc_test_force_anyq) return tpc_launch_insert_anyq(a, lo, hi, gated, test, n_kmers);
#define CALL(Q) launch_insert_q<Q>(a, lo, hi, gated, test, n_kmers)
    TPC_DISPATCH_Q(a.P.q, CALL)
#undef CALL
    return 0;
}

int tpc_launch_query(const TpcLaunch &a, uint32_t *rmask, uint64_t lo, uint64_t hi, bool gated, unsigned long long *n_marks)
{
    if (a.P.q > TPC_KERNEL_MAXQ || tpc_test_force_anyq) return tpc_launch_query_anyq(a, rmask, lo, hi, gated, n_marks);
#define CALL(Q) launch_query_q<Q>(a, rmask, lo, hi, gated, n_marks)
    TPC_DISPATCH_Q(a.P.q, CALL)
#undef CALL
    return 0;
}

// The positions where a (k+1)-mer of 'N' + record + 'N' starts, dispatched records only (VE.h:1177: at least k bases): bit g for
// g in [start - 1, start + len - k] of every record.  One thread per mask word; the records are in text order, so the record a
// word's first position belongs to (or follows) is a binary search away.  (The host used to build this 1-bit-per-position mask in a
// vector and copy it over: 1.1 GB and ~0.8 s at 9 G positions.)
__global__ void __launch_bounds__(256) k_split_emask(const uint64_t *__restrict__ rec_start, const uint64_t *__restrict__ rec_len, uint32_t n_rec, int k, uint64_t n_words,
                                                     uint32_t *__restrict__ emask)
{
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= n_words) return;
    const uint64_t g0 = w << 5;
    // the ranges [a_r, b_r] = [start_r - 1, start_r + len_r - k] of the dispatched records are disjoint and ascending: the first one that
    // ends at or behind this word's first position, then its successors while they begin inside the word
    uint32_t lo = 0, hi = n_rec;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (rec_start[mid] + rec_len[mid] - (uint64_t)k < g0) lo = mid + 1; else hi = mid;
    }
    uint32_t m = 0;
    for (uint32_t r = lo; r < n_rec; r++) {
        const uint64_t a = rec_start[r] - 1, b = rec_start[r] + rec_len[r] - (uint64_t)k;
        if (a > g0 + 31) break;
        const uint64_t x0 = a > g0 ? a : g0, x1 = b < g0 + 31 ? b : g0 + 31;
        const uint32_t len = (uint32_t)(x1 - x0 + 1);
        m |= (len == 32 ? ~0u : ((1u << len) - 1u)) << (uint32_t)(x0 - g0);
    }
    emask[w] = m;
}

int tpc_launch_split_emask(hipStream_t s, const uint64_t *d_rec_start, const uint64_t *d_rec_len, uint32_t n_rec, int k, uint64_t n_words, uint32_t *emask)
{
    if (n_words) hipLaunchKernelGGL(k_split_emask, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s, d_rec_start, d_rec_len, n_rec, k, n_words, emask);
    return 0;
}

int tpc_launch_split(const TpcLaunch &a, uint32_t *emask, uint32_t *bins, uint64_t bin_size)
{
    if (a.P.q > TPC_KERNEL_MAXQ || tpc_test_force_anyq) return tpc_launch_split_anyq(a, emask, bins, bin_size);
#define CALL(Q) launch_split_q<Q>(a, emask, bins, bin_size)
    TPC_DISPATCH_Q(a.P.q, CALL)
#undef CALL
    return 0;
}

int tpc_launch_hash_dump(const TpcLaunch &a, uint64_t g0, uint64_t n, uint64_t *out)
{
    const unsigned blocks = (unsigned)((n + 127) / 128);
    hipLaunchKernelGGL(k_hash_dump, dim3(blocks), dim3(128), 0, a.stream, a.P, a.tab, a.bases, a.nmask, g0, n, out);
    return 0;
}

// tpc_preload: the first use of any kernel of this translation unit makes the runtime load its code object
__global__ void k_warm_pass1() {}
int tpc_warm_pass1() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_warm_pass1)) == hipSuccess ? 0 : -1; }
