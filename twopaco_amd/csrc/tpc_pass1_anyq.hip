// tpc_pass1_anyq.hip -- first pass for MORE than 16 hash functions (-q 17 .. 64).
//
// The reference takes any -q (constructor.cpp:83-90); the rolling kernels of this library are instantiated for 1..16
// functions (2q running hashes per lane in registers).  Beyond that nobody is in a hurry -- a Bloom filter with more than 16
// probes per edge is a curiosity -- so these kernels restate the definitions directly, one thread per packed word of the
// text and every hash evaluated in closed form from the letters:
//     H_i(E)    = fold_t  rotl_L(acc, 1) ^ h_i[E_t]                    (CyclicHash::eat, cyclichash.h:106-109)
//     H_i(rc E) = the same fold over the reverse complement of E      (vertexrollinghash.h:79-102: the negative strand)
//     canonical strand of an edge = the first function whose two values differ, smaller wins, all equal -> positive
//                                                                      (DetermineStrandExtend / Prepend, vertexrollinghash.h:170-200)
// with the same per-position rules as the fast kernels: FilterFillerWorker (VE.h:1035-1083) -> k_insert_anyq,
// CandidateCheckingWorker (VE.h:633-674) -> k_query_anyq, InitialFilterFillerWorker (VE.h:538-571) -> k_split_anyq
// (the exact ballot of tpc_pass1.hip:k_split).  O(k q) per position; filter contents, masks and histograms are those of the
// fast kernels by construction (tests: golden rand6_k9_q20 from the real reference, and q <= 16 forced through this file).
#include "tpc_device.h"
#include "tpc_internal.h"
#include <algorithm>

namespace {

__device__ __forceinline__ bool within(uint64_t v, uint64_t lo, uint64_t hi) { return v >= lo && v <= hi; }  // VE.h:473-476

// A (k+1)-mer of the text model: kind 0 = T[g .. g+k]; kind 1 = v + c (out-edge of the window at g); kind 2 = c + v (in-edge).
struct Edge {
    const uint64_t *bases;
    const uint32_t *nmask;
    uint64_t g;
    int k, kind, c;
    __device__ __forceinline__ int at(int t) const
    {
        if (kind == 1) return t < k ? tpc_text_char(bases, nmask, g + t) : c;
        if (kind == 2) return t == 0 ? c : tpc_text_char(bases, nmask, g + t - 1);
        return tpc_text_char(bases, nmask, g + t);
    }
};

struct HashCtx {
    const uint64_t *tab;  // h[q][5]
    int L, q;
    uint64_t lmask;
    // function i over `len` letters of e, positive strand / reverse complement
    __device__ __forceinline__ uint64_t pos(const Edge &e, int len, int i) const
    {
        uint64_t a = 0;
        for (int t = 0; t < len; t++) a = tpc_rotl1(a, L, lmask) ^ tab[i * 5 + e.at(t)];
        return a;
    }
    __device__ __forceinline__ uint64_t neg(const Edge &e, int len, int i) const
    {
        uint64_t a = 0;
        for (int t = len - 1; t >= 0; t--) a = tpc_rotl1(a, L, lmask) ^ tab[i * 5 + tpc_rc(e.at(t))];
        return a;
    }
    __device__ __forceinline__ bool pick_neg(const Edge &e, int len) const
    {
        for (int i = 0; i < q; i++) {
            const uint64_t p = pos(e, len, i), n = neg(e, len, i);
            if (p != n) return n < p;
        }
        return false;
    }
    __device__ __forceinline__ uint64_t addr(const Edge &e, int len, int i, bool ng) const { return ng ? neg(e, len, i) : pos(e, len, i); }
    // GetVertexHash (vertexrollinghash.h:137-142) of the k letters at g
    __device__ __forceinline__ uint64_t vertex_hash(const uint64_t *bases, const uint32_t *nmask, uint64_t g, int k) const
    {
        const Edge v{bases, nmask, g, k, 0, 0};
        return tpc_min(pos(v, k, 0), neg(v, k, 0));
    }
};

__device__ __forceinline__ int window_ncnt(const uint64_t *bases, const uint32_t *nmask, uint64_t g, int k)
{
    int n = 0;
    for (int t = 0; t < k; t++) n += tpc_text_char(bases, nmask, g + t) == TPC_CODE_N;
    return n;
}

template <bool TEST>
__device__ __forceinline__ void insert_edge(const HashCtx &H, const Edge &e, int len, uint32_t *filter)
{
    const bool ng = H.pick_neg(e, len);
    for (int i = 0; i < H.q; i++) {
        const uint64_t a = H.addr(e, len, i, ng);
        if (TEST && ((filter[a >> 5] >> ((uint32_t)a & 31u)) & 1u)) continue;  // "if(!GetBit) SetBitConcurrently", VE.h:1086-1092
        atomicOr(&filter[a >> 5], 1u << ((uint32_t)a & 31u));
    }
}

template <bool TEST>
__global__ void __launch_bounds__(256)
k_insert_anyq(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases, const uint32_t *__restrict__ nmask, uint64_t n_text,
              uint32_t *filter, uint64_t lo, uint64_t hi, int gated, unsigned long long *n_kmers, uint64_t g_begin)
{   // n_text: the end of the positions this launch covers (the text's, or a rank's chunk's); g_begin: their start
    const uint64_t g = g_begin + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const HashCtx H{tab, P.L, P.q, P.lmask};
    const int k = P.k;
    bool vertex = false;
    if (g < n_text) {
        vertex = window_ncnt(bases, nmask, g, k) == 0;
        bool go = vertex;
        if (gated && go) go = within(H.vertex_hash(bases, nmask, g, k), lo, hi) || within(H.vertex_hash(bases, nmask, g + 1, k), lo, hi);  // VE.h:1063-1073
        if (go) {
            const int c_next = tpc_text_char(bases, nmask, g + k), c_prev = g ? tpc_text_char(bases, nmask, g - 1) : TPC_CODE_N;
            if (c_next != TPC_CODE_N) insert_edge<TEST>(H, Edge{bases, nmask, g, k, 0, 0}, k + 1, filter);
            else { insert_edge<TEST>(H, Edge{bases, nmask, g, k, 1, 0}, k + 1, filter); insert_edge<TEST>(H, Edge{bases, nmask, g, k, 1, 3}, k + 1, filter); }  // VE.h:1048-1052
            if (c_prev == TPC_CODE_N) { insert_edge<TEST>(H, Edge{bases, nmask, g, k, 2, 0}, k + 1, filter); insert_edge<TEST>(H, Edge{bases, nmask, g, k, 2, 3}, k + 1, filter); }  // VE.h:1054-1058
        }
    }
    if (n_kmers) {
        const unsigned long long m = __ballot(vertex);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_kmers, (unsigned long long)__popcll(m));
    }
}

// is the edge in the filter?  (IsOutgoingEdgeInBloomFilter / IsIngoingEdgeInBloomFilter, vertexrollinghash.h:208-234: early exit at the first 0 bit)
__device__ __forceinline__ bool edge_present(const HashCtx &H, const Edge &e, int len, const uint32_t *filter)
{
    const bool ng = H.pick_neg(e, len);
    for (int i = 0; i < H.q; i++) {
        const uint64_t a = H.addr(e, len, i, ng);
        if (!((filter[a >> 5] >> ((uint32_t)a & 31u)) & 1u)) return false;
    }
    return true;
}

__global__ void __launch_bounds__(256)
k_query_anyq(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases, const uint32_t *__restrict__ nmask, uint64_t n_text,
             const uint32_t *__restrict__ filter, uint32_t *__restrict__ rmask, uint64_t n_words, uint64_t lo, uint64_t hi, int gated, unsigned long long *n_marks, uint64_t g_begin)
{   // one lane per position, one wave per 64 positions = two mask words; g_begin (a multiple of 64): the first position of this launch
    const uint64_t g = g_begin + (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const HashCtx H{tab, P.L, P.q, P.lmask};
    const int k = P.k;
    bool mark = false;
    if (g < n_text && window_ncnt(bases, nmask, g, k) == 0 && (!gated || within(H.vertex_hash(bases, nmask, g, k), lo, hi))) {  // VE.h:638
        const int c_next = tpc_text_char(bases, nmask, g + k), c_prev = g ? tpc_text_char(bases, nmask, g - 1) : TPC_CODE_N;
        if (c_prev == TPC_CODE_N || c_next == TPC_CODE_N) mark = true;  // VE.h:640-641: an N neighbour counts 2
        else
            for (int c = 0; c < 4 && !mark; c++) {  // the known edges count 1 each: one more present edge on either side marks (VE.h:656)
                if (c != c_prev && edge_present(H, Edge{bases, nmask, g, k, 2, c}, k + 1, filter)) mark = true;
                if (!mark && c != c_next && edge_present(H, Edge{bases, nmask, g, k, 1, c}, k + 1, filter)) mark = true;
            }
    }
    const unsigned long long m = __ballot(mark);
    const uint64_t w = g >> 5;
    if ((threadIdx.x & 31) == 0 && w < n_words) rmask[w] = (uint32_t)(m >> (threadIdx.x & 32));
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_marks, (unsigned long long)__popcll(m));
}

// the exact ballot of tpc_pass1.hip:k_split, one launch per phase
__global__ void __launch_bounds__(256)
k_split_anyq(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases, const uint32_t *__restrict__ nmask, uint32_t *todo, int phase,
             uint64_t n_text, uint32_t *filter, uint32_t *bins, uint64_t bin_size)
{
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_text) return;
    if (!((todo[g >> 5] >> (g & 31)) & 1u)) return;
    const HashCtx H{tab, P.L, P.q, P.lmask};
    const int k = P.k;
    const Edge e{bases, nmask, g, k, 0, 0};
    const bool ng = H.pick_neg(e, k + 1);
    const uint64_t mine = H.addr(e, k + 1, phase, ng);
    const uint32_t bit = 1u << ((uint32_t)mine & 31u);
    const uint32_t old = atomicOr(&filter[mine >> 5], bit);
    if (!(old & bit)) {  // this occurrence is the edge's first-seen one
        for (int i = 0; i < H.q; i++) {
            if (i == phase) continue;
            const uint64_t a = H.addr(e, k + 1, i, ng);
            atomicOr(&filter[a >> 5], 1u << ((uint32_t)a & 31u));
        }
        const uint64_t b0 = H.vertex_hash(bases, nmask, g, k) / bin_size, b1 = H.vertex_hash(bases, nmask, g + 1, k) / bin_size;
        if (bins[b0] < 0x7FFFFFFFu) atomicAdd(&bins[b0], 1u);
        if (bins[b1] < 0x7FFFFFFFu) atomicAdd(&bins[b1], 1u);
        atomicAnd(&todo[g >> 5], ~(1u << (g & 31)));
    }
}

}  // namespace

int tpc_launch_insert_anyq(const TpcLaunch &a, uint64_t lo, uint64_t hi, bool gated, bool test, unsigned long long *n_kmers)
{
    const uint64_t g0 = std::min(a.g_begin, a.n_text), g1 = std::min(a.g_end, a.n_text);  // (a.g_begin / g_end: a rank's chunk; by default the whole text)
    if (g1 <= g0) return 0;
    const dim3 grid((unsigned)((g1 - g0 + 255) / 256));
    if (test) hipLaunchKernelGGL(k_insert_anyq<true>, grid, dim3(256), 0, a.stream, a.P, a.tab, a.bases, a.nmask, g1, a.filter, lo, hi, gated ? 1 : 0, n_kmers, g0);
    else hipLaunchKernelGGL(k_insert_anyq<false>, grid, dim3(256), 0, a.stream, a.P, a.tab, a.bases, a.nmask, g1, a.filter, lo, hi, gated ? 1 : 0, n_kmers, g0);
    return 0;
}

int tpc_launch_query_anyq(const TpcLaunch &a, uint32_t *rmask, uint64_t lo, uint64_t hi, bool gated, unsigned long long *n_marks)
{
    const uint64_t n_words = (a.n_text >> 5) + 1;
    // whole waves: the grid covers every position of the mask words [0, n_words) -- or of a rank's chunk [g_begin, g_end) (whole tiles: multiples of 64)
    const uint64_t g0 = std::min(a.g_begin, n_words * 32), g1 = std::min(a.g_end, n_words * 32);
    if (g1 <= g0) return 0;
    const dim3 grid((unsigned)((g1 - g0 + 255) / 256));
    hipLaunchKernelGGL(k_query_anyq, grid, dim3(256), 0, a.stream, a.P, a.tab, a.bases, a.nmask, std::min(a.n_text, g1), a.filter, rmask, n_words, lo, hi, gated ? 1 : 0, n_marks, g0);
    return 0;
}

int tpc_launch_split_anyq(const TpcLaunch &a, uint32_t *emask, uint32_t *bins, uint64_t bin_size)
{
    const dim3 grid((unsigned)((a.n_text + 255) / 256));
    const bool crowded = (double)a.P.q * (double)a.n_text > (double)(a.P.lmask >> 3);
    const int phases = (a.P.q < 3 || crowded) ? a.P.q : 3;  // as tpc_pass1.hip:launch_split_q (a phase beyond q would hash with an all-zero table row)
    for (int phase = 0; phase < phases; phase++)
        hipLaunchKernelGGL(k_split_anyq, grid, dim3(256), 0, a.stream, a.P, a.tab, a.bases, a.nmask, emask, phase, a.n_text, a.filter, bins, bin_size);
    return 0;
}
