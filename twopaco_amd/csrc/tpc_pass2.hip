// tpc_pass2.hip -- second pass and output-pass kernels on gfx950: candidate-mask compaction,
// exact open-addressing filter over canonical 2-bit keys, junction key sort and id lookup.
//
// Reference (paths relative to /root/reference/src, VE.h = graphconstructor/vertexenumerator.h):
//   CandidateFinalFilteringWorker VE.h:708-829, CandidateOccurence candidateoccurence.h:8-126
//   TrueBifurcations VE.h:1228-1256, BifurcationStorage bifurcationstorage.h:27-153
//   EdgeConstructionWorker VE.h:856-993 (id lookup part), CompressedString compressedstring.h
//
// Order independence of the bifurcation flag.  The reference keeps the first occurrence's
// (prev,next) and flags a key when a later occurrence differs or both have N on one side
// (VE.h:778-796).  With P / X the sets of prev / next letters (N a letter) of a key,
//     isBif <=> count >= 2 && (|P| > 1 || |X| > 1 || N in P || N in X),
// so each slot carries two 5-bit letter sets (atomicOr) and a count (atomicAdd).
#include "tpc_device.h"
#include "tpc_internal.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>

// This file is compiled twice (csrc/Makefile): part 0 holds everything for keys of one or two words (k <= 60, what nearly
// every run uses); part 1 (-DTPC_PASS2_PART=1 -> tpc_pass2_long.o) only the kernels templated on C = 3..19 words.  The
// runtime loads a code object at the first launch from it, so a run with short keys never loads the 5 MB of long-key
// code (18 -> 3 ms of start-up).
#ifndef TPC_PASS2_PART
#define TPC_PASS2_PART 0
#endif
#if TPC_PASS2_PART == 0
#define TPC_PASS2_FN(name) name
#else
#define TPC_PASS2_FN(name) name##_long
#endif

namespace {

constexpr uint64_t EMPTY = ~0ull;
constexpr int META_NEXT_SHIFT = 5;
constexpr int META_COUNT_SHIFT = 16;
constexpr uint64_t META_MULTI = 1ull << 10;  // a second occurrence was seen (stands in for count >= 2 when no abundance cut applies)

struct Slot { uint64_t key; uint64_t meta; };  // C == 1: key = canonical packed k-mer; C > 1: key = representative mark index

// ---------------------------------------------------------------- block / wave scan helpers
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// exclusive scan over a 256-thread block; returns the block total in `total`
__device__ __forceinline__ uint32_t block_excl_scan256(uint32_t v, uint32_t *s_w, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v);
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int i = 0; i < wv; i++) base += s_w[i];
    total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    return base + inc - v;
}

__device__ __forceinline__ void wave_add64(unsigned long long *dst, unsigned v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst, (unsigned long long)v);
}

// One atomic per workgroup: same-address device atomics serialise at ~12 ns each, so per-wave
// counters from a large grid cost milliseconds (measured: 2 M of them = 19 ms in k_scan2).
__device__ __forceinline__ void block_add64(unsigned long long *dst, unsigned v, uint32_t *s_w)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (t) atomicAdd(dst, (unsigned long long)t);
    }
    __syncthreads();
}

// ---------------------------------------------------------------- mask -> ordered position list
__global__ void __launch_bounds__(256) k_mask_count(const uint32_t *__restrict__ mask, uint64_t n_words, uint64_t *block_sums)
{
    __shared__ uint32_t s_w[4];
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t c = w < n_words ? __popc(mask[w]) : 0;
    uint32_t total;
    block_excl_scan256(c, s_w, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single workgroup: exclusive scan of block_sums in place, total to *n_out (16 consecutive sums per thread and iteration)
__global__ void __launch_bounds__(256) k_scan_sums(uint64_t *block_sums, uint64_t n, unsigned long long *n_out)
{
    __shared__ unsigned long long s_w[4];
    constexpr int PER = 16;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long carry = 0;
    for (uint64_t base = 0; base < n; base += 256 * PER) {
        const uint64_t i0 = base + (uint64_t)threadIdx.x * PER;
        unsigned long long v[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) { v[j] = i0 + j < n ? block_sums[i0 + j] : 0; sum += v[j]; }
        unsigned long long inc = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        unsigned long long run = carry + inc - sum;
        for (int i = 0; i < wv; i++) run += s_w[i];
        carry += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; j++) { if (i0 + j < n) block_sums[i0 + j] = run; run += v[j]; }
    }
    if (threadIdx.x == 0) *n_out = carry;
}

__global__ void __launch_bounds__(256) k_mask_scatter(const uint32_t *__restrict__ mask, uint64_t n_words,
                                                      const uint64_t *__restrict__ block_sums, uint64_t *__restrict__ list)
{
    // the block's marks are gathered in LDS (13-bit local positions) and leave as one contiguous, coalesced run: a lane storing its own
    // 4-5 positions one after the other wrote 8 bytes at a stride of ~36 (172 us for the 44 M marks of the 62-genome workload)
    __shared__ uint32_t s_w[4];
    __shared__ uint16_t s_loc[256 * 32];
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t m = w < n_words ? mask[w] : 0;
    uint32_t total;
    uint32_t o = block_excl_scan256(__popc(m), s_w, total);
    while (m) {
        const int b = __ffs(m) - 1;
        s_loc[o++] = (uint16_t)(threadIdx.x * 32u + (uint32_t)b);
        m &= m - 1;
    }
    __syncthreads();
    const uint64_t base = block_sums[blockIdx.x], pos0 = (uint64_t)blockIdx.x * (256u * 32u);
    for (uint32_t i = threadIdx.x; i < total; i += 256) list[base + i] = pos0 + s_loc[i];
}

__global__ void k_mask_or(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint64_t n_words)
{   // ConcurrentBitVector::MergeOr, concurrentbitvector.cpp:115-122
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) dst[i] |= src[i];
}

// ---------------------------------------------------------------- k-mer -> canonical key
template <int C>
__device__ __forceinline__ void load_kmer(const uint64_t *__restrict__ bases, uint64_t g, int k, uint64_t (&fw)[C])
{   // CompressedString::CopyFromString, compressedstring.h:252-264
#pragma unroll
    for (int w = 0; w < C; w++) {
        const int rem = k - 32 * w;
        uint64_t x = rem > 0 ? tpc_text_word(bases, g + 32 * w) : 0ull;
        if (rem > 0 && rem < 32) x &= (1ull << (2 * rem)) - 1ull;
        fw[w] = x;
    }
}

template <int C>
__device__ __forceinline__ void revcomp_kmer(const uint64_t (&fw)[C], int k, uint64_t (&rc)[C])
{   // CompressedString::CopyFromReverseString, compressedstring.h:266-269
    uint64_t R[C + 2];
#pragma unroll
    for (int w = 0; w < C; w++) R[w] = tpc_revcomp_word(fw[C - 1 - w]);
    R[C] = 0; R[C + 1] = 0;
    const int s = 64 * C - 2 * k;  // < 72 because k + 4 > 32 (C - 1)
    const int ws = s >> 6, bs = s & 63;
#pragma unroll
    for (int w = 0; w < C; w++) {
        const uint64_t lo = ws ? R[w + 1] : R[w];
        const uint64_t hi = ws ? R[w + 2] : R[w + 1];
        rc[w] = bs ? ((lo >> bs) | (hi << (64 - bs))) : lo;
    }
}

template <int C>
__device__ __forceinline__ uint64_t fold_h0(const uint64_t (&x)[C], int k, const uint64_t *s_h0, int L, uint64_t lmask)
{   // CyclicHash::eat over the packed string, cyclichash.h:106-109
    uint64_t h = 0;
#pragma unroll
    for (int w = 0; w < C; w++) {
        int n = k - 32 * w;
        n = n > 32 ? 32 : n;
        uint64_t y = x[w];
        for (int t = 0; t < n; t++) { h = tpc_rotl1(h, L, lmask) ^ s_h0[y & 3]; y >>= 2; }
    }
    return h;
}

template <int C>
__device__ __forceinline__ bool keys_equal(const uint64_t (&a)[C], const uint64_t (&b)[C])
{
    bool eq = true;
#pragma unroll
    for (int w = 0; w < C; w++) eq = eq && a[w] == b[w];
    return eq;
}

// forward strand stored iff posHash0 < negHash0, tie -> LessSelfReverseComplement
// (candidateoccurence.h:34, dnachar.cpp:98-114: first differing base decides, A<C<G<T)
// One-word keys (k <= 32): the fold written out -- H(w) = XOR_t rotl(h0[w_t], k - 1 - t), and for the reverse complement r_t = 3 - w_(k-1-t):
// H(r) = XOR_t rotl(h0[3 - w_t], t) -- so both strands come from ONE walk over the forward word: a 16-byte LDS read {forward term, reverse
// term} and two XORs per base instead of two rotations, two lookups and two XORs (k_filter2 / k_emit were 75-78 % VALU issue on this
// loop: profiles/r05_sq.csv).  The table is built per workgroup from function 0's four letter hashes.
constexpr int CANON_TAB = 32 * 4;
__device__ __forceinline__ uint64_t canon_rotl(uint64_t x, int r, int L, uint64_t lmask) { return r ? ((x << r) & lmask) | (x >> (L - r)) : x; }
__device__ __forceinline__ void canon_tab_build(uint4 *s_ct, const uint64_t *s_h0, int k, int L, uint64_t lmask)  // the caller synchronises before and after
{
    for (int i = threadIdx.x; i < 4 * (k < 32 ? k : 32); i += blockDim.x) {
        const int t = i >> 2, c = i & 3;
        const uint64_t a = canon_rotl(s_h0[c], (k - 1 - t) % L, L, lmask), b = canon_rotl(s_h0[3 - c], t % L, L, lmask);
        s_ct[i] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    }
}

template <int C>
__device__ __forceinline__ bool forward_is_canonical(const uint64_t (&fw)[C], const uint64_t (&rc)[C], int k,
                                                     const uint64_t *s_h0, int L, uint64_t lmask, const uint4 *s_ct)
{
    uint64_t hp, hn;
    if constexpr (C == 1) {
        uint32_t pl = 0, ph = 0, nl = 0, nh = 0;
        uint64_t y = fw[0];
        for (int t = 0; t < k; t++) {
            const uint4 e = s_ct[4 * t + (int)(y & 3)];
            pl ^= e.x; ph ^= e.y; nl ^= e.z; nh ^= e.w;
            y >>= 2;
        }
        hp = (uint64_t)pl | ((uint64_t)ph << 32);
        hn = (uint64_t)nl | ((uint64_t)nh << 32);
    } else {
        hp = fold_h0<C>(fw, k, s_h0, L, lmask);
        hn = fold_h0<C>(rc, k, s_h0, L, lmask);
    }
    if (hp != hn) return hp < hn;
#pragma unroll
    for (int w = 0; w < C; w++) {
        const uint64_t d = fw[w] ^ rc[w];
        if (d) {
            const int grp = (__ffsll((unsigned long long)d) - 1) >> 1;
            return ((fw[w] >> (2 * grp)) & 3) < ((rc[w] >> (2 * grp)) & 3);
        }
    }
    return false;
}

template <int C>
__device__ __forceinline__ uint64_t key_hash(const uint64_t (&key)[C])
{
    uint64_t h = 0;
#pragma unroll
    for (int w = 0; w < C; w++) h = tpc_mix64(h ^ key[w]);
    return h;
}

// ---------------------------------------------------------------- exact filter (a9)
__global__ void k_table_init(Slot *table, uint64_t cap)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += stride) { table[i].key = EMPTY; table[i].meta = 0; }
}

// COUNTED = false: no abundance cut can apply (abundance >= number of marks), so the occurrence
// count collapses to "seen at least twice" = META_MULTI and a thread whose bits are already in the
// slot issues no atomic at all.  Slots are read before they are CAS-ed / OR-ed: a stale read can only
// miss bits (they are set monotonically), in which case the atomic runs as before.
template <int C, bool COUNTED>
__global__ void __launch_bounds__(256)
k_filter2(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases,
          const uint32_t *__restrict__ nmask, const uint64_t *__restrict__ marks, uint64_t n_marks, Slot *table, uint64_t cap,
          unsigned long long *overflow)
{
    __shared__ uint64_t s_h0[4];
    __shared__ uint4 s_ct[C == 1 ? CANON_TAB : 1];
    if (threadIdx.x < 4) s_h0[threadIdx.x] = tab[threadIdx.x];
    __syncthreads();
    if (C == 1) { canon_tab_build(s_ct, s_h0, P.k, P.L, P.lmask); __syncthreads(); }
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_marks) return;
    const uint64_t g = marks[idx];
    uint64_t fw[C], rc[C];
    load_kmer<C>(bases, g, P.k, fw);
    revcomp_kmer<C>(fw, P.k, rc);
    const bool fwd = forward_is_canonical<C>(fw, rc, P.k, s_h0, P.L, P.lmask, s_ct);
    int prev = tpc_text_char(bases, nmask, g - 1);
    int next = tpc_text_char(bases, nmask, g + P.k);
    if (!fwd) {  // candidateoccurence.h:43-45
        const int t = prev;
        prev = tpc_rc(next);
        next = tpc_rc(t);
    }
    uint64_t ck[C];
#pragma unroll
    for (int w = 0; w < C; w++) ck[w] = fwd ? fw[w] : rc[w];
    const uint64_t mask = cap - 1;
    uint64_t slot = key_hash<C>(ck) & mask;
    bool claimed = false;    // this thread put the key into the slot
    uint64_t seen_meta = 0;  // meta bits known to be set already
    // The table is first sized for the usual ratio of marks to distinct keys (tpc_pass2_filter); a probe sequence this long
    // means it is too full: flag it and let the host repeat the pass with the table sized for all marks being distinct.
    for (uint32_t probes = 0;; probes++) {
        if (probes >= TPC_FILTER2_PROBE_LIMIT) { *overflow = 1ull; return; }
        const uint4 raw = *reinterpret_cast<const uint4 *>(&table[slot]);
        uint64_t cur = (uint64_t)raw.x | ((uint64_t)raw.y << 32);
        seen_meta = (uint64_t)raw.z | ((uint64_t)raw.w << 32);
        if (cur == EMPTY) {
            cur = atomicCAS((unsigned long long *)&table[slot].key, (unsigned long long)EMPTY, (unsigned long long)(C == 1 ? ck[0] : idx));
            seen_meta = 0;
            if (cur == EMPTY) { claimed = true; break; }
        }
        if (C == 1) {
            if (cur == ck[0]) break;
        } else {
            // slot key = index of the first mark that claimed it; equality is decided on the immutable
            // text: same k-mer up to reverse complement <=> same canonical key
            uint64_t ofw[C], orc[C];
            load_kmer<C>(bases, marks[cur], P.k, ofw);
            if (keys_equal<C>(ofw, ck)) break;
            revcomp_kmer<C>(ofw, P.k, orc);
            if (keys_equal<C>(orc, ck)) break;
        }
        slot = (slot + 1) & mask;
    }
    unsigned long long *meta = (unsigned long long *)&table[slot].meta;
    uint64_t want = (1ull << prev) | (1ull << (META_NEXT_SHIFT + next));
    if (!COUNTED && !claimed) want |= META_MULTI;
    if ((seen_meta & want) != want) atomicOr(meta, (unsigned long long)want);
    if (COUNTED) atomicAdd(meta, 1ull << META_COUNT_SHIFT);  // CandidateOccurence::Inc, candidateoccurence.h:64-67
}

// Owner rank of every marked position's key when the exact-filter table is sharded by key hash over `world` ranks (the
// address-sharded multi-GPU pass, twopaco_amd/dist.py): all occurrences of a k-mer, on either strand, go to one rank, which
// then sees the complete (prev, next) sets and count of the key.  Uses bits of the key hash the table slot index does not.
template <int C>
__global__ void __launch_bounds__(256)
k_mark_owner(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases, const uint64_t *__restrict__ marks,
             uint64_t n_marks, uint32_t world, int32_t *__restrict__ owner)
{
    __shared__ uint64_t s_h0[4];
    __shared__ uint4 s_ct[C == 1 ? CANON_TAB : 1];
    if (threadIdx.x < 4) s_h0[threadIdx.x] = tab[threadIdx.x];
    __syncthreads();
    if (C == 1) { canon_tab_build(s_ct, s_h0, P.k, P.L, P.lmask); __syncthreads(); }
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_marks) return;
    uint64_t fw[C], rc[C];
    load_kmer<C>(bases, marks[idx], P.k, fw);
    revcomp_kmer<C>(fw, P.k, rc);
    const bool fwd = forward_is_canonical<C>(fw, rc, P.k, s_h0, P.L, P.lmask, s_ct);
    uint64_t ck[C];
#pragma unroll
    for (int w = 0; w < C; w++) ck[w] = fwd ? fw[w] : rc[w];
    owner[idx] = (int32_t)((key_hash<C>(ck) >> 40) % world);
}

// The same, text-free on the owner: every marked position becomes a record of C + 1 words -- the canonical key and
// prev | next << 3 as the canonical strand sees them (candidateoccurence.h:25-50) -- so that the rank that owns the key needs
// none of the text (records travel instead of positions: each rank then holds only its chunk of the packed text).
template <int C>
__global__ void __launch_bounds__(256)
k_mark_records(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases, const uint32_t *__restrict__ nmask,
               const uint64_t *__restrict__ marks, uint64_t n_marks, uint32_t world, uint64_t *__restrict__ records, int32_t *__restrict__ owner)
{
    __shared__ uint64_t s_h0[4];
    __shared__ uint4 s_ct[C == 1 ? CANON_TAB : 1];
    if (threadIdx.x < 4) s_h0[threadIdx.x] = tab[threadIdx.x];
    __syncthreads();
    if (C == 1) { canon_tab_build(s_ct, s_h0, P.k, P.L, P.lmask); __syncthreads(); }
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_marks) return;
    const uint64_t g = marks[idx];
    uint64_t fw[C], rc[C];
    load_kmer<C>(bases, g, P.k, fw);
    revcomp_kmer<C>(fw, P.k, rc);
    const bool fwd = forward_is_canonical<C>(fw, rc, P.k, s_h0, P.L, P.lmask, s_ct);
    int prev = tpc_text_char(bases, nmask, g - 1);
    int next = tpc_text_char(bases, nmask, g + P.k);
    if (!fwd) {  // candidateoccurence.h:43-45
        const int t = prev;
        prev = tpc_rc(next);
        next = tpc_rc(t);
    }
    uint64_t ck[C];
#pragma unroll
    for (int w = 0; w < C; w++) { ck[w] = fwd ? fw[w] : rc[w]; records[idx * (C + 1) + w] = ck[w]; }
    records[idx * (C + 1) + C] = (uint64_t)prev | ((uint64_t)next << 3);
    owner[idx] = (int32_t)((key_hash<C>(ck) >> 40) % world);
}

// k_filter2 over records instead of (position, text): C == 1: slot key = the key; C > 1: slot key = index of the record that
// claimed the slot, equality decided on the records (already canonical: no reverse complement needed).
template <int C, bool COUNTED>
__global__ void __launch_bounds__(256)
k_filter2_rec(const uint64_t *__restrict__ records, uint64_t n, Slot *table, uint64_t cap, unsigned long long *overflow)
{
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    uint64_t ck[C];
#pragma unroll
    for (int w = 0; w < C; w++) ck[w] = records[idx * (C + 1) + w];
    const uint64_t pn = records[idx * (C + 1) + C];
    const bool agg = (pn >> 63) != 0;  // a rank's whole verdict on the key so far (k_table_records): letter sets, "seen twice", count
    const int prev = (int)(pn & 7), next = (int)((pn >> 3) & 7);
    const uint64_t mask = cap - 1;
    uint64_t slot = key_hash<C>(ck) & mask;
    bool claimed = false;
    uint64_t seen_meta = 0;
    for (uint32_t probes = 0;; probes++) {
        if (probes >= TPC_FILTER2_PROBE_LIMIT) { *overflow = 1ull; return; }
        const uint4 raw = *reinterpret_cast<const uint4 *>(&table[slot]);
        uint64_t cur = (uint64_t)raw.x | ((uint64_t)raw.y << 32);
        seen_meta = (uint64_t)raw.z | ((uint64_t)raw.w << 32);
        if (cur == EMPTY) {
            cur = atomicCAS((unsigned long long *)&table[slot].key, (unsigned long long)EMPTY, (unsigned long long)(C == 1 ? ck[0] : idx));
            seen_meta = 0;
            if (cur == EMPTY) { claimed = true; break; }
        }
        if (C == 1) {
            if (cur == ck[0]) break;
        } else {
            uint64_t ok[C];
#pragma unroll
            for (int w = 0; w < C; w++) ok[w] = records[cur * (C + 1) + w];
            if (keys_equal<C>(ok, ck)) break;
        }
        slot = (slot + 1) & mask;
    }
    unsigned long long *meta = (unsigned long long *)&table[slot].meta;
    uint64_t want = agg ? (pn & ((1ull << (2 * META_NEXT_SHIFT)) - 1)) : (1ull << prev) | (1ull << (META_NEXT_SHIFT + next));
    if (!COUNTED && (!claimed || (agg && (pn & META_MULTI)))) want |= META_MULTI;
    if ((seen_meta & want) != want) atomicOr(meta, (unsigned long long)want);
    if (COUNTED) atomicAdd(meta, agg ? (pn & ~(1ull << 63)) >> META_COUNT_SHIFT << META_COUNT_SHIFT : 1ull << META_COUNT_SHIFT);
}

__device__ __forceinline__ bool slot_is_junction(const Slot &sl, uint64_t abundance, bool counted);

// k_scan2_write over a table built by k_filter2_rec
template <int C>
__global__ void __launch_bounds__(256)
k_scan2_write_rec(const uint64_t *__restrict__ records, const Slot *__restrict__ table, uint64_t cap, uint64_t chunk, uint64_t abundance, int counted,
                  const uint64_t *__restrict__ block_off, uint64_t *keys_out)
{
    __shared__ uint32_t s_w[4];
    const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = min(cap, lo + chunk);
    uint64_t base = block_off[blockIdx.x];
    for (uint64_t s0 = lo; s0 < hi; s0 += 256) {
        const uint64_t s = s0 + threadIdx.x;
        Slot sl;
        sl.key = EMPTY; sl.meta = 0;
        if (s < hi) {
            const uint4 raw = *reinterpret_cast<const uint4 *>(&table[s]);
            sl.key = (uint64_t)raw.x | ((uint64_t)raw.y << 32);
            sl.meta = (uint64_t)raw.z | ((uint64_t)raw.w << 32);
        }
        const bool tp = sl.key != EMPTY && slot_is_junction(sl, abundance, counted != 0);
        uint32_t total;
        const uint32_t ex = block_excl_scan256(tp ? 1u : 0u, s_w, total);
        if (tp) {
            const uint64_t o = base + ex;
            if (C == 1) keys_out[o] = sl.key;
            else {
#pragma unroll
                for (int w = 0; w < C; w++) keys_out[o * C + w] = records[sl.key * (C + 1) + w];
            }
        }
        base += total;
    }
}

// TrueBifurcations (VE.h:1228-1256) without global atomics: every workgroup owns a contiguous chunk
// of the table; pass 1 counts (used slots, true junctions) per chunk, a scan turns the counts into
// offsets, pass 2 walks the same chunk again and writes the junction keys at its offset.
__device__ __forceinline__ bool slot_is_junction(const Slot &sl, uint64_t abundance, bool counted)
{
    const uint64_t cnt = sl.meta >> META_COUNT_SHIFT;
    const unsigned pm = (unsigned)sl.meta & 31u, nm = (unsigned)(sl.meta >> META_NEXT_SHIFT) & 31u;
    const bool twice = counted ? cnt >= 2 : (sl.meta & META_MULTI) != 0;
    const bool bif = twice && (__popc(pm) > 1 || __popc(nm) > 1 || (pm & 16u) || (nm & 16u));
    return bif && (!counted || cnt <= abundance);
}

__global__ void __launch_bounds__(256)
k_scan2_count(const Slot *__restrict__ table, uint64_t cap, uint64_t chunk, uint64_t abundance, int counted,
              uint64_t *block_tp, uint64_t *block_used)
{
    __shared__ uint32_t s_w[4];
    const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = min(cap, lo + chunk);
    unsigned used = 0, tp = 0;
    for (uint64_t s = lo + threadIdx.x; s < hi; s += 256) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(&table[s]);
        Slot sl;
        sl.key = (uint64_t)raw.x | ((uint64_t)raw.y << 32);
        sl.meta = (uint64_t)raw.z | ((uint64_t)raw.w << 32);
        if (sl.key == EMPTY) continue;
        used++;
        tp += slot_is_junction(sl, abundance, counted != 0);
    }
    uint32_t total;
    block_excl_scan256(tp, s_w, total);
    if (threadIdx.x == 0) block_tp[blockIdx.x] = total;
    block_excl_scan256(used, s_w, total);
    if (threadIdx.x == 0) block_used[blockIdx.x] = total;
}

template <int C>
__global__ void __launch_bounds__(256)
k_scan2_write(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases,
              const uint64_t *__restrict__ marks, const Slot *__restrict__ table, uint64_t cap, uint64_t chunk, uint64_t abundance,
              int counted, const uint64_t *__restrict__ block_off, uint64_t *keys_out)
{
    __shared__ uint64_t s_h0[4];
    __shared__ uint32_t s_w[4];
    __shared__ uint4 s_ct[C == 1 ? CANON_TAB : 1];
    if (threadIdx.x < 4) s_h0[threadIdx.x] = tab[threadIdx.x];
    __syncthreads();
    if (C == 1) { canon_tab_build(s_ct, s_h0, P.k, P.L, P.lmask); __syncthreads(); }
    const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = min(cap, lo + chunk);
    uint64_t base = block_off[blockIdx.x];
    for (uint64_t s0 = lo; s0 < hi; s0 += 256) {
        const uint64_t s = s0 + threadIdx.x;
        Slot sl;
        sl.key = EMPTY; sl.meta = 0;
        if (s < hi) {
            const uint4 raw = *reinterpret_cast<const uint4 *>(&table[s]);
            sl.key = (uint64_t)raw.x | ((uint64_t)raw.y << 32);
            sl.meta = (uint64_t)raw.z | ((uint64_t)raw.w << 32);
        }
        const bool tp = sl.key != EMPTY && slot_is_junction(sl, abundance, counted != 0);
        uint32_t total;
        const uint32_t ex = block_excl_scan256(tp ? 1u : 0u, s_w, total);
        if (tp) {
            const uint64_t o = base + ex;
            if (C == 1) {
                keys_out[o] = sl.key;
            } else {
                uint64_t fw[C], rc[C];
                load_kmer<C>(bases, marks[sl.key], P.k, fw);
                revcomp_kmer<C>(fw, P.k, rc);
                const bool fwd = forward_is_canonical<C>(fw, rc, P.k, s_h0, P.L, P.lmask, s_ct);
#pragma unroll
                for (int w = 0; w < C; w++) keys_out[o * C + w] = fwd ? fw[w] : rc[w];
            }
        }
        base += total;
    }
}

// Combine before routing, second pass (multi-GPU, table sharded by key hash): a rank first runs k_filter2 over ITS marks into its own
// table and sends every DISTINCT key once, with the letter sets / "seen twice" / count its occurrences add up to (the slot's meta word,
// bit 63 set: an aggregated record for k_filter2_rec), instead of one record per marked position -- on many-genome inputs a key is marked
// dozens of times.  Walks the table as k_scan2_write does; block_off = exclusive scan of k_scan2_count's used slots per chunk.
template <int C>
__global__ void __launch_bounds__(256)
k_table_records(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases, const uint64_t *__restrict__ marks,
                const Slot *__restrict__ table, uint64_t cap, uint64_t chunk, const uint64_t *__restrict__ block_off, uint32_t world,
                uint64_t *__restrict__ records, int32_t *__restrict__ owner)
{
    __shared__ uint64_t s_h0[4];
    __shared__ uint32_t s_w[4];
    __shared__ uint4 s_ct[C == 1 ? CANON_TAB : 1];
    if (threadIdx.x < 4) s_h0[threadIdx.x] = tab[threadIdx.x];
    __syncthreads();
    if (C == 1) { canon_tab_build(s_ct, s_h0, P.k, P.L, P.lmask); __syncthreads(); }
    const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = min(cap, lo + chunk);
    uint64_t base = block_off[blockIdx.x];
    for (uint64_t s0 = lo; s0 < hi; s0 += 256) {
        const uint64_t s = s0 + threadIdx.x;
        Slot sl;
        sl.key = EMPTY; sl.meta = 0;
        if (s < hi) {
            const uint4 raw = *reinterpret_cast<const uint4 *>(&table[s]);
            sl.key = (uint64_t)raw.x | ((uint64_t)raw.y << 32);
            sl.meta = (uint64_t)raw.z | ((uint64_t)raw.w << 32);
        }
        const bool used = sl.key != EMPTY;
        uint32_t total;
        const uint32_t ex = block_excl_scan256(used ? 1u : 0u, s_w, total);
        if (used) {
            const uint64_t o = base + ex;
            uint64_t ck[C];
            if (C == 1) ck[0] = sl.key;
            else {
                uint64_t fw[C], rc[C];
                load_kmer<C>(bases, marks[sl.key], P.k, fw);
                revcomp_kmer<C>(fw, P.k, rc);
                const bool fwd = forward_is_canonical<C>(fw, rc, P.k, s_h0, P.L, P.lmask, s_ct);
#pragma unroll
                for (int w = 0; w < C; w++) ck[w] = fwd ? fw[w] : rc[w];
            }
#pragma unroll
            for (int w = 0; w < C; w++) records[o * (C + 1) + w] = ck[w];
            records[o * (C + 1) + C] = sl.meta | (1ull << 63);
            owner[o] = (int32_t)((key_hash<C>(ck) >> 40) % world);
        }
        base += total;
    }
}

// ---------------------------------------------------------------- id index + output lookup
// One-word keys: 16-byte slots {key, rank} (rank word first claimed by CAS, then the key stored: readers run in a later
// kernel), so that the output pass needs one scattered load per probe.  Multi-word keys: 4-byte rank slots, keys compared
// through the sorted key array.
template <int C>
__global__ void __launch_bounds__(256) k_idtab_build(const uint64_t *__restrict__ keys, uint64_t J, uint32_t *idtab, uint64_t cap)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= J) return;
    uint64_t key[C];
#pragma unroll
    for (int w = 0; w < C; w++) key[w] = keys[i * C + w];
    const uint64_t mask = cap - 1;
    uint64_t slot = key_hash<C>(key) & mask;
    if (C == 1) {
        while (atomicCAS(&idtab[slot * 4 + 2], 0u, (uint32_t)(i + 1)) != 0u) slot = (slot + 1) & mask;
        idtab[slot * 4 + 0] = (uint32_t)key[0];
        idtab[slot * 4 + 1] = (uint32_t)(key[0] >> 32);
        return;
    }
    while (atomicCAS(&idtab[slot], 0u, (uint32_t)(i + 1)) != 0u) slot = (slot + 1) & mask;
}

// BifurcationStorage::GetId (bifurcationstorage.h:100-127) for every marked position: + if the
// forward packed form is the stored key, - if its reverse complement is.
template <int C>
__global__ void __launch_bounds__(256)
k_emit(TpcHashParams P, const uint64_t *__restrict__ tab, const uint64_t *__restrict__ bases,
       const uint64_t *__restrict__ marks, uint64_t n_marks, const uint64_t *__restrict__ keys, uint64_t J,
       const uint32_t *__restrict__ idtab, uint64_t cap, int64_t *__restrict__ ids, unsigned long long *n_valid)
{
    __shared__ uint64_t s_h0[4];
    __shared__ uint4 s_ct[C == 1 ? CANON_TAB : 1];
    if (threadIdx.x < 4) s_h0[threadIdx.x] = tab[threadIdx.x];
    __syncthreads();
    if (C == 1) { canon_tab_build(s_ct, s_h0, P.k, P.L, P.lmask); __syncthreads(); }
    __shared__ uint32_t s_w[4];
    unsigned valid = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_marks; idx += stride) {
        int64_t id = INT64_MAX;
        if (J > 0) {
            const uint64_t g = marks[idx];
            uint64_t fw[C], rc[C], ck[C];
            load_kmer<C>(bases, g, P.k, fw);
            revcomp_kmer<C>(fw, P.k, rc);
            const bool fwd = forward_is_canonical<C>(fw, rc, P.k, s_h0, P.L, P.lmask, s_ct);
#pragma unroll
            for (int w = 0; w < C; w++) ck[w] = fwd ? fw[w] : rc[w];
            const uint64_t mask = cap - 1;
            uint64_t slot = key_hash<C>(ck) & mask;
            for (;;) {
                uint32_t r;
                uint64_t sk[C];
                if (C == 1) {
                    const uint4 sl = reinterpret_cast<const uint4 *>(idtab)[slot];
                    r = sl.z;
                    if (r == 0) break;
                    sk[0] = (uint64_t)sl.x | ((uint64_t)sl.y << 32);
                } else {
                    r = idtab[slot];
                    if (r == 0) break;
#pragma unroll
                    for (int w = 0; w < C; w++) sk[w] = keys[(uint64_t)(r - 1) * C + w];
                }
                if (keys_equal<C>(sk, ck)) {
                    id = keys_equal<C>(ck, fw) ? (int64_t)r : -(int64_t)r;
                    valid++;
                    break;
                }
                slot = (slot + 1) & mask;
            }
        }
        ids[idx] = id;
    }
    block_add64(n_valid, valid, s_w);
}

// gather / permutation helpers for the multi-word key sort
__global__ void k_iota(uint32_t *p, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (uint32_t)i;
}
__global__ void k_gather_word(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ perm, uint64_t n, int C, int w, uint64_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = keys[(uint64_t)perm[i] * C + w];
}
__global__ void k_gather_keys(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ perm, uint64_t n, int C, uint64_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        for (int w = 0; w < C; w++) out[i * C + w] = keys[(uint64_t)perm[i] * C + w];
}

inline unsigned nblk(uint64_t n, unsigned b) { return (unsigned)((n + b - 1) / b); }

}  // namespace

#if TPC_PASS2_PART == 0
#define TPC_DISPATCH_C(C_, CALL)                                                                  \
    switch (C_) {                                                                                 \
    case 1: CALL(1); break; case 2: CALL(2); break;                                               \
    default: return -1;                                                                           \
    }
#else
#define TPC_DISPATCH_C(C_, CALL)                                                                  \
    switch (C_) {                                                                                 \
    case 3: CALL(3); break; case 4: CALL(4); break;                                               \
    case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break;       \
    case 9: CALL(9); break; case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break; \
    case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break; case 16: CALL(16); break; \
    case 17: CALL(17); break; case 18: CALL(18); break; case 19: CALL(19); break;                 \
    default: return -1;                                                                           \
    }
#endif
// the long-key halves of the four dispatching launchers (tpc_pass2_long.o)
int tpc_launch_filter2_long(const TpcLaunch &a, int C, const uint64_t *marks, uint64_t n_marks, void *table, uint64_t cap, bool counted,
                            unsigned long long *overflow);
int tpc_launch_scan2_write_long(const TpcLaunch &a, int C, const uint64_t *marks, const void *table, uint64_t cap, uint64_t abundance, bool counted,
                                const uint64_t *block_off, uint64_t *keys_out);
int tpc_launch_idtab_build_long(hipStream_t s, int C, const uint64_t *keys, uint64_t J, uint32_t *idtab, uint64_t cap);
int tpc_launch_sort_keys_long(hipStream_t s, int C, uint64_t *keys, uint64_t J);  // multi-word keys (C >= 2)
int tpc_launch_emit_long(const TpcLaunch &a, int C, const uint64_t *marks, uint64_t n_marks, const uint64_t *keys, uint64_t J,
                         const uint32_t *idtab, uint64_t cap, int64_t *ids, unsigned long long *n_valid);

#if TPC_PASS2_PART == 0
// The same in-place exclusive scan by up to 64 workgroups (one took 43 us for the 38 K block sums of the 62-genome text, twice per step):
// (a) every workgroup adds up its segment of 4096 sums, (b) adds the totals of the segments before it and scans its own.  seg_tot: 64 words of scratch.
constexpr uint32_t SCAN_SEG = 4096;
__global__ void __launch_bounds__(256) k_scan_sums_seg(const uint64_t *__restrict__ block_sums, uint64_t n, unsigned long long *__restrict__ seg_tot)
{
    __shared__ unsigned long long s64[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * SCAN_SEG, i1 = min(n, i0 + SCAN_SEG);
    unsigned long long sum = 0;
    for (uint64_t i = i0 + threadIdx.x; i < i1; i += 256) sum += block_sums[i];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((threadIdx.x & 63) == 0) s64[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) seg_tot[blockIdx.x] = s64[0] + s64[1] + s64[2] + s64[3];
}
__global__ void __launch_bounds__(256) k_scan_sums_wide(uint64_t *block_sums, uint64_t n, const unsigned long long *__restrict__ seg_tot, unsigned long long *n_out)
{
    __shared__ unsigned long long s_w[4];
    constexpr int PER = SCAN_SEG / 256;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long base = 0, all = 0;
    for (uint32_t g = 0; g < gridDim.x; g++) { const unsigned long long t = seg_tot[g]; if (g < blockIdx.x) base += t; all += t; }
    const uint64_t i0 = (uint64_t)blockIdx.x * SCAN_SEG + (uint64_t)threadIdx.x * PER;
    unsigned long long v[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) { v[j] = i0 + j < n ? block_sums[i0 + j] : 0; sum += v[j]; }
    unsigned long long inc = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    unsigned long long run = base + inc - sum;
    for (int i = 0; i < wv; i++) run += s_w[i];
#pragma unroll
    for (int j = 0; j < PER; j++) { if (i0 + j < n) block_sums[i0 + j] = run; run += v[j]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = all;
}

int tpc_launch_mask_count(hipStream_t s, const uint32_t *mask, uint64_t n_words, uint64_t *block_sums, unsigned long long *n_out)
{   // block_sums: nblk(n_words, 256) sums + 64 words of scratch behind them (TPC_MASK_SUMS_SCRATCH)
    const unsigned nb = nblk(n_words, 256);
    hipLaunchKernelGGL(k_mask_count, dim3(nb), dim3(256), 0, s, mask, n_words, block_sums);
    const unsigned segs = (nb + SCAN_SEG - 1) / SCAN_SEG;
    if (segs < 2 || segs > 64) { hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, s, block_sums, (uint64_t)nb, n_out); return 0; }
    unsigned long long *seg_tot = reinterpret_cast<unsigned long long *>(block_sums + nb);
    hipLaunchKernelGGL(k_scan_sums_seg, dim3(segs), dim3(256), 0, s, block_sums, (uint64_t)nb, seg_tot);
    hipLaunchKernelGGL(k_scan_sums_wide, dim3(segs), dim3(256), 0, s, block_sums, (uint64_t)nb, seg_tot, n_out);
    return 0;
}

int tpc_launch_mask_scatter(hipStream_t s, const uint32_t *mask, uint64_t n_words, const uint64_t *block_sums, uint64_t *list)
{
    hipLaunchKernelGGL(k_mask_scatter, dim3(nblk(n_words, 256)), dim3(256), 0, s, mask, n_words, block_sums, list);
    return 0;
}

int tpc_launch_mask_or(hipStream_t s, uint32_t *dst, const uint32_t *src, uint64_t n_words)
{
    hipLaunchKernelGGL(k_mask_or, dim3(2048), dim3(256), 0, s, dst, src, n_words);
    return 0;
}

size_t tpc_table_slot_bytes(int) { return sizeof(Slot); }

int tpc_launch_table_init(hipStream_t s, void *table, uint64_t cap)
{
    hipLaunchKernelGGL(k_table_init, dim3(2048), dim3(256), 0, s, (Slot *)table, cap);
    return 0;
}

#endif  // part 0 only

int TPC_PASS2_FN(tpc_launch_filter2)(const TpcLaunch &a, int C, const uint64_t *marks, uint64_t n_marks, void *table, uint64_t cap, bool counted,
                       unsigned long long *overflow)
{
    if (n_marks == 0) return 0;
#if TPC_PASS2_PART == 0
    if (C > 2) return tpc_launch_filter2_long(a, C, marks, n_marks, table, cap, counted, overflow);
#endif
#define CALL(C_)                                                                                                                         \
    if (counted) hipLaunchKernelGGL((k_filter2<C_, true>), dim3(nblk(n_marks, 256)), dim3(256), 0, a.stream, a.P, a.tab, a.bases, a.nmask, \
                                    marks, n_marks, (Slot *)table, cap, overflow);                                                       \
    else hipLaunchKernelGGL((k_filter2<C_, false>), dim3(nblk(n_marks, 256)), dim3(256), 0, a.stream, a.P, a.tab, a.bases, a.nmask, marks, \
                            n_marks, (Slot *)table, cap, overflow)
    TPC_DISPATCH_C(C, CALL)
#undef CALL
    return 0;
}

int tpc_launch_mark_owner_long(const TpcLaunch &a, int C, const uint64_t *marks, uint64_t n_marks, uint32_t world, int32_t *owner);
int TPC_PASS2_FN(tpc_launch_mark_owner)(const TpcLaunch &a, int C, const uint64_t *marks, uint64_t n_marks, uint32_t world, int32_t *owner)
{
    if (n_marks == 0) return 0;
#if TPC_PASS2_PART == 0
    if (C > 2) return tpc_launch_mark_owner_long(a, C, marks, n_marks, world, owner);
#endif
#define CALL(C_) hipLaunchKernelGGL((k_mark_owner<C_>), dim3(nblk(n_marks, 256)), dim3(256), 0, a.stream, a.P, a.tab, a.bases, marks, n_marks, world, owner)
    TPC_DISPATCH_C(C, CALL)
#undef CALL
    return 0;
}

int tpc_launch_mark_records_long(const TpcLaunch &a, int C, const uint64_t *marks, uint64_t n_marks, uint32_t world, uint64_t *records, int32_t *owner);
int TPC_PASS2_FN(tpc_launch_mark_records)(const TpcLaunch &a, int C, const uint64_t *marks, uint64_t n_marks, uint32_t world, uint64_t *records, int32_t *owner)
{
    if (n_marks == 0) return 0;
#if TPC_PASS2_PART == 0
    if (C > 2) return tpc_launch_mark_records_long(a, C, marks, n_marks, world, records, owner);
#endif
#define CALL(C_) hipLaunchKernelGGL((k_mark_records<C_>), dim3(nblk(n_marks, 256)), dim3(256), 0, a.stream, a.P, a.tab, a.bases, a.nmask, marks, n_marks, world, records, owner)
    TPC_DISPATCH_C(C, CALL)
#undef CALL
    return 0;
}

int tpc_launch_table_records_long(const TpcLaunch &a, int C, const uint64_t *marks, const void *table, uint64_t cap, const uint64_t *block_off, uint32_t world,
                                  uint64_t *records, int32_t *owner);
uint64_t TPC_PASS2_FN(tpc_scan2_chunk)(uint64_t cap);
int TPC_PASS2_FN(tpc_launch_table_records)(const TpcLaunch &a, int C, const uint64_t *marks, const void *table, uint64_t cap, const uint64_t *block_off, uint32_t world,
                                           uint64_t *records, int32_t *owner)
{
#if TPC_PASS2_PART == 0
    if (C > 2) return tpc_launch_table_records_long(a, C, marks, table, cap, block_off, world, records, owner);
#endif
    const uint64_t chunk = TPC_PASS2_FN(tpc_scan2_chunk)(cap);
#define CALL(C_) hipLaunchKernelGGL((k_table_records<C_>), dim3(TPC_SCAN2_BLOCKS), dim3(256), 0, a.stream, a.P, a.tab, a.bases, marks, (const Slot *)table, cap, chunk, block_off, world, records, owner)
    TPC_DISPATCH_C(C, CALL)
#undef CALL
    return 0;
}

int tpc_launch_filter2_rec_long(const TpcLaunch &a, int C, const uint64_t *records, uint64_t n, void *table, uint64_t cap, bool counted, unsigned long long *overflow);
int TPC_PASS2_FN(tpc_launch_filter2_rec)(const TpcLaunch &a, int C, const uint64_t *records, uint64_t n, void *table, uint64_t cap, bool counted,
                                         unsigned long long *overflow)
{
    if (n == 0) return 0;
#if TPC_PASS2_PART == 0
    if (C > 2) return tpc_launch_filter2_rec_long(a, C, records, n, table, cap, counted, overflow);
#endif
#define CALL(C_)                                                                                                                                  \
    if (counted) hipLaunchKernelGGL((k_filter2_rec<C_, true>), dim3(nblk(n, 256)), dim3(256), 0, a.stream, records, n, (Slot *)table, cap, overflow); \
    else hipLaunchKernelGGL((k_filter2_rec<C_, false>), dim3(nblk(n, 256)), dim3(256), 0, a.stream, records, n, (Slot *)table, cap, overflow)
    TPC_DISPATCH_C(C, CALL)
#undef CALL
    return 0;
}

uint64_t TPC_PASS2_FN(tpc_scan2_chunk)(uint64_t cap);
int tpc_launch_scan2_write_rec_long(const TpcLaunch &a, int C, const uint64_t *records, const void *table, uint64_t cap, uint64_t abundance, bool counted,
                                    const uint64_t *block_off, uint64_t *keys_out);
int TPC_PASS2_FN(tpc_launch_scan2_write_rec)(const TpcLaunch &a, int C, const uint64_t *records, const void *table, uint64_t cap, uint64_t abundance, bool counted,
                                             const uint64_t *block_off, uint64_t *keys_out)
{
#if TPC_PASS2_PART == 0
    if (C > 2) return tpc_launch_scan2_write_rec_long(a, C, records, table, cap, abundance, counted, block_off, keys_out);
#endif
    const uint64_t chunk = TPC_PASS2_FN(tpc_scan2_chunk)(cap);
#define CALL(C_) hipLaunchKernelGGL((k_scan2_write_rec<C_>), dim3(TPC_SCAN2_BLOCKS), dim3(256), 0, a.stream, records, (const Slot *)table, cap, chunk, abundance, counted ? 1 : 0, block_off, keys_out)
    TPC_DISPATCH_C(C, CALL)
#undef CALL
    return 0;
}

uint64_t TPC_PASS2_FN(tpc_scan2_chunk)(uint64_t cap) { return ((cap + TPC_SCAN2_BLOCKS - 1) / TPC_SCAN2_BLOCKS + 255) / 256 * 256; }

#if TPC_PASS2_PART == 0
int tpc_launch_scan2_count(const TpcLaunch &a, const void *table, uint64_t cap, uint64_t abundance, bool counted, uint64_t *block_tp,
                           uint64_t *block_used, unsigned long long *totals)
{
    const uint64_t chunk = tpc_scan2_chunk(cap);
    hipLaunchKernelGGL(k_scan2_count, dim3(TPC_SCAN2_BLOCKS), dim3(256), 0, a.stream, (const Slot *)table, cap, chunk, abundance, counted ? 1 : 0,
                       block_tp, block_used);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, a.stream, block_tp, (uint64_t)TPC_SCAN2_BLOCKS, totals);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, a.stream, block_used, (uint64_t)TPC_SCAN2_BLOCKS, totals + 1);
    return 0;
}

#endif  // part 0 only

int TPC_PASS2_FN(tpc_launch_scan2_write)(const TpcLaunch &a, int C, const uint64_t *marks, const void *table, uint64_t cap, uint64_t abundance, bool counted,
                           const uint64_t *block_off, uint64_t *keys_out)
{
#if TPC_PASS2_PART == 0
    if (C > 2) return tpc_launch_scan2_write_long(a, C, marks, table, cap, abundance, counted, block_off, keys_out);
#endif
    const uint64_t chunk = TPC_PASS2_FN(tpc_scan2_chunk)(cap);
#define CALL(C_) hipLaunchKernelGGL((k_scan2_write<C_>), dim3(TPC_SCAN2_BLOCKS), dim3(256), 0, a.stream, a.P, a.tab, a.bases, marks, (const Slot *)table, cap, chunk, abundance, counted ? 1 : 0, block_off, keys_out)
    TPC_DISPATCH_C(C, CALL)
#undef CALL
    return 0;
}

#if TPC_PASS2_PART == 0
int tpc_launch_sort_keys(hipStream_t s, int C, int k, uint64_t *keys, uint64_t J, void **scratch, size_t *scratch_bytes)
{
    if (J < 2) return 0;
    hipError_t e;
    if (C == 1) {
        // one-word keys (k <= 28): radix sort on the 2k significant bits; output + temporary storage live in the caller's
        // scratch buffer, which only grows (no allocation in the steady state)
        size_t tmp_bytes = 0;
        int end_bit = 2 * k;
        if (end_bit > 64) end_bit = 64;
        uint64_t *out = nullptr;
        if ((e = rocprim::radix_sort_keys(nullptr, tmp_bytes, keys, out, J, 0, end_bit, s)) != hipSuccess) return -2;
        const size_t out_bytes = (J * sizeof(uint64_t) + 255) & ~(size_t)255;
        const size_t need = out_bytes + (tmp_bytes ? tmp_bytes : 8);
        if (need > *scratch_bytes) {
            if (*scratch) (void)hipFree(*scratch);
            *scratch = nullptr; *scratch_bytes = 0;
            if (hipMalloc(scratch, need + need / 4) != hipSuccess) return -3;
            *scratch_bytes = need + need / 4;
        }
        out = (uint64_t *)*scratch;
        void *tmp = (char *)*scratch + out_bytes;
        e = rocprim::radix_sort_keys(tmp, tmp_bytes, keys, out, J, 0, end_bit, s);
        if (e == hipSuccess) e = hipMemcpyAsync(keys, out, J * sizeof(uint64_t), hipMemcpyDeviceToDevice, s);
        return e == hipSuccess ? 0 : -2;
    }
    return tpc_launch_sort_keys_long(s, C, keys, J);
}
#endif  // part 0 only

#if TPC_PASS2_PART == 1
int tpc_launch_sort_keys_long(hipStream_t s, int C, uint64_t *keys, uint64_t J)
{
    // C > 1: CompressedString::Less compares word 0 first (compressedstring.h:93-104), so an LSD
    // pass sequence sorts by word C-1 first and word 0 last, each pass stable.
    uint64_t *kw = nullptr, *kw2 = nullptr, *out = nullptr;
    uint32_t *perm = nullptr, *perm2 = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    int rc = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, kw, kw2, perm, perm2, J, 0, 64, s) != hipSuccess) return -2;
    if (hipMalloc(&kw, J * 8) != hipSuccess || hipMalloc(&kw2, J * 8) != hipSuccess || hipMalloc(&out, J * C * 8) != hipSuccess ||
        hipMalloc(&perm, J * 4) != hipSuccess || hipMalloc(&perm2, J * 4) != hipSuccess || hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8) != hipSuccess) {
        rc = -3;
    } else {
        hipLaunchKernelGGL(k_iota, dim3(nblk(J, 256)), dim3(256), 0, s, perm, J);
        for (int w = C - 1; w >= 0 && rc == 0; w--) {
            hipLaunchKernelGGL(k_gather_word, dim3(nblk(J, 256)), dim3(256), 0, s, keys, perm, J, C, w, kw);
            if (rocprim::radix_sort_pairs(tmp, tmp_bytes, kw, kw2, perm, perm2, J, 0, 64, s) != hipSuccess) rc = -2;
            uint32_t *t = perm; perm = perm2; perm2 = t;
        }
        if (rc == 0) {
            hipLaunchKernelGGL(k_gather_keys, dim3(nblk(J, 256)), dim3(256), 0, s, keys, perm, J, C, out);
            if (hipMemcpyAsync(keys, out, J * C * 8, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = -2;
        }
        (void)hipStreamSynchronize(s);
    }
    (void)hipFree(kw); (void)hipFree(kw2); (void)hipFree(out); (void)hipFree(perm); (void)hipFree(perm2); (void)hipFree(tmp);
    return rc;
}
#endif  // part 1 only

int TPC_PASS2_FN(tpc_launch_idtab_build)(hipStream_t s, int C, const uint64_t *keys, uint64_t J, uint32_t *idtab, uint64_t cap)
{
    if (J == 0) return 0;
#if TPC_PASS2_PART == 0
    if (C > 2) return tpc_launch_idtab_build_long(s, C, keys, J, idtab, cap);
#endif
#define CALL(C_) hipLaunchKernelGGL((k_idtab_build<C_>), dim3(nblk(J, 256)), dim3(256), 0, s, keys, J, idtab, cap)
    TPC_DISPATCH_C(C, CALL)
#undef CALL
    return 0;
}

int TPC_PASS2_FN(tpc_launch_emit)(const TpcLaunch &a, int C, const uint64_t *marks, uint64_t n_marks, const uint64_t *keys, uint64_t J,
                    const uint32_t *idtab, uint64_t cap, int64_t *ids, unsigned long long *n_valid)
{
    if (n_marks == 0) return 0;
#if TPC_PASS2_PART == 0
    if (C > 2) return tpc_launch_emit_long(a, C, marks, n_marks, keys, J, idtab, cap, ids, n_valid);
#endif
#define CALL(C_) hipLaunchKernelGGL((k_emit<C_>), dim3(std::min<unsigned>(nblk(n_marks, 256), 16384u)), dim3(256), 0, a.stream, a.P, a.tab, a.bases, marks, n_marks, keys, J, idtab, cap, ids, n_valid)
    TPC_DISPATCH_C(C, CALL)
#undef CALL
    return 0;
}

#if TPC_PASS2_PART == 0
// tpc_preload: the first use of any kernel of this translation unit makes the runtime load its code object
__global__ void k_warm_pass2() {}
int tpc_warm_pass2() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_warm_pass2)) == hipSuccess ? 0 : -1; }
#endif
