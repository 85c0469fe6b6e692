// tpc_device.h -- device-side building blocks shared by the gfx950 kernels.
//
// Reference semantics restated here (paths relative to /root/reference/src):
//   cyclic-polynomial hash  common/ngramhashing/cyclichash.h:42-121
//   both-strand vertex hash graphconstructor/vertexrollinghash.h:79-200
//   2-bit key packing       graphconstructor/compressedstring.h:188-195,252-264
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TPC_TILE_THREADS 256                 // threads per workgroup = 4 waves of 64
#define TPC_RUN 32                           // vertex positions per thread = one packed word
#define TPC_TILE_POS (TPC_TILE_THREADS * TPC_RUN)
#define TPC_XW_MAX 22                        // halo words for k <= 600
#define TPC_TILE_WORDS (TPC_TILE_THREADS + 1 + TPC_XW_MAX)
#define TPC_CODE_N 4

struct TpcHashParams {
    int k, L, q, rk;   // rk = k % L  (CyclicHash::myr, cyclichash.h:33)
    uint64_t lmask;    // 2^L - 1
};

// rotations inside L bits: fastleftshift1 / fastrightshift1 / fastleftshiftn (cyclichash.h:42-52)
__device__ __forceinline__ uint64_t tpc_rotl1(uint64_t x, int L, uint64_t lmask) { return ((x << 1) & lmask) | (x >> (L - 1)); }
__device__ __forceinline__ uint64_t tpc_rotr1(uint64_t x, int L) { return (x >> 1) | ((x & 1ull) << (L - 1)); }
__device__ __forceinline__ int tpc_rc(int c) { return c == TPC_CODE_N ? TPC_CODE_N : 3 - c; }  // dnachar.cpp:52-58

__device__ __forceinline__ uint64_t tpc_min(uint64_t a, uint64_t b) { return a < b ? a : b; }

// Character at global position g out of the LDS-staged tile (word index relative to wbase).
__device__ __forceinline__ int tpc_tile_char(const uint64_t *sb, const uint32_t *sn, uint64_t g, uint64_t wbase)
{
    const uint32_t lw = (uint32_t)((g >> 5) - wbase);
    const uint32_t o = (uint32_t)g & 31u;
    const uint32_t isn = (sn[lw] >> o) & 1u;
    const uint32_t code = (uint32_t)(sb[lw] >> (2u * o)) & 3u;
    return isn ? TPC_CODE_N : (int)code;
}

// Stage the packed words [wfirst-1, wfirst+256+xw] of the text into LDS.  The host pads both
// arrays with N words, so only word -1 needs a guard.
__device__ __forceinline__ void tpc_stage_tile(uint64_t *sb, uint32_t *sn, const uint64_t *__restrict__ bases,
                                               const uint32_t *__restrict__ nmask, uint64_t wfirst, int xw)
{
    const int nload = TPC_TILE_THREADS + 1 + xw;
    for (int i = threadIdx.x; i < nload; i += TPC_TILE_THREADS) {
        const int64_t w = (int64_t)wfirst - 1 + i;
        sb[i] = w >= 0 ? bases[w] : 0ull;
        sn[i] = w >= 0 ? nmask[w] : 0xFFFFFFFFu;
    }
}

// The 2Q running hashes of one vertex window (VertexRollingHash::posVertexHash_/negVertexHash_).
template <int Q>
struct TpcVHash {
    uint64_t pos[Q], neg[Q];
};

// VertexRollingHash ctor (vertexrollinghash.h:79-102): eat the window left to right on the
// positive strand and its reverse complement on the negative strand (eat: cyclichash.h:106-109).
template <int Q>
__device__ __forceinline__ void tpc_vhash_init(TpcVHash<Q> &v, const TpcHashParams &P, const uint64_t *s_h,
                                               const uint64_t *sb, const uint32_t *sn, uint64_t g0, uint64_t wbase)
{
#pragma unroll
    for (int i = 0; i < Q; i++) { v.pos[i] = 0; v.neg[i] = 0; }
    for (int t = 0; t < P.k; t++) {
        const int c = tpc_tile_char(sb, sn, g0 + t, wbase);
        const int cr = tpc_rc(tpc_tile_char(sb, sn, g0 + P.k - 1 - t, wbase));
#pragma unroll
        for (int i = 0; i < Q; i++) {
            v.pos[i] = tpc_rotl1(v.pos[i], P.L, P.lmask) ^ s_h[i * 5 + c];
            v.neg[i] = tpc_rotl1(v.neg[i], P.L, P.lmask) ^ s_h[i * 5 + cr];
        }
    }
}

// Canonical strand of an edge: first hash function whose two strand values differ decides,
// smaller wins, all equal -> positive (DetermineStrandExtend/Prepend, vertexrollinghash.h:170-200).
template <int Q>
__device__ __forceinline__ bool tpc_pick_neg(const uint64_t (&p)[Q], const uint64_t (&n)[Q])
{
    bool neg = false, decided = false;
#pragma unroll
    for (int i = 0; i < Q; i++) {
        if (!decided && p[i] != n[i]) { neg = n[i] < p[i]; decided = true; }
    }
    return neg;
}

// ---- direct (non-tiled) text access for the per-mark kernels -------------------------------
__device__ __forceinline__ int tpc_text_char(const uint64_t *__restrict__ bases, const uint32_t *__restrict__ nmask, uint64_t g)
{
    const uint32_t o = (uint32_t)g & 31u;
    const uint32_t isn = (nmask[g >> 5] >> o) & 1u;
    const uint32_t code = (uint32_t)(bases[g >> 5] >> (2u * o)) & 3u;
    return isn ? TPC_CODE_N : (int)code;
}

// 32 bases starting at g (any alignment) as one packed word.
__device__ __forceinline__ uint64_t tpc_text_word(const uint64_t *__restrict__ bases, uint64_t g)
{
    const uint32_t o = (uint32_t)g & 31u;
    const uint64_t lo = bases[g >> 5];
    if (o == 0) return lo;
    const uint64_t hi = bases[(g >> 5) + 1];
    return (lo >> (2u * o)) | (hi << (64u - 2u * o));
}

// The same with ONE 16-byte request (an 8-byte aligned dwordx4 load; the packed text is padded, so the second word always exists):
// the per-survivor kernels wait for scattered accesses, and two 8-byte loads of neighbouring words are two requests.
typedef uint64_t tpc_u64x2 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ uint64_t tpc_text_word_x2(const uint64_t *__restrict__ bases, uint64_t g)
{
    const uint32_t o = (uint32_t)g & 31u;
    const tpc_u64x2 v = *reinterpret_cast<const tpc_u64x2 *>(bases + (g >> 5));
    return o ? (v.x >> (2u * o)) | (v.y << (64u - 2u * o)) : v.x;
}

// Reverse the 32 two-bit groups of a word and complement them (A<->T, C<->G = code ^ 3).
__device__ __forceinline__ uint64_t tpc_revcomp_word(uint64_t x)
{
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = __builtin_bswap64(x);
    return ~x;
}

__device__ __forceinline__ uint64_t tpc_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
