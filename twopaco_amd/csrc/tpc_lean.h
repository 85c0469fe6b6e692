// tpc_lean.h -- instruction-lean building blocks for the first-pass hash kernels.
//
// The SQ counters (profiles/r03a_sq.csv) put the binning kernels at 60-77 % VALU issue: they are bound by the NUMBER of
// vector instructions, not by LDS or HBM.  The helpers of tpc_device.h are written for clarity -- one character at a time
// out of LDS with 64-bit index arithmetic (tpc_tile_char: ~12 instructions), rotations as 64-bit shifts, the slice-index
// permutation assembled as a 64-bit address and taken apart again.  Here the same arithmetic (reference cyclichash.h:42-121,
// vertexrollinghash.h:104-200) is restated on 32-bit halves with the characters of a thread's run held in registers.
#pragma once
#include "tpc_bins.h"

// ---- L-bit rotations on a 64-bit value, 5-6 instructions each (L in 2..62, scalars precomputed once per kernel)
struct LeanRot {
    uint32_t lomask, himask;   // 2^L - 1 as two halves
    uint32_t top_shift;        // L - 1
    uint32_t sh, mlo, mhi;     // rotr1: where the wrapped bit lands
    __device__ __forceinline__ void set(int L)
    {
        const uint64_t lm = (1ull << L) - 1ull;
        lomask = (uint32_t)lm; himask = (uint32_t)(lm >> 32);
        top_shift = (uint32_t)(L - 1);
        sh = (uint32_t)(L - 1) & 31u;
        mlo = L - 1 < 32 ? 0xFFFFFFFFu : 0u;
        mhi = ~mlo;
    }
    // fastleftshift1 (cyclichash.h:42-44)
    __device__ __forceinline__ uint64_t rotl1(uint64_t x) const
    {
        const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
        const uint32_t bit = (uint32_t)(x >> top_shift);  // 0 or 1: x < 2^L
        const uint32_t nlo = ((lo << 1) | bit) & lomask;
        const uint32_t nhi = __builtin_amdgcn_alignbit(hi, lo, 31) & himask;
        return ((uint64_t)nhi << 32) | nlo;
    }
    // fastrightshift1 (cyclichash.h:46-52)
    __device__ __forceinline__ uint64_t rotr1(uint64_t x) const
    {
        const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
        const uint32_t t = (lo & 1u) << sh;
        const uint32_t nlo = __builtin_amdgcn_alignbit(hi, lo, 1) | (t & mlo);
        const uint32_t nhi = (hi >> 1) | (t & mhi);
        return ((uint64_t)nhi << 32) | nlo;
    }
};

// ---- the slice-index permutation and the level-1 split of a Bloom address in one go (tpc_bins.h:PtPerm::fwd followed by
// "bin = a' >> shift, remainder = a' & (2^shift - 1)"): bin and remainder of the PERMUTED address, 32-bit arithmetic.
// Needs F = L - slice_bits <= 24 (the multiply is a 24-bit one) and L - log_nb <= 31.
struct LeanSplit {
    uint32_t slice_bits, smask;  // 2^slice_bits - 1
    uint32_t mult24;             // low 24 bits of the odd multiplier
    uint32_t low_bits, lowmask;  // F - log_nb bits of the permuted slice index stay in the remainder
    uint32_t nbmask;             // 2^log_nb - 1
    uint32_t log_nb;
    __host__ __device__ __forceinline__ void set(const PtPerm &perm, int log_nb)
    {
        slice_bits = (uint32_t)perm.slice_bits;
        smask = (1u << perm.slice_bits) - 1u;
        mult24 = perm.mult & 0xFFFFFFu;
        low_bits = (uint32_t)(perm.F - log_nb);
        lowmask = (1u << low_bits) - 1u;
        nbmask = (1u << log_nb) - 1u;
        this->log_nb = (uint32_t)log_nb;
    }
    __device__ __forceinline__ void split(uint64_t a, uint32_t &bin, uint32_t &rem) const
    {
        const uint32_t top = (uint32_t)(a >> slice_bits);       // slice index, < 2^F <= 2^24
        const uint32_t prod = __umul24(top, mult24);            // its low F bits are those of top * mult
        bin = (prod >> low_bits) & nbmask;
        rem = ((prod & lowmask) << slice_bits) | ((uint32_t)a & smask);
    }
};

// ---- round 4: the same rotations and the split on explicit 32-bit halves, specialised on L > 32 (LHI) -------------------------
// The ISA of round 3's kernels showed what the compiler made of LeanRot: `x >> top_shift` and even `lo << 1` had become 64-bit
// shifts again (v_lshlrev_b64 / v_lshrrev_b64), the "wrapped bit lands in lo or hi" selects of rotr1 two v_cndmask -- 6 and 8
// instructions.  With the value kept as two separate 32-bit registers throughout and the case L > 32 / L <= 32 known at compile
// time, rotl1 is 4 instructions (v_lshrrev, v_lshl_or, v_alignbit, v_and) and rotr1 4 (v_alignbit, v_and, v_lshrrev, v_lshl_or);
// for L <= 32 the high halves are identically zero and are never computed (3 + 3, and every XOR is one instruction instead of two).
template <bool LHI>
struct LeanV {  // an L-bit value; hi == 0 (and dead) when !LHI
    uint32_t lo, hi;
};
template <bool LHI> __device__ __forceinline__ LeanV<LHI> lv_make(uint32_t lo, uint32_t hi) { LeanV<LHI> r; r.lo = lo; r.hi = LHI ? hi : 0u; return r; }
template <bool LHI> __device__ __forceinline__ LeanV<LHI> lv_xor(LeanV<LHI> a, uint32_t lo, uint32_t hi) { return lv_make<LHI>(a.lo ^ lo, a.hi ^ hi); }
template <bool LHI> __device__ __forceinline__ LeanV<LHI> lv_xor(LeanV<LHI> a, LeanV<LHI> b) { return lv_make<LHI>(a.lo ^ b.lo, a.hi ^ b.hi); }
template <bool LHI> __device__ __forceinline__ uint64_t lv_u64(LeanV<LHI> a) { return LHI ? (((uint64_t)a.hi << 32) | a.lo) : (uint64_t)a.lo; }
template <bool LHI> __device__ __forceinline__ bool lv_lt(LeanV<LHI> a, LeanV<LHI> b) { return LHI ? lv_u64(a) < lv_u64(b) : a.lo < b.lo; }
template <bool LHI> __device__ __forceinline__ bool lv_eq(LeanV<LHI> a, LeanV<LHI> b) { return LHI ? lv_u64(a) == lv_u64(b) : a.lo == b.lo; }
template <bool LHI> __device__ __forceinline__ LeanV<LHI> lv_sel(bool c, LeanV<LHI> a, LeanV<LHI> b) { return lv_make<LHI>(c ? a.lo : b.lo, c ? a.hi : b.hi); }
template <bool LHI> __device__ __forceinline__ LeanV<LHI> lv_min(LeanV<LHI> a, LeanV<LHI> b) { return lv_sel<LHI>(lv_lt<LHI>(b, a), b, a); }
template <bool LHI> __device__ __forceinline__ LeanV<LHI> lv_from64(uint64_t x) { return lv_make<LHI>((uint32_t)x, (uint32_t)(x >> 32)); }

template <bool LHI>
struct LeanRotH {
    uint32_t lomask, himask;  // 2^L - 1 as two halves
    uint32_t sh;              // LHI: L - 33 (the top bit's place in hi); else L - 1
    __device__ __forceinline__ void set(int L)
    {
        const uint64_t lm = (1ull << L) - 1ull;
        lomask = (uint32_t)lm; himask = (uint32_t)(lm >> 32);
        sh = (uint32_t)(LHI ? L - 33 : L - 1);
    }
    // fastleftshift1 (cyclichash.h:42-44); x < 2^L
    __device__ __forceinline__ LeanV<LHI> rotl1(LeanV<LHI> x) const
    {
        LeanV<LHI> r;
        if constexpr (LHI) {
            const uint32_t bit = x.hi >> sh;  // 0 or 1
            r.lo = (x.lo << 1) | bit;
            r.hi = __builtin_amdgcn_alignbit(x.hi, x.lo, 31) & himask;
        } else {
            r.lo = ((x.lo << 1) | (x.lo >> sh)) & lomask;
            r.hi = 0u;
        }
        return r;
    }
    // fastrightshift1 (cyclichash.h:46-52)
    __device__ __forceinline__ LeanV<LHI> rotr1(LeanV<LHI> x) const
    {
        LeanV<LHI> r;
        if constexpr (LHI) {
            r.lo = __builtin_amdgcn_alignbit(x.hi, x.lo, 1);
            r.hi = ((x.lo & 1u) << sh) | (x.hi >> 1);
        } else {
            r.lo = ((x.lo & 1u) << sh) | (x.lo >> 1);
            r.hi = 0u;
        }
        return r;
    }
};

// LeanSplit::split on halves: slice index by one v_alignbit (slice_bits in 6..20), bin by one bit-field extract.
template <bool LHI>
__device__ __forceinline__ void lean_split_h(const LeanSplit &S, LeanV<LHI> a, uint32_t &bin, uint32_t &rem)
{
    const uint32_t top = LHI ? __builtin_amdgcn_alignbit(a.hi, a.lo, S.slice_bits) : a.lo >> S.slice_bits;  // < 2^F <= 2^24
    const uint32_t prod = __umul24(top, S.mult24);
    bin = __builtin_amdgcn_ubfe(prod, S.low_bits, S.log_nb);
    rem = ((prod & S.lowmask) << S.slice_bits) | (a.lo & S.smask);
}

// 16 two-bit characters starting at position p (relative to the first staged word) as one 32-bit word
__device__ __forceinline__ uint32_t lean_chars16(const uint64_t *sb, uint32_t p)
{
    const uint32_t w = p >> 5, o = 2u * (p & 31u);
    const uint64_t a = sb[w], b = sb[w + 1];
    return (uint32_t)((a >> o) | ((b << 1) << (63u - o)));
}
// 32 N-mask bits starting at position p
__device__ __forceinline__ uint32_t lean_nbits32(const uint32_t *sn, uint32_t p)
{
    const uint32_t w = p >> 5, o = p & 31u;
    return (sn[w] >> o) | ((sn[w + 1] << 1) << (31u - o));
}
// one character code (0..3, 4 = N) at position p
__device__ __forceinline__ uint32_t lean_char(const uint64_t *sb, const uint32_t *sn, uint32_t p)
{
    const uint32_t w = p >> 5, o = p & 31u;
    const uint32_t isn = (sn[w] >> o) & 1u;
    const uint32_t code = (uint32_t)(sb[w] >> (2u * o)) & 3u;
    return isn ? 4u : code;
}
