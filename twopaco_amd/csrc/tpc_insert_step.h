// tpc_insert_step.h -- the per-position body of the first-pass insert, shared by the direct
// (atomicOr) kernel and the partitioned (LDS write-combining) kernel so that the parity-critical
// logic exists once.  Restates FilterFillerWorker::operator(), reference
// src/graphconstructor/vertexenumerator.h:1035-1083.
#pragma once
#include "tpc_device.h"

// Rolling state of one thread's run of consecutive vertex positions.
template <int Q>
struct TpcRoll {
    TpcVHash<Q> v;
    int ncnt;     // N characters inside the current window (k - definiteCount, VE.h:1033)
    int c_prev;   // character before the window
    int c_first;  // first character of the window
};

template <int Q>
__device__ __forceinline__ void tpc_roll_init(TpcRoll<Q> &r, const TpcHashParams &P, const uint64_t *s_h,
                                              const uint64_t *sb, const uint32_t *sn, uint64_t g0, uint64_t wbase)
{
    tpc_vhash_init<Q>(r.v, P, s_h, sb, sn, g0, wbase);
    r.ncnt = 0;
    for (int t = 0; t < P.k; t++) r.ncnt += tpc_tile_char(sb, sn, g0 + t, wbase) == TPC_CODE_N;
    r.c_prev = tpc_tile_char(sb, sn, g0 - 1, wbase);
    r.c_first = tpc_tile_char(sb, sn, g0, wbase);
}

template <int Q, class Emit>
__device__ __forceinline__ void tpc_emit_edge(Emit &emit, const uint64_t (&p)[Q], const uint64_t (&n)[Q])
{
    const bool neg = tpc_pick_neg<Q>(p, n);
    uint64_t a[Q];
#pragma unroll
    for (int i = 0; i < Q; i++) a[i] = neg ? n[i] : p[i];
    emit.template edge<Q>(a);  // the q Bloom addresses of one edge
}

// Rare branch of the step below: the dummy edges beside an N (VE.h:1048-1058), computed from the window's hashes before they roll.
template <int Q, class Emit>
__device__ __forceinline__ void tpc_emit_dummies(const TpcVHash<Q> &v, const TpcHashParams &P, const uint64_t *s_h, const uint64_t *s_hk,
                                              bool out_side, bool in_side, Emit &emit)
{
    uint64_t p[Q], n[Q];
    if (out_side) {  // dummy out-edges v+'A', v+'T' (VE.h:1048-1052)
#pragma unroll
        for (int i = 0; i < Q; i++) { p[i] = tpc_rotl1(v.pos[i], P.L, P.lmask) ^ s_h[i * 5 + 0]; n[i] = v.neg[i] ^ s_hk[i * 5 + 3]; }
        tpc_emit_edge<Q>(emit, p, n);
#pragma unroll
        for (int i = 0; i < Q; i++) { p[i] = tpc_rotl1(v.pos[i], P.L, P.lmask) ^ s_h[i * 5 + 3]; n[i] = v.neg[i] ^ s_hk[i * 5 + 0]; }
        tpc_emit_edge<Q>(emit, p, n);
    }
    if (in_side) {  // dummy in-edges 'A'+v, 'T'+v (VE.h:1054-1058)
#pragma unroll
        for (int i = 0; i < Q; i++) { p[i] = s_hk[i * 5 + 0] ^ v.pos[i]; n[i] = tpc_rotl1(v.neg[i], P.L, P.lmask) ^ s_h[i * 5 + 3]; }
        tpc_emit_edge<Q>(emit, p, n);
#pragma unroll
        for (int i = 0; i < Q; i++) { p[i] = s_hk[i * 5 + 3] ^ v.pos[i]; n[i] = tpc_rotl1(v.neg[i], P.L, P.lmask) ^ s_h[i * 5 + 0]; }
        tpc_emit_edge<Q>(emit, p, n);
    }
}

// One position: emits the Bloom addresses of the window at g (if N-free and inside the round's
// range), then rolls to g+1.  Returns true when the window was an N-free vertex.
// Register diet: hash function 0 is evaluated first -- it decides whether the round takes the edge (GATED) and, unless
// its two strand values tie (probability 2^-L), the canonical strand -- after which every other function is rolled and
// turned into its address on the spot, so only the Q addresses stay live (the all-at-once form held 5 x Q temporaries
// and spilled 37 registers at Q = 5).
template <int Q, bool GATED, class Emit>
__device__ __forceinline__ bool tpc_insert_step(TpcRoll<Q> &r, const TpcHashParams &P, const uint64_t *s_h, const uint64_t *s_hk,
                                                const uint64_t *sb, const uint32_t *sn, uint64_t g, uint64_t wbase,
                                                uint64_t lo, uint64_t hi, Emit &emit)
{
    TpcVHash<Q> &v = r.v;
    const int c_next = tpc_tile_char(sb, sn, g + P.k, wbase);
    const int c_first_nx = tpc_tile_char(sb, sn, g + 1, wbase);
    const int c_first = r.c_first;
    const int rc_next = tpc_rc(c_next), rc_first = tpc_rc(c_first);
    // hash_extend / hash_prepend of the outgoing edge (cyclichash.h:112-121) are the
    // intermediates of update / reverse_update (cyclichash.h:86-102).
    const uint64_t ep0 = tpc_rotl1(v.pos[0], P.L, P.lmask) ^ s_h[c_next];
    const uint64_t en0 = v.neg[0] ^ s_hk[rc_next];
    const uint64_t np0 = ep0 ^ s_hk[c_first];
    const uint64_t nn0 = tpc_rotr1(en0 ^ s_h[rc_first], P.L);
    const bool vertex = r.ncnt == 0;
    bool go = vertex;
    if (GATED && go) {  // VE.h:1063-1073
        const uint64_t first = tpc_min(v.pos[0], v.neg[0]);
        const uint64_t second = tpc_min(np0, nn0);
        go = (first >= lo && first <= hi) || (second >= lo && second <= hi);
    }
    const bool main_edge = go && c_next != TPC_CODE_N;
    if (go && (c_next == TPC_CODE_N || r.c_prev == TPC_CODE_N)) tpc_emit_dummies<Q>(v, P, s_h, s_hk, c_next == TPC_CODE_N, r.c_prev == TPC_CODE_N, emit);
    // canonical strand of the out-edge (DetermineStrandExtend, vertexrollinghash.h:170-184)
    bool ng = en0 < ep0;
    if (main_edge && ep0 == en0) {
        ng = false;
        for (int i = 1; i < Q; i++) {
            const uint64_t p = tpc_rotl1(v.pos[i], P.L, P.lmask) ^ s_h[i * 5 + c_next];
            const uint64_t n = v.neg[i] ^ s_hk[i * 5 + rc_next];
            if (p != n) { ng = n < p; break; }
        }
    }
    uint64_t a[Q];
    a[0] = ng ? en0 : ep0;
    v.pos[0] = np0;
    v.neg[0] = nn0;
#pragma unroll
    for (int i = 1; i < Q; i++) {
        const uint64_t ep = tpc_rotl1(v.pos[i], P.L, P.lmask) ^ s_h[i * 5 + c_next];
        const uint64_t en = v.neg[i] ^ s_hk[i * 5 + rc_next];
        a[i] = ng ? en : ep;
        v.pos[i] = ep ^ s_hk[i * 5 + c_first];
        v.neg[i] = tpc_rotr1(en ^ s_h[i * 5 + rc_first], P.L);
    }
    if (main_edge) emit.template edge<Q>(a);
    r.ncnt += (c_next == TPC_CODE_N) - (c_first == TPC_CODE_N);
    r.c_prev = c_first;
    r.c_first = c_first_nx;
    return vertex;
}
