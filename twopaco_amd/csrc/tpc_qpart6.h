// tpc_qpart6.h -- the lookup side of the partitioned first-pass query (CandidateCheckingWorker, reference
// vertexenumerator.h:586-704) on 6-byte level-2 entries (round 5: the "entry diet").  Included by tpc_qpartition.hip inside its
// anonymous namespace (SurvStage and the verification kernels are shared with the 8-byte path, which stays for sharded filters,
// three-level geometries and levels of fewer than 16 bins).
//
// Every probe of the query crosses HBM four times (tpc_qpartition.hip: A writes it, B reads and rewrites it, C reads it), and B and C
// stream at what HBM gives their access pattern, so they get faster only with fewer bytes per probe.  Level 1 keeps its 8-byte
// entries (k_q_hash2 is bound by its LDS pipe, not by its stores; a 6-byte level 1 was built first and cost that kernel 0.6 ms for
// the extra ring arithmetic: profiles/r05a_*).  Level 2 (k_q_split<.., P6>) writes blocked lines of 20 x 48-bit entries
// (tpc_binsp.h:PFmt6)
//     { slice offset S | edge 3 | low PB2 = 44 - S bits of the position | parity of the position's group g = position >> PB2 }
// and beside every region where the ZONES of its entries start and end: before the first round of every level-1 region that may
// reach a group g not seen so far the rings are emptied (every bin's last line leaves partly filled) and zone g starts at the next
// line (bnd[g]; the zone before it ends at vend[g - 1], what follows up to the line's end is garbage).  An entry at index i lies in
// zone z = #{g >= 1 : bnd[g] <= i}; the zone holds groups z - 1 and z only, so the parity bit decides (Q6Res).  Only the ~3 % of the
// entries that survive the first probe ever need it, and they are resolved when the survivors' staging area is flushed.
// Survivor ids, the survivor sub-lists, the overflow list and the verification are those of the 8-byte path: the mask is the same
// bit for bit (tests/test_gpu_parity.py).
#pragma once

constexpr int Q6_MAX_GROUPS = 64;
constexpr size_t QL6_LDS = QL_LDS + 2 * Q6_MAX_GROUPS * 4;

// A first-probe survivor is staged RAW -- {edge 3 | low position bits PB2 | group parity 1 | index in its level-2 region} -- and becomes
// a survivor id (edge | batch-relative position << 3) when the staging area is flushed (SurvStage), or ~0 if the index lies in the
// garbage behind a zone's last entry.
struct Q6Res {
    const uint32_t *s_bnd;  // LDS: the region's zone starts [n_groups], then its zone ends [n_groups]
    uint32_t n_groups, PB2;
    __device__ __forceinline__ uint64_t raw(uint64_t v, uint32_t S, uint32_t idx) const
    {
        return ((v >> S) & ((1ull << (3u + PB2)) - 1ull)) | ((v >> 47) << (3u + PB2)) | ((uint64_t)idx << (4u + PB2));
    }
    __device__ __forceinline__ uint64_t operator()(uint64_t r) const
    {
        const uint32_t idx = (uint32_t)(r >> (4u + PB2));
        uint32_t z = 0;
        for (uint32_t g = 1; g < n_groups; g++) z += idx >= s_bnd[g] ? 1u : 0u;  // (<= 63 boundaries)
        if (idx >= s_bnd[n_groups + z]) return ~0ull;
        const uint64_t g = (z & 1u) == ((uint32_t)(r >> (3u + PB2)) & 1u) ? z : z - 1u;
        return (r & 7ull) | (((g << PB2) | ((r >> 3) & ((1ull << PB2) - 1ull))) << 3);
    }
};

// One slice at a time per workgroup (k_q_lookup on blocked lines): the slice of the filter in LDS, every entry tests its bit there.
__global__ void __launch_bounds__(PT_APPLY_THREADS)
k_q_lookup6(int slice_bits, int log_nb2, uint32_t wpb, const unsigned char *__restrict__ buf2, const uint32_t *__restrict__ cnt2,
            const uint64_t *__restrict__ off2, const uint32_t *__restrict__ bnd, uint32_t n_groups, uint32_t pb2, const uint32_t *__restrict__ filter, uint64_t *surv,
            unsigned long long *surv_cur, uint64_t surv_cap, PtPerm perm, int group, uint32_t n_slices)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t words = 1u << (slice_bits - 5);
    uint32_t *slice = reinterpret_cast<uint32_t *>(smem);
    SurvStage st;
    uint32_t *s_bnd = reinterpret_cast<uint32_t *>(st.carve(reinterpret_cast<unsigned char *>(slice + ((words + 3u) & ~3u)), slice_bits));
    st.group = group;
    const uint32_t nb2 = 1u << log_nb2;
    // a long-lived workgroup takes every gridDim.x-th slice (tpc_internal.h:tpc_slice_grid)
    for (uint32_t sl = blockIdx.x; sl < n_slices; sl += gridDim.x) {
        const uint32_t b1 = sl >> log_nb2, b2 = sl & (nb2 - 1);
        const uint32_t *src_slice = filter + (uint64_t)perm.slice_of(sl) * words;
        if ((words & 3u) == 0) for (uint32_t i = threadIdx.x; i < words / 4; i += PT_APPLY_THREADS) reinterpret_cast<uint4 *>(slice)[i] = reinterpret_cast<const uint4 *>(src_slice)[i];
        else for (uint32_t i = threadIdx.x; i < words; i += PT_APPLY_THREADS) slice[i] = src_slice[i];
        if (threadIdx.x == 0) st.ctl[0] = 0;
        const uint32_t slice_mask = (1u << slice_bits) - 1u, S = (uint32_t)slice_bits;
        st.list = sl % QS_LISTS;
        st.my_list = surv + (uint64_t)st.list * surv_cap; st.surv0 = surv;
        st.surv_cur = surv_cur; st.surv_cap = surv_cap;
        const Q6Res res{s_bnd, n_groups, pb2};
        for (uint32_t j = 0; j < wpb; j++) {
            const uint64_t r = ((uint64_t)b1 * wpb + j) * nb2 + b2;
            if (j) st.flush(res);  // what is staged belongs to the previous region's boundaries
            __syncthreads();       // (first time round: the slice is loaded)
            if (threadIdx.x < 2u * n_groups) s_bnd[threadIdx.x] = bnd[r * 2u * n_groups + threadIdx.x];
            __syncthreads();
            const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt2[r]);
            auto probe = [&](uint64_t v, uint32_t idx) {
                const uint32_t a = (uint32_t)v & slice_mask;
                if ((slice[a >> 5] >> (a & 31u)) & 1u) st.push(res.raw(v, S, idx), a, res);
            };
            PlStream<PFmt6, PT_APPLY_THREADS, 1> q;
            q.begin(buf2 + off2[r] * PT_LINE, n);
            if (n > QL_LONG_REGION) q.finish_with(probe, [&]() { st.maybe_flush(res); });  // (uniform; see k_q_lookup)
            else q.finish(probe);
            st.maybe_flush(res);
        }
        st.flush(res);
        __syncthreads();
    }
}

// Fused k_part_apply + k_q_lookup6 (deferred apply, tpc_partition.hip): the workgroup of a slice ORs the insert's level-2 entries
// (blocked lines of 40 x 24 bits: PFmt3; or 32-bit entries when I3 is false) into the zeroed LDS slice, writes the slice out and
// tests the query's entries against the slice it still holds.
// LISTS (the combined multi-GPU exchange): the slices are built from imported set-bit lists ALONE (tpc_lists.h; no regions, no overflow
// entries of this rank's own: they are in the lists), and everything whose address depends on the workgroup's index only -- the lists'
// directory entries and first loads, the zone table, the first query region's count, offset and first lines -- is asked for before the
// slice is zeroed: with a rank's share of the probes a workgroup lives ~12 us, most of it waiting for one of these after the other.
template <bool I3, bool LISTS = false>
__global__ void __launch_bounds__(PT_APPLY_THREADS)
k_apply_lookup6(int slice_bits, int log_nb2, uint32_t iwpb, const unsigned char *__restrict__ ibuf2, const uint32_t *__restrict__ icnt2, uint64_t icap2_lines, int fresh,
                const uint64_t *__restrict__ iovf, const uint64_t *__restrict__ iovf_off, uint32_t qwpb, const unsigned char *__restrict__ qbuf2,
                const uint32_t *__restrict__ qcnt2, const uint64_t *__restrict__ qoff2, const uint32_t *__restrict__ bnd, uint32_t n_groups, uint32_t pb2,
                uint32_t *__restrict__ filter, uint64_t *surv, unsigned long long *surv_cur, uint64_t surv_cap, PtPerm perm, int group, TpcListSrc ls, uint32_t n_slices)
{   // ls (ls.n_src > 0: the combined multi-GPU exchange, tpc_lists.h): set-bit lists of the slice, from this and the other ranks' inserts
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t words = 1u << (slice_bits - 5);
    uint32_t *slice = reinterpret_cast<uint32_t *>(smem);
    SurvStage st;
    uint32_t *s_bnd = reinterpret_cast<uint32_t *>(st.carve(reinterpret_cast<unsigned char *>(slice + ((words + 3u) & ~3u)), slice_bits));
    st.group = group;
    const uint32_t nb2 = 1u << log_nb2;
    // a long-lived workgroup takes every gridDim.x-th slice (tpc_internal.h:tpc_slice_grid).  Not the LISTS and I3 instantiations: the
    // loop's live values cost them 40 / 19 spilled registers of the 128 a 1024-thread workgroup has, and the lists' ~12 us per slice are
    // round trips that only separate workgroups overlap (4.6 against 2.9 ms per rank at eight ranks): launched one workgroup per slice.
    constexpr bool LONG_LIVED = !LISTS && !I3;
    for (uint32_t sl = blockIdx.x; sl < n_slices; sl += gridDim.x) {
        const uint32_t b1 = sl >> log_nb2, b2 = sl & (nb2 - 1);
        uint32_t *out = filter + (uint64_t)perm.slice_of(sl) * words;
        const bool wide = (words & 3u) == 0;
        TpcListReader<PT_APPLY_THREADS> lists;
        PlStream<PFmt6, PT_APPLY_THREADS, 1> q0;
        const uint64_t r0 = ((uint64_t)b1 * qwpb) * nb2 + b2;
        if constexpr (LISTS) {
            lists.begin(ls, b1, b2, log_nb2, sl, slice_bits);
            if (threadIdx.x < 2u * n_groups) s_bnd[threadIdx.x] = bnd[r0 * 2u * n_groups + threadIdx.x];
            q0.begin(qbuf2 + qoff2[r0] * PT_LINE, (uint32_t)__builtin_amdgcn_readfirstlane((int)qcnt2[r0]));
        }
        // ---- apply
        if (fresh) {
            if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += PT_APPLY_THREADS) reinterpret_cast<uint4 *>(slice)[i] = make_uint4(0, 0, 0, 0);
            else for (uint32_t i = threadIdx.x; i < words; i += PT_APPLY_THREADS) slice[i] = 0;
        } else {
            if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += PT_APPLY_THREADS) reinterpret_cast<uint4 *>(slice)[i] = reinterpret_cast<const uint4 *>(out)[i];
            else for (uint32_t i = threadIdx.x; i < words; i += PT_APPLY_THREADS) slice[i] = out[i];
        }
        if (threadIdx.x == 0) st.ctl[0] = 0;
        __syncthreads();
        if constexpr (LISTS) lists.finish(ls, slice);
        else
        for (uint32_t j = 0; j < iwpb; j++) {
            const uint64_t r = ((uint64_t)b1 * iwpb + j) * nb2 + b2;
            const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)icnt2[r]);
            if constexpr (I3) {
                PlStream<PFmt3, PT_APPLY_THREADS, 1> is;
                is.begin(ibuf2 + r * icap2_lines * PT_LINE, n);
                is.finish([slice](uint32_t v, uint32_t) { atomicOr(&slice[v >> 5], 1u << (v & 31u)); });
            } else {
                pt_stream_region<PT_APPLY_THREADS, 2>(reinterpret_cast<const uint32_t *>(ibuf2) + r * icap2_lines * 32u, n, [slice](uint32_t v) { atomicOr(&slice[v >> 5], 1u << (v & 31u)); });
            }
        }
        // the insert's overflow entries (permuted addresses that found a ring or region full), grouped by slice beforehand
        if (!LISTS && iovf_off) {
            const uint64_t o0 = iovf_off[sl], o1 = iovf_off[sl + 1];
            for (uint64_t i = o0 + threadIdx.x; i < o1; i += PT_APPLY_THREADS) {
                const uint64_t a = iovf[i];
                atomicOr(&slice[((uint32_t)a & ((1u << slice_bits) - 1u)) >> 5], 1u << ((uint32_t)a & 31u));
            }
        }
        if (!LISTS && threadIdx.x < 2u * n_groups) s_bnd[threadIdx.x] = bnd[r0 * 2u * n_groups + threadIdx.x];
        __syncthreads();
        // the first query region's loads go out before the slice's stores: the 128 KB write-out then drains under them
        if constexpr (!LISTS) q0.begin(qbuf2 + qoff2[r0] * PT_LINE, (uint32_t)__builtin_amdgcn_readfirstlane((int)qcnt2[r0]));
        if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += PT_APPLY_THREADS) reinterpret_cast<uint4 *>(out)[i] = reinterpret_cast<const uint4 *>(slice)[i];
        else for (uint32_t i = threadIdx.x; i < words; i += PT_APPLY_THREADS) out[i] = slice[i];
        // ---- lookup against the slice still in LDS
        const uint32_t slice_mask = (1u << slice_bits) - 1u, S = (uint32_t)slice_bits;
        st.list = sl % QS_LISTS;
        st.my_list = surv + (uint64_t)st.list * surv_cap; st.surv0 = surv;
        st.surv_cur = surv_cur; st.surv_cap = surv_cap;
        const Q6Res res{s_bnd, n_groups, pb2};
        auto probe = [&](uint64_t v, uint32_t idx) {
            const uint32_t a = (uint32_t)v & slice_mask;
            if ((slice[a >> 5] >> (a & 31u)) & 1u) st.push(res.raw(v, S, idx), a, res);
        };
        for (uint32_t j = 0; j < qwpb; j++) {
            const uint64_t r = r0 + (uint64_t)j * nb2;
            if (j == 0) {
                if (q0.n > QL_LONG_REGION) q0.finish_with(probe, [&]() { st.maybe_flush(res); });  // (uniform; see k_q_lookup)
                else q0.finish(probe);
            } else {
                st.flush(res);  // what is staged belongs to the previous region's boundaries
                __syncthreads();
                if (threadIdx.x < 2u * n_groups) s_bnd[threadIdx.x] = bnd[r * 2u * n_groups + threadIdx.x];
                __syncthreads();
                const uint32_t nq = (uint32_t)__builtin_amdgcn_readfirstlane((int)qcnt2[r]);
                PlStream<PFmt6, PT_APPLY_THREADS, 1> q;
                q.begin(qbuf2 + qoff2[r] * PT_LINE, nq);
                if (nq > QL_LONG_REGION) q.finish_with(probe, [&]() { st.maybe_flush(res); });
                else q.finish(probe);
            }
            st.maybe_flush(res);
        }
        st.flush(res);
        if constexpr (!LONG_LIVED) break;
        __syncthreads();
    }
}
