// tpc_combine.hip -- "combine before routing": the multi-GPU first pass with the Bloom filter REPLICATED through sparse set-bit lists.
//
// The reference's threads share one ConcurrentBitVector for free (fetch_or on shared memory, reference
// graphconstructor/concurrentbitvector.cpp:31-45; MergeOr :115-122).  Routing every hash hit of every rank to the rank that owns its
// filter slice (tpc_shard_*: 4 bytes per insert address, 8 per query probe) costs 21 GB per rank and step on the 62-genome workload --
// more wire time than the whole pass takes on one GPU.  But an insert only matters the first time a bit is set, and many-genome inputs
// set the same bits over and over: a rank that first ORs ITS inserts into the LDS slices (the write-combining passes it runs anyway:
// k_part_hash2 -> k_part_split) needs to tell the others only which bits of a slice came out set.  This file is that step:
//
//   k_slice_combine  one workgroup per filter slice: the zeroed slice in LDS, OR of every source -- the rank's own level-2 insert
//                    regions and overflow entries, and / or set-bit lists received from other ranks -- then the slice's set bits
//                    leave as ascending 16-bit offsets, one list per 2^16-bit window of the slice (tpc_lists.h), in the block of the
//                    rank that owns the slice (one atomic per workgroup claims the space; a directory entry per window says where
//                    its list went), and / or the dense slice is written to the filter.
//
// A list block is what travels (tpc_combine_export -> reduce-scatter by owner -> tpc_combine_merge -> all-gather -> tpc_combine_import,
// or a plain all-gather of the exports below four ranks); the query then runs entirely on the rank that hashed it: its fused
// lookup (tpc_qpart6.h:k_apply_lookup6) builds every slice from the imported lists, so no probe and no survivor ever crosses a link.
#include "tpc_internal.h"
#include "tpc_lists.h"

namespace {

constexpr int CB_THREADS = PT_APPLY_THREADS;
constexpr int CB_MAX_WIN = 1 << (20 - TPC_LIST_WINDOW_BITS);

__global__ void __launch_bounds__(CB_THREADS)
k_slice_combine(int slice_bits, int log_nb2, uint32_t iwpb, const uint32_t *__restrict__ ibuf2, const uint32_t *__restrict__ icnt2, uint64_t icap2,
                const uint64_t *__restrict__ iovf, const uint64_t *__restrict__ iovf_off, TpcListSrc ls, uint32_t *__restrict__ filter, int fresh,
                uint16_t *__restrict__ out_payload, uint64_t out_cap, unsigned long long *out_cur, uint64_t *__restrict__ out_dir, uint32_t n_dest, PtPerm perm, PtShard grid)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t words = 1u << (slice_bits - 5);
    uint32_t *slice = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_w = slice + ((words + 3u) & ~3u);   // [CB_THREADS / 64] scan scratch
    uint32_t *s_win = s_w + CB_THREADS / 64;        // [CB_MAX_WIN + 1] entries before every window, then [CB_MAX_WIN] its first unit
    uint32_t *s_ctl = s_win + 2 * CB_MAX_WIN + 1;   // [4]
    const uint32_t nb2 = 1u << log_nb2;
    // the permuted slice this workgroup builds: every slice (grid.world == 1), or the blockIdx-th slice of the level-1 buckets rank grid.rank owns
    uint32_t b1 = blockIdx.x >> log_nb2;
    const uint32_t b2 = blockIdx.x & (nb2 - 1u);
    if (grid.world > 1) b1 = b1 * grid.world + grid.rank;
    const uint32_t sp = (b1 << log_nb2) | b2;
    uint32_t *out = filter ? filter + (uint64_t)perm.slice_of(sp) * words : nullptr;
    const bool wide = (words & 3u) == 0;
    if (fresh || !out) {
        if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += CB_THREADS) reinterpret_cast<uint4 *>(slice)[i] = make_uint4(0, 0, 0, 0);
        else for (uint32_t i = threadIdx.x; i < words; i += CB_THREADS) slice[i] = 0;
    } else {
        if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += CB_THREADS) reinterpret_cast<uint4 *>(slice)[i] = reinterpret_cast<const uint4 *>(out)[i];
        else for (uint32_t i = threadIdx.x; i < words; i += CB_THREADS) slice[i] = out[i];
    }
    __syncthreads();
    auto set = [slice](uint32_t v) { atomicOr(&slice[v >> 5], 1u << (v & 31u)); };
    // ---- this rank's own level-2 insert regions of the slice (32-bit slice offsets) and its overflow entries grouped by slice
    for (uint32_t j = 0; j < iwpb; j++) {
        const uint64_t r = ((uint64_t)b1 * iwpb + j) * nb2 + b2;
        const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)icnt2[r]);
        pt_stream_region<CB_THREADS, 2>(ibuf2 + r * icap2, n, set);
    }
    if (iovf_off) {
        const uint64_t o0 = iovf_off[sp], o1 = iovf_off[sp + 1];
        const uint32_t smask = (1u << slice_bits) - 1u;
        for (uint64_t i = o0 + threadIdx.x; i < o1; i += CB_THREADS) set((uint32_t)iovf[i] & smask);
    }
    // ---- set-bit lists (this kernel's own output format) received from other ranks
    if (ls.n_src) {
        tpc_lists_apply<CB_THREADS>(ls, b1, b2, log_nb2, blockIdx.x, slice, slice_bits);
    }
    __syncthreads();
    if (out) {
        if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += CB_THREADS) reinterpret_cast<uint4 *>(out)[i] = reinterpret_cast<const uint4 *>(slice)[i];
        else for (uint32_t i = threadIdx.x; i < words; i += CB_THREADS) out[i] = slice[i];
    }
    if (!out_payload) return;
    // ---- the slice's set bits: thread t lists the bits of its run of words behind those of the threads before it; the threads of a
    // window (a contiguous range of threads) write that window's list, which starts on a 16-byte unit
    const uint32_t n_win = tpc_list_windows(slice_bits), tpw = (uint32_t)CB_THREADS / n_win;
    const uint32_t per = (words + CB_THREADS - 1u) / CB_THREADS;
    const uint32_t w0 = min(words, threadIdx.x * per), w1 = min(words, w0 + per);
    uint32_t cnt = 0;
    for (uint32_t w = w0; w < w1; w++) cnt += (uint32_t)__popc(slice[w]);
    uint32_t total;
    const uint32_t off = pt_block_excl_scan<CB_THREADS>(cnt, s_w, total);
    const uint32_t win = threadIdx.x / tpw;
    if (threadIdx.x % tpw == 0) s_win[win] = off;  // entries before the window
    if (threadIdx.x == 0) s_win[n_win] = total;
    __syncthreads();
    // destination block and the slice's index there: [local bucket of the destination][b2]
    const uint32_t dest = b1 & (n_dest - 1u);
    const uint64_t key = grid.world > 1 ? (uint64_t)blockIdx.x : ((uint64_t)(b1 / n_dest) << log_nb2) | b2;
    if (threadIdx.x == 0) {
        uint32_t units = 0;
        for (uint32_t v = 0; v < n_win; v++) { s_win[CB_MAX_WIN + 1 + v] = units; units += (s_win[v + 1] - s_win[v] + 7u) >> 3; }
        const uint64_t base = units ? (uint64_t)atomicAdd(&out_cur[dest], (unsigned long long)units) : 0ull;
        const bool ok = base + units <= out_cap;
        if (!ok) out_cur[n_dest] = 1ull;  // the block is too small (the host sizes it from the entry counts: tpc_combine_info)
        s_ctl[0] = (uint32_t)base; s_ctl[1] = (uint32_t)(base >> 32); s_ctl[2] = ok ? 1u : 0u;
    }
    __syncthreads();
    const bool ok = s_ctl[2] != 0;
    const uint64_t base = (uint64_t)s_ctl[0] | ((uint64_t)s_ctl[1] << 32);
    const uint64_t slices_per_dest = ((uint64_t)1 << (perm.F)) / (grid.world > 1 ? grid.world : n_dest);
    uint64_t *dir = out_dir + ((uint64_t)(grid.world > 1 ? 0u : dest) * slices_per_dest + key) * n_win;
    if (threadIdx.x < n_win) {
        const uint32_t n = s_win[threadIdx.x + 1] - s_win[threadIdx.x];
        dir[threadIdx.x] = ok ? ((base + s_win[CB_MAX_WIN + 1 + threadIdx.x]) << 24) | (uint64_t)n : 0ull;
    }
    if (!ok || cnt == 0) return;
    uint16_t *dst = out_payload + (((uint64_t)dest * out_cap + base + s_win[CB_MAX_WIN + 1 + win]) << 3) + (off - s_win[win]);
    const uint32_t wmask = TPC_LIST_WINDOW_WORDS - 1u;
    for (uint32_t w = w0; w < w1; w++) {
        uint32_t x = slice[w];
        while (x) {
            const uint32_t b = (uint32_t)__ffs((int)x) - 1u;
            *dst++ = (uint16_t)(((w & wmask) << 5) | b);
            x &= x - 1u;
        }
    }
    // (the entries behind a list up to its unit's end are never read: readers stop at the count)
}

}  // namespace

// One pass of k_slice_combine.  ipl: the level-2 regions of a deferred insert (or nullptr); iovf / iovf_off: its overflow entries grouped by
// slice (or nullptr); ls: received lists (n_src may be 0); dense: also write every built slice to a.filter (fresh: it starts from zero,
// else from what the filter holds); out: the lists this pass produces (or nullptr).  rank / world: the slices of the level-1 buckets that
// rank owns (world == 1: all of them).
int tpc_launch_slice_combine(const TpcLaunch &a, int slice_bits, int b1, int b2, uint32_t perm_mult, uint32_t perm_inv, const TpcPartPlan *ipl, const uint64_t *iovf,
                             const uint64_t *iovf_off, const TpcListSrc &ls, bool dense, bool fresh, const TpcCombineOut *out, uint32_t rank, uint32_t world)
{
    if (ipl && (ipl->b3 || ipl->fmt2 != 0 || ipl->slice_bits != slice_bits || ipl->b1 != b1 || ipl->b2 != b2 || ipl->world != 1)) return -1;
    if (world == 0 || (world & (world - 1)) || world > (1u << b1) || rank >= world) return -1;
    if (out && (out->n_dest == 0 || (out->n_dest & (out->n_dest - 1)) || out->n_dest > (1u << b1))) return -1;
    if (slice_bits < 6 || slice_bits > 20) return -1;
    const PtPerm perm{slice_bits, b1 + b2, perm_mult, perm_inv};
    const PtShard grid{rank, world};
    const size_t words = (size_t)1 << (slice_bits - 5);
    const size_t lds = ((words + 3) & ~(size_t)3) * 4 + (CB_THREADS / 64 + 2 * CB_MAX_WIN + 1 + 4) * 4;
    (void)hipFuncSetAttribute((const void *)k_slice_combine, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_slice_combine, dim3((1u << (b1 + b2)) / world), dim3(CB_THREADS), lds, a.stream, slice_bits, b2, ipl ? ipl->wpb : 0u, ipl ? ipl->buf2 : nullptr,
                       ipl ? ipl->cnt2 : nullptr, ipl ? ipl->cap2 : 0ull, iovf, iovf_off, ls, dense ? a.filter : nullptr, fresh ? 1 : 0, out ? out->payload : nullptr,
                       out ? out->cap : 0ull, out ? out->cur : nullptr, out ? out->dir : nullptr, out ? out->n_dest : 1u, perm, grid);
    return 0;
}
