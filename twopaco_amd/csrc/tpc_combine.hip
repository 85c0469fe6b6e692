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
//                    leave as 16-bit offsets, one list per 2^16-bit window of the slice (tpc_lists.h), in the block of the
//                    rank that owns the slice (one atomic per workgroup claims the space; a directory entry per window says where
//                    its list went), and / or the dense slice is written to the filter.
//
// A list block is what travels (tpc_combine_export -> reduce-scatter by owner -> tpc_combine_merge -> all-gather -> tpc_combine_import,
// or a plain all-gather of the exports below four ranks); the query then runs entirely on the rank that hashed it: its fused
// lookup (tpc_qpart6.h:k_apply_lookup6) builds every slice from the imported lists, so no probe and no survivor ever crosses a link.
#include "tpc_internal.h"
#include "tpc_lists.h"
#include <cstdlib>

namespace {

constexpr int CB_MAX_WIN = 1 << (20 - TPC_LIST_WINDOW_BITS);

// THREADS = 1024 / parts: a workgroup builds 1 / parts of a slice (whole 2^16-bit windows; a part reads all the entries of its slice's
// regions and keeps its own, and only its own windows' received lists).  The kernel waits for memory most of its time -- region counts,
// entries, the claim of its output space, one after the other: ~10 us per workgroup, 2.7 ms for the 65536 slices of a 2^36-bit filter
// whatever the number of entries.  Parts were built to let two to four workgroups share a CU (a whole slice's 128 KB of LDS leave room
// for one); measured, they buy nothing (tpc_launch_slice_combine), so the launch uses whole slices.
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
k_slice_combine(int slice_bits, int log_nb2, uint32_t iwpb, const uint32_t *__restrict__ ibuf2, const uint32_t *__restrict__ icnt2, uint64_t icap2,
                const uint64_t *__restrict__ iovf, const uint64_t *__restrict__ iovf_off, TpcListSrc ls, uint32_t *__restrict__ filter, int fresh,
                uint16_t *__restrict__ out_payload, uint64_t out_cap, unsigned long long *out_cur, uint64_t *__restrict__ out_dir, uint32_t n_dest, PtPerm perm, PtShard grid)
{
    constexpr uint32_t PARTS = 1024 / THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t words = (1u << (slice_bits - 5)) / PARTS;  // of this part
    uint32_t *slice = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_w = slice + ((words + 3u) & ~3u);   // [THREADS / 64] scan scratch
    uint32_t *s_win = s_w + THREADS / 64;           // [CB_MAX_WIN + 1] entries before every window of the part, then [CB_MAX_WIN] its first unit
    uint32_t *s_ctl = s_win + 2 * CB_MAX_WIN + 1;   // [4]
    const uint32_t nb2 = 1u << log_nb2;
    // the permuted slice this workgroup builds a part of: every slice (grid.world == 1), or the slices of the level-1 buckets rank grid.rank owns
    const uint32_t part = blockIdx.x % PARTS, index = blockIdx.x / PARTS;
    uint32_t b1 = index >> log_nb2;
    const uint32_t b2 = index & (nb2 - 1u);
    if (grid.world > 1) b1 = b1 * grid.world + grid.rank;
    const uint32_t sp = (b1 << log_nb2) | b2;
    const uint32_t word0 = part * words;  // the part's first word in the slice
    uint32_t *out = filter ? filter + (uint64_t)perm.slice_of(sp) * (words * PARTS) + word0 : nullptr;
    const bool wide = (words & 3u) == 0;
    const uint32_t n_win = tpc_list_windows(slice_bits), n_pwin = n_win / PARTS;  // windows of the slice, of the part (the host launches PARTS <= n_win)
    TpcListReader<THREADS> lists;  // the received lists' first loads go out now: their round trips run under the zeroing
    if (ls.n_src) lists.begin(ls, b1, b2, log_nb2, index, slice_bits, part * n_pwin, n_pwin);
    if (fresh || !out) {
        if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += THREADS) reinterpret_cast<uint4 *>(slice)[i] = make_uint4(0, 0, 0, 0);
        else for (uint32_t i = threadIdx.x; i < words; i += THREADS) slice[i] = 0;
    } else {
        if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += THREADS) reinterpret_cast<uint4 *>(slice)[i] = reinterpret_cast<const uint4 *>(out)[i];
        else for (uint32_t i = threadIdx.x; i < words; i += THREADS) slice[i] = out[i];
    }
    __syncthreads();
    auto set = [slice, word0, words](uint32_t v) {
        const uint32_t w = (v >> 5) - word0;  // (unsigned: entries of the parts before this one wrap around and fail the test too)
        if (PARTS == 1 || w < words) atomicOr(&slice[w], 1u << (v & 31u));
    };
    // ---- this rank's own level-2 insert regions of the slice (32-bit slice offsets) and its overflow entries grouped by slice
    for (uint32_t j = 0; j < iwpb; j++) {
        const uint64_t r = ((uint64_t)b1 * iwpb + j) * nb2 + b2;
        const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)icnt2[r]);
        pt_stream_region<THREADS, 2>(ibuf2 + r * icap2, n, set);
    }
    if (iovf_off) {
        const uint64_t o0 = iovf_off[sp], o1 = iovf_off[sp + 1];
        const uint32_t smask = (1u << slice_bits) - 1u;
        for (uint64_t i = o0 + threadIdx.x; i < o1; i += THREADS) set((uint32_t)iovf[i] & smask);
    }
    // ---- set-bit lists (this kernel's own output format) received from other ranks
    if (ls.n_src) lists.finish(ls, slice);
    __syncthreads();
    if (out) {
        if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += THREADS) reinterpret_cast<uint4 *>(out)[i] = reinterpret_cast<const uint4 *>(slice)[i];
        else for (uint32_t i = threadIdx.x; i < words; i += THREADS) out[i] = slice[i];
    }
    if (!out_payload) return;
    // ---- the part's set bits.  The tpw threads of a window (a contiguous range of threads) list that window's bits: thread tl of the
    // window takes its words tl, tl + tpw, tl + 2 tpw, ... -- consecutive lanes read consecutive LDS words (a run of consecutive words per
    // thread put every lane of a wave on one bank: 16 us per slice) -- and writes its bits behind those of the threads before it: a
    // window's list is contiguous and starts on a 16-byte unit, the order of the offsets inside it is of no consequence to a reader.
    const uint32_t tpw = (uint32_t)THREADS / n_pwin;
    const uint32_t win = threadIdx.x / tpw, tl = threadIdx.x % tpw;
    const uint32_t wwords = min(words, (uint32_t)TPC_LIST_WINDOW_WORDS), wfirst = win * wwords;  // the window's words
    uint32_t cnt = 0;
    for (uint32_t w = tl; w < wwords; w += tpw) cnt += (uint32_t)__popc(slice[wfirst + w]);
    uint32_t total;
    const uint32_t off = pt_block_excl_scan<THREADS>(cnt, s_w, total);
    if (tl == 0) s_win[win] = off;  // entries before the window
    if (threadIdx.x == 0) s_win[n_pwin] = total;
    __syncthreads();
    // destination block and the slice's index there: [local bucket of the destination][b2]
    const uint32_t dest = b1 & (n_dest - 1u);
    const uint64_t key = grid.world > 1 ? (uint64_t)index : ((uint64_t)(b1 / n_dest) << log_nb2) | b2;
    if (threadIdx.x == 0) {
        uint32_t units = 0;
        for (uint32_t v = 0; v < n_pwin; v++) { s_win[CB_MAX_WIN + 1 + v] = units; units += (s_win[v + 1] - s_win[v] + 7u) >> 3; }
        const uint64_t base = units ? (uint64_t)atomicAdd(&out_cur[dest], (unsigned long long)units) : 0ull;
        const bool ok = base + units <= out_cap;
        if (!ok) out_cur[n_dest] = 1ull;  // the block is too small (the host sizes it from the entry counts: tpc_combine_info)
        s_ctl[0] = (uint32_t)base; s_ctl[1] = (uint32_t)(base >> 32); s_ctl[2] = ok ? 1u : 0u;
    }
    __syncthreads();
    const bool ok = s_ctl[2] != 0;
    const uint64_t base = (uint64_t)s_ctl[0] | ((uint64_t)s_ctl[1] << 32);
    const uint64_t slices_per_dest = ((uint64_t)1 << (perm.F)) / (grid.world > 1 ? grid.world : n_dest);
    uint64_t *dir = out_dir + ((uint64_t)(grid.world > 1 ? 0u : dest) * slices_per_dest + key) * n_win + part * n_pwin;
    if (threadIdx.x < n_pwin) {
        const uint32_t n = s_win[threadIdx.x + 1] - s_win[threadIdx.x];
        dir[threadIdx.x] = ok ? ((base + s_win[CB_MAX_WIN + 1 + threadIdx.x]) << 24) | (uint64_t)n : 0ull;
    }
    if (!ok || cnt == 0) return;
    uint16_t *dst = out_payload + (((uint64_t)dest * out_cap + base + s_win[CB_MAX_WIN + 1 + win]) << 3) + (off - s_win[win]);
    for (uint32_t w = tl; w < wwords; w += tpw) {
        uint32_t x = slice[wfirst + w];
        while (x) {
            const uint32_t b = (uint32_t)__ffs((int)x) - 1u;
            *dst++ = (uint16_t)((w << 5) | b);
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
    // parts of a slice per workgroup (see k_slice_combine; TPC_COMBINE_PARTS = 2 / 4: measurements).  One is the default: on the 62-genome
    // text at eight ranks the export took 2.74 ms as whole slices and 3.32 ms as quarters (profiles/r06_combine_ab.txt) -- the time is the
    // chain of dependent memory round trips of a workgroup (count, entries, claim), which does not shrink with the part, and four times
    // the workgroups four at a time is no gain.
    const uint32_t n_win = tpc_list_windows(slice_bits);
    static const int parts_env = [] { const char *e = getenv("TPC_COMBINE_PARTS"); return e ? atoi(e) : 0; }();
    uint32_t parts = 1;
    if (parts_env == 1 || parts_env == 2 || parts_env == 4) parts = (uint32_t)parts_env;
    while (parts > n_win) parts >>= 1;
    const size_t words = ((size_t)1 << (slice_bits - 5)) / parts;
    const uint32_t threads = 1024u / parts;
    const size_t lds = ((words + 3) & ~(size_t)3) * 4 + (threads / 64 + 2 * CB_MAX_WIN + 1 + 4) * 4;
    const dim3 blocks(((1u << (b1 + b2)) / world) * parts);
#define TPC_CB_GO(T)                                                                                                                                       \
    do {                                                                                                                                                   \
        (void)hipFuncSetAttribute((const void *)k_slice_combine<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
        hipLaunchKernelGGL(k_slice_combine<T>, blocks, dim3(T), lds, a.stream, slice_bits, b2, ipl ? ipl->wpb : 0u, ipl ? ipl->buf2 : nullptr,              \
                           ipl ? ipl->cnt2 : nullptr, ipl ? ipl->cap2 : 0ull, iovf, iovf_off, ls, dense ? a.filter : nullptr, fresh ? 1 : 0,               \
                           out ? out->payload : nullptr, out ? out->cap : 0ull, out ? out->cur : nullptr, out ? out->dir : nullptr, out ? out->n_dest : 1u, perm, grid); \
    } while (0)
    if (parts == 4) TPC_CB_GO(256); else if (parts == 2) TPC_CB_GO(512); else TPC_CB_GO(1024);
#undef TPC_CB_GO
    return 0;
}
