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
#include <algorithm>
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

// The export of a rank's own insert as ONE LONG-LIVED workgroup per CU (round 6).  k_slice_combine spends a slice's ~10 us waiting for
// memory three times in a row -- the region's count, its entries, the claim of the output space -- with nothing else on the CU (128 KB of
// LDS: one workgroup): 2.7 ms for the 65536 slices of a 2^36-bit filter whatever the entries.  Here a workgroup walks the slices
// blockIdx, blockIdx + grid, ...: the entries of the NEXT slice's first region are in flight (PtStream) while this slice is counted,
// scanned and listed, that region's count was loaded an iteration before; the output space comes from a chunk of CB_CHUNK units the
// workgroup claimed for the destination earlier (one global atomic per ~dozen slices; the directory says where every list went, so the
// unused tail of a chunk is just a few KB that travel for nothing); and listing a word zeroes it for the next slice.
constexpr uint32_t CB_CHUNK = 512;  // 16-byte units per claim (8 KB): what a workgroup leaves unused is at most one window's list per chunk and half a chunk per destination at the end

template <int THREADS>
__global__ void __launch_bounds__(THREADS)
k_slice_export_p(int slice_bits, int log_nb2, uint32_t n_slices, uint32_t iwpb, const uint32_t *__restrict__ ibuf2, const uint32_t *__restrict__ icnt2, uint64_t icap2,
                 const uint64_t *__restrict__ iovf, const uint64_t *__restrict__ iovf_off, uint16_t *__restrict__ out_payload, uint64_t out_cap, unsigned long long *out_cur,
                 uint64_t *__restrict__ out_dir, uint32_t n_dest, PtPerm perm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t words = 1u << (slice_bits - 5);
    uint32_t *slice = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_w = slice + ((words + 3u) & ~3u);   // [THREADS / 64]
    uint32_t *s_win = s_w + THREADS / 64;           // [CB_MAX_WIN + 1], then [CB_MAX_WIN]
    uint32_t *s_ctl = s_win + 2 * CB_MAX_WIN + 1;   // [4]
    uint64_t *s_chunk = reinterpret_cast<uint64_t *>(s_ctl + 4);  // [64] next free unit of this workgroup's chunk per destination, [64] the chunk's end
    uint64_t *s_base = s_chunk + 128;                             // [CB_MAX_WIN] first unit of every window's list of the slice at hand
    const uint32_t nb2 = 1u << log_nb2, smask = (1u << slice_bits) - 1u;
    const bool wide = (words & 3u) == 0;
    if (wide) for (uint32_t i = threadIdx.x; i < words / 4; i += THREADS) reinterpret_cast<uint4 *>(slice)[i] = make_uint4(0, 0, 0, 0);
    else for (uint32_t i = threadIdx.x; i < words; i += THREADS) slice[i] = 0;
    if (threadIdx.x < 128) s_chunk[threadIdx.x] = 0;
    const uint32_t n_win = tpc_list_windows(slice_bits), tpw = (uint32_t)THREADS / n_win;
    const uint32_t win = threadIdx.x / tpw, tl = threadIdx.x % tpw;
    const uint32_t wwords = min(words, (uint32_t)TPC_LIST_WINDOW_WORDS), wfirst = win * wwords;
    const uint64_t slices_per_dest = ((uint64_t)1 << perm.F) / n_dest;
    auto set = [slice](uint32_t v) { atomicOr(&slice[v >> 5], 1u << (v & 31u)); };
    auto region0 = [&](uint32_t sp) { return ((uint64_t)(sp >> log_nb2) * iwpb) * nb2 + (sp & (nb2 - 1u)); };
    // pipeline registers: the first region of the slice at hand (entries in flight), the count of the one after it
    PtStream<THREADS, 2, uint32_t> st;
    uint32_t sp = blockIdx.x;
    uint32_t c_next = 0;
    uint64_t o_lo = 0, o_hi = 0;  // this slice's range of the grouped overflow entries (asked for an iteration ahead, like the count)
    if (iwpb && sp < n_slices) {
        st.begin(ibuf2 + region0(sp) * icap2, (uint32_t)__builtin_amdgcn_readfirstlane((int)icnt2[region0(sp)]));
        if (sp + gridDim.x < n_slices) c_next = icnt2[region0(sp + gridDim.x)];
    }
    if (iovf_off && sp < n_slices) { o_lo = iovf_off[sp]; o_hi = iovf_off[sp + 1]; }
    // 16-byte LDS reads where a window's words divide evenly over its threads (consecutive lanes read consecutive uint4: no bank conflicts)
    const bool vec4 = wwords % (4u * tpw) == 0;
    __syncthreads();
    for (; sp < n_slices; sp += gridDim.x) {
        const uint32_t b1 = sp >> log_nb2, b2 = sp & (nb2 - 1u);
        // ---- OR of this slice's entries (the first region's are, or are about to be, in registers)
        if (iwpb) st.finish(set);
        for (uint32_t j = 1; j < iwpb; j++) {
            const uint64_t r = ((uint64_t)b1 * iwpb + j) * nb2 + b2;
            pt_stream_region<THREADS, 2>(ibuf2 + r * icap2, (uint32_t)__builtin_amdgcn_readfirstlane((int)icnt2[r]), set);
        }
        if (iovf_off) for (uint64_t i = o_lo + threadIdx.x; i < o_hi; i += THREADS) set((uint32_t)iovf[i] & smask);
        // ---- the next slice's first region goes in flight, the count of the one after it is asked for
        const uint32_t sp1 = sp + gridDim.x, sp2 = sp1 + gridDim.x;
        if (iwpb && sp1 < n_slices) {
            st.begin(ibuf2 + region0(sp1) * icap2, (uint32_t)__builtin_amdgcn_readfirstlane((int)c_next));
            if (sp2 < n_slices) c_next = icnt2[region0(sp2)];
        }
        if (iovf_off && sp1 < n_slices) { o_lo = iovf_off[sp1]; o_hi = iovf_off[sp1 + 1]; }
        __syncthreads();
        // ---- count, scan, space, list (and zero)
        uint32_t cnt = 0;
        if (vec4) {
            const uint4 *s4 = reinterpret_cast<const uint4 *>(slice + wfirst);
            for (uint32_t q = tl; q < wwords / 4u; q += tpw) { const uint4 x = s4[q]; cnt += (uint32_t)(__popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w)); }
        } else
        for (uint32_t w = tl; w < wwords; w += tpw) cnt += (uint32_t)__popc(slice[wfirst + w]);
        uint32_t total;
        const uint32_t off = pt_block_excl_scan<THREADS>(cnt, s_w, total);
        if (tl == 0) s_win[win] = off;
        if (threadIdx.x == 0) s_win[n_win] = total;
        __syncthreads();
        const uint32_t dest = b1 & (n_dest - 1u);
        if (threadIdx.x < 64) {  // (first wave) every window's list gets its units from this workgroup's chunk for the destination, window by window
            const uint32_t v = threadIdx.x;
            const uint32_t u = v < n_win ? (s_win[v + 1] - s_win[v] + 7u) >> 3 : 0u;
            uint64_t cur = s_chunk[dest], end = s_chunk[64 + dest];
            bool ok = true;
            uint32_t inc = u;  // the usual case: the whole slice fits what is left of the chunk -- a 16-lane prefix sum places its windows
            for (int o = 1; o < CB_MAX_WIN; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if ((int)v >= o) inc += t; }
            const uint32_t units = (uint32_t)__shfl((int)inc, CB_MAX_WIN - 1, 64);
            if (cur + units <= end) {
                if (v < n_win) s_base[v] = cur + (inc - u);
                cur += units;
            } else
            for (uint32_t x = 0; x < n_win; x++) {  // (uniform over the wave: every lane walks the same sixteen sizes)
                const uint32_t ux = (uint32_t)__shfl((int)u, (int)x, 64);
                if (cur + ux > end) {  // a new chunk (what is left of the old one, less than a window's list, travels unused)
                    const uint64_t claim = ux > CB_CHUNK ? ux : CB_CHUNK;
                    unsigned long long got = 0;
                    if (v == 0) got = atomicAdd(&out_cur[dest], (unsigned long long)claim);
                    cur = (uint64_t)__shfl((long long)got, 0, 64);
                    end = cur + claim;
                    if (end > out_cap) { ok = false; end = cur; if (v == 0) out_cur[n_dest] = 1ull; }
                }
                if (v == x) s_base[x] = cur;
                if (end > cur) cur += ux;
            }
            if (v == 0) { s_chunk[dest] = cur; s_chunk[64 + dest] = end; s_ctl[2] = ok ? 1u : 0u; }
        }
        __syncthreads();
        const bool ok = s_ctl[2] != 0;
        const uint64_t key = ((uint64_t)(b1 / n_dest) << log_nb2) | b2;
        uint64_t *dir = out_dir + ((uint64_t)dest * slices_per_dest + key) * n_win;
        if (threadIdx.x < n_win) {
            const uint32_t n = s_win[threadIdx.x + 1] - s_win[threadIdx.x];
            dir[threadIdx.x] = ok ? (s_base[threadIdx.x] << 24) | (uint64_t)n : 0ull;
        }
        uint16_t *dst = out_payload + (((uint64_t)dest * out_cap + s_base[win]) << 3) + (off - s_win[win]);
        if (cnt && vec4) {
            uint4 *s4 = reinterpret_cast<uint4 *>(slice + wfirst);
            for (uint32_t q = tl; q < wwords / 4u; q += tpw) {
                const uint4 x4 = s4[q];
                if (!(x4.x | x4.y | x4.z | x4.w)) continue;
                s4[q] = make_uint4(0, 0, 0, 0);  // (the next slice starts from zero)
                if (!ok) continue;
                const uint32_t xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    uint32_t x = xs[e];
                    const uint32_t w = 4u * q + (uint32_t)e;
                    while (x) {
                        const uint32_t b = (uint32_t)__ffs((int)x) - 1u;
                        *dst++ = (uint16_t)((w << 5) | b);
                        x &= x - 1u;
                    }
                }
            }
        } else if (cnt) {
            for (uint32_t w = tl; w < wwords; w += tpw) {
                uint32_t x = slice[wfirst + w];
                if (!x) continue;
                slice[wfirst + w] = 0;  // (the next slice starts from zero)
                if (ok) {
                    while (x) {
                        const uint32_t b = (uint32_t)__ffs((int)x) - 1u;
                        *dst++ = (uint16_t)((w << 5) | b);
                        x &= x - 1u;
                    }
                }
            }
        }
        __syncthreads();  // the slice is zero and s_win / s_ctl are free again
    }
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
    static const bool no_persist = getenv("TPC_COMBINE_NO_PERSIST") != nullptr;  // (measurements: A/B against one workgroup per slice)
    const uint32_t n_slices = 1u << (b1 + b2);
    // (from 16384 slices on: a long-lived workgroup leaves up to a chunk unused per destination, which only a large export makes small)
    if (out && ipl && !dense && ls.n_src == 0 && world == 1 && parts == 1 && n_slices >= 16384 && !no_persist) {
        static const int n_cu = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; return n; }();
        const size_t words = (size_t)1 << (slice_bits - 5);
        const size_t lds = ((words + 3) & ~(size_t)3) * 4 + (1024 / 64 + 2 * CB_MAX_WIN + 1 + 4) * 4 + (128 + CB_MAX_WIN) * 8;
        const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(2, ((size_t)160 * 1024) / (lds + 1024)));  // (two 1024-thread workgroups fill a CU's wave slots)
        const uint32_t grid_p = std::min<uint32_t>(n_slices, (uint32_t)n_cu * per_cu);
        (void)hipFuncSetAttribute((const void *)k_slice_export_p<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_slice_export_p<1024>, dim3(grid_p), dim3(1024), lds, a.stream, slice_bits, b2, n_slices, ipl->wpb, ipl->buf2, ipl->cnt2, ipl->cap2, iovf, iovf_off,
                           out->payload, out->cap, out->cur, out->dir, out->n_dest, perm);
        return 0;
    }
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
