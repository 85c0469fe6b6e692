// tpc_internal.h -- host-side launch interface between the C-ABI (tpc_capi*.hip, tpc_ctx.h) and the kernels.
#pragma once
#include "tpc_device.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

// Measurement switches of the kernels' launch and planning code: read from the environment ONCE per process (they used to be looked up
// on every pass).  TPC_NO_LEAN: the generic hash kernels; TPC_RB_HASH: the barrier-free rings in the 512-bin hash; TPC_VERIFY_LAZY=0:
// all q - 1 probes at once; TPC_GATED_FULL_REGIONS: gated rounds sized for all entries; TPC_PPR_INSERT / TPC_GATED_LOADS: round sizes.
struct TpcEnv {
    bool no_lean, rb_hash, verify_eager, gated_full;
    int ppr_insert, gated_loads, split_loads9;  // 0: not set
    int slice_grid;  // workgroups of a one-slice-at-a-time kernel (tpc_slice_grid); 0: one per slice
    static const TpcEnv &get()
    {
        static const TpcEnv e = [] {
            TpcEnv v;
            v.no_lean = getenv("TPC_NO_LEAN") != nullptr;
            v.rb_hash = getenv("TPC_RB_HASH") != nullptr;
            const char *lz = getenv("TPC_VERIFY_LAZY");
            v.verify_eager = lz && lz[0] == '0';
            v.gated_full = getenv("TPC_GATED_FULL_REGIONS") != nullptr;
            const char *p = getenv("TPC_PPR_INSERT"), *g = getenv("TPC_GATED_LOADS");
            v.ppr_insert = p ? atoi(p) : 0;
            v.gated_loads = g ? atoi(g) : 0;
            const char *s9 = getenv("TPC_SPLIT_LOADS9");  // (measurements: entries per thread and round of k_part_split at 512 bins)
            v.split_loads9 = s9 ? atoi(s9) : 0;
            // two long-lived workgroups per CU (one resident beside a 128 KB slice, one queued) unless TPC_LOOKUP_GRID says otherwise
            const char *lg = getenv("TPC_LOOKUP_GRID");
            int n_cu = 0, dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) { (void)hipGetLastError(); n_cu = 256; }
            v.slice_grid = lg ? atoi(lg) : 2 * n_cu;
            return v;
        }();
        return e;
    }
};

// Grid of a kernel that builds or reads one filter slice at a time in LDS: a few long-lived workgroups, each taking every grid-th slice
// (round 6, same-box A/B on the 62-genome step: k_apply_lookup6 5.83 ms as 65536 workgroups, 5.48 as 512 -- the dispatch of a workgroup
// per 22 us of work was not free after all; profiles/r06_lookup_grid_ab.txt).
inline uint32_t tpc_slice_grid(uint32_t n_slices)
{
    const int g = TpcEnv::get().slice_grid;
    return g > 0 ? std::min<uint32_t>(n_slices, (uint32_t)g) : n_slices;
}

#define TPC_TAB_MAXQ 64                // = TPC_MAX_Q (include/twopaco_hip.h)
#define TPC_KERNEL_MAXQ 16             // the rolling kernels are instantiated for 1..16 functions; beyond that tpc_pass1_anyq.hip
#define TPC_TAB_HK (TPC_TAB_MAXQ * 5)  // device table layout: h[64][5] then hk[64][5]
#define TPC_TAB_WORDS (2 * TPC_TAB_MAXQ * 5)

struct TpcLaunch {
    TpcHashParams P;
    const uint64_t *tab;    // device: character tables
    const uint64_t *bases;  // device: packed text
    const uint32_t *nmask;  // device: N mask
    uint64_t n_text;
    uint64_t n_tiles;       // workgroups of 8192 positions covering the text
    uint32_t *filter;       // device: Bloom filter words
    hipStream_t stream;
    // periodic windows (tpc_qpartition.hip:k_periodic_build; nullptr: off): one bit per position, laid out like nmask.  per_i: the
    // (k+1)-mer at i equals the one 1..63 positions earlier -- its insert adds nothing; per_qs: the k + 2 characters around the vertex at i
    // equal those around i - p -- its candidate verdict is that position's, copied after the verification (p: six bit planes, the copy's business)
    const uint32_t *per_i = nullptr, *per_qs = nullptr;
    // the closed-form kernels of tpc_pass1_anyq.hip (q = 17..64) over a range of positions only: a rank's chunk of a replicated multi-GPU pass
    uint64_t g_begin = 0, g_end = ~0ull;
    // measurement: recorded around the k_apply_lookup launch of tpc_launch_query_part_fused_lookup when set (TPC_K_LOOKUP)
    hipEvent_t ev_lookup0 = nullptr, ev_lookup1 = nullptr;
};
constexpr int TPC_PER_MAXP = 63;    // periods of the periodic windows the first pass skips (tpc_qpartition.hip:k_periodic_build)
constexpr int TPC_PER_PLANES = 6;   // bit planes of the copy distance; the masks of a text are [2 + TPC_PER_PLANES][n_words_alloc]: per_qs, the planes, per_i
int tpc_launch_periodic_build(const TpcLaunch &a, uint32_t *qs, uint32_t *qd, uint64_t stride, uint32_t *ins, uint64_t w_begin, uint64_t w_end, uint64_t pos_lo, uint64_t pos_hi,
                              uint32_t *any);
int tpc_launch_periodic_copy(hipStream_t stream, uint32_t *rmask, const uint32_t *qs, const uint32_t *qd, uint64_t stride, uint64_t n_words);

// pass 1 (tpc_pass1.hip)
int tpc_launch_insert(const TpcLaunch &a, uint64_t lo, uint64_t hi, bool gated, bool test, unsigned long long *n_kmers);
int tpc_launch_query(const TpcLaunch &a, uint32_t *rmask, uint64_t lo, uint64_t hi, bool gated, unsigned long long *n_marks);
int tpc_launch_split(const TpcLaunch &a, uint32_t *emask, uint32_t *bins, uint64_t bin_size);  // emask is consumed (occurrences still to be counted)
int tpc_launch_split_emask(hipStream_t s, const uint64_t *d_rec_start, const uint64_t *d_rec_len, uint32_t n_rec, int k, uint64_t n_words, uint32_t *emask);  // the split pass's positions, from the DISPATCHED records (len >= k) in text order
int tpc_launch_hash_dump(const TpcLaunch &a, uint64_t g0, uint64_t n, uint64_t *out);
// more than TPC_KERNEL_MAXQ hash functions (tpc_pass1_anyq.hip): the same three passes with every hash in closed form
int tpc_launch_insert_anyq(const TpcLaunch &a, uint64_t lo, uint64_t hi, bool gated, bool test, unsigned long long *n_kmers);
int tpc_launch_query_anyq(const TpcLaunch &a, uint32_t *rmask, uint64_t lo, uint64_t hi, bool gated, unsigned long long *n_marks);
int tpc_launch_split_anyq(const TpcLaunch &a, uint32_t *emask, uint32_t *bins, uint64_t bin_size);
extern int tpc_test_force_anyq;  // tpc_pass1.hip: option "test_force_anyq"

// partitioned insert (tpc_partition.hip)
struct TpcPartPlan {
    int slice_bits, b1, b2, pos_per_round;
    int b3 = 0;           // > 0: three levels (filters beyond 2^38 bits at the default slice size)
    uint32_t perm_mult, perm_inv;  // slice-index permutation (tpc_bins.h:PtPerm)
    uint64_t tile0, n_tiles;  // 512-word tiles of the text handled by this batch
    uint32_t nwg1, wpb;   // level-1 workgroups; level-2 workgroups per level-1 bucket
    uint32_t wpb3 = 1;    // level-3 workgroups per (b1, b2) bucket
    uint64_t cap1, cap2;  // entries per private region (multiples of 32; cap2 a multiple of 40 = PFmt3::GROUP when fmt2 == 3)
    uint64_t cap3 = 0;
    int fmt2 = 0;         // level-2 regions: 0 = 32-bit entries, 3 = blocked lines of 40 x 24-bit entries (tpc_binsp.h:PFmt3)
    uint64_t ovf_cap;
    uint32_t *buf1, *cnt1, *buf2, *cnt2;
    uint32_t *buf3 = nullptr, *cnt3 = nullptr;
    uint64_t *ovf;
    unsigned long long *ovf_cur;  // [0] count, [1] overflow-of-overflow flag
    // filter sharding (tpc_bins.h:PtShard): this rank, number of ranks; the level-1 regions the split
    // kernel reads (== buf1 / cnt1 when world == 1, the all_to_all receive buffers otherwise)
    uint32_t rank = 0, world = 1;
    const uint32_t *rbuf1 = nullptr, *rcnt1 = nullptr;
    const uint64_t *roff1 = nullptr;  // packed receive buffer: first entry of every received level-1 region (else region index * cap1)
    // this rank's own block read where tpc_shard_hash wrote it (tpc_shard_apply_inplace): the send buffers, same region index
    const uint32_t *rown1 = nullptr, *rowncnt1 = nullptr;
};
bool tpc_part_plan(int L, int q, int slice_bits, uint64_t n_tiles, double frac, TpcPartPlan &pl, int levels = 0);  // n_tiles: 512-word tiles per batch; levels 0 = auto
size_t tpc_part_buf1_bytes(const TpcPartPlan &pl);
size_t tpc_part_cnt1_bytes(const TpcPartPlan &pl);
size_t tpc_part_buf2_bytes(const TpcPartPlan &pl);
size_t tpc_part_cnt2_bytes(const TpcPartPlan &pl);
size_t tpc_part_buf3_bytes(const TpcPartPlan &pl);
size_t tpc_part_cnt3_bytes(const TpcPartPlan &pl);
bool tpc_part_plan_sharded(int L, int q, int slice_bits, uint64_t n_tiles, double frac, uint32_t rank, uint32_t world, TpcPartPlan &pl, int levels = 0, bool tight = false,
                           bool packed = false);  // packed: the one-GPU passes may use the planar entry formats of tpc_binsp.h
int tpc_launch_insert_partitioned(const TpcLaunch &a, const TpcPartPlan &pl, uint64_t lo, uint64_t hi, bool gated, bool fresh,
                                  unsigned long long *n_kmers);
// the two halves of the above, for the sharded path (an all_to_all of the level-1 regions sits between them)
int tpc_launch_insert_part_hash(const TpcLaunch &a, const TpcPartPlan &pl, uint64_t lo, uint64_t hi, bool gated, unsigned long long *n_kmers);
int tpc_launch_insert_part_apply(const TpcLaunch &a, const TpcPartPlan &pl, bool fresh);

// partitioned query (tpc_qpartition.hip)
struct TpcQPlan {
    int slice_bits, b1, b2, pos_per_round, sub_rounds, loads;
    int b3 = 0, loads3 = 1;  // three-level geometry (b3 > 0), as in TpcPartPlan
    uint32_t wpb3 = 1;
    uint64_t cap2 = 0;       // three levels: uniform size of the middle regions
    std::vector<uint64_t> off3_host;  // region offsets of the third level (one region per slice)
    uint64_t buf3_entries = 0;
    uint64_t *buf3 = nullptr;
    uint32_t *cnt3 = nullptr;
    const uint64_t *off3 = nullptr;
    uint32_t perm_mult, perm_inv;
    uint64_t tile0, n_tiles;
    uint64_t tile0_global = 0;  // first tile of the batch (positions in entries are relative to it; == tile0 unless sharded)
    uint32_t nwg1, wpb;
    uint64_t cap1;         // entries (uint64) per level-1 region, multiple of 16
    std::vector<uint64_t> off2_host;  // level-2 region offsets (entries), one per (b1, j, b2) + end
    uint64_t buf2_entries;
    const uint64_t *off2;  // device copy
    uint64_t ovf_cap;      // {address, survivor id} pairs
    uint64_t surv_cap;     // per survivor sub-list (64 of them)
    uint64_t *buf1, *buf2;
    uint32_t *cnt1, *cnt2;
    uint64_t *ovf;
    unsigned long long *ovf_cur;   // [0] count, [1] overflow flag
    uint64_t *surv;
    unsigned long long *surv_cur;  // [0..63] counts, [64] overflow flag
    uint32_t rank = 0, world = 1;  // filter sharding, as in TpcPartPlan
    const uint64_t *rbuf1 = nullptr;
    const uint32_t *rcnt1 = nullptr;
    const uint64_t *roff1 = nullptr;  // packed receive buffer, as in TpcPartPlan
    const uint64_t *rown1 = nullptr;  // own block in place, as in TpcPartPlan
    const uint32_t *rowncnt1 = nullptr;
    bool group_survivors = true;      // the lookup appends its survivors grouped by address (tpc_qpartition.hip:SurvStage)
    // fmt == 6 (round 5; one rank, two levels, 16..512 bins at level 2): the level-2 entries are 48 bits in blocked lines of 20
    // (tpc_binsp.h:PFmt6, tpc_qpart6.h).  buf2 is addressed in 128-byte lines, off2_host counts lines, cnt2 is exact, `bnd` receives
    // n_groups entry counts per level-2 region, and level 1 hashes contiguous blocks of tiles_per_wg tiles per workgroup.
    int fmt = 0;
    bool presplit = false;  // the level-2 (and 3) binning of this batch has already run (tpc_pass1_query_begin): the lookup launches skip it
    uint32_t tiles_per_wg = 0, n_groups = 0, pb2 = 0;  // pb2: position bits a level-2 entry carries (groups of 2^pb2 positions)
    uint32_t *bnd = nullptr;
};
#define TPC_SURV_CUR_WORDS 72  // surv_cur: [0..63] sub-list cursors, [64] overflow flag
bool tpc_qpart_plan(int L, int slice_bits, uint64_t n_tiles, double frac, TpcQPlan &pl, int levels = 0);  // n_tiles: 512-word tiles per batch
size_t tpc_qpart_bytes(const TpcQPlan &pl, int which);  // 0 buf1, 1 cnt1, 2 buf2, 3 cnt2, 4 ovf, 5 ovf_cur, 6 surv, 7 surv_cur, 8 off2, 9 buf3, 10 cnt3, 11 off3, 18 bnd (fmt 6)
bool tpc_qpart_plan_sharded(int L, int slice_bits, uint64_t n_tiles, double frac, uint32_t rank, uint32_t world, TpcQPlan &pl, int levels = 0, bool tight = false,
                            bool packed = false);
int tpc_launch_query_partitioned(const TpcLaunch &a, const TpcQPlan &pl, uint32_t *rmask, uint64_t lo, uint64_t hi, bool gated);
// deferred apply: the insert stops after its level-2 binning (tpc_launch_insert_part_split), the query's lookup builds the slices itself
int tpc_launch_insert_part_split(const TpcLaunch &a, const TpcPartPlan &pl);   // level 2 (and 3) only
int tpc_launch_insert_part_apply_only(const TpcLaunch &a, const TpcPartPlan &pl, bool fresh);  // k_part_apply + k_part_ovf
// iovf_sorted / iovf_off: the insert's overflow entries grouped by slice (tpc_launch_ovf_by_slice), or nullptr when there are none
struct TpcListSrc;
int tpc_launch_query_part_fused_lookup(const TpcLaunch &a, const TpcQPlan &pl, const TpcPartPlan &ipl, bool fresh, const uint64_t *iovf_sorted, const uint64_t *iovf_off,
                                       const TpcListSrc *lists = nullptr);  // k_q_split + k_apply_lookup + k_q_ovf
int tpc_launch_ovf_by_slice(const TpcLaunch &a, const uint64_t *list, uint64_t n, int slice_bits, uint32_t n_slices, uint32_t *cnt, uint32_t *cursor,
                            uint64_t *off, uint64_t *sorted, uint32_t rank = 0, uint32_t world = 1, int log_nb2 = 0);  // world > 1: local slices of this rank's shard
#define TPC_FUSE_MAX_OVF (16ull << 20)  // insert overflow entries (ring or region full) up to which the apply is still deferred
int tpc_launch_query_verify(const TpcLaunch &a, const TpcQPlan &pl, uint32_t *rmask);
extern int tpc_test_tight_pinch;     // tpc_partition.hip: option "test_tight_pinch" (tests: tight regions of a sharded pass at N % of their expected fill + a 64-entry overflow list)
extern int tpc_test_q6_pb2;          // tpc_qpartition.hip: option "test_q6_pb2" (tests: position bits of a 6-byte level-2 entry, to get many groups on small inputs; process-wide)
extern int tpc_test_insert_p3;       // tpc_partition.hip: option "insert_entry_fmt" (3: the level-2 insert entries as blocked 24-bit lines; process-wide)
extern uint32_t tpc_test_sched_cap;  // tpc_partition.hip: option "test_sched_cap" (tests: rounds per schedule segment of the split kernels)
// compacted exchange of the sharded path: off[i] = entries before region i (off[n] = all of them), in 16-byte units of
// `entry_bytes`-byte entries -- counts written by Bins are whole flush groups, so every region stays 128-byte aligned;
// pack copies the used prefix of every fixed-capacity region to its offset
int tpc_launch_region_offsets(const TpcLaunch &a, const uint32_t *cnt, uint32_t n_regions, uint64_t *off);
int tpc_launch_region_pack(const TpcLaunch &a, const void *regions, uint64_t cap_entries, uint32_t entry_bytes, const uint32_t *cnt, const uint64_t *off,
                           uint32_t n_regions, void *packed);
// halves for the sharded path: hash (level 1, marks N-adjacent vertices in rmask), then split + lookup on
// the owned slices (first-probe survivors into pl.surv; no verification: the caller routes them)
int tpc_launch_query_part_hash(const TpcLaunch &a, const TpcQPlan &pl, uint32_t *rmask, uint64_t lo, uint64_t hi, bool gated);
int tpc_launch_query_part_lookup(const TpcLaunch &a, const TpcQPlan &pl);
int tpc_launch_query_part_split(const TpcLaunch &a, const TpcQPlan &pl);  // the binning levels below the first alone (what the lookup launches start with unless pl.presplit)
// sharded verification of first-probe survivors (survivor id = edge | position << 3, batch relative)
int tpc_launch_surv_gather(const TpcLaunch &a, const TpcQPlan &pl, uint64_t *out);  // 64 sub-lists -> one list (sum of min(surv_cur, surv_cap) entries)
int tpc_launch_verify_addrs(const TpcLaunch &a, const TpcQPlan &pl, int fn, int fn_count, const uint64_t *sid, uint64_t n, uint64_t *addr_out, int32_t *owner_out,
                            unsigned long long *owner_counts = nullptr);
#define TPC_V_OWNER_SHIFT 56  // tagged probe addresses (owner_out == nullptr): owner rank above the shard-local bit address
int tpc_launch_finish(hipStream_t s, const TpcQPlan &pl, const uint64_t *sid, uint64_t n, int fn_count, const uint8_t *hit, const uint32_t *perm, uint32_t *rmask,
                      unsigned long long *n_marked);
int tpc_launch_route64(hipStream_t s, const uint64_t *v, uint64_t n, int shift, uint32_t omask, uint64_t keep, unsigned long long *counts, unsigned long long *cursor,
                       uint32_t *perm, uint64_t *dst, int phase);
int tpc_launch_shard_probe(const TpcLaunch &a, const uint64_t *addr, uint64_t n, uint8_t *hit);
int tpc_launch_shard_mark(const TpcLaunch &a, const TpcQPlan &pl, const uint64_t *sid, uint64_t n, uint32_t *rmask);

int tpc_launch_survivor_sources(hipStream_t s, const uint64_t *sid, uint64_t n, uint32_t world, int32_t *src);
// owner routing of survivor probes (world <= 64): counts -> cursors -> perm; dst[perm[i]] = src[i]; survivors with all answers 1
int tpc_launch_route(hipStream_t s, const int32_t *owner, uint64_t n, unsigned long long *counts, unsigned long long *cursor, uint32_t *perm, int phase);
int tpc_launch_permute64(hipStream_t s, const uint64_t *src, const uint32_t *perm, uint64_t n, uint64_t *dst);
int tpc_launch_select(hipStream_t s, const uint64_t *sid, uint64_t n, int fn_count, const uint8_t *hit, const uint32_t *perm, uint64_t *out, unsigned long long *n_out);

// ---- combined multi-GPU exchange: set-bit lists of filter slices (tpc_lists.h, tpc_combine.hip)
constexpr int TPC_LIST_WINDOW_BITS = 16;                           // bits of a slice behind one list
constexpr int TPC_LIST_WINDOW_WORDS = 1 << (TPC_LIST_WINDOW_BITS - 5);

// Where the lists to apply are (POD, passed to kernels by value).  Source s has its block at 16-byte unit base[s] of payload and its
// directory at dir + s * dir_stride.  A workgroup that builds the slice with directory key `key` reads entries
// dir[s * dir_stride + key * n_windows + w].
struct TpcListSrc {
    const uint16_t *payload = nullptr;  // device
    const uint64_t *base = nullptr;     // device, [n_src]: first unit of every source's block
    const uint64_t *dir = nullptr;      // device, [n_src][dir_stride]
    uint64_t dir_stride = 0;
    uint32_t n_src = 0;
    // n_owner == 0: every block lists every slice the reading grid builds, keyed by that grid's workgroup index (the blocks a
    // reduce-scatter delivered to the owner of those slices).  n_owner = W > 0 (all-gathered blocks): block s lists only the slices
    // of the level-1 buckets b1 with b1 % W == s % W, keyed by their local index [b1 / W][b2] -- a slice reads the n_src / W blocks
    // s = b1 % W, b1 % W + W, ... (one per rank that exported, or the one merged block of the bucket's owner).
    uint32_t n_owner = 0;
};

__host__ __device__ __forceinline__ uint32_t tpc_list_windows(int slice_bits) { return slice_bits > TPC_LIST_WINDOW_BITS ? 1u << (slice_bits - TPC_LIST_WINDOW_BITS) : 1u; }

// what a pass of k_slice_combine produces: n_dest blocks of `cap` 16-byte units each in payload, cur[d] = units used of block d
// (cur[n_dest] != 0: a block was too small), dir = [n_dest][slices per destination][windows] (tpc_lists.h)
struct TpcCombineOut { uint16_t *payload; uint64_t cap; unsigned long long *cur; uint64_t *dir; uint32_t n_dest; };
int tpc_launch_slice_combine(const TpcLaunch &a, int slice_bits, int b1, int b2, uint32_t perm_mult, uint32_t perm_inv, const TpcPartPlan *ipl, const uint64_t *iovf,
                             const uint64_t *iovf_off, const TpcListSrc &ls, bool dense, bool fresh, const TpcCombineOut *out, uint32_t rank, uint32_t world);

// pass 2 / output (tpc_pass2.hip)
// Ordered compaction of a bit mask into the list of set positions.  block_sums: scratch of
// n_words/256+2 uint64; *n_out (device) receives the list length; list must hold it (two-phase:
// call with list == nullptr to count only).
int tpc_launch_mask_count(hipStream_t s, const uint32_t *mask, uint64_t n_words, uint64_t *block_sums, unsigned long long *n_out);
int tpc_launch_mask_scatter(hipStream_t s, const uint32_t *mask, uint64_t n_words, const uint64_t *block_sums, uint64_t *list);
int tpc_launch_mask_or(hipStream_t s, uint32_t *dst, const uint32_t *src, uint64_t n_words);

// Exact filter.  Table slot = C key words (C == 1) or one representative mark index (C > 1) plus a
// meta word; see tpc_pass2.hip.
size_t tpc_table_slot_bytes(int C);
int tpc_launch_table_init(hipStream_t s, void *table, uint64_t cap);
// counted: keep exact occurrence counts (needed only when the abundance cut can apply)
// *overflow (device) is set when a probe sequence exceeds TPC_FILTER2_PROBE_LIMIT slots: the table is too small for the
// number of distinct keys and the pass must be repeated with a larger one
#define TPC_FILTER2_PROBE_LIMIT 512u
int tpc_launch_mark_owner(const TpcLaunch &a, int C, const uint64_t *marks, uint64_t n_marks, uint32_t world, int32_t *owner);  // key-hash owner of every mark
// text-free variant: records of C + 1 words (canonical key, prev | next << 3) instead of positions
int tpc_launch_mark_records(const TpcLaunch &a, int C, const uint64_t *marks, uint64_t n_marks, uint32_t world, uint64_t *records, int32_t *owner);
int tpc_launch_table_records(const TpcLaunch &a, int C, const uint64_t *marks, const void *table, uint64_t cap, const uint64_t *block_off, uint32_t world,
                             uint64_t *records, int32_t *owner);  // every used slot of a k_filter2 table as an aggregated record (block_off: scan of the used counts)
int tpc_launch_filter2_rec(const TpcLaunch &a, int C, const uint64_t *records, uint64_t n, void *table, uint64_t cap, bool counted, unsigned long long *overflow);
int tpc_launch_scan2_write_rec(const TpcLaunch &a, int C, const uint64_t *records, const void *table, uint64_t cap, uint64_t abundance, bool counted,
                               const uint64_t *block_off, uint64_t *keys_out);
int tpc_launch_permute_rows(hipStream_t s, const uint64_t *src, const uint32_t *perm, uint64_t n, int row_words, uint64_t *dst);  // dst row perm[i] = src row i
int tpc_launch_filter2(const TpcLaunch &a, int C, const uint64_t *marks, uint64_t n_marks, void *table, uint64_t cap, bool counted,
                       unsigned long long *overflow);
// TrueBifurcations in two atomic-free passes over TPC_SCAN2_BLOCKS chunks of the table.
// count: block_tp / block_used (TPC_SCAN2_BLOCKS uint64 each) become exclusive offsets; totals[0] =
// true junctions, totals[1] = table size.  write: junction keys at block_tp offsets.
#define TPC_SCAN2_BLOCKS 4096
int tpc_launch_scan2_count(const TpcLaunch &a, const void *table, uint64_t cap, uint64_t abundance, bool counted, uint64_t *block_tp,
                           uint64_t *block_used, unsigned long long *totals);
int tpc_launch_scan2_write(const TpcLaunch &a, int C, const uint64_t *marks, const void *table, uint64_t cap, uint64_t abundance, bool counted,
                           const uint64_t *block_off, uint64_t *keys_out);

// Sort junction keys (J x C, in place) in CompressedString::Less order (rocPRIM radix sort).  *scratch / *scratch_bytes:
// a device buffer owned by the caller that the one-word path grows as needed (multi-word keys allocate inside).
int tpc_launch_sort_keys(hipStream_t s, int C, int k, uint64_t *keys, uint64_t J, void **scratch, size_t *scratch_bytes);

// id index over sorted keys + output-pass lookup
int tpc_launch_idtab_build(hipStream_t s, int C, const uint64_t *keys, uint64_t J, uint32_t *idtab, uint64_t cap);
int tpc_launch_emit(const TpcLaunch &a, int C, const uint64_t *marks, uint64_t n_marks, const uint64_t *keys, uint64_t J,
                    const uint32_t *idtab, uint64_t cap, int64_t *ids, unsigned long long *n_valid);

// junction stream = the bytes of the output file (tpc_stream.hip)
size_t tpc_stream_plan_bytes(uint32_t n_rec);
int tpc_launch_stream_plan(hipStream_t s, const uint64_t *d_rec_start, const uint64_t *d_rec_len, uint32_t n_rec, int k, const uint64_t *marks,
                           const int64_t *ids, uint64_t n_marks, uint64_t *vscan, void *rec, uint32_t r_last, uint64_t *totals_host);
int tpc_launch_stream_write(hipStream_t s, const uint64_t *d_rec_start, const uint64_t *d_rec_len, uint32_t n_rec, int k, const uint64_t *marks,
                            const int64_t *ids, uint64_t n_marks, const uint64_t *vscan, const void *rec, uint32_t r_last, uint64_t first_stub, uint32_t *out);

// the stream cut over several ranks by text position (tpc_stream.hip)
int tpc_launch_stream_partial(hipStream_t s, const uint64_t *d_rec_start, const uint64_t *d_rec_len, uint32_t n_rec, int k, const uint64_t *marks,
                              const int64_t *ids, uint64_t n_marks, uint64_t *vscan, uint64_t *cnt, uint32_t *flags, uint64_t *mark_lo);
int tpc_launch_stream_write_part(hipStream_t s, const uint64_t *d_rec_start, const uint64_t *d_rec_len, uint32_t n_rec, int k, const uint64_t *marks,
                                 const int64_t *ids, uint64_t n_marks, const uint64_t *vscan, const uint64_t *mark_lo, const uint32_t *gflags,
                                 const uint64_t *e_scan, const uint64_t *s_scan, const uint64_t *before, uint32_t r_last, uint64_t first_stub,
                                 uint64_t chunk_lo, uint64_t chunk_hi, uint64_t slot0, uint32_t *out);

// code-object warm-up (an attribute query of one kernel per translation unit makes the runtime load its code object), used by tpc_preload
int tpc_warm_pass1();
int tpc_warm_partition();
int tpc_warm_qpartition();
int tpc_warm_pass2();
int tpc_warm_stream();
