// tpc_ctx.h -- the context behind the C-ABI (struct tpc_ctx) and the helpers its translation units share:
//   tpc_capi.hip          context, options, parameters, text, the first pass (tpc_pass1_*), filter / mask transfers, timers
//   tpc_capi_pass2.hip    second pass, junction keys, ids, junction stream (tpc_pass2_*, tpc_junction*, tpc_emit*)
//   tpc_capi_shard.hip    the filter cut by address over ranks (tpc_shard_*), mask unions
//   tpc_capi_combine.hip  the filter replicated through set-bit lists (tpc_combine_*)
// No CPU fallback anywhere: every entry point needs a HIP device.
#pragma once
#include "../../include/twopaco_hip.h"
#include "tpc_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <map>

struct tpc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // parameters
    bool have_params = false;
    TpcHashParams P{};
    uint64_t tab_host[TPC_TAB_WORDS]{};
    uint64_t *tab = nullptr;
    int C = 1;
    // text
    uint64_t *bases = nullptr;
    uint32_t *nmask = nullptr;
    uint64_t n_text = 0, n_words = 0, n_words_alloc = 0, n_tiles = 0;
    // periodic-window masks of the text (tpc_internal.h:TpcLaunch): [8][n_words_alloc] = per_qs, the six bit planes of the copy distance, per_i (at 7 * n_words_alloc); built at the first
    // partitioned pass after an upload / a change of k (ensure_periodic), option "periodic_skip" (default on)
    uint32_t *periodic = nullptr;
    bool periodic_valid = false;
    bool periodic_any_q = false, periodic_any_i = false;  // some position copies its verdict / drops its insert (else the masks are never read)
    int opt_periodic = 1;
    bool opt_shard_periodic = false;  // option shard_periodic_skip: the tpc_shard_hash kernels skip too, the host calls tpc_shard_periodic_copy
    // sharded contexts may hold only the words of the tiles they hash (+ halo): bases / nmask then point text_w0 words BEFORE
    // the allocations, so that kernels keep indexing by global word
    uint64_t *bases_alloc = nullptr;
    uint32_t *nmask_alloc = nullptr;
    int opt_text_window = 0;
    bool text_windowed = false;
    uint64_t text_w0 = 0, text_w1 = 0;
    // filter + masks
    uint32_t *filter = nullptr;
    uint64_t filter_words = 0;
    uint32_t *rmask = nullptr, *mask = nullptr;
    bool mask_dirty = false;   // run-wide mask holds more than one round
    int rounds_done = 0;
    // marks of the current round / final list
    uint64_t *marks = nullptr;
    uint64_t marks_cap = 0, n_marks = 0;
    bool marks_valid = false;  // marks[] is the compaction of rmask
    bool rmask_sums_valid = false;  // block_sums / rmask_sums_n are those of the current rmask (left by the partitioned query's count)
    uint64_t rmask_sums_n = 0;
    uint64_t *block_sums = nullptr;
    uint64_t *scan_blocks = nullptr;  // scan2 per-chunk counts / offsets
    // exact filter table
    void *table = nullptr;
    uint64_t table_cap = 0, table_alloc = 0;
    // junction keys
    uint64_t *keys = nullptr;
    uint64_t n_keys = 0, keys_cap = 0;
    bool finalized = false;
    std::vector<uint64_t> keys_host;
    uint32_t *idtab = nullptr;
    uint64_t idtab_cap = 0;
    size_t idtab_bytes = 0;
    void *sort_scratch = nullptr;
    size_t sort_scratch_bytes = 0;
    // emit
    int64_t *emit_id = nullptr;
    uint64_t emit_cap = 0, n_emit = 0;
    // junction stream (bytes of the output file)
    uint32_t *stream_buf = nullptr;
    uint64_t stream_cap = 0, stream_bytes = 0;
    // per-rank formatting (tpc_emit_stream_partial -> tpc_emit_stream_part): kept between the two calls
    uint64_t *sp_rec = nullptr, *sp_vscan = nullptr, *sp_cnt = nullptr, *sp_lo = nullptr;
    uint32_t *sp_flags = nullptr;
    uint32_t sp_n_rec = 0;
    // scalars
    unsigned long long *counters = nullptr;  // device, 8 words
    unsigned long long *route_scratch = nullptr;  // device, 128 words: tpc_shard_route's per-owner counts and cursors
    uint64_t *sh_off = nullptr;  // device, [regions + 1]: offsets of the level-1 regions in a packed buffer (compacted exchange)
    size_t sh_off_bytes = 0;
    // options
    int opt_test_first = 0;
    int opt_insert_mode = 0;   // 0 auto, 1 direct atomicOr, 2 partitioned (LDS write-combining)
    int opt_slice_bits = 20;
    // partitioned insert
    bool filter_zero_pending = false;  // filter_reset requested, not yet materialised
    static constexpr int NPBUF = 19;  // 0 buf1, 1 cnt1, 2 buf2, 3 cnt2, 4 ovf, 5 ovf_cur, 6 surv, 7 surv_cur, 8 off2, 9 buf3, 10 cnt3, 11 off3; sharded contexts: 12 / 13 the insert's APPLY-side overflow list + cursor, 14 / 15 the query's hash-side list, 16 / 17 its apply-side list (sh_ovf); 18 the group boundaries of the 6-byte query (tpc_qpart6.h)
    void *pbuf[NPBUF] = {};   // shared by insert and query
    size_t pbytes[NPBUF] = {};
    std::vector<uint64_t> off2_uploaded, off3_uploaded;   // region offset tables currently in pbuf[8] / pbuf[11]
    int opt_part_levels = 0;   // 0 auto (three levels when L - slice_bits > 18), 2, 3
    int opt_shard_tight = 1;   // sharded passes: level-1 regions at the expected fill + 6 sigma (they travel whole); 0 = the one-GPU slack of 1.3 x
    // what the last insert / query actually ran (tpc_get_stat)
    int stat_path[2] = {0, 0};       // 1 direct kernel, 2 / 3 partitioned with that many levels (+10: partitioned, then completed by the direct kernel)
    int64_t stat_batches[2] = {0, 0};
    int stat_fmt[2] = {0, 0};        // entry format of the last partitioned insert (level 2: 0 = 32-bit, 3 = planar 24-bit) / query (0 = 8-byte, 6 = planar 48-bit)
    int64_t stat_filter2_retries = 0;  // exact-filter passes repeated with the full-size table (last tpc_pass2_filter)
    int64_t opt_part_min_tiles = 256;  // never cut batches smaller than this many 512-word tiles
    int64_t opt_part_budget = 0;  // bytes of partition buffers per batch; 0 = automatic (part_budget())
    int opt_query_mode = 0;    // 0 auto, 1 direct loads, 2 partitioned
    // deferred apply (insert and query of a round both in one tile batch): the insert stops after its level-2 binning and the
    // query's lookup kernel builds every filter slice itself (k_apply_lookup), so the filter is written once and never read back
    int opt_fuse = 1;
    bool pending_apply = false;   // the filter in HBM does not hold the last insert yet
    bool pending_fresh = false;
    bool pending_shard = false;   // ... and that insert was a sharded one (tpc_shard_apply*): its overflow entries wait in the pass' apply-side list
    TpcPartPlan pending_pl;
    void *ikeep[2] = {nullptr, nullptr};  // the insert's level-2 regions and counts while an apply is pending
    size_t ikeep_bytes[2] = {0, 0};
    uint64_t *ikeep_ovf = nullptr;        // ... and its overflow entries (the query reuses the overflow list), [0, n) as
    uint64_t ikeep_ovf_cap = 0;           //     produced, [cap, cap + n) grouped by slice for the fused kernel
    uint64_t pending_novf = 0;
    uint32_t *iovf_cnt = nullptr;         // [2 x slices] count and cursor of the grouping
    uint64_t *iovf_off = nullptr;         // [slices + 1]
    uint32_t iovf_slices = 0;
    int64_t stat_fused = 0;
    int64_t stat_query_overflow = 0;  // entries the last partitioned query batch handed to its overflow list
    int64_t stat_insert_overflow = 0; // ... and the last partitioned insert batch
    int64_t stat_pbuf_releases = 0;  // times the partition buffers were given back to let a second-pass allocation through
    // address-sharded filter (tpc_shard_*)
    uint32_t sh_rank = 0, sh_world = 1;
    TpcPartPlan sh_ipl;
    TpcQPlan sh_qpl;
    bool sh_have[2] = {false, false};
    uint64_t sh_per[2] = {0, 0}, sh_batches[2] = {0, 0};
    uint64_t sh_nsurv = 0;
    bool sh_defer = false;   // the insert plan at hand may leave its apply to the query's lookup (one batch, room for its level-2 regions)
    // Overflow lists of a sharded pass, two per pass (round 4): the hash kernels append to the PRODUCED list (tpc_shard_overflow_get
    // reads it), tpc_shard_overflow_set writes the gathered entries into the APPLIED list, which the apply side extends (level-2
    // losses) and consumes (k_part_ovf / k_q_ovf).  With one list per pass a hash running under the previous batch's exchange
    // (tpc_shard_hash_begin) would append to the list that exchange is about to overwrite.
    bool sh_ovf_set[2] = {false, false};   // tpc_shard_overflow_set was called since the last apply of the pass
    hipStream_t stream2 = nullptr;         // tpc_shard_hash_begin: the hash of a pass beside the main stream's work
    bool sh_async[2] = {false, false};     // a hash of the pass is in flight on stream2
    unsigned long long sh_ov_host[2][2] = {{0, 0}, {0, 0}};
    // combined exchange (tpc_combine_*, tpc_combine.hip): option replicate_filter keeps the WHOLE filter on every rank of a sharded
    // context (sh_world > 1); tpc_pass1_insert / tpc_pass1_query then run the one-GPU passes over this rank's chunk of the tiles
    int opt_replicate = 0;
    bool qb_valid = false;               // tpc_pass1_query_begin enqueued the first batch's hash and binning of the query of [qb_lo, qb_hi]
    uint64_t qb_lo = 0, qb_hi = 0;
    bool pending_lists = false;          // the pending (deferred) insert lives in imported set-bit lists (cmb_ls), not in level-2 regions
    TpcListSrc cmb_ls;                   // ... these (payload and directories are the caller's device buffers)
    TpcPartPlan cmb_geo;                 // slice geometry of the last deferred insert (tpc_combine_export / _merge / _import agree on it)
    bool cmb_have_geo = false;
    uint64_t *cmb_base = nullptr;        // device, [64]: first unit of every source block
    unsigned long long *cmb_cur = nullptr;  // device, [65]: units claimed per destination block, overflow flag
    // timing
    hipEvent_t ev0[TPC_K_COUNT]{}, ev1[TPC_K_COUNT]{};
    bool ev_used[TPC_K_COUNT]{};
    // debug switches, read from the environment once per context (not on every pass)
    bool dbg_ovf = false, dbg_phases = false, dbg_timing = false, no_lean = false;
    uint64_t reserve_text_bytes = 0;  // tpc_reserve ran before the upload: bytes the text will need, kept out of the buffer budget
};

namespace tpch {

int fail(tpc_ctx *c, int code, const char *fmt, ...);
TpcLaunch make_launch(const tpc_ctx *c);
TpcLaunch make_launch_periodic(const tpc_ctx *c);  // + the periodic-window masks (the hash kernels of tpc_pass1_insert / tpc_pass1_query only)
void ensure_periodic(tpc_ctx *c);
uint64_t rotln_host(uint64_t x, int L, int r);
hipError_t dev_malloc(tpc_ctx *c, void **p, size_t bytes);  // hipMalloc; gives the partition buffers back and tries again when it does not fit
int read_counter(tpc_ctx *c, int i, uint64_t *out);
int materialize_reset(tpc_ctx *c);      // a pending tpc_filter_reset becomes a real zero fill
int flush_pending_apply(tpc_ctx *c);    // the deferred apply of the last insert, for anything that reads the filter other than the fused lookup
double range_mass(const tpc_ctx *c, uint64_t lo, uint64_t hi);
bool ensure_pbuf(tpc_ctx *c, int i, size_t need);
bool release_partition_buffers(tpc_ctx *c);
uint64_t filter_words_for(int L, uint32_t world);
int64_t part_budget(const tpc_ctx *c);
uint64_t next_batches(uint64_t b);
uint64_t text_tiles512(const tpc_ctx *c);
bool replicated(const tpc_ctx *c);      // option replicate_filter on a rank of a sharded context
void pass_tiles(const tpc_ctx *c, uint64_t &t_begin, uint64_t &t_end);
uint64_t pass_tile_count(const tpc_ctx *c);
size_t qpart_need(const TpcQPlan &pl, int i);
bool part_hash_supported(const tpc_ctx *c);
bool plan_query(const tpc_ctx *c, uint64_t lo, uint64_t hi, bool gated, TpcQPlan &pl);
int compact_mask(tpc_ctx *c, const uint32_t *m);
void stream_part_release(tpc_ctx *c);   // tpc_capi_pass2.hip

#define HIPCHK(c, expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return tpch::fail(c, -10, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct Timed {
    tpc_ctx *c;
    int which;
    Timed(tpc_ctx *c_, int w) : c(c_), which(w) { (void)hipEventRecord(c->ev0[w], c->stream); }
    ~Timed() { (void)hipEventRecord(c->ev1[which], c->stream); c->ev_used[which] = true; }
};

template <typename T>
int ensure(tpc_ctx *c, T *&p, uint64_t &cap, uint64_t need)
{
    if (need <= cap && p) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    uint64_t n = need + need / 8 + 16;
    HIPCHK(c, dev_malloc(c, (void **)&p, n * sizeof(T)));
    cap = n;
    return 0;
}

}  // namespace tpch

using namespace tpch;
