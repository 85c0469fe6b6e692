// tpc_capi.hip -- the C-ABI of include/twopaco_hip.h: context, device memory, pass orchestration.
// No CPU fallback: every entry point needs a HIP device.
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

namespace {

bool replicated(const tpc_ctx *c);

int fail(tpc_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(c, -10, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct Timed {
    tpc_ctx *c;
    int which;
    Timed(tpc_ctx *c_, int w) : c(c_), which(w) { (void)hipEventRecord(c->ev0[w], c->stream); }
    ~Timed() { (void)hipEventRecord(c->ev1[which], c->stream); c->ev_used[which] = true; }
};

TpcLaunch make_launch(const tpc_ctx *c)
{
    TpcLaunch a;
    a.P = c->P; a.tab = c->tab; a.bases = c->bases; a.nmask = c->nmask; a.n_text = c->n_text;
    a.n_tiles = c->n_tiles; a.filter = c->filter; a.stream = c->stream;
    return a;
}

// make_launch + the periodic-window masks: for the hash kernels of tpc_pass1_insert / tpc_pass1_query only (their query copies the verdicts
// afterwards; the sharded calls mark through other kernels and keep every position probing)
TpcLaunch make_launch_periodic(const tpc_ctx *c)
{
    TpcLaunch a = make_launch(c);
    if (c->periodic && c->periodic_valid) {
        if (c->periodic_any_q) a.per_qs = c->periodic;
        if (c->periodic_any_i) a.per_i = c->periodic + (size_t)(1 + TPC_PER_PLANES) * c->n_words_alloc;
    }
    return a;
}

// The periodic-window masks of this text and k, built once: a detection launch first (0.6 ms on the 62-genome text, which has none: nothing
// is allocated then), the masks themselves only for a text that has such windows: [8][n_words_alloc] = per_qs, the six planes of the
// copy distance, per_i.  The one-GPU passes use them by default; a sharded context only when its host opted in (option shard_periodic_skip:
// that host calls tpc_shard_periodic_copy after a round's last batch).  A context that holds a window of the text builds the masks of its
// window (characters outside count as N: no skipping across its edges).  A failed allocation just leaves the feature off.
void ensure_periodic(tpc_ctx *c)
{
    if (c->periodic_valid || !c->opt_periodic || !c->bases || !c->have_params || c->n_words_alloc == 0) return;
    if ((c->sh_world > 1 || c->text_windowed) && !c->opt_shard_periodic && !replicated(c)) return;  // (a replicated pass copies the verdicts itself: tpc_pass1_query)
    TpcLaunch a = make_launch(c);
    const uint64_t w0 = c->text_windowed ? c->text_w0 : 0, w1 = c->text_windowed ? std::min(c->text_w1, c->n_words) : c->n_words;
    const uint64_t pos_hi = c->text_windowed ? c->text_w1 << 5 : ~0ull;
    uint32_t any[2] = {0, 0};
    uint32_t *flags = reinterpret_cast<uint32_t *>(c->counters + 6);  // (two words of the context's counter block)
    if (hipMemsetAsync(flags, 0, sizeof any, c->stream) != hipSuccess) return;
    tpc_launch_periodic_build(a, nullptr, nullptr, 0, nullptr, w0, w1, w0 << 5, pos_hi, flags);
    if (hipMemcpyAsync(any, flags, sizeof any, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return;
    c->periodic_any_q = any[0] != 0; c->periodic_any_i = any[1] != 0;
    if (any[0] || any[1]) {
        const size_t words = (size_t)(2 + TPC_PER_PLANES) * c->n_words_alloc;
        if (!c->periodic && hipMalloc((void **)&c->periodic, words * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); c->periodic = nullptr; return; }
        if (hipMemsetAsync(c->periodic, 0, words * sizeof(uint32_t), c->stream) != hipSuccess) return;
        tpc_launch_periodic_build(a, c->periodic, c->periodic + c->n_words_alloc, c->n_words_alloc, c->periodic + (size_t)(1 + TPC_PER_PLANES) * c->n_words_alloc, w0, w1, w0 << 5, pos_hi, flags);
        if (hipStreamSynchronize(c->stream) != hipSuccess) return;  // (a sharded hash may run on the second stream)
    }
    c->periodic_valid = true;
}

uint64_t rotln_host(uint64_t x, int L, int r)
{   // cyclichash.h:42-44
    if (r == 0) return x;
    const uint64_t maskn = (1ull << (L - r)) - 1ull;
    return ((x & maskn) << r) | (x >> (L - r));
}

// Device allocations of the second pass and the output (mark list, exact-filter table, keys, ids, junction stream).  The first
// pass' partition buffers stay allocated between rounds and may hold 60 % of the device (part_budget): when one of these
// allocations does not fit, they are given back (the next first pass allocates them again) and the allocation is repeated.
std::atomic<int> tpc_test_fail_mallocs{0};  // option "test_fail_mallocs" (tests only, process-wide; contexts of a multi-GPU host allocate from several threads): the next N first attempts fail
bool release_partition_buffers(tpc_ctx *c);
hipError_t dev_malloc(tpc_ctx *c, void **p, size_t bytes)
{
    hipError_t e = hipErrorOutOfMemory;
    if (tpc_test_fail_mallocs.load(std::memory_order_relaxed) > 0 && tpc_test_fail_mallocs.fetch_sub(1) > 0) *p = nullptr; else e = hipMalloc(p, bytes);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    if (!release_partition_buffers(c)) return e;
    e = hipMalloc(p, bytes);
    if (e != hipSuccess) *p = nullptr;
    return e;
}

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

int read_counter(tpc_ctx *c, int i, uint64_t *out)
{
    unsigned long long v = 0;
    HIPCHK(c, hipMemcpyAsync(&v, c->counters + i, sizeof v, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *out = v;
    return 0;
}

int flush_pending_apply(tpc_ctx *c);
bool ensure_pbuf(tpc_ctx *c, int i, size_t need);
void stream_part_release(tpc_ctx *c);

int materialize_reset(tpc_ctx *c)
{   // a pending tpc_filter_reset becomes a real zero fill before anything reads the filter
    { int rc0 = flush_pending_apply(c); if (rc0) return rc0; }
    if (!c->filter_zero_pending) return 0;
    Timed t(c, TPC_K_FILTER_RESET);
    HIPCHK(c, hipMemsetAsync(c->filter, 0, c->filter_words * sizeof(uint32_t), c->stream));
    c->filter_zero_pending = false;
    return 0;
}

int flush_pending_apply(tpc_ctx *c)
{   // the deferred apply of the last insert, for anything that reads or extends the filter other than the fused lookup
    if (!c->pending_apply) return 0;
    c->pending_apply = false;
    if (c->pending_lists) {  // the combined exchange's imported lists: every slice built from them and written out (no lookup rides along)
        c->pending_lists = false;
        Timed t(c, TPC_K_FUSED);
        const TpcPartPlan &g = c->pending_pl;
        if (tpc_launch_slice_combine(make_launch(c), g.slice_bits, g.b1, g.b2, g.perm_mult, g.perm_inv, nullptr, nullptr, nullptr, c->cmb_ls, true, c->pending_fresh, nullptr, 0, 1))
            return fail(c, -1, "apply launch failed");
        HIPCHK(c, hipGetLastError());
        return 0;
    }
    if (c->pending_shard) {  // the sharded insert: level-2 regions kept aside, the overflow entries still in the apply-side list the plan points at
        c->pending_shard = false;
        Timed t(c, TPC_K_SHARD_APPLY);
        if (tpc_launch_insert_part_apply_only(make_launch(c), c->pending_pl, c->pending_fresh)) return fail(c, -1, "apply launch failed");
        HIPCHK(c, hipGetLastError());
        return 0;
    }
    Timed t(c, TPC_K_FUSED);
    // the insert's overflow entries were set aside (the list buffer is shared with the query): put them back for k_part_ovf.
    // The shared buffers may have been reallocated since the insert (ensure_pbuf for a query plan of another size): take the
    // pointers from the context as it is now, never the ones saved with the plan
    const size_t ovf_need = std::max<size_t>((size_t)c->pending_novf * sizeof(uint64_t), sizeof(uint64_t));
    if (!ensure_pbuf(c, 4, ovf_need) || !ensure_pbuf(c, 5, 32 * sizeof(unsigned long long))) return fail(c, -10, "out of device memory for the deferred apply's overflow list");
    c->pending_pl.ovf = (uint64_t *)c->pbuf[4];
    c->pending_pl.ovf_cur = (unsigned long long *)c->pbuf[5];
    c->pending_pl.ovf_cap = std::min<uint64_t>(c->pending_pl.ovf_cap, c->pbytes[4] / sizeof(uint64_t));
    const unsigned long long cur[2] = {c->pending_novf, 0};
    if (c->pending_novf) HIPCHK(c, hipMemcpyAsync(c->pending_pl.ovf, c->ikeep_ovf, c->pending_novf * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->pending_pl.ovf_cur, cur, sizeof cur, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // `cur` is on this stack frame
    if (tpc_launch_insert_part_apply_only(make_launch(c), c->pending_pl, c->pending_fresh)) return fail(c, -1, "apply launch failed");
    HIPCHK(c, hipGetLastError());
    return 0;
}

// Share of the vertices whose hash min(H, H') falls into [lo, hi]: the density of the minimum of two
// uniform L-bit values is 2(1 - x), so the mass below x is 1 - (1 - x)^2.
double range_mass(const tpc_ctx *c, uint64_t lo, uint64_t hi)
{
    const double size = (double)(c->P.lmask) + 1.0;
    auto below = [size](double x) { x = std::min(1.0, std::max(0.0, x / size)); return 1.0 - (1.0 - x) * (1.0 - x); };
    return std::min(1.0, std::max(0.0, below((double)hi + 1.0) - below((double)lo)));
}

bool ensure_pbuf(tpc_ctx *c, int i, size_t need)
{
    if (need <= c->pbytes[i]) return true;
    if (c->pbuf[i]) (void)hipFree(c->pbuf[i]);
    c->pbuf[i] = nullptr; c->pbytes[i] = 0;
    if (i == 8) c->off2_uploaded.clear();
    if (i == 11) c->off3_uploaded.clear();
    if (hipMalloc(&c->pbuf[i], need) != hipSuccess) { (void)hipGetLastError(); return false; }
    c->pbytes[i] = need;
    return true;
}

bool release_partition_buffers(tpc_ctx *c)
{   // false: nothing to give back (or a deferred insert still lives in them)
    if (c->pending_apply) return false;
    size_t held = 0;
    for (size_t b : c->pbytes) held += b;
    for (void *p : c->ikeep) if (p) held += 1;
    if (!held) return false;
    (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < tpc_ctx::NPBUF; i++) {
        if (c->pbuf[i]) (void)hipFree(c->pbuf[i]);
        c->pbuf[i] = nullptr; c->pbytes[i] = 0;
    }
    for (void *&p : c->ikeep) { if (p) (void)hipFree(p); p = nullptr; }
    for (size_t &b : c->ikeep_bytes) b = 0;
    c->off2_uploaded.clear();
    c->off3_uploaded.clear();
    c->sh_have[0] = c->sh_have[1] = false;  // the sharded plans hold pointers into the buffers just freed: tpc_shard_plan again before any tpc_shard_* call
    c->stat_pbuf_releases++;
    return true;
}

uint64_t filter_words_for(int L, uint32_t world)
{   // concurrentbitvector.cpp:12; a shard holds 2^L/world bits
    return std::max<uint64_t>(1, ((1ull << L) >> 5) / world) + 1;
}

// Buffer budget of one tile batch.  Automatic: 40 GiB (the first ~48 GiB of hipMalloc are cheap on this system), more when
// the device has room -- fewer batches mean fewer passes over the filter (60 % of what is free plus what is already held;
// the rest stays for the second pass's table, the stream and the overflow lists).
int64_t part_budget(const tpc_ctx *c)
{
    if (c->opt_part_budget > 0) return c->opt_part_budget;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return (int64_t)40 << 30; }
    size_t held = 0;
    for (size_t b : c->pbytes) held += b;
    // tpc_reserve before tpc_seq_upload: the text (bases, N mask, two candidate masks: 20 bytes per 32 positions) is not resident
    // yet and must not be counted as free.  The 40 GiB floor only holds while that much is really there.
    const double avail = std::max(0.0, (double)(free_b + held) - (double)(c->bases ? 0 : c->reserve_text_bytes));
    return std::max<int64_t>(std::min<int64_t>((int64_t)40 << 30, (int64_t)(avail * 0.90)), (int64_t)(avail * 0.60));
}

// Batch counts tried in turn: every count up to 8, then steps of ~1/8 -- each batch streams the whole filter once, so a text that
// needs 17 batches should not be cut into 32.
uint64_t next_batches(uint64_t b) { return b + std::max<uint64_t>(1, b / 8); }

uint64_t text_tiles512(const tpc_ctx *c) { return (c->n_text / TPC_RUN + 512) / 512; }

// The 512-word tiles tpc_pass1_insert / tpc_pass1_query hash: all of them, or -- option replicate_filter on a rank of a sharded
// context -- this rank's chunk (the split of tpc_shard_hash and tpc_shard_chunk; possibly empty on a tiny text)
bool replicated(const tpc_ctx *c) { return c->opt_replicate && c->sh_world > 1; }
void pass_tiles(const tpc_ctx *c, uint64_t &t_begin, uint64_t &t_end)
{
    const uint64_t tiles = text_tiles512(c);
    t_begin = 0; t_end = tiles;
    if (replicated(c)) {
        const uint64_t chunk = (tiles + c->sh_world - 1) / c->sh_world;
        t_begin = std::min<uint64_t>(tiles, (uint64_t)c->sh_rank * chunk);
        t_end = std::min<uint64_t>(tiles, t_begin + chunk);
    }
}
uint64_t pass_tile_count(const tpc_ctx *c) { uint64_t a, b; pass_tiles(c, a, b); return std::max<uint64_t>(1, b - a); }

// Bytes of the single-GPU query's buffer i: with three levels the level-3 regions take the place of the level-1 buffer (dead once
// level 2 has split it), as in the insert -- a batch holds two of the three large buffers, not three.
size_t qpart_need(const TpcQPlan &pl, int i)
{
    if (i == 0) return std::max(tpc_qpart_bytes(pl, 0), tpc_qpart_bytes(pl, 9));
    if (i == 9) return 0;
    return tpc_qpart_bytes(pl, i);
}

// Can the partitioned (LDS write-combining) passes hash with this context's parameters?  One gate for tpc_pass1_insert,
// tpc_pass1_query, tpc_reserve and tpc_shard_plan: the rolling kernels exist for 1..16 functions; 9..16 run on the
// instruction-lean hash kernel only (tpc_partition.hip:launch_hash_q), which takes a slice index of at most 24 bits and can be
// switched off (TPC_NO_LEAN, measurements); the test hook that forces the closed-form kernels means the direct path too.
bool part_hash_supported(const tpc_ctx *c)
{
    if (c->P.q > TPC_KERNEL_MAXQ || tpc_test_force_anyq) return false;
    if (c->P.q > 8 && (c->P.L - c->opt_slice_bits > 24 || c->no_lean)) return false;
    return true;
}

// Tile batching of the partitioned query under the buffer budget; false: use the direct kernel.
bool plan_query(const tpc_ctx *c, uint64_t lo, uint64_t hi, bool gated, TpcQPlan &pl)
{
    const uint64_t tiles = pass_tile_count(c);
    if (c->P.q > TPC_KERNEL_MAXQ || tpc_test_force_anyq) return false;  // the verification kernels are the rolling ones
    if (!replicated(c) && (c->opt_query_mode == 1 || (c->opt_query_mode == 0 && c->P.L < 28))) return false;  // small filters are cache resident: direct loads win
    const int64_t budget = part_budget(c);
    for (uint64_t batches = 1;; batches = next_batches(batches)) {
        const uint64_t per = (tiles + batches - 1) / batches;
        const bool ok = tpc_qpart_plan(c->P.L, c->opt_slice_bits, per, gated ? std::min(1.0, range_mass(c, lo, hi) * 1.15) : 1.0, pl, c->opt_part_levels);
        if (!ok && per * 512 * TPC_RUN <= (1ull << 30)) return false;  // geometry unsupported (not a size problem)
        if (ok && ((int64_t)(qpart_need(pl, 0) + tpc_qpart_bytes(pl, 2)) <= budget || (int64_t)per <= c->opt_part_min_tiles)) {
            // Every batch streams the whole filter through LDS once.  That pays while a slice sees a few thousand probes per
            // batch; below that (sparse huge filters: f >= 39 on the 62-genome input) the direct loads are cheaper.
            // tools/large_filter_bench.py: f=38 3.5 k probes per slice and batch 52 vs 61 ms, f=39 1.8 k 64 vs 61, f=40 0.9 k 99 vs 63.
            const double per_slice = 6.0 * (gated ? range_mass(c, lo, hi) : 1.0) * (double)per * 512 * TPC_RUN / (double)(1ull << (c->P.L - pl.slice_bits));
            if (c->opt_query_mode == 0 && batches > 1 && per_slice < 2500.0 && !replicated(c)) return false;
            return true;
        }
        if (per <= 1) return false;
    }
}

int compact_mask(tpc_ctx *c, const uint32_t *m)
{   // ordered list of the set bits of m -> c->marks / c->n_marks
    Timed t(c, TPC_K_COMPACT);
    uint64_t n = 0;
    int rc = 0;
    if (m == c->rmask && c->rmask_sums_valid) {
        n = c->rmask_sums_n;  // the query has just counted this mask: its block sums are still in place
    } else {
        c->rmask_sums_valid = false;
        tpc_launch_mask_count(c->stream, m, c->n_words, c->block_sums, c->counters + 2);
        if ((rc = read_counter(c, 2, &n))) return rc;
    }
    rc = ensure(c, c->marks, c->marks_cap, n);
    if (rc) return rc;
    if (n) tpc_launch_mask_scatter(c->stream, m, c->n_words, c->block_sums, c->marks);
    c->n_marks = n;
    return 0;
}

}  // namespace

extern "C" {

int tpc_ctx_create(int device, tpc_ctx **out)
{
    if (!out) return -1;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return -2;  // no GPU: fail loudly, no CPU path
    if (device < 0 || device >= n) return -3;
    if (hipSetDevice(device) != hipSuccess) return -4;
    tpc_ctx *c = new tpc_ctx();
    c->device = device;
    c->dbg_ovf = getenv("TPC_DEBUG_OVF") != nullptr; c->dbg_phases = getenv("TPC_PROFILE_PHASES") != nullptr; c->dbg_timing = getenv("TWOPACO_TIMING") != nullptr; c->no_lean = TpcEnv::get().no_lean;
    if (const char *e = getenv("TPC_SHARD_TIGHT")) c->opt_shard_tight = atoi(e) ? 1 : 0;  // measurements: the one-GPU region slack on a sharded context
    if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return -5; }
    for (int i = 0; i < TPC_K_COUNT; i++) {
        if (hipEventCreate(&c->ev0[i]) != hipSuccess || hipEventCreate(&c->ev1[i]) != hipSuccess) { delete c; return -5; }
    }
    if (hipMalloc((void **)&c->tab, TPC_TAB_WORDS * sizeof(uint64_t)) != hipSuccess ||
        hipMalloc((void **)&c->counters, 8 * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void **)&c->route_scratch, 128 * sizeof(unsigned long long)) != hipSuccess) { delete c; return -6; }
    *out = c;
    return 0;
}

void tpc_ctx_destroy(tpc_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    void *ptrs[] = { c->tab, c->bases_alloc, c->nmask_alloc, c->filter, c->rmask, c->mask, c->marks, c->block_sums, c->table,
                     c->keys, c->idtab, c->emit_id, c->stream_buf, c->counters, c->route_scratch, c->sh_off, c->scan_blocks, c->sort_scratch };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    stream_part_release(c);
    for (void *p : c->pbuf) if (p) (void)hipFree(p);
    for (void *p : c->ikeep) if (p) (void)hipFree(p);
    if (c->periodic) (void)hipFree(c->periodic);
    if (c->cmb_base) (void)hipFree(c->cmb_base);
    if (c->cmb_cur) (void)hipFree(c->cmb_cur);
    if (c->ikeep_ovf) (void)hipFree(c->ikeep_ovf);
    if (c->iovf_cnt) (void)hipFree(c->iovf_cnt);
    if (c->iovf_off) (void)hipFree(c->iovf_off);
    for (int i = 0; i < TPC_K_COUNT; i++) { if (c->ev0[i]) (void)hipEventDestroy(c->ev0[i]); if (c->ev1[i]) (void)hipEventDestroy(c->ev1[i]); }
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *tpc_last_error(const tpc_ctx *c) { return c ? c->err.c_str() : "null context"; }

int tpc_set_option(tpc_ctx *c, const char *name, int64_t value)
{
    if (!c || !name) return -1;
    if (!strcmp(name, "insert_test_first")) { c->opt_test_first = value != 0; return 0; }
    if (!strcmp(name, "insert_mode")) { c->opt_insert_mode = (int)value; return 0; }
    if (!strcmp(name, "slice_bits")) { c->opt_slice_bits = (int)value; return 0; }
    if (!strcmp(name, "query_mode")) { c->opt_query_mode = (int)value; return 0; }
    if (!strcmp(name, "part_budget_bytes")) { c->opt_part_budget = value; return 0; }
    if (!strcmp(name, "part_levels")) { c->opt_part_levels = (int)value; return 0; }
    if (!strcmp(name, "shard_tight_regions")) { c->opt_shard_tight = value ? 1 : 0; c->sh_have[0] = c->sh_have[1] = false; return 0; }
    if (!strcmp(name, "part_min_tiles")) { c->opt_part_min_tiles = value < 1 ? 1 : value; return 0; }
    if (!strcmp(name, "fuse_apply_lookup")) { c->opt_fuse = value != 0; return 0; }
    if (!strcmp(name, "test_q6_pb2")) { tpc_test_q6_pb2 = (int)value; return 0; }  // process-wide, tests only
    if (!strcmp(name, "insert_entry_fmt")) { tpc_test_insert_p3 = value == 3; return 0; }  // process-wide; 3 = blocked 24-bit level-2 insert entries (off by default: tpc_partition.hip)
    if (!strcmp(name, "shard_periodic_skip")) { c->opt_shard_periodic = value != 0; return 0; }  // the host promises tpc_shard_periodic_copy after every round's last query batch
    if (!strcmp(name, "periodic_skip")) { c->opt_periodic = value != 0; if (!value) c->periodic_valid = false; return 0; }  // 0: every position inserts and probes for itself
    if (!strcmp(name, "test_tight_pinch")) { tpc_test_tight_pinch = (int)value; c->sh_have[0] = c->sh_have[1] = false; return 0; }  // process-wide, tests only
    if (!strcmp(name, "test_sched_cap")) { tpc_test_sched_cap = value > 0 ? (uint32_t)value : 0; return 0; }  // process-wide, tests only
    if (!strcmp(name, "text_window")) { c->opt_text_window = value != 0; return 0; }
    if (!strcmp(name, "replicate_filter")) { c->opt_replicate = value != 0; return 0; }  // before tpc_shard_config / tpc_set_params
    if (!strcmp(name, "test_force_anyq")) { tpc_test_force_anyq = value != 0; return 0; }  // process-wide, tests only
    if (!strcmp(name, "test_fail_mallocs")) { tpc_test_fail_mallocs.store(value > 0 ? (int)value : 0); return 0; }  // process-wide, tests only
    return fail(c, -1, "unknown option %s", name);
}

int tpc_preload(int device)
{   // needs no context and no stream (a second queue costs ~20 ms to create): attribute queries load the code objects
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return -10; }  // the error is this call's, not the next one's
    const bool timing = getenv("TWOPACO_TIMING") != nullptr;
    // in the order a run needs them: partitioned insert, partitioned query, second pass, junction stream.  Not the direct kernels
    // and the split pass (tpc_pass1.hip: a one-round run on a large filter never launches them), nor the level-1 insert kernels of
    // q != 5 (tpc_partition_q.o) or the long-key second pass: the runtime loads those when a run first launches one of them.
    int (*const warm[4])() = { tpc_warm_partition, tpc_warm_qpartition, tpc_warm_pass2, tpc_warm_stream };
    const char *const name[4] = { "partition", "qpartition", "pass2", "stream" };
    // one after the other: loading them from several host threads at once is no faster (the runtime serialises it) and was seen
    // to stall device allocations made meanwhile by ~0.5 s
    for (int i = 0; i < 4; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        if (warm[i]() != 0) return -10;
        if (timing) fprintf(stderr, "[timing]     code object %s: %.1f ms\n", name[i], std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    return 0;
}

int tpc_warmup(tpc_ctx *c)
{   // touches no context state: may run beside tpc_seq_upload on another host thread
    if (!c) return -1;
    return tpc_preload(c->device);
}

int64_t tpc_get_stat(const tpc_ctx *c, const char *name)
{
    if (!c || !name) return -1;
    if (!strcmp(name, "insert_path")) return c->stat_path[0];
    if (!strcmp(name, "query_path")) return c->stat_path[1];
    if (!strcmp(name, "insert_entry_fmt")) return c->stat_fmt[0];
    if (!strcmp(name, "query_entry_fmt")) return c->stat_fmt[1];
    if (!strcmp(name, "insert_batches")) return c->stat_batches[0];
    if (!strcmp(name, "query_batches")) return c->stat_batches[1];
    if (!strcmp(name, "filter2_retries")) return c->stat_filter2_retries;
    if (!strcmp(name, "fused_lookups")) return c->stat_fused;
    if (!strcmp(name, "query_overflow_entries")) return c->stat_query_overflow;
    if (!strcmp(name, "insert_overflow_entries")) return c->stat_insert_overflow;
    if (!strcmp(name, "periodic_skip")) return c->periodic_valid ? 1 : 0;
    if (!strcmp(name, "pbuf_releases")) return c->stat_pbuf_releases;
    if (!strcmp(name, "text_words")) return (int64_t)(c->text_w1 - c->text_w0);  // packed words of the text this context holds
    if (!strcmp(name, "device_free_bytes") || !strcmp(name, "device_total_bytes")) {  // hipMemGetInfo of the context's device, now
        size_t free_b = 0, total_b = 0;
        if (hipSetDevice(c->device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return -1; }
        return (int64_t)(name[7] == 'f' ? free_b : total_b);
    }
    if (!strcmp(name, "round_marks")) return c->marks_valid ? (int64_t)c->n_marks : -1;  // set bits of the round mask (after tpc_pass2_filter)
    return -1;
}

int tpc_set_params(tpc_ctx *c, int k, int L, int q, const uint64_t *seed_table)
{
    if (!c || !seed_table) return -1;
    if (q < 1 || q > TPC_MAX_Q) return fail(c, -1, "q=%d unsupported (1..%d)", q, TPC_MAX_Q);
    if (L < 2 || L > 62) return fail(c, -1, "filter bits L=%d unsupported (2..62)", L);
    if (k < 1) return fail(c, -1, "k must be positive");
    const int C = (k + 4 + 31) / 32;  // CalculateNeededCapacity, candidateoccurence.h:129-133
    if (C >= 20)                      // vertexenumerator.cpp:56-70
        return fail(c, -1, "The value of K is too big. Please refer to documentaion how to increase the max supported value of K.");
    HIPCHK(c, hipSetDevice(c->device));
    c->P.k = k; c->P.L = L; c->P.q = q; c->P.rk = k % L; c->P.lmask = (1ull << L) - 1ull;
    c->C = C;
    memset(c->tab_host, 0, sizeof c->tab_host);
    for (int i = 0; i < q; i++)
        for (int ch = 0; ch < 5; ch++) {
            const uint64_t h = seed_table[i * 5 + ch];
            if (h > c->P.lmask) return fail(c, -1, "seed table entry exceeds L bits");
            c->tab_host[i * 5 + ch] = h;
            c->tab_host[TPC_TAB_HK + i * 5 + ch] = rotln_host(h, L, c->P.rk);
        }
    HIPCHK(c, hipMemcpy(c->tab, c->tab_host, sizeof c->tab_host, hipMemcpyHostToDevice));
    const uint64_t fw = filter_words_for(L, c->opt_replicate ? 1 : c->sh_world);
    c->sh_have[0] = c->sh_have[1] = false;
    c->pending_lists = false; c->cmb_have_geo = false;
    if (fw != c->filter_words) {
        if (c->filter) (void)hipFree(c->filter);
        c->filter = nullptr;
        HIPCHK(c, hipMalloc((void **)&c->filter, fw * sizeof(uint32_t)));
        c->filter_words = fw;
    }
    c->have_params = true;
    c->periodic_valid = false;  // (k)
    c->pending_apply = false;
    c->n_keys = 0; c->finalized = false; c->rounds_done = 0; c->mask_dirty = false; c->marks_valid = false; c->rmask_sums_valid = false;
    return 0;
}

int tpc_seq_upload(tpc_ctx *c, const uint64_t *bases, const uint32_t *nmask, uint64_t n_text)
{
    if (!c || !bases || !nmask || n_text < 2) return fail(c, -1, "bad text");
    HIPCHK(c, hipSetDevice(c->device));
    const auto t_begin = std::chrono::steady_clock::now();
    const uint64_t nw = (n_text + 31) / 32;
    if (!((nmask[0] & 1u) && ((nmask[(n_text - 1) >> 5] >> ((n_text - 1) & 31)) & 1u)))
        return fail(c, -1, "text must start and end with the N separator");
    const uint64_t tiles = (nw + TPC_TILE_THREADS - 1) / TPC_TILE_THREADS;
    // whole 512-word tiles of text_tiles512() (one more than nw needs when n_text is a multiple of 16384: the kernels of the
    // partitioned passes stage words [tile * 512 - 1, tile * 512 + 512 + xw) and k_q_hash stores a tile's rmask unguarded)
    const uint64_t alloc = ((n_text / TPC_RUN + 512) / 512) * 512 + TPC_XW_MAX + 2;
    // window (option text_window on a sharded context): only the words of this rank's chunk of tiles, plus the halo a tile
    // stages (one word before, TPC_XW_MAX + 1 after).  Everything a windowed context may run reads inside it: the hash
    // kernels over its own tiles and the verification of survivors at its own positions.
    uint64_t w0 = 0, w1 = alloc;
    const bool windowed = c->opt_text_window && c->sh_world > 1;
    if (windowed) {
        const uint64_t t512 = (n_text / TPC_RUN + 512) / 512, chunk = (t512 + c->sh_world - 1) / c->sh_world;
        const uint64_t ta = std::min<uint64_t>(t512, (uint64_t)c->sh_rank * chunk), tb = std::min<uint64_t>(t512, ta + chunk);
        w0 = ta * 512 > 0 ? ta * 512 - 1 : 0;
        w1 = std::min<uint64_t>(alloc, tb * 512 + TPC_XW_MAX + 2);
        if (w1 <= w0) w1 = w0 + 1;
    }
    for (void *p : { (void *)c->bases_alloc, (void *)c->nmask_alloc, (void *)c->rmask, (void *)c->mask, (void *)c->block_sums, (void *)c->periodic })
        if (p) (void)hipFree(p);
    c->periodic = nullptr; c->periodic_valid = false;
    c->bases = nullptr; c->nmask = nullptr; c->bases_alloc = nullptr; c->nmask_alloc = nullptr; c->rmask = nullptr; c->mask = nullptr; c->block_sums = nullptr;
    const uint64_t wn = w1 - w0;
    // through dev_malloc: a reservation made before the upload (tpc_reserve) is given back when the text does not fit beside it
    HIPCHK(c, dev_malloc(c, (void **)&c->bases_alloc, wn * sizeof(uint64_t)));
    HIPCHK(c, dev_malloc(c, (void **)&c->nmask_alloc, wn * sizeof(uint32_t)));
    HIPCHK(c, dev_malloc(c, (void **)&c->rmask, alloc * sizeof(uint32_t)));
    HIPCHK(c, dev_malloc(c, (void **)&c->mask, alloc * sizeof(uint32_t)));
    HIPCHK(c, dev_malloc(c, (void **)&c->block_sums, (alloc / 256 + 2 + 64) * sizeof(uint64_t)));  // (+ 64: tpc_launch_mask_count's scratch behind the sums)
    HIPCHK(c, hipMemsetAsync(c->bases_alloc, 0, wn * sizeof(uint64_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->nmask_alloc, 0xFF, wn * sizeof(uint32_t), c->stream));  // padding = N
    HIPCHK(c, hipMemsetAsync(c->rmask, 0, alloc * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->mask, 0, alloc * sizeof(uint32_t), c->stream));
    const uint64_t ce = std::min(nw, w1);  // host words [w0, ce) exist
    uint32_t last_word = 0;
    if (ce > w0) {
        HIPCHK(c, hipMemcpyAsync(c->bases_alloc, bases + w0, (ce - w0) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->nmask_alloc, nmask + w0, (ce - w0) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        // the last word may be partial: mark the bits past n_text as N (one word patched on the device; the caller's 39 MB of
        // mask used to be copied on the host for this)
        if ((n_text & 31) && ce == nw) {
            last_word = nmask[nw - 1] | (~0u << (n_text & 31));
            HIPCHK(c, hipMemcpyAsync(c->nmask_alloc + (nw - 1 - w0), &last_word, sizeof last_word, hipMemcpyHostToDevice, c->stream));
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->dbg_timing)
        fprintf(stderr, "[timing]   tpc_seq_upload (allocations, %.0f MB host to device): %.1f ms\n", (double)(ce > w0 ? (ce - w0) * 12 : 0) / 1e6,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    c->bases = c->bases_alloc - w0;
    c->nmask = c->nmask_alloc - w0;
    c->text_windowed = windowed; c->text_w0 = w0; c->text_w1 = w1;
    c->pending_apply = false;
    c->n_text = n_text; c->n_words = (n_text >> 5) + 1; c->n_words_alloc = alloc; c->n_tiles = tiles;
    c->n_keys = 0; c->finalized = false; c->rounds_done = 0; c->mask_dirty = false; c->marks_valid = false; c->rmask_sums_valid = false;
    return 0;
}

int tpc_run_begin(tpc_ctx *c)
{
    if (!c) return -1;
    c->n_keys = 0; c->finalized = false; c->rounds_done = 0; c->mask_dirty = false; c->marks_valid = false; c->rmask_sums_valid = false;
    c->n_marks = 0; c->n_emit = 0; c->keys_host.clear();
    return 0;
}

int tpc_filter_reset(tpc_ctx *c)
{
    if (!c || !c->have_params) return fail(c, -1, "set_params first");
    // Lazy: the partitioned insert writes every slice of the filter itself, so the zero fill is
    // only materialised (hipMemsetAsync, timed as TPC_K_FILTER_RESET) when something else needs it.
    c->filter_zero_pending = true;
    c->pending_apply = false;  // an insert nobody looked at is forgotten with the filter
    c->pending_shard = false;
    c->pending_lists = false;
    c->ev_used[TPC_K_FILTER_RESET] = false;
    return 0;
}

int tpc_pass1_insert(tpc_ctx *c, uint64_t lo, uint64_t hi, uint64_t *n_kmers)
{
    if (!c || !c->have_params || !c->bases) return fail(c, -1, "set_params and seq_upload first");
    if (c->sh_world > 1 && !c->opt_replicate) return fail(c, -1, "the filter is sharded: use tpc_shard_hash / tpc_shard_apply");
    HIPCHK(c, hipSetDevice(c->device));
    const bool gated = !(lo == 0 && hi >= c->P.lmask);
    { int rc0 = flush_pending_apply(c); if (rc0) return rc0; }
    ensure_periodic(c);
    if (n_kmers) HIPCHK(c, hipMemsetAsync(c->counters, 0, sizeof(unsigned long long), c->stream));
    uint64_t t_begin, t_end;
    pass_tiles(c, t_begin, t_end);  // every tile, or this rank's chunk (option replicate_filter on a sharded context)
    TpcPartPlan pl;
    const double m_ins = gated ? range_mass(c, lo, hi) : 1.0;
    const double ins_frac = gated ? std::min(1.0, (1.0 - (1.0 - m_ins) * (1.0 - m_ins)) * 1.15) : 1.0;  // either endpoint in range
    const uint64_t tiles = pass_tile_count(c);
    uint64_t batches = 1;
    bool defer = false;
    bool part = ((c->opt_insert_mode != 1 && !(c->opt_insert_mode == 0 && c->P.L < 28)) || replicated(c)) && part_hash_supported(c);  // small filters: the direct kernel is as fast
    // (more than 16 hash functions: the closed-form direct kernels take a rank's chunk as a range of positions; the ranks' dense filters are OR-reduced)
    if (replicated(c) && !part && c->P.q <= TPC_KERNEL_MAXQ)
        return fail(c, -1, "a replicated multi-GPU pass needs the partitioned hash kernels (q=%d, L=%d, slice_bits=%d are outside what they cover)", c->P.q, c->P.L, c->opt_slice_bits);
    if (part) {
        // as few batches of tiles as the buffer budget allows
        const int64_t budget = part_budget(c);
        for (;; batches = next_batches(batches)) {
            const uint64_t per = (tiles + batches - 1) / batches;
            if (!tpc_part_plan(c->P.L, c->P.q, c->opt_slice_bits, per, ins_frac, pl, c->opt_part_levels)) { part = false; break; }
            if ((int64_t)(std::max(tpc_part_buf1_bytes(pl), tpc_part_buf3_bytes(pl)) + tpc_part_buf2_bytes(pl)) <= budget || (int64_t)per <= c->opt_part_min_tiles) break;
        }
        // A batch after the first loads and stores every filter slice (2 x 2^L/8 bytes at ~5 TB/s) to save ~40 ps per address
        // against the direct atomics: worth it only above ~2^slice_bits/800 addresses per slice and batch.
        if (part && c->opt_insert_mode == 0 && batches > 1 && !replicated(c)) {
            const double per_slice = (double)c->P.q * ins_frac * (double)pl.n_tiles * 512 * TPC_RUN / (double)(1ull << (c->P.L - pl.slice_bits));
            if (per_slice < (double)(1ull << pl.slice_bits) / 800.0) part = false;
        }
    }
    if (part) {
        // three levels: the level-3 regions take the level-1 buffer's place (its entries are dead once level 2 has split them),
        // so a batch holds two of the three buffers at a time -- fewer batches, each of which streams the whole filter
        size_t need[tpc_ctx::NPBUF] = { std::max(tpc_part_buf1_bytes(pl), tpc_part_buf3_bytes(pl)), tpc_part_cnt1_bytes(pl), tpc_part_buf2_bytes(pl), tpc_part_cnt2_bytes(pl),
                                        pl.ovf_cap * sizeof(uint64_t), 32 * sizeof(unsigned long long), 0, 0, 0, 0, tpc_part_cnt3_bytes(pl), 0 };
        // the query of the same round shares these buffers: size them for both now (one allocation, not free + grow)
        TpcQPlan qpl;
        const bool qpart = plan_query(c, lo, hi, gated, qpl);
        for (int i = 0; i < tpc_ctx::NPBUF && qpart; i++) need[i] = std::max(need[i], qpart_need(qpl, i));
        for (int i = 0; i < tpc_ctx::NPBUF && part; i++) if (need[i]) part = ensure_pbuf(c, i, need[i]);  // not enough HBM: direct path
        // deferred apply: the insert in one batch, the query partitioned with the same slice geometry (its FIRST batch then
        // builds the slices), and room for the insert's level-2 regions beside the query's buffers
        defer = part && c->opt_fuse && batches == 1 && qpart && pl.b3 == 0 && qpl.b3 == 0 && qpl.slice_bits == pl.slice_bits &&
                qpl.b1 == pl.b1 && qpl.b2 == pl.b2 && (qpl.fmt == 6 || pl.fmt2 == 0);  // (the 8-byte lookup reads 32-bit insert entries only)
        if (defer) {
            const size_t want[2] = { tpc_part_buf2_bytes(pl), tpc_part_cnt2_bytes(pl) };
            for (int i = 0; i < 2 && defer; i++) {
                if (want[i] <= c->ikeep_bytes[i]) continue;
                if (c->ikeep[i]) (void)hipFree(c->ikeep[i]);
                c->ikeep[i] = nullptr; c->ikeep_bytes[i] = 0;
                if (hipMalloc(&c->ikeep[i], want[i]) != hipSuccess) { (void)hipGetLastError(); defer = false; break; }
                c->ikeep_bytes[i] = want[i];
            }
        }
    }
    if (part) {
        pl.buf1 = (uint32_t *)c->pbuf[0]; pl.cnt1 = (uint32_t *)c->pbuf[1]; pl.buf2 = (uint32_t *)c->pbuf[2]; pl.cnt2 = (uint32_t *)c->pbuf[3];
        pl.ovf = (uint64_t *)c->pbuf[4]; pl.ovf_cur = (unsigned long long *)c->pbuf[5];
        pl.buf3 = (uint32_t *)c->pbuf[0]; pl.cnt3 = (uint32_t *)c->pbuf[10];  // level 3 writes where level 1 was (see need[] above)
        if (defer) { pl.buf2 = (uint32_t *)c->ikeep[0]; pl.cnt2 = (uint32_t *)c->ikeep[1]; }
        bool fresh = c->filter_zero_pending;
        unsigned long long ov[2] = {0, 0};
        bool overflowed = false;
        uint64_t per_batch = 1;
        {
            Timed t(c, TPC_K_INSERT);
            if (fresh) HIPCHK(c, hipMemsetAsync(c->filter + (c->filter_words - 1), 0, sizeof(uint32_t), c->stream));
            const uint64_t per = pl.n_tiles;
            per_batch = per;
            for (uint64_t t0 = t_begin; t0 < t_end || t0 == t_begin; t0 += per) {  // (an empty chunk still runs one batch of no tiles: the slices must be written)
                pl.tile0 = t0;
                pl.n_tiles = t0 < t_end ? std::min<uint64_t>(per, t_end - t0) : 0;
                HIPCHK(c, hipMemsetAsync(pl.ovf_cur, 0, 2 * sizeof(unsigned long long), c->stream));
                if (defer) {  // levels 1 and 2 only; whether the apply can wait is known once the overflow count is back
                    TpcPartPlan p1 = pl;
                    p1.rbuf1 = p1.buf1; p1.rcnt1 = p1.cnt1;
                    if (tpc_launch_insert_part_hash(make_launch_periodic(c), p1, lo, hi, gated, n_kmers ? c->counters : nullptr) ||
                        tpc_launch_insert_part_split(make_launch(c), p1)) return fail(c, -1, "partitioned insert launch failed");
                    HIPCHK(c, hipMemcpyAsync(ov, pl.ovf_cur, sizeof ov, hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    bool keep = ov[0] <= TPC_FUSE_MAX_OVF && ov[1] == 0;
                    const uint32_t n_slices = 1u << (p1.b1 + p1.b2);
                    if (keep && ov[0]) {  // the overflow entries wait beside the regions, once as produced and once grouped by slice
                        if (c->ikeep_ovf_cap < ov[0]) {
                            if (c->ikeep_ovf) (void)hipFree(c->ikeep_ovf);
                            c->ikeep_ovf = nullptr; c->ikeep_ovf_cap = 0;
                            const uint64_t cap = std::max<uint64_t>(4096, ov[0] + ov[0] / 4);
                            if (hipMalloc((void **)&c->ikeep_ovf, 2 * cap * sizeof(uint64_t)) == hipSuccess) c->ikeep_ovf_cap = cap; else { (void)hipGetLastError(); keep = false; }
                        }
                        if (keep && c->iovf_slices < n_slices) {
                            if (c->iovf_cnt) (void)hipFree(c->iovf_cnt);
                            if (c->iovf_off) (void)hipFree(c->iovf_off);
                            c->iovf_cnt = nullptr; c->iovf_off = nullptr; c->iovf_slices = 0;
                            if (hipMalloc((void **)&c->iovf_cnt, 2 * (size_t)n_slices * sizeof(uint32_t)) == hipSuccess &&
                                hipMalloc((void **)&c->iovf_off, ((size_t)n_slices + 1) * sizeof(uint64_t)) == hipSuccess) c->iovf_slices = n_slices;
                            else { (void)hipGetLastError(); keep = false; }
                        }
                        if (keep) {
                            HIPCHK(c, hipMemcpyAsync(c->ikeep_ovf, pl.ovf, ov[0] * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
                            if (tpc_launch_ovf_by_slice(make_launch(c), c->ikeep_ovf, ov[0], p1.slice_bits, n_slices, c->iovf_cnt, c->iovf_cnt + n_slices, c->iovf_off,
                                                        c->ikeep_ovf + c->ikeep_ovf_cap)) return fail(c, -1, "overflow grouping launch failed");
                        }
                    }
                    if (keep) { c->pending_apply = true; c->pending_shard = false; c->pending_lists = false; c->pending_fresh = fresh; c->pending_pl = p1; c->pending_novf = ov[0]; }
                    else if (tpc_launch_insert_part_apply_only(make_launch(c), p1, fresh)) return fail(c, -1, "apply launch failed");
                    break;
                }
                if (tpc_launch_insert_partitioned(make_launch_periodic(c), pl, lo, hi, gated, fresh, n_kmers ? c->counters : nullptr))
                    return fail(c, -1, "partitioned insert launch failed");
                fresh = false;  // later batches OR into the slices
                HIPCHK(c, hipMemcpyAsync(ov, pl.ovf_cur, sizeof ov, hipMemcpyDeviceToHost, c->stream));
                if (t0 + per < t_end) { HIPCHK(c, hipStreamSynchronize(c->stream)); overflowed = overflowed || ov[1] != 0; }
            }
        }
        c->filter_zero_pending = false;
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->dbg_ovf)
            fprintf(stderr, "[ovf] insert: %llu overflow entries (cap %llu, flag %llu) b1=%d b2=%d ppr=%d cap1=%llu cap2=%llu nwg1=%u\n", ov[0], (unsigned long long)pl.ovf_cap, ov[1], pl.b1,
                    pl.b2, pl.pos_per_round, (unsigned long long)pl.cap1, (unsigned long long)pl.cap2, pl.nwg1);
        overflowed = overflowed || ov[1] != 0;
        c->stat_path[0] = (pl.b3 ? 3 : 2) + (overflowed ? 10 : 0);
        c->stat_fmt[0] = pl.fmt2;
        c->stat_insert_overflow = (int64_t)ov[0];
        c->stat_batches[0] = (int64_t)((tiles + per_batch - 1) / per_batch);
        if (!overflowed) {
            if (n_kmers) return read_counter(c, 0, n_kmers);
            return 0;
        }
        // the overflow list itself overflowed (pathological skew): OR is idempotent, so running the
        // direct kernel on top completes the filter
        if (replicated(c)) return fail(c, -20, "overflow list overflowed (address skew beyond what the sharded path handles)");  // (the direct kernel scans the whole text)
        if (n_kmers) HIPCHK(c, hipMemsetAsync(c->counters, 0, sizeof(unsigned long long), c->stream));
    }
    int rc = materialize_reset(c);
    if (rc) return rc;
    if (!part) { c->stat_path[0] = 1; c->stat_batches[0] = 1; }
    {
        Timed t(c, TPC_K_INSERT);
        TpcLaunch ad = make_launch(c);
        if (replicated(c)) { ad.g_begin = t_begin * 512ull * TPC_RUN; ad.g_end = t_end * 512ull * TPC_RUN; }  // (q > 16 only: see above)
        if (tpc_launch_insert(ad, lo, hi, gated, c->opt_test_first != 0, n_kmers ? c->counters : nullptr))
            return fail(c, -1, "insert launch failed");
    }
    HIPCHK(c, hipGetLastError());
    if (n_kmers) return read_counter(c, 0, n_kmers);
    return 0;
}

int tpc_reserve(tpc_ctx *c, uint64_t n_text_max)
{
    if (!c || !c->have_params) return fail(c, -1, "set_params first");
    if (c->sh_world > 1 || n_text_max < 2) return 0;  // sharded contexts size their buffers in tpc_shard_plan
    HIPCHK(c, hipSetDevice(c->device));
    // the same planning as an ungated tpc_pass1_insert / tpc_pass1_query of a text of n_text_max positions
    const uint64_t keep = c->n_text;
    c->n_text = n_text_max;
    c->reserve_text_bytes = c->bases ? 0 : (n_text_max / 32 + 1024) * 20;
    const uint64_t tiles = text_tiles512(c);
    size_t need[tpc_ctx::NPBUF] = {};
    TpcPartPlan pl;
    bool part = c->opt_insert_mode != 1 && !(c->opt_insert_mode == 0 && c->P.L < 28) && part_hash_supported(c);
    if (part) {
        const int64_t budget = part_budget(c);
        for (uint64_t batches = 1;; batches = next_batches(batches)) {
            const uint64_t per = (tiles + batches - 1) / batches;
            if (!tpc_part_plan(c->P.L, c->P.q, c->opt_slice_bits, per, 1.0, pl, c->opt_part_levels)) { part = false; break; }
            if ((int64_t)(std::max(tpc_part_buf1_bytes(pl), tpc_part_buf3_bytes(pl)) + tpc_part_buf2_bytes(pl)) <= budget || (int64_t)per <= c->opt_part_min_tiles) break;
        }
    }
    if (part) {
        const size_t ins[tpc_ctx::NPBUF] = { std::max(tpc_part_buf1_bytes(pl), tpc_part_buf3_bytes(pl)), tpc_part_cnt1_bytes(pl), tpc_part_buf2_bytes(pl), tpc_part_cnt2_bytes(pl),
                                             pl.ovf_cap * sizeof(uint64_t), 32 * sizeof(unsigned long long), 0, 0, 0, 0, tpc_part_cnt3_bytes(pl), 0 };
        for (int i = 0; i < tpc_ctx::NPBUF; i++) need[i] = ins[i];
    }
    TpcQPlan qpl;
    const bool qpart = plan_query(c, 0, c->P.lmask + 1, false, qpl);
    for (int i = 0; i < tpc_ctx::NPBUF && qpart; i++) need[i] = std::max(need[i], qpart_need(qpl, i));
    c->n_text = keep;
    for (int i = 0; i < tpc_ctx::NPBUF; i++)
        if (need[i] && !ensure_pbuf(c, i, need[i])) {  // not enough memory now: nothing half-reserved stays behind, the passes decide again when they run
            release_partition_buffers(c);
            return 0;
        }
    // the insert's level-2 regions kept aside while its apply is deferred into the query's lookup
    if (part && qpart && c->opt_fuse && pl.b3 == 0 && qpl.b3 == 0) {
        const size_t want[2] = { tpc_part_buf2_bytes(pl), tpc_part_cnt2_bytes(pl) };
        for (int i = 0; i < 2; i++) {
            if (want[i] <= c->ikeep_bytes[i]) continue;
            if (c->ikeep[i]) (void)hipFree(c->ikeep[i]);
            c->ikeep[i] = nullptr; c->ikeep_bytes[i] = 0;
            if (hipMalloc(&c->ikeep[i], want[i]) != hipSuccess) { (void)hipGetLastError(); break; }
            c->ikeep_bytes[i] = want[i];
        }
    }
    return 0;
}

int tpc_pass1_split_hist(tpc_ctx *c, const uint64_t *rec_start, const uint64_t *rec_len, uint32_t n_rec, uint32_t *bins_host)
{
    if (!c || !c->have_params || !c->bases || !bins_host) return fail(c, -1, "bad arguments");
    if (c->sh_world > 1) return fail(c, -1, "the split pass needs the whole filter as scratch: not available on a sharded context");
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t BINS = 1ull << 24;  // VE.h:471
    c->filter_zero_pending = false;  // the split pass zeroes its scratch filter itself
    c->pending_apply = false;
    // positions where a (k+1)-mer of 'N'+record+'N' starts, for dispatched records only (VE.h:1177): built on the device from the records
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<uint64_t> rs, rl;
    for (uint32_t r = 0; r < n_rec; r++) if (rec_len[r] >= (uint64_t)c->P.k) { rs.push_back(rec_start[r]); rl.push_back(rec_len[r]); }
    uint32_t *d_em = nullptr, *d_bins = nullptr;
    uint64_t *d_rec = nullptr;
    HIPCHK(c, hipMalloc((void **)&d_em, c->n_words_alloc * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void **)&d_bins, BINS * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void **)&d_rec, std::max<size_t>(1, 2 * rs.size()) * sizeof(uint64_t)));
    if (!rs.empty()) {
        HIPCHK(c, hipMemcpyAsync(d_rec, rs.data(), rs.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_rec + rs.size(), rl.data(), rl.size() * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    }
    tpc_launch_split_emask(c->stream, d_rec, d_rec + rs.size(), (uint32_t)rs.size(), c->P.k, c->n_words_alloc, d_em);
    HIPCHK(c, hipMemsetAsync(d_bins, 0, BINS * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(c->filter, 0, c->filter_words * sizeof(uint32_t), c->stream));
    const uint64_t real = 1ull << c->P.L;
    const uint64_t bin_size = std::max<uint64_t>(1, real / BINS);  // VE.h:169
    {
        Timed t(c, TPC_K_SPLIT);
        tpc_launch_split(make_launch(c), d_em, d_bins, bin_size);
    }
    HIPCHK(c, hipMemcpyAsync(bins_host, d_bins, BINS * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(d_em);
    (void)hipFree(d_bins);
    (void)hipFree(d_rec);
    if (c->dbg_timing)
        fprintf(stderr, "[timing]   tpc_pass1_split_hist: %.1f ms in all, the split kernel %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(),
                tpc_kernel_ms(c, TPC_K_SPLIT));
    return 0;
}

namespace {
int pass1_query_impl(tpc_ctx *c, uint64_t lo, uint64_t hi, uint64_t *n_marks, bool begin_only);
}

int tpc_pass1_query(tpc_ctx *c, uint64_t lo, uint64_t hi, uint64_t *n_marks) { return pass1_query_impl(c, lo, hi, n_marks, false); }

int tpc_pass1_query_begin(tpc_ctx *c, uint64_t lo, uint64_t hi) { return pass1_query_impl(c, lo, hi, nullptr, true); }

namespace {

// begin_only: the part of the query that does not need the filter -- the level-1 hash and the level-2 binning of the FIRST tile batch --
// is enqueued and the call returns without waiting (tpc_pass1_query_begin); the tpc_pass1_query of the same range that follows picks
// up there.  The combined multi-GPU exchange runs between the two: the lists travel while the probes are being binned.
int pass1_query_impl(tpc_ctx *c, uint64_t lo, uint64_t hi, uint64_t *n_marks, bool begin_only)
{
    if (!c || !c->have_params || !c->bases) return fail(c, -1, "set_params and seq_upload first");
    if (c->sh_world > 1 && !c->opt_replicate) return fail(c, -1, "the filter is sharded: use tpc_shard_hash / tpc_shard_apply");
    HIPCHK(c, hipSetDevice(c->device));
    const bool gated = !(lo == 0 && hi >= c->P.lmask);
    const bool begun = !begin_only && c->qb_valid && c->qb_lo == lo && c->qb_hi == hi;  // the first batch's hash and binning are on the stream already
    c->qb_valid = false;
    c->marks_valid = false; c->rmask_sums_valid = false;
    ensure_periodic(c);
    TpcQPlan pl;
    uint64_t t_begin, t_end;
    pass_tiles(c, t_begin, t_end);  // every tile, or this rank's chunk (option replicate_filter on a sharded context: the marks of the chunk only)
    bool part = plan_query(c, lo, hi, gated, pl);
    if (replicated(c)) {
        if (!part && c->P.q <= TPC_KERNEL_MAXQ) return fail(c, -1, "a replicated multi-GPU pass needs the partitioned query (q=%d, L=%d, slice_bits=%d)", c->P.q, c->P.L, c->opt_slice_bits);
        if (!begun) HIPCHK(c, hipMemsetAsync(c->rmask, 0, c->n_words_alloc * sizeof(uint32_t), c->stream));  // the hash kernel rewrites the words of this rank's tiles only
    }
    if (begin_only && !part) return 0;  // (the direct kernel needs the filter from its first instruction: nothing to start early)
    if (part)
        for (int i = 0; i < tpc_ctx::NPBUF && part; i++) if (qpart_need(pl, i)) part = ensure_pbuf(c, i, qpart_need(pl, i));  // not enough HBM: direct path
    // deferred apply of this round's insert: the lookup builds the slices (k_apply_lookup) when the geometry still matches
    const bool fused = !begin_only && c->pending_apply && !c->pending_shard && part && pl.b3 == 0 && pl.slice_bits == c->pending_pl.slice_bits &&
                       pl.b1 == c->pending_pl.b1 && pl.b2 == c->pending_pl.b2 && (pl.fmt == 6 || c->pending_pl.fmt2 == 0);
    if (!fused && !begin_only) { int rc0 = materialize_reset(c); if (rc0) return rc0; }
    if (part) {
        pl.buf1 = (uint64_t *)c->pbuf[0]; pl.cnt1 = (uint32_t *)c->pbuf[1]; pl.buf2 = (uint64_t *)c->pbuf[2]; pl.cnt2 = (uint32_t *)c->pbuf[3];
        pl.ovf = (uint64_t *)c->pbuf[4]; pl.ovf_cur = (unsigned long long *)c->pbuf[5];
        pl.surv = (uint64_t *)c->pbuf[6]; pl.surv_cur = (unsigned long long *)c->pbuf[7];
        pl.off2 = (const uint64_t *)c->pbuf[8];
        if (c->off2_uploaded != pl.off2_host) {
            HIPCHK(c, hipMemcpy(c->pbuf[8], pl.off2_host.data(), pl.off2_host.size() * 8, hipMemcpyHostToDevice));
            c->off2_uploaded = pl.off2_host;
        }
        pl.buf3 = (uint64_t *)c->pbuf[0]; pl.cnt3 = (uint32_t *)c->pbuf[10]; pl.off3 = (const uint64_t *)c->pbuf[11];  // qpart_need
        pl.bnd = (uint32_t *)c->pbuf[18];
        if (pl.b3 && c->off3_uploaded != pl.off3_host) {
            HIPCHK(c, hipMemcpy(c->pbuf[11], pl.off3_host.data(), pl.off3_host.size() * 8, hipMemcpyHostToDevice));
            c->off3_uploaded = pl.off3_host;
        }
        unsigned long long f1[2] = {0, 0}, f2 = 0;
        bool overflowed = false;
        uint64_t per_batch = 1;
        {
            Timed t(c, TPC_K_QUERY);
            const uint64_t per = pl.n_tiles;
            per_batch = per;
            for (uint64_t t0 = t_begin; (t0 < t_end || t0 == t_begin) && !overflowed; t0 += per) {  // (an empty chunk: one batch of no tiles -- a pending apply is still carried out)
                pl.tile0 = t0;
                pl.tile0_global = t0;
                pl.n_tiles = t0 < t_end ? std::min<uint64_t>(per, t_end - t0) : 0;
                const bool presplit = begun && t0 == t_begin;
                if (!presplit) {
                    HIPCHK(c, hipMemsetAsync(pl.ovf_cur, 0, 32 * sizeof(unsigned long long), c->stream));
                    HIPCHK(c, hipMemsetAsync(pl.surv_cur, 0, TPC_SURV_CUR_WORDS * sizeof(unsigned long long), c->stream));
                }
                if (begin_only) {  // hash + binning of the first batch, then back to the caller without waiting
                    TpcQPlan p1 = pl;
                    p1.rbuf1 = p1.buf1; p1.rcnt1 = p1.cnt1;
                    if (tpc_launch_query_part_hash(make_launch_periodic(c), p1, c->rmask, lo, hi, gated) || tpc_launch_query_part_split(make_launch(c), p1))
                        return fail(c, -1, "partitioned query launch failed");
                    HIPCHK(c, hipGetLastError());
                    c->qb_valid = true; c->qb_lo = lo; c->qb_hi = hi;
                    return 0;
                }
                if (fused && t0 == t_begin) {  // the first batch's lookup kernel also builds and writes the filter slices; later batches read them
                    TpcQPlan p1 = pl;
                    p1.rbuf1 = p1.buf1; p1.rcnt1 = p1.cnt1;
                    p1.presplit = presplit;
                    c->pending_apply = false;
                    c->stat_fused++;
                    if (!presplit && tpc_launch_query_part_hash(make_launch_periodic(c), p1, c->rmask, lo, hi, gated)) return fail(c, -1, "partitioned query launch failed");
                    {
                        Timed tf(c, TPC_K_FUSED);
                        TpcLaunch af = make_launch(c);
                        af.ev_lookup0 = c->ev0[TPC_K_LOOKUP]; af.ev_lookup1 = c->ev1[TPC_K_LOOKUP]; c->ev_used[TPC_K_LOOKUP] = true;
                        const bool lists = c->pending_lists;  // the combined exchange: the slices are built from the imported set-bit lists
                        c->pending_lists = false;
                        if (tpc_launch_query_part_fused_lookup(af, p1, c->pending_pl, c->pending_fresh, c->pending_novf && !lists ? c->ikeep_ovf + c->ikeep_ovf_cap : nullptr,
                                                               c->pending_novf && !lists ? c->iovf_off : nullptr, lists ? &c->cmb_ls : nullptr)) return fail(c, -1, "fused lookup launch failed");
                    }
                    if (tpc_launch_query_verify(make_launch(c), p1, c->rmask)) return fail(c, -1, "verify launch failed");
                } else if (presplit) {  // (not fused after all -- e.g. the dense form of the exchange: lookup against the filter, verification)
                    TpcQPlan p1 = pl;
                    p1.rbuf1 = p1.buf1; p1.rcnt1 = p1.cnt1;
                    p1.presplit = true;
                    if (tpc_launch_query_part_lookup(make_launch(c), p1) || tpc_launch_query_verify(make_launch(c), p1, c->rmask)) return fail(c, -1, "partitioned query launch failed");
                } else
                if (tpc_launch_query_partitioned(make_launch_periodic(c), pl, c->rmask, lo, hi, gated)) return fail(c, -1, "partitioned query launch failed");
                HIPCHK(c, hipMemcpyAsync(f1, pl.ovf_cur, sizeof f1, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipMemcpyAsync(&f2, pl.surv_cur + 64, sizeof f2, hipMemcpyDeviceToHost, c->stream));
                if (t0 + per < t_end) {
                    // Grouping the survivors by address pays when they are true second edges (every genome's occurrence of an edge probes
                    // the same words); a batch whose first-probe survivors were mostly Bloom false positives -- a well-filled filter: full
                    // configs[3] marks 2 % of them -- has nothing to bring together, and the next batch's lookup skips the sort.  The
                    // batch's marks (a count over its words of the round mask: 0.03 ms) against its survivors say which it was.
                    // (Counting the passes inside k_q_verify2 -- one more register, an atomic per wave -- cost that kernel 2.26 -> 2.6 ms.)
                    unsigned long long sc[TPC_SURV_CUR_WORDS], batch_marks = 0;
                    const uint64_t w0 = t0 * 512, wn = std::min<uint64_t>(pl.n_tiles * 512, c->n_words > w0 ? c->n_words - w0 : 0);
                    if (wn) tpc_launch_mask_count(c->stream, c->rmask + w0, wn, c->block_sums, c->counters + 3);
                    HIPCHK(c, hipMemcpyAsync(sc, pl.surv_cur, sizeof sc, hipMemcpyDeviceToHost, c->stream));
                    if (wn) HIPCHK(c, hipMemcpyAsync(&batch_marks, c->counters + 3, sizeof batch_marks, hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    overflowed = f1[1] != 0 || f2 != 0;
                    unsigned long long surv = 0;
                    for (int i = 0; i < 64; i++) surv += std::min<unsigned long long>(sc[i], pl.surv_cap);
                    if (surv > 0) pl.group_survivors = batch_marks * 4 >= surv;
                }
            }
            if (c->periodic_valid && c->periodic_any_q && c->periodic) tpc_launch_periodic_copy(c->stream, c->rmask, c->periodic, c->periodic + c->n_words_alloc, c->n_words_alloc, c->n_words);  // positions that sent no probes take their twin's verdict
            tpc_launch_mask_count(c->stream, c->rmask, c->n_words, c->block_sums, c->counters + 1);
        }
        HIPCHK(c, hipGetLastError());
        uint64_t n = 0;
        int rc = read_counter(c, 1, &n);
        if (rc) return rc;
        overflowed = overflowed || f1[1] != 0 || f2 != 0;
        c->stat_query_overflow = (int64_t)f1[0];
        c->stat_path[1] = (pl.b3 ? 3 : 2) + (overflowed ? 10 : 0);
        c->stat_fmt[1] = pl.fmt;
        c->stat_batches[1] = (int64_t)((pass_tile_count(c) + per_batch - 1) / per_batch);
        if (c->dbg_ovf) {
            unsigned long long sc[65];
            (void)hipMemcpy(sc, pl.surv_cur, sizeof sc, hipMemcpyDeviceToHost);
            unsigned long long tot = 0, most = 0;
            for (int i = 0; i < 64; i++) { tot += sc[i]; most = std::max(most, sc[i]); }
            fprintf(stderr, "[ovf] query: %llu overflow entries (cap %llu, flag %llu), survivors %llu (fullest list %llu of %llu, flag %llu) b1=%d b2=%d ppr=%d loads=%d\n", f1[0],
                    (unsigned long long)pl.ovf_cap, f1[1], tot, most, (unsigned long long)pl.surv_cap, f2, pl.b1, pl.b2, pl.pos_per_round, pl.loads);
            static const char *const dump_path = getenv("TPC_DUMP_SURV");
            if (const char *path = dump_path) {  // development: the first-probe survivors of the last batch, one id per line
                if (FILE *fp = fopen(path, "w")) {
                    for (int i = 0; i < 64; i++) {
                        const size_t n = (size_t)std::min<unsigned long long>(sc[i], pl.surv_cap);
                        std::vector<uint64_t> ids(n);
                        if (n) (void)hipMemcpy(ids.data(), pl.surv + (size_t)i * pl.surv_cap, n * 8, hipMemcpyDeviceToHost);
                        for (uint64_t v : ids) fprintf(fp, "%llu\n", (unsigned long long)v);
                    }
                    fclose(fp);
                }
            }
            static const bool dbg_regions = getenv("TPC_DEBUG_OVF_REGIONS") != nullptr;
            if (dbg_regions && f1[0]) {  // where the overflow entries go: (permuted) slice histogram of the list
                const size_t n = (size_t)std::min<unsigned long long>(f1[0], 1u << 22);
                std::vector<uint64_t> ent(2 * n);
                (void)hipMemcpy(ent.data(), pl.ovf, 2 * n * 8, hipMemcpyDeviceToHost);
                std::map<uint32_t, uint32_t> h;
                std::map<uint64_t, uint32_t> ha;
                for (size_t i = 0; i < n; i++) { h[(uint32_t)(ent[2 * i] >> pl.slice_bits)]++; ha[ent[2 * i]]++; }
                std::vector<std::pair<uint32_t, uint32_t>> top;
                for (auto &kv : h) top.push_back({kv.second, kv.first});
                std::sort(top.rbegin(), top.rend());
                fprintf(stderr, "[ovf]   %zu entries in %zu permuted slices, %zu distinct addresses; top:", n, h.size(), ha.size());
                for (size_t i = 0; i < std::min<size_t>(top.size(), 10); i++) fprintf(stderr, " slice %u x %u;", top[i].second, top[i].first);
                std::vector<std::pair<uint32_t, uint64_t>> topa;
                for (auto &kv : ha) topa.push_back({kv.second, kv.first});
                std::sort(topa.rbegin(), topa.rend());
                for (size_t i = 0; i < std::min<size_t>(topa.size(), 6); i++) fprintf(stderr, " addr %llx x %u (edge %llu pos %llu);", (unsigned long long)topa[i].second, topa[i].first, 0ull, 0ull);
                fprintf(stderr, "\n");
            }
            if (dbg_regions && !pl.b3 && pl.fmt == 0) {  // which level-2 regions are full
                const size_t nreg = pl.off2_host.size() - 1;
                std::vector<uint32_t> cnt(nreg);
                (void)hipMemcpy(cnt.data(), pl.cnt2, nreg * 4, hipMemcpyDeviceToHost);
                const uint32_t smask = (1u << (pl.b1 + pl.b2)) - 1u;
                size_t full = 0;
                std::vector<uint32_t> hist(64, 0);
                for (size_t r = 0; r < nreg; r++) {
                    const uint64_t cap = pl.off2_host[r + 1] - pl.off2_host[r];
                    const uint32_t s = (uint32_t)(((uint64_t)r * pl.perm_inv) & smask);  // PtPerm::slice_of
                    if (cnt[r] + 16 >= cap) { full++; hist[s >> (pl.b1 + pl.b2 - 6)]++; if (full <= 12) fprintf(stderr, "[ovf]   region %zu (slice %u of %zu): %u of %llu\n", r, s, nreg, cnt[r], (unsigned long long)cap); }
                }
                fprintf(stderr, "[ovf]   %zu full regions; by 64ths of the address space:", full);
                for (int i = 0; i < 64; i++) fprintf(stderr, " %u", hist[i]);
                fprintf(stderr, "\n");
            }
        }
        if (c->dbg_phases) {
            unsigned long long pr[32];
            (void)hipMemcpy(pr, pl.ovf_cur, sizeof pr, hipMemcpyDeviceToHost);
            fprintf(stderr, "[phases] ovf=%llu  split: push %llu book %llu copy %llu rounds %llu | hash: push %llu book %llu copy %llu rounds %llu (10 ns ticks summed over WGs) | lost at level 1: %llu, at level 2/3: %llu\n",
                    pr[0], pr[8], pr[9], pr[10], pr[12], pr[16], pr[17], pr[18], pr[20], pr[25], pr[27]);
        }
        if (!overflowed) {
            c->rmask_sums_valid = true;
            c->rmask_sums_n = n;
            if (n_marks) *n_marks = n;
            return 0;
        }
        // an overflow list overflowed (pathological skew): the direct kernel below rewrites the whole mask
        if (replicated(c)) return fail(c, -20, "overflow or survivor list overflowed (address skew beyond what the sharded path handles)");  // (the direct kernel scans the whole text)
    }
    if (!part) { c->stat_path[1] = 1; c->stat_batches[1] = 1; }
    HIPCHK(c, hipMemsetAsync(c->counters + 1, 0, sizeof(unsigned long long), c->stream));
    {
        Timed t(c, TPC_K_QUERY);
        TpcLaunch ad = make_launch(c);
        if (replicated(c)) { ad.g_begin = t_begin * 512ull * TPC_RUN; ad.g_end = t_end * 512ull * TPC_RUN; }  // (q > 16: the closed-form kernel over this rank's chunk)
        if (tpc_launch_query(ad, c->rmask, lo, hi, gated, c->counters + 1)) return fail(c, -1, "query launch failed");
    }
    HIPCHK(c, hipGetLastError());
    uint64_t n = 0;
    int rc = read_counter(c, 1, &n);
    if (rc) return rc;
    if (n_marks) *n_marks = n;
    return 0;
}

}  // namespace

namespace {
int pass2_filter_impl(tpc_ctx *c, const uint64_t *fmarks, uint64_t n_fmarks, bool external, uint64_t abundance, uint64_t *n_true, uint64_t *n_false, uint64_t *table_size,
                      bool records = false, bool aggregated = false);
// aggregated records (tpc_pass2_aggregate_records): whether occurrences are counted cannot depend on how many records a rank happens to
// hold, so it depends on the cut alone -- any abundance a key could exceed counts
inline bool aggregated_counted(uint64_t abundance) { return abundance < (1ull << 40); }
}

int tpc_pass2_filter(tpc_ctx *c, uint64_t abundance, uint64_t *n_true, uint64_t *n_false, uint64_t *table_size)
{
    return pass2_filter_impl(c, nullptr, 0, false, abundance, n_true, n_false, table_size);
}

int tpc_pass2_marks(tpc_ctx *c, uint64_t *n_marks)
{
    if (!c || !c->have_params || !c->bases || !n_marks) return fail(c, -1, "set_params and seq_upload first");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = compact_mask(c, c->rmask);
    if (rc) return rc;
    c->marks_valid = true;
    *n_marks = c->n_marks;
    return 0;
}

int tpc_pass2_mark_owners(tpc_ctx *c, uint32_t world, uint64_t *pos_dev, int32_t *owner_dev)
{
    if (!c || !c->marks_valid || world == 0 || (c->n_marks && (!pos_dev || !owner_dev))) return fail(c, -1, "tpc_pass2_marks first");
    if (c->text_windowed) return fail(c, -1, "this context holds only its window of the text (option text_window)");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n_marks) {
        HIPCHK(c, hipMemcpyAsync(pos_dev, c->marks, c->n_marks * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
        if (tpc_launch_mark_owner(make_launch(c), c->C, c->marks, c->n_marks, world, owner_dev)) return fail(c, -1, "owner launch failed");
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_pass2_filter_positions(tpc_ctx *c, const uint64_t *pos_dev, uint64_t n, uint64_t abundance, uint64_t *n_true, uint64_t *n_false, uint64_t *table_size)
{
    if (n && !pos_dev) return fail(c, -1, "bad arguments");
    return pass2_filter_impl(c, pos_dev, n, true, abundance, n_true, n_false, table_size);
}

int tpc_pass2_mark_records(tpc_ctx *c, uint32_t world, uint64_t *records_dev, int32_t *owner_dev)
{
    if (!c || !c->marks_valid || world == 0 || (c->n_marks && (!records_dev || !owner_dev))) return fail(c, -1, "tpc_pass2_marks first");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n_marks && tpc_launch_mark_records(make_launch(c), c->C, c->marks, c->n_marks, world, records_dev, owner_dev)) return fail(c, -1, "record launch failed");
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_pass2_filter_records(tpc_ctx *c, const uint64_t *records_dev, uint64_t n, uint64_t abundance, uint64_t *n_true, uint64_t *n_false, uint64_t *table_size)
{
    if (n && !records_dev) return fail(c, -1, "bad arguments");
    return pass2_filter_impl(c, records_dev, n, true, abundance, n_true, n_false, table_size, true);
}

int tpc_pass2_aggregate_records(tpc_ctx *c, uint32_t world, uint64_t abundance, uint64_t *records_dev, int32_t *owner_dev, uint64_t *n_records)
{
    if (!c || !c->marks_valid || world == 0 || !n_records || (c->n_marks && (!records_dev || !owner_dev))) return fail(c, -1, "tpc_pass2_marks first");
    HIPCHK(c, hipSetDevice(c->device));
    *n_records = 0;
    if (c->n_marks == 0) return 0;
    const size_t sb = tpc_table_slot_bytes(c->C);
    TpcLaunch a = make_launch(c);
    const bool counted = aggregated_counted(abundance);
    if (!c->scan_blocks) HIPCHK(c, hipMalloc((void **)&c->scan_blocks, 2 * TPC_SCAN2_BLOCKS * sizeof(uint64_t)));
    uint64_t full = 1024;
    while (full < 2 * c->n_marks + 2) full <<= 1;
    uint64_t cap = 1024;
    while (cap < c->n_marks / 4 + 2) cap <<= 1;
    for (;;) {  // as pass2_filter_impl: sized for the usual ratio of marks to distinct keys, repeated at full size when a probe sequence says so
        if (cap > c->table_alloc) {
            if (c->table) (void)hipFree(c->table);
            c->table = nullptr;
            c->table_alloc = 0;
            HIPCHK(c, dev_malloc(c, &c->table, cap * sb));
            c->table_alloc = cap;
        }
        HIPCHK(c, hipMemsetAsync(c->counters + 6, 0, sizeof(unsigned long long), c->stream));
        Timed t(c, TPC_K_FILTER2);
        tpc_launch_table_init(c->stream, c->table, cap);
        if (tpc_launch_filter2(a, c->C, c->marks, c->n_marks, c->table, cap, counted, c->counters + 6)) return fail(c, -1, "filter2 launch failed");
        if (tpc_launch_scan2_count(a, c->table, cap, abundance, counted, c->scan_blocks, c->scan_blocks + TPC_SCAN2_BLOCKS, c->counters + 4))
            return fail(c, -1, "scan2 launch failed");
        unsigned long long three[3] = {0, 0, 0};
        HIPCHK(c, hipMemcpyAsync(three, c->counters + 4, sizeof three, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!three[2]) {
            *n_records = three[1];
            if (tpc_launch_table_records(a, c->C, c->marks, c->table, cap, c->scan_blocks + TPC_SCAN2_BLOCKS, world, records_dev, owner_dev))
                return fail(c, -1, "record launch failed");
            break;
        }
        if (cap >= full) return fail(c, -1, "exact-filter table overflow at full size");
        cap = full;
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_pass2_filter_aggregated(tpc_ctx *c, const uint64_t *records_dev, uint64_t n, uint64_t abundance, uint64_t *n_true, uint64_t *n_false, uint64_t *table_size)
{
    if (n && !records_dev) return fail(c, -1, "bad arguments");
    return pass2_filter_impl(c, records_dev, n, true, abundance, n_true, n_false, table_size, true, true);
}

int tpc_shard_permute_rows(tpc_ctx *c, const uint64_t *src_dev, const uint32_t *perm_dev, uint64_t n, int row_words, uint64_t *dst_dev)
{
    if (!c || row_words < 1 || (n && (!src_dev || !perm_dev || !dst_dev))) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    tpc_launch_permute_rows(c->stream, src_dev, perm_dev, n, row_words, dst_dev);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

namespace {

// external = false: the positions marked in this round's mask, compacted here; true: the n_fmarks given positions (possibly
// none: the occurrences whose keys this rank owns, tpc_pass2_mark_owners on every rank + an exchange; the round's own marks
// were compacted by tpc_pass2_marks).  Either way the round's mask is then merged into the run-wide one.
// records: the external list holds records of C + 1 words (tpc_pass2_mark_records) instead of positions: no text access at all.
int pass2_filter_impl(tpc_ctx *c, const uint64_t *fmarks, uint64_t n_fmarks, bool external, uint64_t abundance, uint64_t *n_true, uint64_t *n_false, uint64_t *table_size,
                      bool records, bool aggregated)
{
    if (!c || !c->have_params || !c->bases) return fail(c, -1, "set_params and seq_upload first");
    if (c->text_windowed && !records) return fail(c, -1, "this context holds only its window of the text (option text_window): the second pass needs all of it");
    HIPCHK(c, hipSetDevice(c->device));
    if (!external) {
        int rc = compact_mask(c, c->rmask);
        if (rc) return rc;
        c->marks_valid = true;
    } else if (!c->marks_valid) return fail(c, -1, "tpc_pass2_marks first");
    const uint64_t *marks = external ? fmarks : c->marks;
    const uint64_t n_marks = external ? n_fmarks : c->n_marks;
    // Exact-filter table, a power of two.  Sized first for the usual case -- on many-genome inputs a key is marked dozens
    // of times, so marks / 4 slots hold the distinct keys several times over and the table (and TrueBifurcations' scan of it)
    // stays cache sized; a probe sequence longer than TPC_FILTER2_PROBE_LIMIT flags a table that is too full and the pass
    // is repeated with 2 x marks slots, which always suffices.
    const size_t sb = tpc_table_slot_bytes(c->C);
    TpcLaunch a = make_launch(c);
    const bool counted = aggregated ? aggregated_counted(abundance) : abundance < n_marks;  // otherwise no key can exceed the abundance cut
    if (!c->scan_blocks) HIPCHK(c, hipMalloc((void **)&c->scan_blocks, 2 * TPC_SCAN2_BLOCKS * sizeof(uint64_t)));
    uint64_t full = 1024;
    while (full < 2 * n_marks + 2) full <<= 1;
    uint64_t cap = 1024;
    while (cap < n_marks / 4 + 2) cap <<= 1;  // (marks / 8 and / 16 measured the same on M2: k_filter2 0.946 -> 0.943 / 0.944 ms)
    if (aggregated) cap = full;               // a rank sends a key once: the records are distinct up to the number of ranks
    uint64_t tp = 0, used = 0;
    c->stat_filter2_retries = 0;
    for (;;) {
        if (cap > c->table_alloc) {
            if (c->table) (void)hipFree(c->table);
            c->table = nullptr;
            c->table_alloc = 0;
            HIPCHK(c, dev_malloc(c, &c->table, cap * sb));
            c->table_alloc = cap;
        }
        c->table_cap = cap;
        HIPCHK(c, hipMemsetAsync(c->counters + 6, 0, sizeof(unsigned long long), c->stream));
        {
            Timed t(c, TPC_K_FILTER2);
            // key = EMPTY (all ones), meta = 0
            tpc_launch_table_init(c->stream, c->table, cap);
            if (records ? tpc_launch_filter2_rec(a, c->C, marks, n_marks, c->table, cap, counted, c->counters + 6)
                        : tpc_launch_filter2(a, c->C, marks, n_marks, c->table, cap, counted, c->counters + 6)) return fail(c, -1, "filter2 launch failed");
        }
        uint64_t too_full = 0;
        {
            Timed t(c, TPC_K_SCAN2);
            if (tpc_launch_scan2_count(a, c->table, cap, abundance, counted, c->scan_blocks, c->scan_blocks + TPC_SCAN2_BLOCKS, c->counters + 4))
                return fail(c, -1, "scan2 launch failed");
            unsigned long long three[3] = {0, 0, 0};
            HIPCHK(c, hipMemcpyAsync(three, c->counters + 4, sizeof three, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            tp = three[0]; used = three[1]; too_full = three[2];
            if (!too_full && tp) {
                const uint64_t need = c->n_keys + tp;
                if (need > c->keys_cap) {
                    uint64_t *nk = nullptr;
                    const uint64_t ncap = need + need / 4 + 1024;
                    HIPCHK(c, dev_malloc(c, (void **)&nk, ncap * c->C * sizeof(uint64_t)));
                    if (c->n_keys) HIPCHK(c, hipMemcpyAsync(nk, c->keys, c->n_keys * c->C * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    if (c->keys) (void)hipFree(c->keys);
                    c->keys = nk;
                    c->keys_cap = ncap;
                }
                if (records ? tpc_launch_scan2_write_rec(a, c->C, marks, c->table, cap, abundance, counted, c->scan_blocks, c->keys + c->n_keys * c->C)
                            : tpc_launch_scan2_write(a, c->C, marks, c->table, cap, abundance, counted, c->scan_blocks, c->keys + c->n_keys * c->C))
                    return fail(c, -1, "scan2 launch failed");
                c->n_keys += tp;
            }
        }
        if (!too_full) break;
        if (cap >= full) return fail(c, -1, "exact-filter table overflow at full size");
        cap = full;
        c->stat_filter2_retries++;
    }
    // MergeOr into the run-wide mask (VE.h:909-913)
    if (c->rounds_done == 0) {
        HIPCHK(c, hipMemcpyAsync(c->mask, c->rmask, c->n_words_alloc * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    } else {
        tpc_launch_mask_or(c->stream, c->mask, c->rmask, c->n_words_alloc);
        c->mask_dirty = true;
    }
    c->rounds_done++;
    c->finalized = false;
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n_true) *n_true = tp;
    if (n_false) *n_false = used - tp;
    if (table_size) *table_size = used;
    return 0;
}

}  // namespace

int tpc_junctions_finalize(tpc_ctx *c, uint64_t *n_junctions)
{
    if (!c || !c->have_params) return fail(c, -1, "set_params first");
    HIPCHK(c, hipSetDevice(c->device));
    {
        Timed t(c, TPC_K_SORT);
        int rc = tpc_launch_sort_keys(c->stream, c->C, c->P.k, c->keys, c->n_keys, &c->sort_scratch, &c->sort_scratch_bytes);
        if (rc) return fail(c, rc, "key sort failed (%d)", rc);
        uint64_t cap = 1024;
        while (cap < 2 * c->n_keys + 2) cap <<= 1;
        const size_t slot_bytes = c->C == 1 ? 16 : 4;  // one-word keys sit in the slot next to their rank (tpc_pass2.hip:k_idtab_build)
        if (cap * slot_bytes > c->idtab_bytes) {
            if (c->idtab) (void)hipFree(c->idtab);
            c->idtab = nullptr;
            c->idtab_bytes = 0;
            HIPCHK(c, dev_malloc(c, (void **)&c->idtab, cap * slot_bytes));
            c->idtab_bytes = cap * slot_bytes;
        }
        c->idtab_cap = cap;
        HIPCHK(c, hipMemsetAsync(c->idtab, 0, cap * slot_bytes, c->stream));
        if (c->n_keys >= 0xFFFFFFFFull) return fail(c, -1, "too many junctions for the 32-bit id index");
        tpc_launch_idtab_build(c->stream, c->C, c->keys, c->n_keys, c->idtab, cap);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->finalized = true;
    c->keys_host.clear();
    if (n_junctions) *n_junctions = c->n_keys;
    return 0;
}

int tpc_key_words(const tpc_ctx *c) { return c ? c->C : 0; }

int tpc_junction_keys(tpc_ctx *c, uint64_t *keys_host)
{
    if (!c || !c->finalized) return fail(c, -1, "junctions_finalize first");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n_keys) HIPCHK(c, hipMemcpy(keys_host, c->keys, c->n_keys * c->C * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

int tpc_junction_keys_raw(tpc_ctx *c, uint64_t *keys_host, uint64_t *n)
{
    if (!c || !n) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    *n = c->n_keys;
    if (keys_host && c->n_keys) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(keys_host, c->keys, c->n_keys * c->C * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    return 0;
}

int tpc_junction_keys_set(tpc_ctx *c, const uint64_t *keys_host, uint64_t n)
{
    if (!c || (!keys_host && n)) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    if (n > c->keys_cap) {
        if (c->keys) (void)hipFree(c->keys);
        c->keys = nullptr; c->keys_cap = 0;
        HIPCHK(c, dev_malloc(c, (void **)&c->keys, (n + 1024) * c->C * sizeof(uint64_t)));
        c->keys_cap = n + 1024;
    }
    if (n) HIPCHK(c, hipMemcpy(c->keys, keys_host, n * c->C * sizeof(uint64_t), hipMemcpyHostToDevice));
    c->n_keys = n;
    c->finalized = false;
    c->keys_host.clear();
    return 0;
}

int tpc_junction_keys_export(tpc_ctx *c, uint64_t *dst_dev, uint64_t cap_keys, uint64_t *n)
{
    if (!c || !n || (!dst_dev && cap_keys)) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    *n = c->n_keys;
    const uint64_t m = std::min(c->n_keys, cap_keys);
    if (m) HIPCHK(c, hipMemcpyAsync(dst_dev, c->keys, m * c->C * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_junction_keys_import(tpc_ctx *c, const uint64_t *src_dev, uint64_t n, int append)
{
    if (!c || (!src_dev && n)) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t base = append ? c->n_keys : 0;
    const uint64_t need = base + n;
    if (need > c->keys_cap) {
        uint64_t *nk = nullptr;
        const uint64_t ncap = need + need / 2 + 1024;
        HIPCHK(c, dev_malloc(c, (void **)&nk, ncap * c->C * sizeof(uint64_t)));
        if (base) HIPCHK(c, hipMemcpy(nk, c->keys, base * c->C * sizeof(uint64_t), hipMemcpyDeviceToDevice));
        if (c->keys) (void)hipFree(c->keys);
        c->keys = nk;
        c->keys_cap = ncap;
    }
    if (n) HIPCHK(c, hipMemcpyAsync(c->keys + base * c->C, src_dev, n * c->C * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->n_keys = need;
    c->finalized = false;
    c->keys_host.clear();
    return 0;
}

int64_t tpc_get_id(tpc_ctx *c, const char *kmer)
{   // BifurcationStorage::GetId, bifurcationstorage.h:100-127 (host-side: cold query API)
    if (!c || !c->finalized || !kmer) return TPC_INVALID_VERTEX;
    const int k = c->P.k, C = c->C;
    if (c->keys_host.size() != c->n_keys * C) {
        c->keys_host.resize(c->n_keys * C);
        if (tpc_junction_keys(c, c->keys_host.data())) return TPC_INVALID_VERTEX;
    }
    std::vector<uint64_t> fw(C, 0), rc(C, 0);
    for (int i = 0; i < k; i++) {
        int code;
        switch (kmer[i]) { case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break; case 'T': code = 3; break; default: return TPC_INVALID_VERTEX; }
        fw[i >> 5] |= (uint64_t)code << (2 * (i & 31));
        const int j = k - 1 - i;
        rc[j >> 5] |= (uint64_t)(3 - code) << (2 * (j & 31));
    }
    auto find = [&](const std::vector<uint64_t> &key) -> int64_t {
        uint64_t lo = 0, hi = c->n_keys;
        auto less = [&](const uint64_t *a, const uint64_t *b) {  // CompressedString::Less
            for (int w = 0; w < C; w++) if (a[w] != b[w]) return a[w] < b[w];
            return false;
        };
        while (lo < hi) {
            const uint64_t mid = (lo + hi) / 2;
            if (less(&c->keys_host[mid * C], key.data())) lo = mid + 1; else hi = mid;
        }
        if (lo < c->n_keys && !less(key.data(), &c->keys_host[lo * C]) && !less(&c->keys_host[lo * C], key.data())) return (int64_t)lo;
        return -1;
    };
    int64_t r = find(fw);
    if (r >= 0) return r + 1;
    r = find(rc);
    if (r >= 0) return -(r + 1);
    return TPC_INVALID_VERTEX;
}

int tpc_emit(tpc_ctx *c, uint64_t *n_marked, uint64_t *n_valid)
{
    if (!c || !c->finalized) return fail(c, -1, "junctions_finalize first");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    // one round: the round's list is the run-wide list; otherwise compact the merged mask
    if (c->mask_dirty || !c->marks_valid) {
        if ((rc = compact_mask(c, c->mask))) return rc;
        c->marks_valid = false; c->rmask_sums_valid = false;
    }
    if (c->n_marks > c->emit_cap || !c->emit_id) {
        if (c->emit_id) (void)hipFree(c->emit_id);
        c->emit_id = nullptr;
        const uint64_t cap = c->n_marks + c->n_marks / 8 + 16;
        HIPCHK(c, dev_malloc(c, (void **)&c->emit_id, cap * sizeof(int64_t)));
        c->emit_cap = cap;
    }
    HIPCHK(c, hipMemsetAsync(c->counters + 3, 0, sizeof(unsigned long long), c->stream));
    {
        Timed t(c, TPC_K_EMIT);
        if (tpc_launch_emit(make_launch(c), c->C, c->marks, c->n_marks, c->keys, c->n_keys, c->idtab, c->idtab_cap, c->emit_id, c->counters + 3))
            return fail(c, -1, "emit launch failed");
    }
    HIPCHK(c, hipGetLastError());
    uint64_t nv = 0;
    if ((rc = read_counter(c, 3, &nv))) return rc;
    c->n_emit = c->n_marks;
    if (n_marked) *n_marked = c->n_marks;
    if (n_valid) *n_valid = nv;
    return 0;
}

int tpc_emit_stream(tpc_ctx *c, const uint64_t *rec_start, const uint64_t *rec_len, uint32_t n_rec, uint64_t *n_bytes, uint64_t *n_records)
{
    if (!c || !c->finalized || !rec_start || !rec_len || !n_rec) return fail(c, -1, "tpc_emit first; records required");
    if (c->n_emit != c->n_marks || (c->n_marks && !c->emit_id)) return fail(c, -1, "tpc_emit first");
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t r_last = 0;
    for (uint32_t r = 0; r < n_rec; r++) if (rec_len[r] >= (uint64_t)c->P.k) r_last = r;
    uint64_t *d_rec = nullptr, *vscan = nullptr;
    void *plan = nullptr;
    int rc = 0;
    uint64_t totals[2] = {0, 0};
    if (hipMalloc((void **)&d_rec, 2 * (size_t)n_rec * sizeof(uint64_t)) != hipSuccess || dev_malloc(c, (void **)&vscan, (c->n_marks + 1) * sizeof(uint64_t)) != hipSuccess ||
        hipMalloc(&plan, tpc_stream_plan_bytes(n_rec)) != hipSuccess) rc = -10;
    if (rc == 0 && (hipMemcpyAsync(d_rec, rec_start, (size_t)n_rec * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                    hipMemcpyAsync(d_rec + n_rec, rec_len, (size_t)n_rec * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess)) rc = -10;
    if (rc == 0) {
        Timed t(c, TPC_K_STREAM);
        rc = tpc_launch_stream_plan(c->stream, d_rec, d_rec + n_rec, n_rec, c->P.k, c->marks, c->emit_id, c->n_marks, vscan, plan, r_last, totals);
        if (rc == 0) {
            const uint64_t bytes = totals[1] * 12;
            if (bytes > c->stream_cap) {
                if (c->stream_buf) (void)hipFree(c->stream_buf);
                c->stream_buf = nullptr; c->stream_cap = 0;
                if (dev_malloc(c, (void **)&c->stream_buf, bytes + 64) != hipSuccess) rc = -10; else c->stream_cap = bytes;
            }
            if (rc == 0 && bytes)
                rc = tpc_launch_stream_write(c->stream, d_rec, d_rec + n_rec, n_rec, c->P.k, c->marks, c->emit_id, c->n_marks, vscan, plan, r_last,
                                             c->n_keys + 42, c->stream_buf);
            c->stream_bytes = bytes;
        }
    }
    const hipError_t e = hipStreamSynchronize(c->stream);
    for (void *p : { (void *)d_rec, (void *)vscan, plan }) if (p) (void)hipFree(p);
    if (rc) return fail(c, rc, "junction stream failed (%d)", rc);
    HIPCHK(c, e);
    HIPCHK(c, hipGetLastError());
    if (n_bytes) *n_bytes = c->stream_bytes;
    if (n_records) *n_records = totals[0];
    return 0;
}

namespace {
void stream_part_release(tpc_ctx *c)
{
    for (void *p : { (void *)c->sp_rec, (void *)c->sp_vscan, (void *)c->sp_cnt, (void *)c->sp_lo, (void *)c->sp_flags }) if (p) (void)hipFree(p);
    c->sp_rec = c->sp_vscan = c->sp_cnt = c->sp_lo = nullptr;
    c->sp_flags = nullptr;
    c->sp_n_rec = 0;
}
}  // namespace

int tpc_shard_chunk(const tpc_ctx *c, uint64_t *chunk_lo, uint64_t *chunk_hi)
{
    if (!c || !chunk_lo || !chunk_hi || !c->n_text) return -1;
    const uint64_t W = c->sh_world, tiles = text_tiles512(c), chunk = (tiles + W - 1) / W;  // the split of tpc_shard_hash
    const uint64_t t0 = std::min<uint64_t>(tiles, (uint64_t)c->sh_rank * chunk), t1 = std::min<uint64_t>(tiles, t0 + chunk);
    *chunk_lo = t0 * 512 * TPC_RUN;
    *chunk_hi = c->sh_rank + 1 == W ? ~0ull : t1 * 512 * TPC_RUN;
    return 0;
}

int tpc_emit_stream_partial(tpc_ctx *c, const uint64_t *rec_start, const uint64_t *rec_len, uint32_t n_rec, uint64_t *cnt_host, uint32_t *flags_host)
{
    if (!c || !c->finalized || !rec_start || !rec_len || !n_rec || !cnt_host || !flags_host) return fail(c, -1, "tpc_emit first; records required");
    if (c->n_emit != c->n_marks || (c->n_marks && !c->emit_id)) return fail(c, -1, "tpc_emit first");
    HIPCHK(c, hipSetDevice(c->device));
    stream_part_release(c);
    if (hipMalloc((void **)&c->sp_rec, 2 * (size_t)n_rec * sizeof(uint64_t)) != hipSuccess || dev_malloc(c, (void **)&c->sp_vscan, (c->n_marks + 1) * sizeof(uint64_t)) != hipSuccess ||
        hipMalloc((void **)&c->sp_cnt, (size_t)n_rec * sizeof(uint64_t)) != hipSuccess || hipMalloc((void **)&c->sp_lo, (size_t)n_rec * sizeof(uint64_t)) != hipSuccess ||
        hipMalloc((void **)&c->sp_flags, (size_t)n_rec * sizeof(uint32_t)) != hipSuccess) { stream_part_release(c); return fail(c, -10, "out of device memory for the junction stream"); }
    c->sp_n_rec = n_rec;
    HIPCHK(c, hipMemcpyAsync(c->sp_rec, rec_start, (size_t)n_rec * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->sp_rec + n_rec, rec_len, (size_t)n_rec * 8, hipMemcpyHostToDevice, c->stream));
    int rc;
    {
        Timed t(c, TPC_K_STREAM);
        rc = tpc_launch_stream_partial(c->stream, c->sp_rec, c->sp_rec + n_rec, n_rec, c->P.k, c->marks, c->emit_id, c->n_marks, c->sp_vscan, c->sp_cnt, c->sp_flags, c->sp_lo);
    }
    if (rc) return fail(c, rc, "junction stream failed (%d)", rc);
    HIPCHK(c, hipMemcpy(cnt_host, c->sp_cnt, (size_t)n_rec * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(flags_host, c->sp_flags, (size_t)n_rec * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIPCHK(c, hipGetLastError());
    return 0;
}

int tpc_emit_stream_part(tpc_ctx *c, const uint64_t *rec_start, const uint64_t *rec_len, uint32_t n_rec, const uint32_t *gflags_host,
                         const uint64_t *e_scan_host, const uint64_t *s_scan_host, const uint64_t *before_host, uint32_t r_last,
                         uint64_t chunk_lo, uint64_t chunk_hi, uint64_t slot0, uint64_t n_slots, uint64_t *n_bytes)
{
    (void)rec_start; (void)rec_len;
    if (!c || !c->sp_rec || c->sp_n_rec != n_rec || !gflags_host || !e_scan_host || !s_scan_host || !before_host) return fail(c, -1, "tpc_emit_stream_partial first");
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t *d_e = nullptr, *d_s = nullptr, *d_b = nullptr;
    uint32_t *d_f = nullptr;
    int rc = 0;
    if (hipMalloc((void **)&d_e, ((size_t)n_rec + 1) * 8) != hipSuccess || hipMalloc((void **)&d_s, ((size_t)n_rec + 1) * 8) != hipSuccess ||
        hipMalloc((void **)&d_b, (size_t)n_rec * 8) != hipSuccess || hipMalloc((void **)&d_f, (size_t)n_rec * 4) != hipSuccess) rc = -10;
    const uint64_t bytes = n_slots * 12;
    if (rc == 0 && bytes > c->stream_cap) {
        if (c->stream_buf) (void)hipFree(c->stream_buf);
        c->stream_buf = nullptr; c->stream_cap = 0;
        if (dev_malloc(c, (void **)&c->stream_buf, bytes + 64) != hipSuccess) rc = -10; else c->stream_cap = bytes;
    }
    if (rc == 0 && (hipMemcpyAsync(d_e, e_scan_host, ((size_t)n_rec + 1) * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                    hipMemcpyAsync(d_s, s_scan_host, ((size_t)n_rec + 1) * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                    hipMemcpyAsync(d_b, before_host, (size_t)n_rec * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                    hipMemcpyAsync(d_f, gflags_host, (size_t)n_rec * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess)) rc = -10;
    if (rc == 0 && bytes) {
        Timed t(c, TPC_K_STREAM);
        rc = tpc_launch_stream_write_part(c->stream, c->sp_rec, c->sp_rec + n_rec, n_rec, c->P.k, c->marks, c->emit_id, c->n_marks, c->sp_vscan, c->sp_lo, d_f, d_e, d_s, d_b,
                                          r_last, c->n_keys + 42, chunk_lo, chunk_hi, slot0, c->stream_buf);
    }
    const hipError_t e = hipStreamSynchronize(c->stream);
    for (void *p : { (void *)d_e, (void *)d_s, (void *)d_b, (void *)d_f }) if (p) (void)hipFree(p);
    stream_part_release(c);
    if (rc) return fail(c, rc, "junction stream failed (%d)", rc);
    HIPCHK(c, e);
    HIPCHK(c, hipGetLastError());
    c->stream_bytes = bytes;
    if (n_bytes) *n_bytes = bytes;
    return 0;
}

int tpc_emit_stream_fetch(tpc_ctx *c, uint64_t offset, uint64_t nbytes, void *dst_host)
{   // no context state is modified: safe from several host threads at once
    if (!c || (nbytes && !dst_host) || offset + nbytes > c->stream_bytes) return -1;
    if (hipSetDevice(c->device) != hipSuccess) return -10;
    if (nbytes && hipMemcpy(dst_host, (const char *)c->stream_buf + offset, nbytes, hipMemcpyDeviceToHost) != hipSuccess) return -10;
    return 0;
}

int tpc_host_alloc(void **ptr, uint64_t bytes)
{
    if (!ptr) return -1;
    return hipHostMalloc(ptr, bytes, hipHostMallocDefault) == hipSuccess ? 0 : -10;
}

void tpc_host_free(void *ptr)
{
    if (ptr) (void)hipHostFree(ptr);
}

int tpc_emit_fetch(tpc_ctx *c, uint64_t *g_host, int64_t *id_host)
{
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n_emit) {
        HIPCHK(c, hipMemcpy(g_host, c->marks, c->n_emit * sizeof(uint64_t), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(id_host, c->emit_id, c->n_emit * sizeof(int64_t), hipMemcpyDeviceToHost));
    }
    return 0;
}

int tpc_emit_export(tpc_ctx *c, uint64_t *g_dev, int64_t *id_dev)
{
    if (!c || (c->n_emit && (!g_dev || !id_dev))) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n_emit) {
        HIPCHK(c, hipMemcpyAsync(g_dev, c->marks, c->n_emit * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(id_dev, c->emit_id, c->n_emit * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_emit_import(tpc_ctx *c, const uint64_t *g_dev, const int64_t *id_dev, uint64_t n)
{
    if (!c || !c->finalized || (n && (!g_dev || !id_dev))) return fail(c, -1, "junctions_finalize first");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure(c, c->marks, c->marks_cap, n);
    if (rc) return rc;
    if (n > c->emit_cap || !c->emit_id) {
        if (c->emit_id) (void)hipFree(c->emit_id);
        c->emit_id = nullptr; c->emit_cap = 0;
        const uint64_t cap = n + n / 8 + 16;
        HIPCHK(c, dev_malloc(c, (void **)&c->emit_id, cap * sizeof(int64_t)));
        c->emit_cap = cap;
    }
    if (n) {
        HIPCHK(c, hipMemcpyAsync(c->marks, g_dev, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->emit_id, id_dev, n * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->n_marks = n; c->n_emit = n;
    c->marks_valid = false; c->rmask_sums_valid = false;  // the list no longer is this rank's round list
    return 0;
}

// ------------------------------------------------------------------------------------------ address-sharded filter
int tpc_shard_config(tpc_ctx *c, uint32_t rank, uint32_t world)
{
    if (!c) return -1;
    if (world == 0 || (world & (world - 1)) || rank >= world) return fail(c, -1, "world must be a power of two and rank < world");
    HIPCHK(c, hipSetDevice(c->device));
    c->sh_rank = rank; c->sh_world = world;
    c->sh_have[0] = c->sh_have[1] = false;
    c->pending_apply = false;  // the filter is about to be re-cut
    if (c->have_params) {
        const uint64_t fw = filter_words_for(c->P.L, c->opt_replicate ? 1 : world);
        if (fw != c->filter_words) {
            if (c->filter) (void)hipFree(c->filter);
            c->filter = nullptr; c->filter_words = 0;
            HIPCHK(c, hipMalloc((void **)&c->filter, fw * sizeof(uint32_t)));
            c->filter_words = fw;
        }
        c->filter_zero_pending = true;
    }
    return 0;
}

int tpc_shard_plan(tpc_ctx *c, int pass, uint64_t lo, uint64_t hi, uint64_t *geom)
{
    if (!c || !c->have_params || !c->bases || !geom) return fail(c, -1, "set_params and seq_upload first");
    if (pass != TPC_SHARD_INSERT && pass != TPC_SHARD_QUERY) return fail(c, -1, "bad pass");
    HIPCHK(c, hipSetDevice(c->device));
    if (!part_hash_supported(c))
        return fail(c, -1, "a sharded filter needs the partitioned hash kernels: q=%d, L=%d, slice_bits=%d are outside what they cover (1..8 functions, or 9..16 with L - slice_bits <= 24)",
                    c->P.q, c->P.L, c->opt_slice_bits);
    const bool gated = !(lo == 0 && hi >= c->P.lmask);
    if (pass == TPC_SHARD_INSERT) { int rc0 = flush_pending_apply(c); if (rc0) return rc0; }  // (its regions are about to be re-planned)
    if (c->opt_shard_periodic) ensure_periodic(c);
    const uint64_t W = c->sh_world, tiles = text_tiles512(c);
    const uint64_t per_total = (tiles + W - 1) / W;
    const double m = gated ? range_mass(c, lo, hi) : 1.0;
    uint64_t per = per_total;
    if (pass == TPC_SHARD_INSERT) {
        const double frac = gated ? std::min(1.0, (1.0 - (1.0 - m) * (1.0 - m)) * 1.15) : 1.0;
        TpcPartPlan &pl = c->sh_ipl;
        for (uint64_t batches = 1;; batches = next_batches(batches)) {
            per = (per_total + batches - 1) / batches;
            if (!tpc_part_plan_sharded(c->P.L, c->P.q, c->opt_slice_bits, per, frac, c->sh_rank, c->sh_world, pl, c->opt_part_levels, c->opt_shard_tight != 0))
                return fail(c, -1, "no sharded partition geometry for L=%d, slice_bits=%d, world=%u", c->P.L, c->opt_slice_bits, c->sh_world);
            if ((int64_t)(2 * tpc_part_buf1_bytes(pl) + tpc_part_buf2_bytes(pl) + tpc_part_buf3_bytes(pl)) <= part_budget(c) || (int64_t)per <= c->opt_part_min_tiles) break;
        }
        const size_t need[6] = { 0, 0, tpc_part_buf2_bytes(pl), tpc_part_cnt2_bytes(pl), pl.ovf_cap * sizeof(uint64_t), 32 * sizeof(unsigned long long) };
        for (int i = 2; i < 6; i++) if (!ensure_pbuf(c, i, need[i])) return fail(c, -10, "out of device memory for the partition buffers");
        if (!ensure_pbuf(c, 12, need[4]) || !ensure_pbuf(c, 13, need[5])) return fail(c, -10, "out of device memory for the partition buffers");  // the apply-side list
        if (pl.b3 && (!ensure_pbuf(c, 9, tpc_part_buf3_bytes(pl)) || !ensure_pbuf(c, 10, tpc_part_cnt3_bytes(pl)))) return fail(c, -10, "out of device memory for the partition buffers");
        pl.buf2 = (uint32_t *)c->pbuf[2]; pl.cnt2 = (uint32_t *)c->pbuf[3]; pl.ovf = (uint64_t *)c->pbuf[4]; pl.ovf_cur = (unsigned long long *)c->pbuf[5];
        pl.buf3 = (uint32_t *)c->pbuf[9]; pl.cnt3 = (uint32_t *)c->pbuf[10];
        // Deferred apply (as on one GPU, section 3.2): a round whose insert is ONE batch stops after its level-2 binning, the regions kept
        // aside (the query's plan reuses the shared ones), and the first lookup of the round's query builds every owned slice itself
        // (k_apply_lookup on the shard): the shard is written once and not read back.  Room for the regions permitting.
        c->sh_defer = false;
        if (c->opt_fuse && per == per_total && pl.b3 == 0) {
            const size_t want[2] = { tpc_part_buf2_bytes(pl), tpc_part_cnt2_bytes(pl) };
            bool ok = true;
            for (int i = 0; i < 2 && ok; i++) {
                if (want[i] <= c->ikeep_bytes[i]) continue;
                if (c->ikeep[i]) (void)hipFree(c->ikeep[i]);
                c->ikeep[i] = nullptr; c->ikeep_bytes[i] = 0;
                if (hipMalloc(&c->ikeep[i], want[i]) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
                c->ikeep_bytes[i] = want[i];
            }
            c->sh_defer = ok;
        }
        geom[2] = tpc_part_buf1_bytes(pl) / W; geom[3] = tpc_part_cnt1_bytes(pl) / W;
        geom[4] = 0; geom[5] = pl.ovf_cap; geom[6] = 8;
        geom[7] = pl.slice_bits; geom[8] = pl.b1; geom[9] = pl.b2; geom[10] = pl.perm_mult; geom[11] = pl.perm_inv; geom[12] = pl.b3;
    } else {
        TpcQPlan &pl = c->sh_qpl;
        for (uint64_t batches = 1;; batches = next_batches(batches)) {
            per = (per_total + batches - 1) / batches;
            uint32_t log_w = 0;
            while ((1u << log_w) < c->sh_world) ++log_w;
            const bool fits = per * (uint64_t)(512 * TPC_RUN) <= (1ull << (30 - log_w));  // survivor ids: source rank + position relative to its batch in 30 bits
            const bool ok = fits && tpc_qpart_plan_sharded(c->P.L, c->opt_slice_bits, per, gated ? std::min(1.0, m * 1.15) : 1.0, c->sh_rank, c->sh_world, pl, c->opt_part_levels, c->opt_shard_tight != 0);
            if (!ok && fits) return fail(c, -1, "no sharded partition geometry for L=%d, slice_bits=%d, world=%u", c->P.L, c->opt_slice_bits, c->sh_world);
            if (ok && ((int64_t)(2 * tpc_qpart_bytes(pl, 0) + tpc_qpart_bytes(pl, 2) + tpc_qpart_bytes(pl, 9)) <= part_budget(c) || (int64_t)per <= c->opt_part_min_tiles)) break;
            if (per <= 1) return fail(c, -1, "text too large for the sharded query geometry");
        }
        for (int i = 2; i < 12; i++) if (i != 4 && i != 5 && tpc_qpart_bytes(pl, i) && !ensure_pbuf(c, i, tpc_qpart_bytes(pl, i))) return fail(c, -10, "out of device memory for the partition buffers");
        for (int i = 14; i < 18; i++) if (!ensure_pbuf(c, i, tpc_qpart_bytes(pl, 4 + (i & 1)))) return fail(c, -10, "out of device memory for the partition buffers");  // the query's own overflow lists
        pl.buf3 = (uint64_t *)c->pbuf[9]; pl.cnt3 = (uint32_t *)c->pbuf[10]; pl.off3 = (const uint64_t *)c->pbuf[11];
        if (pl.b3 && c->off3_uploaded != pl.off3_host) {
            HIPCHK(c, hipMemcpy(c->pbuf[11], pl.off3_host.data(), pl.off3_host.size() * 8, hipMemcpyHostToDevice));
            c->off3_uploaded = pl.off3_host;
        }
        pl.buf2 = (uint64_t *)c->pbuf[2]; pl.cnt2 = (uint32_t *)c->pbuf[3]; pl.ovf = (uint64_t *)c->pbuf[14]; pl.ovf_cur = (unsigned long long *)c->pbuf[15];
        pl.surv = (uint64_t *)c->pbuf[6]; pl.surv_cur = (unsigned long long *)c->pbuf[7]; pl.off2 = (const uint64_t *)c->pbuf[8];
        if (c->off2_uploaded != pl.off2_host) {
            HIPCHK(c, hipMemcpy(c->pbuf[8], pl.off2_host.data(), pl.off2_host.size() * 8, hipMemcpyHostToDevice));
            c->off2_uploaded = pl.off2_host;
        }
        geom[2] = tpc_qpart_bytes(pl, 0) / W; geom[3] = tpc_qpart_bytes(pl, 1) / W;
        geom[4] = 64 * pl.surv_cap; geom[5] = pl.ovf_cap; geom[6] = 16;
        geom[7] = pl.slice_bits; geom[8] = pl.b1; geom[9] = pl.b2; geom[10] = pl.perm_mult; geom[11] = pl.perm_inv; geom[12] = pl.b3;
    }
    // Tile ownership is contiguous: rank r hashes the chunk [r * per_total, (r + 1) * per_total) of the text, `per` tiles per
    // batch -- so a rank needs only its chunk of the packed text (option text_window) -- and a query entry names its source
    // rank in the top log2(world) bits of its 30-bit position field, the rest being the position relative to that rank's batch.
    c->sh_per[pass] = per;
    c->sh_batches[pass] = (per_total + per - 1) / per;
    c->sh_have[pass] = true;
    c->sh_have[1 - pass] = false;  // the two passes share the partition buffers
    geom[0] = c->sh_batches[pass]; geom[1] = per;
    for (int i = 13; i < 16; i++) geom[i] = 0;
    return 0;
}

int tpc_shard_plan_both(tpc_ctx *c, uint64_t lo, uint64_t hi, uint64_t *geom_insert, uint64_t *geom_query)
{   // both passes planned together: the shared level-2 / level-3 buffers hold the larger of the two needs and BOTH plans stay valid,
    // so that the query's hash may run (tpc_shard_hash_begin) while the insert of the same round is still being exchanged and applied
    int rc = tpc_shard_plan(c, TPC_SHARD_INSERT, lo, hi, geom_insert);
    if (rc) return rc;
    if ((rc = tpc_shard_plan(c, TPC_SHARD_QUERY, lo, hi, geom_query))) return rc;
    TpcPartPlan &pl = c->sh_ipl;  // the query's plan may have grown (reallocated) what the insert's plan pointed at
    pl.buf2 = (uint32_t *)c->pbuf[2]; pl.cnt2 = (uint32_t *)c->pbuf[3]; pl.ovf = (uint64_t *)c->pbuf[4]; pl.ovf_cur = (unsigned long long *)c->pbuf[5];
    pl.buf3 = (uint32_t *)c->pbuf[9]; pl.cnt3 = (uint32_t *)c->pbuf[10];
    if (tpc_part_buf2_bytes(pl) > c->pbytes[2] || tpc_part_cnt2_bytes(pl) > c->pbytes[3] || (pl.b3 && (tpc_part_buf3_bytes(pl) > c->pbytes[9] || tpc_part_cnt3_bytes(pl) > c->pbytes[10])))
        return fail(c, -10, "partition buffers smaller than the insert's plan");
    c->sh_have[TPC_SHARD_INSERT] = true;
    return 0;
}

namespace {

// The level-1 hash of one batch of a sharded pass.  async = false: on the context's stream, synchronised, *n_overflow set.
// async = true (tpc_shard_hash_begin): enqueued on the context's second stream beside whatever the main stream is doing --
// the hash reads the text and writes the caller's send buffers, the pass' PRODUCED overflow list and (query) the round mask,
// nothing an exchange or an apply of another batch or of the other pass touches -- and tpc_shard_hash_end collects it.
int shard_hash_impl(tpc_ctx *c, int pass, uint64_t batch, uint64_t lo, uint64_t hi, void *send_regions, void *send_counts, uint64_t *n_overflow, bool async)
{
    if (!c || (pass != TPC_SHARD_INSERT && pass != TPC_SHARD_QUERY) || !c->sh_have[pass]) return fail(c, -1, "tpc_shard_plan for this pass first");
    if (!send_regions || !send_counts || batch >= c->sh_batches[pass]) return fail(c, -1, "bad arguments");
    if (c->sh_async[pass]) return fail(c, -1, "a hash of this pass is still in flight: tpc_shard_hash_end first");
    HIPCHK(c, hipSetDevice(c->device));
    if (async && !c->stream2) HIPCHK(c, hipStreamCreate(&c->stream2));
    hipStream_t st = async ? c->stream2 : c->stream;
    const bool gated = !(lo == 0 && hi >= c->P.lmask);
    const uint64_t W = c->sh_world, per = c->sh_per[pass], tiles = text_tiles512(c);
    const uint64_t chunk = (tiles + W - 1) / W, c0 = std::min(tiles, c->sh_rank * chunk), c1 = std::min(tiles, c0 + chunk);
    const uint64_t t0 = c0 + batch * per;
    const uint64_t n = t0 < c1 ? std::min<uint64_t>(per, c1 - t0) : 0;
    unsigned long long *ov = c->sh_ov_host[pass];
    ov[0] = ov[1] = 0;
    TpcLaunch a = c->opt_shard_periodic ? make_launch_periodic(c) : make_launch(c);
    a.stream = st;
    if (pass == TPC_SHARD_INSERT) {
        TpcPartPlan pl = c->sh_ipl;
        pl.tile0 = t0; pl.n_tiles = n;
        pl.buf1 = (uint32_t *)send_regions; pl.cnt1 = (uint32_t *)send_counts;
        HIPCHK(c, hipMemsetAsync(pl.ovf_cur, 0, 32 * sizeof(unsigned long long), st));
        if (async) {
            if (tpc_launch_insert_part_hash(a, pl, lo, hi, gated, nullptr)) return fail(c, -1, "hash launch failed");
        } else {
            Timed t(c, TPC_K_SHARD_HASH);
            if (tpc_launch_insert_part_hash(a, pl, lo, hi, gated, nullptr)) return fail(c, -1, "hash launch failed");
        }
        HIPCHK(c, hipMemcpyAsync(ov, pl.ovf_cur, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
#ifdef TPC_BINS3_DEBUG
        if (!async) {
            unsigned long long d[12];
            (void)hipMemcpy(d, pl.ovf_cur, sizeof d, hipMemcpyDeviceToHost);
            fprintf(stderr, "[bins3] insert hash: ring-lost %llu region-lost %llu retry-iterations %llu waited-and-stored %llu (overflow list %llu) ppr=%d cap1=%llu\n", d[8], d[9], d[10], d[11], d[0], pl.pos_per_round, (unsigned long long)pl.cap1);
        }
#endif
    } else {
        TpcQPlan pl = c->sh_qpl;
        pl.tile0 = t0; pl.n_tiles = n; pl.tile0_global = t0;  // positions in the entries are relative to this rank's batch
        pl.buf1 = (uint64_t *)send_regions; pl.cnt1 = (uint32_t *)send_counts;
        // (the hash does not read the filter; the lookup of tpc_shard_apply materialises a pending reset before it probes)
        if (!async && !c->pending_apply) { int rc0 = materialize_reset(c); if (rc0) return rc0; }  // (a deferred apply waits for the lookup)
        HIPCHK(c, hipMemsetAsync(pl.ovf_cur, 0, 32 * sizeof(unsigned long long), st));
        if (!async) HIPCHK(c, hipMemsetAsync(pl.surv_cur, 0, TPC_SURV_CUR_WORDS * sizeof(unsigned long long), st));  // (async: tpc_shard_apply zeroes the survivor cursors itself)
        // marks of this batch's survivors land anywhere in the batch, the hash kernel only rewrites this rank's tiles
        if (batch == 0) HIPCHK(c, hipMemsetAsync(c->rmask, 0, c->n_words_alloc * sizeof(uint32_t), st));
        c->marks_valid = false; c->rmask_sums_valid = false;
        if (async) {
            if (tpc_launch_query_part_hash(a, pl, c->rmask, lo, hi, gated)) return fail(c, -1, "hash launch failed");
        } else {
            Timed t(c, TPC_K_SHARD_HASH);
            if (tpc_launch_query_part_hash(a, pl, c->rmask, lo, hi, gated)) return fail(c, -1, "hash launch failed");
        }
        HIPCHK(c, hipMemcpyAsync(ov, pl.ovf_cur, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipGetLastError());
    if (async) { c->sh_async[pass] = true; return 0; }
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the caller's collective runs on another stream
    if (n_overflow) *n_overflow = ov[1] ? (1ull << 62) : (uint64_t)ov[0];
    return 0;
}

}  // namespace

int tpc_shard_hash(tpc_ctx *c, int pass, uint64_t batch, uint64_t lo, uint64_t hi, void *send_regions, void *send_counts, uint64_t *n_overflow)
{
    return shard_hash_impl(c, pass, batch, lo, hi, send_regions, send_counts, n_overflow, false);
}

int tpc_shard_hash_begin(tpc_ctx *c, int pass, uint64_t batch, uint64_t lo, uint64_t hi, void *send_regions, void *send_counts)
{
    return shard_hash_impl(c, pass, batch, lo, hi, send_regions, send_counts, nullptr, true);
}

int tpc_shard_hash_end(tpc_ctx *c, int pass, uint64_t *n_overflow)
{
    if (!c || (pass != TPC_SHARD_INSERT && pass != TPC_SHARD_QUERY)) return -1;
    if (!c->sh_async[pass]) return fail(c, -1, "no hash of this pass in flight (tpc_shard_hash_begin)");
    HIPCHK(c, hipSetDevice(c->device));
    c->sh_async[pass] = false;
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    const unsigned long long *ov = c->sh_ov_host[pass];
    if (n_overflow) *n_overflow = ov[1] ? (1ull << 62) : (uint64_t)ov[0];
    return 0;
}

int tpc_shard_periodic_copy(tpc_ctx *c)
{
    if (!c || !c->rmask) return fail(c, -1, "no text");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->opt_shard_periodic && c->periodic_valid && c->periodic_any_q && c->periodic) {
        tpc_launch_periodic_copy(c->stream, c->rmask, c->periodic, c->periodic + c->n_words_alloc, c->n_words_alloc, c->n_words);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->marks_valid = false; c->rmask_sums_valid = false;
    }
    return 0;
}

int tpc_shard_overflow_get(tpc_ctx *c, int pass, void *dst, uint64_t n)
{
    if (!c || (pass != TPC_SHARD_INSERT && pass != TPC_SHARD_QUERY) || !c->sh_have[pass] || (!dst && n)) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t cap = pass == TPC_SHARD_INSERT ? c->sh_ipl.ovf_cap : c->sh_qpl.ovf_cap;
    if (n > cap) return fail(c, -1, "overflow list holds at most %llu entries", (unsigned long long)cap);
    if (n) HIPCHK(c, hipMemcpy(dst, c->pbuf[pass == TPC_SHARD_INSERT ? 4 : 14], n * (pass == TPC_SHARD_INSERT ? 8 : 16), hipMemcpyDeviceToDevice));  // the PRODUCED list
    return 0;
}

int tpc_shard_overflow_set(tpc_ctx *c, int pass, const void *src, uint64_t n)
{
    if (!c || (pass != TPC_SHARD_INSERT && pass != TPC_SHARD_QUERY) || !c->sh_have[pass] || (!src && n)) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t cap = pass == TPC_SHARD_INSERT ? c->sh_ipl.ovf_cap : c->sh_qpl.ovf_cap;
    if (n > cap) return fail(c, -1, "gathered overflow lists (%llu entries) exceed the capacity %llu: skew beyond what the sharded path handles",
                             (unsigned long long)n, (unsigned long long)cap);
    const int list = pass == TPC_SHARD_INSERT ? 12 : 16;  // the APPLIED list: what the apply side extends and consumes
    if (n) HIPCHK(c, hipMemcpy(c->pbuf[list], src, n * (pass == TPC_SHARD_INSERT ? 8 : 16), hipMemcpyDeviceToDevice));
    unsigned long long cur[32] = {n, 0};
    HIPCHK(c, hipMemcpy(c->pbuf[list + 1], cur, sizeof cur, hipMemcpyHostToDevice));
    c->sh_ovf_set[pass] = true;
    return 0;
}

namespace {

// level-1 regions of a sharded pass: [rank][local bucket][workgroup], the same number on the sending and the receiving side
uint32_t shard_regions(const tpc_ctx *c, int pass)
{
    return pass == TPC_SHARD_INSERT ? c->sh_ipl.nwg1 << c->sh_ipl.b1 : c->sh_qpl.nwg1 << c->sh_qpl.b1;
}

bool ensure_shard_offsets(tpc_ctx *c, uint32_t n_regions)
{
    const size_t need = ((size_t)n_regions + 1) * sizeof(uint64_t);
    if (c->sh_off_bytes >= need) return true;
    if (c->sh_off) (void)hipFree(c->sh_off);
    c->sh_off = nullptr; c->sh_off_bytes = 0;
    if (hipMalloc((void **)&c->sh_off, need) != hipSuccess) { (void)hipGetLastError(); return false; }
    c->sh_off_bytes = need;
    return true;
}

int shard_apply_impl(tpc_ctx *c, int pass, uint64_t batch, const void *recv_regions, const void *recv_counts, bool packed, uint64_t *n_survivors,
                     const void *own_regions = nullptr, const void *own_counts = nullptr);

}  // namespace

int tpc_shard_apply(tpc_ctx *c, int pass, uint64_t batch, const void *recv_regions, const void *recv_counts, uint64_t *n_survivors)
{
    return shard_apply_impl(c, pass, batch, recv_regions, recv_counts, false, n_survivors);
}

int tpc_shard_apply_packed(tpc_ctx *c, int pass, uint64_t batch, const void *recv_packed, const void *recv_counts, uint64_t *n_survivors)
{
    return shard_apply_impl(c, pass, batch, recv_packed, recv_counts, true, n_survivors);
}

int tpc_shard_apply_inplace(tpc_ctx *c, int pass, uint64_t batch, const void *recv_regions, const void *recv_counts, const void *send_regions,
                            const void *send_counts, uint64_t *n_survivors)
{   // block `rank` of the receive buffers is never read: the entries this rank hashed for itself are taken from the send buffers
    if (!c || !send_regions || !send_counts) return fail(c, -1, "bad arguments");
    if (c->sh_world == 1) return shard_apply_impl(c, pass, batch, send_regions, send_counts, false, n_survivors);  // nothing was exchanged
    return shard_apply_impl(c, pass, batch, recv_regions, recv_counts, false, n_survivors, send_regions, send_counts);
}

int tpc_shard_pack(tpc_ctx *c, int pass, const void *send_regions, const void *send_counts, void *packed, uint64_t *bytes_per_dest)
{
    if (!c || (pass != TPC_SHARD_INSERT && pass != TPC_SHARD_QUERY) || !c->sh_have[pass]) return fail(c, -1, "tpc_shard_plan for this pass first");
    if (!send_regions || !send_counts || !packed || !bytes_per_dest) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t n = shard_regions(c, pass), W = c->sh_world, block = n / W;
    const uint32_t eb = pass == TPC_SHARD_INSERT ? 4 : 8;
    const uint64_t cap1 = pass == TPC_SHARD_INSERT ? c->sh_ipl.cap1 : c->sh_qpl.cap1;
    if (!ensure_shard_offsets(c, n)) return fail(c, -10, "out of device memory for the region offsets");
    const TpcLaunch a = make_launch(c);
    tpc_launch_region_offsets(a, (const uint32_t *)send_counts, n, c->sh_off);
    if (tpc_launch_region_pack(a, send_regions, cap1, eb, (const uint32_t *)send_counts, c->sh_off, n, packed)) return fail(c, -1, "pack launch failed");
    // the regions of destination d are the index range [d * block, (d + 1) * block): its share of the packed buffer
    std::vector<uint64_t> edge(W + 1);
    HIPCHK(c, hipMemcpy2DAsync(edge.data(), sizeof(uint64_t), c->sh_off, (size_t)block * sizeof(uint64_t), sizeof(uint64_t), W + 1, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the caller's collective runs on another stream
    for (uint32_t d = 0; d < W; d++) bytes_per_dest[d] = (edge[d + 1] - edge[d]) * eb;
    return 0;
}

namespace {

int shard_apply_impl(tpc_ctx *c, int pass, uint64_t batch, const void *recv_regions, const void *recv_counts, bool packed, uint64_t *n_survivors,
                     const void *own_regions, const void *own_counts)
{
    if (!c || (pass != TPC_SHARD_INSERT && pass != TPC_SHARD_QUERY) || !c->sh_have[pass]) return fail(c, -1, "tpc_shard_plan for this pass first");
    if (!recv_regions || !recv_counts || batch >= c->sh_batches[pass]) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t *roff1 = nullptr;
    if (packed) {  // the blocks of the source ranks follow one another: the scan over [source][local bucket][workgroup] places every region
        const uint32_t n = shard_regions(c, pass);
        if (!ensure_shard_offsets(c, n)) return fail(c, -10, "out of device memory for the region offsets");
        tpc_launch_region_offsets(make_launch(c), (const uint32_t *)recv_counts, n, c->sh_off);
        roff1 = c->sh_off;
    }
    unsigned long long ov[2] = {0, 0};
    // the apply side's own overflow list: the gathered entries when tpc_shard_overflow_set ran for this batch, empty otherwise
    const int alist = pass == TPC_SHARD_INSERT ? 12 : 16;
    if (!c->sh_ovf_set[pass]) HIPCHK(c, hipMemsetAsync(c->pbuf[alist + 1], 0, 32 * sizeof(unsigned long long), c->stream));
    c->sh_ovf_set[pass] = false;
    if (pass == TPC_SHARD_INSERT) {
        TpcPartPlan pl = c->sh_ipl;
        pl.ovf = (uint64_t *)c->pbuf[alist]; pl.ovf_cur = (unsigned long long *)c->pbuf[alist + 1];
        pl.rbuf1 = (const uint32_t *)recv_regions; pl.rcnt1 = (const uint32_t *)recv_counts; pl.roff1 = roff1;
        pl.rown1 = (const uint32_t *)own_regions; pl.rowncnt1 = (const uint32_t *)own_counts;
        { int rc0 = flush_pending_apply(c); if (rc0) return rc0; }  // (an earlier insert nobody looked up: OR on top of it)
        const bool fresh = c->filter_zero_pending;
        const bool defer = c->sh_defer && c->sh_batches[pass] == 1 && pl.b3 == 0;
        if (defer) { pl.buf2 = (uint32_t *)c->ikeep[0]; pl.cnt2 = (uint32_t *)c->ikeep[1]; }
        {
            Timed t(c, TPC_K_SHARD_APPLY);
            if (defer ? tpc_launch_insert_part_split(make_launch(c), pl) : tpc_launch_insert_part_apply(make_launch(c), pl, fresh)) return fail(c, -1, "apply launch failed");
        }
        HIPCHK(c, hipMemcpyAsync(ov, pl.ovf_cur, sizeof ov, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (ov[1]) return fail(c, -20, "overflow list overflowed (address skew beyond what the sharded path handles)");
        if (defer) {
            // the overflow entries (this rank's and the gathered ones of the others) grouped by local slice for the fused kernel; they also
            // stay where they are, for an apply that has to be completed without a lookup (flush_pending_apply)
            bool keep = ov[0] <= TPC_FUSE_MAX_OVF;
            const uint32_t n_slices = (1u << (pl.b1 + pl.b2)) / pl.world;
            if (keep && ov[0]) {
                if (c->ikeep_ovf_cap < ov[0]) {
                    if (c->ikeep_ovf) (void)hipFree(c->ikeep_ovf);
                    c->ikeep_ovf = nullptr; c->ikeep_ovf_cap = 0;
                    const uint64_t cap = std::max<uint64_t>(4096, ov[0] + ov[0] / 4);
                    if (hipMalloc((void **)&c->ikeep_ovf, 2 * cap * sizeof(uint64_t)) == hipSuccess) c->ikeep_ovf_cap = cap; else { (void)hipGetLastError(); keep = false; }
                }
                if (keep && c->iovf_slices < n_slices) {
                    if (c->iovf_cnt) (void)hipFree(c->iovf_cnt);
                    if (c->iovf_off) (void)hipFree(c->iovf_off);
                    c->iovf_cnt = nullptr; c->iovf_off = nullptr; c->iovf_slices = 0;
                    if (hipMalloc((void **)&c->iovf_cnt, 2 * (size_t)n_slices * sizeof(uint32_t)) == hipSuccess &&
                        hipMalloc((void **)&c->iovf_off, ((size_t)n_slices + 1) * sizeof(uint64_t)) == hipSuccess) c->iovf_slices = n_slices;
                    else { (void)hipGetLastError(); keep = false; }
                }
                if (keep && tpc_launch_ovf_by_slice(make_launch(c), pl.ovf, ov[0], pl.slice_bits, n_slices, c->iovf_cnt, c->iovf_cnt + n_slices, c->iovf_off,
                                                    c->ikeep_ovf + c->ikeep_ovf_cap, pl.rank, pl.world, pl.b2)) return fail(c, -1, "overflow grouping launch failed");
            }
            if (keep) { c->pending_apply = true; c->pending_shard = true; c->pending_fresh = fresh; c->pending_pl = pl; c->pending_novf = ov[0]; }
            else {
                Timed t(c, TPC_K_SHARD_APPLY);
                if (tpc_launch_insert_part_apply_only(make_launch(c), pl, fresh)) return fail(c, -1, "apply launch failed");
                HIPCHK(c, hipGetLastError());
            }
        }
        c->filter_zero_pending = false;  // every owned slice has been (or is about to be) written
        if (n_survivors) *n_survivors = 0;
        return 0;
    }
    TpcQPlan pl = c->sh_qpl;
    pl.ovf = (uint64_t *)c->pbuf[alist]; pl.ovf_cur = (unsigned long long *)c->pbuf[alist + 1];
    pl.rbuf1 = (const uint64_t *)recv_regions; pl.rcnt1 = (const uint32_t *)recv_counts; pl.roff1 = roff1;
    pl.rown1 = (const uint64_t *)own_regions; pl.rowncnt1 = (const uint32_t *)own_counts;
    // the deferred apply of this round's insert: the lookup builds the owned slices itself when the geometry still matches
    const bool fused = c->pending_apply && c->pending_shard && pl.b3 == 0 && pl.slice_bits == c->pending_pl.slice_bits && pl.b1 == c->pending_pl.b1 &&
                       pl.b2 == c->pending_pl.b2 && pl.world == c->pending_pl.world && pl.fmt == 0 && c->pending_pl.fmt2 == 0;
    if (!fused) { int rc0 = materialize_reset(c); if (rc0) return rc0; }  // (a query whose hash ran ahead of the round's insert: tpc_shard_hash_begin)
    HIPCHK(c, hipMemsetAsync(pl.surv_cur, 0, TPC_SURV_CUR_WORDS * sizeof(unsigned long long), c->stream));
    const uint64_t W = c->sh_world, per = c->sh_per[pass], tiles = text_tiles512(c);
    const uint64_t chunk = (tiles + W - 1) / W;
    pl.tile0_global = std::min(tiles, c->sh_rank * chunk) + batch * per;
    c->sh_qpl.tile0_global = pl.tile0_global;  // the survivors that come BACK to this rank (tpc_shard_survivor_sources) are relative to its own batch
    unsigned long long cur[65];
    {
        Timed t(c, TPC_K_SHARD_APPLY);
        if (fused) {
            c->pending_apply = false; c->pending_shard = false;
            c->stat_fused++;
            if (tpc_launch_query_part_fused_lookup(make_launch(c), pl, c->pending_pl, c->pending_fresh, c->pending_novf ? c->ikeep_ovf + c->ikeep_ovf_cap : nullptr,
                                                   c->pending_novf ? c->iovf_off : nullptr)) return fail(c, -1, "fused lookup launch failed");
        } else if (tpc_launch_query_part_lookup(make_launch(c), pl)) return fail(c, -1, "lookup launch failed");
    }
    HIPCHK(c, hipMemcpyAsync(ov, pl.ovf_cur, sizeof ov, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cur, pl.surv_cur, sizeof cur, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (ov[1] || cur[64]) {
        unsigned long long most = 0;
        for (int i = 0; i < 64; i++) most = std::max(most, cur[i]);
        return fail(c, -20, "overflow or survivor list overflowed (address skew beyond what the sharded path handles): %llu overflow entries of %llu, fullest survivor list %llu of %llu",
                    ov[0], (unsigned long long)pl.ovf_cap, most, (unsigned long long)pl.surv_cap);
    }
    uint64_t ns = 0;
    for (int i = 0; i < 64; i++) ns += std::min<uint64_t>(cur[i], pl.surv_cap);
    c->sh_nsurv = ns;
    if (n_survivors) *n_survivors = ns;
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------ combined exchange (tpc_combine.hip)
namespace {

// slices / windows / directory entries of the geometry the combined calls agree on
uint32_t cmb_slices(const TpcPartPlan &g) { return 1u << (g.b1 + g.b2); }

int cmb_sources(tpc_ctx *c, uint32_t n_src, const uint16_t *payload, const uint64_t *base_host, const uint64_t *dir, uint64_t dir_stride, uint32_t n_owner, TpcListSrc &ls)
{
    if (n_src == 0 || n_src > 4096 || !payload || !base_host || !dir) return fail(c, -1, "bad arguments");
    if (n_owner && ((n_owner & (n_owner - 1)) || n_src % n_owner)) return fail(c, -1, "bad arguments: n_owner must be a power of two dividing n_src");
    if (!c->cmb_base) HIPCHK(c, hipMalloc((void **)&c->cmb_base, 4096 * sizeof(uint64_t)));
    HIPCHK(c, hipMemcpy(c->cmb_base, base_host, n_src * sizeof(uint64_t), hipMemcpyHostToDevice));
    ls.payload = payload; ls.base = c->cmb_base; ls.dir = dir; ls.dir_stride = dir_stride; ls.n_src = n_src; ls.n_owner = n_owner;
    return 0;
}

}  // namespace

int tpc_combine_info(tpc_ctx *c, uint32_t n_dest, uint64_t *info)
{
    if (!c || !info || !c->have_params || n_dest == 0 || (n_dest & (n_dest - 1))) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 0; i < 8; i++) info[i] = 0;
    // sparse lists need the insert still in its level-2 regions (32-bit entries, two levels, one batch: the deferred apply)
    if (!(c->pending_apply && !c->pending_shard && !c->pending_lists && c->pending_pl.b3 == 0 && c->pending_pl.fmt2 == 0 && n_dest <= (1u << c->pending_pl.b1))) return 0;
    const TpcPartPlan &g = c->pending_pl;
    const uint32_t n_slices = cmb_slices(g), n_win = tpc_list_windows(g.slice_bits), nb2 = 1u << g.b2;
    // upper bound of a destination block: the entries (duplicates included) of its slices, every window's list rounded up to a unit
    std::vector<uint32_t> cnt((size_t)n_slices * g.wpb);
    std::vector<uint64_t> ovf_off;
    HIPCHK(c, hipMemcpyAsync(cnt.data(), g.cnt2, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (c->pending_novf) {
        ovf_off.resize((size_t)n_slices + 1);
        HIPCHK(c, hipMemcpyAsync(ovf_off.data(), c->iovf_off, ovf_off.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<uint64_t> units(n_dest, 0);
    for (uint32_t b1 = 0; b1 < (1u << g.b1); b1++)
        for (uint32_t b2 = 0; b2 < nb2; b2++) {
            uint64_t e = 0;
            for (uint32_t j = 0; j < g.wpb; j++) e += cnt[((size_t)b1 * g.wpb + j) * nb2 + b2];
            const uint32_t sp = (b1 << g.b2) | b2;
            if (c->pending_novf) e += ovf_off[sp + 1] - ovf_off[sp];
            e = std::min<uint64_t>(e, (uint64_t)1 << g.slice_bits);
            units[b1 & (n_dest - 1)] += (e + 7) / 8 + n_win;
        }
    // (+ the chunks the persistent export claims per workgroup and destination: tpc_combine.hip:CB_CHUNK = 512 units, at most 1024 workgroups)
    info[0] = 1; info[1] = n_slices; info[2] = n_win; info[3] = *std::max_element(units.begin(), units.end()) + 1024 * 512;
    info[4] = (uint64_t)g.slice_bits; info[5] = (uint64_t)g.b1; info[6] = (uint64_t)g.b2; info[7] = (uint64_t)(n_slices / n_dest) * n_win;
    return 0;
}

int tpc_combine_export(tpc_ctx *c, uint32_t n_dest, uint16_t *payload_dev, uint64_t cap_units, uint64_t *dir_dev, uint64_t *units_host)
{
    if (!c || !payload_dev || !dir_dev || !units_host || n_dest == 0 || n_dest > 64 || (n_dest & (n_dest - 1))) return fail(c, -1, "bad arguments");
    if (!(c->pending_apply && !c->pending_shard && !c->pending_lists && c->pending_pl.b3 == 0 && c->pending_pl.fmt2 == 0 && n_dest <= (1u << c->pending_pl.b1)))
        return fail(c, -1, "tpc_combine_export needs the insert of this round still in its level-2 regions (tpc_combine_info says when)");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->cmb_cur) HIPCHK(c, hipMalloc((void **)&c->cmb_cur, 65 * sizeof(unsigned long long)));
    HIPCHK(c, hipMemsetAsync(c->cmb_cur, 0, 65 * sizeof(unsigned long long), c->stream));
    const TpcPartPlan g = c->pending_pl;
    const TpcCombineOut out{payload_dev, cap_units, c->cmb_cur, dir_dev, n_dest};
    {
        Timed t(c, TPC_K_COMBINE);
        if (tpc_launch_slice_combine(make_launch(c), g.slice_bits, g.b1, g.b2, g.perm_mult, g.perm_inv, &g, c->pending_novf ? c->ikeep_ovf + c->ikeep_ovf_cap : nullptr,
                                     c->pending_novf ? c->iovf_off : nullptr, TpcListSrc(), false, true, &out, 0, 1)) return fail(c, -1, "combine launch failed");
    }
    unsigned long long cur[65];
    HIPCHK(c, hipMemcpyAsync(cur, c->cmb_cur, sizeof cur, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (cur[n_dest]) return fail(c, -1, "tpc_combine_export: a destination block of %llu units is too small (size it with tpc_combine_info)", (unsigned long long)cap_units);
    for (uint32_t d = 0; d < n_dest; d++) units_host[d] = cur[d];
    // the lists now hold what the regions held: the insert is no longer pending here -- it comes back, merged with the other ranks',
    // through tpc_combine_import.  The geometry stays for tpc_combine_merge / tpc_combine_import.
    c->cmb_geo = g; c->cmb_geo.wpb = 0; c->cmb_geo.buf2 = nullptr; c->cmb_geo.cnt2 = nullptr; c->cmb_have_geo = true;
    c->pending_apply = false; c->pending_novf = 0;
    c->filter_zero_pending = c->pending_fresh;  // (what the filter held before this insert still counts when it was not reset)
    return 0;
}

int tpc_combine_merge(tpc_ctx *c, uint32_t n_src, const uint16_t *payload_dev, const uint64_t *src_base_host, const uint64_t *dir_dev, uint16_t *out_payload_dev,
                      uint64_t out_cap_units, uint64_t *out_dir_dev, uint64_t *units_host)
{
    if (!c || !out_payload_dev || !out_dir_dev || !units_host) return fail(c, -1, "bad arguments");
    if (!c->cmb_have_geo) return fail(c, -1, "tpc_combine_export first");
    if (!replicated(c) || n_src != c->sh_world) return fail(c, -1, "tpc_combine_merge: one source block per rank of a replicated sharded context");
    HIPCHK(c, hipSetDevice(c->device));
    const TpcPartPlan &g = c->cmb_geo;
    const uint64_t stride = (uint64_t)(cmb_slices(g) / c->sh_world) * tpc_list_windows(g.slice_bits);
    TpcListSrc ls;
    { int rc = cmb_sources(c, n_src, payload_dev, src_base_host, dir_dev, stride, 0, ls); if (rc) return rc; }
    HIPCHK(c, hipMemsetAsync(c->cmb_cur, 0, 65 * sizeof(unsigned long long), c->stream));
    const TpcCombineOut out{out_payload_dev, out_cap_units, c->cmb_cur, out_dir_dev, 1};
    {
        Timed t(c, TPC_K_COMBINE);
        if (tpc_launch_slice_combine(make_launch(c), g.slice_bits, g.b1, g.b2, g.perm_mult, g.perm_inv, nullptr, nullptr, nullptr, ls, false, true, &out, c->sh_rank, c->sh_world))
            return fail(c, -1, "combine launch failed");
    }
    unsigned long long cur[2];
    HIPCHK(c, hipMemcpyAsync(cur, c->cmb_cur, sizeof cur, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (cur[1]) return fail(c, -1, "tpc_combine_merge: the output block of %llu units is too small (the sum of the received blocks always suffices)", (unsigned long long)out_cap_units);
    *units_host = cur[0];
    return 0;
}

int tpc_combine_import(tpc_ctx *c, uint32_t n_src, uint32_t n_owner, const uint16_t *payload_dev, const uint64_t *src_base_host, const uint64_t *dir_dev, uint64_t dir_stride)
{
    if (!c) return -1;
    if (!c->cmb_have_geo) return fail(c, -1, "tpc_combine_export first");
    if (n_owner > (1u << c->cmb_geo.b1)) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc0 = flush_pending_apply(c); if (rc0) return rc0; }
    { int rc = cmb_sources(c, n_src, payload_dev, src_base_host, dir_dev, dir_stride, n_owner, c->cmb_ls); if (rc) return rc; }
    // from here on the round's insert is pending again: the next tpc_pass1_query's first lookup builds every slice from these lists
    // (or whatever reads the filter first: flush_pending_apply)
    c->pending_apply = true; c->pending_shard = false; c->pending_lists = true; c->pending_fresh = c->filter_zero_pending; c->pending_pl = c->cmb_geo; c->pending_novf = 0;
    c->filter_zero_pending = false;
    return 0;
}

int tpc_filter_copy_out(tpc_ctx *c, uint64_t word0, uint64_t n_words, uint32_t *dst_dev)
{
    if (!c || !c->filter || (n_words && !dst_dev) || word0 + n_words > c->filter_words) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc0 = materialize_reset(c); if (rc0) return rc0; }
    if (n_words) HIPCHK(c, hipMemcpyAsync(dst_dev, c->filter + word0, n_words * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_filter_copy_in(tpc_ctx *c, uint64_t word0, uint64_t n_words, const uint32_t *src_dev)
{
    if (!c || !c->filter || (n_words && !src_dev) || word0 + n_words > c->filter_words) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc0 = materialize_reset(c); if (rc0) return rc0; }
    if (n_words) HIPCHK(c, hipMemcpyAsync(c->filter + word0, src_dev, n_words * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_combine_choose(uint32_t world, int L, uint64_t mean_export_units, double *bytes /* [3] */)
{   // bytes a rank RECEIVES per round under each form of the exchange (the directories are small beside the payload and left out):
    //   [0] all-gather of the ranks' exports                      (W - 1) D
    //   [1] reduce-scatter by owner, all-gather of the merged     (W - 1) / W (D + U),  U = all merged lists ~ D W^0.3 (measured on the
    //       lists                                                 62-genome text: 1.33 / 1.62 / 1.84 D at 2 / 4 / 8 ranks; U <= W D always)
    //   [2] the dense filters: OR all-reduce by word ranges       2 (W - 1) / W 2^L / 8
    // D = a rank's export in bytes.  Returns the cheapest: 1, 2 or 3.
    if (world < 2 || !bytes) return 1;
    const double W = (double)world, D = 16.0 * (double)mean_export_units, U = D * std::min(W, std::pow(W, 0.3));
    bytes[0] = (W - 1.0) * D;
    bytes[1] = (W - 1.0) / W * (D + U);
    bytes[2] = 2.0 * (W - 1.0) / W * std::ldexp(1.0, L - 3);
    int best = 0;
    for (int i = 1; i < 3; i++) if (bytes[i] < bytes[best]) best = i;
    return best + 1;
}

int tpc_shard_verify_local(tpc_ctx *c)
{   // One rank: every survivor of the last tpc_shard_apply was hashed here and every probe address of functions 1..q-1 is owned
    // here, so the single-GPU verification kernel runs on the survivor sub-lists as they are -- no gather, no routing, no answers.
    if (!c || !c->sh_have[TPC_SHARD_QUERY]) return fail(c, -1, "tpc_shard_plan / tpc_shard_apply for the query first");
    if (c->sh_world != 1) return fail(c, -1, "tpc_shard_verify_local needs a filter of one shard (world == 1)");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc0 = materialize_reset(c); if (rc0) return rc0; }
    if (c->sh_nsurv) {
        Timed t(c, TPC_K_SHARD_APPLY);
        if (tpc_launch_query_verify(make_launch(c), c->sh_qpl, c->rmask)) return fail(c, -1, "verify launch failed");
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_shard_survivors(tpc_ctx *c, uint64_t *sid_dev)
{
    if (!c || !c->sh_have[TPC_SHARD_QUERY] || (!sid_dev && c->sh_nsurv)) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->sh_nsurv) tpc_launch_surv_gather(make_launch(c), c->sh_qpl, sid_dev);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_shard_verify_addrs(tpc_ctx *c, int fn, int fn_count, const uint64_t *sid_dev, uint64_t n, uint64_t *addr_dev, int32_t *owner_dev)
{
    if (!c || !c->sh_have[TPC_SHARD_QUERY] || (n && (!sid_dev || !addr_dev || !owner_dev))) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (tpc_launch_verify_addrs(make_launch(c), c->sh_qpl, fn, fn_count, sid_dev, n, addr_dev, owner_dev))
        return fail(c, -1, "bad hash function range %d+%d", fn, fn_count);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_shard_probe(tpc_ctx *c, const uint64_t *addr_dev, uint64_t n, uint8_t *hit_dev)
{
    if (!c || !c->filter || (n && (!addr_dev || !hit_dev))) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc0 = materialize_reset(c); if (rc0) return rc0; }
    tpc_launch_shard_probe(make_launch(c), addr_dev, n, hit_dev);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_shard_mark(tpc_ctx *c, const uint64_t *sid_dev, uint64_t n)
{
    if (!c || !c->sh_have[TPC_SHARD_QUERY] || (n && !sid_dev)) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    tpc_launch_shard_mark(make_launch(c), c->sh_qpl, sid_dev, n, c->rmask);
    c->marks_valid = false; c->rmask_sums_valid = false;
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_mask_export(tpc_ctx *c, uint32_t *dst_dev)
{
    if (!c || !c->rmask || !dst_dev) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst_dev, c->rmask, c->n_words * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_mask_merge(tpc_ctx *c, const uint32_t *src_dev, uint32_t count)
{
    if (!c || !c->rmask || (!src_dev && count)) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    for (uint32_t i = 0; i < count; i++) tpc_launch_mask_or(c->stream, c->rmask, src_dev + (uint64_t)i * c->n_words, c->n_words);
    c->marks_valid = false; c->rmask_sums_valid = false;
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_shard_survivor_sources(tpc_ctx *c, const uint64_t *sid_dev, uint64_t n, int32_t *source_dev)
{
    if (!c || !c->sh_have[TPC_SHARD_QUERY] || (n && (!sid_dev || !source_dev))) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    tpc_launch_survivor_sources(c->stream, sid_dev, n, c->sh_world, source_dev);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

namespace {

// owner routing of n tagged items (owner = (v >> shift) & (world - 1)) from src to dst in owner-major order; counted: the per-owner
// counts already sit in route_scratch[0..63] (a producer kernel accumulated them), else a counting pass runs first
int route64(tpc_ctx *c, const uint64_t *src, uint64_t n, int shift, uint64_t keep, bool counted, uint32_t *perm_dev, uint64_t *dst, uint64_t *counts_host)
{
    unsigned long long *d = c->route_scratch;  // [0..63] counts, [64..127] cursors
    if (!counted) {
        HIPCHK(c, hipMemsetAsync(d, 0, 128 * sizeof(unsigned long long), c->stream));
        tpc_launch_route64(c->stream, src, n, shift, c->sh_world - 1, keep, d, d + 64, perm_dev, dst, 0);
    }
    unsigned long long h[64], cur[64];
    HIPCHK(c, hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long acc = 0;
    for (int i = 0; i < 64; i++) { cur[i] = acc; acc += h[i]; if ((uint32_t)i < c->sh_world) counts_host[i] = h[i]; }
    if (acc != n) return fail(c, -1, "owner routing: counted %llu of %llu items", acc, (unsigned long long)n);
    HIPCHK(c, hipMemcpyAsync(d + 64, cur, sizeof cur, hipMemcpyHostToDevice, c->stream));
    tpc_launch_route64(c->stream, src, n, shift, c->sh_world - 1, keep, d, d + 64, perm_dev, dst, 1);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

}  // namespace

int tpc_shard_survivors_home(tpc_ctx *c, uint64_t *tmp_dev, uint64_t *send_dev, uint64_t *counts_host)
{
    if (!c || !c->sh_have[TPC_SHARD_QUERY] || !counts_host || (c->sh_nsurv && (!send_dev || (c->sh_world > 1 && !tmp_dev)))) return fail(c, -1, "bad arguments");
    if (c->sh_world > 64) return fail(c, -1, "routing supports at most 64 ranks");
    if (c->sh_nsurv > 0xFFFFFFFFull) return fail(c, -1, "too many items to route at once");
    HIPCHK(c, hipSetDevice(c->device));
    for (uint32_t r = 0; r < c->sh_world; r++) counts_host[r] = 0;
    if (!c->sh_nsurv) return 0;
    if (c->sh_world == 1) {  // everything was hashed here
        tpc_launch_surv_gather(make_launch(c), c->sh_qpl, send_dev);
        counts_host[0] = c->sh_nsurv;
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    tpc_launch_surv_gather(make_launch(c), c->sh_qpl, tmp_dev);
    uint32_t lw = 0;
    while ((1u << lw) < c->sh_world) ++lw;
    // the rank that hashed the survivor's position: the top log2(world) bits of its 30-bit position field (k_q_hash<SHARDED>)
    return route64(c, tmp_dev, c->sh_nsurv, 3 + 30 - (int)lw, ~0ull, false, nullptr, send_dev, counts_host);
}

int tpc_shard_verify_send(tpc_ctx *c, int fn, int fn_count, const uint64_t *sid_dev, uint64_t n, uint64_t *tmp_dev, uint64_t *send_dev, uint32_t *perm_dev,
                          uint64_t *counts_host)
{
    if (!c || !c->sh_have[TPC_SHARD_QUERY] || !counts_host || fn_count < 1) return fail(c, -1, "bad arguments");
    const bool one = c->sh_world == 1;
    if (n && (!sid_dev || !send_dev || (!one && (!tmp_dev || !perm_dev)))) return fail(c, -1, "bad arguments");
    if (c->sh_world > 64) return fail(c, -1, "routing supports at most 64 ranks");
    const uint64_t total = n * (uint64_t)fn_count;
    if (total > 0xFFFFFFFFull) return fail(c, -1, "too many items to route at once");
    HIPCHK(c, hipSetDevice(c->device));
    for (uint32_t r = 0; r < c->sh_world; r++) counts_host[r] = 0;
    if (!n) return 0;
    unsigned long long *d = c->route_scratch;
    if (!one) HIPCHK(c, hipMemsetAsync(d, 0, 128 * sizeof(unsigned long long), c->stream));
    // one rank: every probe is this rank's own, the natural order is the send order (tags are zero)
    if (tpc_launch_verify_addrs(make_launch(c), c->sh_qpl, fn, fn_count, sid_dev, n, one ? send_dev : tmp_dev, nullptr, one ? nullptr : d))
        return fail(c, -1, "bad hash function range %d+%d", fn, fn_count);
    if (one) {
        counts_host[0] = total;
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    return route64(c, tmp_dev, total, TPC_V_OWNER_SHIFT, (1ull << TPC_V_OWNER_SHIFT) - 1ull, true, perm_dev, send_dev, counts_host);
}

int tpc_shard_finish(tpc_ctx *c, const uint64_t *sid_dev, uint64_t n, int fn_count, const uint8_t *hit_dev, const uint32_t *perm_dev, uint64_t *n_marked)
{
    if (!c || !c->sh_have[TPC_SHARD_QUERY] || fn_count < 1 || (n && (!sid_dev || !hit_dev))) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->counters + 3, 0, sizeof(unsigned long long), c->stream));
    tpc_launch_finish(c->stream, c->sh_qpl, sid_dev, n, fn_count, hit_dev, perm_dev, c->rmask, c->counters + 3);
    c->marks_valid = false; c->rmask_sums_valid = false;
    HIPCHK(c, hipGetLastError());
    uint64_t m = 0;
    const int rc = read_counter(c, 3, &m);
    if (n_marked) *n_marked = m;
    return rc;
}

int tpc_shard_route(tpc_ctx *c, const int32_t *owner_dev, uint64_t n, uint32_t *perm_dev, uint64_t *counts_host)
{
    if (!c || !counts_host || (n && (!owner_dev || !perm_dev))) return fail(c, -1, "bad arguments");
    if (c->sh_world > 64) return fail(c, -1, "routing supports at most 64 ranks");
    if (n > 0xFFFFFFFFull) return fail(c, -1, "too many items to route at once");
    HIPCHK(c, hipSetDevice(c->device));
    unsigned long long *d = c->route_scratch;  // [0..63] counts, [64..127] cursors
    HIPCHK(c, hipMemsetAsync(d, 0, 128 * sizeof(unsigned long long), c->stream));
    tpc_launch_route(c->stream, owner_dev, n, d, d + 64, perm_dev, 0);
    unsigned long long h[64], cur[64];
    HIPCHK(c, hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long acc = 0;
    for (int i = 0; i < 64; i++) { cur[i] = acc; acc += h[i]; if ((uint32_t)i < c->sh_world) counts_host[i] = h[i]; }
    HIPCHK(c, hipMemcpyAsync(d + 64, cur, sizeof cur, hipMemcpyHostToDevice, c->stream));
    tpc_launch_route(c->stream, owner_dev, n, d, d + 64, perm_dev, 1);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_shard_permute64(tpc_ctx *c, const uint64_t *src_dev, const uint32_t *perm_dev, uint64_t n, uint64_t *dst_dev)
{
    if (!c || (n && (!src_dev || !perm_dev || !dst_dev))) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    tpc_launch_permute64(c->stream, src_dev, perm_dev, n, dst_dev);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_shard_select(tpc_ctx *c, const uint64_t *sid_dev, uint64_t n, int fn_count, const uint8_t *hit_dev, const uint32_t *perm_dev, uint64_t *sid_out_dev,
                     uint64_t *n_out)
{
    if (!c || !n_out || fn_count < 1 || (n && (!sid_dev || !hit_dev || !sid_out_dev))) return fail(c, -1, "bad arguments");  // perm_dev may be null: natural order
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->counters + 3, 0, sizeof(unsigned long long), c->stream));
    tpc_launch_select(c->stream, sid_dev, n, fn_count, hit_dev, perm_dev, sid_out_dev, c->counters + 3);
    HIPCHK(c, hipGetLastError());
    return read_counter(c, 3, n_out);
}

// Candidate-mask union by word ranges (an OR all-reduce built from an all_to_all and an all_gather, because RCCL has no
// bitwise reduction): every rank exports its mask padded to world x chunk words, the chunks are exchanged (rank r
// receives chunk r of everyone), tpc_mask_or_blocks folds them, the folded chunks are all-gathered and imported.
int tpc_mask_export_padded(tpc_ctx *c, uint32_t *dst_dev, uint64_t total_words)
{
    if (!c || !c->rmask || !dst_dev || total_words < c->n_words) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst_dev, c->rmask, c->n_words * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    if (total_words > c->n_words) HIPCHK(c, hipMemsetAsync(dst_dev + c->n_words, 0, (total_words - c->n_words) * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_mask_or_blocks(tpc_ctx *c, const uint32_t *blocks_dev, uint32_t count, uint64_t words, uint32_t *out_dev)
{
    if (!c || !blocks_dev || !out_dev || count < 1) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out_dev, blocks_dev, words * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    for (uint32_t i = 1; i < count; i++) tpc_launch_mask_or(c->stream, out_dev, blocks_dev + (uint64_t)i * words, words);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int tpc_mask_import(tpc_ctx *c, const uint32_t *src_dev)
{
    if (!c || !c->rmask || !src_dev) return fail(c, -1, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->rmask, src_dev, c->n_words * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    c->marks_valid = false; c->rmask_sums_valid = false;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

uint64_t tpc_filter_words(const tpc_ctx *c) { return c ? c->filter_words : 0; }

int tpc_filter_download(tpc_ctx *c, uint32_t *words_host)
{
    if (!c || !c->filter) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc0 = materialize_reset(c); if (rc0) return rc0; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(words_host, c->filter, c->filter_words * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}

int tpc_filter_upload(tpc_ctx *c, const uint32_t *words_host)
{
    if (!c || !c->filter || !words_host) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(c->filter, words_host, c->filter_words * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->filter_zero_pending = false;  // the uploaded bits are the filter now
    c->pending_apply = false;
    return 0;
}

uint64_t tpc_mask_words(const tpc_ctx *c) { return c ? c->n_words : 0; }

int tpc_mask_download(tpc_ctx *c, int run_wide, uint32_t *words_host)
{
    if (!c || !c->rmask) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(words_host, run_wide ? c->mask : c->rmask, c->n_words * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}

int tpc_hash_dump(tpc_ctx *c, uint64_t g0, uint64_t n, uint64_t *out_host)
{
    if (!c || !c->have_params || !c->bases) return -1;
    if (g0 + n + c->P.k > c->n_text) return fail(c, -1, "range past the text");
    if (c->text_windowed) return fail(c, -1, "this context holds only its window of the text");
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t *d = nullptr;
    const size_t bytes = n * 2 * c->P.q * sizeof(uint64_t);
    HIPCHK(c, hipMalloc((void **)&d, bytes));
    tpc_launch_hash_dump(make_launch(c), g0, n, d);
    HIPCHK(c, hipMemcpy(out_host, d, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return 0;
}

double tpc_kernel_ms(const tpc_ctx *c, int which)
{
    if (!c || which < 0 || which >= TPC_K_COUNT || !c->ev_used[which]) return -1.0;
    if (hipEventSynchronize(c->ev1[which]) != hipSuccess) return -1.0;
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev0[which], c->ev1[which]) != hipSuccess) return -1.0;
    return ms;
}

}  // extern "C"
