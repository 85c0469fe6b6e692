// tpc_capi.hip -- the C-ABI of include/twopaco_hip.h: context, device memory, first-pass orchestration (the other entry points:
// tpc_capi_pass2.hip, tpc_capi_shard.hip, tpc_capi_combine.hip; the shared context and helpers: tpc_ctx.h).
#include "tpc_ctx.h"

namespace tpch {

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


int read_counter(tpc_ctx *c, int i, uint64_t *out)
{
    unsigned long long v = 0;
    HIPCHK(c, hipMemcpyAsync(&v, c->counters + i, sizeof v, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *out = v;
    return 0;
}


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

}  // namespace tpch

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

