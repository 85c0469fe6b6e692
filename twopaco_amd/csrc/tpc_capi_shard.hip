// tpc_capi_shard.hip -- C-ABI, the Bloom filter cut by bit address over ranks (include/twopaco_hip.h: tpc_shard_*, tpc_mask_*): the halves of each pass; no communication here.
#include "tpc_ctx.h"

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

