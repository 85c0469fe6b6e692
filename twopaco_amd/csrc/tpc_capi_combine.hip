// tpc_capi_combine.hip -- C-ABI, the Bloom filter replicated on every rank through set-bit lists (include/twopaco_hip.h: tpc_combine_*; kernels: tpc_combine.hip).
#include "tpc_ctx.h"

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

