// tpc_capi_pass2.hip -- C-ABI, second pass: exact filter over the marks, junction keys, ids, junction stream (include/twopaco_hip.h: tpc_pass2_*, tpc_junction*, tpc_emit*).
#include "tpc_ctx.h"

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

namespace tpch {
void stream_part_release(tpc_ctx *c)
{
    for (void *p : { (void *)c->sp_rec, (void *)c->sp_vscan, (void *)c->sp_cnt, (void *)c->sp_lo, (void *)c->sp_flags }) if (p) (void)hipFree(p);
    c->sp_rec = c->sp_vscan = c->sp_cnt = c->sp_lo = nullptr;
    c->sp_flags = nullptr;
    c->sp_n_rec = 0;
}
}  // namespace tpch

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

