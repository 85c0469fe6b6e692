"""ctypes bindings over libtwopaco_hip.so (device C-ABI, include/twopaco_hip.h) and
libtwopaco_host.so (C wrappers over the C++ host layer).  Plumbing only: every compute call
goes to the HIP library; if it is missing this module raises -- there is no fallback."""
import ctypes
import os

import numpy as np

from .build import lib_dir

INVALID_VERTEX = (1 << 63) - 1
KERNELS = {"fused": 12, "filter_reset": 0, "insert": 1, "query": 2, "compact": 3, "filter2": 4, "scan2": 5, "sort": 6, "emit": 7, "split": 8,
           "shard_hash": 9, "shard_apply": 10, "stream": 11, "lookup": 13, "combine": 14}

# every symbol include/twopaco_hip.h declares
HIP_SYMBOLS = ["tpc_ctx_create", "tpc_ctx_destroy", "tpc_last_error", "tpc_set_params", "tpc_seq_upload",
               "tpc_run_begin", "tpc_filter_reset", "tpc_pass1_insert", "tpc_pass1_split_hist", "tpc_pass1_query", "tpc_pass2_filter",
               "tpc_junctions_finalize", "tpc_key_words", "tpc_junction_keys", "tpc_junction_keys_raw", "tpc_junction_keys_set", "tpc_get_id", "tpc_emit",
               "tpc_emit_fetch", "tpc_filter_words", "tpc_filter_download", "tpc_mask_words", "tpc_mask_download",
               "tpc_hash_dump", "tpc_kernel_ms", "tpc_set_option",
               "tpc_shard_config", "tpc_shard_plan", "tpc_shard_hash", "tpc_shard_overflow_get", "tpc_shard_overflow_set", "tpc_shard_apply",
               "tpc_shard_pack", "tpc_shard_apply_packed", "tpc_pass2_marks", "tpc_pass2_mark_owners", "tpc_pass2_filter_positions",
               "tpc_pass2_mark_records", "tpc_pass2_filter_records", "tpc_pass2_aggregate_records", "tpc_pass2_filter_aggregated", "tpc_shard_permute_rows", "tpc_emit_export", "tpc_emit_import",
               "tpc_shard_survivors", "tpc_shard_survivor_sources", "tpc_shard_verify_addrs", "tpc_shard_probe", "tpc_shard_mark", "tpc_mask_export", "tpc_mask_merge",
               "tpc_shard_route", "tpc_shard_permute64", "tpc_shard_select", "tpc_mask_export_padded", "tpc_mask_or_blocks", "tpc_mask_import",
               "tpc_emit_stream", "tpc_emit_stream_fetch", "tpc_host_alloc", "tpc_host_free", "tpc_get_stat", "tpc_filter_upload",
               "tpc_junction_keys_export", "tpc_junction_keys_import", "tpc_warmup", "tpc_preload", "tpc_reserve", "tpc_shard_chunk", "tpc_emit_stream_partial", "tpc_emit_stream_part",
               "tpc_shard_plan_both", "tpc_shard_hash_begin", "tpc_shard_hash_end", "tpc_shard_apply_inplace", "tpc_shard_survivors_home", "tpc_shard_verify_send", "tpc_shard_finish", "tpc_shard_verify_local", "tpc_shard_periodic_copy",
               "tpc_pass1_query_begin", "tpc_combine_info", "tpc_combine_export", "tpc_combine_merge", "tpc_combine_import", "tpc_combine_choose", "tpc_filter_copy_out", "tpc_filter_copy_in"]

_hip = None
_host = None


def combine_choose(world, L, mean_export_units):
    """(mode, bytes received per rank for [all-gather of exports, reduce-scatter + all-gather, dense OR all-reduce]); mode 1..3 = the cheapest."""
    b = (ctypes.c_double * 3)()
    mode = hip().tpc_combine_choose(world, L, int(mean_export_units), b)
    return mode, [float(x) for x in b]


def _load(name):
    path = os.path.join(lib_dir(), name)
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback)" % path)
    return ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def hip():
    global _hip
    if _hip is None:
        L = _load("libtwopaco_hip.so")
        u64, i64, p, ci = ctypes.c_uint64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
        L.tpc_ctx_create.argtypes = [ci, ctypes.POINTER(p)]
        L.tpc_ctx_destroy.argtypes = [p]
        L.tpc_last_error.restype = ctypes.c_char_p
        L.tpc_last_error.argtypes = [p]
        L.tpc_set_params.argtypes = [p, ci, ci, ci, p]
        L.tpc_seq_upload.argtypes = [p, p, p, u64]
        L.tpc_filter_reset.argtypes = [p]
        L.tpc_run_begin.argtypes = [p]
        L.tpc_pass1_insert.argtypes = [p, u64, u64, p]
        L.tpc_pass1_split_hist.argtypes = [p, p, p, ctypes.c_uint32, p]
        L.tpc_pass1_query.argtypes = [p, u64, u64, p]
        L.tpc_pass2_filter.argtypes = [p, u64, p, p, p]
        L.tpc_junctions_finalize.argtypes = [p, p]
        L.tpc_key_words.argtypes = [p]
        L.tpc_junction_keys.argtypes = [p, p]
        L.tpc_junction_keys_raw.argtypes = [p, p, p]
        L.tpc_junction_keys_set.argtypes = [p, p, u64]
        L.tpc_junction_keys_export.argtypes = [p, p, u64, p]
        L.tpc_junction_keys_import.argtypes = [p, p, u64, ci]
        L.tpc_get_id.restype = i64
        L.tpc_get_id.argtypes = [p, ctypes.c_char_p]
        L.tpc_emit.argtypes = [p, p, p]
        L.tpc_emit_fetch.argtypes = [p, p, p]
        L.tpc_filter_words.restype = u64
        L.tpc_filter_words.argtypes = [p]
        L.tpc_filter_download.argtypes = [p, p]
        L.tpc_filter_upload.argtypes = [p, p]
        L.tpc_warmup.argtypes = [p]
        L.tpc_preload.argtypes = [ctypes.c_int]
        L.tpc_reserve.argtypes = [p, u64]
        L.tpc_mask_words.restype = u64
        L.tpc_mask_words.argtypes = [p]
        L.tpc_mask_download.argtypes = [p, ci, p]
        L.tpc_hash_dump.argtypes = [p, u64, u64, p]
        L.tpc_kernel_ms.restype = ctypes.c_double
        L.tpc_kernel_ms.argtypes = [p, ci]
        L.tpc_set_option.argtypes = [p, ctypes.c_char_p, i64]
        u32 = ctypes.c_uint32
        L.tpc_shard_config.argtypes = [p, u32, u32]
        L.tpc_shard_plan.argtypes = [p, ci, u64, u64, p]
        L.tpc_shard_hash.argtypes = [p, ci, u64, u64, u64, p, p, p]
        L.tpc_shard_plan_both.argtypes = [p, u64, u64, p, p]
        L.tpc_shard_hash_begin.argtypes = [p, ci, u64, u64, u64, p, p]
        L.tpc_shard_hash_end.argtypes = [p, ci, p]
        L.tpc_shard_overflow_get.argtypes = [p, ci, p, u64]
        L.tpc_shard_overflow_set.argtypes = [p, ci, p, u64]
        L.tpc_shard_apply.argtypes = [p, ci, u64, p, p, p]
        L.tpc_pass2_marks.argtypes = [p, p]
        L.tpc_pass2_mark_owners.argtypes = [p, u32, p, p]
        L.tpc_pass2_filter_positions.argtypes = [p, p, u64, u64, p, p, p]
        L.tpc_pass2_mark_records.argtypes = [p, u32, p, p]
        L.tpc_pass2_filter_records.argtypes = [p, p, u64, u64, p, p, p]
        L.tpc_pass2_aggregate_records.argtypes = [p, u32, u64, p, p, p]
        L.tpc_pass2_filter_aggregated.argtypes = [p, p, u64, u64, p, p, p]
        L.tpc_shard_permute_rows.argtypes = [p, p, p, u64, ci, p]
        L.tpc_shard_pack.argtypes = [p, ci, p, p, p, p]
        L.tpc_shard_apply_packed.argtypes = [p, ci, u64, p, p, p]
        L.tpc_shard_apply_inplace.argtypes = [p, ci, u64, p, p, p, p, p]
        L.tpc_shard_survivors_home.argtypes = [p, p, p, p]
        L.tpc_shard_verify_send.argtypes = [p, ci, ci, p, u64, p, p, p, p]
        L.tpc_shard_finish.argtypes = [p, p, u64, ci, p, p, p]
        L.tpc_shard_verify_local.argtypes = [p]
        L.tpc_shard_periodic_copy.argtypes = [p]
        L.tpc_pass1_query_begin.argtypes = [p, u64, u64]
        L.tpc_combine_info.argtypes = [p, u32, p]
        L.tpc_combine_export.argtypes = [p, u32, p, u64, p, p]
        L.tpc_combine_merge.argtypes = [p, u32, p, p, p, p, u64, p, p]
        L.tpc_combine_import.argtypes = [p, u32, u32, p, p, p, u64]
        L.tpc_combine_choose.argtypes = [u32, ci, u64, p]
        L.tpc_filter_copy_out.argtypes = [p, u64, u64, p]
        L.tpc_filter_copy_in.argtypes = [p, u64, u64, p]
        L.tpc_shard_survivors.argtypes = [p, p]
        L.tpc_shard_verify_addrs.argtypes = [p, ci, ci, p, u64, p, p]
        L.tpc_shard_survivor_sources.argtypes = [p, p, u64, p]
        L.tpc_shard_probe.argtypes = [p, p, u64, p]
        L.tpc_shard_mark.argtypes = [p, p, u64]
        L.tpc_mask_export.argtypes = [p, p]
        L.tpc_mask_merge.argtypes = [p, p, u32]
        L.tpc_shard_route.argtypes = [p, p, u64, p, p]
        L.tpc_shard_permute64.argtypes = [p, p, p, u64, p]
        L.tpc_shard_select.argtypes = [p, p, u64, ci, p, p, p, p]
        L.tpc_mask_export_padded.argtypes = [p, p, u64]
        L.tpc_mask_or_blocks.argtypes = [p, p, u32, u64, p]
        L.tpc_mask_import.argtypes = [p, p]
        L.tpc_emit_stream.argtypes = [p, p, p, u32, p, p]
        L.tpc_emit_stream_fetch.argtypes = [p, u64, u64, p]
        L.tpc_host_alloc.argtypes = [ctypes.POINTER(p), u64]
        L.tpc_host_free.argtypes = [p]
        L.tpc_get_stat.restype = i64
        L.tpc_get_stat.argtypes = [p, ctypes.c_char_p]
        _hip = L
    return _hip


def host():
    global _host
    if _host is None:
        hip()  # dependency of the host library
        L = _load("libtwopaco_host.so")
        u64, i64, p, ci = ctypes.c_uint64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
        L.tpch_last_error.restype = ctypes.c_char_p
        L.tpch_free.argtypes = [p]
        L.tpch_seed_table.argtypes = [u64, ci, ci, ci, p]
        L.tpch_text_new.restype = p
        L.tpch_text_free.argtypes = [p]
        L.tpch_text_add_fasta.argtypes = [p, ctypes.POINTER(ctypes.c_char_p), ci, ci]
        L.tpch_text_add_codes.argtypes = [p, p, u64]
        for name, res in [("tpch_text_length", u64), ("tpch_text_words", u64), ("tpch_text_bases", p), ("tpch_text_nmask", p),
                          ("tpch_text_records", ctypes.c_uint32), ("tpch_text_rec_start", p), ("tpch_text_rec_length", p)]:
            getattr(L, name).restype = res
            getattr(L, name).argtypes = [p]
        L.tpch_create_enumerator.restype = p
        L.tpch_create_enumerator.argtypes = [ctypes.POINTER(ctypes.c_char_p), ci, u64, u64, u64, u64, u64, u64, ctypes.c_char_p,
                                             ctypes.c_char_p, ci, u64, ci, ci, ctypes.POINTER(p)]
        L.tpch_create_enumerator_mgpu.restype = p
        L.tpch_create_enumerator_mgpu.argtypes = [ctypes.POINTER(ctypes.c_char_p), ci, u64, u64, u64, u64, u64, u64, ctypes.c_char_p,
                                                  ctypes.c_char_p, ci, u64, ci, ci, ci, ci, ci, ctypes.POINTER(p)]
        L.tpch_enumerator_free.argtypes = [p]
        L.tpch_vertices_count.restype = u64
        L.tpch_vertices_count.argtypes = [p]
        L.tpch_get_id.restype = i64
        L.tpch_get_id.argtypes = [p, ctypes.c_char_p]
        L.tpch_hash_seed.argtypes = [p, p]
        L.tpch_synth_genome.argtypes = [u64, u64, p]
        L.tpch_synth_substitute.argtypes = [p, u64, u64, u64, p]
        L.tpch_synth_n_runs.argtypes = [p, u64, u64, u64, u64]
        _host = L
    return _host


def _view(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


def seed_table(q, bits, seed=None):
    """q x 5 (A,C,G,T,N) character tables; seed=None draws from /dev/urandom like the reference."""
    t = np.zeros((q, 5), dtype=np.uint64)
    if host().tpch_seed_table(0 if seed is None else seed, 0 if seed is None else 1, q, bits, t.ctypes.data) != 0:
        raise RuntimeError(host().tpch_last_error().decode())
    return t


class PackedText:
    """The packed global text T = N rec0 N rec1 N ... (host/textpack.h)."""

    def __init__(self):
        self._h = host().tpch_text_new()

    def close(self):
        if self._h and host is not None:
            host().tpch_text_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    @classmethod
    def from_fasta(cls, files, threads=1):
        t = cls()
        arr = (ctypes.c_char_p * len(files))(*[f.encode() for f in files])
        if host().tpch_text_add_fasta(t._h, arr, len(files), threads) != 0:
            raise RuntimeError(host().tpch_last_error().decode())
        return t

    @classmethod
    def from_codes(cls, records):
        """records: iterable of uint8 arrays with codes 0..3 (ACGT) and 4 (N)."""
        t = cls()
        for r in records:
            r = np.ascontiguousarray(r, dtype=np.uint8)
            host().tpch_text_add_codes(t._h, r.ctypes.data, r.size)
        return t

    @property
    def length(self):
        return host().tpch_text_length(self._h)

    @property
    def bases(self):
        return _view(host().tpch_text_bases(self._h), host().tpch_text_words(self._h), np.uint64)

    @property
    def nmask(self):
        return _view(host().tpch_text_nmask(self._h), host().tpch_text_words(self._h), np.uint32)

    @property
    def rec_start(self):
        return _view(host().tpch_text_rec_start(self._h), host().tpch_text_records(self._h), np.uint64).copy()

    @property
    def rec_length(self):
        return _view(host().tpch_text_rec_length(self._h), host().tpch_text_records(self._h), np.uint64).copy()


class Context:
    """One device context of the C-ABI (include/twopaco_hip.h)."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        rc = hip().tpc_ctx_create(device, ctypes.byref(self._h))
        if rc != 0:
            self._h = None
            raise RuntimeError("tpc_ctx_create failed (%d): no HIP device -- there is no CPU fallback" % rc)

    def close(self):
        if getattr(self, "_h", None) and hip is not None:
            hip().tpc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError("twopaco_hip: %s (%d)" % (hip().tpc_last_error(self._h).decode(), rc))

    def set_option(self, name, value):
        self._ck(hip().tpc_set_option(self._h, name.encode(), int(value)))

    def stat(self, name):
        return int(hip().tpc_get_stat(self._h, name.encode()))

    def set_params(self, k, L, q, table):
        table = np.ascontiguousarray(table, dtype=np.uint64)
        assert table.shape == (q, 5)
        self.k, self.L, self.q = k, L, q
        self._ck(hip().tpc_set_params(self._h, k, L, q, table.ctypes.data))

    def seq_upload(self, text):
        b, n = np.ascontiguousarray(text.bases), np.ascontiguousarray(text.nmask)
        self._ck(hip().tpc_seq_upload(self._h, b.ctypes.data, n.ctypes.data, text.length))

    def run_begin(self):
        self._ck(hip().tpc_run_begin(self._h))

    def filter_reset(self):
        self._ck(hip().tpc_filter_reset(self._h))

    def pass1_insert(self, lo=0, hi=None, count=True):
        n = ctypes.c_uint64(0)
        self._ck(hip().tpc_pass1_insert(self._h, lo, (1 << self.L) if hi is None else hi, ctypes.byref(n) if count else None))
        return n.value

    def pass1_split_hist(self, rec_start, rec_len):
        rs = np.ascontiguousarray(rec_start, dtype=np.uint64)
        rl = np.ascontiguousarray(rec_len, dtype=np.uint64)
        bins = np.zeros(1 << 24, dtype=np.uint32)
        self._ck(hip().tpc_pass1_split_hist(self._h, rs.ctypes.data, rl.ctypes.data, rs.size, bins.ctypes.data))
        return bins

    def pass1_query(self, lo=0, hi=None):
        n = ctypes.c_uint64(0)
        self._ck(hip().tpc_pass1_query(self._h, lo, (1 << self.L) if hi is None else hi, ctypes.byref(n)))
        return n.value

    def pass1_query_begin(self, lo=0, hi=None):
        self._ck(hip().tpc_pass1_query_begin(self._h, lo, (1 << self.L) if hi is None else hi))

    def pass2_filter(self, abundance=(1 << 64) - 1):
        a, b, c = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._ck(hip().tpc_pass2_filter(self._h, abundance, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"true": a.value, "false": b.value, "table": c.value}

    def pass2_marks(self):
        """Compacts this round's mask; returns the number of marked positions (kept in the context for the output pass)."""
        n = ctypes.c_uint64(0)
        self._ck(hip().tpc_pass2_marks(self._h, ctypes.byref(n)))
        return n.value

    def pass2_mark_owners(self, world, pos_ptr, owner_ptr):
        self._ck(hip().tpc_pass2_mark_owners(self._h, world, pos_ptr, owner_ptr))

    def pass2_filter_positions(self, pos_ptr, n, abundance=(1 << 64) - 1):
        a, b, c = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._ck(hip().tpc_pass2_filter_positions(self._h, pos_ptr, n, abundance, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"true": a.value, "false": b.value, "table": c.value}

    def pass2_mark_records(self, world, rec_ptr, owner_ptr):
        self._ck(hip().tpc_pass2_mark_records(self._h, world, rec_ptr, owner_ptr))

    def pass2_filter_records(self, rec_ptr, n, abundance=(1 << 64) - 1):
        a, b, c = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._ck(hip().tpc_pass2_filter_records(self._h, rec_ptr, n, abundance, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"true": a.value, "false": b.value, "table": c.value}

    def pass2_aggregate_records(self, world, rec_ptr, owner_ptr, abundance=(1 << 64) - 1):
        n = ctypes.c_uint64(0)
        self._ck(hip().tpc_pass2_aggregate_records(self._h, world, abundance, rec_ptr, owner_ptr, ctypes.byref(n)))
        return n.value

    def pass2_filter_aggregated(self, rec_ptr, n, abundance=(1 << 64) - 1):
        a, b, c = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._ck(hip().tpc_pass2_filter_aggregated(self._h, rec_ptr, n, abundance, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"true": a.value, "false": b.value, "table": c.value}

    def shard_permute_rows(self, src_ptr, perm_ptr, n, row_words, dst_ptr):
        self._ck(hip().tpc_shard_permute_rows(self._h, src_ptr, perm_ptr, n, row_words, dst_ptr))

    def junctions_finalize(self):
        n = ctypes.c_uint64(0)
        self._ck(hip().tpc_junctions_finalize(self._h, ctypes.byref(n)))
        self.n_junctions = n.value
        return n.value

    def junction_keys(self):
        C = hip().tpc_key_words(self._h)
        keys = np.zeros((self.n_junctions, C), dtype=np.uint64)
        self._ck(hip().tpc_junction_keys(self._h, keys.ctypes.data))
        return keys

    def junction_keys_raw(self):
        """Keys appended so far (unsorted, before junctions_finalize)."""
        n = ctypes.c_uint64(0)
        self._ck(hip().tpc_junction_keys_raw(self._h, None, ctypes.byref(n)))
        keys = np.zeros((n.value, hip().tpc_key_words(self._h)), dtype=np.uint64)
        self._ck(hip().tpc_junction_keys_raw(self._h, keys.ctypes.data, ctypes.byref(n)))
        return keys

    def junction_keys_set(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        self._ck(hip().tpc_junction_keys_set(self._h, keys.ctypes.data, keys.shape[0]))

    def key_words(self):
        return int(hip().tpc_key_words(self._h))

    def junction_keys_export(self, dst_ptr, cap_keys):
        n = ctypes.c_uint64(0)
        self._ck(hip().tpc_junction_keys_export(self._h, dst_ptr, cap_keys, ctypes.byref(n)))
        return n.value

    def junction_keys_import(self, src_ptr, n, append):
        self._ck(hip().tpc_junction_keys_import(self._h, src_ptr, n, 1 if append else 0))

    def get_id(self, kmer):
        return hip().tpc_get_id(self._h, kmer.encode())

    def emit(self):
        a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._ck(hip().tpc_emit(self._h, ctypes.byref(a), ctypes.byref(b)))
        self.n_marked, self.n_valid = a.value, b.value
        return a.value, b.value

    def emit_fetch(self):
        g = np.zeros(self.n_marked, dtype=np.uint64)
        ids = np.zeros(self.n_marked, dtype=np.int64)
        self._ck(hip().tpc_emit_fetch(self._h, g.ctypes.data, ids.ctypes.data))
        return g, ids

    def emit_stream(self, rec_start, rec_len):
        """The bytes of de_bruijn.bin (after emit()); returns (bytes, records without separators)."""
        rs = np.ascontiguousarray(rec_start, dtype=np.uint64)
        rl = np.ascontiguousarray(rec_len, dtype=np.uint64)
        nb, nr = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._ck(hip().tpc_emit_stream(self._h, rs.ctypes.data, rl.ctypes.data, rs.size, ctypes.byref(nb), ctypes.byref(nr)))
        buf = np.zeros(nb.value, dtype=np.uint8)
        if hip().tpc_emit_stream_fetch(self._h, 0, nb.value, buf.ctypes.data) != 0:
            raise RuntimeError("tpc_emit_stream_fetch failed")
        return buf.tobytes(), nr.value

    def filter_words(self):
        return int(hip().tpc_filter_words(self._h))

    def filter_download(self):
        w = np.zeros(hip().tpc_filter_words(self._h), dtype=np.uint32)
        self._ck(hip().tpc_filter_download(self._h, w.ctypes.data))
        return w

    def filter_upload(self, words):
        w = np.ascontiguousarray(words, dtype=np.uint32)
        assert w.size == hip().tpc_filter_words(self._h)
        self._ck(hip().tpc_filter_upload(self._h, w.ctypes.data))

    def mask_download(self, run_wide=False):
        w = np.zeros(hip().tpc_mask_words(self._h), dtype=np.uint32)
        self._ck(hip().tpc_mask_download(self._h, 1 if run_wide else 0, w.ctypes.data))
        return w

    def hash_dump(self, g0, n):
        out = np.zeros((n, self.q, 2), dtype=np.uint64)
        self._ck(hip().tpc_hash_dump(self._h, g0, n, out.ctypes.data))
        return out

    def kernel_ms(self, name):
        return hip().tpc_kernel_ms(self._h, KERNELS[name])

    # ---- address-sharded filter: device pointers in, no communication here (see twopaco_amd/dist.py)
    def shard_config(self, rank, world):
        self._ck(hip().tpc_shard_config(self._h, rank, world))

    def shard_plan(self, which, lo=0, hi=None):
        g = np.zeros(16, dtype=np.uint64)
        self._ck(hip().tpc_shard_plan(self._h, which, lo, (1 << self.L) if hi is None else hi, g.ctypes.data))
        names = ["batches", "tiles_per_rank", "region_block_bytes", "count_block_bytes", "survivor_cap", "overflow_cap", "overflow_entry_bytes",
                 "slice_bits", "b1", "b2", "perm_mult", "perm_inv", "b3"]
        return {n: int(g[i]) for i, n in enumerate(names)}

    def shard_hash(self, which, batch, send_regions_ptr, send_counts_ptr, lo=0, hi=None):
        n = ctypes.c_uint64(0)
        self._ck(hip().tpc_shard_hash(self._h, which, batch, lo, (1 << self.L) if hi is None else hi, send_regions_ptr, send_counts_ptr, ctypes.byref(n)))
        return n.value

    def shard_overflow_get(self, which, dst_ptr, n):
        self._ck(hip().tpc_shard_overflow_get(self._h, which, dst_ptr, n))

    def shard_overflow_set(self, which, src_ptr, n):
        self._ck(hip().tpc_shard_overflow_set(self._h, which, src_ptr, n))

    def shard_apply(self, which, batch, recv_regions_ptr, recv_counts_ptr):
        n = ctypes.c_uint64(0)
        self._ck(hip().tpc_shard_apply(self._h, which, batch, recv_regions_ptr, recv_counts_ptr, ctypes.byref(n)))
        return n.value

    def shard_pack(self, which, send_regions_ptr, send_counts_ptr, packed_ptr, world):
        """Used prefixes of the level-1 regions, destination major; returns the bytes for every destination rank."""
        out = (ctypes.c_uint64 * world)()
        self._ck(hip().tpc_shard_pack(self._h, which, send_regions_ptr, send_counts_ptr, packed_ptr, out))
        return [int(x) for x in out]

    def shard_apply_packed(self, which, batch, recv_packed_ptr, recv_counts_ptr):
        n = ctypes.c_uint64(0)
        self._ck(hip().tpc_shard_apply_packed(self._h, which, batch, recv_packed_ptr, recv_counts_ptr, ctypes.byref(n)))
        return n.value

    def shard_apply_inplace(self, which, batch, recv_regions_ptr, recv_counts_ptr, send_regions_ptr, send_counts_ptr):
        """tpc_shard_apply with this rank's own block read from the send buffers (the receive buffers' block `rank` is never read)."""
        n = ctypes.c_uint64(0)
        self._ck(hip().tpc_shard_apply_inplace(self._h, which, batch, recv_regions_ptr, recv_counts_ptr, send_regions_ptr, send_counts_ptr, ctypes.byref(n)))
        return n.value

    def shard_survivors_home(self, tmp_ptr, send_ptr, world):
        """Survivors of the last shard_apply grouped by the rank that hashed them; returns the count for every rank."""
        out = (ctypes.c_uint64 * world)()
        self._ck(hip().tpc_shard_survivors_home(self._h, tmp_ptr, send_ptr, out))
        return [int(x) for x in out]

    def shard_verify_send(self, fn, fn_count, sid_ptr, n, tmp_ptr, send_ptr, perm_ptr, world):
        """Probe addresses of functions fn.. in owner-major send order (+ the slot of every probe); returns the count for every rank."""
        out = (ctypes.c_uint64 * world)()
        self._ck(hip().tpc_shard_verify_send(self._h, fn, fn_count, sid_ptr, n, tmp_ptr, send_ptr, perm_ptr, out))
        return [int(x) for x in out]

    def shard_finish(self, sid_ptr, n, fn_count, hit_ptr, perm_ptr):
        """Marks every survivor whose fn_count answers are all 1; returns how many."""
        m = ctypes.c_uint64(0)
        self._ck(hip().tpc_shard_finish(self._h, sid_ptr, n, fn_count, hit_ptr, perm_ptr, ctypes.byref(m)))
        return m.value

    def shard_periodic_copy(self):
        """After a round's last query batch on a context with option shard_periodic_skip: positions that sent no probes take their twin's verdict."""
        self._ck(hip().tpc_shard_periodic_copy(self._h))

    def shard_verify_local(self):
        """One rank: verifies and marks the survivors of the last shard_apply(QUERY) where they are."""
        self._ck(hip().tpc_shard_verify_local(self._h))

    def shard_survivors(self, sid_ptr):
        self._ck(hip().tpc_shard_survivors(self._h, sid_ptr))

    def shard_survivor_sources(self, sid_ptr, n, src_ptr):
        self._ck(hip().tpc_shard_survivor_sources(self._h, sid_ptr, n, src_ptr))

    def shard_verify_addrs(self, fn, fn_count, sid_ptr, n, addr_ptr, owner_ptr):
        self._ck(hip().tpc_shard_verify_addrs(self._h, fn, fn_count, sid_ptr, n, addr_ptr, owner_ptr))

    def shard_probe(self, addr_ptr, n, hit_ptr):
        self._ck(hip().tpc_shard_probe(self._h, addr_ptr, n, hit_ptr))

    def shard_mark(self, sid_ptr, n):
        self._ck(hip().tpc_shard_mark(self._h, sid_ptr, n))

    def mask_export(self, dst_ptr):
        self._ck(hip().tpc_mask_export(self._h, dst_ptr))

    def mask_merge(self, src_ptr, count):
        self._ck(hip().tpc_mask_merge(self._h, src_ptr, count))

    def mask_words(self):
        return int(hip().tpc_mask_words(self._h))

    def shard_route(self, owner_ptr, n, perm_ptr, world):
        counts = np.zeros(64, dtype=np.uint64)
        self._ck(hip().tpc_shard_route(self._h, owner_ptr, n, perm_ptr, counts.ctypes.data))
        return [int(x) for x in counts[:world]]

    def shard_permute64(self, src_ptr, perm_ptr, n, dst_ptr):
        self._ck(hip().tpc_shard_permute64(self._h, src_ptr, perm_ptr, n, dst_ptr))

    def shard_select(self, sid_ptr, n, fn_count, hit_ptr, perm_ptr, out_ptr):
        m = ctypes.c_uint64(0)
        self._ck(hip().tpc_shard_select(self._h, sid_ptr, n, fn_count, hit_ptr, perm_ptr, out_ptr, ctypes.byref(m)))
        return m.value

    def mask_export_padded(self, dst_ptr, total_words):
        self._ck(hip().tpc_mask_export_padded(self._h, dst_ptr, total_words))

    def mask_or_blocks(self, blocks_ptr, count, words, out_ptr):
        self._ck(hip().tpc_mask_or_blocks(self._h, blocks_ptr, count, words, out_ptr))

    def mask_import(self, src_ptr):
        self._ck(hip().tpc_mask_import(self._h, src_ptr))

    # ---- combined exchange: the filter replicated through set-bit lists (include/twopaco_hip.h: tpc_combine_*)
    def combine_info(self, n_dest):
        g = np.zeros(8, dtype=np.uint64)
        self._ck(hip().tpc_combine_info(self._h, n_dest, g.ctypes.data))
        names = ["sparse", "slices", "windows", "cap_units", "slice_bits", "b1", "b2", "dir_entries_per_dest"]
        return {n: int(g[i]) for i, n in enumerate(names)}

    def combine_export(self, n_dest, payload_ptr, cap_units, dir_ptr):
        out = (ctypes.c_uint64 * n_dest)()
        self._ck(hip().tpc_combine_export(self._h, n_dest, payload_ptr, cap_units, dir_ptr, out))
        return [int(x) for x in out]

    def combine_merge(self, n_src, payload_ptr, src_base, dir_ptr, out_payload_ptr, out_cap_units, out_dir_ptr):
        base = (ctypes.c_uint64 * n_src)(*[int(x) for x in src_base])
        n = ctypes.c_uint64(0)
        self._ck(hip().tpc_combine_merge(self._h, n_src, payload_ptr, base, dir_ptr, out_payload_ptr, out_cap_units, out_dir_ptr, ctypes.byref(n)))
        return n.value

    def combine_import(self, n_src, n_owner, payload_ptr, src_base, dir_ptr, dir_stride):
        base = (ctypes.c_uint64 * n_src)(*[int(x) for x in src_base])
        self._ck(hip().tpc_combine_import(self._h, n_src, n_owner, payload_ptr, base, dir_ptr, dir_stride))

    def filter_copy_out(self, word0, n_words, dst_ptr):
        self._ck(hip().tpc_filter_copy_out(self._h, word0, n_words, dst_ptr))

    def filter_copy_in(self, word0, n_words, src_ptr):
        self._ck(hip().tpc_filter_copy_in(self._h, word0, n_words, src_ptr))


class Enumerator:
    """TwoPaCo::CreateEnumerator through the C++ host layer (host/vertexenumerator.h)."""

    def __init__(self, files, k, filter_bits, q=5, rounds=1, threads=1, abundance=(1 << 64) - 1, tmpdir=".",
                 out="de_bruijn.bin", seed=None, device=0, test_first=False, gpus=1, rccl=True, emulate_ranks=False, force_sharded=False):
        arr = (ctypes.c_char_p * len(files))(*[f.encode() for f in files])
        log = ctypes.c_void_p()
        if gpus > 1 or force_sharded:  # host/multigpu.h: the filter sharded by bit address over `gpus` ranks
            self._h = host().tpch_create_enumerator_mgpu(arr, len(files), k, filter_bits, q, rounds, threads, abundance, tmpdir.encode(),
                                                         out.encode(), 0 if seed is None else 1, 0 if seed is None else seed, device,
                                                         gpus, 1 if rccl else 0, 1 if emulate_ranks else 0, 1 if force_sharded else 0, ctypes.byref(log))
        else:
            self._h = host().tpch_create_enumerator(arr, len(files), k, filter_bits, q, rounds, threads, abundance, tmpdir.encode(),
                                                    out.encode(), 0 if seed is None else 1, 0 if seed is None else seed, device,
                                                    1 if test_first else 0, ctypes.byref(log))
        self.log = ctypes.string_at(log.value).decode() if log.value else ""
        if log.value:
            host().tpch_free(log)
        if not self._h:
            raise RuntimeError(host().tpch_last_error().decode())
        self.k, self.q = k, q

    def close(self):
        if getattr(self, "_h", None):
            host().tpch_enumerator_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    def vertices_count(self):
        return host().tpch_vertices_count(self._h)

    def key_words(self):
        return int(hip().tpc_key_words(self._h))

    def junction_keys_export(self, dst_ptr, cap_keys):
        n = ctypes.c_uint64(0)
        self._ck(hip().tpc_junction_keys_export(self._h, dst_ptr, cap_keys, ctypes.byref(n)))
        return n.value

    def junction_keys_import(self, src_ptr, n, append):
        self._ck(hip().tpc_junction_keys_import(self._h, src_ptr, n, 1 if append else 0))

    def get_id(self, kmer):
        return host().tpch_get_id(self._h, kmer.encode())

    def hash_seed(self):
        t = np.zeros((self.q, 5), dtype=np.uint64)
        host().tpch_hash_seed(self._h, t.ctypes.data)
        return t
