"""ctypes wrapper around oracle/libtwopaco_oracle.so + the naive (seed-free) junction oracle.

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from twopaco_amd/.

The naive oracle restates FindJunctionsNaively (reference src/graphconstructor/test.cpp:71-160).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libtwopaco_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")
INVALID_VERTEX = (1 << 63) - 1

_lib = None


def build():
    """Compile the C restatement (and, when /root/reference exists, the real reference)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if os.path.isdir("/root/reference/src") and not os.path.exists(os.path.join(REF_DIR, "twopaco_ref")):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = ctypes.CDLL(LIB_PATH)
        u64, u32, i64, p = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int64, ctypes.c_void_p
        L.orc_seed_table.argtypes = [u64, ctypes.c_int, ctypes.c_int, p]
        L.orc_create.restype = p
        L.orc_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, p]
        L.orc_destroy.argtypes = [p]
        L.orc_error.restype = ctypes.c_char_p
        L.orc_error.argtypes = [p]
        L.orc_add_record.argtypes = [p, ctypes.c_char_p, u64]
        L.orc_add_fasta.argtypes = [p, ctypes.c_char_p]
        L.orc_enumerate.argtypes = [p, ctypes.c_int, u64]
        L.orc_write_bin.argtypes = [p, ctypes.c_char_p]
        L.orc_get_id.restype = i64
        L.orc_get_id.argtypes = [p, ctypes.c_char_p]
        for name, res in [("orc_text_len", u64), ("orc_text", p), ("orc_num_records", u32), ("orc_rec_start", p),
                          ("orc_rec_len", p), ("orc_filter_nwords", u64), ("orc_filter", p), ("orc_mask_nwords", u64),
                          ("orc_mask", p), ("orc_round_mask", p), ("orc_capacity", ctypes.c_int), ("orc_num_keys", u64),
                          ("orc_keys", p), ("orc_num_out", u64), ("orc_out_seq", p), ("orc_out_pos", p),
                          ("orc_out_id", p), ("orc_true_marks", u64)]:
            getattr(L, name).restype = res
            getattr(L, name).argtypes = [p]
        L.orc_round_stat.restype = u64
        L.orc_round_stat.argtypes = [p, ctypes.c_int, ctypes.c_int]
        L.orc_hash_dump.argtypes = [p, u64, p, ctypes.c_int, p]
        L.orc_fill_only.argtypes = [p, u64, u64]
        L.orc_split_bins.argtypes = [p, p]
        L.orc_dist_begin.argtypes = [p]
        L.orc_dist_round.argtypes = [p, u64, u64, u64, p]
        L.orc_set_keys.argtypes = [p, p, u64]
        L.orc_lookup_marks.restype = u64
        L.orc_lookup_marks.argtypes = [p, p, p, u64]
        L.orc_check_only.restype = u64
        L.orc_check_only.argtypes = [p, u64, u64]
        _lib = L
    return _lib


def seed_table(seed, q, L):
    """q x 5 table (A,C,G,T,N) the reference derives from the pinned /dev/urandom stream."""
    t = np.zeros((q, 5), dtype=np.uint64)
    lib().orc_seed_table(seed, q, L, t.ctypes.data)
    return t


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class Oracle:
    """One enumeration run of the CPU restatement."""

    def __init__(self, k, L, q, table):
        self.k, self.L, self.q = k, L, q
        self.table = np.ascontiguousarray(table, dtype=np.uint64)
        self._h = lib().orc_create(k, L, q, self.table.ctypes.data)
        if not self._h:
            raise ValueError("bad oracle parameters")

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(lib().orc_error(self._h).decode())

    def add_record(self, seq):
        if isinstance(seq, str):
            seq = seq.encode()
        self._check(lib().orc_add_record(self._h, seq, len(seq)))

    def add_fasta(self, path):
        self._check(lib().orc_add_fasta(self._h, path.encode()))

    def enumerate(self, rounds=1, abundance=(1 << 64) - 1):
        self._check(lib().orc_enumerate(self._h, rounds, abundance))

    def fill_only(self, low=0, high=None):
        lib().orc_fill_only(self._h, low, (1 << self.L) if high is None else high)

    # --- per-round primitives (multi-process tests) ------------------------------------
    def dist_begin(self):
        lib().orc_dist_begin(self._h)

    def dist_round(self, low, high, abundance=(1 << 64) - 1):
        st = np.zeros(4, dtype=np.uint64)
        lib().orc_dist_round(self._h, low, high, abundance, st.ctypes.data)
        return {"true": int(st[0]), "false": int(st[1]), "table": int(st[2]), "marks": int(st[3])}

    def set_keys(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        lib().orc_set_keys(self._h, keys.ctypes.data, keys.shape[0])

    def lookup_marks(self):
        n = lib().orc_lookup_marks(self._h, None, None, 0)
        g = np.zeros(n, dtype=np.uint64)
        ids = np.zeros(n, dtype=np.int64)
        lib().orc_lookup_marks(self._h, g.ctypes.data, ids.ctypes.data, n)
        return g, ids

    def split_bins(self):
        bins = np.zeros(1 << 24, dtype=np.uint32)
        lib().orc_split_bins(self._h, bins.ctypes.data)
        return bins

    def check_only(self, low=0, high=None):
        return lib().orc_check_only(self._h, low, (1 << self.L) if high is None else high)

    def write_bin(self, path):
        self._check(lib().orc_write_bin(self._h, path.encode()))

    def get_id(self, kmer):
        return lib().orc_get_id(self._h, kmer.encode())

    # --- accessors -------------------------------------------------------------------
    @property
    def text(self):
        return _arr(lib().orc_text(self._h), lib().orc_text_len(self._h), np.uint8)

    @property
    def rec_start(self):
        return _arr(lib().orc_rec_start(self._h), lib().orc_num_records(self._h), np.uint64)

    @property
    def rec_len(self):
        return _arr(lib().orc_rec_len(self._h), lib().orc_num_records(self._h), np.uint64)

    @property
    def filter(self):
        return _arr(lib().orc_filter(self._h), lib().orc_filter_nwords(self._h), np.uint32)

    @property
    def mask(self):
        return _arr(lib().orc_mask(self._h), lib().orc_mask_nwords(self._h), np.uint32)

    @property
    def round_mask(self):
        return _arr(lib().orc_round_mask(self._h), lib().orc_mask_nwords(self._h), np.uint32)

    @property
    def capacity(self):
        return lib().orc_capacity(self._h)

    @property
    def keys(self):
        n = lib().orc_num_keys(self._h)
        return _arr(lib().orc_keys(self._h), n * self.capacity, np.uint64).reshape(n, self.capacity)

    @property
    def records(self):
        n = lib().orc_num_out(self._h)
        return (_arr(lib().orc_out_seq(self._h), n, np.uint32), _arr(lib().orc_out_pos(self._h), n, np.uint32),
                _arr(lib().orc_out_id(self._h), n, np.int64))

    def round_stats(self, rnd):
        names = ["true", "false", "table", "marks", "low", "high"]
        return {nm: lib().orc_round_stat(self._h, rnd, i) for i, nm in enumerate(names)}

    @property
    def true_marks(self):
        return lib().orc_true_marks(self._h)

    def hash_dump(self, g, c=-1):
        pn = np.zeros(2 * self.q, dtype=np.uint64)
        ad = np.zeros(self.q, dtype=np.uint64)
        lib().orc_hash_dump(self._h, g, pn.ctypes.data, c, ad.ctypes.data)
        return pn.reshape(self.q, 2), ad


# ----------------------------------------------------------------------------- .bin I/O
def read_bin(path_or_bytes):
    """JunctionPositionReader (reference src/common/junctionapi.h:81-98): list of (seq,pos,id)."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    rec = np.frombuffer(data, dtype=np.dtype([("pos", "<u4"), ("id", "<i8")]))
    out = []
    seq = 0
    for pos, jid in rec:
        if pos == 0xFFFFFFFF or jid == INVALID_VERTEX:  # junctionapi.h:92 uses &&-negation: either marks a separator
            seq += 1
            continue
        out.append((seq, int(pos), int(jid)))
    return out


# ------------------------------------------------------------------------ naive oracle
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def read_fasta_records(path):
    """Records as upper-cased strings with reference semantics (non-ACGT kept as-is, mapped later)."""
    recs = []
    cur = None
    with open(path, "rb") as f:
        data = f.read().decode("latin-1")
    i = 0
    n = len(data)
    while i < n:
        if data[i] != ">":
            raise ValueError("The FASTA header should start with a '>'")
        j = data.find("\n", i)
        j = n if j < 0 else j
        i = j + 1
        e = data.find(">", i)
        e = n if e < 0 else e
        cur = "".join(data[i:e].split()).upper()
        recs.append(cur)
        i = e
    return recs


def naive_junction_marks(chrs, k):
    """FindJunctionsNaively, test.cpp:71-160: every non-ACGT char and both sequence ends are
    fresh unique symbols; a vertex is a junction iff it has >1 distinct in- or out-symbols
    (both strands); marks = junction occurrences plus pos 0 and pos len-k of every sequence.
    Returns (set of junction k-mers incl. reverse complements, list of per-seq bool arrays)."""
    unknown = [1000]

    def fresh():
        unknown[0] += 1
        return unknown[0]

    genomes = []
    for s in chrs:
        g = [fresh()] + [c if c in _COMP else fresh() for c in s] + [fresh()]
        genomes.append(g)
        genomes.append([_COMP[c] if isinstance(c, str) else fresh() for c in reversed(g)])
    in_e, out_e = {}, {}
    for g in genomes:
        if len(g) < k:
            continue
        nbad = sum(1 for c in g[:k] if not isinstance(c, str))
        for i in range(0, len(g) - k + 1):
            if i > 0:
                nbad += (not isinstance(g[i + k - 1], str)) - (not isinstance(g[i - 1], str))
            if nbad == 0:
                v = "".join(g[i:i + k])
                if i + k < len(g):
                    out_e.setdefault(v, set()).add(g[i + k])
                if i > 0:
                    in_e.setdefault(v, set()).add(g[i - 1])
    junction = set()
    for e in (in_e, out_e):
        for v, s in e.items():
            if len(s) > 1:
                junction.add(v)
                junction.add("".join(_COMP[c] for c in reversed(v)))
    marks = []
    for s in chrs:
        m = np.zeros(len(s), dtype=bool)
        for pos in range(len(s)):
            if pos == 0 or pos == len(s) - k or s[pos:pos + k] in junction:
                m[pos] = True
        marks.append(m)
    return junction, marks


# -------------------------------------------------------------------- reference runner
def run_reference(files, k, L, q=5, rounds=1, threads=1, seed=None, out=None, tmpdir=None, debug=False,
                  abundance=None, timeout=3600):
    """Run the REAL reference binary (oracle/_ref), /dev/urandom pinned when seed is given.
    Returns (stdout text, path of the .bin)."""
    exe = os.path.join(REF_DIR, "twopaco_ref_dbg" if debug else "twopaco_ref")
    env = dict(os.environ)
    if seed is not None:
        env["LD_PRELOAD"] = os.path.join(REF_DIR, "urandom_shim.so")
        env["TPC_URANDOM_SEED"] = str(seed)
    tmpdir = tmpdir or os.path.dirname(out)
    cmd = [exe, "-k", str(k), "-f", str(L), "-q", str(q), "-r", str(rounds), "-t", str(threads),
           "--tmpdir", tmpdir, "-o", out]
    if abundance is not None:
        cmd += ["-a", str(abundance)]
    cmd += list(files)
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    if res.returncode != 0:
        raise RuntimeError("reference failed: " + res.stderr.decode())
    return res.stdout.decode(), out
