"""CPU: host-side logic and the C-ABI library surface (no compute calls: no GPU here)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, golden_cases, text_codes
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi(built):
    from twopaco_amd import capi as m
    return m


def test_abi_exports_every_declared_symbol(capi):
    header = open(os.path.join(ROOT, "include", "twopaco_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(tpc_[a-z0-9_]+)\s*\(", header)) - {"tpc_urandom_word"})
    assert declared == sorted(capi.HIP_SYMBOLS)
    lib = capi.hip()
    for name in declared:
        assert hasattr(lib, name), name


def test_no_cpu_fallback_without_gpu(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no HIP device"):
        capi.Context(0)


@pytest.mark.parametrize("L", [12, 20, 31, 32, 33, 36, 40])
def test_seed_table_matches_oracle(capi, L):
    for q in (1, 5, 8):
        assert (capi.seed_table(q, L, seed=99) == O.seed_table(99, q, L)).all()
    t = capi.seed_table(5, L)  # /dev/urandom
    assert int(t.max()) < (1 << L)


@pytest.mark.parametrize("fa", ["example.fa", "edge.fa", "rand6.fa", "c2.fa"])
def test_text_packer_matches_oracle(capi, fa):
    path = os.path.join(GOLDEN, fa)
    t = capi.PackedText.from_fasta([path])
    o = O.Oracle(5, 16, 1, O.seed_table(1, 1, 16))
    o.add_fasta(path)
    assert t.length == len(o.text)
    assert (text_codes(t.bases, t.nmask, t.length) == o.text).all()
    assert (t.rec_start == o.rec_start).all() and (t.rec_length == o.rec_len).all()
    # N positions carry code 0 in the packed words
    g = np.nonzero(o.text == 4)[0].astype(np.uint64)
    assert (((t.bases[g >> np.uint64(5)] >> (np.uint64(2) * (g & np.uint64(31)))) & np.uint64(3)) == 0).all()


def test_text_packer_from_codes_and_threads(capi, tmp_path):
    from twopaco_amd import synth
    recs, _ = synth.workload("m1", scale=0.001)
    files = []
    for i, r in enumerate(recs):
        p = str(tmp_path / ("g%d.fa" % i))
        synth.write_fasta(p, [r], first_id=i)
        files.append(p)
    a = capi.PackedText.from_codes(recs)
    b = capi.PackedText.from_fasta(files, threads=4)
    assert a.length == b.length and (a.bases == b.bases).all() and (a.nmask == b.nmask).all()
    assert synth.n_kmers(recs, 25) == sum(r.size - 24 for r in recs)


@pytest.mark.parametrize("threads,piece", [(1, None), (5, None), (4, 37), (3, 1)])
def test_text_packer_ragged_records_in_parallel(capi, tmp_path, threads, piece, monkeypatch):
    """Many short and empty records (several per packed word), N runs, lower case, over several files: the
    parallel pieces (host/textpack.cpp; `piece` bytes each, so records are cut at arbitrary places, inside lines
    and packed words) must give the text of the sequential from_codes path."""
    if piece is not None:
        monkeypatch.setenv("TWOPACO_PARSE_PIECE", str(piece))
    rng = np.random.default_rng(7)
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    files, recs = [], []
    for f in range(4):
        path = str(tmp_path / ("r%d.fa" % f))
        with open(path, "w") as out:
            for r in range(60):
                n = int(rng.choice([0, 1, 2, 5, 31, 32, 33, 63, 64, 65, 100, 1000, 4097]))
                codes = rng.integers(0, 4, n).astype(np.uint8)
                if n > 4 and rng.random() < 0.5:
                    a = int(rng.integers(0, n - 1))
                    codes[a:a + int(rng.integers(1, 40))] = 4
                recs.append(codes)
                text = letters[codes].tobytes().decode()
                if rng.random() < 0.3:
                    text = text.lower()
                out.write(">s%d_%d\n" % (f, r))
                for i in range(0, n, 70):
                    out.write(text[i:i + 70] + "\n")
        files.append(path)
    a = capi.PackedText.from_codes(recs)
    b = capi.PackedText.from_fasta(files, threads=threads)
    assert a.length == b.length
    assert (a.rec_start == b.rec_start).all() and (a.rec_length == b.rec_length).all()
    assert (a.bases == b.bases).all() and (a.nmask == b.nmask).all()


def test_fasta_errors(capi, tmp_path):
    bad = tmp_path / "bad.fa"
    bad.write_text(">x\nACGT!ACGT\n")
    with pytest.raises(RuntimeError, match="invalid character"):
        capi.PackedText.from_fasta([str(bad)])
    nohdr = tmp_path / "nohdr.fa"
    nohdr.write_text("ACGT\n")
    with pytest.raises(RuntimeError, match="should start with"):
        capi.PackedText.from_fasta([str(nohdr)])
    with pytest.raises(RuntimeError, match="Can't open"):
        capi.PackedText.from_fasta([str(tmp_path / "missing.fa")])


def test_cli_flags_and_errors(built):
    exe = os.path.join(ROOT, "twopaco_amd", "bin", "twopaco")
    fa = os.path.join(GOLDEN, "example.fa")
    r = subprocess.run([exe, "-k", "12", "-f", "20", fa], capture_output=True, text=True)
    assert r.returncode == 1 and "value of K must be odd" in r.stderr
    r = subprocess.run([exe, "-k", "11", fa], capture_output=True, text=True)
    assert r.returncode == 1 and "Error:" in r.stderr
    r = subprocess.run([exe, "-k", "11", "-f", "20", "--filtermemory", "1", fa], capture_output=True, text=True)
    assert r.returncode == 1
    r = subprocess.run([exe, "-f", "20"], capture_output=True, text=True)
    assert r.returncode == 1 and "filenames" in r.stderr
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "-k", "11", "-f", "20", fa], capture_output=True, text=True)
        assert r.returncode == 1 and "GPU" in r.stderr  # fails loudly, no CPU path
        assert "Vertex length = 11" in r.stdout
    # --filtermemory GB -> filter bits = log2(GB * 8e9) truncated (reference constructor.cpp:158): the header is logged before any device work
    for gb, bits in (("1.0", 32), ("0.5", 31), ("4", 34), ("0.002", 23), ("34.4", 38)):
        r = subprocess.run([exe, "-k", "11", "--filtermemory", gb, fa], capture_output=True, text=True)
        assert "Filter size = %d\n" % (1 << bits) in r.stdout, (gb, r.stdout[:300])
    # the checkpoint flags parse (and need their value)
    r = subprocess.run([exe, "-k", "11", "-f", "20", "--save-filter"], capture_output=True, text=True)
    assert r.returncode == 1 and "Missing a value" in r.stderr
    r = subprocess.run([exe, "-k", "11", "-f", "20", "--load-filter", "/nonexistent/filter.bin", fa], capture_output=True, text=True)
    assert r.returncode == 1 and "Can't open the Bloom filter checkpoint" in r.stderr


def test_junction_api_header_roundtrip(built, tmp_path):
    """host/junctionapi.h reads the reference's bytes and writes them back unchanged."""
    src = r'''
    #include "junctionapi.h"
    #include <iostream>
    int main(int argc, char ** argv) {
        TwoPaCo::JunctionPositionReader reader(argv[1]);
        TwoPaCo::JunctionPositionWriter writer(argv[2]);
        TwoPaCo::JunctionPosition pos;
        size_t n = 0;
        while (reader.NextJunctionPosition(pos)) { writer.WriteJunction(pos); ++n; }
        std::cout << n << std::endl;
        return 0;
    }'''
    cpp = tmp_path / "rt.cpp"
    cpp.write_text(src)
    exe = str(tmp_path / "rt")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "twopaco_amd", "host"), str(cpp), "-o", exe])
    for case in golden_cases():
        if not case.get("bin"):
            continue
        out = str(tmp_path / "copy.bin")
        n = subprocess.check_output([exe, os.path.join(GOLDEN, case["bin"]), out]).decode().strip()
        assert int(n) == case["true_marks"]
        assert open(out, "rb").read() == open(os.path.join(GOLDEN, case["bin"]), "rb").read()


REF_SRC = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="build container only: needs the reference sources where they lie")
def test_reference_main_and_selftest_compile_against_host_headers(built, tmp_path):
    """INTEGRATION.md Option A: the reference's own constructor.cpp (its main) and test.cpp (its --test) compile
    UNCHANGED against twopaco_amd/host/vertexenumerator.h and link against libtwopaco_host.so.  The two files are
    copied to a temporary directory only (a quoted include resolves next to the including file first); nothing of
    the reference enters the repository."""
    import shutil
    for f in ("constructor.cpp", "test.cpp", "test.h"):
        shutil.copy(os.path.join(REF_SRC, "graphconstructor", f), str(tmp_path / f))
    host = os.path.join(ROOT, "twopaco_amd", "host")
    lib = os.path.join(ROOT, "twopaco_amd", "lib")
    objs = []
    for f in ("constructor.cpp", "test.cpp"):
        obj = str(tmp_path / (f[:-4] + ".o"))
        # -I host first: vertexenumerator.h, junctionapi.h, dnachar.h are ours; src/common only supplies the header-only TCLAP
        subprocess.check_call(["g++", "-std=c++14", "-O1", "-w", "-c", str(tmp_path / f), "-I", host, "-I", os.path.join(REF_SRC, "common"), "-o", obj])
        objs.append(obj)
    exe = str(tmp_path / "twopaco_dropin")
    subprocess.check_call(["g++", "-o", exe] + objs + ["-L", lib, "-ltwopaco_host", "-ltwopaco_hip", "-lpthread", "-Wl,-rpath," + lib])
    r = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert "--filtersize" in r.stdout and "--kvalue" in r.stdout  # the reference's own TCLAP usage text
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "-k", "11", "-f", "20", os.path.join(GOLDEN, "example.fa"), "-o", str(tmp_path / "o.bin")], capture_output=True, text=True)
        assert r.returncode == 1 and "GPU" in r.stderr  # the reference's main reports our runtime_error; no CPU path


def test_native_synth_equals_numpy(built, monkeypatch):
    """host/synthgen.cpp generates the workloads of twopaco_amd/synth.py bit for bit (the goldens were made from the numpy code)."""
    from twopaco_amd import synth
    sizes = (("m1", 0.01), ("m2", 0.1), ("m3", 0.002))
    fast = {w: synth.workload(w, scale=s)[0] for w, s in sizes}
    monkeypatch.setenv("TPC_SYNTH_NUMPY", "1")
    for (w, s) in sizes:
        slow = synth.workload(w, scale=s)[0]
        assert len(slow) == len(fast[w])
        for a, b in zip(slow, fast[w]):
            assert a.dtype == b.dtype == np.uint8 and (a == b).all()
    assert sum(int((r == 4).sum()) for r in fast["m2"]) > 0
