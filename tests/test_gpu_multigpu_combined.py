"""GPU (-m gpu): `twopaco --gpus N` with the combined exchange -- its default while the filter fits one GPU (host/multigpu.cpp:
CombinedFirstPass; include/twopaco_hip.h tpc_combine_*): every rank (thread + context, emulated on the one device of a test box
over the loopback transport) keeps the whole filter, inserts its chunk of the text, the set bits of every slice travel as 16-bit
lists, the query is local.  Every run must write the reference's bytes and print its counters; every form of the exchange is
forced in turn (TWOPACO_COMBINE = gather | scatter | dense) besides the bytes model's own choice."""
import os
import subprocess

import pytest

from helpers import GOLDEN, case_files, golden_cases, parse_log, sha256_file

pytestmark = pytest.mark.gpu
CASES = {c["name"]: c for c in golden_cases()}
MAXU = (1 << 64) - 1
EXE = os.path.join(os.path.dirname(GOLDEN), "..", "twopaco_amd", "bin", "twopaco")


@pytest.fixture(scope="module")
def capi():
    from twopaco_amd import capi as m
    m.hip()
    m.host()
    return m


def _check(case, e, out):
    assert os.path.getsize(out) == case["bin_bytes"]
    assert sha256_file(out) == case["bin_sha256"]
    assert e.vertices_count() == case["distinct"]
    log = parse_log(e.log)
    assert log["true_marks"] == case["true_marks"]
    assert "replicated through set-bit lists" in e.log
    return log


@pytest.mark.parametrize("name,ranks,mode", [("rand6_k9_L33", 2, None), ("rand6_k9_L33", 4, None), ("c2_k51_r2", 2, "scatter"), ("edge_k5", 2, "gather"), ("rand6_k25_q3", 8, None),
                                             ("rand6_k9_a3", 4, "dense"), ("c2_k125", 2, None), ("m1_small", 4, None), ("m2_small", 8, None), ("m2_small", 2, None),
                                             ("rand6_k9_q12", 2, "scatter"), ("m2r_small", 4, None), ("tr_k25_L28", 2, None), ("tr_k31_L30_q3", 4, "gather"),
                                             ("rand6_k9_L24_r4", 2, None), ("example_k15_r3", 8, None),
                                             # more than 16 hash functions: every rank runs the closed-form kernels over its chunk, the dense filters are OR-reduced
                                             ("rand6_k9_q20", 2, None), ("rand6_k9_q20_fp_r2", 4, None)])
def test_emulated_ranks_write_reference_bytes(capi, tmp_path, name, ranks, mode, monkeypatch):
    if mode:
        monkeypatch.setenv("TWOPACO_COMBINE", mode)
    case = CASES[name]
    out = str(tmp_path / "mg.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], rounds=case["n_rounds"],
                        abundance=case["abundance"] if case["abundance"] is not None else MAXU, tmpdir=str(tmp_path), out=out,
                        seed=case["seed"], gpus=ranks, emulate_ranks=True)
    log = _check(case, e, out)
    assert "GPUs = %d" % ranks in e.log and "loopback" in e.log
    if case["n_rounds"] == 1:
        assert log["rounds"] == case["rounds"]  # marks, true / false junctions, table size of the one round: the reference's
    else:
        assert [(r["low"], r["high"]) for r in log["rounds"]] == [(r["low"], r["high"]) for r in case["rounds"]]
        assert [r["true"] for r in log["rounds"]] == [r["true"] for r in case["rounds"]]
    e.close()


@pytest.mark.parametrize("name,ranks", [("m2_small", 4), ("c2_k51_r2", 2)])
def test_replicated_second_pass_and_gathered_output(capi, tmp_path, name, ranks, monkeypatch):
    """TWOPACO_REPLICATED_PASS2=1: the ranks' marks are OR-reduced and rank 0 runs the single-GPU second pass on the whole text."""
    monkeypatch.setenv("TWOPACO_REPLICATED_PASS2", "1")
    case = CASES[name]
    out = str(tmp_path / "mg.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], rounds=case["n_rounds"], tmpdir=str(tmp_path), out=out,
                        seed=case["seed"], gpus=ranks, emulate_ranks=True)
    _check(case, e, out)
    e.close()


@pytest.mark.parametrize("name,ranks", [("rand6_k9_a3", 2), ("c2_k125", 4)])
def test_per_position_records_in_the_second_pass(capi, tmp_path, name, ranks, monkeypatch):
    """TWOPACO_PASS2_AGGREGATE=0: a (key, prev | next) record per marked position travels to the key's owner (rounds 3-5) instead of one
    aggregated record per distinct key of a rank's marks (tpc_pass2_aggregate_records, the default): same bytes, same counters -- with
    an abundance cut and with four-word keys."""
    monkeypatch.setenv("TWOPACO_PASS2_AGGREGATE", "0")
    case = CASES[name]
    out = str(tmp_path / "mg.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], rounds=case["n_rounds"],
                        abundance=case["abundance"] if case["abundance"] is not None else MAXU, tmpdir=str(tmp_path), out=out,
                        seed=case["seed"], gpus=ranks, emulate_ranks=True)
    _check(case, e, out)
    e.close()


def test_m1_full_four_emulated_ranks(capi, tmp_path):
    """BASELINE configs[1] at its full size (8 x 5 Mbp, f = 32) on four emulated ranks: sha256 and counters of the real reference."""
    case = CASES["m1_full"]
    out = str(tmp_path / "m1.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], tmpdir=str(tmp_path), out=out, seed=case["seed"],
                        threads=8, gpus=4, emulate_ranks=True)
    log = _check(case, e, out)
    assert log["rounds"] == case["rounds"]
    e.close()


def test_cli_checkpoint_roundtrip(tmp_path):
    """--save-filter / --load-filter with --gpus N under the combined exchange: every rank holds the WHOLE filter, so rank 0 writes
    one unsharded file per round and every rank reads it back instead of inserting; a one-GPU run reads the same files."""
    case = CASES["rand6_k9_fp_r4"]
    fa = os.path.join(GOLDEN, case["fasta"])
    ck = str(tmp_path / "bloom.ckpt")
    base = [EXE, "-k", str(case["k"]), "-f", "20", "-q", str(case["q"]), "-r", "2", "-t", "2", "--tmpdir", str(tmp_path)]
    multi = ["--gpus", "2", "--emulate-ranks"]
    out1, out2, out3 = str(tmp_path / "a.bin"), str(tmp_path / "b.bin"), str(tmp_path / "c.bin")
    r = subprocess.run(base + multi + ["--seed", str(case["seed"]), "--save-filter", ck, "-o", out1, fa], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(ck) and os.path.exists(ck + ".1") and not os.path.exists(ck + ".shard0of2")
    r2 = subprocess.run(base + multi + ["--load-filter", ck, "-o", out2, fa], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr
    assert open(out1, "rb").read() == open(out2, "rb").read()
    r3 = subprocess.run(base + ["--load-filter", ck, "-o", out3, fa], capture_output=True, text=True)  # one GPU, the same checkpoint
    assert r3.returncode == 0, r3.stderr
    assert open(out1, "rb").read() == open(out3, "rb").read()
    rounds = lambda s: [ln for ln in s.splitlines() if ln.startswith("Round ")]
    assert rounds(r.stdout) == rounds(r2.stdout) == rounds(r3.stdout) and len(rounds(r.stdout)) == 2


def test_saturated_filter_falls_back_to_one_gpu(capi, tmp_path):
    """A saturated filter (every first probe survives) overflows the survivor lists of a rank's local query, which -- on a rank that
    holds only its window of the text -- has no scattered-kernel fallback: the run is repeated on one GPU, same bytes."""
    case = CASES["rand6_k9_fp"]
    out = str(tmp_path / "sat.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], tmpdir=str(tmp_path), out=out, seed=case["seed"],
                        gpus=2, emulate_ranks=True)
    assert sha256_file(out) == case["bin_sha256"] and e.vertices_count() == case["distinct"]
    e.close()


def test_timing_line_names_the_exchange(tmp_path):
    """TWOPACO_TIMING=1 prints the bytes model's three figures and its choice (the same arithmetic dist.py prints: tpc_combine_choose)."""
    case = CASES["m2_small"]
    files = case_files(case, tmp_path)
    out = str(tmp_path / "t.bin")
    env = dict(os.environ, TWOPACO_TIMING="1")
    r = subprocess.run([EXE, "-k", str(case["k"]), "-f", str(case["L"]), "-q", str(case["q"]), "--gpus", "4", "--emulate-ranks", "--seed", str(case["seed"]),
                        "--tmpdir", str(tmp_path), "-o", out] + files, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert sha256_file(out) == case["bin_sha256"]
    assert "combined exchange: export" in r.stderr and "reduce-scatter + all-gather" in r.stderr and "combined first pass" in r.stderr


@pytest.mark.parametrize("ranks,mode", [(4, "scatter"), (2, "gather"), (4, "dense")])
def test_both_hosts_move_the_same_bytes(tmp_path, ranks, mode, monkeypatch):
    """One protocol, one implementation of its arithmetic: the C++ host (multigpu.cpp) and the torch.distributed
    driver (dist.py:Combined) take the form of the exchange from the same library call (tpc_combine_choose) and move the same blocks --
    on m2_small at `ranks` emulated ranks rank 0 of either host reports the same number of bytes received, to the byte."""
    import pickle
    import re
    import torch.multiprocessing as mp
    from dist_worker import combined_worker
    from test_dist_cpu import free_port
    case = CASES["m2_small"]
    files = case_files(case, tmp_path)
    env = dict(os.environ, TWOPACO_TIMING="1", TWOPACO_COMBINE=mode)
    out = str(tmp_path / "t.bin")
    r = subprocess.run([EXE, "-k", str(case["k"]), "-f", str(case["L"]), "-q", str(case["q"]), "--gpus", str(ranks), "--emulate-ranks", "--seed", str(case["seed"]),
                        "--tmpdir", str(tmp_path), "-o", out] + files, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert sha256_file(out) == case["bin_sha256"]
    m = re.search(r"combined exchange: (\w+), [0-9.]+ MB received by rank 0 \((\d+) bytes\)", r.stderr)
    assert m and m.group(1) == mode
    cxx_bytes = int(m.group(2))
    from twopaco_amd import synth
    s = case["synth"]
    spec = {"workload": s["workload"], "scale": s["scale"], "k": case["k"], "L": case["L"], "q": case["q"], "seed": case["seed"], "ranges": [(0, 1 << case["L"])],
            "abundance": MAXU, "options": {"slice_bits": min(20, case["L"] - max(2, 2 * (ranks.bit_length() - 1)))}, "mode": mode, "sharded_pass2": "records", "text_window": True}
    res = str(tmp_path / "res.pkl")
    mp.spawn(combined_worker, args=(ranks, free_port(), spec, res), nprocs=ranks, join=True)
    with open(res, "rb") as f:
        gathered = pickle.load(f)
    py = gathered[0]["rounds"][0]["combine"]
    assert py["exchange_bytes_received"] == cxx_bytes, (py, cxx_bytes)
