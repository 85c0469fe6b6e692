"""GPU (-m gpu): the C++ multi-GPU host (twopaco_amd/host/multigpu.cpp) -- Bloom filter sharded by bit address, one rank
(thread + device context) per GPU.  One device is all a test box has, so the ranks are emulated on it over the loopback
transport (RCCL refuses duplicate devices); the RCCL transport itself is exercised with its single possible rank.  Every
run must write the reference's bytes."""
import os
import subprocess

import pytest

from helpers import GOLDEN, case_files, golden_cases, parse_log, sha256_file

pytestmark = pytest.mark.gpu
CASES = {c["name"]: c for c in golden_cases()}
MAXU = (1 << 64) - 1


@pytest.fixture(autouse=True)
def _entry_routing(monkeypatch):
    """This module pins the ENTRY-ROUTING form of the multi-GPU first pass (the filter cut over the ranks, every hash hit of both
    passes routed to the owner of its slice: what filters beyond one GPU need).  The default of `twopaco --gpus N` while the filter
    fits a GPU is the combined exchange: tests/test_gpu_multigpu_combined.py."""
    monkeypatch.setenv("TWOPACO_MULTIGPU", "entries")


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
    return log


@pytest.mark.parametrize("name,ranks", [("rand6_k9_fp", 2), ("rand6_k9_L33", 4), ("c2_k51_r2", 2), ("edge_k5", 2), ("rand6_k25_q3", 8),
                                        ("rand6_k9_a3", 4), ("c2_k125", 2), ("m1_small", 4), ("m2_small", 8), ("rand6_k9_q12", 2), ("m2r_small", 4), ("tr_k25_L28", 2), ("tr_k31_L30_q3", 4)])
def test_emulated_ranks_write_reference_bytes(capi, tmp_path, name, ranks):
    case = CASES[name]
    out = str(tmp_path / "mg.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], rounds=case["n_rounds"],
                        abundance=case["abundance"] if case["abundance"] is not None else MAXU, tmpdir=str(tmp_path), out=out,
                        seed=case["seed"], gpus=ranks, emulate_ranks=True)
    log = _check(case, e, out)
    assert "GPUs = %d" % ranks in e.log and "loopback" in e.log
    if case["n_rounds"] == 1:
        assert log["rounds"] == case["rounds"]  # marks, true / false junctions, table size of the one round
    else:
        assert sum(r["true"] for r in log["rounds"]) == case["distinct"]
    e.close()


def test_text_window_on_sharded_context(capi):
    """Option text_window: rank r of W keeps only the words of its chunk of tiles (+ halo); such a context refuses the
    second pass (the C++ host gives it to every rank but 0)."""
    import numpy as np
    from twopaco_amd import synth
    recs, _ = synth.workload("m1", scale=0.05)   # 8 x 250 kbp = 2 M positions = 123 tiles
    text = capi.PackedText.from_codes(recs)
    held = []
    for rank in range(4):
        ctx = capi.Context(0)
        ctx.shard_config(rank, 4)
        ctx.set_option("text_window", 1)
        ctx.set_params(25, 30, 5, capi.seed_table(5, 30, seed=3))
        ctx.seq_upload(text)
        held.append(ctx.stat("text_words"))
        with pytest.raises(RuntimeError, match="window"):
            ctx.pass2_filter()
        ctx.close()
    full = capi.Context(0)
    full.set_params(25, 30, 5, capi.seed_table(5, 30, seed=3))
    full.seq_upload(text)
    whole = full.stat("text_words")
    full.close()
    assert all(h < 0.3 * whole for h in held) and sum(held) >= text.length // 32


def test_rccl_transport_single_rank(capi, tmp_path):
    """The sharded path over the RCCL transport with the one rank a single-GPU box allows: librccl is loaded, the
    communicator created, and every collective of the pass (send/recv groups to self, all-gather) runs through it."""
    case = CASES["rand6_k25_q3"]  # (not a saturated filter: a sharded filter has no direct-kernel fallback when every probe survives)
    out = str(tmp_path / "rccl.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], tmpdir=str(tmp_path), out=out, seed=case["seed"],
                        gpus=1, force_sharded=True, rccl=True)
    log = _check(case, e, out)
    assert "RCCL" in e.log and log["rounds"] == case["rounds"]
    e.close()


def test_rccl_transport_single_rank_full_size(capi, tmp_path):
    """The bench workload (62 x 5 Mbp, f=36) through the sharded path over RCCL with one rank: 8.6 GB and 19 GB exchange
    buffers cross the transport in 256 MiB messages; sha256 == the real reference's."""
    case = CASES["m2_full"]
    out = str(tmp_path / "m2rccl.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], tmpdir=str(tmp_path), out=out, seed=case["seed"],
                        threads=16, gpus=1, force_sharded=True, rccl=True)
    log = _check(case, e, out)
    assert "RCCL" in e.log and log["rounds"] == case["rounds"]
    e.close()


def test_m1_full_four_emulated_ranks(capi, tmp_path):
    """BASELINE configs[1] at full size with the filter cut over four ranks: sha256 == the real reference's."""
    case = CASES["m1_full"]
    out = str(tmp_path / "m1mg.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], tmpdir=str(tmp_path), out=out, seed=case["seed"],
                        threads=8, gpus=4, emulate_ranks=True)
    log = _check(case, e, out)
    assert log["rounds"] == case["rounds"]
    e.close()


@pytest.mark.parametrize("name,ranks", [("rand6_k25_q3", 4), ("m2_small", 2)])
def test_emulated_ranks_equal_block_exchange(capi, tmp_path, name, ranks, monkeypatch):
    """The equal-block exchange of the level-1 regions forced on (TWOPACO_EQUAL_EXCHANGE; it is the default below eight ranks, the
    exact-size packed exchange from eight on) gives the same bytes as the runs of test_emulated_ranks_write_reference_bytes."""
    monkeypatch.setenv("TWOPACO_EQUAL_EXCHANGE", "1")
    case = CASES[name]
    out = str(tmp_path / "mg.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], rounds=case["n_rounds"],
                        abundance=case["abundance"] if case["abundance"] is not None else MAXU, tmpdir=str(tmp_path), out=out,
                        seed=case["seed"], gpus=ranks, emulate_ranks=True)
    _check(case, e, out)
    e.close()


@pytest.mark.parametrize("name,ranks", [("rand6_k9_fp_r4", 2), ("c2_k51_r2", 4), ("m2_small", 4)])
def test_emulated_ranks_replicated_second_pass(capi, tmp_path, name, ranks, monkeypatch):
    """TWOPACO_REPLICATED_PASS2: union of the candidate masks + the single-GPU second pass on rank 0 (which then keeps the whole
    text) instead of the default key-sharded second pass; same bytes."""
    monkeypatch.setenv("TWOPACO_REPLICATED_PASS2", "1")
    case = CASES[name]
    out = str(tmp_path / "mg.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], rounds=case["n_rounds"],
                        abundance=case["abundance"] if case["abundance"] is not None else MAXU, tmpdir=str(tmp_path), out=out,
                        seed=case["seed"], gpus=ranks, emulate_ranks=True)
    _check(case, e, out)
    e.close()


@pytest.mark.parametrize("name,ranks", [("edge_k5", 4), ("m2_small", 4)])
def test_emulated_ranks_gathered_output(capi, tmp_path, name, ranks, monkeypatch):
    """TWOPACO_GATHER_OUTPUT: the (position, id) lists gathered on rank 0, which formats the whole stream -- the form the
    default replaced (every rank formats and writes the byte range of its own chunk of the text, multigpu.h:ShardedStream,
    covered by every other test of this file); same bytes."""
    monkeypatch.setenv("TWOPACO_GATHER_OUTPUT", "1")
    case = CASES[name]
    out = str(tmp_path / "mg.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], rounds=case["n_rounds"],
                        abundance=case["abundance"] if case["abundance"] is not None else MAXU, tmpdir=str(tmp_path), out=out,
                        seed=case["seed"], gpus=ranks, emulate_ranks=True)
    _check(case, e, out)
    e.close()


def test_saturated_filter_falls_back_to_one_gpu(capi, tmp_path):
    """A saturated filter (every first probe survives) overflows the survivor lists of the sharded pass, which has no
    scattered-kernel fallback: the run is repeated on one GPU and still writes the reference's bytes."""
    case = CASES["rand6_k9_fp"]  # f = 14 on 18 kbp: the filter is full
    out = str(tmp_path / "sat.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], tmpdir=str(tmp_path), out=out, seed=case["seed"],
                        gpus=2, emulate_ranks=True)
    _check(case, e, out)
    e.close()


def test_cli_gpus_flag(tmp_path):
    case = CASES["rand6_k9_fp_r4"]
    exe = os.path.join(os.path.dirname(GOLDEN), "..", "twopaco_amd", "bin", "twopaco")
    out = str(tmp_path / "cli.bin")
    r = subprocess.run([exe, "-k", str(case["k"]), "-f", str(case["L"]), "-q", str(case["q"]), "-r", "2", "-t", "2", "--gpus", "2", "--emulate-ranks",
                        "--seed", str(case["seed"]), "--tmpdir", str(tmp_path), "-o", out, os.path.join(GOLDEN, case["fasta"])],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, case["bin"]), "rb").read()
    assert "GPUs = 2" in r.stdout and "Distinct junctions = %d" % case["distinct"] in r.stdout
    bad = subprocess.run([exe, "-k", "9", "-f", "14", "--gpus", "3", os.path.join(GOLDEN, case["fasta"])], capture_output=True, text=True)
    assert bad.returncode == 1 and "power of two" in bad.stderr


@pytest.mark.parametrize("name,ranks", [("rand6_k9_L24_r4", 2), ("c2_k29_L26_r3", 4), ("lk_k603_r2", 2)])
def test_sharded_run_prints_the_reference_round_ranges(capi, tmp_path, name, ranks, monkeypatch):
    """`twopaco --gpus N -r R`: the split pass (InitialFilterFillerWorker, VE.h:503-583, planner VE.h:206-254) runs on one device
    in a context of its own while a whole scratch filter fits there, so the sharded run prints the SAME `Round n, lo:hi` lines
    as the reference on collision-free goldens (where the histogram is order independent), besides writing its bytes.  With
    TWOPACO_ANALYTIC_SPLIT=1 (what a run falls back to when the scratch filter does not fit) the ranges are the analytic
    quantiles: other lines, the same bytes."""
    case = CASES[name]
    out = str(tmp_path / "mg.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], rounds=case["n_rounds"], tmpdir=str(tmp_path), out=out,
                        seed=case["seed"], gpus=ranks, emulate_ranks=True)
    log = _check(case, e, out)
    assert [(r["low"], r["high"]) for r in log["rounds"]] == [(r["low"], r["high"]) for r in case["rounds"]]
    assert [r["true"] for r in log["rounds"]] == [r["true"] for r in case["rounds"]]
    e.close()
    monkeypatch.setenv("TWOPACO_ANALYTIC_SPLIT", "1")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], rounds=case["n_rounds"], tmpdir=str(tmp_path), out=out,
                        seed=case["seed"], gpus=ranks, emulate_ranks=True)
    _check(case, e, out)
    assert "analytic quantiles" in e.log
    e.close()


def test_cli_sharded_filter_checkpoint_roundtrip(tmp_path):
    """--save-filter / --load-filter with --gpus N: every rank's shard of every round goes to its own file
    (<name>[.<round>].shard<r>of<N>) and comes back instead of the sharded insert; the reloaded run writes the golden bytes; a
    checkpoint of another rank count, or made from other input files, is refused (the header carries the shard layout and a
    fingerprint of the packed text)."""
    case = CASES["rand6_k9_fp_r4"]
    exe = os.path.join(os.path.dirname(GOLDEN), "..", "twopaco_amd", "bin", "twopaco")
    fa = os.path.join(GOLDEN, case["fasta"])
    ck = str(tmp_path / "bloom.ckpt")
    # (f = 20 instead of the golden's saturated 14: the sharded pass has no fallback for a full filter; positions and ids do not depend on f)
    base = [exe, "-k", str(case["k"]), "-f", "20", "-q", str(case["q"]), "-r", "2", "-t", "2", "--tmpdir", str(tmp_path), "--gpus", "2", "--emulate-ranks"]
    out1, out2 = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    r = subprocess.run(base + ["--seed", str(case["seed"]), "--save-filter", ck, "-o", out1, fa], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for name in ("bloom.ckpt.shard0of2", "bloom.ckpt.shard1of2", "bloom.ckpt.1.shard0of2", "bloom.ckpt.1.shard1of2"):
        assert os.path.exists(str(tmp_path / name)), name
    r2 = subprocess.run(base + ["--load-filter", ck, "-o", out2, fa], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr
    assert open(out1, "rb").read() == open(out2, "rb").read()
    rounds = lambda s: [ln for ln in s.splitlines() if ln.startswith("Round ")]
    assert rounds(r.stdout) == rounds(r2.stdout) and len(rounds(r.stdout)) == 2
    # one GPU wrote the same records (ids and positions do not depend on how the filter is cut)
    out3 = str(tmp_path / "c.bin")
    r3 = subprocess.run([exe, "-k", str(case["k"]), "-f", "20", "-q", str(case["q"]), "-r", "2", "--seed", str(case["seed"]), "--tmpdir", str(tmp_path), "-o", out3, fa],
                        capture_output=True, text=True)
    assert r3.returncode == 0 and open(out3, "rb").read() == open(out1, "rb").read()
    # four ranks cannot take a two-rank checkpoint
    bad = subprocess.run(base[:-3] + ["--gpus", "4", "--emulate-ranks", "--load-filter", ck, "-o", out2, fa], capture_output=True, text=True)
    assert bad.returncode == 1 and "checkpoint" in bad.stderr
    # another input: refused by the text fingerprint
    other = os.path.join(GOLDEN, "c2.fa")
    bad = subprocess.run(base + ["--load-filter", ck, "-o", out2, other], capture_output=True, text=True)
    assert bad.returncode == 1 and "other input files" in bad.stderr


@pytest.mark.parametrize("name,ranks", [("rand6_k9_L33", 2), ("c2_k51_r2", 4), ("rand6_k9_q12", 2), ("m2_small", 8), ("edge_k5", 2)])
def test_emulated_ranks_hashes_overlapped(capi, tmp_path, name, ranks, monkeypatch):
    """TWOPACO_OVERLAP=1: every level-1 hash but the first runs on the context's second stream under the exchange and apply of the
    batch before it, and the query's first hash under the insert's last exchange (tpc_shard_plan_both / tpc_shard_hash_begin /
    _end, produced and applied overflow lists kept apart): the reference's bytes and counters, as without it."""
    monkeypatch.setenv("TWOPACO_OVERLAP", "1")
    case = CASES[name]
    out = str(tmp_path / "ov.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], rounds=case["n_rounds"], tmpdir=str(tmp_path), out=out,
                        seed=case["seed"], gpus=ranks, emulate_ranks=True)
    log = _check(case, e, out)
    if case["n_rounds"] == 1:
        assert log["rounds"] == case["rounds"]
    e.close()


def test_m1_full_four_emulated_ranks_overlapped_in_batches(capi, tmp_path, monkeypatch):
    """The same with several tile batches per rank and pass (a 64 MiB buffer budget: three batches of 256 tiles on each of the four
    ranks), so that hashes really run under the exchange of the batch before them, with the skew path's early overflow fetch in
    play: sha256 and every counter == the real reference's for BASELINE configs[1] at full size."""
    monkeypatch.setenv("TWOPACO_OVERLAP", "1")
    monkeypatch.setenv("TWOPACO_PART_BUDGET_GB", "0.0625")
    monkeypatch.setenv("TWOPACO_TIMING", "1")
    case = CASES["m1_full"]
    out = str(tmp_path / "m1ov.bin")
    e = capi.Enumerator(case_files(case, tmp_path), case["k"], case["L"], q=case["q"], tmpdir=str(tmp_path), out=out, seed=case["seed"],
                        threads=8, gpus=4, emulate_ranks=True)
    log = _check(case, e, out)
    assert log["rounds"] == case["rounds"]
    e.close()
