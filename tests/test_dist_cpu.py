"""CPU, world_size 2 (gloo): the multi-GPU orchestration of twopaco_amd/dist.py -- vertex-hash
ranges, union of the per-rank junction keys, id lookup, record merge -- with the oracle standing in
for the HIP kernels.  The merged records must equal the single-process oracle's."""
import os
import pickle
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from helpers import case_files, golden_cases
from oracle import oracle as O
from twopaco_amd import dist as tdist

CASES = {c["name"]: c for c in golden_cases()}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(case, files, world, tmp_path, use_gpu=False):
    from dist_worker import worker
    res = str(tmp_path / "res.pkl")
    mp.spawn(worker, args=(world, free_port(), case, files, use_gpu, res), nprocs=world, join=True)
    with open(res, "rb") as f:
        return pickle.load(f)


def check_against_single(case, files, gathered):
    abundance = case["abundance"] if case["abundance"] is not None else (1 << 64) - 1
    o = O.Oracle(case["k"], case["L"], case["q"], O.seed_table(case["seed"], case["q"], case["L"]))
    for f in files:
        o.add_fasta(f)
    o.enumerate(rounds=1, abundance=abundance)
    J = gathered[0][2]
    assert all(g[2] == J for g in gathered) and J == len(o.keys) == case["distinct"]
    recs = tdist.merge_records([(g[0], g[1]) for g in gathered], o.rec_start, o.rec_len, case["k"], J)
    seq, pos, ids = o.records
    assert recs == list(zip(seq.tolist(), pos.tolist(), ids.tolist()))
    assert len(recs) == case["true_marks"]
    # ranges are disjoint and cover [0, 2^L]
    rng = sorted(g[4] for g in gathered)
    assert rng[0][0] == 0 and rng[-1][1] == 1 << case["L"]
    for a, b in zip(rng, rng[1:]):
        assert a[1] + 1 == b[0]


def test_ranges_cover_and_balance():
    for L in (12, 20, 36):
        for world in (1, 2, 3, 8):
            r = tdist.vertex_hash_ranges(L, world)
            assert r[0][0] == 0 and r[-1][1] == 1 << L
            assert all(a[1] + 1 == b[0] for a, b in zip(r, r[1:]))
    # density of min(u, v) of two uniforms is 2(1-x): each range holds 1/world of the mass
    rng = np.random.default_rng(1)
    x = np.minimum(rng.random(200000), rng.random(200000)) * (1 << 30)
    for lo, hi in tdist.vertex_hash_ranges(30, 4):
        frac = np.mean((x >= lo) & (x <= hi))
        assert abs(frac - 0.25) < 0.01


@pytest.mark.parametrize("name", ["rand6_k9_fp", "edge_k5", "c2_k51_r2", "rand6_k9_a3"])
def test_two_ranks_equal_single_process(name, tmp_path):
    case = CASES[name]
    files = case_files(case, tmp_path)
    gathered = run_world(case, files, 2, tmp_path)
    check_against_single(case, files, gathered)


def test_three_ranks(tmp_path):
    case = CASES["rand6_k25_q3"]
    files = case_files(case, tmp_path)
    check_against_single(case, files, run_world(case, files, 3, tmp_path))


@pytest.mark.parametrize("world", [2, 4])
def test_address_sharded_collectives(world, tmp_path, p2p=False):
    """Routing primitives of the address-sharded driver (equal-block all_to_all of the level-1 regions,
    variable all_to_all of probe addresses and of the answers coming back, all_gather of the masks)."""
    from dist_worker import comm_worker
    res = str(tmp_path / "res.pkl")
    mp.spawn(comm_worker, args=(world, free_port(), res, p2p), nprocs=world, join=True)
    with open(res, "rb") as f:
        got = pickle.load(f)
    for r in range(world):
        assert got[r]["equal"] == [16 * s + r for s in range(world) for _ in range(5)]
        assert got[r]["equal_big"] == [i + 1000 * r + 100000 * s for s in range(world) for i in range(40)]
        assert got[r]["equal_skip"] == [[i + 1000 * r + 100000 * s for i in range(40)] for s in range(world) if s != r]
        assert got[r]["var_big"] == ([i + 1000 * r + 100000 * s for s in range(world) for i in range(7 * s + 3 * r)], [7 * s + 3 * r for s in range(world)])
        assert got[r]["var_out"] == (got[r]["var_big"][0], True if p2p else None)
        want, rc = [], []
        for s in range(world):
            c = (s + r) % 3
            want += [100 * s + r] * c
            rc.append(c)
        assert got[r]["var"] == (want, rc)
        # rank r asked d about counts[d] elements and gets one answer each, in the order sent
        counts = [(r + d) % 3 for d in range(world)]
        assert got[r]["back"] == ([d + 1 for d in range(world) for _ in range(counts[d])], counts)
        assert got[r]["gather"] == [[10 * s + i for i in range(3)] for s in range(world)]
        assert got[r]["max"] == [world - 1, 7]


@pytest.mark.parametrize("world", [2, 3, 4])
def test_address_sharded_collectives_send_recv_form(world, tmp_path):
    """The same exchanges in the grouped send/recv form that runs over RCCL (chunked messages per peer), here over gloo with
    16-byte chunks: only an 8-GPU node runs that form with more than one rank otherwise."""
    test_address_sharded_collectives(world, tmp_path, p2p=True)


def test_bench_gpus_flag_launches_ranks(tmp_path):
    """`bench.py --gpus 2` with no launcher in the environment starts two ranks by itself (bench.launch_ranks:
    torch.distributed.run as a child process) and rank 0 prints one JSON line with n_gpus = ranks = 2, checked against the
    golden counters.  tests/bench_injected.py composes the product's launcher and dist.bench_main with the oracle as the rank
    backend over gloo: what is under test is the launch / timing / reporting path the driver's scaling run depends on."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TPC_DIST_BACKEND="gloo", PYTHONPATH=os.path.join(root, "tests") + os.pathsep + root)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "tests", "bench_injected.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["steps"] == 2 and out["backend"] == "injected"
    assert out["config"]["decomposition"] == "ranges"
    case = CASES["rand6_k9_fp"]
    assert out["result"]["junctions"] == case["distinct"] and out["result"]["junction_occurrences"] == case["true_marks"]
    assert out["result_equals_reference_golden"] is True
    for key in ("metric", "value", "unit", "ms_per_step", "scaling", "kernel_ms_rank0", "phase_ms_rank0_per_step", "all_to_all_GBs_rank0",
                "exchange_bytes_rank0_per_step", "roofline", "result", "rccl_version", "collective_timeout_s"):
        assert key in out, key
    assert out["collective_timeout_s"] > 0  # every collective phase runs under the wall-clock watchdog (TPC_DIST_TIMEOUT_S)
    # the launcher's world must be the --gpus the line claims: a mismatch is an error on every rank, not a mislabelled line
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    r = subprocess.run(cmd, env=env2, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "was launched with WORLD_SIZE = 1" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # a result that differs from the reference's counters: no line, non-zero exit
    r = subprocess.run(cmd + ["--golden-junctions", str(case["distinct"] + 1)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "differs from the reference golden" in r.stderr


def test_watchdog_ends_a_rank_stuck_in_a_collective(tmp_path):
    """A peer that never arrives: the waiting rank's PhaseWatchdog writes the phase's name and ends the process with exit code 17
    within TPC_DIST_TIMEOUT_S -- the first multi-GPU run cannot hang (and torch.distributed.run then stops the other ranks)."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = free_port()
    env = dict(os.environ, TPC_DIST_TIMEOUT_S="3")
    script = os.path.join(root, "tests", "dist_failfast.py")
    late = subprocess.Popen([sys.executable, script, "hang", "1", "2", str(port)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, script, "hang", "0", "2", str(port)], env=env, capture_output=True, text=True, timeout=120)
    finally:
        late.kill()
        late.wait()
    assert r.returncode == 17, (r.returncode, r.stderr[-1500:])
    assert "collective phase 'query batch 0:all_to_all(variable)' exceeded TPC_DIST_TIMEOUT_S = 3 s" in r.stderr
    assert time.time() - t0 < 60


def test_local_failure_is_agreed_before_the_collective(tmp_path):
    """One rank's local work fails between two collectives: its failure flag rides in the count exchange of the next variable
    all_to_all (and in every max all-reduce), so BOTH ranks raise DistAbort naming the phase and the failed rank, no payload moves
    and nobody waits (host/multigpu.cpp does the same behind its rank barrier)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = free_port()
    env = dict(os.environ, TPC_DIST_TIMEOUT_S="30")
    script = os.path.join(root, "tests", "dist_failfast.py")
    procs = [subprocess.Popen([sys.executable, script, "agree", str(r), "2", str(port)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert "tpc_shard_apply: out of device memory" in outs[1][0] and "rank(s) [1] failed" in outs[0][0]


def test_product_has_no_injected_backend_seam():
    """The seam that lets a test backend (the oracle) stand behind a bench line lives under tests/ only: the product's
    distributed driver never names the oracle, and no environment variable swaps the rank backend."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "twopaco_amd", "dist.py")) as f:
        text = f.read()
    assert "TPC_BENCH_BACKEND" not in text and "oracle" not in text.lower()
    with open(os.path.join(root, "bench.py")) as f:
        assert "TPC_BENCH_BACKEND" not in f.read()
