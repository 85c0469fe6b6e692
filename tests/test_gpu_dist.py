"""GPU (-m gpu): two ranks (gloo rendezvous, both contexts on GPU 0) run the sharded step through
the C-ABI; merged records == single-process oracle."""
import pytest

from helpers import case_files, golden_cases
from test_dist_cpu import check_against_single, run_world

pytestmark = pytest.mark.gpu
CASES = {c["name"]: c for c in golden_cases()}


@pytest.mark.parametrize("name", ["rand6_k9_fp", "c2_k51_r2", "rand6_k9_a3", "rand6_k9_L33"])
def test_two_ranks_on_gpu(name, tmp_path):
    case = CASES[name]
    files = case_files(case, tmp_path)
    check_against_single(case, files, run_world(case, files, 2, tmp_path, use_gpu=True))


_SINGLE = {}


def _single_gpu_line(base):
    """The plain single-GPU bench line of the same workload (run once per session: the four decompositions compare with one reference)."""
    import json
    import os
    import subprocess
    key = tuple(base)
    if key not in _SINGLE:
        env1 = {k: v for k, v in os.environ.items() if k != "TPC_FORCE_DIST"}
        out1 = subprocess.run(list(base), env=env1, capture_output=True, text=True, timeout=600)
        assert out1.returncode == 0, out1.stderr[-2000:]
        _SINGLE[key] = json.loads([l for l in out1.stdout.splitlines() if l.startswith("{")][-1])
    return _SINGLE[key]


@pytest.mark.parametrize("decomposition", ["ranges", "address", "address+positions", "address+replicated"])
def test_bench_single_rank_over_rccl(decomposition):
    """bench.py's distributed path with the real backend ("nccl" = RCCL) and one rank: the collectives
    of both decompositions run on device tensors; the result equals the plain single-GPU bench's."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--workload", "m1", "--scale", "0.1",
            "--no-cpu-baseline", "--e2e-runs", "0"]  # (the end-to-end leg has its own tests: test_gpu_e2e / the whole-line tests below)
    env = dict(os.environ, TPC_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    env.pop("TPC_DIST_BACKEND", None)
    if "+" in decomposition:  # the other two forms of the second pass (default: key-sharded, records, text windows)
        decomposition, env["TPC_PASS2"] = decomposition.split("+")
    out = subprocess.run(base + ["--decomposition", decomposition], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    dist_line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    single = _single_gpu_line(base)
    assert dist_line["result"]["junctions"] == single["result"]["junctions"] > 0
    assert dist_line["result"]["junction_occurrences"] == single["result"]["junction_occurrences"]
    assert dist_line["result"]["candidate_marks"] == single["result"]["candidate_marks"]


def test_bench_auto_value_is_the_address_sharded_filter_and_ranges_keep_their_record():
    """`--decomposition auto` times both decompositions; the line's value is ALWAYS the address-sharded filter's (one decomposition at
    every N: the per-N values are a scaling curve of the north-star design), the vertex-hash ranges keep their full record under
    "ranges", and both carry the same (reference) counters."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TPC_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29545", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", TPC_MULTIGPU="entries")
    env.pop("TPC_DIST_BACKEND", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "m1", "--scale", "0.1",
                          "--no-cpu-baseline", "--e2e-runs", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["headline_decomposition"] == "address" == line["config"]["decomposition"]
    assert "ranges" in line and "address" not in line and "headline" not in line["config"]
    assert line["value"] > 0 and line["ranges"]["value"] > 0
    assert line["result"] == line["ranges"]["result"] and line["result"]["junctions"] > 0
    assert "phase_ms_rank0_per_step" in line and line["phase_ms_rank0_per_step"]["query_apply"] > 0


def test_bench_two_ranks_on_one_gpu_whole_line():
    """`bench.py --gpus 2` as the driver starts it, with gloo standing in for RCCL and both ranks on GPU 0 (TPC_DIST_BACKEND=gloo): the launcher,
    both decompositions at world 2 through the C-ABI (address: tight equal blocks with the own block in place, fused verification calls,
    key-sharded second pass; ranges), the address-sharded filter's value in the line, the ranges' record beside it, one JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TPC_FORCE_DIST")}
    env["TPC_DIST_BACKEND"] = "gloo"
    env["TPC_MULTIGPU"] = env["TWOPACO_MULTIGPU"] = "entries"  # this test pins the entry-routing exchange (filters beyond one GPU); the default is below
    env["TPC_E2E_EMULATE_RANKS"] = "1"  # the end-to-end leg: `twopaco --gpus 2 --emulate-ranks` (both ranks of the C++ host on this one device)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "m1", "--scale", "0.1",
                          "--no-cpu-baseline", "--e2e-runs", "1", "--e2e-settle", "0.5"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["backend"] == "gloo"
    assert line["headline_decomposition"] == "address" == line["config"]["decomposition"] and "address" not in line
    assert line["value"] > 0 and line["ranges"]["value"] > 0
    assert line["result"] == line["ranges"]["result"] and line["result"]["junctions"] > 0
    assert line["region_exchange"].startswith("equal blocks") and line["survivors_rank0"][0][0] > 0
    # the product's host beside the driver's figures: its own rounds, phases and bytes
    assert line["e2e_failed"] is False, line["e2e"]
    cx = line["cxx_host"]
    assert cx["rounds_ms"] > 0 and cx["kmers_per_sec"] > 0
    assert {"insert all-to-all", "query all-to-all"} <= set(cx["sharded_first_pass_ms_rank0"]) and cx["region_bytes_sent_rank0"] > 0
    assert cx["all_to_all_GBs_rank0"] is None or cx["all_to_all_GBs_rank0"] > 0  # (None: the loopback copies of this small input took no measurable time)


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_combined_exchange_whole_line(ranks, tmp_path):
    """`bench.py --gpus N` with its default exchange while the filter fits a GPU -- the filter replicated through set-bit lists
    (twopaco_amd/dist.py:Combined) -- gloo standing in for RCCL, the ranks taking turns on GPU 0 (TPC_DIST_SERIALIZE) so that the
    library-call times the link model is built from are those of a rank alone on its device: counters equal to the ranges', the
    model's inputs and prediction in the line, the C++ host's end-to-end leg through the same exchange."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TPC_FORCE_DIST", "TPC_MULTIGPU", "TWOPACO_MULTIGPU")}
    env["TPC_DIST_BACKEND"] = "gloo"
    env["TPC_DIST_SERIALIZE"] = str(tmp_path / "device.lock")
    env["TPC_E2E_EMULATE_RANKS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--workload", "m1", "--scale", "0.1",
                          "--no-cpu-baseline", "--e2e-runs", "1", "--e2e-settle", "0.5"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == ranks and line["backend"] == "gloo" and line["headline_decomposition"] == "address"
    assert line["result"] == line["ranges"]["result"] and line["result"]["junctions"] > 0
    assert ("all-gather of the exports" if ranks == 2 else "reduce-scatter") in line["multi_gpu_exchange"]
    assert line["exchange_bytes_rank0_per_step"] > 0 and line["combine_rank0"]["export_bytes"] > 0
    m = line["model"]
    assert m["compute_ms"] > 0 and m["wire_ms_total"] > 0 and abs(m["predicted_ms_no_overlap"] - m["compute_ms"] - m["wire_ms_total"]) < 1e-6
    assert m["predicted_ms"] <= m["predicted_ms_no_overlap"] and m["query_hash_and_binning_ms_under_the_exchange"] > 0
    assert "pass1_query_begin" in line["call_ms_rank0_per_step"]
    assert {"pass1_insert", "combine_export", "pass1_query"} <= set(line["call_ms_rank0_per_step"])
    assert line["e2e_failed"] is False, line["e2e"]
    cx = line["cxx_host"]
    assert cx["rounds_ms"] > 0 and {"insert (local)", "insert export", "query (local)"} <= set(cx["sharded_first_pass_ms_rank0"])
    assert cx["combined_exchange_rank0"]["bytes_received_rank0"] > 0


def test_bench_address_path_full_size_over_rccl():
    """The address decomposition at the bench's full size (62 x 5 Mbp, f=36) with one RCCL rank: the 8.6 GB and 19 GB exchange
    buffers cross `_Comm` in 256 MiB messages (a single multi-GiB all_to_all_single arrived truncated here), and the counters
    equal the real reference's (tests/golden m2_full)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    case = CASES["m2_full"]
    env = dict(os.environ, TPC_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", TPC_MULTIGPU="entries")
    env.pop("TPC_DIST_BACKEND", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--decomposition", "address",
                          "--no-cpu-baseline", "--e2e-runs", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["backend"] == "rccl" and line["config"]["decomposition"] == "address"
    r = case["rounds"][0]
    assert line["result"] == {"candidate_marks": r["marks"], "junctions": case["distinct"], "junction_occurrences": case["true_marks"]}
