"""GPU (-m gpu): the combined multi-GPU exchange -- the Bloom filter REPLICATED through set-bit lists (include/twopaco_hip.h
tpc_combine_*, csrc/tpc_combine.hip, twopaco_amd/dist.py:Combined) -- on 2, 4 and 8 ranks sharing GPU 0 over a gloo rendezvous.
What the reference's threads get from one shared ConcurrentBitVector (concurrentbitvector.cpp:31-45, MergeOr :115-122) every rank
must end up with here: after a round's query EVERY rank holds the oracle's whole filter bit for bit, the round masks of the ranks
together are the oracle's candidate mask (each rank marks only the positions it hashed), and the final (position, id) list is the
single-process one.  Every form of the exchange is forced in turn: all-gather of the exports, reduce-scatter + merge + all-gather,
dense OR all-reduce."""
import pickle

import numpy as np
import pytest
import torch.multiprocessing as mp

from helpers import case_files, golden_cases
from oracle import oracle as O
from test_dist_cpu import free_port

pytestmark = pytest.mark.gpu
CASES = {c["name"]: c for c in golden_cases()}


def run(spec, world, tmp_path):
    from dist_worker import combined_worker
    res = str(tmp_path / "res.pkl")
    mp.spawn(combined_worker, args=(world, free_port(), spec, res), nprocs=world, join=True)
    with open(res, "rb") as f:
        return pickle.load(f)


def check(spec, o, gathered, world, expect_mode=None):
    for i, (lo, hi) in enumerate(spec["ranges"]):
        o.fill_only(lo, hi)
        marks = o.check_only(lo, hi)
        rounds = [g["rounds"][i] for g in gathered]
        union = np.zeros_like(rounds[0]["mask"])
        for r in rounds:
            assert (r["filter"] == o.filter).all(), (lo, hi)  # the WHOLE filter on every rank
            if r["peek"] is not None:
                assert (r["peek"] == o.filter).all(), (lo, hi)
            if expect_mode:
                assert expect_mode in r["combine"]["mode"], r["combine"]
            union |= r["mask"]
        if spec.get("sharded_pass2"):
            assert sum(int(np.unpackbits(r["mask"].view(np.uint8)).sum()) for r in rounds) == marks  # every mark found by exactly one rank
        else:
            for r in rounds:
                assert (r["mask"] == o.round_mask).all()
        assert (union == o.round_mask).all()
        assert int(np.unpackbits(union.view(np.uint8)).sum()) == marks
    o.enumerate(rounds=1, abundance=spec["abundance"])
    seq, pos, ids = o.records
    start = np.asarray(o.rec_start, dtype=np.int64)
    want = sorted((int(start[s] + p), int(i)) for s, p, i in zip(seq.tolist(), pos.tolist(), ids.tolist()) if abs(i) <= len(o.keys))
    if spec.get("sharded_pass2"):
        got = []
        for g in gathered:
            assert g["junctions"] == len(o.keys)
            keep = g["ids"] != (1 << 63) - 1
            got += list(zip(g["g"][keep].tolist(), g["ids"][keep].tolist()))
        assert sorted(got) == want and len(set(p for p, _ in got)) == len(got)
        assert sum(g["true"] for g in gathered) == len(o.keys)
        return
    for g in gathered:
        assert g["junctions"] == len(o.keys)
        keep = g["ids"] != (1 << 63) - 1
        assert sorted(zip(g["g"][keep].tolist(), g["ids"][keep].tolist())) == want


def golden_spec(name, slice_bits, tmp_path, **extra):
    case = CASES[name]
    files = case_files(case, tmp_path)
    ranges = [(0, 1 << case["L"])] + [(r["low"], r["high"]) for r in case["rounds"] if case["n_rounds"] > 1]
    spec = dict({"files": files, "k": case["k"], "L": case["L"], "q": case["q"], "seed": case["seed"], "ranges": ranges,
                 "abundance": case["abundance"] if case["abundance"] is not None else (1 << 64) - 1, "options": {"slice_bits": slice_bits}}, **extra)
    o = O.Oracle(case["k"], case["L"], case["q"], O.seed_table(case["seed"], case["q"], case["L"]))
    for f in files:
        o.add_fasta(f)
    return spec, o


MODES = {"gather": "all-gather of the exports", "scatter": "reduce-scatter", "dense": "dense"}


@pytest.mark.parametrize("name,slice_bits,world,mode", [("rand6_k9_fp", 8, 2, "gather"), ("rand6_k9_fp", 8, 4, "scatter"), ("rand6_k9_q8", 9, 2, "scatter"),
                                                        ("rand6_k9_q1", 8, 2, "gather"), ("rand6_k25_q3", 12, 4, "gather"), ("c2_k51_r2", 16, 2, "scatter"),
                                                        ("edge_k5", 7, 2, "scatter"), ("rand6_k9_a3", 8, 2, "dense"), ("c2_k125", 14, 4, "scatter"),
                                                        ("c2_k51_r2", 13, 8, "scatter"), ("example_k11", 8, 4, "gather"), ("rand6_k9_fp", 8, 4, "dense")])
def test_combined_golden_cases(name, slice_bits, world, mode, tmp_path):
    """Golden inputs of the real reference (N runs, IUPAC, sequence ends, multi-word keys, several rounds with their exact ranges, an
    abundance cut): filter, masks and ids through every form of the exchange."""
    spec, o = golden_spec(name, slice_bits, tmp_path, mode=mode)
    check(spec, o, run(spec, world, tmp_path), world, MODES[mode])


@pytest.mark.parametrize("name,slice_bits,world,mode", [("rand6_k9_fp", 8, 2, "gather"), ("rand6_k25_q3", 12, 4, "scatter")])
def test_combined_lists_applied_without_a_lookup(name, slice_bits, world, mode, tmp_path):
    """The filter is read between insert and query: the imported lists are applied by k_slice_combine alone (no lookup riding
    along) and the query then probes the dense filter."""
    spec, o = golden_spec(name, slice_bits, tmp_path, mode=mode, peek=True)
    check(spec, o, run(spec, world, tmp_path), world, MODES[mode])


@pytest.mark.parametrize("name,slice_bits,world,aggregate", [("rand6_k9_fp", 8, 2, True), ("c2_k51_r2", 16, 2, True), ("rand6_k9_a3", 8, 4, True), ("edge_k5", 7, 4, True),
                                                             ("c2_k125", 14, 2, True), ("rand6_k9_a3", 8, 2, False), ("c2_k51_r2", 16, 4, False)])
def test_combined_text_free_pass2(name, slice_bits, world, aggregate, tmp_path, monkeypatch):
    """The whole enumeration with the text sharded as well (option text_window): the marks stay on the rank that hashed them, the exact
    filter's table is sharded by key hash, every rank looks up the ids of its own positions.  What travels to the key owners: one record
    per DISTINCT key of a rank's marks with the letter sets and count they add up to (tpc_pass2_aggregate_records: the rank's own exact
    filter first -- an abundance cut, two-word and four-word keys, N runs), or, aggregate = False, a (key, prev | next) record per mark."""
    if not aggregate:
        monkeypatch.setenv("TPC_PASS2_AGGREGATE", "0")
    case = CASES[name]
    spec, o = golden_spec(name, slice_bits, tmp_path, sharded_pass2="records", text_window=True)
    spec["ranges"] = [(0, 1 << case["L"])]
    check(spec, o, run(spec, world, tmp_path), world)


def _synthetic(workload, scale, L, seed, options, **extra):
    from twopaco_amd import synth
    recs, _ = synth.workload(workload, scale=scale)
    spec = dict({"workload": workload, "scale": scale, "k": 25, "L": L, "q": 5, "seed": seed, "ranges": [(0, 1 << L)], "abundance": (1 << 64) - 1, "options": options}, **extra)
    o = O.Oracle(25, L, 5, O.seed_table(seed, 5, L))
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for r in recs:
        o.add_record(letters[r].tobytes())
    return spec, o


@pytest.mark.parametrize("world,slice_bits,mode", [(2, 14, None), (4, 18, None), (8, 20, None), (4, 20, "gather"), (2, 18, "scatter")])
def test_combined_synthetic(world, slice_bits, mode, tmp_path):
    """8 x 50 kbp genomes with 1 % SNPs, filter of 2^26 bits: slices of one, four and sixteen 2^16-bit windows; the automatic choice
    of the bytes model (all-gather at two ranks, reduce-scatter beyond) and both forms forced."""
    spec, o = _synthetic("m1", 0.01, 26, 11, {"slice_bits": slice_bits, "part_min_tiles": 1}, mode=mode, sharded_pass2="records", text_window=True)
    gathered = run(spec, world, tmp_path)
    want = MODES[mode] if mode else (MODES["gather"] if world == 2 else MODES["scatter"])
    check(spec, o, gathered, world, want)
    c = gathered[0]["rounds"][0]["combine"]
    assert gathered[0]["rounds"][0]["fused"] >= 1  # the lookup built the slices from the lists
    # what travelled is far below 4 bytes per insert address: 2 bytes per DISTINCT bit of a chunk
    n_addr = 5 * 8 * 50000
    assert 0 < c["export_bytes"] * world < 3 * n_addr, c
    # second pass: a rank sends every distinct key of its marks once (eight genomes: a key is marked up to eight times)
    for g in gathered:
        assert g["pass2_records_sent"] <= g["step_marks"]
    if world <= 2:  # (at eight ranks a rank holds ONE genome: every key of its marks is already distinct)
        assert sum(g["pass2_records_sent"] for g in gathered) < 0.7 * sum(g["step_marks"] for g in gathered)


@pytest.mark.parametrize("world,budget", [(2, 3 << 20), (4, 3 << 20)])
def test_combined_several_batches_take_the_dense_road(world, budget, tmp_path):
    """A buffer budget that cuts a rank's chunk into several tile batches: the insert cannot stay in its level-2 regions, every rank
    applies it to its dense filter and the filters are OR-reduced by word ranges; the query runs in several batches against the
    complete filter."""
    spec, o = _synthetic("m1", 0.01, 26, 11, {"slice_bits": 14, "part_min_tiles": 1, "part_budget_bytes": budget})
    gathered = run(spec, world, tmp_path)
    assert gathered[0]["rounds"][0]["insert_batches"] > 1 and gathered[0]["rounds"][0]["query_batches"] > 1
    check(spec, o, gathered, world, "dense")


def test_combined_three_levels_take_the_dense_road(tmp_path):
    """The three-level geometry (f = 39 / 40), forced on a small filter: no deferred apply, so the dense OR all-reduce."""
    spec, o = golden_spec("rand6_k25_q3", 12, tmp_path)
    spec["options"]["part_levels"] = 3
    check(spec, o, run(spec, 2, tmp_path), 2, "dense")


@pytest.mark.parametrize("world,mode", [(2, None), (4, None)])
def test_combined_low_complexity_input(world, mode, tmp_path):
    """m2r at 1/50 scale -- repeat families, poly-A / poly-T / (CA)n / (GT)n tracts, two genomes on the other strand, contigs: the one-GPU
    passes skip periodic windows on their chunk and copy the verdicts (on ranks that hold only their window of the text too)."""
    spec, o = _synthetic("m2r", 0.02, 26, 7, {"slice_bits": 14, "part_min_tiles": 1}, mode=mode, sharded_pass2="records", text_window=True)
    check(spec, o, run(spec, world, tmp_path), world)


@pytest.mark.parametrize("name,world", [("rand6_k9_q20", 2), ("rand6_k9_q20_fp_r2", 4)])
def test_combined_more_than_16_hash_functions(name, world, tmp_path):
    """-q 20 (the reference takes any -q, constructor.cpp:83-90): beyond the rolling kernels' 16 functions every rank runs the closed-form
    kernels (csrc/tpc_pass1_anyq.hip) over ITS chunk of the positions, the ranks' dense filters are OR-reduced, the query is local --
    no one-GPU fallback."""
    spec, o = golden_spec(name, 12, tmp_path)
    check(spec, o, run(spec, world, tmp_path), world, "dense")


def test_combined_more_ranks_than_tiles(tmp_path):
    """example.fa is one tile: seven of eight ranks hash nothing and still build the whole filter from the one export."""
    spec, o = golden_spec("example_k15_dbg", 8, tmp_path, mode="scatter")
    check(spec, o, run(spec, 8, tmp_path), 8, "reduce-scatter")
