"""GPU (-m gpu): the address-sharded filter (include/twopaco_hip.h tpc_shard_*, twopaco_amd/dist.py
AddressSharded) on 2 and 4 ranks sharing GPU 0 over a gloo rendezvous.  Checked against the oracle:
the shards reassemble into the oracle's Bloom filter bit for bit, every rank ends the query with the
oracle's candidate mask, and the final (position, id) list is the single-process one."""
import pickle

import numpy as np
import pytest
import torch.multiprocessing as mp

from helpers import case_files, golden_cases
from oracle import oracle as O
from test_dist_cpu import free_port

pytestmark = pytest.mark.gpu
CASES = {c["name"]: c for c in golden_cases()}


def assemble(shards, geom, L, world):
    """Filter words of the whole filter from the per-rank shards: shard layout is [local bucket][b2][slice],
    rank r owns the level-1 buckets b1 = bl * world + r of the PERMUTED slice index."""
    sb, b1, b2 = geom["slice_bits"], geom["b1"], geom["b2"] + geom.get("b3", 0)  # the levels below the first are local to the owner
    F = b1 + b2
    words = 1 << (sb - 5)
    full = np.zeros(((1 << L) >> 5) + 1, dtype=np.uint32)
    for r in range(world):
        sh = shards[r][:-1].reshape(-1, words)
        assert sh.shape[0] == (1 << F) // world
        ls = np.arange(sh.shape[0], dtype=np.uint64)
        permuted = (((ls >> np.uint64(b2)) * np.uint64(world) + np.uint64(r)) << np.uint64(b2)) | (ls & np.uint64((1 << b2) - 1))
        s = (permuted * np.uint64(geom["perm_inv"])) & np.uint64((1 << F) - 1)
        full[:-1].reshape(-1, words)[s.astype(np.int64)] = sh
    return full


def run(spec, world, tmp_path):
    from dist_worker import addr_worker
    res = str(tmp_path / "res.pkl")
    mp.spawn(addr_worker, args=(world, free_port(), spec, res), nprocs=world, join=True)
    with open(res, "rb") as f:
        return pickle.load(f)


def check(spec, o, gathered, world):
    for i, (lo, hi) in enumerate(spec["ranges"]):
        o.fill_only(lo, hi)
        marks = o.check_only(lo, hi)
        rounds = [g["rounds"][i] for g in gathered]
        full = assemble([r["shard"] for r in rounds], rounds[0]["geom"], spec["L"], world)
        assert (full == o.filter).all(), (lo, hi)
        for r in rounds:
            assert (r["mask"] == o.round_mask).all()
            assert int(np.unpackbits(r["mask"].view(np.uint8)).sum()) == marks
    o.enumerate(rounds=1, abundance=spec["abundance"])
    seq, pos, ids = o.records
    start = np.asarray(o.rec_start, dtype=np.int64)
    want = sorted((int(start[s] + p), int(i)) for s, p, i in zip(seq.tolist(), pos.tolist(), ids.tolist()) if abs(i) <= len(o.keys))
    if spec.get("sharded_pass2"):
        # the exact filter's table is sharded by key hash: every rank knows all junction keys but reports the ids of ITS marked
        # positions only; together they are the result, the true-junction counts add up, each mark was found by one rank
        got = []
        for g in gathered:
            assert g["junctions"] == len(o.keys)
            keep = g["ids"] != (1 << 63) - 1
            got += list(zip(g["g"][keep].tolist(), g["ids"][keep].tolist()))
        assert sorted(got) == want and len(set(p for p, _ in got)) == len(got)
        assert sum(g["true"] for g in gathered) == len(o.keys)
        return
    for g in gathered:  # every rank holds the complete result
        assert g["junctions"] == len(o.keys)
        keep = g["ids"] != (1 << 63) - 1
        got = sorted(zip(g["g"][keep].tolist(), g["ids"][keep].tolist()))
        assert got == want


@pytest.mark.parametrize("name,slice_bits,world", [("rand6_k9_fp", 8, 2), ("rand6_k9_fp", 8, 4), ("rand6_k9_q8", 9, 2), ("rand6_k9_q1", 12, 2),
                                                   ("rand6_k25_q3", 12, 4), ("c2_k51_r2", 16, 2), ("edge_k5", 7, 2), ("rand6_k9_a3", 8, 2),
                                                   ("c2_k125", 14, 4)])
def test_address_sharded_golden_cases(name, slice_bits, world, tmp_path):
    case = CASES[name]
    files = case_files(case, tmp_path)
    ranges = [(0, 1 << case["L"])] + [(r["low"], r["high"]) for r in case["rounds"] if case["n_rounds"] > 1]
    spec = {"files": files, "k": case["k"], "L": case["L"], "q": case["q"], "seed": case["seed"], "ranges": ranges,
            "abundance": case["abundance"] if case["abundance"] is not None else (1 << 64) - 1, "options": {"slice_bits": slice_bits}}
    o = O.Oracle(case["k"], case["L"], case["q"], O.seed_table(case["seed"], case["q"], case["L"]))
    for f in files:
        o.add_fasta(f)
    check(spec, o, run(spec, world, tmp_path), world)


@pytest.mark.parametrize("name,slice_bits,world", [("rand6_k9_fp", 8, 2), ("rand6_k25_q3", 12, 4), ("rand6_k9_L33", 20, 2), ("c2_k51_r2", 13, 4), ("c2_k125", 12, 2)])
def test_address_sharded_three_levels(name, slice_bits, world, tmp_path):
    """The three-level geometry (filters beyond 2^38 bits: f = 39 / 40, config 5) under sharding, forced on small filters:
    level 1 is exchanged, levels 2 and 3 run on the owner.  Shards reassemble to the oracle's filter, masks and ids as above."""
    case = CASES[name]
    files = case_files(case, tmp_path)
    ranges = [(0, 1 << case["L"])] + [(r["low"], r["high"]) for r in case["rounds"] if case["n_rounds"] > 1]
    spec = {"files": files, "k": case["k"], "L": case["L"], "q": case["q"], "seed": case["seed"], "ranges": ranges,
            "abundance": case["abundance"] if case["abundance"] is not None else (1 << 64) - 1,
            "options": {"slice_bits": slice_bits, "part_levels": 3}}
    o = O.Oracle(case["k"], case["L"], case["q"], O.seed_table(case["seed"], case["q"], case["L"]))
    for f in files:
        o.add_fasta(f)
    gathered = run(spec, world, tmp_path)
    assert gathered[0]["rounds"][0]["geom"]["b3"] > 0 and gathered[0]["rounds"][0]["qgeom"]["b3"] > 0
    check(spec, o, gathered, world)


@pytest.mark.parametrize("world,budget", [(2, 40 << 30), (4, 3 << 20)])
def test_address_sharded_synthetic_batches(world, budget, tmp_path):
    """8 x 50 kbp genomes, default slice size; the small budget cuts each pass into several batches."""
    from twopaco_amd import synth
    recs, _ = synth.workload("m1", scale=0.01)
    spec = {"workload": "m1", "scale": 0.01, "k": 25, "L": 26, "q": 5, "seed": 11, "ranges": [(0, 1 << 26)], "abundance": (1 << 64) - 1,
            "options": {"slice_bits": 14, "part_min_tiles": 1, "part_budget_bytes": budget}}
    o = O.Oracle(25, 26, 5, O.seed_table(11, 5, 26))
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for r in recs:
        o.add_record(letters[r].tobytes())
    gathered = run(spec, world, tmp_path)
    if budget < (1 << 30):
        assert gathered[0]["rounds"][0]["geom"]["batches"] > 1 and gathered[0]["rounds"][0]["qgeom"]["batches"] > 1
    # the survivor lists shrink from one hash function to the next
    tr = gathered[0]["rounds"][0]["survivors"][0]
    assert len(tr) == 3 and tr[2] <= tr[1] <= tr[0]
    check(spec, o, gathered, world)


def _synthetic(workload, scale, L, seed, options, **extra):
    from twopaco_amd import synth
    recs, _ = synth.workload(workload, scale=scale)
    spec = dict({"workload": workload, "scale": scale, "k": 25, "L": L, "q": 5, "seed": seed, "ranges": [(0, 1 << L)], "abundance": (1 << 64) - 1, "options": options}, **extra)
    o = O.Oracle(25, L, 5, O.seed_table(seed, 5, L))
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for r in recs:
        o.add_record(letters[r].tobytes())
    return spec, o


@pytest.mark.parametrize("world", [2, 4])
def test_overflowing_tight_regions_are_replanned_once(world, tmp_path):
    """A sharded filter has no direct-kernel fallback behind its overflow lists.  With the tight level-1 regions pinched
    to half their expected fill and a 64-entry list (option test_tight_pinch) both passes overflow on every rank; AddressSharded then
    goes back to the one-GPU region size ONCE, for the rest of the object's life, and starts the pass again: oracle's filter, masks, ids."""
    spec, o = _synthetic("m1", 0.01, 26, 11, {"slice_bits": 14, "part_min_tiles": 1, "test_tight_pinch": 50}, compact_exchange=False)
    gathered = run(spec, world, tmp_path)
    assert all(g["relaxed_regions"] == 1 for g in gathered)
    check(spec, o, gathered, world)


@pytest.mark.parametrize("world,compact", [(2, False), (4, True)])
def test_low_complexity_input_through_the_sharded_path(world, compact, tmp_path):
    """m2r at 1/50 scale -- repeat families, poly-A / poly-T / (CA)n / (GT)n tracts, two genomes on the other strand, contigs -- with the
    default tight regions, twice in one launch.  periodic = 0: every position probes; the tracts' entries overflow their regions
    (thousands of identical addresses per workgroup), travel as all-gathered lists and are applied by their owners.  periodic = 1 (the
    default): positions that repeat their neighbour's window send nothing (option shard_periodic_skip reaches the SHARDED hash kernels:
    the overflow lists must shrink -- ADVICE round 5 found the option inert at world > 1) and tpc_shard_periodic_copy gives them their
    twin's verdict after the last batch -- on ranks that hold only their window of the text as well.  Filter, masks and ids are the oracle's."""
    spec0, o = _synthetic("m2r", 0.02, 26, 7, {"slice_bits": 14, "part_min_tiles": 1}, compact_exchange=compact, periodic=0)
    spec1 = dict(spec0, periodic=1, sharded_pass2="records" if world == 4 else False, text_window=world == 4)
    g0, g1 = run([spec0, spec1], world, tmp_path)
    ovf0, ovf1 = sum(g["overflow_entries"] for g in g0), sum(g["overflow_entries"] for g in g1)
    assert ovf0 > 0 and ovf1 * 2 < ovf0, (ovf0, ovf1)
    assert all(g["periodic_skip"] == 0 for g in g0) and all(g["periodic_skip"] == 1 for g in g1)
    check(spec0, o, g0, world)
    check(spec1, o, g1, world)


@pytest.mark.parametrize("name,slice_bits,world", [("rand6_k25_q3", 12, 4), ("c2_k51_r2", 16, 2), ("rand6_k9_a3", 8, 4), ("edge_k5", 7, 2), ("example_k11", 8, 4)])
def test_address_sharded_key_sharded_pass2(name, slice_bits, world, tmp_path):
    """Second pass with the exact filter's table sharded by key hash (tpc_pass2_mark_owners / tpc_pass2_filter_positions): no
    mask union, 8 bytes per marked position to the key's owner, all-gather of the junction keys.  One- to five-word keys, an
    abundance cut (the counts must be complete on the owner), sequence ends and N runs."""
    case = CASES[name]
    files = case_files(case, tmp_path)
    spec = {"files": files, "k": case["k"], "L": case["L"], "q": case["q"], "seed": case["seed"], "ranges": [(0, 1 << case["L"])],
            "abundance": case["abundance"] if case["abundance"] is not None else (1 << 64) - 1, "sharded_pass2": True,
            "options": {"slice_bits": slice_bits}}
    o = O.Oracle(case["k"], case["L"], case["q"], O.seed_table(case["seed"], case["q"], case["L"]))
    for f in files:
        o.add_fasta(f)
    gathered = run(spec, world, tmp_path)
    check(spec, o, gathered, world)


@pytest.mark.parametrize("name,slice_bits,world", [("rand6_k9_fp", 8, 2), ("rand6_k25_q3", 12, 4), ("c2_k125", 14, 2), ("edge_k5", 7, 4)])
def test_address_sharded_text_free_pass2(name, slice_bits, world, tmp_path):
    """The whole enumeration with the text sharded as well: every rank uploads only its chunk of the packed text (option
    text_window), (key, prev | next) records travel to the key owners (tpc_pass2_mark_records / tpc_pass2_filter_records), the
    junction keys are all-gathered and each rank looks up the ids of its own positions."""
    case = CASES[name]
    files = case_files(case, tmp_path)
    spec = {"files": files, "k": case["k"], "L": case["L"], "q": case["q"], "seed": case["seed"], "ranges": [(0, 1 << case["L"])],
            "abundance": case["abundance"] if case["abundance"] is not None else (1 << 64) - 1, "sharded_pass2": "records", "text_window": True,
            "options": {"slice_bits": slice_bits}}
    o = O.Oracle(case["k"], case["L"], case["q"], O.seed_table(case["seed"], case["q"], case["L"]))
    for f in files:
        o.add_fasta(f)
    gathered = run(spec, world, tmp_path)
    check(spec, o, gathered, world)


def test_text_free_pass2_synthetic(tmp_path):
    """8 x 100 kbp genomes on 4 ranks, text windows + records."""
    from twopaco_amd import synth
    recs, _ = synth.workload("m1", scale=0.02)
    spec = {"workload": "m1", "scale": 0.02, "k": 25, "L": 28, "q": 5, "seed": 12, "ranges": [(0, 1 << 28)], "abundance": (1 << 64) - 1,
            "sharded_pass2": "records", "text_window": True, "options": {"slice_bits": 14}}
    o = O.Oracle(25, 28, 5, O.seed_table(12, 5, 28))
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for r in recs:
        o.add_record(letters[r].tobytes())
    gathered = run(spec, 4, tmp_path)
    check(spec, o, gathered, 4)


def test_key_sharded_pass2_synthetic(tmp_path):
    """8 x 100 kbp genomes, 4 ranks: many occurrences per junction spread over all ranks."""
    from twopaco_amd import synth
    recs, _ = synth.workload("m1", scale=0.02)
    spec = {"workload": "m1", "scale": 0.02, "k": 25, "L": 28, "q": 5, "seed": 12, "ranges": [(0, 1 << 28)], "abundance": (1 << 64) - 1,
            "sharded_pass2": True, "options": {"slice_bits": 14}}
    o = O.Oracle(25, 28, 5, O.seed_table(12, 5, 28))
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for r in recs:
        o.add_record(letters[r].tobytes())
    gathered = run(spec, 4, tmp_path)
    check(spec, o, gathered, 4)
    assert sum(g["step_marks"] for g in gathered) == int(np.unpackbits(gathered[0]["rounds"][0]["mask"].view(np.uint8)).sum())


def test_exchange_forms_agree_and_tight_regions_move_fewer_bytes(tmp_path):
    """The three forms of the level-1 exchange give the same filter, masks and ids: packed to the exact sizes (tpc_shard_pack /
    tpc_shard_apply_packed), equal blocks of tightly sized regions with the own block read in place (tpc_shard_apply_inplace; the
    default below eight ranks), and the same with the one-GPU slack of 1.3 x (option shard_tight_regions = 0).  Bytes that leave a
    rank: packed <= tight < 0.9 x loose."""
    from twopaco_amd import synth
    recs, _ = synth.workload("m1", scale=0.02)
    base = {"workload": "m1", "scale": 0.02, "k": 25, "L": 28, "q": 5, "seed": 12, "ranges": [(0, 1 << 28)], "abundance": (1 << 64) - 1,
            "options": {"slice_bits": 14}}
    specs = [dict(base, compact_exchange=True), dict(base, compact_exchange=False),
             dict(base, compact_exchange=False, options={"slice_bits": 14, "shard_tight_regions": 0})]
    o = O.Oracle(25, 28, 5, O.seed_table(12, 5, 28))
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for r in recs:
        o.add_record(letters[r].tobytes())
    results = run(specs, 2, tmp_path)
    for sp, gathered in zip(specs, results):
        check(sp, o, gathered, 2)
    packed, tight, loose = (results[i][0]["region_bytes_sent"] for i in range(3))
    assert 0 < packed <= tight < 0.9 * loose, (packed, tight, loose)


@pytest.mark.parametrize("world", [2, 4])
def test_address_sharded_randomized(world, tmp_path):
    """Random texts (N runs, skew) x random k, L, q, slice size, tile batches and gated ranges, all in one process group."""
    rng = np.random.default_rng(99 + world)
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
    specs = []
    for trial in range(10):
        k = int(rng.choice([5, 9, 15, 25, 31, 33]))
        L = int(rng.integers(14, 24))
        q = int(rng.integers(1, 8))
        slice_bits = int(rng.integers(6, min(12, L - 6) + 1))
        base = alphabet[rng.integers(0, 4, int(rng.integers(3000, 50000)))].copy()
        recs = []
        for r in range(int(rng.integers(1, 5))):
            s = base.copy()
            hits = rng.random(s.size) < 0.02
            s[hits] = alphabet[rng.integers(0, 4, int(hits.sum()))]
            if rng.random() < 0.5:
                a = int(rng.integers(0, s.size)); s[a:a + int(rng.integers(1, 60))] = ord("N")
            if rng.random() < 0.2:
                s[:int(rng.integers(1, s.size // 4))] = ord("A")
            recs.append(s.tobytes())
        size = 1 << L
        cut = sorted(int(x) for x in rng.integers(0, size, 2))
        specs.append({"records": recs, "k": k, "L": L, "q": q, "seed": int(rng.integers(1, 1 << 40)), "ranges": [(0, size), (cut[0], cut[1])],
                      "abundance": (1 << 64) - 1, "compact_exchange": trial % 3 != 2,  # both the exact-size and the equal-block exchange
                      "fused_verify": trial % 2 == 0,  # the fused verification calls and the step-by-step ones
                      "options": {"slice_bits": slice_bits, "part_min_tiles": 1, "part_budget_bytes": int(rng.choice([40 << 30, 1 << 20]))}})
    results = run(specs, world, tmp_path)
    for sp, gathered in zip(specs, results):
        o = O.Oracle(sp["k"], sp["L"], sp["q"], O.seed_table(sp["seed"], sp["q"], sp["L"]))
        for r in sp["records"]:
            o.add_record(r)
        check(sp, o, gathered, world)
        o.close()
