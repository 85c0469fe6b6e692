"""CPU: the oracle (oracle/twopaco_oracle.c) against the golden vectors produced by the REAL
reference binary (tests/golden/make_golden.py: pinned /dev/urandom, -t 1).  Byte-exact
de_bruijn.bin, the reference's own log counters and round ranges."""
import os

import numpy as np
import pytest

from helpers import BIG, GOLDEN, case_files, golden_cases, sha256_file
from oracle import oracle as O

CASES = golden_cases()
FAST = [c for c in CASES if c["name"] not in BIG]


@pytest.mark.parametrize("case", FAST, ids=[c["name"] for c in FAST])
def test_oracle_matches_reference(case, tmp_path):
    table = O.seed_table(case["seed"], case["q"], case["L"])
    o = O.Oracle(case["k"], case["L"], case["q"], table)
    for f in case_files(case, tmp_path):
        o.add_fasta(f)
    o.enumerate(rounds=case["n_rounds"], abundance=case["abundance"] if case["abundance"] is not None else (1 << 64) - 1)
    out = str(tmp_path / "orc.bin")
    o.write_bin(out)
    assert os.path.getsize(out) == case["bin_bytes"]
    assert sha256_file(out) == case["bin_sha256"]
    if case.get("bin"):
        assert open(out, "rb").read() == open(os.path.join(GOLDEN, case["bin"]), "rb").read()
    for i, r in enumerate(case["rounds"]):
        st = o.round_stats(i)
        assert (st["low"], st["high"]) == (r["low"], r["high"])
        assert (st["true"], st["false"], st["table"], st["marks"]) == (r["true"], r["false"], r["table"], r["marks"])
    assert o.true_marks == case["true_marks"]
    assert len(o.keys) == case["distinct"]
    o.close()


def test_bin_reader_roundtrip():
    recs = O.read_bin(os.path.join(GOLDEN, "example_k11.bin"))
    # the reference's shipped example/example.seq pins the positions for k=11 (ids are seed dependent)
    assert [(s, p) for s, p, _ in recs] == [(0, 0), (0, 129), (0, 140), (0, 269), (0, 280), (0, 350), (0, 409), (1, 0),
                                            (1, 129), (1, 140), (1, 269), (1, 280), (1, 350), (1, 409), (1, 420), (1, 479)]


def test_naive_oracle_agrees_on_positions():
    """Seed-free check: positions of the golden .bin == FindJunctionsNaively (reference test.cpp:71-160)."""
    for name, fa, k in [("rand6_k9_fp", "rand6.fa", 9), ("edge_k5", "edge.fa", 5), ("rand6_k3", "rand6.fa", 3)]:
        chrs = O.read_fasta_records(os.path.join(GOLDEN, fa))
        _, marks = O.naive_junction_marks(chrs, k)
        got = [np.zeros(len(c), dtype=bool) for c in chrs]
        for s, p, _ in O.read_bin(os.path.join(GOLDEN, name + ".bin")):
            got[s][p] = True
        for i, c in enumerate(chrs):
            if len(c) < k:
                assert not got[i].any()
                continue
            assert (got[i] == marks[i]).all(), (name, i)
