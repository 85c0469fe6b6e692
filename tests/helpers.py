"""Shared helpers of the test-suite."""
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


# full-size synthetic workloads: minutes of work and GiBs of filter each -- dedicated tests, not the parametrised sweeps
BIG = ("m1_full", "m2_full", "m2_s05_f38", "m3_f38", "m2_x15", "m2r_full", "m2r2_full")


def golden_cases():
    with open(os.path.join(GOLDEN, "cases.json")) as f:
        return json.load(f)


def case_files(case, tmpdir):
    """FASTA paths of a golden case (synthetic workloads are regenerated into tmpdir)."""
    if case.get("fasta"):
        return [os.path.join(GOLDEN, case["fasta"])]
    from twopaco_amd import synth
    s = case["synth"]
    recs, p = synth.workload(s["workload"], seed=s["seed"], scale=s["scale"])
    return synth.fasta_files(recs, p, tmpdir, prefix=case["name"] + "_")  # one record per file, or a genome's contigs per file (m2r)


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def parse_log(log):
    import re
    rounds = []
    for m in re.finditer(r"Round (\d+), (\d+):(\d+)", log):
        rounds.append({"low": int(m.group(2)), "high": int(m.group(3))})
    for key, pat in [("true", r"True junctions count = (\d+)"), ("false", r"False junctions count = (\d+)"),
                     ("table", r"Hash table size = (\d+)"), ("marks", r"Candidate marks count = (\d+)")]:
        for i, m in enumerate(re.finditer(pat, log)):
            rounds[i][key] = int(m.group(1))
    tm = re.search(r"True marks count: (\d+)", log)
    return {"rounds": rounds, "true_marks": int(tm.group(1)) if tm else None}


def text_codes(bases, nmask, length):
    g = np.arange(length, dtype=np.uint64)
    codes = ((bases[g >> np.uint64(5)] >> (np.uint64(2) * (g & np.uint64(31)))) & np.uint64(3)).astype(np.uint8)
    isn = ((nmask[g >> np.uint64(5)] >> (g & np.uint64(31)).astype(np.uint32)) & np.uint32(1)).astype(bool)
    codes[isn] = 4
    return codes
