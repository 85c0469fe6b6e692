"""Development aid: one golden case through the partitioned first pass with a forced slice size, against the oracle
(what tests/test_gpu_parity.py::test_partitioned_query_matches_oracle does, outside pytest so that stderr is visible).
   python tools/dev_case.py rand6_k9_q8 9"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import golden_cases, case_files
from oracle import oracle as O
from twopaco_amd import capi

name, slice_bits = sys.argv[1], int(sys.argv[2])
case = [c for c in golden_cases() if c["name"] == name][0]
tmp = tempfile.mkdtemp()
o = O.Oracle(case["k"], case["L"], case["q"], O.seed_table(case["seed"], case["q"], case["L"]))
for f in case_files(case, tmp):
    o.add_fasta(f)
text = capi.PackedText.from_fasta(case_files(case, tmp))
ctx = capi.Context(0)
ctx.set_option("insert_mode", 2)
ctx.set_option("query_mode", 2)
ctx.set_option("slice_bits", slice_bits)
for a in sys.argv[3:]:
    k, v = a.split("=")
    ctx.set_option(k, int(v))
ctx.set_params(case["k"], case["L"], case["q"], capi.seed_table(case["q"], case["L"], seed=case["seed"]))
ctx.seq_upload(text)
ranges = [(0, 1 << case["L"])] + [(r["low"], r["high"]) for r in case["rounds"] if case["n_rounds"] > 1]
for lo, hi in ranges:
    o.fill_only(lo, hi)
    marks = o.check_only(lo, hi)
    ctx.filter_reset()
    ctx.pass1_insert(lo, hi)
    print("insert ok: path", ctx.stat("insert_path"), "fmt", ctx.stat("insert_entry_fmt"), flush=True)
    got = ctx.pass1_query(lo, hi)
    print("query: path", ctx.stat("query_path"), "fmt", ctx.stat("query_entry_fmt"), "marks", got, "oracle", marks, "overflow", ctx.stat("query_overflow_entries"), flush=True)
    m = ctx.mask_download(False)
    bad = np.nonzero(m != o.round_mask)[0]
    print("mask words differing:", len(bad), bad[:10], flush=True)
    f = ctx.filter_download()
    print("filter equal:", bool((f == o.filter).all()), flush=True)
ctx.close()
