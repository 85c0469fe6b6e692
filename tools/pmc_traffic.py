#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 --pmc runs (FETCH_SIZE and WRITE_SIZE, collected in separate
passes as /opt/skills/guides/MI355X_MICROARCH.md prescribes):
   python tools/pmc_traffic.py gpurun_out/prof_X_f/f_results.db gpurun_out/prof_X_w/w_results.db > profiles/X_pmc_traffic.json
Counters are KB per dispatch.  gfx950 correction from the guide: FETCH_SIZE reports half of a wide coalesced read
stream, so the streaming kernels get 2 x raw; kernels whose reads are scattered 64-byte probes are left as counted."""
import hashlib
import json
import os
import re
import sqlite3
import sys


def csrc_signature():
    """The same signature bench.py computes: byte counts are only valid for the kernel sources they were measured on."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    d = os.path.join(root, "twopaco_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]

SCATTERED = ("k_q_verify", "k_emit", "k_filter2", "k_insert", "k_query", "k_part_ovf", "k_q_ovf", "k_v_probe")


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, value in c.execute("select name, counter_value from pmc_events where counter_name = ?", (counter,)):
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*", "", short).replace("void ", "")
        if "rocprim" in short or short.startswith("__amd"):
            continue
        tot, n = out.get(short, (0.0, 0))
        out[short] = (tot + value * 1024.0, n + 1)
    return {k: t / n for k, (t, n) in out.items()}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, 0) + write.get(k, 0))):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        fc = f if k.startswith(SCATTERED) else 2.0 * f
        kernels[k] = {"fetch_bytes_raw": f, "write_bytes": w, "fetch_bytes_corrected": fc, "hbm_bytes": fc + w}
    groups = {"insert": ("k_part_hash", "k_part_split", "k_part_apply", "k_part_ovf"), "query": ("k_q_hash", "k_q_split", "k_q_lookup", "k_q_verify", "k_q_ovf"),
              "fused": ("k_apply_lookup",)}
    out = {"csrc_signature": csrc_signature(),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0 on m2 (62x5 Mbp, k=25, q=5, f=36); "
                   "bytes per launch; fetch corrected x2 for streaming kernels (MI355X_MICROARCH.md), scattered-probe kernels as counted",
           "kernels": kernels,
           "groups": {g: sum(v["hbm_bytes"] for k, v in kernels.items() if k.startswith(names)) for g, names in groups.items()}}
    # deferred apply: k_apply_lookup does the insert's apply (its entries in, the filter out) and the query's lookup (its entries
    # in); its bytes are split in that proportion, as bench.py splits its time
    fk = [v for k, v in kernels.items() if k.startswith("k_apply_lookup")]
    if fk:
        w = sum(v["write_bytes"] for v in fk)                       # the filter (+ a few survivors)
        ins_entries = sum(v["write_bytes"] for k, v in kernels.items() if k.startswith("k_part_split"))   # = what the apply reads
        share = min(1.0, (w + ins_entries) / max(out["groups"]["fused"], 1.0))
        out["fused_share_insert"] = share
        out["groups"]["insert"] += share * out["groups"]["fused"]
        out["groups"]["query"] += (1.0 - share) * out["groups"]["fused"]
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
