#!/usr/bin/env python3
"""Golden vectors for the graphdump formats (twopaco_amd/host/junctiondump.cpp): runs the REAL reference
graphdump (oracle/_ref/graphdump_ref, built by `make -C oracle ref` from /root/reference -- only in the
build container) on the committed golden .bin files and records exit code, stderr and the sha256 / size of
stdout.  File names are passed relative to tests/golden so that the GFA1 UR:Z: tags do not embed a
container path.  Output: tests/golden/graphdump.json"""
import hashlib
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "..", "oracle", "_ref", "graphdump_ref")
FORMATS = ["seq", "group", "dot", "gfa1", "gfa2", "fasta"]


def record(case, args):
    r = subprocess.run([REF] + args, cwd=HERE, capture_output=True)
    return {"case": case, "args": args, "rc": r.returncode, "stderr": r.stderr.decode(), "stdout_sha256": hashlib.sha256(r.stdout).hexdigest(),
            "stdout_bytes": len(r.stdout), "stdout_head": r.stdout[:200].decode(errors="replace")}


def main():
    cases = json.load(open(os.path.join(HERE, "cases.json")))
    out = []
    for c in cases:
        if not c.get("bin") or not c.get("fasta") or not os.path.exists(os.path.join(HERE, c["bin"])):
            continue
        for fmt in FORMATS:
            for prefix in ([False, True] if fmt in ("gfa1", "gfa2") else [False]):
                out.append(record(c["name"], [c["bin"], "-f", fmt, "-k", str(c["k"]), "-s", c["fasta"]] + (["--prefix"] if prefix else [])))
    # command-line errors
    for args in (["example_k11.bin", "-f", "gfa1", "-k", "11"], ["example_k11.bin", "-f", "seq"], ["missing.bin", "-f", "seq", "-k", "11"],
                 ["example_k11.bin", "-f", "xml", "-k", "11"]):
        out.append(record("cli", args))
    json.dump(out, open(os.path.join(HERE, "graphdump.json"), "w"), indent=1)
    print(len(out), "vectors")


if __name__ == "__main__":
    main()
