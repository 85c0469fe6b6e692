"""CPU: bin/graphdump (twopaco_amd/host/junctiondump.cpp) against golden vectors of the REAL reference graphdump
(tests/golden/graphdump.json, made by tests/golden/make_graphdump_golden.py): all six formats over the
golden .bin files -- stdout byte for byte (sha256), exit code, error text -- and against the GFA / FASTA
files the reference ships in example/ for the structure that does not depend on the hash seed."""
import hashlib
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
VECTORS = json.load(open(os.path.join(GOLDEN, "graphdump.json")))


@pytest.fixture(scope="module")
def exe(built):
    path = os.path.join(os.path.dirname(HERE), "twopaco_amd", "bin", "graphdump")
    assert os.path.exists(path)
    return path


def run(exe, args):
    return subprocess.run([exe] + args, cwd=GOLDEN, capture_output=True, timeout=300)


@pytest.mark.parametrize("fmt", ["seq", "group", "dot", "gfa1", "gfa2", "fasta"])
def test_formats_equal_reference_bytes(exe, fmt):
    n = 0
    for v in VECTORS:
        if v["case"] == "cli" or v["args"][2] != fmt:
            continue
        r = run(exe, v["args"])
        assert r.returncode == v["rc"], (v["args"], r.stderr)
        assert r.stderr.decode() == v["stderr"], v["args"]  # "error: The input is corrupted" for skipped sequence ids
        if v["rc"] == 0:  # (what the reference prints before that error comes from out-of-bounds reads)
            assert len(r.stdout) == v["stdout_bytes"] and hashlib.sha256(r.stdout).hexdigest() == v["stdout_sha256"], v["args"]
        n += 1
    assert n >= 23


def test_command_line_errors(exe):
    for v in VECTORS:
        if v["case"] != "cli":
            continue
        r = run(exe, v["args"])
        assert r.returncode == v["rc"] == 1 and r.stdout == b""
        err = r.stderr.decode()
        if v["stderr"].startswith("PARSE ERROR"):
            # TCLAP prints the program path in its usage text: compare the diagnosis lines only
            assert err.split("\n")[:2] == v["stderr"].split("\n")[:2]
        else:
            assert err == v["stderr"]


def test_gfa_structure_is_consistent(exe):
    """Seed-free checks on a GFA1 dump: every occurrence / link names a declared segment, a path spells its
    sequence back (segments overlap by k), both strands of a segment give reverse-complementary bodies."""
    k = 11
    out = run(exe, ["example_k11.bin", "-f", "gfa1", "-k", str(k), "-s", "example.fa"]).stdout.decode().splitlines()
    seg = {}
    for line in out:
        f = line.split("\t")
        if f[0] == "S" and f[2] != "*":
            seg[f[1]] = f[2]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = lambda s: "".join(comp.get(c, "N") for c in reversed(s))
    seqs = {}
    name = None
    for line in open(os.path.join(GOLDEN, "example.fa")):
        if line.startswith(">"):
            name = line[1:].split()[0]
            seqs[name] = ""
        else:
            seqs[name] += line.strip().upper()
    paths = 0
    for line in out:
        f = line.split("\t")
        if f[0] in ("C", "L"):
            assert f[1] in seg and (f[0] == "C" or f[3] in seg)
        if f[0] == "P":
            spelled = ""
            for item in f[2].split(","):
                body = seg[item[:-1]] if item[-1] == "+" else rc(seg[item[:-1]])
                spelled = body if not spelled else spelled + body[k:]
            assert spelled == seqs[f[1]]
            paths += 1
    assert paths == len(seqs)
