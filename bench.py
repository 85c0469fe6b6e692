#!/usr/bin/env python3
"""bench.py -- junction-enumeration hot path on MI355X, BASELINE.json's metric
("k-mers hashed/sec + end-to-end junctions/sec, 62 E. coli k=25 f=36, 1/2/4/8 GPU").

A "step" is one full pass of the hot path over the synthetic workload, input already packed and
resident in HBM: Bloom filter reset, first-pass insert, first-pass query, candidate compaction,
second-pass exact filter, junction key sort + id index, output-pass id lookup (junction records
left in HBM).  value = vertex k-mers through the whole path per second, whole job.
The second half of the metric -- end-to-end junction occurrences per second, process start to file
close -- is measured at N = 1 by running the `twopaco` CLI as a child process on the same workload
written out as FASTA files (`e2e` in the JSON line).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload m2|m1] [--scale S]
                  [--decomposition ranges|address] [--cpu-baseline sample|full|none] [--e2e-runs R]

--gpus N > 1 without a launcher (no WORLD_SIZE in the environment) starts
`python -m torch.distributed.run --nproc-per-node N` on this file as a CHILD process (before anything
touches the GPU) and passes its output and exit code through; under a launcher every rank runs
twopaco_amd/dist.py:bench_main (one process per GPU, RCCL).  Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import re
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

G_BYTES = 64           # HBM access granule of a scattered 4-byte access (SURVEY 8d planning value)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s
GOLDEN_SEED = 20240229  # tests/golden/make_golden.py: the seed the reference goldens were made with
PMC_PROFILE = "r06_pmc_traffic.json"


def launch_ranks(args, script=None):
    """bench.py --gpus N, N > 1, no launcher: become the parent of torch.distributed.run (child process)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def csrc_signature():
    """sha256 over the kernel sources: PMC byte counts collected on other kernels must not be reported as this run's."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "twopaco_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def golden_case(workload, scale):
    name = {("m2", 1.0): "m2_full", ("m1", 1.0): "m1_full", ("m2r", 1.0): "m2r_full", ("m2r2", 1.0): "m2r2_full"}.get((workload, scale))
    path = os.path.join(ROOT, "tests", "golden", "cases.json")
    if not name or not os.path.exists(path):
        return None
    with open(path) as f:
        for c in json.load(f):
            if c["name"] == name:
                return c
    return None


def one_step(ctx, abundance=(1 << 64) - 1):
    ctx.run_begin()
    ctx.filter_reset()
    ctx.pass1_insert(count=False)
    marks = ctx.pass1_query()
    st = ctx.pass2_filter(abundance)
    J = ctx.junctions_finalize()
    n_marked, n_valid = ctx.emit()
    return marks, st, J, n_valid


def write_fasta_files(recs, tmp, n=None, p=None):
    """One FASTA file per genome: a record each, or the genome's contigs (p["files"], workload m2r); n: the first n genomes only."""
    from twopaco_amd import synth
    pp = dict(p or {})
    if n is not None:
        pp["files"] = (pp.get("files") or [(i, i + 1) for i in range(len(recs))])[:n]
    return synth.fasta_files(recs, pp, tmp)


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


BREAKDOWN_KEYS = [("exec_to_main_ms", "exec -> main"), ("parse_pack_fasta_ms", "parse + pack FASTA"),
                  ("hip_startup_context_ms_parallel_thread", "setup thread: HIP start-up + context"),
                  ("filter_allocation_ms_parallel_thread", "setup thread: parameters + filter allocation"),
                  ("code_objects_ms_parallel_thread", "warm-up thread: code objects"),
                  ("partition_buffers_ms_parallel_thread", "setup thread: partition buffers"),
                  ("context_upload_ms", "context + upload"), ("rounds_ms", "rounds (insert, query, exact filter)"),
                  ("insert_ms", "round: insert"), ("query_ms", "round: query"), ("exact_filter_ms", "round: exact filter"),
                  ("sharded_insert_query_ms", "round: sharded insert + query"),
                  ("sort_ids_stream_ms", "sort + id lookup + junction stream"), ("write_ms", "write junction stream"),
                  ("exec_to_output_complete_ms", "exec -> output complete"), ("exec_to_context_destroyed_ms", "exec -> context destroyed")]


def e2e_cli(files, p, golden, runs, tmp, gpus=1, settle_s=0.0):
    """End-to-end junction occurrences per second (SURVEY 8d metric 2; reference path constructor.cpp:161-176 -> VE ctor ->
    junctionapi.h:118-132): the `twopaco` CLI as a fresh child process, wall clock from process start to exit (output
    file closed), FASTA files in the page cache, output sha256 checked against the reference golden.  gpus > 1: the C++
    multi-GPU host (`twopaco --gpus N`: one rank thread per device, the Bloom filter sharded by bit address, RCCL between them,
    host/multigpu.cpp).  Reports the median run with its own phase timers, p50 / max, and the SLOWEST run's timers beside them
    (a tail run must show which phase ate it)."""
    exe = os.path.join(ROOT, "twopaco_amd", "bin", "twopaco")
    threads = str(min(64, os.cpu_count() or 1))
    walls, occ, sha_ok, phases = [], None, None, []
    env = dict(os.environ, TWOPACO_TIMING="1")  # the CLI's own phase timers on stderr ([timing] lines, milliseconds)
    for rep in range(runs):
        out = os.path.join(tmp, "e2e_%d.bin" % rep)  # a fresh file each time
        cmd = [exe, "-k", str(p["k"]), "-f", str(p["L"]), "-q", str(p["q"]), "-t", threads, "--seed", str(GOLDEN_SEED), "--tmpdir", tmp, "-o", out]
        if gpus > 1:
            cmd += ["--gpus", str(gpus)]
            if os.environ.get("TPC_E2E_EMULATE_RANKS"):  # tests: the N ranks of the C++ host on ONE device (loopback transport)
                cmd += ["--emulate-ranks"]
        cmd += files
        t0 = time.perf_counter()
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        wall = time.perf_counter() - t0
        if res.returncode != 0:
            return {"error": res.stderr.decode()[-400:]}
        walls.append(wall)
        err = res.stderr.decode()
        ph = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"\[timing\]\s+(.*): ([0-9.eE+-]+) ms", err)}
        # the C++ multi-GPU host's own phases of the sharded first pass (rank 0, summed over the rounds): "name ms;" pairs on one line
        sharded = {}
        for m in re.finditer(r"\[timing\]\s+(?:sharded|combined) first pass[^:]*\(rank 0, ms\):(.*)", err):
            for name, val in re.findall(r"\s*([^;]*?) ([0-9.eE+-]+);", m.group(1)):
                sharded[name.strip()] = sharded.get(name.strip(), 0.0) + float(val)
        if sharded:
            ph["__sharded__"] = sharded
        m = re.search(r"\[timing\]\s+combined exchange: (\S.*?), ([0-9.eE+-]+) MB received by rank 0", err)
        if m:
            ph["__combined__"] = {"mode": m.group(1), "bytes_received_rank0": int(float(m.group(2)) * 1e6)}
        m = re.search(r"\[timing\] device memory in use after the rounds: ([0-9.eE+-]+) GB", err)
        if m:
            ph["__device_GB__"] = float(m.group(1))
        phases.append((wall, ph))
        occ = int(re.search(r"True marks count: (\d+)", res.stdout.decode()).group(1))
        if rep == 0 and golden:
            sha_ok = sha256_file(out) == golden["bin_sha256"] and occ == golden["true_marks"]
            if not sha_ok:
                return {"error": "e2e output differs from the reference golden %s" % golden["name"]}
        os.unlink(out)
        if settle_s > 0:
            time.sleep(settle_s)
    order = sorted(phases, key=lambda x: x[0])
    walls.sort()
    med = walls[len(walls) // 2]

    def breakdown(ph):
        return {k: ph.get(v) for k, v in BREAKDOWN_KEYS if ph.get(v) is not None or k in ("exec_to_main_ms", "rounds_ms", "write_ms")}

    median_ph = order[len(order) // 2][1]
    sharded = median_ph.get("__sharded__")
    detail = {k: v for k, v in median_ph.items() if k.startswith("code object")}  # per translation unit, median run
    detail["all runs"] = [ph.get("warm-up thread: code objects") for _, ph in order]
    return {"e2e_wall_s": med, "breakdown_ms": breakdown(order[len(order) // 2][1]), "code_objects_ms": detail, "e2e_wall_s_min": walls[0], "e2e_wall_s_p50": med, "e2e_wall_s_max": walls[-1],
            "e2e_wall_s_all": walls, "slowest_run_breakdown_ms": breakdown(order[-1][1]), "settle_s_between_runs": settle_s,
            "device_bytes_allocated": int(median_ph["__device_GB__"] * 1e9) if median_ph.get("__device_GB__") else None,
            "e2e_junction_occurrences_per_sec": occ / med,
            "junction_occurrences": occ, "runs": runs, "host_threads": int(threads), "gpus": gpus,
            "output_sha256_equals_reference": sha_ok,
            "sharded_first_pass_ms_rank0": {k: v for k, v in sharded.items() if k != "region bytes sent"} if sharded else None,
            "region_bytes_sent_rank0": int(sharded["region bytes sent"]) if sharded and "region bytes sent" in sharded else None,
            "combined_exchange_rank0": median_ph.get("__combined__"),
            "what": "twopaco CLI child process%s, process start -> exit (output file closed), %d FASTA files in the page cache, median of %d runs" % (
                " --gpus %d (C++ host, RCCL transport)" % gpus if gpus > 1 else "", len(files), runs)}


def cpu_baseline(recs, p, tmp, mode="sample", timeout=240):
    """The REAL reference binary (oracle/_ref/twopaco_ref, built from /root/reference) on this host's cores.
    sample: the first 6 genomes (bounded: ~1 min); full: the whole workload (several minutes; also times the fixed
    cost -- the serial filter zeroing of concurrentbitvector.cpp:11-24 -- with a one-record input).
    Falls back to the C oracle (kind 'port') when the reference build is absent."""
    from twopaco_amd import synth
    cores = os.cpu_count() or 1
    ref = os.path.join(ROOT, "oracle", "_ref", "twopaco_ref")
    groups = p.get("files") or [(i, i + 1) for i in range(len(recs))]
    n_genomes = len(groups) if mode == "full" else min(6, len(groups))
    sample = recs[:groups[n_genomes - 1][1]]
    kmers = synth.n_kmers(sample, p["k"])
    if os.path.exists(ref):
        files = write_fasta_files(recs, tmp, n=n_genomes, p=p)

        def run(fs, L, to):
            cmd = [ref, "-k", str(p["k"]), "-f", str(L), "-q", str(p["q"]), "-t", str(cores), "--tmpdir", tmp, "-o", os.path.join(tmp, "ref.bin")] + fs
            t0 = time.time()
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=to)
            return time.time() - t0, res

        fixed = None
        if mode in ("full", "sample"):  # the reference's fixed cost at this filter size: allocation + serial zeroing (~11 s at f = 36)
            tiny = os.path.join(tmp, "tiny.fa")
            synth.write_fasta(tiny, [recs[0][:2000]])
            try:
                fixed, _ = run([tiny], p["L"], 1200)
            except subprocess.TimeoutExpired:
                fixed = None
        for L in (p["L"], 32):
            try:
                wall, res = run(files, L, 3600 if mode == "full" else timeout)
            except subprocess.TimeoutExpired:
                continue
            if res.returncode != 0:
                continue
            log = res.stdout.decode()
            m = re.search(r"\n1\t(\d+)\t(\d+)\t", log)
            occ = int(re.search(r"True marks count: (\d+)", log).group(1))
            out = {"value": kmers / wall, "unit": "k-mers/s", "cores": cores, "kind": "reference",
                   "sample": "%s of the workload (%d genomes, %d k-mers), k=%d q=%d f=%d, reference binary -t %d, wall %.1f s "
                             "(its log: fill %s s incl. serial filter zeroing, query %s s); %d junction occurrences"
                             % ("all" if mode == "full" else "first 6 genomes", n_genomes, kmers, p["k"], p["q"], L, cores, wall,
                                m.group(1) if m else "?", m.group(2) if m else "?", occ),
                   "junction_occurrences_per_sec": occ / wall, "wall_s": wall}
            if fixed is not None and L == p["L"]:
                out["fixed_cost_s"] = fixed
                out["value_without_fixed_cost"] = kmers / max(wall - fixed, 1e-9)
                out["sample"] += "; a 2 kbp input takes %.1f s (filter allocation + serial zeroing, concurrentbitvector.cpp:11-24)" % fixed
            return out
    import numpy as np
    from oracle import oracle as O
    o = O.Oracle(p["k"], min(p["L"], 32), p["q"], O.seed_table(1, p["q"], min(p["L"], 32)))
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for r in sample[:2]:
        o.add_record(letters[r].tobytes())
    kmers = synth.n_kmers(sample[:2], p["k"])
    t0 = time.time()
    o.enumerate()
    wall = time.time() - t0
    return {"value": kmers / wall, "unit": "k-mers/s", "cores": 1, "kind": "port",
            "sample": "first 2 genomes (%d k-mers), scalar C oracle, f=%d, wall %.1f s" % (kmers, min(p["L"], 32), wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="m2", help="m2 (the headline: BASELINE configs[2]), m1, m3, m2r2 = m2r + minisatellite tracts (units of 7..60 bp), or m2r = m2 with repeat families, low-complexity tracts, "
                                                      "two reverse-complemented genomes and 50-300 contigs per genome (synth.py)")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--test-first", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="sample", choices=["sample", "full", "none"])
    ap.add_argument("--e2e-runs", type=int, default=5, help="runs of the twopaco CLI for the end-to-end figure (0 = skip)")
    ap.add_argument("--e2e-settle", type=float, default=3.5,
                    help="seconds between two CLI runs of the headline series: the driver wipes a finished process's device memory "
                         "asynchronously (~3 s for the 59 GB of one run) and an allocation that is handed one of those blocks waits for it (profiles/r04f_e2e_back_to_back.txt)")
    ap.add_argument("--e2e-b2b-runs", type=int, default=6, help="runs of the second series, started right behind one another (reported as e2e_back_to_back; 0 = skip)")
    ap.add_argument("--decomposition", default="auto", choices=["auto", "ranges", "address"],
                    help="multi-GPU: the Bloom filter sharded by bit address with an all-to-all per pass (the north-star decomposition; "
                         "power-of-two N), or vertex-hash ranges (the reference's rounds side by side, no data-path exchange).  auto: "
                         "address at every power-of-two N (the scaling curve is one decomposition); below 8 GPUs the range "
                         "decomposition is timed as well and reported under the key 'ranges' (DESIGN.md section 5)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and os.environ.get("TPC_FORCE_DIST") and "RANK" not in os.environ:  # one rank without a launcher: its own rendezvous
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s.getsockname()[1]))
        s.close()
    if world > 1 or os.environ.get("TPC_FORCE_DIST"):  # TPC_FORCE_DIST: exercise the distributed path with one rank
        from twopaco_amd import dist as tdist
        base = None
        if not args.no_cpu_baseline and args.cpu_baseline != "none":
            def base(recs, p):
                tmp = tempfile.mkdtemp(prefix="tpc_bench_")
                try:
                    return cpu_baseline(recs, p, tmp, args.cpu_baseline)
                finally:
                    shutil.rmtree(tmp, ignore_errors=True)
        e2e = None
        if args.e2e_runs > 0:
            def e2e(recs, p, gpus):
                tmp = tempfile.mkdtemp(prefix="tpc_bench_")
                try:
                    return e2e_cli(write_fasta_files(recs, tmp, p=p), p, golden_case(args.workload, args.scale), min(args.e2e_runs, 3), tmp, gpus=gpus,
                                   settle_s=args.e2e_settle)
                finally:
                    shutil.rmtree(tmp, ignore_errors=True)
        return tdist.bench_main(args, rank, world, local_rank, golden=golden_case(args.workload, args.scale), cpu_baseline=base, e2e=e2e)

    import torch
    from twopaco_amd import capi, synth

    torch.cuda.set_device(0)
    recs, p = synth.workload(args.workload, scale=args.scale)
    n_kmers = synth.n_kmers(recs, p["k"])
    golden = golden_case(args.workload, args.scale)
    text = capi.PackedText.from_codes(recs)
    ctx = capi.Context(0)
    ctx.set_option("insert_test_first", args.test_first)
    ctx.set_params(p["k"], p["L"], p["q"], capi.seed_table(p["q"], p["L"], seed=GOLDEN_SEED))
    t0 = time.time()
    ctx.seq_upload(text)
    torch.cuda.synchronize()
    upload_s = time.time() - t0

    for _ in range(args.warmup):
        one_step(ctx)
    names = ["filter_reset", "insert", "query", "compact", "filter2", "scan2", "sort", "emit", "fused", "lookup"]
    kms = {n: 0.0 for n in names}
    fused0 = ctx.stat("fused_lookups")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        marks, st, J, n_valid = one_step(ctx)
        for n in names:
            kms[n] += max(ctx.kernel_ms(n), 0.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = {n: v / args.steps for n, v in kms.items()}
    fused = ctx.stat("fused_lookups") - fused0 == args.steps  # deferred apply: the query's lookup built the filter slices
    paths = {"insert_path": ctx.stat("insert_path"), "query_path": ctx.stat("query_path"), "insert_overflow_entries": ctx.stat("insert_overflow_entries"),
             "query_overflow_entries": ctx.stat("query_overflow_entries"), "insert_batches": ctx.stat("insert_batches"), "query_batches": ctx.stat("query_batches")}
    # bytes per level-2 entry as the passes ran them (tpc_binsp.h: 48-bit query entries in lines of 20, 24-bit insert entries in lines of 40)
    q_l2_bytes = 128 / 20 if ctx.stat("query_entry_fmt") == 6 else 8.0
    i_l2_bytes = 128 / 40 if ctx.stat("insert_entry_fmt") == 3 else 4.0
    result = {"candidate_marks": marks, "junctions": J, "junction_occurrences": n_valid, **st}
    result_ok = None
    if golden:  # the reference's own counters for this workload (VE.h:384-388), tests/golden/cases.json
        r = golden["rounds"][0]
        result_ok = (marks, J, st["true"], st["false"], st["table"]) == (r["marks"], golden["distinct"], r["true"], r["false"], r["table"])
        if not result_ok:
            print("bench: result %r differs from the reference golden %r" % (result, r), file=sys.stderr)
            sys.exit(3)
    ctx.close()

    # HBM bytes per launch: (1) what the write-combining design has to move by construction (DESIGN.md section 3:
    # every Bloom address is written once and read once per partition level, read once more where it is applied,
    # plus one pass over the filter and the packed text) -- the figure `achieved` is priced with, <= measured bytes
    # by construction so frac <= 1; (2) the PMC counters of a rocprofv3 run of this same command (profiles/), only
    # reported when the kernel sources have not changed since.  SURVEY 8d's scattered-atomic model (2 x 64 B per
    # probe) describes the direct kernels; it is printed as `survey_model_GBs` for reference, never as a fraction.
    q = p["q"]
    filter_bytes = (1 << p["L"]) // 8
    ins_addr, qry_addr = q * n_kmers, 6 * n_kmers
    design_ins = 0.375 * n_kmers + ins_addr * (4 + 4 + 2 * i_l2_bytes) + filter_bytes          # W l1, R+W l2, R apply; filter written once
    design_qry = 0.375 * n_kmers + qry_addr * (8 + 8 + 2 * q_l2_bytes) + (0 if fused else filter_bytes)  # 8-byte level-1 entries; filter read once unless fused
    # With the apply deferred into the query's lookup kernel, ONE kernel is shared by the two groups: k_apply_lookup ("lookup": its own
    # event pair inside "fused" = k_q_split + k_apply_lookup, itself inside "query").  The insert's share of THAT KERNEL's time is its
    # share of that kernel's bytes -- the insert's level-2 entries in and the filter out, against the query's level-2 entries in --
    # exactly as tools/pmc_traffic.py splits the kernel's counter bytes (fused_share_insert); k_q_split is the query's alone.
    ins_ms, qry_ms = kms["insert"], kms["query"]
    share = None
    if fused:
        share = (ins_addr * i_l2_bytes + filter_bytes) / (ins_addr * i_l2_bytes + filter_bytes + qry_addr * q_l2_bytes)
    traffic_ins = traffic_qry = None
    pmc_tag = None
    pmc = os.path.join(ROOT, "profiles", PMC_PROFILE)
    if os.path.exists(pmc) and args.workload == "m2" and args.scale == 1.0:
        with open(pmc) as f:
            t = json.load(f)
        if t.get("csrc_signature") == csrc_signature():
            traffic_ins, traffic_qry = t["groups"]["insert"], t["groups"]["query"]
            pmc_tag = PMC_PROFILE
            if fused and t.get("fused_share_insert"):
                share = t["fused_share_insert"]  # the counters' own split of the shared kernel's bytes
        else:
            pmc_tag = "stale: %s was collected on other kernel sources" % PMC_PROFILE

    if fused:
        ins_ms, qry_ms = kms["insert"] + share * kms["lookup"], kms["query"] - share * kms["lookup"]

    def roof(kernel, ms, design, traffic, survey_bytes, floor_per_kmer):
        # `achieved` / `frac`: the HBM bytes the rocprofv3 counters saw per launch of this kernel group (profiles/<PMC_PROFILE>, same
        # command, same kernel sources) over the group's time measured live with HIP events in this run -- the north star's
        # "rocprof achieved-HBM-GB/s".  No counters for these sources (stale profile): no fraction, rather than a number priced with
        # something else.  What the design has to move by construction is reported beside it (design_*), never as `frac`.
        meas = (traffic / (ms * 1e-3) / 1e9) if traffic else None
        des = design / (ms * 1e-3) / 1e9
        floor = floor_per_kmer * n_kmers
        d = {"bound": "hbm", "kernel": kernel, "achieved": meas, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (meas / HBM_PEAK_GBS) if meas else None,
             "traffic": traffic, "traffic_source": pmc_tag, "launch_ms": ms, "algorithmic_bytes_per_launch": design,
             "algorithmic_bytes_per_kmer": design / n_kmers,
             "measured_hbm_GBs": meas,
             "measured_hbm_frac": (meas / HBM_PEAK_GBS) if meas else None,
             "design_GBs": des, "design_frac": des / HBM_PEAK_GBS,
             "survey_model_GBs": n_kmers * survey_bytes / (ms * 1e-3) / 1e9,
             # the implementation-independent floor ("the word-level lower bound ... what a perfectly write-combined /
             # bucketed implementation approaches"): every address crosses HBM once as an 8-byte word, plus the packed text.  `frac`
             # above is priced with what THIS design moves, so a design change that moves fewer bytes lowers numerator and time
             # together; frac_of_floor only moves when the pass gets faster.
             "floor_bytes": floor, "floor_bytes_per_kmer": floor_per_kmer, "frac_of_floor": floor / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "bytes_over_floor": design / floor, "measured_bytes_over_floor": (traffic / floor) if traffic else None}
        return d

    out = {
        "metric": "kmers_hashed_per_sec", "value": n_kmers * args.steps / dt, "unit": "k-mers/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%s: %d genomes x %d bp E. coli-like synthetic (twopaco_amd/synth.py)%s, k=%d q=%d f=%d, 1 round"
                               % (args.workload, len(p.get("files") or recs), sum(r.size for r in recs) // len(p.get("files") or recs),
                                  " in %d records (repeat families, low-complexity tracts, two genomes reverse-complemented)" % len(recs) if p.get("files") else "",
                                  p["k"], p["q"], p["L"]),
                   "kmers": n_kmers, "filter_bytes": filter_bytes, "insert_test_first": args.test_first, "decomposition": "single GPU",
                   "apply_fused_into_lookup": fused, "query_level2_entry_bytes": q_l2_bytes, "insert_level2_entry_bytes": i_l2_bytes},
        "junction_occurrences_per_sec": n_valid * args.steps / dt,
        "insert_kmers_per_sec": n_kmers / (ins_ms * 1e-3),
        "query_kmers_per_sec": n_kmers / (qry_ms * 1e-3),
        "kernel_ms": kms,
        "paths": paths,  # 2 = partitioned with two levels, +10 = completed by the direct kernel; entries that went the overflow lists' way (last step)
        "ps_per_kmer": dt / args.steps / n_kmers * 1e12,
        "insert_ms_with_its_share_of_fused": ins_ms, "query_ms_without_it": qry_ms,
        "shared_kernel": {"kernel": "k_apply_lookup", "ms": kms["lookup"], "insert_share_of_its_bytes": share,
                          "what": "insert group = k_part_hash + k_part_split (kernel_ms.insert) + this share of the shared kernel's own time; k_q_split (kernel_ms.fused - this kernel) is the query's"} if fused else None,
        "result": result,
        "result_equals_reference_golden": result_ok,
        "upload_s_pcie": upload_s,
        # the dominant kernel group of a step is the first-pass query (k_q_hash + k_q_split + k_q_lookup + k_q_verify)
        "roofline": roof("first-pass query (k_q_hash, k_q_split, k_q_lookup / its share of k_apply_lookup, k_q_verify)", qry_ms, design_qry, traffic_qry, 0.375 + 6 * G_BYTES, 0.375 + 6 * 8),
        # the north star's roofline kernel: first-pass Bloom insert
        "roofline_insert": roof("first-pass insert (k_part_hash, k_part_split, k_part_apply / its share of k_apply_lookup)", ins_ms, design_ins, traffic_ins, 0.25 + q * 2 * G_BYTES, 0.25 + 8 * q),
    }
    tmp = tempfile.mkdtemp(prefix="tpc_bench_")
    try:
        if args.e2e_runs > 0:
            files = write_fasta_files(recs, tmp, p=p)
            time.sleep(args.e2e_settle)  # this process's own context (60 GB) has just been closed
            out["e2e"] = e2e_cli(files, p, golden, args.e2e_runs, tmp, settle_s=args.e2e_settle)
            if "error" in out["e2e"]:
                print("bench: " + out["e2e"]["error"], file=sys.stderr)
                sys.exit(4)
            if args.e2e_b2b_runs > 0:  # the same CLI run started right behind its predecessor: p50 / max and the slowest run's own timers
                b2b = e2e_cli(files, p, golden, args.e2e_b2b_runs, tmp, settle_s=0.0)
                if "error" in b2b:
                    print("bench: " + b2b["error"], file=sys.stderr)
                    sys.exit(4)
                out["e2e_back_to_back"] = {k: b2b[k] for k in ("e2e_wall_s_p50", "e2e_wall_s_min", "e2e_wall_s_max", "e2e_wall_s_all", "breakdown_ms",
                                                              "slowest_run_breakdown_ms", "runs", "output_sha256_equals_reference")}
                out["e2e_back_to_back"]["what"] = ("the same CLI runs with no pause between them: an allocation of a run that is handed device memory the driver is still "
                                                    "wiping for the process before it waits for that wipe (~3 s for 59 GB; profiles/r04e_e2e_budget_sweep.txt)")
            out["e2e_junction_occurrences_per_sec"] = out["e2e"]["e2e_junction_occurrences_per_sec"]
            out["e2e_wall_s"] = out["e2e"]["e2e_wall_s"]
        if not args.no_cpu_baseline and args.cpu_baseline != "none":
            out["cpu_baseline"] = cpu_baseline(recs, p, tmp, args.cpu_baseline)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
