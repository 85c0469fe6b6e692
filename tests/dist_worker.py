"""Worker of the world_size-2 tests (spawned by test_dist_cpu.py / test_gpu_dist.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleBackend:
    """CPU stand-in for the HIP backend -- TEST ONLY: lets the rank orchestration (ranges, key
    union, id lookup, record merge) run under gloo without a GPU."""

    def __init__(self, oracle):
        self.o = oracle

    def run_begin(self):
        self.o.dist_begin()

    def round(self, lo, hi, abundance):
        return self.o.dist_round(lo, hi, abundance)

    def local_keys(self):
        return self.o.keys

    def set_keys(self, keys):
        self.o.set_keys(keys)

    def finalize(self):
        return len(self.o.keys)

    def emit(self):
        self._g, self._ids = self.o.lookup_marks()
        return len(self._g), int((self._ids != (1 << 63) - 1).sum())

    def emit_fetch(self):
        return self._g, self._ids


def worker(rank, world, port, case, files, use_gpu, result_path):
    import pickle

    import torch.distributed as dist

    from oracle import oracle as O
    from twopaco_amd import dist as tdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    abundance = case["abundance"] if case["abundance"] is not None else (1 << 64) - 1
    if use_gpu:
        from twopaco_amd import capi
        text = capi.PackedText.from_fasta(files)
        ctx = capi.Context(0)
        ctx.set_params(case["k"], case["L"], case["q"], capi.seed_table(case["q"], case["L"], seed=case["seed"]))
        ctx.seq_upload(text)
        be = tdist.HipBackend(ctx)
    else:
        o = O.Oracle(case["k"], case["L"], case["q"], O.seed_table(case["seed"], case["q"], case["L"]))
        for f in files:
            o.add_fasta(f)
        be = OracleBackend(o)
    st = tdist.sharded_step(be, dist, case["L"], abundance, fetch=True)
    gathered = [None] * world
    dist.all_gather_object(gathered, (st["g"], st["ids"], st["junctions"], st["marks"], st["range"]))
    if rank == 0:
        with open(result_path, "wb") as f:
            pickle.dump(gathered, f)
    dist.barrier()
    dist.destroy_process_group()


def addr_worker(rank, world, port, spec, result_path):
    """Address-sharded filter: `world` ranks (gloo rendezvous, every context on GPU 0).  Each rank
    reports its filter shard after the insert, the merged candidate mask after the query and the
    final (position, id) list; the test reassembles the shards and compares with the oracle.
    `spec` is one configuration or a list of them (run one after the other in the same process group)."""
    import pickle

    import numpy as np
    import torch
    import torch.distributed as dist

    from twopaco_amd import capi, synth
    from twopaco_amd import dist as tdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    code_of = np.zeros(256, dtype=np.uint8)
    code_of[letters] = np.arange(5, dtype=np.uint8)
    results = []
    for sp in (spec if isinstance(spec, list) else [spec]):
        if sp.get("files"):
            text = capi.PackedText.from_fasta(sp["files"])
        elif sp.get("records"):
            text = capi.PackedText.from_codes([code_of[np.frombuffer(r, dtype=np.uint8)] for r in sp["records"]])
        else:
            recs, _ = synth.workload(sp["workload"], scale=sp["scale"])
            text = capi.PackedText.from_codes(recs)
        ctx = capi.Context(0)
        for opt, val in sp.get("options", {}).items():
            ctx.set_option(opt, val)
        windowed = bool(sp.get("text_window"))
        if windowed:  # every rank keeps only its chunk of the packed text: the sharding must be known before the upload
            ctx.set_option("text_window", 1)
            ctx.shard_config(rank, world)
        ctx.set_params(sp["k"], sp["L"], sp["q"], capi.seed_table(sp["q"], sp["L"], seed=sp["seed"]))
        ctx.seq_upload(text)
        if windowed:
            if world > 1 and len(text.bases) > 8 * 512 * world:  # (a window is whole 512-word tiles + a halo: only meaningful on texts of many tiles)
                assert ctx.stat("text_words") < len(text.bases) * (1.0 / world + 0.25) + 600, (ctx.stat("text_words"), len(text.bases))
        if "periodic" in sp:  # (read by AddressSharded: positions inside periodic windows send nothing)
            os.environ["TPC_SHARD_PERIODIC"] = str(int(sp["periodic"]))
        sh = tdist.AddressSharded(ctx, dist, torch.device("cuda", 0), compact=sp.get("compact_exchange", True), configure=not windowed, fused=sp.get("fused_verify", True))
        out = {"rounds": []}
        for lo, hi in sp["ranges"]:
            geom = sh.insert(lo, hi)
            shard = ctx.filter_download()
            qgeom = sh.query(lo, hi)
            out["rounds"].append({"geom": geom, "qgeom": qgeom, "shard": shard, "mask": ctx.mask_download(False),
                                  "survivors": sh.stats["survivors"]})
        st = tdist.address_sharded_step(sh, sp["abundance"], fetch=True, sharded_pass2=sp.get("sharded_pass2", False))
        out.update(g=st["g"], ids=st["ids"], junctions=st["junctions"], true=st["true"], step_marks=st["marks"], moved=sh.comm.bytes_moved, region_bytes_sent=sh.stats.get("region_bytes_sent", 0),
                   relaxed_regions=sh.stats.get("relaxed_regions", 0), overflow_entries=sh.stats.get("overflow_entries", 0), periodic_skip=ctx.stat("periodic_skip"))
        gathered = [None] * world
        dist.all_gather_object(gathered, out)
        results.append(gathered)
        ctx.close()
    if rank == 0:
        with open(result_path, "wb") as f:
            pickle.dump(results if isinstance(spec, list) else results[0], f)
    dist.barrier()
    dist.destroy_process_group()


def combined_worker(rank, world, port, spec, result_path):
    """The combined exchange (twopaco_amd/dist.py:Combined; include/twopaco_hip.h tpc_combine_*): `world` ranks over gloo, every context
    on GPU 0 with option replicate_filter.  Each rank reports its WHOLE filter and its round mask after every round's query, and the
    final (position, id) list.  spec["peek"]: the filter is also downloaded between insert and query (the imported lists are then
    applied without a lookup riding along, and the query reads the dense filter)."""
    import pickle

    import numpy as np
    import torch
    import torch.distributed as dist

    from twopaco_amd import capi, synth
    from twopaco_amd import dist as tdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    letters = np.frombuffer(b"ACGTN", dtype=np.uint8)
    code_of = np.zeros(256, dtype=np.uint8)
    code_of[letters] = np.arange(5, dtype=np.uint8)
    results = []
    for sp in (spec if isinstance(spec, list) else [spec]):
        if sp.get("files"):
            text = capi.PackedText.from_fasta(sp["files"])
        elif sp.get("records"):
            text = capi.PackedText.from_codes([code_of[np.frombuffer(r, dtype=np.uint8)] for r in sp["records"]])
        else:
            recs, _ = synth.workload(sp["workload"], scale=sp["scale"])
            text = capi.PackedText.from_codes(recs)
        ctx = capi.Context(0)
        for opt, val in sp.get("options", {}).items():
            ctx.set_option(opt, val)
        ctx.set_option("replicate_filter", 1)
        windowed = bool(sp.get("text_window"))
        if windowed:
            ctx.set_option("text_window", 1)
        ctx.shard_config(rank, world)  # before the upload (the window) and the parameters (the filter stays whole)
        ctx.set_params(sp["k"], sp["L"], sp["q"], capi.seed_table(sp["q"], sp["L"], seed=sp["seed"]))
        ctx.seq_upload(text)
        sh = tdist.Combined(ctx, dist, torch.device("cuda", 0), configure=False, mode=sp.get("mode"))
        out = {"rounds": []}
        for lo, hi in sp["ranges"]:
            sh.insert(lo, hi)
            peek = ctx.filter_download() if sp.get("peek") else None
            sh.query(lo, hi, union=not sp.get("sharded_pass2"))
            out["rounds"].append({"filter": ctx.filter_download(), "peek": peek, "mask": ctx.mask_download(False), "combine": dict(sh.stats["combine"]),
                                  "fused": ctx.stat("fused_lookups"), "query_batches": ctx.stat("query_batches"), "insert_batches": ctx.stat("insert_batches")})
        st = tdist.address_sharded_step(sh, sp["abundance"], fetch=True, sharded_pass2=sp.get("sharded_pass2", False))
        lo_hi = ctx.shard_chunk() if hasattr(ctx, "shard_chunk") else None
        out.update(g=st["g"], ids=st["ids"], junctions=st["junctions"], true=st["true"], step_marks=st["marks"], moved=sh.comm.bytes_moved, chunk=lo_hi,
                   combine=dict(sh.stats["combine"]), pass2_records_sent=sh.stats.get("pass2_records_sent"))
        gathered = [None] * world
        dist.all_gather_object(gathered, out)
        results.append(gathered)
        ctx.close()
    if rank == 0:
        with open(result_path, "wb") as f:
            pickle.dump(results if isinstance(spec, list) else results[0], f)
    dist.barrier()
    dist.destroy_process_group()


def comm_worker(rank, world, port, result_path, p2p=False):
    """The collectives of the address-sharded driver (twopaco_amd/dist.py:_Comm) over gloo on host tensors."""
    import pickle

    import torch
    import torch.distributed as dist

    from twopaco_amd import dist as tdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # p2p: the grouped send/recv form that runs over RCCL, here over gloo, with 16-byte chunks so that every block takes several messages
    comm = tdist._Comm(dist, torch.device("cpu"), p2p=p2p, chunk=16 if p2p else None)
    out = {}
    # equal blocks: block d of rank r holds (r, d)
    send = torch.stack([torch.full((5,), 16 * rank + d, dtype=torch.uint8) for d in range(world)]).reshape(-1)
    out["equal"] = comm.a2a_equal(send).tolist()
    # larger equal blocks of 8-byte elements with distinct values: (rank, destination, index)
    big = torch.stack([torch.arange(40, dtype=torch.int64) + 1000 * d + 100000 * rank for d in range(world)]).reshape(-1)
    out["equal_big"] = comm.a2a_equal(big).tolist()
    # the same with the own block left where it is (tpc_shard_apply_inplace reads it from the send buffer): block `rank` of the result is undefined
    if world > 1:
        skipped = comm.a2a_equal(big, skip_self=True).reshape(world, -1)
        out["equal_skip"] = [skipped[s].tolist() for s in range(world) if s != rank]
    else:
        out["equal_skip"] = []
    # variable counts well beyond one chunk: rank r sends 7 r + 3 d elements to d
    bc = [7 * rank + 3 * d for d in range(world)]
    bsend = torch.cat([torch.arange(c, dtype=torch.int64) + 1000 * d + 100000 * rank for d, c in enumerate(bc)] + [torch.zeros(0, dtype=torch.int64)])
    brecv, brc = comm.a2a_var(bsend, bc)
    out["var_big"] = (brecv.tolist(), brc)
    # ... received into a buffer the caller provides (the send/recv form only; AddressSharded._out_buf)
    held = torch.full((4096,), -1, dtype=torch.int64)
    brecv2, _ = comm.a2a_var(bsend, bc, out=lambda n, dt: held.view(dt)[:n])
    out["var_out"] = (brecv2.tolist(), (brecv2.data_ptr() == held.data_ptr()) if p2p else None)
    # variable: rank r sends (r + d) % 3 elements to d, values 100 r + d
    counts = [(rank + d) % 3 for d in range(world)]
    send = torch.cat([torch.full((c,), 100 * rank + d, dtype=torch.int64) for d, c in enumerate(counts)] + [torch.zeros(0, dtype=torch.int64)])
    recv, rc = comm.a2a_var(send, counts)
    out["var"] = (recv.tolist(), rc)
    # answers travel back along the same route
    back, bc = comm.a2a_var((recv % 100 + 1).to(torch.uint8), rc)
    out["back"] = (back.tolist(), bc)
    out["gather"] = comm.all_gather(torch.arange(3, dtype=torch.int32) + 10 * rank).tolist()
    out["max"] = comm.max_ints([rank, 7 - rank])
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        with open(result_path, "wb") as f:
            pickle.dump(gathered, f)
    dist.barrier()
    dist.destroy_process_group()


def bench_backend(args, rank, world):
    """TPC_BENCH_BACKEND hook for tests/test_dist_cpu.py::test_bench_gpus_flag_launches_ranks: the oracle as the rank
    backend of `bench.py --gpus N` (TEST ONLY -- exercises the launcher, the rendezvous, the timed loop and the JSON
    line on a machine without GPUs; the product path never loads the oracle)."""
    from helpers import GOLDEN, golden_cases
    from oracle import oracle as O

    case = [c for c in golden_cases() if c["name"] == "rand6_k9_fp"][0]
    o = O.Oracle(case["k"], case["L"], case["q"], O.seed_table(case["seed"], case["q"], case["L"]))
    o.add_fasta(os.path.join(GOLDEN, case["fasta"]))
    p = {"k": case["k"], "L": case["L"], "q": case["q"]}
    n_kmers = int((o.text != 4).sum())
    return OracleBackend(o), n_kmers, p, "rand6.fa k=9 f=14 (test input)"
