"""Multi-GPU driver: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the
GPU node, "gloo" in the CPU tests).

Decomposition (this round): the vertex set is cut into `world` disjoint vertex-hash ranges -- the
reference's own `-r` decomposition (reference vertexenumerator.h:206-254,1063-1073,638) with one
round per GPU instead of one round after the other.  Rank r inserts only the edges that touch its
range, queries only its vertices and exact-filters only its candidates; rounds are independent,
so the only exchange is the union of the per-rank junction key sets (an all-gather of a few MB),
after which every rank sorts the same key list, owns the same ids and looks up the ids of its own
candidates.  The union of the per-rank (position, id) lists is exactly the single-GPU result: the
reference's output does not depend on where the round boundaries are.

Range boundaries: the reference balances rounds with a split-pass histogram; the vertex hash is
min(H(v), H(rc v)) of two well-mixed L-bit hashes, whose density on [0, 2^L) is 2(1-x), so the
equal-mass quantiles x_r = 1 - sqrt(1 - r/world) give the same balance without the extra pass.

The address-sharded filter with an all-to-all of Bloom addresses (BASELINE.json north_star) builds on
the partitioned insert (csrc/tpc_partition.hip: level-1 buckets are the unit that would travel);
it is the next step and is described in DESIGN.md.
"""
import json
import math
import os
import time

import numpy as np


def vertex_hash_ranges(L, world):
    """[(lo, hi)] inclusive, disjoint, covering [0, 2^L] (reference ranges are inclusive, VE.h:473-476)."""
    size = 1 << L
    cuts = [int(size * (1.0 - math.sqrt(1.0 - r / world))) for r in range(world)] + [size + 1]
    return [(cuts[r], cuts[r + 1] - 1) for r in range(world)]


class HipBackend:
    """The product backend: every call goes to the HIP library through the C-ABI."""

    def __init__(self, ctx):
        self.ctx = ctx

    def run_begin(self):
        self.ctx.run_begin()

    def round(self, lo, hi, abundance):
        c = self.ctx
        c.filter_reset()
        c.pass1_insert(lo, hi, count=False)
        marks = c.pass1_query(lo, hi)
        st = c.pass2_filter(abundance)
        st["marks"] = marks
        return st

    def local_keys(self):
        return self.ctx.junction_keys_raw()

    def set_keys(self, keys):
        self.ctx.junction_keys_set(keys)

    def finalize(self):
        return self.ctx.junctions_finalize()

    def emit(self):
        return self.ctx.emit()

    def emit_fetch(self):
        return self.ctx.emit_fetch()


def _dev(dist):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def allgather_keys(dist, keys):
    """Union (concatenation: ranges are disjoint, so are the key sets) of the per-rank key arrays."""
    import torch
    world = dist.get_world_size()
    dev = _dev(dist)
    C = keys.shape[1]
    n = torch.tensor([keys.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(x.item()) for x in sizes]
    m = max(max(sizes), 1)
    pad = np.zeros((m, C), dtype=np.int64)
    pad[:keys.shape[0]] = keys.view(np.int64)
    mine = torch.from_numpy(pad).to(dev)
    parts = [torch.zeros((m, C), dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = [parts[r][:sizes[r]].cpu().numpy().view(np.uint64) for r in range(world)]
    return np.concatenate(out, axis=0) if out else np.zeros((0, C), dtype=np.uint64)


def sharded_step(backend, dist, L, abundance=(1 << 64) - 1, fetch=False):
    """One enumeration with the vertex-hash ranges spread over the ranks.  Returns per-rank stats."""
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = vertex_hash_ranges(L, world)[rank]
    backend.run_begin()
    st = backend.round(lo, hi, abundance)
    keys = allgather_keys(dist, backend.local_keys())
    backend.set_keys(keys)
    st["junctions"] = backend.finalize()
    st["n_marked"], st["n_valid"] = backend.emit()
    st["range"] = (lo, hi)
    if fetch:
        st["g"], st["ids"] = backend.emit_fetch()
    return st


def merge_records(parts, rec_start, rec_length, k, n_junctions):
    """Host-side output pass over the gathered (g, id) lists of all ranks: sort by position, drop
    Bloom false positives, add the stub ids of sequence ends (reference vertexenumerator.h:927-948).
    Returns [(seq, pos, id)] in output order."""
    INVALID = (1 << 63) - 1
    g = np.concatenate([p[0] for p in parts])
    ids = np.concatenate([p[1] for p in parts])
    keep = ids != INVALID
    g, ids = g[keep], ids[keep]
    order = np.argsort(g, kind="stable")
    g, ids = g[order], ids[order]
    out = []
    stub = n_junctions + 42
    cur = 0
    for r in range(len(rec_start)):
        n = int(rec_length[r])
        if n < k:
            continue
        first = int(rec_start[r])
        last = first + n - k
        while cur < len(g) and g[cur] < first:
            cur += 1
        end = cur
        while end < len(g) and g[end] <= last:
            end += 1
        has_first = cur < end and int(g[cur]) == first
        has_last = cur < end and int(g[end - 1]) == last
        if not has_first:
            out.append((r, 0, stub))
            stub += 1
        for i in range(cur, end):
            out.append((r, int(g[i]) - first, int(ids[i])))
        if last != first and not has_last:
            out.append((r, last - first, stub))
            stub += 1
        cur = end
    return out


def bench_main(args, rank, world, local_rank):
    """bench.py --gpus N under torch.distributed.run: strong scaling of the same workload."""
    import torch
    import torch.distributed as dist

    from . import capi, synth

    backend = os.environ.get("TPC_DIST_BACKEND", "nccl")  # "gloo": several ranks on one GPU (testing only)
    ngpu = torch.cuda.device_count()
    device = local_rank % max(ngpu, 1)
    torch.cuda.set_device(device)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    else:
        dist.init_process_group(backend)
    recs, p = synth.workload(args.workload, scale=args.scale)
    n_kmers = synth.n_kmers(recs, p["k"])
    text = capi.PackedText.from_codes(recs)
    ctx = capi.Context(device)
    ctx.set_params(p["k"], p["L"], p["q"], capi.seed_table(p["q"], p["L"], seed=12345))
    ctx.seq_upload(text)
    be = HipBackend(ctx)
    for _ in range(args.warmup):
        sharded_step(be, dist, p["L"])
    names = ["filter_reset", "insert", "query", "compact", "filter2", "scan2", "sort", "emit"]
    kms = {n: 0.0 for n in names}
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = sharded_step(be, dist, p["L"])
        for n in names:
            kms[n] += max(ctx.kernel_ms(n), 0.0) / args.steps
    torch.cuda.synchronize()
    dist.barrier()
    dev = _dev(dist)
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    tot = torch.tensor([st["n_valid"], st["marks"]], dtype=torch.int64, device=dev)
    dist.all_reduce(tot)
    dt = float(dt.item())
    if rank == 0:
        out = {
            "metric": "kmers_hashed_per_sec", "value": n_kmers * args.steps / dt, "unit": "k-mers/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s: %d genomes x %d bp E. coli-like synthetic (twopaco_amd/synth.py), k=%d q=%d f=%d"
                                   % (args.workload, len(recs), recs[0].size, p["k"], p["q"], p["L"]),
                       "kmers": n_kmers, "filter_bytes": (1 << p["L"]) // 8,
                       "parallelism": "%d vertex-hash ranges, one per GPU (reference rounds run side by side); all-gather of junction keys over RCCL" % world},
            "junction_occurrences_per_sec": int(tot[0].item()) * args.steps / dt,
            "kernel_ms_rank0": kms,
            "result": {"candidate_marks": int(tot[1].item()), "junctions": st["junctions"], "junction_occurrences": int(tot[0].item())},
        }
        print(json.dumps(out))
    dist.destroy_process_group()
