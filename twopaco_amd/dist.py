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

A second decomposition, `address_sharded_round`, cuts the Bloom filter itself over the ranks by bit
address (BASELINE.json north_star): the filter of 2^L bits no longer has to fit one GPU.  Every rank
hashes 1/world of the text; the level-1 regions of the partitioned passes (csrc/tpc_partition.hip,
tpc_qpartition.hip) are exactly "the addresses for rank d", so one equal-split all_to_all per pass
moves them to the owner of their filter slices.  See the docstring of `address_sharded_round`.
"""
import json
import math
import os
import sys
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

    def union_keys_on_device(self, dist):
        """All-gather of the per-rank junction keys without a host round trip: one message per rank =
        [count, keys...] in a device buffer of `cap` keys that only grows (every rank sees every count, so all
        ranks agree when a second, larger exchange is needed)."""
        import torch
        ctx, C = self.ctx, self.ctx.key_words()
        dev = torch.device("cuda", torch.cuda.current_device())
        if not hasattr(self, "_comm"):
            self._comm = _Comm(dist, dev)
            self._cap = 1 << 17
        while True:
            cap = self._cap
            buf = getattr(self, "_keybuf", None)
            if buf is None or buf.numel() != 1 + cap * C:
                buf = self._keybuf = torch.empty(1 + cap * C, dtype=torch.int64, device=dev)
            self._comm.phase = "junction key union"
            try:
                n = ctx.junction_keys_export(buf.data_ptr() + 8, cap)
            except Exception as e:  # noqa: BLE001 -- the count slot carries the failure to every rank (agreement without an extra collective)
                self._comm.fail(e)
                n = -1
            buf[0] = n
            allb = self._comm.all_gather(buf)
            counts = [int(x) for x in allb[:, 0].cpu().tolist()]
            self._comm._agreed([1 if c < 0 else 0 for c in counts], "all_gather")
            if max(counts) <= cap:
                break
            self._cap = int(max(counts) * 1.5) + 1024
        self._comm.sync()
        for r, m in enumerate(counts):
            ctx.junction_keys_import(allb[r].data_ptr() + 8, m, append=r > 0)

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
    if hasattr(backend, "union_keys_on_device"):
        backend.union_keys_on_device(dist)
    else:
        backend.set_keys(allgather_keys(dist, backend.local_keys()))
    st["junctions"] = backend.finalize()
    st["n_marked"], st["n_valid"] = backend.emit()
    st["range"] = (lo, hi)
    if fetch:
        st["g"], st["ids"] = backend.emit_fetch()
    return st


_SERIAL = [False, None]


def _serial_lock():
    """File descriptor of the lock the ranks of an emulated run take turns with (TPC_DIST_SERIALIZE), or None."""
    if not _SERIAL[0]:
        _SERIAL[0] = True
        path = os.environ.get("TPC_DIST_SERIALIZE")
        if path:
            _SERIAL[1] = os.open(path, os.O_CREAT | os.O_RDWR, 0o600)
    return _SERIAL[1]


LINK_GBS = 50.0  # usable GB/s per xGMI link and direction the link model assumes (7 links of 76.8 GB/s peak per direction on an MI355X; RCCL send / recv)


def link_model(world, compute_ms, recv_bytes_per_peer_phases, hidden_ms=0.0):
    """Predicted milliseconds per step of a rank: its own library calls (measured: alone on a GPU) plus, for every exchange phase, the
    bytes it receives from ONE peer over that peer's link at LINK_GBS (the W - 1 links of a fully connected node work side by side;
    nothing is assumed to overlap with compute).  recv_bytes_per_peer_phases: {phase: bytes from the busiest peer}."""
    wire = {k: v / (LINK_GBS * 1e9) * 1e3 for k, v in recv_bytes_per_peer_phases.items()} if world > 1 else {}
    # hidden_ms: the query's hash and binning, enqueued before the insert's exchange (tpc_pass1_query_begin): that much of the exchange's
    # wire time (the phases before the query; not the second pass's records) costs nothing
    early = sum(v for k, v in wire.items() if not k.startswith("second pass"))
    hidden = min(hidden_ms, early)
    return {"compute_ms": compute_ms, "wire_ms": wire, "wire_ms_total": sum(wire.values()), "predicted_ms_no_overlap": compute_ms + sum(wire.values()),
            "query_hash_and_binning_ms_under_the_exchange": hidden_ms, "predicted_ms": compute_ms + sum(wire.values()) - hidden,
            "link_GBs_assumed": LINK_GBS,
            "what": "measured library-call time of the slowest rank (ranks taking turns on the device) + modelled wire time (bytes from the busiest peer over one link); "
                    "only the query's hash and binning are assumed to run under the insert's exchange"}


class _ListOverflow(RuntimeError):
    """An overflow list of a sharded pass overflowed (seen by every rank in the same all-reduce): AddressSharded re-plans once."""


class DistAbort(RuntimeError):
    """Every rank raises this together: some rank's local work failed before the collective named in the message."""


class PhaseWatchdog:
    """Wall-clock limit per collective phase (TPC_DIST_TIMEOUT_S, default 120 s; 0 = off).  A collective whose peers never
    arrive blocks inside the communication library, where no Python exception can reach it; one monitor thread per process
    watches the deadline of the phase its rank is in and, when it passes, writes the phase's name to stderr and ends the
    process with exit code 17 (os._exit: no re-exec, no attempt to unwind through the hung call).  Under torch.distributed.run
    the agent then stops the other ranks, so the launch as a whole exits non-zero instead of hanging; without a launcher
    every rank's own watchdog does the same within the limit.  What the reference has in this place is a shared fetch_or
    (concurrentbitvector.cpp:31-45) that cannot wait for anybody."""

    EXIT_CODE = 17

    def __init__(self, rank, timeout_s=None):
        import threading
        if timeout_s is None:
            timeout_s = float(os.environ.get("TPC_DIST_TIMEOUT_S", "120"))
        self.rank, self.timeout = rank, timeout_s
        self._phase, self._deadline = None, None
        self._lock = threading.Lock()
        if self.timeout > 0:
            t = threading.Thread(target=self._run, name="tpc-dist-watchdog", daemon=True)
            t.start()

    def enter(self, phase, scale=1.0):
        """scale: multiple of the limit for phases that also wait for a peer's SETUP (workload synthesis, context creation, the
        lazy RCCL communicator of the first collective): TPC_DIST_SETUP_TIMEOUT_S (default 5 x the limit) for the first barrier."""
        with self._lock:
            self._phase, self._deadline = phase, time.monotonic() + self.timeout * scale
            self._started = time.time()

    def leave(self):
        with self._lock:
            self._phase, self._deadline = None, None

    def _run(self):
        period = max(0.05, min(0.5, self.timeout / 4))
        while True:
            time.sleep(period)
            with self._lock:
                phase, deadline = self._phase, self._deadline
            if deadline is not None and time.monotonic() > deadline:
                sys.stderr.write("twopaco_amd.dist: rank %d: collective phase '%s' exceeded TPC_DIST_TIMEOUT_S = %g s "
                                 "(entered at %s; a peer never arrived): exiting with code %d\n" % (
                                     self.rank, phase, self.timeout, time.strftime("%H:%M:%S", time.localtime(getattr(self, "_started", 0))), self.EXIT_CODE))
                sys.stderr.flush()
                os._exit(self.EXIT_CODE)


class _Comm:
    """The collectives of the address-sharded path over torch.distributed.  backend "nccl" (RCCL) moves
    device tensors directly; with "gloo" (tests: several ranks on one GPU) they are staged through
    the host."""

    def __init__(self, dist, device, p2p=None, chunk=None):
        import torch
        self.dist, self.torch, self.device = dist, torch, device
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.direct = dist.get_backend() == "nccl"
        # p2p: the grouped send/recv form of the all_to_alls (what runs over RCCL).  Selectable under gloo too, on host
        # tensors, so that its block / chunk arithmetic is exercised by the CPU tests with several ranks
        self.p2p = self.direct if p2p is None else bool(p2p)
        if chunk:
            self.CHUNK = int(chunk)
        self.bytes_moved = 0
        self.phase = "setup"     # set by the driver: names the pass a collective belongs to (watchdog / abort messages)
        self.rc, self.err = 0, ""  # this rank's local failure since the last agreement (AddressSharded._try)
        self.watchdog = PhaseWatchdog(self.rank)

    def _enter(self, op):
        self.watchdog.enter("%s:%s" % (self.phase, op))

    def _leave(self):
        self.watchdog.leave()

    def fail(self, exc):
        """A local failure of this rank: remembered, and agreed upon by every rank at the next collective."""
        if not self.rc:
            self.rc, self.err = 1, "%s: %s" % (type(exc).__name__, exc)
            sys.stderr.write("twopaco_amd.dist: rank %d: local failure in phase '%s': %s\n" % (self.rank, self.phase, self.err))
            sys.stderr.flush()

    def _agreed(self, flags, op):
        """flags[r] != 0: rank r failed before this collective.  Every rank sees the same flags and raises the same DistAbort."""
        bad = [r for r, f in enumerate(flags) if f]
        if bad:
            mine = " (this rank: %s)" % self.err if self.rc else ""
            raise DistAbort("phase '%s:%s': rank(s) %s failed before the collective%s" % (self.phase, op, bad, mine))

    def sync(self):
        if self.torch.device(self.device).type == "cuda":
            self.torch.cuda.current_stream(self.device).synchronize()

    def _in(self, t):
        return t if self.direct else t.cpu()

    def _out(self, t):
        return t if self.direct else t.to(self.device)

    CHUNK = 1 << 28  # bytes per peer and message: multi-GiB all_to_all_single calls arrived truncated under RCCL (zeros past a point)

    def a2a_equal(self, send, skip_self=False):
        """Block d of `send` goes to rank d; block s of the result came from rank s.  Over RCCL: grouped send/recv of at
        most CHUNK bytes per peer (what a C++ host issues as ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd).
        skip_self: the caller reads its own block from `send` (tpc_shard_apply_inplace); block `rank` of the result is then
        undefined (not copied)."""
        torch = self.torch
        self.sync()
        if self.world == 1:
            return send  # one rank: the block is already where it is read
        self.bytes_moved += send.numel() * send.element_size() * ((self.world - 1) / self.world if skip_self else 1)
        self._enter("all_to_all(equal)")
        try:
            return self._a2a_equal(send, skip_self)
        finally:
            self._leave()

    def _a2a_equal(self, send, skip_self=False):
        torch = self.torch
        if not self.p2p:
            s = self._in(send)
            r = torch.empty_like(s)
            self.dist.all_to_all_single(r, s)
            out = self._out(r)
            self.sync()
            return out
        shape, dtype = send.shape, send.dtype
        send = self._in(send)
        sb = send.contiguous().view(torch.uint8)
        wide = sb.numel() % (8 * self.world) == 0   # 8-byte elements: byte-wise copy kernels are slow
        sb = (sb.view(torch.int64) if wide else sb).view(self.world, -1)
        recv = torch.empty_like(sb)
        block, step = sb.shape[1], self.CHUNK // (8 if wide else 1)
        for c0 in range(0, block, step):
            c1 = min(block, c0 + step)
            ops = []
            for p in range(self.world):
                if p == self.rank:
                    if not skip_self:
                        recv[p, c0:c1].copy_(sb[p, c0:c1])
                else:
                    ops.append(self.dist.P2POp(self.dist.isend, sb[p, c0:c1], p))
                    ops.append(self.dist.P2POp(self.dist.irecv, recv[p, c0:c1], p))
            if ops:
                for w in self.dist.batch_isend_irecv(ops):
                    w.wait()
        out = self._out(recv.view(-1).view(torch.uint8).view(dtype).view(shape))
        self.sync()  # inside the watchdog's phase: over RCCL the waits above only order the stream
        return out

    def a2a_var(self, send, counts, out=None):
        """`send` holds counts[d] elements for rank d, in rank order.  Returns (received, counts per source).  out(numel, dtype): where
        to receive (device path); one rank: nothing moves and `send` itself is returned."""
        torch = self.torch
        self.sync()
        if self.world == 1:
            self._agreed([self.rc], "all_to_all(variable)")
            return send, [int(counts[0])]
        self._enter("all_to_all(variable)")
        try:
            return self._a2a_var(send, counts, out)
        finally:
            self._leave()

    def _a2a_var(self, send, counts, out=None):
        torch = self.torch
        # the count exchange carries every rank's failure flag: {count, rc} per peer, so all ranks agree before any payload moves
        # (a rank whose local work failed sends zero counts and rc = 1; nobody is left waiting for its data)
        if self.rc:
            counts = [0] * self.world
            send = send[:0]
        sc2 = torch.empty((self.world, 2), dtype=torch.int64)
        sc2[:, 0] = torch.as_tensor(counts, dtype=torch.int64)
        sc2[:, 1] = self.rc
        if self.direct:
            scd = sc2.to(self.device)
            rcd = torch.empty_like(scd)
            self.dist.all_to_all_single(rcd, scd)
            rc2 = rcd.cpu()
        else:
            rc2 = torch.empty_like(sc2)
            self.dist.all_to_all_single(rc2, sc2)
        self._agreed([int(x) for x in rc2[:, 1].tolist()], "all_to_all(variable)")
        rcl, scl = [int(x) for x in rc2[:, 0].tolist()], [int(x) for x in counts]
        self.bytes_moved += send.numel() * send.element_size()
        if not self.p2p:
            s = self._in(send)
            r = torch.empty(sum(rcl), dtype=send.dtype, device=s.device)
            self.dist.all_to_all_single(r, s, output_split_sizes=rcl, input_split_sizes=scl)
            out = self._out(r)
            self.sync()
            return out, rcl
        # RCCL: grouped send/recv, at most CHUNK bytes per peer and message
        send = self._in(send)
        r = out(sum(rcl), send.dtype) if out is not None else torch.empty(sum(rcl), dtype=send.dtype, device=send.device)
        step = max(1, self.CHUNK // send.element_size())
        so, ro = [0], [0]
        for p in range(self.world):
            so.append(so[-1] + scl[p]); ro.append(ro[-1] + rcl[p])
        if scl[self.rank] != rcl[self.rank]:
            raise RuntimeError("a2a_var: a rank's own block must have one size")
        if scl[self.rank]:
            r[ro[self.rank]:ro[self.rank + 1]].copy_(send[so[self.rank]:so[self.rank + 1]])
        for c0 in range(0, max(max(scl), max(rcl), 1), step):
            ops = []
            for p in range(self.world):
                if p == self.rank:
                    continue
                a, b = min(c0, scl[p]), min(c0 + step, scl[p])
                if b > a:
                    ops.append(self.dist.P2POp(self.dist.isend, send[so[p] + a:so[p] + b], p))
                a, b = min(c0, rcl[p]), min(c0 + step, rcl[p])
                if b > a:
                    ops.append(self.dist.P2POp(self.dist.irecv, r[ro[p] + a:ro[p] + b], p))
            if ops:
                for w in self.dist.batch_isend_irecv(ops):
                    w.wait()
        out = self._out(r)
        self.sync()
        return out, rcl

    def all_gather(self, t):
        """[world, *t.shape]"""
        torch = self.torch
        self.sync()
        self._enter("all_gather")
        try:
            s = self._in(t.contiguous())
            out = torch.empty((self.world,) + tuple(s.shape), dtype=s.dtype, device=s.device)
            self.dist.all_gather_into_tensor(out, s) if self.direct else self.dist.all_gather(list(out.unbind(0)), s)
            self.bytes_moved += t.numel() * t.element_size() * self.world
            out = self._out(out)
            self.sync()
            return out
        finally:
            self._leave()

    def sum_ints(self, values):
        """Element-wise sum over the ranks (the failure flag is agreed separately: a max)."""
        torch = self.torch
        self._enter("all_reduce(sum)")
        try:
            v = torch.as_tensor(list(values), dtype=torch.int64)
            v = v.to(self.device) if self.direct else v
            self.dist.all_reduce(v, op=self.dist.ReduceOp.SUM)
            out = [int(x) for x in v.cpu().tolist()]
        finally:
            self._leave()
        self.max_ints([])  # agreement: every rank learns of a failed peer before the next payload moves
        return out

    def max_ints(self, values):
        """Element-wise maximum over the ranks; this rank's failure flag rides along as one more element (agreement)."""
        torch = self.torch
        self._enter("all_reduce(max)")
        try:
            v = torch.as_tensor(list(values) + [self.rc], dtype=torch.int64)
            v = v.to(self.device) if self.direct else v
            self.dist.all_reduce(v, op=self.dist.ReduceOp.MAX)
            out = [int(x) for x in v.cpu().tolist()]
        finally:
            self._leave()
        if out[-1]:
            mine = " (this rank: %s)" % self.err if self.rc else ""
            raise DistAbort("phase '%s:all_reduce(max)': a rank failed before the collective%s" % (self.phase, mine))
        return out[:-1]

    def agree(self):
        """Agreement on its own (one tiny all-reduce), before collectives that carry no counts: equal-block exchanges, all_gathers."""
        self.max_ints([])

    def barrier(self, what="barrier"):
        self._enter(what)
        try:
            self.dist.barrier()
        finally:
            self._leave()


INSERT, QUERY = 0, 1


class AddressSharded:
    """Address-sharded first pass for one rank (context `ctx`, already holding parameters and text).

    Per pass and batch of tiles:
      hash    tpc_shard_hash: this rank's tiles -> level-1 regions, destination major
      move    all_to_all (equal blocks) of the regions and their fill counts
      apply   tpc_shard_apply: levels 2-3 on the owned filter slices (insert: OR; query: first probe)
    Query only: the survivors of the first probe (edge ids) return to the rank that hashed them (tpc_shard_survivor_sources,
    variable all_to_all) and are checked there against hash functions
    1..q-1 -- addresses to their owners (variable all_to_all), one byte back per address; function 1
    alone first, which rejects most Bloom false positives, then the rest in one exchange -- and the
    survivors of all q functions are the candidate marks.  The grouping of the probes by owner is done by the library
    (tpc_shard_route / _permute64 / _select).  Finally the per-rank masks are OR-ed (an all_to_all of word ranges, a fold,
    an all_gather: RCCL has no bitwise-OR reduction), after which every rank holds the mask tpc_pass1_query would have
    produced and the second pass runs as on one GPU.
    Overflowing level-1 regions (skewed addresses) travel as an all-gathered list; beyond the
    list capacity the library fails loudly (there is no direct-kernel fallback on a sharded filter)."""

    def __init__(self, ctx, dist, device, compact=None, configure=True, fused=None):
        import torch
        self.ctx, self.torch = ctx, torch
        self.comm = _Comm(dist, device)
        # compact: exact-size exchange of the level-1 regions (tpc_shard_pack: a read and a write of every entry, then a variable
        # all_to_all) instead of the equal blocks moved as they are, the own block read in place (tpc_shard_apply_inplace).  The
        # regions are sized at the expected fill + 6 sigma + 128 entries (option shard_tight_regions): a few per cent of slack up to
        # four ranks, ~12 % at eight (a region then holds 3.6 K entries on the 62-genome workload), where the wire is what a pass
        # waits for: packing pays there and not below.  TPC_SHARD_EXCHANGE = packed / equal overrides.
        if compact is None:
            mode = os.environ.get("TPC_SHARD_EXCHANGE", "auto")
            compact = mode == "packed" or (mode == "auto" and self.comm.world >= 8)
        self.compact = bool(compact)
        # fused: the verification's bookkeeping through the library's fused calls (_verify) or step by step (_verify_unfused)
        self.fused = os.environ.get("TPC_VERIFY_FUSED", "1") != "0" if fused is None else bool(fused)
        self.device = device
        self.rank, self.world = self.comm.rank, self.comm.world
        if configure:  # False: the caller did it before tpc_seq_upload (needed for option text_window to take effect)
            ctx.shard_config(self.rank, self.world)
        # positions inside homopolymer / dinucleotide tracts repeat their neighbour's window: they send nothing, query() copies their verdicts
        # after the last batch (tpc_shard_periodic_copy).  A sharded filter has no fallback behind its overflow lists, and tracts are what fills them.
        self.periodic = os.environ.get("TPC_SHARD_PERIODIC", "1") != "0"
        if self.periodic:
            ctx.set_option("shard_periodic_skip", 1)
        self._bufs = {}
        self.stats = {}
        self.t = {}   # seconds per phase, accumulated (host clock; every phase ends synchronised)
        self.call_ms = {}  # milliseconds per library call, accumulated (_try)

    def _tick(self, name, t0):
        self.t[name] = self.t.get(name, 0.0) + (time.perf_counter() - t0)
        return time.perf_counter()

    def _try(self, fn, *args, default=0):
        """This rank's local work between two collectives.  A failure (a library call's error code, an out-of-memory) must not
        leave the peers waiting in the next collective for data that will never come: it is remembered (comm.fail) and `default`
        returned, later local work is skipped, and the next collective's count exchange / all-reduce carries the flag, so that
        EVERY rank raises DistAbort there with the phase's name (multigpu.cpp does the same behind its rank barrier)."""
        if self.comm.rc:
            return default
        # every library call's own wall time (the calls end synchronised): call_ms[name], what the link model adds the wire to.  With
        # TPC_DIST_SERIALIZE=<lock file> the ranks of an emulated run (several ranks on ONE device) take turns on the device, so that
        # a call's time is that of a rank alone on its GPU.
        try:
            return self._timed(fn, *args)
        except Exception as e:  # noqa: BLE001 -- whatever it is, the peers must hear of it
            self.comm.fail(e)
            return default

    def _timed(self, fn, *args):
        """fn(*args) with its wall time added to call_ms[fn's name]; under TPC_DIST_SERIALIZE while holding the ranks' device lock."""
        lock = _serial_lock()
        if lock is not None:
            import fcntl
            fcntl.flock(lock, fcntl.LOCK_EX)
        t0 = time.perf_counter()
        try:
            return fn(*args)
        finally:
            name = getattr(fn, "__name__", "call")
            self.call_ms[name] = self.call_ms.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
            if lock is not None:
                fcntl.flock(lock, fcntl.LOCK_UN)

    def _buf(self, name, nbytes):
        b = self._bufs.get(name)
        if b is None or b.numel() < nbytes:
            b = self.torch.empty(max(nbytes, 16), dtype=self.torch.uint8, device=self.device)
            self._bufs[name] = b
        return b[:nbytes]

    def _out_buf(self, name):
        """Receive-buffer provider for _Comm.a2a_var: a named persistent buffer instead of a fresh allocation per call."""
        def alloc(numel, dtype):
            size = self.torch.empty(0, dtype=dtype).element_size()
            return self._buf(name, max(numel, 1) * size).view(dtype)[:numel]
        return alloc

    def _exchange(self, which, geom, batch, lo, hi):
        W = self.world
        send_r = self._buf("send_r", W * geom["region_block_bytes"])
        send_c = self._buf("send_c", W * geom["count_block_bytes"])
        tag = "insert" if which == INSERT else "query"
        self.comm.phase = "%s batch %d" % (tag, batch)
        t0 = time.perf_counter()
        n_ovf = self._try(self.ctx.shard_hash, which, batch, send_r.data_ptr(), send_c.data_ptr(), lo, hi)
        t0 = self._tick(tag + "_hash", t0)
        # one tiny all-reduce BEFORE anything moves: the largest overflow list of the batch and every rank's failure flag
        m = self.comm.max_ints([n_ovf])[0]
        self.stats["overflow_entries"] = self.stats.get("overflow_entries", 0) + (n_ovf if n_ovf < (1 << 62) else 0)
        if m >= (1 << 62):  # (the all-reduced maximum: every rank is here together)
            raise _ListOverflow("address-sharded pass: an overflow list overflowed (adversarial address skew); use the vertex-hash-range decomposition")
        recv_c = self.comm.a2a_equal(send_c)
        if self.compact:
            # the regions are ~3/4 full: pack their used prefixes, move exactly those (block sizes are multiples of 128 bytes)
            packed = self._buf("packed", W * geom["region_block_bytes"])
            nbytes = self._try(self.ctx.shard_pack, which, send_r.data_ptr(), send_c.data_ptr(), packed.data_ptr(), W, default=[0] * W)
            t0 = self._tick(tag + "_pack", t0)
            recv_r, _ = self.comm.a2a_var(packed[:sum(nbytes)].view(self.torch.int64), [b // 8 for b in nbytes])
            recv_r = recv_r.view(self.torch.uint8)
            if recv_r.numel() == 0:
                recv_r = self._buf("recv_empty", 16)
            self.stats["region_bytes_sent"] = self.stats.get("region_bytes_sent", 0) + sum(nbytes) - nbytes[self.rank]  # what leaves this rank
        else:
            # equal blocks as they are; this rank's own block stays where it was hashed (one rank: nothing moves at all)
            recv_r = self.comm.a2a_equal(send_r, skip_self=True) if W > 1 else None
            self.stats["region_bytes_sent"] = self.stats.get("region_bytes_sent", 0) + send_r.numel() * (W - 1) // W
        self.comm.sync()
        t0 = self._tick(tag + "_all_to_all", t0)
        # skew path: entries that did not fit their level-1 region, for any owner
        if m > 0:
            eb = geom["overflow_entry_bytes"]
            mine = self.torch.zeros(m * eb + 8, dtype=self.torch.uint8, device=self.device)
            self._try(self.ctx.shard_overflow_get, which, mine.data_ptr() + 8, n_ovf)
            mine[:8] = self.torch.tensor([n_ovf], dtype=self.torch.int64).view(self.torch.uint8).to(self.device)
            self.comm.agree()
            allv = self.comm.all_gather(mine)
            parts = []
            for r in range(W):
                n = int(allv[r, :8].view(self.torch.int64).item())
                parts.append(allv[r, 8:8 + n * eb])
            cat = self.torch.cat(parts).contiguous()
            self.comm.sync()
            self._try(self.ctx.shard_overflow_set, which, cat.data_ptr(), cat.numel() // eb)
        self.comm.sync()
        return recv_r, recv_c, send_r, send_c

    def _apply(self, which, b, x):
        recv_r, recv_c, send_r, send_c = x
        # The apply side extends the gathered overflow list with its own level-2 losses (repeat-rich input): when THAT overflows
        # (the library's -20) the pass is re-planned once like a hash-side overflow -- the ranks agree on it in one tiny all-reduce,
        # any other failure goes the usual way (comm.fail -> DistAbort at the next collective)
        n, lost = 0, 0
        if not self.comm.rc:
            try:
                if self.compact:
                    n = self.ctx.shard_apply_packed(which, b, recv_r.data_ptr(), recv_c.data_ptr())
                else:
                    n = self.ctx.shard_apply_inplace(which, b, recv_r.data_ptr() if recv_r is not None else 0, recv_c.data_ptr(), send_r.data_ptr(), send_c.data_ptr())
            except Exception as e:  # noqa: BLE001
                if "list overflowed" in str(e):
                    lost = 1
                else:
                    self.comm.fail(e)
        if self.comm.max_ints([lost])[0]:
            raise _ListOverflow("address-sharded pass: an overflow list overflowed on the apply side (address skew); use the vertex-hash-range decomposition")
        return n

    def _relax(self, exc):
        """An overflow list overflowed on some rank (every rank sees it in the same all-reduce).  Once per object: the level-1 regions go
        back from the tight size (expected fill of the densest bucket + 6 sigma) to the one-GPU size (1.3 x + 8 sigma: 30 % more bytes
        on the wire below eight ranks) and the pass starts again -- a sharded filter has no direct-kernel fallback behind its lists, and
        repeat-rich input (microsatellites: one k-mer in the same bucket thousands of times per workgroup) is what overflows them first."""
        if getattr(self, "_relaxed", False):
            raise RuntimeError(str(exc))
        self._relaxed = True
        for key, val in getattr(self, "_stats_at_pass", {}).items():  # the failed attempt's share of the per-pass counters
            self.stats[key] = val
        self.stats["relaxed_regions"] = 1
        self._try(self.ctx.set_option, "shard_tight_regions", 0)
        self._bufs.clear()  # the block sizes of the exchange buffers change with the plan

    def _pass_begins(self):
        self._stats_at_pass = {key: self.stats.get(key, 0) for key in ("overflow_entries", "region_bytes_sent")}

    def insert(self, lo=0, hi=None):
        self._pass_begins()
        try:
            return self._insert(lo, hi)
        except _ListOverflow as e:
            self._relax(e)
            return self._insert(lo, hi)  # (filter_reset first: what the failed attempt applied is gone)

    def _insert(self, lo=0, hi=None):
        self.comm.phase = "insert plan"
        geom = self._try(self.ctx.shard_plan, INSERT, lo, hi, default=None)
        self.comm.agree()  # every rank has a plan (the batch geometry must agree) before the first exchange
        self._try(self.ctx.filter_reset)
        for b in range(geom["batches"]):
            x = self._exchange(INSERT, geom, b, lo, hi)
            t0 = time.perf_counter()
            self._apply(INSERT, b, x)
            self._tick("insert_apply", t0)
        return geom

    def _verify_rounds(self, n_first):
        """(first function, count) of the verification's round trips.  Lazy -- function 1 alone, then 2..q-1 of what is left -- pays
        when most first-probe survivors are Bloom false positives of function 0 (a well-filled filter: the second probe rejects
        them); when most are true second edges (the 62-genome workload: 54 of 58 M pass every probe) every survivor makes both
        trips and one trip with all q - 1 addresses is cheaper.  The pass rate of function 1 on THIS rank's first batch decides
        for the batches after it; the ranks agree through an all-reduce (they must issue the same collectives)."""
        q = self.ctx.q
        if q < 2:
            return []
        mode = os.environ.get("TPC_VERIFY_ROUNDS", "auto")  # "lazy", "eager", "auto"
        eager = mode == "eager" or (mode == "auto" and getattr(self, "_fn1_pass_rate", 0.0) > 0.5)
        if eager or q == 2:
            return [(1, q - 1)]
        return [(1, 1), (2, q - 2)]

    def _verify(self, b, n, t0):
        """Survivors of batch b's first probe: home, verified against functions 1..q-1, marked -- through the library's fused calls
        (tpc_shard_survivors_home / _verify_send / _finish).  Returns the survivor counts after every step."""
        torch, ctx, W = self.torch, self.ctx, self.world
        zeros = [0] * W
        if W == 1 and not os.environ.get("TPC_VERIFY_ROUTED"):  # one rank: every survivor is home and every probe is owned here
            self.comm.phase = "query batch %d: local verification" % b
            self._try(ctx.shard_verify_local)
            self._tick("query_verify_finish", t0)
            return [n]
        self.comm.phase = "query batch %d: survivors home" % b
        # survivors go back to the rank that hashed their position (it rides in the id) and are verified there, where
        # their text is: a rank then needs only its own chunk of the packed text.  The library groups them by that rank.
        home = self._buf("v_home", max(n, 1) * 8).view(torch.int64)
        tmp = self._buf("v_tmp", max(n, 1) * 8 if W > 1 else 8).view(torch.int64)
        counts = self._try(ctx.shard_survivors_home, tmp.data_ptr(), home.data_ptr(), W, default=zeros)
        sid, _ = self.comm.a2a_var(home[:n], counts, out=self._out_buf("v_sid"))
        self.comm.sync()
        t0 = self._tick("query_survivors_home", t0)
        trace = [n]
        # function 1 alone (drops the Bloom false positives), then functions 2..q-1 in one exchange -- or all at once (_verify_rounds)
        rounds = self._verify_rounds(n)
        for ri, (fn, cnt) in enumerate(rounds):
            self.comm.phase = "query batch %d: probes of functions %d..%d" % (b, fn, fn + cnt - 1)
            n = sid.numel()
            # probe addresses in owner-major send order + the slot of every probe (answers come back in that order)
            send = self._buf("v_send", max(n * cnt, 1) * 8).view(torch.int64)
            tmp = self._buf("v_tmp", max(n * cnt, 1) * 8 if W > 1 else 8).view(torch.int64)
            perm = self._buf("v_perm", max(n * cnt, 1) * 4 if W > 1 else 4).view(torch.int32)
            counts = self._try(ctx.shard_verify_send, fn, cnt, sid.data_ptr(), n, tmp.data_ptr(), send.data_ptr(), perm.data_ptr(), W, default=zeros)
            t0 = self._tick("query_verify_addrs_route", t0)
            req, rcounts = self.comm.a2a_var(send[:n * cnt], counts, out=self._out_buf("v_req"))
            hit = self._buf("v_hit", max(req.numel(), 1))
            self.comm.sync()
            t0 = self._tick("query_verify_all_to_all", t0)
            self._try(ctx.shard_probe, req.data_ptr(), req.numel(), hit.data_ptr())
            t0 = self._tick("query_verify_probe", t0)
            back, _ = self.comm.a2a_var(hit[:req.numel()], rcounts, out=self._out_buf("v_back"))
            self.comm.sync()
            t0 = self._tick("query_verify_all_to_all", t0)
            perm_ptr = perm.data_ptr() if W > 1 else 0  # one rank: natural order
            if ri + 1 == len(rounds):  # the last round marks what passed
                m = self._try(ctx.shard_finish, sid.data_ptr(), n, cnt, back.data_ptr(), perm_ptr)
                trace.append(m)
                t0 = self._tick("query_verify_finish", t0)
            else:
                kept = self._buf("v_kept%d" % (ri & 1), max(n, 1) * 8).view(torch.int64)
                m = self._try(ctx.shard_select, sid.data_ptr(), n, cnt, back.data_ptr(), perm_ptr, kept.data_ptr())
                sid = kept[:m]
                trace.append(m)
                t0 = self._tick("query_verify_select", t0)
        if not rounds:  # a single hash function: the first probe was the only one
            self._try(ctx.shard_mark, sid.data_ptr(), sid.numel())
            t0 = self._tick("query_verify_finish", t0)
        self.comm.sync()
        return trace

    def _verify_unfused(self, b, n, t0):
        """The same protocol through the individual calls (tpc_shard_survivors, _survivor_sources, _route, _permute64, _verify_addrs,
        _select, _mark) -- what a host written against them does; kept so that both forms stay under the parity tests."""
        torch, ctx, W = self.torch, self.ctx, self.world
        zeros = [0] * W
        self.comm.phase = "query batch %d: survivors home" % b
        sid = torch.empty(n, dtype=torch.int64, device=self.device)
        self._try(ctx.shard_survivors, sid.data_ptr())
        src = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        self._try(ctx.shard_survivor_sources, sid.data_ptr(), n, src.data_ptr())
        perm = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        counts = self._try(ctx.shard_route, src.data_ptr(), n, perm.data_ptr(), W, default=zeros)
        send = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        self._try(ctx.shard_permute64, sid.data_ptr(), perm.data_ptr(), n, send.data_ptr())
        sid, _ = self.comm.a2a_var(send[:n].contiguous(), counts)
        sid = sid.contiguous()
        self.comm.sync()
        t0 = self._tick("query_survivors_home", t0)
        trace = [n]
        for fn, cnt in self._verify_rounds(n):
            self.comm.phase = "query batch %d: probes of functions %d..%d" % (b, fn, fn + cnt - 1)
            n = sid.numel()
            addr = torch.empty(n * cnt, dtype=torch.int64, device=self.device)
            owner = torch.empty(n * cnt, dtype=torch.int32, device=self.device)
            self._try(ctx.shard_verify_addrs, fn, cnt, sid.data_ptr(), n, addr.data_ptr(), owner.data_ptr())
            perm = torch.empty(n * cnt, dtype=torch.int32, device=self.device)
            counts = self._try(ctx.shard_route, owner.data_ptr(), n * cnt, perm.data_ptr(), W, default=zeros)
            send = torch.empty(n * cnt, dtype=torch.int64, device=self.device)
            self._try(ctx.shard_permute64, addr.data_ptr(), perm.data_ptr(), n * cnt, send.data_ptr())
            t0 = self._tick("query_verify_addrs_route", t0)
            req, rcounts = self.comm.a2a_var(send, counts)
            hit = torch.empty(req.numel(), dtype=torch.uint8, device=self.device)
            self.comm.sync()
            t0 = self._tick("query_verify_all_to_all", t0)
            self._try(ctx.shard_probe, req.data_ptr(), req.numel(), hit.data_ptr())
            t0 = self._tick("query_verify_probe", t0)
            back, _ = self.comm.a2a_var(hit, rcounts)
            back = back.contiguous()
            kept = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
            self.comm.sync()
            t0 = self._tick("query_verify_all_to_all", t0)
            m = self._try(ctx.shard_select, sid.data_ptr(), n, cnt, back.data_ptr(), perm.data_ptr(), kept.data_ptr())
            sid = kept[:m].contiguous()
            trace.append(sid.numel())
            t0 = self._tick("query_verify_select", t0)
        self.comm.sync()
        self._try(ctx.shard_mark, sid.data_ptr(), sid.numel())
        self._tick("query_verify_finish", t0)
        return trace

    def query(self, lo=0, hi=None, union=True):
        """union = False: every rank keeps only the marks of the positions it hashed (for the key-sharded second pass)."""
        self._pass_begins()
        try:
            return self._query(lo, hi, union)
        except _ListOverflow as e:
            self._relax(e)
            return self._query(lo, hi, union)  # (marks of the failed attempt are marks of this round: setting them again changes nothing)

    def _query(self, lo=0, hi=None, union=True):
        torch, ctx, W = self.torch, self.ctx, self.world
        self.comm.phase = "query plan"
        geom = self._try(ctx.shard_plan, QUERY, lo, hi, default=None)
        self.comm.agree()
        survivors = []
        zeros = [0] * W
        for b in range(geom["batches"]):
            x = self._exchange(QUERY, geom, b, lo, hi)
            t0 = time.perf_counter()
            n = self._apply(QUERY, b, x)
            t0 = self._tick("query_apply", t0)
            trace = (self._verify if self.fused else self._verify_unfused)(b, n, t0)
            survivors.append(trace)
            if len(trace) > 2 and not hasattr(self, "_fn1_pass_rate"):  # a lazy batch was measured: survivors of function 1 / first-probe survivors, all ranks
                tot = self.comm.sum_ints([trace[1], trace[0]])  # (sums over the ranks, as host/multigpu.cpp: both hosts take the same decision on the same input)
                self._fn1_pass_rate = tot[0] / max(tot[1], 1)
        self.stats["survivors"] = survivors
        if self.periodic:
            t0 = time.perf_counter()
            self._try(ctx.shard_periodic_copy)
            self._tick("query_periodic_copy", t0)
        if not union:
            return geom
        t0 = time.perf_counter()
        # OR all-reduce of the candidate masks by word ranges: rank r folds chunk r of every rank's mask, the folded
        # chunks are all-gathered (RCCL has no bitwise reduction; an all_gather of whole masks moved W mask sizes per rank)
        words = ctx.mask_words()
        chunk = (words + W - 1) // W
        self.comm.phase = "mask union"
        mine = torch.empty(W * chunk, dtype=torch.int32, device=self.device)
        self._try(ctx.mask_export_padded, mine.data_ptr(), W * chunk)
        self.comm.agree()
        parts = self.comm.a2a_equal(mine).contiguous()
        folded = torch.empty(chunk, dtype=torch.int32, device=self.device)
        self.comm.sync()
        self._try(ctx.mask_or_blocks, parts.data_ptr(), W, chunk, folded.data_ptr())
        allm = self.comm.all_gather(folded).contiguous()
        self.comm.sync()
        self._try(ctx.mask_import, allm.data_ptr())
        self._tick("mask_union", t0)
        self.stats["survivors"] = survivors
        return geom

    def round(self, lo=0, hi=None, abundance=(1 << 64) - 1):
        self.insert(lo, hi)
        self.query(lo, hi)
        return self.ctx.pass2_filter(abundance)

    def round_sharded_pass2(self, lo=0, hi=None, abundance=(1 << 64) - 1, records=False):
        """The round with the exact filter's table sharded by key hash ("shard the key table by key hash, all-to-all
        for marked positions only"): no mask union; every rank lists the positions it marked and the owner rank of each
        position's canonical key (tpc_pass2_mark_owners), the positions travel to the owners (8 bytes each), and each owner
        runs the exact filter over what it received -- all occurrences of its keys, so the reference's (prev, next) rule and
        abundance cut see the same sets.  The per-rank junction keys are then all-gathered (HipBackend.union_keys_on_device).
        records = True: (key, prev | next) records travel instead of positions, the owner reads no text: every rank can hold
        just its chunk of the packed text (option text_window)."""
        torch, ctx, W = self.torch, self.ctx, self.world
        self.insert(lo, hi)
        self.query(lo, hi, union=False)
        t0 = time.perf_counter()
        self.comm.phase = "second pass: marked positions to the key owners"
        zeros = [0] * W
        none = {"true": 0, "false": 0, "table": 0}
        if W == 1 and not os.environ.get("TPC_PASS2_ROUTED"):  # one rank owns every key and holds the whole text: nothing to route (as _verify does for the probes)
            st = self._try(ctx.pass2_filter, abundance, default=dict(none))
            st["marks"] = self._try(ctx.stat, "round_marks")
            self.stats["pass2_positions_received"] = st["marks"]
            self._tick("pass2_sharded", t0)
            return st
        n = self._try(ctx.pass2_marks)
        if records:
            rw = ctx.key_words() + 1
            rec = torch.empty(max(n, 1) * rw, dtype=torch.int64, device=self.device)
            owner = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
            # combine before routing: one record per DISTINCT key of this rank's marks (its own exact filter first), not one per mark
            agg = os.environ.get("TPC_PASS2_AGGREGATE", "1") != "0"
            nr = n
            if agg:
                nr = self._try(ctx.pass2_aggregate_records, W, rec.data_ptr(), owner.data_ptr(), abundance, default=0)
            else:
                self._try(ctx.pass2_mark_records, W, rec.data_ptr(), owner.data_ptr())
            perm = torch.empty(max(nr, 1), dtype=torch.int32, device=self.device)
            counts = self._try(ctx.shard_route, owner.data_ptr(), nr, perm.data_ptr(), W, default=zeros)
            send = torch.empty(max(nr, 1) * rw, dtype=torch.int64, device=self.device)
            self._try(ctx.shard_permute_rows, rec.data_ptr(), perm.data_ptr(), nr, rw, send.data_ptr())
            recv, rcl = self.comm.a2a_var(send[:nr * rw].contiguous(), [c * rw for c in counts])
            recv = recv.contiguous()
            self.comm.sync()
            self.stats["pass2_recv_bytes_busiest_peer"] = 8 * max([c for r, c in enumerate(rcl) if r != self.rank] or [0])
            self.stats["pass2_records_sent"] = nr
            st = self._try(ctx.pass2_filter_aggregated if agg else ctx.pass2_filter_records, recv.data_ptr(), recv.numel() // rw, abundance, default=dict(none))
            st["marks"] = n
            self.stats["pass2_positions_received"] = recv.numel() // rw
            self._tick("pass2_sharded", t0)
            return st
        pos = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        owner = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        self._try(ctx.pass2_mark_owners, W, pos.data_ptr(), owner.data_ptr())
        perm = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        counts = self._try(ctx.shard_route, owner.data_ptr(), n, perm.data_ptr(), W, default=zeros)
        send = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        self._try(ctx.shard_permute64, pos.data_ptr(), perm.data_ptr(), n, send.data_ptr())
        recv, _ = self.comm.a2a_var(send[:n].contiguous(), counts)
        recv = recv.contiguous()
        self.comm.sync()
        st = self._try(ctx.pass2_filter_positions, recv.data_ptr(), recv.numel(), abundance, default=dict(none))
        st["marks"] = n
        self.stats["pass2_positions_received"] = recv.numel()
        self._tick("pass2_sharded", t0)
        return st


class Combined(AddressSharded):
    """The first pass with the filter REPLICATED through set-bit lists ("combine before routing"; include/twopaco_hip.h: tpc_combine_*,
    csrc/tpc_combine.hip).  The reference's threads share one ConcurrentBitVector for free (concurrentbitvector.cpp:31-45, MergeOr
    :115-122); AddressSharded pays 4 bytes per insert address and 8 per query probe to imitate that over links.  Here every rank
      insert   runs the one-GPU insert over ITS chunk of the text up to the level-2 regions (tpc_pass1_insert on a context with option
               replicate_filter), builds every filter slice in LDS and exports only the slice's SET BITS (tpc_combine_export: 2 bytes per
               distinct bit, owner-major blocks)
      move     the bytes model (tpc_combine_choose, the same arithmetic in both hosts) picks: an all-gather of the exports (two ranks),
               or a reduce-scatter by owner -- the north star's "route hits to owning shards", now deduplicated -- a merge on the owner
               (tpc_combine_merge) and an all-gather of the merged lists; or, when the insert could not stay in its regions (several
               batches, three levels) or the lists would be larger, an OR all-reduce of the dense filters by word ranges
      query    imports the lists (tpc_combine_import) and runs the one-GPU query over its chunk: the first lookup builds every slice
               from the lists, writes the whole filter locally and tests the chunk's probes -- no probe, survivor or answer crosses a link.
    The marks stay on the rank that hashed them (second pass sharded by key hash as in AddressSharded) or are OR-reduced (union = True).
    The context must have been configured with option replicate_filter = 1 before tpc_shard_config / tpc_set_params."""

    def __init__(self, ctx, dist, device, configure=True, mode=None):
        ctx.set_option("replicate_filter", 1)
        super().__init__(ctx, dist, device, compact=False, configure=configure)
        # "auto" (the bytes model), "gather" (all-gather of exports), "scatter" (reduce-scatter + all-gather), "dense"
        self.mode = mode or os.environ.get("TPC_COMBINE", "auto")
        self.stats["combine"] = {}

    # ---- insert: local insert, export, exchange, import
    def insert(self, lo=0, hi=None):
        torch, ctx, W, comm = self.torch, self.ctx, self.world, self.comm
        comm.phase = "combined insert"
        t0 = time.perf_counter()
        self._try(ctx.filter_reset)
        self._try(ctx.pass1_insert, lo, hi, False)
        t0 = self._tick("insert_local", t0)
        if W == 1:  # one rank: the one-GPU pass as it is (the insert waits in its regions for the query's lookup)
            self.stats["combine"].update(mode="one rank: nothing to exchange", exchange_bytes_received=0)
            return {"mode": "one rank"}
        info = self._try(ctx.combine_info, W, default=None) or {"sparse": 0}
        if not info["sparse"]:
            self._begin_query(lo, hi)  # (the insert went to the dense filter: the query's hash and binning run under the filters' OR-reduce)
        # every rank must take the same road: sparse lists only if every rank's insert stayed in its regions
        dense = comm.max_ints([0 if info["sparse"] else 1])[0] == 1 or self.mode == "dense"
        st = self.stats["combine"]
        if not dense:
            n_win, spd = info["windows"], info["slices"] // W  # windows per slice, slices per destination
            cap = info["cap_units"]
            payload = self._buf("c_payload", W * cap * 16)
            cdir = self._buf("c_dir", info["slices"] * n_win * 8)
            units = self._try(ctx.combine_export, W, payload.data_ptr(), cap, cdir.data_ptr(), default=[0] * W)
            t0 = self._tick("insert_export", t0)
            self._begin_query(lo, hi)  # the query's level-1 hash and level-2 binning do not read the filter: they run while the lists travel
            # everybody's block sizes: [source][destination] units
            comm.agree()
            allu = comm.all_gather(torch.tensor(units, dtype=torch.int64, device=self.device)).cpu().tolist()
            mean_units = sum(sum(r) for r in allu) // W
            from . import capi
            mode, model = capi.combine_choose(W, ctx.L, mean_units)  # (pure arithmetic in the library: both hosts take the same road)
            if self.mode == "gather":
                mode = 1
            elif self.mode == "scatter":
                mode = 2
            if W == 1:
                mode = 1
            st.update(mode={1: "all-gather of the exports", 2: "reduce-scatter by owner + all-gather of the merged lists", 3: "dense OR all-reduce"}[mode],
                      model_bytes_received={"gather": model[0], "scatter": model[1], "dense": model[2]}, export_units=units, export_bytes=16 * sum(units))
            dense = mode == 3
            if dense:  # the export took the insert out of its regions: this rank's own lists bring it back, to be applied to its dense filter
                self._try(ctx.combine_import, W, W, payload.data_ptr(), [d * cap for d in range(W)], cdir.data_ptr(), spd * n_win)
        if dense:
            st.setdefault("mode", "dense OR all-reduce")
            self._dense_reduce()
            self._tick("insert_exchange", t0)
            return {"mode": st["mode"]}
        pay64 = payload.view(torch.int64)  # (2 x int64 per unit)
        dir64 = cdir.view(torch.int64)
        if mode == 1:
            # all-gather of every rank's W blocks (used prefixes back to back) and of the directories
            comm.phase = "combined insert: all-gather of the exports"
            mine = torch.cat([pay64[2 * d * cap:2 * (d * cap + units[d])] for d in range(W)]) if sum(units) else pay64[:0]
            most = max(sum(r) for r in allu)
            pad = self._buf("c_send", max(most, 1) * 16).view(torch.int64)
            pad[:mine.numel()] = mine
            allp = comm.all_gather(pad[:2 * most]) if W > 1 else pad[:2 * most].unsqueeze(0)
            alld = comm.all_gather(dir64) if W > 1 else dir64.unsqueeze(0)
            self._keep = (allp, alld)  # the query's lookup reads them
            base = []
            for r in range(W):
                o = r * most
                for d in range(W):
                    base.append(o)
                    o += allu[r][d]
            comm.sync()
            self._try(ctx.combine_import, W * W, W, allp.data_ptr(), base, alld.data_ptr(), spd * n_win)
            st["exchange_bytes_received"] = 16 * (sum(sum(r) for r in allu) - sum(units)) + 8 * (W - 1) * dir64.numel()
            st["recv_bytes_busiest_peer"] = {"all-gather of the exports": 16 * max(sum(allu[r]) for r in range(W) if r != self.rank or W == 1) + 8 * dir64.numel()}
        else:
            # reduce-scatter: block d and its directory to rank d; the owner merges; all-gather of the merged lists
            comm.phase = "combined insert: reduce-scatter of the exports"
            send = torch.cat([pay64[2 * d * cap:2 * (d * cap + units[d])] for d in range(W)]) if sum(units) else pay64[:0]
            recv, rc = comm.a2a_var(send.contiguous(), [2 * u for u in units], out=self._out_buf("c_recv"))
            rdir = comm.a2a_equal(dir64.contiguous())
            base, o = [], 0
            for s in range(W):
                base.append(o)
                o += rc[s] // 2
            comm.sync()
            t0 = self._tick("insert_exchange", t0)
            merged = self._buf("c_merged", max(o, 1) * 16)
            mdir = self._buf("c_mdir", spd * n_win * 8)
            recv_ptr = recv.data_ptr() if recv.numel() else merged.data_ptr()
            mu = self._try(ctx.combine_merge, W, recv_ptr, base, rdir.data_ptr(), merged.data_ptr(), max(o, 1), mdir.data_ptr())
            t0 = self._tick("insert_merge", t0)
            comm.phase = "combined insert: all-gather of the merged lists"
            comm.agree()
            allm = comm.all_gather(torch.tensor([mu], dtype=torch.int64, device=self.device)).cpu().view(-1).tolist()
            most = max(max(allm), 1)
            padm = self._buf("c_mpad", most * 16).view(torch.int64)  # (the largest merged block sets the all-gather's block size)
            padm[:2 * mu] = merged.view(torch.int64)[:2 * mu]
            allp = comm.all_gather(padm)
            alld = comm.all_gather(mdir.view(torch.int64))
            self._keep = (allp, alld)
            comm.sync()
            self._try(ctx.combine_import, W, W, allp.data_ptr(), [r * most for r in range(W)], alld.data_ptr(), spd * n_win)
            st["merged_units"] = allm
            st["exchange_bytes_received"] = 16 * (o - units[self.rank]) + 16 * (sum(allm) - mu) + 8 * (W - 1) * (rdir.numel() // W + mdir.numel() // 8)
            others = [r for r in range(W) if r != self.rank] or [self.rank]
            st["recv_bytes_busiest_peer"] = {"reduce-scatter of the exports": 16 * max(allu[r][self.rank] for r in others) + 8 * (rdir.numel() // W),
                                             "all-gather of the merged lists": 16 * max(allm[r] for r in others) + mdir.numel()}
        self._tick("insert_exchange", t0)
        return {"mode": st["mode"]}

    def _begin_query(self, lo, hi):
        """tpc_pass1_query_begin: the filter-independent part of the round's query, enqueued before the exchange (TPC_COMBINE_OVERLAP=0: not).
        With the ranks of an emulated run taking turns on one device (TPC_DIST_SERIALIZE) the call waits for the kernels, so that its
        time in call_ms is theirs and nobody else's calls run beside them: what the link model may hide behind the wire."""
        if os.environ.get("TPC_COMBINE_OVERLAP", "1") == "0":
            return
        ctx, torch = self.ctx, self.torch

        def pass1_query_begin():
            ctx.pass1_query_begin(lo, hi)
            if _serial_lock() is not None:
                torch.cuda.synchronize()
        self._try(pass1_query_begin)

    def _dense_reduce(self):
        """OR all-reduce of the ranks' dense filters by word ranges (the mask union's scheme on 2^L / 8 bytes)."""
        torch, ctx, W, comm = self.torch, self.ctx, self.world, self.comm
        if W == 1:
            return
        comm.phase = "combined insert: dense OR all-reduce"
        words = ctx.filter_words() - 1  # 2^L / 32: a multiple of W (the last word is the reference's spare)
        chunk = words // W
        mine = self._buf("c_dense", words * 4).view(torch.int32)
        self._try(ctx.filter_copy_out, 0, words, mine.data_ptr())
        comm.agree()
        parts = comm.a2a_equal(mine).contiguous()
        folded = self._buf("c_fold", chunk * 4).view(torch.int32)
        comm.sync()
        self._try(ctx.mask_or_blocks, parts.data_ptr(), W, chunk, folded.data_ptr())
        allm = comm.all_gather(folded).contiguous()
        comm.sync()
        self._try(ctx.filter_copy_in, 0, words, allm.data_ptr())
        self.stats["combine"]["exchange_bytes_received"] = 2 * (W - 1) * chunk * 4
        self.stats["combine"]["recv_bytes_busiest_peer"] = {"dense reduce-scatter": chunk * 4, "dense all-gather": chunk * 4}

    # ---- query: entirely local
    def query(self, lo=0, hi=None, union=True):
        torch, ctx, W = self.torch, self.ctx, self.world
        self.comm.phase = "combined query"
        t0 = time.perf_counter()
        n = self._try(ctx.pass1_query, lo, hi)
        self.comm.agree()
        self._keep = None
        self.stats["local_marks"] = n
        self.stats["survivors"] = []
        t0 = self._tick("query_local", t0)
        if union and W > 1:
            words = ctx.mask_words()
            chunk = (words + W - 1) // W
            self.comm.phase = "mask union"
            mine = torch.empty(W * chunk, dtype=torch.int32, device=self.device)
            self._try(ctx.mask_export_padded, mine.data_ptr(), W * chunk)
            self.comm.agree()
            parts = self.comm.a2a_equal(mine).contiguous()
            folded = torch.empty(chunk, dtype=torch.int32, device=self.device)
            self.comm.sync()
            self._try(ctx.mask_or_blocks, parts.data_ptr(), W, chunk, folded.data_ptr())
            allm = self.comm.all_gather(folded).contiguous()
            self.comm.sync()
            self._try(ctx.mask_import, allm.data_ptr())
            self._tick("mask_union", t0)
        return {"batches": ctx.stat("query_batches")}


def address_sharded_step(sharded, abundance=(1 << 64) - 1, fetch=False, sharded_pass2=False):
    # sharded_pass2: False, True / "positions", or "records" (text-free: see AddressSharded.round_sharded_pass2)
    """One whole enumeration with the filter sharded by address.
    sharded_pass2 = False: after the first pass every rank holds the full candidate mask, so pass 2, the key sort and the id
    lookup run replicated and every rank ends with the complete result (rank 0 writes it).
    sharded_pass2 = True: the marks stay where they were found, the exact filter's table is sharded by key hash
    (round_sharded_pass2), the junction keys are all-gathered and every rank looks up the ids of ITS marked positions; the
    records of all ranks together (merge_records) are the result."""
    ctx = sharded.ctx
    ctx.run_begin()
    if sharded_pass2:
        st = sharded.round_sharded_pass2(0, None, abundance, records=sharded_pass2 == "records")
        if not hasattr(sharded, "_keys_backend"):
            sharded._keys_backend = HipBackend(ctx)
            sharded._keys_backend._comm = sharded.comm
            sharded._keys_backend._cap = 1 << 17
        sharded._keys_backend.union_keys_on_device(sharded.comm.dist)
    else:
        st = sharded.round(0, None, abundance)
    st["junctions"] = sharded._timed(ctx.junctions_finalize)
    st["n_marked"], st["n_valid"] = sharded._timed(ctx.emit)
    if not sharded_pass2:
        st["marks"] = st["n_marked"]
    if fetch:
        st["g"], st["ids"] = ctx.emit_fetch()
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


_LINE_FD = None  # the process's real stdout while a bench run keeps descriptor 1 on stderr (bench_main)


def bench_main(args, rank, world, local_rank, backend_factory=None, golden=None, cpu_baseline=None, e2e=None):
    """Fail-fast wrapper: any exception on any rank ends THAT process with a non-zero exit code at once (the phase it was in on
    stderr) -- under torch.distributed.run the agent then stops the other ranks; ranks blocked in a collective are ended by
    their own watchdog (PhaseWatchdog).  No line is printed by a run that did not complete on every rank."""
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner there when its first communicator comes up (at the first
    # collective, not at init_process_group), and whatever else a library writes to file descriptor 1 would land beside the line too: for
    # the length of the run descriptor 1 points at stderr, and the line goes to the real stdout (_LINE_FD) at the very end.
    global _LINE_FD
    sys.stdout.flush()
    if _LINE_FD is None:
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)
    try:
        return _bench_main(args, rank, world, local_rank, backend_factory, golden, cpu_baseline, e2e)
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        sys.stderr.write("twopaco_amd.dist: rank %d failed (%s: %s): exiting with code 4\n" % (rank, type(e).__name__, e))
        sys.stderr.flush()
        os._exit(4)  # not sys.exit: interpreter teardown would try to destroy a process group whose peers are gone


def _bench_main(args, rank, world, local_rank, backend_factory=None, golden=None, cpu_baseline=None, e2e=None):
    """bench.py --gpus N under torch.distributed.run: strong scaling of the same workload, one process per GPU over RCCL.

    The north star's decomposition -- the Bloom filter sharded by bit address, all-to-all of the level-1 regions per pass -- is timed at
    every N that is a power of two; the vertex-hash-range decomposition (no data-path exchange) is timed as well, the
    line's value is the FASTER of the two and the other one keeps its full record under its own name ("address" / "ranges"):
    routing every hash hit to its owner pays where the links outnumber the work, which by the link model is not at two ranks
    (DESIGN.md section 6).  `--decomposition address|ranges` pins one.
    golden: the reference's counters for this workload (tests/golden/cases.json); a result that differs ends the run with a
    non-zero exit code instead of a line.  cpu_baseline(recs, params) -> dict: timed on rank 0 after the timed region.
    backend_factory(args, rank, world) -> (backend, n_kmers, params, description): what tests/ use to drive the launch /
    rendezvous / timing / reporting path over gloo without a GPU (such a line says "backend": "injected")."""
    import torch
    import torch.distributed as dist

    from . import capi, synth

    backend = os.environ.get("TPC_DIST_BACKEND", "nccl")  # "gloo": several ranks on one GPU (testing only)
    injected = backend_factory is not None
    if getattr(args, "gpus", world) != world:  # the launcher's world must be the --gpus the line will claim
        raise RuntimeError("bench.py --gpus %d was launched with WORLD_SIZE = %d" % (args.gpus, world))
    pow2 = world & (world - 1) == 0
    decomposition = getattr(args, "decomposition", "auto")
    if decomposition == "auto":
        decomposition = "address" if pow2 and not injected else "ranges"
    address = decomposition == "address"
    also_ranges = address and getattr(args, "decomposition", "auto") == "auto"  # both are timed at every N; the line's value is the address-sharded filter's, the ranges keep their record under "ranges"
    ctx = ctx_r = None
    sharded2 = False
    combined = False
    recs = None
    pass2 = "replicated"
    steps = {}
    if injected:
        be, n_kmers, p, workload_desc = backend_factory(args, rank, world)
        dist.init_process_group(backend)
        if address:
            raise RuntimeError("the injected test backend only drives the ranges decomposition")
        steps["ranges"] = lambda: sharded_step(be, dist, p["L"])
    else:
        ngpu = torch.cuda.device_count()
        device = local_rank % max(ngpu, 1)
        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
        recs, p = synth.workload(args.workload, scale=args.scale)
        n_kmers = synth.n_kmers(recs, p["k"])
        workload_desc = "%s: %d genomes x %d bp E. coli-like synthetic (twopaco_amd/synth.py), k=%d q=%d f=%d" % (
            args.workload, len(recs), recs[0].size, p["k"], p["q"], p["L"])
        text = capi.PackedText.from_codes(recs)
        ctx = capi.Context(device)
        # second pass of the address decomposition (TPC_PASS2): "records" (default) = exact filter's table sharded by key hash,
        # (key, prev | next) records travel, every rank holds only its chunk of the text; "positions" = the same with 8-byte
        # positions and the whole text on every rank; "replicated" = union of the masks, every rank runs the whole second pass
        pass2 = os.environ.get("TPC_PASS2", "records") if address else "replicated"
        if pass2 not in ("records", "positions", "replicated"):
            raise RuntimeError("TPC_PASS2 must be records, positions or replicated")
        sharded2 = pass2 != "replicated"
        # how the ranks share the filter (the rule of host/vertexenumerator.cpp): combined while the whole filter fits a GPU with room to
        # spare and the passes have two levels -- every rank keeps the filter, only the set bits of every slice travel (Combined) -- else
        # (or TPC_MULTIGPU=entries) every hash hit of both passes is routed to the owner of its slice (AddressSharded)
        combined = address and p["L"] <= 38 and os.environ.get("TPC_MULTIGPU", os.environ.get("TWOPACO_MULTIGPU", "combined")) != "entries"
        if combined:
            ctx.set_option("replicate_filter", 1)
        if address and pass2 == "records":
            ctx.set_option("text_window", 1)
            ctx.shard_config(rank, world)  # before the upload: the window depends on it
        ctx.set_params(p["k"], p["L"], p["q"], capi.seed_table(p["q"], p["L"], seed=20240229))
        ctx.seq_upload(text)
        if address:
            sh = (Combined if combined else AddressSharded)(ctx, dist, torch.device("cuda", device), configure=pass2 != "records")
            steps["address"] = lambda: address_sharded_step(sh, sharded_pass2=pass2 if sharded2 else False)
        if not address:
            be = HipBackend(ctx)
            steps["ranges"] = lambda: sharded_step(be, dist, p["L"])
        elif also_ranges:  # its own context: a whole filter and the whole text on every rank
            ctx_r = capi.Context(device)
            ctx_r.set_params(p["k"], p["L"], p["q"], capi.seed_table(p["q"], p["L"], seed=20240229))
            ctx_r.seq_upload(text)
            be = HipBackend(ctx_r)
            steps["ranges"] = lambda: sharded_step(be, dist, p["L"])
    if dist.get_world_size() != world:
        raise RuntimeError("process group of %d ranks, expected %d" % (dist.get_world_size(), world))
    dev = _dev(dist)
    wd = PhaseWatchdog(rank)  # the bench's own barriers and reductions (the exchanges inside a step have theirs in _Comm)
    names = ["filter_reset", "insert", "query", "compact", "filter2", "scan2", "sort", "emit", "shard_hash", "shard_apply"]

    setup_scale = float(os.environ.get("TPC_DIST_SETUP_TIMEOUT_S", str(5 * wd.timeout))) / max(wd.timeout, 1e-9) if wd.timeout > 0 else 1.0
    first = [True]

    def guarded(name, fn, *a, **kw):
        # the first guarded collective also waits for the slowest rank's setup (synthesis of the workload, context, the lazy communicator)
        wd.enter(name, setup_scale if first[0] else 1.0)
        first[0] = False
        try:
            return fn(*a, **kw)
        finally:
            wd.leave()

    def timed(which):
        step = steps[which]
        kctx = ctx if which == "address" or not address else ctx_r
        for _ in range(args.warmup):
            step()
        kms = {n: 0.0 for n in names}
        if which == "address":
            sh.t.clear()
            sh.call_ms.clear()
            sh.stats["region_bytes_sent"] = 0
            sh.stats["overflow_entries"] = 0
            moved0 = sh.comm.bytes_moved
        guarded("bench: barrier before the timed steps", dist.barrier)
        if ctx is not None:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        per_step_calls = []
        for _ in range(args.steps):
            if which == "address":
                sh.call_ms.clear()
            st = step()
            if which == "address":
                per_step_calls.append(dict(sh.call_ms))
            for n in names:
                if kctx is not None:
                    kms[n] += max(kctx.kernel_ms(n), 0.0) / args.steps
        if ctx is not None:
            torch.cuda.synchronize()
        guarded("bench: barrier after the timed steps", dist.barrier)
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        guarded("bench: max of the step times", dist.all_reduce, dt, op=dist.ReduceOp.MAX)
        tot = torch.tensor([st["n_valid"], st["marks"]], dtype=torch.int64, device=dev)
        if which == "ranges" or sharded2:  # (address-sharded with the replicated second pass: every rank already holds the whole result)
            guarded("bench: sum of the per-rank counters", dist.all_reduce, tot)
        r = {"dt": float(dt.item()), "kms": kms, "result": {"candidate_marks": int(tot[1].item()), "junctions": st["junctions"], "junction_occurrences": int(tot[0].item())}}
        if which == "address":
            r["phase_ms"] = {k: v * 1e3 / args.steps for k, v in sh.t.items()}
            r["exchange_bytes"] = (sh.comm.bytes_moved - moved0) // args.steps
            a2a_s = sum(v for k, v in sh.t.items() if k.endswith("_all_to_all")) / args.steps
            r["region_bytes_sent"] = sh.stats.get("region_bytes_sent", 0) // args.steps
            # level-1 regions this rank SENT per second of all-to-all wall time (the exchange waits for every peer: a rate per
            # rank, what the link model of DESIGN.md section 5.2 assumed as 50 GB/s per link and direction)
            r["all_to_all_GBs"] = r["region_bytes_sent"] / max(a2a_s, 1e-9) / 1e9  # (region_bytes_sent: what left this rank)
            r["overflow_entries"] = sh.stats.get("overflow_entries", 0) // args.steps
            r["exchange"] = "packed to exact sizes (tpc_shard_pack)" if sh.compact else "equal blocks of tight regions, own block in place"

            r["survivors"] = sh.stats.get("survivors")
            # the link model's inputs: this rank's library-call time (alone on its device when the ranks take turns: TPC_DIST_SERIALIZE)
            # and, per exchange phase, the bytes from its busiest peer; the slowest rank's figures make the line
            # per library call the BEST of the timed steps (an emulated run shares one device and this host's cores among W processes:
            # a stalled call is not the design's; on real ranks the steps agree and best = mean)
            r["call_ms"] = {k: min(c.get(k, 0.0) for c in per_step_calls) for k in per_step_calls[0]}
            r["combine"] = sh.stats.get("combine")
            peer = dict((sh.stats.get("combine") or {}).get("recv_bytes_busiest_peer") or {})
            if sh.stats.get("pass2_recv_bytes_busiest_peer"):
                peer["second pass: records to the key owners"] = sh.stats["pass2_recv_bytes_busiest_peer"]
            mine = torch.tensor([sum(r["call_ms"].values()), r["call_ms"].get("pass1_query_begin", 0.0)] + [float(v) for v in peer.values()], dtype=torch.float64, device=dev)
            guarded("bench: max of the ranks' model inputs", dist.all_reduce, mine, op=dist.ReduceOp.MAX)
            vals = mine.cpu().tolist()
            # (every rank's own sum as well: on an emulated run -- W processes sharing one device and this host's cores -- one rank's
            #  sum can carry a stall that is not the design's; the line keeps the slowest and the median rank)
            own = torch.tensor([sum(r["call_ms"].values())], dtype=torch.float64, device=dev)
            allc = [torch.zeros_like(own) for _ in range(world)]
            guarded("bench: all ranks' library-call sums", dist.all_gather, allc, own)
            per_rank = sorted(float(x.item()) for x in allc)
            r["model"] = link_model(world, per_rank[len(per_rank) // 2], dict(zip(peer.keys(), vals[2:])), hidden_ms=vals[1])
            r["model"]["compute_ms_per_rank_sorted"] = per_rank
            r["model"]["compute_ms_is"] = "the MEDIAN rank's library-call sum (slowest rank: %.2f ms)" % per_rank[-1]
        return r

    head = timed("address" if address else "ranges")
    second = timed("ranges") if also_ranges else None
    ok = True
    if golden:  # the reference's own counters (VE.h:384-388, 413): every decomposition must reproduce them
        for which, r in (("address" if address else "ranges", head), ("ranges", second)):
            if r is None:
                continue
            res = r["result"]
            want = {"junctions": golden["distinct"], "junction_occurrences": golden["true_marks"]}
            if which == "address":  # one global filter: the candidate marks are the reference's too (every range has its own filter)
                want["candidate_marks"] = golden["rounds"][0]["marks"]
            bad = {k: (res[k], v) for k, v in want.items() if res[k] != v}
            if bad:
                ok = False
                if rank == 0:
                    print("bench: %s decomposition: result differs from the reference golden (got, want): %r" % (which, bad), file=sys.stderr)
    dt = head["dt"]
    # Every rank gives its device back -- contexts closed, process group left -- before rank 0 runs the end-to-end leg: the
    # `twopaco --gpus N` child process takes all N devices itself.  After this point no collective is issued: ranks other than 0
    # leave (exit code 3 when the golden check failed), rank 0 finishes the line alone, so nobody waits in a barrier while a CPU
    # baseline of a minute runs.
    rccl_version = None
    try:
        if not injected and backend == "nccl":
            rccl_version = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        pass
    for c in (ctx, ctx_r):
        if c is not None:
            c.close()
    if ctx is not None:
        torch.cuda.empty_cache()
    guarded("bench: barrier before leaving the process group", dist.barrier)
    dist.destroy_process_group()
    if not ok:
        sys.exit(3)
    if rank != 0:
        return 0
    kms = head["kms"]
    # The first pass of rank 0 (both kernel groups; the exchanges are not in it): time = the phases' wall clock between device synchronisations
    # (address) or the library's event timers (ranges); bytes = what the design moves by construction (insert entries 4 B x 4
    # transfers, query entries 8 B x 4 in the sharded level 2, the packed text, the filter written once by the fused apply + lookup).  No
    # PMC profile exists for the sharded path, so `traffic` is null and `frac` is a DESIGN-byte fraction -- `frac_kind` says so.
    fb = (1 << p["L"]) // 8
    ph = head.get("phase_ms") or {}
    if address and combined:  # a rank's own first-pass library calls (the exchanges are not in them); every rank writes the whole filter once
        cm = head.get("call_ms") or {}
        qms = sum(cm.get(k, 0.0) for k in ("pass1_insert", "combine_export", "combine_merge", "pass1_query"))
        design = (2 * 0.375 + p["q"] * 16 + 6 * 32) * n_kmers / world + fb
    elif address:
        qms = sum(ph.get(k, 0.0) for k in ("insert_hash", "insert_apply", "query_hash", "query_apply", "query_verify_finish"))
        design = (2 * 0.375 + p["q"] * 16 + 6 * 32) * n_kmers / world + fb / world
    else:
        qms = kms["insert"] + kms["query"]
        design = 2 * 0.375 * n_kmers + (p["q"] * 16 + 6 * 32) * n_kmers / world + fb
    qms = max(qms, 1e-9)
    out = {
        "metric": "kmers_hashed_per_sec", "value": n_kmers * args.steps / dt, "unit": "k-mers/s", "n_gpus": world,
        "ranks": world, "backend": "injected" if injected else ("rccl" if backend == "nccl" else backend), "rccl_version": rccl_version,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": workload_desc,
                   "kmers": n_kmers, "filter_bytes": fb, "decomposition": "address" if address else "ranges",
                   "parallelism": (("filter replicated over %d GPUs through set-bit lists combined by bit-address owner (every rank inserts its chunk of the text, only the distinct set bits of every slice travel, the query is local); " % world) if address and combined else
                                   ("filter sharded by bit address over %d GPUs; all_to_all of the level-1 regions per pass, per-function survivor probes; " % world)) +
                                  ({"records": "text sharded too; exact-filter table sharded by key hash ((key, prev|next) records to the key's owner), all_gather of the junction keys",
                                    "positions": "exact-filter table sharded by key hash (8 B per marked position to the key's owner), all_gather of the junction keys",
                                    "replicated": "OR all-reduce of the candidate mask, replicated second pass"}[pass2] if address else "")
                   if address else ("%d vertex-hash ranges, one per GPU (reference rounds run side by side); all-gather of junction keys over RCCL" % world)},
        "junction_occurrences_per_sec": head["result"]["junction_occurrences"] * args.steps / dt,
        "kernel_ms_rank0": kms,
        "roofline": {"bound": "hbm", "kernel": "first pass (insert + query kernel groups) on rank 0, exchanges excluded", "achieved": design / (qms * 1e-3) / 1e9,
                     "peak": 8000.0, "unit": "GB/s", "frac": design / (qms * 1e-3) / 1e9 / 8000.0, "frac_kind": "design bytes (no counter profile of the sharded path)",
                     "traffic": None, "algorithmic_bytes_per_launch": design, "launch_ms": qms} if ctx is not None else None,
        "exchange_bytes_rank0_per_step": head.get("exchange_bytes"),
        "multi_gpu_exchange": (head.get("combine") or {}).get("mode") if address else None,
        "combine_rank0": head.get("combine"),
        "call_ms_rank0_per_step": head.get("call_ms"),
        "model": head.get("model"),
        "region_bytes_sent_rank0_per_step": head.get("region_bytes_sent"),
        "all_to_all_GBs_rank0": head.get("all_to_all_GBs"),
        "region_exchange": head.get("exchange"),
        "overflow_entries_rank0_per_step": head.get("overflow_entries"),
        "phase_ms_rank0_per_step": head.get("phase_ms"),
        "survivors_rank0": head.get("survivors"),
        "collective_timeout_s": wd.timeout,
        "result": head["result"],
        "result_equals_reference_golden": True if golden else None,
        # `value` is ONE decomposition at every N, so that the per-N values form a scaling curve of one design: the filter sharded by bit
        # address (the north star's) unless --decomposition ranges was asked for.  The other one, when it was timed too, keeps its full
        # record under the key "ranges" -- by the link model (README) it is the faster one below eight ranks.
        "headline_decomposition": "address" if address else "ranges",
    }
    if second is not None:
        out["ranges"] = {"value": n_kmers * args.steps / second["dt"], "unit": "k-mers/s", "ms_per_step": second["dt"] / args.steps * 1e3,
                         "decomposition": "%d vertex-hash ranges, one per GPU, no data-path exchange (DESIGN.md section 5.1)" % world,
                         "kernel_ms_rank0": second["kms"], "result": second["result"]}
    # the metric's second half at N GPUs: the C++ host end to end (`twopaco --gpus N`, host/multigpu.cpp: RCCL transport), fresh child
    # processes on the FASTA files, output sha256 == the reference golden.  A failure here is reported IN the line (e2e.error, e2e_failed = true, no end-to-end figure).
    if e2e is not None and recs is not None:
        time.sleep(2.0)  # the other ranks' processes are on their way out: let the driver have their device memory back
        try:
            out["e2e"] = e2e(recs, p, world)
        except Exception as ex:  # noqa: BLE001
            out["e2e"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if "error" in out["e2e"]:
            # The k-mers/s above were measured and checked against the reference's counters by the ranks themselves; the C++ host's
            # leg failing (it has never met more than one real device) must not take that line away -- it is reported IN the line,
            # loudly, with no end-to-end figure.
            sys.stderr.write("bench: end-to-end leg (twopaco --gpus %d) FAILED, no e2e figure in the line: %s\n" % (world, out["e2e"]["error"]))
            out["e2e_junction_occurrences_per_sec"] = None
            out["e2e_wall_s"] = None
        else:
            out["e2e_junction_occurrences_per_sec"] = out["e2e"]["e2e_junction_occurrences_per_sec"]
            out["e2e_wall_s"] = out["e2e"]["e2e_wall_s"]
            # the PRODUCT's host at this N (host/multigpu.cpp: C++, RCCL directly): its own timers, median of the CLI runs above, beside
            # the torch.distributed driver's figures -- the first hardware curve must be readable for the host that ships
            bd = out["e2e"].get("breakdown_ms") or {}
            rounds_ms = bd.get("rounds_ms")
            ph = out["e2e"].get("sharded_first_pass_ms_rank0") or {}
            a2a_ms = sum(v for k, v in ph.items() if k.endswith("all-to-all"))
            sent = out["e2e"].get("region_bytes_sent_rank0")
            out["cxx_host"] = {"what": "twopaco --gpus %d (C++ host, one thread per GPU, RCCL send/recv groups), its own TWOPACO_TIMING figures, median run" % world,
                               "rounds_ms": rounds_ms, "kmers_per_sec": (n_kmers / (rounds_ms * 1e-3)) if rounds_ms else None,
                               "sharded_first_pass_ms_rank0": ph or None, "region_bytes_sent_rank0": sent, "combined_exchange_rank0": out["e2e"].get("combined_exchange_rank0"),
                               "all_to_all_GBs_rank0": (sent / (a2a_ms * 1e-3) / 1e9) if sent and a2a_ms > 0 else None,
                               "runs": out["e2e"].get("runs")}
        out["e2e_failed"] = "error" in out["e2e"]
    if cpu_baseline is not None and recs is not None:
        try:
            out["cpu_baseline"] = cpu_baseline(recs, p)
        except Exception as ex:  # noqa: BLE001
            out["cpu_baseline"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    try:  # librccl prints a version banner through C stdio, which would otherwise land behind the JSON line when the process ends
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    sys.stdout.flush()
    line = (json.dumps(out) + "\n").encode()
    if _LINE_FD is not None:
        os.write(_LINE_FD, line)
    else:
        os.write(1, line)
    return 0
