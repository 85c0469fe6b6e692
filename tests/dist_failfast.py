"""TEST ONLY.  One rank of a 2-rank gloo group that misbehaves on purpose (tests/test_dist_cpu.py):
  python tests/dist_failfast.py <mode> <rank> <world> <port>
  hang   rank 1 joins the group and then never arrives at the collective; rank 0 waits in it -> rank 0's watchdog
         (twopaco_amd/dist.py:PhaseWatchdog) must end the process with exit code 17 and name the phase
  agree  rank 1's local work "fails" (comm.fail) before a variable all_to_all; BOTH ranks must raise DistAbort naming the
         phase, with no data moved -- exit code 0 when they did"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode, rank, world, port = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import torch
    import torch.distributed as dist

    from twopaco_amd import dist as tdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = tdist._Comm(dist, torch.device("cpu"))
    comm.phase = "query batch 0"
    if mode == "hang":
        if rank == 1:
            time.sleep(60)  # never arrives (the test kills this process)
            return 1
        comm.a2a_var(torch.arange(4, dtype=torch.int64), [2, 2])
        return 1  # unreachable: the watchdog ends the process
    if mode == "agree":
        if rank == 1:
            comm.fail(RuntimeError("tpc_shard_apply: out of device memory"))
        try:
            comm.a2a_var(torch.arange(4, dtype=torch.int64), [2, 2])
        except tdist.DistAbort as e:
            ok = "query batch 0:all_to_all(variable)" in str(e) and "[1]" in str(e)
            print("rank %d: %s" % (rank, e), flush=True)
            # the same through the all-reduce form (max_ints carries the flag) on the still-standing group
            comm.rc, comm.err = (1, "again") if rank == 0 else (0, "")
            try:
                comm.max_ints([rank])
                ok = False
            except tdist.DistAbort:
                pass
            dist.destroy_process_group()
            return 0 if ok else 1
        return 1
    return 2


if __name__ == "__main__":
    sys.exit(main())
