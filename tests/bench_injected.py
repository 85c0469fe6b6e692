"""TEST ONLY.  `bench.py --gpus N` with the oracle standing in for the HIP library, so that the launch path the driver's
scaling run depends on -- bench.launch_ranks (torch.distributed.run as a child process), the rendezvous, the timed loop
with its barrier and max-over-ranks clock, the golden check and the JSON line of twopaco_amd/dist.py:bench_main -- runs
over gloo on a machine without GPUs (tests/test_dist_cpu.py::test_bench_gpus_flag_launches_ranks).  The product has no
such seam: bench.py / twopaco_amd never load anything under oracle/; this file composes their functions with a fake
rank backend.  Usage: python tests/bench_injected.py --gpus 2 --steps 2 --warmup 1 [--golden-junctions J]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def main():
    import argparse
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--golden-junctions", type=int, default=-1, help="pretend the reference found this many junctions (to test the non-zero exit)")
    args = ap.parse_args()
    args.workload, args.scale, args.decomposition = "test", 1.0, "ranges"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(bench.launch_ranks(args, script=os.path.abspath(__file__)))
    from dist_worker import bench_backend
    from helpers import golden_cases
    from twopaco_amd import dist as tdist
    case = dict([c for c in golden_cases() if c["name"] == "rand6_k9_fp"][0])
    if args.golden_junctions >= 0:
        case["distinct"] = args.golden_junctions
    tdist.bench_main(args, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")),
                     backend_factory=bench_backend, golden=case)


if __name__ == "__main__":
    main()
