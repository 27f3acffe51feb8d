"""TEST INFRASTRUCTURE: CPU-only exercise of bench.py's LAUNCHER (`python bench.py --gpus N` from a plain shell re-executes
itself under torch.distributed.run on 127.0.0.1; the driver's pre-launched ranks read RANK / WORLD_SIZE) with the NumPy
checker backend of the test-suite in place of the device engine.  Until round 5 this lived in bench.py as `--backend
checker`; a NumPy stand-in has no place in the measurement script.  The launcher code itself --
bench.self_launch, the WORLD_SIZE / --gpus guard -- is bench.py's own, imported here.  The line printed is marked
data = "checker" and is never a measurement (tests/test_bench_launcher.py)."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "krylovkit.jl_amd"))
import bench                      # noqa: E402
from bench import laplacian_rows  # noqa: E402


def main_checker(args, rank: int, world: int):
    """CPU-only exercise of the launcher: rendezvous (gloo), row partition, ghost exchange and the two all-reduces of a
    sharded Lanczos sweep with the NumPy checker backend of the test-suite.  Prints a line marked data = "checker";
    it is NOT a measurement (tests/test_bench_launcher.py)."""
    import torch.distributed as dist
    sys.path.insert(0, str(ROOT / "tests"))
    sys.path.insert(0, str(ROOT / "oracle"))
    from dist_checker_backend import CheckerBackend
    import splitphase_dist as kd          # test-side exerciser of the split-phase C entry points (tests/splitphase_dist.py)
    import krylovkit_hip as kk

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    be = CheckerBackend()
    nx, nyl, kd_ = 32, max(4, min(args.ny, 16)), 12
    part = kd.Partition.even(nx * nyl * world, world, rank, align=nx)
    A = laplacian_rows(nx, nyl * world, rank * nyl, (rank + 1) * nyl)
    dop = kd.DistSparseOperator(A, part, be)
    x0 = np.random.default_rng(3 + rank).random(nx * nyl)
    it = kd.DistLanczosIterator(dop, x0, kk.Orthogonalizer(args.orth), capacity=kd_ + 2)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        f = it.initialize()
        for _ in range(kd_ - 1):
            f = it.expand(f)
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "lanczos_iterations_per_second", "value": round((kd_ - 1) * args.steps * world / elapsed, 3),
                          "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "data": "checker",
                          "higher_is_better": True, "scaling": "weak", "last_alpha": f.alphas[-1], "last_beta": f.betas[-1],
                          "config": {"workload": f"launcher self-test: {nx}x{nyl * world} Laplacian, NumPy checker backend, gloo"}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--orth", default="mgs2", choices=["cgs2", "mgs2"])
    ap.add_argument("--ny", type=int, default=8)
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus > 1 and world == 1:     # the same three lines as bench.main()
        if os.environ.get("KK_BENCH_SPAWNED"):
            raise SystemExit("bench_checker.py: spawned without WORLD_SIZE")
        raise SystemExit(bench.self_launch(args, script=__file__))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    return main_checker(args, rank, world)


if __name__ == "__main__":
    main()
