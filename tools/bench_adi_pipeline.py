#!/usr/bin/env python3
"""N1 end to end: batched get_random_samples (walks + expansion + one-hots + value-net forward + target assembly)
with a random-init stand-in of the reference's DeepCube (model.py, hidden [1024,256,128]).  The reference does
393 samples/s on one CPU core with the same net; its own size is 200 cubes x depth 30
(config/config.yaml:7-8, called once per epoch by train.py:152-155).

    python tools/bench_adi_pipeline.py [walks ...] [--graph] [--reps R] [--depth D] [--cube-size 2|3]

Under `rocprofv3 --kernel-trace` every timed call is bracketed by three k_fill_solved launches on a 1-cube buffer (a kernel the
pipeline itself never launches), so tools/adi_split.py can cut the trace at the call's boundaries."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from bench_cfg5 import DeepCubeStandIn
from rubiks_cube_solver_amd import ops
from rubiks_cube_solver_amd.adi import adi_samples

SIZES = ((200, 30), (20_000, 30), (100_000, 30))


def run(sizes=SIZES, reps=3, graph=False, model=None, dev=None, markers=False, cube_size=3):
    """-> {"WxD": {"seconds": median wall time of one adi_samples call (synchronised), "samples_per_s": ...}}
    cube_size 2: the shipped 2x2x2 checkpoint's layer sizes (pretrained/222model.pt: 147 -> 512 -> 128 -> {64 -> 6, 64 -> 1})."""
    dev = dev or torch.device("cuda")
    if model is None:
        model = (DeepCubeStandIn() if cube_size == 3 else DeepCubeStandIn((7, 21), 6, (512, 128, 64))).to(dev).eval()
    mark = ops.alloc_states(1, 3, dev)
    kw = {"graph": True} if graph else {}
    out = {}
    for walks, depth in sizes:
        for _ in range(2):
            adi_samples(model, cube_size, walks if graph else min(walks, 2000), depth, 1.0, device=dev, seed=1, **kw)   # warm-up (graph: the capture)
        torch.cuda.synchronize()
        times = []
        for r in range(reps):
            if markers:
                for _ in range(3):
                    ops.fill_solved(mark, 1, 3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = adi_samples(model, cube_size, walks, depth, 1.0, device=dev, seed=2 + r, **kw)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            if markers:
                for _ in range(3):
                    ops.fill_solved(mark, 1, 3)
                torch.cuda.synchronize()
            assert res["target_value"].shape == (walks, depth)
        dt = sorted(times)[len(times) // 2]
        out[f"{walks}x{depth}"] = {"seconds": round(dt, 5), "samples_per_s": round(walks * depth / dt, 1), "best_seconds": round(min(times), 5)}
    return out


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("walks", type=int, nargs="*")
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--depth", type=int, default=30)
    ap.add_argument("--cube-size", type=int, default=3)
    a = ap.parse_args()
    sizes = tuple((w, a.depth) for w in a.walks) or SIZES
    print(json.dumps(run(sizes, a.reps, graph=a.graph, markers=True, cube_size=a.cube_size)))


if __name__ == "__main__":
    main()
