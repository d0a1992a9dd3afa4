#!/usr/bin/env python3
"""N1 end to end: batched get_random_samples (walks + expansion + one-hots + value-net forward + target assembly)
with a random-init stand-in of the reference's DeepCube (model.py, hidden [1024,256,128]).  The reference does
393 samples/s on one CPU core with the same net."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from bench_cfg5 import DeepCubeStandIn
from rubiks_cube_solver_amd.adi import adi_samples


def main():
    dev = torch.device("cuda")
    model = DeepCubeStandIn().to(dev).eval()
    out = {}
    for walks, depth in ((200, 30), (20_000, 30), (100_000, 30)):
        adi_samples(model, 3, min(walks, 2000), depth, 1.0, device=dev, seed=1)   # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = adi_samples(model, 3, walks, depth, 1.0, device=dev, seed=2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert res["target_value"].shape == (walks, depth)
        out[f"{walks}x{depth}"] = {"seconds": round(dt, 4), "samples_per_s": round(walks * depth / dt, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
