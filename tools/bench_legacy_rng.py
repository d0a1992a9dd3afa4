#!/usr/bin/env python3
"""reset(seed, k)'s draws on the device (rc_legacy_scramble_actions: numpy's legacy MT19937 + masked rejection per env, cube_env.py:62-65):
microseconds per launch for the default route and for each generator form on its own, HIP events, and VecCubeEnv.reset(seeds) end to end."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from rubiks_cube_solver_amd import VecCubeEnv, ops


def timed(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    out = {}
    for n, k in ((300, 30), (1 << 16, 30), (1 << 20, 30), (1 << 20, 100), (1 << 20, 200), (1 << 20, 400), (1 << 16, 1000)):
        seeds = torch.arange(n, dtype=torch.int64, device="cuda") * 10
        row = {}
        for name, variant in (("default", 0), ("lds_lazy_twist", 1), ("streaming_plus_fixup", 2)):
            if variant == 2 and k > 400:
                continue                                          # would overflow in most waves: not a route the default rule takes
            row[name + "_us"] = round(timed(lambda: ops.legacy_scramble_actions(seeds, 3, k, device="cuda", variant=variant)), 1)
        out[f"{n}x{k}"] = row
    n = 1 << 20
    env = VecCubeEnv(n, "cuda", 3, obs=None)
    seeds = torch.arange(n, dtype=torch.int64, device="cuda")
    env.reset(seeds=seeds, scramble_count=30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        env.reset(seeds=seeds, scramble_count=30)
    torch.cuda.synchronize()
    out["VecCubeEnv.reset(seeds, 30) 1M envs ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    out["note"] = "round 4 (eager 624-word twist in LDS): 3.0 ms for 1M envs x k = 30"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
