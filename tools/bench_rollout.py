#!/usr/bin/env python3
"""N3: validation-style greedy rollouts (train.py:167-198: 30 scramble depths x 10 cubes, 200 steps) eager vs hipGraph,
and a large batch.  The net is a random-init stand-in of the reference's DeepCube (it rarely solves: all steps run)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from bench_cfg5 import DeepCubeStandIn
from rubiks_cube_solver_amd import VecCubeEnv
from rubiks_cube_solver_amd.rollout import greedy_rollout


def run(n, T, graph, model):
    env = VecCubeEnv(n, "cuda", 3, obs="onehot")
    env.reset(scramble_count=15)
    greedy_rollout(model, env, 4, graph=graph)          # warm-up
    env.reset(scramble_count=15)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = greedy_rollout(model, env, T, sync_every=50, graph=graph)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = res["actions"].shape[0]
    return {"seconds": round(dt, 4), "us_per_timestep": round(dt / steps * 1e6, 1), "cube_steps_per_s": round(n * steps / dt, 1)}


def run_all(T=200, model=None):
    """n = 300 is train.py:167-198's validation batch (30 scramble depths x 10 cubes); 65536 a large batch."""
    model = model or DeepCubeStandIn().cuda().eval()
    out = {}
    for n in (300, 65536):
        for graph in (False, True):
            out[f"n{n}_{'hipgraph' if graph else 'eager'}"] = run(n, T, graph, model)
    return out


def main():
    print(json.dumps(run_all()))


if __name__ == "__main__":
    main()
