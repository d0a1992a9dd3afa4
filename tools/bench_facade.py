#!/usr/bin/env python3
"""Latency breakdown of the batch-1 facade (CubeEnv.step / MCTS.train) on one GPU (development tool)."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import rubiks_cube_solver_amd as rc
from rubiks_cube_solver_amd import _lib


def per_call(fn, n=5000, warm=200):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    out = {}
    env = rc.make_env(torch.device("cpu"), 3)
    env.reset(seed=1, scramble_count=20)
    acts = np.random.default_rng(0).integers(0, 12, 100000)
    it = iter(acts)
    out["CubeEnv.step_us"] = per_call(lambda: env.step(int(next(it))))
    seeds = iter(range(10 ** 6))
    out["CubeEnv.reset_seed_k30_us"] = per_call(lambda: env.reset(seed=next(seeds), scramble_count=30), n=2000)
    f = env._facade()
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    seq = [f[5]]

    def raw():
        seq[0] = seq[0] % 0xFFFFFFFF + 1
        f[6](f[3], f[4], 3, 5, f[2], seq[0], 1, sp)
    out["rc_facade_step_ctypes_us"] = per_call(raw)
    f[5] = seq[0]
    out["current_stream_lookup_us"] = per_call(lambda: ctypes.c_void_p(torch.cuda.current_stream(env._vec.device).cuda_stream))
    h = f[1]
    out["numpy_postprocess_us"] = per_call(lambda: h[:480].reshape(20, 24).astype(np.int64))
    out["expand_host_us"] = per_call(lambda: env.expand_host())
    out["expand_host_dense_us"] = per_call(lambda: env.expand_host(dense=True))
    import copy
    out["deepcopy_env_us"] = per_call(lambda: copy.deepcopy(env), n=1000)
    env2 = rc.make_env(torch.device("cpu"), 2)
    it2 = iter(np.random.default_rng(0).integers(0, 6, 100000))
    out["CubeEnv222.step_us"] = per_call(lambda: env2.step(int(next(it2))))
    # MCTS.train per simulation with a tiny host model (the reference's tree code path)
    from rubiks_cube_solver_amd.mcts_batched import MCTS

    class M:
        def predict(self, x):
            return np.float32(-1.0), np.full(12, 1 / 12, np.float32)
    cfg = {"mcts": {"virtual_loss_const": 150, "cpuct": 1.0, "value_min": -10.0}, "test": {"cube_size": 3}}
    env.reset(seed=3, scramble_count=12)
    state = env.cube

    def search():                                   # test.py:140-142: numMCTSSim = 50 simulations on a fresh tree
        tree = MCTS(M(), cfg)
        for _ in range(50):
            if tree.train(state, env) is not None:
                break
    out["MCTS_50_simulations_us"] = per_call(search, n=40, warm=3)
    out["MCTS_us_per_simulation"] = out["MCTS_50_simulations_us"] / 50
    print(json.dumps({k: round(v, 2) for k, v in out.items()}))


if __name__ == "__main__":
    main()
