import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from rubiks_cube_solver_amd import ops, VecCubeEnv
for n, k in ((1 << 16, 30), (1 << 20, 30), (1 << 20, 200)):
    seeds = torch.arange(n, dtype=torch.int64, device="cuda")
    ops.legacy_scramble_actions(seeds, 3, k, device="cuda"); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ops.legacy_scramble_actions(seeds, 3, k, device="cuda")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"legacy reset draws: {n} envs x k={k}: {dt*1e3:.2f} ms  ({n/dt/1e6:.1f} M envs/s)")
env = VecCubeEnv(1 << 20, "cuda", 3, obs=None)
seeds = list(range(1 << 20))
t0 = time.perf_counter(); env.reset(seeds=seeds, scramble_count=30); torch.cuda.synchronize()
print(f"VecCubeEnv.reset(seeds=1M ints, k=30) end to end: {(time.perf_counter()-t0)*1e3:.1f} ms")
