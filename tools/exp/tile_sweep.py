"""State-tile size sweep for the step kernel at 2^22 cubes with the round-2 buffer addressing and cache policy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rubiks_cube_solver_amd import _lib, ops
from microbench import timeit
n = 1 << 22
acts = torch.randint(0, 12, (n,), dtype=torch.uint8, device="cuda")
done = torch.empty(n, dtype=torch.uint8, device="cuda")
for rep in range(2):
    for pitch in (4096, 8192, 16384, 32768, 65536, 131072, 1 << 20, 1 << 22):
        a = ops.alloc_states(n, 3, "cuda", pitch); b = torch.empty_like(a)
        ops.fill_solved(a, n, 3); ops.scramble(a, n, 3, 20, seed=1)
        buf = [a, b]
        def f():
            ops.apply_moves(buf[0], buf[1], acts, n, 3, None, done); buf.reverse()
        t = timeit(f, iters=50)
        print("tile", pitch, round(t * 1e6, 1), "us", round(110 * n / t / 1e9), "GB/s", flush=True)
        del a, b, buf
