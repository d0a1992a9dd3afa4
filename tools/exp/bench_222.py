import os, sys
sys.path.insert(0, os.getcwd())
import torch
from rubiks_cube_solver_amd import ops, _lib
def timeit(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for n in (1 << 22, 1 << 24):
    a = ops.alloc_states(n, 2, "cuda"); b = torch.empty_like(a)
    ops.fill_solved(a, n, 2); ops.scramble(a, n, 2, 20, seed=3)
    acts = torch.randint(0, 6, (n,), dtype=torch.uint8, device="cuda"); done = torch.empty(n, dtype=torch.uint8, device="cuda")
    bufs = [a, b]
    def f():
        ops.apply_moves(bufs[0], bufs[1], acts, n, 2, None, done); bufs.reverse()
    t = timeit(f)
    print(f"2x2x2 step n={n}: {t*1e6:.1f} us  {n/t/1e9:.1f} G steps/s  {50*n/t/1e9:.0f} GB/s (50 B/step)")
