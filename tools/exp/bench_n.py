import os, sys
sys.path.insert(0, os.getcwd())
import torch
from rubiks_cube_solver_amd import ops
def timeit(fn, iters=60, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for n in [int(x) for x in os.environ.get("NS", "4194304,2097152,4194304,8388608,4194304").split(",")]:
    a = ops.alloc_states(n, 3, "cuda"); b = torch.empty_like(a)
    ops.fill_solved(a, n, 3); ops.scramble(a, n, 3, 20, seed=3)
    acts = torch.randint(0, 12, (n,), dtype=torch.uint8, device="cuda"); done = torch.empty(n, dtype=torch.uint8, device="cuda")
    bufs = [a, b]
    def f():
        ops.apply_moves(bufs[0], bufs[1], acts, n, 3, None, done); bufs.reverse()
    t = timeit(f)
    print(f"3x3x3 step n=2^{n.bit_length()-1}: {t*1e6:.1f} us  {n/t/1e9:.2f} G steps/s  {110*n/t/1e9:.0f} GB/s")
    del a, b
