"""Is the bimodal dense one-hot time (315 vs 380 us for 1M cubes f32) a function of the output buffer's address?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rubiks_cube_solver_amd import _lib, ops
from microbench import timeit
m = 1 << 20
a = ops.alloc_states(m, 3, "cuda"); ops.fill_solved(a, m, 3); ops.scramble(a, m, 3, 20, seed=1)
code = ops.alloc_code(m, 3, "cuda"); ops.encode(a, m, 3, code, _lib.FMT_CODE)
nbytes = m * 480 * 4
pool = torch.empty(nbytes * 4, dtype=torch.uint8, device="cuda")
print("pool base %x" % pool.data_ptr())
for off in (0, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 1 << 28, (1 << 28) + (1 << 20), 1 << 30, (1 << 30) + 65536, 3 << 29, nbytes, 2 * nbytes, 3 * nbytes - 4096):
    if off + nbytes > pool.numel():
        continue
    oh = pool[off:off + nbytes].view(torch.float32).view(m, 20, 24)
    t = timeit(lambda: ops.onehot_from_code(code, m, 3, oh), iters=10)
    print("offset %11d (addr %x): %.1f us" % (off, oh.data_ptr(), t * 1e6), flush=True)
for i in range(6):
    oh = torch.empty((m, 20, 24), dtype=torch.float32, device="cuda")
    t = timeit(lambda: ops.onehot_from_code(code, m, 3, oh), iters=10)
    print("fresh alloc %d addr %x: %.1f us" % (i, oh.data_ptr(), t * 1e6), flush=True)
    keep = torch.empty(((i + 1) * 37) << 20, dtype=torch.uint8, device="cuda")   # perturb the allocator
