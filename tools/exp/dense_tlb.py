"""Per-dispatch UTCL1/UTCL2 counters of the dense writer into several separately allocated buffers (run under rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rubiks_cube_solver_amd import _lib, ops
from microbench import timeit
m = 1 << 20
a = ops.alloc_states(m, 3, "cuda"); ops.fill_solved(a, m, 3); ops.scramble(a, m, 3, 20, seed=1)
code = ops.alloc_code(m, 3, "cuda"); ops.encode(a, m, 3, code, _lib.FMT_CODE)
bufs, pads = [], []
for i in range(6):
    bufs.append(torch.empty((m, 20, 24), dtype=torch.float32, device="cuda"))
    pads.append(torch.empty(((i + 1) * 37) << 20, dtype=torch.uint8, device="cuda"))
for i, oh in enumerate(bufs):
    t = timeit(lambda: ops.onehot_from_code(code, m, 3, oh), iters=4, warm=1)
    print("buffer %d addr %x: %.1f us" % (i, oh.data_ptr(), t * 1e6), flush=True)
