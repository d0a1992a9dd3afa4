import os, sys, json
sys.path.insert(0, "/root/repo")
import torch
from rubiks_cube_solver_amd import _lib, ops
sys.path.insert(0, "/root/repo/tools")
from microbench import timeit
m = 1 << 20
a = ops.alloc_states(m, 3, "cuda"); ops.fill_solved(a, m, 3); ops.scramble(a, m, 3, 20, seed=1)
code = ops.alloc_code(m, 3, "cuda"); ops.encode(a, m, 3, code, _lib.FMT_CODE)
b2 = torch.empty_like(a)
acts = torch.randint(0, 12, (m,), dtype=torch.uint8, device="cuda")
done = torch.empty(m, dtype=torch.uint8, device="cuda")
for dt, b in ((torch.float32, 1920), (torch.bfloat16, 960), (torch.uint8, 480)):
    oh = torch.empty((m, 20, 24), dtype=dt, device="cuda")
    for rep in range(2):
        for cap in (0, 512, 1024, 1536, 2048, 2560, 3072):
            os.environ["RC_EXP_DENSE_GRID"] = str(cap)
            t = timeit(lambda: ops.onehot_from_code(code, m, 3, oh), iters=10)
            t2 = timeit(lambda: ops.apply_moves(a, b2, acts, m, 3, None, done, oh, _lib.fmt_of(dt)), iters=10)
            print(dt, "grid cap", cap, "code->dense", round(t * 1e6, 1), "us", round((20 + b) * m / t / 1e9), "GB/s | step+dense", round(t2 * 1e6, 1), "us",
                  round((110 + b) * m / t2 / 1e9), "GB/s", flush=True)
