import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
from rubiks_cube_solver_amd import _lib, ops
if os.environ.get('RC_LIB'): _lib.LIB_PATH = os.environ['RC_LIB']
print('lib', _lib.LIB_PATH)
def timed(fn, iters=5, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(iters): fn()
    s1.record(); torch.cuda.synchronize()
    return s0.elapsed_time(s1) / iters
W, D = 100_000, 30
dev = torch.device("cuda", 0)
for trial in range(3):
    pt, ab = ops.adi_buffers(W, D, 3, dev, parents=True, children=True)
    print("trial", trial, "ptr%2MB", ab["children"].data_ptr() % (2<<20), "ms", [round(timed(lambda: ops.adi_generate(W, D, 3, pt, dev, seed=2024, **ab)),4) for _ in range(3)])
    del ab
big = torch.empty((1<<20, 20, 24), dtype=torch.float32, device=dev); big.fill_(1.0); del big
pt, ab = ops.adi_buffers(W, D, 3, dev, parents=True, children=True)
print("after 8GB alloc/free: ptr%2MB", ab["children"].data_ptr() % (2<<20), "ms", [round(timed(lambda: ops.adi_generate(W, D, 3, pt, dev, seed=2024, **ab)),4) for _ in range(3)])
del ab; torch.cuda.empty_cache()
pt, ab = ops.adi_buffers(W, D, 3, dev, parents=True, children=True)
print("after empty_cache: ms", [round(timed(lambda: ops.adi_generate(W, D, 3, pt, dev, seed=2024, **ab)),4) for _ in range(3)])
for iters in (5, 20, 50):
    print("iters", iters, round(timed(lambda: ops.adi_generate(W, D, 3, pt, dev, seed=2024, **ab), iters=iters), 4))
