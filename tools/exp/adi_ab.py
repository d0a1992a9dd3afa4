"""Interleaved A/B of ADI layouts / part counts in ONE process (rule 24), median and min per variant."""
import os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from rubiks_cube_solver_amd import _lib, ops
if os.environ.get('RC_LIB'): _lib.LIB_PATH = os.environ['RC_LIB']
print('lib', _lib.LIB_PATH)
L = _lib.lib()
W, D, dev = 100_000, 30, torch.device("cuda", 0)
def run(fn, iters):
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(iters): fn()
    s1.record(); torch.cuda.synchronize()
    return s0.elapsed_time(s1) / iters
variants = {}
combos = [(f"{nm}_p{pp}", t, pp * 1000) for nm, t in (("plain", _lib.pitch_for(W)), ("tile4096", 4096)) for pp in (6, 12)]
for name, pitch, parts in combos:
    pt, ab = ops.adi_buffers(W, D, 3, dev, pitch, parents=True, children=True)
    variants[name] = (pt, ab, parts)
    print(name, hex(ab["children"].data_ptr()), hex(ab["parents"].data_ptr()))
n = 1 << 22
a = ops.alloc_states(n, 3, dev); b = torch.empty_like(a); ops.fill_solved(a, n, 3); ops.scramble(a, n, 3, 20, seed=1)
acts = torch.randint(0, 12, (n,), dtype=torch.uint8, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
res = {k: [] for k in list(variants) + ["step4M"]}
for r in range(6):
    for name, (pt, ab, parts) in variants.items():
        fn = lambda: ops.adi_generate(W, D, 3, pt, dev, seed=2024, variant=parts, **ab)   # per-call override (round 2)
        if r == 0: run(fn, 3)
        res[name].append(run(fn, 20))
    fn = lambda: ops.apply_moves(a, b, acts, n, 3, None, done)
    res["step4M"].append(run(fn, 50))
for k, v in res.items():
    by = 110 * n if k == "step4M" else 715 * W * D
    print(f"{k:16s} median {statistics.median(v):.4f} ms  min {min(v):.4f}  max {max(v):.4f}   {by / statistics.median(v) / 1e6:.0f} GB/s (median)")
