"""2x2x2 compact code -> dense one-hot [N,7,21]: the three writer forms side by side (64-cube tiles, 256-cube tiles, the 2352-byte pass form),
microseconds and fraction of the 8 TB/s peak (7 B read + 147 * esize written per cube); every form's output is compared with the 64-cube form's."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rubiks_cube_solver_amd import _lib, ops

def timed(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

out = []
for lg in (12, 14, 15, 16, 17, 18, 20, 22):
    n = (1 << lg) + (3 if lg < 20 else 0)
    st = ops.alloc_states(n, 2, "cuda"); ops.fill_solved(st, n, 2); ops.scramble(st, n, 2, 11, seed=lg)
    code = ops.alloc_code(n, 2, "cuda"); ops.encode(st, n, 2, code, _lib.FMT_CODE)
    for dt, esz in ((torch.float32, 4), (torch.bfloat16, 2), (torch.uint8, 1)):
        ref = torch.zeros((n, 7, 21), dtype=dt, device="cuda"); ops.onehot_from_code(code, n, 2, ref, variant=100000)
        row = {"n": n, "dtype": str(dt).split(".")[-1]}
        for name, v in (("tile64", 100000), ("tile256", 200000), ("default", 0)):
            oh = torch.zeros((n, 7, 21), dtype=dt, device="cuda")
            ops.onehot_from_code(code, n, 2, oh, variant=v)
            assert torch.equal(oh, ref), (n, dt, name)
            t = timed(lambda: ops.onehot_from_code(code, n, 2, oh, variant=v))
            row[name + "_us"] = round(t, 1); row[name + "_frac"] = round(n * (7 + 147 * esz) / t / 8e6, 3)
        row["default_kernel"] = _lib.describe(_lib.OP_CODE_TO_DENSE, 2, n, fmt=_lib.fmt_of(dt)).split(" ")[0]
        out.append(row)
print(json.dumps(out, indent=0))
