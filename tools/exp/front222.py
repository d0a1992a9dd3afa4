"""Round 6, bounded attempt 2: the 2x2x2 FRONT writer (k_code_to_dense_front222: NP passes of 2352 bytes per workgroup-front, F fronts per XCD
per workgroup, code dwords through LDS) against the tile kernels, three dense formats, 2^12 .. 2^22 cubes; every output must equal the
64-cube tile form's.  Prints one JSON object (fractions of the 8 TB/s peak: 7 B read + 147 * esize written per cube)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rubiks_cube_solver_amd import _lib, ops

def timed(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    vals = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        vals.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(vals)[1]

FORMS = [("tile64", 100000), ("tile256", 200000), ("front_np2_f1", 400001), ("front_np2_f2", 400002), ("front_np1_f1", 400031), ("front_np1_f2", 400032),
         ("front_np4_f1", 400041), ("front_np4_f2", 400042), ("front_np2_linear", 400021), ("default", 0)]
out = []
for n in ((1 << 12) + 3, (1 << 15) + 5, (1 << 16) + 3, 1 << 18, 1 << 20, 1 << 22):
    st = ops.alloc_states(n, 2, "cuda"); ops.fill_solved(st, n, 2); ops.scramble(st, n, 2, 11, seed=n & 31)
    code = ops.alloc_code(n, 2, "cuda"); ops.encode(st, n, 2, code, _lib.FMT_CODE)
    for dt, esz in ((torch.float32, 4), (torch.bfloat16, 2), (torch.uint8, 1)):
        ref = torch.zeros((n, 7, 21), dtype=dt, device="cuda"); ops.onehot_from_code(code, n, 2, ref, variant=100000)
        assert float(ref.float().sum()) == 7 * n
        row = {"n": n, "dtype": str(dt).split(".")[-1]}
        for name, v in FORMS:
            oh = torch.full((n + 8, 7, 21), 3, dtype=dt, device="cuda")
            ops.onehot_from_code(code, n, 2, oh[:n], variant=v)
            assert torch.equal(oh[:n], ref) and float(oh[n:].float().min()) == 3, (n, dt, name)       # same bytes, nothing past the end
            t = timed(lambda: ops.onehot_from_code(code, n, 2, oh[:n], variant=v))
            row[name] = [round(t, 1), round(n * (7 + 147 * esz) / t / 8e6, 3)]
        row["default_kernel"] = _lib.describe(_lib.OP_CODE_TO_DENSE, 2, n, fmt=_lib.fmt_of(dt)).split(" ")[0]
        out.append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    del st, code
print(json.dumps(out))
