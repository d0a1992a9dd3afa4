"""rc_onehot_from_family_depths at the ADI pipeline's shapes: us per launch and fraction of the 8 TB/s peak (bytes = 51 R + 13 * 480 * esize W per walk-depth),
next to the plain code -> dense front writer writing the same number of bytes."""
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
for wn, gd, dt, esz in ((43008, 1, torch.float32, 4), (20000, 2, torch.float32, 4), (200, 30, torch.float32, 4), (43008, 2, torch.bfloat16, 2), (43008, 2, torch.uint8, 1)):
    pitch, bufs = ops.adi_buffers(wn, gd, 3, "cuda", family=True)
    ops.adi_generate(wn, gd, 3, pitch, "cuda", seed=1, **bufs)
    bs = -(-wn // 8) * 8
    blocks = torch.empty((gd * 13 * bs, 20, 24), dtype=dt, device="cuda")
    t = timed(lambda: ops.onehot_from_family(bufs["family"], wn, 3, blocks, block_stride=bs, n_depths=gd))
    byts = wn * gd * (51 + 13 * 480 * esz)
    m = wn * gd * 13
    code = ops.alloc_code(m, 3, "cuda"); code.random_(0, 24)
    flat = torch.empty((m, 20, 24), dtype=dt, device="cuda")
    t2 = timed(lambda: ops.onehot_from_code(code, m, 3, flat))
    out.append({"walks": wn, "depths": gd, "dtype": str(dt), "family_us": round(t, 1), "family_frac": round(byts / t / 8e6, 3), "plain_same_cubes_us": round(t2, 1),
                "plain_frac": round(m * (20 + 480 * esz) / t2 / 8e6, 3), "kernel": _lib.describe(_lib.OP_FAMILY_TO_DENSE, 3, wn, gd, fmt=_lib.fmt_of(dt))})
print(json.dumps(out, indent=1))
