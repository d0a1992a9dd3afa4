"""Experiment (not shipped): the ADI kernel with 16 walks per lane.  Needs a build in which rc_adi_generate_ex maps variant digit 3 to
launch_adi<T, 4> (see git history: "Record the re-measured 16-walks-per-lane ADI experiment"); measured 0.44 ms against 0.32 ms for the shipped
8-walks-per-lane form."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from rubiks_cube_solver_amd import _lib, ops
from microbench import timeit
W, D = 100_000, 30
for pitch in (8192, 16384):
    pt, bufs = ops.adi_buffers(W, D, 3, "cuda", pitch, parents=True, children=True)
    for var in (0, 1003, 2003, 3003, 1002, 0):
        t = timeit(lambda: ops.adi_generate(W, D, 3, pt, "cuda", seed=2024, variant=var, **bufs), iters=5, warm=2)
        print("pitch", pitch, "variant", var, round(t * 1e6, 1), "us", round(715 * W * D / t / 1e9), "GB/s", flush=True)
    # parity of the V4 path against the default
    ref = {k: v.clone() for k, v in bufs.items()}
    ops.adi_generate(W, D, 3, pt, "cuda", seed=2024, variant=0, **ref)
    ops.adi_generate(W, D, 3, pt, "cuda", seed=2024, variant=1003, **bufs)
    print("V4 == default:", all(torch.equal(ref[k], bufs[k]) for k in bufs))
    del bufs, ref
