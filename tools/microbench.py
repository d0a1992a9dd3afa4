#!/usr/bin/env python3
"""Kernel micro-benchmarks on one GPU (development tool; bench.py is the contract)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rubiks_cube_solver_amd import _lib, ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    out = []
    L = _lib.lib()
    which = sys.argv[1:] or ["step", "code", "dense", "adi", "expand", "copy"]
    n = 1 << 22
    a = ops.alloc_states(n, 3, "cuda")
    b = torch.empty_like(a)
    ops.fill_solved(a, n, 3)
    ops.scramble(a, n, 3, 20, seed=1234)
    acts = torch.randint(0, 12, (n,), dtype=torch.uint8, device="cuda")
    done = torch.empty(n, dtype=torch.uint8, device="cuda")
    rew = torch.empty(n, dtype=torch.float32, device="cuda")
    if "copy" in which:
        t = timeit(lambda: b.copy_(a))
        out.append(dict(k="torch_copy_54rows", ms=t * 1e3, GBps=2 * a.numel() / t / 1e9))
    if "step" in which:
        for var in (0, 1, 2, 12, 22, 32):
            buf = [a, b]
            def f():
                ops.apply_moves(buf[0], buf[1], acts, n, 3, None, done, variant=var)
                buf.reverse()
            t = timeit(f)
            out.append(dict(k=f"step_v{var}", ms=t * 1e3, Gsteps=n / t / 1e9, GBps=110 * n / t / 1e9))
            def f2():
                ops.apply_moves(buf[0], buf[1], acts, n, 3, rew, done, variant=var)
                buf.reverse()
            t = timeit(f2)
            out.append(dict(k=f"step_reward_v{var}", ms=t * 1e3, Gsteps=n / t / 1e9, GBps=114 * n / t / 1e9))
            def f3():
                ops.apply_moves(a, a, acts, n, 3, None, done, variant=var)
            t = timeit(f3)
            out.append(dict(k=f"step_inplace_v{var}", ms=t * 1e3, Gsteps=n / t / 1e9, GBps=110 * n / t / 1e9))
    if "code" in which:
        code = ops.alloc_code(n, 3, "cuda")
        for var in (0, 1, 2):
            buf = [a, b]
            def f():
                ops.apply_moves(buf[0], buf[1], acts, n, 3, None, done, code, _lib.FMT_CODE, variant=var)
                buf.reverse()
            t = timeit(f)
            out.append(dict(k=f"step_code_v{var}", ms=t * 1e3, Gsteps=n / t / 1e9, GBps=130 * n / t / 1e9))
    if "densetile" in which:
        m = 1 << 20
        oh = torch.empty((m, 20, 24), dtype=torch.float32, device="cuda")
        oh8 = torch.empty((m, 20, 24), dtype=torch.uint8, device="cuda")
        code = ops.alloc_code(m, 3, "cuda")
        ops.encode(a, m, 3, code, _lib.FMT_CODE)
        for rep in range(2):
            for forced, name in ((100000, 64), (200000, 256)):
                t = timeit(lambda: ops.apply_moves(a, b, acts, m, 3, None, done, oh, _lib.FMT_F32, variant=forced), iters=10)
                out.append(dict(k=f"step_dense_f32_1M_tile{name}", ms=t * 1e3, GBps=(110 + 1920) * m / t / 1e9))
                t = timeit(lambda: ops.apply_moves(a, b, acts, m, 3, None, done, oh8, _lib.FMT_U8, variant=forced), iters=10)
                out.append(dict(k=f"step_dense_u8_1M_tile{name}", ms=t * 1e3, GBps=(110 + 480) * m / t / 1e9))
                t = timeit(lambda: ops.onehot_from_code(code, m, 3, oh), iters=10)
                out.append(dict(k=f"code_to_dense_f32_1M_tile{name}", ms=t * 1e3, GBps=(20 + 1920) * m / t / 1e9))
    if "dense" in which:
        m = 1 << 20
        for fmt, name, bpc in ((_lib.FMT_U8, "u8", 480), (_lib.FMT_F16, "f16", 960), (_lib.FMT_F32, "f32", 1920)):
            oh = torch.empty((m, 20, 24), dtype=_lib.dense_dtype(fmt), device="cuda")
            t = timeit(lambda: ops.apply_moves(a, b, acts, m, 3, None, done, oh, fmt), iters=10)
            out.append(dict(k=f"step_dense_{name}_1M", ms=t * 1e3, Gsteps=m / t / 1e9, GBps=(110 + bpc) * m / t / 1e9))
            del oh
    if "adi" in which or "adiplain" in which:
        W, D = 100_000, 30
        for pitch in ((ops.ADI_TILE,) if "adi" in which else (None,)):       # "adi": the default tiling only (fresh-process repeats)
            tag = "plain" if pitch is None else f"tile{pitch}"
            pt, bufs = ops.adi_buffers(W, D, 3, "cuda", pitch or _lib.pitch_for(W), parents=True, children=True)
            t = timeit(lambda: ops.adi_generate(W, D, 3, pt, "cuda", seed=2024, **bufs), iters=5, warm=2)
            out.append(dict(k=f"adi_100k_x30_stickers_{tag}", ms=t * 1e3, Gunits=W * D / t / 1e9, GBps=715 * W * D / t / 1e9))
            del bufs
            pt, bufs = ops.adi_buffers(W, D, 3, "cuda", pitch or _lib.pitch_for(W), parents=True, parent_code=True, child_code=True)
            t = timeit(lambda: ops.adi_generate(W, D, 3, pt, "cuda", seed=2024, **bufs), iters=5, warm=2)
            out.append(dict(k=f"adi_100k_x30_codes_{tag}", ms=t * 1e3, Gunits=W * D / t / 1e9,
                            GBps=(54 + 1 + 12 + 13 * 20) * W * D / t / 1e9))
            del bufs
    if "facade" in which:
        import time
        import numpy as np
        import rubiks_cube_solver_amd as rc
        env = rc.make_env(torch.device("cpu"), 3)
        env.reset(seed=1, scramble_count=20)
        acts = np.random.default_rng(0).integers(0, 12, 2000)
        for a_ in acts[:100]:
            env.step(int(a_))
        t0 = time.perf_counter()
        for a_ in acts:
            env.step(int(a_))
        dt = (time.perf_counter() - t0) / len(acts)
        out.append(dict(k="facade_batch1_step", us=dt * 1e6, steps_per_s=1 / dt))
        m = 1 << 20
        venv = rc.VecCubeEnv(m, "cuda", 3, obs="code")
        venv.reset(scramble_count=20)
        ha = torch.from_numpy(np.random.default_rng(1).integers(0, 12, m, dtype=np.uint8)).pin_memory()
        hd = torch.empty(m, dtype=torch.uint8).pin_memory()
        hcode = torch.empty(venv._obs_buf.shape, dtype=torch.uint8).pin_memory()
        def pcie_step():
            o, r, d, _ = venv.step(ha.to("cuda", non_blocking=True))
            hd.copy_(d, non_blocking=True)
            hcode.copy_(o, non_blocking=True)
            torch.cuda.synchronize()
        for _ in range(3):
            pcie_step()
        t0 = time.perf_counter()
        for _ in range(20):
            pcie_step()
        dt = (time.perf_counter() - t0) / 20
        out.append(dict(k="vec_1M_step_pcie_inclusive(actions H2D, done+code D2H)", ms=dt * 1e3, Gsteps=m / dt / 1e9))
    if "adigeo" in which:
        # launch geometry of the ADI kernel: pack width (units digit 1,2,3 -> 4,8,16 walks per lane) x parts x output tile
        W, D = 100_000, 30
        for pitch in (4096, 8192, 16384, 32768):
            pt, bufs = ops.adi_buffers(W, D, 3, "cuda", pitch, parents=True, children=True)
            for v in (1, 2):
                for parts in (1, 2, 3, 6, 12):
                    var = parts * 1000 + v
                    t = timeit(lambda: ops.adi_generate(W, D, 3, pt, "cuda", seed=2024, variant=var, **bufs), iters=5, warm=2)
                    out.append(dict(k=f"adi_stickers_pitch{pitch}_V{v}_parts{parts}", ms=t * 1e3, Gunits=W * D / t / 1e9, GBps=715 * W * D / t / 1e9))
            t = timeit(lambda: ops.adi_generate(W, D, 3, pt, "cuda", seed=2024, **bufs), iters=5, warm=2)
            out.append(dict(k=f"adi_stickers_pitch{pitch}_default", ms=t * 1e3, Gunits=W * D / t / 1e9, GBps=715 * W * D / t / 1e9))
            del bufs
        pt, bufs = ops.adi_buffers(W, D, 3, "cuda", parents=True, parent_code=True, child_code=True)
        for v in (1, 2):
            for parts in (1, 2, 3, 6, 12):
                var = parts * 1000 + v
                t = timeit(lambda: ops.adi_generate(W, D, 3, pt, "cuda", seed=2024, variant=var, **bufs), iters=5, warm=2)
                out.append(dict(k=f"adi_codes_V{v}_parts{parts}", ms=t * 1e3, Gunits=W * D / t / 1e9))
        t = timeit(lambda: ops.adi_generate(W, D, 3, pt, "cuda", seed=2024, **bufs), iters=5, warm=2)
        out.append(dict(k="adi_codes_default", ms=t * 1e3, Gunits=W * D / t / 1e9))
        del bufs
    if "expandgeo" in which:
        m = 1 << 20
        src = ops.alloc_states(m, 3, "cuda")
        ops.fill_solved(src, m, 3)
        ops.scramble(src, m, 3, 20, seed=5)
        for pitch in (4096, 8192, 16384, 32768):
            o = ops.expand_buffers(m, 3, "cuda", pitch, children=True, codes=False)
            for v in (1, 2):
                for parts in (1, 2, 6):
                    var = parts * 1000 + v
                    t = timeit(lambda: ops.expand_children(src, m, 3, o["children"], o["child_solved"], pitch=pitch, variant=var), iters=20)
                    out.append(dict(k=f"expand_1M_pitch{pitch}_V{v}_parts{parts}", us=t * 1e6, GBps=(54 + 12 * 54 + 12) * m / t / 1e9))
            t = timeit(lambda: ops.expand_children(src, m, 3, o["children"], o["child_solved"], pitch=pitch), iters=20)
            out.append(dict(k=f"expand_1M_pitch{pitch}_default", us=t * 1e6, GBps=(54 + 12 * 54 + 12) * m / t / 1e9))
            del o
    if "expand" in which:
        for m in (4096, 1 << 20):
            src = ops.alloc_states(m, 3, "cuda")
            ops.fill_solved(src, m, 3)
            ops.scramble(src, m, 3, 20, seed=5)
            for pitch in (None, 1024, 16384):
                if pitch is not None and m <= pitch:
                    continue
                o = ops.expand_buffers(m, 3, "cuda", pitch or _lib.pitch_for(m), children=True, codes=False)
                pt = o["children"].shape[-1]
                t = timeit(lambda: ops.expand_children(src, m, 3, o["children"], o["child_solved"], pitch=pt), iters=20)
                out.append(dict(k=f"expand_{m}_{'plain' if pitch is None else pitch}", us=t * 1e6, GBps=(54 + 12 * 54 + 12) * m / t / 1e9))
                del o
    for r in out:
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))


if __name__ == "__main__":
    main()
