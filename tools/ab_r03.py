#!/usr/bin/env python3
"""Round-3 A/B measurements through the shipped library (per-call `variant` of the *_ex entry points; RUBIKHIP_LIB selects
another build of the same sources for before/after rows).  One JSON line per row.

    python tools/ab_r03.py stepcode stepn hbm16 dense adiseg adipitch adirep widegroups wideskew

The dense-writer groups r03f-r03k and r03v of profiles/r03_ab.json came from experiment builds of the same sources (environment-
selected group counts, strided / paced / window forms) that are not in the tree any more; their rows are kept in profiles/r03_ab.json.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rubiks_cube_solver_amd import _lib, ops

TAG = os.environ.get("AB_TAG", os.path.basename(_lib.LIB_PATH))


def timeit(fn, iters=20, warm=3, reps=3):
    best = []
    for _ in range(reps):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / iters * 1e-3)
    best.sort()
    return best[len(best) // 2]


def emit(**kw):
    kw["lib"] = TAG
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in kw.items()}), flush=True)


def main():
    which = sys.argv[1:]
    n = 1 << 22
    if "stepcode" in which:
        a = ops.alloc_states(n, 3, "cuda")
        b = torch.empty_like(a)
        ops.fill_solved(a, n, 3)
        ops.scramble(a, n, 3, 20, seed=1234)
        acts = torch.randint(0, 12, (n,), dtype=torch.uint8, device="cuda")
        done = torch.empty(n, dtype=torch.uint8, device="cuda")
        rew = torch.empty(n, dtype=torch.float32, device="cuda")
        code = ops.alloc_code(n, 3, "cuda")
        buf = [a, b]
        for var in (0, 1, 2, 11, 12, 31, 32):
            def f0():
                ops.apply_moves(buf[0], buf[1], acts, n, 3, None, done, variant=var); buf.reverse()
            def f1():
                ops.apply_moves(buf[0], buf[1], acts, n, 3, rew, done, variant=var); buf.reverse()
            def f2():
                ops.apply_moves(buf[0], buf[1], acts, n, 3, rew, done, code, _lib.FMT_CODE, variant=var); buf.reverse()
            for name, f, bpc in (("step_done", f0, 110), ("step_reward_done", f1, 114), ("step_reward_done_code", f2, 134)):
                t = timeit(f, iters=50)
                emit(k=f"{name}_4M_var{var}", us=t * 1e6, frac=bpc * n / t / 8e12)
        del a, b, code, buf
    if "stepn" in which:
        for lg in (18, 19, 20, 21, 22, 23, 24):
            nn = 1 << lg
            a = ops.alloc_states(nn, 3, "cuda")
            b = torch.empty_like(a)
            ops.fill_solved(a, nn, 3)
            ops.scramble(a, nn, 3, 20, seed=1234)
            acts = torch.randint(0, 12, (nn,), dtype=torch.uint8, device="cuda")
            done = torch.empty(nn, dtype=torch.uint8, device="cuda")
            rew = torch.empty(nn, dtype=torch.float32, device="cuda")
            code = ops.alloc_code(nn, 3, "cuda")
            buf = [a, b]
            for var in (1, 2):
                def f0():
                    ops.apply_moves(buf[0], buf[1], acts, nn, 3, None, done, variant=var); buf.reverse()
                def f1():
                    ops.apply_moves(buf[0], buf[1], acts, nn, 3, rew, done, variant=var); buf.reverse()
                def f2():
                    ops.apply_moves(buf[0], buf[1], acts, nn, 3, rew, done, code, _lib.FMT_CODE, variant=var); buf.reverse()
                def f3():
                    ops.apply_moves(a, a, acts, nn, 3, rew, done, code, _lib.FMT_CODE, variant=var)
                def f4():
                    ops.apply_moves(a, a, acts, nn, 3, None, done, variant=var)
                for name, f, bpc in (("done", f0, 110), ("reward_done", f1, 114), ("reward_done_code", f2, 134), ("inplace_reward_done_code", f3, 134), ("inplace_done", f4, 110)):
                    t = timeit(f, iters=30 if lg < 24 else 10)
                    emit(k=f"step_{name}_2^{lg}_V{var}", us=t * 1e6, frac=bpc * nn / t / 8e12)
            del a, b, code, buf, acts, done, rew
    if "adipitch" in which:
        W, D = 100_000, 30
        for label, kw, bpu in (("codes", dict(parent_code=True, child_code=True), 1 + 12 + 13 * 20),
                               ("codes+parents", dict(parents=True, parent_code=True, child_code=True), 54 + 1 + 12 + 13 * 20),
                               ("stickers", dict(parents=True, children=True), 715)):
            for pitch in (512, 1024, 2048, 4096, 8192, 16384, 32768):
                pt, bufs = ops.adi_buffers(W, D, 3, "cuda", pitch, **kw)
                for v, parts, segs in ((1, 2, 1), (1, 1, 1), (2, 1, 1), (2, 1, 2), (2, 1, 3), (2, 1, 4), (1, 1, 2)):
                    var = segs * 1000000 + parts * 1000 + v
                    t = timeit(lambda: ops.adi_generate(W, D, 3, pt, "cuda", seed=2024, variant=var, **bufs), iters=5, warm=2)
                    emit(k=f"adi_{label}_pitch{pitch}_V{v}_parts{parts}_segs{segs}", us=t * 1e6, frac=bpu * W * D / t / 8e12)
                del bufs
    if "widegroups" in which:
        m = 1 << 20
        a = ops.alloc_states(m, 3, "cuda")
        b = torch.empty_like(a)
        ops.fill_solved(a, m, 3)
        ops.scramble(a, m, 3, 20, seed=1234)
        acts = torch.randint(0, 12, (m,), dtype=torch.uint8, device="cuda")
        done = torch.empty(m, dtype=torch.uint8, device="cuda")
        rew = torch.empty(m, dtype=torch.float32, device="cuda")
        code = ops.alloc_code(m, 3, "cuda")
        ops.encode(a, m, 3, code, _lib.FMT_CODE)
        for fmt, name, bpc in ((_lib.FMT_U8, "u8", 480), (_lib.FMT_BF16, "bf16", 960), (_lib.FMT_F32, "f32", 1920)):
            for bi in range(2):
                oh = torch.empty((m, 20, 24), dtype=_lib.dense_dtype(fmt), device="cuda")
                for groups16 in (0, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 32):
                    var = 300000 + groups16 * 1000
                    t = timeit(lambda: ops.apply_moves(a, b, acts, m, 3, rew, done, oh, fmt, variant=var), iters=10)
                    emit(k=f"fused_{name}", buf=bi, groups=groups16 * 16, us=t * 1e6, frac=(114 + bpc) * m / t / 8e12)
                    t = timeit(lambda: ops.onehot_from_code(code, m, 3, oh, variant=var), iters=10)
                    emit(k=f"c2d_{name}", buf=bi, groups=groups16 * 16, us=t * 1e6, frac=(20 + bpc) * m / t / 8e12)
                t = timeit(lambda: ops.apply_moves(a, b, acts, m, 3, rew, done, oh, fmt, variant=200000), iters=10)
                emit(k=f"fused_{name}", buf=bi, groups=-256, us=t * 1e6, frac=(114 + bpc) * m / t / 8e12)
                t = timeit(lambda: ops.onehot_from_code(code, m, 3, oh, variant=200000), iters=10)
                emit(k=f"c2d_{name}", buf=bi, groups=-256, us=t * 1e6, frac=(20 + bpc) * m / t / 8e12)
    if "wideskew" in which:
        for m in (1 << 20, 1_300_000, 1 << 21):
            a = ops.alloc_states(m, 3, "cuda")
            ops.fill_solved(a, m, 3)
            ops.scramble(a, m, 3, 20, seed=1234)
            code = ops.alloc_code(m, 3, "cuda")
            ops.encode(a, m, 3, code, _lib.FMT_CODE)
            for fmt, name, bpc in ((_lib.FMT_U8, "u8", 480), (_lib.FMT_BF16, "bf16", 960), (_lib.FMT_F32, "f32", 1920)):
                for bi in range(2):
                    oh = torch.empty((m, 20, 24), dtype=_lib.dense_dtype(fmt), device="cuda")
                    for g16 in (6, 7, 8, 10):
                        for skew in (0, 1, 3, 5, 7, 9):
                            var = 300000 + g16 * 1000 + skew * 10
                            t = timeit(lambda: ops.onehot_from_code(code, m, 3, oh, variant=var), iters=10)
                            emit(k=f"c2d_{name}", n=m, buf=bi, groups=g16 * 16, skew=skew, us=t * 1e6, frac=(20 + bpc) * m / t / 8e12)
                    t = timeit(lambda: ops.onehot_from_code(code, m, 3, oh, variant=200000), iters=10)
                    emit(k=f"c2d_{name}", n=m, buf=bi, groups=-256, skew=0, us=t * 1e6, frac=(20 + bpc) * m / t / 8e12)
                    del oh
            del a, code
    if "expandstream" in which:
        m = 1 << 20
        src = ops.alloc_states(m, 3, "cuda")
        ops.fill_solved(src, m, 3)
        ops.scramble(src, m, 3, 20, seed=5)
        for rep in range(3):
            o = ops.expand_buffers(m, 3, "cuda", children=True, codes=False)
            pitch = o["children"].shape[-1]
            for label, var in (("k_expand_2048_waves", 800), ("stream128", 100), ("stream192", 200), ("stream256", 300), ("stream384", 400), ("stream512", 500),
                               ("stream768", 600), ("stream1024", 700), ("default", 0)):
                t = timeit(lambda: ops.expand_children(src, m, 3, o["children"], o["child_solved"], pitch=pitch, variant=var), iters=20)
                emit(k=f"expand_1M_{label}", rep=rep, us=t * 1e6, frac=(54 + 12 * 54 + 12) * m / t / 8e12, kernel=_lib.describe(_lib.OP_EXPAND, 3, m, outputs=_lib.OUT_STATES | _lib.OUT_FLAGS, variant=var))
            del o
    if "adirep" in which:
        W, D = 100_000, 30
        for rep in range(3):
            for label, kw, bpu in (("codes", dict(parent_code=True, child_code=True), 1 + 12 + 13 * 20),
                                   ("codes+parents", dict(parents=True, parent_code=True, child_code=True), 54 + 1 + 12 + 13 * 20)):
                pt, bufs = ops.adi_buffers(W, D, 3, "cuda", **kw)
                for name, var in (("default", 0), ("V1_parts2_segs1", 1002001), ("V1_parts1_segs2", 2001001), ("V2_parts1_segs3", 3001002), ("V1_parts1_segs1", 1001001)):
                    t = timeit(lambda: ops.adi_generate(W, D, 3, pt, "cuda", seed=2024, variant=var, **bufs), iters=5, warm=2)
                    emit(k=f"adi_{label}_{name}", rep=rep, us=t * 1e6, frac=bpu * W * D / t / 8e12,
                         kernel=_lib.describe(_lib.OP_ADI, 3, W, D, outputs=_lib.OUT_CODE | _lib.OUT_FLAGS, variant=var))
                del bufs
                torch.cuda.empty_cache()
    if "hbm16" in which:
        n16 = 1 << 24
        a = ops.alloc_states(n16, 3, "cuda")
        b = torch.empty_like(a)
        ops.fill_solved(a, n16, 3)
        ops.scramble(a, n16, 3, 20, seed=1234)
        acts16 = torch.randint(0, 12, (n16,), dtype=torch.uint8, device="cuda")
        done16 = torch.empty(n16, dtype=torch.uint8, device="cuda")
        buf = [a, b]
        for var in (0, 1, 2):
            def f():
                ops.apply_moves(buf[0], buf[1], acts16, n16, 3, None, done16, variant=var); buf.reverse()
            t = timeit(f, iters=20)
            emit(k=f"step_done_16M_var{var}", us=t * 1e6, frac=110 * n16 / t / 8e12)
        del a, b, buf
    if "dense" in which:
        m = 1 << 20
        a = ops.alloc_states(m, 3, "cuda")
        b = torch.empty_like(a)
        ops.fill_solved(a, m, 3)
        ops.scramble(a, m, 3, 20, seed=1234)
        acts = torch.randint(0, 12, (m,), dtype=torch.uint8, device="cuda")
        done = torch.empty(m, dtype=torch.uint8, device="cuda")
        rew = torch.empty(m, dtype=torch.float32, device="cuda")
        code = ops.alloc_code(m, 3, "cuda")
        ops.encode(a, m, 3, code, _lib.FMT_CODE)
        for rep in range(2):
            for fmt, name, bpc in ((_lib.FMT_U8, "u8", 480), (_lib.FMT_F16, "f16", 960), (_lib.FMT_BF16, "bf16", 960), (_lib.FMT_F32, "f32", 1920)):
                oh = torch.empty((m, 20, 24), dtype=_lib.dense_dtype(fmt), device="cuda")
                t = timeit(lambda: ops.apply_moves(a, b, acts, m, 3, rew, done, oh, fmt), iters=10)
                emit(k=f"step_dense_{name}_1M", rep=rep, us=t * 1e6, frac=(114 + bpc) * m / t / 8e12)
                t = timeit(lambda: ops.onehot_from_code(code, m, 3, oh), iters=10)
                emit(k=f"code_to_dense_{name}_1M", rep=rep, us=t * 1e6, frac=(20 + bpc) * m / t / 8e12)
                del oh
    if "adiseg" in which:
        W, D = 100_000, 30
        for label, kw, bpu in (("codes+parents", dict(parents=True, parent_code=True, child_code=True), 54 + 1 + 12 + 13 * 20),
                               ("codes", dict(parent_code=True, child_code=True), 1 + 12 + 13 * 20)):
            pt, bufs = ops.adi_buffers(W, D, 3, "cuda", **kw)
            for v in (1, 2):
                for parts in (1, 2):
                    for segs in (1, 2, 3, 4, 6, 8, 12, 16):
                        var = segs * 1000000 + parts * 1000 + v
                        t = timeit(lambda: ops.adi_generate(W, D, 3, pt, "cuda", seed=2024, variant=var, **bufs), iters=5, warm=2)
                        emit(k=f"adi_{label}_V{v}_parts{parts}_segs{segs}", us=t * 1e6, frac=bpu * W * D / t / 8e12)
            t = timeit(lambda: ops.adi_generate(W, D, 3, pt, "cuda", seed=2024, **bufs), iters=5, warm=2)
            emit(k=f"adi_{label}_default", us=t * 1e6, frac=bpu * W * D / t / 8e12)
            del bufs
        pt, bufs = ops.adi_buffers(W, D, 3, "cuda", parents=True, children=True)
        for segs in (1, 2, 4):
            for v, parts in ((2, 1), (1, 1), (1, 2)):
                var = segs * 1000000 + parts * 1000 + v
                t = timeit(lambda: ops.adi_generate(W, D, 3, pt, "cuda", seed=2024, variant=var, **bufs), iters=5, warm=2)
                emit(k=f"adi_stickers_V{v}_parts{parts}_segs{segs}", us=t * 1e6, frac=715 * W * D / t / 8e12)
        del bufs


if __name__ == "__main__":
    main()
