#!/usr/bin/env python3
"""Differential fuzz of round 4's kernels on random shapes (GPU): every front-writer shape and the workspace routes against the
64-cube-tile form / the one-launch kernels, the family record against the picked codes.  Seeded; prints one line per failure and a
summary.   python tools/fuzz_round4.py [iterations] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rubiks_cube_solver_amd import _lib as L, ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda"
fmts = ((L.FMT_U8, torch.uint8), (L.FMT_F16, torch.float16), (L.FMT_BF16, torch.bfloat16), (L.FMT_F32, torch.float32))
fails = 0


def sizes():
    kind = rng.integers(0, 5)
    if kind == 0:
        return int(rng.integers(1, 600))
    if kind == 1:
        base = int(rng.choice([256, 512, 3840, 4096, 16384, 32768, 65536, 131072, 262144]))
        return max(1, base + int(rng.integers(-9, 10)))
    if kind == 2:
        return int(rng.integers(1, 40000))
    if kind == 3:
        return int(rng.integers(120000, 140000))
    return int(rng.integers(1, 400000))


for it in range(iters):
    n = sizes()
    pitch = None if rng.integers(0, 2) else (L.pitch_for(n) if rng.integers(0, 2) else int(rng.choice([512, 1024, 4096])))
    st = ops.alloc_states(n, 3, dev, pitch=pitch)
    ops.fill_solved(st, n, 3)
    ops.scramble(st, n, 3, int(rng.integers(0, 25)), seed=int(rng.integers(0, 1 << 30)))
    code = ops.alloc_code(n, 3, dev, pitch=pitch)
    ops.encode(st, n, 3, code, L.FMT_CODE)
    fmt, dt = fmts[int(rng.integers(0, 4))]
    ref = torch.full((n, 20, 24), 3, dtype=dt, device=dev)
    ops.onehot_from_code(code, n, 3, ref, variant=100000)
    for form in (0, 400031, 400032, 400034, 400041, 400042, 400044, 400020):
        oh = torch.full((n + 2, 20, 24), 3, dtype=dt, device=dev)
        ops.onehot_from_code(code, n, 3, oh[:n], variant=form)
        if not torch.equal(oh[:n], ref) or float(oh[n:].float().min()) != 3.0:
            fails += 1
            print("FAIL c2d", n, pitch, fmt, form, flush=True)
    # fused: default (workspace where used) against the forced one-launch kernel; in place too; encode-only
    acts = torch.randint(0, 13, (n,), dtype=torch.uint8, device=dev)
    outs = []
    for variant in (0, 200000):
        dst = torch.zeros_like(st)
        oh = torch.full((n, 20, 24), 3, dtype=dt, device=dev)
        rew = torch.zeros(n, dtype=torch.float32, device=dev)
        done = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        ops.apply_moves(st, dst, acts, n, 3, rew, done, oh, fmt, variant=variant)
        outs.append((ops.to_aos(dst, n).clone(), oh, rew, done))
    if not all(torch.equal(a, b) for a, b in zip(*outs)):
        fails += 1
        print("FAIL fused", n, pitch, fmt, flush=True)
    work = st.clone()
    oh = torch.full((n, 20, 24), 3, dtype=dt, device=dev)
    ops.apply_moves(work, work, acts, n, 3, None, None, oh, fmt)
    enc = torch.full((n, 20, 24), 3, dtype=dt, device=dev)
    ops.encode(work, n, 3, enc, fmt)
    if not (torch.equal(ops.to_aos(work, n), outs[0][0]) and torch.equal(oh, outs[0][1]) and torch.equal(enc, outs[0][1])):
        fails += 1
        print("FAIL in place / encode", n, pitch, fmt, flush=True)
    # family record against the picked codes (small depth), both cube sizes
    if it % 4 == 0:
        for cs in (3, 2):
            w, d = min(n, 60000), int(rng.integers(1, 6))
            A = 12 if cs == 3 else 6
            apitch = None if rng.integers(0, 2) else int(rng.choice([512, 2048]))
            variant = int(rng.choice([0, 1, 2, 2001001, 3001002]))
            seed = int(rng.integers(0, 1 << 30))
            pt, cb = ops.adi_buffers(w, d, cs, dev, pitch=apitch, parents=True, parent_code=True, child_code=True)
            ops.adi_generate(w, d, cs, pt, dev, seed=seed, stream_id=3, **cb)
            pt2, fb = ops.adi_buffers(w, d, cs, dev, pitch=apitch, parents=True, family=True)
            ops.adi_generate(w, d, cs, pt2, dev, seed=seed, stream_id=3, variant=variant, **fb)
            nf, rows = L.family_layout(cs)
            aos = lambda t, k: ops.to_aos(t[k], w)                               # pad columns of the buffers are uninitialised: compare cubes only
            ok = all(torch.equal(aos(fb["parents"], k), aos(cb["parents"], k)) for k in range(d))
            ok = ok and torch.equal(fb["child_solved"][..., :w], cb["child_solved"][..., :w]) and torch.equal(fb["actions_out"][:, :w], cb["actions_out"][:, :w])
            prow = torch.from_numpy(rows[A].astype(np.int64)).to(dev)
            ok = ok and all(torch.equal(ops.to_aos(fb["family"][k].index_select(1, prow), w), aos(cb["parent_code"], k)) for k in range(d))
            for a in range(A):
                arow = torch.from_numpy(rows[a].astype(np.int64)).to(dev)
                got = torch.stack([ops.to_aos(fb["family"][k].index_select(1, arow), w) for k in range(d)])
                want = torch.stack([ops.to_aos(cb["child_code"][k, a], w) for k in range(d)])
                ok = ok and torch.equal(got, want)
            if cs == 3:
                p = fb["actions_out"].shape[1]
                dense = torch.full((13 * p, 20, 24), 3, dtype=dt, device=dev)
                k = int(rng.integers(0, d))
                ops.onehot_from_family(fb["family"][k], w, 3, dense, block_stride=p)
                want = torch.full((13 * p, 20, 24), 3, dtype=dt, device=dev)
                for a in range(12):
                    ops.onehot_from_code(cb["child_code"][k, a], w, 3, want[a * p:a * p + w], variant=100000)
                ops.onehot_from_code(cb["parent_code"][k], w, 3, want[12 * p:12 * p + w], variant=100000)
                ok = ok and torch.equal(dense, want)                             # incl. the untouched pad cubes (still 3)
            if not ok:
                fails += 1
                print("FAIL family", cs, w, d, apitch, variant, flush=True)
    if L.read_status() != 0:
        fails += 1
        print("FAIL status", n, flush=True)
    if it % 20 == 0:
        print("iteration", it, "n", n, "fails", fails, flush=True)
    del st, code, ref
print("fuzz done:", iters, "iterations,", fails, "failures")
sys.exit(1 if fails else 0)
