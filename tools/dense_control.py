#!/usr/bin/env python3
"""Round-4 control experiment for the dense one-hot writers (task 3): what bounds them?

ONE process, the SAME output buffers, at 2^20 cubes (bf16: 1.0 GB, f32: 2.0 GB per buffer), NB separately allocated buffers per format:

  (i)   hipMemsetAsync of the buffer, and torch's fill kernel                       -- what the box's write path gives a plain fill
  (ii)  the writers with the LDS read replaced by a register value (RC_DENSE_CTRL 1) and as pure store kernels
        (RC_DENSE_CTRL 2: no code loads, no LDS, no barriers)                        -- the SHAPE without the LDS -> store dependency
  (iii) the real kernels: round 3's loop (one store in flight per LDS round trip, RC_DENSE_PIPE 1) and the software-pipelined
        loop (RC_DENSE_PIPE 2 / 4 / 8), other cache policies of the dense stores (RC_DENSE_AUX), wide group counts 96 .. 512

The variants are builds of the SAME sources with -D switches (tools/dense_control.py --build puts them into tools/_ctl/), loaded
side by side with ctypes; the shipped librubikhip.so is row "shipped".  One JSON line per measurement on stdout.

    python tools/dense_control.py --build            # here (hipcc cross-compiles), before gpurun
    python tools/dense_control.py [--quick]          # on the GPU box
    python tools/dense_control.py --pmc              # few launches per kernel, for `rocprofv3 --pmc ... -- python3 tools/dense_control.py --pmc`
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CTL = os.path.join(ROOT, "tools", "_ctl")
SRC = os.path.join(ROOT, "rubiks-cube-solver_amd", "csrc", "rubikhip.hip")

BUILDS = {                      # name -> -D switches
    "pipe1": ["-DRC_DENSE_PIPE=1"],
    "pipe2": ["-DRC_DENSE_PIPE=2"],
    "pipe8": ["-DRC_DENSE_PIPE=8"],
    "ctrl1_pipe1": ["-DRC_DENSE_PIPE=1", "-DRC_DENSE_CTRL=1"],
    "ctrl2_pipe1": ["-DRC_DENSE_PIPE=1", "-DRC_DENSE_CTRL=2"],
    "ctrl2_pipe4": ["-DRC_DENSE_PIPE=4", "-DRC_DENSE_CTRL=2"],
    "aux0_pipe4": ["-DRC_DENSE_PIPE=4", "-DRC_DENSE_AUX=0"],
    "aux2_pipe4": ["-DRC_DENSE_PIPE=4", "-DRC_DENSE_AUX=2"],
    "aux17_pipe4": ["-DRC_DENSE_PIPE=4", "-DRC_DENSE_AUX=17"],
}


def build(jobs=4):
    os.makedirs(CTL, exist_ok=True)
    procs = []
    for name, defs in BUILDS.items():
        out = os.path.join(CTL, f"librubikhip_{name}.so")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *defs, "-o", out, SRC]
        procs.append((name, subprocess.Popen(cmd)))
        if len(procs) >= jobs:
            n, p = procs.pop(0)
            assert p.wait() == 0, n
    for n, p in procs:
        assert p.wait() == 0, n
    print("built", sorted(os.listdir(CTL)))


def sizes():
    """Every dense form of the shipped library over batch sizes (two output buffers each): where do the defaults belong?"""
    import torch
    from rubiks_cube_solver_amd import _lib, ops
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    _lib.init(dev)
    sp = _lib.stream_ptr(dev)

    def timeit(fn, iters=10, warm=3, reps=3):
        vals = []
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
            vals.append(e0.elapsed_time(e1) / iters * 1e3)
        return sorted(vals)[len(vals) // 2]

    for log2n in (15, 16, 17, 18, 19, 20, 21, 22):
        for n in ((1 << log2n), (1 << log2n) + (1 << (log2n - 1)) + 1234) if log2n in (17, 19) else ((1 << log2n),):
            st = ops.alloc_states(n, 3, dev)
            st2 = torch.empty_like(st)
            ops.fill_solved(st, n, 3)
            ops.scramble(st, n, 3, 20, seed=1234)
            code = ops.alloc_code(n, 3, dev)
            ops.encode(st, n, 3, code, _lib.FMT_CODE)
            acts = torch.randint(0, 12, (n,), dtype=torch.uint8, device=dev)
            rew = torch.empty(n, dtype=torch.float32, device=dev)
            done = torch.empty(n, dtype=torch.uint8, device=dev)
            for name, dt, fmt, bpc in (("bf16", torch.bfloat16, _lib.FMT_BF16, 960), ("f32", torch.float32, _lib.FMT_F32, 1920), ("u8", torch.uint8, _lib.FMT_U8, 480)):
                for bi in range(2):
                    oh = torch.empty((n, 20, 24), dtype=dt, device=dev)
                    row = dict(n=n, fmt=name, buf=bi)
                    for what, v in (("c2d_default", 0), ("c2d_tile64", 100000), ("c2d_256", 200000), ("c2d_wide", 300000), ("front_gatherF1", 400031),
                                    ("front_ldsF1", 400041), ("front_ldsF2", 400042)):
                        t = timeit(lambda: ops.onehot_from_code(code, n, 3, oh, variant=v))
                        row[what] = round((bpc + 20) * n / (t * 1e-6) / 8e12, 3)
                    for what, v in (("fused_default(ws)", 0), ("fused_tile64", 100000), ("fused_256", 200000)):
                        t = timeit(lambda: ops.apply_moves(st, st2, acts, n, 3, rew, done, oh, fmt, variant=v))
                        row[what] = round((bpc + 114) * n / (t * 1e-6) / 8e12, 3)
                    print(json.dumps(row), flush=True)
                    del oh
            del st, st2, code
            torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--buffers", type=int, default=4)
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--sizes", action="store_true", help="size sweep 2^15 .. 2^22 of the shipped library's dense forms (dispatch thresholds)")
    args = ap.parse_args()
    if args.build:
        return build()
    if args.sizes:
        return sizes()

    import torch
    from rubiks_cube_solver_amd import _lib, ops

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n = 1 << args.log2n
    hip = ctypes.CDLL("libamdhip64.so")       # the runtime torch already loaded
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]

    libs = {"shipped": _lib.lib()}
    _lib.init(dev)
    for name in BUILDS:
        path = os.path.join(CTL, f"librubikhip_{name}.so")
        if os.path.exists(path) and not args.quick:
            L = ctypes.CDLL(path)
            if not hasattr(L, "rc_onehot_from_family"):
                print(f"# {path} is a build of older sources: rebuild with --build", file=sys.stderr)
                continue
            _lib._declare(L)
            assert L.rc_init(0) == 0
            libs[name] = L

    def emit(**kw):
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in kw.items()}), flush=True)

    iters, warm, reps = (3, 1, 1) if args.pmc else (10, 3, 3)

    def timeit(fn):
        vals = []
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
            vals.append(e0.elapsed_time(e1) / iters * 1e3)
        vals.sort()
        return vals[len(vals) // 2]

    # inputs: 20-move scrambles, their compact codes, actions
    st = ops.alloc_states(n, 3, dev)
    st2 = torch.empty_like(st)
    ops.fill_solved(st, n, 3)
    ops.scramble(st, n, 3, 20, seed=1234)
    code = ops.alloc_code(n, 3, dev)
    ops.encode(st, n, 3, code, _lib.FMT_CODE)
    acts = torch.randint(0, 12, (n,), dtype=torch.uint8, device=dev)
    rew = torch.empty(n, dtype=torch.float32, device=dev)
    done = torch.empty(n, dtype=torch.uint8, device=dev)
    sp = _lib.stream_ptr(dev)
    cp = code.shape[-1]
    pin, pout = st.shape[-1], st2.shape[-1]

    def c2d(L, oh, fmt, variant):
        rc = L.rc_onehot_from_code_ex(code.data_ptr(), n, cp, 3, oh.data_ptr(), fmt, sp, variant)
        assert rc == 0, L.rc_last_error()

    def fused(L, oh, fmt, variant):
        rc = L.rc_apply_moves_ex(st.data_ptr(), st2.data_ptr(), acts.data_ptr(), n, pin, pout, 3, rew.data_ptr(), done.data_ptr(),
                                 oh.data_ptr(), fmt, 0, sp, variant)
        assert rc == 0, L.rc_last_error()

    fmts = [("bf16", torch.bfloat16, _lib.FMT_BF16, 960), ("f32", torch.float32, _lib.FMT_F32, 1920)]
    if not args.pmc:
        fmts.append(("u8", torch.uint8, _lib.FMT_U8, 480))
    group_fields = [6, 7, 8, 10, 12, 14, 16, 20, 24, 32]          # wide form: wanted groups / 16 (thousands field of `variant`)
    emit(what="header", n=n, device=torch.cuda.get_device_name(0), libs=sorted(libs), iters=iters, reps=reps)
    for name, dt, fmt, bpc in fmts:
        bufs = [torch.empty((n, 20, 24), dtype=dt, device=dev) for _ in range(args.buffers)]
        for bi, oh in enumerate(bufs):
            nbytes = oh.numel() * oh.element_size()
            common = dict(fmt=name, buf=bi, addr=hex(oh.data_ptr()), out_bytes=nbytes)

            def frac(us, extra):
                return (nbytes + extra * n) / (us * 1e-6) / 8e12

            t = timeit(lambda: hip.hipMemsetAsync(oh.data_ptr(), 0, nbytes, sp))
            emit(what="hipMemsetAsync", us=t, frac=frac(t, 0), **common)
            t = timeit(lambda: oh.fill_(1))
            emit(what="torch.fill_", us=t, frac=frac(t, 0), **common)
            L = libs["shipped"]
            ws = ops.workspace(dev, max(16, L.rc_workspace_bytes(_lib.OP_STEP, 3, n, fmt)))

            def fused_ws(oh):
                rc = L.rc_apply_moves_ws(st.data_ptr(), st2.data_ptr(), acts.data_ptr(), n, pin, pout, 3, rew.data_ptr(), done.data_ptr(),
                                         oh.data_ptr(), fmt, 0, ws.data_ptr(), ws.numel(), sp)
                assert rc == 0, L.rc_last_error()

            for what, fn, extra in (("c2d_front_xcd (default)", lambda: c2d(L, oh, fmt, 0), 20), ("c2d_front_linear", lambda: c2d(L, oh, fmt, 400020), 20),
                                    ("c2d_front_xcd F=1", lambda: c2d(L, oh, fmt, 400001), 20), ("c2d_front_xcd F=2", lambda: c2d(L, oh, fmt, 400002), 20),
                                    ("c2d_front_xcd F=4", lambda: c2d(L, oh, fmt, 400004), 20),
                                    ("fused_ws: step+code, front (rc_apply_moves_ws)", lambda: fused_ws(oh), 114)):
                t = timeit(fn)
                emit(what=what, lib="shipped", us=t, frac=frac(t, extra), **common)
            # how the front writer fetches its code bytes, on the SHIPPED library: RC_VARIANT_DENSE_FRONT_FETCH 3 = a byte gather per
            # lane, 4 = one load of wave 0 + LDS (round 4 compared a -DRC_FRONT_LDS=0 build; that macro is gone, the variant replaced it)
            for what, v in (("c2d_front_xcd fetch=gather", 400030), ("c2d_front_xcd fetch=lds", 400040)):
                t = timeit(lambda: c2d(libs["shipped"], oh, fmt, v))
                emit(what=what, lib="shipped", us=t, frac=frac(t, 20), **common)
            for lname, L in libs.items():
                if args.pmc and lname not in ("shipped", "pipe1", "ctrl1_pipe1", "ctrl2_pipe1", "ctrl2_pipe4"):
                    continue
                if args.quick and lname != "shipped":
                    continue
                t = timeit(lambda: c2d(L, oh, fmt, 300000))
                emit(what="c2d_wide112", lib=lname, us=t, frac=frac(t, 20), **common)
                t = timeit(lambda: c2d(L, oh, fmt, 200000))
                emit(what="c2d_256thread", lib=lname, us=t, frac=frac(t, 20), **common)
                t = timeit(lambda: c2d(L, oh, fmt, 100000))
                emit(what="c2d_tile64", lib=lname, us=t, frac=frac(t, 20), **common)
                t = timeit(lambda: fused(L, oh, fmt, 100000))
                emit(what="fused_tile64", lib=lname, us=t, frac=frac(t, 114), **common)
                t = timeit(lambda: fused(L, oh, fmt, 0))
                emit(what="fused_step_dense", lib=lname, us=t, frac=frac(t, 114), **common)
                if args.pmc or args.quick:
                    continue
                if lname in ("shipped", "pipe1", "ctrl2_pipe4", "ctrl2_pipe1", "pipe8"):
                    for f in group_fields:
                        t = timeit(lambda: c2d(L, oh, fmt, 300000 + f * 1000))
                        emit(what=f"c2d_wide{f * 16}", lib=lname, us=t, frac=frac(t, 20), **common)
            # correctness on this buffer: every form equals the 64-cube-tile form, the workspace route equals the one-launch kernel
            ref = torch.empty_like(oh)
            c2d(libs["shipped"], ref, fmt, 100000)
            for what, v in (("front_xcd", 0), ("front_linear", 400020), ("front_F2", 400002), ("front_F4", 400004), ("wide", 300000)):
                oh.zero_()
                c2d(libs["shipped"], oh, fmt, v)
                emit(what=f"check_{what}_equals_tile64", ok=bool(torch.equal(oh.view(torch.uint8), ref.view(torch.uint8))), **common)
            fused(libs["shipped"], ref, fmt, 0)
            s_ref, r_ref, d_ref = st2.clone(), rew.clone(), done.clone()
            oh.zero_(); st2.zero_(); rew.zero_(); done.zero_()
            fused_ws(oh)
            emit(what="check_fused_ws_equals_one_launch", ok=bool(torch.equal(oh.view(torch.uint8), ref.view(torch.uint8)) and torch.equal(st2, s_ref)
                                                                 and torch.equal(rew, r_ref) and torch.equal(done, d_ref)), **common)
            del ref
        del bufs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
