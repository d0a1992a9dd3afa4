#!/usr/bin/env python3
"""Headline benchmark: cube-move steps/s, 3x3x3, batch 4M per GPU (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--cubes-per-gpu C] [--no-cpu] [--no-configs] [--force-dist] [--backend nccl|gloo]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--cubes-per-gpu 1048576]

One bench "step" = one pass of the hot path over one batch: ONE rc_apply_moves launch that moves
C cubes per GPU (each by its own random face turn) and writes their solved flags, ping-ponging two
HBM-resident state buffers.  C defaults to 2^22 (the metric's batch: working set 453 MB > the 256 MB Infinity
Cache); --cubes-per-gpu 1048576 is BASELINE config 4's shape (1M cubes per GPU x N GPUs).
Inputs are resident in HBM before the timed region.  value = cubes * K * N / max-over-ranks time of the K timed steps.
Every rank owns its own batch and RNG stream (stream_id = rank); there is no collective on the
env path ("scaling": "weak").

OUTPUT.  stdout carries exactly ONE line, printed last by rank 0: the COMPACT record (compact_record(): the contract's keys,
`roofline`, `cpu_baseline`, ten scalar `extras`, under 4 KB, no string over 120 characters -- round 5's 23.5 KB line came back from
the driver unparsed).  The FULL record (every per-config roofline record, the notes, per-rank details) is written to --full-out
(default bench_full.json next to this file; the builder's copies live under profiles/).  In the distributed branch the full record's
`per_gpu` carries each rank's stream_id, device index, PCI bus id, rate and a sha256 of its first 65536 scrambled cubes (the tests
compare it with the oracle's (seed, rank) stream); the compact line carries the short form and `config.distinct_devices`.  One rank
per GPU: a launch with more ranks than visible devices is REFUSED unless --share-gpu marks it as a rehearsal.  At N > 1 the other
configs are skipped unless --configs is given, so rank 0 exits with its peers.
The distributed branch (process group, barrier, all_gather, MAX over ranks) runs for N > 1, under a launcher at N = 1
(torch.distributed.run --nproc-per-node 1) and with --force-dist, so the code the 2/4/8-GPU runs execute can be exercised on one GPU.

roofline.launch_us is the MEDIAN of R >= 7 further batches of K launches, each bracketed by HIP events on the launch stream, run
right after the timed region (launch_us_min / _max beside it, launch_us_timed_region = the K timed steps themselves).

roofline.per_config and "configs" (rank 0, after the process group is gone): the other BASELINE.json workloads, each timed OUTSIDE
the headline's timed region with HIP events on the launch stream (median of 3 batches) and with a roofline record of its own
(kernel = what the library's own dispatch reports, algorithmic bytes per launch, achieved GB/s, fraction of the 8 TB/s HBM peak):
config 2 (1M cubes, move + reward + done), the 4M step in place, with the reward, with the fused compact code, the 16M-cube step
whose 1.8 GB ping-pong defeats the Infinity Cache (the HBM-only point: roofline.frac_hbm_only), the fused dense one-hot (f32 /
bf16 / u8), code -> dense (f32 / bf16 / f16 / u8), 2x2x2 step / expansion / dense, config 3 (ADI 100k walks x 30, 715 B per (walk,
depth)) and its code / family forms, the family -> dense-block launch of the ADI pipeline, the 1M-parent expansion.  Then the loops
the reference actually runs, end to end on the host clock: `adi_pipeline` (get_random_samples batched: 200 x 30 = the reference's
own size, 20k x 30, 100k x 30, and the hipGraph replay), `config5_mcts_4096_leaves` (config 5: us per MCTS step eager / hipGraph,
the lockstep search's device step + transfers and whole simulations split select / device / update), `rollout`
(greedy validation rollouts), `reset_seeds_1M_k30` (reset(seed, 30) for 1M envs, numpy's legacy generator on the device) and the
batch-1 facade latency.  --no-configs skips them.  RC_BENCH_DRY=1 (with --backend gloo) runs the N-rank plumbing without a GPU.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CUBES = 1 << 22
CUBE = 3
BYTES_PER_STEP = 54 + 54 + 1 + 1  # stickers R + W, action, done flag
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBPS = 6300.0      # same guide: "8 TB/s peak (spec); about 6.3 TB/s achievable"


def cpu_share():
    """Cores this job may use: the smaller of its CPU affinity and its cgroup CPU quota; a one-GPU box of this pool hands a job
    16 cores without pinning it (affinity still lists every host thread), so without a visible quota that figure is used."""
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(period)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, q // period)
        except Exception:
            pass
    share = min(affinity, quota) if quota else min(affinity, int(os.environ.get("RC_CPU_SHARE", "16")))
    return share, affinity, quota


def _numpy_env_worker(seconds):
    """One process = one core: the reference-style numpy env stepping one cube for `seconds` (spawned, never touches the GPU)."""
    import numpy as np
    from oracle.oracle_np import OracleCubeEnv
    env = OracleCubeEnv(None, CUBE)
    acts = np.random.default_rng(os.getpid()).integers(0, 12, 500)
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for a in acts:
            env.step(int(a))
        k += len(acts)
    return k / (time.perf_counter() - t0)


def numpy_env_all_cores(seconds=1.5, procs=None):
    import multiprocessing as mp
    # the GPU box gives a one-GPU job a CPU share of 16 cores: size the pool to that, not to the host's 256 threads
    procs = procs or cpu_share()[0]
    # the workers are plain CPU processes: do not let a profiler's preload (rocprofv3 sets LD_PRELOAD) make each of
    # them open the GPU
    saved = {k: os.environ.pop(k) for k in list(os.environ) if k == "LD_PRELOAD" or k.startswith(("ROCP", "ROCPROF"))}
    try:
        with mp.get_context("spawn").Pool(procs) as pool:
            rates = pool.map(_numpy_env_worker, [seconds] * procs)
    finally:
        os.environ.update(saved)
    return float(sum(rates)), procs


def cpu_baseline(budget_s=5.0):
    """The C oracle (kind "port") timed on this box's host cores, same workload shape, bounded sample."""
    import numpy as np
    from oracle.oracle_np import Oracle, OracleCubeEnv

    orc = Oracle()
    share, affinity, quota = cpu_share()
    omp_threads = max(1, min(orc.max_threads(), share))          # the cores this job may use, not the host's thread count
    n = 1 << 20
    rng = np.random.default_rng(1)
    walk = rng.integers(0, 12, (n, 1), dtype=np.uint8)
    states = orc.adi(CUBE, n, 1, actions_in=walk, want_children=False, threads=omp_threads)["parents"][:, 0]
    acts = rng.integers(0, 12, n, dtype=np.uint8)
    res = {}
    for label, threads in (("single", 1), ("all", omp_threads)):
        t1 = orc.time_steps(CUBE, states, acts, 1, False, threads)
        iters = max(1, min(2000, int(budget_s / max(t1, 1e-4))))
        t = orc.time_steps(CUBE, states, acts, iters, False, threads)
        res[label] = dict(steps_per_s=n * iters / t, threads=threads, iters=iters, seconds=t)
    # reference-style numpy env (one cube at a time, Python loop): the reference's own structure
    env = OracleCubeEnv(None, CUBE)
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < 2.0:
        for a in acts[:500]:
            env.step(int(a))
        k += 500
    np_rate = k / (time.perf_counter() - t0)
    try:
        np_all, np_procs = numpy_env_all_cores(1.5 if budget_s >= 4 else 0.3, None if budget_s >= 4 else 2)
    except Exception:  # a sandbox without process spawning: report what we have
        np_all, np_procs = None, 0
    return {
        "value": res["all"]["steps_per_s"], "unit": "steps/s", "cores": res["all"]["threads"], "kind": "port",
        "affinity_cores": affinity, "cgroup_cpu_quota": quota, "cpu_share_used": share, "host_logical_cores": os.cpu_count(),
        "sample": f"C oracle (oracle/rc_oracle.c, OpenMP, {res['all']['threads']} threads = this job's CPU share) step = move+solved flag "
                  f"on 2^20 cubes x {res['all']['iters']} passes ({res['all']['seconds']:.1f} s)",
        "sample_short": f"C oracle, OpenMP x{res['all']['threads']}: move + solved flag, 2^20 cubes x {res['all']['iters']} passes ({res['all']['seconds']:.1f} s)",
        "single_core_steps_per_s": res["single"]["steps_per_s"],
        "numpy_env_1core_steps_per_s": np_rate,
        "numpy_env_allcores_steps_per_s": np_all, "numpy_env_processes": np_procs,
        "numpy_env_note": "reference-style per-cube numpy env (oracle_np.OracleCubeEnv.step: move + one-hot + solved), 2 s sample",
    }


def other_configs(torch, ops, _lib, dev, acts):
    """BASELINE.json configs 2, 3, 5, the fused-output variants and (pipeline_configs) the end-to-end loops, each with its own record.
    Timed with HIP events on the launch stream, outside the headline's timed region.  `kernel` of every record is what
    rc_describe_dispatch reports for exactly that call (the library's own pick_* functions), never a literal."""
    recs = []
    D = _lib.describe
    ST, CODE, FLAGS, REW, INPL = _lib.OUT_STATES | _lib.OUT_DONE, _lib.OUT_CODE, _lib.OUT_FLAGS, _lib.OUT_REWARD, _lib.OUT_INPLACE

    def timed(fn, iters, warm=5, reps=3):
        """seconds per launch: the median of `reps` event-timed batches of `iters` launches"""
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        vals = []
        for _ in range(reps):
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            for _ in range(iters):
                fn()
            s1.record()
            torch.cuda.synchronize()
            vals.append(s0.elapsed_time(s1) / iters * 1e-3)
        return sorted(vals)[len(vals) // 2]

    def rec(short, name, kernel, units, unit_name, bytes_per_unit, t, note=None, bound="hbm"):
        achieved = bytes_per_unit * units / t / 1e9
        r = {"short": short, "config": name, "kernel": kernel, "launch_us": t * 1e6, "value": units / t, "unit": f"{unit_name}/s",
             "roofline": {"bound": bound, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                          "algorithmic_bytes_per_launch": bytes_per_unit * units, "bytes_per_unit": bytes_per_unit}}
        if note:
            r["note"] = note
        recs.append(r)
        return r

    m = 1 << 20
    n = N_CUBES
    acts = acts[:n] if acts.numel() >= n else torch.randint(0, 12, (n,), device=dev, dtype=torch.uint8)
    rew = torch.empty(n, dtype=torch.float32, device=dev)
    done = torch.empty(n, dtype=torch.uint8, device=dev)
    # config 2: batch 1M, single apply_move + reward (+ done) -- the 113 MB ping-pong working set sits in the Infinity Cache
    a1, b1 = ops.alloc_states(m, CUBE, dev), ops.alloc_states(m, CUBE, dev)
    ops.fill_solved(a1, m, CUBE)
    ops.scramble(a1, m, CUBE, 20, seed=1234)
    pp = [a1, b1]
    t = timed(lambda: (ops.apply_moves(pp[0], pp[1], acts, m, CUBE, rew, done), pp.reverse()), 200)
    rec("cfg2 1M step+reward", "config 2: 3x3x3 batch 1M, apply_move + reward + done", D(_lib.OP_STEP, CUBE, m, outputs=ST | REW), m, "steps", 114, t,
        "working set 113 MB: served from the 256 MB Infinity Cache, not HBM")
    # the metric's batch: in place, with the reward, with the fused compact one-hot code
    a4, b4 = ops.alloc_states(n, CUBE, dev), ops.alloc_states(n, CUBE, dev)
    ops.fill_solved(a4, n, CUBE)
    ops.scramble(a4, n, CUBE, 20, seed=1234)
    p4 = [a4, b4]
    t = timed(lambda: ops.apply_moves(a4, a4, acts, n, CUBE, None, done), 50)
    rec("4M step in place", "3x3x3 batch 4M, apply_move + done IN PLACE (what VecCubeEnv.step launches; 226 MB working set)", D(_lib.OP_STEP, CUBE, n, outputs=ST | INPL),
        n, "steps", 110, t, "one buffer read and rewritten: the working set fits the Infinity Cache")
    t = timed(lambda: (ops.apply_moves(p4[0], p4[1], acts, n, CUBE, rew, done), p4.reverse()), 50)
    rec("4M step+reward", "3x3x3 batch 4M, apply_move + reward + done", D(_lib.OP_STEP, CUBE, n, outputs=ST | REW), n, "steps", 114, t)
    code = ops.alloc_code(n, CUBE, dev)
    t = timed(lambda: (ops.apply_moves(p4[0], p4[1], acts, n, CUBE, rew, done, code, _lib.FMT_CODE), p4.reverse()), 50)
    rec("4M step+reward+code", "3x3x3 batch 4M, apply_move + reward + done + fused compact one-hot code (20 B)", D(_lib.OP_STEP, CUBE, n, outputs=ST | REW | CODE, fmt=_lib.FMT_CODE),
        n, "steps", 134, t)
    del code, a4, b4, p4
    # the HBM-only point: 2^24 cubes, 1.8 GB ping-pong -- nothing of it survives in the 256 MB Infinity Cache between launches
    n16 = 1 << 24
    a16, b16 = ops.alloc_states(n16, CUBE, dev), ops.alloc_states(n16, CUBE, dev)
    ops.fill_solved(a16, n16, CUBE)
    ops.scramble(a16, n16, CUBE, 20, seed=1234)
    acts16 = acts.repeat(n16 // n)
    done16 = torch.empty(n16, dtype=torch.uint8, device=dev)
    p16 = [a16, b16]
    t = timed(lambda: (ops.apply_moves(p16[0], p16[1], acts16, n16, CUBE, None, done16), p16.reverse()), 20, 3)
    hbm_only = rec("16M step (HBM only)", "3x3x3 batch 16M, apply_move + done, nothing cached (1.8 GB ping-pong: the HBM-only point)", D(_lib.OP_STEP, CUBE, n16, outputs=ST),
                   n16, "steps", 110, t, "inputs and outputs streamed (nt / sc0 sc1 nt): every byte comes from and goes to HBM")
    del a16, b16, p16, acts16, done16
    torch.cuda.empty_cache()
    # 2x2x2 (24 stickers, 6 actions): 24 R + 24 W + 1 + 1 = 50 B per step
    a2, b2 = ops.alloc_states(n, 2, dev), ops.alloc_states(n, 2, dev)
    ops.fill_solved(a2, n, 2)
    ops.scramble(a2, n, 2, 20, seed=1234)
    acts2 = acts % 6
    p2 = [a2, b2]
    t = timed(lambda: (ops.apply_moves(p2[0], p2[1], acts2, n, 2, None, done), p2.reverse()), 100)
    rec("2x2x2 4M step", "2x2x2 batch 4M, apply_move + done (201 MB ping-pong: Infinity-Cache resident)", D(_lib.OP_STEP, 2, n, outputs=ST), n, "steps", 50, t)
    del a2, b2, p2, acts2
    # fused dense one-hot in the layout model.py consumes
    for dt, fmt, name, bpc in ((torch.float32, _lib.FMT_F32, "f32", 1920), (torch.bfloat16, _lib.FMT_BF16, "bf16", 960)):
        oh = torch.empty((m, 20, 24), dtype=dt, device=dev)
        t = timed(lambda: ops.apply_moves(a1, b1, acts, m, CUBE, rew, done, oh, fmt), 10, 2)
        rec(f"1M step+dense {name}", f"3x3x3 batch 1M, apply_move + reward + done + fused dense {name} one-hot [N,20,24]",
            D(_lib.OP_STEP, CUBE, m, outputs=ST | REW | _lib.OUT_WORKSPACE, fmt=fmt), m, "steps", 114 + bpc, t,
            "ops.apply_moves = rc_apply_moves_ws: ONE call, two launches (step + reward + done + compact code into a caller-owned workspace, "
            "then the front writer: one 3840-byte pass per workgroup)")
        del oh
    # compact code -> dense one-hot: what the replay sink and the lockstep search launch (the front writer; adi_samples uses its family form)
    code1 = ops.alloc_code(m, CUBE, dev)
    ops.encode(a1, m, CUBE, code1, _lib.FMT_CODE)
    for dt, fmt, name, bpc in ((torch.float32, _lib.FMT_F32, "f32", 1920), (torch.bfloat16, _lib.FMT_BF16, "bf16", 960), (torch.float16, _lib.FMT_F16, "f16", 960),
                               (torch.uint8, _lib.FMT_U8, "u8", 480)):
        oh = torch.empty((m, 20, 24), dtype=dt, device=dev)
        t = timed(lambda: ops.onehot_from_code(code1, m, CUBE, oh), 10, 2)
        rec(f"1M code->dense {name}", f"3x3x3 batch 1M, compact code -> dense {name} one-hot [N,20,24] (rc_onehot_from_code)", D(_lib.OP_CODE_TO_DENSE, CUBE, m, fmt=fmt), m, "cubes",
            20 + bpc, t)
        del oh
    del code1
    # 1M-parent expansion (the MCTS / ADI child loop at scale)
    ex = ops.expand_buffers(m, CUBE, dev, children=True, codes=False)
    t = timed(lambda: ops.expand_children(a1, m, CUBE, ex["children"], ex["child_solved"], pitch=ex["children"].shape[-1]), 30)
    rec("1M expansion", "3x3x3 expansion of 1M parents to all 12 children + solved flags", D(_lib.OP_EXPAND, CUBE, m, outputs=ST | FLAGS), m, "parents", 54 + 12 * 54 + 12, t)
    del ex
    oh8 = torch.empty((m, 20, 24), dtype=torch.uint8, device=dev)
    t = timed(lambda: ops.apply_moves(a1, b1, acts, m, CUBE, rew, done, oh8, _lib.FMT_U8), 10, 2)
    rec("1M step+dense u8", "3x3x3 batch 1M, apply_move + reward + done + fused dense u8 one-hot [N,20,24] (one launch: k_step_dense)",
        D(_lib.OP_STEP, CUBE, m, outputs=ST | REW | _lib.OUT_WORKSPACE, fmt=_lib.FMT_U8), m, "steps", 114 + 480, t)
    del oh8, a1, b1, pp
    # 2x2x2 at 1M cubes: expansion to the 6 children (24 R + 6 * 24 W + 6 flags) and code -> dense f32 (7 R + 147 * 4 W)
    s2 = ops.alloc_states(m, 2, dev)
    ops.fill_solved(s2, m, 2)
    ops.scramble(s2, m, 2, 14, seed=1234)
    ex2 = ops.expand_buffers(m, 2, dev, children=True, codes=False)
    t = timed(lambda: ops.expand_children(s2, m, 2, ex2["children"], ex2["child_solved"], pitch=ex2["children"].shape[-1]), 30)
    rec("2x2x2 1M expansion", "2x2x2 expansion of 1M parents to all 6 children + solved flags", D(_lib.OP_EXPAND, 2, m, outputs=ST | FLAGS), m, "parents", 24 + 6 * 24 + 6, t)
    del ex2
    code2 = ops.alloc_code(m, 2, dev)
    ops.encode(s2, m, 2, code2, _lib.FMT_CODE)
    oh2 = torch.empty((m, 7, 21), dtype=torch.float32, device=dev)
    t = timed(lambda: ops.onehot_from_code(code2, m, 2, oh2), 10, 2)
    rec("2x2x2 1M code->dense f32", "2x2x2 batch 1M, compact code -> dense f32 one-hot [N,7,21] (rc_onehot_from_code)", D(_lib.OP_CODE_TO_DENSE, 2, m, fmt=_lib.FMT_F32), m, "cubes",
        7 + 147 * 4, t)
    del code2, oh2
    # the 2x2x2 ADI plan's one-hot launch (round 6): 12 equally tiled child-code buffers (2 depths x 6 children) of 20000 walks -> packed
    # blocks of ceil16(walks) rows in ONE launch (rc_onehot_from_code_blocks: the tile kernel with blockIdx.y = block)
    wn2, nb2 = 20_000, 12
    pt2, ab2 = ops.adi_buffers(wn2, 2, 2, dev, child_code=True)
    ops.adi_generate(wn2, 2, 2, pt2, dev, seed=2024, **ab2)
    bs2 = -(-wn2 // 16) * 16
    blk2 = torch.zeros((nb2 * bs2, 7, 21), dtype=torch.float32, device=dev)
    cc2 = ab2["child_code"].view(nb2, -1, 7, pt2)
    t = timed(lambda: ops.onehot_from_code_blocks(cc2, wn2, 2, blk2, bs2), 20, 3)
    rec("2x2x2 ADI blocks f32 20000x12", f"2x2x2 ADI child codes -> packed dense f32 blocks: {nb2} code buffers of {wn2} walks in one launch (rc_onehot_from_code_blocks)",
        D(_lib.OP_CODE_TO_DENSE, 2, wn2, fmt=_lib.FMT_F32) + f" x {nb2} blocks (grid.y)", wn2 * nb2, "cubes", 7 + 147 * 4, t)
    del s2, ab2, blk2, cc2
    # config 3: ADI data generation
    W, DEPTH = 100_000, 30
    pt, ab = ops.adi_buffers(W, DEPTH, CUBE, dev, parents=True, children=True)
    t = timed(lambda: ops.adi_generate(W, DEPTH, CUBE, pt, dev, seed=2024, **ab), 10, 3)
    rec("cfg3 ADI 100k x 30", "config 3: ADI 100k walks x depth 30, parents + 12 children + flags + actions", D(_lib.OP_ADI, CUBE, W, DEPTH, outputs=ST | FLAGS), W * DEPTH, "walk-depths", 715, t,
        f"{13 * W * DEPTH / t:.4g} cube-move steps/s; output tiles of {pt} walks; 0 bytes read")
    del ab
    pt, ab = ops.adi_buffers(W, DEPTH, CUBE, dev, **ADI_CODE_OUTPUTS)
    t = timed(lambda: ops.adi_generate(W, DEPTH, CUBE, pt, dev, seed=2024, **ab), 10, 3)
    rec("ADI 100k x 30 codes", "ADI 100k x 30 with compact codes instead of child stickers (parent stickers + 13 codes + flags + actions)", D(_lib.OP_ADI, CUBE, W, DEPTH, outputs=CODE | FLAGS),
        W * DEPTH, "walk-depths", 54 + 1 + 12 + 13 * 20, t, f"output tiles of {pt} walks")
    del ab
    pt, ab = ops.adi_buffers(W, DEPTH, CUBE, dev, family=True)
    t = timed(lambda: ops.adi_generate(W, DEPTH, CUBE, pt, dev, seed=2024, **ab), 10, 3)
    rec("ADI 100k x 30 family", "ADI 100k x 30 as adi_samples launches it: the 51-byte FAMILY record (the shared look-ups behind the 13 codes) + child flags + actions, "
        "no stickers", D(_lib.OP_ADI, CUBE, W, DEPTH, outputs=_lib.OUT_FAMILY | FLAGS), W * DEPTH, "walk-depths", 51 + 12 + 1, t,
        f"output tiles of {pt} walks; 64 B per (walk, depth) against 327 B with the picked codes: the launch is VALU-bound, read the time, not the fraction")
    # the family record -> the [depth][13][walks] dense input of the value net (rc_onehot_from_family_depths): what AdiPlan launches per group of depths
    for wn, gd, dt, name, esz in ((43008, 1, torch.float32, "f32", 4), (20000, 2, torch.float32, "f32", 4), (43008, 2, torch.bfloat16, "bf16", 2)):
        bs = -(-wn // 8) * 8
        blocks = torch.empty((gd * 13 * bs, 20, 24), dtype=dt, device=dev)
        fam = ab["family"][:gd, :-(-wn // pt)].contiguous()
        t = timed(lambda: ops.onehot_from_family(fam, wn, CUBE, blocks, block_stride=bs, n_depths=gd), 10, 3)
        rec(f"ADI dense blocks {name} {wn}x{gd}", f"ADI family record -> dense {name} one-hot blocks [depth][13][walks] of {wn} walks x {gd} depth(s): the value net's input, one launch "
            "(rc_onehot_from_family_depths)", D(_lib.OP_FAMILY_TO_DENSE, CUBE, wn, gd, fmt=_lib.fmt_of(dt)), wn * gd, "walk-depths", 51 + 13 * 480 * esz, t)
        del blocks, fam
    del ab
    torch.cuda.empty_cache()
    out = {"records": recs, "hbm_only_frac": hbm_only["roofline"]["frac"]}
    try:                                                           # config 5 (latency-bound: microseconds, not GB/s) + the lockstep search's whole simulation
        from tools.bench_cfg5 import rounded, run as cfg5
        out["config5_mcts_4096_leaves"] = rounded(cfg5(short=True))
    except Exception as e:
        out["config5_mcts_4096_leaves"] = {"error": str(e)[:200]}
    out.update(pipeline_configs(torch, ops, _lib, dev))
    try:                                                           # batch-1 facade (the reference-shaped CubeEnv.step)
        import numpy as np
        import rubiks_cube_solver_amd as rc
        env = rc.make_env(torch.device("cpu"), CUBE)
        env.reset(seed=1, scramble_count=20)
        seq = np.random.default_rng(0).integers(0, 12, 6000)
        for a_ in seq[:500]:
            env.step(int(a_))
        batches = []                                              # wall-clock on a shared host: five batches, median and best
        for b_ in range(5):
            t0 = time.perf_counter()
            for a_ in seq[500 + 1100 * b_:500 + 1100 * (b_ + 1)]:
                env.step(int(a_))
            batches.append((time.perf_counter() - t0) / 1100)
        batches.sort()
        dt = batches[2]
        out["facade_batch1"] = {"CubeEnv.step_us": dt * 1e6, "CubeEnv.step_us_best_batch": batches[0] * 1e6, "steps_per_s": 1 / dt,
                                "note": "median / best of five 1100-step batches; reference: 24.6 us/step on one CPU core (SURVEY.md section 6); "
                                        "rc_facade_step, results via host-mapped memory"}
        # BASELINE config 1's shape (plumbing): 2x2x2, batch 1, reset(seed, 20) then step() + solved flag, through the same facade
        env2 = rc.make_env(torch.device("cpu"), 2)
        t0 = time.perf_counter()
        for sd in range(200):
            env2.reset(seed=sd, scramble_count=20)
        reset_us = (time.perf_counter() - t0) / 200 * 1e6
        seq2 = np.random.default_rng(1).integers(0, 6, 3000)
        for a_ in seq2[:300]:
            env2.step(int(a_))
        t0 = time.perf_counter()
        for a_ in seq2[300:]:
            _, _, solved2, _ = env2.step(int(a_))
        out["facade_batch1"]["config1_2x2x2"] = {"reset_seed_k20_us": reset_us, "step_us": (time.perf_counter() - t0) / (len(seq2) - 300) * 1e6,
                                                 "note": "2x2x2 values are unpinned (the reference ships no py222): plumbing only"}
    except Exception as e:
        out["facade_batch1"] = {"error": str(e)[:200]}
    return out


def pipeline_configs(torch, ops, _lib, dev):
    """End-to-end figures of the loops the reference actually runs (outside the timed region, a few seconds in all), wall-clock with
    a synchronisation on both sides:
      adi_pipeline   get_random_samples batched (adi.adi_samples: walks + expansion + one-hots + DeepCube forward + targets) at the
                     reference's own size 200 x 30 (config/config.yaml:7-8, train.py:152-155), 20k x 30 and config 3's 100k x 30
      rollout        greedy validation rollouts (train.py:167-198): microseconds per time step at n = 300 and 65536
      reset_seeds    VecCubeEnv.reset(seeds=..., 30): numpy's legacy MT19937 draws on the device + the scramble, 1M envs"""
    out = {}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from tools.bench_adi_pipeline import run as adi_run
        from tools.bench_cfg5 import DeepCubeStandIn
        model = DeepCubeStandIn().to(dev).eval()
        ap = adi_run(reps=3, model=model, dev=dev)
        ap["200x30_hipgraph"] = adi_run(sizes=((200, 30),), reps=5, graph=True, model=model, dev=dev)["200x30"]
        flops_per_state = 2 * (480 * 1024 + 1024 * 256 + 2 * 256 * 128 + 128 * 12 + 128)            # the net's multiply-adds per one-hot state
        for key in ("200x30", "20000x30", "100000x30", "200x30_hipgraph"):                            # 13 states per sample: 12 children + the parent
            tf = 13 * ap[key]["samples_per_s"] * flops_per_state / 1e12
            ap[key]["net_TFLOPs_end_to_end"] = round(tf, 1)
            ap[key]["frac_of_fp32_mfma_peak"] = round(tf / 157.3, 3)
        ap["bound"] = ("the caller's value net: 1.64 MFLOP per state in float32 against the 157.3 TFLOP/s fp32 MFMA peak (MI355X_MICROARCH.md); the env kernels are "
                       "under 5 % of a call (profiles/r05_adi_pipeline.json)")
        # the drop-in call itself (cube_env.py:177-194 as train.py:152-155 issues it): host draws with numpy's global generator, the plan,
        # the replay sink, and the env left on the last walk's final state
        import numpy as np
        import rubiks_cube_solver_amd as rc
        env = rc.make_env(dev, CUBE)
        sink = rc.TensorReplayBuffer(500_000, 100_000, CUBE)
        for graph in (False, True):
            env.adi_graph = graph
            for _ in range(3):
                env.get_random_samples(sink, model, 30, 200, 1.0)
            torch.cuda.synchronize()
            times = []
            for _ in range(7):
                t0 = time.perf_counter()
                env.get_random_samples(sink, model, 30, 200, 1.0)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            ap["CubeEnv.get_random_samples_200x30" + ("_hipgraph" if graph else "")] = {
                "seconds": round(sorted(times)[3], 5), "samples_per_s": round(6000 / sorted(times)[3], 1),
                "what": "env.get_random_samples(TensorReplayBuffer, model, 30, 200, 1.0): 200 np.random.randint draws on the host + upload + AdiPlan.run + append_batch + "
                        "the env's own state update; the reference's call takes 15 s on one CPU core (393 samples/s)"}
        env.close()
        del sink
        ap["2x2x2_20000x14"] = adi_run(sizes=((20_000, 14),), reps=3, dev=dev, cube_size=2)["20000x14"]      # the shipped checkpoint's layer sizes (147-512-128-64)
        ap["note"] = ("median wall time of one adi_samples call incl. its final synchronisation; net = random-init DeepCube [1024,256,128] in float32 (the reference's "
                      "config); the reference does 393 samples/s on one CPU core (SURVEY.md section 6)")
        out["adi_pipeline"] = ap
    except Exception as e:
        out["adi_pipeline"] = {"error": str(e)[:200]}
    try:
        from tools.bench_rollout import run_all
        out["rollout"] = run_all(T=100, model=model)
        out["rollout"]["note"] = "greedy_rollout, 100 time steps from 15-move scrambles, random-init DeepCube: one net forward + one rc_apply_moves per time step"
    except Exception as e:
        out["rollout"] = {"error": str(e)[:200]}
    try:
        import rubiks_cube_solver_amd as rc
        n, k = 1 << 20, 30
        env = rc.VecCubeEnv(n, dev, CUBE, obs=None)
        seeds = torch.arange(n, dtype=torch.int64) * 10                # train.py:180 seeds i * 10
        seeds_dev = seeds.to(dev)
        env.reset(seeds=seeds_dev, scramble_count=k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            env.reset(seeds=seeds_dev, scramble_count=k)
        torch.cuda.synchronize()
        whole = (time.perf_counter() - t0) / 5
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            ops.legacy_scramble_actions(seeds_dev, CUBE, k, device=dev)
        ev[1].record()
        torch.cuda.synchronize()
        out["reset_seeds_1M_k30"] = {"envs": n, "scramble_count": k, "reset_ms": whole * 1e3, "legacy_actions_us": ev[0].elapsed_time(ev[1]) / 5 * 1e3,
                                     "resets_per_s": n / whole,
                                     "note": "VecCubeEnv.reset(seeds=[...], 30): np.random.seed(s); randint(12, size=30) per env regenerated on the device "
                                             "(rc_legacy_scramble_actions: streaming MT19937 + fix-up) then rc_scramble; cube_env.py:62-68"}
        del env
    except Exception as e:
        out["reset_seeds_1M_k30"] = {"error": str(e)[:200]}
    return out


ADI_CODE_OUTPUTS = dict(parents=True, parent_code=True, child_code=True)
SAMPLE_CUBES = 1 << 16      # per-rank sample whose sha256 the N>1 line carries (tests replay it through the oracle)
SCRAMBLE_SEED, SCRAMBLE_DEPTH = 1234, 20


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return str(sock.getsockname()[1])


def per_config_summary(configs):
    """Compact per-config table for the `roofline` object: name, kernel, launch time, algorithmic bytes per launch, fraction of the
    8 TB/s peak -- frac = bytes / (launch_us * 1e-6) / 8e12, recomputable from the row alone."""
    import re
    return [{"name": r["short"], "kernel": re.sub(r" (grid|block|tiles_per_group|cubes_per_pass)=\d+(x\d+)?", "", r["kernel"]), "launch_us": round(r["launch_us"], 2),
             "bytes": r["roofline"]["algorithmic_bytes_per_launch"], "frac": round(r["roofline"]["frac"], 4)}
            for r in configs["records"]]


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # before anything initialises the HIP runtime (RCCL / IPC, N > 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--cubes-per-gpu", type=int, default=N_CUBES,
                    help="batch per GPU (default 2^22 = the metric's batch; 1048576 = BASELINE config 4: 1M cubes per GPU x N GPUs)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs (timed outside the headline region)")
    ap.add_argument("--extras", action="store_true", help="(kept for compatibility: the configs are on by default)")
    ap.add_argument("--backend", default="nccl", help="process-group backend of the distributed branch (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the distributed branch (process group, barrier, all_gather, MAX over ranks) even at world size 1")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: let ranks fold onto the same GPU (local_rank %% device_count) when world > device_count; without it "
                         "such a launch is refused")
    ap.add_argument("--configs", action="store_true", help="N > 1: run the other BASELINE configs on rank 0 too (default there: skipped, rank 0 exits with the others)")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "bench_full.json"),
                    help="where the FULL record (per-config records, notes, per-rank details) is written; stdout carries only the compact line")
    ap.add_argument("--batches", type=int, default=0, help="event-timed batches of --steps launches after the timed region (0 = max(7, 140 / steps))")
    args = ap.parse_args()
    if args.cubes_per_gpu < 1:
        sys.exit("--cubes-per-gpu must be positive")

    import hashlib

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # the distributed branch also runs at world size 1 when asked to (--force-dist) or when a launcher set the rendezvous up
    # (torch.distributed.run --nproc-per-node 1): the same code the N = 2/4/8 runs execute, on the one GPU a builder has
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ and "MASTER_PORT" in os.environ
    use_dist = world > 1 or args.force_dist or launched
    if use_dist and args.backend == "nccl" and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
        # checked before anything touches the GPU: an exported value other than 0 would send RCCL into legacy IPC mode
        sys.exit("HSA_ENABLE_IPC_MODE_LEGACY must be 0 for RCCL on this driver stack (dmabuf IPC); it is "
                 f"{os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')!r}")
    # RC_BENCH_DRY=1: the plumbing of the N-rank line without a GPU (process group, shards, barrier, all_gather, MAX over ranks, the
    # JSON keys) -- no kernel is launched and `value` means nothing; tests/test_bench_contract.py runs the 8-rank shape this way
    dry = os.environ.get("RC_BENCH_DRY") == "1"
    if dry and args.backend != "gloo":
        sys.exit("RC_BENCH_DRY=1 needs --backend gloo (no GPU is touched)")
    # one rank per GPU: rank r asks for device LOCAL_RANK and nothing else.  A launch with more ranks than devices is refused (it
    # would silently fold ranks onto one GPU and report an "8-GPU" line from fewer) unless --share-gpu says it is a rehearsal.
    n_devices = torch.cuda.device_count()
    dev_index = local_rank
    if not dry:
        if n_devices < 1:
            sys.exit("bench.py needs a GPU (RC_BENCH_DRY=1 --backend gloo rehearses the plumbing without one)")
        if local_rank >= n_devices:
            masked = any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
            if args.share_gpu:
                dev_index = local_rank % n_devices
            elif masked and n_devices == 1:
                dev_index = 0      # a launcher that shows every rank ONE card of its own: verified below (distinct PCI bus ids) before a line is printed
            else:
                sys.exit(f"rank {rank}: LOCAL_RANK {local_rank} >= {n_devices} visible device(s); one rank per GPU is the contract "
                         "(--share-gpu folds ranks onto the same GPU for a rehearsal)")
        torch.cuda.set_device(dev_index)
    if world > 1 and not args.configs:
        args.no_configs = True                                        # rank 0 leaves with the others; N = 1 carries the per-config records
    dev = torch.device("cuda", dev_index)
    if use_dist:
        # reporting only (barrier + MAX of elapsed time): the env path itself has no collective
        if not launched:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from rubiks_cube_solver_amd import _lib, ops  # raises if librubikhip.so is missing: no fallback
    from rubiks_cube_solver_amd import dist as rcdist

    n = args.cubes_per_gpu
    stream_id = rcdist.rng_stream(rank)                              # rank-distinct RNG stream, no exchange between ranks
    k = min(n, SAMPLE_CUBES)
    if dry:
        device_name, sample_sha, pci = "dry run (no GPU)", "dry-run", None

        def step():
            pass

        def sync_all():
            if use_dist:
                dist.barrier()
        for _ in range(args.warmup):
            step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync_all()
        elapsed = max(time.perf_counter() - t0, 1e-9)
        dev_ms = elapsed * 1e3
        n_batches = args.batches if args.batches > 0 else max(7, -(-140 // max(1, args.steps)))
        launch_us = launch_min = launch_max = elapsed / args.steps * 1e6
    else:
        device_name = torch.cuda.get_device_name(dev_index)
        props = torch.cuda.get_device_properties(dev_index)
        pci = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}"      # which physical card this rank ran on
        a = ops.alloc_states(n, CUBE, dev)
        b = torch.empty_like(a)
        ops.fill_solved(a, n, CUBE)
        ops.scramble(a, n, CUBE, SCRAMBLE_DEPTH, seed=SCRAMBLE_SEED, stream_id=stream_id)   # 20-move scrambles, rank-distinct streams
        sample_sha = hashlib.sha256(ops.to_aos(a, k).contiguous().cpu().numpy().tobytes()).hexdigest()
        g = torch.Generator(device=dev).manual_seed(1 + rank)
        acts = torch.randint(0, 12, (n,), generator=g, device=dev, dtype=torch.uint8)
        done = torch.empty(n, dtype=torch.uint8, device=dev)
        bufs = [a, b]

        def step():
            ops.apply_moves(bufs[0], bufs[1], acts, n, CUBE, None, done)
            bufs.reverse()

        def sync_all():
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync_all()                                                       # untimed: the first barrier of a process group builds RCCL's communicator (seconds)
        sync_all()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(args.steps):
            step()
        e1.record()
        sync_all()
        elapsed = time.perf_counter() - t0
        dev_ms = e0.elapsed_time(e1)                                     # HIP events on the launch stream
        assert _lib.read_status(dev) == 0
        # The roofline's launch duration: R more batches of `steps` launches each, back to back, every batch bracketed by HIP events on
        # the launch stream (outside the timed region above, same buffers, same kernel): the MEDIAN batch is roofline.launch_us, so a
        # 20-step driver run does not hang the headline fraction on one 1.4 ms sample.
        n_batches = args.batches if args.batches > 0 else max(7, -(-140 // max(1, args.steps)))
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_batches + 1)]
        evs[0].record()
        for bi in range(n_batches):
            for _ in range(args.steps):
                step()
            evs[bi + 1].record()
        torch.cuda.synchronize()
        batch_us = sorted(evs[i].elapsed_time(evs[i + 1]) / args.steps * 1e3 for i in range(n_batches))
        launch_us, launch_min, launch_max = batch_us[len(batch_us) // 2], batch_us[0], batch_us[-1]
        assert _lib.read_status(dev) == 0
    per_rank = None
    if use_dist:
        t = torch.tensor([elapsed, dev_ms, launch_us, launch_min, launch_max], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)                                     # reporting only: per-GPU figures beside the aggregate (config 4)
        shas = [None] * world
        dist.all_gather_object(shas, {"stream_id": stream_id, "sha": sample_sha, "device": device_name, "device_index": dev_index,
                                      "requested_device_index": local_rank, "pci_bus_id": pci, "pid": os.getpid()})
        per_rank = [{"rank": r, "stream_id": shas[r]["stream_id"], "device_index": shas[r]["device_index"],
                     "requested_device_index": shas[r]["requested_device_index"], "pci_bus_id": shas[r]["pci_bus_id"], "device": shas[r]["device"],
                     "ms_per_step": float(x[0]) / args.steps * 1e3,
                     "launch_us": float(x[2]), "launch_us_timed_region": float(x[1]) / args.steps * 1e3, "steps_per_s": n * args.steps / float(x[0]),
                     "GBps": BYTES_PER_STEP * n / (float(x[2]) * 1e-6) / 1e9,
                     "initial_state_sha256_first_cubes": shas[r]["sha"], "sha_cubes": k}
                    for r, x in enumerate(every)]
        cards = {r["pci_bus_id"] for r in per_rank}
        if not dry and not args.share_gpu and len(cards) < world:
            # never an "N-GPU" line from fewer cards: every rank sees the same gathered list and leaves with the same error
            sys.exit(f"rank {rank}: {world} ranks ran on {len(cards)} distinct device(s) {sorted(cards)}: one rank per GPU is the contract "
                     "(--share-gpu marks a rehearsal)")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_ms, launch_us, launch_max = float(t[0]), float(t[1]), float(t[2]), float(t[4])
        launch_min = min(float(x[3]) for x in every)
        # every rank is done with the process group here: nothing below (rank 0's extra configs, the CPU leg) keeps a peer waiting
        # inside a collective
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    # the measured device-copy ceiling SURVEY 8d asks for beside the vendor peak: the runtime's own D2D copy of the same buffers
    copy_gbps = None
    if not dry:
        for _ in range(3):
            bufs[1].copy_(bufs[0])
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(20):
            bufs[1].copy_(bufs[0])
        c1.record()
        torch.cuda.synchronize()
        copy_gbps = 2 * bufs[0].numel() / (c0.elapsed_time(c1) / 20 * 1e-3) / 1e9
    kernel = _lib.describe(_lib.OP_STEP, CUBE, n, outputs=_lib.OUT_STATES | _lib.OUT_DONE)   # what the headline launches, from the library's own dispatch
    configs = None
    if not args.no_configs and not dry:
        del a, b, bufs
        torch.cuda.empty_cache()
        configs = other_configs(torch, ops, _lib, dev, acts)
        assert _lib.read_status(dev) == 0

    steps_per_s = n * args.steps * world / elapsed
    launch_s = launch_us * 1e-6
    achieved = BYTES_PER_STEP * n / launch_s / 1e9
    traffic, traffic_source = None, None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")       # PMC-derived bytes per launch, if profiled
    if os.path.exists(tfile) and n == N_CUBES:
        try:
            tj = json.load(open(tfile))
            traffic = tj.get("k_step_bytes_per_launch")
            traffic_source = (f"profiles/traffic.json: builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of round {tj.get('round')} "
                              "over this same command (not re-measured in this run)")
        except Exception:
            traffic = None
    state_mb = n * 54 / 1e6
    pol = kernel.split("POL=")[1][0] if "POL=" in kernel else "?"
    served = {"0": f"working set {2 * state_mb:.0f} MB fits the 256 MB Infinity Cache: rows default-cached, served on-die, NOT an HBM figure",
              "1": "input rows streamed (nt), output rows written through and kept (sc0 sc1): the next launch finds part of its input in the "
                   "256 MB Infinity Cache, so DRAM traffic is below the fabric traffic the counters show; the nothing-cached figure is "
                   "roofline.frac_hbm_only",
              "2": "input rows streamed (nt), output rows streamed (sc0 sc1 nt): every byte comes from and goes to HBM"}.get(pol, "")
    workload = (f"3x3x3 apply_move + solved flag, {n} cubes per GPU per launch x {world} GPU(s), uint8 SoA [54][N] in 32768-cube tiles, "
                "ping-pong of two buffers; ")
    workload += ("configs[1] shape at the metric's batch 4M" if n == N_CUBES else
                 "BASELINE configs[3] shape: 1M cubes per GPU, rank-distinct RNG streams, no collective" if n == 1 << 20 else "custom batch")
    out = {
        "metric": "cube-move steps/sec, 3x3x3 batch 4M; HBM GB/s vs roofline at 1/2/4/8 GPU",
        "value": steps_per_s, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic" if not dry else "synthetic (RC_BENCH_DRY=1: NO kernel ran, the numbers mean nothing)",
        "config": {"workload": workload, "cubes_per_gpu": n, "total_cubes": n * world, "bytes_per_step_algorithmic": BYTES_PER_STEP,
                   "parallelism": f"{world} independent ranks, stream_id = rank, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": kernel, "launch_us": launch_us, "launch_us_min": launch_min, "launch_us_max": launch_max,
                     "launch_us_timed_region": dev_ms / args.steps * 1e3,
                     "frac_basis": "launch_us: the median event-timed batch AFTER the timed region (the kernel's steady launch duration)",
                     "frac_timed_region": BYTES_PER_STEP * n / (dev_ms / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "frac_wall_clock": BYTES_PER_STEP * n / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS,
                     "frac_note": "three durations, three fractions: frac (launch_us), frac_timed_region (HIP events around the K timed steps themselves), "
                                  "frac_wall_clock (ms_per_step: host clock incl. the barrier and synchronisation on both sides = what `value` uses)",
                     "launch_batches": n_batches, "launches_per_batch": args.steps,
                     "launch_note": "launch_us = median over `launch_batches` back-to-back batches of `launches_per_batch` launches, each bracketed by "
                                    "HIP events on the launch stream, run right after the timed region (N > 1: the slowest rank's median)",
                     "algorithmic_bytes_per_launch": BYTES_PER_STEP * n,
                     "achievable_GBps": HBM_ACHIEVABLE_GBPS, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBPS,
                     "achievable_note": "MI355X_MICROARCH.md: HBM 8 TB/s peak (spec), about 6.3 TB/s achievable",
                     "served_by": "hbm" if pol == "2" else "hbm + infinity cache" if pol == "1" else "infinity cache",
                     "frac_hbm_only": configs["hbm_only_frac"] if configs else None,
                     "frac_hbm_only_note": "the same kernel at 2^24 cubes (1.8 GB ping-pong, nothing cached): per_config '16M step (HBM only)'",
                     "device_copy_GBps": copy_gbps, "frac_of_device_copy": achieved / copy_gbps if copy_gbps else None,
                     "device_copy_note": f"hipMemcpy D2D (torch copy_) of the same {state_mb:.0f} MB state buffer, read + write bytes",
                     "note": served},
    }
    if configs:
        out["roofline"]["per_config"] = per_config_summary(configs)
        out["roofline"]["per_config_note"] = ("every other workload of BASELINE.json / SURVEY 8d, timed outside the headline region (median of 3 event-timed "
                                              "batches): frac = bytes / (launch_us * 1e-6) / 8e12; full records under `configs`")
    if per_rank:
        out["per_gpu"] = per_rank
        out["roofline"]["aggregate_GBps"] = sum(r["GBps"] for r in per_rank)
        out["roofline"]["aggregate_frac_of_n_x_peak"] = out["roofline"]["aggregate_GBps"] / (HBM_PEAK_GBPS * world)
        out["config"]["process_group"] = args.backend
        # which card every rank ran on: distinct_devices < n_gpus means ranks shared a GPU (a --share-gpu rehearsal), never silently
        cards = {(r["pci_bus_id"] or f"requested:{r['requested_device_index']}") for r in per_rank}
        out["config"]["distinct_devices"] = len(cards)
        out["config"]["shared_gpu_rehearsal"] = bool(args.share_gpu and len(cards) < world)
    else:
        out["config"]["distinct_devices"] = 1
        out["per_gpu"] = [{"rank": 0, "stream_id": stream_id, "device_index": dev_index, "requested_device_index": local_rank, "pci_bus_id": pci,
                           "device": device_name}]
    if dry:
        out["dry_run"] = True
    if not args.no_cpu and world == 1 and not dry:
        out["cpu_baseline"] = cpu_baseline()
    if configs:
        out["configs"] = configs
    full_path = None
    if args.full_out:
        try:
            with open(args.full_out, "w") as f:
                json.dump(out, f)
                f.write("\n")
            full_path = os.path.relpath(args.full_out, ROOT) if os.path.abspath(args.full_out).startswith(ROOT + os.sep) else args.full_out
            print(f"bench.py: full record ({os.path.getsize(args.full_out)} bytes) -> {args.full_out}", file=sys.stderr)
        except OSError as e:                                         # a read-only tree: the compact line is still the record
            print(f"bench.py: could not write the full record: {e}", file=sys.stderr)
    line = json.dumps(compact_record(out, full_path), allow_nan=False, separators=(",", ":"))
    assert len(line) < COMPACT_LIMIT, len(line)
    print(line, flush=True)


COMPACT_LIMIT = 4096      # the driver parses the LAST stdout line and keeps 8 KB of tail: round 5's 23.5 KB line came back unparsed


def compact_record(full, full_path=None):
    """The FINAL stdout line: the contract's keys, `roofline`, `cpu_baseline` and ten scalar extras -- no notes, no per-config table,
    every string under 120 characters, the whole line under COMPACT_LIMIT bytes.  Everything else lives in the full record
    (--full-out, default bench_full.json next to this file; the builder's copies are under profiles/)."""
    def cut(v, n=118):
        return v if not isinstance(v, str) or len(v) <= n else v[:n - 1] + "~"

    def pick(d, keys):
        return {k: cut(d[k]) for k in keys if k in d}

    def sig(v, digits=6):
        return float(f"{v:.{digits}g}") if isinstance(v, float) else v

    out = pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    n, world = full["config"]["cubes_per_gpu"], full["n_gpus"]
    out["config"] = {"workload": cut(f"3x3x3 apply_move + solved flag, {n} cubes/GPU/launch x {world} GPU(s), u8 SoA [54][N], ping-pong; "
                                     + ("configs[1] shape at the metric's batch 4M" if n == N_CUBES else "configs[3] shape: 1M cubes per GPU" if n == 1 << 20 else "custom batch"), 200),
                     **pick(full["config"], ("cubes_per_gpu", "total_cubes", "bytes_per_step_algorithmic", "parallelism", "process_group", "distinct_devices",
                                             "shared_gpu_rehearsal"))}
    r = full["roofline"]
    out["roofline"] = pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launch_us", "launch_us_min", "launch_us_max", "launch_us_timed_region",
                               "launch_batches", "launches_per_batch", "frac_timed_region", "frac_wall_clock", "frac_hbm_only", "algorithmic_bytes_per_launch",
                               "served_by", "achievable_GBps", "frac_of_achievable", "device_copy_GBps", "frac_of_device_copy", "aggregate_GBps",
                               "aggregate_frac_of_n_x_peak"))
    out["roofline"]["frac_basis"] = "launch_us = median event-timed batch of K launches after the timed region"
    out["roofline"]["traffic_source"] = "profiles/traffic.json (builder's rocprofv3 --pmc passes)" if r.get("traffic") is not None else None
    if "cpu_baseline" in full:
        c = full["cpu_baseline"]
        out["cpu_baseline"] = pick(c, ("value", "unit", "cores", "kind", "single_core_steps_per_s", "numpy_env_1core_steps_per_s", "numpy_env_allcores_steps_per_s",
                                       "numpy_env_processes"))
        out["cpu_baseline"]["sample"] = cut(c.get("sample_short") or c.get("sample", ""))
    if not (world > 1 or full["config"].get("process_group")):
        out["config"]["pci_bus_id"] = full["per_gpu"][0]["pci_bus_id"] if full.get("per_gpu") else None
    else:
        out["per_gpu"] = [{"rank": p["rank"], "dev": p["device_index"], "pci": p["pci_bus_id"], "stream": p["stream_id"], "launch_us": sig(p["launch_us"], 5),
                           "gsteps_per_s": sig(p["steps_per_s"] / 1e9, 5), "sha12": p["initial_state_sha256_first_cubes"][:12]} for p in full["per_gpu"]]
    cf = full.get("configs")
    if cf:                                                            # at most ten scalars of the other workloads; their records are in the full file
        pc = {x["name"]: x for x in r.get("per_config", [])}
        ap, c5 = cf.get("adi_pipeline", {}), cf.get("config5_mcts_4096_leaves", {})
        ex = {"cfg2_1M_step_reward_frac": pc.get("cfg2 1M step+reward", {}).get("frac"),
              "cfg3_adi_100kx30_frac": pc.get("cfg3 ADI 100k x 30", {}).get("frac"),
              "cfg3_adi_100kx30_us": pc.get("cfg3 ADI 100k x 30", {}).get("launch_us"),
              "dense_f32_1M_step_frac": pc.get("1M step+dense f32", {}).get("frac"),
              "adi_pipeline_200x30_s": ap.get("200x30", {}).get("seconds"),
              "adi_pipeline_100kx30_samples_per_s": ap.get("100000x30", {}).get("samples_per_s"),
              "cfg5_leaves_step_hipgraph_us": c5.get("batched_mcts_leaves_step_with_d2h_hipgraph_us"),
              "cfg5_serial_step_hipgraph_us": c5.get("hipgraph_serial_step_us"),
              "reset_seeds_1M_k30_ms": cf.get("reset_seeds_1M_k30", {}).get("reset_ms"),
              "facade_step_us": cf.get("facade_batch1", {}).get("CubeEnv.step_us")}
        out["extras"] = {k: sig(v, 5) for k, v in ex.items() if v is not None}
    if full.get("dry_run"):
        out["dry_run"] = True
    if full_path:
        out["full_record"] = cut(full_path)
    for key in ("roofline", "cpu_baseline"):
        if key in out:
            out[key] = {k: sig(v, 7) for k, v in out[key].items()}
    return out


if __name__ == "__main__":
    main()
