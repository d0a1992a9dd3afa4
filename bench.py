#!/usr/bin/env python3
"""Headline benchmark: cube-move steps/s, 3x3x3, batch 4M per GPU (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu] [--extras]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One bench "step" = one pass of the hot path over one batch: ONE rc_apply_moves launch that moves
2^22 cubes (each by its own random face turn) and writes their solved flags, ping-ponging two
HBM-resident state buffers (working set 453 MB > the 256 MB Infinity Cache, so HBM-bound).
Inputs are resident in HBM before the timed region.  value = cubes * K * N / max-over-ranks time.
Every rank owns its own batch and RNG stream (stream_id = rank); there is no collective on the
env path ("scaling": "weak").  One JSON line is printed by rank 0.
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
    procs = procs or min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    # the workers are plain CPU processes: do not let a profiler's preload (rocprofv3 sets LD_PRELOAD) make each of
    # them open the GPU
    saved = {k: os.environ.pop(k) for k in list(os.environ) if k == "LD_PRELOAD" or k.startswith(("ROCP", "ROCPROF"))}
    try:
        with mp.get_context("spawn").Pool(procs) as pool:
            rates = pool.map(_numpy_env_worker, [seconds] * procs)
    finally:
        os.environ.update(saved)
    return float(sum(rates)), procs


def cpu_baseline(budget_s=8.0):
    """The C oracle (kind "port") timed on this box's host cores, same workload shape, bounded sample."""
    import numpy as np
    from oracle.oracle_np import Oracle, OracleCubeEnv

    orc = Oracle()
    n = 1 << 20
    rng = np.random.default_rng(1)
    walk = rng.integers(0, 12, (n, 1), dtype=np.uint8)
    states = orc.adi(CUBE, n, 1, actions_in=walk, want_children=False, threads=orc.max_threads())["parents"][:, 0]
    acts = rng.integers(0, 12, n, dtype=np.uint8)
    res = {}
    for label, threads in (("single", 1), ("all", orc.max_threads())):
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
        "sample": f"C oracle (oracle/rc_oracle.c, OpenMP) step = move+solved flag on 2^20 cubes x {res['all']['iters']} passes "
                  f"({res['all']['seconds']:.1f} s); host has {os.cpu_count()} logical cores",
        "single_core_steps_per_s": res["single"]["steps_per_s"],
        "numpy_env_1core_steps_per_s": np_rate,
        "numpy_env_allcores_steps_per_s": np_all, "numpy_env_processes": np_procs,
        "numpy_env_note": "reference-style per-cube numpy env (oracle_np.OracleCubeEnv.step: move + one-hot + solved), 2 s sample",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--extras", action="store_true", help="also time fused reward / code / ADI variants (outside the timed region)")
    ap.add_argument("--backend", default="nccl", help="process-group backend for N>1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    dev_index = local_rank % max(1, torch.cuda.device_count())      # one rank per GPU (rehearsals may share one)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # reporting only (barrier + MAX of elapsed time): the env path itself has no collective
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from rubiks_cube_solver_amd import _lib, ops  # raises if librubikhip.so is missing: no fallback

    n = N_CUBES
    a = ops.alloc_states(n, CUBE, dev)
    b = torch.empty_like(a)
    ops.fill_solved(a, n, CUBE)
    ops.scramble(a, n, CUBE, 20, seed=1234, stream_id=rank)         # 20-move scrambles, rank-distinct streams
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    acts = torch.randint(0, 12, (n,), generator=g, device=dev, dtype=torch.uint8)
    done = torch.empty(n, dtype=torch.uint8, device=dev)
    bufs = [a, b]

    def step():
        ops.apply_moves(bufs[0], bufs[1], acts, n, CUBE, None, done)
        bufs.reverse()

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
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
    if world > 1:
        t = torch.tensor([elapsed, dev_ms], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_ms = float(t[0]), float(t[1])

    extras = {}
    if args.extras and rank == 0:
        def timed(fn, iters=50):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            for _ in range(iters):
                fn()
            s1.record()
            torch.cuda.synchronize()
            return s0.elapsed_time(s1) / iters * 1e-3
        rew = torch.empty(n, dtype=torch.float32, device=dev)
        code = ops.alloc_code(n, CUBE, dev)
        t = timed(lambda: (ops.apply_moves(bufs[0], bufs[1], acts, n, CUBE, rew, done), bufs.reverse()))
        extras["step_reward_done"] = {"steps_per_s": n / t, "GBps": 114 * n / t / 1e9}
        t = timed(lambda: (ops.apply_moves(bufs[0], bufs[1], acts, n, CUBE, rew, done, code, _lib.FMT_CODE), bufs.reverse()))
        extras["step_reward_done_code"] = {"steps_per_s": n / t, "GBps": 134 * n / t / 1e9}
        m = 1 << 20                                                  # BASELINE config 2: batch 1M, apply_move + reward
        a1, b1 = ops.alloc_states(m, CUBE, dev), ops.alloc_states(m, CUBE, dev)
        ops.fill_solved(a1, m, CUBE)
        ops.scramble(a1, m, CUBE, 20, seed=1234)
        pp = [a1, b1]
        t = timed(lambda: (ops.apply_moves(pp[0], pp[1], acts, m, CUBE, rew, done), pp.reverse()), iters=200)
        extras["cfg2_step_reward_done_1M"] = {"steps_per_s": m / t, "GBps": 114 * m / t / 1e9, "launch_us": t * 1e6,
                                              "note": "226 MB ping-pong working set fits the 256 MB Infinity Cache"}
        oh = torch.empty((m, 20, 24), dtype=torch.float32, device=dev)
        t = timed(lambda: ops.apply_moves(a, b, acts, m, CUBE, rew, done, oh, _lib.FMT_F32), iters=10)
        extras["step_dense_f32_1M"] = {"steps_per_s": m / t, "GBps": (114 + 1920) * m / t / 1e9}
        del oh
        W, D = 100_000, 30
        pt, ab = ops.adi_buffers(W, D, CUBE, dev, parents=True, children=True)      # 4096-walk tiles per (depth, child)
        t = timed(lambda: ops.adi_generate(W, D, CUBE, pt, dev, seed=2024, **ab), iters=5)
        extras["adi_100k_x30"] = {"units_per_s": W * D / t, "steps_per_s": 13 * W * D / t, "GBps": 715 * W * D / t / 1e9}
        del ab

    if rank == 0:
        steps_per_s = n * args.steps * world / elapsed
        launch_s = dev_ms * 1e-3 / args.steps
        achieved = BYTES_PER_STEP * n / launch_s / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")       # PMC-derived bytes per launch, if profiled
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("k_step_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "cube-move steps/sec, 3x3x3 batch 4M; HBM GB/s vs roofline at 1/2/4/8 GPU",
            "value": steps_per_s, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "3x3x3 apply_move + solved flag, 2^22 cubes per GPU per launch, uint8 SoA [54][N], "
                                   "ping-pong of two buffers (configs[1] shape at the metric's batch 4M)",
                       "cubes_per_gpu": n, "bytes_per_step_algorithmic": BYTES_PER_STEP,
                       "parallelism": f"{world} independent ranks, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "k_step<Cube3,V,move,store>", "launch_us": launch_s * 1e6,
                         "algorithmic_bytes_per_launch": BYTES_PER_STEP * n},
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        if extras:
            out["extras"] = extras
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
