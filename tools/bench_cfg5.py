#!/usr/bin/env python3
"""BASELINE.json config 5: MCTS batched expand, 4096 leaves x 12 children per step, interleaved with
the value-net forward on [4096, 480] fp32 (latency-bound: report microseconds and overlap, not GB/s).

The net is a random-init stand-in with the reference's architecture (model.py:7-45 with
config.yaml hidden_dim [1024, 256, 128]): the product consumes the caller's model unchanged."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch import nn

from rubiks_cube_solver_amd import _lib, ops


class DeepCubeStandIn(nn.Module):
    def __init__(self, state_dim=(20, 24), action_dim=12, hidden=(1024, 256, 128)):
        super().__init__()
        d = state_dim[0] * state_dim[1]
        self.encoder_net = nn.Sequential(nn.Flatten(), nn.Linear(d, hidden[0]), nn.ELU(), nn.Linear(hidden[0], hidden[1]), nn.ELU())
        self.policy_net = nn.Sequential(nn.Linear(hidden[1], hidden[2]), nn.ELU(), nn.Linear(hidden[2], action_dim))
        self.value_net = nn.Sequential(nn.Linear(hidden[1], hidden[2]), nn.ELU(), nn.Linear(hidden[2], 1))

    def forward(self, x):
        if x.dim() == 2:
            x = x.unsqueeze(0)
        h = self.encoder_net(x)
        return self.value_net(h), self.policy_net(h)


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


@torch.no_grad()
def run(short=False, iters=None, sims=20):
    """-> dict of microsecond timings.  short: fewer repetitions (bench.py's default line); the lockstep-search legs always run.
    Legs: the kernels of one MCTS step on 4096 leaves (expansion, dense encode, net forward) one by one, as the serial sequence the
    product launches (eager and as a hipGraph), then the product's lockstep search: device step, device step +
    upload + packed download, and whole simulations split into select / device + transfers / update."""
    import time

    import numpy as np
    iters = iters or (60 if short else 200)
    warm = 10 if short else 20
    T = lambda fn: timeit(fn, iters, warm)
    n, cs, dev = 4096, 3, torch.device("cuda")
    model = DeepCubeStandIn().to(dev).eval()
    leaves = ops.alloc_states(n, cs, dev)
    ops.fill_solved(leaves, n, cs)
    ops.scramble(leaves, n, cs, 20, seed=7)
    ex = ops.expand_buffers(n, cs, dev, children=True, codes=True)
    pitch = ex["children"].shape[-1]
    onehot = torch.empty((n, 20, 24), dtype=torch.float32, device=dev)

    def expand():
        ops.expand_children(leaves, n, cs, ex["children"], ex["child_solved"], ex["child_code"], pitch=pitch)

    def expand_flags_codes():
        ops.expand_children(leaves, n, cs, None, ex["child_solved"], ex["child_code"], pitch=pitch)

    def encode():
        ops.encode(leaves, n, cs, onehot, _lib.FMT_F32)

    def forward():
        return model(onehot)

    def serial():
        expand(); encode(); forward()

    res = {
        "leaves": n, "children_per_step": n * 12,
        "expand_stickers_codes_flags_us": T(expand),
        "expand_codes_flags_us": T(expand_flags_codes),
        "encode_dense_f32_us": T(encode),
        "forward_us": T(forward),
        "serial_step_us": T(serial),
    }
    # BASELINE config 5's "interleaved with the value-net forward" as two streams was measured in rounds 3-5 and lost (serial 164 us, two streams
    # 207 us, hipGraph 123 us: EXPERIMENTS.md); the product runs one stream / one hipGraph and the leg is gone from the bench
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            serial()
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            serial()
        res["hipgraph_serial_step_us"] = T(g.replay)
    except Exception as e:  # capture of foreign launches may be refused by the runtime
        res["hipgraph_serial_step_us"] = None
        res["hipgraph_error"] = str(e)[:200]
    # the product's lockstep search: device part of one simulation (replay 8 moves + expand + one-hot + forward + result packing)
    from rubiks_cube_solver_amd.mcts_batched import BatchedMCTS
    paths = np.random.default_rng(0).integers(0, 12, (n, 8), dtype=np.uint8)
    for name, kw in (("eager", {}), ("hipgraph", {"graph": True})):
        bm = BatchedMCTS(model, leaves, n, cs, **kw)
        bm.leaves_step(paths)
        bm.leaves_step(paths)
        torch.cuda.synchronize()
        res[f"batched_mcts_device_step_{name}_us"] = T(bm._graphs[8].replay) if kw else T(lambda: bm._device_step(8))
        calls = []
        for _ in range(max(sims, 40)):
            t0 = time.perf_counter()
            bm.leaves_step(paths, copy=False)
            calls.append((time.perf_counter() - t0) * 1e6)
        res[f"batched_mcts_leaves_step_with_d2h_{name}_us"] = sorted(calls)[len(calls) // 2]        # median call: one host hiccup must not move it
        res[f"batched_mcts_leaves_step_with_d2h_{name}_max_us"] = max(calls)
    res["batched_mcts_transfers_us"] = res["batched_mcts_leaves_step_with_d2h_hipgraph_us"] - res["batched_mcts_device_step_hipgraph_us"]
    # whole simulations of the lockstep search (host trees in librubiktree.so + the device step + transfers), split per phase
    import random
    bm = BatchedMCTS(model, leaves, n, cs, graph=True, rngs=[random.Random(r) for r in range(n)])
    for _ in range(30):                                              # past the first graph captures (depth buckets 4, 8, 16)
        bm.simulate()
    torch.cuda.synchronize()
    buckets = sorted(bm._graphs)
    for copy in (False, True):                                       # the trees read the pinned download block in place / private copies of it
        split = {"select": [], "device_and_transfers": [], "update": []}
        t_all = time.perf_counter()
        for _ in range(sims):
            t0 = time.perf_counter()
            p = bm.native.select()
            t1 = time.perf_counter()
            out = bm.leaves_step(p, copy=copy)
            t2 = time.perf_counter()
            bm.native.update(*out)
            t3 = time.perf_counter()
            split["select"].append(t1 - t0)
            split["device_and_transfers"].append(t2 - t1)
            split["update"].append(t3 - t2)
        total = time.perf_counter() - t_all
        tag = "_copied_results" if copy else ""
        res[f"batched_mcts_simulate_native_tree{tag}_ms"] = total / sims * 1e3
        res[f"batched_mcts_simulate_split{tag}_us"] = {k: sorted(v)[len(v) // 2] * 1e6 for k, v in split.items()}      # medians
    res["batched_mcts_graph_buckets"] = {"before": buckets, "after": sorted(bm._graphs)}
    res["batched_mcts_note"] = (f"simulations 31-{30 + 2 * sims} of 4096 roots, per-root generators, hipGraph device step, one packed download; select / update = "
                                "librubiktree.so (C++ / OpenMP on this job's CPU share); the pure-Python tree needs 100-500 ms per simulation")
    return res


def rounded(res):
    def r(v):
        return round(v, 2) if isinstance(v, float) else {a: r(b) for a, b in v.items()} if isinstance(v, dict) else [r(x) for x in v] if isinstance(v, list) else v
    return {k: r(v) for k, v in res.items()}


def main():
    print(json.dumps(rounded(run())))


if __name__ == "__main__":
    main()
