"""BatchedMCTS.leaves_step (4096 roots, depth-8 descents): host-clock microseconds per call, hipGraph and eager, with the result block
downloaded in one piece behind the net or split (codes + flags on a side stream while the net runs)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from bench_cfg5 import DeepCubeStandIn
from rubiks_cube_solver_amd import ops
from rubiks_cube_solver_amd.mcts_batched import BatchedMCTS

n, cs, dev = 4096, 3, torch.device("cuda")
model = DeepCubeStandIn().to(dev).eval()
leaves = ops.alloc_states(n, cs, dev); ops.fill_solved(leaves, n, cs); ops.scramble(leaves, n, cs, 20, seed=7)
paths = np.random.default_rng(0).integers(0, 12, (n, 8), dtype=np.uint8)
cur = torch.cuda.current_stream()
def wall(fn, it=300):
    for _ in range(30): fn()
    cur.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    cur.synchronize(); return round((time.perf_counter() - t0) / it * 1e6, 1)
out = {}
ref = None
for rep in range(2):
    for graph in (True, False):
        for split in (False,):
            bm = BatchedMCTS(model, leaves, n, cs, graph=graph)
            res = [x.copy() for x in bm.leaves_step(paths, copy=False)]
            if ref is None:
                ref = res
            assert all((a == b).all() for a, b in zip(ref, res)), (graph, split)
            key = f"{'hipgraph' if graph else 'eager'}_{'split' if split else 'one'}_download"
            out.setdefault(key, []).append(wall(lambda: bm.leaves_step(paths, copy=False)))
            if graph:
                out.setdefault(key + "_replay_only", []).append(wall(lambda: (bm._graphs[8].replay(), cur.synchronize())))
print(json.dumps(out))
