import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from rubiks_cube_solver_amd import _lib, ops
W, D, dev = 100_000, 30, torch.device("cuda", 0)
for pitch in (_lib.pitch_for(W), 4096, 1024):
    pt, ab = ops.adi_buffers(W, D, 3, dev, pitch, parents=True, children=True)
    for _ in range(2):
        ops.adi_generate(W, D, 3, pt, dev, seed=2024, **ab)
    torch.cuda.synchronize()
    del ab
n = 1 << 22
for tile in (n, 32768, 1024):
    a = ops.alloc_states(n, 3, dev, tile); b = torch.empty_like(a)
    ops.fill_solved(a, n, 3)
    acts = torch.randint(0, 12, (n,), dtype=torch.uint8, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
    for _ in range(2):
        ops.apply_moves(a, b, acts, n, 3, None, done)
    torch.cuda.synchronize()
