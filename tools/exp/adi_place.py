"""Does ADI write speed depend on where the children buffer sits?  One pool, many base offsets (development experiment)."""
import os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from rubiks_cube_solver_amd import _lib, ops
W, D, dev = 100_000, 30, torch.device("cuda", 0)
pitch = 4096
tiles = -(-W // pitch)
Wp = tiles * pitch
size_ch = D * 12 * tiles * 54 * pitch
MODE = os.environ.get("MODE", "A")
if MODE == "A":      # small buffers first, pool after (as before)
    parents = torch.empty((D, tiles, 54, pitch), dtype=torch.uint8, device=dev)
    cs = torch.empty((D, 12, Wp), dtype=torch.uint8, device=dev)
    ao = torch.empty((D, Wp), dtype=torch.uint8, device=dev)
    pool = torch.empty(size_ch + (1 << 30), dtype=torch.uint8, device=dev)
elif MODE == "B":    # pool first, small buffers after
    pool = torch.empty(size_ch + (1 << 30), dtype=torch.uint8, device=dev)
    parents = torch.empty((D, tiles, 54, pitch), dtype=torch.uint8, device=dev)
    cs = torch.empty((D, 12, Wp), dtype=torch.uint8, device=dev)
    ao = torch.empty((D, Wp), dtype=torch.uint8, device=dev)
elif MODE == "D":    # small buffers first, then a 5 GB pool; probe offsets up to the very end
    parents = torch.empty((D, tiles, 54, pitch), dtype=torch.uint8, device=dev)
    cs = torch.empty((D, 12, Wp), dtype=torch.uint8, device=dev)
    ao = torch.empty((D, Wp), dtype=torch.uint8, device=dev)
    pool = torch.empty(size_ch + (3 << 30), dtype=torch.uint8, device=dev)
else:                # everything inside ONE pool: [children | gap | parents | cs | ao]
    pool = torch.empty(size_ch + (3 << 30), dtype=torch.uint8, device=dev)
    base = size_ch + (2 << 30)
    parents = pool[base:base + D * tiles * 54 * pitch].view(D, tiles, 54, pitch); base += parents.numel()
    cs = pool[base:base + D * 12 * Wp].view(D, 12, Wp); base += cs.numel()
    ao = pool[base:base + D * Wp].view(D, Wp)
print("pool base", hex(pool.data_ptr()), "parents", hex(parents.data_ptr()))
def run(fn, iters):
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(iters): fn()
    s1.record(); torch.cuda.synchronize()
    return s0.elapsed_time(s1) / iters
offs = [0, 64 << 20, 128 << 20, 256 << 20, 384 << 20, 512 << 20, 768 << 20, 1 << 30]
if MODE == "D":
    offs = [0, 512 << 20, 1 << 30, 1536 << 20, 2 << 30, 2560 << 20, 3 << 30]
print("MODE", MODE)
res = {}
for rep in range(3):
    for off in offs:
        ch = pool[off:off + size_ch].view(D, 12, tiles, 54, pitch)
        fn = lambda: ops.adi_generate(W, D, 3, pitch, dev, seed=2024, actions_out=ao, parents=parents, children=ch, child_solved=cs)
        if rep == 0: run(fn, 3)
        res.setdefault(off, []).append(run(fn, 10))
for off in offs:
    print(f"offset {off:>11d} ({off/2**20:8.3f} MiB): {statistics.median(res[off]):.4f} ms  {715*W*D/statistics.median(res[off])/1e6:.0f} GB/s")
