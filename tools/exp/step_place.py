"""Does the step kernel's speed depend on where its two buffers sit inside a big allocation?  (development experiment)"""
import os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from rubiks_cube_solver_amd import _lib, ops
n, dev = 1 << 22, torch.device("cuda", 0)
tiles, pitch = n // 32768, 32768
size = tiles * 54 * pitch
acts = torch.randint(0, 12, (n,), dtype=torch.uint8, device=dev)
done = torch.empty(n, dtype=torch.uint8, device=dev)
POOL = int(os.environ.get("POOL_GB", "5")) << 30
pool = torch.empty(POOL, dtype=torch.uint8, device=dev)
print("pool", hex(pool.data_ptr()), POOL >> 30, "GB; buffer", size >> 20, "MiB")
def run(fn, iters):
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(iters): fn()
    s1.record(); torch.cuda.synchronize()
    return s0.elapsed_time(s1) / iters
def view(off):
    return pool[off:off + size].view(tiles, 54, pitch)
src0 = view(0); ops.fill_solved(src0, n, 3); ops.scramble(src0, n, 3, 20, seed=1)
cands = [0, 256 << 20, 1 << 30, 2 << 30, 3 << 30, 4 << 30, POOL - 2 * size, POOL - 2 * size - (256 << 20)]
res = {}
for rep in range(3):
    for off in cands:
        a, b = view(off), view(off + size)
        if rep == 0:
            a.copy_(src0)
        bufs = [a, b]
        def fn():
            ops.apply_moves(bufs[0], bufs[1], acts, n, 3, None, done); bufs.reverse()
        run(fn, 6)
        res.setdefault(off, []).append(run(fn, 50))
for off in cands:
    t = statistics.median(res[off])
    print(f"offset {off/2**30:7.3f} GiB: {t*1e3:.2f} us  {110*n/t/1e6:.0f} GB/s")
# separately allocated buffers (what bench.py does)
a = ops.alloc_states(n, 3, dev); b = torch.empty_like(a); a.copy_(src0)
bufs = [a, b]
def fn():
    ops.apply_moves(bufs[0], bufs[1], acts, n, 3, None, done); bufs.reverse()
run(fn, 6); t = statistics.median([run(fn, 50) for _ in range(3)])
print(f"separate torch allocations {hex(a.data_ptr())} {hex(b.data_ptr())}: {t*1e3:.2f} us  {110*n/t/1e6:.0f} GB/s")
