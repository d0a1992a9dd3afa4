"""librubiktree.so (the C++ trees of the lockstep search, include/rubiktree.h) against the pure-Python tree of
mcts_batched.BatchedMCTS, on the CPU: both are driven by the same synthetic "device step" (leaf keys with transpositions,
values, policies and rare solved children are a deterministic function of the path), with per-root generators and with
the shared global generator.  Every root must end identically: simulations used, solution, root visit counts and the
exact root values (mcts.py:52-154 semantics incl. float32 PUCT arithmetic and random.randint draws)."""
import random

import numpy as np
import pytest

A, SL, N = 12, 20, 40


def leaf_data(paths):
    lc = np.zeros((N, SL), np.uint8)
    cc = np.zeros((N, A, SL), np.uint8)
    so = np.zeros((N, A), np.uint8)
    va = np.zeros(N, np.float32)
    po = np.zeros((N, A), np.float32)

    def norm(p):                      # a move followed by its inverse returns to the same state: transpositions
        out = []
        for a in p:
            if out and out[-1] == (a ^ 1):
                out.pop()
            else:
                out.append(a)
        return out[-SL:]

    def key(p):
        k = np.zeros(SL, np.uint8)
        k[:len(p)] = np.array(p, np.uint8) + 1
        return k
    for r in range(N):
        p = norm([int(a) for a in paths[r] if a != A])
        lc[r] = key(p)
        for a in range(A):
            cc[r, a] = key(norm(p + [a]))
        g = np.random.default_rng([r] + p)
        va[r] = np.float32(g.normal())
        x = g.normal(size=A).astype(np.float32)
        po[r] = np.exp(x) / np.exp(x).sum()
        so[r] = g.random(A) < 0.001
    return lc, cc, so, va, po


def python_tree(rngs):
    from rubiks_cube_solver_amd import _lib, mcts_batched as mb

    class Host(mb.BatchedMCTS):       # the Python tree with the synthetic device step
        def __init__(self):
            self.n, self.A, self.c, self.vl, self.vmin = N, A, 1.0, 150.0, -10.0
            self.native, self.rngs, self.dev = None, rngs, None
            self.trees = [dict() for _ in range(N)]
            self.solution = [None] * N
            self.sims_used = [0] * N

        def leaves_step(self, pad):
            return leaf_data(pad)
    return Host(), _lib


@pytest.mark.parametrize("shared", [False, True])
def test_native_trees_equal_python_trees(monkeypatch, shared):
    from rubiks_cube_solver_amd._tree import NativeTrees
    py, _lib = python_tree(None if shared else [random.Random(1000 + r) for r in range(N)])
    monkeypatch.setattr(_lib, "read_status", lambda dev=None: 0)
    nat = NativeTrees(N, A, SL, 1.0, 150.0, -10.0, None if shared else [random.Random(1000 + r) for r in range(N)])
    saved = random.getstate()
    try:
        for sim in range(150):
            if shared:
                random.seed(777 + sim)            # both sides consume the SAME global stream from the same point
            py.simulate()
            if shared:
                random.seed(777 + sim)
            paths = nat.select()
            pad = np.full((N, paths.shape[1]), A, np.uint8)
            pad[:, :paths.shape[1]] = paths
            nat.update(*leaf_data(pad))
    finally:
        random.setstate(saved)
    deep = 0
    for r in range(N):
        visits, values, nodes = nat.root_stats(r)
        root = py.trees[r][b"root"]
        assert nat.solution(r) == py.solution[r], r
        assert int(nat.sims_used()[r]) == py.sims_used[r], r
        assert visits == root.visits, r
        assert [float(v) for v in values] == [float(v) for v in root.value], r      # exact, not approximate
        assert nodes == len({id(v) for v in py.trees[r].values()}), r
        deep += nodes > 100
    assert deep >= 5                                     # some searches really ran long (transpositions, deep PUCT)


def test_native_rng_is_cpython_randint():
    """The generator continues a random.Random stream exactly: after a search, sync_rngs() leaves every Python generator
    where a pure-Python run would have left it."""
    from rubiks_cube_solver_amd._tree import NativeTrees
    gens = [random.Random(5 + r) for r in range(N)]
    ref = [random.Random(5 + r) for r in range(N)]
    py, _ = python_tree(ref)
    import rubiks_cube_solver_amd._lib as L
    orig = L.read_status
    L.read_status = lambda dev=None: 0
    try:
        nat = NativeTrees(N, A, SL, 1.0, 150.0, -10.0, gens)
        for _ in range(40):
            py.simulate()
            paths = nat.select()
            nat.update(*leaf_data(paths if paths.shape[1] else np.full((N, 0), A, np.uint8)))
    finally:
        L.read_status = orig
    nat.sync_rngs()
    for g, r in zip(gens, ref):
        assert g.getstate() == r.getstate()


def test_tree_driver_thread_count_invariance(tmp_path):
    """tests/abi/tree_driver.cpp: a plain C++ consumer of include/rubiktree.h (no Python in the process) -- the program
    tools/sanitize_cpu.sh runs under ASan / UBSan / TSan -- built here with g++ and run plain: 1 thread and 4 threads must leave every
    root identical (simulations, visit counts, values, solutions, generator states) with per-root generators and with the shared
    generator consumed in root order (mcts.py:52-154 semantics; random.randint draws at mcts.py:69-70)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "tree_driver")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-Wall", "-Wextra", "-Werror", f"-I{root}/include",
                           os.path.join(root, "tests", "abi", "tree_driver.cpp"), os.path.join(root, "rubiks-cube-solver_amd", "csrc", "rc_tree.cpp"), "-o", exe])
    for args in (["96", "40", "4"], ["33", "25", "3"]):
        out = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="4"))
        assert out.returncode == 0 and out.stdout.startswith("tree_driver ok:") and "identical" in out.stdout, out.stdout + out.stderr
