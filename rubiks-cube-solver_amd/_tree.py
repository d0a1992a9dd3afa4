"""ctypes binding of librubiktree.so (include/rubiktree.h): the host-side trees of the lockstep search.

Built by __graft_entry__.build() with g++ (no GPU code).  Like _lib.py there is no silent fallback: if the library is
missing, tree() raises."""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RUBIKTREE_LIB") or os.path.join(_HERE, "librubiktree.so")   # env override: sanitizer builds (tools/sanitize_cpu.sh)
_lib = None


def tree_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        L = ctypes.CDLL(LIB_PATH)
        vp, i32, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        from . import _build
        L.rc_tree_build_id.restype = ctypes.c_char_p
        _build.check_loaded(LIB_PATH, L.rc_tree_build_id().decode(), _build.TREE_SOURCES)      # a stale build is refused, not used
        L.rc_tree_create.restype = vp
        L.rc_tree_create.argtypes = [i32, i32, i32, dbl, dbl, dbl]
        L.rc_tree_destroy.argtypes = [vp]
        L.rc_tree_destroy.restype = None
        for name, args in (("rc_tree_set_threads", [vp, i32]), ("rc_tree_set_rng", [vp, i32, vp]), ("rc_tree_get_rng", [vp, vp]), ("rc_tree_select", [vp]),
                           ("rc_tree_paths", [vp, vp, i32]), ("rc_tree_update", [vp, vp, vp, vp, vp, vp]),
                           ("rc_tree_solution", [vp, i32, vp, i32]), ("rc_tree_sims_used", [vp, vp]),
                           ("rc_tree_root_stats", [vp, i32, vp, vp])):
            f = getattr(L, name)
            f.argtypes, f.restype = args, i32
        _lib = L
    return _lib


def cpu_share():
    """Cores this process may use: the smaller of its CPU affinity and its cgroup CPU quota (16 when neither limits it)."""
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(period)))
    except Exception:
        pass
    return max(1, min(affinity, quota) if quota else min(affinity, 16))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class NativeTrees:
    """R trees in librubiktree.so.  rngs: None (the global `random` module, consumed in root order) or one
    random.Random per root; their streams are continued exactly (getstate), and written back by sync_rngs()."""

    def __init__(self, n_roots, n_actions, n_slots, cpuct, virtual_loss, value_min, rngs=None, threads=None):
        import random
        if int(np.__version__.split(".")[0]) < 2:
            # the native PUCT computes c * P * sqrt(N) / (1 + n) + W - L in float32, which is what the reference's expression
            # evaluates to under numpy >= 2 (NEP 50: python float * np.float32 stays float32); numpy 1.x promotes it to float64
            raise RuntimeError("the native search trees reproduce numpy >= 2 scalar arithmetic; with numpy 1.x use "
                               "BatchedMCTS(native=False)")
        self.L = tree_lib()
        self.n, self.A, self.slots = int(n_roots), int(n_actions), int(n_slots)
        self.h = self.L.rc_tree_create(self.n, self.A, self.slots, float(cpuct), float(virtual_loss), float(value_min))
        if not self.h:
            raise ValueError("rc_tree_create: bad arguments")
        self.threads = self.L.rc_tree_set_threads(self.h, int(threads) if threads else cpu_share())
        self._rngs = rngs
        self._shared = rngs is None
        gens = [random] if self._shared else list(rngs)
        if len(gens) != (1 if self._shared else self.n):
            raise ValueError("need one generator per root")
        self._push_rng(gens)

    def _push_rng(self, gens):
        st = np.empty((len(gens), 625), np.uint32)
        for i, g in enumerate(gens):
            version, words, gauss = g.getstate()
            if version != 3:
                raise ValueError("unexpected random.Random state version")
            st[i] = words
        if self.L.rc_tree_set_rng(self.h, int(self._shared), _p(st)) != 0:
            raise ValueError("rc_tree_set_rng failed")

    def sync_rngs(self):
        """Write the generators' current states back into the Python objects they were taken from."""
        import random
        gens = [random] if self._shared else list(self._rngs)
        st = np.empty((len(gens), 625), np.uint32)
        if self.L.rc_tree_get_rng(self.h, _p(st)) != 0:
            raise RuntimeError("rc_tree_get_rng failed")
        for i, g in enumerate(gens):
            g.setstate((3, tuple(int(x) for x in st[i]), g.getstate()[2]))

    def select(self):
        """One descent per unfinished root -> uint8 [n, depth] action paths padded with the no-op."""
        if self._shared:                     # the global generator may have been used by others since the last call
            import random
            self._push_rng([random])
        depth = self.L.rc_tree_select(self.h)
        if depth < 0:
            raise RuntimeError("rc_tree_select failed")
        if self._shared:
            self.sync_rngs()
        paths = np.empty((self.n, max(depth, 1)), np.uint8)
        if self.L.rc_tree_paths(self.h, _p(paths), paths.shape[1]) != 0:
            raise RuntimeError("rc_tree_paths failed")
        return paths[:, :depth]

    def update(self, leaf_code, child_code, solved, value, policy):
        lc = np.ascontiguousarray(leaf_code, np.uint8)
        cc = np.ascontiguousarray(child_code, np.uint8)
        so = np.ascontiguousarray(solved, np.uint8)
        va = np.ascontiguousarray(value, np.float32)
        po = np.ascontiguousarray(policy, np.float32)
        assert lc.shape == (self.n, self.slots) and cc.shape == (self.n, self.A, self.slots) and so.shape == (self.n, self.A)
        assert va.shape == (self.n,) and po.shape == (self.n, self.A)
        done = self.L.rc_tree_update(self.h, _p(lc), _p(cc), _p(so), _p(va), _p(po))
        if done < 0:
            raise RuntimeError("rc_tree_update failed")
        return done

    def solution(self, r):
        buf = np.empty(4096, np.uint8)
        k = self.L.rc_tree_solution(self.h, int(r), _p(buf), len(buf))
        if k > len(buf):                                  # the call reports the full length: fetch again with room for it
            buf = np.empty(k, np.uint8)
            k = self.L.rc_tree_solution(self.h, int(r), _p(buf), len(buf))
        return None if k < 0 else [int(a) for a in buf[:k]]

    def sims_used(self):
        out = np.empty(self.n, np.int32)
        self.L.rc_tree_sims_used(self.h, _p(out))
        return out

    def root_stats(self, r):
        vis, val = np.empty(self.A, np.int32), np.empty(self.A, np.float64)
        nodes = self.L.rc_tree_root_stats(self.h, int(r), _p(vis), _p(val))
        return vis.tolist(), val.tolist(), nodes

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self.L.rc_tree_destroy(h)
