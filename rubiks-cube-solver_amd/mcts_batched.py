"""MCTS on the batched env (config 5 of BASELINE.json).

The reference's tree search (mcts.py:12-147) spends its env time in `expand` (mcts.py:83-113):
13 `copy.deepcopy(env)` and 12 `env.step` per leaf, with children keyed by
`np.array2string(one-hot)`.  Here a leaf's 12 children, their solved flags and their compact
one-hot codes come from ONE rc_expand_children launch, and a node's key is its 20-byte (7-byte)
code -- lossless, 24x smaller than the printed one-hot.

  MCTS         same class surface as the reference (train / traverse / expand / backpropagate /
               get_most_promising_action_index) over one CubeEnv; statistics follow mcts.py.
  BatchedMCTS  R independent roots searched in lockstep: per simulation ONE replay launch brings all
               R leaves into a device buffer (paths padded with the no-op), ONE expansion launch
               produces R x A children (optionally on a side stream) and the value/policy net runs
               once on the R leaves; tree statistics stay on the host, by default in librubiktree.so
               (include/rubiktree.h: C++ / OpenMP, the reference's arithmetic and random draws bit for bit;
               25 ms -> well under 1 ms of host time per simulation of 4096 roots), or in the pure-Python
               tree below (`native=False`, kept as the cross-check).

Tree rules restated from mcts.py: PUCT score U + W - L with U = c * P * sqrt(sum N) / (1 + N)
(:148-169); W is the MAX of backed-up values (:124-125); a traversed edge gains the virtual loss and
loses it again on back-propagation (:77,127); an untried node picks a uniformly random action while
its visit counts are all zero (:69-70).
"""
from __future__ import annotations

import copy
import math
import random

import numpy as np
import torch

from . import _lib, ops
from .tables import get_env_config


class _Node:
    __slots__ = ("children", "policy", "value", "visits", "vloss", "done")

    def __init__(self, children, policy, value_min, done):
        a = len(children)
        self.children, self.policy, self.done = children, policy, done
        self.value = [value_min] * a
        self.visits = [0] * a
        self.vloss = [0] * a


def _puct_best(node, c):
    """mcts.py:132-154 with the reference's arithmetic types: `policy[i]` stays the numpy float32 scalar it is in the
    reference, so `c * P * x + W - L` is evaluated exactly as there (float32 under numpy >= 2) and near ties fall the
    same way; `max(range(A), key=...)` = first maximal index."""
    total = sum(node.visits)
    root = math.sqrt(total)
    score = [c * node.policy[i] * (root / (1 + node.visits[i])) + node.value[i] - node.vloss[i] for i in range(len(node.children))]
    return max(range(len(score)), key=score.__getitem__)


class MCTS:
    """Drop-in for the reference's `MCTS(model, cfg)`; `train(state, env)` runs one simulation and
    returns the action list to a solved child if one was found, else None (mcts.py:36-50)."""

    def __init__(self, model, cfg):
        self.model = model
        self.children_and_data = dict()  # key (bytes of the compact code) -> _Node
        self.loss_constant = cfg["mcts"]["virtual_loss_const"]
        self.exploration_constant = cfg["mcts"]["cpuct"]
        self.value_min = cfg["mcts"]["value_min"]
        self.cube_size = cfg["test"]["cube_size"]
        _, self.action_dim = get_env_config(self.cube_size)

    @staticmethod
    def key_of(env):
        """Node key of the env's current state: its compact one-hot code (one launch, no download)."""
        return env.expand_host()[0]

    def key_of_state(self, state):
        """Node key of a one-hot state array, on the host: the reference keys nodes by np.array2string(state)
        (mcts.py:66); the compact code is the same information (one column index per row)."""
        s = np.asarray(state)
        if self.cube_size == 3:
            return s.argmax(-1).astype(np.uint8).tobytes()          # row = slot, column = code
        col = s.argmax(-1)                                           # row = piece: column = slot*3 + ori (cube_env.py:143-147)
        code = np.zeros(len(col), np.uint8)
        code[col // 3] = np.arange(len(col)) * 3 + col % 3
        return code.tobytes()

    def train(self, state, env):
        sim = copy.deepcopy(env)  # mcts.py:37 (one clone per simulation instead of 14)
        path, actions, leaf_key = self.traverse(state, sim)
        value = self.expand(leaf_key, sim)
        self.backpropagate(path, actions, value)
        node = self.children_and_data[leaf_key]
        for i, d in enumerate(node.done):
            if d:
                actions.append(i)
                return actions
        return None

    def traverse(self, state, env):
        """mcts.py:52-81.  The reference steps the env once per tree level; the descent only needs the tree (child keys are
        stored in the nodes), so the actions are collected on the host and the env makes the whole descent in ONE launch
        (CubeEnv.step_many) -- it ends on the same leaf state."""
        path, actions = [], []
        current = self.key_of_state(state)                          # mcts.py:66: the root is keyed by the `state` argument
        while True:
            node = self.children_and_data.get(current)
            if node is None or not node.children:
                break
            a = random.randint(0, self.action_dim - 1) if sum(node.visits) == 0 else _puct_best(node, self.exploration_constant)
            path.append(current)
            actions.append(a)
            node.vloss[a] += self.loss_constant
            current = node.children[a]
        if actions:
            if hasattr(env, "step_many"):
                env.step_many(actions)
            else:
                for a in actions:
                    env.step(a)
        return path, actions, current

    def expand(self, leaf_key, env):
        """mcts.py:83-113 with ONE launch (rc_facade_expand: child codes + solved flags land in pinned host memory)
        instead of 12 steps + 13 deep copies."""
        value, policy = self.model.predict(env.cube)
        _, codes, solved = env.expand_host()
        self.children_and_data[leaf_key] = _Node([c.tobytes() for c in codes], policy, self.value_min, [bool(x) for x in solved])
        return value

    def backpropagate(self, path, actions, reward):
        r = np.asarray(reward).reshape(-1)[0]                       # stays the model's float32 (mcts.py:124-125)
        for key, a in zip(reversed(path), reversed(actions)):
            node = self.children_and_data[key]
            node.value[a] = max(node.value[a], r)
            node.vloss[a] -= 150  # mcts.py:127: the literal 150, NOT virtual_loss_const (bug-compatible: a config with another
            #                       constant keeps a residual virtual loss in the reference, and so here)
            node.visits[a] += 1

    def get_most_promising_action_index(self, key):
        return _puct_best(self.children_and_data[key], self.exploration_constant)


class _RootView:
    """trees[r][b"root"] of a native search: the root node's visit counts and values."""

    def __init__(self, nat, r):
        self._nat, self._r = nat, r

    def __getitem__(self, key):
        if key != b"root":
            raise KeyError(key)
        visits, value, _ = self._nat.root_stats(self._r)

        class _N:
            pass
        node = _N()
        node.visits, node.value = visits, value
        return node


class BatchedMCTS:
    """R roots searched in lockstep on one GPU (config 5: 4096 roots x 12 children per step)."""

    def __init__(self, model, root_stickers, n_roots, cube_size=3, cpuct=1.0, virtual_loss=150.0, value_min=-10.0,
                 device="cuda", rngs=None, graph=False, native=True):
        """rngs: optional list of `random.Random` (one per root) for the untried-node draws (mcts.py:69-70); with
        root r's generator seeded like a stand-alone run, root r's search is that run (default: the global `random`).
        With native=True the generators' states are COPIED into the C++ trees and advanced there: call sync_rngs() before
        using the Python objects again (native=False consumes them in place).
        graph: capture the per-simulation kernel sequence as a hipGraph and replay it (launch-bound loop).
        Everything runs on ONE stream: expansion on a side stream next to the net forward was measured and loses at this
        size (8-9 us of expansion against the cost of two cross-stream dependencies; tools/bench_cfg5.py still reports the
        interleaved shape for BASELINE config 5, DESIGN.md "Config 5")."""
        self.model, self.cube_size, self.n = model, cube_size, int(n_roots)
        self.c, self.vl, self.vmin = cpuct, virtual_loss, value_min
        self.dev = torch.device(device)
        (self.R, self.C), self.A = get_env_config(cube_size)
        self.roots = root_stickers                      # tiled state buffer [tiles, S, pitch]
        self.work = torch.empty_like(root_stickers)
        self.rngs = rngs
        self.native = None
        if native:
            from ._tree import NativeTrees
            self.native = NativeTrees(self.n, self.A, ops.N_SLOTS[cube_size], cpuct, virtual_loss, value_min, rngs)
        else:
            self.trees = [dict() for _ in range(self.n)]
            self.solution = [None] * self.n
            self.sims_used = [0] * self.n
        self.onehot = torch.empty((self.n, self.R, self.C), dtype=torch.float32, device=self.dev)
        self.code = ops.alloc_code(self.n, cube_size, self.dev, root_stickers.shape[-1])
        self.ex = ops.expand_buffers(self.n, cube_size, self.dev, root_stickers.shape[-1], children=False, codes=True)
        SL = ops.N_SLOTS[cube_size]
        # Results travel to the host as ONE block: leaf codes, child codes, solved flags, leaf values and policies are produced in the
        # host's layout ([root][...]) ON the device (small transposing copies inside the captured graph) as views of one device
        # buffer, which one D2H copy brings into one pinned buffer; the trees read the pinned views in place (round 4 issued five
        # copies and then copied every array again on the host: 172 us per simulation of 4096 roots).
        n, A = self.n, self.A
        parts = (("leaf", n * SL), ("child", n * A * SL), ("solved", n * A), ("value", 4 * n), ("policy", 4 * n * A))
        off, total = {}, 0
        for name, size in parts:
            off[name] = (total, size)
            total = -(-(total + size) // 256) * 256
        self._pack_dev = torch.zeros(total, dtype=torch.uint8, device=self.dev)
        self._pack_host = torch.zeros(total, dtype=torch.uint8).pin_memory() if self.dev.type == "cuda" else torch.zeros(total, dtype=torch.uint8)
        dv = lambda name: self._pack_dev[off[name][0]:off[name][0] + off[name][1]]
        self._leaf_aos = dv("leaf").view(n, SL)
        self._child_aos = dv("child").view(n, A, SL)
        self._solved_aos = dv("solved").view(n, A)
        self._value_dev = dv("value").view(torch.float32)
        self._policy_dev = dv("policy").view(torch.float32).view(n, A)
        hp = self._pack_host.numpy()
        hv = lambda name: hp[off[name][0]:off[name][0] + off[name][1]]
        self._host = (hv("leaf").reshape(n, SL), hv("child").reshape(n, A, SL), hv("solved").reshape(n, A),
                      hv("value").view(np.float32), hv("policy").view(np.float32).reshape(n, A))
        self.graph, self._graphs, self._paths = bool(graph), {}, None

    def __getattr__(self, name):
        # result views of the native trees under the names the Python tree uses (solution, sims_used, trees[r][b"root"]);
        # fetched once per simulation, not once per access
        nat = self.__dict__.get("native")
        if nat is not None and name in ("solution", "sims_used", "trees"):
            cache = self.__dict__.setdefault("_views", {})
            stamp = self.__dict__.get("_sims", 0)
            if cache.get("stamp") != stamp:
                cache.clear()
                cache["stamp"] = stamp
            if name not in cache:
                if name == "solution":
                    cache[name] = [nat.solution(r) for r in range(self.n)]
                elif name == "sims_used":
                    cache[name] = nat.sims_used().tolist()
                else:
                    cache[name] = [_RootView(nat, r) for r in range(self.n)]
            return cache[name]
        raise AttributeError(name)

    def _device_step(self, depth):
        """Kernels and the download of one simulation, all on the current stream (capturable): roots replayed `depth` moves into work
        (rc_scramble_from reads self._paths from pinned host memory), expansion, leaf code, ONE launch that lays codes and flags out
        per root (rc_search_pack), one-hot, net forward, value + policy into the same block, one D2H copy of the block.
        (Sending codes + flags on a side stream while the net runs was measured and loses: 246 us against 211 us per call under
        hipGraph -- as every two-branch graph on this stack; profiles/r05_mcts.json.)"""
        n, cs = self.n, self.cube_size
        ops.scramble(self.work, n, cs, depth, actions_in=self._paths[:depth] if depth else None, src=self.roots)
        pitch = self.work.shape[-1]
        ops.expand_children(self.work, n, cs, None, self.ex["child_solved"], self.ex["child_code"], pitch=pitch)
        ops.encode(self.work, n, cs, self.code, _lib.FMT_CODE)
        ops.search_pack(self.code, self.ex["child_code"], self.ex["child_solved"], n, cs, self._leaf_aos, self._child_aos, self._solved_aos)
        ops.onehot_from_code(self.code, n, cs, self.onehot)
        value, logits = self.model(self.onehot)
        if logits.dtype == torch.float32:
            torch.softmax(logits, dim=-1, out=self._policy_dev)                 # model.py:89, straight into the download block
        else:
            self._policy_dev.copy_(torch.softmax(logits, dim=-1))
        self._value_dev.copy_(value.reshape(-1))
        self._pack_host.copy_(self._pack_dev, non_blocking=True)

    @torch.no_grad()
    def leaves_step(self, paths, copy=True):
        """Device part of one simulation for `paths` (uint8 [R, depth], no-op padded).  With graph=True the
        kernel sequence is captured once per depth bucket as a hipGraph (paths padded with the no-op up to the
        bucket) and replayed: one launch instead of ~15 (DESIGN.md "Config 5").  No upload (the replay kernel reads the paths from
        pinned host memory), one download (the packed result block), one synchronisation.  Returns host arrays (leaf code [R, SLOTS], child code
        [R, A, SLOTS], solved [R, A], value [R], policy [R, A]); copy=False hands out views of the pinned block, valid until
        the next call."""
        n = self.n
        depth = paths.shape[1]
        if self._paths is None or self._paths.shape[0] < depth:
            # the descents live in PINNED host memory that the replay kernel reads in place (rc_host_alias): no upload
            cap = max(16, 1 << max(depth - 1, 0).bit_length())
            self._paths = torch.full((cap, _lib.pitch_for(n)), self.A, dtype=torch.uint8).pin_memory()
            self._paths_np = self._paths.numpy()
            self._path_rows = cap                                                          # rows that may hold moves (the rest is the no-op already)
            self._graphs = {}
        rows = min(self._paths.shape[0], max(4, 1 << max(depth - 1, 0).bit_length())) if depth else 0   # what a replay of this depth reads
        self._paths_np[depth:max(rows, self._path_rows)].fill(self.A)                     # rows behind the descents: back to the no-op
        self._path_rows = rows
        if depth:
            self._paths_np[:depth, :n] = paths.T
        if self.graph:
            bucket = 0 if depth == 0 else max(4, 1 << (depth - 1).bit_length())      # replay length: no-op padded
            bucket = min(bucket, self._paths.shape[0])
            if bucket not in self._graphs:
                s = torch.cuda.Stream(self.dev)
                s.wait_stream(torch.cuda.current_stream(self.dev))
                with torch.cuda.stream(s):
                    self._device_step(bucket)                                             # warm-up outside capture
                torch.cuda.current_stream(self.dev).wait_stream(s)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._device_step(bucket)
                self._graphs[bucket] = g
            self._graphs[bucket].replay()
        else:
            self._device_step(depth)
        torch.cuda.current_stream(self.dev).synchronize()                          # the downloads are part of the device step
        code_h, cc_h, cs_h, v_h, p_h = self._host
        if not copy:
            return code_h, cc_h, cs_h, v_h, p_h
        # copies: the pinned block is overwritten by the next simulation (the Python tree keeps policy rows)
        return code_h.copy(), cc_h.copy(), cs_h.astype(bool), v_h.copy(), p_h.copy()

    def sync_rngs(self):
        """Write the per-root generators' current states back into the `random.Random` objects passed as `rngs` (native trees
        advance private copies; the Python trees consume the objects in place, nothing to do)."""
        if self.native is not None:
            self.native.sync_rngs()

    def simulate(self):
        """One simulation for every unsolved root.  Returns the number of roots solved so far."""
        if self.native is not None:
            paths = self.native.select()                                          # R tree descents (C++)
            leaf_code, child_code, solved, value, policy = self.leaves_step(paths, copy=False)   # device: replay, expand, encode, net
            done = self.native.update(leaf_code, child_code, solved, value, policy)  # R insertions + back-propagations (C++)
            self._sims = getattr(self, "_sims", 0) + 1
            if self._sims % 16 == 0 and _lib.read_status(self.dev) & _lib.STATUS_BAD_ACTION:
                raise IndexError("action out of range")                 # cube_env.py:86,96
            return done
        n = self.n
        paths, trails = [], []
        for r in range(n):
            tree, key, acts, trail = self.trees[r], b"root", [], []
            if self.solution[r] is None:
                self.sims_used[r] += 1
                rng = self.rngs[r] if self.rngs is not None else random
                while True:
                    node = tree.get(key)
                    if node is None or not node.children:
                        break
                    a = rng.randint(0, self.A - 1) if sum(node.visits) == 0 else _puct_best(node, self.c)
                    node.vloss[a] += self.vl
                    trail.append((node, a))
                    acts.append(a)
                    key = node.children[a]
            paths.append(acts)
            trails.append((key, trail))
        depth = max((len(p) for p in paths), default=0)
        pad = np.full((n, depth), self.A, np.uint8)
        for r, p in enumerate(paths):
            pad[r, :len(p)] = p
        leaf_code, child_code, solved, value, policy = self.leaves_step(pad)
        for r in range(n):
            if self.solution[r] is not None:
                continue
            key, trail = trails[r]
            tree = self.trees[r]
            kids = [child_code[r, a].tobytes() for a in range(self.A)]
            tree[key] = _Node(kids, policy[r], self.vmin, list(solved[r]))
            if key == b"root":
                tree[leaf_code[r].tobytes()] = tree[key]
            for node, a in reversed(trail):
                node.value[a] = max(node.value[a], value[r])         # numpy float32, as in the reference
                node.vloss[a] -= 150                                 # mcts.py:127 (literal)
                node.visits[a] += 1
            if solved[r].any():
                self.solution[r] = paths[r] + [int(np.argmax(solved[r]))]
        if self.sims_used and max(self.sims_used) % 16 == 0 and _lib.read_status(self.dev) & _lib.STATUS_BAD_ACTION:
            raise IndexError("action out of range")                 # cube_env.py:86,96
        return sum(s is not None for s in self.solution)
