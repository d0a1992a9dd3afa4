"""CPU oracle, Python side.  TEST INFRASTRUCTURE ONLY (see rc_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module,
and only as the checker / timed comparator; the product package never does.

Two restatements live here:
  * ``Oracle``          -- ctypes front end of rc_oracle.c (batched, array-of-structures);
  * ``OracleCubeEnv``   -- a numpy, one-cube-at-a-time env with the structure of the
                           reference's CubeEnv (gym-cube/gym_cube/envs/cube_env.py:12-252):
                           the "reference-style CPU env" that SURVEY.md section 8d times.

3x3x3 tables come from tests/golden/tables_333.npz = the reference's own arrays
(PINNED).  2x2x2 tables restate the public MeepMoop/py222 algorithm: the six
permutations are the corner-sticker restriction of the golden 3x3x3 table, piece
definitions and the 58-row hash table are py222's published ones (PARITY UNPINNED: the
reference does not ship its py222, cube_env.py:8).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_GOLDEN = os.path.join(os.path.dirname(_HERE), "tests", "golden")
_LIB = os.environ.get("RC_ORACLE_LIB") or os.path.join(_HERE, "librc_oracle.so")   # env override: sanitizer builds (tools/sanitize_cpu.sh)

ACTION_NAMES = {2: ["U", "U'", "F", "F'", "R", "R'"],
                3: ["U", "U'", "F", "F'", "R", "R'", "D", "D'", "B", "B'", "L", "L'"]}  # cube_env.py:24-27
STATE_DIM = {2: (7, 21), 3: (20, 24)}  # utils.py:177-182


def build_library(force: bool = False) -> str:
    src = os.path.join(_HERE, "rc_oracle.c")
    if os.environ.get("RC_ORACLE_LIB"):
        return _LIB                                                  # a caller-provided build is used as it is
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "librc_oracle.so"])
    return _LIB


def tables_333():
    g = np.load(os.path.join(_GOLDEN, "tables_333.npz"))
    return dict(
        S=54, A=12, perm=g["moveDefs"].astype(np.uint8),
        corner_defs=g["corner_pieceDefs"].astype(np.uint8), edge_defs=g["edge_pieceDefs"].astype(np.uint8),
        corner_lut=g["corner_pieceInds"].astype(np.uint8), edge_lut=g["edge_pieceInds"].astype(np.uint8),
    )


def tables_222():
    """Public py222: stickers U0-3 R4-7 F8-11 D12-15 L16-19 B20-23; DLB cubie never moves."""
    t3 = tables_333()
    corner_idx = [9 * f + k for f in range(6) for k in (0, 2, 6, 8)]
    pos = {s: i for i, s in enumerate(corner_idx)}
    perm = np.array([[pos[int(t3["perm"][a][s])] for s in corner_idx] for a in range(6)], np.uint8)
    piece_defs = np.array([[0, 21, 16], [2, 17, 8], [3, 9, 4], [1, 5, 20],
                           [12, 10, 19], [13, 6, 11], [15, 22, 7]], np.uint8)
    lut = np.zeros((58, 2), np.uint8)
    rows = {50: (0, 0), 54: (0, 1), 13: (0, 2), 28: (1, 0), 42: (1, 1), 8: (1, 2),
            14: (2, 0), 21: (2, 1), 4: (2, 2), 52: (3, 0), 15: (3, 1), 11: (3, 2),
            47: (4, 0), 30: (4, 1), 40: (4, 2), 25: (5, 0), 18: (5, 1), 35: (5, 2),
            23: (6, 0), 57: (6, 1), 37: (6, 2)}
    for h, v in rows.items():
        lut[h] = v
    return dict(S=24, A=6, perm=perm, corner_defs=piece_defs, edge_defs=np.zeros((0, 2), np.uint8),
                corner_lut=lut, edge_lut=np.zeros((0, 2), np.uint8))


def tables(cube_size):
    if cube_size == 3:
        return tables_333()
    if cube_size == 2:
        return tables_222()
    raise NotImplementedError


class Oracle:
    """ctypes wrapper over rc_oracle.c.  Arrays are array-of-structures: [n][S] uint8."""

    def __init__(self):
        self.lib = ctypes.CDLL(build_library())
        L = self.lib
        u8p, i64, i32, u64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64
        L.orc_set_tables.argtypes = [i32, i32, i32, u8p, i32, u8p, i32, u8p, i32, u8p, i32, u8p]
        L.orc_step_batch.argtypes = [i32, u8p, u8p, i64, u8p, u8p, u8p, i32]
        L.orc_expand_batch.argtypes = [i32, u8p, i64, u8p, u8p, u8p, i32]
        L.orc_adi_generate.argtypes = [i32, u64, u64, i64, i64, i32, u8p, u8p, u8p, u8p, u8p, u8p, u8p, i32]
        L.orc_time_steps.argtypes = [i32, u8p, u8p, i64, i32, i32, u8p, u8p, i32]
        L.orc_time_steps.restype = ctypes.c_double
        L.orc_is_solved.argtypes = [i32, u8p]
        L.orc_onehot.argtypes = [i32, u8p, u8p, u8p]
        L.orc_walk_rng_seed.argtypes = [u64, u64, u64, u8p]
        L.orc_walk_rng_action.argtypes = [u8p, ctypes.c_uint32]
        L.orc_walk_rng_action.restype = ctypes.c_uint32
        self.t = {}
        for cs in (2, 3):
            t = tables(cs)
            self.t[cs] = t
            rc = L.orc_set_tables(
                cs, t["S"], t["A"], _p(t["perm"]), len(t["corner_defs"]), _p(t["corner_defs"]),
                len(t["edge_defs"]), _p(t["edge_defs"]), len(t["corner_lut"]), _p(t["corner_lut"]),
                len(t["edge_lut"]), _p(t["edge_lut"]))
            assert rc == 0

    @staticmethod
    def slots(cube_size):
        return 20 if cube_size == 3 else 7

    def max_threads(self):
        return int(self.lib.orc_max_threads())

    def solved(self, cube_size, n=1):
        S = self.t[cube_size]["S"]
        return np.tile(np.repeat(np.arange(6, dtype=np.uint8), S // 6), (n, 1))

    def step(self, cube_size, states, actions, threads=1):
        """states [n][S] (copied), actions [n] -> (new_states, code [n][slots], done [n], reward [n])."""
        st = np.ascontiguousarray(states, np.uint8).copy()
        ac = np.ascontiguousarray(actions, np.uint8)
        n = st.shape[0]
        code = np.zeros((n, self.slots(cube_size)), np.uint8)
        done = np.zeros(n, np.uint8)
        rew = np.zeros(n, np.float32)
        rc = self.lib.orc_step_batch(cube_size, _p(st), _p(ac), n, _p(code), _p(done), _p(rew), threads)
        if rc:
            raise IndexError("action out of range or hash outside the LUT")
        return st, code, done, rew

    def is_solved(self, cube_size, states):
        st = np.ascontiguousarray(states, np.uint8)
        return np.array([self.lib.orc_is_solved(cube_size, _p(s)) for s in st], np.uint8)

    def encode(self, cube_size, states):
        """-> (code [n][slots], onehot [n][R][C] uint8)."""
        st = np.ascontiguousarray(states, np.uint8)
        R, C = STATE_DIM[cube_size]
        code = np.zeros((len(st), self.slots(cube_size)), np.uint8)
        oh = np.zeros((len(st), R, C), np.uint8)
        for i in range(len(st)):
            if self.lib.orc_onehot(cube_size, _p(st[i]), _p(oh[i]), _p(code[i])):
                raise IndexError("hash outside the LUT")
        return code, oh

    def expand(self, cube_size, parents, threads=1):
        st = np.ascontiguousarray(parents, np.uint8)
        n, S, A = len(st), self.t[cube_size]["S"], self.t[cube_size]["A"]
        ch = np.zeros((n, A, S), np.uint8)
        cc = np.zeros((n, A, self.slots(cube_size)), np.uint8)
        cs = np.zeros((n, A), np.uint8)
        if self.lib.orc_expand_batch(cube_size, _p(st), n, _p(ch), _p(cc), _p(cs), threads):
            raise IndexError("hash outside the LUT")
        return ch, cc, cs

    def adi(self, cube_size, n_walks, depth, seed=0, stream=0, walk0=0, actions_in=None, threads=1,
            want_children=True):
        S, A, sl = self.t[cube_size]["S"], self.t[cube_size]["A"], self.slots(cube_size)
        out = dict(
            actions=np.zeros((n_walks, depth), np.uint8),
            parents=np.zeros((n_walks, depth, S), np.uint8),
            parent_code=np.zeros((n_walks, depth, sl), np.uint8),
            child_code=np.zeros((n_walks, depth, A, sl), np.uint8),
            child_solved=np.zeros((n_walks, depth, A), np.uint8),
        )
        if want_children:
            out["children"] = np.zeros((n_walks, depth, A, S), np.uint8)
        ai = None if actions_in is None else np.ascontiguousarray(actions_in, np.uint8)
        rc = self.lib.orc_adi_generate(
            cube_size, seed, stream, walk0, n_walks, depth, _p(ai) if ai is not None else None,
            _p(out["actions"]), _p(out["parents"]), _p(out["parent_code"]),
            _p(out["children"]) if want_children else None, _p(out["child_code"]), _p(out["child_solved"]), threads)
        if rc:
            raise IndexError("action out of range or hash outside the LUT")
        return out

    def rng_actions(self, seed, stream, walk, n, A):
        s = np.zeros(2, np.uint64)
        self.lib.orc_walk_rng_seed(seed, stream, walk, _p(s))
        return np.array([self.lib.orc_walk_rng_action(_p(s), A) for _ in range(n)], np.uint8)

    def time_steps(self, cube_size, states, actions, iters, with_code, threads):
        st = np.ascontiguousarray(states, np.uint8).copy()
        ac = np.ascontiguousarray(actions, np.uint8)
        n = len(st)
        code = np.zeros((n, self.slots(cube_size)), np.uint8)
        done = np.zeros(n, np.uint8)
        return float(self.lib.orc_time_steps(cube_size, _p(st), _p(ac), n, iters, int(with_code), _p(code), _p(done), threads))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------- numpy, per cube
class OracleCubeEnv:
    """One cube at a time, numpy fancy indexing, Python loops: the reference's structure.

    cube_env.py:12-252 restated (render hooks left out: out of scope)."""

    def __init__(self, device=None, cube_size=3):
        if cube_size not in (2, 3):
            raise NotImplementedError  # cube_env.py:44
        self.cube_size, self.device = cube_size, device
        self.t = tables(cube_size)
        self.action_to_sim_action = {2: ACTION_NAMES[2], 3: ACTION_NAMES[3]}
        self.move_index = {n: i for i, n in enumerate(ACTION_NAMES[3 if cube_size == 3 else 2])}
        self.state_dim, self.action_dim = list(STATE_DIM[cube_size]), self.t["A"]  # utils.py:162-186
        self._perm = self.t["perm"].astype(np.int64)
        self._cdefs = self.t["corner_defs"].astype(np.int64)
        self._edefs = self.t["edge_defs"].astype(np.int64)
        self._clut = self.t["corner_lut"].astype(np.int64)
        self._elut = self.t["edge_lut"].astype(np.int64)
        self.init_state()

    # py333.py:211-218 / cube_env.py:33-42
    def init_state(self):
        self.sim_cube = np.repeat(np.arange(6), self.t["S"] // 6)
        self.cube = self.sim_state_to_state(self.sim_cube)

    # py333.py:224-227
    def _get_op(self, s):
        c = self._clut[s[self._cdefs] @ np.array([1, 2, 10])]
        if len(self._edefs):
            e = self._elut[s[self._edefs] @ np.array([1, 10])]
            return np.concatenate((c, e))
        return c

    # cube_env.py:132-152 + py333.py:235-246
    def sim_state_to_state(self, s):
        op = self._get_op(s)
        if self.cube_size == 3:
            state = np.zeros((20, 24), dtype=int)
            for slot, (piece, ori) in enumerate(op):
                state[slot][piece * (3 if slot < 8 else 2) + ori] = 1
        else:
            state = np.zeros(self.state_dim)
            for slot, (piece, ori) in enumerate(op):
                state[piece][slot * 3 + ori] = 1.0
        return state

    # py333.py:229-233
    def _is_solved(self, s):
        f = self.t["S"] // 6
        return all((s[f * i:f * i + f] == s[f * i]).all() for i in range(6))

    # cube_env.py:71-111
    def step(self, action):
        name = self.action_to_sim_action[self.cube_size][action]
        self.sim_cube = self.sim_cube[self._perm[self.move_index[name]]]  # py333.py:220-222
        self.cube = self.sim_state_to_state(self.sim_cube)
        done = bool(self._is_solved(self.sim_cube))
        return self.cube, (1.0 if done else -1.0), done, {}

    # cube_env.py:50-69
    def reset(self, seed=None, scramble_count=2):
        self.init_state()
        saved = np.random.get_state()
        if seed is not None:
            np.random.seed(seed)
        for a in np.random.randint(self.action_dim, size=scramble_count):
            state, _, _, _ = self.step(a)
        np.random.set_state(saved)
        return state

    # cube_env.py:196-252
    def get_target_value(self, model, scramble_count, temperature):
        import torch

        rewards, nexts = [], []
        for a in range(self.action_dim):
            child = self.sim_cube[self._perm[a]]
            if self._is_solved(child):
                reward, target_value, target_policy = 1.0, 1.0, a
                break
            reward = -1.0
            nexts.append(self.sim_state_to_state(child))
            rewards.append(reward)
        if reward != 1.0:
            x = torch.tensor(np.array(nexts), device=self.device).float()
            with torch.no_grad():
                v = model(x)[0].squeeze(dim=-1).detach() + torch.tensor(rewards, device=self.device)
            tv, tp = torch.max(v, -1, keepdim=True)
            target_value, target_policy = tv.item(), tp.item()
        weight = scramble_count ** (-1 * temperature)
        with torch.no_grad():
            v = model(torch.tensor(self.cube, device=self.device).float())[0]
        return target_value, target_policy, abs(v.detach().item() - target_value) * weight

    # cube_env.py:177-194
    def get_random_samples(self, replay_buffer, model, sample_scramble_count, sample_cube_count, temperature):
        for _ in range(sample_cube_count):
            self.init_state()
            for d, a in enumerate(np.random.randint(self.action_dim, size=sample_scramble_count)):
                state, _, _, _ = self.step(a)
                tv, tp, err = self.get_target_value(model, d + 1, temperature)
                replay_buffer.append({"state": state, "target_value": tv, "target_policy": tp,
                                      "scramble_count": d + 1, "error": err})
