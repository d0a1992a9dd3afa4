"""VecCubeEnv: N independent cubes resident in HBM, stepped by one HIP launch.

The batched counterpart of the reference's CubeEnv (gym-cube/gym_cube/envs/cube_env.py:12-111):
same action order, reward (+1.0 solved / -1.0 otherwise), done flag and one-hot state
convention, for millions of cubes at once.  All cube arithmetic runs in librubikhip.so; this
class only owns tensors and forwards pointers.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops
from .tables import ACTION_NAMES, get_env_config


def legacy_scramble_actions(seeds, scramble_count, action_dim):
    """The reference's reset() draw for each seed: np.random.seed(s); randint(A, size=k), with the
    caller's global legacy RNG state saved and restored (cube_env.py:62-68).  -> uint8 [len(seeds), k]."""
    saved = np.random.get_state()
    try:
        out = np.empty((len(seeds), scramble_count), np.uint8)
        for i, s in enumerate(seeds):
            np.random.seed(int(s))
            out[i] = np.random.randint(action_dim, size=scramble_count)
    finally:
        np.random.set_state(saved)
    return out


class VecCubeEnv:
    """num_envs cubes on one GPU.

    obs: "onehot" -> step/reset return the dense one-hot [N, R, C] (`onehot_dtype`), the layout
         model.py:31-45 consumes;  "code" -> the compact uint8 code buffer [tiles, SLOTS, pitch]
         (lossless, 20 B per cube instead of 1920);  None -> no observation is produced.
    seed / stream_id: device RNG stream for reset() without explicit seeds (stream_id = rank in
         multi-GPU jobs gives every rank an independent stream, no communication).
    """

    def __init__(self, num_envs, device="cuda", cube_size=3, obs="onehot", onehot_dtype=torch.float32,
                 seed=0, stream_id=0, debug_check_every=0):
        self.state_dim, self.action_dim = get_env_config(cube_size)
        self.cube_size = cube_size
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.RubikHipError("VecCubeEnv runs on a HIP device only (there is no CPU fallback)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if obs not in ("onehot", "code", None):
            raise ValueError("obs must be 'onehot', 'code' or None")
        self.obs = obs
        self.action_names = list(ACTION_NAMES[cube_size])
        self.seed, self.stream_id = int(seed), int(stream_id)
        self._resets = 0
        self._steps, self.debug_check_every = 0, int(debug_check_every)
        n, dev = self.num_envs, self.device
        self.stickers = ops.alloc_states(n, cube_size, dev)
        self.reward = torch.empty(n, dtype=torch.float32, device=dev)
        self.done = torch.empty(n, dtype=torch.uint8, device=dev)
        self._fmt = _lib.FMT_NONE
        self._obs_buf = None
        if obs == "onehot":
            self._fmt = _lib.fmt_of(onehot_dtype)
            self._obs_buf = torch.empty((n, *self.state_dim), dtype=onehot_dtype, device=dev)
        elif obs == "code":
            self._fmt = _lib.FMT_CODE
            self._obs_buf = ops.alloc_code(n, cube_size, dev)
        self.init_state()

    # ------------------------------------------------------------------ reference surface
    def init_state(self):
        """All cubes solved (cube_env.py:33-42)."""
        ops.fill_solved(self.stickers, self.num_envs, self.cube_size)
        return self._observe()

    def reset(self, seeds=None, scramble_count=2, actions=None):
        """Solved, then `scramble_count` random face turns per cube (cube_env.py:50-69).

        seeds   : one int per env -> each env i gets exactly the reference's reset(seed=seeds[i],
                  scramble_count[i]) move sequence; numpy's legacy generator runs on the device;
        actions : explicit uint8 [N, K] moves (overrides seeds); the value `action_dim` is a no-op,
                  so rows may be padded to a common length;
        neither : moves drawn on the device from (seed, stream_id, reset counter) -- reproducible,
                  rank-independent streams, no host work.
        scramble_count may be one int or (with seeds) one int per env.
        Returns the observation of the scrambled cubes."""
        n = self.num_envs
        counts = np.broadcast_to(np.asarray(scramble_count, dtype=np.int64), (n,)) if np.ndim(scramble_count) else None
        kmax = int(counts.max()) if counts is not None else int(scramble_count)
        if kmax <= 0 or (counts is not None and int(counts.min()) <= 0):
            # the reference returns an unbound `state` here (UnboundLocalError, cube_env.py:69)
            raise UnboundLocalError("reset(scramble_count=0): the reference has no state to return")
        ops.fill_solved(self.stickers, n, self.cube_size)
        if actions is None and seeds is not None:
            if len(seeds) != n:
                raise ValueError("need one seed per env")
            # the reference's np.random.seed(s); randint(A, size=k) per env, generated on the device
            # (rc_legacy_scramble_actions): bit-exact and independent of the host's global RNG state
            s_t = seeds if isinstance(seeds, torch.Tensor) else torch.as_tensor(np.asarray(seeds, dtype=np.int64))
            buf, kk = ops.legacy_scramble_actions(s_t, self.cube_size,
                                                  kmax if counts is None else counts.tolist(), device=self.device)
            ops.scramble(self.stickers, n, self.cube_size, kk, actions_in=buf, done=self.done, reward=self.reward)
            return self._observe()
        elif counts is not None:
            raise ValueError("per-env scramble counts need seeds (or pad explicit actions with the no-op)")
        if actions is not None:
            a = torch.as_tensor(actions, dtype=torch.uint8)
            if a.dim() != 2 or a.shape[0] != n:
                raise ValueError(f"actions must be [{n}, K]")
            if int(a.max()) > self.action_dim:
                raise IndexError("action out of range")  # cube_env.py:86,96
            k = a.shape[1]
            buf = torch.full((k, _lib.pitch_for(n)), self.action_dim, dtype=torch.uint8)
            buf[:, :n] = a.t()
            ops.scramble(self.stickers, n, self.cube_size, k, actions_in=buf.to(self.device),
                         done=self.done, reward=self.reward)
        else:
            self._resets += 1
            ops.scramble(self.stickers, n, self.cube_size, kmax, seed=self.seed, stream_id=self.stream_id,
                         walk_offset=self._resets * n, done=self.done, reward=self.reward)
        return self._observe()

    def step(self, actions, active=None):
        """One face turn per cube.  actions: uint8 tensor [N] on the env's device (anything else is
        converted and range-checked).  active: optional bool tensor [N]; cubes where it is False get
        the no-op (they keep their state; used by batched rollouts to park solved cubes).
        Returns (obs, reward float32 [N] of +-1.0, done uint8 [N], {}) -- cube_env.py:71-111.

        The three returned tensors are the env's OWN buffers, overwritten by the next step / reset: clone what
        must outlive it.  A uint8 device tensor is not range-checked on the host (that would synchronise): an
        action > action_dim leaves that cube unspecified and raises IndexError at the next check_actions()
        (`debug_check_every=K` in the constructor runs that check every K steps)."""
        a = self._actions(actions)
        self._steps += 1
        if self.debug_check_every and self._steps % self.debug_check_every == 0:
            self.check_actions()
        if active is not None:
            a = torch.where(active.to(self.device), a, torch.full_like(a, self.action_dim))
        ops.apply_moves(self.stickers, self.stickers, a, self.num_envs, self.cube_size, self.reward, self.done,
                        self._obs_buf, self._fmt)
        return self._obs_buf, self.reward, self.done, {}

    def is_solved(self):
        ops.is_solved(self.stickers, self.num_envs, self.cube_size, self.done, self.reward)
        return self.done

    def sim_state_to_state(self, stickers=None, out=None, dtype=None):
        """Dense one-hot of the given (default: current) sticker buffer (cube_env.py:132-152)."""
        st = self.stickers if stickers is None else stickers
        if out is None:
            out = torch.empty((self.num_envs, *self.state_dim), dtype=dtype or torch.float32, device=self.device)
        ops.encode(st, self.num_envs, self.cube_size, out, _lib.fmt_of(out.dtype))
        return out

    # ------------------------------------------------------------------------- batched extras
    @property
    def sim_cube(self):
        """[N, S] uint8 tensor: one row per cube (a copy; the live buffer is `stickers`)."""
        return ops.to_aos(self.stickers, self.num_envs).contiguous()

    def set_sim_cube(self, states):
        """Load [N, S] sticker rows (host or device)."""
        t = torch.as_tensor(states, dtype=torch.uint8).cpu()
        self.stickers.copy_(ops.from_aos(t, self.device, self.stickers.shape[2]))

    def expand(self, children=False, codes=True, pitch=None):
        """All A children of every cube (cube_env.py:212-236, mcts.py:96-101).
        Returns dict(child_solved [A, Wp], child_code [A, tiles, SLOTS, pitch], children [A, tiles, S, pitch]);
        ops.to_aos(buf[a], n) turns one child's tiled buffer into [n, rows]."""
        n, cs = self.num_envs, self.cube_size
        _, pitch = ops._tile_shape(n, pitch)
        out = ops.expand_buffers(n, cs, self.device, pitch, children=children, codes=codes)
        ops.expand_children(self.stickers, n, cs, out.get("children"), out["child_solved"], out.get("child_code"), pitch=pitch)
        return out

    def check_actions(self):
        """Raise IndexError if any kernel since the last check saw an out-of-range action (synchronises)."""
        if _lib.read_status(self.device) & _lib.STATUS_BAD_ACTION:
            raise IndexError("action out of range")  # cube_env.py:86,96

    def clone(self, lean=False):
        """Independent copy of the cubes.  lean: copy only the sticker buffer (the state); reward / done / observation
        are outputs of the next step and get fresh uninitialised buffers -- one copy kernel instead of four (the batch-1
        facade, which mcts.py:37 deep-copies once per simulation, keeps its observation on the host)."""
        other = object.__new__(VecCubeEnv)
        other.__dict__.update(self.__dict__)
        other.stickers = self.stickers.clone()
        for k in ("reward", "done", "_obs_buf"):
            v = getattr(self, k)
            setattr(other, k, None if v is None else (torch.empty_like(v) if lean else v.clone()))
        return other

    __copy__ = clone

    def __deepcopy__(self, memo):
        return self.clone()

    # ------------------------------------------------------------------------------- helpers
    def _actions(self, actions):
        a = actions
        if isinstance(a, torch.Tensor) and a.numel() != self.num_envs:
            raise ValueError(f"need {self.num_envs} actions")
        if not (isinstance(a, torch.Tensor) and a.dtype == torch.uint8 and a.device == self.device):
            a = torch.as_tensor(np.asarray(actions) if not isinstance(actions, torch.Tensor) else actions)
            if a.numel() != self.num_envs:
                raise ValueError(f"need {self.num_envs} actions")
            if a.numel() and (int(a.max()) >= self.action_dim or int(a.min()) < 0):
                raise IndexError("action out of range")
            a = a.to(device=self.device, dtype=torch.uint8)
        return a.contiguous().reshape(-1)

    def _observe(self):
        if self._obs_buf is not None:
            ops.encode(self.stickers, self.num_envs, self.cube_size, self._obs_buf, self._fmt)
        return self._obs_buf
