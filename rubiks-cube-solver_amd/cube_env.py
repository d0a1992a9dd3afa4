"""CubeEnv / make_env: the reference's batch-1 env surface on top of the HIP library.

Mirrors gym-cube/gym_cube/envs/cube_env.py:12-252 and env.py:3-5 -- same method names, argument
meaning, return types and error behaviour -- so train.py / mcts.py / test.py style callers run
unchanged, while every cube operation (move, solved test, one-hot, child expansion, ADI walks)
executes in librubikhip.so on the GPU.  There is no CPU implementation behind this class.

Differences that are deliberate and documented in DESIGN.md:
  * `device` keeps the reference's meaning (where model tensors are created, cube_env.py:240,249);
    the cubes themselves live on `compute_device` (default: the current HIP device);
  * rendering (render / close_render / save_video, cube_env.py:113-130,254-275) is out of scope.
"""
from __future__ import annotations

import copy
import ctypes

import numpy as np
import torch

from . import _lib, ops
from .tables import ACTION_NAMES, get_env_config, get_tables


try:                       # the reference's class is a gym.Env (cube_env.py:12); gym is not a dependency of this package, but where a caller's
    import gym as _gym     # environment has it, CubeEnv is one too (isinstance checks, gym.make's registry entry: register_gym below)
    _EnvBase = _gym.Env
except Exception:          # not installed (or a broken install): a plain class with the same surface
    _EnvBase = object


class CubeEnv(_EnvBase):
    metadata = {"render_modes": ["human", "rgb_array"]}

    def __init__(self, device, cube_size=2, compute_device=None):
        """device: torch device for model tensors (e.g. torch.device('cpu:0')); cube_size: 2 or 3;
        compute_device: the HIP device the cube lives on (default: `device` if it is one, else the current HIP device)."""
        self.cube_size = cube_size
        self.device = device
        self.action_to_sim_action = {  # cube_env.py:24-28
            2: list(ACTION_NAMES[2]),
            3: list(ACTION_NAMES[3]),
            "render": [["U", 1], ["U", -1], ["F", 1], ["F", -1], ["R", 1], ["R", -1],
                       ["D", 1], ["D", -1], ["B", 1], ["B", -1], ["L", 1], ["L", -1]],
        }
        self.show_cube = False
        if cube_size not in (2, 3):
            raise NotImplementedError  # get_env_config / init_state, cube_env.py:30,44
        self.state_dim, self.action_dim = get_env_config(cube_size)
        self._vec = self._make_vec(compute_device)
        self._sim_cache = None
        self._cube_cache = None
        self._fast = None  # pinned result buffer + cached call arguments of the facade step path
        self._adi_plans = {}  # get_random_samples: static buffers (and, with adi_graph, the captured hipGraph) per call shape
        self.adi_graph = False  # True: get_random_samples replays its one-hot -> net -> targets body as a hipGraph (adi.AdiPlan)
        self.init_state()

    def _make_vec(self, compute_device):
        """The one-cube device state behind the facade: a VecCubeEnv(1) on a HIP device.  There is no other implementation."""
        from .vec_env import VecCubeEnv
        if compute_device is None:
            d = torch.device(self.device) if not isinstance(self.device, torch.device) else self.device
            compute_device = d if d.type == "cuda" else "cuda"
        return VecCubeEnv(1, compute_device, self.cube_size, obs="onehot", onehot_dtype=torch.uint8)

    # ------------------------------------------------------------------ state attributes
    @property
    def sim_cube(self):
        """Sticker vector, int64 ndarray [54] | [24] (py333.py:212-218 dtype)."""
        if self._sim_cache is None:
            self._sim_cache = self._vec.sim_cube[0].cpu().numpy().astype(np.int64)
        return self._sim_cache

    @sim_cube.setter
    def sim_cube(self, value):
        v = np.asarray(value)
        self._vec.set_sim_cube(v.reshape(1, -1).astype(np.uint8))
        self._sim_cache = v.astype(np.int64).copy()
        self._cube_cache = None

    @property
    def cube(self):
        """One-hot state: int64 [20,24] (py333.py:238) or float64 [7,21] (cube_env.py:143)."""
        if self._cube_cache is None:
            self._cube_cache = self._typed(self._vec.sim_state_to_state(dtype=torch.uint8)[0])
        return self._cube_cache

    def _typed(self, onehot_u8):
        a = onehot_u8.cpu().numpy()
        return a.astype(np.int64) if self.cube_size == 3 else a.astype(np.float64)

    # ------------------------------------------------------------------ reference surface
    def init_state(self):
        """Initialize state (cube_env.py:33-48)."""
        if self.cube_size not in (2, 3):
            raise NotImplementedError
        obs = self._vec.init_state()
        self._sim_cache = None
        self._cube_cache = self._typed(obs[0])

    def reset(self, seed=None, scramble_count=2):
        """Reset to a randomly scrambled cube (cube_env.py:50-69): the global legacy numpy RNG is
        saved, optionally seeded, used for `randint(action_dim, size=scramble_count)` and restored.
        Device work: one fill launch + one rc_facade_steps launch for the whole scramble."""
        origin_state = np.random.get_state()
        if seed is not None:
            np.random.seed(seed)
        action_sequence = np.random.randint(self.action_dim, size=scramble_count)
        np.random.set_state(origin_state)
        self._fill_solved()
        self._sim_cache = None
        if len(action_sequence) == 0:
            self._cube_cache = None
            raise UnboundLocalError("local variable 'state' referenced before assignment")  # cube_env.py:69
        return self.step_many([int(a) for a in action_sequence])[0]

    def step(self, action):
        """action: int 0..A-1 in the order U,U',F,F',R,R'[,D,D',B,B',L,L'].
        Returns (state ndarray, reward +1.0/-1.0, done bool, {}) -- cube_env.py:71-111."""
        if self.cube_size not in (2, 3):
            raise NotImplementedError
        names = self.action_to_sim_action[self.cube_size]
        sim_action = names[action]  # IndexError / TypeError exactly like the reference's list lookup
        idx = names.index(sim_action)  # moveInds, py333.py:41-44,221
        onehot, solved = self._step_device(idx)
        self._sim_cache = None
        self._cube_cache = onehot.astype(np.int64) if self.cube_size == 3 else onehot.astype(np.float64)
        return self._cube_cache, (1.0 if solved else -1.0), solved, {}

    def _facade(self):
        """Pinned host buffer + cached call arguments of the two batch-1 entry points (rc_facade_step / rc_facade_expand):
        the kernels write their results straight into host memory and the library spins on a sequence word -- no
        upload, no download, no stream synchronisation (include/rubikhip.h)."""
        if self._fast is None:
            v = self._vec
            host = torch.zeros(8192, dtype=torch.uint8).pin_memory()
            L = _lib.lib()
            _lib.init(v.device)
            self._fast = [host, host.numpy(), ctypes.c_void_p(host.data_ptr()), ctypes.c_void_p(v.stickers.data_ptr()),
                          int(v.stickers.shape[-1]), 0, L.rc_facade_step, L.rc_facade_expand, L.rc_facade_steps]
        f = self._fast
        f[5] = (f[5] % 0xFFFFFFFF) + 1
        return f

    # The four methods below are the facade's whole contact with the device: each is ONE launch of librubikhip.so.
    def _fill_solved(self):
        ops.fill_solved(self._vec.stickers, 1, self.cube_size)

    def _step_device(self, idx):
        f = self._facade()
        v = self._vec
        rc = f[6](f[3], f[4], self.cube_size, idx, f[2], f[5], 1, _lib.stream_ptr(v.device))
        if rc:
            _lib.check(rc)
        R, C = self.state_dim
        h = f[1]
        return h[:R * C].reshape(R, C), bool(h[496])

    def step_many(self, actions):
        """Apply a whole action sequence in ONE launch (per 60 moves) and return (state, reward, done, {}) of the final
        state, exactly what len(actions) successive step() calls leave behind -- a tree descent of MCTS.traverse
        (mcts.py:52-81).  actions: ints 0..A-1 (IndexError otherwise, like step)."""
        names = self.action_to_sim_action[self.cube_size]
        acts = bytes(names.index(names[a]) for a in actions)        # same IndexError / TypeError as step()
        onehot, solved = self._steps_device(acts)
        self._sim_cache = None
        self._cube_cache = onehot.astype(np.int64) if self.cube_size == 3 else onehot.astype(np.float64)
        return self._cube_cache, (1.0 if solved else -1.0), solved, {}

    def _steps_device(self, acts):
        f = self._facade()
        v = self._vec
        rc = f[8](f[3], f[4], self.cube_size, acts, len(acts), f[2], f[5], 1, _lib.stream_ptr(v.device))
        if rc:
            _lib.check(rc)
        R, C = self.state_dim
        return f[1][:R * C].reshape(R, C), bool(f[1][496])

    def expand_host(self, dense=False):
        """All children of the CURRENT state in one launch, results on the host (mcts.py:83-113, cube_env.py:212-236):
        (own compact code bytes, child codes uint8 [A, SLOTS], child solved bool [A][, child one-hots uint8 [A, R, C]])."""
        f = self._facade()
        v = self._vec
        rc = f[7](f[3], f[4], self.cube_size, f[2], f[5], int(dense), 1, _lib.stream_ptr(v.device))
        if rc:
            _lib.check(rc)
        A, SL = self.action_dim, ops.N_SLOTS[self.cube_size]
        h = f[1]
        out = (h[:SL].tobytes(), h[32:32 + A * SL].reshape(A, SL).copy(), h[288:288 + A].astype(bool))
        if dense:
            R, C = self.state_dim
            out += (h[512:512 + A * R * C].reshape(A, R, C).copy(),)
        return out

    def sim_state_to_state(self, sim_state):
        """One-hot of an arbitrary sticker vector (cube_env.py:132-152)."""
        if self.cube_size not in (2, 3):
            raise NotImplementedError
        st = ops.from_aos(np.asarray(sim_state).reshape(1, -1).astype(np.uint8), self._vec.device)
        out = torch.empty((1, *self.state_dim), dtype=torch.uint8, device=self._vec.device)
        ops.encode(st, 1, self.cube_size, out, _lib.FMT_U8)
        return self._typed(out[0])

    def state_to_sim_state(self, state):
        """One-hot -> stickers.  2x2x2 only (cube_env.py:154-175; unused by the reference's callers);
        a host-side table inversion, not part of the hot path."""
        if self.cube_size == 3:
            raise NotImplementedError
        if self.cube_size != 2:
            raise NotImplementedError
        t = get_tables(2)
        stickers = np.array(t.solved, dtype=np.int64)
        for cubelet, row in enumerate(np.asarray(state)):
            col = int(np.where(row == 1.0)[0][0])
            position, ori = col // 3, col % 3
            colours = [int(t.solved[i]) for i in t.corner_defs[cubelet]]
            rot = colours[-ori:] + colours[:-ori] if ori else colours
            for k in range(3):
                stickers[t.corner_defs[position][k]] = rot[k]
        return stickers

    def get_random_samples(self, replay_buffer, model, sample_scramble_count, sample_cube_count, temperature):
        """ADI samples into replay_buffer (cube_env.py:177-194): for each of sample_cube_count cubes the
        moves come from np.random.randint(action_dim, size=sample_scramble_count) on the global legacy
        RNG, exactly as the reference draws them; walks, expansion, one-hots and targets run on the GPU.  A sink with
        `append_batch` (replay.TensorReplayBuffer) receives the batch as tensors; anything else gets the reference's dicts."""
        from .adi import _module_device, _module_dtype, samples_to_dicts

        if sample_cube_count <= 0:
            return
        # the reference draws np.random.randint(action_dim, size=depth) once per cube (cube_env.py:189): ONE call of size [cubes, depth]
        # consumes numpy's legacy stream identically (same values, same final state -- the golden test checks it) in 1/16 of the time
        actions = np.random.randint(self.action_dim, size=(sample_cube_count, sample_scramble_count)).astype(np.uint8)
        if sample_scramble_count > 0:
            tensor_sink = hasattr(replay_buffer, "append_batch")         # replay.TensorReplayBuffer: no per-sample dicts
            # train.py:152-155 calls this every epoch with one shape and one model object: the buffers (and with adi_graph the
            # captured hipGraph) of that shape are kept between calls -- ONE plan per env (up to the 1 GiB dense budget of device memory,
            # about 150 MB at 200 x 30, held until close()).  The plan freezes the dtype the net computes in and where its parameters
            # live, so both are part of the key: after an in-place model.half() / model.to(...) the next call builds a new plan
            key = (id(model), sample_scramble_count, sample_cube_count, float(temperature), not tensor_sink, bool(self.adi_graph),
                   str(_module_dtype(model)), str(_module_device(model, None)))
            plan = self._adi_plans.get(key)
            if plan is None or plan.model is not model:
                self._adi_plans.clear()
                plan = self._adi_plans[key] = self._new_adi_plan(model, sample_cube_count, sample_scramble_count, temperature, not tensor_sink)
            res = plan.run(actions)
            if tensor_sink:
                replay_buffer.append_batch(res)
            else:                                                         # the reference's own ReplayBuffer / any list-like sink
                for sample in samples_to_dicts(res, self.cube_size):
                    replay_buffer.append(sample)
            # the env is left on the last walk's final state, as in the reference
            self._vec.reset(actions=actions[-1:], scramble_count=sample_scramble_count)
        else:
            self._vec.init_state()
        self._sim_cache = None
        self._cube_cache = None

    def _new_adi_plan(self, model, n_walks, depth, temperature, want_state_dense):
        """The device plan behind get_random_samples (adi.AdiPlan: generator launch, one-hot blocks, the caller's net, target assembly)."""
        from .adi import AdiPlan
        return AdiPlan(model, self.cube_size, n_walks, depth, temperature, device=self._vec.device, model_device=self.device,
                       want_state_dense=want_state_dense, graph=bool(self.adi_graph))

    def get_target_value(self, model, scramble_count, temperature):
        """(target_value, target_policy, error) of the CURRENT state (cube_env.py:196-252)."""
        if self.cube_size not in (2, 3):
            raise NotImplementedError
        _, _, solved, child_onehot = self.expand_host(dense=True)
        reward = -1.0
        if solved.any():  # lowest solved action wins, value exactly 1.0 (cube_env.py:229-232)
            reward, target_value, target_policy = 1.0, 1.0, int(np.argmax(solved))
        if reward != 1.0:
            A = self.action_dim
            next_state_tensor = torch.from_numpy(child_onehot).float().to(self.device)
            reward_tensor = torch.tensor([-1.0] * A, device=self.device)
            with torch.no_grad():
                next_value, _ = model(next_state_tensor)
                value = next_value.squeeze(dim=-1).detach() + reward_tensor
            target_value, target_policy = torch.max(value, -1, keepdim=True)
            target_value, target_policy = target_value.item(), target_policy.item()
        weight = scramble_count ** (-1 * temperature)
        with torch.no_grad():
            state_tensor = torch.tensor(self.cube, device=self.device).float()
            value, _ = model(state_tensor)
            error = abs(value.detach().item() - target_value) * weight
        return target_value, target_policy, error

    def close(self):
        """Drop the pinned result buffer of the batch-1 entry points and the library's cached device alias of it
        (rc_facade_release), so that the address can be reused by anyone, and the plan get_random_samples keeps between calls.
        Called on garbage collection too; the env stays usable (both are re-created on the next use)."""
        self._adi_plans.clear()                  # get_random_samples' static buffers (up to the 1 GiB dense budget) and captured graph
        fast, self._fast = self._fast, None
        if fast is not None:
            try:
                _lib.lib().rc_facade_release(fast[2])
            except Exception:  # interpreter shutdown: the library may be gone already
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ out of scope: rendering
    def render(self, mode=None):
        raise NotImplementedError("rendering (cube_env.py:113-122) is out of scope of the MI355X env path")

    def close_render(self):
        raise NotImplementedError("rendering (cube_env.py:124-130) is out of scope of the MI355X env path")

    def save_video(self, *args, **kwargs):
        raise NotImplementedError("rendering (cube_env.py:254-275) is out of scope of the MI355X env path")

    # mcts.py:37,96,101 deep-copies the env
    def __deepcopy__(self, memo):
        other = object.__new__(type(self))         # a subclass stays itself through mcts.py's copies
        for k, v in self.__dict__.items():
            if k == "_vec":
                other._vec = self._vec.clone(lean=True)
            elif k == "_adi_plans":
                other._adi_plans = {}  # static device buffers are not shared between copies
            elif k == "device":
                other.device = self.device
            elif k == "_fast":
                other._fast = None  # re-created lazily; buffers are not shared between copies
            else:
                setattr(other, k, copy.deepcopy(v, memo))
        return other


def register_gym(env_id="cube-v0"):
    """The reference's registry entry (gym-cube/gym_cube/__init__.py:4-7) pointing at this CubeEnv, for callers that
    keep `gym.make('cube-v0', cube_size=..., device=...)` (env.py:3-5).  Needs the `gym` package (not a dependency)."""
    from gym.envs.registration import register

    register(id=env_id, entry_point="rubiks_cube_solver_amd.cube_env:CubeEnv")
    return env_id


def make_env(device, cube_size):
    """env.py:3-5: gym.make('cube-v0', cube_size=cube_size, device=device) -> CubeEnv."""
    return CubeEnv(device=device, cube_size=cube_size)
