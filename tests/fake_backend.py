"""TEST-ONLY stand-in for VecCubeEnv(1) built on the CPU oracle, so CubeEnv's host logic
(RNG handling, action lookup, return types, errors, deepcopy) can be tested without a GPU.
The product never imports this file."""
import numpy as np
import torch

from oracle.oracle_np import Oracle, STATE_DIM

_ORC = None


class OracleBackend:
    num_envs = 1

    def __init__(self, cube_size):
        global _ORC
        _ORC = _ORC or Oracle()
        self.cube_size = cube_size
        self.device = torch.device("cpu")
        self.state = _ORC.solved(cube_size, 1)
        self.action_dim = 12 if cube_size == 3 else 6

    def _obs(self):
        return torch.from_numpy(_ORC.encode(self.cube_size, self.state)[1])

    def init_state(self):
        self.state = _ORC.solved(self.cube_size, 1)
        return self._obs()

    def reset(self, actions=None, scramble_count=2, seeds=None):
        self.state = _ORC.solved(self.cube_size, 1)
        for a in np.asarray(actions).reshape(-1):
            self.state = _ORC.step(self.cube_size, self.state, np.array([a], np.uint8))[0]
        return self._obs()

    def step(self, actions):
        self.state, _, done, rew = _ORC.step(self.cube_size, self.state, np.asarray(actions, np.uint8).reshape(1))
        return self._obs(), torch.from_numpy(rew), torch.from_numpy(done), {}

    @property
    def sim_cube(self):
        return torch.from_numpy(self.state.copy())

    def set_sim_cube(self, states):
        self.state = np.asarray(states, np.uint8).reshape(1, -1).copy()

    def sim_state_to_state(self, dtype=None):
        return self._obs()

    def clone(self):
        o = OracleBackend(self.cube_size)
        o.state = self.state.copy()
        return o
