"""MI355X-native vectorised Rubik's-cube environment (gfx950 HIP kernels behind a C ABI).

Drop-in surface for the env path of SUNGBEOMCHOI/Rubiks-Cube-Solver:
  make_env / CubeEnv  (env.py:3-5, gym-cube/gym_cube/envs/cube_env.py)   batch-1 facade
  VecCubeEnv                                                            the batched env
  TensorReplayBuffer  (utils.py:203-270 ReplayBuffer)                   device-resident sink of ADI samples
  get_env_config      (utils.py:162-186)
  ops                 batched operator layer (assets/py333.py:211-246)
  py333               the same operators under the reference's names, one cube per call
  py222               the six names cube_env.py:8 imports from the file the reference does not ship (2x2x2, one cube per call)
The directory is named `rubiks-cube-solver_amd`; import it as `rubiks_cube_solver_amd`.
"""
from .tables import ACTION_NAMES, get_env_config, get_tables  # noqa: F401

__all__ = ["get_env_config", "get_tables", "ACTION_NAMES", "ops", "make_env", "CubeEnv", "VecCubeEnv", "TensorReplayBuffer"]


def __getattr__(name):  # torch-dependent parts load lazily
    import importlib

    if name in ("ops", "_lib", "vec_env", "cube_env", "adi", "mcts_batched", "rollout", "dist", "py333", "py222", "replay"):
        return importlib.import_module(f"{__name__}.{name}")
    if name == "VecCubeEnv":
        return importlib.import_module(f"{__name__}.vec_env").VecCubeEnv
    if name == "TensorReplayBuffer":
        return importlib.import_module(f"{__name__}.replay").TensorReplayBuffer
    if name in ("CubeEnv", "make_env"):
        return getattr(importlib.import_module(f"{__name__}.cube_env"), name)
    raise AttributeError(name)
