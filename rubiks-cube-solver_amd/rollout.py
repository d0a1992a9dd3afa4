"""Vectorised greedy rollouts: the solve loops of train.py:167-198
(`validation`) and test.py:103-158 (`trial`, greedy branch) for thousands of cubes at once.

Per time step: ONE value/policy-net forward on all active cubes and ONE rc_apply_moves launch;
solved cubes are parked with the no-op action.  Action choice restates model.py:47-76
(`get_action`): the arg-max of the policy output, or the runner-up when the arg-max would undo
the previous move (`pre_action` mask, test.py:131-133).
"""
from __future__ import annotations

import torch

from .vec_env import VecCubeEnv


@torch.no_grad()
def greedy_rollout(model, env: VecCubeEnv, max_timesteps, mask=False, sync_every=8):
    """Roll every cube of `env` (obs must be 'onehot') forward greedily from its current state.

    Returns dict(solved bool [N], solve_step int32 [N] (0 = not solved within max_timesteps, else the 1-based
    step at which done came, as test.py:151), actions uint8 [T, N] (no-op = action_dim after a cube is done))."""
    if env.obs != "onehot":
        raise ValueError("greedy_rollout needs VecCubeEnv(obs='onehot')")
    n, dev, A = env.num_envs, env.device, env.action_dim
    obs = env._observe()
    active = torch.ones(n, dtype=torch.bool, device=dev)          # the reference starts every trial with done = False
    solve_step = torch.zeros(n, dtype=torch.int32, device=dev)
    pre = torch.full((n,), -1, dtype=torch.int64, device=dev)
    taken = []
    for t in range(1, max_timesteps + 1):
        logits = model(obs.float())[1]
        top2 = torch.topk(logits, 2, dim=-1).indices               # model.py:71-74: best, else second best
        a = top2[:, 0]
        if mask:
            invalid = torch.where(pre >= 0, pre ^ 1, pre)          # model.py:64-69: U<->U', F<->F', ...
            a = torch.where(a == invalid, top2[:, 1], a)
            pre = torch.where(active, a, pre)
        a8 = torch.where(active, a, torch.full_like(a, A)).to(torch.uint8)
        obs, _, done, _ = env.step(a8)
        taken.append(a8)
        newly = active & (done != 0)
        solve_step = torch.where(newly, torch.full_like(solve_step, t), solve_step)
        active = active & ~newly
        if t % sync_every == 0 and not bool(active.any()):
            break
    return {"solved": solve_step > 0, "solve_step": solve_step, "actions": torch.stack(taken) if taken else None}


@torch.no_grad()
def solve_percentage(model, cube_size, sample_scramble_count, sample_cube_count, max_timesteps, device="cuda",
                     mask=False, seeds=None):
    """train.py:167-198: for scramble_count = 1..sample_scramble_count, the percentage of the
    sample_cube_count cubes (seeds i*10, train.py:180) the greedy policy solves within max_timesteps.
    All scramble_count x cube pairs run as ONE batch.  -> list of percentages, as valid_history stores."""
    seeds = list(seeds) if seeds is not None else [i * 10 for i in range(sample_cube_count)]
    ks = [k for k in range(1, sample_scramble_count + 1) for _ in seeds]
    env = VecCubeEnv(len(ks), device, cube_size, obs="onehot", onehot_dtype=torch.float32)
    env.reset(seeds=seeds * sample_scramble_count, scramble_count=ks)
    res = greedy_rollout(model, env, max_timesteps, mask=mask)
    solved = res["solved"].view(sample_scramble_count, len(seeds)).float().mean(1) * 100.0
    return [float(x) for x in solved.cpu()]
