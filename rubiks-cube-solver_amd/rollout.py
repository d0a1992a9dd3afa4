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
def greedy_rollout(model, env: VecCubeEnv, max_timesteps, mask=False, sync_every=8, graph=False):
    """Roll every cube of `env` (obs must be 'onehot') forward greedily from its current state.

    graph=True captures ONE time step (net forward, action choice, rc_apply_moves, bookkeeping) as a hipGraph and
    replays it: small batches are launch-bound (~15 launches per step otherwise).

    Returns dict(solved bool [N], solve_step int32 [N] (0 = not solved within max_timesteps, else the 1-based
    step at which done came, as test.py:151), actions uint8 [T, N] (no-op = action_dim after a cube is done))."""
    if env.obs != "onehot":
        raise ValueError("greedy_rollout needs VecCubeEnv(obs='onehot')")
    n, dev, A = env.num_envs, env.device, env.action_dim
    env._observe()
    active = torch.ones(n, dtype=torch.bool, device=dev)          # the reference starts every trial with done = False
    solve_step = torch.zeros(n, dtype=torch.int32, device=dev)
    pre = torch.full((n,), -1, dtype=torch.int64, device=dev)
    taken = torch.full((max_timesteps, n), A, dtype=torch.uint8, device=dev)
    t_dev = torch.zeros(1, dtype=torch.int64, device=dev)         # 0-based index of the step being taken
    noop = torch.full((n,), A, dtype=torch.int64, device=dev)

    def one_step():
        logits = model(env._obs_buf.float())[1]
        top2 = torch.topk(logits, 2, dim=-1).indices               # model.py:71-74: best, else second best
        a = top2[:, 0]
        if mask:
            invalid = torch.where(pre >= 0, pre ^ 1, pre)          # model.py:64-69: U<->U', F<->F', ...
            a = torch.where(a == invalid, top2[:, 1], a)
            pre.copy_(torch.where(active, a, pre))
        a8 = torch.where(active, a, noop).to(torch.uint8)
        _, _, done, _ = env.step(a8)
        taken.scatter_(0, t_dev.expand(1, n), a8.unsqueeze(0))
        newly = active & (done != 0)
        solve_step.copy_(torch.where(newly, (t_dev + 1).to(torch.int32).expand(n), solve_step))
        active.logical_and_(~newly)
        t_dev.add_(1)

    g = None
    if graph and max_timesteps > 1:
        snap = [x.clone() for x in (env.stickers, env._obs_buf, active, solve_step, pre, taken, t_dev)]
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            one_step()                                             # warm-up outside capture ...
        torch.cuda.current_stream(dev).wait_stream(s)
        for x, y in zip((env.stickers, env._obs_buf, active, solve_step, pre, taken, t_dev), snap):
            x.copy_(y)                                             # ... then rewind to the start state
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            one_step()
        for x, y in zip((env.stickers, env._obs_buf, active, solve_step, pre, taken, t_dev), snap):
            x.copy_(y)                                             # capture does not execute, but keep it exact
    steps_done = 0
    for t in range(1, max_timesteps + 1):
        g.replay() if g is not None else one_step()
        steps_done = t
        if t % sync_every == 0 and not bool(active.any()):
            break
    env.check_actions()   # the device cannot raise mid-launch: surface an out-of-range action (IndexError, cube_env.py:86,96) here
    return {"solved": solve_step > 0, "solve_step": solve_step, "actions": taken[:steps_done]}


@torch.no_grad()
def solve_percentage(model, cube_size, sample_scramble_count, sample_cube_count, max_timesteps, device="cuda",
                     mask=False, seeds=None, graph=False):
    """train.py:167-198: for scramble_count = 1..sample_scramble_count, the percentage of the
    sample_cube_count cubes (seeds i*10, train.py:180) the greedy policy solves within max_timesteps.
    All scramble_count x cube pairs run as ONE batch.  -> list of percentages, as valid_history stores.
    graph: replay one time step as a hipGraph (greedy_rollout(graph=True)): 125 us instead of 307 us per step at the reference's 300 cubes."""
    seeds = list(seeds) if seeds is not None else [i * 10 for i in range(sample_cube_count)]
    ks = [k for k in range(1, sample_scramble_count + 1) for _ in seeds]
    env = VecCubeEnv(len(ks), device, cube_size, obs="onehot", onehot_dtype=torch.float32)
    env.reset(seeds=seeds * sample_scramble_count, scramble_count=ks)
    res = greedy_rollout(model, env, max_timesteps, mask=mask, graph=graph)
    solved = res["solved"].view(sample_scramble_count, len(seeds)).float().mean(1) * 100.0
    return [float(x) for x in solved.cpu()]
