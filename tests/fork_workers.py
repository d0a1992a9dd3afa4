"""The reference's worker pattern, executed: a parent that imports the package WITHOUT touching the GPU, then FORKS its workers
(train.py:89-91: mp.Process, fork on Linux), each of which builds its own env inside the child (train.py:141, env.py:3-5).

    python tests/fork_workers.py OUT_DIR [N_WORKERS]

Worker `rank`: make_env(torch.device("cpu"), 3); reset(seed=rank*10, scramble_count=7); 50 step()s with actions from
default_rng(rank); writes OUT_DIR/w<rank>.npz (stickers after the reset, the actions, stickers / one-hot / reward / done of every
step).  The parent never initialises HIP: a process that has is never forked (INTEGRATION.md "Worker processes").
Driven by tests/test_gpu_env.py::test_forked_workers_build_their_own_envs, which checks the files against G4 and the oracle.
"""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import rubiks_cube_solver_amd as rc  # noqa: E402  (import only: no GPU call, the HIP library is not even loaded yet)

STEPS = 50


def worker(rank, out_dir):
    torch.set_num_threads(1)                                       # train.py:122
    env = rc.make_env(torch.device("cpu"), 3)                      # train.py:141
    first = env.reset(seed=rank * 10, scramble_count=7)
    after_reset = env.sim_cube.copy()
    acts = np.random.default_rng(rank).integers(0, 12, STEPS)
    stickers, onehots, rewards, dones = [], [], [], []
    for a in acts:
        state, reward, done, info = env.step(int(a))
        assert isinstance(reward, float) and isinstance(done, bool) and info == {}
        stickers.append(env.sim_cube.copy())
        onehots.append(state.copy())
        rewards.append(reward)
        dones.append(done)
    np.savez(os.path.join(out_dir, f"w{rank}.npz"), first=first, after_reset=after_reset, actions=acts, stickers=np.stack(stickers),
             onehots=np.stack(onehots), rewards=np.array(rewards), dones=np.array(dones), pid=os.getpid(), ppid=os.getppid())


def main():
    out_dir, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3
    assert not torch.cuda.is_initialized(), "the parent must not have touched the GPU before forking"
    ctx = mp.get_context("fork")
    procs = [ctx.Process(target=worker, args=(rank, out_dir)) for rank in range(n)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    assert not torch.cuda.is_initialized()                          # still true: the envs live in the children only
    codes = [p.exitcode for p in procs]
    print("parent", os.getpid(), "worker exit codes", codes, flush=True)
    sys.exit(0 if all(c == 0 for c in codes) else 1)


if __name__ == "__main__":
    main()
