#!/usr/bin/env python3
"""Fixture for SURVEY.md section 8f N4 on the HIP path: greedy solve traces of the reference's shipped 2x2x2
checkpoint (pretrained/222model.pt, model.py:47-76 arg-max policy) on the restated 2x2x2 env, for seeds 0..39 x
scramble depths {1,2,3,4,6,8} (reset(seed, k), cube_env.py:50-69).

Runs only in the build container (needs /root/reference); writes tests/golden/crosscheck_222.npz holding ACTIONS AND
OUTCOMES ONLY -- no weights, no reference source:
    seeds [N], ks [N], scramble [N, 8] (no-op 6 padded), actions [N, T] (no-op 6 after the solve / at the horizon),
    cols [N, T, 7] (arg-max column of every one-hot row after each step), done [N, T], solve_step [N] (0 = unsolved).
The checkpoint is read statically (tests/test_crosscheck_222.py: pickle opcodes + raw float32 zip members, nothing is
unpickled or executed).

    python tests/golden/make_crosscheck_222.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from test_crosscheck_222 import CKPT, policy_logits, read_state_dict_statically  # noqa: E402
from oracle.oracle_np import OracleCubeEnv  # noqa: E402

T = 30


def main():
    sd = read_state_dict_statically(CKPT)
    env = OracleCubeEnv(None, 2)
    seeds, ks, scr, acts, cols, done, solve = [], [], [], [], [], [], []
    for k in (1, 2, 3, 4, 6, 8):
        for seed in range(40):
            saved = np.random.get_state()
            np.random.seed(seed)
            draw = np.random.randint(6, size=k)                   # what reset(seed, k) draws (cube_env.py:64-65)
            np.random.set_state(saved)
            state = env.reset(seed=seed, scramble_count=k)
            a_row, c_row, d_row, s_at = np.full(T, 6, np.uint8), np.zeros((T, 7), np.uint8), np.zeros(T, np.uint8), 0
            for t in range(T):
                a = int(np.argmax(policy_logits(sd, state[None].astype(np.float32))[0]))
                state, _, d, _ = env.step(a)
                a_row[t], c_row[t], d_row[t] = a, np.argmax(state, 1), d
                if d:
                    s_at = t + 1
                    c_row[t + 1:], d_row[t + 1:] = c_row[t], 1     # parked with the no-op: state and flag stay
                    break
            seeds.append(seed); ks.append(k)
            scr.append(np.concatenate([draw, np.full(8 - k, 6)]).astype(np.uint8))
            acts.append(a_row); cols.append(c_row); done.append(d_row); solve.append(s_at)
    out = dict(seeds=np.array(seeds, np.int64), ks=np.array(ks, np.int32), scramble=np.stack(scr), actions=np.stack(acts),
               cols=np.stack(cols), done=np.stack(done), solve_step=np.array(solve, np.int32))
    np.savez_compressed(os.path.join(HERE, "crosscheck_222.npz"), **out)
    rate = {int(k): float((out["solve_step"][out["ks"] == k] > 0).mean()) for k in (1, 2, 3, 4, 6, 8)}
    print("solve rates", rate, "file bytes", os.path.getsize(os.path.join(HERE, "crosscheck_222.npz")))


if __name__ == "__main__":
    main()
