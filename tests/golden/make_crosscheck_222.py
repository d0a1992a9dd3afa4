#!/usr/bin/env python3
"""Fixture for SURVEY.md section 8f N4: what the reference's shipped 2x2x2 checkpoint (pretrained/222model.pt) does on the
restated 2x2x2 env -- and, as NEGATIVE CONTROLS, on four plausible but different one-hot conventions.

The reference imports assets.py222 (cube_env.py:8) but does not ship it, so the 2x2x2 state encoding is unpinned; the
checkpoint is the only artefact that encodes the authors' convention.  If the restated convention is theirs, their net
solves our cubes; if a wrong convention solved them just as well the evidence would be worthless.  Conventions run here
(all on the same sticker arithmetic, only sim_state_to_state differs, cube_env.py:142-147):

    shipped          state[cubelet][position*3 + ori] = 1, ori = number of right rotations of the colour triple  (the product)
    ori_left         ori counted as LEFT rotations                      ((3 - ori) % 3)
    ori_shifted      orientation measured against the next sticker      ((ori + 1) % 3)
    pieces_reversed  cubelets and positions numbered in the opposite order (6 - index)
    not_transposed   the 3x3x3 file's convention: row = position, column = cubelet*3 + ori (py333.py:239-241)

Two solvers, as test.py:103-158 runs them: the greedy arg-max policy (model.py:47-76, no mask, 30 steps) for seeds 0..39, and
the REFERENCE'S OWN MCTS class (mcts.py, imported unmodified; config.yaml:29-32: 50 simulations, cpuct 1, virtual loss 150,
value_min -10) for seeds 0..19, on reset(seed, k) scrambles of depth k in DEPTHS.

Runs only in the build container (needs /root/reference); writes tests/golden/crosscheck_222.npz holding ACTIONS, OUTCOMES AND
RATES ONLY -- no weights, no reference source:
    seeds, ks, scramble [N, 14] (no-op 6 padded), actions [N, T], cols [N, T, 7], done [N, T], solve_step [N]      greedy, shipped
    mcts_seeds, mcts_ks, mcts_scramble [M, 14], mcts_found [M], mcts_sims [M], mcts_solution [M, 56] (6 padded),
    mcts_cols [M, 56, 7]                                                                                         MCTS, shipped
    conventions [5], depths [9], greedy_rate [5, 9], mcts_rate [5, 9]                                            every convention
The checkpoint is read statically (tests/test_crosscheck_222.py: pickle opcodes + raw float32 zip members, nothing is
unpickled or executed).

    python tests/golden/make_crosscheck_222.py
"""
import os
import random
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from test_crosscheck_222 import CKPT, elu, policy_logits, read_state_dict_statically  # noqa: E402
from oracle.oracle_np import OracleCubeEnv  # noqa: E402

T = 30
DEPTHS = (1, 2, 3, 4, 6, 8, 10, 12, 14)
KMAX = 14
LSOL = 56          # longest action list MCTS.train can return within 50 simulations (one tree level per simulation at most)
CONVENTIONS = ("shipped", "ori_left", "ori_shifted", "pieces_reversed", "not_transposed")
GREEDY_SEEDS, MCTS_SEEDS = range(40), range(20)
MCTS_CFG = {"mcts": {"virtual_loss_const": 150, "cpuct": 1.0, "value_min": -10.0, "numMCTSSim": 50}, "test": {"cube_size": 2}}


class ConventionEnv(OracleCubeEnv):
    """The restated 2x2x2 env with a selectable one-hot convention (the sticker arithmetic is shared)."""

    def __init__(self, convention):
        self.convention = convention
        super().__init__(None, 2)

    def sim_state_to_state(self, s):
        state = np.zeros(self.state_dim)
        for slot, (piece, ori) in enumerate(self._get_op(s)):
            c = self.convention
            if c == "ori_left":
                ori = (3 - ori) % 3
            elif c == "ori_shifted":
                ori = (ori + 1) % 3
            elif c == "pieces_reversed":
                piece, slot = 6 - piece, 6 - slot
            if c == "not_transposed":
                state[slot][piece * 3 + ori] = 1.0
            else:
                state[piece][slot * 3 + ori] = 1.0                    # cube_env.py:143-147
        return state


class StaticNet:
    """model.py:31-45 / :78-91 on the statically read float32 weights: predict(state) -> (value [1], softmax policy [6])."""

    def __init__(self, sd):
        self.sd = sd

    def predict(self, x):
        sd = self.sd
        x = np.asarray(x, np.float32).reshape(1, -1)
        h = elu(x @ sd["encoder_net.1.weight"].T + sd["encoder_net.1.bias"])
        h = elu(h @ sd["encoder_net.3.weight"].T + sd["encoder_net.3.bias"])
        v = elu(h @ sd["value_net.0.weight"].T + sd["value_net.0.bias"]) @ sd["value_net.2.weight"].T + sd["value_net.2.bias"]
        p = elu(h @ sd["policy_net.0.weight"].T + sd["policy_net.0.bias"]) @ sd["policy_net.2.weight"].T + sd["policy_net.2.bias"]
        e = np.exp(p[0] - p[0].max())
        return v[0].astype(np.float32), (e / e.sum()).astype(np.float32)


def import_reference_mcts():
    gym = types.ModuleType("gym")
    gym.Env = type("Env", (), {})
    sys.modules.setdefault("gym", gym)
    np.int = int
    sys.path[:0] = [REF]
    import mcts as ref_mcts                                             # the reference's file, unmodified

    return ref_mcts


def draws(seed, k):
    saved = np.random.get_state()
    np.random.seed(seed)
    d = np.random.randint(6, size=k)                                    # what reset(seed, k) draws (cube_env.py:64-65)
    np.random.set_state(saved)
    return np.concatenate([d, np.full(KMAX - k, 6)]).astype(np.uint8)


def greedy(sd, env, seed, k):
    state = env.reset(seed=seed, scramble_count=k)
    a_row, c_row, d_row, s_at = np.full(T, 6, np.uint8), np.zeros((T, 7), np.uint8), np.zeros(T, np.uint8), 0
    for t in range(T):
        a = int(np.argmax(policy_logits(sd, state[None].astype(np.float32))[0]))
        state, _, d, _ = env.step(a)
        a_row[t], c_row[t], d_row[t] = a, np.argmax(state, 1), d
        if d:
            s_at = t + 1
            c_row[t + 1:], d_row[t + 1:] = c_row[t], 1                 # parked with the no-op: state and flag stay
            break
    return a_row, c_row, d_row, s_at


def mcts_trial(ref_mcts, net, env, seed, k):
    """test.py:120-142 with mcts_=True: numMCTSSim calls of MCTS.train on the scrambled root."""
    state = env.reset(seed=seed, scramble_count=k)
    random.seed(7000 + 31 * seed + k)
    tree = ref_mcts.MCTS(net, MCTS_CFG)
    for sim in range(1, MCTS_CFG["mcts"]["numMCTSSim"] + 1):
        found = tree.train(state, env)
        if found is not None:
            return list(found), sim
    return None, MCTS_CFG["mcts"]["numMCTSSim"]


def main():
    sd = read_state_dict_statically(CKPT)
    ref_mcts = import_reference_mcts()
    net = StaticNet(sd)
    out = {}
    g_rate, m_rate = np.zeros((len(CONVENTIONS), len(DEPTHS))), np.zeros((len(CONVENTIONS), len(DEPTHS)))
    seeds, ks, scr, acts, cols, done, solve = [], [], [], [], [], [], []
    m_seeds, m_ks, m_scr, m_found, m_sims, m_sol, m_cols = [], [], [], [], [], [], []
    for ci, conv in enumerate(CONVENTIONS):
        env = ConventionEnv(conv)
        for di, k in enumerate(DEPTHS):
            ok = 0
            for seed in GREEDY_SEEDS:
                a_row, c_row, d_row, s_at = greedy(sd, env, seed, k)
                ok += s_at > 0
                if conv == "shipped":
                    seeds.append(seed); ks.append(k); scr.append(draws(seed, k))
                    acts.append(a_row); cols.append(c_row); done.append(d_row); solve.append(s_at)
            g_rate[ci, di] = ok / len(GREEDY_SEEDS)
            ok = 0
            for seed in MCTS_SEEDS:
                found, sims = mcts_trial(ref_mcts, net, env, seed, k)
                ok += found is not None
                if conv == "shipped":
                    sol, pc = np.full(LSOL, 6, np.uint8), np.zeros((LSOL, 7), np.uint8)
                    if found is not None:
                        assert len(found) <= LSOL
                        env.reset(seed=seed, scramble_count=k)
                        for t, a in enumerate(found):
                            st, _, d, _ = env.step(int(a))
                            sol[t], pc[t] = a, np.argmax(st, 1)
                        assert d                                        # the returned action list solves the cube
                        pc[len(found):] = pc[len(found) - 1]
                    m_seeds.append(seed); m_ks.append(k); m_scr.append(draws(seed, k)); m_found.append(found is not None)
                    m_sims.append(sims); m_sol.append(sol); m_cols.append(pc)
            m_rate[ci, di] = ok / len(MCTS_SEEDS)
        print(f"{conv:16s} greedy {np.round(g_rate[ci], 2)}  mcts {np.round(m_rate[ci], 2)}", flush=True)
    out.update(seeds=np.array(seeds, np.int64), ks=np.array(ks, np.int32), scramble=np.stack(scr), actions=np.stack(acts),
               cols=np.stack(cols), done=np.stack(done), solve_step=np.array(solve, np.int32),
               mcts_seeds=np.array(m_seeds, np.int64), mcts_ks=np.array(m_ks, np.int32), mcts_scramble=np.stack(m_scr),
               mcts_found=np.array(m_found, np.uint8), mcts_sims=np.array(m_sims, np.int32), mcts_solution=np.stack(m_sol),
               mcts_cols=np.stack(m_cols), conventions=np.array(CONVENTIONS), depths=np.array(DEPTHS, np.int32),
               greedy_rate=g_rate, mcts_rate=m_rate)
    np.savez_compressed(os.path.join(HERE, "crosscheck_222.npz"), **out)
    print("file bytes", os.path.getsize(os.path.join(HERE, "crosscheck_222.npz")))


if __name__ == "__main__":
    main()
