"""the only evidence available for the UNPINNED 2x2x2 convention.

The reference ships a trained 2x2x2 DeepCube checkpoint (pretrained/222model.pt) but not the
py222 module that defined its state encoding.  If our restated 2x2x2 tables / one-hot convention
match the authors', their policy net must solve shallow scrambles of OUR cubes greedily.

Runs only where /root/reference exists (the build container); the checkpoint never travels.
The pickle is NOT executed: tensor names, storage keys and shapes are read from its opcodes with
pickletools, the float32 payloads straight from the zip members."""
import os
import pickletools
import zipfile

import numpy as np
import pytest

CKPT = "/root/reference/pretrained/222model.pt"
pytestmark = pytest.mark.skipif(not os.path.exists(CKPT), reason="reference checkpoint not present on this box")


def read_state_dict_statically(path):
    z = zipfile.ZipFile(path)
    pkl = [n for n in z.namelist() if n.endswith("data.pkl")][0]
    prefix = pkl[: -len("data.pkl")]
    ops = list(pickletools.genops(z.read(pkl)))
    out, name, i = {}, None, 0
    while i < len(ops):
        op, arg, _ = ops[i]
        if op.name == "BINUNICODE" and arg == "optimizer_state_dict":
            break
        if op.name == "BINUNICODE" and "_net." in str(arg):
            name = arg
        if op.name == "BINPERSID" and name is not None:
            j = i - 1                                   # storage key = last all-digit string before the persid
            while not (ops[j][0].name == "BINUNICODE" and str(ops[j][1]).isdigit()):
                j -= 1
            key = ops[j][1]
            ints, k = [], i + 1                         # after persid: offset, then the shape tuple
            while ops[k][0].name.startswith("BININT"):
                ints.append(ops[k][1])
                k += 1
            assert ops[k][0].name.startswith("TUPLE")
            offset, shape = ints[0], tuple(ints[1:])
            raw = np.frombuffer(z.read(f"{prefix}data/{key}"), dtype="<f4")
            out[name] = raw[offset: offset + int(np.prod(shape))].reshape(shape).copy()
            name = None
        i += 1
    return out


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def policy_logits(sd, x):
    """model.py:13-45: Flatten -> Linear/ELU x2 -> policy head Linear/ELU/Linear."""
    h = elu(x.reshape(len(x), -1) @ sd["encoder_net.1.weight"].T + sd["encoder_net.1.bias"])
    h = elu(h @ sd["encoder_net.3.weight"].T + sd["encoder_net.3.bias"])
    p = elu(h @ sd["policy_net.0.weight"].T + sd["policy_net.0.bias"])
    return p @ sd["policy_net.2.weight"].T + sd["policy_net.2.bias"]


def test_checkpoint_policy_solves_our_222_cubes():
    from oracle.oracle_np import OracleCubeEnv
    sd = read_state_dict_statically(CKPT)
    assert sd["encoder_net.1.weight"].shape == (512, 147) and sd["policy_net.2.weight"].shape == (6, 64)
    env = OracleCubeEnv(None, 2)
    rates = {}
    for k in (1, 2, 3, 4, 6, 8):
        solved = 0
        for seed in range(40):
            state, done = env.reset(seed=seed, scramble_count=k), False
            for _ in range(30):
                a = int(np.argmax(policy_logits(sd, state[None].astype(np.float32))[0]))
                state, _, done, _ = env.step(a)
                if done:
                    break
            solved += done
        rates[k] = solved / 40
    print("2x2x2 greedy solve rate of the shipped checkpoint on the restated env:", rates)
    # a convention mismatch leaves the net blind: solve rates would sit near chance even at depth 1-2
    assert rates[1] >= 0.9 and rates[2] >= 0.9 and rates[4] >= 0.8


def _fixture():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "crosscheck_222.npz"))


def test_fixture_matches_live_checkpoint_run():
    """tests/golden/crosscheck_222.npz (actions and outcomes of the run above, no weights) is what the checkpoint does
    on the restated env today: regenerate a slice and compare."""
    from oracle.oracle_np import OracleCubeEnv
    g = _fixture()
    sd = read_state_dict_statically(CKPT)
    env = OracleCubeEnv(None, 2)
    for i in range(0, len(g["seeds"]), 7):
        state = env.reset(seed=int(g["seeds"][i]), scramble_count=int(g["ks"][i]))
        for t in range(int(g["solve_step"][i])):
            a = int(np.argmax(policy_logits(sd, state[None].astype(np.float32))[0]))
            assert a == g["actions"][i, t]
            state, _, d, _ = env.step(a)
            assert (np.argmax(state, 1) == g["cols"][i, t]).all() and d == g["done"][i, t]


def test_negative_controls_regenerate():
    """The control conventions of the fixture are re-run here on a slice (greedy, depths 2 and 6, 20 seeds): the live rates
    equal what the fixture recorded for those seeds' population within sampling, and every control stays far below the
    shipped convention."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_crosscheck_222 as mk
    sd = read_state_dict_statically(CKPT)
    live = {}
    for conv in mk.CONVENTIONS:
        env = mk.ConventionEnv(conv)
        live[conv] = [np.mean([mk.greedy(sd, env, seed, k)[3] > 0 for seed in range(20)]) for k in (2, 6)]
    assert live["shipped"] == [1.0, 1.0]
    for conv in mk.CONVENTIONS[1:]:
        assert live[conv][0] <= 0.5 and live[conv][1] <= 0.15, (conv, live[conv])
