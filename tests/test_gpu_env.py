"""Env surface on the GPU: VecCubeEnv and the CubeEnv facade against the oracle env and the
reference's golden vectors (reset KATs, ADI samples with the stub model).  GPU only."""
import copy
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def stub_model(g, device="cpu"):
    w, b = torch.tensor(g["w"], device=device), torch.tensor(g["b"], device=device)

    def model(x):
        if x.dim() == 2:
            x = x.unsqueeze(0)
        return (x.reshape(x.shape[0], -1) @ w + b).unsqueeze(-1), torch.zeros(x.shape[0], 12, device=x.device)
    return model


@pytest.fixture(scope="module")
def mod():
    import rubiks_cube_solver_amd as r
    return r


def test_make_env_facade_matches_oracle_env(mod, golden):
    from oracle.oracle_np import OracleCubeEnv
    env = mod.make_env(torch.device("cpu"), 3)
    ref = OracleCubeEnv(None, 3)
    assert env.state_dim == ref.state_dim and env.action_dim == ref.action_dim
    assert (env.sim_cube == ref.sim_cube).all() and (env.cube == ref.cube).all() and env.cube.dtype == ref.cube.dtype
    rng = np.random.default_rng(0)
    for _ in range(60):
        a = int(rng.integers(0, 12))
        s, r, d, info = env.step(a)
        s2, r2, d2, _ = ref.step(a)
        assert (s == s2).all() and s.dtype == s2.dtype and r == r2 and d == d2 and info == {}
        assert (env.sim_cube == ref.sim_cube).all()
    g = golden("reset_333")
    for i, seed in enumerate(g["seeds"][:4]):
        for j in (0, 6, 29):
            s = env.reset(seed=int(seed), scramble_count=int(g["ks"][j]))
            assert (env.sim_cube == g["stickers"][i, j]).all() and (np.argmax(s, 1) == g["cols"][i, j]).all()
    arb = golden("encode_333")
    for k in range(5):
        oh = env.sim_state_to_state(arb["stickers"][k])
        assert (np.argmax(oh, 1) == arb["cols"][k]).all() and oh.dtype == np.int64
    with pytest.raises(IndexError):
        env.step(12)
    with pytest.raises(UnboundLocalError):
        env.reset(seed=0, scramble_count=0)


def test_facade_222_and_deepcopy(mod):
    from oracle.oracle_np import OracleCubeEnv
    env, ref = mod.make_env(torch.device("cpu"), 2), OracleCubeEnv(None, 2)
    for a in (0, 3, 4, 1, 2, 5, 5, 0):
        s, r, d, _ = env.step(a)
        s2, r2, d2, _ = ref.step(a)
        assert (s == s2).all() and s.dtype == np.float64 and (r, d) == (r2, d2)
    assert (env.reset(seed=4, scramble_count=20) == ref.reset(seed=4, scramble_count=20)).all()
    snap = env.sim_cube.copy()
    other = copy.deepcopy(env)
    env.step(1)
    assert (other.sim_cube == snap).all() and not (env.sim_cube == snap).all()
    other.step(1)
    assert (other.sim_cube == env.sim_cube).all()
    assert (env.state_to_sim_state(env.cube) == env.sim_cube).all()


def test_get_target_value_matches_reference_golden(mod, golden):
    g = golden("adi_333")
    model = stub_model(g)
    env = mod.make_env(torch.device("cpu"), 3)
    T = float(g["temperature"])
    for c in range(6):
        env.init_state()
        for d in range(30):
            env.step(int(g["actions"][c, d]))
            tv, tp, err = env.get_target_value(model, d + 1, T)
            assert tp == g["target_policy"][c, d]
            assert tv == pytest.approx(g["target_value"][c, d], abs=1e-6)
            assert err == pytest.approx(g["error"][c, d], abs=1e-6)
            assert isinstance(tv, float) and isinstance(tp, int)


def test_get_random_samples_matches_reference_golden(mod, golden):
    """Same global numpy seed as the golden run -> the same samples, in the same order."""
    g = golden("adi_333")
    n = g["actions"].shape[0]
    env = mod.make_env(torch.device("cpu"), 3)
    buf = []
    np.random.seed(int(g["seed"]))
    env.get_random_samples(buf, stub_model(g), 30, n, float(g["temperature"]))
    assert len(buf) == n * 30
    after = np.random.get_state()[1].copy()
    np.random.seed(int(g["seed"]))
    for _ in range(n):
        np.random.randint(12, size=30)
    assert (np.random.get_state()[1] == after).all()             # consumed exactly the reference's draws
    # child values of the stub model, from the oracle's child codes: a policy mismatch is only acceptable on an arg-max TIE
    from oracle.oracle_np import Oracle
    out = Oracle().adi(3, n, 30, actions_in=g["actions"], want_children=False)
    wv = g["w"].reshape(20, 24).astype(np.float64)
    v = np.sort(wv[np.arange(20), out["child_code"].astype(np.int64)].sum(-1), -1)                 # [n, 30, 12] ascending
    gap = v[..., -1] - v[..., -2]
    for i, smp in enumerate(buf):
        c, d = divmod(i, 30)
        assert set(smp) == {"state", "target_value", "target_policy", "scramble_count", "error"}
        assert smp["state"].shape == (20, 24) and smp["state"].dtype == np.int64
        assert (np.argmax(smp["state"], 1) == g["cols"][c, d]).all() and (smp["state"].sum(1) == 1).all()
        assert smp["scramble_count"] == d + 1
        assert smp["target_value"] == pytest.approx(g["target_value"][c, d], abs=1e-5)
        assert smp["error"] == pytest.approx(g["error"][c, d], abs=1e-5)
        if smp["target_policy"] != g["target_policy"][c, d]:
            assert gap[c, d] < 1e-5 and not out["child_solved"][c, d].any(), (c, d, gap[c, d])   # mismatch => top-2 tie
    assert all(b["target_value"] == 1.0 for b in buf[::30])      # depth 1: the inverse move solves
    assert (env.sim_cube == _final_state(g)).all()


def _final_state(g):
    from oracle.oracle_np import OracleCubeEnv
    ref = OracleCubeEnv(None, 3)
    for a in g["actions"][-1]:
        ref.step(int(a))
    return ref.sim_cube


def test_adi_samples_device_rng_and_gpu_model(mod, oracle, golden):
    from rubiks_cube_solver_amd.adi import adi_samples
    g = golden("adi_333")
    model = stub_model(g, "cuda")
    W, D = 700, 9
    res = adi_samples(model, 3, W, D, 0.3, device="cuda", seed=11, stream_id=2, dense_budget_bytes=13 * 1920 * 256,
                      want_state_dense=True)
    exp = oracle.adi(3, W, D, seed=11, stream=2, want_children=False)
    assert (res["actions"].cpu().numpy() == exp["actions"]).all()
    assert (res["state_code"].cpu().numpy() == exp["parent_code"]).all()
    assert (res["state"].argmax(-1).cpu().numpy() == exp["parent_code"]).all()
    w = g["w"].reshape(20, 24).astype(np.float64)
    v_child = w[np.arange(20), exp["child_code"].astype(np.int64)].sum(-1) + float(g["b"]) - 1.0   # [W, D, 12]
    solved = exp["child_solved"].astype(bool)
    tv = np.where(solved.any(-1), 1.0, v_child.max(-1))
    assert np.allclose(res["target_value"].cpu().numpy(), tv, atol=1e-5)
    tp = np.where(solved.any(-1), np.argmax(solved, -1), np.argmax(v_child, -1))
    got_tp = res["target_policy"].cpu().numpy()
    srt = np.sort(v_child, -1)
    tie = (srt[..., -1] - srt[..., -2] < 1e-5) & ~solved.any(-1)
    assert ((got_tp == tp) | tie).all()                                 # a mismatch is a top-2 tie within 1e-5, nothing else
    assert (got_tp == tp).mean() > 0.995
    v_par = w[np.arange(20), exp["parent_code"].astype(np.int64)].sum(-1) + float(g["b"])
    err = np.abs(v_par - tv) * np.arange(1, D + 1, dtype=np.float64)[None, :] ** -0.3
    assert np.allclose(res["error"].cpu().numpy(), err, atol=1e-5)
    assert (res["scramble_count"][0].cpu().numpy() == np.arange(1, D + 1)).all()


@pytest.mark.parametrize("cs", [3, 2])
@pytest.mark.parametrize("obs", ["onehot", "code", None])
def test_vec_env_vs_oracle(mod, oracle, cs, obs):
    n = 40000
    A = 12 if cs == 3 else 6
    env = mod.VecCubeEnv(n, "cuda", cs, obs=obs, onehot_dtype=torch.float16, seed=5, stream_id=1)
    assert env.stickers.shape[0] > 1                                                   # tiled buffer
    assert (env.sim_cube.cpu().numpy() == oracle.solved(cs, n)).all()
    o = env.reset(scramble_count=15)
    exp = oracle.adi(cs, n, 15, seed=5, stream=1, walk0=n, want_children=False)["parents"][:, -1]
    assert (env.sim_cube.cpu().numpy() == exp).all()
    acts = np.random.default_rng(1).integers(0, A, n, dtype=np.uint8)
    o, r, d, info = env.step(acts)
    e_st, e_code, e_done, e_rew = oracle.step(cs, exp, acts)
    assert (env.sim_cube.cpu().numpy() == e_st).all() and info == {}
    assert (r.cpu().numpy() == e_rew).all() and (d.cpu().numpy() == e_done).all()
    if obs == "onehot":
        assert o.dtype == torch.float16 and tuple(o.shape) == (n, *env.state_dim)
        _, e_oh = oracle.encode(cs, e_st)
        assert (o.cpu().numpy() == e_oh.astype(np.float16)).all()
    elif obs == "code":
        from rubiks_cube_solver_amd import ops
        assert (ops.to_aos(o, n).cpu().numpy() == e_code).all()
    else:
        assert o is None
    ex = env.expand(children=True)
    ch, cc, cs_ = oracle.expand(cs, e_st, threads=4)
    from rubiks_cube_solver_amd import ops as _ops
    assert (ex["child_solved"][:, :n].cpu().numpy().T == cs_).all()
    for a in range(A):
        assert (_ops.to_aos(ex["child_code"][a], n).cpu().numpy() == cc[:, a]).all()
        assert (_ops.to_aos(ex["children"][a], n).cpu().numpy() == ch[:, a]).all()
    with pytest.raises(IndexError):
        env.step(np.full(n, A))                                                        # host-converted actions are range-checked
    bad = torch.zeros(n, dtype=torch.uint8, device="cuda")
    bad[17] = A + 1                                                                    # (A itself is the no-op)
    env.step(bad)                                                                      # device path cannot raise...
    with pytest.raises(IndexError):
        env.check_actions()                                                            # ...the status word does
    clone = copy.deepcopy(env)
    env.init_state()
    assert bool(env.is_solved().all()) and not bool(clone.is_solved().all())


def test_vec_env_seeded_reset_is_reference_reset(mod, golden):
    g = golden("reset_333")
    seeds = [int(s) for s in g["seeds"]]
    env = mod.VecCubeEnv(len(seeds), "cuda", 3, obs="code")
    np.random.seed(99)
    before = np.random.get_state()[1].copy()
    for j in (0, 13, 29):
        env.reset(seeds=seeds, scramble_count=int(g["ks"][j]))
        assert (env.sim_cube.cpu().numpy() == g["stickers"][:, j]).all()
    assert (np.random.get_state()[1] == before).all()
    with pytest.raises(UnboundLocalError):
        env.reset(scramble_count=0)
    env.reset(seeds=torch.tensor(seeds, device="cuda"), scramble_count=30)       # seeds may already live on the device
    assert (env.sim_cube.cpu().numpy() == g["stickers"][:, 29]).all()


# ------------------------------------------------------------------ no-op, rollouts (N3), MCTS (N2)
class TinyNet(torch.nn.Module):
    """DeepCube-shaped stand-in (model.py:7-45) with float64 maths so CPU and GPU agree on every arg-max."""

    def __init__(self, state_dim, action_dim, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        d = state_dim[0] * state_dim[1]
        self.w1 = torch.nn.Parameter(torch.randn(d, 64, generator=g, dtype=torch.float64))
        self.wp = torch.nn.Parameter(torch.randn(64, action_dim, generator=g, dtype=torch.float64))
        self.wv = torch.nn.Parameter(torch.randn(64, 1, generator=g, dtype=torch.float64))

    def forward(self, x):
        if x.dim() == 2:
            x = x.unsqueeze(0)
        h = torch.nn.functional.elu(x.reshape(x.shape[0], -1).double() @ self.w1)
        return h @ self.wv, h @ self.wp

    def get_action(self, x, pre_action=None):          # model.py:47-76 restated for the per-cube reference loop
        order = self.forward(x)[1].sort(descending=True)[1][0]
        invalid = None if pre_action is None else (pre_action - 1 if pre_action % 2 == 1 else pre_action + 1)
        return order[1].item() if invalid == order[0] else order[0].item()

    def predict(self, x):                               # model.py:78-91
        v, p = self.forward(torch.tensor(np.asarray(x)).float())
        return v.detach().cpu().numpy()[0], torch.softmax(p, -1).detach().cpu().numpy()[0]


def test_noop_action_and_active_mask(mod, oracle):
    for cs, A in ((3, 12), (2, 6)):
        n = 3000
        env = mod.VecCubeEnv(n, "cuda", cs, obs=None, seed=3)
        env.reset(scramble_count=9)
        before = env.sim_cube.cpu().numpy()
        acts = np.random.default_rng(2).integers(0, A, n, dtype=np.uint8)
        active = torch.from_numpy(np.arange(n) % 3 != 0).cuda()
        env.step(torch.from_numpy(acts).cuda(), active=active)
        exp = oracle.step(cs, before, acts)[0]
        keep = (np.arange(n) % 3 == 0)
        exp[keep] = before[keep]
        assert (env.sim_cube.cpu().numpy() == exp).all()
        env.check_actions()                                  # the no-op is not an error
        env.step(torch.full((n,), A, dtype=torch.uint8, device="cuda"))
        assert (env.sim_cube.cpu().numpy() == exp).all()
        env.check_actions()
        env.step(torch.full((n,), A + 1, dtype=torch.uint8, device="cuda"))
        with pytest.raises(IndexError):
            env.check_actions()


@pytest.mark.parametrize("mask", [False, True])
def test_greedy_rollout_matches_per_cube_loop(mod, mask):
    from oracle.oracle_np import OracleCubeEnv
    from rubiks_cube_solver_amd.rollout import greedy_rollout
    net = TinyNet([20, 24], 12, seed=4)
    seeds, ks, T = list(range(360)), [1 + (i % 3) for i in range(360)], 14
    env = mod.VecCubeEnv(len(seeds), "cuda", 3, obs="onehot")
    env.reset(seeds=seeds, scramble_count=ks)
    res = greedy_rollout(net.cuda(), env, T, mask=mask, sync_every=1)
    steps = res["solve_step"].cpu().numpy()
    net.cpu()
    ref = OracleCubeEnv(None, 3)
    n_solved = 0
    for i, (s, k) in enumerate(zip(seeds, ks)):
        state, pre, solved_at = ref.reset(seed=s, scramble_count=k), None, 0
        for t in range(1, T + 1):                            # test.py:126-151 / train.py:183-193
            with torch.no_grad():
                a = net.get_action(torch.tensor(state).float(), pre if mask else None)
            if mask:
                pre = a
            state, _, done, _ = ref.step(a)
            if done:
                solved_at = t
                break
        assert steps[i] == solved_at, (i, steps[i], solved_at)
        n_solved += solved_at > 0
    assert n_solved >= 2                                      # the parking path was exercised
    assert (env.is_solved().cpu().numpy().astype(bool) == (steps > 0)).all()


def test_solve_percentage_shape(mod):
    from rubiks_cube_solver_amd.rollout import solve_percentage
    net = TinyNet([7, 21], 6, seed=1).cuda()
    pct = solve_percentage(net, 2, 4, 25, 12, device="cuda")
    assert len(pct) == 4 and all(0.0 <= p <= 100.0 for p in pct)
    assert solve_percentage(net, 2, 4, 25, 12, device="cuda", graph=True) == pct       # one time step replayed as a hipGraph: the same rollouts


def test_mcts_class_and_batched_mcts(mod, oracle):
    from rubiks_cube_solver_amd.mcts_batched import MCTS, BatchedMCTS
    cfg = {"mcts": {"virtual_loss_const": 150, "cpuct": 1.0, "value_min": -10.0, "numMCTSSim": 50}, "test": {"cube_size": 3}}
    net = TinyNet([20, 24], 12, seed=2)
    env = mod.make_env(torch.device("cpu"), 3)
    state = env.reset(seed=3, scramble_count=1)
    tree = MCTS(net, cfg)
    actions = tree.train(state, env)                          # depth-1 scramble: the first expansion sees the solved child
    assert actions is not None and len(actions) == 1
    chk = oracle.step(3, env.sim_cube[None].astype(np.uint8), np.array(actions[-1:], np.uint8))
    assert chk[2][0] == 1
    root = tree.children_and_data[MCTS.key_of(env)]
    _, cc, cs = oracle.expand(3, env.sim_cube[None].astype(np.uint8))
    assert [c.tobytes() for c in cc[0]] == root.children and list(cs[0].astype(bool)) == root.done
    state = env.reset(seed=8, scramble_count=3)
    tree, found = MCTS(net, cfg), None
    for _ in range(400):
        found = tree.train(state, env)
        if found is not None:
            break
    root_key = tree.key_of_state(state)
    if root_key in tree.children_and_data and sum(tree.children_and_data[root_key].visits):
        node = tree.children_and_data[root_key]                   # mcts.py:132-154 under its reference name
        tot = np.sqrt(sum(node.visits))
        score = [np.float32(1.0) * node.policy[i] * np.float32(tot / (1 + node.visits[i])) + np.float32(node.value[i]) - np.float32(node.vloss[i]) for i in range(12)]
        assert tree.get_most_promising_action_index(root_key) == int(np.argmax(score))
    if found is not None:                                     # replay the returned path: it must solve the cube
        s = env.sim_cube[None].astype(np.uint8)
        for a in found:
            s, _, d, _ = oracle.step(3, s, np.array([a], np.uint8))
        assert d[0] == 1
    # lockstep search over many roots
    n = 256
    venv = mod.VecCubeEnv(n, "cuda", 3, obs=None, seed=21)
    venv.reset(seeds=list(range(n)), scramble_count=[1 + (i % 2) for i in range(n)])
    roots = venv.sim_cube.cpu().numpy()
    bm = BatchedMCTS(net.cuda(), venv.stickers, n, 3)
    solved = 0
    for _ in range(20):
        solved = bm.simulate()
        if solved == n:
            break
    # every depth-1 root is solved by its first expansion; deeper ones depend on the (random) net's guidance
    assert all(bm.solution[r] is not None and len(bm.solution[r]) == 1 for r in range(0, n, 2)) and solved >= n // 2
    for r in range(n):
        if bm.solution[r] is not None:
            s = roots[r:r + 1]
            for a in bm.solution[r]:
                s, _, d, _ = oracle.step(3, s, np.array([a], np.uint8))
            assert d[0] == 1, r


def test_mcts_matches_reference_golden(mod, golden):
    """G8: the reference's own MCTS (mcts.py) vs mcts_batched.MCTS on the same scrambles, the same stub model and
    the same seeded `random`: simulations needed, returned action list and the root's visit counts / values."""
    import random

    from rubiks_cube_solver_amd.mcts_batched import MCTS
    g = golden("mcts_333")
    wv, wp = g["wv"], g["wp"]

    class Stub:
        def predict(self, x):
            f = np.asarray(x, dtype=np.float32).reshape(-1)
            logits = f @ wp
            e = np.exp(logits - logits.max())
            return np.array([f @ wv], np.float32), (e / e.sum()).astype(np.float32)

    cfg = {"mcts": {"virtual_loss_const": 150, "cpuct": 1.0, "value_min": -10.0, "numMCTSSim": 50}, "test": {"cube_size": 3}}
    env = mod.make_env(torch.device("cpu"), 3)
    for i, (seed, k) in enumerate(zip(g["seeds"], g["ks"])):
        state = env.reset(seed=int(seed), scramble_count=int(k))
        random.seed(int(g["random_seed"][i]))
        tree, found, used = MCTS(Stub(), cfg), None, 0
        for s in range(60):
            used = s + 1
            found = tree.train(state, env)
            if found is not None:
                break
        assert used == int(g["sims"][i]), (i, used, int(g["sims"][i]))
        exp = [int(a) for a in g["solution"][i] if a != 255]
        assert (found or []) == exp
        root = tree.children_and_data[MCTS.key_of(env)]
        assert root.visits == g["root_visits"][i].tolist()
        assert np.allclose(root.value, g["root_values"][i], rtol=0, atol=1e-6)
    venv = mod.VecCubeEnv(len(g["long_seeds"]), "cuda", 3, obs=None)       # reset(seed, 1000): test.py:279 style scrambles
    venv.reset(seeds=[int(s) for s in g["long_seeds"]], scramble_count=int(g["long_k"]))
    assert (venv.sim_cube.cpu().numpy() == g["long_stickers"]).all()


@pytest.mark.parametrize("graph", [False, True])
def test_greedy_rollout_matches_reference_golden(mod, golden, graph):
    """G9: the reference's DeepCube (weights from the fixture) driving greedy_rollout on the GPU reproduces the
    reference's per-cube solve loops: every action taken and the step at which each cube was solved."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from bench_cfg5 import DeepCubeStandIn          # same layer layout as model.py:7-45 -> loads the reference state_dict
    from rubiks_cube_solver_amd.rollout import greedy_rollout
    g = golden("rollout_333")
    net = DeepCubeStandIn((20, 24), 12, (64, 32, 16))
    net.load_state_dict({k[3:]: torch.tensor(g[k]) for k in g.files if k.startswith("sd_")})
    net = net.cuda().eval()
    ks, n_seeds, T = [int(k) for k in g["ks"]], int(g["n_seeds"]), int(g["T"])
    seeds = [j * 10 for j in range(n_seeds)] * len(ks)
    counts = [k for k in ks for _ in range(n_seeds)]
    for mask, a_key, s_key in ((False, "actions", "solved_at"), (True, "actions_mask", "solved_at_mask")):
        env = mod.VecCubeEnv(len(seeds), "cuda", 3, obs="onehot")
        env.reset(seeds=seeds, scramble_count=counts)
        res = greedy_rollout(net, env, T, mask=mask, sync_every=T, graph=graph)
        steps = res["solve_step"].cpu().numpy().reshape(len(ks), n_seeds)
        assert (steps == g[s_key]).all()
        acts = res["actions"].cpu().numpy().T.reshape(len(ks), n_seeds, -1)          # [k, seed, t]
        exp = g[a_key]
        took = exp != 255
        assert (acts[..., :exp.shape[-1]][took] == exp[took]).all()
        assert (acts[..., :exp.shape[-1]][~took] == 12).all()                         # parked with the no-op afterwards
        assert int((steps > 0).sum()) == 14


@pytest.mark.parametrize("native", [True, False])
@pytest.mark.parametrize("graph", [False, True])
def test_batched_mcts_reproduces_reference_runs_in_lockstep(mod, golden, graph, native):
    """All 24 scrambles of fixture G8 searched TOGETHER by BatchedMCTS (one replay + one expansion launch + one net
    forward per simulation), each root with its own seeded generator: every root ends like the reference's
    stand-alone MCTS run (simulations used, action list, root visit counts, root values)."""
    import random

    from rubiks_cube_solver_amd.mcts_batched import BatchedMCTS
    g = golden("mcts_333")
    wv, wp = torch.tensor(g["wv"]).cuda(), torch.tensor(g["wp"]).cuda()

    def model(x):                                            # the fixture's stub as a batched torch callable
        f = x.reshape(x.shape[0], -1).float()
        return (f @ wv).unsqueeze(-1), f @ wp

    n = len(g["seeds"])
    venv = mod.VecCubeEnv(n, "cuda", 3, obs=None)
    venv.reset(seeds=[int(s) for s in g["seeds"]], scramble_count=[int(k) for k in g["ks"]])
    bm = BatchedMCTS(model, venv.stickers, n, 3, rngs=[random.Random(int(s)) for s in g["random_seed"]], graph=graph, native=native)   # librubiktree.so | the Python tree
    for _ in range(60):
        bm.simulate()
    for r in range(n):
        assert bm.sims_used[r] == int(g["sims"][r]), r
        exp = [int(a) for a in g["solution"][r] if a != 255]
        assert (bm.solution[r] or []) == exp
        root = bm.trees[r][b"root"]
        assert root.visits == g["root_visits"][r].tolist(), r
        assert np.allclose(root.value, g["root_values"][r], rtol=0, atol=1e-5)


def test_get_random_samples_with_reference_deepcube(mod, golden):
    """G10: same global numpy seed, the reference's DeepCube weights -> the reference's replay-buffer records
    (values to 1e-5: the net runs batched on 13 x walks states per depth instead of 12 + 1 per sample)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from bench_cfg5 import DeepCubeStandIn
    g = golden("adi_deepcube_333")
    net = DeepCubeStandIn((20, 24), 12, (64, 32, 16))
    net.load_state_dict({k[3:]: torch.tensor(g[k]) for k in g.files if k.startswith("sd_")})
    net.eval()
    n, depth = g["target_value"].shape
    for device in ("cpu", "cuda"):                           # model on the host (the reference's config) and on the GPU
        env = mod.make_env(torch.device(device), 3)
        buf = []
        np.random.seed(int(g["seed"]))
        env.get_random_samples(buf, net.to(device), depth, n, float(g["temperature"]))
        assert len(buf) == n * depth
        for i, smp in enumerate(buf):
            c, d = divmod(i, depth)
            assert (np.argmax(smp["state"], 1) == g["cols"][c, d]).all()
            assert smp["target_value"] == pytest.approx(g["target_value"][c, d], abs=1e-5)
            assert smp["error"] == pytest.approx(g["error"][c, d], abs=1e-5)
            if smp["target_policy"] != g["target_policy"][c, d]:       # only on a top-2 tie of the child values (< 1e-5)
                from oracle.oracle_np import Oracle
                orc = Oracle()
                st = orc.solved(3, 1)
                np.random.seed(int(g["seed"]))
                walks = [np.random.randint(12, size=depth) for _ in range(n)]
                for a in walks[c][:d + 1]:
                    st = orc.step(3, st, np.array([a], np.uint8))[0]
                ch, cc, cso = orc.expand(3, st)
                oh = torch.nn.functional.one_hot(torch.from_numpy(cc[0].astype(np.int64)), 24).float()
                with torch.no_grad():
                    vals = np.sort(net.cpu()(oh)[0].reshape(-1).double().numpy())
                assert vals[-1] - vals[-2] < 1e-5 and not cso.any(), (c, d)


def test_reference_named_operators(mod, golden):
    """py333-named operators (one cube per call) on the reference's vectors; reads like the reference's call sites."""
    from rubiks_cube_solver_amd import py333 as P
    g = golden("walks_333")
    s = P.initState_3()
    assert s.dtype == np.int64 and s.tolist() == sum([[c] * 9 for c in range(6)], []) and P.isSolved_3(s)
    names = list(g["action_names"]) if "action_names" in g.files else ["U", "U'", "F", "F'", "R", "R'", "D", "D'", "B", "B'", "L", "L'"]
    for d in range(30):
        s2 = P.doMove_3(s, names[int(g["actions"][7, d])])
        assert s2 is not s and (s2 == g["stickers"][7, d]).all()
        s = s2
        op = P.getOP_3(s)
        assert op.shape == (20, 2)
        oh = P.pos_to_state_3(op)
        assert oh.dtype == np.int64 and (np.argmax(oh, 1) == g["cols"][7, d]).all() and (oh.sum(1) == 1).all()
        assert P.isSolved_3(s) == bool(g["done"][7, d])
    with pytest.raises(KeyError):
        P.doMove_3(s, "X")
    t = golden("tables_333")
    for k in range(6):                                        # getOP_3 == the reference LUT rows on arbitrary colourings
        st = golden("encode_333")["stickers"][k].astype(np.int64)
        ref_c = t["corner_pieceInds"][st[t["corner_pieceDefs"].astype(int)] @ np.array([1, 2, 10])]
        ref_e = t["edge_pieceInds"][st[t["edge_pieceDefs"].astype(int)] @ np.array([1, 10])]
        assert (P.getOP_3(st) == np.concatenate([ref_c, ref_e])).all()
    s = P.initState()
    for m in ("R", "U", "R'", "U'") * 6:
        s = P.doMove(s, m)
    assert P.isSolved(s) and (P.getStickers(P.getOP(P.doMove(s, "F"))) == P.doMove(s, "F")).all()
    with pytest.raises(KeyError):
        P.doMove(s, "D")                                       # the 2x2x2 env only turns U, F, R (cube_env.py:25)


@pytest.mark.parametrize("cs", [3, 2])
def test_facade_step_and_expand_entry_points(mod, oracle, cs):
    """rc_facade_step / rc_facade_expand (the batch-1 latency path: results land in pinned host memory, no copies, no
    stream sync) against the oracle over a long random walk, incl. solved hits; keys from the one-hot state equal the
    device code (cube_env.py:71-111, mcts.py:66,83-113)."""
    from rubiks_cube_solver_amd.mcts_batched import MCTS
    A = 12 if cs == 3 else 6
    env = mod.make_env(torch.device("cpu"), cs)
    cfg = {"mcts": {"virtual_loss_const": 150, "cpuct": 1.0, "value_min": -10.0}, "test": {"cube_size": cs}}
    tree = MCTS(None, cfg)
    rng = np.random.default_rng(cs)
    acts = list(rng.integers(0, A, 300))
    acts[10:12] = [2, 3]                                   # X then X': back to the state 10 moves in
    st = oracle.solved(cs, 1)
    n_solved = 0
    for i, a in enumerate([0, 1] + acts):                  # the first two moves end on the solved cube
        st, code, done, rew = oracle.step(cs, st, np.array([a], np.uint8))
        s, r, d, _ = env.step(int(a))
        assert r == float(rew[0]) and d == bool(done[0])
        n_solved += d
        own, child_code, child_solved = env.expand_host()
        assert own == code[0].tobytes() == tree.key_of_state(s) == MCTS.key_of(env)
        ch, cc, cso = oracle.expand(cs, st)
        assert (child_code == cc[0]).all() and (child_solved == cso[0].astype(bool)).all()
        if i % 50 == 0:
            _, _, _, dense = env.expand_host(dense=True)
            _, oh = oracle.encode(cs, ch[0])
            assert (dense == oh).all()
    assert n_solved >= 1
    assert (env.sim_cube == st[0]).all()
    for k in (0, 1, 7, 60, 61, 131):                        # whole descents in one launch (60 moves per launch)
        seq = [int(a) for a in rng.integers(0, A, k)]
        for a in seq:
            st, code, done, rew = oracle.step(cs, st, np.array([a], np.uint8))
        s, r, d, info = env.step_many(seq)
        assert tree.key_of_state(s) == code[0].tobytes() and d == bool(done[0]) and r == float(rew[0]) and info == {}
        assert (env.sim_cube == st[0]).all() and (env.cube == s).all()
    with pytest.raises(IndexError):
        env.step_many([0, A])


def _mp_child(rank, n, steps, seed, q):
    """Fresh process (spawn): its own VecCubeEnv on the shared GPU with stream_id = rank; reports its action draws and
    final stickers."""
    import numpy as np
    import torch
    import rubiks_cube_solver_amd as r
    env = r.VecCubeEnv(n, "cuda", 3, obs="code", seed=seed, stream_id=rank)
    env.reset(scramble_count=5)
    g = torch.Generator(device="cuda").manual_seed(100 + rank)
    for _ in range(steps):
        env.step(torch.randint(0, 12, (n,), generator=g, device="cuda", dtype=torch.uint8))
    env.check_actions()
    env2 = r.make_env(torch.device("cpu"), 3)               # the batch-1 facade in the same worker, like train.py:141
    env2.reset(seed=rank * 10, scramble_count=7)
    q.put((rank, env.sim_cube.cpu().numpy(), env2.sim_cube.copy()))


def test_envs_in_several_processes_share_one_gpu(oracle, golden):
    """train.py:85-92,141-147 runs one env per worker process.  Here 3 spawned processes each own a VecCubeEnv
    (stream_id = rank) and a CubeEnv on the SAME GPU at once, 200 steps each; every rank's result equals the oracle's
    replay of its own (seed, rank) stream and the ranks' streams differ.  Children are fresh processes (spawn): a
    process that has touched the GPU is never forked."""
    import multiprocessing as mp
    import os
    ctx = mp.get_context("spawn")
    n, steps, seed, world = 3000, 200, 4242, 3
    q = ctx.Queue()
    env = dict(os.environ)
    procs = [ctx.Process(target=_mp_child, args=(rank, n, steps, seed, q)) for rank in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, st, st1 = q.get(timeout=300)
        got[rank] = (st, st1)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    finals = []
    for rank in range(world):
        # the device RNG stream of reset(): (seed, stream_id = rank, walk = resets * n + i), restated by the oracle
        exp = oracle.adi(3, n, 5, seed=seed, stream=rank, walk0=n, want_children=False)["parents"][:, -1]
        g = torch.Generator(device="cuda").manual_seed(100 + rank)
        st = exp
        for _ in range(steps):
            a = torch.randint(0, 12, (n,), generator=g, device="cuda", dtype=torch.uint8).cpu().numpy()
            st = oracle.step(3, st, a, threads=4)[0]
        assert (got[rank][0] == st).all(), rank
        finals.append(exp)
        gr = golden("reset_333")                             # the facade's reset(seed = rank*10, 7) is the reference's
        i, j = list(gr["seeds"]).index(rank * 10), list(gr["ks"]).index(7)
        assert (got[rank][1] == gr["stickers"][i, j]).all(), rank
    assert not (finals[0] == finals[1]).all() and not (finals[1] == finals[2]).all()


def test_forked_workers_build_their_own_envs(oracle, golden, tmp_path):
    """The reference starts its workers with mp.Process -- fork on Linux -- and each builds its env inside the child
    (train.py:89-91,141; SURVEY 8b "usable after fork in 14 workers").  tests/fork_workers.py does exactly that with this package:
    the parent imports it, never touches the GPU, forks 3 workers; every worker's reset(seed = rank*10, 7) is the reference's
    (G4, reset_333.npz) and its 50 steps equal the oracle's replay (stickers, one-hot columns, reward, done)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fork_workers.py"), str(tmp_path), "3"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    gr = golden("reset_333")
    pids = set()
    for rank in range(3):
        w = np.load(tmp_path / f"w{rank}.npz")
        i, j = list(gr["seeds"]).index(rank * 10), list(gr["ks"]).index(7)
        assert (w["after_reset"] == gr["stickers"][i, j]).all(), rank
        assert (w["first"].argmax(-1) == gr["cols"][i, j]).all() and w["first"].sum() == 20, rank
        st = gr["stickers"][i, j].astype(np.uint8)[None]
        for t, a in enumerate(w["actions"]):
            st, code, done, rew = oracle.step(3, st, np.array([a], dtype=np.uint8))
            assert (w["stickers"][t] == st[0]).all(), (rank, t)
            assert (w["onehots"][t].argmax(-1) == code[0]).all() and w["onehots"][t].sum() == 20 and w["onehots"][t].dtype == np.int64
            assert w["rewards"][t] == rew[0] and bool(w["dones"][t]) == bool(done[0])
        pids.add(int(w["pid"]))
        assert int(w["ppid"]) not in pids
    assert len(pids) == 3                                      # three different worker processes, one forking parent


def test_checkpoint_traces_replay_on_the_hip_222_path(mod, golden):
    """SURVEY 8f N4 on the product path.  tests/golden/crosscheck_222.npz holds the greedy solve traces of the
    reference's shipped 2x2x2 checkpoint (pretrained/222model.pt, model.py:47-76) for seeds 0..39 x k in
    {1,2,3,4,6,8,10,12,14}: the scramble draws of reset(seed, k), the policy's actions and, per step, the arg-max column of every
    one-hot row and the done flag.  Here the Cube2 KERNELS replay them: VecCubeEnv(cube_size=2).reset(seeds, ks) (numpy's
    legacy generator on the device), then one rc_apply_moves per time step.  The authors' policy solving every one of
    these cubes is the only external evidence for the unpinned 2x2x2 convention (cube_env.py:142-147); this test
    carries it to the HIP code."""
    g = golden("crosscheck_222")
    n, T = g["actions"].shape
    env = mod.VecCubeEnv(n, "cuda", 2, obs="onehot", onehot_dtype=torch.uint8)
    env.reset(seeds=[int(s) for s in g["seeds"]], scramble_count=[int(k) for k in g["ks"]])
    env2 = mod.VecCubeEnv(n, "cuda", 2, obs="onehot", onehot_dtype=torch.uint8)
    env2.reset(actions=g["scramble"], scramble_count=g["scramble"].shape[1])     # the recorded draws, no-op padded
    assert torch.equal(env.sim_cube, env2.sim_cube)                    # device MT19937 == the draws the fixture recorded
    solve_step = np.zeros(n, np.int32)
    for t in range(T):
        obs, rew, done, _ = env.step(torch.from_numpy(np.ascontiguousarray(g["actions"][:, t])).cuda())
        assert (obs.argmax(-1).cpu().numpy() == g["cols"][:, t]).all(), t
        assert int(obs.sum()) == n * 7                                  # exactly one 1 per row
        d = done.cpu().numpy()
        assert (d == g["done"][:, t]).all(), t
        assert (rew.cpu().numpy() == np.where(d, 1.0, -1.0)).all()
        solve_step[(solve_step == 0) & (d != 0)] = t + 1
    assert (solve_step == g["solve_step"]).all()
    assert (solve_step[g["ks"] <= 8] > 0).all()                         # the authors' policy solves every scramble up to depth 8
    assert (solve_step > 0).mean() >= 0.9
    env.check_actions()
    # the action lists the REFERENCE'S OWN MCTS returned on the restated env (50 simulations, depths up to 14) solve the same
    # scrambles on the Cube2 kernels, through the same one-hots
    found = g["mcts_found"].astype(bool)
    sol, cols = g["mcts_solution"][found], g["mcts_cols"][found]
    m = int(found.sum())
    menv = mod.VecCubeEnv(m, "cuda", 2, obs="onehot", onehot_dtype=torch.uint8)
    menv.reset(seeds=[int(x) for x in g["mcts_seeds"][found]], scramble_count=[int(k) for k in g["mcts_ks"][found]])
    chk = mod.VecCubeEnv(m, "cuda", 2, obs=None)
    chk.reset(actions=g["mcts_scramble"][found], scramble_count=g["mcts_scramble"].shape[1])
    assert torch.equal(menv.sim_cube, chk.sim_cube)
    last = (sol < 6).sum(1)                                             # length of each action list
    for t in range(int(last.max())):
        obs, rew, done, _ = menv.step(torch.from_numpy(np.ascontiguousarray(sol[:, t])).cuda())   # the no-op parks finished lists
        assert (obs.argmax(-1).cpu().numpy() == cols[:, t]).all(), t
        d = done.cpu().numpy().astype(bool)
        assert (d[last == t + 1]).all() and not d[last > t + 1].any(), t      # solved exactly by the last move of its list
    assert menv.is_solved().cpu().numpy().all()


# ----------------------------------------------------------------------------- N2: guided searches (fixture G12)
_POW31 = [pow(31, i, 1 << 63) for i in range(20)]


def _guided_table(g):
    """Fixture G12's stub as data: hash of the 20 one-hot columns -> (distance to solved, move back towards solved)."""
    cols = g["table_cols"].astype(np.int64)
    h = (cols * np.array(_POW31, np.int64)).sum(1)
    order = np.argsort(h)
    assert len(np.unique(h)) == len(h)
    return cols, h[order], g["table_depth"][order].astype(np.int64), g["table_back"][order].astype(np.int64)


class _GuidedHostStub:
    """predict(state) exactly as the fixture's generator defined it (make_golden.py golden_mcts_guided)."""

    def __init__(self, g):
        self.table = {tuple(c): (int(d), int(b)) for c, d, b in zip(g["table_cols"].tolist(), g["table_depth"], g["table_back"])}

    def predict(self, x):
        hit = self.table.get(tuple(np.asarray(x).argmax(-1).tolist()))
        logits, value = np.zeros(12, np.float32), np.float32(-9.0)
        if hit is not None:
            value = np.float32(-float(hit[0]))
            logits[hit[1]] = 2.0
        e = np.exp(logits - logits.max())
        return np.array([value], np.float32), (e / e.sum()).astype(np.float32)


def _guided_device_model(g):
    """The same stub as a batched torch callable: (value [B,1], LOGITS [B,12]); BatchedMCTS applies the softmax."""
    _, h_sorted, depth, back = _guided_table(g)
    hs, dp, bk = (torch.from_numpy(a).cuda() for a in (h_sorted, depth, back))
    powers = torch.tensor(_POW31, dtype=torch.int64, device="cuda")

    def model(x):
        h = (x.argmax(-1).to(torch.int64) * powers).sum(-1)
        idx = torch.searchsorted(hs, h).clamp_(max=len(hs) - 1)
        hit = hs[idx] == h
        value = torch.where(hit, -dp[idx].float(), torch.full_like(h, -9, dtype=torch.float32))
        logits = torch.nn.functional.one_hot(bk[idx], 12).float() * (2.0 * hit.float()).unsqueeze(-1)   # capturable: no data-dependent shapes
        return value.unsqueeze(-1), logits
    return model


def test_mcts_guided_matches_reference_along_the_returned_path(mod, golden):
    """G12: the reference's MCTS with a guiding stub on depth 3-5 scrambles (27 of 36 searches FIND a solution, paths up to 12
    moves, up to 44 simulations).  mcts_batched.MCTS with the same stub and the same seeded `random`: simulations used, the
    returned action list, and visits / values / virtual losses of EVERY node from the root to the expanded leaf
    (mcts.py:52-130: deep descents, transpositions through the code-keyed table, the literal -= 150)."""
    import random

    from rubiks_cube_solver_amd.mcts_batched import MCTS
    g = golden("mcts_guided_333")
    stub = _GuidedHostStub(g)
    cfg = {"mcts": {"virtual_loss_const": 150, "cpuct": 1.0, "value_min": -10.0, "numMCTSSim": 50}, "test": {"cube_size": 3}}
    env = mod.make_env(torch.device("cpu"), 3)
    n_found = 0
    for i, (seed, k) in enumerate(zip(g["seeds"], g["ks"])):
        state = env.reset(seed=int(seed), scramble_count=int(k))
        random.seed(int(g["random_seed"][i]))
        tree, found, used = MCTS(stub, cfg), None, 0
        for sim in range(60):
            used = sim + 1
            found = tree.train(state, env)
            if found is not None:
                break
        assert used == int(g["sims"][i]), (i, used)
        exp = [int(a) for a in g["solution"][i] if a != 255]
        assert (found or []) == exp, i
        if found is None:
            continue
        n_found += 1
        key = tree.key_of_state(state)
        assert int(g["path_nodes"][i]) == len(found)
        for t, a in enumerate(found):
            node = tree.children_and_data[key]
            assert node.visits == g["path_visits"][i, t].tolist(), (i, t)
            assert [float(v) for v in node.value] == g["path_values"][i, t].tolist(), (i, t)
            assert [float(v) for v in node.vloss] == g["path_vloss"][i, t].tolist(), (i, t)
            key = node.children[a]
    assert n_found == 27 and max(int(x) for x in g["path_nodes"]) >= 12


@pytest.mark.parametrize("native", [True, False])
def test_batched_mcts_guided_lockstep(mod, golden, native):
    """The 36 searches of fixture G12 run TOGETHER (one replay + one expansion + one stub evaluation per simulation): every
    root ends like the reference's stand-alone run; with the Python trees the statistics of every node on the returned path
    are compared too."""
    import random

    from rubiks_cube_solver_amd.mcts_batched import BatchedMCTS
    g = golden("mcts_guided_333")
    n = len(g["seeds"])
    venv = mod.VecCubeEnv(n, "cuda", 3, obs=None)
    venv.reset(seeds=[int(s) for s in g["seeds"]], scramble_count=[int(k) for k in g["ks"]])
    rngs = [random.Random(int(s)) for s in g["random_seed"]]
    bm = BatchedMCTS(_guided_device_model(g), venv.stickers, n, 3, rngs=rngs, graph=False, native=native)
    for _ in range(60):
        bm.simulate()
    bm.sync_rngs()
    for r in range(n):
        assert bm.sims_used[r] == int(g["sims"][r]), r
        exp = [int(a) for a in g["solution"][r] if a != 255]
        assert (bm.solution[r] or []) == exp, r
        if not exp:
            continue
        root = bm.trees[r][b"root"]
        assert list(root.visits) == g["path_visits"][r, 0].tolist(), r
        assert [float(v) for v in root.value] == g["path_values"][r, 0].tolist(), r
        if not native:
            node = root
            for t, a in enumerate(exp):
                assert node.visits == g["path_visits"][r, t].tolist() and [float(v) for v in node.vloss] == g["path_vloss"][r, t].tolist(), (r, t)
                assert [float(v) for v in node.value] == g["path_values"][r, t].tolist(), (r, t)
                node = bm.trees[r].get(node.children[a])
    # the generators were advanced like the stand-alone runs' global `random`: a fresh run from the synced state continues
    # where the reference's would (the first case drew at least once)
    assert rngs[0].getstate() != random.Random(int(g["random_seed"][0])).getstate()


@pytest.mark.parametrize("graph", [False, True])
def test_batched_mcts_4096_roots_every_replica_matches_its_reference_run(mod, golden, graph):
    """BASELINE config 5's size, checked (not only timed): 4096 roots = fixture G12's 36 scrambles tiled 113.8 times, every
    replica with its own generator seeded like the reference's stand-alone run (mcts.py:36-154, config.yaml:29-32).  After 60
    lockstep simulations every replica has used the same number of simulations, returns the same action list and has the same
    root statistics as that run."""
    import random

    from rubiks_cube_solver_amd.mcts_batched import BatchedMCTS
    g = golden("mcts_guided_333")
    n, m = 4096, len(g["seeds"])
    case = np.arange(n) % m
    venv = mod.VecCubeEnv(n, "cuda", 3, obs=None)
    venv.reset(seeds=[int(g["seeds"][c]) for c in case], scramble_count=[int(g["ks"][c]) for c in case])
    bm = BatchedMCTS(_guided_device_model(g), venv.stickers, n, 3, rngs=[random.Random(int(g["random_seed"][c])) for c in case],
                     graph=graph, native=True)
    solved = 0
    for _ in range(60):
        solved = bm.simulate()
    sims, sols = bm.sims_used, bm.solution
    exp_sol = [[int(a) for a in g["solution"][c] if a != 255] for c in range(m)]
    assert solved == int(sum(bool(exp_sol[c]) for c in case))
    for r in range(n):
        c = int(case[r])
        assert sims[r] == int(g["sims"][c]), (r, c)
        assert (sols[r] or []) == exp_sol[c], (r, c)
    for r in list(range(0, n, 97)) + list(range(n - 36, n)):                  # root statistics on a strided sample + the tail
        c = int(case[r])
        if exp_sol[c]:
            root = bm.trees[r][b"root"]
            assert list(root.visits) == g["path_visits"][c, 0].tolist() and [float(v) for v in root.value] == g["path_values"][c, 0].tolist(), r


def test_vec_env_debug_action_check_and_clone(mod):
    """debug_check_every=K surfaces an out-of-range device action as the reference's IndexError (cube_env.py:86,96) within K
    steps; clone() is an independent deep copy (lean=True copies the state only)."""
    env = mod.VecCubeEnv(64, "cuda", 3, obs="code", debug_check_every=2)
    env.reset(scramble_count=3)
    good = torch.zeros(64, dtype=torch.uint8, device="cuda")
    bad = good.clone()
    bad[7] = 200
    env.step(good)
    twin, lean = env.clone(), env.clone(lean=True)
    assert torch.equal(twin.stickers, env.stickers) and torch.equal(twin.done, env.done) and torch.equal(lean.stickers, env.stickers)
    env.step(bad)                                 # step 2: the check runs before the launch, nothing bad seen yet
    with pytest.raises(IndexError):
        env.step(good)                            # step 3 launches; step 4's check ...
        env.step(good)                            # ... raises here at the latest
    assert not torch.equal(twin.stickers, env.stickers)        # the copies did not follow
    twin.step(good)
    lean.step(good)
    assert torch.equal(twin.sim_cube, lean.sim_cube)
    with pytest.raises(ValueError):
        env.step(torch.zeros(63, dtype=torch.uint8, device="cuda"))   # wrong length is refused on the fast path too


# ----------------------------------------------------------------------------- N1: tensor replay sink (fixture G11)
def test_tensor_replay_buffer_matches_reference_class(golden):
    """Fixture G11 = the reference's ReplayBuffer (utils.py:203-270) fed by its own get_random_samples with G5's stub model.
    The tensor sink, fed by the PRODUCT's get_random_samples under the same global seed, must hold the same samples after the
    deque eviction, draw the same prioritised indices under the same legacy-RNG seeds, return the same 5-tuples (values and
    dtypes), visit them in the DataLoader's order, and follow update() and later appends exactly."""
    import rubiks_cube_solver_amd as rc
    g, g5 = golden("replay_333"), golden("adi_333")
    w, b = torch.tensor(g5["w"]), torch.tensor(g5["b"])

    class Stub(torch.nn.Module):
        def forward(self, x):
            if x.dim() == 2:
                x = x.unsqueeze(0)
            return (x.reshape(x.shape[0], -1).float().cpu() @ w + b).unsqueeze(-1), torch.zeros(x.shape[0], 12)

    env = rc.make_env(torch.device("cpu"), 3)
    rb = rc.TensorReplayBuffer(int(g["buf_size"]), int(g["sample_size"]), cube_size=3, device="cuda")
    np.random.seed(int(g5["seed"]))
    env.get_random_samples(rb, Stub(), 30, 64, float(g5["temperature"]))          # 1920 samples into a ring of 1500
    n = rb.size
    assert n == int(g["buf_size"])
    phys = torch.from_numpy(rb._phys(np.arange(n))).cuda()
    assert (rb.code[phys].cpu().numpy() == g["mem_cols"]).all()                    # same samples survive, oldest first
    got_err = rb.error_memory[rb._phys(np.arange(n))]
    assert np.allclose(got_err, g["mem_err"], rtol=0, atol=2e-6)                   # fp32 stub on the device vs the reference's CPU dot
    # from here on the DRAWS are what is tested: give both sides bit-identical probabilities
    rb.error_memory[rb._phys(np.arange(n))] = g["mem_err"]
    np.random.seed(31337)
    idx1 = rb.get_prioritized_sample()
    assert (idx1 == g["idx1"]).all() and len(rb) == int(g["sample_size"])
    items = [rb[i] for i in range(len(rb))]
    assert [str(t.dtype) for t in items[0]] == list(g["item_dtypes"])
    assert (torch.stack([it[0] for it in items]).argmax(-1).cpu().numpy() == g["item_cols"]).all()
    assert all(int(it[0].sum()) == 20 and tuple(it[0].shape) == (20, 24) for it in items[:8])
    tv = torch.stack([it[1] for it in items]).cpu().numpy()
    solved = g["item_tv"] == 1.0
    assert (tv[solved] == 1.0).all() and np.allclose(tv, g["item_tv"], rtol=0, atol=2e-6)
    assert (torch.stack([it[2] for it in items]).cpu().numpy() == g["item_tp"]).all()
    assert (torch.stack([it[3] for it in items]).cpu().numpy() == g["item_sc"]).all()
    assert (torch.stack([it[4] for it in items]).cpu().numpy() == g["item_idx"]).all()
    # DataLoader(replay_buffer, batch_size, shuffle=True) of update_params (utils.py:296-303), and the batched iterator
    from torch.utils.data import DataLoader
    torch.manual_seed(4242)
    assert (np.concatenate([bt[4].cpu().numpy() for bt in DataLoader(rb, batch_size=100, shuffle=True)]) == g["loader_idx"]).all()
    torch.manual_seed(4242)
    seen = []
    for st, v, pol, sc, mem in rb.batches(100, shuffle=True, dtype=torch.bfloat16):
        assert st.dtype == torch.bfloat16 and st.shape[1:] == (20, 24) and float(st.float().sum()) == 20 * st.shape[0]
        where = {int(m): k for k, m in enumerate(g["item_idx"])}
        rows = [where[int(m)] for m in mem.cpu().numpy()]
        assert (st.float().argmax(-1).cpu().numpy() == g["item_cols"][rows]).all() and (sc.cpu().numpy() == g["item_sc"][rows]).all()
        seen.append(mem.cpu().numpy())
    assert (np.concatenate(seen) == g["loader_idx"]).all()
    # update() then a second draw; scalar and batched forms
    half = len(g["upd_idx"]) // 2
    for i, e in zip(g["upd_idx"][:half], g["upd_err"][:half]):
        rb.update(int(i), float(e))
    rb.update(torch.from_numpy(g["upd_idx"][half:]), torch.from_numpy(g["upd_err"][half:]))
    np.random.seed(99)
    assert (rb.get_prioritized_sample() == g["idx2"]).all()
    # more samples arrive: the oldest 300 leave, indices shift as in a deque
    np.random.seed(7)
    env.get_random_samples(rb, Stub(), 30, 10, float(g5["temperature"]))
    n = rb.size
    phys = torch.from_numpy(rb._phys(np.arange(n))).cuda()
    assert (rb.code[phys].cpu().numpy() == g["mem3_cols"]).all()
    assert np.allclose(rb.error_memory[rb._phys(np.arange(n))], g["mem3_err"], rtol=0, atol=2e-6)
    rb.error_memory[rb._phys(np.arange(n))] = g["mem3_err"]
    np.random.seed(123)
    assert (rb.get_prioritized_sample() == g["idx3"]).all()
    # a buffer that is not full returns every index in order; single reference-style dicts are accepted too
    small = rc.TensorReplayBuffer(5000, 4000, cube_size=3, device="cuda")
    np.random.seed(5)
    env.get_random_samples(small, Stub(), 5, 8, float(g5["temperature"]))
    assert (small.get_prioritized_sample() == g["small_idx"]).all() and small.size == 40
    ref_style = []
    np.random.seed(5)
    env.get_random_samples(ref_style, Stub(), 5, 8, float(g5["temperature"]))
    one = rc.TensorReplayBuffer(64, 64, cube_size=3, device="cuda")
    for smp in ref_style:
        one.append(smp)
    one.get_prioritized_sample()
    assert torch.equal(one.code[:40], small.code[:40]) and torch.equal(one.target_policy[:40], small.target_policy[:40])
    assert np.array_equal(one.error_memory[:40], small.error_memory[:40]) and torch.equal(one[3][0], small[3][0])


def test_tensor_replay_buffer_222_and_wraparound():
    """2x2x2 samples ([7,21] float64 one-hots, transposed convention cube_env.py:142-147) through the sink, with the ring
    wrapping several times in one append and across appends; every stored sample equals the dict adapter's."""
    import rubiks_cube_solver_amd as rc

    class Stub(torch.nn.Module):
        def forward(self, x):
            if x.dim() == 2:
                x = x.unsqueeze(0)
            f = x.reshape(x.shape[0], -1).float().cpu()
            return (f @ torch.linspace(-0.3, 0.4, f.shape[1]) + 0.05).unsqueeze(-1), torch.zeros(x.shape[0], 6)

    env = rc.make_env(torch.device("cpu"), 2)
    ref_style, rb = [], rc.TensorReplayBuffer(37, 16, cube_size=2, device="cuda")
    for seed, (depth, cubes) in enumerate(((6, 4), (9, 11), (3, 2))):              # 24, 99 (> capacity), 6 samples
        np.random.seed(seed)
        env.get_random_samples(ref_style, Stub(), depth, cubes, 0.5)
        np.random.seed(seed)
        env.get_random_samples(rb, Stub(), depth, cubes, 0.5)
        keep = ref_style[-37:]
        assert rb.size == len(keep)
        np.random.seed(1000 + seed)
        idx = rb.get_prioritized_sample()
        for j, i in enumerate(idx[:16]):
            st, tv, tp, sc, mi = rb[j]
            smp = keep[int(i)]
            assert st.dtype == torch.float64 and (st.cpu().numpy() == smp["state"]).all()
            assert float(tv) == np.float32(smp["target_value"]) and int(tp) == smp["target_policy"] and int(sc) == smp["scramble_count"]
            assert int(mi) == int(i) and rb.error_memory[rb._phys(int(i))] == smp["error"]


def test_tensor_replay_buffer_append_between_sampling_and_reading():
    """The reference reads state AND targets from the same deque entry at __getitem__ time (utils.py:226-243), so an append
    between get_prioritized_sample() and the reads -- which shifts a full deque under the logical indices -- still returns
    consistent samples.  The tensor ring does the same: the dense cache is dropped by every append."""
    from collections import deque
    import rubiks_cube_solver_amd as rc

    class Stub(torch.nn.Module):
        def forward(self, x):
            if x.dim() == 2:
                x = x.unsqueeze(0)
            f = x.reshape(x.shape[0], -1).float().cpu()
            return (f @ torch.linspace(-0.2, 0.3, f.shape[1]) + 0.01).unsqueeze(-1), torch.zeros(x.shape[0], 12)

    env = rc.make_env(torch.device("cpu"), 3)
    cap = 40
    ref, rb = deque(maxlen=cap), rc.TensorReplayBuffer(cap, 12, cube_size=3, device="cuda")
    for seed, (depth, cubes) in enumerate(((6, 6), (5, 3))):                       # 36 samples, then 15 more: the ring wraps by 11
        np.random.seed(seed)
        env.get_random_samples(ref, Stub(), depth, cubes, 0.5)
        np.random.seed(seed)
        env.get_random_samples(rb, Stub(), depth, cubes, 0.5)
        if seed == 0:
            np.random.seed(5)
            idx = rb.get_prioritized_sample().copy()                                # sampled BEFORE the second append
    assert rb.size == cap == len(ref)
    got = list(rb.batches(5, shuffle=False, dtype=torch.int64))
    flat = [torch.cat([b[k] for b in got]).cpu().numpy() for k in range(5)]
    for j, i in enumerate(idx):
        smp = ref[int(i)]                                                           # the reference: the entry that index names NOW
        st, tv, tp, sc, mi = rb[j]
        assert (st.cpu().numpy() == smp["state"]).all() and float(tv) == np.float32(smp["target_value"])
        assert int(tp) == smp["target_policy"] and int(sc) == smp["scramble_count"] and int(mi) == int(i)
        assert (flat[0][j] == smp["state"]).all() and flat[1][j] == np.float32(smp["target_value"]) and flat[2][j] == smp["target_policy"]
    rb.update(idx, np.arange(len(idx)) + 0.5)                                       # update() names the same live entries
    assert [rb.error_memory[rb._phys(int(i))] for i in idx] == [k + 0.5 for k in range(len(idx))]


def test_adi_samples_feeds_the_net_in_its_own_dtype(mod):
    """adi_samples writes the dense one-hot stream in the dtype the value net computes in (a bfloat16 net gets bfloat16
    one-hots: half the bytes of the 13 x walks x 480 elements per depth); the samples are those of the float32 route up to the
    net's own rounding."""
    from rubiks_cube_solver_amd.adi import adi_samples

    class Net(torch.nn.Module):
        def __init__(self, dtype):
            super().__init__()
            g = torch.Generator().manual_seed(3)
            self.w = torch.nn.Parameter((torch.randn(480, generator=g) * 0.05).to(dtype))
            self.seen = set()

        def forward(self, x):
            self.seen.add(x.dtype)
            v = x.reshape(x.shape[0], -1) @ self.w
            return v.unsqueeze(-1), torch.zeros(x.shape[0], 12, device=x.device, dtype=x.dtype)

    lo, hi = Net(torch.bfloat16).cuda(), Net(torch.float32).cuda()
    with torch.no_grad():
        hi.w.copy_(lo.w.float())                                   # the same weights, exactly
    a = adi_samples(lo, 3, 3000, 9, 0.5, device="cuda", seed=11)
    b = adi_samples(hi, 3, 3000, 9, 0.5, device="cuda", seed=11)
    assert lo.seen == {torch.bfloat16} and hi.seen == {torch.float32}
    assert torch.equal(a["state_code"], b["state_code"]) and torch.equal(a["actions"], b["actions"])
    assert a["target_value"].dtype == torch.float32 and torch.allclose(a["target_value"], b["target_value"], atol=2e-2, rtol=0)
    solved = b["target_value"] == 1.0
    assert torch.equal(a["target_value"][solved], b["target_value"][solved]) and torch.equal(a["target_policy"][solved], b["target_policy"][solved])
    assert (a["target_policy"] == b["target_policy"]).float().mean() > 0.9          # near ties may fall differently after bf16 rounding
    c = adi_samples(hi, 3, 500, 4, 0.5, device="cuda", seed=11, dense_dtype=torch.float32)
    assert torch.equal(c["state_code"], b["state_code"][:500, :4])


# ------------------------------------------------------------------------------ G13: the reference's own 2x2x2 branches
def _stub_222(g, device="cpu"):
    w, b = torch.tensor(g["adi_w"], device=device), torch.tensor(g["adi_b"], device=device)

    class StubModel(torch.nn.Module):
        def forward(self, x):
            if x.dim() == 2:
                x = x.unsqueeze(0)
            return (x.reshape(x.shape[0], -1).float() @ w + b).unsqueeze(-1), torch.zeros(x.shape[0], 6, device=x.device)
    return StubModel()


def test_facade_222_matches_the_references_own_222_branches(mod, golden):
    """G13 = the REFERENCE's CubeEnv run with cube_size = 2 (stand-in py222 under it).  The product facade on the GPU: every reset(seed, k),
    walks step by step (stickers, the transposed float64 one-hot of cube_env.py:143-147, reward, done), state_to_sim_state
    (cube_env.py:154-175), get_target_value (cube_env.py:196-252; the model is called as the reference calls it: 1e-6)."""
    g = golden("env222_via_reference")
    env = mod.make_env(torch.device("cpu"), 2)
    assert (env.sim_cube == g["solved_stickers"]).all() and (np.argmax(env.cube, 1) == g["solved_cols"]).all() and str(env.cube.dtype) == str(g["state_dtype"])
    np.random.seed(5)
    before = np.random.get_state()[1].copy()
    for i, sd in enumerate(g["reset_seeds"]):
        for j, k in enumerate(g["reset_ks"]):
            s = env.reset(seed=int(sd), scramble_count=int(k))
            assert s.dtype == np.float64 and s.shape == (7, 21) and (np.argmax(s, 1) == g["reset_cols"][i, j]).all() and (s.sum(1) == 1).all()
            assert (env.sim_cube == g["reset_stickers"][i, j]).all()
    assert (np.random.get_state()[1] == before).all()
    for w in range(0, 300, 6):
        env.init_state()
        for d in range(14):
            s, r, dn, info = env.step(int(g["walk_actions"][w, d]))
            assert (env.sim_cube == g["walk_stickers"][w, d]).all() and (np.argmax(s, 1) == g["walk_cols"][w, d]).all() and (s.sum(1) == 1).all()
            assert isinstance(r, float) and isinstance(dn, bool) and (r, dn) == (g["walk_reward"][w, d], bool(g["walk_done"][w, d])) and info == {}
            if d % 5 == 0:
                assert (env.state_to_sim_state(env.cube) == g["roundtrip_stickers"][w, d // 5]).all()
    model, T = _stub_222(g), float(g["adi_temperature"])
    for c in (0, 7, 31):
        env.init_state()
        for d in range(g["adi_actions"].shape[1]):
            env.step(int(g["adi_actions"][c, d]))
            tv, tp, err = env.get_target_value(model, d + 1, T)
            assert isinstance(tv, float) and isinstance(tp, int) and tp == g["adi_target_policy"][c, d]
            assert tv == pytest.approx(g["adi_target_value"][c, d], abs=1e-6) and err == pytest.approx(g["adi_error"][c, d], abs=1e-6)


def test_batched_222_walks_and_adi_match_the_references_own_222_branches(mod, oracle, golden):
    """The batched 2x2x2 kernels against G13: 300 walks stepped together (rc_apply_moves with the fused dense float32 one-hot [N, 7, 21],
    reward, done), get_random_samples through the facade (dict records, legacy-RNG draws) and adi_samples with replayed moves (AdiPlan's
    packed blocks, rc_onehot_from_code_blocks): integer outputs exact, value-net floats within 1e-5, policy equal except on a top-2 tie."""
    from rubiks_cube_solver_amd import _lib, ops
    from rubiks_cube_solver_amd.adi import adi_samples
    g = golden("env222_via_reference")
    W, D = g["walk_actions"].shape
    st = ops.alloc_states(W, 2, "cuda")
    ops.fill_solved(st, W, 2)
    oh = torch.empty((W, 7, 21), dtype=torch.float32, device="cuda")
    rew = torch.empty(W, dtype=torch.float32, device="cuda")
    done = torch.empty(W, dtype=torch.uint8, device="cuda")
    for d in range(D):
        ops.apply_moves(st, st, torch.from_numpy(g["walk_actions"][:, d].copy()).cuda(), W, 2, rew, done, oh, _lib.FMT_F32)
        assert (ops.to_aos(st, W).cpu().numpy() == g["walk_stickers"][:, d]).all()
        assert (oh.argmax(-1).cpu().numpy() == g["walk_cols"][:, d]).all() and float(oh.sum()) == 7 * W
        assert (done.cpu().numpy() == g["walk_done"][:, d]).all() and (rew.cpu().numpy().astype(np.float64) == g["walk_reward"][:, d]).all()
    n, depth = g["adi_actions"].shape
    T = float(g["adi_temperature"])
    # child values of the stub on the host (float64) for the tie rule
    out = oracle.adi(2, n, depth, actions_in=g["adi_actions"], want_children=False)
    wv = g["adi_w"].reshape(7, 21).astype(np.float64)
    code = out["child_code"].astype(np.int64)
    v = np.sort(wv[code // 3, np.arange(7) * 3 + code % 3].sum(-1), -1)
    tie = (v[..., -1] - v[..., -2] < 1e-5) & ~out["child_solved"].any(-1)
    # (a) the drop-in call: env.get_random_samples with a list sink
    env = mod.make_env(torch.device("cpu"), 2)
    buf = []
    np.random.seed(int(g["adi_seed"]))
    env.get_random_samples(buf, _stub_222(g), depth, n, T)
    assert len(buf) == n * depth and (env.sim_cube == g["adi_final_stickers"]).all()
    for i, smp in enumerate(buf):
        c, d = divmod(i, depth)
        assert list(smp) == ["state", "target_value", "target_policy", "scramble_count", "error"] and smp["state"].dtype == np.float64
        assert (np.argmax(smp["state"], 1) == g["adi_cols"][c, d]).all() and (smp["state"].sum(1) == 1).all() and smp["scramble_count"] == d + 1
        assert smp["target_value"] == pytest.approx(g["adi_target_value"][c, d], abs=1e-5) and smp["error"] == pytest.approx(g["adi_error"][c, d], abs=1e-5)
        assert smp["target_policy"] == g["adi_target_policy"][c, d] or tie[c, d], (c, d)
    assert all(b["target_value"] == 1.0 for b in buf[::depth])
    # (b) adi_samples with the model on the GPU and the same moves replayed
    res = adi_samples(_stub_222(g, "cuda"), 2, n, depth, T, device="cuda", actions=g["adi_actions"], want_state_dense=True)
    assert (res["state"].argmax(-1).cpu().numpy() == g["adi_cols"]).all() and int(res["state"].sum()) == n * depth * 7
    assert np.allclose(res["target_value"].cpu().numpy(), g["adi_target_value"], rtol=0, atol=1e-5)
    assert np.allclose(res["error"].cpu().numpy(), g["adi_error"], rtol=0, atol=1e-5)
    got = res["target_policy"].cpu().numpy()
    assert ((got == g["adi_target_policy"]) | tie).all() and (got == g["adi_target_policy"]).mean() > 0.99
    assert (res["target_value"].cpu().numpy()[g["adi_target_value"] == 1.0] == 1.0).all()


def test_mcts_222_matches_the_references_own_mcts(mod, golden):
    """G13: the reference's MCTS (mcts.py, action_dim 6) over ITS OWN CubeEnv(cube_size=2) vs mcts_batched.MCTS over the product facade:
    simulations used, action lists, root visit counts and values for 30 searches (k = 1..5)."""
    import random

    from rubiks_cube_solver_amd.mcts_batched import MCTS
    g = golden("env222_via_reference")
    wv, wp = g["mcts_wv"], g["mcts_wp"]

    class Stub:
        def predict(self, x):
            f = np.asarray(x, dtype=np.float32).reshape(-1)
            logits = f @ wp
            e = np.exp(logits - logits.max())
            return np.array([f @ wv], np.float32), (e / e.sum()).astype(np.float32)

    cfg = {"mcts": {"virtual_loss_const": 150, "cpuct": 1.0, "value_min": -10.0, "numMCTSSim": 50}, "test": {"cube_size": 2}}
    env = mod.make_env(torch.device("cpu"), 2)
    for i, (seed, k) in enumerate(zip(g["mcts_seeds"], g["mcts_ks"])):
        state = env.reset(seed=int(seed), scramble_count=int(k))
        random.seed(int(g["mcts_random_seed"][i]))
        tree, found, used = MCTS(Stub(), cfg), None, 0
        for s in range(60):
            used = s + 1
            found = tree.train(state, env)
            if found is not None:
                break
        assert used == int(g["mcts_sims"][i]), (i, used, int(g["mcts_sims"][i]))
        assert (found or []) == [int(a) for a in g["mcts_solution"][i] if a != 255]
        root = tree.children_and_data[MCTS.key_of(env)]
        assert root.visits == g["mcts_root_visits"][i].tolist()
        assert np.allclose(root.value, g["mcts_root_values"][i], rtol=0, atol=1e-6)


def test_py222_names_on_the_device_match_the_standin(mod, golden, capsys):
    """rubiks_cube_solver_amd.py222: the six names of the reference's missing assets/py222.py (cube_env.py:8) on the device, against the
    harness stand-in that produced G13 (tests/golden/py222_standin.py) on random walks, and against G13's walk states."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import py222_standin as ref
    from rubiks_cube_solver_amd import py222 as P
    assert sorted(P.__all__) == sorted(["initState", "getOP", "doMove", "isSolved", "getStickers", "printCube"])
    g = golden("env222_via_reference")
    s, t = P.initState(), ref.initState()
    assert s.dtype == np.int64 and (s == t).all() and P.isSolved(s) is True and (P.getOP(s) == ref.getOP(t)).all()
    names = ["U", "U'", "F", "F'", "R", "R'"]
    for d in range(14):
        mv = names[int(g["walk_actions"][11, d])]
        s2, t = P.doMove(s, mv), ref.doMove(t, mv)
        assert s2 is not s and (s2 == t).all() and (s2 == g["walk_stickers"][11, d]).all()
        s = s2
        op = P.getOP(s)
        assert op.shape == (7, 2) and (op == ref.getOP(t)).all() and P.isSolved(s) == ref.isSolved(t) == bool(g["walk_done"][11, d])
        assert (P.getStickers(op) == s).all() and (ref.getStickers(op) == s).all()
    with pytest.raises(KeyError):
        P.doMove(s, "D")                                             # the 2x2x2 model has no D / B / L turns (cube_env.py:25)
    P.printCube(s)
    assert len(capsys.readouterr().out.splitlines()) == 6
