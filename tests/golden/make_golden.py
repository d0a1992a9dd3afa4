#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Runs ONLY in the build container (needs /root/reference, which never travels to the
GPU box).  The reference is imported in-process, unmodified, behind three harness-side
shims:
  1. numpy.int = int            (py333.py:171,182,238 use the removed alias)
  2. a stub `gym` module         (cube_env.py:1; gym is not installed)
  3. a stub `assets.py222`       (cube_env.py:8; the file is absent from the reference)
Nothing from the reference is copied: the fixtures hold inputs (action sequences,
seeds, stub-model weights) and the outputs the reference computed for them.

    python -B tests/golden/make_golden.py                 # everything, into tests/golden/
    python -B tests/golden/make_golden.py mcts            # only the named groups (tables walks reset adi expand encode mcts rollout adi_deepcube replay mcts_guided)
    python -B tests/golden/make_golden.py --out DIR       # write somewhere else
    python -B tests/golden/make_golden.py --check         # regenerate into a temp dir, compare every array with the committed fixtures
                                                          # (12 x IDENTICAL; exit status 1 otherwise) -- tests/test_oracle.py runs it

Fixtures written (all small, np.savez_compressed):
  tables_333.npz   G1  tables as data (perm table, piece defs, hash weights, LUTs)
  walks_333.npz    G2+G3 single moves from solved; 1000 random 30-move walks, per step
                   stickers / one-hot column index / done / reward
  reset_333.npz    G4  CubeEnv.reset(seed, k) for seeds 0,10..90 x k in 1..30
  adi_333.npz      G5  get_random_samples / get_target_value with a deterministic
                   linear stub model (incl. solved-child `break` cases)
  expand_333.npz   G6  12-child expansion of random leaves (MCTS.expand's env work)
  encode_333.npz   G7  getOP_3/pos_to_state_3 on arbitrary (unreachable) colourings
                   whose hashes stay inside the LUTs; isSolved_3 on recoloured cubes
  rollout_333.npz  G9  greedy solve loops of train.py:183-193 / test.py:126-151 with the reference's own
                   DeepCube (model.py, small hidden dims, seeded init): actions taken and solve step
  adi_deepcube_333.npz G10 get_random_samples with the reference's DeepCube as the model
  replay_333.npz   G11 the reference's ReplayBuffer (utils.py:203-270) fed by get_random_samples (G5's stub model): deque
                   eviction, prioritised indices under seeded legacy draws, __getitem__ tuples, update(), DataLoader order
  mcts_guided_333.npz G12 the reference's MCTS with a guiding stub (value = rows at home): depth 3-5 scrambles that ARE
                   solved; simulations, action lists and the statistics of every node on the returned path
  mcts_333.npz     G8  the reference's MCTS (mcts.py) driven by a deterministic stub model and a
                   seeded `random`: simulations needed, returned action lists, root statistics;
                   plus reset(seed, 1000) end states for seeds 0..19 (test.py:166,279 style)
  env222_via_reference.npz G13 the reference's OWN CubeEnv and MCTS run with cube_size = 2, the six names of the file it imports but
                   does not ship (assets/py222.py, cube_env.py:8) supplied by tests/golden/py222_standin.py (the oracle's restated
                   tables): reset(seed, k), 300 random walks step by step, state_to_sim_state round trips, get_random_samples /
                   get_target_value with a linear stub model, MCTS searches.  Pins every 2x2x2 LINE of cube_env.py / mcts.py by
                   execution (RNG handling, reward / done, the transposed one-hot, the inversion, the target rule, action_dim 6);
                   the CONTENT of the authors' py222 tables stays unpinned (the stand-in is the build's restatement)
"""
import hashlib
import os
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = HERE          # --out DIR / --check write somewhere else; the committed fixtures live in HERE
FIXTURES = ("tables", "walks", "reset", "adi", "expand", "encode", "mcts", "mcts_guided", "rollout", "adi_deepcube", "replay", "env222")


def fixture_file(group):
    return "env222_via_reference.npz" if group == "env222" else f"{group}_333.npz"


def out_path(name):
    return os.path.join(OUT, name)


def earlier_fixture(name):
    """A fixture an earlier group wrote in this run (OUT), else the committed one."""
    p = os.path.join(OUT, name)
    return p if os.path.exists(p) else os.path.join(HERE, name)


def import_reference():
    np.int = int  # shim 1
    gym = types.ModuleType("gym")  # shim 2

    class Env:  # noqa: D401 - empty base class, as gym.Env is only subclassed
        pass

    gym.Env = Env
    sys.modules["gym"] = gym
    sys.path[:0] = [REF, os.path.join(REF, "gym-cube/gym_cube/envs")]
    import assets  # noqa: F401  (namespace package of the reference)

    stub = types.ModuleType("assets.py222")  # shim 3
    for name in ("initState", "getOP", "doMove", "isSolved", "getStickers", "printCube"):
        setattr(stub, name, None)
    sys.modules["assets.py222"] = stub
    import torch
    import cube_env
    from assets import py333

    return torch, cube_env, py333


def cols_of(onehot):
    """Column index of the single 1 in each of the 20 rows (py333.py:239-245)."""
    oh = np.asarray(onehot)
    assert oh.shape == (20, 24) and (oh.sum(1) == 1).all()
    return np.argmax(oh, 1).astype(np.uint8)


def generate(groups):
    """Run the reference and write the named fixture groups (all when empty) into OUT.  The groups that only need the env and a
    model (mcts, rollout, adi_deepcube, replay, mcts_guided) run LAST: `replay` reads the stub-model weights of G5's adi_333.npz."""
    want = lambda g: not groups or g in groups
    torch, cube_env, py333 = import_reference()
    torch.set_num_threads(1)
    dev = torch.device("cpu")
    A = 12
    names = ["U", "U'", "F", "F'", "R", "R'", "D", "D'", "B", "B'", "L", "L'"]

    # ---------------------------------------------------------------- G1 tables
    env = cube_env.CubeEnv(dev, cube_size=3)
    assert env.action_to_sim_action[3] == names
    assert [py333.moveInds[n] for n in names] == list(range(12))
    late = {"mcts", "rollout", "adi_deepcube", "replay", "mcts_guided", "env222"}
    if not groups or groups - late:
        base_fixtures(torch, cube_env, py333, env, dev, A, names)
    if want("mcts"):
        golden_mcts(torch, cube_env, env)
    if want("rollout"):
        golden_rollout(torch, env)
    if want("adi_deepcube"):
        golden_adi_deepcube(torch, env)
    if want("replay"):
        golden_replay(torch, env)
    if want("mcts_guided"):
        golden_mcts_guided(torch, cube_env, env)
    if want("env222"):
        golden_env222(torch, cube_env)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


def base_fixtures(torch, cube_env, py333, env, dev, A, names):
    """G1-G7 (tables, walks, reset, adi, expand, encode): always regenerated together, in this order (the legacy-RNG state of one
    block is not used by the next: every block seeds what it draws)."""
    np.savez_compressed(
        out_path("tables_333.npz"),
        moveDefs=py333.moveDefs.astype(np.uint8),
        corner_pieceDefs=py333.corner_pieceDefs.astype(np.uint8),
        edge_pieceDefs=py333.edge_pieceDefs.astype(np.uint8),
        corner_hashOP=py333.corner_hashOP.astype(np.uint8),
        edge_hashOP=py333.edge_hashOP.astype(np.uint8),
        corner_pieceInds=py333.corner_pieceInds.astype(np.uint8),
        edge_pieceInds=py333.edge_pieceInds.astype(np.uint8),
        initState=py333.initState_3().astype(np.uint8),
        action_names=np.array(names),
        state_dim=np.array(env.state_dim),
        action_dim=np.array(env.action_dim),
    )

    # ------------------------------------------------- G2 + G3 moves and walks
    single = np.zeros((A, 54), np.uint8)
    single_cols = np.zeros((A, 20), np.uint8)
    single_done = np.zeros(A, np.uint8)
    for a in range(A):
        env.init_state()
        st, r, d, _ = env.step(a)
        single[a] = env.sim_cube
        single_cols[a] = cols_of(st)
        single_done[a] = d
    W, D = 1000, 30
    actions = np.random.default_rng(12345).integers(0, A, (W, D), dtype=np.uint8)
    stickers = np.zeros((W, D, 54), np.uint8)
    cols = np.zeros((W, D, 20), np.uint8)
    done = np.zeros((W, D), np.uint8)
    reward = np.zeros((W, D), np.float32)
    h = hashlib.sha256()
    for w in range(W):
        env.init_state()
        for d in range(D):
            st, r, dn, info = env.step(int(actions[w, d]))
            assert isinstance(r, float) and isinstance(dn, bool) and info == {}
            stickers[w, d] = env.sim_cube
            cols[w, d] = cols_of(st)
            done[w, d] = dn
            reward[w, d] = r
            h.update(stickers[w, d].tobytes() + cols[w, d].tobytes() + bytes([int(dn)]))
    # KAT-D of SURVEY.md section 8c
    assert h.hexdigest() == "bba6d49d3dac850b34dd2efa3477c7af775612ccd2a229af982e5712dd276649"
    assert int(done.sum()) == 92
    # onehot dtype / values as the reference emits them
    st, _, _, _ = env.step(0)
    np.savez_compressed(
        out_path("walks_333.npz"),
        single_stickers=single, single_cols=single_cols, single_done=single_done,
        actions=actions, stickers=stickers, cols=cols, done=done, reward=reward,
        sha256=np.array(h.hexdigest()), onehot_dtype=np.array(str(st.dtype)),
    )

    # ------------------------------------------------------------- G4 reset()
    seeds = np.arange(0, 100, 10)
    ks = np.arange(1, 31)
    r_actions = np.full((len(seeds), len(ks), 30), 255, np.uint8)
    r_stickers = np.zeros((len(seeds), len(ks), 54), np.uint8)
    r_cols = np.zeros((len(seeds), len(ks), 20), np.uint8)
    np.random.seed(777)
    before = np.random.get_state()[1].copy()
    for i, s in enumerate(seeds):
        for j, k in enumerate(ks):
            state = env.reset(seed=int(s), scramble_count=int(k))
            r_stickers[i, j] = env.sim_cube
            r_cols[i, j] = cols_of(state)
            # the action draw itself, reproduced with the same legacy RNG calls
            np.random.seed(int(s))
            r_actions[i, j, :k] = np.random.randint(A, size=int(k))
    np.random.seed(777)
    assert (np.random.get_state()[1] == before).all()
    np.savez_compressed(
        out_path("reset_333.npz"),
        seeds=seeds, ks=ks, actions=r_actions, stickers=r_stickers, cols=r_cols,
    )

    # ---------------------------------------------------------------- G5 ADI
    rng = np.random.default_rng(99)
    w_lin = (rng.standard_normal(480) * 0.05).astype(np.float32)
    b_lin = np.float32(0.125)

    class StubModel(torch.nn.Module):
        """value = <onehot, w> + b ; policy output unused by the env path."""

        def __init__(self):
            super().__init__()
            self.w = torch.tensor(w_lin)
            self.b = torch.tensor(b_lin)

        def forward(self, x):
            if x.dim() == 2:
                x = x.unsqueeze(0)
            v = x.reshape(x.shape[0], -1) @ self.w + self.b
            return v.unsqueeze(-1), torch.zeros(x.shape[0], 12)

    class ListBuffer(list):
        pass

    model = StubModel()
    temperature = 0.3
    n_cubes, depth = 64, 30
    buf = ListBuffer()
    np.random.seed(2024)
    env.get_random_samples(buf, model, depth, n_cubes, temperature)
    np.random.seed(2024)
    adi_actions = np.stack([np.random.randint(A, size=depth) for _ in range(n_cubes)]).astype(np.uint8)
    assert len(buf) == n_cubes * depth
    assert set(buf[0].keys()) == {"state", "target_value", "target_policy", "scramble_count", "error"}
    adi_cols = np.stack([cols_of(b["state"]) for b in buf]).reshape(n_cubes, depth, 20)
    adi_tv = np.array([b["target_value"] for b in buf], np.float64).reshape(n_cubes, depth)
    adi_tp = np.array([b["target_policy"] for b in buf], np.int64).reshape(n_cubes, depth)
    adi_sc = np.array([b["scramble_count"] for b in buf], np.int64).reshape(n_cubes, depth)
    adi_err = np.array([b["error"] for b in buf], np.float64).reshape(n_cubes, depth)
    assert (adi_tv[:, 0] == 1.0).all()  # depth 1: the inverse move solves the cube
    # dedicated get_target_value cases: 2 moves that cancel etc.
    np.savez_compressed(
        out_path("adi_333.npz"),
        w=w_lin, b=b_lin, temperature=np.float64(temperature), seed=np.int64(2024),
        actions=adi_actions, cols=adi_cols, target_value=adi_tv, target_policy=adi_tp,
        scramble_count=adi_sc, error=adi_err,
    )

    # ------------------------------------------------------------- G6 expand
    L = 512
    leaf_actions = np.random.default_rng(4096).integers(0, A, (L, 20), dtype=np.uint8)
    leaves = np.zeros((L, 54), np.uint8)
    ch_st = np.zeros((L, A, 54), np.uint8)
    ch_cols = np.zeros((L, A, 20), np.uint8)
    ch_done = np.zeros((L, A), np.uint8)
    for i in range(L):
        env.init_state()
        for a in leaf_actions[i]:
            env.step(int(a))
        leaf = env.sim_cube.copy()
        leaves[i] = leaf
        for a in range(A):
            env.sim_cube = leaf.copy()
            st, r, dn, _ = env.step(a)
            ch_st[i, a] = env.sim_cube
            ch_cols[i, a] = cols_of(st)
            ch_done[i, a] = dn
    np.savez_compressed(
        out_path("expand_333.npz"),
        leaf_actions=leaf_actions, leaves=leaves, child_stickers=ch_st,
        child_cols=ch_cols, child_done=ch_done,
    )

    # ------------------------------------- G7 encode / solved on arbitrary colourings
    rng = np.random.default_rng(7)
    arb = []
    while len(arb) < 2048:
        s = rng.integers(0, 6, 54)
        hc = s[py333.corner_pieceDefs] @ py333.corner_hashOP
        he = s[py333.edge_pieceDefs] @ py333.edge_hashOP
        if hc.max() < 62 and he.max() < 55:
            arb.append(s)
    arb = np.array(arb)
    arb_cols = np.stack([cols_of(py333.pos_to_state_3(py333.getOP_3(s))) for s in arb])
    arb_solved = np.array([py333.isSolved_3(s) for s in arb], np.uint8)
    # recoloured solved cubes: every face uniform but not the canonical colour
    rec = np.stack([np.repeat(rng.permutation(6), 9) for _ in range(32)])
    rec = np.concatenate([rec, np.repeat(rng.integers(0, 6, (32, 6)), 9, axis=1)])
    rec_solved = np.array([py333.isSolved_3(s) for s in rec], np.uint8)
    np.savez_compressed(
        out_path("encode_333.npz"),
        stickers=arb.astype(np.uint8), cols=arb_cols, solved=arb_solved,
        recoloured=rec.astype(np.uint8), recoloured_solved=rec_solved,
    )


def golden_rollout(torch, env):
    """G9: the reference's model class + its greedy loops (no mask: train.py:183-193; mask: test.py:126-151)."""
    import model as ref_model

    torch.manual_seed(7)
    net = ref_model.DeepCube([20, 24], 12, [64, 32, 16]).eval()
    T, ks, n_seeds = 12, (1, 2, 3), 30
    out = {}
    for mask in (False, True):
        acts = np.full((len(ks), n_seeds, T), 255, np.uint8)
        solved_at = np.zeros((len(ks), n_seeds), np.int32)
        for i, k in enumerate(ks):
            for j in range(n_seeds):
                state, pre = env.reset(seed=j * 10, scramble_count=k), None
                for t in range(1, T + 1):
                    with torch.no_grad():
                        x = torch.tensor(state).float().detach()
                        a = net.get_action(x, pre) if mask else net.get_action(x)
                    if mask:
                        pre = a
                    acts[i, j, t - 1] = a
                    state, _, done, _ = env.step(a)
                    if done:
                        solved_at[i, j] = t
                        break
        out["actions_mask" if mask else "actions"] = acts
        out["solved_at_mask" if mask else "solved_at"] = solved_at
    sd = {"sd_" + k: v.numpy() for k, v in net.state_dict().items()}
    np.savez_compressed(out_path("rollout_333.npz"), ks=np.array(ks), n_seeds=np.int64(n_seeds), T=np.int64(T), **out, **sd)
    print("rollout: solved", int((out["solved_at"] > 0).sum()), "masked", int((out["solved_at_mask"] > 0).sum()))


def golden_adi_deepcube(torch, env):
    """G10: get_random_samples (cube_env.py:177-252) with the reference's own DeepCube as the model."""
    import model as ref_model

    torch.manual_seed(11)
    net = ref_model.DeepCube([20, 24], 12, [64, 32, 16]).eval()
    n_cubes, depth, temperature, seed = 16, 12, 0.7, 555
    buf = []
    np.random.seed(seed)
    env.get_random_samples(buf, net, depth, n_cubes, temperature)
    shape = (n_cubes, depth)
    sd = {"sd_" + k: v.numpy() for k, v in net.state_dict().items()}
    np.savez_compressed(
        out_path("adi_deepcube_333.npz"), seed=np.int64(seed), temperature=np.float64(temperature),
        cols=np.stack([cols_of(b["state"]) for b in buf]).reshape(*shape, 20),
        target_value=np.array([b["target_value"] for b in buf], np.float64).reshape(shape),
        target_policy=np.array([b["target_policy"] for b in buf], np.int64).reshape(shape),
        error=np.array([b["error"] for b in buf], np.float64).reshape(shape), **sd)
    print("adi_deepcube:", len(buf), "samples")


def golden_replay(torch, env):
    """G11: the reference's ReplayBuffer (utils.py:203-270) fed by get_random_samples with G5's stub model: eviction at
    maxlen, prioritised sampling on the global legacy RNG, __getitem__ 5-tuples, update(), and the DataLoader order of
    update_params (utils.py:296-303)."""
    import utils as ref_utils
    from torch.utils.data import DataLoader

    g5 = np.load(earlier_fixture("adi_333.npz"))
    w_lin, b_lin = torch.tensor(g5["w"]), torch.tensor(g5["b"])

    class StubModel(torch.nn.Module):
        def forward(self, x):
            if x.dim() == 2:
                x = x.unsqueeze(0)
            return (x.reshape(x.shape[0], -1) @ w_lin + b_lin).unsqueeze(-1), torch.zeros(x.shape[0], 12)

    n_cubes, depth, temperature = 64, 30, float(g5["temperature"])
    buf_size, sample_size = 1500, 256
    rb = ref_utils.ReplayBuffer(buf_size, sample_size)
    np.random.seed(int(g5["seed"]))
    env.get_random_samples(rb, StubModel(), depth, n_cubes, temperature)          # 1920 samples into a deque of 1500
    assert len(rb.memory) == buf_size
    mem_cols = np.stack([cols_of(m["state"]) for m in rb.memory])
    mem_err = np.array(rb.error_memory, np.float64)
    out = {}
    np.random.seed(31337)
    rb.get_prioritized_sample()
    out["idx1"] = np.asarray(rb.prioritized_idx, np.int64)
    items = [rb[i] for i in range(len(rb))]
    out["item_dtypes"] = np.array([str(t.dtype) for t in items[0]])
    out["item_cols"] = np.stack([cols_of(it[0].numpy()) for it in items])
    out["item_tv"] = np.array([it[1].item() for it in items], np.float32)
    out["item_tp"] = np.array([it[2].item() for it in items], np.int64)
    out["item_sc"] = np.array([it[3].item() for it in items], np.int64)
    out["item_idx"] = np.array([it[4].item() for it in items], np.int64)
    # the mini-batch order update_params sees: DataLoader(replay_buffer, batch_size, shuffle=True) under a torch seed
    torch.manual_seed(4242)
    out["loader_idx"] = np.concatenate([b[4].numpy() for b in DataLoader(rb, batch_size=100, shuffle=True)])
    # update() as update_params does, then a second prioritised draw
    upd_idx = out["idx1"][::3]
    upd_err = (np.arange(len(upd_idx), dtype=np.float64) % 7 + 1) * 0.03125
    for i, e in zip(upd_idx, upd_err):
        rb.update(int(i), float(e))
    np.random.seed(99)
    rb.get_prioritized_sample()
    out["idx2"] = np.asarray(rb.prioritized_idx, np.int64)
    # more samples arrive (the deque drops its oldest 300), third draw
    np.random.seed(7)
    env.get_random_samples(rb, StubModel(), depth, 10, temperature)
    np.random.seed(123)
    rb.get_prioritized_sample()
    out["idx3"] = np.asarray(rb.prioritized_idx, np.int64)
    out["mem3_cols"] = np.stack([cols_of(m["state"]) for m in rb.memory])
    out["mem3_err"] = np.array(rb.error_memory, np.float64)
    # a buffer that is not full: every index, in order
    small = ref_utils.ReplayBuffer(5000, 4000)
    np.random.seed(5)
    env.get_random_samples(small, StubModel(), 5, 8, temperature)
    small.get_prioritized_sample()
    out["small_idx"] = np.asarray(small.prioritized_idx, np.int64)
    np.savez_compressed(out_path("replay_333.npz"), buf_size=np.int64(buf_size), sample_size=np.int64(sample_size),
                        mem_cols=mem_cols, mem_err=mem_err, upd_idx=upd_idx, upd_err=upd_err, **out)
    print("replay:", len(out["idx1"]), "prioritised of", buf_size, "dtypes", list(out["item_dtypes"]))


def golden_mcts_guided(torch, cube_env, env):
    """G12: the reference's MCTS (mcts.py:36-154) with a GUIDING stub so that deep searches succeed and the found-solution path
    is exercised: a table of every state within 4 moves of the solved cube (breadth-first with the reference env itself:
    one-hot columns -> distance d and the move that leads back towards solved).  predict(state) = (value -d, policy softmax
    with logit 2 on that move) for tabled states, (-9, uniform) otherwise.  For seeds 0..11 x k in {3, 4, 5}: simulations
    used, the returned action list, and the statistics (visits, values, virtual losses) of EVERY node on the returned path."""
    import random

    import mcts as ref_mcts

    env.init_state()
    start = env.sim_cube.copy()
    table = {tuple(cols_of(env.cube)): (0, 0)}
    frontier = [start]
    for depth in range(1, 5):
        nxt = []
        for st in frontier:
            for a in range(12):
                env.sim_cube = st.copy()
                oh, _, _, _ = env.step(a)
                key = tuple(cols_of(oh))
                if key not in table:
                    table[key] = (depth, a ^ 1)                       # X' undoes X: actions come in (2f, 2f+1) pairs
                    nxt.append(env.sim_cube.copy())
        frontier = nxt
    t_cols = np.array(list(table.keys()), np.uint8)
    t_depth = np.array([v[0] for v in table.values()], np.uint8)
    t_back = np.array([v[1] for v in table.values()], np.uint8)

    class Stub:
        def predict(self, x):
            hit = table.get(tuple(cols_of(np.asarray(x))))
            logits = np.zeros(12, np.float32)
            value = np.float32(-9.0)
            if hit is not None:
                value = np.float32(-float(hit[0]))
                logits[hit[1]] = 2.0
            e = np.exp(logits - logits.max())
            return np.array([value], np.float32), (e / e.sum()).astype(np.float32)

    cfg = {"mcts": {"virtual_loss_const": 150, "cpuct": 1.0, "value_min": -10.0, "numMCTSSim": 50}, "test": {"cube_size": 3}}
    cases = [(s, k) for k in (3, 4, 5) for s in range(12)]
    LMAX = 24
    sims, sol, p_vis, p_val, p_vl, p_len = [], [], [], [], [], []
    for seed, k in cases:
        state = env.reset(seed=seed, scramble_count=k)
        random.seed(5000 + 13 * seed + k)
        tree = ref_mcts.MCTS(Stub(), cfg)
        found, used = None, 0
        for i in range(60):
            used = i + 1
            found = tree.train(state, env)
            if found is not None:
                break
        sims.append(used)
        a = np.full(LMAX, 255, np.uint8)
        vis, val, vl = np.zeros((LMAX, 12), np.int64), np.zeros((LMAX, 12), np.float64), np.zeros((LMAX, 12), np.float64)
        n_nodes = 0
        if found is not None:
            assert len(found) <= LMAX
            a[:len(found)] = found
            key = np.array2string(state)
            for t in range(len(found)):                               # every node from the root to the expanded leaf
                node = tree.children_and_data[key]
                vis[t] = node[tree.n_of_v_i]
                val[t] = [float(np.asarray(v).reshape(-1)[0]) for v in node[tree.s_i]]
                vl[t] = node[tree.v_l_i]
                key = node[tree.ch_i][found[t]]
                n_nodes += 1
        sol.append(a); p_vis.append(vis); p_val.append(val); p_vl.append(vl); p_len.append(n_nodes)
    np.savez_compressed(
        out_path("mcts_guided_333.npz"),
        table_cols=t_cols, table_depth=t_depth, table_back=t_back,
        seeds=np.array([c[0] for c in cases]), ks=np.array([c[1] for c in cases]),
        random_seed=np.array([5000 + 13 * s + k for s, k in cases]), sims=np.array(sims), solution=np.stack(sol),
        path_nodes=np.array(p_len), path_visits=np.stack(p_vis), path_values=np.stack(p_val), path_vloss=np.stack(p_vl))
    found_by_k = {k: int(sum(1 for (s_, k_), n in zip(cases, p_len) if k_ == k and n)) for k in (3, 4, 5)}
    print("mcts_guided: table", len(table), "found per depth", found_by_k, "sims", sims, "path lengths", p_len)


def golden_mcts(torch, cube_env, env):
    """G8: run the reference's own MCTS class with a deterministic stub model."""
    import random

    import mcts as ref_mcts

    rng = np.random.default_rng(4321)
    wv = (rng.standard_normal(480) * 0.05).astype(np.float32)
    wp = (rng.standard_normal((480, 12)) * 0.3).astype(np.float32)

    class Stub:
        """predict(state) -> (value[1], softmax policy[12]) as model.py:78-91 returns them."""

        def predict(self, x):
            f = np.asarray(x, dtype=np.float32).reshape(-1)
            logits = f @ wp
            e = np.exp(logits - logits.max())
            return np.array([f @ wv], np.float32), (e / e.sum()).astype(np.float32)

    cfg = {"mcts": {"virtual_loss_const": 150, "cpuct": 1.0, "value_min": -10.0, "numMCTSSim": 50}, "test": {"cube_size": 3}}
    cases = [(s, k) for k in (1, 2, 3, 4) for s in range(6)]
    sims, sol, root_visits, root_values = [], [], [], []
    for seed, k in cases:
        state = env.reset(seed=seed, scramble_count=k)
        random.seed(1000 + 17 * seed + k)
        tree = ref_mcts.MCTS(Stub(), cfg)
        found, used = None, 0
        for i in range(60):
            used = i + 1
            found = tree.train(state, env)
            if found is not None:
                break
        sims.append(used)
        a = np.full(16, 255, np.uint8)
        if found is not None:
            a[:len(found)] = found
        sol.append(a)
        root = tree.children_and_data[np.array2string(state)]
        root_visits.append(np.array(root[tree.n_of_v_i], np.int64))
        root_values.append(np.array([float(np.asarray(v).reshape(-1)[0]) for v in root[tree.s_i]], np.float64))
    seeds_long = np.arange(20)
    long_st = np.zeros((20, 54), np.uint8)
    for i, s in enumerate(seeds_long):
        env.reset(seed=int(s), scramble_count=1000)
        long_st[i] = env.sim_cube
    np.savez_compressed(
        out_path("mcts_333.npz"),
        wv=wv, wp=wp, seeds=np.array([c[0] for c in cases]), ks=np.array([c[1] for c in cases]),
        random_seed=np.array([1000 + 17 * s + k for s, k in cases]), sims=np.array(sims), solution=np.stack(sol),
        root_visits=np.stack(root_visits), root_values=np.stack(root_values),
        long_seeds=seeds_long, long_k=np.int64(1000), long_stickers=long_st,
    )
    print("mcts:", list(zip(cases, sims)))


def cols_222(onehot):
    """Column index of the single 1.0 in each of the 7 cubelet rows of the 2x2x2 one-hot (cube_env.py:143-147: row = cubelet,
    column = position * 3 + orientation)."""
    oh = np.asarray(onehot)
    assert oh.shape == (7, 21) and oh.dtype == np.float64 and (oh.sum(1) == 1).all() and set(np.unique(oh)) <= {0.0, 1.0}
    return np.argmax(oh, 1).astype(np.uint8)


def golden_env222(torch, cube_env):
    """G13: the reference's own CubeEnv / MCTS with cube_size = 2.  cube_env.py binds the six py222 names at import time (to the
    harness stub's None); here they are re-bound, in the imported module's namespace, to tests/golden/py222_standin.py."""
    import random

    import mcts as ref_mcts
    import py222_standin as standin

    for name in ("initState", "getOP", "doMove", "isSolved", "getStickers", "printCube"):
        setattr(cube_env, name, getattr(standin, name))
    dev = torch.device("cpu")
    env = cube_env.CubeEnv(dev, cube_size=2)
    assert env.state_dim == [7, 21] and env.action_dim == 6 and env.action_to_sim_action[2] == ["U", "U'", "F", "F'", "R", "R'"]
    out = {"action_names": np.array(env.action_to_sim_action[2]), "solved_stickers": np.asarray(env.sim_cube, np.uint8), "solved_cols": cols_222(env.cube),
           "state_dtype": np.array(str(env.cube.dtype))}
    # reset(seed, k): the legacy global generator is saved, seeded, used for randint(6, size=k) and restored (cube_env.py:62-68)
    seeds, ks = np.arange(0, 100, 10), np.arange(1, 15)
    r_st, r_cols = np.zeros((len(seeds), len(ks), 24), np.uint8), np.zeros((len(seeds), len(ks), 7), np.uint8)
    np.random.seed(424242)
    before = np.random.get_state()[1].copy()
    for i, sd in enumerate(seeds):
        for j, k in enumerate(ks):
            state = env.reset(seed=int(sd), scramble_count=int(k))
            r_st[i, j], r_cols[i, j] = env.sim_cube, cols_222(state)
    assert (np.random.get_state()[1] == before).all()
    out.update(reset_seeds=seeds, reset_ks=ks, reset_stickers=r_st, reset_cols=r_cols)
    # 300 random walks of 14 moves, step by step; every 5th state through state_to_sim_state (cube_env.py:154-175, getStickers)
    W, D = 300, 14
    acts = np.random.default_rng(222).integers(0, 6, (W, D), dtype=np.uint8)
    w_st, w_cols = np.zeros((W, D, 24), np.uint8), np.zeros((W, D, 7), np.uint8)
    w_done, w_rew, rt = np.zeros((W, D), np.uint8), np.zeros((W, D), np.float64), np.zeros((W, D // 5 + 1, 24), np.uint8)
    for w in range(W):
        env.init_state()
        for d in range(D):
            state, reward, done, info = env.step(int(acts[w, d]))
            assert isinstance(reward, float) and isinstance(done, bool) and info == {} and state is env.cube
            w_st[w, d], w_cols[w, d], w_done[w, d], w_rew[w, d] = env.sim_cube, cols_222(state), done, reward
            if d % 5 == 0:
                rt[w, d // 5] = env.state_to_sim_state(env.cube)
    out.update(walk_actions=acts, walk_stickers=w_st, walk_cols=w_cols, walk_done=w_done, walk_reward=w_rew, roundtrip_stickers=rt)
    # get_random_samples / get_target_value with a deterministic linear stub (value = <one-hot, w> + b), incl. solved-child breaks at depth 1
    rng = np.random.default_rng(2222)
    w_lin, b_lin = torch.tensor((rng.standard_normal(147) * 0.3).astype(np.float32)), torch.tensor(np.float32(0.125))

    class StubModel(torch.nn.Module):
        def forward(self, x):
            if x.dim() == 2:
                x = x.unsqueeze(0)
            return (x.reshape(x.shape[0], -1) @ w_lin + b_lin).unsqueeze(-1), torch.zeros(x.shape[0], 6)

    n_cubes, depth, temperature, seed = 48, 10, 0.7, 777
    buf = []
    np.random.seed(seed)
    env.get_random_samples(buf, StubModel(), depth, n_cubes, temperature)
    np.random.seed(seed)
    adi_actions = np.stack([np.random.randint(6, size=depth) for _ in range(n_cubes)]).astype(np.uint8)   # the draws the call made (cube_env.py:189)
    shape = (n_cubes, depth)
    assert len(buf) == n_cubes * depth and list(buf[0].keys()) == ["state", "target_value", "target_policy", "scramble_count", "error"]
    out.update(adi_w=w_lin.numpy(), adi_b=b_lin.numpy(), adi_seed=np.int64(seed), adi_temperature=np.float64(temperature), adi_actions=adi_actions,
               adi_cols=np.stack([cols_222(b["state"]) for b in buf]).reshape(*shape, 7),
               adi_target_value=np.array([b["target_value"] for b in buf], np.float64).reshape(shape),
               adi_target_policy=np.array([b["target_policy"] for b in buf], np.int64).reshape(shape),
               adi_scramble_count=np.array([b["scramble_count"] for b in buf], np.int64).reshape(shape),
               adi_error=np.array([b["error"] for b in buf], np.float64).reshape(shape),
               adi_final_stickers=np.asarray(env.sim_cube, np.uint8))
    import utils as ref_utils                                          # the reference's ReplayBuffer over those records: the 5-tuple's dtypes
    rb = ref_utils.ReplayBuffer(10_000, 5_000)
    for b in buf:
        rb.append(b)
    rb.get_prioritized_sample()
    out["adi_item_dtypes"] = np.array([str(t.dtype) for t in rb[0]])
    # the reference's MCTS on the 2x2x2 (action_dim 6 from cfg['test']['cube_size'], mcts.py:33-34)
    wv = (rng.standard_normal(147) * 0.05).astype(np.float32)
    wp = (rng.standard_normal((147, 6)) * 0.3).astype(np.float32)

    class Stub:
        def predict(self, x):
            f = np.asarray(x, dtype=np.float32).reshape(-1)
            logits = f @ wp
            e = np.exp(logits - logits.max())
            return np.array([f @ wv], np.float32), (e / e.sum()).astype(np.float32)

    cfg = {"mcts": {"virtual_loss_const": 150, "cpuct": 1.0, "value_min": -10.0, "numMCTSSim": 50}, "test": {"cube_size": 2}}
    cases = [(s, k) for k in (1, 2, 3, 4, 5) for s in range(6)]
    sims, sol, root_visits, root_values = [], [], [], []
    for sd, k in cases:
        state = env.reset(seed=sd, scramble_count=k)
        random.seed(2000 + 17 * sd + k)
        tree = ref_mcts.MCTS(Stub(), cfg)
        found, used = None, 0
        for i in range(60):
            used = i + 1
            found = tree.train(state, env)
            if found is not None:
                break
        sims.append(used)
        a = np.full(16, 255, np.uint8)
        if found is not None:
            a[:len(found)] = found
        sol.append(a)
        root = tree.children_and_data[np.array2string(state)]
        root_visits.append(np.array(root[tree.n_of_v_i], np.int64))
        root_values.append(np.array([float(np.asarray(v).reshape(-1)[0]) for v in root[tree.s_i]], np.float64))
    out.update(mcts_wv=wv, mcts_wp=wp, mcts_seeds=np.array([c[0] for c in cases]), mcts_ks=np.array([c[1] for c in cases]),
               mcts_random_seed=np.array([2000 + 17 * s_ + k for s_, k in cases]), mcts_sims=np.array(sims), mcts_solution=np.stack(sol),
               mcts_root_visits=np.stack(root_visits), mcts_root_values=np.stack(root_values))
    np.savez_compressed(out_path("env222_via_reference.npz"), **out)
    print("env222: walks solved", int(w_done.sum()), "adi solved-child samples", int((out["adi_target_value"] == 1.0).sum()),
          "mcts sims", sims)


def compare(dir_a, dir_b, names):
    """Array-by-array comparison (keys, dtype, shape, values) of the named fixtures in two directories."""
    bad = 0
    for name in names:
        fa, fb = os.path.join(dir_a, name), os.path.join(dir_b, name)
        if not (os.path.exists(fa) and os.path.exists(fb)):
            print(f"{name}: MISSING ({'regenerated' if not os.path.exists(fa) else 'committed'} file absent)")
            bad += 1
            continue
        a, b = np.load(fa), np.load(fb)
        diffs = [k for k in sorted(set(a.files) | set(b.files))
                 if k not in a.files or k not in b.files or a[k].dtype != b[k].dtype or a[k].shape != b[k].shape
                 or not np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f")]
        print(f"{name}: {'IDENTICAL' if not diffs else 'DIFFERENT in ' + ', '.join(diffs)} ({len(a.files)} arrays)")
        bad += bool(diffs)
    return bad


def main():
    import argparse
    import tempfile
    global OUT
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("groups", nargs="*", help=f"fixture groups to (re)generate, default all: {' '.join(FIXTURES)}")
    ap.add_argument("--out", default=None, help="directory to write into (default: tests/golden itself)")
    ap.add_argument("--check", action="store_true",
                    help="regenerate into a temporary directory and compare every array (dtype, shape, values) with the committed fixtures; "
                         "exit status 1 on any difference")
    args = ap.parse_args()
    unknown = set(args.groups) - set(FIXTURES)
    if unknown:
        sys.exit(f"unknown groups: {sorted(unknown)}")
    if not os.path.isdir(REF):
        sys.exit(f"{REF} is not here: the fixtures can only be regenerated in the build container")
    if args.check:
        with tempfile.TemporaryDirectory(prefix="golden_check_") as tmp:
            OUT = tmp
            generate(set(args.groups))
            base = {"tables", "walks", "reset", "adi", "expand", "encode"}
            which = set(args.groups) or set(FIXTURES)
            if which & base:
                which |= base                                      # G1-G7 are regenerated together
            names = [fixture_file(g) for g in FIXTURES if g in which]
            bad = compare(tmp, HERE, names)
        print(f"{len(names) - bad} of {len(names)} fixtures IDENTICAL to the committed ones")
        sys.exit(1 if bad else 0)
    if args.out:
        os.makedirs(args.out, exist_ok=True)
        OUT = args.out
    generate(set(args.groups))


if __name__ == "__main__":
    main()
