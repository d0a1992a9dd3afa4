"""Round 5: the ADI pipeline as ONE generator launch + one net forward per GROUP of depths (adi.AdiPlan), the multi-depth forms of the
family writer and of the target assembly, the hipGraph path, and the two forms of the legacy-numpy generator.  Every check is against
the CPU oracle, numpy itself, or the committed reference fixtures."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def ops():
    from rubiks_cube_solver_amd import ops as o
    return o


@pytest.fixture(scope="module")
def L():
    from rubiks_cube_solver_amd import _lib
    return _lib


def _linear_model(cs, device="cuda", seed=0):
    """value = <one-hot, w> + b as a batched torch callable (exact in float32 up to summation order)."""
    R, C = (20, 24) if cs == 3 else (7, 21)
    A = 12 if cs == 3 else 6
    rng = np.random.default_rng(seed)
    w = torch.tensor(rng.standard_normal(R * C).astype(np.float32), device=device)
    b = 0.25

    def model(x):
        if x.dim() == 2:
            x = x.unsqueeze(0)
        return (x.reshape(x.shape[0], -1).float() @ w + b).unsqueeze(-1), torch.zeros(x.shape[0], A, device=x.device)
    return model, w.cpu().numpy().astype(np.float64).reshape(R, C), b


def _onehot_value(code, w, cs):
    """<one-hot(code), w> on the host: 3x3x3 row = slot, column = code; 2x2x2 row = code // 3, column = slot * 3 + code % 3."""
    code = code.astype(np.int64)
    if cs == 3:
        return w[np.arange(20), code].sum(-1)
    return w[code // 3, np.arange(7) * 3 + code % 3].sum(-1)


def _expected(oracle, cs, W, D, T, w, b, **adi_kw):
    exp = oracle.adi(cs, W, D, want_children=False, threads=4, **adi_kw)
    v_child = _onehot_value(exp["child_code"], w, cs) + b - 1.0                        # [W, D, A]
    solved = exp["child_solved"].astype(bool)
    tv = np.where(solved.any(-1), 1.0, v_child.max(-1))
    tp = np.where(solved.any(-1), np.argmax(solved, -1), np.argmax(v_child, -1))
    srt = np.sort(v_child, -1)
    tie = (srt[..., -1] - srt[..., -2] < 1e-4) & ~solved.any(-1)
    v_par = _onehot_value(exp["parent_code"], w, cs) + b
    err = np.abs(v_par - tv) * np.arange(1, D + 1, dtype=np.float64)[None, :] ** -T
    return exp, tv, tp, tie, err


def _check(res, exp, tv, tp, tie, err, D):
    assert (res["actions"].cpu().numpy() == exp["actions"]).all()
    assert (res["state_code"].cpu().numpy() == exp["parent_code"]).all()
    assert np.allclose(res["target_value"].cpu().numpy(), tv, atol=2e-5)
    got = res["target_policy"].cpu().numpy()
    assert ((got == tp) | tie).all() and (got == tp).mean() > 0.99
    assert np.allclose(res["error"].cpu().numpy(), err, atol=2e-5)
    assert (res["scramble_count"].cpu().numpy() == np.arange(1, D + 1)[None, :]).all()


@pytest.mark.parametrize("cs", [3, 2])
def test_adi_plan_groups_chunks_and_graph(oracle, cs):
    """The same samples whatever the grouping: every depth in one forward, one depth per forward, several groups, several chunks of
    walks, and the hipGraph replay (two runs with different seeds through ONE captured graph)."""
    from rubiks_cube_solver_amd.adi import AdiPlan, adi_samples
    model, w, b = _linear_model(cs)
    T = 0.7
    R, C = (20, 24) if cs == 3 else (7, 21)
    A1 = 13 if cs == 3 else 7
    row = R * C * 4
    cases = [(333, 11, 1 << 30, "one group"), (333, 11, A1 * row * 336, "one depth per forward"), (333, 11, 4 * A1 * row * 336 + 5, "groups of 4 + a rest of 3"),
             (2500, 5, A1 * row * 1100, "chunks of 1024 walks + a rest"), (20000, 3, 1 << 30, "several generator tiles")]
    for W, D, budget, what in cases:
        exp = _expected(oracle, cs, W, D, T, w, b, seed=5, stream=3)
        res = adi_samples(model, cs, W, D, T, device="cuda", seed=5, stream_id=3, dense_budget_bytes=budget, want_state_dense=W < 1000)
        _check(res, *exp, D)
        if W < 1000:
            oh = res["state"]
            assert tuple(oh.shape) == (W, D, R, C) and int(oh.sum()) == W * D * (20 if cs == 3 else 7)
            code = exp[0]["parent_code"].astype(np.int64)
            want = np.zeros((W, D, R, C), np.uint8)
            if cs == 3:
                np.put_along_axis(want, code[..., None], 1, -1)
            else:
                wi, di, si = np.meshgrid(np.arange(W), np.arange(D), np.arange(7), indexing="ij")
                want[wi, di, code // 3, si * 3 + code % 3] = 1
            assert (oh.cpu().numpy() == want).all(), what
    # the plan object: geometry, then eager and graph runs over the same static buffers
    plan = AdiPlan(model, cs, 333, 11, T, dense_budget_bytes=4 * A1 * row * 336 + 5)
    assert plan.group == 4 and plan.chunk == 333 and len(plan.chunks) == 1     # packed blocks for both sizes (round 6: 2x2x2 blocks were padded to 512 walks)
    assert plan.chunks[0][2]["bs"] == 336 and plan.dense.shape[0] == 4 * A1 * 336
    plan = AdiPlan(model, cs, 2500, 5, T, dense_budget_bytes=A1 * row * 1100)
    assert plan.chunk == 1024 and [c[1] for c in plan.chunks] == [1024, 1024, 452]
    gplan = AdiPlan(model, cs, 777, 9, T, graph=True)
    for seed in (21, 22, 23):                                          # run 1 captures, runs 2 and 3 replay
        res = gplan.run(seed=seed, stream_id=1)
        _check(res, *_expected(oracle, cs, 777, 9, T, w, b, seed=seed, stream=1), 9)
    assert len(gplan._graphs) == 1
    acts = np.random.default_rng(9).integers(0, 12 if cs == 3 else 6, (777, 9), dtype=np.uint8)
    res = gplan.run(actions=acts)                                      # replayed moves through the same graph (host upload outside it)
    _check(res, *_expected(oracle, cs, 777, 9, T, w, b, actions_in=acts), 9)
    again = adi_samples(model, cs, 777, 9, T, graph=True, actions=torch.from_numpy(acts).cuda())
    again2 = adi_samples(model, cs, 777, 9, T, graph=True, actions=acts)
    for k in ("state_code", "target_value", "target_policy", "error", "actions"):
        assert torch.equal(again[k], again2[k]) and torch.equal(again[k], res[k])
    assert again["error"].data_ptr() != again2["error"].data_ptr()    # adi_samples hands out copies, not the plan's buffers
    from rubiks_cube_solver_amd import adi
    assert len(adi._plans) == 1
    adi.release_plans()
    with pytest.raises(ValueError):
        AdiPlan(model, cs, 10, 2, T, model_device="cpu", graph=True)
    empty = adi_samples(model, cs, 0, 4, T)
    assert tuple(empty["target_value"].shape) == (0, 4) and tuple(empty["state_code"].shape) == (0, 4, 20 if cs == 3 else 7)


def test_family_and_targets_depth_groups_match_single_depth_launches(ops, L, oracle):
    """rc_onehot_from_family_depths / rc_adi_targets_depths against their one-depth forms (which earlier tests pin to the oracle):
    identical bytes, for packed block strides, every dense format, one tile and several tiles."""
    for n, D, bs in ((333, 6, 336), (5000, 3, 5000), (40000, 2, 40008)):
        pitch, bufs = ops.adi_buffers(n, D, 3, "cuda", family=True)
        ops.adi_generate(n, D, 3, pitch, "cuda", seed=3, stream_id=1, **bufs)
        for dt in (torch.float32, torch.bfloat16, torch.float16, torch.uint8):
            one = torch.zeros((D, 13 * bs, 20, 24), dtype=dt, device="cuda")
            for d in range(D):
                ops.onehot_from_family(bufs["family"][d], n, 3, one[d], block_stride=bs)
            many = torch.zeros((D * 13 * bs, 20, 24), dtype=dt, device="cuda")
            ops.onehot_from_family(bufs["family"], n, 3, many, block_stride=bs, n_depths=D)
            assert torch.equal(many.view(D, 13 * bs, 20, 24), one), (n, dt)
            part = torch.zeros(((D - 1) * 13 * bs, 20, 24), dtype=dt, device="cuda")          # a sub-range of depths
            ops.onehot_from_family(bufs["family"][1:], n, 3, part, block_stride=bs, n_depths=D - 1)
            assert torch.equal(part.view(D - 1, 13 * bs, 20, 24), one[1:])
        # parent block == the oracle's parent code, every depth
        exp = oracle.adi(3, n, D, seed=3, stream=1, want_children=False, threads=4)
        got = many.view(D, 13, bs, 20, 24)[:, 12, :n].argmax(-1).cpu().numpy()
        assert (got.transpose(1, 0, 2) == exp["parent_code"]).all()
        # targets: random values laid out as the net's output [D][13][bs]
        g = torch.Generator(device="cuda").manual_seed(n)
        v = torch.randn(D * 13 * bs, generator=g, device="cuda")
        wgt = torch.tensor([float(d + 1) ** -0.5 for d in range(D)], dtype=torch.float64, device="cuda")
        tv = torch.zeros((n + 3, D + 2), dtype=torch.float32, device="cuda")
        tp = torch.zeros((n + 3, D + 2), dtype=torch.int32, device="cuda")
        er = torch.zeros((n + 3, D + 2), dtype=torch.float64, device="cuda")
        ops.adi_targets_depths(v, 13 * bs, bs, bufs["child_solved"], v[12 * bs:], 13 * bs, wgt, n, D, 3, tv[3:, 1:], tp[3:, 1:], er[3:, 1:])
        vv = v.view(D, 13, bs)
        for d in range(D):
            cv = torch.zeros((12, bufs["child_solved"].shape[-1]), dtype=torch.float32, device="cuda")
            cv[:, :n] = vv[d, :12, :n]
            a, b_, c = ops.adi_targets(cv, bufs["child_solved"][d], n, 3, vv[d, 12, :n].contiguous(), torch.full((n,), float(wgt[d]), dtype=torch.float64, device="cuda"))
            assert torch.equal(tv[3:3 + n, 1 + d], a) and torch.equal(tp[3:3 + n, 1 + d], b_) and torch.equal(er[3:3 + n, 1 + d], c)
        assert float(tv[:3].abs().sum()) == 0 and float(tv[:, 0].abs().sum()) == 0 and float(tv[:, D + 1].abs().sum()) == 0   # nothing outside the view
    with pytest.raises(L.RubikHipError):
        ops.onehot_from_family(bufs["family"], n, 3, many[:100], block_stride=bs, n_depths=D)
    # a record with MORE tiles per depth than the n walks converted (the first n < W walks of a W-walk record) is refused: the library
    # derives the per-depth source stride from n, so depths >= 1 would be read from the wrong offsets
    few = 20000                                                        # 2 of the 3 tiles of the 40000-walk record
    with pytest.raises(L.RubikHipError, match="tiles per depth"):
        ops.onehot_from_family(bufs["family"], few, 3, many, block_stride=bs, n_depths=D)
    sliced = bufs["family"][:, :2].contiguous()
    ops.onehot_from_family(sliced, few, 3, many, block_stride=bs, n_depths=D)
    assert (many.view(D, 13, bs, 20, 24)[:, 12, :few].argmax(-1).cpu().numpy().transpose(1, 0, 2) == exp["parent_code"][:few]).all()
    with pytest.raises(L.RubikHipError):
        ops.adi_targets_depths(v[:10], 13 * bs, bs, bufs["child_solved"], v[12 * bs:], 13 * bs, wgt, n, D, 3, tv[3:, 1:], tp[3:, 1:], er[3:, 1:])


@pytest.mark.parametrize("cs", [2, 3])
def test_onehot_from_code_blocks_packs_equally_tiled_buffers(ops, L, oracle, cs):
    """rc_onehot_from_code_blocks (round 6): the A child-code buffers of several depths of ONE rc_adi_generate call, each [tile][SLOTS][pitch],
    become packed dense blocks in one launch -- every block against the oracle's child codes, every dense format, one tile and several, pad
    rows between blocks untouched.  (cube_env.py:143-147 / py333.py:235-246 define the one-hot; the 2x2x2 ADI plan launches this.)"""
    A, SL, (R, C) = (6, 7, (7, 21)) if cs == 2 else (12, 20, (20, 24))
    for n, D in ((333, 3), (20001, 2)):
        pitch, bufs = ops.adi_buffers(n, D, cs, "cuda", parent_code=True, child_code=True)
        ops.adi_generate(n, D, cs, pitch, "cuda", seed=8, stream_id=2, **bufs)
        exp = oracle.adi(cs, n, D, seed=8, stream=2, want_children=False, threads=4)
        tiles = bufs["child_code"].shape[2]
        bs = -(-n // 16) * 16 + 16                                     # a stride with a whole pad chunk behind every block
        for dt in (torch.float32, torch.bfloat16, torch.float16, torch.uint8):
            x = torch.full((D * A * bs + D * bs, R, C), 7, dtype=dt, device="cuda")
            ops.onehot_from_code_blocks(bufs["child_code"].view(D * A, tiles, SL, pitch), n, cs, x[:D * A * bs], bs)
            ops.onehot_from_code_blocks(bufs["parent_code"], n, cs, x[D * A * bs:], bs)
            kids = x[:D * A * bs].view(D, A, bs, R, C)
            pars = x[D * A * bs:].view(D, bs, R, C)
            assert float(kids[:, :, n:].float().min()) == 7 and float(pars[:, n:].float().min()) == 7       # pad rows untouched
            assert float(kids[:, :, :n].float().sum()) == D * A * n * SL and float(pars[:, :n].float().sum()) == D * n * SL
            for got, want in ((kids[:, :, :n], exp["child_code"].transpose(1, 2, 0, 3)), (pars[:, :n], exp["parent_code"].transpose(1, 0, 2))):
                g = got.float().cpu().numpy()
                code = want.astype(np.int64)
                if cs == 3:
                    assert (g.argmax(-1) == code).all(), (n, dt)
                else:                                                  # row = piece = code // 3, column = slot * 3 + code % 3
                    idx = np.broadcast_to(np.arange(7), code.shape)
                    assert (np.take_along_axis(g.reshape(*g.shape[:-2], R * C), (code // 3) * C + idx * 3 + code % 3, -1) == 1).all(), (n, dt)
        one = torch.zeros((n, R, C), dtype=torch.float32, device="cuda")
        ops.onehot_from_code(bufs["parent_code"][1], n, cs, one)
        assert torch.equal(x[D * A * bs + bs:D * A * bs + bs + n].float(), one)                             # == the single-buffer entry point
    with pytest.raises(L.RubikHipError):
        ops.onehot_from_code_blocks(bufs["parent_code"], n, cs, x[:10], bs)
    with pytest.raises(L.RubikHipError):
        ops.onehot_from_code_blocks(bufs["parent_code"], n, cs, x, n - 1)
    lib, sp, P = L.lib(), L.stream_ptr(torch.device("cuda", torch.cuda.current_device())), lambda t: t.data_ptr()
    pc = bufs["parent_code"]
    assert lib.rc_onehot_from_code_blocks(P(pc), n, pitch, cs, P(x), L.FMT_F32, 0, pc.stride(0), bs, sp) != 0          # no blocks
    assert lib.rc_onehot_from_code_blocks(P(pc), n, pitch, cs, P(x), L.FMT_F32, 2, pc.stride(0) + 8, bs, sp) != 0      # misaligned source stride
    assert lib.rc_onehot_from_code_blocks(P(pc), n, pitch, cs, P(x), L.FMT_U8, 2, pc.stride(0), n if cs == 2 else n - 1, sp) != 0   # 20001 x 147 B blocks are not 16-byte aligned / stride < n
    assert lib.rc_onehot_from_code_blocks(P(pc), 0, pitch, cs, P(x), L.FMT_F32, 2, pc.stride(0), bs, sp) == 0          # nothing to do
    assert L.read_status() == 0


def _numpy_legacy(seeds, ks, A):
    out = []
    saved = np.random.get_state()
    try:
        for s, k in zip(seeds, ks):
            np.random.seed(int(s))
            out.append(np.random.randint(A, size=int(k)))
    finally:
        np.random.set_state(saved)
    return out


@pytest.mark.parametrize("cs", [3, 2])
def test_legacy_generator_forms_agree_with_numpy(ops, L, cs):
    """reset(seed, k)'s draws (cube_env.py:62-65) from both device forms of numpy's legacy generator: the LDS form (lazy twist), the
    streaming form at its default limit, at the boundaries of its three phases (227 / 454 outputs) and with a limit so low that most
    waves overflow and are redone by the fix-up launch -- every byte equal to numpy's, pad rows = the no-op."""
    A = 12 if cs == 3 else 6
    rng = np.random.default_rng(cs)
    n = 1500
    seeds = rng.integers(0, 2 ** 32, n, dtype=np.uint64)
    seeds[:4] = (0, 1, 2 ** 32 - 1, 10)
    for ks, label in ((np.full(n, 30), "k=30"), (rng.integers(0, 61, n), "mixed 0..60"), (np.r_[rng.integers(1, 40, n - 3), [170, 171, 200]], "up to 200"),
                      (np.r_[rng.integers(150, 420, n - 4), [455, 460, 470, 340]], "up to 470: all three phases of the streaming form, overflow into the fix-up")):
        want = _numpy_legacy(seeds, ks, A)
        kmax = int(ks.max())
        uniform = int(ks[0]) if (ks == ks[0]).all() else None
        for variant in (0, 1, 2, 2 + 16 * 40, 2 + 16 * 7, 2 + 16 * 227, 2 + 16 * 228, 2 + 16 * 454, 2 + 16 * 455, 2 + 16 * 623):
            buf, kk = ops.legacy_scramble_actions(torch.from_numpy(seeds.astype(np.int64)), cs, uniform if uniform is not None else ks.tolist(), device="cuda", variant=variant)
            got = buf.cpu().numpy()
            assert kk == kmax
            for i in range(n):
                assert (got[:ks[i], i] == want[i]).all(), (label, variant, i)
                assert (got[ks[i]:kmax, i] == A).all(), (label, variant, i)
    # more than one generation (624 outputs) and the k = 1000 scramble of test.py:279: the LDS form, chosen by the default rule
    ks = np.array([1000, 700, 469, 468, 5])
    want = _numpy_legacy(seeds[:5], ks, A)
    for variant in (0, 1):
        buf, _ = ops.legacy_scramble_actions(torch.from_numpy(seeds[:5].astype(np.int64)), cs, ks.tolist(), device="cuda", variant=variant)
        got = buf.cpu().numpy()
        for i in range(5):
            assert (got[:ks[i], i] == want[i]).all() and (got[ks[i]:, i] == A).all(), (variant, i)
    for bad in (3, 2 + 16 * 624, -1, 1 + 16):
        with pytest.raises(L.RubikHipError):
            ops.legacy_scramble_actions(torch.arange(4), cs, 3, device="cuda", variant=bad)


def test_legacy_generator_large_batch_matches_host_function(ops):
    """2^17 envs x k = 30 on the default route (streaming + fix-up) against vec_env.legacy_scramble_actions (numpy itself)."""
    from rubiks_cube_solver_amd.vec_env import legacy_scramble_actions
    n = 1 << 17
    seeds = np.random.default_rng(1).integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.int64)
    buf, _ = ops.legacy_scramble_actions(torch.from_numpy(seeds), 3, 30, device="cuda")
    want = legacy_scramble_actions(seeds[:20000], 30, 12)
    assert (buf[:, :20000].cpu().numpy().T == want).all()
    lds, _ = ops.legacy_scramble_actions(torch.from_numpy(seeds), 3, 30, device="cuda", variant=1)
    assert torch.equal(buf[:, :n], lds[:, :n])


def test_workspace_must_not_alias_operands(ops, L):
    """rc_apply_moves_ws / rc_encode_ws reject a workspace carved out of an operand instead of producing wrong rows."""
    n = 1 << 17
    st = ops.alloc_states(n, 3, "cuda")
    ops.fill_solved(st, n, 3)
    dst = torch.empty_like(st)
    acts = torch.zeros(n, dtype=torch.uint8, device="cuda")
    oh = torch.empty((n, 20, 24), dtype=torch.float32, device="cuda")
    need = L.lib().rc_workspace_bytes(L.OP_STEP, 3, n, L.FMT_F32)
    assert need > 0
    sp = L.stream_ptr(torch.device("cuda"))
    P = L.ptr
    L.init(torch.device("cuda", torch.cuda.current_device()))
    good = torch.empty(need, dtype=torch.uint8, device="cuda")
    assert L.lib().rc_apply_moves_ws(P(st), P(dst), P(acts), n, st.shape[2], dst.shape[2], 3, None, None, P(oh), L.FMT_F32, 0, P(good), need, sp) == 0
    for alias in (oh.view(torch.uint8).view(-1)[1024:], st.view(-1), dst.view(-1)[16:]):
        rc = L.lib().rc_apply_moves_ws(P(st), P(dst), P(acts), n, st.shape[2], dst.shape[2], 3, None, None, P(oh), L.FMT_F32, 0, P(alias), need, sp)
        assert rc == -1 and b"overlaps" in L.lib().rc_last_error()
    rc = L.lib().rc_encode_ws(P(st), n, st.shape[2], 3, P(oh), L.FMT_F32, 0, P(oh.view(torch.uint8).view(-1)[4096:]), need, sp)
    assert rc == -1 and b"overlaps" in L.lib().rc_last_error()
    torch.cuda.synchronize()


def test_facade_release_from_another_thread(L):
    """the alias cache of the batch-1 entry points is process-wide, so a release issued by whichever thread runs the
    garbage collector takes effect for the thread that used the buffer."""
    import threading

    import rubiks_cube_solver_amd as rc
    env = rc.make_env(torch.device("cpu"), 3)
    env.step(3)
    host = env._fast[2]
    t = threading.Thread(target=lambda: L.lib().rc_facade_release(host))
    t.start()
    t.join()
    s, r, d, _ = env.step(2)                        # re-validated, still correct
    ref = rc.make_env(torch.device("cpu"), 3)
    ref.step(3)
    s2, _, _, _ = ref.step(2)
    assert (s == s2).all()
    results = []

    def worker():                                   # a second thread stepping its own env through the shared table
        e = rc.make_env(torch.device("cpu"), 3)
        for a in (3, 2):
            out = e.step(a)
        results.append(out[0])
    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert (results[0] == s2).all()
    env.close()
    ref.close()


@pytest.mark.parametrize("cs", [3, 2])
def test_scramble_from_pinned_paths_and_search_pack(ops, L, oracle, cs):
    """The lockstep search's device step in pieces: rc_scramble_from replays no-op padded descents that it reads from PINNED HOST memory
    (rc_host_alias) out of untouched root states, and rc_search_pack lays the expansion's codes and flags out per root -- against the
    oracle and against the plain torch transposes round 4 used."""
    A, SL = (12, 20) if cs == 3 else (6, 7)
    for n, pitch in ((4096, None), (5000, 1024), (70000, None)):
        roots = ops.alloc_states(n, cs, "cuda", pitch)
        ops.fill_solved(roots, n, cs)
        ops.scramble(roots, n, cs, 9, seed=4, stream_id=2)
        snap = roots.clone()
        rng = np.random.default_rng(n)
        depth = 6
        lens = rng.integers(0, depth + 1, n)
        paths = rng.integers(0, A, (n, depth), dtype=np.uint8)
        paths[np.arange(depth)[None, :] >= lens[:, None]] = A                      # ragged descents, padded with the no-op
        host = torch.full((8, L.pitch_for(n)), A, dtype=torch.uint8).pin_memory()
        host.numpy()[:depth, :n] = paths.T
        work = torch.full_like(roots, 9)
        done = torch.empty(n, dtype=torch.uint8, device="cuda")
        ops.scramble(work, n, cs, depth, actions_in=host[:depth], src=roots, done=done)
        assert torch.equal(roots, snap)                                            # the roots are read only
        st = ops.to_aos(snap, n).cpu().numpy()
        for d in range(depth):
            live = paths[:, d] < A
            if live.any():
                st[live] = oracle.step(cs, st[live], paths[live, d], threads=4)[0]
        assert (ops.to_aos(work, n).cpu().numpy() == st).all()
        assert (done.cpu().numpy() == oracle.is_solved(cs, st)).all() if n <= 5000 else True
        dev_paths = host.cuda()
        work2 = torch.empty_like(roots)
        ops.scramble(work2, n, cs, depth, actions_in=dev_paths[:depth], src=roots)  # the same from device memory
        assert torch.equal(ops.to_aos(work2, n), ops.to_aos(work, n))               # (columns past n are never written)
        ops.scramble(work2, n, cs, 0, src=roots)                                    # depth 0: a copy
        assert torch.equal(ops.to_aos(work2, n), ops.to_aos(roots, n))
        # expansion of the leaves -> one record per root
        p = work.shape[-1]
        ex = ops.expand_buffers(n, cs, "cuda", p, children=False, codes=True)
        ops.expand_children(work, n, cs, None, ex["child_solved"], ex["child_code"], pitch=p)
        code = ops.alloc_code(n, cs, "cuda", p)
        ops.encode(work, n, cs, code, L.FMT_CODE)
        leaf = torch.zeros((n, SL), dtype=torch.uint8, device="cuda")
        child = torch.zeros((n, A, SL), dtype=torch.uint8, device="cuda")
        solved = torch.zeros((n, A), dtype=torch.uint8, device="cuda")
        ops.search_pack(code, ex["child_code"], ex["child_solved"], n, cs, leaf, child, solved)
        cc = ex["child_code"]
        assert torch.equal(leaf, ops.to_aos(code, n)) and torch.equal(solved, ex["child_solved"][:, :n].t().contiguous())
        assert torch.equal(child, cc.permute(1, 3, 0, 2).reshape(-1, A, SL)[:n].contiguous())
        _, e_cc, e_cs = oracle.expand(cs, st[:3000], threads=4)
        assert (child[:3000].cpu().numpy() == e_cc).all() and (solved[:3000].cpu().numpy() == e_cs).all()
    with pytest.raises(L.RubikHipError):
        ops.scramble(work, n, cs, depth, actions_in=torch.zeros((depth, L.pitch_for(n)), dtype=torch.uint8), src=roots)   # pageable host memory
    with pytest.raises(L.RubikHipError):
        ops.scramble(work, n, cs, depth, actions_in=host[:depth], src=roots[:1])
    with pytest.raises(L.RubikHipError):
        ops.search_pack(code, ex["child_code"], ex["child_solved"], n, cs, leaf, child[:, :3], solved)


def test_workspace_cache_is_locked_bounded_and_releasable(ops, L, oracle):
    """two host threads driving ONE stream through the workspace route (step + code, then the front writer) get their own
    results -- the cache look-up and both launches of a call are one critical section -- and the cache keeps at most _WS_MAX entries."""
    import threading
    n, cs = (1 << 17) + 3, 3
    rng = np.random.default_rng(5)
    jobs = []
    for t in range(2):
        st = oracle.adi(cs, n, 5, seed=40 + t, stream=t, want_children=False, threads=4)["parents"][:, -1]
        acts = rng.integers(0, 12, n, dtype=np.uint8)
        jobs.append((st, acts, ops.from_aos(st, "cuda"), torch.from_numpy(acts).cuda(), torch.empty((n, 20, 24), dtype=torch.float32, device="cuda")))
    errors = []

    def work(job):
        try:
            _, _, src, a_d, oh = job
            dst = torch.empty_like(src)
            for _ in range(25):
                ops.apply_moves(src, dst, a_d, n, cs, None, None, oh, L.FMT_F32)
        except Exception as e:                                        # surfaced below: a thread must not die silently
            errors.append(e)
    ops.release_workspaces()
    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for st, acts, _, _, oh in jobs:
        _, e_code, _, _ = oracle.step(cs, st, acts, threads=4)
        assert (oh.argmax(-1).cpu().numpy() == e_code).all() and float(oh.sum()) == n * 20
    assert len(ops._workspaces) == 1                                  # one stream, one workspace
    streams = [torch.cuda.Stream() for _ in range(ops._WS_MAX + 3)]
    _, _, src, a_d, oh = jobs[0]
    dst = torch.empty_like(src)
    torch.cuda.synchronize()
    for s in streams:
        with torch.cuda.stream(s):
            ops.apply_moves(src, dst, a_d, n, cs, None, None, oh, L.FMT_F32)
        s.synchronize()
    assert len(ops._workspaces) == ops._WS_MAX
    ops.release_workspaces()
    assert len(ops._workspaces) == 0


def test_cube_env_keeps_its_adi_plan_and_replays_it_as_a_graph(oracle):
    """CubeEnv.get_random_samples (cube_env.py:177-194) called epoch after epoch with one shape (train.py:152-155): the plan's buffers are
    kept between calls, with env.adi_graph the captured hipGraph is replayed, the samples are those of the eager env for the same
    numpy seed, the global generator ends where the reference leaves it, and the env sits on the last walk's final state."""
    import rubiks_cube_solver_amd as rc
    model, w, b = _linear_model(3)
    envs = [rc.make_env(torch.device("cuda"), 3) for _ in range(2)]
    envs[1].adi_graph = True
    bufs = [rc.TensorReplayBuffer(10_000, 512, 3), rc.TensorReplayBuffer(10_000, 512, 3)]
    states = []
    for env, buf in zip(envs, bufs):
        np.random.seed(77)
        for epoch in range(3):
            env.get_random_samples(buf, model, 12, 50, 0.5)
        states.append(np.random.get_state()[1].copy())
        assert len(env._adi_plans) == 1 and buf.size == 3 * 50 * 12
    assert (states[0] == states[1]).all()
    plan = next(iter(envs[1]._adi_plans.values()))
    assert plan.graph and len(plan._graphs) == 1 and not next(iter(envs[0]._adi_plans.values())).graph
    for k in ("code", "target_policy", "scramble_count"):
        assert torch.equal(getattr(bufs[0], k), getattr(bufs[1], k)), k
    assert torch.allclose(bufs[0].target_value, bufs[1].target_value, atol=1e-6) and np.allclose(bufs[0].error_memory, bufs[1].error_memory, atol=1e-6)
    # the samples themselves, against the oracle: the third epoch's draws
    np.random.seed(77)
    for _ in range(2 * 50):
        np.random.randint(12, size=12)
    acts = np.stack([np.random.randint(12, size=12) for _ in range(50)]).astype(np.uint8)
    exp, tv, tp, tie, err = _expected(oracle, 3, 50, 12, 0.5, w, b, actions_in=acts)
    last = slice(2 * 600, 3 * 600)
    assert (bufs[1].code[last].cpu().numpy().reshape(50, 12, 20) == exp["parent_code"]).all()
    assert np.allclose(bufs[1].target_value[last].cpu().numpy().reshape(50, 12), tv, atol=2e-5)
    assert np.allclose(bufs[1].error_memory[last].reshape(50, 12), err, atol=2e-5)
    from oracle.oracle_np import OracleCubeEnv
    ref = OracleCubeEnv(None, 3)
    for a in acts[-1]:
        ref.step(int(a))
    assert (envs[1].sim_cube == ref.sim_cube).all() and (envs[0].sim_cube == ref.sim_cube).all()
    other = torch.nn.Linear(480, 1).cuda()                             # another model object: the cached plan is dropped, not reused

    def model2(x):
        return other(x.reshape(x.shape[0], -1).float()), torch.zeros(x.shape[0], 12, device=x.device)
    envs[1].get_random_samples(bufs[1], model2, 12, 50, 0.5)
    assert len(envs[1]._adi_plans) == 1 and next(iter(envs[1]._adi_plans.values())).model is model2
    for e in envs:
        e.close()


def test_round5_entry_points_random_shapes(ops, L, oracle):
    """A bounded fuzz over the round-5 launches: random walk counts around the tile / pack / pass boundaries, depth groups, block
    strides and formats for the family block writer and the grouped target rule (against the oracle's codes and a numpy restatement of
    cube_env.py:229-251), random seeds and counts for both generator forms (against numpy)."""
    rng = np.random.default_rng(2025)
    dts = (torch.float32, torch.bfloat16, torch.float16, torch.uint8)
    for case in range(24):
        n = int(rng.choice([1, 2, 7, 8, 9, 63, 255, 256, 257, 1023, 1025, 4097, 16383, 16384, 16385, 33000]))
        D = int(rng.integers(1, 6))
        bs = n + int(rng.choice([0, 1, 7, 8, 100]))
        pitch, bufs = ops.adi_buffers(n, D, 3, "cuda", family=True)
        ops.adi_generate(n, D, 3, pitch, "cuda", seed=case, stream_id=9, **bufs)
        exp = oracle.adi(3, n, D, seed=case, stream=9, want_children=False, threads=4)
        dt = dts[case % 4]
        g0 = int(rng.integers(0, D))
        gd = D - g0
        blocks = torch.zeros((gd * 13 * bs, 20, 24), dtype=dt, device="cuda")
        ops.onehot_from_family(bufs["family"][g0:], n, 3, blocks, block_stride=bs, n_depths=gd)
        v = blocks.view(gd, 13, bs, 20, 24)
        got = v[:, :, :n].float().argmax(-1).cpu().numpy()
        assert (got[:, 12].transpose(1, 0, 2) == exp["parent_code"][:, g0:]).all(), (case, n, D, bs, dt)
        assert (got[:, :12].transpose(2, 0, 1, 3) == exp["child_code"][:, g0:]).all(), (case, n, D, bs, dt)
        assert float(v[:, :, :n].float().sum()) == gd * 13 * n * 20 and float(v[:, :, n:].float().abs().sum()) == 0    # one 1 per row, pad rows untouched
        vals = torch.randn(gd * 13 * bs, device="cuda")
        wgt = torch.rand(gd, dtype=torch.float64, device="cuda") + 0.1
        tv = torch.zeros((n, D), dtype=torch.float32, device="cuda")
        tp = torch.zeros((n, D), dtype=torch.int32, device="cuda")
        er = torch.zeros((n, D), dtype=torch.float64, device="cuda")
        ops.adi_targets_depths(vals, 13 * bs, bs, bufs["child_solved"][g0:], vals[12 * bs:], 13 * bs, wgt, n, gd, 3, tv[:, g0:], tp[:, g0:], er[:, g0:])
        vv = vals.view(gd, 13, bs)[:, :, :n].cpu().numpy()
        solved = exp["child_solved"][:, g0:].astype(bool)                                   # [n, gd, 12]
        cv = vv[:, :12].transpose(2, 0, 1) + np.float32(-1.0)
        want_tv = np.where(solved.any(-1), np.float32(1.0), cv.max(-1))
        want_tp = np.where(solved.any(-1), solved.argmax(-1), cv.argmax(-1))
        assert (tv[:, g0:].cpu().numpy() == want_tv).all() and (tp[:, g0:].cpu().numpy() == want_tp).all(), (case, n, D)
        want_er = np.abs(vv[:, 12].T.astype(np.float64) - want_tv.astype(np.float64)) * wgt.cpu().numpy()[None, :]
        assert (er[:, g0:].cpu().numpy() == want_er).all() and float(tv[:, :g0].abs().sum()) == 0
    for case in range(12):
        cs = 3 if case % 2 else 2
        A = 12 if cs == 3 else 6
        n = int(rng.choice([1, 63, 64, 65, 255, 257, 1000, 4099]))
        seeds = rng.integers(0, 2 ** 32, n, dtype=np.uint64)
        kmax = int(rng.choice([1, 2, 30, 64, 129, 260, 401, 480]))
        ks = rng.integers(0, kmax + 1, n)
        ks[0] = kmax
        want = _numpy_legacy(seeds, ks, A)
        variant = [0, 1, 2, 2 + 16 * int(rng.integers(1, 624))][case % 4]
        buf, kk = ops.legacy_scramble_actions(torch.from_numpy(seeds.astype(np.int64)), cs, ks.tolist(), device="cuda", variant=variant)
        got = buf.cpu().numpy()
        for i in range(n):
            assert (got[:ks[i], i] == want[i]).all() and (got[ks[i]:kk, i] == A).all(), (case, cs, n, kmax, variant, i)


def test_graph_plan_notices_reallocated_parameters():
    """A captured plan holds the net's parameter ADDRESSES: an in-place update (an optimizer step) is seen by the replay, a re-allocated
    parameter (model.to / .half / assigning .data) must trigger a new capture, not a replay over the freed tensor."""
    from rubiks_cube_solver_amd.adi import AdiPlan
    lin = torch.nn.Linear(480, 1).cuda()

    def model(x):
        return lin(x.reshape(x.shape[0], -1).float()), torch.zeros(x.shape[0], 12, device=x.device)
    model.parameters = lin.parameters
    gp = AdiPlan(model, 3, 300, 5, 0.5, graph=True)
    ep = AdiPlan(model, 3, 300, 5, 0.5)
    for step in range(3):
        a, b = gp.run(seed=step), ep.run(seed=step)
        assert torch.allclose(a["target_value"], b["target_value"], atol=1e-5) and torch.allclose(a["error"], b["error"], atol=1e-5)
        with torch.no_grad():
            lin.weight.add_(0.01 * (step + 1))                                  # in place: same address, the replay reads the new values
    assert len(gp._graphs) == 1
    first = next(iter(gp._graphs.values()))
    gp_before = gp.run(seed=9)["target_value"].clone()
    lin.weight.data = lin.weight.data.clone() * -1.5                            # a NEW tensor behind the parameter
    a, b = gp.run(seed=9), ep.run(seed=9)
    assert torch.allclose(a["target_value"], b["target_value"], atol=1e-5) and torch.allclose(a["error"], b["error"], atol=1e-5)
    assert float((a["target_value"] - gp_before).abs().max()) > 0.1              # (and the values did change with the weights)
    assert next(iter(gp._graphs.values())) is not first                         # captured again


def test_round5_entry_points_empty_and_error_paths(ops, L):
    """n = 0 / depth 0 are no-ops, bad pointers / strides / formats are RC_EINVAL with a message, through the raw C ABI."""
    lib, P = L.lib(), L.ptr
    L.init(torch.device("cuda", torch.cuda.current_device()))
    sp = L.stream_ptr(torch.device("cuda"))
    z8 = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    zf = torch.zeros(1 << 14, dtype=torch.float32, device="cuda")
    zd = torch.zeros(64, dtype=torch.float64, device="cuda")
    zi = torch.zeros(1 << 12, dtype=torch.int32, device="cuda")
    nf = L.family_layout(3)[0]
    # rc_onehot_from_family_depths
    assert lib.rc_onehot_from_family_depths(P(z8), 0, 256, 3, P(zf), L.FMT_F32, 0, 3, sp) == 0                # no walks
    assert lib.rc_onehot_from_family_depths(P(z8), 8, 256, 3, P(zf), L.FMT_F32, 8, 0, sp) == 0                # no depths
    for args in ((None, 8, 256, 3, P(zf), L.FMT_F32, 8, 1), (P(z8), 8, 256, 2, P(zf), L.FMT_F32, 8, 1), (P(z8), 8, 256, 3, P(zf), L.FMT_CODE, 8, 1),
                 (P(z8), 8, 256, 3, None, L.FMT_F32, 8, 1), (P(z8), 8, 256, 3, P(zf), L.FMT_F32, 7, 1), (P(z8), 8, 100, 3, P(zf), L.FMT_F32, 8, 1),
                 (P(z8), 8, 256, 3, P(zf), L.FMT_F32, 8, -1), (P(z8), 8, 256, 3, P(zf), L.FMT_F32, 8, 6000), (P(z8), -1, 256, 3, P(zf), L.FMT_F32, 8, 1)):
        assert lib.rc_onehot_from_family_depths(*args, sp) == -1 and lib.rc_last_error(), args
    assert nf * 256 <= z8.numel()
    # rc_adi_targets_depths
    assert lib.rc_adi_targets_depths(P(zf), 13 * 8, 8, P(z8), 256, P(zf), 13 * 8, P(zd), 0, 2, 3, P(zf), P(zi), P(zd), 2, sp) == 0
    assert lib.rc_adi_targets_depths(P(zf), 13 * 8, 8, P(z8), 256, P(zf), 13 * 8, P(zd), 8, 0, 3, P(zf), P(zi), P(zd), 2, sp) == 0
    ok = [P(zf), 13 * 8, 8, P(z8), 256, P(zf), 13 * 8, P(zd), 8, 2, 3, P(zf[4096:]), P(zi), P(zd[32:]), 2]
    assert lib.rc_adi_targets_depths(*ok, sp) == 0
    for i, bad in ((0, None), (3, None), (11, None), (12, None), (4, 4), (2, 4), (14, 1), (8, -1), (9, -1), (5, None), (7, None)):
        args = list(ok)
        args[i] = bad
        assert lib.rc_adi_targets_depths(*args, sp) == -1 and lib.rc_last_error(), (i, bad)
    args = list(ok); args[13] = None                                                      # no error output: parent / weight not needed
    args[5] = None; args[7] = None
    assert lib.rc_adi_targets_depths(*args, sp) == 0
    args = list(ok); args[10] = 4
    assert lib.rc_adi_targets_depths(*args, sp) == -1
    # rc_scramble_from / rc_search_pack / rc_host_alias / rc_legacy_scramble_actions_ex
    st = ops.alloc_states(16, 3, "cuda")
    ops.fill_solved(st, 16, 3)
    assert lib.rc_scramble_from(P(st), P(st), 0, 256, 3, 4, 0, 0, 0, None, None, 0, None, None, sp) == 0
    assert lib.rc_scramble_from(None, P(st), 16, 256, 3, 4, 0, 0, 0, None, None, 0, None, None, sp) == -1
    assert lib.rc_scramble_from(P(st), P(st), 16, 100, 3, 4, 0, 0, 0, None, None, 0, None, None, sp) == -1
    assert lib.rc_scramble_from(P(st), P(st), 16, 256, 5, 4, 0, 0, 0, None, None, 0, None, None, sp) == -1
    assert lib.rc_search_pack(P(z8), P(z8), P(z8), 0, 256, 3, P(z8), P(z8), P(z8), sp) == 0
    for i in range(3):
        args = [P(z8)] * 3 + [16, 256, 3] + [P(z8)] * 3
        args[(0, 4, 7)[i]] = (None, 100, None)[i]
        assert lib.rc_search_pack(*args, sp) == -1 and lib.rc_last_error()
    assert lib.rc_search_pack(P(z8), P(z8), P(z8), 16, 256, 4, P(z8), P(z8), P(z8), sp) == -1
    import ctypes
    out = ctypes.c_void_p(0)
    assert lib.rc_host_alias(None, ctypes.byref(out)) == -1 and lib.rc_host_alias(P(z8), None) == -1
    assert lib.rc_host_alias(P(z8), ctypes.byref(out)) == -1                               # device memory is not host-mapped memory
    seeds = torch.zeros(16, dtype=torch.int32, device="cuda")
    assert lib.rc_legacy_scramble_actions_ex(P(seeds), None, 3, 3, 0, 3, P(z8), 256, sp, 0) == 0
    assert lib.rc_legacy_scramble_actions_ex(P(seeds), None, 0, 0, 16, 3, P(z8), 256, sp, 2) == 0        # kmax 0: nothing to draw
    assert lib.rc_legacy_scramble_actions_ex(P(seeds), None, 4, 3, 16, 3, P(z8), 256, sp, 0) == -1       # count_uniform > kmax
    assert lib.rc_legacy_scramble_actions_ex(P(seeds), None, 3, 3, 16, 3, P(z8[1:]), 256, sp, 0) == -1   # misaligned output
    assert lib.rc_legacy_scramble_actions_ex(P(seeds), None, 3, 3, 16, 4, P(z8), 256, sp, 0) == -1
    assert lib.rc_legacy_scramble_actions_ex(None, None, 3, 3, 16, 3, P(z8), 256, sp, 0) == -1
    torch.cuda.synchronize()
    assert L.read_status() == 0
