"""Every launch geometry the host code can choose is executed and compared with the CPU oracle.  GPU only.

Round-1 review: the default dispatch only ever took `parts = 12` and pack width 1 in the tests, so the instantiations
that run at BASELINE sizes (wide packs, few parts, 16384-walk tiles) were never compared.  Here:
  * rc_expand_children_ex / rc_adi_generate_ex force parts in {1,2,3,4,6,12} ({1,2,3,6} for 2x2x2) x pack width
    {1,2} and compare every output with the oracle;
  * the default dispatch runs at its real sizes: expansion of 2^20 (+3, ragged) parents, BASELINE config 3
    (100 000 walks x depth 30) in full, and rc_apply_moves at 1M and 4M cubes on ALL cubes.
Reference lines restated by the oracle: gym-cube/gym_cube/envs/cube_env.py:177-194,212-236 (ADI / children),
assets/py333.py:220-246 (move, solved, one-hot)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

S_OF = {2: 24, 3: 54}
A_OF = {2: 6, 3: 12}
SL_OF = {2: 7, 3: 20}
PARTS = {3: (1, 2, 3, 4, 6, 12), 2: (1, 2, 3, 6)}


@pytest.fixture(scope="module")
def ops():
    from rubiks_cube_solver_amd import ops as o
    return o


@pytest.fixture(scope="module")
def L():
    from rubiks_cube_solver_amd import _lib
    return _lib


def untile(ops, t, n, lead):
    """[*lead, tiles, rows, pitch] (or [*lead, rows, pitch]) -> numpy [*lead, n, rows]."""
    if t.dim() == lead + 2:
        t = t.unsqueeze(lead)
    flat = t.reshape(-1, *t.shape[lead:])
    out = torch.stack([ops.to_aos(x, n) for x in flat]).reshape(*t.shape[:lead], n, t.shape[-2])
    return out.cpu().numpy()


def walk_states(oracle, cs, n, depth, seed, threads=8):
    """n random states, `depth` oracle moves from solved (host memory stays at a few state arrays)."""
    rng = np.random.default_rng(seed)
    st = oracle.solved(cs, n)
    for _ in range(depth):
        st = oracle.step(cs, st, rng.integers(0, A_OF[cs], n, dtype=np.uint8), threads=threads)[0]
    return st


@pytest.mark.parametrize("cs", [3, 2])
@pytest.mark.parametrize("v", [1, 2])
@pytest.mark.parametrize("n,pitch", [(2500, None), (9000, 512), (9000, 1024), (9000, 4096)])
def test_expand_every_parts_value(ops, L, oracle, cs, v, n, pitch):
    S, A, SL = S_OF[cs], A_OF[cs], SL_OF[cs]
    states = walk_states(oracle, cs, n, 20, seed=n + v)
    k = n // 9
    states[:k] = oracle.step(cs, oracle.solved(cs, k), np.arange(k) % A)[0]        # parents with a solved child
    ch, cc, cso = oracle.expand(cs, states, threads=8)
    src = ops.from_aos(states, "cuda")
    for parts in PARTS[cs]:
        out = ops.expand_buffers(n, cs, "cuda", pitch or L.pitch_for(n), children=True, codes=True)
        for t in out.values():
            t.fill_(9)
        ops.expand_children(src, n, cs, out["children"], out["child_solved"], out["child_code"], pitch=out["children"].shape[-1],
                            variant=parts * 1000 + v)
        assert (untile(ops, out["children"], n, 1).transpose(1, 0, 2) == ch).all(), parts
        assert (untile(ops, out["child_code"], n, 1).transpose(1, 0, 2) == cc).all(), parts
        assert (out["child_solved"][:, :n].cpu().numpy().T == cso).all() and cso.any(), parts
    if v == 2:                                                   # the streaming form (few persistent waves), forced for this small batch
        for h in (1, 3, 7):
            out = ops.expand_buffers(n, cs, "cuda", pitch or L.pitch_for(n), children=True, codes=False)
            for t in out.values():
                t.fill_(9)
            ops.expand_children(src, n, cs, out["children"], out["child_solved"], None, pitch=out["children"].shape[-1], variant=h * 100)
            assert (untile(ops, out["children"], n, 1).transpose(1, 0, 2) == ch).all(), h
            assert (out["child_solved"][:, :n].cpu().numpy().T == cso).all(), h
    assert L.read_status() == 0


@pytest.mark.parametrize("cs", [3, 2])
@pytest.mark.parametrize("v", [1, 2])
@pytest.mark.parametrize("n_walks,depth,pitch", [(1500, 6, None), (5000, 5, 512), (5000, 5, 1024), (5000, 5, 2048)])
def test_adi_every_parts_value(ops, L, oracle, cs, v, n_walks, depth, pitch):
    exp = oracle.adi(cs, n_walks, depth, seed=77, stream=2, walk0=5, threads=8)
    for parts in PARTS[cs]:
        pt, bufs = ops.adi_buffers(n_walks, depth, cs, "cuda", pitch or L.pitch_for(n_walks), parents=True, parent_code=True,
                                   children=True, child_code=True)
        for t in bufs.values():
            t.fill_(7)
        ops.adi_generate(n_walks, depth, cs, pt, "cuda", seed=77, stream_id=2, walk_offset=5, variant=parts * 1000 + v, **bufs)
        assert (bufs["actions_out"][:, :n_walks].cpu().numpy().T == exp["actions"]).all(), parts
        assert (untile(ops, bufs["parents"], n_walks, 1).transpose(1, 0, 2) == exp["parents"]).all(), parts
        assert (untile(ops, bufs["parent_code"], n_walks, 1).transpose(1, 0, 2) == exp["parent_code"]).all(), parts
        assert (untile(ops, bufs["children"], n_walks, 2).transpose(2, 0, 1, 3) == exp["children"]).all(), parts
        assert (untile(ops, bufs["child_code"], n_walks, 2).transpose(2, 0, 1, 3) == exp["child_code"]).all(), parts
        assert (bufs["child_solved"][..., :n_walks].cpu().numpy().transpose(2, 0, 1) == exp["child_solved"]).all(), parts
        assert exp["child_solved"].any()
        # stickers only (the byte-bound instantiation without code look-ups)
        pt, b2 = ops.adi_buffers(n_walks, depth, cs, "cuda", pitch or L.pitch_for(n_walks), parents=True, children=True)
        ops.adi_generate(n_walks, depth, cs, pt, "cuda", seed=77, stream_id=2, walk_offset=5, variant=parts * 1000 + v, **b2)
        assert torch.equal(b2["children"], bufs["children"]) or \
            (untile(ops, b2["children"], n_walks, 2).transpose(2, 0, 1, 3) == exp["children"]).all()
        assert (b2["child_solved"][..., :n_walks].cpu().numpy().transpose(2, 0, 1) == exp["child_solved"]).all(), parts
        if parts in (1, 2):   # replaying the recorded moves (actions_in) through the same instantiation gives the same bytes
            wp = b2["actions_out"].shape[1]
            a_in = torch.zeros((depth, wp), dtype=torch.uint8, device="cuda")
            a_in[:, :n_walks] = torch.from_numpy(np.ascontiguousarray(exp["actions"].T)).cuda()
            pt, b3 = ops.adi_buffers(n_walks, depth, cs, "cuda", pitch or L.pitch_for(n_walks), parents=True, children=True)
            ops.adi_generate(n_walks, depth, cs, pt, "cuda", actions_in=a_in, variant=parts * 1000 + v, **b3)
            assert (untile(ops, b3["children"], n_walks, 2).transpose(2, 0, 1, 3) == exp["children"]).all(), parts
            assert (b3["actions_out"][:, :n_walks].cpu().numpy().T == exp["actions"]).all(), parts
    assert L.read_status() == 0


@pytest.mark.parametrize("cs", [3, 2])
@pytest.mark.parametrize("v", [1, 2])
def test_adi_depth_segments(ops, L, oracle, cs, v):
    """Depth segments of the ADI kernel (a wave replays the moves of the earlier depths and emits only its own range): every
    segment count 1..depth and beyond (clamped to the depth), alone and combined with parts, drawn and replayed moves -- all outputs
    against the oracle (cube_env.py:177-194,212-236)."""
    n_walks, depth = 3000, 7
    exp = oracle.adi(cs, n_walks, depth, seed=91, stream=4, walk0=11, threads=8)
    wp = None
    with pytest.raises(L.RubikHipError, match="variant"):           # the segment field is 1..16 (RC_VARIANT_ADI_SEGS): the launcher rejects the rest
        pt, bufs = ops.adi_buffers(n_walks, depth, cs, "cuda", 1024, parents=True)
        ops.adi_generate(n_walks, depth, cs, pt, "cuda", seed=91, variant=40 * 1000000 + v, **bufs)
    for segs, parts in ((1, 1), (2, 1), (3, 2), (4, 1), (5, 3), (7, 1), (7, 6), (12, 1), (16, 2), (16, 1)):
        pt, bufs = ops.adi_buffers(n_walks, depth, cs, "cuda", 1024, parents=True, parent_code=True, children=True, child_code=True)
        for t in bufs.values():
            t.fill_(7)
        var = segs * 1000000 + parts * 1000 + v
        ops.adi_generate(n_walks, depth, cs, pt, "cuda", seed=91, stream_id=4, walk_offset=11, variant=var, **bufs)
        tag = (segs, parts)
        assert (bufs["actions_out"][:, :n_walks].cpu().numpy().T == exp["actions"]).all(), tag
        assert (untile(ops, bufs["parents"], n_walks, 1).transpose(1, 0, 2) == exp["parents"]).all(), tag
        assert (untile(ops, bufs["parent_code"], n_walks, 1).transpose(1, 0, 2) == exp["parent_code"]).all(), tag
        assert (untile(ops, bufs["children"], n_walks, 2).transpose(2, 0, 1, 3) == exp["children"]).all(), tag
        assert (untile(ops, bufs["child_code"], n_walks, 2).transpose(2, 0, 1, 3) == exp["child_code"]).all(), tag
        assert (bufs["child_solved"][..., :n_walks].cpu().numpy().transpose(2, 0, 1) == exp["child_solved"]).all(), tag
        # the same through replayed moves and the code-only instantiation
        wp = bufs["actions_out"].shape[1]
        a_in = torch.zeros((depth, wp), dtype=torch.uint8, device="cuda")
        a_in[:, :n_walks] = torch.from_numpy(np.ascontiguousarray(exp["actions"].T)).cuda()
        pt, b2 = ops.adi_buffers(n_walks, depth, cs, "cuda", 1024, parent_code=True, child_code=True)
        for t in b2.values():
            t.fill_(7)
        ops.adi_generate(n_walks, depth, cs, pt, "cuda", actions_in=a_in, variant=var, **b2)
        assert (untile(ops, b2["parent_code"], n_walks, 1).transpose(1, 0, 2) == exp["parent_code"]).all(), tag
        assert (untile(ops, b2["child_code"], n_walks, 2).transpose(2, 0, 1, 3) == exp["child_code"]).all(), tag
        assert (b2["child_solved"][..., :n_walks].cpu().numpy().transpose(2, 0, 1) == exp["child_solved"]).all(), tag
        assert (b2["actions_out"][:, :n_walks].cpu().numpy().T == exp["actions"]).all(), tag
    assert L.read_status() == 0


@pytest.mark.parametrize("n", [(1 << 20) + 3])
@pytest.mark.parametrize("codes", [False, True])
def test_expand_one_million_parents_default_dispatch(ops, L, oracle, n, codes):
    """The instantiation the MCTS / ADI callers get at scale (wide pack, one part, 32768-cube tiles): all children of
    2^20 + 3 parents against the oracle (cube_env.py:212-236)."""
    cs, A = 3, 12
    states = walk_states(oracle, cs, n, 14, seed=3, threads=oracle.max_threads())
    states[:1000] = oracle.step(cs, oracle.solved(cs, 1000), np.arange(1000) % A)[0]
    ch, cc, cso = oracle.expand(cs, states, threads=oracle.max_threads())
    src = ops.from_aos(states, "cuda")
    out = ops.expand_buffers(n, cs, "cuda", children=True, codes=codes)
    ops.expand_children(src, n, cs, out["children"], out["child_solved"], out.get("child_code"), pitch=out["children"].shape[-1])
    assert out["children"].shape[1] > 1                                           # tiled
    assert (out["child_solved"][:, :n].cpu().numpy().T == cso).all() and cso.sum() >= 1000
    for a in range(A):                                                            # child by child: keeps host memory bounded
        assert (ops.to_aos(out["children"][a], n).cpu().numpy() == ch[:, a]).all(), a
        if codes:
            assert (ops.to_aos(out["child_code"][a], n).cpu().numpy() == cc[:, a]).all(), a
    assert L.read_status() == 0


def test_adi_config3_full_size(ops, L, oracle):
    """BASELINE config 3 exactly as bench.py launches it: 100 000 walks x depth 30 from solved, default dispatch and
    default tiling, every output byte against the oracle (streamed in walk chunks): actions, parents, all 12 children
    and their solved flags (cube_env.py:177-194,212-236)."""
    cs, W, D, A, S = 3, 100_000, 30, 12, 54
    pt, bufs = ops.adi_buffers(W, D, cs, "cuda", parents=True, children=True)
    assert bufs["children"].shape[2] > 1                                          # the tiled layout bench.py uses
    ops.adi_generate(W, D, cs, pt, "cuda", seed=2024, stream_id=0, **bufs)
    assert L.read_status() == 0
    chunk = pt                                                                    # one output tile of walks at a time
    n_solved = 0
    for w0 in range(0, W, chunk):
        m = min(chunk, W - w0)
        exp = oracle.adi(cs, m, D, seed=2024, stream=0, walk0=w0, threads=oracle.max_threads())
        t = w0 // pt
        assert (bufs["actions_out"][:, w0:w0 + m].cpu().numpy().T == exp["actions"]).all(), w0
        par = bufs["parents"][:, t, :, :m].cpu().numpy()                          # [D, S, m]
        assert (par.transpose(2, 0, 1) == exp["parents"]).all(), w0
        kids = bufs["children"][:, :, t, :, :m].cpu().numpy()                     # [D, A, S, m]
        assert (kids.transpose(3, 0, 1, 2) == exp["children"]).all(), w0
        flags = bufs["child_solved"][:, :, w0:w0 + m].cpu().numpy()               # [D, A, m]
        assert (flags.transpose(2, 0, 1) == exp["child_solved"]).all(), w0
        n_solved += int(exp["child_solved"].sum())
    assert n_solved >= W                                                          # depth 1: the inverse move solves every walk


@pytest.mark.parametrize("cs,log2n", [(3, 20), (3, 22), (2, 22)])
def test_apply_moves_full_batch_vs_oracle(ops, L, oracle, cs, log2n):
    """BASELINE config 2 (2^20 cubes) and the metric's batch (2^22; also for 2x2x2): one rc_apply_moves launch with reward,
    done and the compact code, default dispatch, compared with the oracle on ALL cubes (SURVEY 8d asks for >= 64 K)."""
    n = 1 << log2n
    thr = oracle.max_threads()
    states = walk_states(oracle, cs, n, 20, seed=log2n, threads=thr)
    acts = np.random.default_rng(log2n).integers(0, A_OF[cs], n, dtype=np.uint8)
    k = 4096
    states[:k] = oracle.step(cs, oracle.solved(cs, k), (acts[:k] ^ 1))[0]         # these become solved again
    exp_st, exp_code, exp_done, exp_rew = oracle.step(cs, states, acts, threads=thr)
    src = ops.from_aos(states, "cuda")
    dst = torch.empty_like(src)
    a_d = torch.from_numpy(acts).cuda()
    rew = torch.empty(n, dtype=torch.float32, device="cuda")
    done = torch.empty(n, dtype=torch.uint8, device="cuda")
    code = ops.alloc_code(n, cs, "cuda")
    ops.apply_moves(src, dst, a_d, n, cs, rew, done, code, L.FMT_CODE)
    assert src.shape[0] > 1
    assert (ops.to_aos(dst, n).cpu().numpy() == exp_st).all()
    assert (ops.to_aos(code, n).cpu().numpy() == exp_code).all()
    assert (done.cpu().numpy() == exp_done).all() and exp_done[:k].all()
    assert (rew.cpu().numpy() == exp_rew).all()
    # the bench's launch shape: move + done only, and in place
    done2 = torch.empty_like(done)
    dst2 = torch.empty_like(src)
    ops.apply_moves(src, dst2, a_d, n, cs, None, done2)
    assert torch.equal(dst2, dst) and torch.equal(done2, done)
    ops.apply_moves(src, src, a_d, n, cs, None, done2)
    assert torch.equal(src, dst) and torch.equal(done2, done)
    assert L.read_status() == 0


@pytest.mark.parametrize("cs", [3, 2])
def test_step_policies_agree_at_scale(ops, L, cs):
    """The three row-traffic policies and both pack widths of the step kernel produce identical bytes on a batch large
    enough to take the full-wave fast path everywhere (2^21 + 5 cubes, ragged tail)."""
    n = (1 << 21) + 5
    A = A_OF[cs]
    st = ops.alloc_states(n, cs, "cuda")
    ops.fill_solved(st, n, cs)
    ops.scramble(st, n, cs, 15, seed=cs)
    acts = torch.randint(0, A, (n,), dtype=torch.uint8, device="cuda")
    ref = torch.empty_like(st)
    ref_done = torch.empty(n, dtype=torch.uint8, device="cuda")
    ops.apply_moves(st, ref, acts, n, cs, None, ref_done, variant=21)              # narrow pack, default-cached
    valid = ops.to_aos(ref, n)
    ref_code = ops.alloc_code(n, cs, "cuda")
    ref_rew = torch.empty(n, dtype=torch.float32, device="cuda")
    ops.apply_moves(st, ref, acts, n, cs, ref_rew, ref_done, ref_code, L.FMT_CODE, variant=21)
    for variant in (0, 1, 2, 11, 12, 22, 31, 32, 41, 42):
        out = torch.zeros_like(st)
        done = torch.zeros_like(ref_done)
        ops.apply_moves(st, out, acts, n, cs, None, done, variant=variant)
        assert torch.equal(ops.to_aos(out, n), valid), variant
        assert torch.equal(done, ref_done), variant
        # every side output (reward, done, compact code) through every policy's store forms, ping-pong and in place
        out.zero_(); done.zero_()
        code, rew = torch.zeros_like(ref_code), torch.zeros_like(ref_rew)
        ops.apply_moves(st, out, acts, n, cs, rew, done, code, L.FMT_CODE, variant=variant)
        assert torch.equal(ops.to_aos(out, n), valid) and torch.equal(done, ref_done) and torch.equal(rew, ref_rew), variant
        assert torch.equal(ops.to_aos(code, n), ops.to_aos(ref_code, n)), variant
        work = st.clone()
        code.zero_(); rew.zero_()
        ops.apply_moves(work, work, acts, n, cs, rew, done, code, L.FMT_CODE, variant=variant)
        assert torch.equal(ops.to_aos(work, n), valid) and torch.equal(rew, ref_rew) and torch.equal(ops.to_aos(code, n), ops.to_aos(ref_code, n)), variant
    assert L.read_status() == 0


def test_adi_codes_default_dispatch_100k(ops, L, oracle):
    """The code-only ADI instantiation at config 3's walk count (default dispatch: narrow packs, two parts): parent and
    child codes, flags and actions of 100 000 walks x depth 4 against the oracle."""
    cs, W, D = 3, 100_000, 4
    pt, bufs = ops.adi_buffers(W, D, cs, "cuda", parents=True, parent_code=True, child_code=True)
    ops.adi_generate(W, D, cs, pt, "cuda", seed=5, stream_id=1, **bufs)
    exp = oracle.adi(cs, W, D, seed=5, stream=1, threads=oracle.max_threads(), want_children=False)
    assert (bufs["actions_out"][:, :W].cpu().numpy().T == exp["actions"]).all()
    assert (untile(ops, bufs["parents"], W, 1).transpose(1, 0, 2) == exp["parents"]).all()
    assert (untile(ops, bufs["parent_code"], W, 1).transpose(1, 0, 2) == exp["parent_code"]).all()
    assert (untile(ops, bufs["child_code"], W, 2).transpose(2, 0, 1, 3) == exp["child_code"]).all()
    assert (bufs["child_solved"][..., :W].cpu().numpy().transpose(2, 0, 1) == exp["child_solved"]).all()
    # the same walks as FAMILY records (51 shared look-ups per state instead of 13 x 20 picked codes): every slot code of the parent
    # and of every child is the family row the layout table names, and one rc_onehot_from_family launch per depth gives the dense
    # one-hots of all 12 children and the parent
    nf, rows = L.family_layout(cs)
    assert nf == 51 and rows.shape == (13, 20)
    pt2, fb = ops.adi_buffers(W, D, cs, "cuda", parents=True, family=True)
    ops.adi_generate(W, D, cs, pt2, "cuda", seed=5, stream_id=1, **fb)
    assert (fb["actions_out"][:, :W].cpu().numpy().T == exp["actions"]).all()
    assert (untile(ops, fb["parents"], W, 1).transpose(1, 0, 2) == exp["parents"]).all()
    assert (fb["child_solved"][..., :W].cpu().numpy().transpose(2, 0, 1) == exp["child_solved"]).all()
    fam = untile(ops, fb["family"], W, 1).transpose(1, 0, 2)                       # [W, D, NF]
    assert (fam[:, :, rows[12]] == exp["parent_code"]).all()
    for a in range(12):
        assert (fam[:, :, rows[a]] == exp["child_code"][:, :, a]).all(), a
    p = fb["actions_out"].shape[1]
    for dt in (torch.float32, torch.bfloat16, torch.uint8):
        dense = torch.full((13 * p, 20, 24), 3, dtype=dt, device="cuda")
        for d in (0, D - 1):
            ops.onehot_from_family(fb["family"][d], W, cs, dense, block_stride=p)
            got = dense.view(13, p, 20, 24)[:, :W].float().argmax(-1).to(torch.uint8).cpu().numpy()   # [13, W, 20]
            assert (got[12] == exp["parent_code"][:, d]).all() and (got[:12].transpose(1, 0, 2) == exp["child_code"][:, d]).all(), (dt, d)
            assert float(dense.view(13, p, 20, 24)[:, :W].float().sum()) == 13.0 * 20 * W
            assert float(dense.view(13, p, 20, 24)[:, W:].float().min()) == 3.0 if p > W else True       # pad cubes untouched
    assert "k_adi<Cube3,V=1,code,family> parts=1 segs=4 grid=1564" in L.describe(L.OP_ADI, cs, W, 30, outputs=L.OUT_FAMILY | L.OUT_FLAGS)
    assert L.read_status() == 0


@pytest.mark.parametrize("cs", [3, 2])
@pytest.mark.parametrize("n,pitch,variant", [(1, None, 0), (700, None, 0), (5000, 512, 0), (5000, 1024, 2001001), (40000, None, 1)])
def test_adi_family_records_small_and_tiled(ops, L, oracle, cs, n, pitch, variant):
    """FAMILY records on small, ragged and tiled batches, both cube sizes, forced pack widths / depth segments: the rows the layout
    table names equal the oracle's parent and child codes (cube_env.py:212-236, py333.py:224-227), the other outputs are unchanged;
    3x3x3: the dense expansion of every block equals the oracle's codes."""
    D, A, SL = 5, A_OF[cs], 20 if cs == 3 else 7
    nf, rows = L.family_layout(cs)
    assert nf == (51 if cs == 3 else 15) and rows.shape == (A + 1, SL) and int(rows.max()) == nf - 1
    pt, fb = ops.adi_buffers(n, D, cs, "cuda", pitch=pitch, parents=True, family=True)
    ops.adi_generate(n, D, cs, pt, "cuda", seed=9, stream_id=2, variant=variant, **fb)
    exp = oracle.adi(cs, n, D, seed=9, stream=2, want_children=False)
    assert (fb["actions_out"][:, :n].cpu().numpy().T == exp["actions"]).all()
    assert (untile(ops, fb["parents"], n, 1).transpose(1, 0, 2) == exp["parents"]).all()
    assert (fb["child_solved"][..., :n].cpu().numpy().transpose(2, 0, 1) == exp["child_solved"]).all()
    fam = untile(ops, fb["family"], n, 1).transpose(1, 0, 2)
    assert (fam[:, :, rows[A]] == exp["parent_code"]).all()
    for a in range(A):
        assert (fam[:, :, rows[a]] == exp["child_code"][:, :, a]).all(), a
    if cs == 3:
        p = fb["actions_out"].shape[1]
        dense = torch.full((13 * p + 3, 20, 24), 3, dtype=torch.float16, device="cuda")
        ops.onehot_from_family(fb["family"][2], n, cs, dense, block_stride=p)
        got = dense[:13 * p].view(13, p, 20, 24)[:, :n].float().argmax(-1).to(torch.uint8).cpu().numpy()
        assert (got[12] == exp["parent_code"][:, 2]).all() and (got[:12].transpose(1, 0, 2) == exp["child_code"][:, 2]).all()
        assert float(dense[13 * p:].float().min()) == 3.0
        if n > 1:                                                                    # a block stride smaller than the batch is refused, nothing written
            rc = L.lib().rc_onehot_from_family(L.ptr(fb["family"][2]), n, fb["family"].shape[-1], cs, L.ptr(dense), L.FMT_F16, n - 1, L.stream_ptr(dense.device))
            assert rc == -1 and b"block_stride" in L.lib().rc_last_error()
        rc = L.lib().rc_adi_generate_family(9, 2, 0, n, D, cs, pt, None, None, None, None, None, L.stream_ptr(dense.device), 0)
        assert rc == -1 and b"family is NULL" in L.lib().rc_last_error()
    else:
        with pytest.raises(L.RubikHipError):
            ops.onehot_from_family(fb["family"][0], n, cs, torch.empty((7 * n + n, 7, 21), dtype=torch.float32, device="cuda"), block_stride=n)
    assert L.read_status() == 0


@pytest.mark.parametrize("n", [1, 255, 1000, 3841, 70001, (1 << 17) + 77, 300_001])
def test_dense_writer_forms(ops, L, oracle, n):
    """Every form of the dense one-hot writers (64-cube tiles, 256-cube tiles, the wide 960-thread code -> dense form and the
    default dispatch) x every element type, fused with the move (ping-pong and in place), encode-only and code -> dense, ragged sizes
    around the tile (256), round (15 tiles = 3840 cubes) and group boundaries: the one-hot's arg-max is the oracle's code,
    every row holds exactly one 1, states / done / reward equal the oracle's (py333.py:220-246, cube_env.py:71-111)."""
    cs = 3
    states = walk_states(oracle, cs, n, 12, seed=n % 1000)
    acts = np.random.default_rng(n).integers(0, 12, n, dtype=np.uint8)
    k = min(n, 64)
    states[:k] = oracle.step(cs, oracle.solved(cs, k), acts[:k] ^ 1)[0]                  # these become solved
    exp_st, exp_code, exp_done, exp_rew = oracle.step(cs, states, acts, threads=8)
    src = ops.from_aos(states, "cuda")
    a_d = torch.from_numpy(acts).cuda()
    exp_code_t = torch.from_numpy(exp_code).cuda()
    code_buf = ops.alloc_code(n, cs, "cuda")
    ops.encode(src, n, cs, code_buf, L.FMT_CODE)
    src_code = ops.to_aos(code_buf, n)
    fmts = ((L.FMT_U8, torch.uint8), (L.FMT_F16, torch.float16), (L.FMT_BF16, torch.bfloat16), (L.FMT_F32, torch.float32))
    for form in (0, 100000, 200000, 300000):
        if n > 100000 and form == 100000:
            continue                                                                     # 64-cube tiles at 300k cubes: covered at 70001
        for fmt, dt in fmts:
            tag = (n, form, str(dt))
            oh = torch.full((n, 20, 24), 3, dtype=dt, device="cuda")
            dst = torch.zeros_like(src)
            rew = torch.zeros(n, dtype=torch.float32, device="cuda")
            done = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
            ops.apply_moves(src, dst, a_d, n, cs, rew, done, oh, fmt, variant=min(form, 200000))     # the fused writer has tile forms 1, 2 only
            assert torch.equal(oh.float().argmax(-1).to(torch.uint8), exp_code_t), tag
            assert float(oh.float().sum()) == 20.0 * n and float(oh.float().max()) == 1.0, tag
            assert (ops.to_aos(dst, n).cpu().numpy() == exp_st).all(), tag
            assert (done.cpu().numpy() == exp_done).all() and (rew.cpu().numpy() == exp_rew).all(), tag
            # in place
            work = src.clone()
            oh.fill_(3)
            ops.apply_moves(work, work, a_d, n, cs, None, done, oh, fmt, variant=min(form, 200000))
            assert torch.equal(work, dst) and torch.equal(oh.float().argmax(-1).to(torch.uint8), exp_code_t), tag
            # code -> dense
            oh.fill_(3)
            ops.onehot_from_code(code_buf, n, cs, oh, variant=form)
            assert torch.equal(oh.float().argmax(-1).to(torch.uint8), src_code) and float(oh.float().sum()) == 20.0 * n, tag
        # encode-only (no move) goes through the default dispatch of rc_encode
    oh = torch.full((n, 20, 24), 3, dtype=torch.float32, device="cuda")
    ops.encode(src, n, cs, oh, L.FMT_F32)
    assert torch.equal(oh.argmax(-1).to(torch.uint8), src_code) and float(oh.sum()) == 20.0 * n
    assert L.read_status() == 0
    if n >= 1 << 17:                # what form 0 (the default dispatch) was above: round 4's defaults, the wide form is form 300000
        assert "k_step_dense<Cube3,bf16,move,store,TILE=256>" in L.describe(L.OP_STEP, 3, n, outputs=L.OUT_STATES, fmt=L.FMT_BF16)
        assert "k_code_to_dense_front<Cube3,f32,F=1,gather>" in L.describe(L.OP_CODE_TO_DENSE, 3, n, fmt=L.FMT_F32)
        assert "k_code_to_dense_wide" in L.describe(L.OP_CODE_TO_DENSE, 3, n, fmt=L.FMT_F32, variant=300000)


def _random_shapes(seed, count):
    """Seeded ragged sizes around the places where the kernels change behaviour: pack width (4 / 8 cubes per lane), wave span
    (256 / 512), tile boundaries (1024-cube tiles), single cubes."""
    rng = np.random.default_rng(seed)
    special = [1, 2, 3, 4, 5, 7, 8, 9, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2049, 4095, 4097]
    out = []
    for _ in range(count):
        n = int(rng.choice(special)) if rng.random() < 0.5 else int(rng.integers(1, 9000))
        pitch = None if n <= 1024 or rng.random() < 0.4 else int(rng.choice([512, 1024, 2048, 4096]))
        out.append((n, pitch, int(rng.choice([3, 2])), int(rng.choice([0, 1, 2])), int(rng.choice([0, 1, 2, 3, 6]))))
    return out


def test_randomised_ragged_sizes_all_kernels(ops, L, oracle):
    """48 seeded (size, tiling, cube size, pack width, parts) combinations through step (+ code), expansion and ADI."""
    for case, (n, pitch, cs, v, parts) in enumerate(_random_shapes(20260, 48)):
        A = A_OF[cs]
        tag = (case, n, pitch, cs, v, parts)
        states = walk_states(oracle, cs, n, 9, seed=case)
        acts = np.random.default_rng(case).integers(0, A, n, dtype=np.uint8)
        exp_st, exp_code, exp_done, exp_rew = oracle.step(cs, states, acts)
        src = ops.from_aos(states, "cuda", pitch)
        dst = torch.zeros_like(src)
        code = torch.zeros((src.shape[0], SL_OF[cs], src.shape[2]), dtype=torch.uint8, device="cuda")
        rew = torch.zeros(n, dtype=torch.float32, device="cuda")
        done = torch.zeros(n, dtype=torch.uint8, device="cuda")
        ops.apply_moves(src, dst, torch.from_numpy(acts).cuda(), n, cs, rew, done, code, L.FMT_CODE, variant=v)
        assert (ops.to_aos(dst, n).cpu().numpy() == exp_st).all(), tag
        assert (ops.to_aos(code, n).cpu().numpy() == exp_code).all(), tag
        assert (done.cpu().numpy() == exp_done).all() and (rew.cpu().numpy() == exp_rew).all(), tag
        ch, cc, cso = oracle.expand(cs, exp_st, threads=4)
        out = ops.expand_buffers(n, cs, "cuda", src.shape[2], children=True, codes=True)
        ops.expand_children(dst, n, cs, out["children"], out["child_solved"], out["child_code"], pitch=src.shape[2], variant=parts * 1000 + v)
        assert (untile(ops, out["children"], n, 1).transpose(1, 0, 2) == ch).all(), tag
        assert (untile(ops, out["child_code"], n, 1).transpose(1, 0, 2) == cc).all(), tag
        assert (out["child_solved"][:, :n].cpu().numpy().T == cso).all(), tag
        depth = 1 + case % 4
        m = min(n, 3000)
        adi_pitch = src.shape[2] if m > 1024 and pitch else L.pitch_for(m)
        pt, bufs = ops.adi_buffers(m, depth, cs, "cuda", adi_pitch, parents=True, children=True, child_code=True, parent_code=True)
        segs = (0, 1, 2, 3, 5)[case % 5]                                                  # depth segments: clamped to the depth by the library
        ops.adi_generate(m, depth, cs, pt, "cuda", seed=case, stream_id=7, walk_offset=case * 13, variant=segs * 1000000 + parts * 1000 + v, **bufs)
        ex = oracle.adi(cs, m, depth, seed=case, stream=7, walk0=case * 13, threads=4)
        assert (untile(ops, bufs["children"], m, 2).transpose(2, 0, 1, 3) == ex["children"]).all(), tag
        assert (untile(ops, bufs["child_code"], m, 2).transpose(2, 0, 1, 3) == ex["child_code"]).all(), tag
        assert (untile(ops, bufs["parents"], m, 1).transpose(1, 0, 2) == ex["parents"]).all(), tag
        assert (bufs["child_solved"][..., :m].cpu().numpy().transpose(2, 0, 1) == ex["child_solved"]).all(), tag
        # dense one-hot of the stepped states in a random element type, fused and from the code buffer, every writer form
        fmt, dt = ((L.FMT_U8, torch.uint8), (L.FMT_F16, torch.float16), (L.FMT_BF16, torch.bfloat16), (L.FMT_F32, torch.float32))[case % 4]
        form = (0, 100000, 200000, 300000)[(case // 4) % 4]
        R, C = (20, 24) if cs == 3 else (7, 21)
        oh = torch.full((n, R, C), 2, dtype=dt, device="cuda")
        ops.apply_moves(src, torch.empty_like(src), torch.from_numpy(acts).cuda(), n, cs, None, None, oh, fmt, variant=min(form, 200000))
        oh2 = torch.full((n, R, C), 2, dtype=dt, device="cuda")
        ops.onehot_from_code(code, n, cs, oh2, variant=form)
        assert torch.equal(oh, oh2) and float(oh.float().sum()) == float(n * (20 if cs == 3 else 7)), tag
        if cs == 3:
            assert (oh.float().argmax(-1).cpu().numpy() == exp_code).all(), tag
    assert L.read_status() == 0


def test_ninety_million_cubes_cross_the_4gib_line(ops, L, oracle):
    """Maximum-size edge: 90 000 001 cubes = 4.86 GB per state buffer, so tile offsets pass 2^32 bytes (cube 79.5 M).  Device-drawn
    3-move scrambles are compared with the oracle on three windows (start, around the 4-GiB line, the ragged end), a move followed
    by its inverse restores every byte, and the fused compact code of the windows equals the oracle's (py333.py:220-246)."""
    cs, n, depth = 3, 90_000_001, 3
    st = ops.alloc_states(n, cs, "cuda")
    assert st.numel() > (1 << 32)
    done = torch.empty(n, dtype=torch.uint8, device="cuda")
    ops.fill_solved(st, n, cs)
    ops.is_solved(st, n, cs, done)
    assert int(done.sum()) == n
    ops.scramble(st, n, cs, depth, seed=77, stream_id=3)
    pitch = st.shape[-1]
    line = ((1 << 32) // (54 * pitch)) * pitch                       # first cube of the tile that straddles byte 2^32
    windows = [(0, 70_000), (line - 40_000, 80_000), (n - 70_001, 70_001)]
    aos = lambda t, w0, m: ops.to_aos(t[w0 // pitch:(w0 + m - 1) // pitch + 1], (w0 % pitch) + m)[w0 % pitch:].cpu().numpy()
    expect = {}
    for w0, m in windows:
        exp = oracle.adi(cs, m, depth, seed=77, stream=3, walk0=w0, threads=8, want_children=False)
        expect[w0] = (exp["parents"][:, -1], exp["parent_code"][:, -1])
        assert (aos(st, w0, m) == expect[w0][0]).all(), w0
    g = torch.Generator(device="cuda").manual_seed(5)
    acts = torch.randint(0, 12, (n,), generator=g, device="cuda", dtype=torch.uint8)
    out = torch.empty_like(st)
    code = ops.alloc_code(n, cs, "cuda")
    ops.apply_moves(st, out, acts, n, cs, None, done)
    ops.apply_moves(out, out, acts ^ 1, n, cs, None, done, code, L.FMT_CODE)          # X' undoes X, in place, with the fused code
    flat = lambda t: t.permute(0, 2, 1).reshape(-1, t.shape[1])[:n]                      # ignore the pad columns of the last tile
    assert torch.equal(flat(out), flat(st))
    for w0, m in windows:
        assert (aos(code, w0, m) == expect[w0][1]).all(), w0
    assert L.read_status() == 0


def test_outputs_beyond_4gib_adi_and_expansion(ops, L, oracle):
    """Write-once output streams larger than 4 GiB: ADI 230 000 walks x 30 (children 4.47 GB) and the expansion of 2^23 parents
    (children 5.4 GB).  Windows at the start and at the very end of the streams against the oracle (64-bit base offsets of the per-
    child descriptors; cube_env.py:177-194,212-236)."""
    cs, W, D = 3, 230_000, 30
    pt, bufs = ops.adi_buffers(W, D, cs, "cuda", parents=True, children=True)
    assert bufs["children"].numel() > (1 << 32)
    ops.adi_generate(W, D, cs, pt, "cuda", seed=31, stream_id=2, **bufs)
    for w0, m in ((0, 16384), (W - 16384 - 176, 16384 + 176)):        # first tile; the last (ragged) tiles
        exp = oracle.adi(cs, m, D, seed=31, stream=2, walk0=w0, threads=8)
        t0, t1 = w0 // pt, (w0 + m - 1) // pt
        for d in (0, D - 1):
            for a in (0, 11):
                got = ops.to_aos(bufs["children"][d, a, t0:t1 + 1], (w0 - t0 * pt) + m)[w0 - t0 * pt:].cpu().numpy()
                assert (got == exp["children"][:, d, a]).all(), (w0, d, a)
            assert (bufs["child_solved"][d, :, w0:w0 + m].cpu().numpy().T == exp["child_solved"][:, d]).all(), (w0, d)
        assert (bufs["actions_out"][:, w0:w0 + m].cpu().numpy().T == exp["actions"]).all(), w0
    del bufs
    torch.cuda.empty_cache()
    n = (1 << 23) + 5
    st = ops.alloc_states(n, cs, "cuda")
    ops.fill_solved(st, n, cs)
    ops.scramble(st, n, cs, 6, seed=8, stream_id=1)
    ex = ops.expand_buffers(n, cs, "cuda", children=True, codes=False)
    assert ex["children"].numel() > (1 << 32)
    ops.expand_children(st, n, cs, ex["children"], ex["child_solved"], pitch=ex["children"].shape[-1])
    p = st.shape[-1]
    for w0, m in ((0, 40_000), (n - 40_000, 40_000)):
        t0, t1 = w0 // p, (w0 + m - 1) // p
        parents = ops.to_aos(st[t0:t1 + 1], (w0 - t0 * p) + m)[w0 - t0 * p:].cpu().numpy()
        ch, _, cso = oracle.expand(cs, parents, threads=8)
        for a in (0, 5, 11):
            got = ops.to_aos(ex["children"][a, t0:t1 + 1], (w0 - t0 * p) + m)[w0 - t0 * p:].cpu().numpy()
            assert (got == ch[:, a]).all(), (w0, a)
        assert (ex["child_solved"][:, w0:w0 + m].cpu().numpy().T == cso).all(), w0
    assert "k_expand_stream" in L.describe(L.OP_EXPAND, cs, n, outputs=L.OUT_STATES | L.OUT_FLAGS)
    assert L.read_status() == 0


@pytest.mark.parametrize("n", [1, 7, 3841, 70001, (1 << 17) + 77, 300_001])
def test_dense_front_forms_and_workspace_route(ops, L, oracle, n):
    """Round 4's FRONT writer (one 3840-byte pass per workgroup: 2 / 4 / 8 whole cubes) in every shape -- one linear front, one
    front per XCD, 2 and 4 fronts per XCD per workgroup, code bytes gathered per lane or fetched once through LDS -- x every element type, on ragged sizes (a last pass with 1 .. 7 cubes, XCD ranges of
    unequal length, grids padded to a multiple of 8), from tiled and single-tile code buffers; and the two-launch route of
    rc_apply_moves_ws (step + compact code into the caller's workspace, then the front writer), ping-pong and in place, against
    the one-launch kernel and the oracle (py333.py:220-246, cube_env.py:71-111)."""
    cs = 3
    states = walk_states(oracle, cs, n, 12, seed=n % 1000 + 1)
    acts = np.random.default_rng(n + 1).integers(0, 12, n, dtype=np.uint8)
    k = min(n, 64)
    states[:k] = oracle.step(cs, oracle.solved(cs, k), acts[:k] ^ 1)[0]                  # these become solved
    exp_st, exp_code, exp_done, exp_rew = oracle.step(cs, states, acts, threads=8)
    src = ops.from_aos(states, "cuda")
    a_d = torch.from_numpy(acts).cuda()
    exp_code_t = torch.from_numpy(exp_code).cuda()
    fmts = ((L.FMT_U8, torch.uint8), (L.FMT_F16, torch.float16), (L.FMT_BF16, torch.bfloat16), (L.FMT_F32, torch.float32))
    for pitch in (None, L.pitch_for(n)):                                                  # 32768-cube tiles / one tile
        code_buf = ops.alloc_code(n, cs, "cuda", pitch=pitch)
        ops.encode(src, n, cs, code_buf, L.FMT_CODE)
        src_code = ops.to_aos(code_buf, n)
        # units digit: fronts per XCD per workgroup; tens digit: 2 one linear front, 3 byte gather per lane, 4 one load + LDS
        for form in (400000, 400031, 400032, 400034, 400041, 400042, 400044, 400020):
            for fmt, dt in fmts:
                oh = torch.full((n + 3, 20, 24), 3, dtype=dt, device="cuda")            # three guard cubes behind the batch
                ops.onehot_from_code(code_buf, n, cs, oh[:n], variant=form)
                tag = (n, pitch, form, str(dt))
                assert torch.equal(oh[:n].float().argmax(-1).to(torch.uint8), src_code), tag
                assert float(oh[:n].float().sum()) == 20.0 * n and float(oh[:n].float().max()) == 1.0, tag
                assert float(oh[n:].float().min()) == 3.0 and float(oh[n:].float().max()) == 3.0, tag     # nothing written past the batch
    # the workspace route (float32, from 2^17 cubes; smaller batches and other formats must take the one-launch kernel unchanged)
    lib = L.lib()
    for fmt, dt in fmts:
        need = lib.rc_workspace_bytes(L.OP_STEP, cs, n, fmt)
        assert (need > 0) == (fmt != L.FMT_U8 and n >= 1 << 17), (n, fmt, need)       # (u8: from 2^22 cubes, test_dense_outputs_u8_4m_workspace)
        oh = torch.full((n, 20, 24), 3, dtype=dt, device="cuda")
        dst = torch.zeros_like(src)
        rew = torch.zeros(n, dtype=torch.float32, device="cuda")
        done = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
        ops.apply_moves(src, dst, a_d, n, cs, rew, done, oh, fmt)                         # default dispatch: with a workspace where one is used
        tag = (n, "ws" if need else "one launch", str(dt))
        assert torch.equal(oh.float().argmax(-1).to(torch.uint8), exp_code_t) and float(oh.float().sum()) == 20.0 * n, tag
        assert (ops.to_aos(dst, n).cpu().numpy() == exp_st).all(), tag
        assert (done.cpu().numpy() == exp_done).all() and (rew.cpu().numpy() == exp_rew).all(), tag
        if need:
            ref = torch.full((n, 20, 24), 3, dtype=dt, device="cuda")
            dst2 = torch.zeros_like(src)
            ops.apply_moves(src, dst2, a_d, n, cs, rew, done, ref, fmt, variant=200000)   # the one-launch kernel
            assert torch.equal(ref, oh) and torch.equal(dst2, dst), tag
            work = src.clone()
            oh.fill_(3)
            ops.apply_moves(work, work, a_d, n, cs, None, done, oh, fmt)                  # in place, with the workspace
            assert torch.equal(work, dst) and torch.equal(ref, oh), tag
            # encode-only with the workspace (ops.encode = rc_encode_ws): the one-hot of the MOVED states, no state written
            enc = torch.full((n, 20, 24), 3, dtype=dt, device="cuda")
            keep = dst.clone()
            ops.encode(dst, n, cs, enc, fmt)
            assert torch.equal(enc, ref) and torch.equal(dst, keep), tag
            # a workspace that is too small (or NULL) falls back to the one-launch kernel: same results
            ws = torch.empty(need - 16, dtype=torch.uint8, device="cuda")
            oh.fill_(3)
            p_in, p_out = src.shape[-1], dst2.shape[-1]
            L.check(lib.rc_apply_moves_ws(L.ptr(src), L.ptr(dst2), L.ptr(a_d), n, p_in, p_out, cs, L.ptr(rew), L.ptr(done), L.ptr(oh), fmt, 0,
                                          L.ptr(ws), ws.numel(), L.stream_ptr(src.device)))
            assert torch.equal(ref, oh), tag
            oh.fill_(3)
            L.check(lib.rc_apply_moves_ws(L.ptr(src), L.ptr(dst2), L.ptr(a_d), n, p_in, p_out, cs, L.ptr(rew), L.ptr(done), L.ptr(oh), fmt, 0,
                                          None, 0, L.stream_ptr(src.device)))
            assert torch.equal(ref, oh), tag
    assert L.read_status() == 0


def test_dense_outputs_beyond_4gib(ops, L):
    """Dense one-hot streams larger than 4 GiB: 2^22 cubes as float32 (8 GB) through the wide code -> dense writer and through the
    fused step; every cube's arg-max against the compact code of the same states, one 1 per row (py333.py:235-246)."""
    cs, n = 3, 1 << 22
    st = ops.alloc_states(n, cs, "cuda")
    ops.fill_solved(st, n, cs)
    ops.scramble(st, n, cs, 9, seed=4)
    code = ops.alloc_code(n, cs, "cuda")
    ops.encode(st, n, cs, code, L.FMT_CODE)
    want = ops.to_aos(code, n)
    oh = torch.empty((n, 20, 24), dtype=torch.float32, device="cuda")
    assert oh.numel() * 4 > (1 << 32)
    for how in ("code_to_dense", "fused"):
        oh.fill_(3.0)
        if how == "fused":
            ops.apply_moves(st, st, torch.full((n,), 12, dtype=torch.uint8, device="cuda"), n, cs, None, None, oh, L.FMT_F32)   # the no-op action
        else:
            ops.onehot_from_code(code, n, cs, oh)
        for lo_ in range(0, n, 1 << 20):                                  # chunked: keeps the temporaries small
            blk = oh[lo_:lo_ + (1 << 20)]
            assert torch.equal(blk.argmax(-1).to(torch.uint8), want[lo_:lo_ + (1 << 20)]), (how, lo_)
            assert float(blk.sum()) == 20.0 * blk.shape[0] and float(blk.max()) == 1.0, (how, lo_)
    assert L.read_status() == 0


def test_dense_u8_4m_workspace_route(ops, L):
    """u8 dense steps take the workspace route from 2^22 cubes (their state ping-pong no longer fits the Infinity Cache): the
    two-launch result equals the one-launch kernel's, byte for byte (states, flags, reward, one-hot)."""
    cs, n = 3, 1 << 22
    assert L.lib().rc_workspace_bytes(L.OP_STEP, cs, n, L.FMT_U8) == 20 * n
    st = ops.alloc_states(n, cs, "cuda")
    ops.fill_solved(st, n, cs)
    ops.scramble(st, n, cs, 7, seed=11)
    acts = torch.randint(0, 13, (n,), dtype=torch.uint8, device="cuda")              # incl. the no-op action 12
    outs = []
    for variant in (0, 200000):                                                        # workspace route / one-launch kernel
        dst = torch.zeros_like(st)
        oh = torch.full((n, 20, 24), 3, dtype=torch.uint8, device="cuda")
        rew = torch.zeros(n, dtype=torch.float32, device="cuda")
        done = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
        ops.apply_moves(st, dst, acts, n, cs, rew, done, oh, L.FMT_U8, variant=variant)
        outs.append((dst, oh, rew, done))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert int(outs[0][1].sum()) == 20 * n and L.read_status() == 0


def test_workspace_route_under_hipgraph_capture(ops, L):
    """ops.apply_moves with a dense float32 output on 2^17 cubes takes the workspace route; captured into a hipGraph the workspace
    is allocated from the graph's pool (never from the per-stream cache) and every replay reproduces the eager result."""
    cs, n = 3, 1 << 17
    st = ops.alloc_states(n, cs, "cuda")
    ops.fill_solved(st, n, cs)
    ops.scramble(st, n, cs, 11, seed=5)
    acts = torch.randint(0, 12, (n,), dtype=torch.uint8, device="cuda")
    dst_e, dst_g = torch.zeros_like(st), torch.zeros_like(st)
    oh_e = torch.empty((n, 20, 24), dtype=torch.float32, device="cuda")
    oh_g = torch.full((n, 20, 24), 3.0, dtype=torch.float32, device="cuda")
    done = torch.empty(n, dtype=torch.uint8, device="cuda")
    ops.apply_moves(st, dst_e, acts, n, cs, None, done, oh_e, L.FMT_F32)
    cached = dict(ops._workspaces)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.apply_moves(st, dst_g, acts, n, cs, None, done, oh_g, L.FMT_F32)      # warm-up on the capture stream
        with torch.cuda.graph(g, stream=side):
            ops.apply_moves(st, dst_g, acts, n, cs, None, done, oh_g, L.FMT_F32)
    torch.cuda.current_stream().wait_stream(side)
    assert all(k in ops._workspaces and ops._workspaces[k] is v for k, v in cached.items())   # the capture added nothing it keeps
    for _ in range(3):
        oh_g.fill_(3.0)
        dst_g.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(oh_g, oh_e) and torch.equal(dst_g, dst_e)
    assert L.read_status() == 0
