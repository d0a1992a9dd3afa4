"""Host-side logic on CPU: tables, the C-ABI surface, CubeEnv's Python semantics (with a test-only
oracle backend), sharding helpers incl. a world_size-2 gloo run.  No GPU compute is called."""
import copy
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------------------- tables
def test_tables_match_reference_golden(golden):
    import rubiks_cube_solver_amd as r
    g, t = golden("tables_333"), r.get_tables(3)
    assert (t.perm == g["moveDefs"]).all()                      # derived from geometry == reference table
    assert (t.corner_defs == g["corner_pieceDefs"]).all() and (t.edge_defs == g["edge_pieceDefs"]).all()
    assert (t.corner_lut == g["corner_pieceInds"]).all() and (t.edge_lut == g["edge_pieceInds"]).all()
    assert (t.solved == g["initState"]).all()
    assert list(t.action_names) == list(g["action_names"])
    assert list(t.state_dim) == list(g["state_dim"]) and t.n_actions == int(g["action_dim"])
    assert (t.corner_code[:62] == g["corner_pieceInds"][:, 0] * 3 + g["corner_pieceInds"][:, 1]).all()
    assert (t.edge_code[:55] == g["edge_pieceInds"][:, 0] * 2 + g["edge_pieceInds"][:, 1]).all()
    assert not t.corner_code[62:].any() and not t.edge_code[55:].any()


def test_tables_222_match_oracle_py222():
    import rubiks_cube_solver_amd as r
    from oracle.oracle_np import tables_222
    t, o = r.get_tables(2), tables_222()
    assert (t.perm == o["perm"]).all() and (t.corner_defs == o["corner_defs"]).all()
    assert (t.corner_lut == o["corner_lut"]).all()
    assert r.get_env_config(2) == ([7, 21], 6) and r.get_env_config(3) == ([20, 24], 12)
    with pytest.raises(NotImplementedError):
        r.get_env_config(4)


def test_move_group_properties():
    import rubiks_cube_solver_amd as r
    for cs in (2, 3):
        p = r.get_tables(cs).perm.astype(int)
        S = p.shape[1]
        ident = np.arange(S)
        for a in range(0, len(p), 2):
            assert (p[a][p[a + 1]] == ident).all()              # X' undoes X
            q = ident
            for _ in range(4):
                q = q[p[a]]
            assert (q == ident).all()                           # order 4
            assert (p[a] != ident).sum() == (20 if cs == 3 else 12)
        s = ident
        for _ in range(6):                                      # (R U R' U')^6 = identity
            for a in (4, 0, 5, 1):
                s = s[p[a]]
        assert (s == ident).all()
    p3 = r.get_tables(3).perm.astype(int)
    assert all((p3[a][[4, 13, 22, 31, 40, 49]] == [4, 13, 22, 31, 40, 49]).all() for a in range(12))  # centres fixed
    assert (p3[0][p3[6]] == p3[6][p3[0]]).all()                 # opposite faces commute (U, D)


def test_pair_tables_reproduce_the_reference_luts(golden, oracle):
    """The two-colour code tables the kernels use (tables.pair_tables -> rc_tables.h) against the reference's own outputs:
    every step of fixture G3's 1000 walks (reachable states: corners and edges) and fixture G7's arbitrary colourings
    (edges only: the edge table is exact for any colouring), plus 2x2x2 walks against the oracle."""
    import rubiks_cube_solver_amd as r
    from rubiks_cube_solver_amd.tables import SIGMAS, pair_tables
    t = r.get_tables(3)
    epair, cpair, cid = pair_tables(3)
    assert len(cpair) == 2 and cid.shape == (8, 6)
    w = golden("walks_333")
    st, cols = w["stickers"].reshape(-1, 54), w["cols"].reshape(-1, 20)
    for q in range(8):
        for sid, sg in enumerate(SIGMAS):
            c = [st[:, t.corner_defs[q][j]] for j in sg]
            literal = t.corner_code[c[0] + 2 * c[1] + 10 * c[2]]
            assert (cpair[cid[q, sid]][c[1], c[0]] == literal).all(), (q, sid)
            if sid == 0:
                assert (literal == cols[:, q]).all()
    for e in range(12):
        c0, c1 = st[:, t.edge_defs[e][0]], st[:, t.edge_defs[e][1]]
        assert (epair[c1, c0] == cols[:, 8 + e]).all() and (epair[c0, c1] == t.edge_code[c1 + 10 * c0]).all()
    g7 = golden("encode_333")
    for e in range(12):
        c0, c1 = g7["stickers"][:, t.edge_defs[e][0]], g7["stickers"][:, t.edge_defs[e][1]]
        assert (epair[c1, c0] == g7["cols"][:, 8 + e]).all()
    t2 = r.get_tables(2)
    _, cpair2, cid2 = pair_tables(2)
    ex = oracle.adi(2, 3000, 14, seed=3, stream=0, want_children=False)
    st2, code2 = ex["parents"].reshape(-1, 24), ex["parent_code"].reshape(-1, 7)
    for q in range(7):
        c0, c1 = st2[:, t2.corner_defs[q][0]], st2[:, t2.corner_defs[q][1]]
        assert (cpair2[cid2[q, 0]][c1, c0] == code2[:, q]).all(), q


def test_generated_header_is_current():
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_tables.py"), "--check"]).returncode == 0


# ------------------------------------------------------------------------------ C ABI
def test_abi_exports_every_declared_symbol():
    from rubiks_cube_solver_amd import _lib
    L = _lib.lib()                                              # loads without a GPU
    header = open(os.path.join(ROOT, "include", "rubikhip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char \*)\s*(rc_\w+)\(", header, re.M))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(L, name), name
    assert L.rc_version() >= 600 and len(declared) == 38 and {"rc_build_id", "rc_onehot_from_code_blocks", "rc_describe_dispatch", "rc_facade_release", "rc_apply_moves_ws", "rc_encode_ws", "rc_workspace_bytes",
                                                              "rc_adi_generate_family", "rc_family_layout", "rc_onehot_from_family", "rc_onehot_from_family_depths",
                                                              "rc_adi_targets_depths", "rc_legacy_scramble_actions_ex", "rc_host_alias", "rc_scramble_from", "rc_search_pack"} <= declared
    # the binary is the tree's sources: the id embedded at build time (rc_build_id) = the hash of rubikhip.hip + rc_device.h + rc_tables.h +
    # rubikhip.h as they are on disk (a stale library would not even have loaded: _lib.lib() refuses it)
    from rubiks_cube_solver_amd import _build
    assert _lib.build_id() == _build.source_hash(_build.HIP_SOURCES) == _build.embedded_id(_lib.LIB_PATH) and len(_lib.build_id()) == 16
    # every rc_* the library exports is declared in the header, and nothing else leaves it
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in nm.splitlines() if " T " in l}
    assert {e for e in exported if e.startswith("rc_")} == declared, exported ^ declared
    # the workspace query needs no GPU: 20 bytes per cube in 32768-cube tiles for large 3x3x3 dense steps (f32 / 16-bit from 2^17
    # cubes, u8 from 2^22), 0 where no workspace is used (small batches, 2x2x2, the compact code, other operations)
    W = lambda *a: L.rc_workspace_bytes(*a)
    assert W(_lib.OP_STEP, 3, 1 << 20, _lib.FMT_F32) == 20 << 20 and W(_lib.OP_STEP, 3, (1 << 17) + 1, _lib.FMT_F32) == 20 * ((1 << 17) + 32768)
    assert W(_lib.OP_STEP, 3, 1 << 16, _lib.FMT_F32) == 0 and W(_lib.OP_STEP, 2, 1 << 20, _lib.FMT_F32) == 0
    assert W(_lib.OP_STEP, 3, 1 << 20, _lib.FMT_BF16) == 20 << 20 and W(_lib.OP_STEP, 3, 1 << 20, _lib.FMT_F16) == 20 << 20
    assert W(_lib.OP_STEP, 3, 1 << 20, _lib.FMT_U8) == 0 and W(_lib.OP_STEP, 3, 1 << 22, _lib.FMT_U8) == 20 << 22
    assert W(_lib.OP_STEP, 3, 1 << 20, _lib.FMT_CODE) == 0 and W(_lib.OP_EXPAND, 3, 1 << 20, _lib.FMT_F32) == 0
    # include/rubiktree.h <-> librubiktree.so (host-side trees of the lockstep search)
    from rubiks_cube_solver_amd import _tree
    T = _tree.tree_lib()
    header = open(os.path.join(ROOT, "include", "rubiktree.h")).read()
    declared = set(re.findall(r"^(?:int|void|rc_tree \*)\s*(rc_tree_\w+)\(", header, re.M))
    assert len(declared) == 11, declared
    for name in declared:
        assert hasattr(T, name), name


def test_abi_tables_equal_package_tables():
    import rubiks_cube_solver_amd as r
    from rubiks_cube_solver_amd import _lib
    for cs in (2, 3):
        t, p = _lib.get_tables(cs), r.get_tables(cs)
        assert (t["perm"] == p.perm).all() and (t["corner_code"] == p.corner_code).all()
        assert (t["edge_code"] == p.edge_code).all() and (t["corner_defs"] == p.corner_defs).all()
        assert t["dims"][:2] == (p.n_stickers, p.n_actions)
    assert _lib.lib().rc_get_tables(4, None, None, None, None, None, None, None) == -1
    assert b"cube_size" in _lib.lib().rc_last_error()


def test_dispatch_description_and_enodev_without_gpu():
    """rc_describe_dispatch runs the library's own dispatch functions on the host (no GPU needed): the shapes the bench labels
    its records with.  Launching entry points return RC_ENODEV while rc_init has not succeeded for the current device."""
    from rubiks_cube_solver_amd import _lib as L
    st = L.OUT_STATES | L.OUT_DONE
    assert L.describe(L.OP_STEP, 3, 1 << 22, outputs=st).startswith("k_step<Cube3,V=2,move,store,POL=1> grid=8192")
    assert "POL=2" in L.describe(L.OP_STEP, 3, 1 << 24, outputs=st) and "POL=0" in L.describe(L.OP_STEP, 3, 1 << 20, outputs=st)
    assert "POL=0" in L.describe(L.OP_STEP, 3, 1 << 22, outputs=st | L.OUT_INPLACE)             # 226 MB in place: resident
    assert "POL=3" in L.describe(L.OP_STEP, 3, 1 << 22, outputs=st | L.OUT_INPLACE, fmt=L.FMT_CODE)   # + 84 MB of code: streamed past
    assert "V=2,move,store,code,POL=1" in L.describe(L.OP_STEP, 3, 1 << 22, outputs=st | L.OUT_REWARD, fmt=L.FMT_CODE)
    assert L.describe(L.OP_STEP, 2, 1 << 22, outputs=st).startswith("k_step<Cube2,")
    assert L.describe(L.OP_STEP, 3, 1 << 20, outputs=st, fmt=L.FMT_BF16).startswith("k_step_dense<Cube3,bf16,move,store,TILE=64> grid=16384")
    assert L.describe(L.OP_STEP, 3, 1 << 20, outputs=st, fmt=L.FMT_U8).startswith("k_step_dense<Cube3,u8,move,store,TILE=256> grid=4096")
    assert L.describe(L.OP_STEP, 3, 1 << 20, outputs=st, fmt=L.FMT_BF16, variant=200000).startswith("k_step_dense<Cube3,bf16,move,store,TILE=256> grid=4096")
    assert "TILE=64" in L.describe(L.OP_STEP, 3, 1 << 20, outputs=st, fmt=L.FMT_BF16, variant=100000)
    assert L.describe(L.OP_STEP, 2, 1 << 20, outputs=st, fmt=L.FMT_BF16).startswith("k_step_dense<Cube2,bf16,move,store,TILE=256>")
    assert L.describe(L.OP_STEP, 3, 4096, outputs=0, fmt=L.FMT_F32).startswith("k_step_dense<Cube3,f32,encode,TILE=64>")
    assert L.describe(L.OP_CODE_TO_DENSE, 3, 1 << 20, fmt=L.FMT_F32, variant=300000).startswith("k_code_to_dense_wide<Cube3,f32> tiles_per_group=37 grid=111")
    # large 3x3x3 batches: the front writer, one 3840-byte pass per workgroup (2 / 4 / 8 cubes), one front per XCD; f32 gathers its code
    # bytes per lane, the 1- and 2-byte formats fetch them once per workgroup through LDS, u8 with two fronts per XCD per workgroup
    assert L.describe(L.OP_CODE_TO_DENSE, 3, 1 << 20, fmt=L.FMT_F32).startswith("k_code_to_dense_front<Cube3,f32,F=1,gather> cubes_per_pass=2 xcd grid=524288")
    assert L.describe(L.OP_CODE_TO_DENSE, 3, 1 << 20, fmt=L.FMT_BF16).startswith("k_code_to_dense_front<Cube3,bf16,F=1,lds> cubes_per_pass=4 xcd grid=262144")
    assert L.describe(L.OP_CODE_TO_DENSE, 3, 1 << 20, fmt=L.FMT_U8).startswith("k_code_to_dense_front<Cube3,u8,F=2,lds> cubes_per_pass=8 xcd grid=65536")
    assert L.describe(L.OP_CODE_TO_DENSE, 3, 1 << 20, fmt=L.FMT_BF16, variant=20).startswith("k_code_to_dense_front<Cube3,bf16,F=1,lds> cubes_per_pass=4 grid=262144")
    assert L.describe(L.OP_CODE_TO_DENSE, 2, 1 << 20, fmt=L.FMT_F32).startswith("k_code_to_dense<Cube2,f32,TILE=64> grid=2048 block=640")   # round 6: dense_write_222 on 640 threads
    assert L.describe(L.OP_CODE_TO_DENSE, 2, 1 << 20, fmt=L.FMT_BF16).startswith("k_code_to_dense<Cube2,bf16,TILE=64> grid=16384 block=320")   # the 147-chunk pass writer
    assert L.describe(L.OP_CODE_TO_DENSE, 2, 1 << 20, fmt=L.FMT_U8).startswith("k_code_to_dense<Cube2,u8,TILE=256> grid=4096 block=320")
    # thresholds of the front writer: 2^15 (f32), 2^16 (16-bit), 2^18 (u8); 64-cube tiles below
    assert L.describe(L.OP_CODE_TO_DENSE, 3, 1 << 14, fmt=L.FMT_F32).startswith("k_code_to_dense<Cube3,f32,TILE=64> grid=256")
    assert L.describe(L.OP_CODE_TO_DENSE, 3, 1 << 15, fmt=L.FMT_F32).startswith("k_code_to_dense_front<Cube3,f32,F=1,gather>")
    assert L.describe(L.OP_CODE_TO_DENSE, 3, 1 << 15, fmt=L.FMT_BF16).startswith("k_code_to_dense<Cube3,bf16,TILE=64> grid=512")
    assert L.describe(L.OP_CODE_TO_DENSE, 3, 1 << 16, fmt=L.FMT_BF16).startswith("k_code_to_dense_front<Cube3,bf16,F=1,lds>")
    assert L.describe(L.OP_CODE_TO_DENSE, 3, 1 << 17, fmt=L.FMT_U8).startswith("k_code_to_dense<Cube3,u8,TILE=64>")
    assert L.describe(L.OP_CODE_TO_DENSE, 3, 1 << 18, fmt=L.FMT_U8).startswith("k_code_to_dense_front<Cube3,u8,F=2,lds>")
    # rc_apply_moves_ws with its workspace: step + code, then the front writer; without (or below 2^17 cubes) the one-launch kernel
    d = L.describe(L.OP_STEP, 3, 1 << 20, outputs=st | L.OUT_REWARD | L.OUT_WORKSPACE, fmt=L.FMT_F32)
    assert d.startswith("k_step<Cube3,V=2,move,store,code,POL=0> grid=2048 block=64 + k_code_to_dense_front<Cube3,f32,F=1,gather> xcd grid=524288"), d
    d = L.describe(L.OP_STEP, 3, 1 << 20, outputs=st | L.OUT_REWARD | L.OUT_WORKSPACE, fmt=L.FMT_BF16)
    assert d.startswith("k_step<Cube3,V=2,move,store,code,POL=0> grid=2048 block=64 + k_code_to_dense_front<Cube3,bf16,F=1,lds> xcd grid=262144"), d
    assert L.describe(L.OP_STEP, 3, 1 << 16, outputs=st | L.OUT_WORKSPACE, fmt=L.FMT_F32).startswith("k_step_dense<Cube3,f32,move,store,TILE=64>")
    assert L.describe(L.OP_STEP, 3, 1 << 20, outputs=st | L.OUT_WORKSPACE, fmt=L.FMT_U8).startswith("k_step_dense<Cube3,u8,move,store,TILE=256> grid=4096")
    assert L.describe(L.OP_STEP, 2, 1 << 20, outputs=st | L.OUT_WORKSPACE, fmt=L.FMT_F32).startswith("k_step_dense<Cube2,f32")
    assert L.describe(L.OP_STEP, 3, 1 << 18, outputs=st, fmt=L.FMT_BF16).startswith("k_step_dense<Cube3,bf16,move,store,TILE=256>")   # no workspace: 16-bit 64-cube tiles from 2^19
    assert "code,POL=4" in L.describe(L.OP_STEP, 3, 1 << 22, outputs=st | L.OUT_REWARD | L.OUT_WORKSPACE, fmt=L.FMT_F32)      # beyond the cache: code kept
    d = L.describe(L.OP_STEP, 3, 1 << 20, outputs=L.OUT_WORKSPACE, fmt=L.FMT_F32)                                              # rc_encode_ws
    assert d.startswith("k_step<Cube3,V=2,code,POL=0> grid=2048 block=64 + k_code_to_dense_front<Cube3,f32,F=1,gather> xcd"), d
    assert L.describe(L.OP_EXPAND, 3, 1 << 20, outputs=L.OUT_STATES | L.OUT_FLAGS).startswith("k_expand_stream<Cube3> grid=512")
    assert L.describe(L.OP_EXPAND, 3, 1 << 20, outputs=L.OUT_STATES | L.OUT_FLAGS, variant=800).startswith("k_expand<Cube3,V=2> parts=1 grid=2048")
    assert L.describe(L.OP_EXPAND, 3, 4096, outputs=L.OUT_STATES | L.OUT_FLAGS).startswith("k_expand<Cube3,V=1>")
    assert L.describe(L.OP_ADI, 3, 100000, 30, outputs=L.OUT_STATES | L.OUT_FLAGS).startswith("k_adi<Cube3,V=2> parts=1 segs=1 grid=196")
    assert L.describe(L.OP_ADI, 3, 100000, 30, outputs=L.OUT_CODE | L.OUT_FLAGS).startswith("k_adi<Cube3,V=2,code> parts=1 segs=3 grid=588")
    assert "segs=5 " in L.describe(L.OP_ADI, 3, 100000, 5, outputs=L.OUT_CODE, variant=9001001)       # clamped to the depth
    assert "V=1," in L.describe(L.OP_STEP, 3, 1 << 22, outputs=st, variant=1) and "POL=0" in L.describe(L.OP_STEP, 3, 1 << 22, outputs=st, variant=20)
    # the ADI pipeline's block writer: 13 blocks per depth, several depths per launch, the front writer's shapes per format
    assert L.describe(L.OP_FAMILY_TO_DENSE, 3, 43008, 1, fmt=L.FMT_F32) == "k_code_to_dense_front<Cube3,f32,F=1,gather,family> depths=1 cubes_per_pass=2 xcd grid=21504x13 block=256"
    assert L.describe(L.OP_FAMILY_TO_DENSE, 3, 200, 30, fmt=L.FMT_BF16).startswith("k_code_to_dense_front<Cube3,bf16,F=1,lds,family> depths=30 cubes_per_pass=4 xcd grid=56x390 ")
    assert L.describe(L.OP_FAMILY_TO_DENSE, 3, 200, 30, fmt=L.FMT_U8).startswith("k_code_to_dense_front<Cube3,u8,F=2,lds,family> depths=30 cubes_per_pass=8 xcd grid=16x390 ")
    for bad in (dict(cube_size=2, fmt=L.FMT_F32), dict(cube_size=3, fmt=L.FMT_CODE), dict(cube_size=3, fmt=L.FMT_F32, variant=1), dict(cube_size=3, fmt=L.FMT_F32, depth=0)):
        with pytest.raises(L.RubikHipError):
            L.describe(L.OP_FAMILY_TO_DENSE, bad["cube_size"], 100, bad.get("depth", 2), fmt=bad["fmt"], variant=bad.get("variant", 0))
    with pytest.raises(L.RubikHipError):
        L.describe(99, 3, 10)
    with pytest.raises(L.RubikHipError):
        L.describe(L.OP_ADI, 3, 10, 0)
    if not torch.cuda.is_available():
        lib = L.lib()
        assert lib.rc_fill_solved(None, 1, 256, 3, None) == -3 and b"rc_init" in lib.rc_last_error() or b"device" in lib.rc_last_error()
        out = ctypes.c_uint32(0)
        assert lib.rc_read_status(ctypes.byref(out), None) == -3


def test_no_cpu_fallback_without_gpu():
    """Compute entry points refuse host tensors / a missing device instead of falling back."""
    from rubiks_cube_solver_amd import _lib, ops
    from rubiks_cube_solver_amd.vec_env import VecCubeEnv
    with pytest.raises(_lib.RubikHipError):
        ops.fill_solved(torch.zeros((54, 256), dtype=torch.uint8), 10, 3)
    with pytest.raises(_lib.RubikHipError):
        VecCubeEnv(4, "cpu", 3)
    from rubiks_cube_solver_amd.replay import TensorReplayBuffer
    with pytest.raises(_lib.RubikHipError):
        TensorReplayBuffer(16, 8, cube_size=3, device="cpu")      # the sink lives on the device: no host-side twin
    src = open(os.path.join(ROOT, "rubiks-cube-solver_amd", "ops.py")).read()
    assert "oracle" not in src
    for f in os.listdir(os.path.join(ROOT, "rubiks-cube-solver_amd")):
        if f.endswith(".py"):
            txt = open(os.path.join(ROOT, "rubiks-cube-solver_amd", f)).read()
            assert "import oracle" not in txt and "from oracle" not in txt, f


def test_family_layout_table():
    """The FAMILY record's layout (rc_family_layout, no GPU needed): 51 | 15 shared look-ups; the parent's 20 | 7 slot codes are
    distinct rows; every row is some (child, slot)'s code; a child's slot reads the row of the parent slot its cubie came from."""
    from rubiks_cube_solver_amd import _lib
    from rubiks_cube_solver_amd.tables import get_tables
    for cs, nf_want in ((3, 51), (2, 15)):
        nf, rows = _lib.family_layout(cs)
        t = get_tables(cs)
        A, SL, NC = len(t.perm), rows.shape[1], len(t.corner_defs)
        assert nf == nf_want and rows.shape == (A + 1, SL)
        assert len(set(rows[A].tolist())) == SL                                   # the parent's slots: SL different look-ups
        assert set(rows.reshape(-1).tolist()) == set(range(nf))                   # no unused row, none out of range
        assert (rows[:, :NC] < rows[A, :NC].max() + 6).all() and (rows[:, NC:] > rows[:, :NC].max()).all() if SL > NC else True
        for a in range(A):                                                        # slots a turn does not touch keep the parent's row
            moved = set(int(i) for i in np.nonzero(np.asarray(t.perm[a]) != np.arange(len(t.perm[a])))[0])
            for p_, d in enumerate(list(t.corner_defs) + list(t.edge_defs)):
                if not (set(int(x) for x in d) & moved):
                    assert rows[a, p_] == rows[A, p_], (cs, a, p_)


def _variant_macros():
    """RC_VARIANT_* of include/rubikhip.h as Python callables (the bodies are plain integer arithmetic)."""
    src = open(os.path.join(ROOT, "include", "rubikhip.h")).read()
    out = {}
    for name, arg, body in re.findall(r"^#define (RC_VARIANT_\w+)\((\w+)\) (.+?)\s*(?:/\*.*)?$", src, re.M):
        out[name] = eval(f"lambda {arg}: {body}")
    for name, body in re.findall(r"^#define (RC_VARIANT_\w+) (\d+)\s", src, re.M):
        out[name] = int(body)
    return out


def test_variant_macros_round_trip_through_describe_dispatch():
    """Every RC_VARIANT_* macro of the header, through rc_describe_dispatch (no GPU needed): the field shows up in the dispatch of ITS
    entry point and is rejected by every entry point whose group does not define it."""
    from rubiks_cube_solver_amd import _lib as L
    M = _variant_macros()
    assert set(M) == {"RC_VARIANT_STEP_PACK", "RC_VARIANT_STEP_POLICY", "RC_VARIANT_STEP_DENSE_TILE", "RC_VARIANT_EXPAND_PACK", "RC_VARIANT_EXPAND_STREAM",
                      "RC_VARIANT_EXPAND_PARTS", "RC_VARIANT_ADI_PACK", "RC_VARIANT_ADI_PARTS", "RC_VARIANT_ADI_SEGS", "RC_VARIANT_DENSE_FORM",
                      "RC_VARIANT_DENSE_WIDE_GROUPS16", "RC_VARIANT_DENSE_WIDE_SKEW", "RC_VARIANT_DENSE_FRONTS", "RC_VARIANT_DENSE_FRONT_FETCH",
                      "RC_VARIANT_LEGACY_LDS", "RC_VARIANT_LEGACY_STREAM"}
    st = L.OUT_STATES | L.OUT_DONE
    n = 1 << 20
    step = lambda v, **kw: L.describe(L.OP_STEP, 3, n, outputs=st, variant=v, **kw)
    expand = lambda v: L.describe(L.OP_EXPAND, 3, n, outputs=L.OUT_STATES | L.OUT_FLAGS, variant=v)
    adi = lambda v: L.describe(L.OP_ADI, 3, 100000, 30, outputs=L.OUT_CODE | L.OUT_FLAGS, variant=v)
    dense = lambda v, fmt=L.FMT_BF16: L.describe(L.OP_CODE_TO_DENSE, 3, n, fmt=fmt, variant=v)
    for v in (1, 2):
        assert f"V={v}," in step(M["RC_VARIANT_STEP_PACK"](v)) and f"V={v}" in expand(M["RC_VARIANT_EXPAND_PACK"](v) + M["RC_VARIANT_EXPAND_STREAM"](8))
        assert f"V={v}," in adi(M["RC_VARIANT_ADI_PACK"](v))
    for p, pol in ((1, 2), (2, 0), (3, 1), (4, 3)):
        assert f"POL={pol}>" in step(M["RC_VARIANT_STEP_POLICY"](p))
    assert "TILE=64>" in step(M["RC_VARIANT_STEP_DENSE_TILE"](1), fmt=L.FMT_BF16) and "TILE=256>" in step(M["RC_VARIANT_STEP_DENSE_TILE"](2), fmt=L.FMT_BF16)
    for h, waves in ((1, 128), (5, 512), (7, 1024)):
        assert expand(M["RC_VARIANT_EXPAND_STREAM"](h)).startswith(f"k_expand_stream<Cube3> grid={waves}")
    assert expand(M["RC_VARIANT_EXPAND_STREAM"](8)).startswith("k_expand<")
    for parts in (1, 3, 12):
        assert f"parts={parts} " in expand(M["RC_VARIANT_EXPAND_PARTS"](parts)) and f"parts={parts} " in adi(M["RC_VARIANT_ADI_PARTS"](parts) + M["RC_VARIANT_ADI_PACK"](1))
    for segs in (1, 4, 16):
        assert f"segs={segs} " in adi(M["RC_VARIANT_ADI_SEGS"](segs))
    assert "k_code_to_dense<Cube3,bf16,TILE=64>" in dense(M["RC_VARIANT_DENSE_FORM"](1)) and "TILE=256>" in dense(M["RC_VARIANT_DENSE_FORM"](2))
    assert dense(M["RC_VARIANT_DENSE_FORM"](3) + M["RC_VARIANT_DENSE_WIDE_GROUPS16"](4) + M["RC_VARIANT_DENSE_WIDE_SKEW"](3)).startswith("k_code_to_dense_wide<Cube3,bf16> tiles_per_group=")
    assert "grid=64 " in dense(M["RC_VARIANT_DENSE_FORM"](3) + M["RC_VARIANT_DENSE_WIDE_GROUPS16"](4))
    for f in (1, 2, 4):
        assert f"k_code_to_dense_front<Cube3,bf16,F={f}," in dense(M["RC_VARIANT_DENSE_FORM"](4) + M["RC_VARIANT_DENSE_FRONTS"](f))
    assert ",gather>" in dense(M["RC_VARIANT_DENSE_FRONT_FETCH"](3)) and ",lds>" in dense(M["RC_VARIANT_DENSE_FRONT_FETCH"](4), L.FMT_F32)
    assert " xcd " not in dense(M["RC_VARIANT_DENSE_FRONT_FETCH"](2)) and " xcd " in dense(0)
    # fields of another group, and digits no group defines, are rejected -- by the describing call here, by the launchers on the GPU
    wrong = [(step, M["RC_VARIANT_ADI_SEGS"](2)), (step, M["RC_VARIANT_EXPAND_STREAM"](3)), (step, M["RC_VARIANT_EXPAND_PARTS"](2)), (step, M["RC_VARIANT_DENSE_FORM"](3)),
             (step, M["RC_VARIANT_DENSE_FORM"](4)), (step, 3), (step, M["RC_VARIANT_STEP_POLICY"](5)), (expand, M["RC_VARIANT_STEP_POLICY"](1)),
             (expand, M["RC_VARIANT_EXPAND_STREAM"](9)), (expand, M["RC_VARIANT_EXPAND_PARTS"](13)), (expand, M["RC_VARIANT_DENSE_FORM"](1)), (expand, M["RC_VARIANT_ADI_SEGS"](1)),
             (adi, M["RC_VARIANT_STEP_POLICY"](2)), (adi, M["RC_VARIANT_EXPAND_STREAM"](1)), (adi, M["RC_VARIANT_ADI_SEGS"](17)), (adi, M["RC_VARIANT_ADI_PARTS"](13)),
             (adi, M["RC_VARIANT_DENSE_FORM"](2)), (dense, M["RC_VARIANT_ADI_SEGS"](1)), (dense, M["RC_VARIANT_EXPAND_STREAM"](1)), (dense, M["RC_VARIANT_DENSE_FORM"](5)),
             (dense, M["RC_VARIANT_DENSE_FORM"](4) + 3), (dense, M["RC_VARIANT_DENSE_FORM"](4) + M["RC_VARIANT_DENSE_FRONT_FETCH"](1)),
             (dense, M["RC_VARIANT_DENSE_FORM"](1) + 1), (dense, M["RC_VARIANT_DENSE_FORM"](3) + 1), (dense, M["RC_VARIANT_DENSE_FORM"](4) + M["RC_VARIANT_DENSE_WIDE_GROUPS16"](2)),
             (step, -1), (step, 100000000), (adi, 1 << 30)]
    for fn, v in wrong:
        with pytest.raises(L.RubikHipError, match="variant"):
            fn(v)
    # 2x2x2: six actions bound the parts field
    assert "parts=6 " in L.describe(L.OP_EXPAND, 2, n, outputs=L.OUT_STATES, variant=M["RC_VARIANT_EXPAND_PARTS"](6))
    with pytest.raises(L.RubikHipError, match="variant"):
        L.describe(L.OP_EXPAND, 2, n, outputs=L.OUT_STATES, variant=M["RC_VARIANT_EXPAND_PARTS"](7))
    assert M["RC_VARIANT_LEGACY_LDS"] == 1 and M["RC_VARIANT_LEGACY_STREAM"](0) == 2 and M["RC_VARIANT_LEGACY_STREAM"](40) == 642 and M["RC_VARIANT_LEGACY_STREAM"](623) == 9970


def test_cube_env_has_no_backend_parameter():
    """The facade has ONE implementation: CubeEnv takes no backend argument and carries no branch for one (round 4's `_backend=`
    seam is gone); the CPU tests below subclass it in tests/fake_backend.py and override its device hooks instead."""
    import inspect
    from rubiks_cube_solver_amd.cube_env import CubeEnv, make_env
    assert list(inspect.signature(CubeEnv.__init__).parameters) == ["self", "device", "cube_size", "compute_device"]
    assert list(inspect.signature(make_env).parameters) == ["device", "cube_size"]            # env.py:3-5
    with pytest.raises(TypeError):
        CubeEnv(torch.device("cpu"), cube_size=3, _backend=object())
    for dp, _, fs in os.walk(os.path.join(ROOT, "rubiks-cube-solver_amd")):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"\b_backend\b", src) and "test-only" not in src.lower(), f


def test_tile_helpers():
    from rubiks_cube_solver_amd import ops
    for n, pitch in ((10, None), (5000, 1024), (40000, None), (16384, None)):
        a = np.random.default_rng(n).integers(0, 6, (n, 54), dtype=np.uint8)
        t = ops.from_aos(a, "cpu", pitch)
        assert t.dim() == 3 and t.shape[1] == 54 and t.shape[0] * t.shape[2] >= n
        assert (ops.to_aos(t, n).numpy() == a).all()
        if t.shape[0] > 1:
            k = t.shape[2]
            assert (t[1, :, 0].numpy() == a[k]).all()           # cube `pitch` opens tile 1


# ---------------------------------------------------- CubeEnv host semantics (oracle-backed, CPU)
def _env(cs):
    from tests.fake_backend import HostLogicCubeEnv
    return HostLogicCubeEnv(torch.device("cpu"), cube_size=cs)


def test_cube_env_surface_and_types(golden):
    env = _env(3)
    assert env.state_dim == [20, 24] and env.action_dim == 12 and env.cube_size == 3 and env.show_cube is False
    assert env.action_to_sim_action[3] == ["U", "U'", "F", "F'", "R", "R'", "D", "D'", "B", "B'", "L", "L'"]
    assert env.action_to_sim_action[2] == ["U", "U'", "F", "F'", "R", "R'"]
    assert env.sim_cube.dtype == np.int64 and env.sim_cube.tolist() == sum([[c] * 9 for c in range(6)], [])
    s, r, d, info = env.step(0)
    assert s.shape == (20, 24) and s.dtype == np.int64 and isinstance(r, float) and isinstance(d, bool) and info == {}
    assert (r, d) == (-1.0, False) and env.step(1)[1:3] == (1.0, True)          # KAT-C
    assert env.step(-1)[0] is not None                                            # list indexing: -1 = L'
    for bad in (12, 99, -13):
        with pytest.raises(IndexError):
            env.step(bad)
    with pytest.raises(TypeError):
        env.step("U")
    g = golden("walks_333")
    env.init_state()
    for d_ in range(30):
        s, r, dn, _ = env.step(int(g["actions"][3, d_]))
        assert (env.sim_cube == g["stickers"][3, d_]).all() and (np.argmax(s, 1) == g["cols"][3, d_]).all()
        assert (env.cube == s).all()
    e2 = _env(2)
    s2, _, _, _ = e2.step(2)
    assert s2.shape == (7, 21) and s2.dtype == np.float64 and e2.state_dim == [7, 21] and e2.action_dim == 6
    with pytest.raises(IndexError):
        e2.step(6)
    from tests.fake_backend import HostLogicCubeEnv
    with pytest.raises(NotImplementedError):
        HostLogicCubeEnv(torch.device("cpu"), cube_size=4)


def test_cube_env_reset_matches_reference_rng(golden):
    g = golden("reset_333")
    env = _env(3)
    np.random.seed(31337)
    before = np.random.get_state()[1].copy()
    for i, seed in enumerate(g["seeds"]):
        for j in (0, 1, 9, 29):
            s = env.reset(seed=int(seed), scramble_count=int(g["ks"][j]))
            assert (env.sim_cube == g["stickers"][i, j]).all() and (np.argmax(s, 1) == g["cols"][i, j]).all()
    assert (np.random.get_state()[1] == before).all()                             # cube_env.py:62,68
    a = env.reset(scramble_count=7).copy()                                        # unseeded: draws, then restores
    assert (env.reset(scramble_count=7) == a).all()
    with pytest.raises(UnboundLocalError):
        env.reset(seed=1, scramble_count=0)
    s = env.reset(seed=10, scramble_count=30)                                     # KAT-B
    assert "".join(map(str, env.sim_cube)) == "503401005122111541220425001153533522404445432413352330"


def test_cube_env_deepcopy_and_assignment():
    env = _env(3)
    env.reset(seed=5, scramble_count=11)
    snap = env.sim_cube.copy()
    other = copy.deepcopy(env)                                                    # mcts.py:37,96,101
    env.step(3)
    assert (other.sim_cube == snap).all() and not (env.sim_cube == snap).all()
    env.sim_cube = snap                                                           # callers may restore a state
    assert (env.sim_cube == snap).all() and (env.cube == other.cube).all()
    for fn in (env.render, env.close_render, env.save_video):
        with pytest.raises(NotImplementedError):
            fn()
    e2 = _env(2)
    e2.reset(seed=2, scramble_count=9)
    assert (e2.state_to_sim_state(e2.cube) == e2.sim_cube).all()
    with pytest.raises(NotImplementedError):
        env.state_to_sim_state(env.cube)


def test_make_env_needs_gpu():
    from rubiks_cube_solver_amd import _lib
    from rubiks_cube_solver_amd.cube_env import make_env
    if not torch.cuda.is_available():
        with pytest.raises((_lib.RubikHipError, RuntimeError, AssertionError)):
            make_env(torch.device("cpu"), 3)


def test_register_gym_entry_point(monkeypatch):
    import types
    calls = {}
    gym = types.ModuleType("gym")
    envs = types.ModuleType("gym.envs")
    reg = types.ModuleType("gym.envs.registration")
    reg.register = lambda id, entry_point: calls.update(id=id, entry_point=entry_point)
    for name, m in (("gym", gym), ("gym.envs", envs), ("gym.envs.registration", reg)):
        monkeypatch.setitem(sys.modules, name, m)
    from rubiks_cube_solver_amd.cube_env import register_gym
    assert register_gym() == "cube-v0"
    assert calls == {"id": "cube-v0", "entry_point": "rubiks_cube_solver_amd.cube_env:CubeEnv"}
    mod_name, cls_name = calls["entry_point"].split(":")
    import importlib
    assert getattr(importlib.import_module(mod_name), cls_name).__name__ == "CubeEnv"


def test_cube_env_is_a_gym_env_where_gym_is_importable(tmp_path):
    """The reference's CubeEnv subclasses gym.Env (cube_env.py:12) and is built through gym.make (env.py:3-5).  gym is not installed here
    and is not a dependency; where it IS importable the product class derives from gym.Env and register_gym() points the registry at it.
    Executed in a child process against a minimal stand-in `gym` package on sys.path (Env, envs.registration.register, make): NOT a test
    against a real gym release (INTEGRATION.md says so)."""
    import subprocess
    pkg = tmp_path / "gym"
    (pkg / "envs").mkdir(parents=True)
    (pkg / "__init__.py").write_text(
        "class Env:\n    metadata = {}\n"
        "from gym.envs.registration import register, make, registry\n")
    (pkg / "envs" / "__init__.py").write_text("")
    (pkg / "envs" / "registration.py").write_text(
        "import importlib\nregistry = {}\n"
        "def register(id, entry_point):\n    registry[id] = entry_point\n"
        "def make(id, **kw):\n    mod, cls = registry[id].split(':')\n    return getattr(importlib.import_module(mod), cls)(**kw)\n")
    code = (
        "import sys, torch, gym\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from rubiks_cube_solver_amd.cube_env import CubeEnv, register_gym\n"
        "from tests.fake_backend import HostLogicCubeEnv\n"
        "assert issubclass(CubeEnv, gym.Env) and issubclass(HostLogicCubeEnv, gym.Env)\n"
        "assert register_gym() == 'cube-v0' and gym.registry['cube-v0'] == 'rubiks_cube_solver_amd.cube_env:CubeEnv'\n"
        "gym.registry['cube-test-v0'] = 'tests.fake_backend:HostLogicCubeEnv'      # the same class over the oracle backend (no GPU here)\n"
        "env = gym.make('cube-test-v0', cube_size=3, device=torch.device('cpu'))   # env.py:3-5's call shape\n"
        "assert isinstance(env, gym.Env) and isinstance(env, CubeEnv)\n"
        "s = env.reset(seed=10, scramble_count=30)\n"
        "assert ''.join(map(str, env.sim_cube)) == '503401005122111541220425001153533522404445432413352330' and s.shape == (20, 24)   # KAT-B\n"
        "import copy\n"
        "assert type(copy.deepcopy(env)) is type(env)\n"
        "print('gym ok')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT,
                         env=dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", "")))
    assert out.returncode == 0 and "gym ok" in out.stdout, out.stdout + out.stderr[-3000:]


def test_legacy_scramble_actions(golden):
    from rubiks_cube_solver_amd.vec_env import legacy_scramble_actions
    g = golden("reset_333")
    np.random.seed(1)
    before = np.random.get_state()[1].copy()
    a = legacy_scramble_actions(g["seeds"], 30, 12)
    assert (a == g["actions"][:, 29, :]).all() and (np.random.get_state()[1] == before).all()
    assert legacy_scramble_actions([0], 5, 12).tolist() == [[5, 0, 3, 11, 3]]


# -------------------------------------------------------------------------- sharding
def test_shard_partitions():
    from rubiks_cube_solver_amd import dist as d
    for n in (0, 1, 7, 8, 100_000, (1 << 23) + 5):
        for ws in (1, 2, 3, 8):
            parts = [d.shard(n, r, ws) for r in range(ws)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(ws - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    assert len({d.rng_stream(r) for r in range(8)}) == 8


_GLOO_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["RC_ROOT"])
import numpy as np
from rubiks_cube_solver_amd import dist as d
from oracle.oracle_np import Oracle
rank, ws, local = d.init(backend="gloo")
assert ws == int(os.environ["RC_WS"])
lo, hi = d.shard(1001, rank, ws)
orc = Oracle()
out = orc.adi(3, hi - lo, 6, seed=77, stream=d.rng_stream(rank), walk0=0, want_children=False)   # rank-local walks, own stream
d.barrier()
tot, = d.reduce_scalars([float(hi - lo)], op="sum")
mx, = d.reduce_scalars([float(rank + 1) * 0.5], op="max")
json.dump({"rank": rank, "lo": lo, "hi": hi, "total": tot, "max": mx,
           "first_actions": out["actions"][0].tolist(), "chk": int(out["parents"].astype(np.int64).sum())},
          open(os.path.join(os.environ["RC_OUT"], f"r{rank}.json"), "w"))
d.barrier()
'''


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return str(sock.getsockname()[1])


def _run_gloo_ranks(tmp_path, ws):
    import json
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, RC_ROOT=ROOT, RC_OUT=str(tmp_path), RC_WS=str(ws), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ws}", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), str(script)]
    subprocess.run(cmd, check=True, env=env, timeout=480, capture_output=True)
    return [json.load(open(tmp_path / f"r{i}.json")) for i in range(ws)]


def test_two_rank_gloo_sharding(tmp_path):
    """world_size 2 on CPU (gloo): disjoint shards, rank-distinct RNG streams, reporting reductions only.
    This exercises the HOST side of the N>1 path only -- dist.py's shard / rng_stream / reductions, with the oracle
    standing in for the per-rank generator; no kernel of librubikhip.so runs here.  The kernels' side of the same
    contract (stream_id = rank on a shared GPU, one process each) is tests/test_gpu_env.py::
    test_envs_in_several_processes_share_one_gpu and tests/test_bench_contract.py::test_bench_two_ranks_rehearsal."""
    r0, r1 = _run_gloo_ranks(tmp_path, 2)
    assert (r0["lo"], r0["hi"], r1["lo"], r1["hi"]) == (0, 501, 501, 1001)
    assert r0["total"] == r1["total"] == 1001.0 and r0["max"] == r1["max"] == 1.0
    assert r0["first_actions"] != r1["first_actions"]           # independent streams per rank
    from oracle.oracle_np import Oracle
    assert r0["first_actions"] == Oracle().rng_actions(77, 0, 0, 6, 12).tolist()  # reproducible


def test_dist_init_refuses_legacy_ipc_for_nccl(monkeypatch):
    """dist.init for the nccl backend refuses an exported HSA_ENABLE_IPC_MODE_LEGACY other than 0: no process group
    is created, nothing touches a GPU."""
    from rubiks_cube_solver_amd import dist as d
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    with pytest.raises(RuntimeError, match="HSA_ENABLE_IPC_MODE_LEGACY"):
        d.init(backend="nccl")
    assert not d.dist.is_initialized()


def test_eight_rank_streams_and_shards(tmp_path):
    """BASELINE config 4's world size (8 ranks) on CPU over gloo: eight contiguous shards that tile the batch, eight RNG streams
    that are each the oracle's (seed, stream_id = rank) stream and pairwise different, SUM / MAX reductions over all eight.
    (A GPU box of this pool admits at most 6 GPU processes per job, so the 8-rank shape is rehearsed here and the kernels' side
    with 4 ranks in tests/test_bench_contract.py::test_bench_many_ranks_rehearsal.)"""
    from oracle.oracle_np import Oracle
    from rubiks_cube_solver_amd import dist as d
    rs = _run_gloo_ranks(tmp_path, 8)
    assert [(r["lo"], r["hi"]) for r in rs] == [d.shard(1001, k, 8) for k in range(8)] and rs[0]["lo"] == 0 and rs[-1]["hi"] == 1001
    assert all(r["total"] == 1001.0 and r["max"] == 4.0 for r in rs)
    orc = Oracle()
    for k, r in enumerate(rs):
        assert r["first_actions"] == orc.rng_actions(77, d.rng_stream(k), 0, 6, 12).tolist(), k
    assert len({tuple(r["first_actions"]) for r in rs}) == 8


def test_reference_callers_run_unmodified_against_the_product_cube_env():
    """The drop-in claim, executed: the REFERENCE's own mcts.MCTS.train (mcts.py:36-154), train.validation (train.py:167-198),
    test.trial (test.py:103-158, plain / masked / MCTS) and utils.ReplayBuffer + DataLoader (utils.py:203-270,296-303) are imported
    unmodified (make_golden.py's three harness shims) and driven with the product's CubeEnv class -- here its host logic over the
    oracle backend, tests/fake_backend.HostLogicCubeEnv -- and every observable (simulations used, action lists, node statistics,
    every env.step validation issues, solve steps, the replay deque, prioritised indices, __getitem__ dtypes, DataLoader order,
    get_target_value) equals what the reference computed with its OWN env (G5, G8, G9, G11, G12; 2x2x2: G13).  33 checks; runs in a child
    process because the shims rebind `gym`, numpy.int and put the reference's `utils` / `test` / `model` modules on sys.path.
    The reference never travels: skipped where /root/reference is absent (the GPU box)."""
    import json
    import os
    import subprocess
    import sys
    if not os.path.isdir("/root/reference/gym-cube"):
        pytest.skip("the reference is only present in the build container")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-B", os.path.join(root, "tests", "golden", "run_reference_callers.py")],
                         capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["failed"] == [] and len(d["checks"]) == 33 and all(d["checks"].values())
    assert d["env_class"] == "rubiks_cube_solver_amd.cube_env" and d["reference_modules"] == ["mcts", "model", "test", "train", "utils"]


def test_cube_env_subclass_survives_deepcopy_and_plan_key():
    """mcts.py:37,96,101 deep-copies the env 14 times per simulation: a subclass must stay itself (round 6: the copy was a plain
    CubeEnv, found by running the reference's MCTS against HostLogicCubeEnv).  get_random_samples keeps one plan per call shape AND
    per (dtype, device) of the model (an in-place model.half() / .to() must not meet the stale plan)."""
    import copy
    env = _env(3)
    assert type(copy.deepcopy(env)) is type(env)
    made = []
    orig = type(env)._new_adi_plan

    def counting(self, model, *a):
        made.append(str(next(model.parameters()).dtype))
        return orig(self, model, *a)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(480, 1)

        def forward(self, x):
            if x.dim() == 2:
                x = x.unsqueeze(0)
            return self.lin(x.reshape(x.shape[0], -1).to(self.lin.weight.dtype)).float(), torch.zeros(x.shape[0], 12)

    net = Net()
    type(env)._new_adi_plan = counting
    try:
        sink = []
        env.get_random_samples(sink, net, 3, 2, 1.0)
        env.get_random_samples(sink, net, 3, 2, 1.0)                     # same shape, same model: the kept plan
        assert made == ["torch.float32"] and len(sink) == 12
        net.bfloat16()                                                   # in place: same object, other dtype
        env.get_random_samples(sink, net, 3, 2, 1.0)
        assert made == ["torch.float32", "torch.bfloat16"] and len(sink) == 18 and len(env._adi_plans) == 1
    finally:
        type(env)._new_adi_plan = orig


def test_a_stale_library_is_refused_and_rebuilt_by_id_not_by_mtime(tmp_path):
    """rc_build_id (round 6): the binding loads only a library whose embedded source hash equals the hash of the tree's sources, and
    build() decides by that id, never by modification times.  A copy of the shipped library with ONE hex digit of its id changed (= a
    binary built from other sources) and the newest mtime of all: refused by _lib.lib() in a fresh process, accepted only with
    RC_ALLOW_STALE=1 (A/B experiments), and seen as stale by the build's own check."""
    import shutil
    import subprocess
    from rubiks_cube_solver_amd import _build, _lib
    want = _build.source_hash(_build.HIP_SOURCES)
    assert want and _build.embedded_id(_lib.LIB_PATH) == want
    fake = str(tmp_path / "librubikhip.so")
    data = bytearray(open(_lib.LIB_PATH, "rb").read())
    i = data.find(_build.MARKER) + len(_build.MARKER)
    data[i] = ord("0") if data[i] != ord("0") else ord("1")
    open(fake, "wb").write(bytes(data))
    os.utime(fake, None)                                               # newer than every source: an mtime rule would call it current
    assert _build.embedded_id(fake) != want and os.path.getmtime(fake) >= max(os.path.getmtime(p) for p in _build.HIP_SOURCES)
    code = "from rubiks_cube_solver_amd import _lib; L = _lib.lib(); print('loaded', _lib.build_id())"
    env = dict(os.environ, RUBIKHIP_LIB=fake)
    env.pop("RC_ALLOW_STALE", None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "is stale" in out.stderr and "loaded" not in out.stdout, out.stderr[-2000:]
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(env, RC_ALLOW_STALE="1"))
    assert out.returncode == 0 and "loaded " + _build.embedded_id(fake) in out.stdout, out.stderr[-2000:]
    # the tree library goes through the same check
    from rubiks_cube_solver_amd import _tree
    tfake = str(tmp_path / "librubiktree.so")
    tdata = bytearray(open(_tree.LIB_PATH, "rb").read())
    j = tdata.find(_build.MARKER) + len(_build.MARKER)
    tdata[j] = ord("0") if tdata[j] != ord("0") else ord("1")
    open(tfake, "wb").write(bytes(tdata))
    out = subprocess.run([sys.executable, "-c", "from rubiks_cube_solver_amd import _tree; _tree.tree_lib()"], capture_output=True, text=True, timeout=300,
                         cwd=ROOT, env=dict(os.environ, RUBIKTREE_LIB=tfake))
    assert out.returncode != 0 and "is stale" in out.stderr, out.stderr[-2000:]
    # a library without any id (built by hand without -DRC_SRC_HASH, or one that predates round 6) is stale too
    plain = str(tmp_path / "no_id.so")
    open(plain, "wb").write(b"\x7fELF" + b"\0" * 64)
    assert _build.embedded_id(plain) is None and _build.embedded_id(str(tmp_path / "missing.so")) is None


def test_cube_env_222_host_logic_against_the_references_own_222_branches(golden):
    """Fixture G13 = the reference's OWN CubeEnv run with cube_size = 2 (stand-in py222).  The product's CubeEnv host logic (oracle backend)
    against it: reset(seed, k) incl. the untouched global generator, step's 4-tuple and float64 [7, 21] one-hot (row = cubelet, column =
    position * 3 + orientation: cube_env.py:143-147), state_to_sim_state (cube_env.py:154-175), get_random_samples records and the state
    the env is left in (cube_env.py:177-194), get_target_value (cube_env.py:196-252) -- float-EQUAL: the host path calls the model as the
    reference does."""
    g = golden("env222_via_reference")
    env = _env(2)
    assert (env.sim_cube == g["solved_stickers"]).all() and (np.argmax(env.cube, 1) == g["solved_cols"]).all() and str(env.cube.dtype) == str(g["state_dtype"])
    np.random.seed(99)
    before = np.random.get_state()[1].copy()
    for i, sd in enumerate(g["reset_seeds"]):
        for j, k in enumerate(g["reset_ks"]):
            s = env.reset(seed=int(sd), scramble_count=int(k))
            assert s.dtype == np.float64 and (np.argmax(s, 1) == g["reset_cols"][i, j]).all() and (s.sum(1) == 1).all()
            assert (env.sim_cube == g["reset_stickers"][i, j]).all()
    assert (np.random.get_state()[1] == before).all()
    for w in range(0, 300, 7):
        env.init_state()
        for d in range(14):
            s, r, dn, info = env.step(int(g["walk_actions"][w, d]))
            assert (env.sim_cube == g["walk_stickers"][w, d]).all() and (np.argmax(s, 1) == g["walk_cols"][w, d]).all()
            assert isinstance(r, float) and isinstance(dn, bool) and (r, dn) == (g["walk_reward"][w, d], bool(g["walk_done"][w, d])) and info == {}
            if d % 5 == 0:
                assert (env.state_to_sim_state(env.cube) == g["roundtrip_stickers"][w, d // 5]).all()
    w_, b_ = torch.tensor(g["adi_w"]), torch.tensor(g["adi_b"])

    class StubModel(torch.nn.Module):
        def forward(self, x):
            if x.dim() == 2:
                x = x.unsqueeze(0)
            return (x.reshape(x.shape[0], -1) @ w_ + b_).unsqueeze(-1), torch.zeros(x.shape[0], 6)

    model, T = StubModel(), float(g["adi_temperature"])
    n, depth = g["adi_actions"].shape
    buf = []
    np.random.seed(int(g["adi_seed"]))
    env.get_random_samples(buf, model, depth, n, T)
    assert len(buf) == n * depth and (env.sim_cube == g["adi_final_stickers"]).all()
    assert buf[0]["state"].dtype == np.float64 and buf[0]["state"].shape == (7, 21) and type(buf[0]["target_policy"]) is int
    assert (np.stack([np.argmax(x["state"], 1) for x in buf]).reshape(n, depth, 7) == g["adi_cols"]).all()
    assert (np.array([x["target_value"] for x in buf]).reshape(n, depth) == g["adi_target_value"]).all()
    assert (np.array([x["target_policy"] for x in buf]).reshape(n, depth) == g["adi_target_policy"]).all()
    assert (np.array([x["scramble_count"] for x in buf]).reshape(n, depth) == g["adi_scramble_count"]).all()
    assert (np.array([x["error"] for x in buf]).reshape(n, depth) == g["adi_error"]).all()
    env.init_state()
    for d in range(depth):
        env.step(int(g["adi_actions"][3, d]))
        tv, tp, er = env.get_target_value(model, d + 1, T)
        assert (tv, tp, er) == (g["adi_target_value"][3, d], g["adi_target_policy"][3, d], g["adi_error"][3, d])
