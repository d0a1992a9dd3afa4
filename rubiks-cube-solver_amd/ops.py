"""Batched cube operators on PyTorch-ROCm tensors: the device analogue of the reference's
operator layer (gym-cube/gym_cube/envs/assets/py333.py:211-246: initState_3, doMove_3,
isSolved_3, getOP_3 + pos_to_state_3) plus the env loops built on it.  Each function
validates its tensors and forwards raw pointers + the current HIP stream to librubikhip.so.

State layout (include/rubikhip.h "State layout"): uint8 tensor [tiles, S, pitch] -- structure of
arrays in tiles: one row per sticker, one column per cube, `pitch` cubes per tile; cube n sits in
tile n // pitch, column n % pitch.  A 2-D [S, pitch] tensor is the one-tile case.  Buffers with
several tiles need a power-of-two pitch >= MIN_TILE = 512 (DEFAULT_TILE = 32768 measured best on MI355X).
Compact code buffers [tiles, SLOTS, pitch] follow the same rule.  Expansion / ADI outputs are
one-tile ("plain") buffers.
"""
from __future__ import annotations

import threading

import torch

from . import _lib
from ._lib import FMT_BF16, FMT_CODE, FMT_F16, FMT_F32, FMT_NONE, FMT_U8, RubikHipError, check, lib, ptr, stream_ptr
from .tables import ACTION_DIM, STATE_DIM

N_STICKERS = {2: 24, 3: 54}
N_SLOTS = {2: 7, 3: 20}


def _size(cube_size):
    if cube_size not in (2, 3):
        raise NotImplementedError(f"cube_size {cube_size}")  # cube_env.py:44,106,151
    return N_STICKERS[cube_size], ACTION_DIM[cube_size], N_SLOTS[cube_size]


DEFAULT_TILE = 32768
MIN_TILE = 512        # the widest wave span (8 cubes per lane): a wave never straddles tiles


def _rows(t, rows, n, what):
    """Plain (one-tile) buffer: contiguous uint8 device tensor [..., rows, pitch], pitch >= n, pitch % 16 == 0."""
    if t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous():
        raise RubikHipError(f"{what}: need a contiguous uint8 HIP tensor")
    if t.dim() < 2 or t.shape[-2] != rows or t.shape[-1] < n or t.shape[-1] % 16:
        raise RubikHipError(f"{what}: shape {tuple(t.shape)} is not [..., {rows}, pitch>=n, pitch%16==0]")
    return t.shape[-1]


def _tiled(t, rows, n, what):
    """State / code buffer: [tiles, rows, pitch] (or [rows, pitch] = one tile).  Returns the pitch."""
    if t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous():
        raise RubikHipError(f"{what}: need a contiguous uint8 HIP tensor")
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.dim() != 3 or t.shape[1] != rows:
        raise RubikHipError(f"{what}: shape {tuple(t.shape)} is not [tiles, {rows}, pitch]")
    tiles, _, pitch = t.shape
    if pitch % 16 or tiles * pitch < n:
        raise RubikHipError(f"{what}: pitch {pitch} x {tiles} tiles cannot hold {n} cubes (pitch % 16 must be 0)")
    if n > pitch and (pitch < MIN_TILE or pitch & (pitch - 1)):
        raise RubikHipError(f"{what}: a buffer with several tiles needs a power-of-two pitch >= {MIN_TILE}, got {pitch}")
    return pitch


def _tile_shape(n, pitch):
    if pitch is None:
        pitch = _lib.pitch_for(n) if n <= DEFAULT_TILE else DEFAULT_TILE
    return max(1, -(-n // pitch)), pitch


def to_aos(t, n):
    """[tiles, rows, pitch] (or [rows, pitch]) device buffer -> [n, rows] tensor (one cube per row)."""
    if t.dim() == 2:
        t = t.unsqueeze(0)
    return t.permute(0, 2, 1).reshape(-1, t.shape[1])[:n]


def from_aos(a, device, pitch=None):
    """[n, rows] uint8 (numpy or tensor, host) -> tiled device buffer [tiles, rows, pitch]."""
    a = torch.as_tensor(a, dtype=torch.uint8)
    n, rows = a.shape
    tiles, pitch = _tile_shape(n, pitch)
    buf = torch.zeros((tiles * pitch, rows), dtype=torch.uint8)
    buf[:n] = a
    return buf.reshape(tiles, pitch, rows).permute(0, 2, 1).contiguous().to(device)


def alloc_code(n, cube_size, device, pitch=None):
    _, _, SL = _size(cube_size)
    tiles, pitch = _tile_shape(n, pitch)
    return torch.empty((tiles, SL, pitch), dtype=torch.uint8, device=device)


def _vec(t, n, dtype, what):
    if t is None:
        return None
    if t.dtype != dtype or not t.is_cuda or not t.is_contiguous() or t.numel() < n:
        raise RubikHipError(f"{what}: need a contiguous {dtype} HIP tensor with >= {n} elements")
    return t


def alloc_states(n, cube_size, device, pitch=None):
    S, _, _ = _size(cube_size)
    tiles, pitch = _tile_shape(n, pitch)
    return torch.empty((tiles, S, pitch), dtype=torch.uint8, device=device)


def fill_solved(st, n, cube_size):
    S, _, _ = _size(cube_size)
    pitch = _tiled(st, S, n, "fill_solved")
    _lib.init(st.device)
    check(lib().rc_fill_solved(ptr(st), n, pitch, cube_size, stream_ptr(st.device)))
    return st


def _onehot_args(onehot, fmt, n, cube_size, what):
    if fmt == FMT_NONE:
        if onehot is not None:
            raise RubikHipError(f"{what}: onehot given but fmt is FMT_NONE")
        return None, 0
    if onehot is None:
        raise RubikHipError(f"{what}: fmt needs an output tensor")
    if fmt == FMT_CODE:
        return onehot, _tiled(onehot, N_SLOTS[cube_size], n, what + " code")
    R, C = STATE_DIM[cube_size]
    if onehot.dtype != _lib.dense_dtype(fmt) or not onehot.is_cuda or not onehot.is_contiguous() or onehot.numel() < n * R * C:
        raise RubikHipError(f"{what}: dense one-hot must be a contiguous {_lib.dense_dtype(fmt)} HIP tensor [n,{R},{C}]")
    return onehot, 0


_workspaces = {}     # (device index, stream handle) -> uint8 tensor; grown on demand, never shared between streams
_ws_lock = threading.Lock()
_WS_MAX = 8          # cached workspaces (streams come and go: the oldest entry is dropped beyond this)


def workspace(device, nbytes):
    """Scratch tensor for rc_apply_moves_ws on `device`'s CURRENT stream (the library allocates nothing: the caller owns it).
    One per (device, stream): launches of one stream are ordered, so reusing it between calls is safe -- provided the two launches of
    one call are queued back to back, which is why apply_moves / encode hold `_ws_lock` from this look-up until the C call has
    returned (two host threads driving ONE stream would otherwise interleave step(T1), step(T2), front(T1)).  At most _WS_MAX
    workspaces are kept (20 bytes per cube each); release_workspaces() drops them all."""
    device = torch.device(device)
    if torch.cuda.is_current_stream_capturing():
        # under hipGraph capture the allocation belongs to the graph's private pool and lives exactly as long as the graph: never cached
        return torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream_ptr(device).value)
    ws = _workspaces.pop(key, None)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    _workspaces[key] = ws                                         # most recently used last
    while len(_workspaces) > _WS_MAX:
        _workspaces.pop(next(iter(_workspaces)))                   # the caching allocator keeps the block alive for launches still queued
    return ws


def release_workspaces():
    """Drop every cached rc_apply_moves_ws workspace (they are re-created on demand)."""
    with _ws_lock:
        _workspaces.clear()


def apply_moves(src, dst, actions, n, cube_size, reward=None, done=None, onehot=None, fmt=FMT_NONE, variant=0):
    """CubeEnv.step for n cubes (cube_env.py:71-111).  dst may be src (in place).
    A dense `onehot` on a large 3x3x3 batch goes through rc_apply_moves_ws with a cached workspace (two launches: step + compact
    code, then the front writer: same results, the dense stream at 0.8-0.94 of the HBM peak wherever the buffer lives).
    variant: per-call tuning override (include/rubikhip.h "Tuning override"; tests and benchmarks only)."""
    S, _, _ = _size(cube_size)
    p_in, p_out = _tiled(src, S, n, "apply_moves src"), _tiled(dst, S, n, "apply_moves dst")
    _vec(actions, n, torch.uint8, "actions")
    _vec(reward, n, torch.float32, "reward")
    _vec(done, n, torch.uint8, "done")
    oh, cp = _onehot_args(onehot, fmt, n, cube_size, "apply_moves")
    _lib.init(src.device)
    if variant:
        check(lib().rc_apply_moves_ex(ptr(src), ptr(dst), ptr(actions), n, p_in, p_out, cube_size, ptr(reward), ptr(done),
                                      ptr(oh), fmt, cp, stream_ptr(src.device), variant))
        return
    need = lib().rc_workspace_bytes(_lib.OP_STEP, cube_size, n, fmt) if fmt >= _lib.FMT_U8 else 0
    if need > 0:
        with _ws_lock:
            ws = workspace(src.device, need)
            check(lib().rc_apply_moves_ws(ptr(src), ptr(dst), ptr(actions), n, p_in, p_out, cube_size, ptr(reward), ptr(done),
                                          ptr(oh), fmt, cp, ptr(ws), ws.numel(), stream_ptr(src.device)))
        return
    check(lib().rc_apply_moves(ptr(src), ptr(dst), ptr(actions), n, p_in, p_out, cube_size, ptr(reward), ptr(done),
                               ptr(oh), fmt, cp, stream_ptr(src.device)))


def scramble(st, n, cube_size, depth, seed=0, stream_id=0, walk_offset=0, actions_in=None, actions_out=None,
             done=None, reward=None, src=None):
    """reset()'s scramble loop in place (cube_env.py:65-67); actions_* are [depth, pitch] uint8.
    actions_in may also be a PINNED host tensor: the kernel then reads the moves straight from host memory (no upload; the lockstep
    search hands its tree descents over this way).
    src: read the start states from this buffer (same shape as st) instead of st: st = src moved, src untouched (rc_scramble_from;
    with depth 0 a plain copy)."""
    S, _, _ = _size(cube_size)
    pitch = _tiled(st, S, n, "scramble")
    if src is not None and (_tiled(src, S, n, "scramble src") != pitch or src.shape != st.shape):
        raise RubikHipError("scramble: src and st must share one shape")
    ap = 0
    a_in = ptr(actions_in)
    if actions_in is not None and not actions_in.is_cuda:
        if actions_in.dtype != torch.uint8 or actions_in.dim() != 2 or actions_in.shape[0] != depth or not actions_in.is_contiguous() or \
                actions_in.shape[1] < n or actions_in.shape[1] % 16:
            raise RubikHipError(f"scramble actions (host): need a contiguous pinned uint8 tensor [{depth}, pitch >= n, pitch % 16 == 0]")
        a_in, ap = _lib.host_alias(actions_in), actions_in.shape[1]
    elif actions_in is not None:
        ap = _rows(actions_in, depth, n, "scramble actions")
    if actions_out is not None:
        ap2 = _rows(actions_out, depth, n, "scramble actions")
        if actions_in is not None and ap2 != ap:
            raise RubikHipError("scramble: actions_in and actions_out must share one pitch")
        ap = ap2
    _vec(reward, n, torch.float32, "reward")
    _vec(done, n, torch.uint8, "done")
    _lib.init(st.device)
    if src is not None:
        check(lib().rc_scramble_from(ptr(src), ptr(st), n, pitch, cube_size, depth, seed, stream_id, walk_offset, a_in,
                                     ptr(actions_out), ap, ptr(done), ptr(reward), stream_ptr(st.device)))
        return
    check(lib().rc_scramble(ptr(st), n, pitch, cube_size, depth, seed, stream_id, walk_offset, a_in,
                            ptr(actions_out), ap, ptr(done), ptr(reward), stream_ptr(st.device)))


def search_pack(leaf_code, child_code, child_solved, n, cube_size, leaf_out, child_out, solved_out):
    """The results of one expansion launch, laid out per root for the host trees of a lockstep search (rc_search_pack):
    leaf_code [tiles, SLOTS, pitch], child_code [A, tiles, SLOTS, pitch], child_solved [A, tiles * pitch]  ->
    leaf_out [n, SLOTS], child_out [n, A, SLOTS], solved_out [n, A] (uint8, contiguous)."""
    _, A, SL = _size(cube_size)
    pitch = _tiled(leaf_code, SL, n, "search_pack leaf_code")
    tiles = _tiles_of(n, pitch)
    _out(child_code, (A,), tiles, SL, pitch, "search_pack child_code")
    _out(child_solved, (A,), tiles, 0, pitch, "search_pack child_solved")
    for t, shape, what in ((leaf_out, (n, SL), "leaf_out"), (child_out, (n, A, SL), "child_out"), (solved_out, (n, A), "solved_out")):
        if t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous() or tuple(t.shape) != shape:
            raise RubikHipError(f"search_pack: {what} must be a contiguous uint8 HIP tensor {shape}")
    _lib.init(leaf_code.device)
    check(lib().rc_search_pack(ptr(leaf_code), ptr(child_code), ptr(child_solved), n, pitch, cube_size, ptr(leaf_out), ptr(child_out), ptr(solved_out),
                               stream_ptr(leaf_code.device)))


def legacy_scramble_actions(seeds, cube_size, scramble_count, device=None, variant=0):
    """The reference's reset(seed, k) draws for every env, computed on the device (numpy legacy MT19937 +
    masked rejection, cube_env.py:62-68).  seeds: int tensor / sequence [n] (0 <= seed < 2**32);
    scramble_count: one int or one int per env.  Returns (actions uint8 [kmax, pitch] with no-op padding, kmax).
    variant: which generator form runs (include/rubikhip.h RC_VARIANT_LEGACY_*; tests and benchmarks only)."""
    _, A, _ = _size(cube_size)
    dev = torch.device(device) if device is not None else (seeds.device if isinstance(seeds, torch.Tensor) else torch.device("cuda"))
    s = torch.as_tensor(seeds)
    if s.numel() and (int(s.min()) < 0 or int(s.max()) > 0xFFFFFFFF):
        raise ValueError("Seed must be between 0 and 2**32 - 1")          # numpy's legacy seeding error
    n = s.numel()
    s32 = (s.to(torch.int64) & 0xFFFFFFFF).to(torch.int64)
    s32 = torch.where(s32 >= 2 ** 31, s32 - 2 ** 32, s32).to(torch.int32).to(dev).contiguous()   # raw 32-bit pattern
    counts = None
    if isinstance(scramble_count, int):
        kmax, uniform = scramble_count, scramble_count
    else:
        c = torch.as_tensor(scramble_count, dtype=torch.int32)
        if c.numel() != n:
            raise ValueError("need one scramble count per env")
        kmax, uniform, counts = int(c.max()) if n else 0, 0, c.to(dev).contiguous()
    pitch = _lib.pitch_for(n)
    out = torch.full((max(kmax, 1), pitch), A, dtype=torch.uint8, device=dev)      # pad columns / rows hold the no-op
    _lib.init(dev)
    check(lib().rc_legacy_scramble_actions_ex(ptr(s32), ptr(counts), uniform, kmax, n, cube_size, ptr(out), pitch, stream_ptr(dev), variant))
    return out, kmax


def is_solved(st, n, cube_size, done=None, reward=None):
    S, _, _ = _size(cube_size)
    pitch = _tiled(st, S, n, "is_solved")
    _vec(reward, n, torch.float32, "reward")
    _vec(done, n, torch.uint8, "done")
    _lib.init(st.device)
    check(lib().rc_is_solved(ptr(st), n, pitch, cube_size, ptr(done), ptr(reward), stream_ptr(st.device)))


def encode(st, n, cube_size, onehot, fmt):
    S, _, _ = _size(cube_size)
    pitch = _tiled(st, S, n, "encode")
    oh, cp = _onehot_args(onehot, fmt, n, cube_size, "encode")
    _lib.init(st.device)
    need = lib().rc_workspace_bytes(_lib.OP_STEP, cube_size, n, fmt) if fmt >= _lib.FMT_U8 else 0
    if need > 0:                                                 # large dense batches: codes into the workspace, then the front writer
        with _ws_lock:
            ws = workspace(st.device, need)
            check(lib().rc_encode_ws(ptr(st), n, pitch, cube_size, ptr(oh), fmt, cp, ptr(ws), ws.numel(), stream_ptr(st.device)))
        return
    check(lib().rc_encode(ptr(st), n, pitch, cube_size, ptr(oh), fmt, cp, stream_ptr(st.device)))


def onehot_from_code(code, n, cube_size, onehot, variant=0):
    _size(cube_size)
    cp = _tiled(code, N_SLOTS[cube_size], n, "onehot_from_code")
    fmt = _lib.fmt_of(onehot.dtype)
    _onehot_args(onehot, fmt, n, cube_size, "onehot_from_code")
    _lib.init(code.device)
    check(lib().rc_onehot_from_code_ex(ptr(code), n, cp, cube_size, ptr(onehot), fmt, stream_ptr(code.device), variant))


def onehot_from_code_blocks(code, n, cube_size, onehot, block_stride):
    """Several equally tiled code buffers -> packed dense blocks in ONE launch (rc_onehot_from_code_blocks).  code: contiguous uint8
    [n_blocks, tiles, SLOTS, pitch] (n cubes each); block b fills onehot[b * block_stride : b * block_stride + n] (block_stride >= n and
    every block 16-byte aligned: a multiple of 16 cubes always is).  The 2x2x2 ADI pipeline's one-hot launch."""
    _size(cube_size)
    if code.dim() != 4 or not code.is_contiguous():
        raise RubikHipError("onehot_from_code_blocks: need a contiguous [n_blocks, tiles, SLOTS, pitch] code buffer")
    cp = _tiled(code[0], N_SLOTS[cube_size], n, "onehot_from_code_blocks")
    nb = code.shape[0]
    if code.shape[1] != _tiles_of(n, cp):
        raise RubikHipError(f"onehot_from_code_blocks: {code.shape[1]} tiles per block, {n} cubes need {_tiles_of(n, cp)}")
    fmt = _lib.fmt_of(onehot.dtype)
    R, C = STATE_DIM[cube_size]
    rows = (nb - 1) * block_stride + n
    if not onehot.is_cuda or not onehot.is_contiguous() or onehot.dim() != 3 or tuple(onehot.shape[1:]) != (R, C) or onehot.shape[0] < rows or block_stride < n:
        raise RubikHipError(f"onehot_from_code_blocks: need a contiguous HIP tensor [>= (n_blocks - 1) * block_stride + n, {R}, {C}] and block_stride >= n")
    _lib.init(code.device)
    check(lib().rc_onehot_from_code_blocks(ptr(code), n, cp, cube_size, ptr(onehot), fmt, nb, code.stride(0), block_stride, stream_ptr(code.device)))


def onehot_from_family(family, n, cube_size, onehot, block_stride, n_depths=1):
    """FAMILY rows of `n_depths` consecutive depths ([n_depths, tiles, NF, pitch]; one depth may drop the leading axis; n walks) -> the
    dense one-hots of all A children and the parent of every depth in ONE launch: depth g, child a fills
    onehot[(g * (A + 1) + a) * block_stride : ... + n], the parent is block a = A (3x3x3)."""
    _, A, _ = _size(cube_size)
    nf = _lib.family_layout(cube_size)[0]
    one = family if n_depths == 1 and family.dim() <= 3 else family[0]
    pitch = _tiled(one, nf, n, "onehot_from_family")
    if n_depths != 1 or family.dim() > 3:
        if family.dim() != 4 or family.shape[0] != n_depths or not family.is_contiguous():
            raise RubikHipError(f"onehot_from_family: {n_depths} depths need a contiguous [n_depths, tiles, {nf}, pitch] buffer")
        if n_depths > 1 and family.shape[1] != _tiles_of(n, pitch):
            # the library derives the per-depth source stride from n (ceil(n / pitch) tiles): a record with more tiles than that
            # (e.g. the first n < W walks of a W-walk record) would be read at the wrong offsets from depth 1 on
            raise RubikHipError(f"onehot_from_family: {family.shape[1]} tiles per depth, but {n} walks at pitch {pitch} are {_tiles_of(n, pitch)}: "
                                "slice the record to that many tiles (contiguous) first")
    fmt = _lib.fmt_of(onehot.dtype)
    R, C = STATE_DIM[cube_size]
    rows = (n_depths * (A + 1) - 1) * block_stride + n
    if not onehot.is_cuda or not onehot.is_contiguous() or onehot.dim() != 3 or tuple(onehot.shape[1:]) != (R, C) or onehot.shape[0] < rows:
        raise RubikHipError(f"onehot_from_family: need a contiguous HIP tensor [>= (n_depths * {A + 1} - 1) * block_stride + n, {R}, {C}]")
    _lib.init(family.device)
    check(lib().rc_onehot_from_family_depths(ptr(family), n, pitch, cube_size, ptr(onehot), fmt, block_stride, n_depths, stream_ptr(family.device)))


def _tiles_of(n, pitch):
    if pitch % 16:
        raise RubikHipError(f"pitch {pitch} must be a multiple of 16")
    if n <= pitch:
        return 1
    if pitch < MIN_TILE or pitch & (pitch - 1):
        raise RubikHipError(f"{n} cubes in tiles of {pitch}: several tiles need a power-of-two pitch >= {MIN_TILE}")
    return -(-n // pitch)


def _out(t, lead, tiles, rows, pitch, what):
    """Output buffer [*lead, tiles, rows, pitch] (the tile axis may be dropped when tiles == 1; rows == 0 means
    a flag / action array [*lead, tiles * pitch])."""
    if t is None:
        return
    if t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous():
        raise RubikHipError(f"{what}: need a contiguous uint8 HIP tensor")
    want = (*lead, tiles * pitch) if rows == 0 else (*lead, tiles, rows, pitch)
    ok = tuple(t.shape) == want or (rows and tiles == 1 and tuple(t.shape) == (*lead, rows, pitch))
    if not ok:
        raise RubikHipError(f"{what}: shape {tuple(t.shape)}, expected {want}")


def expand_buffers(n, cube_size, device, pitch=None, children=False, codes=True):
    """Allocate rc_expand_children outputs: child_solved [A, Wp], child_code [A, tiles, SLOTS, pitch],
    children [A, tiles, S, pitch]."""
    S, A, SL = _size(cube_size)
    tiles, pitch = _tile_shape(n, pitch)
    e = lambda *s: torch.empty(s, dtype=torch.uint8, device=device)
    out = {"child_solved": e(A, tiles * pitch)}
    if codes:
        out["child_code"] = e(A, tiles, SL, pitch)
    if children:
        out["children"] = e(A, tiles, S, pitch)
    return out


def expand_children(st, n, cube_size, children=None, child_solved=None, child_code=None, pitch=None, variant=0):
    """All A children of every cube (cube_env.py:212-236, mcts.py:96-101).  Outputs share one tiling:
    children [A, tiles, S, pitch], child_code [A, tiles, SLOTS, pitch], child_solved [A, tiles * pitch]
    (tile axis optional when tiles == 1).  `pitch` defaults to the last dim of the first output given."""
    S, A, SL = _size(cube_size)
    p_in = _tiled(st, S, n, "expand src")
    if pitch is None:
        ref = children if children is not None else child_code
        if ref is not None:
            pitch = ref.shape[-1]
        elif child_solved is not None:
            pitch = child_solved.shape[-1] if n <= child_solved.shape[-1] and child_solved.shape[-1] % 16 == 0 else None
        if pitch is None:
            raise RubikHipError("expand_children: cannot infer the output pitch, pass pitch=")
        if child_solved is not None and children is None and child_code is None and n > pitch:
            raise RubikHipError("expand_children: pass pitch= for a tiled flag buffer")
    tiles = _tiles_of(n, pitch)
    _out(children, (A,), tiles, S, pitch, "children")
    _out(child_code, (A,), tiles, SL, pitch, "child_code")
    _out(child_solved, (A,), tiles, 0, pitch, "child_solved")
    if children is None and child_solved is None and child_code is None:
        raise RubikHipError("expand_children: nothing to write")
    _lib.init(st.device)
    check(lib().rc_expand_children_ex(ptr(st), n, p_in, cube_size, ptr(children), ptr(child_solved), ptr(child_code),
                                      pitch, stream_ptr(st.device), variant))


ADI_TILE = 16384   # walks per output tile: measured best for the ADI kernel's output stream (profiles/r02_design_ab.json)


def adi_buffers(n_walks, depth, cube_size, device, pitch=None, actions=True, parents=False, parent_code=False,
                children=False, child_code=False, child_solved=True, family=False):
    """Allocate rc_adi_generate outputs with one tiling (see include/rubikhip.h).  Returns (pitch, dict).
    family: the [depth, tiles, NF, pitch] FAMILY record of rc_adi_generate_family (instead of parent_code / child_code)."""
    S, A, SL = _size(cube_size)
    if pitch is None:
        pitch = _lib.pitch_for(n_walks) if n_walks <= ADI_TILE else ADI_TILE
    tiles = _tiles_of(n_walks, pitch)
    e = lambda *s: torch.empty(s, dtype=torch.uint8, device=device)
    out = {}
    if actions:
        out["actions_out"] = e(depth, tiles * pitch)
    if parents:
        out["parents"] = e(depth, tiles, S, pitch)
    if parent_code:
        out["parent_code"] = e(depth, tiles, SL, pitch)
    if children:
        out["children"] = e(depth, A, tiles, S, pitch)
    if child_code:
        out["child_code"] = e(depth, A, tiles, SL, pitch)
    if child_solved:
        out["child_solved"] = e(depth, A, tiles * pitch)
    if family:
        out["family"] = e(depth, tiles, _lib.family_layout(cube_size)[0], pitch)
    return pitch, out


def adi_generate(n_walks, depth, cube_size, pitch, device, seed=0, stream_id=0, walk_offset=0, actions_in=None,
                 actions_out=None, parents=None, parent_code=None, children=None, child_code=None, child_solved=None, family=None, variant=0):
    """ADI walks + expansion (cube_env.py:177-194,212-236); layouts in include/rubikhip.h / adi_buffers.
    family: emit the FAMILY record (rc_adi_generate_family) -- then no parent_code / child_code / children."""
    S, A, SL = _size(cube_size)
    tiles = _tiles_of(n_walks, pitch)
    _out(actions_in, (depth,), tiles, 0, pitch, "actions_in")
    _out(actions_out, (depth,), tiles, 0, pitch, "actions_out")
    _out(parents, (depth,), tiles, S, pitch, "parents")
    _out(parent_code, (depth,), tiles, SL, pitch, "parent_code")
    _out(children, (depth, A), tiles, S, pitch, "children")
    _out(child_code, (depth, A), tiles, SL, pitch, "child_code")
    _out(child_solved, (depth, A), tiles, 0, pitch, "child_solved")
    dev = torch.device(device)
    _lib.init(dev)
    if family is not None:
        if parent_code is not None or child_code is not None or children is not None:
            raise RubikHipError("adi_generate: the family record replaces parent_code / child_code / children")
        _out(family, (depth,), tiles, _lib.family_layout(cube_size)[0], pitch, "family")
        check(lib().rc_adi_generate_family(seed, stream_id, walk_offset, n_walks, depth, cube_size, pitch, ptr(actions_in), ptr(actions_out),
                                           ptr(parents), ptr(family), ptr(child_solved), stream_ptr(dev), variant))
        return
    check(lib().rc_adi_generate_ex(seed, stream_id, walk_offset, n_walks, depth, cube_size, pitch, ptr(actions_in),
                                   ptr(actions_out), ptr(parents), ptr(parent_code), ptr(children), ptr(child_code),
                                   ptr(child_solved), stream_ptr(dev), variant))


def adi_targets(child_value, child_solved, n, cube_size, parent_value=None, weight=None):
    """Target value / policy / error (cube_env.py:229-232,239-251).
    child_value [A,pitch] f32, child_solved [A,pitch] u8, parent_value [n] f32, weight [n] f64."""
    _, A, _ = _size(cube_size)
    pitch = child_value.shape[-1]
    if child_value.dtype != torch.float32 or tuple(child_value.shape) != (A, pitch) or not child_value.is_contiguous():
        raise RubikHipError("adi_targets: child_value must be contiguous float32 [A, pitch]")
    if child_solved.dtype != torch.uint8 or tuple(child_solved.shape) != (A, pitch) or not child_solved.is_contiguous():
        raise RubikHipError("adi_targets: child_solved must be contiguous uint8 [A, pitch]")
    dev = child_value.device
    tv = torch.empty(n, dtype=torch.float32, device=dev)
    tp = torch.empty(n, dtype=torch.int32, device=dev)
    err = None
    if parent_value is not None:
        _vec(parent_value, n, torch.float32, "parent_value")
        _vec(weight, n, torch.float64, "weight")
        err = torch.empty(n, dtype=torch.float64, device=dev)
    _lib.init(dev)
    check(lib().rc_adi_targets(ptr(child_value), ptr(child_solved), ptr(parent_value), ptr(weight), n, pitch, cube_size,
                               ptr(tv), ptr(tp), ptr(err), stream_ptr(dev)))
    return tv, tp, err


def adi_targets_depths(child_value, cv_depth_stride, cv_child_stride, child_solved, parent_value, pv_depth_stride, weight, n, n_depths, cube_size,
                       target_value, target_policy, error):
    """rc_adi_targets for a GROUP of depths straight out of the value net's output (cube_env.py:229-232,239-251), results walk-major.
    child_value / parent_value: float32 views INTO the net's output; element [g, a, w] of the former sits at g * cv_depth_stride +
    a * cv_child_stride + w, [g, w] of the latter at g * pv_depth_stride + w.  child_solved: uint8 [n_depths, A, Wp] (contiguous slice of
    the generator's flags).  weight: float64 [n_depths] (d ** -temperature per depth).  target_value / target_policy / error: 2-D views
    [>= n, >= n_depths] of walk-major outputs whose last dimension is contiguous (stride(0) = elements between walks)."""
    _, A, _ = _size(cube_size)
    if child_solved.dtype != torch.uint8 or child_solved.dim() != 3 or tuple(child_solved.shape[:2]) != (n_depths, A) or not child_solved.is_contiguous():
        raise RubikHipError("adi_targets_depths: child_solved must be contiguous uint8 [n_depths, A, Wp]")
    wp = child_solved.shape[2]
    for t, dt, what in ((child_value, torch.float32, "child_value"), (parent_value, torch.float32, "parent_value"), (weight, torch.float64, "weight")):
        if t.dtype != dt or not t.is_cuda:
            raise RubikHipError(f"adi_targets_depths: {what} must be a {dt} HIP tensor")
    if weight.numel() < n_depths or not weight.is_contiguous():
        raise RubikHipError("adi_targets_depths: need one contiguous weight per depth")
    if child_value.numel() < (n_depths - 1) * cv_depth_stride + (A - 1) * cv_child_stride + n or parent_value.numel() < (n_depths - 1) * pv_depth_stride + n:
        raise RubikHipError("adi_targets_depths: value views are too short for the strides given")
    stride = None
    for t, dt, what in ((target_value, torch.float32, "target_value"), (target_policy, torch.int32, "target_policy"), (error, torch.float64, "error")):
        if t.dtype != dt or not t.is_cuda or t.dim() != 2 or t.shape[0] < n or t.shape[1] < n_depths or t.stride(1) != 1:
            raise RubikHipError(f"adi_targets_depths: {what} must be a {dt} HIP view [>= n, >= n_depths] with a contiguous last dimension")
        if stride is not None and t.stride(0) != stride:
            raise RubikHipError("adi_targets_depths: the three outputs must share one walk stride")
        stride = t.stride(0)
    dev = child_solved.device
    _lib.init(dev)
    check(lib().rc_adi_targets_depths(ptr(child_value), cv_depth_stride, cv_child_stride, ptr(child_solved), wp, ptr(parent_value), pv_depth_stride,
                                      ptr(weight), n, n_depths, cube_size, ptr(target_value), ptr(target_policy), ptr(error), stride, stream_ptr(dev)))
