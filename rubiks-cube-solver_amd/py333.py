"""Reference-named operator layer on the device: one cube per call, numpy in / numpy out.

Same names, argument meaning and error behaviour as the reference's operator modules --
3x3x3: gym-cube/gym_cube/envs/assets/py333.py:211-246 (`initState_3`, `doMove_3`, `getOP_3`,
`isSolved_3`, `pos_to_state_3`); 2x2x2: the names cube_env.py:8 imports from the absent
`assets.py222` (`initState`, `doMove`, `isSolved`, `getOP`, `getStickers`; convention unpinned,
DESIGN.md section 2).  Every function is a thin round trip through librubikhip.so (upload one
cube, one kernel, download); the batched forms live in `ops`.  Useful for parity tests that read
like the reference's own call sites, not for throughput.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops
from .tables import ACTION_NAMES, get_tables

moveInds = {name: i for i, name in enumerate(ACTION_NAMES[3])}      # py333.py:41-44
moveInds_2 = {name: i for i, name in enumerate(ACTION_NAMES[2])}


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _up(s, cube_size):
    a = np.asarray(s).reshape(1, -1).astype(np.uint8)
    if a.shape[1] != ops.N_STICKERS[cube_size]:
        raise IndexError(f"a {cube_size}x{cube_size}x{cube_size} state has {ops.N_STICKERS[cube_size]} stickers")
    return ops.from_aos(a, _dev())


def _init(cube_size):
    st = ops.alloc_states(1, cube_size, _dev())
    ops.fill_solved(st, 1, cube_size)
    return ops.to_aos(st, 1)[0].cpu().numpy().astype(np.int64)


def _move(s, move, cube_size, table):
    a = table[move]                                               # KeyError on an unknown move string (py333.py:221)
    st = _up(s, cube_size)
    ops.apply_moves(st, st, torch.tensor([a], dtype=torch.uint8, device=st.device), 1, cube_size)
    return ops.to_aos(st, 1)[0].cpu().numpy().astype(np.int64)   # a fresh array, like s[moveDefs[move]]


def _solved(s, cube_size):
    done = torch.zeros(16, dtype=torch.uint8, device=_dev())
    ops.is_solved(_up(s, cube_size), 1, cube_size, done)
    return bool(done[0].item())


def _op(s, cube_size):
    code = ops.alloc_code(1, cube_size, _dev())
    ops.encode(_up(s, cube_size), 1, cube_size, code, _lib.FMT_CODE)
    c = ops.to_aos(code, 1)[0].cpu().numpy().astype(np.int64)
    nc = len(get_tables(cube_size).corner_defs)
    mult = np.where(np.arange(len(c)) < nc, 3, 2)
    return np.stack([c // mult, c % mult], 1)                     # rows (piece, orientation)


# ------------------------------------------------------------------------------- 3x3x3
def initState_3():
    return _init(3)


def doMove_3(s, move):
    return _move(s, move, 3, moveInds)


def isSolved_3(s):
    return _solved(s, 3)


def getOP_3(s):
    return _op(s, 3)


def pos_to_state_3(pos):
    """[20,2] (piece, orientation) rows -> int one-hot [20,24] (py333.py:235-246)."""
    pos = np.asarray(pos).astype(np.int64)
    code = (pos[:, 0] * np.where(np.arange(len(pos)) < 8, 3, 2) + pos[:, 1]).astype(np.uint8)
    if code.max() >= 24:
        raise IndexError("index out of bounds for axis with size 24")
    buf = ops.from_aos(code.reshape(1, -1), _dev())
    out = torch.empty((1, 20, 24), dtype=torch.uint8, device=buf.device)
    ops.onehot_from_code(buf, 1, 3, out)
    return out[0].cpu().numpy().astype(np.int64)


# ------------------------------------------------------------------------------- 2x2x2
def initState():
    return _init(2)


def doMove(s, move):
    return _move(s, move, 2, moveInds_2)


def isSolved(s):
    return _solved(s, 2)


def getOP(s):
    return _op(s, 2)


def getStickers(sOP):
    """[7,2] (piece, orientation) per slot -> 24 stickers; host-side table inversion (no caller in the reference
    besides the unused state_to_sim_state, cube_env.py:165-170)."""
    t = get_tables(2)
    s = np.array(t.solved, dtype=np.int64)
    for slot, (piece, ori) in enumerate(np.asarray(sOP).astype(np.int64)):
        col = [int(t.solved[i]) for i in t.corner_defs[piece]]
        rot = col[-ori:] + col[:-ori] if ori else col
        for k in range(3):
            s[t.corner_defs[slot][k]] = rot[k]
    return s
