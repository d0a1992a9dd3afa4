"""A stand-in for the file the reference imports but does not ship: `assets/py222.py` (cube_env.py:8 imports initState, getOP, doMove,
isSolved, getStickers, printCube from it; gym_cube.egg-info/SOURCES.txt:15 lists it, the tree does not hold it).

HARNESS SIDE ONLY (tests/golden/make_golden.py plugs these six names into the imported reference module so that the reference's OWN
CubeEnv runs with cube_size = 2).  The functions restate the public MeepMoop/py222 2x2x2 model from the oracle's tables
(oracle/oracle_np.tables_222: the six U / F / R permutations = the corner-sticker restriction of the reference's 3x3x3 table, the seven
piece definitions, the 58-row hash table) in plain numpy, with the call shapes the reference's call sites need:

    initState()            -> int64[24]                 cube_env.py:38
    doMove(s, "U'")        -> a NEW int64[24]           cube_env.py:86-87,215 (the authors' version takes the move STRING)
    isSolved(s)            -> bool                      cube_env.py:89,217
    getOP(s)               -> int64[7, 2] (piece, ori)  cube_env.py:144-145
    getStickers(int[7, 2]) -> int64[24]                 cube_env.py:165-170

What a fixture made THROUGH this stand-in pins: every line of the reference's 2x2x2 branches in cube_env.py (init_state, reset's RNG
handling, step's reward / done, the TRANSPOSED one-hot convention of sim_state_to_state, state_to_sim_state's inversion, the child loop
and target rule of get_target_value, get_random_samples) and of mcts.py with action_dim 6 -- executed, not restated.  What it cannot pin:
the CONTENT of the authors' py222 tables (sticker numbering, piece order, orientation numbering): PARITY UNPINNED there, as everywhere
for 2x2x2; the shipped 2x2x2 checkpoint is the only evidence for those (tests/golden/make_crosscheck_222.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.oracle_np import ACTION_NAMES, tables_222

_T = tables_222()
_PERM = _T["perm"].astype(np.int64)
_DEFS = _T["corner_defs"].astype(np.int64)
_LUT = _T["corner_lut"].astype(np.int64)
_MOVE = {name: i for i, name in enumerate(ACTION_NAMES[2])}
_HASH = np.array([1, 2, 10])


def initState():
    return np.repeat(np.arange(6), 4)


def doMove(s, move):
    return np.asarray(s)[_PERM[_MOVE[move]]]          # KeyError on an unknown move string


def isSolved(s):
    s = np.asarray(s)
    return bool(all((s[4 * f:4 * f + 4] == s[4 * f]).all() for f in range(6)))


def getOP(s):
    return _LUT[np.asarray(s)[_DEFS] @ _HASH]


def getStickers(sOP):
    s = np.repeat(np.arange(6), 4)
    solved = s.copy()
    for slot, (piece, ori) in enumerate(np.asarray(sOP).astype(np.int64)):
        col = [int(solved[i]) for i in _DEFS[piece]]
        rot = col[-ori:] + col[:-ori] if ori else col
        for k in range(3):
            s[_DEFS[slot][k]] = rot[k]
    return s


def printCube(s):
    print(" ".join(map(str, np.asarray(s))))
