"""Cube tables: the single source of truth for the HIP kernels, the C-ABI and the facade.

Nothing here is a transcription of the reference's 12x54 table.  The sticker
permutations are DERIVED from cube geometry given only the reference's sticker
numbering and colour order (docstring of gym-cube/gym_cube/envs/assets/py333.py:3-19:
faces in the order U,R,F,D,L,B, nine stickers per face numbered row-major as seen on
the unfolded net, colour of a face = its index) and the gather convention
``new[i] = old[perm[a][i]]`` (py333.py:220-222).  tests/test_tables.py checks the result
against the reference's own table captured in tests/golden/tables_333.npz.

What cannot be derived is restated as data, citing where it comes from:
  * the order of the 8 corner / 12 edge slots and of the stickers inside each slot
    (py333.py:140-164), written here as face-letter strings;
  * the hash -> (piece, orientation) look-up tables (py333.py:171-198), INCLUDING the
    reference's wrong / missing corner entries: rows never assigned stay (0, 0).
    Reproducing them is required for bit-exact one-hot states.

2x2x2: the reference imports ``assets.py222`` (cube_env.py:8) but does not ship it
 -> PARITY UNPINNED for 2x2x2 values.  We follow the published
algorithm of the public MeepMoop/py222 solver (no version is pinned by the reference):
24 stickers numbered like the 3x3x3 net, the DLB cubie fixed, moves U,U',F,F',R,R'
(cube_env.py:25), 7 piece slots, hash c0+2*c1+10*c2, orientation k = colour triple
rotated right k times.  The six permutations equal the corner-sticker restriction of
the 3x3x3 ones (asserted in tests).
"""
from __future__ import annotations

import functools
from dataclasses import dataclass

import numpy as np

FACES = "URFDLB"  # face order == colour order (py333.py:12-19)
_NORMAL = {
    "U": (0, 1, 0), "D": (0, -1, 0),
    "R": (1, 0, 0), "L": (-1, 0, 0),
    "F": (0, 0, 1), "B": (0, 0, -1),
}

# action order of the env API (cube_env.py:24-28, py333.py:41-44)
ACTION_NAMES = {
    2: ["U", "U'", "F", "F'", "R", "R'"],
    3: ["U", "U'", "F", "F'", "R", "R'", "D", "D'", "B", "B'", "L", "L'"],
}
# get_env_config (utils.py:162-186)
STATE_DIM = {2: (7, 21), 3: (20, 24)}
ACTION_DIM = {2: 6, 3: 12}

# slot order and sticker order inside a slot, as face letters (first letter = the
# sticker that is hashed with weight 1).  3x3x3: py333.py:140-164.
CORNER_SLOTS_3 = ["UBL", "ULF", "UFR", "URB", "DLB", "DFL", "DFR", "DBR"]
EDGE_SLOTS_3 = ["UB", "UL", "UF", "UR", "DB", "DL", "DF", "DR", "FL", "FR", "BR", "BL"]
# 2x2x2 (public py222 pieceDefs): DLB is the fixed cubie and has no slot.
CORNER_SLOTS_2 = ["UBL", "ULF", "UFR", "URB", "DFL", "DRF", "DBR"]

CORNER_HASH = (1, 2, 10)  # py333.py:167
EDGE_HASH = (1, 10)  # py333.py:168

# hash -> (piece, orientation), exactly the rows the reference assigns (py333.py:172-180);
# every other row of the 62-row table is (0, 0).
_CORNER_LUT_3 = {
    50: (0, 0), 54: (0, 1), 13: (0, 2), 28: (1, 0), 8: (1, 1), 42: (1, 2),
    14: (2, 0), 5: (2, 1), 12: (2, 2), 52: (3, 0), 11: (3, 1), 15: (3, 2),
    61: (4, 0), 44: (4, 1), 51: (4, 2), 47: (5, 0), 30: (5, 1), 40: (5, 2),
    17: (6, 0), 35: (6, 1), 18: (6, 2), 23: (7, 0), 56: (7, 1), 21: (7, 2),
}
_CORNER_LUT_3_ROWS = 62
# py333.py:184-198 (55-row table)
_EDGE_LUT_3 = {
    50: (0, 0), 5: (0, 1), 40: (1, 0), 4: (1, 1), 20: (2, 0), 2: (2, 1),
    10: (3, 0), 1: (3, 1), 53: (4, 0), 35: (4, 1), 43: (5, 0), 34: (5, 1),
    23: (6, 0), 32: (6, 1), 13: (7, 0), 31: (7, 1), 42: (8, 0), 24: (8, 1),
    12: (9, 0), 21: (9, 1), 15: (10, 0), 51: (10, 1), 45: (11, 0), 54: (11, 1),
}
_EDGE_LUT_3_ROWS = 55
_CORNER_LUT_2_ROWS = 58  # public py222: pieceInds = zeros([58, 2])

LUT_PAD = 72  # device LUTs are padded to 72 bytes (9 groups of 8) ; rows >= table size -> 0


# --------------------------------------------------------------------------- geometry
def _coords(n: int):
    return [2 * k - (n - 1) for k in range(n)]


def _sticker_geometry(n: int):
    """For every sticker index: (cubie position, outward normal), doubled integer coords.

    Row-major numbering of each face as drawn on the net of py333.py:3-11 (U on top,
    L F R B in a row, D below)."""
    v = _coords(n)
    geo = []
    for f in FACES:
        for r in range(n):
            for c in range(n):
                top = v[n - 1 - r]
                if f == "U":
                    p = (v[c], n - 1, v[r])
                elif f == "D":
                    p = (v[c], -(n - 1), v[n - 1 - r])
                elif f == "F":
                    p = (v[c], top, n - 1)
                elif f == "B":
                    p = (v[n - 1 - c], top, -(n - 1))
                elif f == "R":
                    p = (n - 1, top, v[n - 1 - c])
                else:  # L
                    p = (-(n - 1), top, v[c])
                geo.append((p, _NORMAL[f]))
    return geo


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _rot_cw(axis, vec):
    """Quarter turn, clockwise when looking at the face whose outward normal is `axis`
    (= -90 degrees about `axis` by the right-hand rule): v' = n(n.v) - n x v."""
    d = sum(a * b for a, b in zip(axis, vec))
    cr = _cross(axis, vec)
    return tuple(axis[i] * d - cr[i] for i in range(3))


def _face_turn(n: int, face: str) -> np.ndarray:
    geo = _sticker_geometry(n)
    where = {g: i for i, g in enumerate(geo)}
    axis = _NORMAL[face]
    perm = np.arange(len(geo))
    for j, (p, nm) in enumerate(geo):
        if sum(a * b for a, b in zip(axis, p)) == n - 1:  # cubie in the turning layer
            i = where[(_rot_cw(axis, p), _rot_cw(axis, nm))]
            perm[i] = j  # the sticker that was at j is now at i: new[i] = old[j]
    return perm


def _slot_stickers(n: int, slots):
    """Sticker indices of each slot: the cubie shared by the named faces; sticker k lies
    on the face named by letter k."""
    geo = _sticker_geometry(n)
    out = []
    for name in slots:
        row = []
        for letter in name:
            hits = [
                i for i, (p, nm) in enumerate(geo)
                if nm == _NORMAL[letter]
                and {l for l in FACES if sum(a * b for a, b in zip(_NORMAL[l], p)) == n - 1}
                == set(name)
            ]
            assert len(hits) == 1, (name, letter, hits)
            row.append(hits[0])
        out.append(row)
    return np.array(out, dtype=np.uint8)


# ------------------------------------------------------------------------------ tables
@dataclass(frozen=True)
class CubeTables:
    cube_size: int
    n_stickers: int  # S
    n_actions: int  # A
    action_names: tuple
    perm: np.ndarray  # uint8 [A][S]   new[i] = old[perm[a][i]]
    solved: np.ndarray  # uint8 [S]
    corner_defs: np.ndarray  # uint8 [n_corner_slots][3]
    edge_defs: np.ndarray  # uint8 [n_edge_slots][2]  (empty for 2x2x2)
    corner_lut: np.ndarray  # uint8 [rows][2]  (piece, ori), reference layout
    edge_lut: np.ndarray  # uint8 [rows][2]
    corner_code: np.ndarray  # uint8 [LUT_PAD]  piece*3+ori ; 0 beyond the table
    edge_code: np.ndarray  # uint8 [LUT_PAD]  piece*2+ori
    state_dim: tuple  # one-hot shape (R, C)
    n_slots: int  # compact code bytes per cube (20 | 7)

    @property
    def face_size(self):
        return self.cube_size * self.cube_size


def _lut_array(entries, rows):
    t = np.zeros((rows, 2), np.uint8)
    for h, (p, o) in entries.items():
        t[h] = (p, o)
    return t


def _code_array(lut, mult):
    c = np.zeros(LUT_PAD, np.uint8)
    c[: len(lut)] = lut[:, 0] * mult + lut[:, 1]
    return c


@functools.lru_cache(maxsize=None)
def get_tables(cube_size: int) -> CubeTables:
    if cube_size not in (2, 3):
        raise NotImplementedError(f"cube_size {cube_size}")  # cube_env.py:44
    n = cube_size
    names = ACTION_NAMES[n]
    turns = {f: _face_turn(n, f) for f in "UFRDBL"}
    perm = []
    for nm in names:
        p = turns[nm[0]]
        if nm.endswith("'"):
            p = np.argsort(p)  # inverse permutation
        perm.append(p)
    perm = np.array(perm, dtype=np.uint8)
    solved = np.repeat(np.arange(6, dtype=np.uint8), n * n)  # py333.py:211-218
    if n == 3:
        cdefs = _slot_stickers(3, CORNER_SLOTS_3)
        edefs = _slot_stickers(3, EDGE_SLOTS_3)
        clut = _lut_array(_CORNER_LUT_3, _CORNER_LUT_3_ROWS)
        elut = _lut_array(_EDGE_LUT_3, _EDGE_LUT_3_ROWS)
    else:
        cdefs = _slot_stickers(2, CORNER_SLOTS_2)
        edefs = np.zeros((0, 2), np.uint8)
        ent = {}
        for piece, d in enumerate(cdefs):
            col = [int(solved[i]) for i in d]
            for ori in range(3):  # colour triple rotated right `ori` times
                c = col[-ori:] + col[:-ori] if ori else col
                ent[sum(w * x for w, x in zip(CORNER_HASH, c))] = (piece, ori)
        assert len(ent) == 21
        clut = _lut_array(ent, _CORNER_LUT_2_ROWS)
        elut = np.zeros((1, 2), np.uint8)
    for a in range(0, len(names), 2):  # X' undoes X
        assert (perm[a][perm[a + 1]] == np.arange(perm.shape[1])).all()
    return CubeTables(
        cube_size=n, n_stickers=6 * n * n, n_actions=len(names), action_names=tuple(names),
        perm=perm, solved=solved, corner_defs=cdefs, edge_defs=edefs,
        corner_lut=clut, edge_lut=elut,
        corner_code=_code_array(clut, 3), edge_code=_code_array(elut, 2),
        state_dim=STATE_DIM[n], n_slots=len(cdefs) + len(edefs),
    )


# ------------------------------------------------------ pair-indexed code tables (device look-ups)
# sigma id of a reading order (j0, j1, j2) of a corner slot's three stickers, as rc_device.h numbers them
SIGMAS = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]


@functools.lru_cache(maxsize=None)
def pair_tables(cube_size: int):
    """Code look-ups indexed by TWO colours instead of the hash (the kernels' v_perm tables have 8 entries: a row per second
    colour, the first colour selects inside the row).

    edge_pair[c1][c0] = edge_code[c0 + 10*c1]: the hash is injective in (c0, c1), so this is exact for ANY colouring.
    Corners: on states reachable from the solved cube the first two colours of a slot read in a fixed order determine the
    third (a cubie is a fixed colour triple and a slot reads it in a fixed handedness), so
    corner_pair[table][c1][c0] = corner_code[c0 + 2*c1 + 10*c2(c0, c1)] -- including the reference's missing rows (zeros).
    The handedness depends on the slot and on the reading order: corner_table[q][sigma] names the table of slot q read in
    order SIGMAS[sigma].  Built by enumeration: random walks from the solved cube visit all 24 (21) triples of every slot; the
    generator asserts that (c0, c1) determines c2 and that walks of twice the length find nothing new.
    Returns (edge_pair uint8 [6][8], corner_pair uint8 [K][6][8], corner_table uint8 [NC][6])."""
    t = get_tables(cube_size)
    rng = np.random.default_rng(20260 + cube_size)
    A = t.n_actions
    seen = [set() for _ in t.corner_defs]
    st = np.tile(t.solved, (4096, 1))
    for step in range(60):
        acts = rng.integers(0, A, len(st))
        st = np.stack([s[t.perm[a]] for s, a in zip(st, acts)]) if step == 0 else st[np.arange(len(st))[:, None], t.perm[acts]]
        for q, d in enumerate(t.corner_defs):
            seen[q].update(map(tuple, st[:, d].tolist()))
        if step == 29:
            counts = [len(x) for x in seen]
    assert counts == [len(x) for x in seen] and set(counts) == {3 * (8 if cube_size == 3 else 7)}, counts
    tables, index = [], np.zeros((len(t.corner_defs), 6), np.uint8)
    for q in range(len(t.corner_defs)):
        for sid, sg in enumerate(SIGMAS):
            tab = np.zeros((6, 8), np.uint8)
            third = {}
            for tri in seen[q]:
                c0, c1, c2 = tri[sg[0]], tri[sg[1]], tri[sg[2]]
                assert third.setdefault((c0, c1), c2) == c2          # two colours determine the third
                h = c0 + 2 * c1 + 10 * c2
                tab[c1][c0] = t.corner_code[h] if h < LUT_PAD else 0
            for k, have in enumerate(tables):
                if (have == tab).all():
                    index[q, sid] = k
                    break
            else:
                index[q, sid] = len(tables)
                tables.append(tab)
    epair = np.zeros((6, 8), np.uint8)
    if len(t.edge_defs):
        for c1 in range(6):
            for c0 in range(6):
                epair[c1][c0] = t.edge_code[c0 + 10 * c1]
    return epair, np.stack(tables), index


def get_env_config(cube_size: int = 3):
    """([rows, cols] of the one-hot state, number of actions) -- utils.py:162-186."""
    if cube_size not in (2, 3):
        raise NotImplementedError(f"cube_size {cube_size}")
    return list(STATE_DIM[cube_size]), ACTION_DIM[cube_size]
