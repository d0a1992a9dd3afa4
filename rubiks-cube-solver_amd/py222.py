"""The six names the reference imports from the file it does not ship -- `from assets.py222 import initState, getOP, doMove, isSolved,
getStickers, printCube` (gym-cube/gym_cube/envs/cube_env.py:8; listed in gym_cube.egg-info/SOURCES.txt:15, absent from the tree) --
on the device, one cube per call, with the call shapes the reference's call sites use (cube_env.py:38,86-89,144-145,165-170,215-217):

    initState() -> int64[24]      doMove(s, "U'") -> a new int64[24]      isSolved(s) -> bool
    getOP(s) -> int64[7, 2] rows (piece, orientation) per position        getStickers(int[7, 2]) -> int64[24]      printCube(s)

A maintainer of the reference who wants ITS OWN CubeEnv(cube_size=2) to run can drop `from rubiks_cube_solver_amd.py222 import *`
into `gym_cube/envs/assets/py222.py`.  The sticker numbering, piece order and orientation numbering are the build's restatement of the
public MeepMoop/py222 model (PARITY UNPINNED: DESIGN.md section 2); every function is a thin round trip through librubikhip.so
(py333.py in this package holds the implementations next to the 3x3x3 names)."""
from .py333 import doMove, getOP, getStickers, initState, isSolved  # noqa: F401

__all__ = ["initState", "getOP", "doMove", "isSolved", "getStickers", "printCube"]


def printCube(s):
    """The 24 stickers as an unfolded cube (U on top, then L F R B in a row, D below); faces U0-3 R4-7 F8-11 D12-15 L16-19 B20-23."""
    s = [int(x) for x in s]
    f = lambda k: (f"{s[k]} {s[k + 1]}", f"{s[k + 2]} {s[k + 3]}")
    u, r, fr, d, l, b = f(0), f(4), f(8), f(12), f(16), f(20)
    print("\n".join(["    " + u[0], "    " + u[1], " ".join((l[0], fr[0], r[0], b[0])), " ".join((l[1], fr[1], r[1], b[1])), "    " + d[0], "    " + d[1]]))
