"""Import alias: the package lives in the directory `rubiks-cube-solver_amd/` (not a valid
Python identifier); this shim makes it importable as `rubiks_cube_solver_amd`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "rubiks-cube-solver_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _f.name, "exec"))
