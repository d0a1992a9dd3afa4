"""The C ABI from plain C: compile + link with gcc here (no GPU), run on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "abi", "abi_smoke.c")
EXE = os.path.join(ROOT, "tests", "abi", "abi_smoke.out")
LIBDIR = os.path.join(ROOT, "rubiks-cube-solver_amd")


def build():
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-O1", f"-I{rocm}/include", f"-I{ROOT}/include", SRC, "-o", EXE,
           f"-L{LIBDIR}", "-lrubikhip", f"-L{rocm}/lib", "-lamdhip64", f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{rocm}/lib"]
    subprocess.check_call(cmd)
    return EXE


def test_c_consumer_compiles_and_links():
    if not os.path.exists(os.path.join(LIBDIR, "librubikhip.so")):
        pytest.skip("librubikhip.so not built")
    assert os.path.exists(build())


@pytest.mark.gpu
def test_c_consumer_runs():
    exe = build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "abi_smoke ok" in out.stdout
