#!/usr/bin/env python3
"""Compile librubikhip.so with -Rpass-analysis=kernel-resource-usage and print one line per kernel."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rubiks-cube-solver_amd", "csrc", "rubikhip.hip")
OUT = os.path.join(ROOT, "rubiks-cube-solver_amd", "librubikhip.so")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", OUT, SRC,
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
p = subprocess.run(cmd, capture_output=True, text=True)
rows, cur = [], None
for l in p.stderr.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", l)
    if not m:
        if "error" in l or "warning" in l:
            print(l)
        continue
    t = m.group(1)
    if t.startswith("Function Name"):
        cur = {"name": t.split(": ")[1]}
        rows.append(cur)
    elif ":" in t and cur is not None:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").replace("rc::", "")
    n = re.sub(r"\(.*$", "", n).replace("void ", "")
    print(f"{n:62s} vgpr={r.get('VGPRs'):>4} agpr={r.get('AGPRs'):>3} scratch={r.get('ScratchSize [bytes/lane]'):>4} "
          f"occ={r.get('Occupancy [waves/SIMD]')} lds={r.get('LDS Size [bytes/block]')}")
sys.exit(p.returncode)
