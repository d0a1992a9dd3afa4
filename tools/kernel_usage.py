#!/usr/bin/env python3
"""Cross-compile rubikhip.hip for gfx950 to assembly (no GPU needed) and print one line per kernel: registers, scratch,
LDS, VALU count and the FORM of every memory instruction (flat / global / buffer / scratch, 64-bit VALU address adds).

    python tools/kernel_usage.py [filter ...] [--json profiles/r02_isa_summary.json]

This is the check behind DESIGN.md's claims "0 scratch", "0 flat_store / no per-access v_lshl_add_u64 on the row paths"."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rubiks-cube-solver_amd", "csrc", "rubikhip.hip")
args = sys.argv[1:]
out_json = None
if "--json" in args:
    i = args.index("--json")
    out_json = args[i + 1]
    del args[i:i + 2]
with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "rubikhip.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", asm, SRC],
                          stderr=subprocess.DEVNULL)
    s = open(asm).read()
kern = re.findall(r"^(\S+):\s*; @\S+\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M)
names = subprocess.run(["c++filt"], input="\n".join(k for k, _ in kern), capture_output=True, text=True).stdout.splitlines()
rows = []
for (mangled, body), dem in zip(kern, names):
    n = dem.replace("(anonymous namespace)::", "").replace("rc::", "").replace("void ", "")
    n = re.sub(r"\(.*$", "", n)
    if args and not any(a in n for a in args):
        continue
    g = lambda k: int((re.search(r"\.amdhsa_" + k + r" (\d+)", body) or [0, 0])[1])
    c = collections.Counter(re.findall(r"^\s+((?:flat|global|buffer|scratch)_(?:load|store|atomic)\w*|v_lshl_add_u64|v_accvgpr_\w+)\b", body, re.M))
    mem = collections.Counter()
    for k, v in c.items():
        mem[k.split("_")[0] + "_" + k.split("_")[1] if not k.startswith("v_") else k] += v
    rows.append({"kernel": n, "vgpr_incl_agpr": g("next_free_vgpr"), "sgpr": g("next_free_sgpr"), "scratch_bytes": g("private_segment_fixed_size"),
                 "lds_bytes": g("group_segment_fixed_size"), "valu_instructions": len(re.findall(r"^\s+v_\w+", body, re.M)), "memory_forms": dict(mem)})
for r in rows:
    print(f"{r['kernel']:58s} vgpr={r['vgpr_incl_agpr']:>4} sgpr={r['sgpr']:>4} scratch={r['scratch_bytes']:>4} lds={r['lds_bytes']:>6} "
          f"valu={r['valu_instructions']:>6} {r['memory_forms']}")
if out_json:
    json.dump({"command": "hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S rubikhip.hip (tools/kernel_usage.py)", "kernels": rows},
              open(out_json, "w"), indent=1)
