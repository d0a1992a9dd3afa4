#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of one gpu_round.sh run from gpurun_out/ into profiles/ (tracked).

    python tools/summarize_profiles.py r01a r01      # <gpurun tag> <round name>
Writes profiles/<round>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <round>_pmc.json (mean
FETCH_SIZE / WRITE_SIZE per kernel, separate passes) and refreshes profiles/traffic.json, which bench.py
reads for roofline.traffic.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 wide
coalesced streaming reads (it tallies 128-B requests at 64 B); WRITE_SIZE is exact; both are in KiB."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
stats = glob.glob(os.path.join(G, f"{tag}_prof_stats", "**", "*_kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(P, f"{rnd}_kernel_stats.csv"))
micro = glob.glob(os.path.join(G, f"{tag}_prof_micro", "**", "*_kernel_stats.csv"), recursive=True)
if micro:
    shutil.copy(micro[0], os.path.join(P, f"{rnd}_kernel_stats_microbench.csv"))
pmc = {}
for name in ("fetch", "write", "sq", "microwrite"):
    for f in glob.glob(os.path.join(G, f"{tag}_prof_{name}", "**", "*_counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            short = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:80]
            pmc.setdefault(short, {})[c] = {"mean": sum(v) / len(v), "launches": len(v)}
json.dump(pmc, open(os.path.join(P, f"{rnd}_pmc.json"), "w"), indent=1, sort_keys=True)
step = [v for k, v in pmc.items() if k.startswith("k_step<rc::Cube3") and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
if step:
    s = max(step, key=lambda v: v["FETCH_SIZE"]["launches"])
    rd, wr = s["FETCH_SIZE"]["mean"] * 1024 * 2, s["WRITE_SIZE"]["mean"] * 1024
    json.dump({"round": rnd, "k_step_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
               "note": "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE KiB x 1024, separate --pmc passes, "
                       "bench.py workload (2^22 cubes, move + done flag); algorithmic = 110 B x 2^22 = 461373440"},
              open(os.path.join(P, "traffic.json"), "w"), indent=1)
print(open(os.path.join(P, f"{rnd}_kernel_stats.csv")).read()[:1500] if stats else "no stats")
print(json.dumps(pmc, indent=1)[:1500])
