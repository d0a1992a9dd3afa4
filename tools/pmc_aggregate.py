#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc output ON THE GPU BOX: <dir>/**/*_counter_collection.csv -> <dir>/pmc_aggregate.json
{kernel: {counter: {"mean": ..., "launches": ...}}} for the library's kernels, then delete the raw CSVs (a bench.py pass with every
config writes tens of MB of them; gpurun merges at most 64 MiB back).   python tools/pmc_aggregate.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import json
import os
import sys


def short(k):
    return k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:90]


for d in sys.argv[1:]:
    out = {}
    files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    for f in files:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            out.setdefault(short(k), {})[c] = {"mean": sum(v) / len(v), "launches": len(v), "min": min(v), "max": max(v)}
        os.remove(f)
    out = {k: v for k, v in out.items() if k.startswith("k_")}
    if files:
        json.dump(out, open(os.path.join(d, "pmc_aggregate.json"), "w"), indent=1, sort_keys=True)
    print(d, len(files), "file(s)", len(out), "kernel(s)")
