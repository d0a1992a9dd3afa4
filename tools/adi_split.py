#!/usr/bin/env python3
"""Split a rocprofv3 kernel trace of tools/bench_adi_pipeline.py into env / model / glue / idle microseconds per adi_samples call.

    python tools/adi_split.py <dir with *_kernel_trace.csv> [label]   -> one JSON object on stdout

bench_adi_pipeline.py brackets every timed call with three k_fill_solved launches; the window of a call runs from the END of
the last marker before it to the START of the first marker after it.  Classes:
  env    librubikhip kernels of the path (k_adi*, k_code_to_dense*, k_adi_targets*, k_step*, ...)
  gemm   the value net's matrix products (hipBLASLt / Tensile "Cijk_*", rocBLAS gemm)
  net_elementwise   the net's bias / ELU kernels (ATen elementwise over float tensors launched between GEMMs)
  glue   every other ATen kernel (copies, fills, index, cat)
  idle   window time during which no kernel runs (launch gaps: the host is the bottleneck there)"""
import csv
import glob
import json
import os
import sys


def classify(name):
    n = name
    if "k_fill_solved" in n:
        return "marker"
    if "(anonymous namespace)::k_" in n or n.startswith("k_") or "rc::" in n:
        return "env"
    if "Cijk_" in n or "gemm" in n.lower() or "hipblaslt" in n.lower():
        return "gemm"
    if "elu" in n.lower() or "add" in n.lower() and "elementwise" in n.lower():
        return "net_elementwise"
    return "glue"


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit(f"no kernel trace under {d}")
    rows = []
    for r in csv.DictReader(open(files[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if classify(r[2]) == "marker"]
    # groups of three consecutive markers; calls sit between group 2k and group 2k + 1
    groups = []
    for i in marks:
        if groups and i == groups[-1][-1] + 1:
            groups[-1].append(i)
        else:
            groups.append([i])
    groups = [g[i:i + 3] for g in groups for i in range(0, len(g) - 2, 3)]          # back-to-back calls: runs of six
    calls = []
    for k in range(0, len(groups) - 1, 2):
        lo, hi = groups[k][-1], groups[k + 1][0]
        t0, t1 = rows[lo][1], rows[hi][0]
        cls = {"env": 0, "gemm": 0, "net_elementwise": 0, "glue": 0}
        env_detail = {"k_adi": 0, "k_code_to_dense_front": 0, "k_adi_targets": 0, "other": 0}
        per_kernel = {}
        busy, cur_end, launches = 0, t0, 0
        for s, e, name in rows[lo + 1:hi]:
            c = classify(name)
            cls[c] += e - s
            if c == "env":
                key = "k_adi_targets" if "k_adi_targets" in name else "k_adi" if "k_adi<" in name else "k_code_to_dense_front" if "k_code_to_dense_front" in name else "other"
                env_detail[key] += e - s
            launches += 1
            short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
            pk = per_kernel.setdefault(short, [0, 0])
            pk[0] += 1
            pk[1] += e - s
            s2 = max(s, cur_end)
            if e > s2:
                busy += e - s2
                cur_end = e
        window = t1 - t0
        calls.append({"window_us": window / 1e3, "launches": launches, **{k_ + "_us": v / 1e3 for k_, v in cls.items()},
                      "env_detail_us": {k_: round(v / 1e3, 1) for k_, v in env_detail.items()},
                      "idle_us": (window - busy) / 1e3,
                      "top_kernels": {k_: {"calls": v[0], "us": round(v[1] / 1e3, 1)} for k_, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:8]}})
    calls.sort(key=lambda c: c["window_us"])
    out = {"label": sys.argv[2] if len(sys.argv) > 2 else os.path.basename(d.rstrip("/")), "calls_traced": len(calls)}
    if calls:
        med = calls[len(calls) // 2]
        out.update({k: (round(v, 1) if isinstance(v, float) else v) for k, v in med.items()})
        out["note"] = "median call (by window) of the traced ones; window = end of the last marker before the call to the start of the first after it"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
