#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of one tools/gpu_round.sh run from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/summarize_profiles.py r02a r02      # <gpurun tag> <round name>

Writes
  profiles/<round>_bench.json               the bench.py JSON line of the session
  profiles/<round>_kernel_stats.csv         rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu --no-configs --steps 200 --warmup 20`
  profiles/<round>_kernel_stats_configs.csv the same with the configs (k_adi, k_expand, dense kernels, ...)
  profiles/<round>_pmc.json                 mean FETCH_SIZE / WRITE_SIZE / SQ counters per kernel (separate --pmc passes)
  profiles/<round>_adi_fresh_processes.json k_adi / k_expand averages of FIVE fresh processes (rocprofv3 kernel stats each)
  profiles/<round>_design_kernel_stats.csv  kernel-stat rows of the design A/B harness (tools/exp/exp_step2 22 0 1)
  profiles/<round>_{cfg5,rollout,facade,adi_pipeline}.json   tool outputs of the same session
  profiles/traffic.json                     HBM-side bytes per k_step launch, read by bench.py for roofline.traffic
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 wide coalesced streaming reads (it tallies
128-B requests at 64 B); WRITE_SIZE is exact; both are in KiB."""
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


def short(k):
    return k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:90]


def first(pattern):
    f = glob.glob(os.path.join(G, pattern), recursive=True)
    return f[0] if f else None


for name in ("bench", "cfg5", "rollout", "facade"):      # (adi_pipeline: profiles/rNN_adi_pipeline.json is the trace split, written by hand)
    src = os.path.join(G, f"{tag}_{name}.json")
    if os.path.exists(src) and os.path.getsize(src):
        lines = [l for l in open(src).read().splitlines() if l.startswith("{")]
        if lines:
            json.dump(json.loads(lines[-1]), open(os.path.join(P, f"{rnd}_{name}.json"), "w"), indent=1)

stats = first(f"{tag}_prof_stats/**/*_kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(P, f"{rnd}_kernel_stats.csv"))
statscfg = first(f"{tag}_prof_statscfg/**/*_kernel_stats.csv")
if statscfg:
    shutil.copy(statscfg, os.path.join(P, f"{rnd}_kernel_stats_configs.csv"))
design = first(f"{tag}_prof_design/**/*_kernel_stats.csv")
if design:
    shutil.copy(design, os.path.join(P, f"{rnd}_design_kernel_stats.csv"))

fresh = []
for i in range(1, 6):
    f = first(f"{tag}_prof_adi{i}/**/*_kernel_stats.csv")
    if not f:
        continue
    row = {"process": i}
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k.startswith(("k_adi<", "k_expand<")):
            row[k] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
    fresh.append(row)
if fresh:
    W, D = 100_000, 30
    key = "k_adi<rc::Cube3, 2, false, false>"
    avgs = [r[key]["avg_us"] for r in fresh if key in r]
    json.dump({"command": "rocprofv3 --kernel-trace --stats -- python3 tools/microbench.py adi expand   (x5, a fresh process each)",
               "workload": "k_adi: 100000 walks x depth 30, parents + 12 children + flags + actions, default dispatch and tiling; "
                           "k_expand<..., 2, false>: 2^20 parents, 3 output tilings",
               "algorithmic_bytes_per_launch_k_adi": 715 * W * D,
               "k_adi_avg_us_per_process": avgs,
               "k_adi_GBps_per_process": [715 * W * D / (a * 1e-6) / 1e9 for a in avgs],
               "k_adi_frac_of_8TBps_per_process": [715 * W * D / (a * 1e-6) / 8e12 for a in avgs],
               "processes": fresh}, open(os.path.join(P, f"{rnd}_adi_fresh_processes.json"), "w"), indent=1)

def read_pmc(names):
    out = {}
    for name in names:
        agg_file = os.path.join(G, f"{tag}_prof_{name}", "pmc_aggregate.json")     # reduced on the GPU box by tools/pmc_aggregate.py
        if os.path.exists(agg_file):
            for k, v in json.load(open(agg_file)).items():
                out.setdefault(k, {}).update({c: {"mean": x["mean"], "launches": x["launches"]} for c, x in v.items()})
            continue
        for f in glob.glob(os.path.join(G, f"{tag}_prof_{name}", "**", "*_counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, c), v in agg.items():
                out.setdefault(short(k), {})[c] = {"mean": sum(v) / len(v), "launches": len(v)}
    return {k: v for k, v in out.items() if k.startswith("k_")}     # our kernels only (torch's GEMMs etc. are not the subject)


headline = read_pmc(("fetch", "write"))            # bench.py --no-configs: only the headline launches of k_step
pmc = read_pmc(("fetchcfg", "writecfg", "sq"))     # with the configs: k_adi, k_expand, dense kernels, ...
HEAD = "k_step<rc::Cube3, 2, true, true, false, 1, 64>"
if HEAD in headline:
    pmc[HEAD + " [headline launches only]"] = headline[HEAD]
    pmc.setdefault(HEAD, {})["note"] = ("means over the headline AND the reward-bearing config launches of the same instantiation (+4 B per cube on "
                                        "some): use the [headline launches only] entry for the bench's roofline.traffic")
if pmc:
    json.dump(pmc, open(os.path.join(P, f"{rnd}_pmc.json"), "w"), indent=1, sort_keys=True)
if HEAD in headline and "FETCH_SIZE" in headline[HEAD] and "WRITE_SIZE" in headline[HEAD]:
    s_ = headline[HEAD]
    rd, wr = s_["FETCH_SIZE"]["mean"] * 1024 * 2, s_["WRITE_SIZE"]["mean"] * 1024
    json.dump({"round": rnd, "kernel": HEAD, "k_step_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
               "launches_counted": {"FETCH_SIZE": s_["FETCH_SIZE"]["launches"], "WRITE_SIZE": s_["WRITE_SIZE"]["launches"]},
               "note": "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE KiB x 1024, separate --pmc passes over "
                       "`python3 bench.py --no-cpu --no-configs --steps 10 --warmup 2` (2^22 cubes, move + done flag; the headline launches only); "
                       "algorithmic = 110 B x 2^22 = 461373440.  These are the L2's fabric-side request counters: Infinity-Cache hits are counted, "
                       "so this is fabric traffic, an upper bound of DRAM traffic"},
              open(os.path.join(P, "traffic.json"), "w"), indent=1)
for k in sorted(pmc):
    print(k, {c: round(v["mean"], 1) for c, v in pmc[k].items() if c in ("FETCH_SIZE", "WRITE_SIZE")})
print(open(os.path.join(P, f"{rnd}_kernel_stats.csv")).read()[:600] if stats else "no stats")
