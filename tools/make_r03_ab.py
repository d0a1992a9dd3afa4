#!/usr/bin/env python3
"""Collect round 3's A/B measurements (tools/ab_r03.py runs on the GPU box, one JSON line per row in gpurun_out/r03*_*.jsonl)
into profiles/r03_ab.json, grouped by question, with the reading of each group.

    python tools/make_r03_ab.py
"""
import collections
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")


def rows(name):
    f = os.path.join(G, name)
    return [json.loads(l) for l in open(f)] if os.path.exists(f) else []


def table(rs, key, col, val="us"):
    t = collections.OrderedDict()
    for r in rs:
        t.setdefault(key(r), collections.OrderedDict()).setdefault(col(r), []).append(round(r[val], 1 if val == "us" else 3))
    return t


out = {"hardware": "MI355X (gfx950), one GPU per session, ROCm 7.2; every group is measured inside ONE process on one box (boxes differ by up to 10 % "
                   "on write-bound kernels)",
       "units": "us per launch (HIP events on the launch stream, median of 3 x N launches); frac = algorithmic bytes / time / 8 TB/s",
       "harness": "tools/ab_r03.py through the shipped library's per-call `variant` (and RUBIKHIP_LIB for other builds of the same sources)"}

# 1. side-output policy and reward store shape of the step kernel
a_new, a_old = rows("r03a_ab_new.jsonl"), {r["k"] + str(r.get("rep", "")): r for r in rows("r03a_ab_r02.jsonl")}
out["step_side_outputs_4M"] = {
    "what": "2^22 cubes ping-pong; round 2's build (side outputs = code / done / reward stored with the state's keep policy) against the first round-3 "
            "build (side outputs streamed, sc0 sc1 nt); var = pack width (1, 2) + 10 x forced policy",
    "rows": [{"k": r["k"], "r03_first_us": r["us"], "r03_first_frac": r["frac"], "r02_us": a_old[r["k"] + str(r.get("rep", ""))]["us"],
              "r02_frac": a_old[r["k"] + str(r.get("rep", ""))]["frac"]} for r in a_new if r["k"].startswith("step_") and r["k"] + str(r.get("rep", "")) in a_old],
    "reading": "streaming the 84 MB of code past the Infinity Cache took the fused-code launch from 107.6 to 82.2 us at 4 cubes per lane; at 8 cubes per "
               "lane nothing moved (92 us) and the reward variant got SLOWER (69.8 -> 75.5 us): a lane's two 16-byte reward stores sit 32 bytes apart, "
               "so every streamed store instruction wrote half lines.  With the rewards shuffled so that each instruction writes 1 KiB of contiguous "
               "floats (next group) 8 cubes per lane win everywhere"}
out["step_size_sweep_final"] = {
    "what": "pack width 4 (V1) / 8 (V2) cubes per lane x {done, reward + done, reward + done + code, in place} x 2^18 .. 2^24 cubes, default policy "
            "(POL 0 resident, 3 state resident + side streamed, 1 keep output, 2 stream); build with contiguous reward stores, POL 3 and the two-colour tables",
    "table_us": table([r for r in rows("r03d_ab.jsonl") if r["k"].startswith("step_")], lambda r: r["k"].rsplit("_", 1)[0], lambda r: r["k"].rsplit("_", 1)[1]),
    "before_contiguous_reward_stores_us": table([r for r in rows("r03b_ab.jsonl") if r["k"].startswith("step_")], lambda r: r["k"].rsplit("_", 1)[0],
                                                 lambda r: r["k"].rsplit("_", 1)[1]),
    "reading": "V2 wins or ties from 2^18 cubes up (V1 is ~2 % ahead around 2^21); 4M with the fused code 81-82 us = 0.86-0.87 (round 2: 93.1 us, 0.755); "
               "16M (nothing cached) 0.81-0.83 for every output set (before the reward fix: 0.73 with the reward)"}
# 2. ADI
out["adi_depth_segments_parts_packs"] = {
    "what": "code-only ADI, 100k walks x 30: pack width x parts x depth segments (first round-3 build, hash + 72-entry look-ups)",
    "table_us": table([r for r in a_new if r["k"].startswith("adi_")], lambda r: r["k"].rsplit("_segs", 1)[0], lambda r: "segs" + r["k"].rsplit("_segs", 1)[1] if "_segs" in r["k"] else "default"),
    "reading": "one wave per 256 walks (391 waves) leaves most SIMDs idle: 193 us; 2 parts 161 us (duplicated look-ups); 8 walks per lane x 3 segments "
               "151 us; more segments lose to the replayed moves and to waves sharing a SIMD (the counts are not monotone: 782 waves fit one per SIMD, 1173 do not)"}
out["adi_output_tile_pitch"] = {
    "what": "the same over output tile pitches 512 .. 32768 walks (pitch 512 = every wave's 20 code rows contiguous)",
    "table_us": table([r for r in rows("r03b_ab.jsonl") if r["k"].startswith("adi_")], lambda r: r["k"].split("_pitch")[0] + "_" + r["k"].split("_", 3)[3], lambda r: r["k"].split("_")[2]),
    "reading": "code-only generation does not care where its rows land (flat within 2 %): it is not a write-pattern problem; the sticker stream prefers 8192-16384"}
out["adi_repeats"] = {
    "what": "three repeats, buffers re-allocated: before (r03c) and after (r03d) the two-colour tables",
    "hash_lut_us": table(rows("r03c_ab.jsonl")[-30:], lambda r: r["k"], lambda r: f"rep{r['rep']}"),
    "two_colour_tables_us": table([r for r in rows("r03d_ab.jsonl") if r["k"].startswith("adi_")], lambda r: r["k"], lambda r: f"rep{r['rep']}"),
    "reading": "default (8 walks per lane x 3 segments) 143-148 us for codes (0.69-0.71 of 273 B/unit), 175-179 us with parent stickers (0.69-0.70 of 327 B/unit); "
               "round 2: 160.8 / 186.5 us.  The look-up rewrite cut the kernel's static VALU count from 3013 to 2205 (V1) and 5699 to 4159 (V2)"}
# 3. dense writers
out["dense_chunk_builder"] = {
    "what": "16-byte chunk of a 2-byte one-hot: eight compares + selects (round 2) against one 64-bit shift + two compares (r03a new)",
    "rows": [{"k": r["k"] + str(r.get("rep", "")), "shift_us": r["us"], "compare_us": a_old[r["k"] + str(r.get("rep", ""))]["us"]} for r in a_new
             if "dense" in r["k"] and r["k"] + str(r.get("rep", "")) in a_old],
    "reading": "no difference (bf16 191.3 vs 191.3 us): the dense writers are not VALU-bound (30 M VALU instructions over 1024 SIMDs are ~25 us of a 190 us launch)"}
out["dense_wide_first_sweeps"] = {
    "what": "code -> dense at 2^20 cubes: 960-thread workgroups, G groups with contiguous tile ranges (r03f, r03h), strided tiles (r03g), paced stores (r03i: s_sleep after "
            "every store), six separately allocated buffers x two processes (r03j)",
    "contiguous_us": table(rows("r03f_dense.jsonl") + rows("r03h_dense.jsonl"), lambda r: r["k"], lambda r: r["lib"]),
    "strided_us": table(rows("r03g_dense.jsonl"), lambda r: r["k"], lambda r: r["lib"]),
    "paced_f32_us": table([r for r in rows("r03i_dense.jsonl") if "f32" in r["k"]], lambda r: r["k"], lambda r: r["lib"]),
    "six_buffers_two_processes_us": table(rows("r03j_dense.jsonl"), lambda r: r["k"], lambda r: r["lib"]),
    "reading": "the result depends on the address stride between the concurrent streams, not on pacing; on every one of 12 buffers the wide contiguous form beat the "
               "256-thread form (bf16 159-182 against 173-194 us, f32 305-335 against 306-380 us)"}
out["dense_wide_sizes"] = {
    "what": "code -> dense, base (256-thread) against wide with 64 / 96 / 112 / 160 / 256 groups, 2^17 .. 2^22 cubes incl. sizes that are not powers of two (fraction of 8 TB/s)",
    "table_frac": table(rows("r03k_dense.jsonl"), lambda r: f"{r['k']}_n{r['n']}", lambda r: r["form"], "frac"),
    "reading": "~112 groups is the best or close to it everywhere and never below the base form: adopted for rc_onehot_from_code from 2^17 cubes"}
out["dense_wide_groups"] = {
    "what": "FUSED step + dense (wide form: every wave produces a tile, then all 960 threads sweep 15 tiles) and code -> dense over group counts, two buffers each; "
            "groups = -256 is the 256-thread form",
    "table_frac": table(rows("r03m_wide.jsonl"), lambda r: f"{r['k']}_buf{r['buf']}", lambda r: str(r["groups"]), "frac"),
    "reading": "fused: the wide form loses for every format and group count (bf16 0.62-0.74 against 0.74-0.76): ~110 workgroups funnel the state traffic and the dense "
               "stream stalls while they produce -- not shipped.  code -> dense: wide 112 wins on this box too (bf16 0.77-0.81 against 0.68-0.70)"}
out["adi_store_wave"] = {
    "what": "code-emitting ADI as a store-wave kernel (experiment build, parity-green): workgroup = 3 compute waves (one per depth segment, families handed over "
            "through 153 KB of LDS, one barrier per emitted depth) + 1 store wave issuing all code rows; variant 100 = store wave, 900 = the shipped waves per segment; "
            "walks x depth, us per launch (host clock around 3 launches)",
    "rows": {"512x3": [39.4, 16.3], "512x30": [104.6, 38.5], "5120x30": [106.9, 41.4], "20000x30": [106.8, 52.6], "100000x3": [29.1, 25.7], "100000x30": [166.7, 155.0]},
    "columns": ["store_wave_us", "shipped_us"],
    "reading": "slower at every size: ~200 storing waves with at most 63 stores of 512 B in flight each cap the stream at 4.9 TB/s, and small batches pay the "
               "lock-step barriers; not shipped"}
t = collections.OrderedDict()
for r in rows("r03q_skew.jsonl"):
    t.setdefault(f"{r['k']}_n{r['n']}_buf{r['buf']}", collections.OrderedDict())[f"g{r['groups']}_skew{r['skew']}"] = round(r["frac"], 3)
out["dense_wide_skew"] = {
    "what": "code -> dense, wide form: groups x sweep-start skew (group g starts g*skew tiles into its range and wraps); g-256 = the 256-thread form; fraction of 8 TB/s",
    "table_frac": t,
    "reading": "a fast box: 112 groups 0.89-0.91 (bf16, f32 at 2^20), skew neutral there; 128 groups (equal power-of-two ranges) 0.90-0.93 without and 0.93-0.95 with "
               "skew; 1.3M cubes and f32 at 2^21 cubes (4 GB) stay at 0.72-0.83 for every form"}
t = collections.OrderedDict()
for f, tag in (("r03t_aosoa.jsonl", "lds_transposed_16B_per_lane"), ("r03t_base.jsonl", "shipped_rows_8B_per_lane")):
    for r in rows(f):
        if "rep" in r:
            t.setdefault(r["k"], collections.OrderedDict()).setdefault(tag, []).append(round(r["us"], 1))
out["adi_code_store_shape"] = {
    "what": "code-emitting ADI, 100k x 30: the 20 code rows of every state transposed through a wave-private LDS block and stored as five 16-byte-per-lane instructions "
            "of 1 KiB contiguous each (timing experiment: the shape of a [n/4][20][4] code layout) against the shipped 20 row stores of 512 B; same box",
    "table_us": t,
    "reading": "the wide shape is 25-35 % SLOWER (190-200 us against 146-157 us): not adopted, the code layout stays [slot][pitch]",
    "also": "a cheap necessary pre-test for the child flags (some face entirely home) cut the dynamic VALU work by ~20 % and changed nothing "
            "(r03s: 149-155 us / 170 us): the launch is bound by its store stream, not by VALU"}
out["expand_streaming"] = {
    "what": "expansion of 2^20 parents to 12 children + flags: 2048 short-lived waves (k_expand) against the streaming form with 128 .. 1024 persistent waves "
            "(next group's rows prefetched under the current group's stores), three repeats",
    "table_us": table(rows("r03x_expand.jsonl"), lambda r: r["k"], lambda r: f"rep{r['rep']}"),
    "reading": "512 waves 113 us (0.83) against 116.4 us (0.80); 128 waves are too few (155 us), 384 do not divide the 2048 walk groups evenly (123 us)"}
out["dense_fused_wide_pipelined"] = {
    "what": "second attempt at a FUSED wide dense kernel (experiment build): every wave issues the 54 row loads of its next tile, all 960 threads sweep the current "
            "round's 15 tiles out of one LDS buffer, then every wave finishes its next tile into the other buffer (156 KB of LDS); group counts 80 .. 256 against "
            "the 256-thread form, three buffers each; outputs verified equal",
    "table_frac": table(rows("r03A_fused.jsonl"), lambda r: f"{r['k']}_buf{r['buf']}", lambda r: r["form"], "frac"),
    "reading": "bf16 0.63-0.70 against 0.72-0.73, u8 0.54-0.75 against 0.81-0.82, f32 0.77 against 0.74 (112 groups): not shipped; the fused launch keeps the "
               "256-thread form, whose ~1000 concurrent producers hide the state traffic better than ~110 wide workgroups can"}
out["dense_fused_contiguous_ranges"] = {
    "what": "the shipped 256-thread fused step + dense kernel with CONTIGUOUS tile ranges per workgroup and 128 .. 3072 workgroups (experiment build) against its "
            "shipped strided form (groups 0), three buffers each, outputs verified equal",
    "table_frac": table(rows("r03D_fc.jsonl"), lambda r: f"{r['k']}_buf{r['buf']}", lambda r: str(r["groups"]), "frac"),
    "reading": "no group count beats the shipped form (bf16 0.57-0.73 against 0.73-0.74, f32 0.60-0.69 against 0.70, u8 0.42-0.79 against 0.80-0.82)"}
out["dense_window_form"] = {
    "what": "code -> dense as a memset-like WINDOW: passes of 960 threads (15 KiB of output) dealt round-robin to G workgroups, code bytes read straight from global "
            "memory (experiment build); against the 256-thread form and the wide form (112 / 128 groups), three buffers each, on a session whose allocations were "
            "all of the slow kind (second skew sweep r03u: the same script, another session -- f32 at 2^20 0.71-0.73 where the first session had 0.91-0.94)",
    "table_frac": table(rows("r03v_win.jsonl"), lambda r: f"{r['k']}_n{r['n']}_buf{r['buf']}", lambda r: r["form"], "frac"),
    "second_skew_session_frac": table(rows("r03u_skew.jsonl"), lambda r: f"{r['k']}_n{r['n']}_buf{r['buf']}", lambda r: f"g{r['groups']}_skew{r['skew']}", "frac"),
    "reading": "on slow allocations every writer shape lands at 0.65-0.73 (window 256 groups 0.725 bf16, wide 112 0.70, 256-thread form 0.665): the spread between "
               "sessions is a property of where the output buffer lives, not of the kernel; the wide form is the best or within 3 % of it in both kinds of session"}
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_ab.json"), "w"), indent=1)
print("wrote profiles/r03_ab.json", os.path.getsize(os.path.join(ROOT, "profiles", "r03_ab.json")), "bytes")
