"""Summarise rocprofv3 CSV output (kernel trace stats + PMC passes) into small text/JSON files."""
import csv, glob, json, os, sys
from collections import defaultdict

raw, out = sys.argv[1], sys.argv[2]

import re

def short(n, templ=False):
    """k_name, or (templ) k_name<template arguments> - the instantiations of one kernel are different programs (k_render_bwd whole
    tiles / long-tile segments, k_fwd_long_seg<0|1>): the text summaries keep them apart."""
    m = re.search(r"(k_[a-z_0-9]+)(<[^()]*>)?", n)
    if m:
        return m.group(1) + ((m.group(2) or "").replace(" ", "").replace("(anonymousnamespace)::", "") if templ else "")
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = n.split("(")[0].split("<")[0]
    return n.split("::")[-1].strip() or "?"

# kernel stats
for f in glob.glob(os.path.join(raw, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, "kernel_stats.txt"), "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats summary (times in ns)\n")
        o.write(f"{'kernel':60s} {'calls':>7s} {'total_ns':>14s} {'avg_ns':>12s} {'pct':>7s} {'min_ns':>10s} {'max_ns':>10s}\n")
        for r in rows:
            o.write(f"{short(r['Name'], True)[:60]:60s} {r['Calls']:>7s} {r['TotalDurationNs']:>14s} {float(r['AverageNs']):12.1f} "
                    f"{float(r['Percentage']):7.2f} {r['MinNs']:>10s} {r['MaxNs']:>10s}\n")
    print(open(os.path.join(out, "kernel_stats.txt")).read())

# timeline of one step: gaps between consecutive kernels (from the kernel trace)
for f in glob.glob(os.path.join(raw, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # find the last occurrence of k_preprocess .. next k_preprocess
    idx = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) == "k_preprocess"]
    if len(idx) >= 3:
        a, b = idx[-3], idx[-2]
        with open(os.path.join(out, "step_timeline.txt"), "w") as o:
            o.write("# one steady-state step: kernel, start offset (us), duration (us), gap to previous kernel end (us)\n")
            t0 = int(rows[a]["Start_Timestamp"]); prev_end = None
            for r in rows[a:b + 1]:
                st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                gap = 0.0 if prev_end is None else (st - prev_end) / 1e3
                o.write(f"{short(r['Kernel_Name'], True)[:40]:40s} {(st - t0) / 1e3:10.1f} {(en - st) / 1e3:10.1f} {gap:8.1f}\n")
                prev_end = en
        print(open(os.path.join(out, "step_timeline.txt")).read())

# PMC passes
summary = {}
for p in sorted(glob.glob(os.path.join(raw, "pmc*"))):
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(lambda: defaultdict(int))
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"], True)
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
        for k in acc:
            summary.setdefault(k, {})
            for c in acc[k]:
                summary[k][c] = {"avg_per_launch": acc[k][c] / cnt[k][c], "launches": cnt[k][c]}
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
with open(os.path.join(out, "pmc_summary.txt"), "w") as o:
    for k in summary:
        if not k.startswith("k_"):
            continue
        o.write(k + "\n")
        for c, v in sorted(summary[k].items()):
            o.write(f"    {c:28s} {v['avg_per_launch']:18.1f}  (n={v['launches']})\n")
print(open(os.path.join(out, "pmc_summary.txt")).read())

# compact per-kernel counters for bench.py: HBM-side traffic (bytes per launch) and the vector-ALU picture
# (tools/merge_counters.py copies them into profiles/traffic.json and profiles/valu.json under a config name)
# (per kernel NAME: the instantiations of a name that run in one step - a dense one-view launch runs k_render_bwd twice, whole tiles
# and long-tile segments - are summed, weighted by how often each ran relative to the most frequent one)
def _avg(k, c):
    insts = [(n, v[c]) for n, v in summary.items() if n.split("<")[0] == k and c in v]
    if not insts:
        return None
    most = max(v["launches"] for _, v in insts)
    return sum(v["avg_per_launch"] * v["launches"] / most for _, v in insts)
counters = {"traffic": {}, "valu": {}}
for k in sorted({n.split("<")[0] for n in summary}):
    if not k.startswith("k_"):
        continue
    f, w = _avg(k, "FETCH_SIZE"), _avg(k, "WRITE_SIZE")
    if f is not None and w is not None:
        counters["traffic"][k] = int((2 * f + w) * 1024)        # FETCH_SIZE doubled: MI355X_MICROARCH.md, HBM section
    insts, act, gui = _avg(k, "SQ_INSTS_VALU"), _avg(k, "SQ_ACTIVE_INST_VALU"), _avg(k, "GRBM_GUI_ACTIVE")
    if insts is not None and act is not None and gui:
        cycles = gui / 8.0                                      # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        counters["valu"][k] = {"SQ_INSTS_VALU": int(insts), "SQ_ACTIVE_INST_VALU_quad_cycles": int(act),
                               "kernel_cycles": int(cycles), "valu_busy": round(act * 4.0 / (1024 * cycles), 4),
                               "valu_cycles_per_inst": round(act * 4.0 / insts, 2),
                               "SQ_INSTS_SALU": int(_avg(k, "SQ_INSTS_SALU") or 0), "SQ_INSTS_LDS": int(_avg(k, "SQ_INSTS_LDS") or 0),
                               "SQ_WAVES": int(_avg(k, "SQ_WAVES") or 0)}
json.dump(counters, open(os.path.join(out, "counters.json"), "w"), indent=1)
