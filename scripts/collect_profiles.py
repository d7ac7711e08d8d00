#!/usr/bin/env python3
"""Condense rocprofv3 output directories (scripts/prof_bench.sh) into small summaries:
   <out>/summary_kernel_stats.csv   per-kernel calls / total / average / min / max (ns)
   <out>/summary_pmc.json           per-kernel mean counter values per launch + derived HBM traffic
HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so reads are
doubled; separate --pmc passes.  traffic_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024."""
import collections
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
stats = glob.glob(os.path.join(out, "trace", "*", "*_kernel_stats.csv"))
rows = []
if stats:
    for r in csv.DictReader(open(stats[0])):
        rows.append([r["Name"].replace("(anonymous namespace)::", "").split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    with open(os.path.join(out, "summary_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "average_ns", "percent", "min_ns", "max_ns"])
        w.writerows(rows)
pmc = {}
for d in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq", "pmc_sq2"):
    fs = glob.glob(os.path.join(out, d, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        # the same kernel name covers very different launches (a 4096-query and a 1 M-query batch): one entry per (kernel, grid size)
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        agg[f'{name} @grid={r.get("Grid_Size", "?")}'][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "pct::" in k:
            pmc.setdefault(k, {}).update({c: {"mean_per_launch": sum(x) / len(x), "launches": len(x)} for c, x in v.items()})
for k, v in pmc.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["derived_hbm_traffic_bytes_per_launch"] = 2 * v["FETCH_SIZE"]["mean_per_launch"] * 1024 + v["WRITE_SIZE"]["mean_per_launch"] * 1024
    if "TCC_HIT_sum" in v:
        h, m = v["TCC_HIT_sum"]["mean_per_launch"], v["TCC_MISS_sum"]["mean_per_launch"]
        v["derived_l2_hit_rate"] = h / max(h + m, 1)
json.dump(pmc, open(os.path.join(out, "summary_pmc.json"), "w"), indent=1)
for r in rows[:12]:
    print(f"{r[0][:60]:60s} calls={r[1]:>4s} avg_us={float(r[3])/1e3:10.1f}")
for k, v in pmc.items():
    if "grid" in k or "tile" in k or "nn_stream" in k:
        print(k, {c: (round(x["mean_per_launch"]) if isinstance(x, dict) else x) for c, x in v.items()})
