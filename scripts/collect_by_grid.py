#!/usr/bin/env python3
"""Per-launch-shape summary of one kernel from a rocprofv3 output directory made by scripts/prof_r2.sh: the same kernel name covers
very different launches (a 4096-query and a 1 M-query batch, the 10 M- and the 100 M-point cloud), so durations and PMC counters
are grouped by (kernel, grid size).  usage: collect_by_grid.py <dir> <kernel substring> [...]  -> JSON on stdout"""
import collections, csv, glob, json, os, sys

out, pats = sys.argv[1], sys.argv[2:]
clean = lambda n: n.replace("(anonymous namespace)::", "").split("(")[0]
res = collections.defaultdict(dict)
for f in glob.glob(os.path.join(out, "trace", "*", "*_kernel_trace.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = clean(r["Kernel_Name"])
        if any(p in n for p in pats):
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            agg[(n, g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for (n, g), v in agg.items():
        res[f"{n} @grid={g}"].update({"launches": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3})
for d in ("pmc_fetch", "pmc_write", "pmc_l2"):
    for f in glob.glob(os.path.join(out, d, "*", "*_counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = clean(r["Kernel_Name"])
            if any(p in n for p in pats):
                agg[(n, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for (n, g), c in agg.items():
            for k, v in c.items():
                res[f"{n} @grid={g}"][k + "_mean_per_launch"] = sum(v) / len(v)
for k, v in res.items():
    if "FETCH_SIZE_mean_per_launch" in v and "WRITE_SIZE_mean_per_launch" in v:
        v["derived_hbm_traffic_bytes_per_launch"] = 2 * v["FETCH_SIZE_mean_per_launch"] * 1024 + v["WRITE_SIZE_mean_per_launch"] * 1024
        v["derived_traffic_GBs"] = v["derived_hbm_traffic_bytes_per_launch"] / (v["avg_us"] * 1e-6) / 1e9 if "avg_us" in v else None
    if "TCC_HIT_sum_mean_per_launch" in v:
        h, m = v["TCC_HIT_sum_mean_per_launch"], v["TCC_MISS_sum_mean_per_launch"]
        v["derived_l2_hit_rate"] = h / max(h + m, 1)
print(json.dumps(dict(sorted(res.items())), indent=1))
