# kernel-trace only of the index build probe (one rocprofv3 pass): bash scripts/prof_build_quick.sh <tag> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pbq_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/probe_build.py 10000000 100000000 > $OUT/run.log 2>&1
grep build_grid $OUT/run.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/*/*_kernel_trace.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if "gb_" in k or "bbox" in k or "cell_" in k:
        agg[(k, r.get("Grid_Size") or r.get("Grid_Size_X"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    print("%-40s grid %-10s n=%d  mean %.1f us  min %.1f" % (k[0], k[1], len(v), sum(v) / len(v), min(v)))
PY
