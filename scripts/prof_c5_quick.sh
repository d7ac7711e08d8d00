# kernel-trace of the C5 probe (one rocprofv3 pass): bash scripts/prof_c5_quick.sh <tag> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pc5_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/probe_c5.py 100 > $OUT/run.log 2>&1
grep -E "ingest_ms_p50|ms_per_tick_p50" $OUT/run.log
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-60s calls %-6s avg %.1f us  min %.1f  max %.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
