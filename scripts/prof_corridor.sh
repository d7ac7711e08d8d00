cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d gpurun_out/prof_corr -- python3 scripts/probe_corridor.py > gpurun_out/prof_corr.log 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_corr/*/*_kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(f"{r['Name'][:50]:50s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.1f}")
for f in glob.glob('gpurun_out/prof_corr/*/*_hip_api_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(f"{r['Name'][:40]:40s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.1f}")
PY
