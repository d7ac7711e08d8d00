cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ck && mkdir -p gpurun_out/prof_ck
rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d gpurun_out/prof_ck -- python3 scripts/probe_corridor_k.py 256 > gpurun_out/prof_ck/run.log 2>&1
tail -3 gpurun_out/prof_ck/run.log
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_ck/*/*_kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(f"{r['Name'][:50]:50s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
for f in glob.glob('gpurun_out/prof_ck/*/*_hip_api_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(f"{r['Name'][:40]:40s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
