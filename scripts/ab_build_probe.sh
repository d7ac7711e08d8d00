# per-frame index rebuild (bench.py index_build_probe, 10 M points) under settings of ONE environment variable: bash scripts/ab_build_probe.sh VAR v1 v2 ...
VAR=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    env $VAR=$v python bench.py --steps 5 --warmup 2 --replan-probe 0 --c4-probe 0 --clustered-probe 0 --cpu-queries 0 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['index_build_probe']; print('$VAR=$v', 'build ms_median %.4f points/s %.3e' % (p['ms_median'], p['points_per_s']))"
  done
done
