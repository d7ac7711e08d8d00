# headline step under settings of ONE environment variable: bash scripts/ab_env.sh VAR v1 v2 ...   (each value twice, alternating)
VAR=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    env $VAR=$v python bench.py --steps 100 --warmup 50 --stream-probe 0 --replan-probe 0 --c4-probe 0 --clustered-probe 0 --cpu-queries 0 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', 'ms_per_step %.4f kernel_ms %.4f q/s %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
  done
done
