# radius-count batch through the index (bench.py's radius_count_probe) with each engine build given: bash scripts/ab_count.sh so1 so2 ...
for rep in 1 2; do
  for so in "$@"; do
    PCT_ENGINE_SO=$so python bench.py --steps 5 --warmup 2 --replan-probe 0 --c4-probe 0 --clustered-probe 0 --cpu-queries 0 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['radius_count_probe']; print('$so', 'ms_per_batch %.4f counts/s %.3e mean_count %.2f' % (p['ms_per_batch'], p['queries_per_s'], p['mean_count']))"
  done
done
