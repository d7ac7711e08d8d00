# headline step (10 M uniform points, 1 M queries, 4 batches in rotation) with each engine build given: bash scripts/ab_headline.sh so1 so2 ...
for so in "$@"; do
  for rep in 1 2; do
    PCT_ENGINE_SO=$so python bench.py --steps 100 --warmup 50 --stream-probe 0 --replan-probe 0 --c4-probe 0 --clustered-probe 0 --cpu-queries 0 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$so', 'ms_per_step %.4f kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done
