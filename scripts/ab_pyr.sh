# 10 M pillar-surface cloud through the pyramid kernel with each engine build given: bash scripts/ab_pyr.sh so1 so2 ...
for rep in 1 2; do
  for so in "$@"; do
    echo -n "$so "; PCT_ENGINE_SO=$so python scripts/probe_pyr.py pillar10m 2>/dev/null | grep -o "queries .*kernel [0-9.]* ms"
  done
done
