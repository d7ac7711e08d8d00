# same-box comparison of environment switches of ONE build: scripts/ab_env.sh "NAME=VAL ..." "NAME=VAL ..." ...  (each argument one
# variant; "-" = defaults), two rounds, wall-clock ms per 1M-query step (scripts/ab_step.py)
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
    echo "variant$i round$round [$v] $(env $envs timeout -k 10 200 python scripts/ab_step.py 2>&1 | tail -1)"
  done
done
