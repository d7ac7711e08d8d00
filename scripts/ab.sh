# same-box comparison of engine builds: every pointcloudtraj_amd/lib/ab/*.so is copied over the live library in turn (two rounds)
cd $GRAFT_REPO_ROOT
L=pointcloudtraj_amd/lib
cp $L/libpct_engine.so $L/ab/_live_backup
for round in 1 2; do
  for f in $L/ab/*.so; do
    v=$(basename $f .so)
    cp $f $L/libpct_engine.so
    if [ -n "$AB_STEP_ONLY" ]; then
      timeout -k 10 200 python scripts/ab_step.py > gpurun_out/ab_$v$round.log 2>&1; echo "$v$round $(tail -1 gpurun_out/ab_$v$round.log)"
    else
    timeout -k 10 200 python bench.py --cpu-queries 0 --replan-probe 0 --stream-probe 0 --steps 30 > gpurun_out/ab_$v$round.log 2>&1
    tail -1 gpurun_out/ab_$v$round.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v$round', 'value %.4e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'])"
    fi
  done
done
cp $L/ab/_live_backup $L/libpct_engine.so
