# same-box A/B of two engine builds: pointcloudtraj_amd/lib/ab/{old,new}.so are copied over the live library in turn
cd $GRAFT_REPO_ROOT
L=pointcloudtraj_amd/lib
for round in 1 2; do
  for v in old new; do
    cp $L/ab/$v.so $L/libpct_engine.so
    timeout -k 10 200 python bench.py --cpu-queries 0 --replan-probe 0 --stream-probe 0 --steps 30 > gpurun_out/ab_$v$round.log 2>&1
    tail -1 gpurun_out/ab_$v$round.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v$round', 'value %.4e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'])"
  done
done
cp $L/ab/new.so $L/libpct_engine.so
