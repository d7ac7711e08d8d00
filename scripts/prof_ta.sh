cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA_[A-Z_]+|TCP_[A-Z_]+)\b" | sort -u | head -80 > gpurun_out/counters_ta_tcp.txt
wc -l gpurun_out/counters_ta_tcp.txt
ARGS="--steps 3 --warmup 1 --cpu-queries 0 --stream-probe 0 --replan-probe 0"
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_ta/a -- python3 bench.py $ARGS > /dev/null 2>&1; echo rc=$?
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d gpurun_out/prof_ta/b -- python3 bench.py $ARGS > /dev/null 2>&1; echo rc=$?
python3 - <<'PY'
import csv,glob,collections
for d in ('a','b'):
    fs=glob.glob(f'gpurun_out/prof_ta/{d}/*/*_counter_collection.csv')
    if not fs: print('none',d); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r['Kernel_Name'][:45]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'nn_grid' in k: print(k,{c:round(sum(x)/len(x)) for c,x in v.items()})
PY
