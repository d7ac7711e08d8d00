cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stream -- python3 scripts/debug_filter.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d gpurun_out/prof_stream_pmc -- python3 scripts/debug_filter.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_stream_pmc2 -- python3 scripts/debug_filter.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/prof_stream/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>3s} avg_us={float(r['AverageNs'])/1e3:10.1f}")
for d in ('prof_stream_pmc','prof_stream_pmc2'):
    fs=glob.glob(f'gpurun_out/{d}/*/*_counter_collection.csv')
    if not fs: print('no pmc', d); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'nn_tile' in k: print(k,{c:round(sum(x)/len(x)) for c,x in v.items()})
PY
