cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1/trace -- python3 bench.py --steps 10 --warmup 2 --cpu-queries 0 > gpurun_out/prof_r1/bench_trace.log 2>&1
echo trace_rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_r1/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-queries 0 > gpurun_out/prof_r1/bench_pmc1.log 2>&1
echo pmc1_rc=$?
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/prof_r1/pmc_l2 -- python3 bench.py --steps 3 --warmup 1 --cpu-queries 0 > gpurun_out/prof_r1/bench_pmc2.log 2>&1
echo pmc2_rc=$?
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/prof_r1/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --cpu-queries 0 > gpurun_out/prof_r1/bench_pmc3.log 2>&1
echo pmc3_rc=$?
find gpurun_out/prof_r1 -name "*.csv" | head -30
