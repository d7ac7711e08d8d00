# rocprofv3 passes over the default bench.py run; summaries are copied to profiles/ by scripts/collect_profiles.py
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
ARGS="--steps 10 --warmup 2 --cpu-queries 0 --stream-probe 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
echo trace_rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_pmc_fetch.log 2>&1
echo fetch_rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_pmc_write.log 2>&1
echo write_rc=$?
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 bench.py $ARGS > $OUT/bench_pmc_l2.log 2>&1
echo l2_rc=$?
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_pmc_sq.log 2>&1
echo sq_rc=$?
python3 scripts/collect_profiles.py $OUT $TAG
