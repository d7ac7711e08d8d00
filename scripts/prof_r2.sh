# rocprofv3 passes of round 2 (program directly after `--`); summaries are condensed by scripts/collect_profiles.py into
# gpurun_out/prof_<tag>/summary_* and copied to profiles/ by hand.   usage: bash scripts/prof_r2.sh <what> [tag]
#   what = bench | bench_sorted | c4 | c5 | brute | build
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
WHAT=${1:-bench}
TAG=${2:-r02_$WHAT}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
case $WHAT in
  bench)        CMD="bench.py --steps 10 --warmup 2 --cpu-queries 0 --stream-probe 0 --replan-probe 0 --c4-probe 0" ;;
  bench_sorted) export PCT_SORTED_WRITES=1; CMD="bench.py --steps 10 --warmup 2 --cpu-queries 0 --stream-probe 0 --replan-probe 0 --c4-probe 0" ;;
  c4)           CMD="scripts/probe_c4.py" ;;
  c5)           CMD="scripts/probe_c5.py 100" ;;
  brute)        CMD="scripts/probe_brute.py" ;;
  build)        CMD="scripts/probe_build.py 10000000 100000000" ;;
esac
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $CMD > $OUT/run_trace.log 2>&1
echo trace_rc=$?
if [ "$WHAT" != "c5" ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $CMD > $OUT/run_pmc_fetch.log 2>&1
echo fetch_rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $CMD > $OUT/run_pmc_write.log 2>&1
echo write_rc=$?
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $CMD > $OUT/run_pmc_l2.log 2>&1
echo l2_rc=$?
fi
if [ "$WHAT" = "bench" ] || [ "$WHAT" = "brute" ]; then
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $CMD > $OUT/run_pmc_sq.log 2>&1
echo sq_rc=$?
fi
python3 scripts/collect_profiles.py $OUT $TAG
tail -2 $OUT/run_trace.log
