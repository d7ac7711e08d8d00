# rocprofv3 passes (kernel trace + separate PMC passes, as scripts/prof_bench.sh) over an arbitrary python command:
#   bash scripts/prof_any.sh TAG script.py [args...]   -> gpurun_out/prof_TAG/summary_*.{csv,json}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/run_trace.log 2>&1
echo trace_rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 "$@" > $OUT/run_pmc_fetch.log 2>&1
echo fetch_rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 "$@" > $OUT/run_pmc_write.log 2>&1
echo write_rc=$?
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 "$@" > $OUT/run_pmc_l2.log 2>&1
echo l2_rc=$?
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 "$@" > $OUT/run_pmc_sq.log 2>&1
echo sq_rc=$?
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq2 -- python3 "$@" > $OUT/run_pmc_sq2.log 2>&1
echo sq2_rc=$?
python3 scripts/collect_profiles.py $OUT $TAG
