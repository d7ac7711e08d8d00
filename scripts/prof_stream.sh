cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "binning or golden or full_size" > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
python3 scripts/probe.py grid 2>&1 | grep -E "Q= 1048576|work|SORTED"
