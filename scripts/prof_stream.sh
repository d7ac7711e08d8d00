cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
python3 scripts/probe.py grid 2>&1 | grep -E "Q= 1048576|Q=   65536|SORTED" | grep -E "ppc= 1.0 shift=1|ppc= 2.0 shift=1|ppc= 2.0 shift=0|SORTED"
