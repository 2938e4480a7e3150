cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kcycle.py -m gpu -x -q > gpurun_out/r3_t26.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3_t26.log
