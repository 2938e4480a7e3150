cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t11.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3_t11.log
