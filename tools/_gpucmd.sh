cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t32.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/r3_t32.log
