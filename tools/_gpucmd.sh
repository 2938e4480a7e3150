cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_slab.py tests/test_gpu_u1.py tests/test_gpu_wilson_direct.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r3_t24.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/r3_t24.log
