cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_f32.py tests/test_gpu_batch.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3_t21.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3_t21.log
