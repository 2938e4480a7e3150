cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_f32.py -m gpu -x -q -k "16_bit_fine" > gpurun_out/r3_t22.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r3_t22.log
