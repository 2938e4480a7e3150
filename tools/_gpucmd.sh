cd $GRAFT_REPO_ROOT
python tools/kernelc_bench.py 24 8 16 12 > gpurun_out/r3_kc_b32pf.txt 2>&1
grep "k=1 " gpurun_out/r3_kc_b32pf.txt
timeout -k 10 900 python -m pytest tests/test_gpu_f32.py tests/test_gpu_parity.py tests/test_gpu_epilogue.py tests/test_gpu_batch.py -m gpu -x -q > gpurun_out/r3_t25.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3_t25.log
