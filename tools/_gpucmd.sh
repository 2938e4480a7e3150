cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_f32.py tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_kcycle.py -m gpu -x -q > gpurun_out/r3_t33.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3_t33.log
python tools/xfer_bench.py > gpurun_out/r3_xfer8.txt 2>&1; grep "mfma=1" gpurun_out/r3_xfer8.txt | grep "prolong" | grep "k=8"
python tools/kernelc_bench.py 24 8 > gpurun_out/r3_kc_raw3.txt 2>&1; grep "k=16" gpurun_out/r3_kc_raw3.txt | grep -v "mat16"
