cd $GRAFT_REPO_ROOT
python tools/xfer_bench.py > gpurun_out/r3_xfer3.txt 2>&1
QMG_TUNING=xfer_mfma=1 timeout -k 10 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_f32.py -m gpu -x -q > gpurun_out/r3_t8.log 2>&1; echo "pytest(mfma on) rc=$?"; tail -3 gpurun_out/r3_t8.log
echo done
