F=$GRAFT_REPO_ROOT/tests/golden/l64t64b60_heatbath.dat
cd $GRAFT_REPO_ROOT/quantum-mg_amd/drivers
for e in "QMG_GCR_FUSED=1" "QMG_GCR_FUSED=0"; do
  for i in 1 2; do
    echo "== n13 C3 [$e] run $i"; env QMG_QUIET=1 $e ./n13_wilson_kcycle 2048 -0.07 6.0 2 24 $F 64 2>&1 | grep -E "converged|QMG-TIMING|ERROR|FATAL|OPS-STATS"
    echo "== n22 C5 schur [$e] run $i"; env QMG_QUIET=1 $e ./n22_wilson_kcycle_adaptive 4096 -0.07 6.0 3 1 $F 64 schur nrhs=1 f32 2>&1 | grep -E "converged|QMG-TIMING|ERROR|FATAL"
  done
  echo "== n13 mrhs 8 [$e]"; env QMG_QUIET=1 $e ./n13_wilson_kcycle_mrhs 2048 -0.07 6.0 2 24 $F 64 8 2>&1 | grep -E "QMG-TIMING|ERROR|FATAL"
done
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_kcycle.py tests/test_gpu_f32.py -m gpu -x -q > gpurun_out/r3_t12.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3_t12.log
echo done
