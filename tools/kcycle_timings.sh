# The two K-cycle configurations the bench quotes, straight through the drivers (no profiler), each variant REPS times (the pool's boxes differ
# run to run): C3 (n13 2048^2 nc=24) and C5 in its red-black form (n22 4096^2, fp64 then the fp32 K-cycle).
#   gpurun -- 'bash tools/kcycle_timings.sh > gpurun_out/kc.txt'
F=$GRAFT_REPO_ROOT/tests/golden/l64t64b60_heatbath.dat
REPS=${REPS:-3}
cd $GRAFT_REPO_ROOT/quantum-mg_amd/drivers
for e in "" "QMG_APPLY_EPILOGUE=0" "QMG_KCYCLE_DEVICE_SCALARS=0" "QMG_KCYCLE_ENGINE=single" "QMG_COARSE_F32=0"; do
  for i in $(seq $REPS); do
    echo "== n13 C3 [$e] run $i"; env QMG_QUIET=1 $e ./n13_wilson_kcycle 2048 -0.07 6.0 2 24 $F 64 2>&1 | grep -E "converged|QMG-TIMING|ERROR|FATAL"
  done
done
for e in "" "QMG_APPLY_EPILOGUE=0" "QMG_KCYCLE_DEVICE_SCALARS=0" "QMG_KCYCLE_ENGINE=single"; do
  for i in $(seq $REPS); do
    echo "== n22 C5 schur [$e] run $i"; env QMG_QUIET=1 $e ./n22_wilson_kcycle_adaptive 4096 -0.07 6.0 3 1 $F 64 schur nrhs=1 f32 2>&1 | grep -E "converged|QMG-TIMING|ERROR|FATAL"
  done
done
