# scratch: the command of the builder's last ad-hoc GPU call (gpurun -- 'bash tools/_gpucmd.sh'); not part of the product or of the collection scripts
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_f32.py tests/test_gpu_kcycle.py -x -q -k "narrow_null or n13 or engines or storage" > gpurun_out/t.log 2>&1; echo rc $?; tail -5 gpurun_out/t.log
cd quantum-mg_amd/drivers
G=../../tests/golden/l64t64b60_heatbath.dat
for i in 1 2; do
  for b in 32 64; do
    echo "-- C3 bits=$b"; QMG_COARSE_BITS=$b QMG_QUIET=1 timeout -k 10 300 ./n13_wilson_kcycle 2048 -0.07 6.0 2 24 $G 64 2>&1 | grep -E "converged|^\[QMG-TIMING\]" | cut -c1-110
  done
done
for i in 1 2; do
  echo "-- C5"; QMG_QUIET=1 timeout -k 10 300 ./n22_wilson_kcycle_adaptive 4096 -0.07 6.0 3 1 $G 64 schur nrhs=1 f32 2>&1 | grep -E "converged|TIMING\]" | cut -c1-150
done
