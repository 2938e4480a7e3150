# scratch: the command of the builder's last ad-hoc GPU call (gpurun -- 'bash tools/_gpucmd.sh'); not part of the product or of the collection scripts
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -3
cd quantum-mg_amd/drivers
G=../../tests/golden/l64t64b60_heatbath.dat
for i in 1 2; do
  echo "-- new C3"; QMG_QUIET=1 timeout -k 10 300 ./n13_wilson_kcycle 2048 -0.07 6.0 2 24 $G 64 2>&1 | grep -E "^\[QMG-TIMING\]"
  echo "-- head C3"; LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_ab:$LD_LIBRARY_PATH QMG_QUIET=1 timeout -k 10 300 ./n13_wilson_kcycle 2048 -0.07 6.0 2 24 $G 64 2>&1 | grep -E "^\[QMG-TIMING\]"
done
for i in 1 2; do
  echo "-- new C5"; QMG_QUIET=1 timeout -k 10 300 ./n22_wilson_kcycle_adaptive 4096 -0.07 6.0 3 1 $G 64 schur nrhs=1 f32 2>&1 | grep -E "TIMING\]|MRHS\]: (solve|.*iterations/s)"
  echo "-- head C5"; LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_ab:$LD_LIBRARY_PATH QMG_QUIET=1 timeout -k 10 300 ./n22_wilson_kcycle_adaptive 4096 -0.07 6.0 3 1 $G 64 schur nrhs=1 f32 2>&1 | grep -E "TIMING\]|MRHS\]: (solve|.*iterations/s)"
done
