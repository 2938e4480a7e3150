cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t7.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3_t7.log
python tools/xfer_bench.py > gpurun_out/r3_xfer2.txt 2>&1
F=$GRAFT_REPO_ROOT/tests/golden/l64t64b60_heatbath.dat
cd quantum-mg_amd/drivers
for i in 1 2; do echo "== C5 $i"; env QMG_QUIET=1 ./n22_wilson_kcycle_adaptive 4096 -0.07 6.0 3 1 $F 64 schur nrhs=1 f32 2>&1 | grep -E "converged|QMG-TIMING"; done > ../../gpurun_out/r3_kc6.txt 2>&1
echo "== C3 batched 8" >> ../../gpurun_out/r3_kc6.txt; env QMG_QUIET=1 ./n13_wilson_kcycle_mrhs 2048 -0.07 6.0 2 24 $F 64 8 2>&1 | grep -E "converged|QMG-TIMING|ERROR" >> ../../gpurun_out/r3_kc6.txt
echo done
