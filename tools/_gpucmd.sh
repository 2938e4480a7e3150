cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_apply_norm.py -m gpu -x -q > gpurun_out/r3_t14.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3_t14.log
bash tools/profile_kcycles.sh r03 > gpurun_out/prof_kc.log 2>&1; tail -2 gpurun_out/prof_kc.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out
for v in "" f32; do
  rm -rf $O/prof_m
  QMG_QUIET=1 rocprofv3 --kernel-trace --output-format csv -d $O/prof_m -- quantum-mg_amd/drivers/n13_wilson_kcycle_mrhs 2048 -0.07 6.0 2 24 tests/golden/l64t64b60_heatbath.dat 64 8 $v > $O/r03_n13_mrhs8${v:+_}$v.log 2>&1
  python tools/solve_phase_profile.py $O/prof_m > $O/r03_n13_mrhs8${v:+_}${v}_solve_phase.json
  rm -rf $O/prof_m
done
echo done
