# rocprofv3 kernel traces of the two K-cycle configurations the bench quotes, summarised on the box (the traces are too big to carry back):
#   gpurun -- 'bash tools/profile_kcycles.sh r03'   ->  gpurun_out/<tag>_n13_solve_phase.json, gpurun_out/<tag>_n22_c5_schur_f32_solve_phase.json
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out
F=tests/golden/l64t64b60_heatbath.dat
rm -rf $O/prof_n13 $O/prof_n22
QMG_QUIET=1 rocprofv3 --kernel-trace --output-format csv -d $O/prof_n13 -- quantum-mg_amd/drivers/n13_wilson_kcycle 2048 -0.07 6.0 2 24 $F 64 > $O/${TAG}_n13_kcycle_2048_nc24.log 2>&1
python tools/solve_phase_profile.py $O/prof_n13 > $O/${TAG}_n13_solve_phase.json
rm -rf $O/prof_n13
echo n13 done
QMG_QUIET=1 rocprofv3 --kernel-trace --output-format csv -d $O/prof_n22 -- quantum-mg_amd/drivers/n22_wilson_kcycle_adaptive 4096 -0.07 6.0 3 1 $F 64 schur nrhs=1 f32 > $O/${TAG}_n22_c5_schur_f32.log 2>&1
python tools/solve_phase_profile.py $O/prof_n22 last > $O/${TAG}_n22_c5_schur_f32_solve_phase.json
rm -rf $O/prof_n22
echo n22 done
