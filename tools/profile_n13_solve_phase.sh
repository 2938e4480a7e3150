set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $R
rm -rf gpurun_out/prof_n13
QMG_QUIET=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_n13 -- quantum-mg_amd/drivers/n13_wilson_kcycle 2048 -0.07 6.0 2 24 tests/golden/l64t64b60_heatbath.dat 64 > gpurun_out/r02_n13_kcycle_2048_nc24.log 2>&1
python tools/solve_phase_profile.py gpurun_out/prof_n13 > gpurun_out/r02_n13_solve_phase.json
rm -rf gpurun_out/prof_n13
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_n13_solve_phase.json'))
for ph in ('setup','solve'):
    s=d[ph]; print(ph, round(s['wall_ms']), round(s['gpu_busy_ms']), round(s['idle_frac'],3), s['launches'])
    for k in s['kernels'][:12]: print('   ', k['kernel'][:70], k['calls'], round(k['total_ms'],1), round(k['pct_of_wall'],1))
PY
