# One GPU call that produces every round file under profiles/ (run on the GPU box: gpurun -- 'bash tools/collect_round_profiles.sh r02').
# rocprofv3 gets the program itself after `--` (python3 file / a binary), counters in their own passes.
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_n13 $O/pmc_w_fetch $O/pmc_w_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-also > $O/prof_stats.log 2>&1
echo stats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-also > $O/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-also > $O/prof_write.log 2>&1
echo pmc done
python tools/summarize_profiles.py $TAG > $O/summarize.log 2>&1
cp profiles/${TAG}_summary.md profiles/${TAG}_pmc_traffic.json profiles/${TAG}_kernel_stats.csv $O/
python tools/bench_r02.py > $O/${TAG}_kernel_rooflines.json 2> $O/bench_r02.err
echo kernels done
python tools/site_kernel_ab.py > $O/${TAG}_site_kernel_ab.txt 2>&1
python tools/apply_norm_ab.py > $O/${TAG}_apply_norm_ab.txt 2>&1
bash tools/pmc_ab.sh tools/apply_norm_ab.py >> $O/${TAG}_apply_norm_ab.txt 2>&1
bash tools/pmc_wilson_direct.sh > $O/${TAG}_wilson_direct_pmc.txt 2>&1
python tools/wilson_direct_bench.py >> $O/${TAG}_wilson_direct_pmc.txt 2>&1
echo ab done
QMG_QUIET=1 rocprofv3 --kernel-trace --output-format csv -d $O/prof_n13 -- quantum-mg_amd/drivers/n13_wilson_kcycle 2048 -0.07 6.0 2 24 tests/golden/l64t64b60_heatbath.dat 64 > $O/${TAG}_n13_kcycle_2048_nc24.log 2>&1
python tools/solve_phase_profile.py $O/prof_n13 > $O/${TAG}_n13_solve_phase.json
rm -rf $O/prof_n13
echo n13 done
QMG_QUIET=1 rocprofv3 --kernel-trace --output-format csv -d $O/prof_n22 -- quantum-mg_amd/drivers/n22_wilson_kcycle_adaptive 4096 -0.07 6.0 3 1 tests/golden/l64t64b60_heatbath.dat 64 schur nrhs=1 f32 > $O/${TAG}_n22_c5_schur_f32.log 2>&1
python tools/solve_phase_profile.py $O/prof_n22 last > $O/${TAG}_n22_c5_schur_f32_solve_phase.json
rm -rf $O/prof_n22
echo n22 done
for v in "" f32; do
  rm -rf $O/prof_m
  QMG_QUIET=1 rocprofv3 --kernel-trace --output-format csv -d $O/prof_m -- quantum-mg_amd/drivers/n13_wilson_kcycle_mrhs 2048 -0.07 6.0 2 24 tests/golden/l64t64b60_heatbath.dat 64 8 $v > $O/${TAG}_n13_mrhs8${v:+_}$v.log 2>&1
  python tools/solve_phase_profile.py $O/prof_m > $O/${TAG}_n13_mrhs8${v:+_}${v}_solve_phase.json
  rm -rf $O/prof_m
done
echo mrhs done
python tools/xfer_bench.py > $O/${TAG}_xfer_mfma.txt 2>&1
python tools/kernelc_bench.py > $O/${TAG}_kernelC_multi_rhs.txt 2>&1
echo xfer done
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
echo bench done
python -c "
import json
d=json.load(open('$O/${TAG}_bench.json'))
print(d['value'], d['roofline']['frac'], d['roofline']['traffic'])
print({k:(v.get('value') if isinstance(v,dict) else None) for k,v in d.items() if k.startswith('also_kcycle')})
"
