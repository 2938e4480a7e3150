set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $R
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-also > gpurun_out/prof_stats.log 2>&1
echo stats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-also > gpurun_out/prof_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-also > gpurun_out/prof_write.log 2>&1
echo write done
python tools/summarize_profiles.py r02 > gpurun_out/summarize.log 2>&1
cp profiles/r02_summary.md profiles/r02_pmc_traffic.json profiles/r02_kernel_stats.csv gpurun_out/ 
echo summarized
python tools/bench_r02.py > gpurun_out/r02_kernel_rooflines.json 2> gpurun_out/bench_r02.err
echo kernels done
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
echo bench done
tail -c 600 gpurun_out/r02_bench.json
