cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t19.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/r3_t19.log
if [ $rc -eq 0 ]; then python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; tail -c 600 gpurun_out/r03_bench.json; fi
