cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t35.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r3_t35.log
if [ $rc -eq 0 ]; then bash tools/collect_round_profiles.sh r03 > gpurun_out/collect_r03.log 2>&1; tail -4 gpurun_out/collect_r03.log; fi
