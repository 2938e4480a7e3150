cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_slab.py tests/test_gpu_kcycle.py -m gpu -x -q > gpurun_out/r3_t34.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3_t34.log
cd quantum-mg_amd/drivers
for i in 1 2; do env QMG_QUIET=1 ./slab_wilson_solve 4096 0.05 6.0 200 1337 1e-10 1 1 2>&1 | grep -E "BiCGStab" ; done
