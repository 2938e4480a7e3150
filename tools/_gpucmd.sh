# scratch: the command of the builder's last ad-hoc GPU call (gpurun -- 'bash tools/_gpucmd.sh'); not part of the product or of the collection scripts
cd $GRAFT_REPO_ROOT
(cd quantum-mg_amd/drivers && ./facade_selftest ../../tests/golden/l32t32b60_heatbath.dat 2>&1 | tail -30) > gpurun_out/selftest.txt 2>&1
tail -5 gpurun_out/selftest.txt
timeout -k 10 900 python -m pytest tests/test_gpu_kcycle.py -x -q -k "right_jacobi or coarsest_cg or cgne" 2>&1 | tail -30
