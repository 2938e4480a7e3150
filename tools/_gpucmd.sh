# scratch: the command of the builder's last ad-hoc GPU call (gpurun -- 'bash tools/_gpucmd.sh'); not part of the product or of the collection scripts
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_slab.py tests/test_gpu_kcycle.py -x -v -k "default_engine or batched_right_jacobi" > gpurun_out/t.log 2>&1; echo rc $?; tail -14 gpurun_out/t.log
