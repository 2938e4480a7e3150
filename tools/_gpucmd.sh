# scratch: the command of the builder's last ad-hoc GPU call (gpurun -- 'bash tools/_gpucmd.sh'); not part of the product or of the collection scripts
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/full.log 2>&1; echo rc $?; tail -5 gpurun_out/full.log
