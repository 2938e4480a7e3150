# scratch: the command of the builder's last ad-hoc GPU call (gpurun -- 'bash tools/_gpucmd.sh'); not part of the product or of the collection scripts
cd $GRAFT_REPO_ROOT
timeout -k 10 1150 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
