# scratch: the command of the builder's last ad-hoc GPU call (gpurun -- 'bash tools/_gpucmd.sh'); not part of the product or of the collection scripts
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo rc $?; tail -3 gpurun_out/smoke.log
