# scratch: the command of the builder's last ad-hoc GPU call (gpurun -- 'bash tools/_gpucmd.sh'); not part of the product or of the collection scripts
cd $GRAFT_REPO_ROOT/quantum-mg_amd/drivers
timeout -k 5 200 ./facade_selftest ../../tests/golden/l32t32b60_heatbath.dat > ../../gpurun_out/selftest.txt 2>&1; echo rc $?; grep -i "bcg_core\|FAIL\|SELFTEST" ../../gpurun_out/selftest.txt
