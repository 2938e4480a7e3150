cd $GRAFT_REPO_ROOT
python tools/kernelc_bench.py 24 16 12 > gpurun_out/r3_kc_a.txt 2>&1
cp quantum-mg_amd/libqmg_hip.so /tmp/keep.so
cp quantum-mg_amd/libqmg_hip_pf2.so quantum-mg_amd/libqmg_hip.so
python tools/kernelc_bench.py 24 16 12 > gpurun_out/r3_kc_b.txt 2>&1
cp /tmp/keep.so quantum-mg_amd/libqmg_hip.so
paste -d'|' gpurun_out/r3_kc_a.txt gpurun_out/r3_kc_b.txt | grep "fp32 \|m16v32" | grep "k=8"
