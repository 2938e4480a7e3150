cd quantum-mg_amd/drivers
G=../../tests/golden
run() { name="$1"; shift; QMG_QUIET=1 timeout -k 5 100 "$@" > ../../gpurun_out/sweep_tmp.log 2>&1; rc=$?; echo "$name rc=$rc $(grep -E 'Multigrid|Check tol|worst relative|ERROR|WARNING|QMG-SLAB\]: (BiCG|world)|measurements,|MRHS\]: rhs 0' ../../gpurun_out/sweep_tmp.log | tr '\n' ' ' | cut -c1-260)"; }
run n13_32_1_4 ./n13_wilson_kcycle 32 -0.03 6.0 1 4 $G/l32t32b60_heatbath.dat 32
run n13_64_1_16 ./n13_wilson_kcycle 64 -0.07 6.0 1 16 $G/l64t64b60_heatbath.dat 64
run n13_128_1_12 ./n13_wilson_kcycle 128 -0.07 6.0 1 12 $G/l128t128b60_heatbath.dat 128
run n13_256_2_32 ./n13_wilson_kcycle 256 -0.07 6.0 2 32 $G/l64t64b60_heatbath.dat 64
run n13_512_3_8 ./n13_wilson_kcycle 512 -0.07 6.0 3 8 $G/l64t64b60_heatbath.dat 64
run n19_64_2 ./n19_wilson_kcycle_precond 64 2 $G/l64t64b60_heatbath.dat 64
run n19_256_3 ./n19_wilson_kcycle_precond 256 3 $G/l64t64b60_heatbath.dat 64 nrhs=5
run n22_64_2_2 ./n22_wilson_kcycle_adaptive 64 -0.07 6.0 2 2 $G/l64t64b60_heatbath.dat 64
run n22_512_3_2 ./n22_wilson_kcycle_adaptive 512 -0.07 6.0 3 2 $G/l64t64b60_heatbath.dat 64 nrhs=4
run n22_256_2_1_schur ./n22_wilson_kcycle_adaptive 256 -0.07 6.0 2 1 $G/l64t64b60_heatbath.dat 64 schur
run mrhs_256_16 ./n13_wilson_kcycle_mrhs 256 -0.07 6.0 2 16 $G/l64t64b60_heatbath.dat 64 16 verify
run mrhs_512_nc12_7 ./n13_wilson_kcycle_mrhs 512 -0.07 6.0 2 12 $G/l64t64b60_heatbath.dat 64 7 verify
run mrhs_256_nc6_5 ./n13_wilson_kcycle_mrhs 256 -0.07 6.0 2 6 $G/l64t64b60_heatbath.dat 64 5 verify
run mrhs_f32 env QMG_COARSE_F32=1 ./n13_wilson_kcycle_mrhs 512 -0.07 6.0 2 24 $G/l64t64b60_heatbath.dat 64 4 verify
# ---- round 2: fp32 / 16-bit K-cycle, the stored-stencil path, and everything again on y-slabs (ranks emulated by host threads)
run mrhs_f32_kcycle ./n13_wilson_kcycle_mrhs 512 -0.07 6.0 2 24 $G/l64t64b60_heatbath.dat 64 4 verify f32
run mrhs_f32_kcycle_h16 env QMG_F16_FINE=1 ./n13_wilson_kcycle_mrhs 512 -0.07 6.0 2 24 $G/l64t64b60_heatbath.dat 64 4 f32
run n13_stored_stencil env QMG_WILSON_DIRECT=0 ./n13_wilson_kcycle 256 -0.07 6.0 2 8 $G/l64t64b60_heatbath.dat 64
run n13_sequential_nullvecs env QMG_NULL_BATCH=1 ./n13_wilson_kcycle 256 -0.07 6.0 2 8 $G/l64t64b60_heatbath.dat 64
run n22_1024_3_schur_f32 ./n22_wilson_kcycle_adaptive 1024 -0.07 6.0 3 1 $G/l64t64b60_heatbath.dat 64 schur f32
run slab_bicgstab_r4 env QMG_COMM_EMULATE=4 ./slab_wilson_solve 256 0.05 6.0 100 7
run slab_n13_r1 ./n13_wilson_kcycle_slab 512 -0.07 6.0 2 8 $G/l64t64b60_heatbath.dat 64
run slab_n13_r4 env QMG_COMM_EMULATE=4 ./n13_wilson_kcycle_slab 512 -0.07 6.0 2 8 $G/l64t64b60_heatbath.dat 64
run slab_n13_r8_nc24 env QMG_COMM_EMULATE=8 ./n13_wilson_kcycle_slab 512 -0.07 6.0 2 24 $G/l64t64b60_heatbath.dat 64
run slab_n13_r4_batched_f32 env QMG_COMM_EMULATE=4 ./n13_wilson_kcycle_slab 256 -0.07 6.0 2 8 $G/l64t64b60_heatbath.dat 64 nrhs=4 f32
run slab_n19_r4 env QMG_COMM_EMULATE=4 ./n19_wilson_kcycle_precond 256 2 $G/l64t64b60_heatbath.dat 64 nrhs=3
run n15_pion ./n15_wilson_goldstone_u1_heatbath 32 0.01 6.0 50 100 500 1337
run n20_pion ./n20_staggered_goldstone_u1_heatbath 32 0.1 6.0 50 100 500 1337
