"""Every BASELINE config at the size BASELINE.json names, on the one GPU of the test box.

At these sizes the CPU oracle cannot be run whole, so the checks are the size-independent ones: true
residuals of the ORIGINAL system, the periodic-image gate (a 64-periodic right-hand side on the 64-periodic tiled gauge
field gives the 64-periodic image of the oracle's 64^2 result), batched = single iteration counts, per-RHS norms against
numpy.  The small-size oracle comparisons of the same code paths are in test_gpu_parity / test_gpu_kcycle / test_gpu_f32.
"""
import ctypes as C
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
qmg = importlib.import_module("quantum-mg_amd")
DRIVERS = os.path.join(ROOT, "quantum-mg_amd", "drivers")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "quantum-mg_amd"), "-j4", "libqmg_hip.so"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", DRIVERS, "-j4"], stdout=subprocess.DEVNULL)
    qmg.init(0)


def test_c2_wilson_apply_2048_and_4096_fp64_and_fp32(golden_dir):
    """configs[1] (2048^2) and the north-star size (4096^2): periodic-image gate of the fp64 apply (1e-13) and of the fp32
    instantiation (5e-6) -- bench.py's own gates, run here so that a green -m gpu covers them."""
    import bench
    fixture = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    for L in (2048, 4096):
        wl = bench.Workload(qmg, L, fixture, 1337)
        assert wl.parity_gate(fixture) < 1e-13
        wl.free()
    barrier = lambda: qmg.sync()
    out = bench.f32_fine_apply(qmg, 4096, fixture, 3, 1, barrier)
    assert out["parity_gate_rel_l2_vs_fp64_oracle"] < 5e-6


def test_c3_wilson_kcycle_2048_three_levels_nc24(golden_dir):
    """configs[2]: n13 K-cycle, 2048^2 -> 512^2 -> 128^2, coarse nc = 24.  Four systems in lock step; system 0 is then
    re-solved alone by the single-vector path: converged, every true residual <= 1e-10, iterations batched = single +-1,
    solutions equal to solver accuracy."""
    fixture = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    out = subprocess.run([os.path.join(DRIVERS, "n13_wilson_kcycle_mrhs"), "2048", "-0.07", "6.0", "2", "24", fixture, "64", "4", "verify0"], cwd=DRIVERS,
                         env=dict(os.environ, QMG_QUIET="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "[QMG-ERROR]" not in out.stdout and "[QMG-WARNING]" not in out.stdout
    rows = re.findall(r"\[QMG-MRHS\]: rhs (\d+) converged in (\d+) iterations ; alleged tolerance ([-\d.e+]+) ; check tolerance ([-\d.e+]+)", out.stdout)
    assert len(rows) == 4 and all(float(r[3]) <= 1.05e-10 for r in rows)
    assert all(5 <= int(r[1]) <= 30 for r in rows)
    ver = re.findall(r"\[QMG-MRHS-VERIFY\]: rhs 0 single-path iterations (\d+) \(batched (\d+)\) ; relative solution difference ([-\d.e+]+)", out.stdout)
    assert len(ver) == 1 and abs(int(ver[0][0]) - int(ver[0][1])) <= 1 and float(ver[0][2]) < 1e-7
    assert out.stdout.count("[QMG-OPS-STATS]") == 3


def test_c4_staggered_4096_eight_rhs_norms_and_allreduce(golden_dir, monkeypatch):
    """configs[3], one GPU's share: staggered Dslash at 4096^2 for 8 right-hand sides sharing one read of the hopping
    matrices, the per-RHS norm2sq (separately, and fused into the apply as bench.py runs it), and ONE qmg_allreduce_sum_f64 of the norm buffer through a (forced) one-rank RCCL
    communicator obtained by qmg_comm_init_env.  Gate: periodic image of the oracle's 64^2 result on the LAST right-hand
    side (1e-13); every norm against numpy (1e-12); the all-reduce leaves the one rank's values unchanged."""
    import bench
    L, nrhs = 4096, 8
    vol = L * L
    fixture = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    g = qmg.DeviceArray.from_host(bench.tiled_gauge(L, fixture))
    hop = qmg.DeviceArray(4 * vol)
    qmg.staggered_fill(hop, g, L, L)
    qmg.sync()
    g.free()
    desc = qmg.make_desc(L, L, 1, None, hop, 0.04)
    rhs, lhs = qmg.DeviceArray(nrhs * vol), qmg.DeviceArray(nrhs * vol)
    qmg.gaussian(rhs, nrhs * vol, 99)
    ph = np.loadtxt(fixture)
    hop64 = ol.staggered_fill(ol.phases_to_gauge_u1(ph, 64, 64), 64, 64)
    rng = np.random.default_rng(1337)
    v = rng.standard_normal(64 * 64) + 1j * rng.standard_normal(64 * 64)
    want = bench.tile_vector(ol.stencil_apply(ol.make_desc(64, 64, 1, None, hop64, 0.04), v), L, 1)
    lib = qmg.lib()
    qmg.check(lib.qmg_memcpy_h2d(C.c_void_p(rhs.offset((nrhs - 1) * vol)), bench.tile_vector(v, L, 1).ctypes.data_as(C.c_void_p), C.c_size_t(16 * vol), None))
    qmg.stencil_apply(desc, lhs, rhs, qmg.P_ALL | qmg.P_ZERO, nrhs=nrhs, vec_stride=vol)
    norms = qmg.DeviceArray(nrhs, np.float64)
    for k in range(nrhs):
        qmg.check(lib.qmg_norm2sq(C.c_void_p(lhs.offset(k * vol)), C.c_size_t(vol), C.c_void_p(norms.offset(k)), None, None))
    got = lhs.to_host()
    assert np.linalg.norm(got[(nrhs - 1) * vol:] - want) / np.linalg.norm(want) < 1e-13
    ref = np.array([np.vdot(got[k * vol:(k + 1) * vol], got[k * vol:(k + 1) * vol]).real for k in range(nrhs)])
    assert np.allclose(norms.to_host(), ref, rtol=1e-12)
    # the step as bench.py runs it: the apply leaves the norms from the same pass (qmg_stencil_apply_norm2) -- same bytes in lhs,
    # norms to rounding of the separate reductions, bit-reproducible from launch to launch
    lhs2, fused = qmg.DeviceArray(nrhs * vol), qmg.DeviceArray(nrhs, np.float64)
    qmg.stencil_apply_norm2(desc, lhs2, rhs, qmg.P_ALL | qmg.P_ZERO, nrhs, vol, norms_dev=fused.ptr)
    qmg.sync()
    assert np.array_equal(lhs2.to_host(), got)
    first = fused.to_host()
    assert np.allclose(first, ref, rtol=1e-13)
    qmg.stencil_apply_norm2(desc, lhs2, rhs, qmg.P_ALL | qmg.P_ZERO, nrhs, vol, norms_dev=fused.ptr)
    qmg.sync()
    assert np.array_equal(fused.to_host(), first)
    lhs2.free(); fused.free()
    monkeypatch.setenv("QMG_COMM_FORCE_RCCL", "1")
    qmg.comm_init_env(1, 0)
    assert qmg.comm_all_ok(True)
    qmg.check(lib.qmg_allreduce_sum_f64(C.c_void_p(norms.ptr), C.c_size_t(nrhs), None))
    qmg.sync()
    assert np.allclose(norms.to_host(), ref, rtol=1e-12)
    assert lib.qmg_comm_finalize() == 0
    for a in (hop, rhs, lhs, norms):
        a.free()


def test_c5_adaptive_schur_4096_fp64_and_fp32_kcycle(golden_dir):
    """configs[4] on one GPU: adaptive n22 setup (one pass), 4096^2 -> 1024^2 -> 256^2 -> 64^2, nc = 8, solved in the red-black
    (right-block-Jacobi Schur) form of n19 -- (i) all fp64 through the single-vector path, (ii) one more system through the
    batch engine with the K-cycle preconditioner in fp32.  Both: true residual of the ORIGINAL system <= 1e-10."""
    fixture = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    out = subprocess.run([os.path.join(DRIVERS, "n22_wilson_kcycle_adaptive"), "4096", "-0.07", "6.0", "3", "1", fixture, "64", "schur", "nrhs=1", "f32"], cwd=DRIVERS,
                         env=dict(os.environ, QMG_QUIET="1"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "[QMG-ERROR]" not in out.stdout and "nan" not in out.stdout.lower()
    it = int(re.search(r"Multigrid converged in (\d+) iterations", out.stdout).group(1))
    res = float(re.search(r"Check tolerance ([-\d.e+]+)", out.stdout).group(1))
    assert res <= 1.05e-10 and 10 < it < 150
    assert out.stdout.count("[QMG-OPS-STATS]") == 4
    assert "K-cycle preconditioner in fp32" in out.stdout
    f = re.search(r"\[QMG-MRHS\]: rhs 0 converged in (\d+) iterations ; alleged tolerance [-\d.e+]+ ; check tolerance ([-\d.e+]+)", out.stdout)
    assert f and float(f.group(2)) <= 1.05e-10
    assert abs(int(f.group(1)) - it) <= max(4, it // 10)       # another right-hand side, fp32 preconditioner: about the same count


def test_apply_norm2_wilson_2048_four_systems(golden_dir):
    """qmg_stencil_apply_norm2 on the stored Wilson stencil (nc = 2) at 2048^2 for 4 systems: the bytes of qmg_stencil_apply, the norms of the
    separate reductions to 1e-13 -- the size-independent properties of tests/test_gpu_apply_norm.py at a BASELINE size."""
    import bench
    L, nrhs = 2048, 4
    fixture = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    wl = bench.Workload(qmg, L, fixture, 1337)
    n = 2 * L * L
    d = qmg.make_desc(L, L, 2, wl.clover, wl.hopping, bench.MASS)
    rhs, a, b = qmg.DeviceArray(nrhs * n), qmg.DeviceArray(nrhs * n), qmg.DeviceArray(nrhs * n)
    qmg.gaussian(rhs, nrhs * n, 5)
    qmg.stencil_apply(d, a, rhs, qmg.P_ALL | qmg.P_ZERO, nrhs, n)
    norms = qmg.stencil_apply_norm2(d, b, rhs, qmg.P_ALL | qmg.P_ZERO, nrhs, n)
    for k in range(nrhs):
        ref = qmg.norm2sq(a.offset(k * n), n)
        assert abs(norms[k] - ref) <= 1e-13 * ref
    qmg.batch_blas(qmg.BOP_CAXPY, b, n, nrhs, n, (1 << nrhs) - 1, a=[-1.0] * nrhs, x=a)      # b -= a: every byte equal <=> exactly zero
    assert all(v == 0.0 for v in qmg.batch_reduce(qmg.BRED_NORM2, b, None, n, nrhs, n, (1 << nrhs) - 1).real)
    for x in (rhs, a, b):
        x.free()
    wl.free()

