"""GPU K-cycle parity: the C++ facade's StatefulMultigridMG::mg_preconditioner + Krylov drivers (product path,
quantum-mg_amd/drivers/n13_wilson_kcycle, every step a HIP kernel through the C-ABI) against the CPU oracle's
K-cycle on the SAME hierarchy: the driver dumps its null vectors and right-hand side, the oracle rebuilds the
transfer operators and Galerkin coarse operators from them and solves the same system.

Bar: true residual <= the requested 1e-10; outer iteration count equal to the oracle's +-1
(bit-different reductions can move a restart); solutions agree to the solve tolerance.
Also runs the n02 counterpart (known answers 4.01 / -1 / 20.0801 / -8.02 / 1 through the facade)."""
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

import coordspace as cs
import oracle_lib as ol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVERS = os.path.join(ROOT, "quantum-mg_amd", "drivers")


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "quantum-mg_amd"), "-j4", "libqmg_hip.so"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", DRIVERS, "-j4"], stdout=subprocess.DEVNULL)


def test_n02_free_laplace_driver_known_answers():
    out = subprocess.run([os.path.join(DRIVERS, "n02_free_laplace")], cwd=DRIVERS, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    selfs = re.findall(r"Self: \(([-\d.e+]+),", out.stdout)
    assert [float(s) for s in selfs] == pytest.approx([4.01, 4.01, 20.0801], abs=1e-12)
    assert "Algorithm CG took" in out.stdout and "[QMG-ERROR]" not in out.stdout


def test_facade_selftest_reference_identities(golden_dir):
    """n00 / n03 / n04 / n05 / n08 / n17 / n18 / n21 identities and solves through the C++ facade on the GPU."""
    out = subprocess.run([os.path.join(DRIVERS, "facade_selftest"), os.path.join(golden_dir, "l32t32b60_heatbath.dat")], cwd=DRIVERS,
                         capture_output=True, text=True, timeout=150)
    assert out.returncode == 0 and "[SELFTEST PASSED]" in out.stdout, out.stdout[-4000:] + out.stderr[-2000:]
    assert "[QMG-ERROR]" not in out.stdout and "[QMG-WARNING]" not in out.stdout
    assert out.stdout.count("[ OK ]") >= 18


@pytest.mark.parametrize("L,n_refine", [(128, 3), (64, 2)])
def test_n19_schur_kcycle_solves_the_original_system(golden_dir, L, n_refine):
    """tests/n19_wilson_kcycle_precond: right-block-Jacobi + Schur on every level, coarse operators built from the rbjacobi
    stencil with their own rbjacobi variants.  The reference prints 'Check tolerance' = ||b - A x|| / ||b|| of the ORIGINAL
    operator after reconstruct_M and expects it at the requested 1e-8 (n19:83,378-380)."""
    gauge_file = os.path.join(golden_dir, "l%dt%db60_heatbath.dat" % (L, L))
    out = subprocess.run([os.path.join(DRIVERS, "n19_wilson_kcycle_precond"), str(L), str(n_refine), gauge_file, str(L)], cwd=DRIVERS,
                         env=dict(os.environ, QMG_QUIET="1"), capture_output=True, text=True, timeout=150)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "[QMG-ERROR]" not in out.stdout and "[QMG-WARNING]" not in out.stdout
    iters = int(re.search(r"Multigrid converged in (\d+) iterations", out.stdout).group(1))
    res = float(re.search(r"Check tolerance ([-\d.e+]+)", out.stdout).group(1))
    assert res <= 1.05e-8 and 0 < iters < 40


@pytest.mark.parametrize("L,n_refine,n_setup,variant", [(64, 2, 1, ""), (128, 2, 1, ""), (64, 2, 1, "schur")])
def test_n22_adaptive_kcycle(golden_dir, L, n_refine, n_setup, variant):
    """tests/n22_wilson_kcycle_adaptive: Richardson-relaxed initial vectors, n_setup adaptive passes through the current
    K-cycle, then the outer solve to 1e-10; prints the reference's OPS / ITER stats lines.  (Coarsest lattices are kept
    >= 4x4 here: a 2x2 coarsest level is launch-latency-bound on a GPU, see DESIGN.md 'next'.)"""
    gauge_file = os.path.join(golden_dir, "l%dt%db60_heatbath.dat" % (L, L))
    cmd = [os.path.join(DRIVERS, "n22_wilson_kcycle_adaptive"), str(L), "-0.07", "6.0", str(n_refine), str(n_setup), gauge_file, str(L)]
    if variant:
        cmd.append(variant)
    out = subprocess.run(cmd, cwd=DRIVERS, env=dict(os.environ, QMG_QUIET="1"), capture_output=True, text=True, timeout=150)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "[QMG-ERROR]" not in out.stdout and "[QMG-WARNING]" not in out.stdout
    res = float(re.search(r"Check tolerance ([-\d.e+]+)", out.stdout).group(1))
    assert res <= 1.05e-10
    assert out.stdout.count("[QMG-OPS-STATS]") == n_refine + 1 and out.stdout.count("[QMG-ITER-STATS]") == n_refine + 1
    nullvec0 = int(re.search(r"Level 0 NullVec (\d+)", out.stdout).group(1))
    assert nullvec0 > 0                      # setup work is booked under NullVec (n22:428-431)


@pytest.mark.parametrize("L,n_refine,coarse_dof,mass", [(64, 1, 8, -0.07), (64, 2, 8, -0.07), (32, 1, 4, -0.03), (128, 2, 24, -0.06)])
def test_wilson_kcycle_matches_oracle(golden_dir, L, n_refine, coarse_dof, mass):
    """(128, 2, 24): the BASELINE configs[2] coarse dof -- 128^2 -> 32^2 -> 8^2, nc = 24 on both coarse levels (kernel B at
    nc = 24, Galerkin build of a 24-dof level from a 24-dof level) -- against the oracle on the same dumped null vectors."""
    gauge_file = os.path.join(golden_dir, "l%dt%db60_heatbath.dat" % (L, L))
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, QMG_QUIET="1", QMG_DUMP_DIR=tmp)
        out = subprocess.run([os.path.join(DRIVERS, "n13_wilson_kcycle"), str(L), str(mass), "6.0", str(n_refine), str(coarse_dof), gauge_file, str(L)],
                             cwd=DRIVERS, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        assert "[QMG-ERROR]" not in out.stdout and "[QMG-WARNING]" not in out.stdout
        gpu_iters = int(re.search(r"Multigrid converged in (\d+) iterations", out.stdout).group(1))
        gpu_res = float(re.search(r"Check tolerance ([-\d.e+]+)", out.stdout).group(1))
        nullvecs = [np.fromfile(os.path.join(tmp, "nullvecs_level%d.bin" % l), dtype=np.complex128) for l in range(n_refine)]
        b = np.fromfile(os.path.join(tmp, "b.bin"), dtype=np.complex128)
        x_gpu = np.fromfile(os.path.join(tmp, "x.bin"), dtype=np.complex128)
    assert gpu_res <= 1e-10
    ph = np.loadtxt(gauge_file)
    gauge = ol.phases_to_gauge_u1(ph, L, L)
    it, x_cpu, true_res, ops, its = ol.wilson_kcycle(L, mass, n_refine, coarse_dof, gauge, nullvecs, b)
    assert it > 0 and true_res <= 1e-10
    assert abs(gpu_iters - it) <= 1, (gpu_iters, it)
    # both solve A x = b to 1e-10: the solutions agree to cond(A) * 1e-10
    assert cs.rel_l2(x_gpu, x_cpu) < 1e-7
    # the GPU solution satisfies the ORACLE's operator
    clover, hopping = ol.wilson_fill(gauge, L, L)
    d = ol.make_desc(L, L, 2, clover, hopping, mass)
    assert cs.rel_l2(ol.stencil_apply(d, x_gpu), b) <= 1.1e-10
    # Dslash counts per level as tracked by the facade (stateful_multigrid.h:854-865).  The facade skips the smoothers'
    # opening A*0 (zero initial guess, krylov.hpp ZeroGuess), takes the pre-smoother's recursive residual instead of
    # recomputing rhs - A z1 (batch.hpp bmr_fixed_zero_guess), and counts the applies it really performs:
    # pre = n_pre, post = n_post per outer iteration (the reference's accounting: n_pre + 2 and n_post + 1).
    m = re.search(r"Level 0 NullVec 0 PreSmooth (\d+) Krylov 0 PostSmooth (\d+)", out.stdout)
    assert int(m.group(1)) == 2 * gpu_iters and int(m.group(2)) == 2 * gpu_iters


def test_n13_128_nc12_stagnation_is_a_property_of_the_configuration(golden_dir):
    """`n13_wilson_kcycle 128 -0.07 6.0 1 12` on the reference's l128t128b60 fixture does not converge (1.2e-4 after 1000
    iterations).  It is not a defect of the nc = 12 path: at mass -0.07 this configuration is past critical (a negative
    real eigenvalue, tests/test_oracle_known_answers.py::test_l128_fixture_is_past_critical_at_mass_minus_007), and the
    CPU oracle's K-cycle on the SAME null vectors stagnates the same way -- outer residual histories agree to 1e-5 (the
    first three to 1e-7; once the coarsest solves run into their cap the history amplifies summation-order rounding) and
    the coarsest GCR takes the same number of iterations call by call (18, 125, then the 1000-iteration cap)."""
    L, nit = 128, 8
    gauge_file = os.path.join(golden_dir, "l128t128b60_heatbath.dat")
    with tempfile.TemporaryDirectory() as tmp:
        out = subprocess.run([os.path.join(DRIVERS, "n13_wilson_kcycle"), "128", "-0.07", "6.0", "1", "12", gauge_file, "128"], cwd=DRIVERS,
                             env=dict(os.environ, QMG_DUMP_DIR=tmp, QMG_MAX_ITER=str(nit)), capture_output=True, text=True, timeout=200)
        assert out.returncode == 1 and "Multigrid failed to converge in %d iterations" % nit in out.stdout, out.stdout[-2000:]
        nullvecs = [np.fromfile(os.path.join(tmp, "nullvecs_level0.bin"), dtype=np.complex128)]
        b = np.fromfile(os.path.join(tmp, "b.bin"), dtype=np.complex128)
    gpu_hist = [float(v) for v in re.findall(r"Level 0: VPGCR-restart Iter \d+ RelTol ([-\d.e+]+)", out.stdout)]
    gpu_coarsest = [(ok == "Success", int(it)) for ok, it in re.findall(r"Level 1 GCR-restart (Success|Fail) Iter (\d+)", out.stdout)]
    assert len(gpu_hist) == nit
    gauge = ol.phases_to_gauge_u1(np.loadtxt(gauge_file), L, L)
    it, _, _, _, _, hist, chist = ol.wilson_kcycle_history(L, -0.07, 1, 12, gauge, nullvecs, b, max_iter=nit)
    assert it < 0 and len(hist) == nit
    assert np.allclose(gpu_hist[:3], hist[:3], rtol=1e-7) and np.allclose(gpu_hist, hist, rtol=1e-5), (gpu_hist, list(hist))
    assert hist[-1] > 5e-3 and hist[-1] / hist[-2] > 0.99            # stagnating, in both
    # the coarsest solves: same iteration counts while they converge, and both hit the cap from the third call on
    assert [c for c in gpu_coarsest[:2]] == [(True, int(chist[0])), (True, int(chist[1]))]
    assert all((not ok_) and n == 1000 for ok_, n in gpu_coarsest[2:nit]) and all(c == -1000 for c in chist[2:nit])


@pytest.mark.parametrize("L,n_refine,coarse_dof,nrhs,point,mass", [(64, 2, 8, 5, False, "-0.07"), (64, 2, 8, 3, True, "-0.07"), (64, 1, 12, 16, False, "-0.07"),
                                                                  (64, 1, 24, 2, True, "-0.07"), (128, 1, 12, 4, False, "-0.06"), (128, 2, 8, 3, True, "-0.06")])
def test_batched_kcycle_reproduces_the_single_solves(golden_dir, L, n_refine, coarse_dof, nrhs, point, mass):
    """include/qmg/batch.hpp: up to 16 systems advance through one K-cycle iteration together (coarse applies on the
    f64 matrix cores, null vectors streamed once per step).  `verify` re-solves every system alone through the
    single-vector path: iteration counts equal (+-1), solutions equal to solver accuracy, every true residual <= 1e-10.
    With `point`, system 1 is a point source and converges on its own schedule, so the outer-level freeze masks are
    exercised as well as the inner ones (coarse solves converge per system all the time).  The 128^2 fixture runs at mass
    -0.06: at -0.07 it is past critical (test_n13_128_nc12_stagnation_is_a_property_of_the_configuration)."""
    gauge_file = os.path.join(golden_dir, "l%dt%db60_heatbath.dat" % (L, L))
    env = dict(os.environ, QMG_QUIET="1")
    if point:
        env["QMG_MRHS_POINT"] = "1"
    out = subprocess.run([os.path.join(DRIVERS, "n13_wilson_kcycle_mrhs"), str(L), mass, "6.0", str(n_refine), str(coarse_dof), gauge_file, str(L), str(nrhs), "verify"],
                         cwd=DRIVERS, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "[QMG-ERROR]" not in out.stdout and "[QMG-WARNING]" not in out.stdout
    rows = re.findall(r"\[QMG-MRHS\]: rhs (\d+) converged in (\d+) iterations ; alleged tolerance ([-\d.e+]+) ; check tolerance ([-\d.e+]+)", out.stdout)
    assert len(rows) == nrhs
    assert all(float(r[3]) <= 1.05e-10 for r in rows)
    ver = re.findall(r"\[QMG-MRHS-VERIFY\]: rhs (\d+) single-path iterations (\d+) \(batched (\d+)\) ; relative solution difference ([-\d.e+]+)", out.stdout)
    assert len(ver) == nrhs
    for _, single_it, batch_it, diff in ver:
        assert abs(int(single_it) - int(batch_it)) <= 1
        assert float(diff) < 1e-7


def test_n22_four_levels_and_batched_setup(golden_dir):
    """BASELINE configs[4] shape (four levels) at 256^2 -> 64^2 -> 16^2 -> 4^2 on the tiled l64 config.  The reference's
    build_coarse_by_restrict declares no doubling type (n22:682), which leaves every 'down' null vector of the third
    transfer zero and the hierarchy NaN; the driver declares it (see its comment).  Also: the adaptive relaxations of a
    level run as one lock-step batch -- the sequential setup (QMG_NO_BATCHED_SETUP) must give the same solve."""
    gauge_file = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    cmd = [os.path.join(DRIVERS, "n22_wilson_kcycle_adaptive"), "256", "-0.07", "6.0", "3", "1", gauge_file, "64"]
    res = {}
    for tag, extra in (("batched", {}), ("sequential", {"QMG_NO_BATCHED_SETUP": "1"})):
        out = subprocess.run(cmd, cwd=DRIVERS, env=dict(os.environ, QMG_QUIET="1", **extra), capture_output=True, text=True, timeout=150)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        assert "nan" not in out.stdout.lower()
        assert out.stdout.count("[QMG-OPS-STATS]") == 4
        res[tag] = (int(re.search(r"Multigrid converged in (\d+) iterations", out.stdout).group(1)), float(re.search(r"Check tolerance ([-\d.e+]+)", out.stdout).group(1)),
                    ("(batched)" in out.stdout))
    assert res["batched"][2] and not res["sequential"][2]
    assert res["batched"][1] <= 1.05e-10 and res["sequential"][1] <= 1.05e-10
    assert abs(res["batched"][0] - res["sequential"][0]) <= 1


def test_kcycle_with_f32_stored_coarse_operators(golden_dir):
    """Default of the K-cycle hierarchy (multigrid.hpp; QMG_COARSE_F32=0 / QMG_COARSE_BITS=64 switch it off): the Galerkin operators are streamed as
    complex<float>; the hierarchy only preconditions, so the outer fp64 VPGCR still reaches 1e-10 in (about) the same number
    of iterations as with fp64-stored coarse matrices.  QMG_COARSE_BITS=16 (opt-in): complex<half> storage (kernel B32 / C widen the tile on its
    way into LDS), same bar."""
    gauge_file = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    its = {}
    for tag, extra in (("f64", {"QMG_COARSE_F32": "0"}), ("f32", {}), ("f16", {"QMG_COARSE_BITS": "16"})):
        out = subprocess.run([os.path.join(DRIVERS, "n13_wilson_kcycle_mrhs"), "128", "-0.07", "6.0", "2", "8", gauge_file, "64", "3", "verify"],
                             cwd=DRIVERS, env=dict(os.environ, QMG_QUIET="1", **extra), capture_output=True, text=True, timeout=150)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        assert ("complex<float>" in out.stdout) == (tag == "f32") and ("complex<half>" in out.stdout) == (tag == "f16")
        rows = re.findall(r"\[QMG-MRHS\]: rhs (\d+) converged in (\d+) iterations ; alleged tolerance ([-\d.e+]+) ; check tolerance ([-\d.e+]+)", out.stdout)
        assert len(rows) == 3 and all(float(r[3]) <= 1.05e-10 for r in rows)
        its[tag] = [int(r[1]) for r in rows]
    assert all(abs(a - b) <= 2 and abs(a - c) <= 2 for a, b, c in zip(its["f64"], its["f32"], its["f16"])), its


def test_schur_kcycle_with_narrow_stored_rbjacobi_operators(golden_dir):
    """The red-black (Schur) K-cycle streams the right-block-Jacobi hops and cinv, not the ORIGINAL arrays: a preconditioner hierarchy keeps
    complex<float> (default) or complex<half> (QMG_COARSE_BITS=16) copies of those as well (Stencil2D::narrow_rbjacobi_copies).  n22 in its
    Schur form at 256^2 (4 levels), 24 coarse dof: the same outer iterations (+-1) and the ORIGINAL system's true residual <= 1e-10 with
    fp64, fp32 and 16-bit storage."""
    gauge_file = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    its = {}
    for tag, bits in (("f64", "64"), ("f32", "32"), ("f16", "16")):
        out = subprocess.run([os.path.join(DRIVERS, "n22_wilson_kcycle_adaptive"), "256", "-0.07", "6.0", "3", "1", gauge_file, "64", "schur"], cwd=DRIVERS,
                             env=dict(os.environ, QMG_QUIET="1", QMG_COARSE_BITS=bits), capture_output=True, text=True, timeout=200)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        assert "[QMG-ERROR]" not in out.stdout and "[QMG-WARNING]" not in out.stdout
        assert ("as complex<float>" in out.stdout) == (tag == "f32") and ("as complex<half>" in out.stdout) == (tag == "f16")
        its[tag] = int(re.search(r"Multigrid converged in (\d+) iterations", out.stdout).group(1))
        assert float(re.search(r"Check tolerance ([-\d.e+]+)", out.stdout).group(1)) <= 1.05e-10
    assert abs(its["f32"] - its["f64"]) <= 1 and abs(its["f16"] - its["f64"]) <= 1, its


def test_cgne_smoothers_in_both_engines(golden_dir):
    """LevelSolveMG::pre_cgne / post_cgne (stateful_multigrid.h:847-857, 1032-1042: MR on M M^dagger, then M^dagger; no reference test sets the flags, the
    driver takes QMG_SMOOTHER=cgne).  The lock-step batch engine (dagger stencil by name, the MR dots riding on the second apply, fixed-count
    device-scalar form) against the reference-shaped single-vector code of multigrid.hpp (perform_swap_dagger, minv_vector_minres on
    apply_M_M_dagger): same outer iteration count, the same solution, true residual <= 1e-10; the trackers count what each engine performs
    (per level visit 2 (2 x 2 + 1) smoother applies, plus the two residual applies the single-vector engine spends and the batch engine saves
    one of).  Then three systems in lock step, fp64 and with the K-cycle in complex<float> (the dagger stencil's fp32 shadow)."""
    gauge_file = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    res = {}
    for tag, extra in (("batch", {}), ("single", {"QMG_KCYCLE_ENGINE": "single"}), ("mr", {"QMG_SMOOTHER": "mr"})):
        with tempfile.TemporaryDirectory() as d:
            env = dict(os.environ, QMG_QUIET="1", QMG_SMOOTHER="cgne", QMG_DUMP_DIR=d)
            env.update(extra)
            out = subprocess.run([os.path.join(DRIVERS, "n13_wilson_kcycle"), "128", "-0.07", "6.0", "2", "8", gauge_file, "64"], cwd=DRIVERS, env=env,
                                 capture_output=True, text=True, timeout=150)
            assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
            assert "[QMG-ERROR]" not in out.stdout and "[QMG-WARNING]" not in out.stdout
            assert ("CGNE smoothers" in out.stdout) == (tag != "mr")
            it = int(re.search(r"Multigrid converged in (\d+) iterations", out.stdout).group(1))
            chk = float(re.search(r"Check tolerance ([-\d.e+]+)", out.stdout).group(1))
            m = re.search(r"Level 0 NullVec 0 PreSmooth (\d+) Krylov 0 PostSmooth (\d+)", out.stdout)
            x = np.fromfile(os.path.join(d, "x.bin"), dtype=np.complex128)
            res[tag] = (it, chk, int(m.group(1)), int(m.group(2)), x)
    for tag in res:
        assert res[tag][1] <= 1.05e-10, (tag, res[tag][:4])
    assert res["batch"][0] == res["single"][0], (res["batch"][:4], res["single"][:4])
    rel = np.linalg.norm(res["batch"][4] - res["single"][4]) / np.linalg.norm(res["single"][4])
    assert rel < 1e-8, rel
    it = res["batch"][0]
    # level-0 visits = outer iterations; smoother applies per visit: MR on M M^dagger costs 2 per step, + the closing M^dagger
    assert res["batch"][2] == it * 5 and res["batch"][3] == it * 5, res["batch"][:4]
    assert res["single"][2] == it * 6 and res["single"][3] == it * 5, res["single"][:4]
    assert res["mr"][2] == res["mr"][0] * 2                                                                # (plain MR: 2 applies per visit)
    for extra in ([], ["f32"]):
        out = subprocess.run([os.path.join(DRIVERS, "n13_wilson_kcycle_mrhs"), "128", "-0.07", "6.0", "2", "8", gauge_file, "64", "3", "verify"] + extra, cwd=DRIVERS,
                             env=dict(os.environ, QMG_QUIET="1", QMG_SMOOTHER="cgne"), capture_output=True, text=True, timeout=200)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        assert "[QMG-ERROR]" not in out.stdout and "[QMG-WARNING]" not in out.stdout and "CGNE smoothers" in out.stdout
        rows = re.findall(r"\[QMG-MRHS\]: rhs (\d+) converged in (\d+) iterations ; alleged tolerance ([-\d.e+]+) ; check tolerance ([-\d.e+]+)", out.stdout)
        assert len(rows) == 3 and all(float(r[3]) <= 1.05e-10 for r in rows), out.stdout[-2000:]
        ver = re.findall(r"\[QMG-MRHS-VERIFY\]: rhs (\d+) single-path iterations (\d+) \(batched (\d+)\) ; relative solution difference ([-\d.e+]+)", out.stdout)
        assert len(ver) == 3
        for _, single_it, batch_it, diff in ver:
            assert abs(int(single_it) - int(batch_it)) <= (2 if extra else 1) and float(diff) < 1e-7


def test_batched_schur_kcycle_reproduces_the_single_solves(golden_dir):
    """n19 configuration (even-odd Schur complement of the right-block-Jacobi operator on every level, four levels
    128 -> 32 -> 8 -> 2) for a lock-step batch: prepare / Schur solve / reconstruct per system, inner tolerances per
    system (coarse_tol |r| / |r_prep|), verified against the single-vector path system by system."""
    gauge_file = os.path.join(golden_dir, "l128t128b60_heatbath.dat")
    out = subprocess.run([os.path.join(DRIVERS, "n19_wilson_kcycle_precond"), "128", "3", gauge_file, "128", "nrhs=3"], cwd=DRIVERS,
                         env=dict(os.environ, QMG_QUIET="1", QMG_MRHS_VERIFY="1"), capture_output=True, text=True, timeout=150)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "[QMG-ERROR]" not in out.stdout and "[QMG-WARNING]" not in out.stdout
    rows = re.findall(r"\[QMG-MRHS\]: rhs (\d+) converged in (\d+) iterations ; alleged tolerance ([-\d.e+]+) ; check tolerance ([-\d.e+]+)", out.stdout)
    assert len(rows) == 3 and all(float(r[3]) <= 2e-8 for r in rows)     # n19 solves to 1e-8
    ver = re.findall(r"\[QMG-MRHS-VERIFY\]: rhs (\d+) single-path iterations (\d+) \(batched (\d+)\) ; relative solution difference ([-\d.e+]+)", out.stdout)
    assert len(ver) == 3
    for _, single_it, batch_it, diff in ver:
        assert abs(int(single_it) - int(batch_it)) <= 1 and float(diff) < 1e-6


def test_n22_rank_sharded_setup_through_rccl(golden_dir):
    """SURVEY 8e setup phase in the C++ driver: the adaptive relaxations of a level are owned by ranks (j mod world) and
    exchanged by one RCCL sum all-reduce per level.  One GPU here, so the communicator has one rank -- forced through
    RCCL (QMG_COMM_FORCE_RCCL; qmg_comm_init_env, the all-ok flag and the data all-reduce all go through librccl) -- and
    the run must reproduce the plain run exactly.  The multi-rank rendezvous (TCP) runs in tests/test_distributed_cpu.py."""
    gauge_file = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    cmd = [os.path.join(DRIVERS, "n22_wilson_kcycle_adaptive"), "128", "-0.07", "6.0", "2", "1", gauge_file, "64"]
    plain = subprocess.run(cmd, cwd=DRIVERS, env=dict(os.environ, QMG_QUIET="1"), capture_output=True, text=True, timeout=150)
    forced = subprocess.run(cmd, cwd=DRIVERS, env=dict(os.environ, QMG_QUIET="1", QMG_COMM_FORCE_RCCL="1",
                                                       RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), capture_output=True, text=True, timeout=150)
    assert plain.returncode == 0 and forced.returncode == 0, forced.stdout[-2000:] + forced.stderr[-2000:]
    assert "rank 0 of 1 on device 0" in forced.stdout
    pick = lambda out: (re.search(r"Multigrid converged in (\d+) iterations with alleged tolerance ([-\d.e+]+)", out.stdout).groups(),
                        re.search(r"Check tolerance ([-\d.e+]+)", out.stdout).group(1))
    assert pick(plain) == pick(forced)


@pytest.mark.parametrize("extra", [[], ["nrhs=2"], ["nrhs=2", "f32"]])
def test_rbjacobi_hops_from_the_links_reproduce_the_stored_stencil_solve(golden_dir, extra):
    """Schur-complement K-cycle (n22 schur) with the level-0 right-block-Jacobi hops applied from the links x cinv (qmg_wilson_hops_direct,
    the default) against the same solve streaming the built right-block-Jacobi hopping (QMG_WILSON_DIRECT_RBJ=0): in fp64 the applies are
    bit-identical, so every printed iteration count and residual must be the same text; the fp32 K-cycle (different rounding of the fp32
    entries) must converge in the same number of outer iterations +- 1."""
    gauge_file = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    cmd = [os.path.join(DRIVERS, "n22_wilson_kcycle_adaptive"), "256", "-0.07", "6.0", "2", "1", gauge_file, "64", "schur"] + extra
    outs = {}
    for flag in ("1", "0"):
        p = subprocess.run(cmd, cwd=DRIVERS, env=dict(os.environ, QMG_QUIET="1", QMG_WILSON_DIRECT_RBJ=flag), capture_output=True, text=True, timeout=150)
        assert p.returncode == 0 and "[QMG-ERROR]" not in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
        outs[flag] = [re.sub(r", t = [-\d.e+]+ s", "", l) for l in p.stdout.splitlines() if "TIMING" not in l]   # all but the wall-clock text
    if "f32" not in extra:
        # the APPLIES are bit-identical; the smoothers' <p,r>, <p,p> are not any more: from the links they come out of kernel W's epilogue
        # (per-wavefront partial sums), from the stored stencil out of a separate pass -- same numbers, different summation order.  So:
        # the same lines, the same iteration counts, residuals equal to 1e-6.
        assert len(outs["1"]) == len(outs["0"])
        num = re.compile(r"[-+]?\d+\.\d+(?:e[-+]?\d+)?")
        for a, b in zip(outs["1"], outs["0"]):
            assert num.sub("#", a) == num.sub("#", b), (a, b)
            for u, v in zip(num.findall(a), num.findall(b)):
                assert abs(float(u) - float(v)) <= 1e-6 * max(abs(float(u)), abs(float(v))), (a, b)
    else:
        rows = {f: re.findall(r"\[QMG-MRHS\]: rhs (\d+) converged in (\d+) iterations ; alleged tolerance ([-\d.e+]+) ; check tolerance ([-\d.e+]+)", "\n".join(o)) for f, o in outs.items()}
        assert len(rows["1"]) == 2 and len(rows["0"]) == 2
        for a, b in zip(rows["1"], rows["0"]):
            assert abs(int(a[1]) - int(b[1])) <= 1 and float(a[3]) <= 1.05e-10


def _kcycle_run(driver, args, env_extra, timeout=200):
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, QMG_QUIET="1", QMG_DUMP_DIR=d)
        env.update(env_extra)
        out = subprocess.run([os.path.join(DRIVERS, driver)] + args, cwd=DRIVERS, env=env, capture_output=True, text=True, timeout=timeout)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        assert "[QMG-ERROR]" not in out.stdout and "[QMG-WARNING]" not in out.stdout, out.stdout[-3000:]
        it = int(re.search(r"Multigrid converged in (\d+) iterations", out.stdout).group(1))
        chk = float(re.search(r"Check tolerance ([-\d.e+]+)", out.stdout).group(1))
        x = np.fromfile(os.path.join(d, "x.bin"), dtype=np.complex128)
    return it, chk, x, out.stdout


@pytest.mark.parametrize("hooks", [
    {"QMG_SOLVE_TYPE": "jacobi"},
    {"QMG_SOLVE_TYPE": "jacobi", "QMG_SMOOTHER": "cgne"},
    {"QMG_SOLVE_TYPE": "jacobi", "QMG_COARSEST_TYPE": "rbj_mmd"},
    {"QMG_SOLVE_TYPE": "jacobi", "QMG_COARSEST_TYPE": "rbj_mdm", "QMG_NORMAL_SHIFT": "0.01"},
    {"QMG_COARSEST_TYPE": "rbj_mdm"},
], ids=["jacobi", "jacobi-cgne", "jacobi-coarsest-MMdag", "jacobi-coarsest-MdagM-shifted", "schur-coarsest-MdagM"])
def test_right_jacobi_levels_and_normal_coarsest_solves_in_both_engines(golden_dir, hooks):
    """The branches of StatefulMultigridMG::mg_preconditioner no reference driver selects (stateful_multigrid.h:845-857 CGNE on a RIGHT_JACOBI level, :930-960 the
    coarsest solve by CG on a normal-equation operator with normal_shift; stencil_2d.h:2418-2527 apply / prepare / reconstruct by type), reached through the n19
    counterpart's environment hooks.  The lock-step batch engine (operators by name: right-block-Jacobi hops + unit shift, its dagger stencil, batched CG) against
    the reference-shaped single-vector code of multigrid.hpp (perform_swap_*, minv_vector_cg): the same outer iteration count, the same reconstructed solution,
    true residual against the ORIGINAL operator <= 1e-7 (tol 1e-8 on the preconditioned system).  Parity here is engine against engine: the reference holds no
    output for these branches."""
    gauge_file = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    args = ["128", "2", gauge_file, "64"]
    b = _kcycle_run("n19_wilson_kcycle_precond", args, hooks)
    s = _kcycle_run("n19_wilson_kcycle_precond", args, dict(hooks, QMG_KCYCLE_ENGINE="single"))
    assert "solve type" in b[3]
    assert b[1] <= 1e-7 and s[1] <= 1e-7, (b[:2], s[:2])
    assert abs(b[0] - s[0]) <= 1, (b[:2], s[:2])
    rel = np.linalg.norm(b[2] - s[2]) / np.linalg.norm(s[2])
    assert rel < 1e-6, rel


@pytest.mark.parametrize("ctype", ["mmd", "mdm"])
def test_coarsest_cg_on_the_original_hierarchy_in_both_engines(golden_dir, ctype):
    """n13's hierarchy with the coarsest solve by CG on M M^dagger / M^dagger M (CoarsestSolveMG::coarsest_stencil_app, stateful_multigrid.h:930-960; prepare_M /
    reconstruct_M of those types, stencil_2d.h:1413-1446): batch engine (bcg_core, the dagger stencil by name) against the single-vector engine
    (minv_vector_cg_restart, perform_swap_dagger) -- same outer iterations, same solution, true residual <= 1e-10; then three systems in lock step."""
    gauge_file = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    args = ["128", "-0.07", "6.0", "2", "8", gauge_file, "64"]
    b = _kcycle_run("n13_wilson_kcycle", args, {"QMG_COARSEST_TYPE": ctype})
    s = _kcycle_run("n13_wilson_kcycle", args, {"QMG_COARSEST_TYPE": ctype, "QMG_KCYCLE_ENGINE": "single"})
    assert b[1] <= 1.05e-10 and s[1] <= 1.05e-10, (b[:2], s[:2])
    assert abs(b[0] - s[0]) <= 1, (b[:2], s[:2])
    rel = np.linalg.norm(b[2] - s[2]) / np.linalg.norm(s[2])
    assert rel < 1e-7, rel
    out = subprocess.run([os.path.join(DRIVERS, "n13_wilson_kcycle_mrhs"), "128", "-0.07", "6.0", "2", "8", gauge_file, "64", "3", "verify"], cwd=DRIVERS,
                         env=dict(os.environ, QMG_QUIET="1", QMG_COARSEST_TYPE=ctype), capture_output=True, text=True, timeout=200)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "[QMG-ERROR]" not in out.stdout and "[QMG-WARNING]" not in out.stdout
    rows = re.findall(r"\[QMG-MRHS\]: rhs (\d+) converged in (\d+) iterations ; alleged tolerance ([-\d.e+]+) ; check tolerance ([-\d.e+]+)", out.stdout)
    assert len(rows) == 3 and all(float(r[3]) <= 1.05e-10 for r in rows), out.stdout[-2000:]


def test_batched_right_jacobi_solves_follow_the_single_solves(golden_dir):
    """Three systems in lock step with the outer solve, every level and the coarsest solve on the RIGHT_JACOBI operator and CGNE smoothers (n19 counterpart,
    QMG_SOLVE_TYPE=jacobi, nrhs=3): each system converges, is reconstructed (x = C^-1 y) to a true residual <= 1e-7 against the ORIGINAL operator, and matches its
    single-vector solve (QMG_MRHS_VERIFY: iteration counts within 1, solutions to 1e-6)."""
    gauge_file = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    out = subprocess.run([os.path.join(DRIVERS, "n19_wilson_kcycle_precond"), "128", "2", gauge_file, "64", "nrhs=3"], cwd=DRIVERS,
                         env=dict(os.environ, QMG_QUIET="1", QMG_SOLVE_TYPE="jacobi", QMG_SMOOTHER="cgne", QMG_MRHS_VERIFY="1"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "[QMG-ERROR]" not in out.stdout and "[QMG-WARNING]" not in out.stdout, out.stdout[-3000:]
    rows = re.findall(r"\[QMG-MRHS\]: rhs (\d+) converged in (\d+) iterations ; alleged tolerance ([-\d.e+]+) ; check tolerance ([-\d.e+]+)", out.stdout)
    assert len(rows) == 3 and all(float(r[3]) <= 1e-7 for r in rows), out.stdout[-2000:]
    ver = re.findall(r"\[QMG-MRHS-VERIFY\]: rhs (\d+) single-path iterations (\d+) \(batched (\d+)\) ; relative solution difference ([-\d.e+]+)", out.stdout)
    assert len(ver) == 3 and all(abs(int(v[1]) - int(v[2])) <= 1 and float(v[3]) < 1e-6 for v in ver), ver
