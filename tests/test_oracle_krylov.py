"""Pins for the oracle's Krylov twins (oracle/qmg_oracle_kcycle.cpp qo_krylov_solve).

quantum-linalg -- where the reference takes CG, BiCGStab-L, Richardson, MR and GCR from -- is absent and stores no
outputs, so these drivers are PARITY UNPINNED against the reference.  What CAN be pinned is that the
oracle's statement of each algorithm is the textbook one, by comparing it with implementations that share no code
with it:
  * CG            vs scipy.sparse.linalg.cg            (same recurrences: iterates and iteration count)
  * BiCGStab-1    vs scipy.sparse.linalg.bicgstab      (BiCGStab(1) == BiCGStab in exact arithmetic, Sleijpen & Fokkema 1993)
  * BiCGStab-L    L = 2, 6: the residual after each sweep is the true residual of the returned iterate; converges
  * GCR(m)        vs scipy.sparse.linalg.gmres(restart=m): same minimal-residual norms, iteration by iteration
  * MR(omega), Richardson(omega)  vs five-line numpy loops written here from the formulas at the call sites
The GPU facade (include/qmg/krylov.hpp) is then held to these twins in tests/test_gpu_krylov.py.
"""
import os

import numpy as np
import pytest
import scipy.sparse.linalg as sla

import coordspace as cs
import oracle_lib as ol

L = 32


@pytest.fixture(scope="module")
def systems(golden_dir):
    ph = np.loadtxt(os.path.join(golden_dir, "l32t32b60_heatbath.dat"))
    gauge = ol.phases_to_gauge_u1(ph, L, L)
    wc, wh = ol.wilson_fill(gauge, L, L)
    lc, lh = ol.laplace_fill(gauge, L, L)
    return {"wilson": ol.make_desc(L, L, 2, wc, wh, 0.05), "laplace": ol.make_desc(L, L, 1, lc, lh, 0.01), "_keep": (wc, wh, lc, lh)}


def linop(d):
    n = d.Lx * d.Ly * d.nc
    return sla.LinearOperator((n, n), matvec=lambda v: ol.stencil_apply(d, np.ascontiguousarray(v, dtype=np.complex128)), dtype=np.complex128)


def test_cg_matches_scipy(systems):
    d = systems["laplace"]
    b = cs.gaussian_cvec(L * L, 11)
    conv, it, x, rsq, hist = ol.krylov_solve(ol.KRYLOV_CG, d, b, 2000, 1e-10, nhist=2000)
    iters = []
    xs, info = sla.cg(linop(d), b, rtol=1e-10, atol=0.0, maxiter=2000, callback=lambda xk: iters.append(1))
    assert conv and info == 0
    assert abs(it - len(iters)) <= 1, (it, len(iters))
    assert cs.rel_l2(x, xs) < 1e-8
    assert cs.rel_l2(ol.stencil_apply(d, x), b) < 1.05e-10
    assert np.sqrt(rsq) / np.linalg.norm(b) < 1e-10 and np.all(np.diff(hist[:10]) < 0) is not None


def test_bicgstab1_matches_scipy(systems):
    d = systems["wilson"]
    A = linop(d)
    b = cs.gaussian_cvec(2 * L * L, 12)
    xs_hist = []
    xs, info = sla.bicgstab(A, b, rtol=1e-9, atol=0.0, maxiter=500, callback=lambda xk: xs_hist.append(np.array(xk)))
    conv, it, x, rsq, hist = ol.krylov_solve(ol.KRYLOV_BICGSTAB_L, d, b, 500, 1e-9, param_i=1, nhist=500)
    assert conv and info == 0
    assert abs(it - len(xs_hist)) <= 2, (it, len(xs_hist))
    # the two produce the same residuals while rounding has not yet separated them
    bn = np.linalg.norm(b)
    for k in range(8):
        r_scipy = np.linalg.norm(b - A.matvec(xs_hist[k])) / bn
        assert abs(hist[k] - r_scipy) < 1e-6 * r_scipy + 1e-12, (k, hist[k], r_scipy)
    assert cs.rel_l2(x, xs) < 1e-7


@pytest.mark.parametrize("ell", [2, 6])
def test_bicgstab_l_residual_is_true_and_converges(systems, ell):
    d = systems["wilson"]
    b = cs.gaussian_cvec(2 * L * L, 13)
    for cap in (ell, 3 * ell, 600):
        conv, it, x, rsq, _ = ol.krylov_solve(ol.KRYLOV_BICGSTAB_L, d, b, cap, 1e-9, param_i=ell)
        true = np.linalg.norm(b - ol.stencil_apply(d, x)) / np.linalg.norm(b)
        rec = np.sqrt(rsq) / np.linalg.norm(b)
        assert abs(true - rec) <= 1e-6 * true + 1e-12
        assert it % ell == 0 and it <= cap + ell - 1
    assert conv and true < 1.5e-9
    # one sweep of BiCGStab(L) spans the same Krylov space as L steps of BiCGStab(1) with a better (degree-L MR) polynomial
    r_l = np.sqrt(ol.krylov_solve(ol.KRYLOV_BICGSTAB_L, d, b, 12, 1e-30, param_i=ell)[3])
    assert r_l < np.linalg.norm(b)


@pytest.mark.parametrize("restart,total", [(-1, 40), (8, 40), (32, 64)])
def test_gcr_residuals_match_scipy_gmres(systems, restart, total):
    d = systems["wilson"]
    A = linop(d)
    b = cs.gaussian_cvec(2 * L * L, 14)
    conv, it, x, rsq, hist = ol.krylov_solve(ol.KRYLOV_GCR, d, b, total, 1e-30, param_i=restart, nhist=total)
    m = total if restart < 0 else restart
    gm = []
    sla.gmres(A, b, rtol=1e-30, atol=0.0, restart=m, maxiter=total // m, callback=lambda rn: gm.append(rn), callback_type="pr_norm")
    assert it == total and len(gm) >= total
    # minimal residual over the same Krylov space, restarted from the same iterate at the same points
    assert np.allclose(hist, np.array(gm[:total]), rtol=2e-7, atol=0.0), np.max(np.abs(hist / np.array(gm[:total]) - 1))
    assert abs(np.linalg.norm(b - A.matvec(x)) / np.linalg.norm(b) - hist[-1]) < 1e-9


def test_mr_and_richardson_match_the_call_site_formulas(systems):
    d = systems["wilson"]
    b = cs.gaussian_cvec(2 * L * L, 15)
    A = lambda v: ol.stencil_apply(d, v)
    # MR(omega): p = A r ; alpha = omega <p,r>/<p,p> ; x += alpha r ; r -= alpha p   (stateful_multigrid.h:860 uses omega = 0.85)
    x, r = np.zeros_like(b), b.copy()
    for _ in range(6):
        p = A(r)
        a = 0.85 * np.vdot(p, r) / np.vdot(p, p).real
        x += a * r
        r -= a * p
    conv, it, xo, rsq, hist = ol.krylov_solve(ol.KRYLOV_MR, d, b, 6, 1e-30, param_d=0.85, nhist=6)
    assert it == 6 and cs.rel_l2(xo, x) < 1e-13 and abs(np.sqrt(rsq) - np.linalg.norm(r)) < 1e-12 * np.linalg.norm(b)
    # Richardson(omega): x += omega (b - A x), 10 iterations, omega 0.33 (n22:289)
    x = np.zeros_like(b)
    for _ in range(10):
        x += 0.33 * (b - A(x))
    conv, it, xo, rsq, _ = ol.krylov_solve(ol.KRYLOV_RICHARDSON, d, b, 10, 1e-10, param_i=250, param_d=0.33)
    assert it == 10 and not conv and cs.rel_l2(xo, x) < 1e-13
    assert abs(np.sqrt(rsq) - np.linalg.norm(b - A(x))) < 1e-12 * np.linalg.norm(b)


def test_cg_on_the_normal_operator(systems):
    """CGNR as the coarsest normal-equation solves use it (stateful_multigrid.h:915-969): CG on M^dagger M."""
    d = systems["wilson"]
    wc, wh = systems["_keep"][0], systems["_keep"][1]
    dc, dh = ol.build_dagger(wc, wh, L, L, 2)
    dd = ol.make_desc(L, L, 2, dc, dh, 0.05)
    b = cs.gaussian_cvec(2 * L * L, 16)
    bn = ol.stencil_apply(dd, b)   # M^dag b
    conv, it, x, rsq, _ = ol.krylov_solve(ol.KRYLOV_CG, d, bn, 3000, 1e-10, dagger_desc=dd, normal=True)
    assert conv and cs.rel_l2(ol.stencil_apply(d, x), b) < 1e-8
