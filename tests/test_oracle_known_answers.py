"""Pin the CPU oracle (not gpu): the reference's own known answers + coordinate-space operators.

* n02 known answers: tests/n02_free_laplace_test/free_laplace.cpp:31-100 (32x24, m^2 = 0.01).
* cshift identities: tests/n00_cshift/cshift_2d_test.cpp (6x4, dof 2; shift then inverse shift).
* operators on the reference's own U(1) fixtures (tests/common_cfgs_u1/l32t32b60, l64t64b60)
  against tests/coordspace.py.
"""
import os

import numpy as np
import pytest

import coordspace as cs
import oracle_lib as ol

TOL = 1e-13


def test_index_round_trip():
    for Lx, Ly in ((6, 4), (32, 24), (8, 8)):
        seen = set()
        for x in range(Lx):
            for y in range(Ly):
                i = ol.coord_to_index(Lx, Ly, x, y)
                assert ol.index_to_coord(Lx, Ly, i) == (x, y)
                assert (i >= Lx * Ly // 2) == bool((x + y) & 1)
                seen.add(i)
        assert seen == set(range(Lx * Ly))
        grid = cs.site_index_grid(Lx, Ly)
        assert all(grid[x, y] == ol.coord_to_index(Lx, Ly, x, y) for x in range(Lx) for y in range(Ly))


def test_n02_free_laplace_known_answers():
    Lx, Ly, msq = 32, 24, 0.1 * 0.1
    clover, hopping = ol.free_laplace_fill(Lx, Ly)
    d = ol.make_desc(Lx, Ly, 1, clover, hopping, shift=msq)
    idx = lambda x, y: ol.coord_to_index(Lx, Ly, x % Lx, y % Ly)
    for (x0, y0) in ((Lx // 2, Ly // 2), (Lx // 2, Ly // 2 + 1)):      # even point, odd point
        rhs = ol.cvec(Lx * Ly)
        rhs[idx(x0, y0)] = 1.0
        lhs = ol.stencil_apply(d, rhs, ol.P_ALL)                      # accumulate into zeros, as the test does
        assert lhs[idx(x0, y0)] == pytest.approx(4.01, abs=1e-15)
        for dx, dy in ((1, 0), (0, 1), (-1, 0), (0, -1)):
            assert lhs[idx(x0 + dx, y0 + dy)] == pytest.approx(-1.0, abs=1e-15)
        assert abs(lhs).sum() == pytest.approx(8.01, abs=1e-13)
    twice = ol.stencil_apply(d, lhs, ol.P_ALL)                        # free_laplace.cpp:93-100
    assert twice[idx(x0, y0)] == pytest.approx(20.0801, abs=1e-13)
    assert twice[idx(x0 + 1, y0)] == pytest.approx(-8.02, abs=1e-13)
    assert twice[idx(x0 + 2, y0)] == pytest.approx(1.0, abs=1e-13)


@pytest.mark.parametrize("dof", [1, 2, 4])
def test_n00_cshift_matches_coordinate_roll(dof):
    Lx, Ly = 6, 4
    v = cs.gaussian_cvec(Lx * Ly * dof, 5)
    grid = cs.eo_to_grid(v, Lx, Ly, dof)
    for cdir, want in ((ol.CSHIFT_XP1, cs.fwd(grid, 0)), (ol.CSHIFT_YP1, cs.fwd(grid, 1)),
                       (ol.CSHIFT_XM1, cs.bwd(grid, 0)), (ol.CSHIFT_YM1, cs.bwd(grid, 1))):
        got = ol.cshift(v, cdir, ol.EO_FROM_EVENODD, dof, Lx, Ly)
        assert np.array_equal(got, cs.grid_to_eo(want, Lx, Ly, dof))
    # shift then inverse shift is the identity (what n00 prints)
    for a, b in ((ol.CSHIFT_XP1, ol.CSHIFT_XM1), (ol.CSHIFT_YP1, ol.CSHIFT_YM1)):
        back = ol.cshift(ol.cshift(v, a, ol.EO_FROM_EVENODD, dof, Lx, Ly), b, ol.EO_FROM_EVENODD, dof, Lx, Ly)
        assert np.array_equal(back, v)
    # single-parity shift only touches the opposite half
    half = Lx * Ly * dof // 2
    got = ol.cshift(v, ol.CSHIFT_XP1, ol.EO_FROM_EVEN, dof, Lx, Ly)
    assert np.all(got[:half] == 0) and np.array_equal(got[half:], cs.grid_to_eo(cs.fwd(grid, 0), Lx, Ly, dof)[half:])


def _fixture_links(golden_dir, L):
    ph = np.loadtxt(os.path.join(golden_dir, "l%dt%db60_heatbath.dat" % (L, L)))
    assert ph.size == 2 * L * L
    return ph, cs.phases_to_links(ph, L, L)


@pytest.mark.parametrize("L", [32, 64])
def test_gauge_reader_layout(golden_dir, L):
    ph, (Ux, Uy) = _fixture_links(golden_dir, L)
    g_file = ol.read_gauge_u1(os.path.join(golden_dir, "l%dt%db60_heatbath.dat" % (L, L)), L, L)
    g_np = cs.links_to_eo_gauge(Ux, Uy, L, L)
    assert cs.rel_l2(g_file, g_np) < 1e-15
    assert np.array_equal(ol.phases_to_gauge_u1(ph, L, L), g_file)


@pytest.mark.parametrize("L", [32, 64])
def test_wilson_vs_coordinate_space(golden_dir, L):
    ph, (Ux, Uy) = _fixture_links(golden_dir, L)
    gauge = ol.phases_to_gauge_u1(ph, L, L)
    mass = -0.07
    clover, hopping = ol.wilson_fill(gauge, L, L, 1.0)
    d = ol.make_desc(L, L, 2, clover, hopping, shift=mass)
    rhs = cs.gaussian_cvec(L * L * 2, 1337)
    got = ol.stencil_apply(d, rhs)
    want = cs.grid_to_eo(cs.wilson_apply(cs.eo_to_grid(rhs, L, L, 2), Ux, Uy, mass), L, L, 2)
    assert cs.rel_l2(got, want) < TOL
    # gamma5-hermiticity: g5 D g5 = D^dag  <=>  <y, g5 D g5 x> = conj(<x, D y>)... checked as <g5 y, D g5 x> = conj <x, D^... >
    y = cs.gaussian_cvec(L * L * 2, 7)
    g5 = np.tile([1.0, -1.0], L * L)
    lhs1 = np.vdot(y, g5 * ol.stencil_apply(d, g5 * rhs))
    lhs2 = np.conj(np.vdot(rhs, ol.stencil_apply(d, y)))
    assert abs(lhs1 - lhs2) / abs(lhs1) < 1e-12


@pytest.mark.parametrize("L", [32, 64])
def test_staggered_and_laplace_vs_coordinate_space(golden_dir, L):
    ph, (Ux, Uy) = _fixture_links(golden_dir, L)
    gauge = ol.phases_to_gauge_u1(ph, L, L)
    rhs = cs.gaussian_cvec(L * L, 1337)
    hop = ol.staggered_fill(gauge, L, L)
    d = ol.make_desc(L, L, 1, None, hop, shift=0.04)
    want = cs.grid_to_eo(cs.staggered_apply(cs.eo_to_grid(rhs, L, L, 1), Ux, Uy, 0.04), L, L, 1)
    assert cs.rel_l2(ol.stencil_apply(d, rhs), want) < TOL
    clover, hop = ol.laplace_fill(gauge, L, L)
    d = ol.make_desc(L, L, 1, clover, hop, shift=0.01)
    want = cs.grid_to_eo(cs.laplace_apply(cs.eo_to_grid(rhs, L, L, 1), Ux, Uy, 0.01), L, L, 1)
    assert cs.rel_l2(ol.stencil_apply(d, rhs), want) < TOL


def test_pieces_sum_to_full_apply(golden_dir):
    L = 32
    ph, _ = _fixture_links(golden_dir, L)
    clover, hopping = ol.wilson_fill(ol.phases_to_gauge_u1(ph, L, L), L, L)
    d = ol.make_desc(L, L, 2, clover, hopping, shift=0.1 + 0.02j, eo_shift=0.03, dof_shift=-0.05j)
    rhs = cs.gaussian_cvec(L * L * 2, 3)
    full = ol.stencil_apply(d, rhs)
    acc = ol.cvec(L * L * 2)
    pieces = [ol.P_CLOVER_E, ol.P_CLOVER_O, ol.P_SHIFT_E, ol.P_SHIFT_O]
    pieces += [ol.P_EO_XP1 << k for k in range(4)] + [ol.P_OE_XP1 << k for k in range(4)]
    for p in pieces:
        ol.stencil_apply(d, rhs, p, lhs=acc)
    assert cs.rel_l2(acc, full) < TOL
    # eo_shift flips sign on odd sites, dof_shift on the bottom half of the dof (stencil_2d.h:890-908)
    only_shift = ol.stencil_apply(d, rhs, ol.P_SHIFT | ol.P_ZERO)
    half = L * L
    sgn_c = np.tile([1.0, -1.0], L * L)
    want = (0.1 + 0.02j) * rhs + 0.03 * np.concatenate([rhs[:half], -rhs[half:]]) + (-0.05j) * sgn_c * rhs
    assert cs.rel_l2(only_shift, want) < TOL


def test_n17_dagger_and_n18_rbjacobi_identities(golden_dir):
    L = 32
    ph, _ = _fixture_links(golden_dir, L)
    clover, hopping = ol.wilson_fill(ol.phases_to_gauge_u1(ph, L, L), L, L)
    mass = -0.07 + 0.01j
    d = ol.make_desc(L, L, 2, clover, hopping, shift=mass)
    x, y = cs.gaussian_cvec(L * L * 2, 11), cs.gaussian_cvec(L * L * 2, 12)
    dc, dh = ol.build_dagger(clover, hopping, L, L, 2)
    dd = ol.make_desc(L, L, 2, dc, dh, shift=np.conj(mass))           # perform_swap_dagger conjugates shifts (:1159)
    a = np.vdot(y, ol.stencil_apply(d, x))
    b = np.vdot(ol.stencil_apply(dd, y), x)
    assert abs(a - b) / abs(a) < 1e-12                                 # n17: <y, M x> = <M^dag y, x>
    # right block Jacobi: M_rbj = M . C^-1, C = clover + shift  (stencil_2d.h:1452-1601)
    cinv, rclover, rhopping = ol.build_rbjacobi(d)
    drb = ol.make_desc(L, L, 2, rclover, rhopping)                     # shifts are zero after the swap (:1620-1622)
    dcinv = ol.make_desc(L, L, 2, cinv, None)
    cinv_x = ol.stencil_apply(dcinv, x, ol.P_CLOVER | ol.P_ZERO)
    assert cs.rel_l2(ol.stencil_apply(drb, x), ol.stencil_apply(d, cinv_x)) < 1e-12
    # Schur: solving via prepare/solve/reconstruct satisfies the ORIGINAL system (n18:153-231): check operator identity
    half = L * L
    ye = ol.cvec(2 * half)
    ye[:half] = x[:half]
    t = ol.stencil_apply(drb, ye, ol.P_OE | ol.P_ZERO)                 # D'_oe y_e
    t2 = ol.stencil_apply(drb, t, ol.P_EO | ol.P_ZERO)                 # D'_eo D'_oe y_e
    schur = ye[:half] - t2[:half]
    # with x_o chosen so the odd rows vanish (x_o = -D'_oe y_e), the full rbjacobi operator reproduces the Schur op
    full_in = ye.copy()
    full_in[half:] = -t[half:]
    full = ol.stencil_apply(drb, full_in)
    assert cs.rel_l2(full[:half], schur) < 1e-12 and np.linalg.norm(full[half:]) < 1e-12 * np.linalg.norm(schur)


def test_reductions():
    L, nc = 8, 2
    a, b = cs.gaussian_cvec(L * L * nc, 1), cs.gaussian_cvec(L * L * nc, 2)
    assert ol.norm2sq(a) == pytest.approx(np.vdot(a, a).real, rel=1e-14)
    assert ol.dot(a, b) == pytest.approx(np.vdot(a, b), rel=1e-14)
    assert ol.diffnorm2sq(a, b) == pytest.approx(np.linalg.norm(a - b) ** 2, rel=1e-14)
    assert ol.norminf(a) == pytest.approx(np.abs(a).max(), rel=1e-15)
    grid = cs.eo_to_grid(a, L, L, nc)
    assert np.allclose(ol.norm2sq_cv_timeslice(a, L, L, nc), (np.abs(grid) ** 2).sum(axis=(0, 2)), rtol=1e-13)
    gb = cs.eo_to_grid(b, L, L, nc)
    assert np.allclose(ol.dot_cv_timeslice(a, b, L, L, nc), (np.conj(grid) * gb).sum(axis=(0, 2)), rtol=1e-13)


def test_l128_fixture_is_past_critical_at_mass_minus_007(golden_dir):
    """Evidence for profiles/r01_driver_sweep.log `n13_128_1_12 ... failed to converge`: on the reference's own 128^2
    configuration (tests/common_cfgs_u1/l128t128b60_heatbath.dat) the Wilson operator at mass -0.07 has a NEGATIVE real
    eigenvalue (-2.5e-4): this configuration's critical mass is -0.06975, inside the ensemble's -0.0706(15)
    (tests/n15_wilson_goldstone_u1_heatbath/critical_mass.txt), so mass -0.07 is past critical and the Galerkin coarse
    operator inherits an indefinite near-null mode on which restarted GCR(32) stagnates.  Computed with the independent
    coordinate-space operator (tests/coordspace.py) and scipy's shift-invert Arnoldi -- no oracle, no HIP code."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as sla
    Lq = 128
    ph = np.loadtxt(os.path.join(golden_dir, "l128t128b60_heatbath.dat"))
    Ux, Uy = cs.phases_to_links(ph, Lq, Lq)
    N = 2 * Lq * Lq
    X, Y = [a.ravel() for a in np.meshgrid(np.arange(Lq), np.arange(Lq), indexing="ij")]
    idx = lambda x, y, c: ((x % Lq) * Lq + (y % Lq)) * 2 + c
    s1 = np.array([[0, 1], [1, 0]], dtype=complex)
    s2 = np.array([[0, -1j], [1j, 0]])
    rows, cols, vals = [], [], []
    for r in range(2):
        rows.append(idx(X, Y, r)); cols.append(idx(X, Y, r)); vals.append(np.full(Lq * Lq, 2.0 - 0.07, dtype=complex))
    for mu, (U, sig) in enumerate(((Ux, s1), (Uy, s2))):
        Hp, Hm = 0.5 * (-np.eye(2) + sig), 0.5 * (-np.eye(2) - sig)
        dx, dy = (1, 0) if mu == 0 else (0, 1)
        Uf, Ub = U[X, Y], np.conj(U[(X - dx) % Lq, (Y - dy) % Lq])
        for r in range(2):
            for c in range(2):
                rows.append(idx(X, Y, r)); cols.append(idx(X + dx, Y + dy, c)); vals.append(Uf * Hp[r, c])
                rows.append(idx(X, Y, r)); cols.append(idx(X - dx, Y - dy, c)); vals.append(Ub * Hm[r, c])
    A = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N))
    v = cs.gaussian_cvec(N, 3).reshape(Lq, Lq, 2)
    assert np.linalg.norm(A @ v.reshape(-1) - cs.wilson_apply(v, Ux, Uy, -0.07).reshape(-1)) < 1e-11 * np.linalg.norm(v)
    ev = sla.eigs(A, k=4, sigma=0.0, which="LM", return_eigenvectors=False)
    real_ev = np.sort(ev[np.abs(ev.imag) < 1e-9].real)
    assert real_ev[0] < 0.0 and abs(real_ev[0] + 2.5315e-4) < 2e-6, real_ev      # past critical
    assert real_ev[1] > 0.0                                                      # indefinite


def test_n21_rbj_dagger_identity_on_the_oracle(golden_dir):
    """a11 (stencil_2d.h:1989-2060): the oracle's rbj-dagger stencil is the adjoint of its right-block-Jacobi stencil,
    <y, M_rbj x> = <M_rbj^dag y, x> (what n21 checks before its CGNE / CGNR solves), cinv^dag is the conj-transpose per site,
    and the dagger of the identity clover is the identity."""
    L = 32
    ph, _ = _fixture_links(golden_dir, L)
    clover, hopping = ol.wilson_fill(ol.phases_to_gauge_u1(ph, L, L), L, L)
    d = ol.make_desc(L, L, 2, clover, hopping, shift=-0.07 + 0.02j)
    cinv, rclover, rhopping = ol.build_rbjacobi(d)
    dcinv, dcl, dho = ol.build_rbj_dagger(cinv, rclover, rhopping, L, L, 2)
    assert np.array_equal(dcl, rclover)                                                   # identity blocks
    assert np.array_equal(dcinv.reshape(-1, 2, 2), np.conj(cinv.reshape(-1, 2, 2).transpose(0, 2, 1)))
    x, y = cs.gaussian_cvec(L * L * 2, 21), cs.gaussian_cvec(L * L * 2, 22)
    a = np.vdot(y, ol.stencil_apply(ol.make_desc(L, L, 2, rclover, rhopping), x))
    b = np.vdot(ol.stencil_apply(ol.make_desc(L, L, 2, dcl, dho), y), x)
    assert abs(a - b) / abs(a) < 1e-12


def test_redot_timeslice_and_wall_source():
    """reductions/reductions.h:47-66 and :90-162 on the oracle: redot = Re(dot) per timeslice against a coordinate-space sum; the
    wall source is real, lives on ONE timeslice and ONE component, is zero elsewhere, has the requested mean / deviation, and an
    out-of-range timeslice or colour is refused."""
    L, nc = 16, 2
    a, b = cs.gaussian_cvec(L * L * nc, 3), cs.gaussian_cvec(L * L * nc, 4)
    ga, gb = cs.eo_to_grid(a, L, L, nc), cs.eo_to_grid(b, L, L, nc)
    assert np.allclose(ol.redot_cv_timeslice(a, b, L, L, nc), (np.conj(ga) * gb).sum(axis=(0, 2)).real, rtol=1e-13)
    assert np.allclose(ol.redot_cv_timeslice(a, b, L, L, nc), ol.dot_cv_timeslice(a, b, L, L, nc).real, rtol=1e-13)
    Lx, Ly = 64, 32
    w = ol.gaussian_wall_source(Lx, Ly, nc, 5, 1, 1337, deviation=2.0, mean=0.5)
    g = cs.eo_to_grid(w, Lx, Ly, nc)                  # [x, y, c]
    assert not g.imag.any()
    mask = np.zeros(g.shape, dtype=bool)
    mask[:, 5, 1] = True
    assert not g[~mask].any() and np.all(g[mask] != 0)
    vals = np.concatenate([cs.eo_to_grid(ol.gaussian_wall_source(Lx, Ly, nc, t, 1, 7 + t, deviation=2.0, mean=0.5), Lx, Ly, nc)[:, t, 1].real for t in range(Ly)])
    assert abs(vals.mean() - 0.5) < 0.15 and abs(vals.std() - 2.0) < 0.15          # 2048 draws
    assert ol.gaussian_wall_source(Lx, Ly, nc, Ly, 0, 1) is None and ol.gaussian_wall_source(Lx, Ly, nc, 0, nc, 1) is None


def test_volume_one_lattice_is_the_shift_corner_case():
    """stencil_2d.h:870-888 on the oracle: a 1 x 1 lattice applies lhs[c] += (shift + eo_shift +- dof_shift) rhs[c] and nothing else (every
    half-volume loop of the clover / hopping passes has count volume / 2 = 0)."""
    for nc in (1, 2, 3, 8):
        rng = np.random.default_rng(nc)
        clover = (rng.normal(size=nc * nc) + 1j * rng.normal(size=nc * nc)).astype(np.complex128)
        hopping = (rng.normal(size=4 * nc * nc) + 1j * rng.normal(size=4 * nc * nc)).astype(np.complex128)
        rhs = (rng.normal(size=nc) + 1j * rng.normal(size=nc)).astype(np.complex128)
        sh, eo, ds = 0.3 - 0.1j, 0.05 + 0.02j, 0.7 + 0.4j
        d = ol.make_desc(1, 1, nc, clover, hopping, sh, eo, ds)
        fac = np.full(nc, sh + eo, dtype=np.complex128)
        if nc % 2 == 0:
            fac[:nc // 2] += ds
            fac[nc // 2:] -= ds
        out = ol.stencil_apply(d, rhs, ol.P_ALL | ol.P_ZERO)
        assert np.allclose(out, fac * rhs, rtol=1e-15)
        lhs = rhs.copy()
        ol.stencil_apply(d, rhs, ol.P_CLOVER | ol.P_HOPPING, lhs=lhs)
        assert np.array_equal(lhs, rhs)
