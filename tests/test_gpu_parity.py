"""GPU parity tests: the HIP path (through the C-ABI of libqmg_hip.so) against the CPU oracle on the
same seeded inputs and on the reference's own U(1) fixtures.

Tolerance: fp64, relative L2 <= 1e-13 for operator / transfer outputs (only summation order and
FMA contraction differ), <= 1e-12 for global reductions.  Pure data movement
(cshift, operator fills that only copy / negate / conjugate) must be bit-exact.
"""
import importlib
import os

import numpy as np
import pytest

import coordspace as cs
import oracle_lib as ol

qmg = importlib.import_module("quantum-mg_amd")

pytestmark = pytest.mark.gpu

TOL = 1e-13
RTOL_RED = 1e-12


@pytest.fixture(scope="module", autouse=True)
def _device():
    qmg.build()
    qmg.init(0)
    yield
    qmg.sync()


def D(a):
    return qmg.DeviceArray.from_host(np.ascontiguousarray(a, dtype=np.complex128))


def fixture_gauge(golden_dir, L):
    ph = np.loadtxt(os.path.join(golden_dir, "l%dt%db60_heatbath.dat" % (L, L)))
    return ol.phases_to_gauge_u1(ph, L, L)


def random_gauge(Lx, Ly, seed):
    rng = np.random.default_rng(seed)
    return ol.phases_to_gauge_u1(rng.uniform(-np.pi, np.pi, 2 * Lx * Ly), Lx, Ly)


# ------------------------------------------------------------------ cshift (a2)
@pytest.mark.parametrize("Lx,Ly", [(6, 4), (32, 24), (34, 10), (2, 2), (64, 64)])
@pytest.mark.parametrize("dof", [1, 2, 4, 9])
def test_cshift_bit_exact(Lx, Ly, dof):
    v = cs.gaussian_cvec(Lx * Ly * dof, 100 + dof)
    dv = D(v)
    for cdir in (ol.CSHIFT_XP1, ol.CSHIFT_YP1, ol.CSHIFT_XM1, ol.CSHIFT_YM1):
        for eo in (ol.EO_FROM_EVEN, ol.EO_FROM_ODD, ol.EO_FROM_EVENODD):
            sentinel = np.full(v.size, 7.0 + 3.0j)
            want = ol.cshift(v, cdir, eo, dof, Lx, Ly, lhs=sentinel.copy())
            out = D(sentinel)
            qmg.cshift(out, dv, cdir, eo, dof, Lx, Ly)
            assert np.array_equal(out.to_host(), want), (cdir, eo)
    # FROM_0 quirk (cshift_2d.h:58,147): half_size elements, same half
    sentinel = np.full(v.size, 7.0 + 3.0j)
    want = ol.cshift(v, ol.CSHIFT_FROM_0, ol.EO_FROM_EVENODD, dof, Lx, Ly, lhs=sentinel.copy())
    out = D(sentinel)
    qmg.cshift(out, dv, qmg.CSHIFT_FROM_0, qmg.EO_FROM_EVENODD, dof, Lx, Ly)
    assert np.array_equal(out.to_host(), want)


def test_cshift_rejects_bad_arguments():
    v = D(np.zeros(64))
    L = qmg.lib()
    import ctypes as C
    assert L.qmg_cshift(C.c_void_p(v.ptr), C.c_void_p(v.ptr), 2, 3, 1, 7, 4, None) == 1      # odd Lx
    assert L.qmg_cshift(C.c_void_p(v.ptr), C.c_void_p(v.ptr), 6, 3, 1, 8, 4, None) == 3      # distance-2: unsupported
    assert L.qmg_cshift(None, C.c_void_p(v.ptr), 2, 3, 1, 8, 4, None) == 1


# ------------------------------------------------------------------ operator fills (a13-a15)
@pytest.mark.parametrize("L", [32, 64])
def test_fills_match_oracle_on_reference_fixtures(golden_dir, L):
    gauge = fixture_gauge(golden_dir, L)
    dg = D(gauge)
    vol = L * L
    clover, hopping = ol.wilson_fill(gauge, L, L, 1.0)
    dc, dh = qmg.DeviceArray(4 * vol), qmg.DeviceArray(16 * vol)
    qmg.wilson_fill(dc, dh, dg, L, L, 1.0)
    assert np.array_equal(dc.to_host(), clover)
    assert cs.rel_l2(dh.to_host(), hopping) < 1e-16 or np.array_equal(dh.to_host(), hopping)
    hop = ol.staggered_fill(gauge, L, L)
    dh1 = qmg.DeviceArray(4 * vol)
    qmg.staggered_fill(dh1, dg, L, L)
    assert np.array_equal(dh1.to_host(), hop)
    clover, hop = ol.laplace_fill(gauge, L, L)
    dc1 = qmg.DeviceArray(vol)
    qmg.laplace_fill(dc1, dh1, dg, L, L)
    assert np.array_equal(dc1.to_host(), clover) and np.array_equal(dh1.to_host(), hop)


def test_wilson_fill_nonunit_coeff_ragged():
    Lx, Ly = 34, 10
    gauge = random_gauge(Lx, Ly, 5)
    clover, hopping = ol.wilson_fill(gauge, Lx, Ly, 0.7)
    dc, dh = qmg.DeviceArray(4 * Lx * Ly), qmg.DeviceArray(16 * Lx * Ly)
    qmg.wilson_fill(dc, dh, D(gauge), Lx, Ly, 0.7)
    assert cs.rel_l2(dc.to_host(), clover) < 1e-16
    assert cs.rel_l2(dh.to_host(), hopping) < 1e-15


# ------------------------------------------------------------------ stencil apply (a4-a8)
def _apply_both(Lx, Ly, nc, clover, hopping, rhs, pieces, shifts=(0, 0, 0), lhs0=None, nrhs=1):
    """Run oracle and HIP on identical inputs; returns (want, got)."""
    size = Lx * Ly * nc
    od = ol.make_desc(Lx, Ly, nc, clover, hopping, *shifts)
    want = np.zeros(size * nrhs, dtype=np.complex128) if lhs0 is None else lhs0.copy()
    for k in range(nrhs):
        ol.stencil_apply(od, np.ascontiguousarray(rhs[k * size:(k + 1) * size]), pieces, lhs=want[k * size:(k + 1) * size])
    dcl = None if clover is None else D(clover)
    dho = None if hopping is None else D(hopping)
    gd = qmg.make_desc(Lx, Ly, nc, dcl, dho, *shifts)
    dl = D(np.zeros(size * nrhs) if lhs0 is None else lhs0)
    qmg.stencil_apply(gd, dl, D(rhs), pieces, nrhs=nrhs, vec_stride=size)
    return want, dl.to_host()


@pytest.mark.parametrize("L", [32, 64])
def test_wilson_apply_on_reference_fixture(golden_dir, L):
    gauge = fixture_gauge(golden_dir, L)
    clover, hopping = ol.wilson_fill(gauge, L, L)
    rhs = cs.gaussian_cvec(L * L * 2, 1337)
    want, got = _apply_both(L, L, 2, clover, hopping, rhs, ol.P_ALL | ol.P_ZERO, (-0.07, 0, 0))
    assert cs.rel_l2(got, want) < TOL
    # and against the independent coordinate-space operator
    ph = np.loadtxt(os.path.join(golden_dir, "l%dt%db60_heatbath.dat" % (L, L)))
    Ux, Uy = cs.phases_to_links(ph, L, L)
    coord = cs.grid_to_eo(cs.wilson_apply(cs.eo_to_grid(rhs, L, L, 2), Ux, Uy, -0.07), L, L, 2)
    assert cs.rel_l2(got, coord) < TOL


def test_n02_free_laplace_known_answers_on_gpu():
    """tests/n02_free_laplace_test/free_laplace.cpp:31-100 through the HIP path."""
    Lx, Ly, msq = 32, 24, 0.1 * 0.1
    clover, hopping = ol.free_laplace_fill(Lx, Ly)
    gd = qmg.make_desc(Lx, Ly, 1, D(clover), D(hopping), msq)
    idx = lambda x, y: ol.coord_to_index(Lx, Ly, x % Lx, y % Ly)
    x0, y0 = Lx // 2, Ly // 2 + 1
    rhs = np.zeros(Lx * Ly, dtype=np.complex128)
    rhs[idx(x0, y0)] = 1.0
    dl, dr = qmg.DeviceArray.zeros(Lx * Ly), D(rhs)
    qmg.stencil_apply(gd, dl, dr, qmg.P_ALL)
    lhs = dl.to_host()
    assert lhs[idx(x0, y0)] == pytest.approx(4.01, abs=1e-15)
    for dx, dy in ((1, 0), (0, 1), (-1, 0), (0, -1)):
        assert lhs[idx(x0 + dx, y0 + dy)] == pytest.approx(-1.0, abs=1e-15)
    qmg.zero_vector(dr, Lx * Ly)
    qmg.stencil_apply(gd, dr, dl, qmg.P_ALL)
    twice = dr.to_host()
    assert twice[idx(x0, y0)] == pytest.approx(20.0801, abs=1e-13)
    assert twice[idx(x0 + 1, y0)] == pytest.approx(-8.02, abs=1e-13)
    assert twice[idx(x0 + 2, y0)] == pytest.approx(1.0, abs=1e-13)


PIECE_SETS = [
    ("all_zero", ol.P_ALL | ol.P_ZERO), ("all_accumulate", ol.P_ALL), ("clover", ol.P_CLOVER), ("hopping", ol.P_HOPPING),
    ("eo", ol.P_EO), ("oe", ol.P_OE), ("shift", ol.P_SHIFT), ("ee", ol.P_CLOVER_E | ol.P_SHIFT_E),
    ("oo_zero", ol.P_CLOVER_O | ol.P_SHIFT_O | ol.P_ZERO_O), ("eo_xp1", ol.P_EO_XP1), ("oe_ym1", ol.P_OE_XP1 << 3),
    ("dir_xm1_both", (ol.P_EO_XP1 << 2) | (ol.P_OE_XP1 << 2)), ("zero_only", ol.P_ZERO_E),
]


@pytest.mark.parametrize("name,pieces", PIECE_SETS)
@pytest.mark.parametrize("Lx,Ly,nc", [(32, 32, 2), (6, 4, 2), (34, 10, 1), (16, 12, 4), (12, 8, 3), (16, 6, 8), (8, 8, 24)])
def test_every_piece_mask_every_kernel(name, pieces, Lx, Ly, nc):
    """Random dense stencils: exercises kernel A (nc 1,2,4) and kernel B (nc 3,8,24), accumulate
    vs overwrite, untouched halves, ragged tiles, all three shifts."""
    vol = Lx * Ly
    clover = cs.gaussian_cvec(vol * nc * nc, 1)
    hopping = cs.gaussian_cvec(4 * vol * nc * nc, 2)
    rhs = cs.gaussian_cvec(vol * nc, 3)
    lhs0 = cs.gaussian_cvec(vol * nc, 4)
    shifts = (0.3 - 0.1j, 0.05 + 0.02j, -0.07j)
    want, got = _apply_both(Lx, Ly, nc, clover, hopping, rhs, pieces, shifts, lhs0)
    assert cs.rel_l2(got, want) < TOL


@pytest.mark.parametrize("Lx,Ly,nc", [(2, 2, 1), (2, 2, 2), (2, 2, 8), (2, 4, 2), (4, 2, 3), (2, 6, 24)])
def test_minimum_lattices(Lx, Ly, nc):
    """The smallest lattices the reference's cshift admits (both extents even, cshift_2d.h:62): on a 2-wide lattice the +x
    and -x neighbours are the SAME site (half-row length 1), as on the coarsest level of an n19-style hierarchy."""
    vol = Lx * Ly
    clover = cs.gaussian_cvec(vol * nc * nc, 1)
    hopping = cs.gaussian_cvec(4 * vol * nc * nc, 2)
    rhs = cs.gaussian_cvec(vol * nc, 3)
    lhs0 = cs.gaussian_cvec(vol * nc, 4)
    for pieces in (ol.P_ALL | ol.P_ZERO, ol.P_ALL, ol.P_EO, ol.P_OE | ol.P_ZERO_O, (ol.P_EO_XP1 << 2) | (ol.P_OE_XP1 << 2)):
        want, got = _apply_both(Lx, Ly, nc, clover, hopping, rhs, pieces, (0.1, 0.02, 0.0), lhs0)
        assert cs.rel_l2(got, want) < TOL, pieces


@pytest.mark.parametrize("nc", [1, 2, 3, 8])
def test_volume_one_lattice(nc):
    """lattice.h:77,201 / stencil_2d.h:870-888: on the 1 x 1 lattice every half-volume loop of the reference (clover sweeps, cshifts, hopping
    products: counts volume / 2 = 0) touches nothing, so apply_M is the shift term in its corner form -- the one site counts as even,
    lhs[c] += (shift + eo_shift +- dof_shift) rhs[c] with the dof term for even nc only.  Device against the oracle and against the formula;
    the stored matrices must not matter."""
    clover, hopping = cs.gaussian_cvec(nc * nc, 1), cs.gaussian_cvec(4 * nc * nc, 2)
    rhs, lhs0 = cs.gaussian_cvec(nc, 3), cs.gaussian_cvec(nc, 4)
    sh, eo, ds = 0.3 - 0.1j, 0.05 + 0.02j, 0.7 + 0.4j
    fac = np.full(nc, sh + eo, dtype=np.complex128)
    if nc % 2 == 0:
        fac[:nc // 2] += ds
        fac[nc // 2:] -= ds
    for pieces, want in ((ol.P_ALL | ol.P_ZERO, fac * rhs), (ol.P_ALL, lhs0 + fac * rhs), (ol.P_HOPPING | ol.P_CLOVER, lhs0), (ol.P_SHIFT | ol.P_ZERO, fac * rhs),
                         (ol.P_ZERO, np.zeros(nc))):
        w, g = _apply_both(1, 1, nc, clover, hopping, rhs, pieces, (sh, eo, ds), lhs0)
        assert np.array_equal(w, want) or cs.rel_l2(w, want) < 1e-15, pieces
        assert np.allclose(g, w, rtol=4e-16, atol=1e-300), (pieces, g, w)
    # a batch of three systems with a gap in the mask, both storage precisions
    X = cs.gaussian_cvec(3 * nc, 5)
    out = D(np.zeros(3 * nc))
    gd = qmg.make_desc(1, 1, nc, None, None, sh, eo, ds)
    qmg.stencil_apply_batch(gd, out, D(X), ol.P_ALL | ol.P_ZERO, 3, nc, 0b101)
    got = out.to_host().reshape(3, nc)
    assert np.allclose(got[0], fac * X[:nc], rtol=4e-16) and np.allclose(got[2], fac * X[2 * nc:], rtol=4e-16) and np.all(got[1] == 0)
    x32 = qmg.DeviceArray.from_host(X.astype(np.complex64))
    o32 = qmg.DeviceArray.from_host(np.zeros(3 * nc, dtype=np.complex64))
    qmg.stencil_apply_t(qmg.C32, gd, o32, x32, ol.P_ALL | ol.P_ZERO, nrhs=3, vec_stride=nc, mask=0b111)
    assert np.allclose(o32.to_host(), (np.tile(fac, 3) * X).astype(np.complex64), rtol=3e-7)


def test_zero_length_and_strided_inputs():
    """Empty inputs are accepted and do nothing; multi-RHS batches may be padded (vec_stride > size_cv)."""
    import ctypes as C
    v = qmg.DeviceArray.zeros(16)
    L = qmg.lib()
    assert L.qmg_caxpy(C.c_double(1.0), C.c_double(0.0), C.c_void_p(v.ptr), C.c_void_p(v.ptr), C.c_size_t(0), None) == 0
    assert L.qmg_zero_vector(None, C.c_size_t(0), None) == 0
    out = C.c_double(-1.0)
    assert L.qmg_norm2sq(C.c_void_p(v.ptr), C.c_size_t(0), None, C.byref(out), None) == 0 and out.value == 0.0
    assert L.qmg_multi_caxpy(None, None, 0, C.c_void_p(v.ptr), C.c_size_t(16), None) == 0
    assert not v.to_host().any()
    Lx, Ly, nc, nrhs, pad = 8, 6, 2, 3, 5
    vol = Lx * Ly
    size, stride = vol * nc, vol * nc + pad
    clover, hopping = cs.gaussian_cvec(vol * nc * nc, 1), cs.gaussian_cvec(4 * vol * nc * nc, 2)
    rhs = cs.gaussian_cvec(stride * nrhs, 3)
    sentinel = np.full(stride * nrhs, 9.0 - 2.0j)
    od = ol.make_desc(Lx, Ly, nc, clover, hopping, 0.1)
    gd = qmg.make_desc(Lx, Ly, nc, D(clover), D(hopping), 0.1)
    dl = D(sentinel)
    qmg.stencil_apply(gd, dl, D(rhs), nrhs=nrhs, vec_stride=stride)
    got = dl.to_host()
    for k in range(nrhs):
        want = ol.stencil_apply(od, np.ascontiguousarray(rhs[k * stride:k * stride + size]))
        assert cs.rel_l2(got[k * stride:k * stride + size], want) < TOL
        assert np.all(got[k * stride + size:(k + 1) * stride] == 9.0 - 2.0j)      # padding untouched


@pytest.mark.parametrize("nc", [1, 2, 8])
def test_missing_clover_or_hopping(nc):
    Lx, Ly = 16, 8
    vol = Lx * Ly
    hopping = cs.gaussian_cvec(4 * vol * nc * nc, 2)
    clover = cs.gaussian_cvec(vol * nc * nc, 1)
    rhs = cs.gaussian_cvec(vol * nc, 3)
    want, got = _apply_both(Lx, Ly, nc, None, hopping, rhs, ol.P_ALL | ol.P_ZERO, (0.04, 0, 0))   # staggered-like
    assert cs.rel_l2(got, want) < TOL
    want, got = _apply_both(Lx, Ly, nc, clover, None, rhs, ol.P_ALL | ol.P_ZERO, (0.04, 0, 0))
    assert cs.rel_l2(got, want) < TOL


@pytest.mark.parametrize("nc", [1, 2, 8])
def test_multi_rhs_shares_matrices(nc):
    Lx, Ly, nrhs = 16, 16, 5
    vol = Lx * Ly
    clover = cs.gaussian_cvec(vol * nc * nc, 1)
    hopping = cs.gaussian_cvec(4 * vol * nc * nc, 2)
    rhs = cs.gaussian_cvec(vol * nc * nrhs, 3)
    want, got = _apply_both(Lx, Ly, nc, clover, hopping, rhs, ol.P_ALL | ol.P_ZERO, (0.1, 0, 0), nrhs=nrhs)
    assert cs.rel_l2(got, want) < TOL


@pytest.mark.parametrize("nc,nrhs", [(8, 2), (8, 16), (12, 5), (16, 16), (24, 8), (24, 19), (32, 3)])
def test_multi_rhs_coarse_apply_on_matrix_cores(nc, nrhs):
    """Kernel C (v_mfma_f64_16x16x4_f64): every piece subset the facade launches, all three shifts, overwrite and
    accumulate, ragged rhs counts (19 = one full pass of 16 + 3), against the oracle applied per right-hand side."""
    Lx, Ly = 12, 6
    vol = Lx * Ly
    clover = cs.gaussian_cvec(vol * nc * nc, 1)
    hopping = cs.gaussian_cvec(4 * vol * nc * nc, 2)
    rhs = cs.gaussian_cvec(vol * nc * nrhs, 3)
    lhs0 = cs.gaussian_cvec(vol * nc * nrhs, 4)
    shifts = (0.3 - 0.2j, 0.11 + 0.05j, -0.07 + 0.02j)
    for pieces, l0 in ((ol.P_ALL | ol.P_ZERO, None), (ol.P_ALL, lhs0), (ol.P_EO | ol.P_ZERO_E, lhs0), (ol.P_OE, lhs0),
                       (ol.P_CLOVER | ol.P_SHIFT | ol.P_ZERO, None), (ol.P_HOPPING | ol.P_SHIFT | ol.P_ZERO, None), (ol.P_EO_XP1 | (ol.P_OE_XP1 << 3), lhs0)):   # +x into even, -y into odd
        want, got = _apply_both(Lx, Ly, nc, clover, hopping, rhs, pieces, shifts, lhs0=l0, nrhs=nrhs)
        assert cs.rel_l2(got, want) < TOL, hex(pieces)
    # and it is the same operator as the one-rhs-at-a-time kernel B
    qmg.set_tuning("stencil_mfma", 0)
    try:
        _, got_b = _apply_both(Lx, Ly, nc, clover, hopping, rhs, ol.P_ALL | ol.P_ZERO, shifts, nrhs=nrhs)
    finally:
        qmg.set_tuning("stencil_mfma", 1)
    _, got_c = _apply_both(Lx, Ly, nc, clover, hopping, rhs, ol.P_ALL | ol.P_ZERO, shifts, nrhs=nrhs)
    assert cs.rel_l2(got_c, got_b) < TOL
    # default mode (packed re|im columns for <= 8 rhs) vs the plain four-MFMA product
    qmg.set_tuning("stencil_mfma", 2)
    try:
        _, got_plain = _apply_both(Lx, Ly, nc, clover, hopping, rhs, ol.P_ALL | ol.P_ZERO, shifts, nrhs=nrhs)
    finally:
        qmg.set_tuning("stencil_mfma", 1)
    assert cs.rel_l2(got_plain, got_b) < TOL and cs.rel_l2(got_c, got_plain) < TOL


@pytest.mark.parametrize("nc", [1, 2, 8])
def test_in_place_eo_like_the_reference_schur(nc):
    """apply_M_rbjacobi_eo(eo_cvector, eo_cvector): reads the odd half, accumulates into the even half
    (stencil_2d.h:1904); staggered reconstruct_x does the same with oe (staggered.h:236)."""
    Lx, Ly = 16, 12
    vol = Lx * Ly
    hopping = cs.gaussian_cvec(4 * vol * nc * nc, 2)
    v = cs.gaussian_cvec(vol * nc, 3)
    od = ol.make_desc(Lx, Ly, nc, None, hopping)
    gd = qmg.make_desc(Lx, Ly, nc, None, D(hopping))
    for pieces in (ol.P_EO, ol.P_OE):
        want = v.copy()
        ol.stencil_apply(od, v.copy(), pieces, lhs=want)
        dv = D(v)
        qmg.stencil_apply(gd, dv, dv, pieces)
        assert cs.rel_l2(dv.to_host(), want) < TOL


def test_staggered_and_laplace_on_fixture(golden_dir):
    L = 64
    gauge = fixture_gauge(golden_dir, L)
    rhs = cs.gaussian_cvec(L * L, 1337)
    hop = ol.staggered_fill(gauge, L, L)
    want, got = _apply_both(L, L, 1, None, hop, rhs, ol.P_ALL | ol.P_ZERO, (0.04, 0, 0))
    assert cs.rel_l2(got, want) < TOL
    clover, hop = ol.laplace_fill(gauge, L, L)
    want, got = _apply_both(L, L, 1, clover, hop, rhs, ol.P_ALL | ol.P_ZERO, (0.01, 0, 0))
    assert cs.rel_l2(got, want) < TOL


def test_apply_rejects_bad_arguments():
    import ctypes as C
    v = qmg.DeviceArray.zeros(64)
    d = qmg.make_desc(7, 4, 1, v, v)
    assert qmg.lib().qmg_stencil_apply(C.byref(d), C.c_void_p(v.ptr), C.c_void_p(v.ptr), 0xFFF, 1, C.c_size_t(0), None) == 1
    d = qmg.make_desc(8, 4, 1, v, v)
    assert qmg.lib().qmg_stencil_apply(C.byref(d), None, C.c_void_p(v.ptr), 0xFFF, 1, C.c_size_t(0), None) == 1
    assert qmg.lib().qmg_stencil_apply(C.byref(d), C.c_void_p(v.ptr), C.c_void_p(v.ptr), 0xFFF, 2, C.c_size_t(3), None) == 1


# ------------------------------------------------------------------ stencil variants (a9-a11)
@pytest.mark.parametrize("Lx,Ly,nc", [(32, 32, 2), (12, 8, 3), (8, 8, 8)])
def test_dagger_and_rbjacobi_builds(Lx, Ly, nc):
    vol = Lx * Ly
    clover = cs.gaussian_cvec(vol * nc * nc, 1) + 4.0 * np.tile(np.eye(nc).reshape(-1), vol)
    hopping = cs.gaussian_cvec(4 * vol * nc * nc, 2)
    dc, dh = ol.build_dagger(clover, hopping, Lx, Ly, nc)
    gdc, gdh = qmg.DeviceArray(clover.size), qmg.DeviceArray(hopping.size)
    qmg.build_dagger(gdc, gdh, D(clover), D(hopping), Lx, Ly, nc)
    assert np.array_equal(gdc.to_host(), dc) and np.array_equal(gdh.to_host(), dh)
    shifts = (0.2 + 0.1j, 0.03, 0.05 if nc % 2 == 0 else 0.0)
    od = ol.make_desc(Lx, Ly, nc, clover, hopping, *shifts)
    cinv, rclover, rhopping = ol.build_rbjacobi(od)
    gcl, gho = D(clover), D(hopping)
    gd = qmg.make_desc(Lx, Ly, nc, gcl, gho, *shifts)
    gcinv, grc, grh = qmg.DeviceArray(clover.size), qmg.DeviceArray(clover.size), qmg.DeviceArray(hopping.size)
    qmg.build_rbjacobi(gcinv, grc, grh, gd)
    assert cs.rel_l2(gcinv.to_host(), cinv) < 1e-12
    assert np.array_equal(grc.to_host(), rclover)
    assert cs.rel_l2(grh.to_host(), rhopping) < 1e-12
    # <y, M x> = <M^dag y, x> on the device (n17)
    x, y = cs.gaussian_cvec(vol * nc, 5), cs.gaussian_cvec(vol * nc, 6)
    dx, dy, t = D(x), D(y), qmg.DeviceArray(vol * nc)
    qmg.stencil_apply(gd, t, dx)
    a = qmg.dot(dy, t, vol * nc)
    gdd = qmg.make_desc(Lx, Ly, nc, gdc, gdh, *[np.conj(s) for s in shifts])
    qmg.stencil_apply(gdd, t, dy)
    b = qmg.dot(t, dx, vol * nc)
    assert abs(a - b) / abs(a) < 1e-12


@pytest.mark.parametrize("Lx,Ly,nc", [(32, 32, 2), (12, 8, 3), (8, 8, 8)])
def test_rbj_dagger_build_and_normal_operators(Lx, Ly, nc):
    """a11 (stencil_2d.h:1989-2060, 2282-2411): the dagger of the right-block-Jacobi stencil -- conj-transposed cinv and (identity)
    clover, hopping daggered with the neighbour shift -- against the oracle's restatement, bit for bit (pure data movement); then the two
    normal operators the CGNE / CGNR smoothers apply, M_rbj^dag M_rbj (MDM, :2282-2299) and M_rbj M_rbj^dag (MMD, :2354-2371), as the
    facade launches them (identity clover as a unit shift, hops from the built arrays) against the oracle applying the stored matrices
    pass by pass, and <y, M_rbj x> = <M_rbj^dag y, x> (n21)."""
    vol = Lx * Ly
    clover = cs.gaussian_cvec(vol * nc * nc, 11) + 4.0 * np.tile(np.eye(nc).reshape(-1), vol)
    hopping = cs.gaussian_cvec(4 * vol * nc * nc, 12)
    shifts = (0.2 + 0.1j, 0.03, 0.05 if nc % 2 == 0 else 0.0)
    cinv, rclover, rhopping = ol.build_rbjacobi(ol.make_desc(Lx, Ly, nc, clover, hopping, *shifts))
    want_cinv, want_cl, want_ho = ol.build_rbj_dagger(cinv, rclover, rhopping, Lx, Ly, nc)
    # device: the same two entry points Stencil2D::build_rbj_dagger_stencil calls
    g_cinv, g_rcl, g_rho = D(cinv), D(rclover), D(rhopping)
    d_cinv, d_cl, d_ho = qmg.DeviceArray(cinv.size), qmg.DeviceArray(rclover.size), qmg.DeviceArray(rhopping.size)
    qmg.build_dagger(d_cl, d_ho, g_rcl, g_rho, Lx, Ly, nc)
    qmg.cmat_conjtrans(d_cinv, g_cinv, vol, nc)
    assert np.array_equal(d_cinv.to_host(), want_cinv)
    assert np.array_equal(d_cl.to_host(), want_cl)
    assert np.array_equal(d_ho.to_host(), want_ho)
    # normal operators: oracle applies clover (= identity matrices) + hopping of each stored stencil; device: unit shift + hops
    n = vol * nc
    x, y = cs.gaussian_cvec(n, 13), cs.gaussian_cvec(n, 14)
    o_rbj = ol.make_desc(Lx, Ly, nc, rclover, rhopping, 0.0)
    o_dag = ol.make_desc(Lx, Ly, nc, want_cl, want_ho, 0.0)
    g_rbj = qmg.make_desc(Lx, Ly, nc, None, g_rho, 1.0)
    g_dag = qmg.make_desc(Lx, Ly, nc, None, d_ho, 1.0)
    pieces = qmg.P_HOPPING | qmg.P_SHIFT | qmg.P_ZERO
    dx, t, u = D(x), qmg.DeviceArray(n), qmg.DeviceArray(n)
    qmg.stencil_apply(g_rbj, t, dx, pieces); qmg.stencil_apply(g_dag, u, t, pieces)          # MDM
    want = ol.stencil_apply(o_dag, ol.stencil_apply(o_rbj, x))
    assert cs.rel_l2(u.to_host(), want) < TOL
    qmg.stencil_apply(g_dag, t, dx, pieces); qmg.stencil_apply(g_rbj, u, t, pieces)          # MMD
    want = ol.stencil_apply(o_rbj, ol.stencil_apply(o_dag, x))
    assert cs.rel_l2(u.to_host(), want) < TOL
    # reconstruct_M_rbjacobi_MMD (:2373-2392): x = cinv . M_rbj^dag y
    qmg.stencil_apply(g_dag, t, D(y), pieces)
    qmg.stencil_apply(qmg.make_desc(Lx, Ly, nc, g_cinv, None, 0.0), u, t, qmg.P_CLOVER | qmg.P_ZERO)
    want = ol.stencil_apply(ol.make_desc(Lx, Ly, nc, cinv, None, 0.0), ol.stencil_apply(o_dag, y), ol.P_CLOVER | ol.P_ZERO)
    assert cs.rel_l2(u.to_host(), want) < TOL
    # adjointness (n21)
    dy = D(y)
    qmg.stencil_apply(g_rbj, t, dx, pieces)
    a = qmg.dot(dy, t, n)
    qmg.stencil_apply(g_dag, t, dy, pieces)
    b = qmg.dot(t, dx, n)
    assert abs(a - b) / abs(a) < 1e-12


# ------------------------------------------------------------------ BLAS-1 and reductions (a21)
def test_blas1_leaves():
    n = 100003
    x, y = cs.gaussian_cvec(n, 1), cs.gaussian_cvec(n, 2)
    a, b = 0.3 - 1.1j, -0.8 + 0.25j
    dx, dy, dz = D(x), D(y), qmg.DeviceArray(n)
    qmg.caxpy(a, dx, dy, n); assert cs.rel_l2(dy.to_host(), y + a * x) < 1e-15
    dy.upload(y); qmg.cxpay(dx, a, dy, n); assert cs.rel_l2(dy.to_host(), x + a * y) < 1e-15
    dy.upload(y); qmg.caxpby(a, dx, b, dy, n); assert cs.rel_l2(dy.to_host(), a * x + b * y) < 1e-15
    dy.upload(y); qmg.caxpbyz(a, dx, b, dy, dz, n); assert cs.rel_l2(dz.to_host(), a * x + b * y) < 1e-15
    qmg.cxpyz(dx, dy, dz, n); assert np.array_equal(dz.to_host(), x + y)
    qmg.cxpy(dx, dy, n); assert np.array_equal(dy.to_host(), x + y)
    qmg.caxy(a, dx, dz, n); assert cs.rel_l2(dz.to_host(), a * x) < 1e-15
    qmg.cax(b, dz, n); assert cs.rel_l2(dz.to_host(), a * b * x) < 1e-15
    qmg.copy_vector(dz, dx, n); assert np.array_equal(dz.to_host(), x)
    qmg.zero_vector(dz, n); assert not dz.to_host().any()
    # gamma5 (wilson.h:83-93) and sigma1 (:138-143) as patterns
    nsite = 5000
    v = cs.gaussian_cvec(2 * nsite, 9)
    dv, dw = D(v), qmg.DeviceArray(2 * nsite)
    qmg.caxy_pattern([1.0, -1.0], [0, 1], dv, dw, nsite)
    assert np.array_equal(dw.to_host(), v * np.tile([1.0, -1.0], nsite))
    qmg.caxy_pattern([1.0, 1.0], [1, 0], dv, dw, nsite)
    assert np.array_equal(dw.to_host(), v.reshape(-1, 2)[:, ::-1].reshape(-1))


@pytest.mark.parametrize("n", [1, 63, 4096, 1000003])
def test_global_reductions(n):
    x, y = cs.gaussian_cvec(n, 1), cs.gaussian_cvec(n, 2)
    dx, dy = D(x), D(y)
    assert qmg.norm2sq(dx, n) == pytest.approx(ol.norm2sq(x), rel=RTOL_RED)
    assert qmg.diffnorm2sq(dx, dy, n) == pytest.approx(ol.diffnorm2sq(x, y), rel=RTOL_RED)
    assert qmg.norminf(dx, n) == pytest.approx(ol.norminf(x), rel=1e-15)
    want = ol.dot(x, y)
    assert abs(qmg.dot(dx, dy, n) - want) <= RTOL_RED * np.sqrt(ol.norm2sq(x) * ol.norm2sq(y))
    # deterministic: two runs are bit-identical
    assert qmg.dot(dx, dy, n) == qmg.dot(dx, dy, n)


@pytest.mark.parametrize("k", [1, 2, 3, 7, 32])
def test_multidot(k):
    n = 50021
    xs = [cs.gaussian_cvec(n, 10 + i) for i in range(k)]
    y = cs.gaussian_cvec(n, 99)
    got = qmg.multidot([D(x) for x in xs], D(y), n)
    want = np.array([ol.dot(x, y) for x in xs])
    assert np.max(np.abs(got - want)) <= RTOL_RED * n


@pytest.mark.parametrize("k", [1, 5, 32, 40])
def test_multi_caxpy(k):
    n = 30011
    xs = [cs.gaussian_cvec(n, 10 + i) for i in range(k)]
    a = cs.gaussian_cvec(k, 77)
    y = cs.gaussian_cvec(n, 99)
    dy = D(y)
    qmg.multi_caxpy(list(a), [D(x) for x in xs], dy, n)
    want = y + sum(ai * xi for ai, xi in zip(a, xs))
    assert cs.rel_l2(dy.to_host(), want) < 1e-14


def test_timeslice_reductions():
    Lx, Ly, nc = 16, 12, 2
    a, b = cs.gaussian_cvec(Lx * Ly * nc, 1), cs.gaussian_cvec(Lx * Ly * nc, 2)
    assert np.allclose(qmg.norm2sq_cv_timeslice(D(a), Lx, Ly, nc), ol.norm2sq_cv_timeslice(a, Lx, Ly, nc), rtol=1e-13)
    assert np.allclose(qmg.dot_cv_timeslice(D(a), D(b), Lx, Ly, nc), ol.dot_cv_timeslice(a, b, Lx, Ly, nc), rtol=1e-12, atol=1e-12)


def test_gaussian_fill_statistics():
    n = 1 << 20
    d = qmg.DeviceArray(n)
    qmg.gaussian(d, n, 1337)
    v = d.to_host()
    assert abs(v.real.mean()) < 5e-3 and abs(v.imag.mean()) < 5e-3
    assert abs(v.real.var() - 1.0) < 1e-2 and abs(v.imag.var() - 1.0) < 1e-2
    d2 = qmg.DeviceArray(n)
    qmg.gaussian(d2, n, 1337)
    assert np.array_equal(d2.to_host(), v)


# ------------------------------------------------------------------ transfer (a17-a20) and coarse build (a16)
XFER_CASES = [((16, 16, 2), (4, 4, 8)), ((16, 12, 2), (4, 6, 4)), ((8, 8, 8), (2, 2, 6)), ((16, 8, 1), (4, 4, 2)),
              ((12, 8, 2), (4, 2, 4))]   # last: bx = 3 (odd) -> generic restrict


@pytest.mark.parametrize("fdims,cdims", XFER_CASES)
def test_prolong_restrict(fdims, cdims):
    fsize, csize = fdims[0] * fdims[1] * fdims[2], cdims[0] * cdims[1] * cdims[2]
    nvec = cdims[2]
    nv = cs.gaussian_cvec(nvec * fsize, 1)
    coarse, fine = cs.gaussian_cvec(csize, 2), cs.gaussian_cvec(fsize, 3)
    fine0, coarse0 = cs.gaussian_cvec(fsize, 4), cs.gaussian_cvec(csize, 5)
    dnv = D(nv)
    want = ol.prolong(nv, coarse, fdims, cdims, fine=fine0.copy())
    df = D(fine0)
    qmg.prolong(dnv, nvec, D(coarse), df, fdims, cdims)
    assert cs.rel_l2(df.to_host(), want) < TOL
    want = ol.restrict(nv, fine, fdims, cdims, coarse=coarse0.copy())
    dc = D(coarse0)
    qmg.restrict(dnv, nvec, D(fine), dc, fdims, cdims)
    assert cs.rel_l2(dc.to_host(), want) < TOL
    # single-vector form used by block-ortho (nvec = 1 < cnc): only colour 0 of each coarse site changes
    want = ol.restrict(nv, fine, fdims, cdims, coarse=coarse0.copy(), nvec=1)
    dc = D(coarse0)
    qmg.restrict(dnv, 1, D(fine), dc, fdims, cdims)
    assert cs.rel_l2(dc.to_host(), want) < TOL


@pytest.mark.parametrize("fdims,cdims", XFER_CASES[:4])
def test_block_orthonormalize_and_n05_identities(fdims, cdims):
    fsize, csize = fdims[0] * fdims[1] * fdims[2], cdims[0] * cdims[1] * cdims[2]
    nvec = cdims[2]
    nv = cs.gaussian_cvec(nvec * fsize, 1)
    chol = np.zeros(cdims[0] * cdims[1] * nvec * nvec, dtype=np.complex128)
    want = ol.block_orthonormalize(nv.copy(), fdims, cdims, cholesky=chol)
    dnv, dchol = D(nv), qmg.DeviceArray.zeros(chol.size)
    qmg.block_orthonormalize(dnv, nvec, fdims, cdims[0], cdims[1], cholesky=dchol)
    assert cs.rel_l2(dnv.to_host(), want) < 1e-12
    assert cs.rel_l2(dchol.to_host(), chol) < 1e-12
    qmg.block_orthonormalize(dnv, nvec, fdims, cdims[0], cdims[1])          # second pass (transfer.h:160-174)
    # n05: P^dag P = 1 on the coarse space
    vc = cs.gaussian_cvec(csize, 7)
    df, dc = qmg.DeviceArray.zeros(fsize), qmg.DeviceArray.zeros(csize)
    qmg.prolong(dnv, nvec, D(vc), df, fdims, cdims)
    qmg.restrict(dnv, nvec, df, dc, fdims, cdims)
    assert cs.rel_l2(dc.to_host(), vc) < 1e-13


@pytest.mark.parametrize("fdims,cdims", [((32, 32, 2), (8, 8, 24)), ((16, 16, 24), (4, 4, 24)), ((16, 8, 8), (4, 2, 8)), ((8, 8, 2), (4, 4, 6)), ((12, 12, 2), (4, 4, 4))])
def test_block_local_setup_kernels_match_the_oracle_and_the_full_lattice_passes(fdims, cdims):
    """csrc/qmg_setup.hip: block orthonormalisation with the tile in LDS, BOTH passes of the TransferMG
    constructor in one launch (32-, 128- and 256-thread groups: nel = 32, 128, 384), and the Galerkin build as per-block
    products, with a separate restrictor -- against the oracle (1e-12) and against the first round's full-lattice passes
    ("setup_fused" 0; also the fallback for the odd block width of the last case)."""
    fLx, fLy, fnc = fdims
    fvol, fsize = fLx * fLy, fLx * fLy * fnc
    nvec = cdims[2]
    nv = cs.gaussian_cvec(nvec * fsize, 21)
    ccm = cdims[0] * cdims[1] * nvec * nvec
    chol = np.zeros(ccm, dtype=np.complex128)
    want = ol.block_orthonormalize(nv.copy(), fdims, cdims, cholesky=chol)
    want = ol.block_orthonormalize(want, fdims, cdims)                       # second pass, no factor
    got = {}
    for fused in (1, 0):
        qmg.set_tuning("setup_fused", fused)
        dnv, dchol = D(nv), qmg.DeviceArray.zeros(ccm)
        qmg.check(qmg.lib().qmg_block_orthonormalize_n(qmg._vp(dnv), nvec, fLx, fLy, fnc, cdims[0], cdims[1], qmg._vp(dchol), 2, None))
        got[fused] = (dnv.to_host(), dchol.to_host())
        assert cs.rel_l2(got[fused][0], want) < 1e-12 and cs.rel_l2(got[fused][1], chol) < 1e-12, fused
    assert cs.rel_l2(got[1][0], got[0][0]) < 1e-12
    # Galerkin build, restrictor != prolongator
    clover, hopping = cs.gaussian_cvec(fvol * fnc * fnc, 1), cs.gaussian_cvec(4 * fvol * fnc * fnc, 2)
    pv = want
    rv = want + 0.2 * cs.gaussian_cvec(nvec * fsize, 22)
    od = ol.make_desc(fLx, fLy, fnc, clover, hopping, 0.1)
    for rvecs in (None, rv):
        cclover, chopping = ol.coarse_build(od, pv, cdims, restrict_vecs=rvecs)
        dcl, dho, dpv = D(clover), D(hopping), D(pv)
        drv = None if rvecs is None else D(rvecs)
        gd = qmg.make_desc(fLx, fLy, fnc, dcl, dho, 0.1)
        for fused in (1, 0):
            qmg.set_tuning("setup_fused", fused)
            gcc, gch = qmg.DeviceArray(cclover.size), qmg.DeviceArray(chopping.size)
            qmg.coarse_build(gcc, gch, gd, dpv, cdims, restrict_vecs=drv)
            assert cs.rel_l2(gcc.to_host(), cclover) < 1e-12 and cs.rel_l2(gch.to_host(), chopping) < 1e-12, (fused, rvecs is None)
    qmg.set_tuning("setup_fused", 1)


@pytest.mark.parametrize("fdims,cdims", XFER_CASES[:4])
def test_block_bi_orthonormalize_asymmetric_transfer(fdims, cdims):
    """transfer.h:610-769 (P != R^dag, tests/n05_prolong_restrict_test:105-139): after bi-orthonormalisation R^dag P = 1
    on the coarse space; L and U reproduce the block Gram matrix R^dag P of the ORIGINAL vectors."""
    fsize, csize = fdims[0] * fdims[1] * fdims[2], cdims[0] * cdims[1] * cdims[2]
    nvec = cdims[2]
    pv = cs.gaussian_cvec(nvec * fsize, 1)
    rv = pv + 0.3 * cs.gaussian_cvec(nvec * fsize, 2)          # restrictor vectors close to, but not equal to, the prolongator's
    cm = cdims[0] * cdims[1] * nvec * nvec
    Lh, Uh = np.zeros(cm, dtype=np.complex128), np.zeros(cm, dtype=np.complex128)
    wp, wr = ol.block_bi_orthonormalize(pv.copy(), rv.copy(), fdims, cdims, Lh, Uh)
    dp, dr, dL, dU = D(pv), D(rv), qmg.DeviceArray.zeros(cm), qmg.DeviceArray.zeros(cm)
    qmg.block_bi_orthonormalize(dp, dr, nvec, fdims, cdims[0], cdims[1], dL, dU)
    assert cs.rel_l2(dp.to_host(), wp) < 1e-11 and cs.rel_l2(dr.to_host(), wr) < 1e-11
    assert cs.rel_l2(dL.to_host(), Lh) < 1e-11 and cs.rel_l2(dU.to_host(), Uh) < 1e-11
    qmg.block_bi_orthonormalize(dp, dr, nvec, fdims, cdims[0], cdims[1])          # second pass, as the constructor does
    vc = cs.gaussian_cvec(csize, 7)
    df, dc = qmg.DeviceArray.zeros(fsize), qmg.DeviceArray.zeros(csize)
    qmg.prolong(dp, nvec, D(vc), df, fdims, cdims)
    qmg.restrict(dr, nvec, df, dc, fdims, cdims)
    assert cs.rel_l2(dc.to_host(), vc) < 1e-12
    # L U = block Gram matrix of the original vectors: G[s][i][j] = sum_{k in block s} conj(r_i[k]) p_j[k]
    m = ol.transfer_build_map(fdims[0], fdims[1], fdims[2], cdims[0], cdims[1])
    P0, R0 = pv.reshape(nvec, fsize), rv.reshape(nvec, fsize)
    Lg, Ug = dL.to_host().reshape(-1, nvec, nvec), dU.to_host().reshape(-1, nvec, nvec)
    for s_ in range(0, m.shape[0], max(1, m.shape[0] // 5)):
        G = np.conj(R0[:, m[s_]]) @ P0[:, m[s_]].T
        assert np.allclose(Lg[s_] @ Ug[s_], G, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("fdims,cdims", [((16, 16, 2), (4, 4, 4)), ((8, 8, 4), (4, 2, 6)), ((16, 8, 1), (4, 4, 2))])
def test_coarse_build_matches_oracle_and_n08_galerkin(fdims, cdims):
    fLx, fLy, fnc = fdims
    fvol, fsize, csize = fLx * fLy, fLx * fLy * fnc, cdims[0] * cdims[1] * cdims[2]
    nvec = cdims[2]
    clover = cs.gaussian_cvec(fvol * fnc * fnc, 1)
    hopping = cs.gaussian_cvec(4 * fvol * fnc * fnc, 2)
    nv = ol.block_orthonormalize(cs.gaussian_cvec(nvec * fsize, 3), fdims, cdims)
    od = ol.make_desc(fLx, fLy, fnc, clover, hopping, 0.1)
    cclover, chopping = ol.coarse_build(od, nv, cdims)
    dcl, dho, dnv = D(clover), D(hopping), D(nv)
    gd = qmg.make_desc(fLx, fLy, fnc, dcl, dho, 0.1)
    gcc, gch = qmg.DeviceArray(cclover.size), qmg.DeviceArray(chopping.size)
    qmg.coarse_build(gcc, gch, gd, dnv, cdims)
    assert cs.rel_l2(gcc.to_host(), cclover) < 1e-12
    assert cs.rel_l2(gch.to_host(), chopping) < 1e-12
    # n08 (distance1_build_test.cpp:117-147): coarse apply == restrict . fine apply . prolong
    vc = cs.gaussian_cvec(csize, 9)
    dvc, tf, taf = D(vc), qmg.DeviceArray.zeros(fsize), qmg.DeviceArray(fsize)
    emul, direct = qmg.DeviceArray.zeros(csize), qmg.DeviceArray(csize)
    qmg.prolong(dnv, nvec, dvc, tf, fdims, cdims)
    qmg.stencil_apply(gd, taf, tf)
    qmg.restrict(dnv, nvec, taf, emul, fdims, cdims)
    gcd = qmg.make_desc(cdims[0], cdims[1], cdims[2], gcc, gch, 0.1)    # shift copied, coarse.h:131
    qmg.stencil_apply(gcd, direct, dvc)
    assert cs.rel_l2(direct.to_host(), emul.to_host()) < 1e-12


def test_coarse_apply_beyond_2_31_matrix_elements_spot_checked():
    """Maximum sizes: 1024^2 with nc = 24 has 4 * 1024^2 * 576 = 2.4e9 hopping elements -- where the reference's `int
    size_hopping` overflows (lattice.h:23,40).  48 GB of random matrices are generated on the device; the output of one
    apply (kernel B), and of a 5-rhs batch (kernel C, f64 MFMA), is checked at sampled sites -- first, last, row ends,
    random -- against a host computation from the rows of the five matrices and the neighbour vectors fetched for those
    sites.  Size-independent, exact to 1e-13."""
    L, nc = 1024, 24
    vol, hr, half = L * L, L // 2, L * L // 2
    nc2 = nc * nc
    cl = qmg.DeviceArray(vol * nc2); qmg.gaussian(cl, vol * nc2, 11)
    ho = qmg.DeviceArray(4 * vol * nc2); qmg.gaussian(ho, 4 * vol * nc2, 12)
    nrhs = 5
    x = qmg.DeviceArray(nrhs * vol * nc); qmg.gaussian(x, nrhs * vol * nc, 13)
    y1 = qmg.DeviceArray(vol * nc)
    yb = qmg.DeviceArray(nrhs * vol * nc)
    shift = 0.3 - 0.2j
    d = qmg.make_desc(L, L, nc, cl, ho, shift)
    qmg.stencil_apply(d, y1, x, qmg.P_ALL | qmg.P_ZERO)
    qmg.stencil_apply_batch(d, yb, x, qmg.P_ALL | qmg.P_ZERO, nrhs, vol * nc, (1 << nrhs) - 1)
    rng = np.random.default_rng(5)
    samples = [(0, 0, 0), (1, L - 1, hr - 1), (0, L - 1, 0), (1, 0, hr - 1), (0, 511, 255), (1, 512, 256)]
    samples += [(int(rng.integers(2)), int(rng.integers(L)), int(rng.integers(hr))) for _ in range(10)]
    for p, y, j in samples:
        site = p * half + y * hr + j
        s = (y + p) & 1
        opp = (1 - p) * half
        nb = [site, opp + y * hr + (j + s) % hr, opp + ((y + 1) % L) * hr + j, opp + y * hr + (j + s - 1) % hr, opp + ((y - 1) % L) * hr + j]
        mats = [cl.read(site * nc2, nc2).reshape(nc, nc)] + [ho.read(mu * vol * nc2 + site * nc2, nc2).reshape(nc, nc) for mu in range(4)]
        for k in range(nrhs):
            vecs = [x.read(k * vol * nc + n * nc, nc) for n in nb]
            want = sum(m @ v for m, v in zip(mats, vecs)) + shift * vecs[0]
            got_b = yb.read(k * vol * nc + site * nc, nc)
            assert cs.rel_l2(got_b, want) < TOL, (p, y, j, k)
            if k == 0:
                assert cs.rel_l2(y1.read(site * nc, nc), want) < TOL, (p, y, j)
    for a in (cl, ho, x, y1, yb):
        a.free()


def test_wilson_apply_8192_periodic_image(golden_dir):
    """Maximum sizes for the fine operator: 8192^2 (67M sites, 17 GiB of hopping matrices).  The 64-periodic gauge field
    with a 64-periodic right-hand side must give the 64-periodic image of the oracle's 64^2 result (bench.py's gate)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    fixture = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    wl = bench.Workload(qmg, 8192, fixture, seed=7)
    try:
        assert wl.parity_gate(fixture) < TOL
    finally:
        wl.free()


def test_transfer_at_4096_spot_checked():
    """Maximum sizes for the transfer: 4096^2 (nc 2) -> 1024^2 (nc 24): 24 null vectors of 33.5M complex (12.9 GB, byte
    offsets far beyond 2^32).  prolong and restrict outputs are checked at sampled sites against a host computation from
    the fetched null-vector entries (blocks = 4x4 fine sites, transfer.h:391-395)."""
    fL, cL, fnc, cnc = 4096, 1024, 2, 24
    fsize, csize = fL * fL * fnc, cL * cL * cnc
    nv = qmg.DeviceArray(cnc * fsize); qmg.gaussian(nv, cnc * fsize, 21)
    cv = qmg.DeviceArray(csize); qmg.gaussian(cv, csize, 22)
    fv = qmg.DeviceArray(fsize); qmg.gaussian(fv, fsize, 23)
    fout = qmg.DeviceArray.zeros(fsize)
    cout = qmg.DeviceArray.zeros(csize)
    fd, cd = (fL, fL, fnc), (cL, cL, cnc)
    qmg.prolong(nv, cnc, cv, fout, fd, cd)
    qmg.restrict(nv, cnc, fv, cout, fd, cd)
    fidx = lambda x, y: ((y + ((x + y) & 1) * fL) * (fL // 2) + x // 2)
    cidx = lambda x, y: ((y + ((x + y) & 1) * cL) * (cL // 2) + x // 2)
    rng = np.random.default_rng(9)
    coarse_sites = [(0, 0), (cL - 1, cL - 1), (cL - 1, 0), (511, 512)] + [(int(rng.integers(cL)), int(rng.integers(cL))) for _ in range(3)]
    for cx, cy in coarse_sites:
        ci = cidx(cx, cy)
        cvals = cv.read(ci * cnc, cnc)
        # restrict: coarse[ci, d] = sum over the 4x4 block of conj(null[d][e]) fine[e]
        want = np.zeros(cnc, dtype=np.complex128)
        block = [(4 * cx + dx, 4 * cy + dy) for dy in range(4) for dx in range(4)]
        for (x, y) in block:
            e0 = fidx(x, y) * fnc
            f = fv.read(e0, fnc)
            for dd in range(cnc):
                want[dd] += np.vdot(nv.read(dd * fsize + e0, fnc), f)
        assert cs.rel_l2(cout.read(ci * cnc, cnc), want) < TOL, (cx, cy)
        # prolong at two fine sites of the block: fine[e] = sum_d null[d][e] coarse[ci, d]
        for (x, y) in (block[0], block[-1]):
            e0 = fidx(x, y) * fnc
            wantf = sum(nv.read(dd * fsize + e0, fnc) * cvals[dd] for dd in range(cnc))
            assert cs.rel_l2(fout.read(e0, fnc), wantf) < TOL, (x, y)
    for a in (nv, cv, fv, fout, cout):
        a.free()


def test_blas_non_temporal_reads_change_no_bit():
    """Vectors of `blas_nt_mb` MiB and more are streamed with non-temporal loads on their read-only operands (csrc/qmg_blas.hip): a cache hint,
    so every BLAS-1 result and every reduction must be the same bits with the threshold at 1 MiB (on for these 4.8 MB vectors) and at 0 (never)."""
    n = 300001
    x, y, z0 = cs.gaussian_cvec(n, 41), cs.gaussian_cvec(n, 42), cs.gaussian_cvec(n, 43)
    dx, dy = D(x), D(y)
    res = {}
    try:
        for mb in (0, 1):
            qmg.set_tuning("blas_nt_mb", mb)
            out = []
            for fn in (lambda z: qmg.caxpy(0.3 - 0.2j, dx, z, n), lambda z: qmg.cxpay(dx, 0.7 + 0.1j, z, n), lambda z: qmg.caxpbyz(0.3, dx, -1.1j, dy, z, n),
                       lambda z: qmg.cxpy(dx, z, n), lambda z: qmg.multi_caxpy([0.1, 0.2j], [dx, dy], z, n)):
                dz = D(z0)
                fn(dz)
                out.append(dz.to_host())
            out.append(np.array([qmg.norm2sq(dx, n), qmg.diffnorm2sq(dx, dy, n), qmg.norminf(dx, n)]))
            out.append(np.array([qmg.dot(dx, dy, n)] + list(qmg.multidot([dx, dy, D(z0)], dy, n))))
            res[mb] = out
    finally:
        qmg.set_tuning("blas_nt_mb", 256)
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
