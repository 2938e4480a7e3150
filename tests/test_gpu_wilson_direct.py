"""The Wilson operator applied straight from the gauge links (csrc/qmg_wilson.hip, kernel W) against the stored-stencil path
(qmg_wilson_fill + qmg_stencil_apply: operators/wilson.h:153-209, stencil_2d.h:912-936) and against the CPU oracle.

fp64: the matrix entries are formed by the same multiplications the fill kernel does and enter the same FMA sequence as the
site kernel, so the results must be IDENTICAL BIT FOR BIT to the stored path through that kernel -- for every piece set the
kernel serves, with shifts, accumulating, in batches with masks, and on y-slabs with halos.  fp32: 2e-6 of the fp64 oracle
(the stored fp32 matrices are rounded fp64 products, the direct ones fp32 products: same accuracy, different bits)."""
import importlib

import numpy as np
import pytest

import coordspace as cs
import oracle_lib as ol

qmg = importlib.import_module("quantum-mg_amd")
pytestmark = pytest.mark.gpu
D = qmg.DeviceArray.from_host
P = qmg


@pytest.fixture(scope="module", autouse=True)
def _device():
    qmg.build()
    qmg.init(0)
    qmg.set_tuning("stencil_site", 7)       # the stored path through the site kernel: the bit-for-bit twin
    yield
    qmg.set_tuning("stencil_site", 3)
    qmg.sync()


def gauge(Lx, Ly, seed):
    rng = np.random.default_rng(seed)
    return np.exp(1j * rng.uniform(-np.pi, np.pi, size=2 * Lx * Ly))


SERVED = [P.P_ALL | P.P_ZERO, P.P_ALL, P.P_CLOVER | P.P_HOPPING | P.P_ZERO, P.P_EO | P.P_ZERO_E, P.P_OE | P.P_ZERO_O, P.P_EO, P.P_HOPPING | P.P_ZERO,
          P.P_CLOVER_E | P.P_EO | P.P_SHIFT_E | P.P_ZERO_E, P.P_CLOVER_O | P.P_OE | P.P_ZERO_O]


@pytest.mark.parametrize("Lx,Ly", [(16, 16), (24, 10), (130, 6)])
def test_direct_apply_is_bit_identical_to_the_stored_stencil_in_fp64(Lx, Ly):
    n = 2 * Lx * Ly
    g = D(gauge(Lx, Ly, 3))
    cl, hp = qmg.DeviceArray(4 * Lx * Ly), qmg.DeviceArray(16 * Lx * Ly)
    w = 0.9
    qmg.wilson_fill(cl, hp, g, Lx, Ly, w)
    d = qmg.make_desc(Lx, Ly, 2, cl, hp, -0.07 + 0.02j, 0.011, 0.023 - 0.01j)
    nrhs, mask = 3, 0b110
    x, l0 = cs.gaussian_cvec(n * nrhs, 1), cs.gaussian_cvec(n * nrhs, 2)
    dx = D(x)
    for pieces in SERVED:
        want, got = D(l0), D(l0)
        qmg.stencil_apply_t(qmg.C64, d, want, dx, pieces, nrhs, n, mask)
        qmg.wilson_apply_direct(qmg.C64, d, g, got, dx, pieces, w, nrhs, n, mask)
        assert np.array_equal(got.to_host(), want.to_host()), hex(pieces)
        want1, got1 = D(l0[:n]), D(l0[:n])          # and one system through the non-batch variant
        qmg.stencil_apply(d, want1, dx, pieces)
        qmg.wilson_apply_direct(qmg.C64, d, g, got1, dx, pieces, w)
        assert np.array_equal(got1.to_host(), want1.to_host()), hex(pieces)


def test_direct_apply_against_the_oracle_and_in_fp32():
    L = 32
    ph = np.random.default_rng(8).uniform(-np.pi, np.pi, size=2 * L * L)
    g64 = ol.phases_to_gauge_u1(ph, L, L)
    clover, hopping = ol.wilson_fill(g64, L, L)
    n = 2 * L * L
    x = cs.gaussian_cvec(n, 4)
    want = ol.stencil_apply(ol.make_desc(L, L, 2, clover, hopping, -0.07), x)
    d = qmg.make_desc(L, L, 2, None, None, -0.07)
    got = qmg.DeviceArray(n)
    qmg.wilson_apply_direct(qmg.C64, d, D(g64), got, D(x), P.P_ALL | P.P_ZERO)
    assert cs.rel_l2(got.to_host(), want) < 1e-15
    got32 = qmg.DeviceArray(n, np.complex64)
    qmg.wilson_apply_direct(qmg.C32, d, D(g64.astype(np.complex64)), got32, D(x.astype(np.complex64)), P.P_ALL | P.P_ZERO)
    assert cs.rel_l2(got32.to_host().astype(np.complex128), want) < 2e-6
    # D_eo in place (the reference's aliased use, stencil_2d.h:1904): even rows written from the odd half
    inplace = D(x)
    qmg.wilson_apply_direct(qmg.C64, d, D(g64), inplace, inplace, P.P_EO | P.P_ZERO_E)
    ref = ol.stencil_apply(ol.make_desc(L, L, 2, clover, hopping, -0.07), x, P.P_EO | P.P_ZERO_E, lhs=x.copy())
    h = inplace.to_host()
    assert np.array_equal(h[n // 2:], x[n // 2:])
    assert cs.rel_l2(h[:n // 2], ref[:n // 2]) < 1e-15


def test_direct_apply_refuses_what_it_does_not_serve():
    import ctypes as C
    L = 16
    g, v, o = D(gauge(L, L, 1)), D(cs.gaussian_cvec(2 * L * L, 1)), qmg.DeviceArray(2 * L * L)
    d = qmg.make_desc(L, L, 2, None, None, 0.1)
    lib = qmg.lib()

    def call(pieces, lhs=o, dtype=qmg.C64, desc=d):
        return lib.qmg_wilson_apply_direct(dtype, C.byref(desc), C.c_void_p(g.ptr), L, 0, C.c_double(1.0), C.c_void_p(lhs.ptr), C.c_void_p(v.ptr), None, None,
                                           C.c_uint(pieces), 1, C.c_size_t(0), C.c_size_t(0), C.c_uint(1), 0, None)
    assert call(P.P_ALL | P.P_ZERO) == 0
    assert call(P.P_EO_XP1 | P.P_ZERO_E) == 3          # a single direction: the stored stencil serves it
    assert call(P.P_CLOVER | P.P_ZERO) == 3            # clover alone
    assert call(P.P_SHIFT | P.P_ZERO) == 3
    assert call(P.P_ALL | P.P_ZERO, lhs=v) == 1        # the full operator in place
    d4 = qmg.make_desc(L, L, 4, None, None, 0.1)
    assert call(P.P_ALL | P.P_ZERO, desc=d4) == 3


@pytest.mark.parametrize("R", [2, 4])
def test_direct_apply_on_slabs_reads_the_global_links(R):
    """Slab r applies rows [r L/R, (r+1) L/R) from the GLOBAL gauge field with the neighbour rows as halos: bit for bit the rows of the
    single-domain direct apply."""
    L, nrhs = 32, 2
    n = 2 * L * L
    g = D(gauge(L, L, 6))
    d = qmg.make_desc(L, L, 2, None, None, 0.05)
    x = cs.gaussian_cvec(n * nrhs, 9)
    want = qmg.DeviceArray(n * nrhs)
    qmg.wilson_apply_direct(qmg.C64, d, g, want, D(x), P.P_ALL | P.P_ZERO, 1.0, nrhs, n, 0b11)
    want = want.to_host()
    Ll, row = L // R, L
    nl = 2 * L * Ll
    xs = x.reshape(nrhs, 2, L, row)

    def rows(a, y0):
        return a.reshape(2, L, row)[:, y0:y0 + Ll].reshape(-1)
    for r in range(R):
        y0 = r * Ll
        dl = qmg.make_desc(L, Ll, 2, None, None, 0.05)
        dx = D(np.concatenate([rows(x[k * n:(k + 1) * n], y0) for k in range(nrhs)]))
        out = qmg.DeviceArray(nl * nrhs)
        lo, hi = D(xs[:, :, (y0 - 1) % L].reshape(-1)), D(xs[:, :, (y0 + Ll) % L].reshape(-1))
        for rows_mode in (1, 2):
            qmg.wilson_apply_direct(qmg.C64, dl, g, out, dx, P.P_ALL | P.P_ZERO, 1.0, nrhs, nl, 0b11, gauge_Ly=L, y0=y0, halo_lo=lo, halo_hi=hi,
                                    halo_stride=2 * row, rows=rows_mode)
        got = out.to_host()
        for k in range(nrhs):
            assert np.array_equal(got[k * nl:(k + 1) * nl], rows(want[k * n:(k + 1) * n], y0)), (R, r, k)
        # one system: the two-rows-per-lane-group form of kernel W2 (row pairs starting at y = 0 with both halos for rows = 0, at the odd
        # row y = 1 for the interior rows) must leave the same bytes
        lo1, hi1 = D(xs[0, :, (y0 - 1) % L].reshape(-1)), D(xs[0, :, (y0 + Ll) % L].reshape(-1))
        dx1 = D(rows(x[:n], y0))
        for modes in ((0,), (1, 2)):
            out1 = qmg.DeviceArray.zeros(nl)
            for rows_mode in modes:
                qmg.wilson_apply_direct(qmg.C64, dl, g, out1, dx1, P.P_ALL | P.P_ZERO, 1.0, gauge_Ly=L, y0=y0, halo_lo=lo1, halo_hi=hi1, halo_stride=2 * row,
                                        rows=rows_mode)
            assert np.array_equal(out1.to_host(), rows(want[:n], y0)), (R, r, modes)


RBJ_SERVED = [P.P_EO | P.P_ZERO_E, P.P_OE | P.P_ZERO_O, P.P_EO, P.P_OE, P.P_HOPPING | P.P_ZERO, P.P_HOPPING]


@pytest.mark.parametrize("Lx,Ly,mass", [(16, 16, -0.07), (24, 10, 0.3), (130, 6, -0.2)])
def test_rbjacobi_hops_from_the_links_are_bit_identical_to_the_built_stencil(Lx, Ly, mass):
    """Right-block-Jacobi Wilson (stencil_2d.h:1452-1601): cinv = 1 / (2w + m) times the identity, so H' = H * cinv(x + mu) has the stored
    hop entries times ONE number -- qmg_wilson_hops_direct forms them from the links with that number read back from the built cinv."""
    n, vol = 2 * Lx * Ly, Lx * Ly
    g = D(gauge(Lx, Ly, 5))
    cl, hp = qmg.DeviceArray(4 * vol), qmg.DeviceArray(16 * vol)
    w = 0.9
    qmg.wilson_fill(cl, hp, g, Lx, Ly, w)
    d = qmg.make_desc(Lx, Ly, 2, cl, hp, mass)
    cinv, rcl, rhp = qmg.DeviceArray(4 * vol), qmg.DeviceArray(4 * vol), qmg.DeviceArray(16 * vol)
    qmg.build_rbjacobi(cinv, rcl, rhp, d)
    ci = cinv.to_host().reshape(vol, 4)
    scale = ci[0, 0].real
    assert np.all(ci[:, 0] == ci[0, 0]) and np.all(ci[:, 3] == ci[0, 0]) and np.all(ci[:, 1] == 0) and np.all(ci[:, 2] == 0) and ci[0, 0].imag == 0
    assert abs(scale - 1.0 / (2 * w + mass)) < 1e-15
    drb = qmg.make_desc(Lx, Ly, 2, None, rhp)
    nrhs, mask = 3, 0b101
    x, l0 = cs.gaussian_cvec(n * nrhs, 1), cs.gaussian_cvec(n * nrhs, 2)
    dx = D(x)
    for pieces in RBJ_SERVED:
        want, got = D(l0), D(l0)
        qmg.stencil_apply_t(qmg.C64, drb, want, dx, pieces, nrhs, n, mask)
        qmg.wilson_hops_direct(qmg.C64, drb, g, got, dx, pieces, w, scale, nrhs, n, mask)
        assert np.array_equal(got.to_host(), want.to_host()), hex(pieces)
        want1, got1 = D(l0[:n]), D(l0[:n])
        qmg.stencil_apply(drb, want1, dx, pieces)
        qmg.wilson_hops_direct(qmg.C64, drb, g, got1, dx, pieces, w, scale)
        assert np.array_equal(got1.to_host(), want1.to_host()), hex(pieces)
    # in place (the Schur complement's second hop, stencil_2d.h:1904) and fp32
    a, b = D(x[:n]), D(x[:n])
    qmg.stencil_apply(drb, a, a, P.P_EO | P.P_ZERO_E)
    qmg.wilson_hops_direct(qmg.C64, drb, g, b, b, P.P_EO | P.P_ZERO_E, w, scale)
    assert np.array_equal(a.to_host(), b.to_host())
    g32 = D(g.to_host().astype(np.complex64))
    out32 = qmg.DeviceArray(n, np.complex64)
    qmg.wilson_hops_direct(qmg.C32, drb, g32, out32, D(x[:n].astype(np.complex64)), P.P_HOPPING | P.P_ZERO, w, scale)
    ref = qmg.DeviceArray(n)
    qmg.stencil_apply(drb, ref, dx, P.P_HOPPING | P.P_ZERO)
    assert cs.rel_l2(out32.to_host().astype(np.complex128), ref.to_host()) < 2e-6
    # what it does not serve
    import ctypes as C
    def call(pieces):
        return qmg.lib().qmg_wilson_hops_direct(qmg.C64, C.byref(drb), C.c_void_p(g.ptr), Ly, 0, C.c_double(w), C.c_double(scale), C.c_void_p(got.ptr), C.c_void_p(dx.ptr),
                                                None, None, C.c_uint(pieces), 1, C.c_size_t(0), C.c_size_t(0), C.c_uint(1), 0, None)
    assert call(P.P_ALL | P.P_ZERO) == 3 and call(P.P_HOPPING | P.P_SHIFT | P.P_ZERO) == 3 and call(P.P_EO_XP1 | P.P_ZERO_E) == 3


@pytest.mark.parametrize("R", [2, 4])
def test_rbjacobi_hops_from_the_links_on_slabs(R):
    """qmg_wilson_hops_direct on y-slabs (global links, halo rows of the right-hand side): bit for bit the rows of the single-domain call,
    for D'_eo, D'_oe and both, all rows / interior + boundary rows."""
    L, nrhs, w, scale = 32, 2, 1.0, 1.0 / (2.0 - 0.07)
    n = 2 * L * L
    g = D(gauge(L, L, 16))
    d = qmg.make_desc(L, L, 2, None, None)
    x = cs.gaussian_cvec(n * nrhs, 19)
    Ll, row = L // R, L
    nl = 2 * L * Ll
    xs = x.reshape(nrhs, 2, L, row)

    def rows(a, y0):
        return a.reshape(2, L, row)[:, y0:y0 + Ll].reshape(-1)
    for pieces in (P.P_EO | P.P_ZERO_E, P.P_OE | P.P_ZERO_O, P.P_HOPPING | P.P_ZERO):
        want = qmg.DeviceArray.zeros(n * nrhs)
        qmg.wilson_hops_direct(qmg.C64, d, g, want, D(x), pieces, w, scale, nrhs, n, 0b11)
        want = want.to_host()
        for r in range(R):
            y0 = r * Ll
            dl = qmg.make_desc(L, Ll, 2, None, None)
            dx = D(np.concatenate([rows(x[k * n:(k + 1) * n], y0) for k in range(nrhs)]))
            lo, hi = D(xs[:, :, (y0 - 1) % L].reshape(-1)), D(xs[:, :, (y0 + Ll) % L].reshape(-1))
            for modes in ((0,), (1, 2)):
                out = qmg.DeviceArray.zeros(nl * nrhs)
                for rows_mode in modes:
                    qmg.wilson_hops_direct(qmg.C64, dl, g, out, dx, pieces, w, scale, nrhs, nl, 0b11, gauge_Ly=L, y0=y0, halo_lo=lo, halo_hi=hi,
                                           halo_stride=2 * row, rows=rows_mode)
                got = out.to_host()
                for k in range(nrhs):
                    assert np.array_equal(got[k * nl:(k + 1) * nl], rows(want[k * n:(k + 1) * n], y0)), (hex(pieces), R, r, k, modes)
