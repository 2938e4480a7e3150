"""Stencil applies with an EPILOGUE (include/qmg_hip.h: qmg_apply_epilogue; csrc/qmg_stencil.hip kernels B / B32, csrc/qmg_wilson.hip kernel W):
the finished site values become other_scale * other + acc_scale * acc -- the residual b - A x (stateful_multigrid.h:863-866, 1023-1029), the
Schur combination r_e - D'_eo t (stencil_2d.h:1894-1907) -- and / or leave minv_vector_minres's <p,r>, <p,p> in the device slot that
qmg_batch_mr_update_t consumes.  The pin is the unfused composition through the entry points that are themselves held to the oracle
(tests/test_gpu_parity.py, test_gpu_wilson_direct.py, test_gpu_f32.py): apply, then caxpbyz, then the dots of the STORED result.
fp64: the combined value is bit for bit the two-pass value (one rounding in both); the dots differ from a separate pass by summation order
only (1e-12).  fp32: the fused value is rounded once instead of twice (3e-7), its dots are those of the stored values (1e-12 of them)."""
import importlib

import numpy as np
import pytest

import coordspace as cs

qmg = importlib.import_module("quantum-mg_amd")
pytestmark = pytest.mark.gpu
P = qmg


@pytest.fixture(scope="module", autouse=True)
def _device():
    qmg.build()
    qmg.init(0)
    yield
    qmg.sync()


def D(a, dtype=np.complex128):
    return qmg.DeviceArray.from_host(np.ascontiguousarray(a, dtype=dtype))


def check_dots(slot_row, out, r, rel=1e-12):
    """slot = (Re<p,r>, Im<p,r>, <p,p>) with <p,r> = sum conj(p) r, p = out as stored"""
    o, rr = out.astype(np.complex128), r.astype(np.complex128)
    pr, pp = np.vdot(o, rr), np.vdot(o, o).real
    scale = np.sqrt(pp * np.vdot(rr, rr).real)
    assert abs(complex(slot_row[0], slot_row[1]) - pr) <= rel * scale and abs(slot_row[2] - pp) <= rel * pp, (slot_row, pr, pp)


@pytest.mark.parametrize("nc,Lx,Ly", [(3, 12, 8), (8, 16, 16), (24, 8, 12), (12, 34, 10)])
@pytest.mark.parametrize("storage", ["f64", "mat32", "c32"])
def test_kernel_B_epilogue(nc, Lx, Ly, storage):
    if storage != "f64" and nc % 2:
        pytest.skip("fp32-stored matrices: even nc (kernel B32)")
    vol = Lx * Ly
    n = vol * nc
    clover = cs.gaussian_cvec(vol * nc * nc, 1)
    hopping = cs.gaussian_cvec(4 * vol * nc * nc, 2)
    vt = np.complex64 if storage == "c32" else np.complex128
    mt = np.complex128 if storage == "f64" else np.complex64
    dt = qmg.C32 if storage == "c32" else qmg.C64
    mat32 = storage != "f64"
    dcl, dho = D(clover, mt), D(hopping, mt)
    d = qmg.make_desc(Lx, Ly, nc, dcl, dho, 0.3 - 0.1j, 0.02, 0.05 if nc % 2 == 0 else 0.0)
    nsys, system = 3, 2
    stride = n + 16
    x = cs.gaussian_cvec(stride * nsys, 3).astype(vt)
    b = cs.gaussian_cvec(stride * nsys, 4).astype(vt)
    dx, db = D(x, vt), D(b, vt)
    sl = slice(system * stride, system * stride + n)

    def plain(pieces):      # the unfused apply of system `system`
        tmp = D(np.zeros(stride * nsys), vt)
        if storage == "f64":
            qmg.stencil_apply_batch(d, tmp, dx, pieces, nsys, stride, 1 << system)
        elif storage == "mat32":
            qmg.stencil_apply_mat32(d, tmp, dx, pieces, nsys, stride, 1 << system)
        else:
            qmg.stencil_apply_t(qmg.C32, d, tmp, dx, pieces, nsys, stride, 1 << system)
        return tmp

    half = n // 2
    for pieces, region in ((P.P_ALL | P.P_ZERO, slice(0, n)), (P.P_EO | P.P_ZERO_E, slice(0, half)), (P.P_OE | P.P_CLOVER_O | P.P_ZERO_O, slice(half, n))):
        tmp = plain(pieces)
        ax = tmp.to_host()[sl]
        # (i) b - A x
        out = D(np.full(stride * nsys, 7.0), vt)
        assert qmg.stencil_apply_epi(dt, mat32, d, out, dx, pieces, qmg.make_epilogue(db, 1.0, -1.0, None), stride, system) == 0
        got = out.to_host()
        want = (b[sl].astype(np.complex128) - ax.astype(np.complex128))
        if storage == "c32":
            assert cs.rel_l2(got[sl][region], want[region]) < 3e-7
        else:
            assert np.array_equal(got[sl][region], want[region])
        untouched = np.ones(stride * nsys, dtype=bool)
        untouched[system * stride + region.start:system * stride + region.stop] = False
        assert np.all(got[untouched] == 7.0)                                           # other parity, other systems, padding: not written
        # (ii) p = A x with the MR dots against x
        out2 = D(np.zeros(stride * nsys), vt)
        assert qmg.stencil_apply_epi(dt, mat32, d, out2, dx, pieces, qmg.make_epilogue(None, 0.0, 1.0, dx), stride, system) == 0
        got2 = out2.to_host()[sl]
        assert np.array_equal(got2[region], ax[region])
        check_dots(qmg.batch_mr_read_dots(nsys)[system], got2[region], x[sl][region])
        # (iii) both at once, dots against b
        out3 = D(np.zeros(stride * nsys), vt)
        assert qmg.stencil_apply_epi(dt, mat32, d, out3, dx, pieces, qmg.make_epilogue(db, 1.0, -1.0, db), stride, system) == 0
        check_dots(qmg.batch_mr_read_dots(nsys)[system], out3.to_host()[sl][region], b[sl][region])
    # not served / refused
    assert qmg.stencil_apply_epi(dt, mat32, d, D(np.zeros(stride * nsys), vt), dx, P.P_ALL, qmg.make_epilogue(db, 1.0, -1.0, None), stride, system) == 1   # accumulate + epilogue
    if storage == "f64":
        d2 = qmg.make_desc(Lx, Ly, 2, D(cs.gaussian_cvec(vol * 4, 5)), D(cs.gaussian_cvec(4 * vol * 4, 6)), 0.1)
        v2 = D(cs.gaussian_cvec(2 * vol, 7))
        assert qmg.stencil_apply_epi(qmg.C64, False, d2, D(np.zeros(2 * vol)), v2, P.P_ALL | P.P_ZERO, qmg.make_epilogue(v2, 1.0, -1.0, None)) == 3     # nc = 2: kernel W's job


def gauge(Lx, Ly, seed):
    rng = np.random.default_rng(seed)
    return np.exp(1j * rng.uniform(-np.pi, np.pi, size=2 * Lx * Ly))


@pytest.mark.parametrize("Lx,Ly", [(16, 16), (130, 6), (24, 10), (66, 7 * 2)])
@pytest.mark.parametrize("dtype", ["c64", "c32"])
@pytest.mark.parametrize("pair", [2, 1, 0])
def test_kernel_W_epilogue(Lx, Ly, dtype, pair):
    """Kernel W in its three forms (wilson_pair = 2: two rows per lane group, 1: both parities of a column, 0: one site) with the epilogue,
    incl. half rows that do not fill their last wavefront (Lx = 130, 66: padding lanes must neither store nor count), hops-only launches
    (D_eo with r_e - D_eo t on the even sites) and the right-block-Jacobi hops (qmg_wilson_hops_direct_epi)."""
    qmg.set_tuning("wilson_pair", pair)
    try:
        n = 2 * Lx * Ly
        vt = np.complex128 if dtype == "c64" else np.complex64
        dt = qmg.C64 if dtype == "c64" else qmg.C32
        g = D(gauge(Lx, Ly, 3), vt)
        w = 0.9
        d = qmg.make_desc(Lx, Ly, 2, None, None, -0.07 + 0.02j, 0.011, 0.023 - 0.01j)
        nsys, system = 2, 1
        stride = n + 8
        x = cs.gaussian_cvec(stride * nsys, 1).astype(vt)
        b = cs.gaussian_cvec(stride * nsys, 2).astype(vt)
        dx, db = D(x, vt), D(b, vt)
        sl = slice(system * stride, system * stride + n)
        half = n // 2
        tol = 0 if dtype == "c64" else 3e-7
        cases = [("apply", P.P_ALL | P.P_ZERO, slice(0, n)), ("apply", P.P_CLOVER | P.P_HOPPING | P.P_ZERO, slice(0, n)), ("apply", P.P_EO | P.P_ZERO_E, slice(0, half)),
                 ("apply", P.P_CLOVER_O | P.P_OE | P.P_SHIFT_O | P.P_ZERO_O, slice(half, n)), ("hops", P.P_EO | P.P_ZERO_E, slice(0, half)), ("hops", P.P_HOPPING | P.P_ZERO, slice(0, n))]
        for kind, pieces, region in cases:
            tmp = D(np.zeros(stride * nsys), vt)
            if kind == "apply":
                qmg.wilson_apply_direct(dt, d, g, tmp, dx, pieces, w, nsys, stride, 1 << system)
            else:
                qmg.wilson_hops_direct(dt, d, g, tmp, dx, pieces, w, 0.517, nsys, stride, 1 << system)
            ax = tmp.to_host()[sl]

            def fused(epi, fill=0.0):
                out = D(np.full(stride * nsys, fill), vt)
                if kind == "apply":
                    rc = qmg.wilson_apply_direct_epi(dt, d, g, out, dx, pieces, epi, nsys, stride, 1 << system, w)
                else:
                    rc = qmg.wilson_hops_direct_epi(dt, d, g, 0.517, out, dx, pieces, epi, nsys, stride, 1 << system, w)
                assert rc == 0, (kind, hex(pieces), rc)
                return out.to_host()

            got = fused(qmg.make_epilogue(db, 1.0, -1.0, None), 7.0)
            want = b[sl].astype(np.complex128) - ax.astype(np.complex128)
            if tol == 0:
                assert np.array_equal(got[sl][region], want[region]), (kind, hex(pieces))
            else:
                assert cs.rel_l2(got[sl][region], want[region]) < tol
            untouched = np.ones(stride * nsys, dtype=bool)
            untouched[system * stride + region.start:system * stride + region.stop] = False
            assert np.all(got[untouched] == 7.0)
            got2 = fused(qmg.make_epilogue(None, 0.0, 1.0, dx))[sl]
            assert np.array_equal(got2[region], ax[region])
            check_dots(qmg.batch_mr_read_dots(nsys)[system], got2[region], x[sl][region])
            got3 = fused(qmg.make_epilogue(db, 1.0, -1.0, db))[sl]
            check_dots(qmg.batch_mr_read_dots(nsys)[system], got3[region], b[sl][region])
        # two active systems, accumulate: refused / not served
        assert qmg.wilson_apply_direct_epi(dt, d, g, D(np.zeros(stride * nsys), vt), dx, P.P_ALL | P.P_ZERO, qmg.make_epilogue(db, 1.0, -1.0, None), nsys, stride, 0b11, w) == 3
        assert qmg.wilson_apply_direct_epi(dt, d, g, D(np.zeros(stride * nsys), vt), dx, P.P_ALL, qmg.make_epilogue(db, 1.0, -1.0, None), nsys, stride, 0b10, w) == 1
    finally:
        qmg.set_tuning("wilson_pair", 2)


@pytest.mark.parametrize("dtype", ["c64", "c32"])
def test_mr_step_from_a_fused_apply_equals_the_separate_passes(dtype):
    """One MR(omega) step of the K-cycle's smoother both ways on a Wilson operator from the links: (a) apply, qmg_batch_mr_dots_t, qmg_batch_mr_update_t;
    (b) apply with the MR epilogue, qmg_batch_mr_update_t.  p is the same vector bit for bit; alpha comes from dots that differ by summation order,
    so x and r agree to 1e-13 (fp64) / fp32 rounding."""
    Lx, Ly = 64, 32
    n = 2 * Lx * Ly
    vt = np.complex128 if dtype == "c64" else np.complex64
    dt = qmg.C64 if dtype == "c64" else qmg.C32
    g = D(gauge(Lx, Ly, 9), vt)
    d = qmg.make_desc(Lx, Ly, 2, None, None, -0.05)
    r = cs.gaussian_cvec(n, 11).astype(vt)
    res = []
    for fusedp in (False, True):
        dr, dp, dxx = D(r, vt), D(np.zeros(n), vt), D(np.zeros(n), vt)
        if fusedp:
            assert qmg.wilson_apply_direct_epi(dt, d, g, dp, dr, P.P_ALL | P.P_ZERO, qmg.make_epilogue(None, 0.0, 1.0, dr)) == 0
        else:
            qmg.wilson_apply_direct(dt, d, g, dp, dr, P.P_ALL | P.P_ZERO)
            qmg.batch_mr_dots(dt, dr, dp, n, 1, n, 1)
        qmg.batch_mr_update(dt, 0.85, dxx, dr, dr, dp, True, n, 1, n, 1)
        res.append((dp.to_host(), dxx.to_host(), dr.to_host()))
    assert np.array_equal(res[0][0], res[1][0])
    tol = 1e-13 if dtype == "c64" else 2e-7
    assert cs.rel_l2(res[1][1], res[0][1]) < tol and cs.rel_l2(res[1][2], res[0][2]) < tol
    # and the step is MR's: x = alpha r with alpha = omega <p,r>/<p,p>, new residual orthogonal-ish: <p, r_new> = (1 - omega) <p, r>
    p, xx, rn = (v.astype(np.complex128) for v in res[1])
    alpha = 0.85 * np.vdot(p, r.astype(np.complex128)) / np.vdot(p, p).real
    assert cs.rel_l2(xx, alpha * r.astype(np.complex128)) < (1e-13 if dtype == "c64" else 3e-7)
    assert abs(np.vdot(p, rn) - 0.15 * np.vdot(p, r.astype(np.complex128))) < (1e-12 if dtype == "c64" else 1e-5) * abs(np.vdot(p, r.astype(np.complex128)))
