"""Lock-step batches (qmg_batch.hip, qmg_stencil_apply_batch): every batched entry point against the single-vector
entry point applied per active system -- bit-identical for element-wise ops and reductions (same block partition and
fixed-order second stage), 1e-13 for the MFMA / register-blocked kernels whose summation order differs -- and frozen
(masked-out) systems must come back untouched."""
import importlib

import numpy as np
import pytest

import coordspace as cs
import oracle_lib as ol

qmg = importlib.import_module("quantum-mg_amd")
pytestmark = pytest.mark.gpu
TOL = 1e-13


@pytest.fixture(scope="module", autouse=True)
def _device():
    qmg.build()
    qmg.init(0)
    yield
    qmg.sync()


def D(a):
    return qmg.DeviceArray.from_host(np.ascontiguousarray(a, dtype=np.complex128))


def active(mask, nrhs):
    return [k for k in range(nrhs) if (mask >> k) & 1]


MASKS = [(5, 0b11111), (5, 0b01101), (16, 0xFFFF), (16, 0x8421), (3, 0b010), (9, 0b111111111)]


@pytest.mark.parametrize("nrhs,mask", MASKS)
def test_batch_blas_bit_identical_and_masked(nrhs, mask):
    n, pad = 1000, 7
    stride = n + pad
    x = cs.gaussian_cvec(stride * nrhs, 1)
    y = cs.gaussian_cvec(stride * nrhs, 2)
    z0 = cs.gaussian_cvec(stride * nrhs, 3)
    a = cs.gaussian_cvec(nrhs, 4)
    b = cs.gaussian_cvec(nrhs, 5)
    dx, dy = D(x), D(y)
    single = {qmg.BOP_ZERO: lambda zk, xk, yk, k: qmg.zero_vector(zk, n), qmg.BOP_COPY: lambda zk, xk, yk, k: qmg.copy_vector(zk, xk, n),
              qmg.BOP_CAX: lambda zk, xk, yk, k: qmg.cax(a[k], zk, n), qmg.BOP_CAXPY: lambda zk, xk, yk, k: qmg.caxpy(a[k], xk, zk, n),
              qmg.BOP_CXPY: lambda zk, xk, yk, k: qmg.cxpy(xk, zk, n), qmg.BOP_CAXPBYZ: lambda zk, xk, yk, k: qmg.caxpbyz(a[k], xk, b[k], yk, zk, n)}
    for op, fn in single.items():
        dz = D(z0)
        qmg.batch_blas(op, dz, n, nrhs, stride, mask, a=a, b=b, x=dx, y=dy)
        got = dz.to_host()
        want = z0.copy()
        for k in active(mask, nrhs):
            zk, xk, yk = D(z0[k * stride:k * stride + n]), D(x[k * stride:k * stride + n]), D(y[k * stride:k * stride + n])
            fn(zk, xk, yk, k)
            want[k * stride:k * stride + n] = zk.to_host()
        assert np.array_equal(got, want), op


@pytest.mark.parametrize("nrhs,mask", MASKS)
@pytest.mark.parametrize("n", [1, 777, 300001])
def test_batch_reductions_bit_identical(nrhs, mask, n):
    stride = n + 3
    x = cs.gaussian_cvec(stride * nrhs, 11)
    y = cs.gaussian_cvec(stride * nrhs, 12)
    dx, dy = D(x), D(y)
    nrm = qmg.batch_reduce(qmg.BRED_NORM2, dx, None, n, nrhs, stride, mask)
    dt = qmg.batch_reduce(qmg.BRED_DOT, dx, dy, n, nrhs, stride, mask)
    df = qmg.batch_reduce(qmg.BRED_DIFFNORM2, dx, dy, n, nrhs, stride, mask)
    for k in range(nrhs):
        if (mask >> k) & 1:
            xk, yk = D(x[k * stride:k * stride + n]), D(y[k * stride:k * stride + n])
            assert nrm[k].real == qmg.norm2sq(xk, n)
            assert dt[k] == qmg.dot(xk, yk, n)
            assert df[k].real == qmg.diffnorm2sq(xk, yk, n)
            assert abs(dt[k] - np.vdot(x[k * stride:k * stride + n], y[k * stride:k * stride + n])) <= 1e-12 * abs(dt[k]) + 1e-12
        else:
            assert np.isnan(nrm[k].real) and np.isnan(dt[k].real)   # untouched


@pytest.mark.parametrize("nj", [1, 2, 5, 11, 32])
def test_batch_multidot_and_multi_caxpy(nj):
    nrhs, mask, n = 6, 0b101101, 4099
    stride = n + 5
    xs = [cs.gaussian_cvec(stride * nrhs, 20 + j) for j in range(nj)]
    y = cs.gaussian_cvec(stride * nrhs, 99)
    dxs, dy = [D(v) for v in xs], D(y)
    got = qmg.batch_multidot(dxs, dy, n, nrhs, stride, mask)
    for k in range(nrhs):
        if (mask >> k) & 1:
            single = qmg.multidot([D(v[k * stride:k * stride + n]) for v in xs], D(y[k * stride:k * stride + n]), n)
            assert np.array_equal(got[k], single)
        else:
            assert np.all(np.isnan(got[k].real))
    coeffs = cs.gaussian_cvec(nj * nrhs, 7).reshape(nj, nrhs)
    qmg.batch_multi_caxpy(coeffs, dxs, dy, n, nrhs, stride, mask)
    out = dy.to_host()
    want = y.copy()
    for k in active(mask, nrhs):
        sl = slice(k * stride, k * stride + n)
        for j in range(nj):
            want[sl] += coeffs[j, k] * xs[j][sl]
    assert np.array_equal(out[[i for k in range(nrhs) if not (mask >> k) & 1 for i in range(k * stride, (k + 1) * stride)]],
                          y[[i for k in range(nrhs) if not (mask >> k) & 1 for i in range(k * stride, (k + 1) * stride)]])
    assert cs.rel_l2(out, want) < TOL


@pytest.mark.parametrize("nc,nrhs,mask", [(2, 4, 0b1011), (1, 3, 0b101), (8, 6, 0b110101), (24, 16, 0xFFFF), (24, 12, 0b101010111011), (3, 4, 0b0110), (16, 2, 0b10),
                                          (8, 16, 0xFFFF), (8, 11, 0x7FF), (24, 9, 0x1FF), (12, 16, 0xFFFF), (32, 10, 0x3FF)])
def test_stencil_apply_batch_masked(nc, nrhs, mask):
    Lx, Ly = 12, 6
    vol = Lx * Ly
    size = vol * nc
    stride = size + 4
    clover = cs.gaussian_cvec(vol * nc * nc, 1)
    hopping = cs.gaussian_cvec(4 * vol * nc * nc, 2)
    rhs = cs.gaussian_cvec(stride * nrhs, 3)
    lhs0 = cs.gaussian_cvec(stride * nrhs, 4)
    shifts = (0.3 - 0.2j, 0.11, -0.07 + 0.02j)
    od = ol.make_desc(Lx, Ly, nc, clover, hopping, *shifts)
    gd = qmg.make_desc(Lx, Ly, nc, D(clover), D(hopping), *shifts)
    for pieces in (ol.P_ALL | ol.P_ZERO, ol.P_ALL, ol.P_EO | ol.P_ZERO_E):
        want = lhs0.copy()
        for k in active(mask, nrhs):
            ol.stencil_apply(od, np.ascontiguousarray(rhs[k * stride:k * stride + size]), pieces, lhs=want[k * stride:k * stride + size])
        dl = D(lhs0)
        qmg.stencil_apply_batch(gd, dl, D(rhs), pieces, nrhs, stride, mask)
        got = dl.to_host()
        assert cs.rel_l2(got, want) < TOL
        for k in range(nrhs):
            if not (mask >> k) & 1:
                assert np.array_equal(got[k * stride:(k + 1) * stride], lhs0[k * stride:(k + 1) * stride])


@pytest.mark.parametrize("fd,cd,nrhs,mask", [((16, 16, 2), (4, 4, 8), 5, 0b10111), ((16, 8, 2), (8, 4, 4), 9, 0b110110101), ((8, 8, 8), (2, 2, 12), 16, 0xFFFF),
                                             ((16, 16, 24), (4, 4, 24), 3, 0b101), ((32, 32, 2), (8, 8, 24), 8, 0xFF), ((8, 8, 6), (4, 4, 6), 2, 0b11)])
def test_transfer_batch_matches_single(fd, cd, nrhs, mask):
    fsize, csize = fd[0] * fd[1] * fd[2], cd[0] * cd[1] * cd[2]
    nvec = cd[2]
    fstride, cstride = fsize + 6, csize + 2
    nv = cs.gaussian_cvec(nvec * fsize, 1)
    fine0 = cs.gaussian_cvec(fstride * nrhs, 2)
    coarse0 = cs.gaussian_cvec(cstride * nrhs, 3)
    dnv = D(nv)
    # prolong: fine += P coarse
    df = D(fine0)
    qmg.prolong_batch(dnv, nvec, D(coarse0), df, fd, cd, nrhs, cstride, fstride, mask)
    got = df.to_host()
    want = fine0.copy()
    for k in active(mask, nrhs):
        fk = D(fine0[k * fstride:k * fstride + fsize])
        qmg.prolong(dnv, nvec, D(coarse0[k * cstride:k * cstride + csize]), fk, fd, cd)
        want[k * fstride:k * fstride + fsize] = fk.to_host()
    assert cs.rel_l2(got, want) < TOL
    for k in range(nrhs):
        if not (mask >> k) & 1:
            assert np.array_equal(got[k * fstride:(k + 1) * fstride], fine0[k * fstride:(k + 1) * fstride])
    # and against the oracle for the first active system
    k = active(mask, nrhs)[0]
    o = ol.prolong(nv, coarse0[k * cstride:k * cstride + csize].copy(), fd, cd, fine=fine0[k * fstride:k * fstride + fsize].copy())
    assert cs.rel_l2(got[k * fstride:k * fstride + fsize], o) < TOL
    # restrict: coarse += R fine
    dc = D(coarse0)
    qmg.restrict_batch(dnv, nvec, D(fine0), dc, fd, cd, nrhs, fstride, cstride, mask)
    got = dc.to_host()
    want = coarse0.copy()
    for k in active(mask, nrhs):
        ck = D(coarse0[k * cstride:k * cstride + csize])
        qmg.restrict(dnv, nvec, D(fine0[k * fstride:k * fstride + fsize]), ck, fd, cd)
        want[k * cstride:k * cstride + csize] = ck.to_host()
    assert cs.rel_l2(got, want) < TOL
    for k in range(nrhs):
        if not (mask >> k) & 1:
            assert np.array_equal(got[k * cstride:(k + 1) * cstride], coarse0[k * cstride:(k + 1) * cstride])


@pytest.mark.parametrize("nc,nrhs,mask", [(8, 1, 0b1), (24, 1, 0b1), (3, 1, 0b1), (8, 6, 0b101101), (24, 16, 0xFFFF), (12, 3, 0b111), (32, 2, 0b10), (6, 4, 0b1111)])
def test_stencil_apply_with_f32_stored_matrices(nc, nrhs, mask):
    """qmg_stencil_apply_mat32 (opt-in storage format): the matrices are read as complex<float>, everything else is fp64.
    Parity is exact in the sense that matters: equal (1e-13) to the ORACLE's fp64 apply of the matrices rounded to
    float -- kernels B (one rhs, or nc outside the MFMA set) and C (several rhs, f64 MFMA)."""
    Lx, Ly = 12, 6
    vol = Lx * Ly
    size = vol * nc
    stride = size + 2
    clover = cs.gaussian_cvec(vol * nc * nc, 1)
    hopping = cs.gaussian_cvec(4 * vol * nc * nc, 2)
    rhs = cs.gaussian_cvec(stride * nrhs, 3)
    lhs0 = cs.gaussian_cvec(stride * nrhs, 4)
    shifts = (0.3 - 0.2j, 0.11, -0.07 + 0.02j)
    dc, dh = D(clover), D(hopping)
    dc32, dh32 = qmg.DeviceArray(vol * nc * nc // 2 + 1), qmg.DeviceArray(4 * vol * nc * nc // 2 + 1)   # 8 bytes per element
    qmg.c64_to_c32(dc32, dc, vol * nc * nc)
    qmg.c64_to_c32(dh32, dh, 4 * vol * nc * nc)
    r32 = lambda a: a.astype(np.complex64).astype(np.complex128)
    od = ol.make_desc(Lx, Ly, nc, r32(clover), r32(hopping), *shifts)
    gd = qmg.make_desc(Lx, Ly, nc, dc32, dh32, *shifts)
    for pieces in (ol.P_ALL | ol.P_ZERO, ol.P_ALL, ol.P_EO | ol.P_ZERO_E, ol.P_CLOVER | ol.P_SHIFT):
        want = lhs0.copy()
        for k in active(mask, nrhs):
            ol.stencil_apply(od, np.ascontiguousarray(rhs[k * stride:k * stride + size]), pieces, lhs=want[k * stride:k * stride + size])
        dl = D(lhs0)
        qmg.stencil_apply_mat32(gd, dl, D(rhs), pieces, nrhs, stride, mask)
        assert cs.rel_l2(dl.to_host(), want) < TOL, hex(pieces)
    # nc = 1, 2, 4 are not served
    import ctypes as C
    d2 = qmg.make_desc(Lx, Ly, 2, dc32, dh32)
    assert qmg.lib().qmg_stencil_apply_mat32(C.byref(d2), C.c_void_p(dc.ptr), C.c_void_p(dh.ptr), C.c_uint(0xFFF), 1, C.c_size_t(0), C.c_uint(1), None) == 3


def test_rccl_allreduce_through_the_c_abi(monkeypatch):
    """qmg_comm_* (csrc/qmg_comm.hip): librccl is dlopen'ed, a ONE-rank communicator is created for real
    (QMG_COMM_FORCE_RCCL), and the in-place sum all-reduce of a small double vector in HBM returns the vector -- this pins
    the hand-declared RCCL signatures and enum values (ncclDouble = 8, ncclSum = 0) against the library on the box.
    (Two ranks need two GPUs: the multi-rank logic runs on gloo in tests/test_distributed_cpu.py.)"""
    import ctypes as C
    monkeypatch.setenv("QMG_COMM_FORCE_RCCL", "1")
    L = qmg.lib()
    uid = (C.c_char * 128)()
    assert L.qmg_comm_get_unique_id(uid) == 0
    assert L.qmg_comm_init(uid, 1, 0) == 0
    w, r = C.c_int(-1), C.c_int(-1)
    assert L.qmg_comm_world(C.byref(w), C.byref(r)) == 0 and (w.value, r.value) == (1, 0)
    vals = np.arange(1.0, 9.0) * 0.125 + 1j * np.arange(8.0)          # 16 doubles
    d = D(vals)
    assert L.qmg_allreduce_sum_f64(C.c_void_p(d.ptr), C.c_size_t(16), None) == 0
    qmg.sync()
    assert np.array_equal(d.to_host(), vals)
    assert L.qmg_comm_finalize() == 0


def test_batch_entry_points_edge_cases():
    """Empty masks and empty vectors are no-ops; one-system batches equal the single-vector calls; bad arguments are
    refused with a status, not a crash."""
    import ctypes as C
    L = qmg.lib()
    n, nrhs = 33, 3
    x = cs.gaussian_cvec(n * nrhs, 1)
    y0 = cs.gaussian_cvec(n * nrhs, 2)
    dx, dy = D(x), D(y0)
    qmg.batch_blas(qmg.BOP_CAXPY, dy, n, nrhs, n, 0, a=[1.0, 2.0, 3.0], x=dx)          # mask 0: nothing happens
    assert np.array_equal(dy.to_host(), y0)
    qmg.batch_blas(qmg.BOP_CAXPY, dy, 0, nrhs, n, 0b111, a=[1.0, 2.0, 3.0], x=dx)      # n = 0: nothing happens
    assert np.array_equal(dy.to_host(), y0)
    out = qmg.batch_reduce(qmg.BRED_NORM2, dx, None, n, nrhs, n, 0)
    assert np.all(np.isnan(out.real))
    one = qmg.batch_reduce(qmg.BRED_NORM2, dx, None, n, 1, n, 1)                        # one-system batch == single call
    assert one[0].real == qmg.norm2sq(D(x[:n]), n)
    # refused, not crashed
    assert L.qmg_batch_blas(99, None, None, C.c_void_p(dx.ptr), None, C.c_void_p(dy.ptr), C.c_size_t(n), nrhs, C.c_size_t(n), C.c_uint(1), None) == 1
    assert L.qmg_batch_blas(qmg.BOP_COPY, None, None, C.c_void_p(dx.ptr), None, C.c_void_p(dy.ptr), C.c_size_t(n), 17, C.c_size_t(n), C.c_uint(1), None) == 1
    assert L.qmg_batch_reduce(qmg.BRED_DOT, C.c_void_p(dx.ptr), None, C.c_size_t(n), nrhs, C.c_size_t(n), C.c_uint(1), (C.c_double * 6)(), None) == 1
    d = qmg.make_desc(8, 8, 8, dx, dx)
    assert L.qmg_stencil_apply_batch(C.byref(d), C.c_void_p(dy.ptr), C.c_void_p(dx.ptr), C.c_uint(0xFFF), 17, C.c_size_t(n), C.c_uint(1), None) == 1
    assert L.qmg_stencil_apply_batch(C.byref(d), C.c_void_p(dy.ptr), C.c_void_p(dx.ptr), C.c_uint(0xFFF), 4, C.c_size_t(8 * 8 * 8), C.c_uint(0), None) == 0   # empty mask
    free_b, total_b = C.c_size_t(0), C.c_size_t(0)
    assert L.qmg_mem_info(C.byref(free_b), C.byref(total_b)) == 0 and 0 < free_b.value <= total_b.value and total_b.value > 200e9   # 288 GB part


@pytest.mark.parametrize("dtype", ["c64", "c32"])
def test_batch_non_temporal_reads_change_no_bit(dtype):
    """A batch whose active systems add up to `blas_nt_mb` MiB reads its read-only operands non-temporally (csrc/qmg_batch.hip): the same
    bits with the threshold at 1 MiB (on: 5 systems x 100 001 elements) and at 0 (never), in both storage precisions, aliased z = a x + b z included."""
    n, nrhs, mask = 100001, 5, 0b10111
    stride = n + 3
    np_t = np.complex128 if dtype == "c64" else np.complex64
    dt = qmg.C64 if dtype == "c64" else qmg.C32
    mk = lambda seed: qmg.DeviceArray.from_host(cs.gaussian_cvec(stride * nrhs, seed).astype(np_t))
    x, y = mk(1), mk(2)
    z0 = cs.gaussian_cvec(stride * nrhs, 3).astype(np_t)
    a, b = cs.gaussian_cvec(nrhs, 4), cs.gaussian_cvec(nrhs, 5)
    res = {}
    try:
        for mb in (0, 1):
            qmg.set_tuning("blas_nt_mb", mb)
            out = []
            for op, kw in ((qmg.BOP_CAXPY, dict(a=a, x=x)), (qmg.BOP_CAXPBYZ, dict(a=a, b=b, x=x, y=y)), (qmg.BOP_COPY, dict(x=x))):
                z = qmg.DeviceArray.from_host(z0)
                qmg.batch_blas_t(dt, op, z, n, nrhs, stride, mask, **kw)
                out.append(z.to_host())
            z = qmg.DeviceArray.from_host(z0)
            qmg.batch_blas_t(dt, qmg.BOP_CAXPBYZ, z, n, nrhs, stride, mask, a=a, b=b, x=x, y=z)      # z = a x + b z
            out.append(z.to_host())
            z = qmg.DeviceArray.from_host(z0)
            qmg.batch_multi_caxpy_t(dt, np.outer(np.array([0.3, -0.2j]), a), [x, y], z, n, nrhs, stride, mask)
            out.append(z.to_host())
            out.append(np.nan_to_num(qmg.batch_reduce_t(dt, qmg.BRED_DOT, x, y, n, nrhs, stride, mask)))
            out.append(np.nan_to_num(qmg.batch_multidot_t(dt, [x, y], y, n, nrhs, stride, mask)))
            res[mb] = out
    finally:
        qmg.set_tuning("blas_nt_mb", 256)
    for u, v in zip(res[0], res[1]):
        assert np.array_equal(u, v)


@pytest.mark.parametrize("nrhs,mask", [(1, 0b1), (5, 0b01101), (16, 0xFFFF)])
@pytest.mark.parametrize("dtype", ["c64", "c32"])
def test_mr_step_with_device_scalars_is_the_host_scalar_step(nrhs, mask, dtype):
    """qmg_batch_mr_dots_t + qmg_batch_mr_update_t (the K-cycle's fixed-count smoother, scalars never leave the device) against the
    host-scalar step of bminv_vector_minres_zero_guess built from the existing entry points: multidot of {r, p} against p, alpha =
    omega conj(<r,p>) / <p,p> on the host, two caxpy.  Same reduction order and the same (omega pr) / pp: x and r BIT FOR BIT, in both
    storage precisions; the dots in the device slot equal the multidot's; frozen systems untouched; the x_set form (x = alpha r) equals
    the accumulate form on x = 0; <p,p> = 0 leaves a system as it is."""
    dt = qmg.C64 if dtype == "c64" else qmg.C32
    n, pad, omega = 3000, 8, 0.85
    stride = n + pad
    tohost = (lambda d: d.to_host()) if dtype == "c64" else (lambda d: d.to_host())
    mk = (lambda a: D(a)) if dtype == "c64" else (lambda a: qmg.DeviceArray.from_host(np.ascontiguousarray(a, dtype=np.complex64)))
    x0, r0, p0 = cs.gaussian_cvec(stride * nrhs, 1), cs.gaussian_cvec(stride * nrhs, 2), cs.gaussian_cvec(stride * nrhs, 3)
    if dtype == "c32":
        x0, r0, p0 = (v.astype(np.complex64) for v in (x0, r0, p0))
    act = active(mask, nrhs)
    # host-scalar reference through the existing batched entry points
    dr, dp = mk(r0), mk(p0)
    d2 = qmg.batch_multidot_t(dt, [dr, dp], dp, n, nrhs, stride, mask)
    alpha = np.zeros(nrhs, dtype=np.complex128)
    for k in act:   # alpha = (omega <p,r>) / <p,p> component by component, as std::complex / double divides (numpy multiplies by a reciprocal)
        pr, pp = np.conj(d2[k, 0]), d2[k, 1].real
        alpha[k] = complex((omega * pr.real) / pp, (omega * pr.imag) / pp)
    wx, wr = mk(x0), mk(r0)
    qmg.batch_blas_t(dt, qmg.BOP_CAXPY, wx, n, nrhs, stride, mask, a=alpha, x=dr)
    qmg.batch_blas_t(dt, qmg.BOP_CAXPY, wr, n, nrhs, stride, mask, a=-alpha, x=dp)
    # device-scalar step
    gx, gr = mk(x0), mk(r0)
    qmg.batch_mr_dots(dt, gr, dp, n, nrhs, stride, mask)
    slot = qmg.batch_mr_read_dots(nrhs)
    for k in act:
        assert slot[k, 0] == d2[k, 0].real and slot[k, 1] == -d2[k, 0].imag and slot[k, 2] == d2[k, 1].real, k   # <p,r> = conj(<r,p>)
    qmg.batch_mr_update(dt, omega, gx, gr, gr, dp, False, n, nrhs, stride, mask)
    assert np.array_equal(gx.to_host(), wx.to_host()) and np.array_equal(gr.to_host(), wr.to_host())
    for k in range(nrhs):
        if k not in act:
            assert np.array_equal(gx.to_host()[k * stride:(k + 1) * stride], x0[k * stride:(k + 1) * stride])
    # x_set: x = alpha r_in on whatever x held; r_out a different vector; r_out = NULL leaves r alone
    gx2, gr2, gout = mk(x0), mk(r0), mk(np.zeros_like(r0))
    qmg.batch_mr_dots(dt, gr2, dp, n, nrhs, stride, mask)
    qmg.batch_mr_update(dt, omega, gx2, gr2, gout, dp, True, n, nrhs, stride, mask)
    zx = mk(np.zeros_like(x0))
    qmg.batch_blas_t(dt, qmg.BOP_CAXPY, zx, n, nrhs, stride, mask, a=alpha, x=dr)
    for k in act:
        sl = slice(k * stride, k * stride + n)
        assert np.array_equal(gx2.to_host()[sl], zx.to_host()[sl]) and np.array_equal(gout.to_host()[sl], wr.to_host()[sl])
    assert np.array_equal(gr2.to_host(), r0)
    qmg.batch_mr_update(dt, omega, gx2, gr2, None, dp, True, n, nrhs, stride, mask)
    assert np.array_equal(gr2.to_host(), r0)
    # breakdown: p = 0 -> <p,p> = 0 -> alpha = 0: x and r unchanged (accumulate form), x = 0 (x_set form)
    zp = mk(np.zeros_like(p0))
    gx3, gr3 = mk(x0), mk(r0)
    qmg.batch_mr_dots(dt, gr3, zp, n, nrhs, stride, mask)
    qmg.batch_mr_update(dt, omega, gx3, gr3, gr3, zp, False, n, nrhs, stride, mask)
    assert np.array_equal(gx3.to_host(), x0) and np.array_equal(gr3.to_host(), r0)


@pytest.mark.parametrize("nj,f32,with_z", [(0, False, True), (1, False, False), (3, False, True), (8, True, True), (11, False, True), (19, True, False)])
def test_gcr_update_is_the_three_separate_passes_bit_for_bit(nj, f32, with_z):
    """qmg_batch_gcr_update_t (w += sum_j c_j W_j ; r += a w ; z_next = r in one launch: the vector updates of one flexible-GCR iteration, the
    caller of the K-cycle on both sides of the hot path) against qmg_batch_multi_caxpy_t, qmg_batch_blas_t(CAXPY), qmg_batch_blas_t(COPY) in that
    order: identical bits in both storage precisions, frozen systems and the padding between systems untouched."""
    nrhs, mask, n = 5, 0b10111, 4098
    stride = n + 6
    vt, dt = (np.complex64, qmg.C32) if f32 else (np.complex128, qmg.C64)
    Ws = [cs.gaussian_cvec(stride * nrhs, 40 + j).astype(vt) for j in range(nj)]
    w0, r0, z0 = (cs.gaussian_cvec(stride * nrhs, s).astype(vt) for s in (71, 72, 73))
    coeffs = cs.gaussian_cvec(max(nj, 1) * nrhs, 8).reshape(max(nj, 1), nrhs)[:nj]
    if nj >= 3:
        coeffs[1, 2] = 0.0                      # "system 2 does not own direction 1"
    a = cs.gaussian_cvec(nrhs, 9)
    dW = [D(v) for v in Ws]
    w1, r1, z1 = D(w0), D(r0), D(z0)
    if nj:
        qmg.batch_multi_caxpy_t(dt, coeffs, dW, w1, n, nrhs, stride, mask)
    qmg.batch_blas_t(dt, qmg.BOP_CAXPY, r1, n, nrhs, stride, mask, a=a, x=w1)
    if with_z:
        qmg.batch_blas_t(dt, qmg.BOP_COPY, z1, n, nrhs, stride, mask, x=r1)
    w2, r2, z2 = D(w0), D(r0), D(z0)
    qmg.batch_gcr_update_t(dt, coeffs, dW, w2, a, r2, z2 if with_z else None, n, nrhs, stride, mask)
    assert np.array_equal(w1.to_host().view(np.uint8), w2.to_host().view(np.uint8))
    assert np.array_equal(r1.to_host().view(np.uint8), r2.to_host().view(np.uint8))
    assert np.array_equal(z1.to_host().view(np.uint8), z2.to_host().view(np.uint8))
    assert not np.array_equal(r2.to_host(), r0)
