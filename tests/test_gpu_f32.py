"""GPU parity of the fp32 instantiation (qmg_dtype QMG_C32; BASELINE configs[4] "fp32"): every `_t` entry point with
complex<float> storage against the fp64 CPU oracle applied to the SAME inputs after rounding them to fp32.

Bar: relative L2 <= 5e-6 per apply.  The fine kernels (nc = 1, 2, 4) compute in fp32; the coarse kernels,
the transfer and the BLAS-1 leaves keep fp64 registers and round once on store (~6e-8); reductions accumulate in fp64,
so on fp32-representable inputs they agree with the oracle to 1e-12."""
import importlib
import os
import re
import subprocess

import numpy as np
import pytest

import coordspace as cs
import oracle_lib as ol

qmg = importlib.import_module("quantum-mg_amd")

pytestmark = pytest.mark.gpu

TOL32 = 5e-6
TOL32_ROUND = 3e-7    # fp64 arithmetic, one fp32 rounding of the result
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVERS = os.path.join(ROOT, "quantum-mg_amd", "drivers")


@pytest.fixture(scope="module", autouse=True)
def _device():
    qmg.build()
    qmg.init(0)
    yield
    qmg.sync()


def r32(a):
    """round to complex<float> and widen back: what the device arrays hold"""
    return np.ascontiguousarray(a, dtype=np.complex64).astype(np.complex128)


def D32(a):
    return qmg.DeviceArray.from_host(np.ascontiguousarray(a, dtype=np.complex64))


def H(dev):
    return dev.to_host().astype(np.complex128)


PIECE_SETS = [
    ("all_zero", ol.P_ALL | ol.P_ZERO), ("all_accumulate", ol.P_ALL), ("clover", ol.P_CLOVER), ("hopping", ol.P_HOPPING),
    ("eo", ol.P_EO), ("oe", ol.P_OE), ("shift", ol.P_SHIFT), ("ee", ol.P_CLOVER_E | ol.P_SHIFT_E),
    ("oo_zero", ol.P_CLOVER_O | ol.P_SHIFT_O | ol.P_ZERO_O), ("eo_xp1", ol.P_EO_XP1), ("oe_ym1", ol.P_OE_XP1 << 3),
    ("dir_xm1_both", (ol.P_EO_XP1 << 2) | (ol.P_OE_XP1 << 2)), ("zero_only", ol.P_ZERO_E),
]


@pytest.mark.parametrize("name,pieces", PIECE_SETS)
@pytest.mark.parametrize("Lx,Ly,nc", [(32, 32, 2), (6, 4, 2), (34, 10, 1), (16, 12, 4), (12, 8, 3), (16, 6, 8), (8, 8, 24)])
def test_every_piece_mask_every_kernel_f32(name, pieces, Lx, Ly, nc):
    """The piece-mask matrix of test_gpu_parity.py::test_every_piece_mask_every_kernel in fp32: kernel A in fp32 arithmetic
    (nc 1, 2, 4), kernels B / B32 with fp32 tiles and fp32 vectors (nc 3, 8, 24); accumulate vs overwrite, untouched
    halves, ragged tiles, all three shifts."""
    vol = Lx * Ly
    clover, hopping = r32(cs.gaussian_cvec(vol * nc * nc, 1)), r32(cs.gaussian_cvec(4 * vol * nc * nc, 2))
    rhs, lhs0 = r32(cs.gaussian_cvec(vol * nc, 3)), r32(cs.gaussian_cvec(vol * nc, 4))
    shifts = (0.3 - 0.1j, 0.05 + 0.02j, -0.07j)
    want = ol.stencil_apply(ol.make_desc(Lx, Ly, nc, clover, hopping, *shifts), rhs, pieces, lhs=lhs0.copy())
    dc, dh, dr, dl = D32(clover), D32(hopping), D32(rhs), D32(lhs0)
    qmg.stencil_apply_t(qmg.C32, qmg.make_desc(Lx, Ly, nc, dc, dh, *shifts), dl, dr, pieces)
    got = H(dl)
    assert cs.rel_l2(got, want) < (TOL32 if nc in (1, 2, 4) else TOL32_ROUND), cs.rel_l2(got, want)
    # a half no piece touches is bit-for-bit untouched
    if name == "eo":
        assert np.array_equal(got[vol * nc // 2:], lhs0[vol * nc // 2:])


@pytest.mark.parametrize("nc,nrhs,mask", [(2, 3, 0b101), (1, 8, 0xFF), (8, 3, 0b110), (8, 7, 0x7F), (12, 6, 0b111011), (24, 5, 0b11111), (24, 16, 0xFFFF), (16, 9, 0x1FF), (7, 4, 0b1011)])
def test_stencil_apply_f32_batches(nc, nrhs, mask):
    """Masked lock-step batches in fp32: kernel A's rhs loop, kernel B32 with 4 / 8 accumulators, kernel C (f64 MFMA over
    fp32 tiles and fp32 vectors, both the 2-MFMA and the 4-MFMA product).  Frozen systems are not written."""
    Lx, Ly = 16, 12
    vol = Lx * Ly
    size, stride = vol * nc, vol * nc + 6
    clover, hopping = r32(cs.gaussian_cvec(vol * nc * nc, 1)), r32(cs.gaussian_cvec(4 * vol * nc * nc, 2))
    rhs, lhs0 = r32(cs.gaussian_cvec(nrhs * stride, 3)), r32(cs.gaussian_cvec(nrhs * stride, 4))
    shifts = (0.2 + 0.1j, 0.0, 0.03)
    d = ol.make_desc(Lx, Ly, nc, clover, hopping, *shifts)
    dc, dh, dr, dl = D32(clover), D32(hopping), D32(rhs), D32(lhs0)
    qmg.stencil_apply_t(qmg.C32, qmg.make_desc(Lx, Ly, nc, dc, dh, *shifts), dl, dr, ol.P_ALL | ol.P_ZERO, nrhs=nrhs, vec_stride=stride, mask=mask)
    got = H(dl)
    for k in range(nrhs):
        seg = slice(k * stride, k * stride + size)
        if (mask >> k) & 1:
            want = ol.stencil_apply(d, rhs[seg].copy())
            assert cs.rel_l2(got[seg], want) < (TOL32 if nc in (1, 2, 4) else TOL32_ROUND), (k, cs.rel_l2(got[seg], want))
        else:
            assert np.array_equal(got[seg], lhs0[seg])
        assert np.array_equal(got[k * stride + size:(k + 1) * stride], lhs0[k * stride + size:(k + 1) * stride])   # padding untouched


def test_wilson_fixture_apply_f32(golden_dir):
    """The reference's l64 U(1) fixture: fp32 Wilson apply (192 B/site) vs the fp64 oracle on the rounded operator."""
    L = 64
    gauge = ol.phases_to_gauge_u1(np.loadtxt(os.path.join(golden_dir, "l64t64b60_heatbath.dat")), L, L)
    clover, hopping = ol.wilson_fill(gauge, L, L)
    clover, hopping = r32(clover), r32(hopping)
    rhs = r32(cs.gaussian_cvec(2 * L * L, 5))
    want = ol.stencil_apply(ol.make_desc(L, L, 2, clover, hopping, -0.07), rhs)
    dl = qmg.DeviceArray(2 * L * L, np.complex64)
    qmg.stencil_apply_t(qmg.C32, qmg.make_desc(L, L, 2, D32(clover), D32(hopping), -0.07), dl, D32(rhs), ol.P_ALL | ol.P_ZERO)
    assert cs.rel_l2(H(dl), want) < 1e-6


@pytest.mark.parametrize("n,stride_pad,nrhs,mask", [(4096, 0, 1, 1), (1000, 2, 5, 0b10111), (4097, 1, 3, 0b101), (1 << 20, 0, 2, 0b11)])
def test_batch_blas_and_reductions_f32(n, stride_pad, nrhs, mask):
    """Element-wise leaves round once (3e-7); reductions accumulate in fp64 (1e-12 on fp32-representable inputs).  Odd n /
    odd stride exercise the 8-byte fallback of the 16-byte-per-lane kernels."""
    stride = n + stride_pad
    x, y, z = (r32(cs.gaussian_cvec(nrhs * stride, s)) for s in (1, 2, 3))
    a = np.array([0.3 - 0.2j + 0.1 * k for k in range(nrhs)])
    b = np.array([-0.7 + 0.05j * k for k in range(nrhs)])
    act = [k for k in range(nrhs) if (mask >> k) & 1]
    seg = lambda v, k: v[k * stride:k * stride + n]
    dx, dy = D32(x), D32(y)
    for op, ref in ((qmg.BOP_CAXPY, lambda k: seg(z, k) + a[k] * seg(x, k)), (qmg.BOP_CAXPBYZ, lambda k: a[k] * seg(x, k) + b[k] * seg(y, k)),
                    (qmg.BOP_COPY, lambda k: seg(x, k)), (qmg.BOP_CAX, lambda k: a[k] * seg(z, k)), (qmg.BOP_CXPY, lambda k: seg(z, k) + seg(x, k)),
                    (qmg.BOP_ZERO, lambda k: 0 * seg(z, k))):
        dz = D32(z)
        qmg.batch_blas_t(qmg.C32, op, dz, n, nrhs, stride, mask, a=a, b=b, x=dx, y=dy)
        got = H(dz)
        for k in range(nrhs):
            if k in act:
                assert np.linalg.norm(seg(got, k) - ref(k)) <= TOL32_ROUND * max(np.linalg.norm(ref(k)), 1.0), op
            else:
                assert np.array_equal(seg(got, k), seg(z, k))
    nrm = qmg.batch_reduce_t(qmg.C32, qmg.BRED_NORM2, dx, None, n, nrhs, stride, mask)
    dt = qmg.batch_reduce_t(qmg.C32, qmg.BRED_DOT, dx, dy, n, nrhs, stride, mask)
    df = qmg.batch_reduce_t(qmg.C32, qmg.BRED_DIFFNORM2, dx, dy, n, nrhs, stride, mask)
    for k in act:
        assert abs(nrm[k].real - np.vdot(seg(x, k), seg(x, k)).real) < 1e-12 * n
        assert abs(dt[k] - np.vdot(seg(x, k), seg(y, k))) < 1e-12 * n
        assert abs(df[k].real - np.linalg.norm(seg(x, k) - seg(y, k)) ** 2) < 1e-12 * n
    # multidot / multi_caxpy over 5 vector sets
    xs = [r32(cs.gaussian_cvec(nrhs * stride, 10 + j)) for j in range(5)]
    dxs = [D32(v) for v in xs]
    md = qmg.batch_multidot_t(qmg.C32, dxs, dy, n, nrhs, stride, mask)
    coeffs = np.array([[0.1 * (j + 1) - 0.05j * k for k in range(nrhs)] for j in range(5)])
    dz = D32(z)
    qmg.batch_multi_caxpy_t(qmg.C32, coeffs, dxs, dz, n, nrhs, stride, mask)
    got = H(dz)
    for k in act:
        for j in range(5):
            assert abs(md[k][j] - np.vdot(seg(xs[j], k), seg(y, k))) < 1e-12 * n
        ref = seg(z, k) + sum(coeffs[j][k] * seg(xs[j], k) for j in range(5))
        assert cs.rel_l2(seg(got, k), ref) < TOL32_ROUND


@pytest.mark.parametrize("fd,cd,nrhs,mask", [((32, 32, 2), (8, 8, 8), 1, 1), ((32, 32, 2), (8, 8, 8), 8, 0xFF), ((32, 32, 2), (8, 8, 24), 5, 0b11011), ((16, 16, 8), (4, 4, 8), 3, 0b111),
                                            ((16, 8, 24), (4, 2, 24), 2, 0b11), ((24, 12, 2), (12, 6, 6), 4, 0b1111), ((12, 12, 2), (4, 4, 6), 2, 0b11)])
def test_transfer_f32_single_and_tiled(fd, cd, nrhs, mask):
    """restrict / prolong with complex<float> null vectors and vectors: the one-system kernels (nrhs = 1), the LDS-tiled
    batch kernels (2, 4, 8 accumulators), 2x2 blocks, and the odd-block-width fallback (12 -> 4: bx = 3)."""
    fsize, csize = fd[0] * fd[1] * fd[2], cd[0] * cd[1] * cd[2]
    nvec = cd[2]
    nv = r32(cs.gaussian_cvec(nvec * fsize, 1))
    fine, coarse = r32(cs.gaussian_cvec(nrhs * fsize, 2)), r32(cs.gaussian_cvec(nrhs * csize, 3))
    dn = D32(nv)
    df, dc = D32(fine), D32(coarse)
    qmg.prolong_batch_t(qmg.C32, dn, nvec, dc, df, fd, cd, nrhs, csize, fsize, mask)
    got_f = H(df)
    df2, dc2 = D32(fine), D32(coarse)
    qmg.restrict_batch_t(qmg.C32, dn, nvec, df2, dc2, fd, cd, nrhs, fsize, csize, mask)
    got_c = H(dc2)
    for k in range(nrhs):
        fs, csl = slice(k * fsize, (k + 1) * fsize), slice(k * csize, (k + 1) * csize)
        if (mask >> k) & 1:
            assert cs.rel_l2(got_f[fs], ol.prolong(nv, coarse[csl].copy(), fd, cd, fine=fine[fs].copy())) < TOL32_ROUND
            assert cs.rel_l2(got_c[csl], ol.restrict(nv, fine[fs].copy(), fd, cd, coarse=coarse[csl].copy())) < TOL32_ROUND
        else:
            assert np.array_equal(got_f[fs], fine[fs]) and np.array_equal(got_c[csl], coarse[csl])


@pytest.mark.parametrize("fd,cd,nrhs,mask", [((32, 32, 2), (8, 8, 8), 1, 1), ((64, 32, 2), (16, 8, 24), 3, 0b010), ((16, 16, 8), (4, 4, 8), 2, 0b11), ((16, 8, 24), (4, 2, 24), 1, 1),
                                            ((24, 12, 2), (12, 6, 6), 2, 0b01)])
def test_transfer_with_narrow_null_vectors_under_fp64_vectors(fd, cd, nrhs, mask):
    """qmg_prolong_batch_nv32 / qmg_restrict_batch_nv32 (transfer/transfer.h:455-511 on complex<double> vectors with the null vectors stored as complex<float>:
    what the K-cycle's own transfers stream in a hierarchy that only preconditions): the fp64 oracle on the ROUNDED null vectors to 1e-13 -- the arithmetic is
    fp64, only the storage of the null vectors is narrow; frozen systems untouched."""
    fsize, csize = fd[0] * fd[1] * fd[2], cd[0] * cd[1] * cd[2]
    nvec = cd[2]
    nv = r32(cs.gaussian_cvec(nvec * fsize, 1))                      # (values exactly representable in complex<float>)
    fine, coarse = cs.gaussian_cvec(nrhs * fsize, 2), cs.gaussian_cvec(nrhs * csize, 3)
    dn = D32(nv)
    D = qmg.DeviceArray.from_host
    df, dc = D(fine), D(coarse)
    qmg.prolong_batch_nv32(dn, nvec, dc, df, fd, cd, nrhs, csize, fsize, mask)
    got_f = df.to_host()
    df2, dc2 = D(fine), D(coarse)
    qmg.restrict_batch_nv32(dn, nvec, df2, dc2, fd, cd, nrhs, fsize, csize, mask)
    got_c = dc2.to_host()
    nv64 = nv.astype(np.complex128)
    for k in range(nrhs):
        fs, csl = slice(k * fsize, (k + 1) * fsize), slice(k * csize, (k + 1) * csize)
        if (mask >> k) & 1:
            assert cs.rel_l2(got_f[fs], ol.prolong(nv64, coarse[csl].copy(), fd, cd, fine=fine[fs].copy())) < 1e-13
            assert cs.rel_l2(got_c[csl], ol.restrict(nv64, fine[fs].copy(), fd, cd, coarse=coarse[csl].copy())) < 1e-13
        else:
            assert np.array_equal(got_f[fs], fine[fs]) and np.array_equal(got_c[csl], coarse[csl])


def test_convert_round_trip():
    x = cs.gaussian_cvec(10007, 9)
    d64, d32, back = qmg.DeviceArray.from_host(x), qmg.DeviceArray(10007, np.complex64), qmg.DeviceArray(10007)
    qmg.convert(d32, qmg.C32, d64, qmg.C64, 10007)
    qmg.convert(back, qmg.C64, d32, qmg.C32, 10007)
    assert np.array_equal(d32.to_host(), x.astype(np.complex64))
    assert np.array_equal(back.to_host(), x.astype(np.complex64).astype(np.complex128))


@pytest.mark.parametrize("args,extra", [(["128", "-0.06", "6.0", "2", "8"], "128"), (["256", "-0.07", "6.0", "2", "24"], "64")])
def test_fp32_kcycle_preconditions_the_fp64_solve(golden_dir, args, extra):
    """n13 K-cycle with the whole preconditioner in fp32 (complex<float> hierarchy, mg_preconditioner_batch_mixed) inside
    the fp64 outer VPGCR: every system still reaches the fp64 tolerance 1e-10 (true residual), in (about) the iteration
    count of the all-fp64 solve."""
    gauge_file = os.path.join(golden_dir, "l%st%sb60_heatbath.dat" % (extra, extra))
    its = {}
    for tag, tail in (("f64", []), ("f32", ["f32"])):
        out = subprocess.run([os.path.join(DRIVERS, "n13_wilson_kcycle_mrhs")] + args + [gauge_file, extra, "3"] + tail, cwd=DRIVERS,
                             env=dict(os.environ, QMG_QUIET="1"), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        assert ("K-cycle preconditioner in fp32" in out.stdout) == (tag == "f32")
        rows = re.findall(r"\[QMG-MRHS\]: rhs (\d+) converged in (\d+) iterations ; alleged tolerance ([-\d.e+]+) ; check tolerance ([-\d.e+]+)", out.stdout)
        assert len(rows) == 3 and all(float(r[3]) <= 1.05e-10 for r in rows), out.stdout[-1500:]
        its[tag] = [int(r[1]) for r in rows]
    assert all(abs(a - b) <= 2 for a, b in zip(its["f64"], its["f32"])), its


HALF_PIECES = [("all_zero", ol.P_ALL | ol.P_ZERO), ("all_accumulate", ol.P_ALL), ("eo_inplace_like", ol.P_EO | ol.P_ZERO_E), ("oe", ol.P_OE), ("clover_shift_o", ol.P_CLOVER_O | ol.P_SHIFT_O | ol.P_ZERO_O),
               ("dir_ym1_both", (ol.P_EO_XP1 << 3) | (ol.P_OE_XP1 << 3))]


@pytest.mark.parametrize("name,pieces", HALF_PIECES)
@pytest.mark.parametrize("Lx,Ly", [(32, 32), (6, 4), (520, 10)])
def test_fine_apply_with_16_bit_stored_matrices(name, pieces, Lx, Ly):
    """qmg_stencil_apply_h16: complex<half> matrices, complex<float> vectors, fp32 arithmetic, one lane per site.
    Against the fp64 oracle applied to the SAME matrices after rounding to half and the same fp32 vectors: 5e-6."""
    nc, vol = 2, Lx * Ly
    h16 = lambda v: (v.real.astype(np.float16).astype(np.float64) + 1j * v.imag.astype(np.float16).astype(np.float64))
    clover, hopping = h16(cs.gaussian_cvec(vol * 4, 1)), h16(cs.gaussian_cvec(4 * vol * 4, 2))
    rhs, lhs0 = r32(cs.gaussian_cvec(vol * nc, 3)), r32(cs.gaussian_cvec(vol * nc, 4))
    shifts = (0.3 - 0.1j, 0.05 + 0.02j, -0.07j)
    want = ol.stencil_apply(ol.make_desc(Lx, Ly, nc, clover, hopping, *shifts), rhs, pieces, lhs=lhs0.copy())
    dc64, dh64 = qmg.DeviceArray.from_host(clover), qmg.DeviceArray.from_host(hopping)
    dc, dh = qmg.DeviceArray(vol * 4, np.float32), qmg.DeviceArray(4 * vol * 4, np.float32)      # 4 bytes per complex<half>
    qmg.convert_to_c16(dc, dc64, qmg.C64, vol * 4)
    qmg.convert_to_c16(dh, dh64, qmg.C64, 4 * vol * 4)
    dr, dl = D32(rhs), D32(lhs0)
    qmg.stencil_apply_h16(qmg.make_desc(Lx, Ly, nc, dc, dh, *shifts), dl, dr, pieces)
    assert cs.rel_l2(H(dl), want) < TOL32
    # masked batch: system 1 of 3 frozen
    size = vol * nc
    rb, lb = r32(cs.gaussian_cvec(3 * size, 5)), r32(cs.gaussian_cvec(3 * size, 6))
    drb, dlb = D32(rb), D32(lb)
    qmg.stencil_apply_h16(qmg.make_desc(Lx, Ly, nc, dc, dh, *shifts), dlb, drb, ol.P_ALL | ol.P_ZERO, nrhs=3, vec_stride=size, mask=0b101)
    got = H(dlb)
    for k in range(3):
        seg = slice(k * size, (k + 1) * size)
        if k == 1:
            assert np.array_equal(got[seg], lb[seg])
        else:
            assert cs.rel_l2(got[seg], ol.stencil_apply(ol.make_desc(Lx, Ly, nc, clover, hopping, *shifts), rb[seg].copy())) < TOL32


@pytest.mark.parametrize("name,pieces", HALF_PIECES)
@pytest.mark.parametrize("Lx,Ly,nc", [(16, 6, 8), (12, 8, 12), (8, 8, 24), (34, 4, 16)])
def test_coarse_apply_with_16_bit_stored_matrices(name, pieces, Lx, Ly, nc):
    """qmg_stencil_apply_mat16_t (kernel B32 with complex<half> matrices widened on their way into LDS; nc a multiple of 4): fp64 vectors against
    the fp64 oracle on the SAME matrices after rounding to half (1e-13: only summation order differs), complex<float> vectors to 2e-6; a masked
    batch of 3 (the 4-system pass), a batch of 7 (the 8-system pass), and the MR / residual epilogue (mat32 = 2)."""
    vol, size = Lx * Ly, Lx * Ly * nc
    h16 = lambda v: (v.real.astype(np.float16).astype(np.float64) + 1j * v.imag.astype(np.float16).astype(np.float64))
    clover, hopping = h16(cs.gaussian_cvec(vol * nc * nc, 1)), h16(cs.gaussian_cvec(4 * vol * nc * nc, 2))
    shifts = (0.3 - 0.1j, 0.05 + 0.02j, -0.07j)
    od = ol.make_desc(Lx, Ly, nc, clover, hopping, *shifts)
    dc64, dh64 = qmg.DeviceArray.from_host(clover), qmg.DeviceArray.from_host(hopping)
    dc, dh = qmg.DeviceArray(vol * nc * nc, np.float32), qmg.DeviceArray(4 * vol * nc * nc, np.float32)      # 4 bytes per complex<half>
    qmg.convert_to_c16(dc, dc64, qmg.C64, vol * nc * nc)
    qmg.convert_to_c16(dh, dh64, qmg.C64, 4 * vol * nc * nc)
    back = qmg.DeviceArray(vol * nc * nc)
    qmg.convert_from_c16(back, qmg.C64, dc, vol * nc * nc)
    assert np.array_equal(back.to_host(), clover)                     # the conversion pair is exact on half-representable values
    gd = qmg.make_desc(Lx, Ly, nc, dc, dh, *shifts)
    rhs, lhs0 = cs.gaussian_cvec(size, 3), cs.gaussian_cvec(size, 4)
    want = ol.stencil_apply(od, rhs, pieces, lhs=lhs0.copy())
    dl = qmg.DeviceArray.from_host(lhs0)
    assert qmg.stencil_apply_mat16(qmg.C64, gd, dl, qmg.DeviceArray.from_host(rhs), pieces) == 0
    assert cs.rel_l2(dl.to_host(), want) < 1e-13
    r32v, l32v = r32(rhs), r32(lhs0)
    dl32 = D32(l32v)
    assert qmg.stencil_apply_mat16(qmg.C32, gd, dl32, D32(r32v), pieces) == 0
    assert cs.rel_l2(H(dl32), ol.stencil_apply(od, r32v, pieces, lhs=l32v.copy())) < 2e-6
    # batches: 2 of 3 (kernel B32's 4-system pass), 7 (kernel C, packed columns), 12 of 13 (kernel C, plain columns)
    for nrhs, mask in ((3, 0b101), (7, 0x7F), (13, 0x1FFF & ~0b100)):
        rb, lb = cs.gaussian_cvec(nrhs * size, 5), cs.gaussian_cvec(nrhs * size, 6)
        for vdt, tol in ((qmg.C64, 1e-13), (qmg.C32, 2e-6)):
            rbv, lbv = (rb, lb) if vdt == qmg.C64 else (r32(rb), r32(lb))
            dlb = qmg.DeviceArray.from_host(lbv) if vdt == qmg.C64 else D32(lbv)
            drb = qmg.DeviceArray.from_host(rbv) if vdt == qmg.C64 else D32(rbv)
            assert qmg.stencil_apply_mat16(vdt, gd, dlb, drb, pieces, nrhs=nrhs, vec_stride=size, mask=mask) == 0
            got = dlb.to_host() if vdt == qmg.C64 else H(dlb)
            for k in range(nrhs):
                seg = slice(k * size, (k + 1) * size)
                if not (mask >> k) & 1:
                    assert np.array_equal(got[seg], lbv[seg])
                else:
                    assert cs.rel_l2(got[seg], ol.stencil_apply(od, rbv[seg].copy(), pieces, lhs=lbv[seg].copy())) < tol, (nrhs, k, vdt)
    if name == "all_zero":   # epilogue on the 16-bit storage: out = b - A x, and p = A r with the MR dots
        b = cs.gaussian_cvec(size, 7)
        out = qmg.DeviceArray.zeros(size)
        db, dx = qmg.DeviceArray.from_host(b), qmg.DeviceArray.from_host(rhs)
        assert qmg.stencil_apply_epi(qmg.C64, 2, gd, out, dx, pieces, qmg.make_epilogue(db, 1.0, -1.0, None)) == 0
        Ax = ol.stencil_apply(od, rhs)
        assert cs.rel_l2(out.to_host(), b - Ax) < 1e-13
        assert qmg.stencil_apply_epi(qmg.C64, 2, gd, out, dx, pieces, qmg.make_epilogue(None, 0.0, 1.0, dx)) == 0
        dots = qmg.batch_mr_read_dots(1)[0]
        p = out.to_host()
        assert abs(complex(dots[0], dots[1]) - np.vdot(p, rhs)) <= 1e-12 * abs(np.vdot(p, rhs)) and abs(dots[2] - np.vdot(p, p).real) <= 1e-12 * np.vdot(p, p).real
    # refused where kernel B32 does not serve the storage
    assert qmg.stencil_apply_mat16(qmg.C64, qmg.make_desc(8, 8, 6, dc, dh), dl, dl, pieces) == 3


@pytest.mark.parametrize("variant", ["", "schur"])
def test_kcycle_with_16_bit_fine_matrices_still_reaches_fp64_tolerance(golden_dir, variant):
    """QMG_F16_FINE=1: inside the fp32 K-cycle the level-0 matrices are streamed in 16 bits (operator perturbed by ~5e-4);
    the fp64 outer VPGCR still converges to 1e-10 (true residual of the ORIGINAL fp64 operator) in about the same count."""
    gauge_file = os.path.join(golden_dir, "l64t64b60_heatbath.dat")
    its = {}
    for tag, env in (("f32", {}), ("f16", {"QMG_F16_FINE": "1"}), ("f16c", {"QMG_F16_COARSE": "1"}), ("f16fc", {"QMG_F16_FINE": "1", "QMG_F16_COARSE": "1"})):
        cmd = [os.path.join(DRIVERS, "n22_wilson_kcycle_adaptive"), "256", "-0.07", "6.0", "2", "1", gauge_file, "64", "nrhs=2", "f32"] + ([variant] if variant else [])
        out = subprocess.run(cmd, cwd=DRIVERS, env=dict(os.environ, QMG_QUIET="1", **env), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        assert "[QMG-ERROR]" not in out.stdout
        assert ("fine-level matrices of the K-cycle stored in 16 bits" in out.stdout) == ("QMG_F16_FINE" in env)
        assert ("coarse-level matrices of the K-cycle stored in 16 bits" in out.stdout) == ("QMG_F16_COARSE" in env)     # (kernels B32 / C with complex<half> matrices)
        rows = re.findall(r"\[QMG-MRHS\]: rhs (\d+) converged in (\d+) iterations ; alleged tolerance ([-\d.e+]+) ; check tolerance ([-\d.e+]+)", out.stdout)
        assert len(rows) == 2 and all(float(r[3]) <= 1.05e-10 for r in rows), out.stdout[-1500:]
        its[tag] = [int(r[1]) for r in rows]
    for tag in ("f16", "f16c", "f16fc"):
        assert all(b <= a + max(3, a // 8) for a, b in zip(its["f32"], its[tag])), its
