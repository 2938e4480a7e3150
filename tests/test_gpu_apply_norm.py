"""qmg_stencil_apply_norm2: the apply with |lhs_k|^2 of its results from the same pass (kernel A2 with NORM).
The vectors must be the BYTES qmg_stencil_apply writes (same kernel body, same order of operations), the norms must agree
with the oracle's norm of those vectors to rounding (the kernel's fixed summation order is not qmg_norm2sq's) and be
reproducible from run to run; the cases cover half rows that do not fill a block (dead lane groups stay in the wavefront
reductions), odd row counts per group, accumulation into lhs, and the refusals."""
import importlib

import numpy as np
import pytest

import coordspace as cs
import oracle_lib as ol

qmg = importlib.import_module("quantum-mg_amd")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _device():
    qmg.build()
    qmg.init(0)
    yield
    qmg.sync()


def D(a):
    return qmg.DeviceArray.from_host(np.ascontiguousarray(a, dtype=np.complex128))


def operator(Lx, Ly, nc, seed):
    vol = Lx * Ly
    clover = cs.gaussian_cvec(vol * nc * nc, seed)
    hopping = cs.gaussian_cvec(4 * vol * nc * nc, seed + 1)
    shifts = (0.3 - 0.2j, 0.11, -0.07 + 0.02j)
    return clover, hopping, shifts


@pytest.mark.parametrize("Lx,Ly,nc,nrhs", [(12, 6, 1, 3), (12, 6, 2, 5), (64, 10, 1, 8), (1030, 4, 1, 2), (260, 6, 2, 16), (32, 32, 1, 1), (36, 2, 2, 7)])
def test_apply_norm2_matches_apply_then_norm(Lx, Ly, nc, nrhs):
    vol = Lx * Ly
    size = vol * nc
    stride = size + 4
    clover, hopping, shifts = operator(Lx, Ly, nc, 1)
    rhs = cs.gaussian_cvec(stride * nrhs, 3)
    lhs0 = cs.gaussian_cvec(stride * nrhs, 4)
    gd = qmg.make_desc(Lx, Ly, nc, D(clover), D(hopping), *shifts)
    od = ol.make_desc(Lx, Ly, nc, clover, hopping, *shifts)
    drhs = D(rhs)
    for pieces in (ol.P_ALL | ol.P_ZERO, ol.P_ALL, ol.P_EO | ol.P_OE | ol.P_ZERO, ol.P_CLOVER | ol.P_SHIFT | ol.P_ZERO):
        ref = D(lhs0)
        qmg.stencil_apply(gd, ref, drhs, pieces, nrhs, stride)
        want = ref.to_host()
        dl = D(lhs0)
        norms = qmg.stencil_apply_norm2(gd, dl, drhs, pieces, nrhs, stride)
        got = dl.to_host()
        assert np.array_equal(got, want), hex(pieces)                 # the bytes of the plain apply (padding untouched too)
        for k in range(nrhs):
            vk = want[k * stride:k * stride + size]
            exact = float(np.vdot(vk, vk).real)
            assert abs(norms[k] - exact) <= 1e-13 * exact, (k, norms[k], exact)
            assert abs(norms[k] - qmg.norm2sq(D(vk), size)) <= 1e-13 * exact
        # against the oracle's apply as well (stencil_2d.h:666-936), one system
        o = lhs0[:size].copy()
        ol.stencil_apply(od, np.ascontiguousarray(rhs[:size]), pieces, lhs=o)
        assert cs.rel_l2(got[:size], o) < 1e-13
        # run-to-run reproducible: fixed-order partial sums, no atomics
        again = qmg.stencil_apply_norm2(gd, D(lhs0), drhs, pieces, nrhs, stride)
        assert np.array_equal(norms, again)


def test_apply_norm2_device_result_staggered_shape():
    """norms left in device memory (no synchronisation), on a staggered operator (nc = 1, hops only + mass shift)."""
    L, nrhs = 32, 8
    vol = L * L
    hopping = cs.gaussian_cvec(4 * vol, 21)
    gd = qmg.make_desc(L, L, 1, None, D(hopping), 0.1)
    rhs = cs.gaussian_cvec(vol * nrhs, 22)
    dl = qmg.DeviceArray.zeros(vol * nrhs)
    dn = D(np.zeros(nrhs // 2 + nrhs % 2, dtype=np.complex128))    # nrhs doubles
    assert qmg.stencil_apply_norm2(gd, dl, D(rhs), ol.P_ALL | ol.P_ZERO, nrhs, vol, norms_dev=dn.ptr) is None
    qmg.sync()
    got = dn.to_host().view(np.float64)[:nrhs]
    out = dl.to_host()
    for k in range(nrhs):
        vk = out[k * vol:(k + 1) * vol]
        exact = float(np.vdot(vk, vk).real)
        assert abs(got[k] - exact) <= 1e-13 * exact


def test_apply_norm2_on_two_streams_of_one_thread():
    """The fused-norm partials live in ONE buffer per calling thread: calls that alternate between two streams must not overlap on it (the library
    orders them with an event).  Two operators of different size, 24 alternating launches with device-resident results and no host
    synchronisation in between; every result must be the one the same call gives alone."""
    s1, s2 = qmg.stream_create(), qmg.stream_create()
    try:
        ops = []
        for L, nc, nrhs, seed in ((64, 2, 4, 31), (48, 1, 8, 41)):
            vol = L * L
            clover = cs.gaussian_cvec(vol * nc * nc, seed) if nc == 2 else None
            hopping = cs.gaussian_cvec(4 * vol * nc * nc, seed + 1)
            gd = qmg.make_desc(L, L, nc, D(clover) if clover is not None else None, D(hopping), 0.1)
            rhs = D(cs.gaussian_cvec(vol * nc * nrhs, seed + 2))
            lhs = qmg.DeviceArray.zeros(vol * nc * nrhs)
            alone = qmg.stencil_apply_norm2(gd, lhs, rhs, ol.P_ALL | ol.P_ZERO, nrhs, vol * nc)
            outs = [D(np.zeros(nrhs // 2 + nrhs % 2, dtype=np.complex128)) for _ in range(12)]
            ops.append((gd, lhs, rhs, nrhs, vol * nc, alone, outs))
        for i in range(12):
            for (gd, lhs, rhs, nrhs, stride, alone, outs), st in zip(ops, (s1, s2)):
                qmg.stencil_apply_norm2(gd, lhs, rhs, ol.P_ALL | ol.P_ZERO, nrhs, stride, norms_dev=outs[i].ptr, stream=st)
        qmg.sync(s1)
        qmg.sync(s2)
        for gd, lhs, rhs, nrhs, stride, alone, outs in ops:
            for o in outs:
                assert np.array_equal(o.to_host().view(np.float64)[:nrhs], np.asarray(alone)), (nrhs, o.to_host(), alone)
    finally:
        qmg.stream_destroy(s1)
        qmg.stream_destroy(s2)


def test_apply_norm2_refusals():
    Lx, Ly = 12, 6
    for nc in (1, 4):
        vol = Lx * Ly
        clover, hopping, shifts = operator(Lx, Ly, nc, 5)
        gd = qmg.make_desc(Lx, Ly, nc, D(clover), D(hopping), *shifts)
        v = D(cs.gaussian_cvec(vol * nc, 6))
        w = D(cs.gaussian_cvec(vol * nc, 7))
        before = w.to_host()
        if nc == 4:      # only nc = 1, 2
            with pytest.raises(qmg.QmgError):
                qmg.stencil_apply_norm2(gd, w, v, ol.P_ALL | ol.P_ZERO)
        else:
            with pytest.raises(qmg.QmgError):   # one parity untouched: its |lhs|^2 would be missing
                qmg.stencil_apply_norm2(gd, w, v, ol.P_EO | ol.P_ZERO_E)
            with pytest.raises(qmg.QmgError):   # in place
                qmg.stencil_apply_norm2(gd, v, v, ol.P_ALL | ol.P_ZERO)
            with pytest.raises(qmg.QmgError):   # more than 16 systems
                qmg.stencil_apply_norm2(gd, w, v, ol.P_ALL | ol.P_ZERO, 17, vol)
        assert np.array_equal(w.to_host(), before)


@pytest.mark.parametrize("Lx,Ly,nrhs", [(12, 6, 3), (1030, 4, 8), (64, 12, 16), (20, 3 * 2, 2)])
def test_next_system_prefetch_changes_no_byte(Lx, Ly, nrhs):
    """Kernel A2's prefetch of system k+1 (nc = 1 batches, tuning key "pair_prefetch") is a scheduling change only."""
    vol = Lx * Ly
    stride = vol + 2
    clover, hopping, shifts = operator(Lx, Ly, 1, 11)
    gd = qmg.make_desc(Lx, Ly, 1, D(clover), D(hopping), *shifts)
    rhs = D(cs.gaussian_cvec(stride * nrhs, 12))
    lhs0 = cs.gaussian_cvec(stride * nrhs, 13)
    out = {}
    try:
        for pf in (0, 1):
            qmg.set_tuning("pair_prefetch", pf)
            for pieces in (ol.P_ALL | ol.P_ZERO, ol.P_ALL):
                a, b = D(lhs0), D(lhs0)
                qmg.stencil_apply(gd, a, rhs, pieces, nrhs, stride)
                n = qmg.stencil_apply_norm2(gd, b, rhs, pieces, nrhs, stride)
                out[pf, pieces] = (a.to_host(), b.to_host(), n)
    finally:
        qmg.set_tuning("pair_prefetch", 1)
    for pieces in (ol.P_ALL | ol.P_ZERO, ol.P_ALL):
        for i in range(3):
            assert np.array_equal(out[0, pieces][i], out[1, pieces][i])
        assert np.array_equal(out[0, pieces][0], out[0, pieces][1])
