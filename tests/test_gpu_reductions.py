"""reductions/reductions.h on the GPU against the oracle (VERDICT r02 missing 3): the three per-timeslice reductions
(`norm2sq_cv_timeslice` :24-41, `redot_cv_timeslice` :47-66, `dot_cv_timeslice` :69-87) and `gaussian_wall_source` (:90-162).
Tolerance: reductions 1e-12 relative (summation order differs: one block per timeslice against the reference's element loop);
the wall source's support pattern (which elements are non-zero, zero imaginary parts) is exact, its values agree to 1e-13
(device and host libm differ in the last bits of log / cos)."""
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


@pytest.mark.parametrize("Lx,Ly,nc", [(6, 4, 2), (32, 24, 1), (34, 10, 3), (2, 2, 2), (64, 64, 2), (16, 16, 24), (256, 128, 2)])
def test_timeslice_reductions(Lx, Ly, nc):
    n = Lx * Ly * nc
    a, b = cs.gaussian_cvec(n, 31), cs.gaussian_cvec(n, 32)
    da, db = D(a), D(b)
    want_n = ol.norm2sq_cv_timeslice(a, Lx, Ly, nc)
    want_d = ol.dot_cv_timeslice(a, b, Lx, Ly, nc)
    want_r = ol.redot_cv_timeslice(a, b, Lx, Ly, nc)
    scale = np.sqrt(want_n * ol.norm2sq_cv_timeslice(b, Lx, Ly, nc))         # |<a,b>| <= |a| |b| per slice: the natural scale of the sums
    assert np.allclose(qmg.norm2sq_cv_timeslice(da, Lx, Ly, nc), want_n, rtol=1e-12, atol=0)
    assert np.all(np.abs(qmg.dot_cv_timeslice(da, db, Lx, Ly, nc) - want_d) <= 1e-12 * scale)
    got_r = qmg.redot_cv_timeslice(da, db, Lx, Ly, nc)
    assert got_r.shape == (Ly,) and np.all(np.abs(got_r - want_r) <= 1e-12 * scale)
    assert np.array_equal(got_r, qmg.dot_cv_timeslice(da, db, Lx, Ly, nc).real)          # same kernel, same order: bit for bit the real parts
    # the slices add up to the global reductions
    assert abs(qmg.norm2sq_cv_timeslice(da, Lx, Ly, nc).sum() - ol.norm2sq(a)) <= 1e-12 * ol.norm2sq(a)


@pytest.mark.parametrize("Lx,Ly,nc,t,c", [(6, 4, 2, 3, 1), (32, 24, 1, 0, 0), (34, 10, 3, 9, 2), (64, 64, 2, 17, 0), (16, 16, 24, 5, 23)])
def test_gaussian_wall_source(Lx, Ly, nc, t, c):
    n = Lx * Ly * nc
    dv = qmg.DeviceArray.from_host(cs.gaussian_cvec(n, 5))                     # stale contents must be overwritten everywhere
    assert qmg.gaussian_wall_source(dv, Lx, Ly, nc, t, c, 1337, 2.0, 0.5) == 0
    got = dv.to_host()
    want = ol.gaussian_wall_source(Lx, Ly, nc, t, c, 1337, deviation=2.0, mean=0.5)
    assert np.array_equal(got != 0, want != 0) and not got.imag.any()
    assert np.allclose(got.real, want.real, rtol=1e-13, atol=1e-13)
    g = cs.eo_to_grid(got, Lx, Ly, nc)
    mask = np.zeros(g.shape, dtype=bool)
    mask[:, t, c] = True
    assert not g[~mask].any() and np.all(g[mask] != 0)
    # out of range: refused, vector untouched (reductions.h:94-107 print and return)
    before = dv.to_host()
    assert qmg.gaussian_wall_source(dv, Lx, Ly, nc, Ly, c, 1) != 0 and qmg.gaussian_wall_source(dv, Lx, Ly, nc, t, nc, 1) != 0
    assert np.array_equal(dv.to_host(), before)


def test_wall_source_correlator_is_a_timeslice_sum():
    """How n15 / n16 / n20 use the pair: a wall source on timeslice t0, and norm2sq_cv_timeslice of a vector built from it; here the
    'propagator' is the source itself, whose norm lives on t0 alone and equals the global norm."""
    Lx, Ly, nc = 64, 32, 2
    dv = qmg.DeviceArray(Lx * Ly * nc)
    assert qmg.gaussian_wall_source(dv, Lx, Ly, nc, 7, 1, 99) == 0
    sl = qmg.norm2sq_cv_timeslice(dv, Lx, Ly, nc)
    assert sl[7] > 0 and not np.delete(sl, 7).any()
    assert abs(sl[7] - qmg.norm2sq(dv, Lx * Ly * nc)) <= 1e-12 * sl[7]
