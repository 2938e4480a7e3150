"""y-slab domain decomposition of one lattice (csrc/qmg_site.hip, csrc/qmg_comm.hip, csrc/qmg_fill.hip).

The reference is single-process (cshift/cshift_2d.h:39-42,72,89 only says "Becomes MPI"), so the pin is the path itself:
a lattice cut into R slabs, each slab applied with its neighbours' boundary rows as halos, must reproduce the rows of the
single-domain apply BIT FOR BIT (the per-site arithmetic is the same code).  One GPU box has one GPU, so the R slabs of
these tests live in one process and the "exchange" between different slabs is done by the test; the library's own exchange
is exercised on one rank (device copies = the periodic wrap) and through a real one-rank RCCL communicator
(QMG_COMM_FORCE_RCCL: ncclSend / ncclRecv to self inside a group, and the all-reduce of the distributed reductions).
More ranks need more GPUs: the driver's multi-GPU bench runs `also_slab_solve` (bench.py) on them."""
import importlib

import numpy as np
import pytest

import coordspace as cs

qmg = importlib.import_module("quantum-mg_amd")
pytestmark = pytest.mark.gpu
D = qmg.DeviceArray.from_host


@pytest.fixture(scope="module", autouse=True)
def _device():
    qmg.build()
    qmg.init(0)
    yield
    qmg.sync()


@pytest.fixture(autouse=True)
def _fp64_coarse_storage(monkeypatch):
    """The single-domain drivers store the Galerkin matrices of their preconditioner levels as complex<float> by default; slab mode
    streams the fp64 arrays (multigrid.hpp: coarse_f32_wanted).  These tests compare the two runs digit by digit, so the children
    of this module run with fp64 coarse storage on both sides."""
    monkeypatch.setenv("QMG_COARSE_F32", "0")
    # ... and with the same arithmetic on both sides: a slab run keeps the single-vector K-cycle of multigrid.hpp (its exchanges overlap the
    # interior) and the separate BLAS-1 passes, while the single-domain default is the batch engine with apply epilogues -- the same numbers
    # summed in a different order.  The digit-for-digit comparisons of this module are about the DECOMPOSITION, so both sides run the former.
    monkeypatch.setenv("QMG_KCYCLE_ENGINE", "single")
    monkeypatch.setenv("QMG_APPLY_EPILOGUE", "0")


def rows(a, Ly, per_row, y0, n):
    """rows y0 .. y0+n-1 of every (array, parity) plane of an even-odd array: a = [planes][2][Ly][per_row]"""
    planes = a.size // (2 * Ly * per_row)
    return a.reshape(planes, 2, Ly, per_row)[:, :, y0:y0 + n].reshape(-1).copy()


def random_gauge(L, seed):
    rng = np.random.default_rng(seed)
    return np.exp(1j * rng.uniform(-np.pi, np.pi, size=2 * L * L))


def to_storage(a, storage):
    if storage == "c64":
        return D(a)
    if storage == "c32":
        return D(a.astype(np.complex64))
    d = qmg.DeviceArray(a.size, np.float32)           # complex<half>: 4 bytes per element
    qmg.convert_to_c16(d, D(a), qmg.C64, a.size)
    return d


STORAGE = {"c64": (qmg.C64, np.complex128), "c32": (qmg.C32, np.complex64), "h16": (qmg.C32 | qmg.SLAB_H16, np.complex64)}


@pytest.mark.parametrize("R", [2, 4])
def test_slab_fill_equals_the_rows_of_the_global_fill(R):
    L = 32
    g = D(random_gauge(L, 3))
    cl, hp = qmg.DeviceArray(4 * L * L), qmg.DeviceArray(16 * L * L)
    qmg.wilson_fill(cl, hp, g, L, L, 1.0)
    cl_h, hp_h = cl.to_host(), hp.to_host()
    Ll = L // R
    for r in range(R):
        c, h = qmg.DeviceArray(4 * L * Ll), qmg.DeviceArray(16 * L * Ll)
        qmg.wilson_fill_slab(c, h, g, L, L, r * Ll, Ll, 1.0)
        assert np.array_equal(c.to_host(), rows(cl_h, L, (L // 2) * 4, r * Ll, Ll))
        assert np.array_equal(h.to_host(), rows(hp_h, L, (L // 2) * 4, r * Ll, Ll))
    import ctypes as C
    assert qmg.lib().qmg_wilson_fill_slab(C.c_void_p(cl.ptr), C.c_void_p(hp.ptr), C.c_void_p(g.ptr), L, L, 3, Ll, C.c_double(1.0), None) == 1   # odd y0


def single_domain_apply(storage, L, cl, hp, x, pieces, lhs0, nrhs, mask, shifts):
    dtype, vt = STORAGE[storage]
    dc, dh = to_storage(cl, storage), to_storage(hp, storage)
    d = qmg.make_desc(L, L, 2, dc, dh, *shifts)
    dx, dl = D(x.astype(vt)), D(lhs0.astype(vt))
    n = 2 * L * L
    if storage == "h16":
        qmg.stencil_apply_h16(d, dl, dx, pieces, nrhs, n, mask)
    elif storage == "c32":
        qmg.stencil_apply_t(qmg.C32, d, dl, dx, pieces, nrhs, n, mask)
    else:
        qmg.set_tuning("stencil_site", 7)             # the same kernel as the slab path, so the comparison is bit for bit
        qmg.stencil_apply_t(qmg.C64, d, dl, dx, pieces, nrhs, n, mask)
        qmg.set_tuning("stencil_site", 3)
    return dl.to_host()


@pytest.mark.parametrize("storage", ["c64", "c32", "h16"])
@pytest.mark.parametrize("R,split", [(2, False), (4, True), (8, True)])
def test_slab_apply_with_neighbour_rows_as_halos_equals_the_single_domain_apply(storage, R, split):
    L, nrhs, mask = 32, 3, 0b101
    dtype, vt = STORAGE[storage]
    g = D(random_gauge(L, 11))
    cl, hp = qmg.DeviceArray(4 * L * L), qmg.DeviceArray(16 * L * L)
    qmg.wilson_fill(cl, hp, g, L, L, 1.0)
    cl_h, hp_h = cl.to_host(), hp.to_host()
    n = 2 * L * L
    x = cs.gaussian_cvec(n * nrhs, 5)
    lhs0 = cs.gaussian_cvec(n * nrhs, 6)
    Ll = L // R
    nl = 2 * L * Ll
    row = (L // 2) * 2                                # complex elements of one parity's row
    shifts = (-0.07, 0.013, 0.021)
    P = qmg
    for pieces in (P.P_ALL | P.P_ZERO, P.P_ALL, P.P_EO | P.P_ZERO_E, P.P_OE | P.P_ZERO_O, P.P_EO | P.P_CLOVER_E | P.P_SHIFT_E,
                   P.P_CLOVER_E | P.P_CLOVER_O | P.P_EO | P.P_OE | P.P_ZERO):
        want = single_domain_apply(storage, L, cl_h, hp_h, x, pieces, lhs0, nrhs, mask, shifts)
        xs = x.reshape(nrhs, 2, L, row)
        for r in range(R):
            y0 = r * Ll
            dc, dh = to_storage(rows(cl_h, L, (L // 2) * 4, y0, Ll), storage), to_storage(rows(hp_h, L, (L // 2) * 4, y0, Ll), storage)
            d = qmg.make_desc(L, Ll, 2, dc, dh, *shifts)
            dx = D(np.concatenate([rows(x[k * n:(k + 1) * n], L, row, y0, Ll) for k in range(nrhs)]).astype(vt))
            dl = D(np.concatenate([rows(lhs0[k * n:(k + 1) * n], L, row, y0, Ll) for k in range(nrhs)]).astype(vt))
            lo = D(xs[:, :, (y0 - 1) % L].reshape(-1).astype(vt))            # [system][parity][hr][nc]
            hi = D(xs[:, :, (y0 + Ll) % L].reshape(-1).astype(vt))
            if split:   # interior first (it needs no halo), then the two boundary rows
                qmg.stencil_apply_slab(dtype, d, dl, dx, lo, hi, pieces, nrhs, nl, 2 * row, mask, rows=1)
                qmg.stencil_apply_slab(dtype, d, dl, dx, lo, hi, pieces, nrhs, nl, 2 * row, mask, rows=2)
            else:
                qmg.stencil_apply_slab(dtype, d, dl, dx, lo, hi, pieces, nrhs, nl, 2 * row, mask, rows=0)
            got = dl.to_host()
            for k in range(nrhs):
                assert np.array_equal(got[k * nl:(k + 1) * nl], rows(want[k * n:(k + 1) * n], L, row, y0, Ll)), (storage, R, r, hex(pieces), k)


@pytest.mark.parametrize("forced_rccl", [False, True])
def test_library_exchange_on_one_rank_is_the_periodic_wrap(monkeypatch, forced_rccl):
    """qmg_halo_exchange with one rank: device copies, or -- with a real one-rank communicator -- ncclSend / ncclRecv to self;
    either way the slab IS the lattice and apply_slab must equal the periodic apply.  With the communicator up, the
    distributed reductions run their all-reduce (of one rank: the value itself)."""
    import ctypes as C
    Lx, Ly, nrhs = 48, 16, 2
    lib = qmg.lib()
    if forced_rccl:
        monkeypatch.setenv("QMG_COMM_FORCE_RCCL", "1")
        uid = (C.c_char * 128)()
        assert lib.qmg_comm_get_unique_id(uid) == 0 and lib.qmg_comm_init(uid, 1, 0) == 0
        qmg.comm_set_distributed_reductions(True)
    try:
        rng = np.random.default_rng(2)
        g = D(np.exp(1j * rng.uniform(-np.pi, np.pi, size=2 * Lx * Ly)))
        cl, hp = qmg.DeviceArray(4 * Lx * Ly), qmg.DeviceArray(16 * Lx * Ly)
        qmg.wilson_fill(cl, hp, g, Lx, Ly, 1.0)
        d = qmg.make_desc(Lx, Ly, 2, cl, hp, 0.1)
        n, row = 2 * Lx * Ly, Lx
        x = cs.gaussian_cvec(n * nrhs, 8)
        dx, want, got = D(x), qmg.DeviceArray(n * nrhs), qmg.DeviceArray(n * nrhs)
        qmg.set_tuning("stencil_site", 7)
        qmg.stencil_apply(d, want, dx, qmg.P_ALL | qmg.P_ZERO, nrhs, n)
        qmg.set_tuning("stencil_site", 3)
        lo, hi = qmg.DeviceArray(2 * row * nrhs), qmg.DeviceArray(2 * row * nrhs)
        qmg.halo_exchange(qmg.C64, dx, Lx, Ly, 2, lo, hi, nrhs, n, 2 * row)
        xs = x.reshape(nrhs, 2, Ly, row)
        assert np.array_equal(lo.to_host(), xs[:, :, Ly - 1].reshape(-1)) and np.array_equal(hi.to_host(), xs[:, :, 0].reshape(-1))
        qmg.stencil_apply_slab(qmg.C64, d, got, dx, lo, hi, qmg.P_ALL | qmg.P_ZERO, nrhs, n, 2 * row, 0b11)
        assert np.array_equal(got.to_host(), want.to_host())
        # reductions: one rank's sum over ranks is the local value, through the all-reduce when the communicator is real
        assert qmg.norm2sq(dx, n) == pytest.approx(np.vdot(x[:n], x[:n]).real, rel=1e-13)
        raw = qmg.batch_reduce(qmg.BRED_NORM2, dx, None, n, nrhs, n, 0b11)
        assert raw[1].real == pytest.approx(np.vdot(x[n:], x[n:]).real, rel=1e-13)
    finally:
        if forced_rccl:
            qmg.comm_set_distributed_reductions(False)
            assert lib.qmg_comm_finalize() == 0


def test_slab_entry_points_refuse_what_they_do_not_serve():
    import ctypes as C
    L = 16
    cl, hp = qmg.DeviceArray(16 * L * L), qmg.DeviceArray(64 * L * L)
    x, y = qmg.DeviceArray(4 * L * L), qmg.DeviceArray(4 * L * L)
    halo = qmg.DeviceArray(4 * L)
    d4 = qmg.make_desc(L, L, 4, cl, hp)
    lib = qmg.lib()

    def call(storage, desc, rows=0):
        return lib.qmg_stencil_apply_slab(storage, C.byref(desc), C.c_void_p(y.ptr), C.c_void_p(x.ptr), C.c_void_p(halo.ptr), C.c_void_p(halo.ptr), C.c_uint(0xFFF), 1,
                                          C.c_size_t(0), C.c_size_t(0), C.c_uint(1), rows, None)
    assert call(qmg.C64, d4) == 0                                # any nc: kernel B with halos
    assert call(qmg.C64, d4, rows=1) == 3                        # ... but only all rows in one launch
    assert call(qmg.C32 | qmg.SLAB_H16, d4) == 3                 # 16-bit matrices: nc = 2, or the Galerkin levels (nc > 4, a multiple of 4)
    assert call(qmg.C64 | qmg.SLAB_M32, d4) == 3 and call(qmg.C64 | qmg.SLAB_M16, d4) == 3
    assert call(qmg.C64 | qmg.SLAB_M32 | qmg.SLAB_M16, d4) == 1  # one width
    d2 = qmg.make_desc(L, L, 2, cl, hp)
    assert call(qmg.C64 | qmg.SLAB_H16, d2) == 1                 # 16-bit matrices come with fp32 vectors
    assert call(qmg.C64 | qmg.SLAB_M32, d2) == 1                 # narrow copies with wider vectors are not an nc = 2 format
    assert call(7, d2) == 1
    assert lib.qmg_stencil_apply_slab(qmg.C64, C.byref(d2), C.c_void_p(x.ptr), C.c_void_p(x.ptr), C.c_void_p(halo.ptr), C.c_void_p(halo.ptr), C.c_uint(0xFFF), 1,
                                      C.c_size_t(0), C.c_size_t(0), C.c_uint(1), 0, None) == 1          # in place
    assert lib.qmg_halo_exchange(qmg.C64, C.c_void_p(x.ptr), L, 15, 2, C.c_void_p(halo.ptr), C.c_void_p(halo.ptr), 1, C.c_size_t(0), C.c_size_t(0), None) == 1


@pytest.mark.parametrize("forced_rccl", [False, True])
def test_slab_solve_driver_on_one_rank(forced_rccl):
    """drivers/slab_wilson_solve.cpp: gauge generation, slab operator, the single-domain cross-check, the overlapped apply and a
    BiCGStab-6 solve with the library's distributed reductions -- on one rank, without and with a real RCCL communicator (the
    send / recv / all-reduce calls are then made for real, to self).  Both runs must tell the same story digit for digit."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drivers = os.path.join(root, "quantum-mg_amd", "drivers")
    subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    if forced_rccl:
        env["QMG_COMM_FORCE_RCCL"] = "1"
    out = subprocess.run([os.path.join(drivers, "slab_wilson_solve"), "128", "0.05", "6.0", "100", "7"], cwd=drivers, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "rel diff 0 (ok)" in out.stdout or re.search(r"rel diff [\d.]+e-1[4-9] \(ok\)", out.stdout)
    m = re.search(r"BiCGStab-6 converged in (\d+) iterations, .* true relative residual ([\d.e+-]+), \|b\| ([\d.e+-]+), \|x\|\^2 ([\d.e+-]+), world 1", out.stdout)
    assert m, out.stdout[-1500:]
    assert float(m.group(2)) < 1e-9
    # the same numbers with and without the communicator (kept across the two parametrisations)
    key = (m.group(1), m.group(3), m.group(4))
    seen = test_slab_solve_driver_on_one_rank.__dict__.setdefault("seen", key)
    assert seen == key
    # a world the rows do not divide into is refused by every rank together
    bad = subprocess.run([os.path.join(drivers, "slab_wilson_solve"), "6", "0.05", "6.0", "1"], cwd=drivers, env=dict(env, WORLD_SIZE="1"), capture_output=True, text=True, timeout=60)
    assert bad.returncode == 0 or "do not split" in bad.stdout      # 6 rows on one rank are fine; the refusal needs world > 1 (not reachable on one GPU)


@pytest.mark.parametrize("L,R", [(64, 2), (64, 4), (128, 8), (96, 3), (96, 6)])
def test_slab_solve_with_ranks_emulated_by_threads(L, R):
    """More than one rank on a one-GPU box: R host threads attach as ranks (qmg_comm_emulate_*, csrc/qmg_comm.hip), the transport
    becomes device copies + host sums behind thread barriers, and everything above it is the code the RCCL path runs -- peer
    selection and halo layout of qmg_halo_exchange, the library's distributed reductions, the slab operator, BiCGStab-6 in lock
    step.  Every rank's slab apply (with REAL neighbours) must equal its rows of the single-domain apply bit for bit, and the
    solve must agree with the one-rank solve: same |b|, the solution's norm to 1e-9, iteration counts within 2 (the
    reductions sum the slabs' partial results, so the rounding -- not the arithmetic -- differs)."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drivers = os.path.join(root, "quantum-mg_amd", "drivers")
    subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    exe = os.path.join(drivers, "slab_wilson_solve")
    args = [exe, str(L), "0.05", "6.0", "100", "7"]
    pat = r"BiCGStab-6 converged in (\d+) iterations, .* true relative residual ([\d.e+-]+), \|b\| ([\d.e+-]+), \|x\|\^2 ([\d.e+-]+), world (\d+)"
    one = subprocess.run(args, cwd=drivers, env=dict(os.environ, RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=300)
    many = subprocess.run(args, cwd=drivers, env=dict(os.environ, QMG_COMM_EMULATE=str(R)), capture_output=True, text=True, timeout=600)
    assert one.returncode == 0 and many.returncode == 0, many.stdout[-2500:] + many.stderr[-1500:]
    checks = re.findall(r"rank (\d+) rows \[(\d+), (\d+)\): slab apply vs single-domain apply, rel diff ([\d.e+-]+) \((ok|MISMATCH)\)", many.stdout)
    assert sorted(int(c[0]) for c in checks) == list(range(R)) and all(c[4] == "ok" and float(c[3]) == 0.0 for c in checks), checks
    assert sorted((int(c[1]), int(c[2])) for c in checks) == [(r * L // R, (r + 1) * L // R) for r in range(R)]
    m1, mR = re.search(pat, one.stdout), re.search(pat, many.stdout)
    assert m1 and mR, many.stdout[-1500:]
    assert int(mR.group(5)) == R and float(mR.group(2)) < 1e-9
    assert m1.group(3) == mR.group(3) or abs(float(m1.group(3)) - float(mR.group(3))) < 1e-12 * float(m1.group(3))      # |b|: a sum in another order
    assert abs(float(m1.group(4)) - float(mR.group(4))) < 1e-9 * float(m1.group(4))
    assert abs(int(m1.group(1)) - int(mR.group(1))) <= 2


@pytest.mark.parametrize("nc,nrhs,mask", [(8, 1, 1), (8, 3, 0b101), (12, 6, 0b111111), (1, 2, 0b11), (4, 1, 1)])
def test_generic_nc_slab_apply_through_kernel_B(nc, nrhs, mask):
    """The Galerkin coarse operators (any nc) on slabs: kernel B with the right-hand side's rows -1 / Ly from the halo buffers
    reproduces the rows of its own single-domain apply bit for bit (qmg_stencil_apply_slab, nc != 2)."""
    Lx, Ly, R = 16, 16, 4
    vol = Lx * Ly
    n = vol * nc
    clover, hopping = cs.gaussian_cvec(vol * nc * nc, 1), cs.gaussian_cvec(4 * vol * nc * nc, 2)
    x, l0 = cs.gaussian_cvec(n * nrhs, 3), cs.gaussian_cvec(n * nrhs, 4)
    shifts = (0.1 + 0.05j, 0.02, 0.03 if nc % 2 == 0 else 0.0)
    d = qmg.make_desc(Lx, Ly, nc, D(clover), D(hopping), *shifts)
    qmg.set_tuning("stencil_mfma", 0)          # the single-domain twin through kernel B as well (kernel C sums in another order)
    qmg.set_tuning("stencil_pair", 0)
    try:
        for pieces in (qmg.P_ALL | qmg.P_ZERO, qmg.P_ALL, qmg.P_OE | qmg.P_ZERO_O):
            want = D(l0)
            qmg.stencil_apply_t(qmg.C64, d, want, D(x), pieces, nrhs, n, mask)
            want = want.to_host()
            Ll, row = Ly // R, (Lx // 2) * nc
            nl = Lx * Ll * nc
            xs = x.reshape(nrhs, 2, Ly, row)
            for r in range(R):
                y0 = r * Ll
                dl = qmg.make_desc(Lx, Ll, nc, D(rows(clover, Ly, (Lx // 2) * nc * nc, y0, Ll)), D(rows(hopping, Ly, (Lx // 2) * nc * nc, y0, Ll)), *shifts)
                dx = D(np.concatenate([rows(x[k * n:(k + 1) * n], Ly, row, y0, Ll) for k in range(nrhs)]))
                out = D(np.concatenate([rows(l0[k * n:(k + 1) * n], Ly, row, y0, Ll) for k in range(nrhs)]))
                lo, hi = D(xs[:, :, (y0 - 1) % Ly].reshape(-1)), D(xs[:, :, (y0 + Ll) % Ly].reshape(-1))
                qmg.stencil_apply_slab(qmg.C64, dl, out, dx, lo, hi, pieces, nrhs, nl, 2 * row, mask, rows=0)
                got = out.to_host()
                for k in range(nrhs):
                    if nc in (1, 4):    # single-domain nc = 1, 4 run kernel A (other summation order): 1e-14 instead of bit equality
                        assert cs.rel_l2(got[k * nl:(k + 1) * nl], rows(want[k * n:(k + 1) * n], Ly, row, y0, Ll)) < 1e-14
                    else:
                        assert np.array_equal(got[k * nl:(k + 1) * nl], rows(want[k * n:(k + 1) * n], Ly, row, y0, Ll)), (nc, r, hex(pieces), k)
    finally:
        qmg.set_tuning("stencil_mfma", 1)
        qmg.set_tuning("stencil_pair", 2)


@pytest.mark.parametrize("R", [2, 4])
def test_galerkin_build_and_gaussian_on_slabs(R):
    """Setup on slabs: the block-local Galerkin build with the null vectors' halo rows gives the slab's rows of the single-domain
    coarse operator bit for bit, and qmg_gaussian_slab draws the slab's rows of the single-domain Gaussian vector."""
    L, nvec = 32, 8
    cL = L // 4
    vol = L * L
    g = D(random_gauge(L, 4))
    cl, hp = qmg.DeviceArray(4 * vol), qmg.DeviceArray(16 * vol)
    qmg.wilson_fill(cl, hp, g, L, L, 1.0)
    fsize = 2 * vol
    P = cs.gaussian_cvec(nvec * fsize, 21)
    dP = D(P)
    qmg.block_orthonormalize(dP, nvec, (L, L, 2), cL, cL)
    P = dP.to_host()
    cc, ch = qmg.DeviceArray(cL * cL * nvec * nvec), qmg.DeviceArray(4 * cL * cL * nvec * nvec)
    qmg.coarse_build(cc, ch, qmg.make_desc(L, L, 2, cl, hp, 0.0), dP, (cL, cL, nvec))
    want_c, want_h = cc.to_host(), ch.to_host()
    cl_h, hp_h = cl.to_host(), hp.to_host()
    Ll, cLl, row = L // R, cL // R, L
    Ps = P.reshape(nvec, 2, L, row)
    for r in range(R):
        y0 = r * Ll
        Pl = D(np.concatenate([rows(P[v * fsize:(v + 1) * fsize], L, row, y0, Ll) for v in range(nvec)]))
        lo, hi = D(Ps[:, :, (y0 - 1) % L].reshape(-1)), D(Ps[:, :, (y0 + Ll) % L].reshape(-1))
        dl = qmg.make_desc(L, Ll, 2, D(rows(cl_h, L, (L // 2) * 4, y0, Ll)), D(rows(hp_h, L, (L // 2) * 4, y0, Ll)), 0.0)
        sc, sh = qmg.DeviceArray(cL * cLl * nvec * nvec), qmg.DeviceArray(4 * cL * cLl * nvec * nvec)
        qmg.coarse_build_slab(sc, sh, dl, Pl, (cL, cLl, nvec), lo, hi, 2 * row)
        per = (cL // 2) * nvec * nvec
        assert np.array_equal(sc.to_host(), rows(want_c, cL, per, r * cLl, cLl)), r
        assert np.array_equal(sh.to_host(), rows(want_h, cL, per, r * cLl, cLl)), r
        v = qmg.DeviceArray(2 * L * Ll)
        qmg.gaussian_slab(v, L, L, y0, Ll, 2, 99)
        full = qmg.DeviceArray(2 * vol)
        qmg.gaussian(full, 2 * vol, 99)
        assert np.array_equal(v.to_host(), rows(full.to_host(), L, row, y0, Ll))


@pytest.mark.parametrize("L,levels,nc,R,batched", [(128, 2, 8, 2, False), (128, 2, 8, 4, False), (256, 2, 8, 8, False), (128, 1, 24, 4, False),
                                                   (128, 2, 8, 4, True), (256, 2, 24, 4, True)])
def test_kcycle_on_slabs_follows_the_single_domain_kcycle(L, levels, nc, R, batched):
    """The whole n13 K-cycle with ONE lattice cut into R y-slabs (drivers/n13_wilson_kcycle_slab.cpp, facade slab mode): setup
    (null-vector relaxation, block orthonormalisation, Galerkin build with the prolongator's halo rows) and solve on every level
    decomposed, ranks emulated by host threads.  The decomposed run draws the single-domain run's random vectors (gaussian_lattice),
    so it must FOLLOW the single-domain n13 driver (same sequential fp64 null-vector relaxation): same outer iteration count, the
    same right-hand side, the same solution norm to 1e-10, true residual below tolerance -- only the rounding of the reductions
    (per-slab partial sums) differs.  `batched`: the default setup instead (null vectors relaxed 8 at a time on complex<float> copies: ONE
    halo exchange per batch apply, kernel W / kernel B with halos in fp32) -- there the single-domain driver uses the MFMA kernel where
    the slabs use kernel B, so the two hierarchies differ by fp32 rounding: iteration counts within 1, solution norm to 1e-8."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drivers = os.path.join(root, "quantum-mg_amd", "drivers")
    subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    gauge = os.path.join(root, "tests", "golden", "l64t64b60_heatbath.dat")
    args = [str(L), "-0.05", "6.0", str(levels), str(nc), gauge, "64"]
    env = dict(os.environ, QMG_QUIET="1")
    if not batched:
        env["QMG_NULL_BATCH"] = "1"
    plain = subprocess.run([os.path.join(drivers, "n13_wilson_kcycle")] + args, cwd=drivers, env=env, capture_output=True, text=True, timeout=600)
    one = subprocess.run([os.path.join(drivers, "n13_wilson_kcycle_slab")] + args, cwd=drivers, env=dict(env, RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=600)
    many = subprocess.run([os.path.join(drivers, "n13_wilson_kcycle_slab")] + args, cwd=drivers, env=dict(env, QMG_COMM_EMULATE=str(R)), capture_output=True, text=True, timeout=900)
    for o in (plain, one, many):
        assert o.returncode == 0, o.stdout[-2500:] + o.stderr[-1500:]
    it = lambda o: int(re.search(r"Multigrid converged in (\d+) iterations", o.stdout).group(1))
    chk = lambda o: float(re.search(r"Check tolerance ([\d.e+-]+)", o.stdout).group(1))
    slab = lambda o: [float(v) for v in re.search(r"\[QMG-SLAB\]: world \d+ ; \|b\| ([\d.e+-]+) ; \|x\|\^2 ([\d.e+-]+)", o.stdout).groups()]
    # one rank in slab mode IS the single-domain run (its exchanges are device copies): identical to the plain driver
    if batched:
        assert abs(it(one) - it(plain)) <= 1 and abs(it(many) - it(one)) <= 1, (it(plain), it(one), it(many))
    else:
        assert it(one) == it(plain) and chk(one) == chk(plain)
        assert it(many) == it(one), (it(many), it(one))
    assert chk(many) < 1e-9
    b1, x1 = slab(one)
    bR, xR = slab(many)
    assert abs(bR - b1) < 1e-13 * b1 and abs(xR - x1) < (1e-8 if batched else 1e-10) * x1, (b1, bR, x1, xR)
    assert "world %d" % R in many.stdout


@pytest.mark.parametrize("R", [2, 4])
def test_cgne_smoothers_on_slabs(R):
    """LevelSolveMG::pre_cgne / post_cgne (stateful_multigrid.h:847-857, 1032-1042) with ONE lattice cut into y-slabs: the dagger stencils of every
    level are built on the slabs (qmg_build_dagger_slab: the transposed hops across a slab boundary come from the neighbouring rank's rows) and
    the smoothers run MR on M M^dagger followed by M^dagger through the slab applies.  One rank in slab mode = the plain driver digit for
    digit; R thread-emulated ranks: the same outer iterations, the same solution norm to 1e-10."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drivers = os.path.join(root, "quantum-mg_amd", "drivers")
    subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    gauge = os.path.join(root, "tests", "golden", "l64t64b60_heatbath.dat")
    args = ["128", "-0.05", "6.0", "2", "8", gauge, "64"]
    env = dict(os.environ, QMG_QUIET="1", QMG_NULL_BATCH="1", QMG_SMOOTHER="cgne")
    plain = subprocess.run([os.path.join(drivers, "n13_wilson_kcycle")] + args, cwd=drivers, env=env, capture_output=True, text=True, timeout=600)
    one = subprocess.run([os.path.join(drivers, "n13_wilson_kcycle_slab")] + args, cwd=drivers, env=dict(env, RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=600)
    many = subprocess.run([os.path.join(drivers, "n13_wilson_kcycle_slab")] + args, cwd=drivers, env=dict(env, QMG_COMM_EMULATE=str(R)), capture_output=True, text=True, timeout=900)
    for o in (plain, one, many):
        assert o.returncode == 0 and "CGNE smoothers" in o.stdout, o.stdout[-2500:] + o.stderr[-1500:]
        assert "[QMG-ERROR]" not in o.stdout and "[QMG-WARNING]" not in o.stdout, o.stdout[-2500:]
    it = lambda o: int(re.search(r"Multigrid converged in (\d+) iterations", o.stdout).group(1))
    chk = lambda o: float(re.search(r"Check tolerance ([\d.e+-]+)", o.stdout).group(1))
    slab = lambda o: [float(v) for v in re.search(r"\[QMG-SLAB\]: world \d+ ; \|b\| ([\d.e+-]+) ; \|x\|\^2 ([\d.e+-]+)", o.stdout).groups()]
    assert it(one) == it(plain) and chk(one) == chk(plain)
    assert it(many) == it(one) and chk(many) < 1e-9, (it(many), it(one), chk(many))
    (b1, x1), (bR, xR) = slab(one), slab(many)
    assert abs(bR - b1) < 1e-13 * b1 and abs(xR - x1) < 1e-10 * x1, (b1, bR, x1, xR)


def test_two_processes_on_one_device_rendezvous_and_are_refused_by_rccl_without_hanging():
    """The launcher path with world = 2 on a one-GPU box: both processes complete the TCP rendezvous of the RCCL id (qmg_comm_init_env),
    then RCCL refuses the second rank on the same device -- and BOTH processes must come back with an error exit in seconds, not
    hang in a half-built communicator (this is the limit that the thread-emulated ranks work around)."""
    import os
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drivers = os.path.join(root, "quantum-mg_amd", "drivers")
    subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="2", LOCAL_RANK="0", QMG_COMM_TIMEOUT_S="30")
    args = [os.path.join(drivers, "slab_wilson_solve"), "64", "0.05", "6.0", "20", "7"]
    t0 = time.time()
    procs = [subprocess.Popen(args, cwd=drivers, env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert time.time() - t0 < 100
    assert all(p.returncode not in (0, None) for p in procs), [p.returncode for p in procs]
    assert all("qmg_comm_init_env failed" in o for o in outs), outs


@pytest.mark.parametrize("R,f32", [(4, False), (2, True)])
def test_batched_kcycle_solve_on_slabs(R, f32):
    """The lock-step batch engine on slabs (n13_wilson_kcycle_slab ... nrhs=4 [f32]): one halo exchange per batch apply, per-system
    reductions summed over the ranks, the K-cycle optionally in complex<float>.  One rank in slab mode reproduces the plain batched driver
    digit for digit; R thread-emulated ranks converge every system in the same number of iterations (+-1) to the same tolerance."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drivers = os.path.join(root, "quantum-mg_amd", "drivers")
    subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    gauge = os.path.join(root, "tests", "golden", "l64t64b60_heatbath.dat")
    base = ["128", "-0.05", "6.0", "2", "8", gauge, "64"]
    tail = ["f32"] if f32 else []
    env = dict(os.environ, QMG_QUIET="1")
    plain = subprocess.run([os.path.join(drivers, "n13_wilson_kcycle_mrhs")] + base + ["4"] + tail, cwd=drivers, env=env, capture_output=True, text=True, timeout=600)
    one = subprocess.run([os.path.join(drivers, "n13_wilson_kcycle_slab")] + base + ["nrhs=4"] + tail, cwd=drivers, env=dict(env, RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=600)
    many = subprocess.run([os.path.join(drivers, "n13_wilson_kcycle_slab")] + base + ["nrhs=4"] + tail, cwd=drivers, env=dict(env, QMG_COMM_EMULATE=str(R)), capture_output=True, text=True, timeout=900)
    pat = r"\[QMG-MRHS\]: rhs (\d+) converged in (\d+) iterations ; alleged tolerance ([\d.e+-]+) ; check tolerance ([\d.e+-]+)"
    rows_ = [re.findall(pat, o.stdout) for o in (plain, one, many)]
    for o, r in zip((plain, one, many), rows_):
        assert o.returncode == 0 and len(r) == 4, o.stdout[-2000:] + o.stderr[-1000:]
    assert rows_[0] == rows_[1]                                                    # one slab = the whole lattice: the same run
    for a, b in zip(rows_[1], rows_[2]):
        assert abs(int(a[1]) - int(b[1])) <= 1 and float(b[3]) < 1e-9, (a, b)


@pytest.mark.parametrize("R", [2, 4])
def test_schur_kcycle_on_slabs(R):
    """The red-black form on slabs (n19: every level solved as the even-odd Schur complement of its right-block-Jacobi operator, coarse
    operators Galerkin-coarsened from the rbjacobi stencil): the rbjacobi builds exchange the halo rows of cinv (qmg_rb_hopping_slab),
    the applies only the rows of the parity they read (qmg_halo_exchange_parity: the Schur systems' vectors are half-length), the in-place
    D_eo of the Schur apply runs through the slab kernels.  One rank in slab mode = the plain n19 driver digit for digit; R thread
    ranks: the same 11 iterations, the same solution norm to 1e-11, and the batched Schur solve converges every system alike."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drivers = os.path.join(root, "quantum-mg_amd", "drivers")
    subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    gauge = os.path.join(root, "tests", "golden", "l128t128b60_heatbath.dat")
    args = [os.path.join(drivers, "n19_wilson_kcycle_precond"), "128", "2", gauge, "128", "nrhs=2"]
    env = dict(os.environ, QMG_QUIET="1")
    plain = subprocess.run(args, cwd=drivers, env=env, capture_output=True, text=True, timeout=600)
    one = subprocess.run(args, cwd=drivers, env=dict(env, QMG_SLAB="1", RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=600)
    many = subprocess.run(args, cwd=drivers, env=dict(env, QMG_COMM_EMULATE=str(R)), capture_output=True, text=True, timeout=900)
    for o in (plain, one, many):
        assert o.returncode == 0, o.stdout[-2500:] + o.stderr[-1500:]
    it = lambda o: int(re.search(r"Multigrid converged in (\d+) iterations", o.stdout).group(1))
    chk = lambda o: re.search(r"Check tolerance ([\d.e+-]+)", o.stdout).group(1)
    xn = lambda o: float(re.search(r"\|x\|\^2 ([\d.e+-]+)", o.stdout).group(1))
    assert it(one) == it(plain) and chk(one) == chk(plain)
    assert it(many) == it(one) and float(chk(many)) < 1e-7
    assert abs(xn(many) - xn(one)) < 1e-11 * xn(one)
    pat = r"\[QMG-MRHS\]: rhs (\d+) converged in (\d+) iterations"
    assert re.findall(pat, many.stdout) == re.findall(pat, plain.stdout) and len(re.findall(pat, many.stdout)) == 2


@pytest.mark.parametrize("opts,R", [([], 4), (["schur"], 2), (["schur", "nrhs=2", "f32"], 4)])
def test_adaptive_n22_on_slabs(opts, R):
    """BASELINE configs[4]'s driver on slabs (n22: adaptive setup passes through the batched K-cycle, optional red-black form, optional
    fp32 K-cycle for the batched solves): one rank in slab mode = the plain driver digit for digit; R thread ranks: the same outer
    iteration count, the same solution norm to 1e-10."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drivers = os.path.join(root, "quantum-mg_amd", "drivers")
    subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    gauge = os.path.join(root, "tests", "golden", "l64t64b60_heatbath.dat")
    args = [os.path.join(drivers, "n22_wilson_kcycle_adaptive"), "256", "-0.07", "6.0", "2", "1", gauge, "64"] + opts
    env = dict(os.environ, QMG_QUIET="1")
    plain = subprocess.run(args, cwd=drivers, env=env, capture_output=True, text=True, timeout=600)
    one = subprocess.run(args, cwd=drivers, env=dict(env, QMG_SLAB="1", RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=600)
    many = subprocess.run(args, cwd=drivers, env=dict(env, QMG_COMM_EMULATE=str(R)), capture_output=True, text=True, timeout=900)
    for o in (plain, one, many):
        assert o.returncode == 0, o.stdout[-2500:] + o.stderr[-1500:]
    it = lambda o: int(re.search(r"Multigrid converged in (\d+) iterations", o.stdout).group(1))
    chk = lambda o: re.search(r"Check tolerance ([\d.e+-]+)", o.stdout).group(1)
    xn = lambda o: float(re.search(r"\|x\|\^2 ([\d.e+-]+)", o.stdout).group(1))
    assert it(one) == it(plain) and chk(one) == chk(plain)
    assert it(many) == it(one) and float(chk(many)) < 1e-9
    assert abs(xn(many) - xn(one)) < 1e-10 * xn(one)
    if "nrhs=2" in opts:
        pat = r"\[QMG-MRHS\]: rhs (\d+) converged in (\d+) iterations"
        a, b = re.findall(pat, one.stdout), re.findall(pat, many.stdout)
        assert len(a) == 2 and len(b) == 2 and all(abs(int(u[1]) - int(v[1])) <= 1 for u, v in zip(a, b)), (a, b)


@pytest.mark.parametrize("nc,nrhs,mask,f32", [(8, 6, 0b111111, False), (12, 8, 0xFF, False), (24, 5, 0b11011, False), (24, 16, 0xFFFF, False), (8, 8, 0xFF, True), (24, 7, 0x7F, True)])
def test_coarse_batches_on_slabs_through_the_mfma_kernel(nc, nrhs, mask, f32):
    """Kernel C (the multi-rhs coarse apply on the matrix cores) with the halo step: a slab's batch of right-hand sides takes rows -1 / Ly
    from the halo buffers; bit for bit the rows of the single-domain kernel C, in fp64 (f64 MFMA) and with complex<float> matrices and
    vectors (f32 MFMA)."""
    Lx, Ly, R = 16, 16, 2
    vol = Lx * Ly
    n = vol * nc
    vt = np.complex64 if f32 else np.complex128
    dt = qmg.C32 if f32 else qmg.C64
    clover, hopping = cs.gaussian_cvec(vol * nc * nc, 1).astype(vt), cs.gaussian_cvec(4 * vol * nc * nc, 2).astype(vt)
    x, l0 = cs.gaussian_cvec(n * nrhs, 3).astype(vt), cs.gaussian_cvec(n * nrhs, 4).astype(vt)
    shifts = (0.1 + 0.05j, 0.02, 0.03)
    d = qmg.make_desc(Lx, Ly, nc, D(clover), D(hopping), *shifts)
    pieces = qmg.P_ALL | qmg.P_ZERO
    want = D(l0)
    qmg.stencil_apply_t(dt, d, want, D(x), pieces, nrhs, n, mask)
    want = want.to_host()
    Ll, row = Ly // R, (Lx // 2) * nc
    nl = Lx * Ll * nc
    xs = x.reshape(nrhs, 2, Ly, row)
    for r in range(R):
        y0 = r * Ll
        dl = qmg.make_desc(Lx, Ll, nc, D(rows(clover, Ly, (Lx // 2) * nc * nc, y0, Ll)), D(rows(hopping, Ly, (Lx // 2) * nc * nc, y0, Ll)), *shifts)
        dx = D(np.concatenate([rows(x[k * n:(k + 1) * n], Ly, row, y0, Ll) for k in range(nrhs)]))
        out = D(np.concatenate([rows(l0[k * n:(k + 1) * n], Ly, row, y0, Ll) for k in range(nrhs)]))
        lo, hi = D(np.ascontiguousarray(xs[:, :, (y0 - 1) % Ly]).reshape(-1)), D(np.ascontiguousarray(xs[:, :, (y0 + Ll) % Ly]).reshape(-1))
        qmg.stencil_apply_slab(dt, dl, out, dx, lo, hi, pieces, nrhs, nl, 2 * row, mask, rows=0)
        got = out.to_host()
        for k in range(nrhs):
            assert np.array_equal(got[k * nl:(k + 1) * nl], rows(want[k * n:(k + 1) * n], Ly, row, y0, Ll)), (nc, r, k)


@pytest.mark.parametrize("R,nc", [(2, 2), (4, 3), (2, 8)])
def test_dagger_build_on_slabs_matches_the_rows_of_the_single_domain_build(R, nc):
    """build_dagger_stencil (stencil_2d.h:1080-1139) on y-slabs: dagger[+y] of a slab's last row is the conjugate transpose of the NEXT rank's
    first-row hopping[-y], dagger[-y] of its first row that of the PREVIOUS rank's last-row hopping[+y] (qmg_build_dagger_slab with the two halo
    rows qmg_halo_exchange delivers).  Bit for bit the rows of the single-domain build."""
    Lx = Ly = 16
    hr, nc2, vol = Lx // 2, nc * nc, Lx * Ly
    clover = cs.gaussian_cvec(vol * nc2, 31)
    hopping = cs.gaussian_cvec(4 * vol * nc2, 32)
    dcl, dho = qmg.DeviceArray(vol * nc2), qmg.DeviceArray(4 * vol * nc2)
    qmg.build_dagger(dcl, dho, D(clover), D(hopping), Lx, Ly, nc)
    want_cl = dcl.to_host().reshape(2, Ly, hr, nc2)
    want_ho = dho.to_host().reshape(4, 2, Ly, hr, nc2)
    cl_g = clover.reshape(2, Ly, hr, nc2)
    ho_g = hopping.reshape(4, 2, Ly, hr, nc2)
    Ll = Ly // R
    for r in range(R):
        y0 = r * Ll
        cl_l = np.ascontiguousarray(cl_g[:, y0:y0 + Ll]).reshape(-1)
        ho_l = np.ascontiguousarray(ho_g[:, :, y0:y0 + Ll]).reshape(-1)
        ym_hi = np.ascontiguousarray(ho_g[3, :, (y0 + Ll) % Ly]).reshape(-1)     # [parity][hr][nc^2]: the next rank's first row of the -y field
        yp_lo = np.ascontiguousarray(ho_g[1, :, (y0 - 1) % Ly]).reshape(-1)      # the previous rank's last row of the +y field
        ocl, oho = qmg.DeviceArray(Lx * Ll * nc2), qmg.DeviceArray(4 * Lx * Ll * nc2)
        qmg.build_dagger_slab(ocl, oho, D(cl_l), D(ho_l), Lx, Ll, nc, D(ym_hi), D(yp_lo))
        assert np.array_equal(ocl.to_host().reshape(2, Ll, hr, nc2), want_cl[:, y0:y0 + Ll]), (R, r)
        assert np.array_equal(oho.to_host().reshape(4, 2, Ll, hr, nc2), want_ho[:, :, y0:y0 + Ll]), (R, r)


@pytest.mark.parametrize("R", [1, 2, 4])
def test_dagger_stencils_on_slabs_satisfy_the_adjoint_identity(R):
    """Facade slab mode: build_dagger_stencil on every level of an n13 hierarchy (Wilson fine level, Galerkin coarse levels, nc = 8) with the
    boundary rows' +-y hops taken from the neighbouring ranks, then <u, M v> = <M^dag u, v> with the reductions summed over the ranks
    (drivers/n13_wilson_kcycle_slab.cpp ... adjoint).  Wiring the boundary rows to the slab's own opposite edge (what the single-domain
    build would do on a slab) breaks the identity at O(1)."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drivers = os.path.join(root, "quantum-mg_amd", "drivers")
    subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    gauge = os.path.join(root, "tests", "golden", "l64t64b60_heatbath.dat")
    args = ["128", "-0.05", "6.0", "2", "8", gauge, "64", "adjoint"]
    env = dict(os.environ, QMG_QUIET="1")
    env.update({"QMG_COMM_EMULATE": str(R)} if R > 1 else {"RANK": "0", "WORLD_SIZE": "1"})
    out = subprocess.run([os.path.join(drivers, "n13_wilson_kcycle_slab")] + args, cwd=drivers, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    rows = re.findall(r"\[QMG-SLAB\]: level (\d) .* rel diff ([-\d.e+]+) \((ok|MISMATCH)\)", out.stdout)
    assert len(rows) == 3 and all(r[2] == "ok" and float(r[1]) < 1e-12 for r in rows), out.stdout[-2000:]


@pytest.mark.parametrize("R", [2, 4])
def test_nc1_operator_fills_on_slabs_are_the_rows_of_the_single_domain_fill(R):
    """qmg_staggered_fill_slab / qmg_laplace_fill_slab (operators/staggered.h:50-72, gaugedlaplace.h:45-68) from the GLOBAL links: bit for bit the
    slab's rows of the single-domain fill -- the back-y hop of a slab's first row is the neighbouring slab's link, eta_y follows the global x."""
    Lx = Ly = 16
    hr, vol = Lx // 2, Lx * Ly
    g = np.exp(1j * np.random.default_rng(7).uniform(-np.pi, np.pi, size=2 * vol))
    dg = D(g)
    sh, lc, lh = qmg.DeviceArray(4 * vol), qmg.DeviceArray(vol), qmg.DeviceArray(4 * vol)
    qmg.staggered_fill(sh, dg, Lx, Ly)
    qmg.laplace_fill(lc, lh, dg, Lx, Ly)
    want_sh, want_lc, want_lh = sh.to_host().reshape(4, 2, Ly, hr), lc.to_host().reshape(2, Ly, hr), lh.to_host().reshape(4, 2, Ly, hr)
    Ll = Ly // R
    for r in range(R):
        y0 = r * Ll
        a, c, h = qmg.DeviceArray(4 * Lx * Ll), qmg.DeviceArray(Lx * Ll), qmg.DeviceArray(4 * Lx * Ll)
        qmg.staggered_fill_slab(a, dg, Lx, Ly, y0, Ll)
        qmg.laplace_fill_slab(c, h, dg, Lx, Ly, y0, Ll)
        assert np.array_equal(a.to_host().reshape(4, 2, Ll, hr), want_sh[:, :, y0:y0 + Ll])
        assert np.array_equal(c.to_host().reshape(2, Ll, hr), want_lc[:, y0:y0 + Ll])
        assert np.array_equal(h.to_host().reshape(4, 2, Ll, hr), want_lh[:, :, y0:y0 + Ll])


def test_staggered_and_laplace_solves_on_slabs_follow_the_one_slab_run():
    """drivers/slab_nc1_solve.cpp: Staggered2D (full operator BiCGStab-6; even-odd preconditioned CG + reconstruct, tests/n04) and GaugedLaplace2D
    (even-odd preconditioned CG, tests/n03) in the facade's slab mode with 1, 2 and 4 thread-emulated ranks: true residual of the FULL system below
    1e-8 everywhere, the CG iteration counts identical and |x|^2 equal to 1e-10 across the decompositions (the right-hand side is the same
    global vector; only the rounding of the per-slab partial sums differs)."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drivers = os.path.join(root, "quantum-mg_amd", "drivers")
    subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    gauge = os.path.join(root, "tests", "golden", "l64t64b60_heatbath.dat")
    res = {}
    for R in (1, 2, 4):
        env = dict(os.environ)
        env.update({"QMG_COMM_EMULATE": str(R)} if R > 1 else {"RANK": "0", "WORLD_SIZE": "1"})
        out = subprocess.run([os.path.join(drivers, "slab_nc1_solve"), "128", "0.1", gauge, "64"], cwd=drivers, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        rows = re.findall(r"\[QMG-SLAB\]: (.*?) : world (\d+) ; iterations (\d+) ; true residual ([-\d.e+]+) ; \|x\|\^2 ([-\d.e+]+)", out.stdout)
        assert len(rows) == 3 and all(int(w) == R and float(t) < 1e-8 for _, w, _, t, _ in rows)
        res[R] = {name: (int(it), float(xn)) for name, _, it, _, xn in rows}
    for name, (it1, xn1) in res[1].items():
        for R in (2, 4):
            it, xn = res[R][name]
            assert abs(xn - xn1) <= 1e-9 * xn1, (name, R)
            if "preconditioned CG" in name:
                assert it == it1, (name, R)
            else:
                assert abs(it - it1) <= 12, (name, R)      # BiCGStab-6 steps come in sixes; its path follows the rounding


@pytest.mark.parametrize("nc,nrhs,mask,bits,f32", [(8, 1, 0b1, 32, False), (24, 3, 0b101, 32, False), (12, 2, 0b11, 16, False), (24, 1, 0b1, 16, False),
                                                   (8, 6, 0b111111, 32, False), (24, 8, 0xFF, 16, False), (8, 2, 0b11, 16, True), (24, 6, 0b111111, 16, True)])
def test_narrow_stored_matrices_on_slabs(nc, nrhs, mask, bits, f32):
    """A preconditioner level's narrow Galerkin copy on a slab (qmg_stencil_apply_slab with QMG_SLAB_M32 / QMG_SLAB_M16: complex<float> / complex<half> matrices,
    complex<double> or complex<float> vectors; kernel B32 with its halo step for few systems, kernel C for batches): bit for bit the rows of the whole-lattice
    apply on the same narrow arrays (qmg_stencil_apply_mat32 / qmg_stencil_apply_mat16_t)."""
    Lx, Ly, R = 16, 16, 2
    vol = Lx * Ly
    n = vol * nc
    vt = np.complex64 if f32 else np.complex128
    dt = qmg.C32 if f32 else qmg.C64
    clover, hopping = 0.3 * cs.gaussian_cvec(vol * nc * nc, 11), 0.3 * cs.gaussian_cvec(4 * vol * nc * nc, 12)
    x, l0 = cs.gaussian_cvec(n * nrhs, 13).astype(vt), cs.gaussian_cvec(n * nrhs, 14).astype(vt)
    shifts = (0.1 + 0.05j, 0.02, 0.03)
    storage = "c32" if bits == 32 else "c16"
    d = qmg.make_desc(Lx, Ly, nc, to_storage(clover, storage), to_storage(hopping, storage), *shifts)
    pieces = qmg.P_ALL | qmg.P_ZERO
    want = D(l0)
    if bits == 16:
        qmg.stencil_apply_mat16(dt, d, want, D(x), pieces, nrhs, n, mask)
    else:
        qmg.stencil_apply_mat32(d, want, D(x), pieces, nrhs, n, mask)
    want = want.to_host()
    Ll, row = Ly // R, (Lx // 2) * nc
    nl = Lx * Ll * nc
    xs = x.reshape(nrhs, 2, Ly, row)
    flag = qmg.SLAB_M16 if bits == 16 else qmg.SLAB_M32
    for r in range(R):
        y0 = r * Ll
        dl = qmg.make_desc(Lx, Ll, nc, to_storage(rows(clover, Ly, (Lx // 2) * nc * nc, y0, Ll), storage), to_storage(rows(hopping, Ly, (Lx // 2) * nc * nc, y0, Ll), storage), *shifts)
        dx = D(np.concatenate([rows(x[k * n:(k + 1) * n], Ly, row, y0, Ll) for k in range(nrhs)]))
        out = D(np.concatenate([rows(l0[k * n:(k + 1) * n], Ly, row, y0, Ll) for k in range(nrhs)]))
        lo, hi = D(np.ascontiguousarray(xs[:, :, (y0 - 1) % Ly]).reshape(-1)), D(np.ascontiguousarray(xs[:, :, (y0 + Ll) % Ly]).reshape(-1))
        qmg.stencil_apply_slab(dt | flag, dl, out, dx, lo, hi, pieces, nrhs, nl, 2 * row, mask, rows=0)
        got = out.to_host()
        for k in range(nrhs):
            assert np.array_equal(got[k * nl:(k + 1) * nl], rows(want[k * n:(k + 1) * n], Ly, row, y0, Ll)), (nc, r, k)


@pytest.mark.parametrize("R,extra", [(2, {}), (4, {"QMG_COARSE_BITS": "16"}), (2, {"QMG_KCYCLE_SLAB_ENGINE": "single"})])
def test_kcycle_on_slabs_with_the_default_engine_and_storage(R, extra, monkeypatch):
    """What a slab run does when nothing is switched off (this module's other tests pin fp64 coarse storage and the single-vector engine to compare digit by digit):
    the lock-step batch engine as a batch of one system (qmg_kcycle_via_batch; QMG_KCYCLE_SLAB_ENGINE=single: the single-vector code) streaming the complex<float>
    (QMG_COARSE_BITS=16: complex<half>) copies of the Galerkin matrices through the slab kernels.  One slab: the plain driver's outer iteration count; R thread-emulated
    slabs: the same count (+-1), true residual < 1e-9, the same solution norm to 1e-9."""
    import os
    import re
    import subprocess
    for k in ("QMG_COARSE_F32", "QMG_KCYCLE_ENGINE", "QMG_APPLY_EPILOGUE"):
        monkeypatch.delenv(k, raising=False)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drivers = os.path.join(root, "quantum-mg_amd", "drivers")
    subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    gauge = os.path.join(root, "tests", "golden", "l64t64b60_heatbath.dat")
    args = ["256", "-0.07", "6.0", "2", "8", gauge, "64"]
    env = dict(os.environ, QMG_QUIET="1", QMG_NULL_BATCH="1")
    env.update(extra)
    plain = subprocess.run([os.path.join(drivers, "n13_wilson_kcycle")] + args, cwd=drivers, env=env, capture_output=True, text=True, timeout=600)
    one = subprocess.run([os.path.join(drivers, "n13_wilson_kcycle_slab")] + args, cwd=drivers, env=dict(env, RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=600)
    many = subprocess.run([os.path.join(drivers, "n13_wilson_kcycle_slab")] + args, cwd=drivers, env=dict(env, QMG_COMM_EMULATE=str(R)), capture_output=True, text=True, timeout=900)
    want_info = "complex<half>" if extra.get("QMG_COARSE_BITS") == "16" else "complex<float>"
    for o in (plain, one, many):
        assert o.returncode == 0 and want_info in o.stdout, o.stdout[-2500:] + o.stderr[-1500:]
        assert "[QMG-ERROR]" not in o.stdout and "[QMG-WARNING]" not in o.stdout, o.stdout[-2500:]
    it = lambda o: int(re.search(r"Multigrid converged in (\d+) iterations", o.stdout).group(1))
    chk = lambda o: float(re.search(r"Check tolerance ([\d.e+-]+)", o.stdout).group(1))
    xn = lambda o: float(re.search(r"\[QMG-SLAB\]: world \d+ ; \|b\| [\d.e+-]+ ; \|x\|\^2 ([\d.e+-]+)", o.stdout).group(1))
    assert it(one) == it(plain) and chk(one) < 1e-9 and chk(plain) < 1e-9, (it(one), it(plain), chk(one), chk(plain))
    assert abs(it(many) - it(one)) <= 1 and chk(many) < 1e-9, (it(many), it(one), chk(many))
    assert abs(xn(many) - xn(one)) < 1e-9 * xn(one)


@pytest.mark.parametrize("R,hooks", [(2, {}), (4, {"QMG_SOLVE_TYPE": "jacobi", "QMG_SMOOTHER": "cgne"}), (2, {"QMG_COARSEST_TYPE": "rbj_mdm"})])
def test_red_black_kcycle_on_slabs_with_the_default_engine_and_storage(R, hooks, monkeypatch):
    """The n19 counterpart on slabs with nothing switched off: the batch engine (Schur / right-block-Jacobi levels, CGNE smoothers, CG on a normal-equation
    coarsest operator: DESIGN 10.9) with the narrow copies of the Galerkin matrices, right-block-Jacobi hops and cinv streamed through the slab kernels.
    One slab: the plain driver's outer iteration count; R thread-emulated slabs: the same count (+-1), true residual < 1e-7 (tol 1e-8 on the
    preconditioned system), the same solution norm to solver accuracy."""
    import os
    import re
    import subprocess
    for k in ("QMG_COARSE_F32", "QMG_KCYCLE_ENGINE", "QMG_APPLY_EPILOGUE"):
        monkeypatch.delenv(k, raising=False)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drivers = os.path.join(root, "quantum-mg_amd", "drivers")
    subprocess.check_call(["make", "-C", drivers, "-j4"], stdout=subprocess.DEVNULL)
    # (the 64^2 fixture tiled to 128^2, as in test_gpu_kcycle: on the 128^2 fixture the restarted CG on M_rbj^dagger M_rbj stalls the outer solve at 1e-2 in BOTH
    # engines -- identical residual histories -- which is the algorithm's business, not a test of the decomposition)
    gauge = os.path.join(root, "tests", "golden", "l64t64b60_heatbath.dat")
    args = [os.path.join(drivers, "n19_wilson_kcycle_precond"), "128", "2", gauge, "64"]
    env = dict(os.environ, QMG_QUIET="1")
    env.update(hooks)
    plain = subprocess.run(args, cwd=drivers, env=env, capture_output=True, text=True, timeout=120)
    one = subprocess.run(args, cwd=drivers, env=dict(env, QMG_SLAB="1", RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=120)
    many = subprocess.run(args, cwd=drivers, env=dict(env, QMG_COMM_EMULATE=str(R)), capture_output=True, text=True, timeout=180)
    for o in (plain, one, many):
        assert o.returncode == 0, o.stdout[-2500:] + o.stderr[-1500:]
        assert "[QMG-ERROR]" not in o.stdout and "[QMG-WARNING]" not in o.stdout, o.stdout[-2500:]
    it = lambda o: int(re.search(r"Multigrid converged in (\d+) iterations", o.stdout).group(1))
    chk = lambda o: float(re.search(r"Check tolerance ([\d.e+-]+)", o.stdout).group(1))
    xn = lambda o: float(re.search(r"\|x\|\^2 ([\d.e+-]+)", o.stdout).group(1))
    assert it(one) == it(plain) and chk(one) < 1e-7 and chk(plain) < 1e-7, (it(one), it(plain), chk(one), chk(plain))
    assert abs(it(many) - it(one)) <= 1 and chk(many) < 1e-7, (it(many), it(one), chk(many))
    assert abs(xn(many) - xn(one)) < 1e-6 * xn(one)          # (two solves to 1e-8: the solutions agree to solver accuracy)
