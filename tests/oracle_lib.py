"""ctypes binding of the CPU oracle (oracle/libqmg_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_SO = os.path.join(ORACLE_DIR, "libqmg_oracle.so")

# piece mask (oracle/qmg_oracle.h; identical bit meaning in include/qmg_hip.h)
P_CLOVER_E, P_CLOVER_O = 1 << 0, 1 << 1
P_EO_XP1, P_OE_XP1 = 1 << 2, 1 << 6
P_SHIFT_E, P_SHIFT_O = 1 << 10, 1 << 11
P_ZERO_E, P_ZERO_O = 1 << 12, 1 << 13
P_CLOVER, P_EO, P_OE, P_HOPPING = 3, 0xF << 2, 0xF << 6, 0xFF << 2
P_SHIFT, P_ZERO, P_ALL = 3 << 10, 3 << 12, 0xFFF

CSHIFT_FROM_0, CSHIFT_XP1, CSHIFT_YP1, CSHIFT_XM1, CSHIFT_YM1 = 1, 2, 3, 4, 5
EO_FROM_EVEN, EO_FROM_ODD, EO_FROM_EVENODD = 1, 2, 3


class StencilDesc(C.Structure):
    _fields_ = [("Lx", C.c_int), ("Ly", C.c_int), ("nc", C.c_int),
                ("clover", C.c_void_p), ("hopping", C.c_void_p),
                ("shift", C.c_double * 2), ("eo_shift", C.c_double * 2), ("dof_shift", C.c_double * 2)]


def build():
    """(Re)build the oracle with its Makefile if the .so is missing or stale."""
    src = [os.path.join(ORACLE_DIR, f) for f in ("qmg_oracle.cpp", "qmg_oracle_kcycle.cpp", "qmg_oracle.h", "Makefile")]
    if (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libqmg_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.qo_norm2sq.restype = C.c_double
        _lib.qo_diffnorm2sq.restype = C.c_double
        _lib.qo_norminf.restype = C.c_double
        _lib.qo_time_apply.restype = C.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def cvec(n):
    return np.zeros(n, dtype=np.complex128)


def make_desc(Lx, Ly, nc, clover, hopping, shift=0.0, eo_shift=0.0, dof_shift=0.0):
    d = StencilDesc()
    d.Lx, d.Ly, d.nc = Lx, Ly, nc
    d.clover = None if clover is None else clover.ctypes.data
    d.hopping = None if hopping is None else hopping.ctypes.data
    for name, v in (("shift", shift), ("eo_shift", eo_shift), ("dof_shift", dof_shift)):
        v = complex(v)
        getattr(d, name)[0], getattr(d, name)[1] = v.real, v.imag
    d._keep = (clover, hopping)
    return d


def coord_to_index(Lx, Ly, x, y):
    return lib().qo_coord_to_index(Lx, Ly, x, y)


def index_to_coord(Lx, Ly, i):
    x, y = C.c_int(), C.c_int()
    lib().qo_index_to_coord(Lx, Ly, i, C.byref(x), C.byref(y))
    return x.value, y.value


def cshift(rhs, cdir, eo, dof, Lx, Ly, lhs=None):
    if lhs is None:
        lhs = np.zeros_like(rhs)
    rc = lib().qo_cshift(_p(lhs), _p(rhs), cdir, eo, dof, Lx, Ly)
    assert rc == 0, rc
    return lhs


def stencil_apply(desc, rhs, pieces=P_ALL | P_ZERO, lhs=None):
    if lhs is None:
        lhs = cvec(desc.Lx * desc.Ly * desc.nc)
    rc = lib().qo_stencil_apply(C.byref(desc), _p(lhs), _p(rhs), C.c_uint(pieces))
    assert rc == 0, rc
    return lhs


def time_apply(desc, rhs, pieces, reps):
    lhs = cvec(desc.Lx * desc.Ly * desc.nc)
    return lib().qo_time_apply(C.byref(desc), _p(lhs), _p(rhs), C.c_uint(pieces), reps)


def wilson_fill(gauge, Lx, Ly, w=1.0):
    vol = Lx * Ly
    clover, hopping = cvec(4 * vol), cvec(16 * vol)
    assert lib().qo_wilson_fill(_p(clover), _p(hopping), _p(gauge), Lx, Ly, C.c_double(w)) == 0
    return clover, hopping


def staggered_fill(gauge, Lx, Ly):
    hopping = cvec(4 * Lx * Ly)
    assert lib().qo_staggered_fill(_p(hopping), _p(gauge), Lx, Ly) == 0
    return hopping


def laplace_fill(gauge, Lx, Ly):
    clover, hopping = cvec(Lx * Ly), cvec(4 * Lx * Ly)
    assert lib().qo_laplace_fill(_p(clover), _p(hopping), _p(gauge), Lx, Ly) == 0
    return clover, hopping


def free_laplace_fill(Lx, Ly):
    clover, hopping = cvec(Lx * Ly), cvec(4 * Lx * Ly)
    assert lib().qo_free_laplace_fill(_p(clover), _p(hopping), Lx, Ly) == 0
    return clover, hopping


def read_gauge_u1(path, Lx, Ly):
    g = cvec(2 * Lx * Ly)
    rc = lib().qo_read_gauge_u1(_p(g), Lx, Ly, path.encode())
    assert rc == 0, rc
    return g


def phases_to_gauge_u1(phases, Lx, Ly):
    g = cvec(2 * Lx * Ly)
    ph = np.ascontiguousarray(phases, dtype=np.float64)
    assert lib().qo_phases_to_gauge_u1(_p(g), _p(ph), Lx, Ly) == 0
    return g


def unit_gauge_u1(Lx, Ly):
    g = cvec(2 * Lx * Ly)
    lib().qo_unit_gauge_u1(_p(g), Lx, Ly)
    return g


def build_dagger(clover, hopping, Lx, Ly, nc):
    dc = None if clover is None else np.zeros_like(clover)
    dh = None if hopping is None else np.zeros_like(hopping)
    assert lib().qo_build_dagger(_p(dc), _p(dh), _p(clover), _p(hopping), Lx, Ly, nc) == 0
    return dc, dh


def build_rbjacobi(desc):
    cm = desc.Lx * desc.Ly * desc.nc * desc.nc
    cinv, rclover, rhopping = cvec(cm), cvec(cm), cvec(4 * cm)
    rc = lib().qo_build_rbjacobi(_p(cinv), _p(rclover), _p(rhopping), C.byref(desc))
    assert rc == 0, rc
    return cinv, rclover, rhopping


def build_rbj_dagger(cinv, rclover, rhopping, Lx, Ly, nc):
    """stencil_2d.h:1989-2060: conj-transpose of cinv and of the identity clover, dagger of the right-block-Jacobi hopping."""
    dcinv, dcl, dho = np.zeros_like(cinv), np.zeros_like(rclover), np.zeros_like(rhopping)
    assert lib().qo_build_rbj_dagger(_p(dcinv), _p(dcl), _p(dho), _p(cinv), _p(rclover), _p(rhopping), Lx, Ly, nc) == 0
    return dcinv, dcl, dho


def norm2sq(x):
    return lib().qo_norm2sq(_p(x), C.c_long(x.size))


def dot(x, y):
    out = (C.c_double * 2)()
    lib().qo_dot(_p(x), _p(y), C.c_long(x.size), out)
    return complex(out[0], out[1])


def diffnorm2sq(x, y):
    return lib().qo_diffnorm2sq(_p(x), _p(y), C.c_long(x.size))


def norminf(x):
    return lib().qo_norminf(_p(x), C.c_long(x.size))


def norm2sq_cv_timeslice(cv, Lx, Ly, nc):
    s = np.zeros(Ly)
    lib().qo_norm2sq_cv_timeslice(_p(s), _p(cv), Lx, Ly, nc)
    return s


def redot_cv_timeslice(a, b, Lx, Ly, nc):
    s = np.zeros(Ly)
    lib().qo_redot_cv_timeslice(_p(s), _p(a), _p(b), Lx, Ly, nc)
    return s


def gaussian_wall_source(Lx, Ly, nc, timeslice, color, seed, deviation=1.0, mean=0.0):
    """None for an out-of-range timeslice / color (the reference prints an error and returns)."""
    cv = cvec(Lx * Ly * nc)
    rc = lib().qo_gaussian_wall_source(_p(cv), Lx, Ly, nc, timeslice, color, C.c_ulonglong(seed), C.c_double(deviation), C.c_double(mean))
    return cv if rc == 0 else None


def dot_cv_timeslice(a, b, Lx, Ly, nc):
    s = cvec(Ly)
    lib().qo_dot_cv_timeslice(_p(s), _p(a), _p(b), Lx, Ly, nc)
    return s


def transfer_build_map(fLx, fLy, fnc, cLx, cLy):
    per = (fLx // cLx) * (fLy // cLy) * fnc
    m = np.zeros((cLx * cLy, per), dtype=np.int32)
    assert lib().qo_transfer_build_map(_p(m), fLx, fLy, fnc, cLx, cLy) == per
    return m


def prolong(nullvecs, coarse, fdims, cdims, fine=None, nvec=None):
    fLx, fLy, fnc = fdims
    cLx, cLy, cnc = cdims
    if fine is None:
        fine = cvec(fLx * fLy * fnc)
    nvec = cnc if nvec is None else nvec
    assert lib().qo_prolong(_p(nullvecs), nvec, _p(coarse), _p(fine), fLx, fLy, fnc, cLx, cLy, cnc) == 0
    return fine


def restrict(nullvecs, fine, fdims, cdims, coarse=None, nvec=None):
    fLx, fLy, fnc = fdims
    cLx, cLy, cnc = cdims
    if coarse is None:
        coarse = cvec(cLx * cLy * cnc)
    nvec = cnc if nvec is None else nvec
    assert lib().qo_restrict(_p(nullvecs), nvec, _p(fine), _p(coarse), fLx, fLy, fnc, cLx, cLy, cnc) == 0
    return coarse


def block_orthonormalize(nullvecs, fdims, cdims, cholesky=None):
    fLx, fLy, fnc = fdims
    cLx, cLy, cnc = cdims
    assert lib().qo_block_orthonormalize(_p(nullvecs), cnc, fLx, fLy, fnc, cLx, cLy, _p(cholesky)) == 0
    return nullvecs


def block_bi_orthonormalize(pvecs, rvecs, fdims, cdims, block_L=None, block_U=None):
    fLx, fLy, fnc = fdims
    cLx, cLy, cnc = cdims
    assert lib().qo_block_bi_orthonormalize(_p(pvecs), _p(rvecs), cnc, fLx, fLy, fnc, cLx, cLy, _p(block_L), _p(block_U)) == 0
    return pvecs, rvecs


def coarse_build(fdesc, nullvecs, cdims, restrict_vecs=None):
    cLx, cLy, cnc = cdims
    ccm = cLx * cLy * cnc * cnc
    cclover, chopping = cvec(ccm), cvec(4 * ccm)
    rc = lib().qo_coarse_build(_p(cclover), _p(chopping), C.byref(fdesc), _p(nullvecs), _p(restrict_vecs), cLx, cLy, cnc)
    assert rc == 0, rc
    return cclover, chopping


def wilson_kcycle(L, mass, n_refine, coarse_dof, gauge, nullvecs, b, tol=1e-10, max_iter=1000, restart=32, inner_tol=0.2,
                  coarsest_tol=0.2, n_smooth=2):
    """CPU K-cycle solve (oracle/qmg_oracle_kcycle.cpp). nullvecs: list of per-level arrays (coarse_dof x level size)."""
    ptrs = (C.c_void_p * n_refine)(*[nv.ctypes.data for nv in nullvecs])
    x = cvec(L * L * 2)
    true_res = C.c_double()
    ops = (C.c_long * (n_refine + 1))()
    its = (C.c_long * (n_refine + 1))()
    it = lib().qo_wilson_kcycle(L, C.c_double(mass), n_refine, coarse_dof, _p(gauge), ptrs, _p(b), C.c_double(tol), max_iter, restart,
                                C.c_double(inner_tol), C.c_double(coarsest_tol), n_smooth, _p(x), C.byref(true_res), ops, its)
    return it, x, true_res.value, list(ops), list(its)


def wilson_kcycle_history(L, mass, n_refine, coarse_dof, gauge, nullvecs, b, tol=1e-10, max_iter=1000, restart=32, inner_tol=0.2,
                          coarsest_tol=0.2, n_smooth=2):
    """wilson_kcycle plus the outer relative-residual history and the iteration count of every coarsest solve
    (negative = that solve hit its 1000-iteration cap)."""
    ptrs = (C.c_void_p * n_refine)(*[nv.ctypes.data for nv in nullvecs])
    x = cvec(L * L * 2)
    true_res = C.c_double()
    ops = (C.c_long * (n_refine + 1))()
    its = (C.c_long * (n_refine + 1))()
    nh, nch = max_iter + 8, 1 << 16
    hist = np.zeros(nh)
    chist = np.zeros(nch, dtype=np.int64)
    nho, ncho = C.c_int(), C.c_int()
    it = lib().qo_wilson_kcycle_history(L, C.c_double(mass), n_refine, coarse_dof, _p(gauge), ptrs, _p(b), C.c_double(tol), max_iter, restart,
                                        C.c_double(inner_tol), C.c_double(coarsest_tol), n_smooth, _p(x), C.byref(true_res), ops, its,
                                        hist.ctypes.data_as(C.POINTER(C.c_double)), nh, C.byref(nho), chist.ctypes.data_as(C.POINTER(C.c_long)), nch, C.byref(ncho))
    return it, x, true_res.value, list(ops), list(its), hist[:nho.value].copy(), chist[:ncho.value].copy()


KRYLOV_CG, KRYLOV_BICGSTAB_L, KRYLOV_RICHARDSON, KRYLOV_MR, KRYLOV_GCR = range(5)


def krylov_solve(kind, desc, b, max_iter, tol, param_i=0, param_d=0.0, x0=None, dagger_desc=None, normal=False, nhist=0):
    """CPU twins of the facade's Krylov drivers on a stencil operator (oracle/qmg_oracle_kcycle.cpp qo_krylov_solve).
    Returns (converged, iterations, x, resSq, history)."""
    n = desc.Lx * desc.Ly * desc.nc
    x = cvec(n) if x0 is None else np.array(x0, dtype=np.complex128)
    it, rsq = C.c_int(), C.c_double()
    hist = np.zeros(max(nhist, 1))
    rc = lib().qo_krylov_solve(kind, C.byref(desc), C.byref(dagger_desc) if dagger_desc is not None else None, 1 if normal else 0, _p(x), _p(b),
                               max_iter, C.c_double(tol), param_i, C.c_double(param_d), C.byref(it), C.byref(rsq),
                               hist.ctypes.data_as(C.POINTER(C.c_double)), nhist)
    assert rc >= 0
    return bool(rc), it.value, x, rsq.value, hist[:min(nhist, it.value)].copy()
