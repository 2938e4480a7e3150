"""quantum-mg_amd -- MI355X-native multigrid hot path behind the quantum-mg operator API.

The product is `libqmg_hip.so` (hand-written HIP kernels for gfx950 + a C-ABI, include/qmg_hip.h)
and the C++ facade in `include/qmg/` that mirrors the reference's Stencil2D / TransferMG /
StatefulMultigridMG classes.  This Python module is only the ctypes binding used by tests/,
bench.py and __graft_entry__.py to drive the C-ABI; it holds no compute and has NO CPU
fallback: if the shared library is missing, importing `lib()` raises.

(The directory name contains a hyphen, so import it with
 `importlib.import_module("quantum-mg_amd")`.)
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO_PATH = os.path.join(HERE, "libqmg_hip.so")

# ---- enums (include/qmg_hip.h) ----
P_CLOVER_E, P_CLOVER_O = 1 << 0, 1 << 1
P_EO_XP1, P_OE_XP1 = 1 << 2, 1 << 6
P_SHIFT_E, P_SHIFT_O = 1 << 10, 1 << 11
P_ZERO_E, P_ZERO_O = 1 << 12, 1 << 13
P_CLOVER, P_EO, P_OE, P_HOPPING = 3, 0xF << 2, 0xF << 6, 0xFF << 2
P_SHIFT, P_ZERO, P_ALL = 3 << 10, 3 << 12, 0xFFF
CSHIFT_FROM_0, CSHIFT_XP1, CSHIFT_YP1, CSHIFT_XM1, CSHIFT_YM1 = 1, 2, 3, 4, 5
EO_FROM_EVEN, EO_FROM_ODD, EO_FROM_EVENODD = 1, 2, 3

# every symbol include/qmg_hip.h declares (checked by tests/test_abi_symbols.py against the header text)
ABI_SYMBOLS = [
    "qmg_init", "qmg_device_count", "qmg_status_string", "qmg_last_hip_error", "qmg_version",
    "qmg_malloc", "qmg_free", "qmg_shutdown", "qmg_mem_info", "qmg_memcpy_h2d", "qmg_memcpy_d2h", "qmg_memcpy_d2d", "qmg_memset_zero",
    "qmg_stream_create", "qmg_stream_destroy", "qmg_stream_sync",
    "qmg_event_create", "qmg_event_destroy", "qmg_event_record", "qmg_event_elapsed_ms", "qmg_stream_wait_event",
    "qmg_cshift", "qmg_stencil_apply", "qmg_stencil_apply_batch", "qmg_stencil_apply_mat32", "qmg_c64_to_c32", "qmg_wilson_fill", "qmg_staggered_fill", "qmg_laplace_fill",
    "qmg_build_dagger", "qmg_build_rbjacobi", "qmg_cmat_conjtrans",
    "qmg_zero_vector", "qmg_copy_vector", "qmg_cax", "qmg_caxy", "qmg_caxpy", "qmg_cxpy", "qmg_cxpay",
    "qmg_caxpby", "qmg_cxpyz", "qmg_caxpbyz", "qmg_multi_caxpy", "qmg_caxy_pattern", "qmg_gaussian",
    "qmg_norm2sq", "qmg_dot", "qmg_diffnorm2sq", "qmg_norminf", "qmg_multidot",
    "qmg_norm2sq_cv_timeslice", "qmg_dot_cv_timeslice", "qmg_redot_cv_timeslice", "qmg_gaussian_wall_source",
    "qmg_prolong", "qmg_restrict", "qmg_block_orthonormalize", "qmg_block_orthonormalize_n", "qmg_block_bi_orthonormalize", "qmg_coarse_build", "qmg_set_tuning",
    "qmg_batch_blas", "qmg_batch_multi_caxpy", "qmg_batch_reduce", "qmg_batch_multidot", "qmg_prolong_batch", "qmg_restrict_batch",
    "qmg_comm_get_unique_id", "qmg_comm_init", "qmg_comm_init_env", "qmg_comm_rendezvous", "qmg_comm_all_ok", "qmg_comm_world", "qmg_allreduce_sum_f64", "qmg_comm_finalize",
    "qmg_convert", "qmg_stencil_apply_t", "qmg_batch_blas_t", "qmg_batch_multi_caxpy_t", "qmg_batch_gcr_update_t", "qmg_prolong_batch_nv32", "qmg_restrict_batch_nv32", "qmg_batch_reduce_t", "qmg_batch_multidot_t",
    "qmg_prolong_batch_t", "qmg_restrict_batch_t",
    "qmg_convert_to_c16", "qmg_convert_from_c16", "qmg_stencil_apply_h16", "qmg_stencil_apply_mat16_t", "qmg_stencil_apply_norm2",
    "qmg_wilson_apply_direct", "qmg_wilson_hops_direct", "qmg_halo_exchange", "qmg_halo_exchange_parity", "qmg_stencil_apply_slab", "qmg_wilson_fill_slab", "qmg_comm_set_distributed_reductions", "qmg_coarse_build_slab", "qmg_gaussian_slab", "qmg_rb_hopping_slab", "qmg_build_dagger_slab", "qmg_staggered_fill_slab", "qmg_laplace_fill_slab", "qmg_comm_emulate_begin", "qmg_comm_emulate_attach", "qmg_comm_emulate_end",
    "qmg_stencil_apply_epi_t", "qmg_wilson_apply_direct_epi", "qmg_wilson_hops_direct_epi", "qmg_batch_mr_dots_t", "qmg_batch_mr_update_t", "qmg_batch_mr_read_dots",
    "qmg_u1_heatbath_noncompact", "qmg_u1_phase_to_gauge", "qmg_u1_gauge_to_phase", "qmg_u1_plaquette", "qmg_u1_noncompact_action",
]


class StencilDesc(C.Structure):
    _fields_ = [("Lx", C.c_int), ("Ly", C.c_int), ("nc", C.c_int),
                ("clover", C.c_void_p), ("hopping", C.c_void_p),
                ("shift", C.c_double * 2), ("eo_shift", C.c_double * 2), ("dof_shift", C.c_double * 2)]


class QmgError(RuntimeError):
    pass


def build(force=False):
    """Compile libqmg_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", HERE, "-j4", "libqmg_hip.so"], stdout=subprocess.DEVNULL)
    return SO_PATH


_lib = None


def lib():
    """The loaded C-ABI library.  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise QmgError("libqmg_hip.so not built (%s); run __graft_entry__.build() -- there is no CPU fallback" % SO_PATH)
        _lib = C.CDLL(SO_PATH)
        _lib.qmg_status_string.restype = C.c_char_p
        _lib.qmg_last_hip_error.restype = C.c_char_p
        _lib.qmg_version.restype = C.c_char_p
    return _lib


def check(status, what=""):
    if status != 0:
        L = lib()
        raise QmgError("%s failed: %s [%s]" % (what or "qmg call", L.qmg_status_string(status).decode(), L.qmg_last_hip_error().decode()))


def init(device=0):
    check(lib().qmg_init(device), "qmg_init")


def device_count():
    n = C.c_int(0)
    lib().qmg_device_count(C.byref(n))
    return n.value


def sync(stream=None):
    check(lib().qmg_stream_sync(C.c_void_p(stream)), "qmg_stream_sync")


def stream_create():
    """A non-default HIP stream as the raw handle every `stream=` argument takes."""
    st = C.c_void_p()
    check(lib().qmg_stream_create(C.byref(st)), "qmg_stream_create")
    return st.value


def stream_destroy(stream):
    check(lib().qmg_stream_destroy(C.c_void_p(stream)), "qmg_stream_destroy")


class DeviceArray:
    """A complex128 (or raw bytes) array in HBM owned through qmg_malloc / qmg_free."""

    def __init__(self, n, dtype=np.complex128):
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        self.nbytes = self.n * self.dtype.itemsize
        p = C.c_void_p()
        check(lib().qmg_malloc(C.byref(p), C.c_size_t(self.nbytes)), "qmg_malloc")
        self.ptr = p.value or 0

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.size, a.dtype)
        check(lib().qmg_memcpy_h2d(C.c_void_p(d.ptr), a.ctypes.data_as(C.c_void_p), C.c_size_t(d.nbytes), None), "h2d")
        return d

    @classmethod
    def zeros(cls, n, dtype=np.complex128):
        d = cls(n, dtype)
        check(lib().qmg_memset_zero(C.c_void_p(d.ptr), C.c_size_t(d.nbytes), None), "memset")
        return d

    def to_host(self):
        out = np.empty(self.n, dtype=self.dtype)
        sync()
        check(lib().qmg_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(self.nbytes), None), "d2h")
        return out

    def read(self, first, count):
        """`count` elements starting at element `first`, copied to the host (spot checks on arrays too large to download)."""
        out = np.empty(int(count), dtype=self.dtype)
        sync()
        check(lib().qmg_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.offset(first)), C.c_size_t(out.nbytes), None), "d2h")
        return out

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.size == self.n
        check(lib().qmg_memcpy_h2d(C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p), C.c_size_t(self.nbytes), None), "h2d")

    def offset(self, elems):
        return self.ptr + int(elems) * self.dtype.itemsize

    def free(self):
        if self.ptr:
            lib().qmg_free(C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _vp(x):
    if x is None:
        return C.c_void_p(None)
    if isinstance(x, DeviceArray):
        return C.c_void_p(x.ptr)
    return C.c_void_p(int(x))


def make_desc(Lx, Ly, nc, clover, hopping, shift=0.0, eo_shift=0.0, dof_shift=0.0):
    d = StencilDesc()
    d.Lx, d.Ly, d.nc = Lx, Ly, nc
    d.clover = None if clover is None else (clover.ptr if isinstance(clover, DeviceArray) else int(clover))
    d.hopping = None if hopping is None else (hopping.ptr if isinstance(hopping, DeviceArray) else int(hopping))
    for name, v in (("shift", shift), ("eo_shift", eo_shift), ("dof_shift", dof_shift)):
        v = complex(v)
        getattr(d, name)[0], getattr(d, name)[1] = v.real, v.imag
    d._keep = (clover, hopping)
    return d


def stencil_apply(desc, lhs, rhs, pieces=P_ALL | P_ZERO, nrhs=1, vec_stride=0, stream=None):
    check(lib().qmg_stencil_apply(C.byref(desc), _vp(lhs), _vp(rhs), C.c_uint(pieces), C.c_int(nrhs),
                                  C.c_size_t(vec_stride), C.c_void_p(stream)), "qmg_stencil_apply")


def stencil_apply_norm2(desc, lhs, rhs, pieces=P_ALL | P_ZERO, nrhs=1, vec_stride=0, norms_dev=None, stream=None):
    """The apply and |lhs_k|^2 of its results in one pass.  With norms_dev (a device pointer to nrhs doubles) nothing
    synchronises and None is returned; otherwise the norms come back as a numpy array."""
    out = None if norms_dev is not None else np.full(nrhs, np.nan)
    check(lib().qmg_stencil_apply_norm2(C.byref(desc), _vp(lhs), _vp(rhs), C.c_uint(pieces), C.c_int(nrhs), C.c_size_t(vec_stride),
                                        C.c_void_p(norms_dev), None if out is None else out.ctypes.data_as(C.POINTER(C.c_double)),
                                        C.c_void_p(stream)), "qmg_stencil_apply_norm2")
    return out


def cshift(lhs, rhs, cdir, eo, dof, Lx, Ly, stream=None):
    check(lib().qmg_cshift(_vp(lhs), _vp(rhs), cdir, eo, dof, Lx, Ly, C.c_void_p(stream)), "qmg_cshift")


def wilson_fill(clover, hopping, gauge, Lx, Ly, w=1.0, stream=None):
    check(lib().qmg_wilson_fill(_vp(clover), _vp(hopping), _vp(gauge), Lx, Ly, C.c_double(w), C.c_void_p(stream)), "qmg_wilson_fill")


def staggered_fill(hopping, gauge, Lx, Ly, stream=None):
    check(lib().qmg_staggered_fill(_vp(hopping), _vp(gauge), Lx, Ly, C.c_void_p(stream)), "qmg_staggered_fill")


def laplace_fill(clover, hopping, gauge, Lx, Ly, stream=None):
    check(lib().qmg_laplace_fill(_vp(clover), _vp(hopping), _vp(gauge), Lx, Ly, C.c_void_p(stream)), "qmg_laplace_fill")


def build_dagger(dclover, dhopping, clover, hopping, Lx, Ly, nc, stream=None):
    check(lib().qmg_build_dagger(_vp(dclover), _vp(dhopping), _vp(clover), _vp(hopping), Lx, Ly, nc, C.c_void_p(stream)), "qmg_build_dagger")


def staggered_fill_slab(hopping, gauge_global, Lx, Ly_global, y0, Ly_local, stream=None):
    check(lib().qmg_staggered_fill_slab(_vp(hopping), _vp(gauge_global), Lx, Ly_global, y0, Ly_local, C.c_void_p(stream)), "qmg_staggered_fill_slab")


def laplace_fill_slab(clover, hopping, gauge_global, Lx, Ly_global, y0, Ly_local, stream=None):
    check(lib().qmg_laplace_fill_slab(_vp(clover), _vp(hopping), _vp(gauge_global), Lx, Ly_global, y0, Ly_local, C.c_void_p(stream)), "qmg_laplace_fill_slab")


def build_dagger_slab(dclover, dhopping, clover, hopping, Lx, Ly, nc, ym_halo_hi, yp_halo_lo, stream=None):
    check(lib().qmg_build_dagger_slab(_vp(dclover), _vp(dhopping), _vp(clover), _vp(hopping), Lx, Ly, nc, _vp(ym_halo_hi), _vp(yp_halo_lo), C.c_void_p(stream)),
          "qmg_build_dagger_slab")


def build_rbjacobi(cinv, rclover, rhopping, desc, stream=None):
    check(lib().qmg_build_rbjacobi(_vp(cinv), _vp(rclover), _vp(rhopping), C.byref(desc), C.c_void_p(stream)), "qmg_build_rbjacobi")


def cmat_conjtrans(out, inp, nsite, nc, stream=None):
    """out[i] = in[i]^dagger for nsite row-major nc x nc blocks (cMATcopy_conjtrans_square; stencil_2d.h:2012-2020: cinv of the rbj-dagger stencil)."""
    check(lib().qmg_cmat_conjtrans(_vp(out), _vp(inp), C.c_size_t(nsite), nc, C.c_void_p(stream)), "qmg_cmat_conjtrans")


def _scalar(a):
    a = complex(a)
    return C.c_double(a.real), C.c_double(a.imag)


def zero_vector(x, n):
    check(lib().qmg_zero_vector(_vp(x), C.c_size_t(n), None))


def copy_vector(dst, src, n):
    check(lib().qmg_copy_vector(_vp(dst), _vp(src), C.c_size_t(n), None))


def cax(a, x, n):
    check(lib().qmg_cax(*_scalar(a), _vp(x), C.c_size_t(n), None))


def caxy(a, x, y, n):
    check(lib().qmg_caxy(*_scalar(a), _vp(x), _vp(y), C.c_size_t(n), None))


def caxpy(a, x, y, n):
    check(lib().qmg_caxpy(*_scalar(a), _vp(x), _vp(y), C.c_size_t(n), None))


def cxpy(x, y, n):
    check(lib().qmg_cxpy(_vp(x), _vp(y), C.c_size_t(n), None))


def cxpay(x, a, y, n):
    check(lib().qmg_cxpay(_vp(x), *_scalar(a), _vp(y), C.c_size_t(n), None))


def caxpby(a, x, b, y, n):
    check(lib().qmg_caxpby(*_scalar(a), _vp(x), *_scalar(b), _vp(y), C.c_size_t(n), None))


def cxpyz(x, y, z, n):
    check(lib().qmg_cxpyz(_vp(x), _vp(y), _vp(z), C.c_size_t(n), None))


def caxpbyz(a, x, b, y, z, n):
    check(lib().qmg_caxpbyz(*_scalar(a), _vp(x), *_scalar(b), _vp(y), _vp(z), C.c_size_t(n), None))


def multi_caxpy(coeffs, xs, y, n):
    k = len(xs)
    cf = (C.c_double * (2 * k))(*[v for a in coeffs for v in (complex(a).real, complex(a).imag)])
    ptrs = (C.c_void_p * k)(*[(x.ptr if isinstance(x, DeviceArray) else int(x)) for x in xs])
    check(lib().qmg_multi_caxpy(cf, ptrs, k, _vp(y), C.c_size_t(n), None), "qmg_multi_caxpy")


def caxy_pattern(scale, shuffle, x, y, nsite):
    nc = len(scale)
    sc = (C.c_double * nc)(*scale)
    sh = (C.c_int * nc)(*shuffle)
    check(lib().qmg_caxy_pattern(sc, sh, nc, _vp(x), _vp(y), C.c_size_t(nsite), None))


def gaussian(x, n, seed):
    check(lib().qmg_gaussian(_vp(x), C.c_size_t(n), C.c_ulonglong(seed), None))


def norm2sq(x, n):
    out = C.c_double()
    check(lib().qmg_norm2sq(_vp(x), C.c_size_t(n), None, C.byref(out), None))
    return out.value


def dot(x, y, n):
    out = (C.c_double * 2)()
    check(lib().qmg_dot(_vp(x), _vp(y), C.c_size_t(n), None, out, None))
    return complex(out[0], out[1])


def diffnorm2sq(x, y, n):
    out = C.c_double()
    check(lib().qmg_diffnorm2sq(_vp(x), _vp(y), C.c_size_t(n), None, C.byref(out), None))
    return out.value


def norminf(x, n):
    out = C.c_double()
    check(lib().qmg_norminf(_vp(x), C.c_size_t(n), None, C.byref(out), None))
    return out.value


def multidot(xs, y, n):
    k = len(xs)
    ptrs = (C.c_void_p * k)(*[(x.ptr if isinstance(x, DeviceArray) else int(x)) for x in xs])
    out = (C.c_double * (2 * k))()
    check(lib().qmg_multidot(ptrs, k, _vp(y), C.c_size_t(n), None, out, None))
    return np.array([complex(out[2 * i], out[2 * i + 1]) for i in range(k)])


def norm2sq_cv_timeslice(cv, Lx, Ly, nc):
    out = np.zeros(Ly)
    check(lib().qmg_norm2sq_cv_timeslice(_vp(cv), Lx, Ly, nc, None, out.ctypes.data_as(C.POINTER(C.c_double)), None))
    return out


def dot_cv_timeslice(a, b, Lx, Ly, nc):
    out = np.zeros(2 * Ly)
    check(lib().qmg_dot_cv_timeslice(_vp(a), _vp(b), Lx, Ly, nc, None, out.ctypes.data_as(C.POINTER(C.c_double)), None))
    return out[0::2] + 1j * out[1::2]


def redot_cv_timeslice(a, b, Lx, Ly, nc):
    out = np.zeros(Ly)
    check(lib().qmg_redot_cv_timeslice(_vp(a), _vp(b), Lx, Ly, nc, None, out.ctypes.data_as(C.POINTER(C.c_double)), None))
    return out


def gaussian_wall_source(cv, Lx, Ly, nc, timeslice, color, seed, deviation=1.0, mean=0.0):
    """reductions/reductions.h:90-162; returns the status instead of raising for an out-of-range timeslice / color (the reference prints and returns)."""
    return lib().qmg_gaussian_wall_source(_vp(cv), Lx, Ly, nc, timeslice, color, C.c_ulonglong(seed), C.c_double(deviation), C.c_double(mean), None)


def prolong(nullvecs, nvec, coarse, fine, fdims, cdims):
    check(lib().qmg_prolong(_vp(nullvecs), nvec, _vp(coarse), _vp(fine), *fdims, *cdims, None), "qmg_prolong")


def restrict(nullvecs, nvec, fine, coarse, fdims, cdims):
    check(lib().qmg_restrict(_vp(nullvecs), nvec, _vp(fine), _vp(coarse), *fdims, *cdims, None), "qmg_restrict")


def block_orthonormalize(nullvecs, nvec, fdims, cLx, cLy, cholesky=None):
    check(lib().qmg_block_orthonormalize(_vp(nullvecs), nvec, *fdims, cLx, cLy, _vp(cholesky), None), "qmg_block_orthonormalize")


def block_bi_orthonormalize(pvecs, rvecs, nvec, fdims, cLx, cLy, block_L=None, block_U=None):
    check(lib().qmg_block_bi_orthonormalize(_vp(pvecs), _vp(rvecs), nvec, *fdims, cLx, cLy, _vp(block_L), _vp(block_U), None), "qmg_block_bi_orthonormalize")


def coarse_build(cclover, chopping, fdesc, nullvecs, cdims, restrict_vecs=None):
    check(lib().qmg_coarse_build(_vp(cclover), _vp(chopping), C.byref(fdesc), _vp(nullvecs), _vp(restrict_vecs), *cdims, None), "qmg_coarse_build")


def coarse_build_slab(cclover, chopping, fdesc, nullvecs, cdims, halo_lo, halo_hi, halo_stride, restrict_vecs=None):
    check(lib().qmg_coarse_build_slab(_vp(cclover), _vp(chopping), C.byref(fdesc), _vp(nullvecs), _vp(restrict_vecs), *cdims, _vp(halo_lo), _vp(halo_hi),
                                      C.c_size_t(halo_stride), None), "qmg_coarse_build_slab")


class ApplyEpilogue(C.Structure):
    _fields_ = [("other", C.c_void_p), ("other_scale", C.c_double), ("acc_scale", C.c_double), ("dotv", C.c_void_p)]


def make_epilogue(other=None, other_scale=1.0, acc_scale=-1.0, dotv=None):
    e = ApplyEpilogue()
    e.other = None if other is None else (other.ptr if isinstance(other, DeviceArray) else int(other))
    e.other_scale, e.acc_scale = other_scale, acc_scale
    e.dotv = None if dotv is None else (dotv.ptr if isinstance(dotv, DeviceArray) else int(dotv))
    return e


def stencil_apply_epi(dtype, mat32, desc, lhs, rhs, pieces, epi, vec_stride=0, system=0):
    """status (0 = done, 3 = this operator is not served with an epilogue) of qmg_stencil_apply_epi_t"""
    return lib().qmg_stencil_apply_epi_t(dtype, int(mat32), C.byref(desc), _vp(lhs), _vp(rhs), C.c_uint(pieces), C.c_size_t(vec_stride), system, C.byref(epi), None)


def wilson_apply_direct_epi(dtype, desc, gauge, lhs, rhs, pieces, epi, nrhs=1, vec_stride=0, mask=1, wilson_coeff=1.0):
    return lib().qmg_wilson_apply_direct_epi(dtype, C.byref(desc), _vp(gauge), desc.Ly, 0, C.c_double(wilson_coeff), _vp(lhs), _vp(rhs), None, None, C.c_uint(pieces), nrhs,
                                             C.c_size_t(vec_stride), C.c_size_t(0), C.c_uint(mask), C.byref(epi), None)


def wilson_hops_direct_epi(dtype, desc, gauge, hop_scale, lhs, rhs, pieces, epi, nrhs=1, vec_stride=0, mask=1, wilson_coeff=1.0):
    return lib().qmg_wilson_hops_direct_epi(dtype, C.byref(desc), _vp(gauge), desc.Ly, 0, C.c_double(wilson_coeff), C.c_double(hop_scale), _vp(lhs), _vp(rhs), None, None,
                                            C.c_uint(pieces), nrhs, C.c_size_t(vec_stride), C.c_size_t(0), C.c_uint(mask), C.byref(epi), None)


def batch_mr_dots(dtype, r, p, n, nrhs, stride, mask):
    check(lib().qmg_batch_mr_dots_t(dtype, _vp(r), _vp(p), C.c_size_t(n), nrhs, C.c_size_t(stride), C.c_uint(mask), None), "qmg_batch_mr_dots_t")


def batch_mr_update(dtype, omega, x, r_in, r_out, p, x_set, n, nrhs, stride, mask):
    check(lib().qmg_batch_mr_update_t(dtype, C.c_double(omega), _vp(x), _vp(r_in), _vp(r_out), _vp(p), int(x_set), C.c_size_t(n), nrhs, C.c_size_t(stride), C.c_uint(mask), None),
          "qmg_batch_mr_update_t")


def batch_mr_read_dots(nrhs):
    out = (C.c_double * (4 * nrhs))()
    check(lib().qmg_batch_mr_read_dots(out, nrhs, None), "qmg_batch_mr_read_dots")
    return np.array(out).reshape(nrhs, 4)


def gaussian_slab(x, Lx, Ly_global, y0, Ly_local, nc, seed):
    check(lib().qmg_gaussian_slab(_vp(x), Lx, Ly_global, y0, Ly_local, nc, C.c_ulonglong(seed), None), "qmg_gaussian_slab")


# ---------------- lock-step batches (qmg_batch.hip) ----------------
BOP_ZERO, BOP_COPY, BOP_CAX, BOP_CAXPY, BOP_CXPY, BOP_CAXPBYZ = range(6)
BRED_NORM2, BRED_DOT, BRED_DIFFNORM2 = range(3)


def _coef(a, nrhs):
    if a is None:
        return None
    v = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.complex128), (nrhs,))).view(np.float64)
    return v.ctypes.data_as(C.POINTER(C.c_double)), v


def stencil_apply_batch(desc, lhs, rhs, pieces, nrhs, vec_stride, mask, stream=None):
    check(lib().qmg_stencil_apply_batch(C.byref(desc), _vp(lhs), _vp(rhs), C.c_uint(pieces), nrhs, C.c_size_t(vec_stride), C.c_uint(mask), stream),
          "qmg_stencil_apply_batch")


def stencil_apply_mat32(desc, lhs, rhs, pieces, nrhs, vec_stride, mask, stream=None):
    check(lib().qmg_stencil_apply_mat32(C.byref(desc), _vp(lhs), _vp(rhs), C.c_uint(pieces), nrhs, C.c_size_t(vec_stride), C.c_uint(mask), stream),
          "qmg_stencil_apply_mat32")


def stencil_apply_mat16(vec_dtype, desc, lhs, rhs, pieces, nrhs=1, vec_stride=0, mask=1, stream=None):
    """status of qmg_stencil_apply_mat16_t: complex<half> matrices (convert_to_c16), vectors of vec_dtype"""
    return lib().qmg_stencil_apply_mat16_t(vec_dtype, C.byref(desc), _vp(lhs), _vp(rhs), C.c_uint(pieces), nrhs, C.c_size_t(vec_stride), C.c_uint(mask), stream)


def convert_from_c16(dst, dst_dtype, src, n):
    check(lib().qmg_convert_from_c16(_vp(dst), dst_dtype, _vp(src), C.c_size_t(n), None), "qmg_convert_from_c16")


def c64_to_c32(dst, src, n):
    check(lib().qmg_c64_to_c32(_vp(dst), _vp(src), C.c_size_t(n), None), "qmg_c64_to_c32")


def batch_blas(op, z, n, nrhs, stride, mask, a=None, b=None, x=None, y=None):
    ca, cb = _coef(a, nrhs), _coef(b, nrhs)
    check(lib().qmg_batch_blas(op, ca[0] if ca else None, cb[0] if cb else None, _vp(x), _vp(y), _vp(z), C.c_size_t(n), nrhs, C.c_size_t(stride),
                               C.c_uint(mask), None), "qmg_batch_blas")


def batch_multi_caxpy(coeffs, xs, y, n, nrhs, stride, mask):
    nj = len(xs)
    cf = np.ascontiguousarray(np.asarray(coeffs, dtype=np.complex128).reshape(nj, nrhs)).view(np.float64)
    ptrs = (C.c_void_p * nj)(*[x.ptr for x in xs])
    check(lib().qmg_batch_multi_caxpy(cf.ctypes.data_as(C.POINTER(C.c_double)), ptrs, nj, _vp(y), C.c_size_t(n), nrhs, C.c_size_t(stride), C.c_uint(mask), None),
          "qmg_batch_multi_caxpy")


def batch_reduce(op, x, y, n, nrhs, stride, mask):
    out = np.full(2 * nrhs, np.nan)
    check(lib().qmg_batch_reduce(op, _vp(x), _vp(y), C.c_size_t(n), nrhs, C.c_size_t(stride), C.c_uint(mask), out.ctypes.data_as(C.POINTER(C.c_double)), None),
          "qmg_batch_reduce")
    return out[0::2] + 1j * out[1::2]


def batch_multidot(xs, y, n, nrhs, stride, mask):
    nj = len(xs)
    ptrs = (C.c_void_p * nj)(*[x.ptr for x in xs])
    out = np.full(2 * nrhs * nj, np.nan)
    check(lib().qmg_batch_multidot(ptrs, nj, _vp(y), C.c_size_t(n), nrhs, C.c_size_t(stride), C.c_uint(mask), out.ctypes.data_as(C.POINTER(C.c_double)), None),
          "qmg_batch_multidot")
    return (out[0::2] + 1j * out[1::2]).reshape(nrhs, nj)


def prolong_batch(nullvecs, nvec, coarse, fine, fdims, cdims, nrhs, cstride, fstride, mask):
    check(lib().qmg_prolong_batch(_vp(nullvecs), nvec, _vp(coarse), _vp(fine), *fdims, *cdims, nrhs, C.c_size_t(cstride), C.c_size_t(fstride), C.c_uint(mask), None),
          "qmg_prolong_batch")


def restrict_batch(nullvecs, nvec, fine, coarse, fdims, cdims, nrhs, fstride, cstride, mask):
    check(lib().qmg_restrict_batch(_vp(nullvecs), nvec, _vp(fine), _vp(coarse), *fdims, *cdims, nrhs, C.c_size_t(fstride), C.c_size_t(cstride), C.c_uint(mask), None),
          "qmg_restrict_batch")


# ---------------- either storage precision (`_t` entry points; dtype = C64 | C32) ----------------
C64, C32 = 0, 1
NP_DTYPE = {C64: np.complex128, C32: np.complex64}


def convert(dst, dst_dtype, src, src_dtype, n):
    check(lib().qmg_convert(_vp(dst), dst_dtype, _vp(src), src_dtype, C.c_size_t(n), None), "qmg_convert")


def stencil_apply_t(dtype, desc, lhs, rhs, pieces, nrhs=1, vec_stride=0, mask=1, stream=None):
    check(lib().qmg_stencil_apply_t(dtype, C.byref(desc), _vp(lhs), _vp(rhs), C.c_uint(pieces), nrhs, C.c_size_t(vec_stride), C.c_uint(mask), stream),
          "qmg_stencil_apply_t")


def batch_blas_t(dtype, op, z, n, nrhs, stride, mask, a=None, b=None, x=None, y=None):
    ca, cb = _coef(a, nrhs), _coef(b, nrhs)
    check(lib().qmg_batch_blas_t(dtype, op, ca[0] if ca else None, cb[0] if cb else None, _vp(x), _vp(y), _vp(z), C.c_size_t(n), nrhs, C.c_size_t(stride),
                                 C.c_uint(mask), None), "qmg_batch_blas_t")


def batch_multi_caxpy_t(dtype, coeffs, xs, y, n, nrhs, stride, mask):
    nj = len(xs)
    cf = np.ascontiguousarray(np.asarray(coeffs, dtype=np.complex128).reshape(nj, nrhs)).view(np.float64)
    ptrs = (C.c_void_p * nj)(*[x.ptr for x in xs])
    check(lib().qmg_batch_multi_caxpy_t(dtype, cf.ctypes.data_as(C.POINTER(C.c_double)), ptrs, nj, _vp(y), C.c_size_t(n), nrhs, C.c_size_t(stride), C.c_uint(mask), None),
          "qmg_batch_multi_caxpy_t")


def batch_gcr_update_t(dtype, coeffs, ws, w, a, r, z_next, n, nrhs, stride, mask):
    """w += sum_j c_j ws_j ; r += a w ; z_next = r (optional) in one pass (qmg_batch_gcr_update_t)"""
    nj = len(ws)
    cf = np.ascontiguousarray(np.asarray(coeffs, dtype=np.complex128).reshape(max(nj, 1), nrhs)).view(np.float64) if nj else None
    ptrs = (C.c_void_p * max(nj, 1))(*[x.ptr for x in ws]) if nj else None
    av = np.ascontiguousarray(np.asarray(a, dtype=np.complex128).reshape(nrhs)).view(np.float64)
    check(lib().qmg_batch_gcr_update_t(dtype, cf.ctypes.data_as(C.POINTER(C.c_double)) if nj else None, ptrs, nj, _vp(w), av.ctypes.data_as(C.POINTER(C.c_double)), _vp(r),
                                       _vp(z_next), C.c_size_t(n), nrhs, C.c_size_t(stride), C.c_uint(mask), None), "qmg_batch_gcr_update_t")


def batch_reduce_t(dtype, op, x, y, n, nrhs, stride, mask):
    out = np.full(2 * nrhs, np.nan)
    check(lib().qmg_batch_reduce_t(dtype, op, _vp(x), _vp(y), C.c_size_t(n), nrhs, C.c_size_t(stride), C.c_uint(mask), out.ctypes.data_as(C.POINTER(C.c_double)), None),
          "qmg_batch_reduce_t")
    return out[0::2] + 1j * out[1::2]


def batch_multidot_t(dtype, xs, y, n, nrhs, stride, mask):
    nj = len(xs)
    ptrs = (C.c_void_p * nj)(*[x.ptr for x in xs])
    out = np.full(2 * nrhs * nj, np.nan)
    check(lib().qmg_batch_multidot_t(dtype, ptrs, nj, _vp(y), C.c_size_t(n), nrhs, C.c_size_t(stride), C.c_uint(mask), out.ctypes.data_as(C.POINTER(C.c_double)), None),
          "qmg_batch_multidot_t")
    return (out[0::2] + 1j * out[1::2]).reshape(nrhs, nj)


def prolong_batch_t(dtype, nullvecs, nvec, coarse, fine, fdims, cdims, nrhs, cstride, fstride, mask):
    check(lib().qmg_prolong_batch_t(dtype, _vp(nullvecs), nvec, _vp(coarse), _vp(fine), *fdims, *cdims, nrhs, C.c_size_t(cstride), C.c_size_t(fstride), C.c_uint(mask), None),
          "qmg_prolong_batch_t")


def restrict_batch_t(dtype, nullvecs, nvec, fine, coarse, fdims, cdims, nrhs, fstride, cstride, mask):
    check(lib().qmg_restrict_batch_t(dtype, _vp(nullvecs), nvec, _vp(fine), _vp(coarse), *fdims, *cdims, nrhs, C.c_size_t(fstride), C.c_size_t(cstride), C.c_uint(mask), None),
          "qmg_restrict_batch_t")


def prolong_batch_nv32(null32, nvec, coarse, fine, fdims, cdims, nrhs, cstride, fstride, mask):
    """complex<double> vectors, complex<float> null vectors (qmg_prolong_batch_nv32)"""
    check(lib().qmg_prolong_batch_nv32(_vp(null32), nvec, _vp(coarse), _vp(fine), *fdims, *cdims, nrhs, C.c_size_t(cstride), C.c_size_t(fstride), C.c_uint(mask), None),
          "qmg_prolong_batch_nv32")


def restrict_batch_nv32(null32, nvec, fine, coarse, fdims, cdims, nrhs, fstride, cstride, mask):
    check(lib().qmg_restrict_batch_nv32(_vp(null32), nvec, _vp(fine), _vp(coarse), *fdims, *cdims, nrhs, C.c_size_t(fstride), C.c_size_t(cstride), C.c_uint(mask), None),
          "qmg_restrict_batch_nv32")


def convert_to_c16(dst, src, src_dtype, n):
    check(lib().qmg_convert_to_c16(_vp(dst), _vp(src), src_dtype, C.c_size_t(n), None), "qmg_convert_to_c16")


def stencil_apply_h16(desc, lhs, rhs, pieces, nrhs=1, vec_stride=0, mask=1, stream=None):
    check(lib().qmg_stencil_apply_h16(C.byref(desc), _vp(lhs), _vp(rhs), C.c_uint(pieces), nrhs, C.c_size_t(vec_stride), C.c_uint(mask), stream),
          "qmg_stencil_apply_h16")


def comm_init_env(world, rank):
    check(lib().qmg_comm_init_env(world, rank), "qmg_comm_init_env")


SLAB_H16 = 0x100
SLAB_M32 = 0x200   # nc != 2: complex<float> matrices whatever the vectors' type
SLAB_M16 = 0x400   # nc != 2, a multiple of 4: complex<half> matrices


def halo_exchange(dtype, vec, Lx, Ly_local, nc, halo_lo, halo_hi, nrhs=1, vec_stride=0, halo_stride=0, stream=None):
    check(lib().qmg_halo_exchange(dtype, _vp(vec), Lx, Ly_local, nc, _vp(halo_lo), _vp(halo_hi), nrhs, C.c_size_t(vec_stride), C.c_size_t(halo_stride), stream),
          "qmg_halo_exchange")


def stencil_apply_slab(storage, desc, lhs, rhs, halo_lo, halo_hi, pieces, nrhs=1, vec_stride=0, halo_stride=0, mask=1, rows=0, stream=None):
    check(lib().qmg_stencil_apply_slab(storage, C.byref(desc), _vp(lhs), _vp(rhs), _vp(halo_lo), _vp(halo_hi), C.c_uint(pieces), nrhs, C.c_size_t(vec_stride),
                                       C.c_size_t(halo_stride), C.c_uint(mask), rows, stream), "qmg_stencil_apply_slab")


def wilson_fill_slab(clover, hopping, gauge_global, Lx, Ly_global, y0, Ly_local, w=1.0, stream=None):
    check(lib().qmg_wilson_fill_slab(_vp(clover), _vp(hopping), _vp(gauge_global), Lx, Ly_global, y0, Ly_local, C.c_double(w), stream), "qmg_wilson_fill_slab")


def wilson_apply_direct(dtype, desc, gauge, lhs, rhs, pieces, w=1.0, nrhs=1, vec_stride=0, mask=1, gauge_Ly=None, y0=0, halo_lo=None, halo_hi=None,
                        halo_stride=0, rows=0, stream=None):
    check(lib().qmg_wilson_apply_direct(dtype, C.byref(desc), _vp(gauge), desc.Ly if gauge_Ly is None else gauge_Ly, y0, C.c_double(w), _vp(lhs), _vp(rhs),
                                        _vp(halo_lo), _vp(halo_hi), C.c_uint(pieces), nrhs, C.c_size_t(vec_stride), C.c_size_t(halo_stride), C.c_uint(mask), rows,
                                        stream), "qmg_wilson_apply_direct")


def wilson_hops_direct(dtype, desc, gauge, lhs, rhs, pieces, w, hop_scale, nrhs=1, vec_stride=0, mask=1, gauge_Ly=None, y0=0, halo_lo=None, halo_hi=None,
                       halo_stride=0, rows=0, stream=None):
    check(lib().qmg_wilson_hops_direct(dtype, C.byref(desc), _vp(gauge), desc.Ly if gauge_Ly is None else gauge_Ly, y0, C.c_double(w), C.c_double(hop_scale),
                                       _vp(lhs), _vp(rhs), _vp(halo_lo), _vp(halo_hi), C.c_uint(pieces), nrhs, C.c_size_t(vec_stride), C.c_size_t(halo_stride),
                                       C.c_uint(mask), rows, stream), "qmg_wilson_hops_direct")


def comm_set_distributed_reductions(on):
    check(lib().qmg_comm_set_distributed_reductions(1 if on else 0), "qmg_comm_set_distributed_reductions")


def comm_all_ok(ok):
    out = C.c_int(0)
    check(lib().qmg_comm_all_ok(1 if ok else 0, C.byref(out)), "qmg_comm_all_ok")
    return bool(out.value)


def u1_heatbath_noncompact(phase, Lx, Ly, beta, n_update, seed, first_sweep=0):
    check(lib().qmg_u1_heatbath_noncompact(_vp(phase), Lx, Ly, C.c_double(beta), n_update, C.c_ulonglong(seed), C.c_ulonglong(first_sweep), None), "qmg_u1_heatbath_noncompact")


def u1_phase_to_gauge(gauge, phase, n):
    check(lib().qmg_u1_phase_to_gauge(_vp(gauge), _vp(phase), C.c_size_t(n), None), "qmg_u1_phase_to_gauge")


def u1_gauge_to_phase(phase, gauge, n):
    check(lib().qmg_u1_gauge_to_phase(_vp(phase), _vp(gauge), C.c_size_t(n), None), "qmg_u1_gauge_to_phase")


def u1_plaquette(gauge, Lx, Ly):
    """(average plaquette, topological charge)"""
    out = (C.c_double * 3)()
    check(lib().qmg_u1_plaquette(_vp(gauge), Lx, Ly, out, None), "qmg_u1_plaquette")
    return complex(out[0], out[1]), out[2]


def u1_noncompact_action(phase, Lx, Ly, beta):
    out = C.c_double()
    check(lib().qmg_u1_noncompact_action(_vp(phase), Lx, Ly, C.c_double(beta), C.byref(out), None), "qmg_u1_noncompact_action")
    return out.value


def set_tuning(key, value):
    check(lib().qmg_set_tuning(key.encode(), int(value)), "qmg_set_tuning")


class Timer:
    """HIP-event timing on the stream the kernels are launched on (NULL stream by default)."""

    def __init__(self):
        self.a, self.b = C.c_void_p(), C.c_void_p()
        check(lib().qmg_event_create(C.byref(self.a)))
        check(lib().qmg_event_create(C.byref(self.b)))

    def start(self, stream=None):
        check(lib().qmg_event_record(self.a, C.c_void_p(stream)))

    def stop_ms(self, stream=None):
        check(lib().qmg_event_record(self.b, C.c_void_p(stream)))
        ms = C.c_float()
        check(lib().qmg_event_elapsed_ms(self.a, self.b, C.byref(ms)))
        return ms.value
