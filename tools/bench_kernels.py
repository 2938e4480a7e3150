"""Per-kernel roofline table for the hot-path rows of SURVEY 8(a): each kernel timed with HIP events on the launch
stream at a BASELINE-relevant size, achieved GB/s from the ALGORITHMIC byte count, and the CPU oracle
(reference pass structure, 1 thread) timed on a bounded sample of the same operation beside it.

    python tools/bench_kernels.py > profiles/rNN_kernel_rooflines.json
"""
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import oracle_lib as ol  # noqa: E402

qmg = importlib.import_module("quantum-mg_amd")
qmg.init(0)
PEAK = 8000.0
C = 16  # bytes per complex128
fixture = os.path.join(ROOT, "tests", "golden", "l64t64b60_heatbath.dat")
rows = []


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    qmg.sync()
    t = qmg.Timer()
    t.start()
    for _ in range(reps):
        fn()
    return t.stop_ms() / reps


def cpu_time(fn, budget=3.0):
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter() - t0
    reps = max(1, min(20, int(budget / max(t1, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def row(name, ref, ms, alg_bytes, flops, cpu_s=None, cpu_scale=1.0, note=""):
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    r = {"kernel": name, "reference": ref, "ms": ms, "algorithmic_GB": alg_bytes / 1e9, "achieved_GBps": gbs, "frac_of_hbm_peak": gbs / PEAK,
         "GFLOPs": flops / (ms * 1e-3) / 1e9, "note": note}
    if cpu_s is not None:
        r["cpu_oracle_ms_same_size"] = cpu_s * cpu_scale * 1e3
        r["cpu_sample_scale"] = cpu_scale
        r["gpu_over_cpu"] = cpu_s * cpu_scale * 1e3 / ms
    rows.append(r)
    print("%-34s %9.3f ms %8.0f GB/s (%.0f%%)" % (name, ms, gbs, 100 * gbs / PEAK), file=sys.stderr)


def gauss(n, seed):
    d = qmg.DeviceArray(n)
    qmg.gaussian(d, n, seed)
    return d


# ---------------- fine operators at 4096^2
L = 4096
vol = L * L
g = qmg.DeviceArray.from_host(bench.tiled_gauge(L, fixture))
# CPU samples at 512^2 scaled by area
Ls = 512
gs = bench.tiled_gauge(Ls, fixture)
scale = (L / Ls) ** 2

cl, ho = qmg.DeviceArray(4 * vol), qmg.DeviceArray(16 * vol)
qmg.wilson_fill(cl, ho, g, L, L)
d = qmg.make_desc(L, L, 2, cl, ho, -0.07)
x, y = gauss(2 * vol, 1), qmg.DeviceArray(2 * vol)
ocl, oho = ol.wilson_fill(gs, Ls, Ls)
od = ol.make_desc(Ls, Ls, 2, ocl, oho, -0.07)
xs = np.random.default_rng(0).standard_normal(2 * Ls * Ls) + 0j
row("wilson apply_M (both parities)", "stencil_2d.h:912-936", timeit(lambda: qmg.stencil_apply(d, y, x)), 384 * vol, 176 * vol,
    cpu_time(lambda: ol.stencil_apply(od, xs)), scale)
row("wilson apply_M_eo (one parity)", "stencil_2d.h:706-733", timeit(lambda: qmg.stencil_apply(d, y, x, qmg.P_EO | qmg.P_ZERO_E)), (4 * 64 + 32 + 32) * vol // 2,
    (8 * 4 * 4 + 0) * vol // 2, cpu_time(lambda: ol.stencil_apply(od, xs, ol.P_EO | ol.P_ZERO_E)), scale)
# right-block-Jacobi apply = identity (as unit shift) + hopping: no clover read
d_rb = qmg.make_desc(L, L, 2, None, ho, 1.0)
row("rbjacobi apply (1 + H')", "stencil_2d.h:1818-1845", timeit(lambda: qmg.stencil_apply(d_rb, y, x, qmg.P_HOPPING | qmg.P_SHIFT | qmg.P_ZERO)), (4 * 64 + 64) * vol, (8 * 4 * 4 + 16) * vol)
cl.free(); ho.free()

ho1 = qmg.DeviceArray(4 * vol)
qmg.staggered_fill(ho1, g, L, L)
d1 = qmg.make_desc(L, L, 1, None, ho1, 0.04)
ohs = ol.staggered_fill(gs, Ls, Ls)
od1 = ol.make_desc(Ls, Ls, 1, None, ohs, 0.04)
xs1 = np.random.default_rng(1).standard_normal(Ls * Ls) + 0j
row("staggered apply_M, 1 rhs", "staggered.h + stencil_2d.h:912", timeit(lambda: qmg.stencil_apply(d1, y, x)), 96 * vol, 40 * vol, cpu_time(lambda: ol.stencil_apply(od1, xs1)), scale)
x8, y8 = gauss(8 * vol, 2), qmg.DeviceArray(8 * vol)
row("staggered apply_M, 8 rhs batched", "SURVEY 8d multi-RHS", timeit(lambda: qmg.stencil_apply(d1, y8, x8, nrhs=8, vec_stride=vol)), (64 + 8 * 32) * vol, 8 * 40 * vol,
    note="matrices read once for 8 right-hand sides")
clp = qmg.DeviceArray(vol)
qmg.laplace_fill(clp, ho1, g, L, L)
d1l = qmg.make_desc(L, L, 1, clp, ho1, 0.01)
row("gauged Laplace apply_M", "gaugedlaplace.h + stencil_2d.h:912", timeit(lambda: qmg.stencil_apply(d1l, y, x)), 112 * vol, 48 * vol)
# cshift (kept for setup paths)
row("cshift FROM_XP1 both parities, dof 2", "cshift_2d.h:45-236", timeit(lambda: qmg.cshift(y, x, qmg.CSHIFT_XP1, qmg.EO_FROM_EVENODD, 2, L, L)), 2 * 32 * vol, 0,
    cpu_time(lambda: ol.cshift(xs.copy(), ol.CSHIFT_XP1, ol.EO_FROM_EVENODD, 2, Ls, Ls)), scale)
for a in (ho1, clp, x8, y8, g):
    a.free()

# ---------------- BLAS-1 / reductions at the fine vector size (2 x 4096^2 complex)
n = 2 * vol
dres = qmg.DeviceArray.zeros(8)
xs_b = np.random.default_rng(2).standard_normal(2 * Ls * Ls) + 0j
ys_b = xs_b[::-1].copy()
row("norm2sq", "qlinalg norm2sq (stateful_multigrid.h:880)", timeit(lambda: qmg.lib().qmg_norm2sq(ctypes.c_void_p(x.ptr), ctypes.c_size_t(n), ctypes.c_void_p(dres.ptr), None, None)), C * n, 4 * n, cpu_time(lambda: ol.norm2sq(xs_b)), scale)
row("dot", "qlinalg dot (stateful_multigrid.h:904)", timeit(lambda: qmg.lib().qmg_dot(ctypes.c_void_p(x.ptr), ctypes.c_void_p(y.ptr), ctypes.c_size_t(n), ctypes.c_void_p(dres.ptr), None, None)), 2 * C * n, 8 * n, cpu_time(lambda: ol.dot(xs_b, ys_b)), scale)
row("caxpy", "qlinalg caxpy", timeit(lambda: qmg.caxpy(0.3 - 0.1j, x, y, n)), 3 * C * n, 8 * n)
row("caxpbyz", "qlinalg caxpbyz (stateful_multigrid.h:866)", timeit(lambda: qmg.caxpbyz(1.0, x, -1.0, y, y, n)), 3 * C * n, 8 * n)
vs = [gauss(n, 10 + i) for i in range(8)]
row("multidot, 8 vectors", "GCR orthogonalisation", timeit(lambda: qmg.multidot(vs, y, n)), (8 + 2) * C * n, 8 * 8 * n, note="y re-read once per 4 vectors")
row("multi_caxpy, 8 vectors", "GCR orthogonalisation", timeit(lambda: qmg.multi_caxpy([0.1] * 8, vs, y, n)), (8 + 2) * C * n, 8 * 8 * n)
for v in vs:
    v.free()
x.free(); y.free()

# ---------------- transfer and coarse operator at BASELINE configs[2] sizes: 2048^2 nc=2 -> 512^2 nc=24
fL, cL, cnc = 2048, 512, 24
fsize, csize = fL * fL * 2, cL * cL * cnc
nv = gauss(cnc * fsize, 5)
fv, cv = gauss(fsize, 6), gauss(csize, 7)
fd, cd = (fL, fL, 2), (cL, cL, cnc)
xfer_bytes = (cnc * fsize + 2 * fsize + csize) * C
# CPU sample: 256^2 -> 64^2 with the same 24 vectors
sfL, scL = 256, 64
snv = np.random.default_rng(3).standard_normal(cnc * sfL * sfL * 2) + 0j
sf = np.random.default_rng(4).standard_normal(sfL * sfL * 2) + 0j
sc = np.random.default_rng(5).standard_normal(scL * scL * cnc) + 0j
xscale = (fL / sfL) ** 2
row("prolong_c2f (24 null vectors)", "transfer.h:455-480", timeit(lambda: qmg.prolong(nv, cnc, cv, fv, fd, cd), reps=10), xfer_bytes, 8 * cnc * fsize,
    cpu_time(lambda: ol.prolong(snv, sc, (sfL, sfL, 2), (scL, scL, cnc))), xscale)
row("restrict_f2c (24 null vectors)", "transfer.h:487-511", timeit(lambda: qmg.restrict(nv, cnc, fv, cv, fd, cd), reps=10), xfer_bytes, 8 * cnc * fsize,
    cpu_time(lambda: ol.restrict(snv, sf, (sfL, sfL, 2), (scL, scL, cnc))), xscale)
nv.free(); fv.free()
ccm = cL * cL * cnc * cnc
ccl, cho = gauss(ccm, 8), gauss(4 * ccm, 9)
cdsc = qmg.make_desc(cL, cL, cnc, ccl, cho, -0.07)
cy = qmg.DeviceArray(csize)
# CPU sample 64^2
socl = np.random.default_rng(6).standard_normal(scL * scL * cnc * cnc) + 0j
soho = np.random.default_rng(7).standard_normal(4 * scL * scL * cnc * cnc) + 0j
sod = ol.make_desc(scL, scL, cnc, socl, soho, -0.07)
row("coarse apply_M nc=24, 512^2", "coarse.h + stencil_2d.h:912", timeit(lambda: qmg.stencil_apply(cdsc, cy, cv), reps=10), 46848 * cL * cL, 23232 * cL * cL,
    cpu_time(lambda: ol.stencil_apply(sod, sc)), (cL / scL) ** 2)

# ---------------- lock-step batches: the coarse apply on the f64 matrix cores, batched transfer
for k in (8, 16):
    xb, yb = gauss(k * csize, 20 + k), qmg.DeviceArray(k * csize)
    row("coarse apply_M nc=24, 512^2, %d rhs per launch (MFMA)" % k, "qmg_stencil_apply_batch (kernel C)",
        timeit(lambda: qmg.stencil_apply_batch(cdsc, yb, xb, qmg.P_ALL | qmg.P_ZERO, k, csize, (1 << k) - 1), reps=10), (5 * cnc * cnc + 2 * cnc * k) * 16 * cL * cL,
        (8 * cnc * cnc * 5 + 8 * cnc) * cL * cL * k, note="%.3f of the one-rhs kernel's bytes per right-hand side" % ((5 * cnc * cnc / k + 2 * cnc) / (5 * cnc * cnc + 2 * cnc)))
    xb.free(); yb.free()
nv = gauss(cnc * fsize, 5)
k = 8
fb, cb = gauss(k * fsize, 31), gauss(k * csize, 32)
row("prolong_c2f, 8 systems per pass", "qmg_prolong_batch", timeit(lambda: qmg.prolong_batch(nv, cnc, cb, fb, fd, cd, k, csize, fsize, 255), reps=5),
    (cnc * fsize + 2 * k * fsize + k * csize) * C, 8 * cnc * fsize * k)
row("restrict_f2c, 8 systems per pass", "qmg_restrict_batch", timeit(lambda: qmg.restrict_batch(nv, cnc, fb, cb, fd, cd, k, fsize, csize, 255), reps=5),
    (cnc * fsize + k * fsize + 2 * k * csize) * C, 8 * cnc * fsize * k)

print(json.dumps({"device": "MI355X (gfx950)", "hbm_peak_GBps": PEAK, "rows": rows}, indent=1))
