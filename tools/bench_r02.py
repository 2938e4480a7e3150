"""Round-2 kernel table: the fp32 instantiation next to its fp64 twin, and the LDS-tiled batched transfer next to the
one-system kernels.  HIP-event timing on the launch stream, achieved GB/s from the ALGORITHMIC byte counts of SURVEY 8(d).

    python tools/bench_r02.py > profiles/r02_kernel_rooflines.json
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

qmg = importlib.import_module("quantum-mg_amd")
qmg.init(0)
PEAK = 8000.0
fixture = os.path.join(ROOT, "tests", "golden", "l64t64b60_heatbath.dat")
rows = []
only = set(sys.argv[1:])


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    qmg.sync()
    t = qmg.Timer()
    t.start()
    for _ in range(reps):
        fn()
    return t.stop_ms() / reps


def row(name, ms, alg_bytes, note=""):
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    rows.append({"kernel": name, "ms": round(ms, 4), "algorithmic_GB": round(alg_bytes / 1e9, 4), "achieved_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / PEAK, 4), "note": note})
    print("# %-58s %8.3f ms %8.0f GB/s  %.3f" % (name, ms, gbs, gbs / PEAK), file=sys.stderr, flush=True)


def gauss(n, seed, dtype=qmg.C64):
    d = qmg.DeviceArray(n)
    qmg.gaussian(d, n, seed)
    if dtype == qmg.C64:
        return d
    f = qmg.DeviceArray(n, np.complex64)
    qmg.convert(f, qmg.C32, d, qmg.C64, n)
    qmg.sync()
    d.free()
    return f


def want(tag):
    return not only or tag in only


# ---------------- fine Wilson apply, fp64 and fp32 ----------------
if want("fine"):
    for L in (4096, 2048):
        vol = L * L
        wl = bench.Workload(qmg, L, fixture, 1337)
        ms = timeit(wl.step, reps=50, warm=10)
        row("k_stencil_pair<double,2,2> Wilson %d^2 fp64" % L, ms, 384 * vol, "headline kernel: the reference's algorithm, a general stored stencil")
        c32, h32 = qmg.DeviceArray(4 * vol, np.complex64), qmg.DeviceArray(16 * vol, np.complex64)
        qmg.convert(c32, qmg.C32, wl.clover, qmg.C64, 4 * vol)
        qmg.convert(h32, qmg.C32, wl.hopping, qmg.C64, 16 * vol)
        r32, l32 = gauss(2 * vol, 7, qmg.C32), qmg.DeviceArray(2 * vol, np.complex64)
        d32 = qmg.make_desc(L, L, 2, c32, h32, bench.MASS)
        ms = timeit(lambda: qmg.stencil_apply_t(qmg.C32, d32, l32, r32, qmg.P_ALL | qmg.P_ZERO), reps=50, warm=10)
        row("k_stencil_site<1,...> (kernel S) Wilson %d^2 fp32" % L, ms, 192 * vol, "192 B/site: fp32 matrices AND vectors, fp32 arithmetic")
        qmg.set_tuning("stencil_site", 0)
        ms = timeit(lambda: qmg.stencil_apply_t(qmg.C32, d32, l32, r32, qmg.P_ALL | qmg.P_ZERO), reps=50, warm=10)
        row("k_stencil_elem<float,2> (kernel A in fp32, stencil_site=0) Wilson %d^2 fp32" % L, ms, 192 * vol, "the kernel S replaced")
        qmg.set_tuning("stencil_site", 3)
        # straight from the links (kernel W): no stored stencil
        rng = np.random.default_rng(1)
        g64 = qmg.DeviceArray.from_host(np.exp(1j * rng.uniform(-np.pi, np.pi, size=2 * vol)))
        g32 = qmg.DeviceArray(2 * vol, np.complex64)
        qmg.convert(g32, qmg.C32, g64, qmg.C64, 2 * vol)
        dW = qmg.make_desc(L, L, 2, None, None, bench.MASS)
        ms = timeit(lambda: qmg.wilson_apply_direct(qmg.C64, dW, g64, wl.lhs, wl.rhs, qmg.P_ALL | qmg.P_ZERO), reps=50, warm=10)
        row("k_wilson_direct<double> (kernel W) Wilson %d^2 fp64, from the links" % L, ms, 96 * vol, "96 B/site: links 32 + rhs 32 + lhs 32; bit-identical to the stored stencil")
        ms = timeit(lambda: qmg.wilson_apply_direct(qmg.C64, dW, g64, wl.lhs, wl.rhs, qmg.P_EO | qmg.P_ZERO_E), reps=50, warm=10)
        row("k_wilson_direct<double> D_eo %d^2 fp64" % L, ms, 128 * vol / 2, "one parity, per written site: its four links 64 (every link of the lattice is read once) + rhs 32 + lhs 32")
        ms = timeit(lambda: qmg.wilson_apply_direct(qmg.C32, dW, g32, l32, r32, qmg.P_ALL | qmg.P_ZERO), reps=50, warm=10)
        row("k_wilson_direct<float> (kernel W) Wilson %d^2 fp32, from the links" % L, ms, 48 * vol, "48 B/site")
        g64.free(); g32.free()
        c16, h16 = qmg.DeviceArray(4 * vol, np.float32), qmg.DeviceArray(16 * vol, np.float32)
        qmg.convert_to_c16(c16, wl.clover, qmg.C64, 4 * vol)
        qmg.convert_to_c16(h16, wl.hopping, qmg.C64, 16 * vol)
        d16 = qmg.make_desc(L, L, 2, c16, h16, bench.MASS)
        ms = timeit(lambda: qmg.stencil_apply_h16(d16, l32, r32, qmg.P_ALL | qmg.P_ZERO), reps=50, warm=10)
        row("k_stencil_site<0,...> (kernel S) Wilson %d^2, complex<half> matrices + complex<float> vectors" % L, ms, 112 * vol, "112 B/site: 16-bit stored operator (preconditioner only)")
        ms = timeit(lambda: qmg.stencil_apply_h16(d16, l32, r32, qmg.P_EO | qmg.P_ZERO_E), reps=50, warm=10)
        row("k_stencil_site<0,2,...> D_eo %d^2, 16-bit matrices" % L, ms, (4 * 16 + 16 + 16) * vol / 2, "one parity")
        c16.free(); h16.free()
        # one-parity (Schur) applies
        ms = timeit(lambda: qmg.stencil_apply_t(qmg.C32, d32, l32, r32, qmg.P_EO | qmg.P_ZERO_E), reps=50, warm=10)
        row("k_stencil_site<1,2,...> D_eo %d^2 fp32" % L, ms, (4 * 32 + 16 + 16) * vol / 2, "one parity: 4 hopping matrices + rhs + lhs")
        for a in (c32, h32, r32, l32):
            a.free()
        wl.free()

if want("fine_variants"):
    L = 4096
    vol = L * L
    wl = bench.Workload(qmg, L, fixture, 1337)
    c32, h32 = qmg.DeviceArray(4 * vol, np.complex64), qmg.DeviceArray(16 * vol, np.complex64)
    qmg.convert(c32, qmg.C32, wl.clover, qmg.C64, 4 * vol)
    qmg.convert(h32, qmg.C32, wl.hopping, qmg.C64, 16 * vol)
    wl.free()
    r32, l32 = gauss(2 * vol, 7, qmg.C32), qmg.DeviceArray(2 * vol, np.complex64)
    d32 = qmg.make_desc(L, L, 2, c32, h32, bench.MASS)
    for pair in (2, 1, 0):
        for rcap in (0, 512, 128):
            qmg.set_tuning("stencil_pair", pair)
            qmg.set_tuning("stencil_rows", rcap)
            ms = timeit(lambda: qmg.stencil_apply_t(qmg.C32, d32, l32, r32, qmg.P_ALL | qmg.P_ZERO), reps=50, warm=10)
            row("fp32 Wilson 4096^2 stencil_pair=%d stencil_rows=%d" % (pair, rcap), ms, 192 * vol)
    qmg.set_tuning("stencil_pair", 2)
    qmg.set_tuning("stencil_rows", 0)

# ---------------- coarse apply nc = 8 (C5 level 1: 1024^2) and nc = 24 (C3 level 1: 512^2), fp64 vs fp32 ----------------
if want("coarse"):
    for L, nc in ((1024, 8), (512, 24)):
        vol = L * L
        for dt, name, esz in ((qmg.C64, "fp64", 16), (qmg.C32, "fp32", 8)):
            cl, ho = gauss(vol * nc * nc, 1, dt), gauss(4 * vol * nc * nc, 2, dt)
            d = qmg.make_desc(L, L, nc, cl, ho, 0.1)
            for k in (1, 8):
                x, y = gauss(k * vol * nc, 3, dt), gauss(k * vol * nc, 4, dt)
                ms = timeit(lambda: qmg.stencil_apply_t(dt, d, y, x, qmg.P_ALL | qmg.P_ZERO, nrhs=k, vec_stride=vol * nc, mask=(1 << k) - 1), reps=10, warm=2)
                row("coarse apply nc=%d %d^2 %s, %d rhs" % (nc, L, name, k), ms, (5 * nc * nc + 2 * nc * k) * esz * vol, "kernel B/B32 (1 rhs) or kernel C (8 rhs)")
                x.free(); y.free()
            cl.free(); ho.free()

# ---------------- transfer: one-system kernels and the LDS-tiled batch kernels ----------------
if want("xfer") or "xfer0" in only:
    for (fL, fnc, cL, cnc) in (((2048, 2, 512, 24),) if "xfer0" in only else ((2048, 2, 512, 24), (4096, 2, 1024, 8), (512, 24, 128, 24))):
        fsize, csize = fL * fL * fnc, cL * cL * cnc
        fd, cd = (fL, fL, fnc), (cL, cL, cnc)
        for dt, name, esz in ((qmg.C64, "fp64", 16), (qmg.C32, "fp32", 8)):
            nv = gauss(cnc * fsize, 5, dt)
            for k in (1, 4, 8):
                fb, cb = gauss(k * fsize, 31, dt), gauss(k * csize, 32, dt)
                b = (cnc * fsize + 2 * k * fsize + k * csize) * esz
                for tile in ((1,) if k == 1 else ((1, 2) if "xfer0" in only else (1, 0))):
                    qmg.set_tuning("xfer_tile", tile)
                    tag = "" if k == 1 else (" tiled" if tile == 1 else " tiled(general restrict)" if tile == 2 else " system-by-system")
                    ms = timeit(lambda: qmg.prolong_batch_t(dt, nv, cnc, cb, fb, fd, cd, k, csize, fsize, (1 << k) - 1), reps=8, warm=2)
                    row("prolong %d^2x%d -> %d^2x%d %s k=%d%s" % (cL, cnc, fL, fnc, name, k, tag), ms, b, "(nvec + 2k) size_cv_f + k size_cv_c")
                    ms = timeit(lambda: qmg.restrict_batch_t(dt, nv, cnc, fb, cb, fd, cd, k, fsize, csize, (1 << k) - 1), reps=8, warm=2)
                    row("restrict %d^2x%d -> %d^2x%d %s k=%d%s" % (fL, fnc, cL, cnc, name, k, tag), ms, b, "")
                qmg.set_tuning("xfer_tile", 1)
                fb.free(); cb.free()
            nv.free()

# ---------------- BLAS-1 / reductions fp32 vs fp64 ----------------
if want("blas"):
    n = 2 * 4096 * 4096
    for dt, name, esz in ((qmg.C64, "fp64", 16), (qmg.C32, "fp32", 8)):
        x, y = gauss(n, 1, dt), gauss(n, 2, dt)
        ms = timeit(lambda: qmg.batch_blas_t(dt, qmg.BOP_CAXPY, y, n, 1, n, 1, a=[0.5 + 0.1j], x=x), reps=20)
        row("caxpy %s, 2x4096^2 elements" % name, ms, 3 * n * esz)
        ms = timeit(lambda: qmg.batch_reduce_t(dt, qmg.BRED_DOT, x, y, n, 1, n, 1), reps=20)
        row("dot %s (fp64 accumulation), 2x4096^2 elements" % name, ms, 2 * n * esz)
        ms = timeit(lambda: qmg.batch_reduce_t(dt, qmg.BRED_NORM2, x, None, n, 1, n, 1), reps=20)
        row("norm2sq %s, 2x4096^2 elements" % name, ms, n * esz)
        x.free(); y.free()

print(json.dumps({"device": "MI355X (gfx950)", "hbm_peak_GBps": PEAK, "rows": rows}, indent=1))
