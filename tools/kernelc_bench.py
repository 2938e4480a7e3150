"""Kernels B / B32 (one system) and C (multi-rhs coarse apply on the matrix cores) at the K-cycle shapes, the storage combinations
(fp64 / fp32-stored matrices with fp64 vectors / fp32 / complex<half> matrices with fp64 or fp32 vectors), 1, 8 and 16 systems: ms and fraction of the 8 TB/s HBM peak on the
algorithmic bytes (5 nc^2 matrix elements + 2 nc k vector elements per site).   gpurun -- 'python tools/kernelc_bench.py'"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
qmg = importlib.import_module("quantum-mg_amd")
qmg.init(0)


def gauss(n, s, dtype):
    d = qmg.DeviceArray(n)
    qmg.gaussian(d, n, s)
    if dtype == qmg.C32:
        f = qmg.DeviceArray(n, np.complex64)
        qmg.convert(f, qmg.C32, d, qmg.C64, n)
        d.free()
        return f
    return d


t = qmg.Timer()
shapes = ((1024, 8), (512, 24), (256, 24), (512, 12), (512, 16))
if len(sys.argv) > 1:   # "python tools/kernelc_bench.py 24 8": only these nc (PMC runs)
    shapes = tuple(sh for sh in shapes if str(sh[1]) in sys.argv[1:] and sh[0] >= 512)
for L, nc in shapes:
    vol = L * L
    for tag, mdt, vdt in (("fp64", qmg.C64, qmg.C64), ("mat32", qmg.C32, qmg.C64), ("fp32", qmg.C32, qmg.C32), ("mat16", 16, qmg.C64), ("m16v32", 16, qmg.C32)):
        if mdt == 16:
            c64, h64 = gauss(vol * nc * nc, 1, qmg.C64), gauss(4 * vol * nc * nc, 2, qmg.C64)
            cl, ho = qmg.DeviceArray(vol * nc * nc, np.float32), qmg.DeviceArray(4 * vol * nc * nc, np.float32)   # 4 bytes per complex<half>
            qmg.convert_to_c16(cl, c64, qmg.C64, vol * nc * nc)
            qmg.convert_to_c16(ho, h64, qmg.C64, 4 * vol * nc * nc)
            c64.free(); h64.free()
        else:
            cl, ho = gauss(vol * nc * nc, 1, mdt), gauss(4 * vol * nc * nc, 2, mdt)
        d = qmg.make_desc(L, L, nc, cl, ho, 0.1)
        for k in (1, 8, 16):
            x, y = gauss(k * vol * nc, 3, vdt), gauss(k * vol * nc, 4, vdt)
            if mdt == 16:
                fn = lambda: qmg.stencil_apply_mat16(vdt, d, y, x, qmg.P_ALL | qmg.P_ZERO, nrhs=k, vec_stride=vol * nc, mask=(1 << k) - 1)
            elif tag == "mat32":
                fn = lambda: qmg.stencil_apply_mat32(d, y, x, qmg.P_ALL | qmg.P_ZERO, nrhs=k, vec_stride=vol * nc, mask=(1 << k) - 1)
            else:
                fn = lambda: qmg.stencil_apply_t(vdt, d, y, x, qmg.P_ALL | qmg.P_ZERO, nrhs=k, vec_stride=vol * nc, mask=(1 << k) - 1)
            for _ in range(3):
                fn()
            qmg.sync()
            t.start()
            for _ in range(10):
                fn()
            ms = t.stop_ms() / 10
            msz, vsz = (4 if mdt == 16 else 16 if mdt == qmg.C64 else 8), (16 if vdt == qmg.C64 else 8)
            b = (5 * nc * nc * msz + 2 * nc * k * vsz) * vol
            print("nc=%d %d^2 %-6s k=%-2d  %.3f ms  %.0f GB/s  %.2f of peak" % (nc, L, tag, k, ms, b / ms / 1e6, b / ms / 1e6 / 8000.0), flush=True)
            x.free(); y.free()
        cl.free(); ho.free()
