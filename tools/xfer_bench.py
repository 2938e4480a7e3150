"""Batched restrict / prolong at the shapes the K-cycle configurations use, both storage precisions, matrix-core kernels on and off
(tuning key "xfer_mfma"): ms per call and the fraction of the 8 TB/s HBM peak on the algorithmic bytes
(nvec + 2 k) size_cv_f + k size_cv_c elements.   gpurun -- 'python tools/xfer_bench.py > gpurun_out/xfer.txt'"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
qmg = importlib.import_module("quantum-mg_amd")
qmg.init(0)
PEAK = 8000.0


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
for (fL, fnc, cL, cnc) in ((2048, 2, 512, 24), (512, 24, 128, 24), (4096, 2, 1024, 8), (1024, 8, 256, 8)):
    fsize, csize = fL * fL * fnc, cL * cL * cnc
    fd, cd = (fL, fL, fnc), (cL, cL, cnc)
    for dtype, name, esz in ((qmg.C64, "fp64", 16), (qmg.C32, "fp32", 8)):
        nv = gauss(cnc * fsize, 5, dtype)
        for k in (4, 8):
            fb, cb = gauss(k * fsize, 31, dtype), gauss(k * csize, 32, dtype)
            for mfma in (1, 2, 0):
                qmg.set_tuning("xfer_mfma", mfma)
                for op, fn in (("prolong", lambda: qmg.prolong_batch_t(dtype, nv, cnc, cb, fb, fd, cd, k, csize, fsize, (1 << k) - 1)),
                               ("restrict", lambda: qmg.restrict_batch_t(dtype, nv, cnc, fb, cb, fd, cd, k, fsize, csize, (1 << k) - 1))):
                    # fresh operands for every measurement: the calls ACCUMULATE (fine += P coarse, coarse += P^dag fine), and a dozen rounds of that
                    # overflow complex<float>
                    fb.free(); cb.free()
                    fb, cb = gauss(k * fsize, 31, dtype), gauss(k * csize, 32, dtype)
                    for _ in range(2):
                        fn()
                    qmg.sync()
                    t.start()
                    for _ in range(5):
                        fn()
                    ms = t.stop_ms() / 5
                    b = (cnc * fsize + 2 * k * fsize + k * csize) * esz
                    print("%dx%dx%d -> %dx%dx%d %s k=%d %-8s mfma=%d  %.3f ms  %.0f GB/s  %.2f of peak" % (fL, fL, fnc, cL, cL, cnc, name, k, op, mfma, ms, b / ms / 1e6, b / ms / 1e6 / PEAK), flush=True)
            qmg.set_tuning("xfer_mfma", 1)
            fb.free()
            cb.free()
        nv.free()
