"""A/B of the staggered step's pieces at 4096^2, 8 right-hand sides: the plain apply, the 8 norms, the apply with fused norms.
   python tools/apply_norm_ab.py     (GPU box)"""
import importlib, sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
qmg = importlib.import_module("quantum-mg_amd"); qmg.init(0)
t = qmg.Timer()

def run(fn, nbytes, label):
    for _ in range(5): fn()
    qmg.sync(); t.start()
    for _ in range(50): fn()
    ms = t.stop_ms() / 50
    print("%-52s %.4f ms %6.0f GB/s  %.3f of 8 TB/s" % (label, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000), flush=True)

for nc, L, K in ((1, 4096, 8), (1, 4096, 16), (1, 4096, 2), (2, 2048, 8), (2, 2048, 3)):
    vol = L * L; size = vol * nc
    hop = qmg.DeviceArray(4 * vol * nc * nc); qmg.gaussian(hop, 4 * vol * nc * nc, 2)
    cl = None
    if nc == 2:
        cl = qmg.DeviceArray(vol * nc * nc); qmg.gaussian(cl, vol * nc * nc, 1)
    d = qmg.make_desc(L, L, nc, cl, hop, 0.04)
    r = qmg.DeviceArray(size * K); l = qmg.DeviceArray(size * K); qmg.gaussian(r, size * K, 3)
    nd = qmg.DeviceArray(16)
    FULL = qmg.P_ALL | qmg.P_ZERO
    mat = (4 + (1 if nc == 2 else 0)) * 16 * nc * nc
    vec = 32 * nc
    def norms():
        for k in range(K):
            qmg.check(qmg.lib().qmg_norm2sq(C.c_void_p(l.offset(k * size)), C.c_size_t(size), C.c_void_p(nd.ptr + 8 * k), None, None))
    tag = "nc=%d %d^2 x%d" % (nc, L, K)
    for pf in (0, 1):
        qmg.set_tuning("pair_prefetch", pf)
        run(lambda: qmg.stencil_apply(d, l, r, FULL, K, size), (mat + vec * K) * vol, tag + " apply  prefetch=%d" % pf)
        run(lambda: qmg.stencil_apply_norm2(d, l, r, FULL, K, size, norms_dev=nd.ptr), (mat + vec * K) * vol, tag + " apply + fused norms  prefetch=%d" % pf)
    run(norms, 16 * nc * K * vol, tag + " %d x norm2sq" % K)
    for a in (hop, r, l, nd): a.free()
    if cl is not None: cl.free()
