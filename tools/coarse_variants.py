"""A/B timing of the generic-nc (coarse) stencil kernel's tuning knobs at the K-cycle's coarse sizes, interleaved rounds
in one process.  usage: coarse_variants.py L nc '[{"gen_sites":0},{"gen_sites":2}]'"""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (before libqmg_hip)
qmg = importlib.import_module("quantum-mg_amd")
qmg.init(0)
qmg.set_tuning("stencil_nt", 3)
L = int(sys.argv[1]); nc = int(sys.argv[2])
variants = json.loads(sys.argv[3])
nrhs = int(sys.argv[4]) if len(sys.argv) > 4 else 1
vol = L * L
cl = qmg.DeviceArray(vol * nc * nc); qmg.gaussian(cl, vol * nc * nc, 1)
ho = qmg.DeviceArray(4 * vol * nc * nc); qmg.gaussian(ho, 4 * vol * nc * nc, 2)
x = qmg.DeviceArray(vol * nc * nrhs); qmg.gaussian(x, vol * nc * nrhs, 3)
y = qmg.DeviceArray(vol * nc * nrhs)
d = qmg.make_desc(L, L, nc, cl, ho, -0.07)
cl32, ho32 = qmg.DeviceArray(vol * nc * nc // 2 + 1), qmg.DeviceArray(4 * vol * nc * nc // 2 + 1)
qmg.c64_to_c32(cl32, cl, vol * nc * nc); qmg.c64_to_c32(ho32, ho, 4 * vol * nc * nc)
d32 = qmg.make_desc(L, L, nc, cl32, ho32, -0.07)


def apply(v):
    if v.get("mat32"):
        qmg.stencil_apply_mat32(d32, y, x, qmg.P_ALL | qmg.P_ZERO, nrhs, vol * nc, (1 << nrhs) - 1)
    else:
        qmg.stencil_apply(d, y, x, qmg.P_ALL | qmg.P_ZERO, nrhs=nrhs, vec_stride=vol * nc)

alg = (5 * nc * nc + 2 * nc * nrhs) * 16 * vol
flops = (8 * nc * nc * 5 + 8 * nc) * vol * nrhs
timer = qmg.Timer()
res = {i: [] for i in range(len(variants))}
ref = None
for rnd in range(6):
    for i, v in enumerate(variants):
        for k, val in v.items():
            if k != "mat32": qmg.set_tuning(k, val)
        for _ in range(3): apply(v)
        qmg.sync(); timer.start()
        for _ in range(20): apply(v)
        res[i].append(timer.stop_ms() / 20)
        if rnd == 0 and not v.get('stencil_ablate') and not v.get('mat32'):
            h = y.to_host()
            if ref is None: ref = h
            else: assert np.linalg.norm(h - ref) <= 1e-13 * np.linalg.norm(ref), "variant changed the result"
for i, v in enumerate(variants):
    t = np.array(res[i])
    print("%-40s nrhs %2d median %.4f ms min %.4f ms -> %.0f GB/s algorithmic, %.2f TFLOP/s, %.4f ms per rhs" % (json.dumps(v), nrhs, np.median(t), t.min(), alg / np.median(t) / 1e6, flops / np.median(t) / 1e9, np.median(t) / nrhs))
