"""A/B of the nc = 2 site kernel (csrc/qmg_site.hip) against kernel A at 4096^2 in the three storage precisions.
   python tools/site_kernel_ab.py     (GPU box)"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
qmg = importlib.import_module("quantum-mg_amd"); qmg.init(0)
L = 4096; vol = L * L
fixture = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/l64t64b60_heatbath.dat')
wl = bench.Workload(qmg, L, fixture, 1337)
t = qmg.Timer()

def run(fn, nbytes, label):
    for _ in range(5): fn()
    qmg.sync(); t.start()
    for _ in range(50): fn()
    ms = t.stop_ms() / 50
    print("%-44s %.4f ms %6.0f GB/s  %.3f of 8 TB/s" % (label, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000), flush=True)

FULL, DEO = qmg.P_ALL | qmg.P_ZERO, qmg.P_EO | qmg.P_ZERO_E
# fp64
d64 = qmg.make_desc(L, L, 2, wl.clover, wl.hopping, -0.07)
r = qmg.DeviceArray(2 * vol); l = qmg.DeviceArray(2 * vol)
qmg.gaussian(r, 2 * vol, 5)
for site in (0, 1):
    qmg.set_tuning("stencil_site", 7 if site else 0)
    run(lambda: qmg.stencil_apply(d64, l, r, FULL), 384 * vol, "fp64 full  site=%d" % site)
    run(lambda: qmg.stencil_apply(d64, l, r, DEO), (256 + 64) * vol / 2, "fp64 D_eo  site=%d" % site)
# fp32
c32, h32 = qmg.DeviceArray(4 * vol, np.complex64), qmg.DeviceArray(16 * vol, np.complex64)
qmg.convert(c32, qmg.C32, wl.clover, qmg.C64, 4 * vol); qmg.convert(h32, qmg.C32, wl.hopping, qmg.C64, 16 * vol)
c16, h16 = qmg.DeviceArray(4 * vol, np.float32), qmg.DeviceArray(16 * vol, np.float32)
qmg.convert_to_c16(c16, wl.clover, qmg.C64, 4 * vol); qmg.convert_to_c16(h16, wl.hopping, qmg.C64, 16 * vol)
wl.free(); del r, l
r = qmg.DeviceArray(2 * vol, np.complex64); l = qmg.DeviceArray(2 * vol, np.complex64)
d32 = qmg.make_desc(L, L, 2, c32, h32, -0.07)
for site in (0, 1):
    qmg.set_tuning("stencil_site", 7 if site else 0)
    run(lambda: qmg.stencil_apply_t(qmg.C32, d32, l, r, FULL), 192 * vol, "fp32 full  site=%d" % site)
    run(lambda: qmg.stencil_apply_t(qmg.C32, d32, l, r, DEO), (128 + 32) * vol / 2, "fp32 D_eo  site=%d" % site)
d16 = qmg.make_desc(L, L, 2, c16, h16, -0.07)
for gen in (0, 1):
    qmg.set_tuning("site_generic", gen)
    run(lambda: qmg.stencil_apply_h16(d16, l, r, FULL), 112 * vol, "16-bit matrices full  generic=%d" % gen)
    run(lambda: qmg.stencil_apply_h16(d16, l, r, DEO), (64 + 32) * vol / 2, "16-bit matrices D_eo  generic=%d" % gen)

# batches of 8 right-hand sides at 2048^2: matrices once per site (320 + 64 n B/site in fp64)
del r, l, c32, h32, c16, h16
L = 2048; vol = L * L; K = 8
cl = qmg.DeviceArray(4 * vol); hp = qmg.DeviceArray(16 * vol)
qmg.gaussian(cl, 4 * vol, 1); qmg.gaussian(hp, 16 * vol, 2)
r = qmg.DeviceArray(2 * vol * K); l = qmg.DeviceArray(2 * vol * K); qmg.gaussian(r, 2 * vol * K, 3)
d = qmg.make_desc(L, L, 2, cl, hp, -0.07)
qmg.set_tuning("stencil_site", 0)
run(lambda: qmg.stencil_apply(d, l, r, FULL, K, 2 * vol), (320 + 64 * K) * vol, "fp64 full 2048^2 x8  kernel A")
qmg.set_tuning("stencil_site", 7)
run(lambda: qmg.stencil_apply(d, l, r, FULL, K, 2 * vol), (320 + 64 * K) * vol, "fp64 full 2048^2 x8  site")
c32, h32 = qmg.DeviceArray(4 * vol, np.complex64), qmg.DeviceArray(16 * vol, np.complex64)
qmg.convert(c32, qmg.C32, cl, qmg.C64, 4 * vol); qmg.convert(h32, qmg.C32, hp, qmg.C64, 16 * vol)
c16, h16 = qmg.DeviceArray(4 * vol, np.float32), qmg.DeviceArray(16 * vol, np.float32)
qmg.convert_to_c16(c16, cl, qmg.C64, 4 * vol); qmg.convert_to_c16(h16, hp, qmg.C64, 16 * vol)
del r, l
r = qmg.DeviceArray(2 * vol * K, np.complex64); l = qmg.DeviceArray(2 * vol * K, np.complex64)
d32 = qmg.make_desc(L, L, 2, c32, h32, -0.07); d16 = qmg.make_desc(L, L, 2, c16, h16, -0.07)
qmg.set_tuning("stencil_site", 0)
run(lambda: qmg.stencil_apply_t(qmg.C32, d32, l, r, FULL, K, 2 * vol, 0xFF), (160 + 32 * K) * vol, "fp32 full 2048^2 x8  kernel A")
qmg.set_tuning("stencil_site", 7)
run(lambda: qmg.stencil_apply_t(qmg.C32, d32, l, r, FULL, K, 2 * vol, 0xFF), (160 + 32 * K) * vol, "fp32 full 2048^2 x8  site")
run(lambda: qmg.stencil_apply_h16(d16, l, r, FULL, K, 2 * vol, 0xFF), (80 + 32 * K) * vol, "16-bit matrices full 2048^2 x8")
