"""Kernel W (Wilson straight from the links, csrc/qmg_wilson.hip) against the stored-stencil kernels at 4096^2 and 2048^2."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
qmg = importlib.import_module("quantum-mg_amd"); qmg.init(0)
t = qmg.Timer()
def run(fn, nbytes, label):
    for _ in range(5): fn()
    qmg.sync(); t.start()
    for _ in range(50): fn()
    ms = t.stop_ms() / 50
    print("%-58s %.4f ms %6.0f GB/s  %.3f of 8 TB/s" % (label, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000), flush=True)
FULL, DEO = qmg.P_ALL | qmg.P_ZERO, qmg.P_EO | qmg.P_ZERO_E
for L in (4096, 2048):
    vol = L * L
    rng = np.random.default_rng(1)
    g = qmg.DeviceArray.from_host(np.exp(1j * rng.uniform(-np.pi, np.pi, size=2 * vol)))
    g32 = qmg.DeviceArray(2 * vol, np.complex64); qmg.convert(g32, qmg.C32, g, qmg.C64, 2 * vol)
    cl, hp = qmg.DeviceArray(4 * vol), qmg.DeviceArray(16 * vol)
    qmg.wilson_fill(cl, hp, g, L, L, 1.0)
    d = qmg.make_desc(L, L, 2, cl, hp, -0.07)
    r, l = qmg.DeviceArray(2 * vol), qmg.DeviceArray(2 * vol); qmg.gaussian(r, 2 * vol, 5)
    run(lambda: qmg.stencil_apply(d, l, r, FULL), 384 * vol, "L=%d fp64 stored stencil (kernel A), 384 B/site" % L)
    for pair in (2, 1, 0):
        qmg.set_tuning("wilson_pair", pair)
        run(lambda: qmg.wilson_apply_direct(qmg.C64, d, g, l, r, FULL), 96 * vol, "L=%d fp64 from the links (kernel W%s), 96 B/site" % (L, ("", "2: paired parities", "2: paired parities x 2 rows")[pair]))
    qmg.set_tuning("wilson_pair", 2)
    run(lambda: qmg.wilson_apply_direct(qmg.C64, d, g, l, r, DEO), 128 * vol / 2, "L=%d fp64 from the links, D_eo, 128 B per written site" % L)
    # the right-block-Jacobi hops (the Schur complement's D'_eo): stored (kernel S, 320 B per written site) against links x cinv
    cinv, rcl, rhp = qmg.DeviceArray(4 * vol), qmg.DeviceArray(4 * vol), qmg.DeviceArray(16 * vol)
    qmg.build_rbjacobi(cinv, rcl, rhp, d)
    scale = cinv.to_host()[0].real
    del cinv, rcl
    drb = qmg.make_desc(L, L, 2, None, rhp)
    run(lambda: qmg.stencil_apply(drb, l, r, DEO), 320 * vol / 2, "L=%d fp64 rbj D'_eo stored (kernel S), 320 B per written site" % L)
    run(lambda: qmg.wilson_hops_direct(qmg.C64, drb, g, l, r, DEO, 1.0, scale), 128 * vol / 2, "L=%d fp64 rbj D'_eo from the links, 128 B per written site" % L)
    r32, l32 = qmg.DeviceArray(2 * vol, np.complex64), qmg.DeviceArray(2 * vol, np.complex64)
    qmg.convert(r32, qmg.C32, r, qmg.C64, 2 * vol)
    h32 = qmg.DeviceArray(16 * vol, np.complex64); qmg.convert(h32, qmg.C32, rhp, qmg.C64, 16 * vol)
    d32 = qmg.make_desc(L, L, 2, None, h32)
    run(lambda: qmg.stencil_apply_t(qmg.C32, d32, l32, r32, DEO), 160 * vol / 2, "L=%d fp32 rbj D'_eo stored (kernel S), 160 B per written site" % L)
    h16 = qmg.DeviceArray(16 * vol, np.float32); qmg.convert_to_c16(h16, rhp, qmg.C64, 16 * vol)
    d16 = qmg.make_desc(L, L, 2, None, h16)
    run(lambda: qmg.stencil_apply_h16(d16, l32, r32, DEO), 96 * vol / 2, "L=%d 16-bit rbj D'_eo stored (kernel S), 96 B per written site" % L)
    run(lambda: qmg.wilson_hops_direct(qmg.C32, drb, g32, l32, r32, DEO, 1.0, scale), 64 * vol / 2, "L=%d fp32 rbj D'_eo from the links, 64 B per written site" % L)
    del cl, hp, rhp, h32, h16, r32, l32
    r32, l32 = qmg.DeviceArray(2 * vol, np.complex64), qmg.DeviceArray(2 * vol, np.complex64)
    qmg.convert(r32, qmg.C32, r, qmg.C64, 2 * vol)
    for pair in (2, 1, 0):
        qmg.set_tuning("wilson_pair", pair)
        run(lambda: qmg.wilson_apply_direct(qmg.C32, d, g32, l32, r32, FULL), 48 * vol, "L=%d fp32 from the links%s, 48 B/site" % (L, ("", " (paired)", " (paired x 2 rows)")[pair]))
    qmg.set_tuning("wilson_pair", 2)
    K = 8 if L == 2048 else 4
    rb, lb = qmg.DeviceArray(2 * vol * K), qmg.DeviceArray(2 * vol * K); qmg.gaussian(rb, 2 * vol * K, 6)
    run(lambda: qmg.wilson_apply_direct(qmg.C64, d, g, lb, rb, FULL, 1.0, K, 2 * vol, (1 << K) - 1), (32 + 64 * K) * vol, "L=%d fp64 from the links, %d systems" % (L, K))
    del g, g32, r, l, r32, l32, rb, lb
