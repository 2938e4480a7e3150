"""Cost of each stencil piece (diagnostic): which stream makes the full apply slower than its byte share?"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
qmg = importlib.import_module("quantum-mg_amd")
qmg.init(0)
L = 4096
wl = bench.Workload(qmg, L, os.path.join(ROOT, "tests", "golden", "l64t64b60_heatbath.dat"), 1337)
P = qmg
cases = [
    ("EO|ZERO_E (4 mats, 1 parity)", P.P_EO | P.P_ZERO_E, 0.5 * (256 + 64)),
    ("EO|CLOVER_E|ZERO_E", P.P_EO | P.P_CLOVER_E | P.P_ZERO_E, 0.5 * (320 + 64)),
    ("EO|CLOVER_E|SHIFT_E|ZERO_E", P.P_EO | P.P_CLOVER_E | P.P_SHIFT_E | P.P_ZERO_E, 0.5 * (320 + 64)),
    ("CLOVER_E|ZERO_E", P.P_CLOVER_E | P.P_ZERO_E, 0.5 * (64 + 64)),
    ("HOPPING|ZERO (both)", P.P_HOPPING | P.P_ZERO, 256 + 64),
    ("HOPPING|SHIFT|ZERO (both)", P.P_HOPPING | P.P_SHIFT | P.P_ZERO, 256 + 64),
    ("ALL|ZERO (both)", P.P_ALL | P.P_ZERO, 384),
    ("ALL accumulate (both)", P.P_ALL, 384 + 32),
]
t = qmg.Timer()
for pair in (2, 0):
    qmg.set_tuning("stencil_pair", pair)
    for name, pieces, bps in cases:
        for _ in range(3): qmg.stencil_apply(wl.desc, wl.lhs, wl.rhs, pieces)
        qmg.sync(); t.start()
        for _ in range(20): qmg.stencil_apply(wl.desc, wl.lhs, wl.rhs, pieces)
        ms = t.stop_ms() / 20
        print("pair=%d %-32s %.3f ms  %.0f GB/s" % (pair, name, ms, bps * L * L / ms / 1e6))
