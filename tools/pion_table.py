"""The would-be pion masses of the reference's critical_mass.txt tables (n15 Wilson, n20 staggered; 32^2, beta = 6.0) through the
counterpart drivers: fitted m_pi and its statistical error per mass.   gpurun -- 'python tools/pion_table.py [nconf] > gpurun_out/pion.txt'"""
import os
import re
import subprocess
import sys

import numpy as np
from scipy.optimize import curve_fit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVERS = os.path.join(ROOT, "quantum-mg_amd", "drivers")
nconf = sys.argv[1] if len(sys.argv) > 1 else "400"
TABLE = {"n15_wilson_goldstone_u1_heatbath": [(0.01, 0.28205, 0.00047), (-0.01, 0.23957, 0.00053), (-0.03, 0.19324, 0.00062), (-0.05, 0.14087, 0.00081), (-0.06, 0.1076, 0.0012)],
         "n20_staggered_goldstone_u1_heatbath": [(0.1, 0.355891, 0.0004116), (0.08, 0.308843, 0.0004178), (0.06, 0.258516, 0.0004829), (0.04, 0.202947, 0.0005526)]}
for drv, rows in TABLE.items():
    for mass, ref, dref in rows:
        out = subprocess.run([os.path.join(DRIVERS, drv), "32", str(mass), "6.0", nconf, "100", "1000", "1337"], cwd=DRIVERS, env=dict(os.environ, QMG_QUIET="1"),
                             capture_output=True, text=True, timeout=1200)
        if out.returncode != 0 or "[QMG-BEGIN-PION]" not in out.stdout:
            print(drv, mass, "FAILED", out.stdout[-300:], flush=True)
            continue
        body = out.stdout[out.stdout.index("[QMG-BEGIN-PION]"):out.stdout.index("[QMG-END-PION]")]
        r = re.findall(r"^(\d+) ([-\d.e+]+) \+/- ([-\d.e+]+)$", body, re.M)
        t = np.array([int(x[0]) for x in r], dtype=float)
        c, dc = np.array([float(x[1]) for x in r]), np.array([float(x[2]) for x in r])
        res = []
        for lo in (5, 7, 9):
            sel = (t >= lo) & (t <= 16)
            (amp, m), cov = curve_fit(lambda tt, a, mm: a * np.cosh(mm * (tt - 16.0)), t[sel], c[sel], p0=(c[16], 0.3), sigma=dc[sel], absolute_sigma=True)
            res.append("t>=%d: %.5f(%.5f)" % (lo, m, np.sqrt(cov[1, 1])))
        unconv = re.search(r"(\d+) measurements, (\d+) unconverged", out.stdout)
        print("%s m=%+.2f ref %.5f(%.5f)  %s  [%s]" % (drv[:3], mass, ref, dref, "  ".join(res), unconv.group(0) if unconv else "?"), flush=True)
