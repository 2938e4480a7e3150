"""Where the GPU idles during a K-cycle solve: gaps between consecutive kernels of a rocprofv3 --kernel-trace, grouped by the kernel
that PRECEDES the gap (a gap after a reduction's final kernel is a host round trip; after anything else it is launch overhead).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_n13 -- quantum-mg_amd/drivers/n13_wilson_kcycle ...
    python tools/gap_analysis.py gpurun_out/prof_n13"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
SETUP_ONLY = ("k_probe_scatter", "k_unit_probe", "k_chol_store", "k_gaussian", "k_inv_real_sqrt", "k_galerkin", "k_block_ortho")
last_setup = max((i for i, r in enumerate(rows) if any(k in r[2] for k in SETUP_ONLY)), default=-1)
rows = rows[last_setup + 1:]
by = defaultdict(lambda: [0, 0.0])
tot = 0.0
for a, b in zip(rows, rows[1:]):
    gap = max(0, b[0] - a[1]) / 1e3
    name = a[2].split("(")[0].replace("void qmg::", "")[:40]
    by[name][0] += 1
    by[name][1] += gap
    tot += gap
wall = (rows[-1][1] - rows[0][0]) / 1e3
print("solve wall %.1f ms, idle %.1f ms (%.1f %%), %d launches" % (wall / 1e3, tot / 1e3, 100 * tot / wall, len(rows)))
for name, (n, g) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-42s gaps after it: %6d  total %.1f ms  avg %.1f us" % (name, n, g / 1e3, g / n))
