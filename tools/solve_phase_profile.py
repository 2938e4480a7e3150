"""Split a rocprofv3 --kernel-trace of a K-cycle driver (n13/n19) into setup and solve and summarise the SOLVE phase:
per-kernel totals, GPU busy time vs wall time (the gap is launch / host-sync latency), and launches per outer iteration.

The solve starts after the last setup-only kernel (Galerkin probes / block-orthonormalisation leaves).  What follows is cut into
SEGMENTS at every gap longer than 5 ms -- no launch or reduction round trip takes that long; such a gap is host work between phases: the
scratch reservation in front of a timed solve (GB-sized hipMallocs: 3 ms or 1.6 s on this pool), the creation of the fp32
shadow hierarchy between the two solves of the n22 driver -- and ONE segment is reported as "solve": the one with the most GPU time
(default) or the last one (`last`: the fp32 solve of `n22 ... f32`, which runs after the fp64 one).  Every segment is listed, so nothing
is hidden.  The full trace is too large to carry back from the GPU box, so this runs there and writes a small JSON:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_n13 -- quantum-mg_amd/drivers/n13_wilson_kcycle ...
    python tools/solve_phase_profile.py gpurun_out/prof_n13 [largest|last] > gpurun_out/n13_solve_phase.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
SETUP_ONLY = ("k_probe_scatter", "k_unit_probe", "k_chol_store", "k_gaussian", "k_inv_real_sqrt")
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"]
        # the stencil / transfer kernels serve every level: tell the levels apart by the launch grid (in blocks)
        if any(k in name for k in ("k_stencil", "k_restrict", "k_prolong", "k_wilson")) and "Grid_Size_X" in r:
            wx = max(1, int(r.get("Workgroup_Size_X", 1) or 1))
            name = name.split("(")[0] + " grid=%dx%d" % (int(r["Grid_Size_X"]) // wx, int(r.get("Grid_Size_Y", 1) or 1))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
last_setup = max((i for i, r in enumerate(rows) if any(k in r[2] for k in SETUP_ONLY)), default=-1)
solve = rows[last_setup + 1:]


def summarise(rs):
    if not rs:
        return {}
    wall = rs[-1][1] - rs[0][0]
    busy = 0
    cur_end = rs[0][0]
    for s, e, _ in rs:   # union of intervals (kernels on one stream rarely overlap, but be exact)
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
    per = defaultdict(lambda: [0, 0])
    for s, e, n in rs:
        key = n.split("(")[0].replace("void qmg::", "").replace("qmg::", "")
        per[key][0] += 1
        per[key][1] += e - s
    top = sorted(per.items(), key=lambda kv: -kv[1][1])
    # where the GPU idles: gaps grouped by the kernel that PRECEDES them (after a reduction's final kernel = a host round trip)
    gaps = defaultdict(lambda: [0, 0])
    for a, b in zip(rs, rs[1:]):
        key = a[2].split("(")[0].replace("void qmg::", "").replace("qmg::", "")
        gaps[key][0] += 1
        gaps[key][1] += max(0, b[0] - a[1])
    gtop = sorted(gaps.items(), key=lambda kv: -kv[1][1])
    return {"gaps_after": [{"kernel": k, "gaps": v[0], "total_ms": v[1] / 1e6, "avg_us": v[1] / v[0] / 1e3} for k, v in gtop[:16]],"wall_ms": wall / 1e6, "gpu_busy_ms": busy / 1e6, "idle_frac": 1.0 - busy / wall, "launches": len(rs),
            "avg_gap_us": (wall - busy) / 1e3 / max(1, len(rs) - 1),
            "kernels": [{"kernel": k, "calls": v[0], "total_ms": v[1] / 1e6, "avg_us": v[1] / v[0] / 1e3, "pct_of_wall": 100.0 * v[1] / wall} for k, v in top[:24]]}


which = sys.argv[2] if len(sys.argv) > 2 else "largest"
segs, cur = [], []
for r in solve:
    if cur and r[0] - max(x[1] for x in cur[-8:]) > 5_000_000:
        segs.append(cur)
        cur = []
    cur.append(r)
if cur:
    segs.append(cur)


def busy_of(rs):
    b, ce = 0, rs[0][0]
    for s_, e_, _ in rs:
        if e_ > ce:
            b += e_ - max(s_, ce)
            ce = e_
    return b


seg_info = [{"launches": len(g), "wall_ms": (g[-1][1] - g[0][0]) / 1e6, "gpu_busy_ms": busy_of(g) / 1e6,
             "gap_before_ms": (g[0][0] - segs[i - 1][-1][1]) / 1e6 if i else 0.0} for i, g in enumerate(segs)]
pick = len(segs) - 1 if which == "last" else max(range(len(segs)), key=lambda i: seg_info[i]["gpu_busy_ms"]) if segs else 0
# ("last": the driver's closing residual check is a handful of launches right behind the solve, inside the same segment)
out = {"trace": os.path.basename(f), "segment_rule": "gaps > 5 ms split the post-setup launches; reported segment: " + which, "segments": seg_info,
       "reported_segment": pick, "setup": summarise(rows[:last_setup + 1]), "solve": summarise(segs[pick] if segs else solve)}
# the driver's tail (true-residual check, dumps) is a handful of launches and stays inside "solve"
print(json.dumps(out, indent=1))
