# HBM traffic (rocprofv3 PMC, FETCH_SIZE and WRITE_SIZE in separate passes) of the kernels an A/B tool launches:
#   bash tools/pmc_ab.sh tools/wilson_direct_bench.py      (GPU box)
# Prints per kernel and grid: launches, average counter value (KiB).  HBM bytes = FETCH_SIZE x 2 (gfx950: 64-byte units
# reported as 32, MI355X_MICROARCH.md) x 1024 + WRITE_SIZE x 1024.
set -e
S=${1:-tools/wilson_direct_bench.py}
T=$(basename $S .py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $R
rm -rf gpurun_out/pmc_${T}_fetch gpurun_out/pmc_${T}_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${T}_fetch -- python3 $S > gpurun_out/pmc_${T}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${T}_write -- python3 $S > gpurun_out/pmc_${T}_write.log 2>&1
python3 - $T <<'PY'
import csv, glob, collections, sys
T = sys.argv[1]
tot = collections.defaultdict(dict)
for name in ("fetch", "write"):
    f = sorted(glob.glob("gpurun_out/pmc_%s_%s/*/*_counter_collection.csv" % (T, name)))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(s in k for s in ("k_wilson_", "k_stencil_pair", "k_stencil_site", "k_reduce<", "k_apply_norm_final")):
            acc[(k.split("(")[0].replace("void qmg::", "")[:80], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(name, k, len(v), "avg KiB", sum(v) / len(v))
        tot[k][name] = sum(v) / len(v)
for k, v in tot.items():
    if "fetch" in v and "write" in v:
        print("hbm_bytes_per_launch", k, "%.0f" % (v["fetch"] * 2 * 1024 + v["write"] * 1024))
PY
rm -rf gpurun_out/pmc_${T}_fetch gpurun_out/pmc_${T}_write
