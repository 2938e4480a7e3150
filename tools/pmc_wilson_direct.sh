set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $R
rm -rf gpurun_out/pmc_w_fetch gpurun_out/pmc_w_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_w_fetch -- python3 tools/wilson_direct_bench.py > gpurun_out/pmc_w_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w_write -- python3 tools/wilson_direct_bench.py > gpurun_out/pmc_w_write.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for name in ("fetch","write"):
    f=sorted(glob.glob("gpurun_out/pmc_w_%s/*/*_counter_collection.csv"%name))[-1]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "k_wilson_direct" in k or "k_stencil_pair" in k:
            acc[(k[:70], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(name, k, len(v), "avg KiB", sum(v)/len(v))
PY
