# Matrix-pipe counters of kernel C (one --pmc counter per pass; rocprofv3 gets python3 directly):
#   bash tools/pmc_mfma.sh 16        (GPU box)  ->  gpurun_out/r02_mfma_kernelC_pmc.json
set -e
NR=${1:-16}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $R
O=gpurun_out
for V in 1 2; do
  for C in MfmaUtil SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
    rm -rf $O/pmc_mfma_${V}_$C
    rocprofv3 --pmc $C --output-format csv -d $O/pmc_mfma_${V}_$C -- python3 tools/coarse_variants.py 512 24 "[{\"stencil_mfma\":$V}]" $NR > $O/pmc_mfma.log 2>&1
  done
done
python3 - $NR <<'PY'
import csv, glob, json, sys
nr = int(sys.argv[1])
out = {"workload": "coarse apply 512x512, nc = 24, fp64, %d right-hand sides per launch (kernel C)" % nr,
       "command": "rocprofv3 --pmc <one counter per pass> -- python3 tools/coarse_variants.py 512 24 '[{\"stencil_mfma\":V}]' %d ; V = 1: MODE 2 (real-form tiles, 36 MFMAs per piece), V = 2: MODE 0 (48)" % nr, "variants": {}}
for v, name in ((1, "MODE 2 (real form)"), (2, "MODE 0 (four real MFMAs per complex tile product)")):
    row = {}
    for c in ("MfmaUtil", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"):
        f = sorted(glob.glob("gpurun_out/pmc_mfma_%d_%s/*/*_counter_collection.csv" % (v, c)))[-1]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_stencil_mfma" in r["Kernel_Name"]]
        row[c] = {"avg": sum(vals) / len(vals), "n": len(vals), "min": min(vals), "max": max(vals)}
    row["mfma_instructions_per_launch"] = row["SQ_INSTS_VALU_MFMA_MOPS_F64"]["avg"] / 4.0
    out["variants"][name] = row
json.dump(out, open("gpurun_out/r02_mfma_kernelC_pmc.json", "w"), indent=1)
print(json.dumps({k: {"MfmaUtil": v["MfmaUtil"]["avg"], "mfma_per_launch": v["mfma_instructions_per_launch"]} for k, v in out["variants"].items()}, indent=1))
PY
rm -rf $O/pmc_mfma_1_* $O/pmc_mfma_2_*
