"""gpurun_out/prof_mfma_{stats,fetch,write}{8,16} (rocprofv3 CSVs of tools/coarse_variants.py 512 24 ... nrhs) ->
profiles/r01_mfma_coarse_apply.json + kernel stats CSVs.  FETCH_SIZE x2 gfx950 correction as in summarize_profiles.py."""
import csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
L, nc = 512, 24
vol = L * L
out = {"workload": "Galerkin coarse operator apply, 512x512, nc=24, fp64, k right-hand sides per launch (kernel C, v_mfma_f64_16x16x4_f64)",
       "command": "rocprofv3 --kernel-trace --stats -- python3 tools/coarse_variants.py 512 24 '[{\"stencil_mfma\":1}]' <k>   (+ separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes)",
       "correction": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request), WRITE_SIZE exact", "hbm_peak_GBps": 8000.0,
       "f64_mfma_rate_measured_TFLOPs": "48 with 2 wavefronts/SIMD issuing back to back (profiles/r01_mfma_f64_rate.txt); vendor figure 78.6", "rows": []}
for nrhs in (8, 16):
    st = sorted(glob.glob(os.path.join(G, "prof_mfma_stats%d" % nrhs, "*", "*_kernel_stats.csv")))[-1]
    shutil.copy(st, os.path.join(P, "r01_mfma_coarse_apply_%drhs_kernel_stats.csv" % nrhs))
    k = [r for r in csv.DictReader(open(st)) if "k_stencil_mfma" in r["Name"]][0]
    def pmc(name):
        f = sorted(glob.glob(os.path.join(G, "prof_mfma_%s%d" % (name, nrhs), "*", "*_counter_collection.csv")))[-1]
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_stencil_mfma" in r["Kernel_Name"]]
        return sum(v) / len(v), len(v)
    fk, fn = pmc("fetch")
    wk, wn = pmc("write")
    alg = (5 * nc * nc + 2 * nc * nrhs) * 16 * vol
    flops = (8 * nc * nc * 5 + 8 * nc) * vol * nrhs
    mfma_per_site = 2 * 6 * (2 if nrhs <= 8 else 4) * 5
    avg = float(k["AverageNs"])
    out["rows"].append({"nrhs": nrhs, "kernel": k["Name"][:64], "calls": int(k["Calls"]), "avg_launch_ns": avg, "ms_per_rhs": avg / nrhs / 1e6,
                        "algorithmic_bytes_per_launch": alg, "achieved_GBps_algorithmic": alg / avg, "frac_of_hbm_peak": alg / avg / 8000.0,
                        "useful_TFLOPs": flops / avg / 1e3, "mfma_per_site": mfma_per_site, "executed_mfma_TFLOPs_incl_padding": mfma_per_site * 2048.0 * vol / avg / 1e3,
                        "hbm_read_bytes_corrected": 2 * fk * 1024, "hbm_write_bytes": wk * 1024, "traffic_over_algorithmic": (2 * fk + wk) * 1024 / alg, "pmc_launches": [fn, wn]})
json.dump(out, open(os.path.join(P, "r01_mfma_coarse_apply.json"), "w"), indent=1)
print(json.dumps(out["rows"], indent=1))
