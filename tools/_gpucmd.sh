# scratch: the command of the builder's last ad-hoc GPU call (gpurun -- 'bash tools/_gpucmd.sh'); not part of the product or of the collection scripts
cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT/quantum-mg_amd/drivers
G=../../tests/golden/l64t64b60_heatbath.dat
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/ab_new $O/ab_head
QMG_QUIET=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ab_new -- ./n13_wilson_kcycle 2048 -0.07 6.0 2 24 $G 64 > $O/ab_new.log 2>&1
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_ab:$LD_LIBRARY_PATH QMG_QUIET=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ab_head -- ./n13_wilson_kcycle 2048 -0.07 6.0 2 24 $G 64 > $O/ab_head.log 2>&1
python3 - <<'PY'
import csv, glob, os, re
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for v in ("new","head"):
    f=glob.glob(O+"/ab_"+v+"/**/*kernel_stats.csv", recursive=True)[0]
    tot={}
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        m=re.search(r"k_b?multi(dot|_caxpy(_small)?)<[^>]*>", n)
        if m: tot[m.group(0)]=(int(r["Calls"]), float(r["TotalDurationNs"])/1e6)
    print("==",v, "sum_ms %.2f"%sum(t for _,t in tot.values()))
    for k,(c,t) in sorted(tot.items()): print("   %-40s calls %6d  total %8.2f ms"%(k,c,t))
PY
rm -rf $O/ab_new $O/ab_head
