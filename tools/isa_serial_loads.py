"""Which kernels wait for their loads one at a time?  Compiles every csrc/*.hip to gfx950 assembly and counts, per kernel, the global loads that are
followed by a full `s_waitcnt vmcnt(0)` with ANOTHER load behind it (load, wait, load: the second load could not be issued until the first had
returned).  The pattern comes from a conversion / mask / shift or a divergent branch placed right behind a load (qmg_common.h, RawC) and is invisible
in the source.  Found that way in round 3: every complex<float> staging loop of kernels B / B32 / C and of the restricts, and the kernel-argument
byte load (`a.ridx[k]`) in front of the matrix prefetch of kernels B / B32.     python tools/isa_serial_loads.py [min_count]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
minc = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tmp = tempfile.mkdtemp()
for src in sorted(glob.glob(os.path.join(ROOT, "quantum-mg_amd", "csrc", "*.hip"))):
    asm = os.path.join(tmp, os.path.basename(src) + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", asm, src], stderr=subprocess.DEVNULL, check=True)
    name, seq, res = None, [], {}
    for ln in open(asm):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, seq = m.group(1), []
            continue
        if name is None:
            continue
        t = ln.strip()
        if t.startswith(("global_load", "buffer_load")):
            seq.append("L")
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            seq.append("W")
        elif t.startswith("s_endpgm"):
            s = "".join(seq)
            res[name] = (len(re.findall(r"LW(?=L)", s)), s.count("L"))
            name = None
    bad = sorted(((n, tot, k) for k, (n, tot) in res.items() if n >= minc), reverse=True)
    print("== %s: %d kernels, %d with >= %d serialised loads" % (os.path.basename(src), len(res), len(bad), minc))
    names = subprocess.run(["c++filt"] + [k for _, _, k in bad[:12]], capture_output=True, text=True).stdout.split("\n") if bad else []
    for (n, tot, k), d in zip(bad[:12], names):
        print("   %3d of %3d loads   %s" % (n, tot, d.split("(")[0].replace("void qmg::", "")))
