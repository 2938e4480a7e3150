"""A/B timing of stencil-kernel tuning knobs, interleaved rounds in ONE process (guide rule 24)."""
import importlib, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
qmg = importlib.import_module("quantum-mg_amd")
qmg.init(0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
variants = json.loads(sys.argv[2]) if len(sys.argv) > 2 else [{"stencil_nt": 0}, {"stencil_nt": 1}]
fixture = os.path.join(ROOT, "tests", "golden", "l64t64b60_heatbath.dat")
wl = bench.Workload(qmg, L, fixture, 1337)
timer = qmg.Timer()
res = {i: [] for i in range(len(variants))}
for rnd in range(int(os.environ.get('ROUNDS', '8'))):
    for i, v in enumerate(variants):
        for k, val in v.items(): qmg.set_tuning(k, val)
        for _ in range(3): wl.step()
        qmg.sync()
        timer.start()
        for _ in range(20): wl.step()
        res[i].append(timer.stop_ms() / 20)
for i, v in enumerate(variants):
    t = np.array(res[i])
    print("   rounds:", " ".join("%.3f" % q for q in t))
    print("%-60s median %.4f ms  min %.4f ms  -> %.0f GB/s (min)" % (json.dumps(v), np.median(t), t.min(), 384.0 * L * L / t.min() / 1e6))
# parity must hold for the last variant too
qmg.set_tuning("stencil_ablate", 0)
print("parity gate:", wl.parity_gate(fixture))
