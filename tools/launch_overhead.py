import importlib, time, sys, os
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    print("torch", torch.__version__, "hip", torch.version.hip)
    if len(sys.argv) > 2: torch.cuda.synchronize()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
qmg = importlib.import_module("quantum-mg_amd")
qmg.init(0)
x = qmg.DeviceArray.zeros(4096)
qmg.zero_vector(x, 4096); qmg.sync()
for n in (100, 1000):
    t = time.perf_counter()
    for _ in range(n): qmg.zero_vector(x, 4096)
    t1 = time.perf_counter(); qmg.sync(); t2 = time.perf_counter()
    print("zero_vector x%d: issue %.1f us/call, total %.1f us/call" % (n, (t1 - t) / n * 1e6, (t2 - t) / n * 1e6))
os.system("grep -i hip /proc/%d/maps | awk '{print $6}' | sort -u | head" % os.getpid())
