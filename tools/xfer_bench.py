import importlib, os, sys
sys.path.insert(0, '/root/repo')
import torch
qmg = importlib.import_module("quantum-mg_amd"); qmg.init(0)
fL, cL, cnc = 2048, 512, 24
fsize, csize = fL*fL*2, cL*cL*cnc
fd, cd = (fL, fL, 2), (cL, cL, cnc)
def gauss(n, s):
    d = qmg.DeviceArray(n); qmg.gaussian(d, n, s); return d
nv = gauss(cnc*fsize, 5)
t = qmg.Timer()
for k in (4, 8):
    fb, cb = gauss(k*fsize, 31), gauss(k*csize, 32)
    for name, fn in (("prolong", lambda: qmg.prolong_batch(nv, cnc, cb, fb, fd, cd, k, csize, fsize, (1<<k)-1)), ("restrict", lambda: qmg.restrict_batch(nv, cnc, fb, cb, fd, cd, k, fsize, csize, (1<<k)-1))):
        for _ in range(2): fn()
        qmg.sync(); t.start()
        for _ in range(5): fn()
        ms = t.stop_ms()/5
        b = (cnc*fsize + 2*k*fsize + k*csize)*16
        print("%s k=%d %.3f ms %.0f GB/s" % (name, k, ms, b/ms/1e6))
    fb.free(); cb.free()
