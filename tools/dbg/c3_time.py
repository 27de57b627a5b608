"""Where BASELINE config 3's run_pv goes: constructor, run_pv, and (under rocprofv3 --kernel-trace --stats) its kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
import pypevoc_amd
from tests.conftest import load_golden
g = load_golden("G7_perlman")
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = g["x"]
print(x.dtype, x.shape)
best = None
for i in range(30):
    t0 = time.perf_counter()
    p = pypevoc_amd.PV(x, g["sr"], nfft=4096, hop=1024, npks=100, progress=False, precision=prec)
    t1 = time.perf_counter()
    p.run_pv()
    t2 = time.perf_counter()
    r = (t1 - t0, t2 - t1)
    best = r if best is None or sum(r) < sum(best) else best
print("ctor %.1f us  run_pv %.1f us" % (best[0] * 1e6, best[1] * 1e6))
x32 = x.astype(np.float32)
best = None
for i in range(30):
    t0 = time.perf_counter()
    p = pypevoc_amd.PV(x32, g["sr"], nfft=4096, hop=1024, npks=100, progress=False, precision=prec)
    t1 = time.perf_counter()
    p.run_pv()
    t2 = time.perf_counter()
    r = (t1 - t0, t2 - t1)
    best = r if best is None or sum(r) < sum(best) else best
print("float32 samples: ctor %.1f us  run_pv %.1f us" % (best[0] * 1e6, best[1] * 1e6))
