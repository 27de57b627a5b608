import sys, os, time, numpy as np
sys.path.insert(0, "/root/repo")
import pypevoc_amd
from tests.conftest import load_golden
g = load_golden("G7_perlman")
x = g["x"].astype(np.float32)
for it in range(6):
    t0 = time.perf_counter()
    p = pypevoc_amd.PV(x, g["sr"], nfft=4096, hop=1024, npks=100, progress=False)
    t0b = time.perf_counter()
    p.run_pv(); t1 = time.perf_counter()
    ss = p.toSinSum(); t2 = time.perf_counter()
    w = ss.synth(g["sr"], p.hop / 1); t3 = time.perf_counter()
    print("ctor %.3f run_pv %.3f toSinSum %.3f synth %.3f ms" % ((t0b - t0) * 1e3, (t1 - t0b) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), file=sys.stderr)
