#!/usr/bin/env python3
"""Host-side time stamps of one config-3 round trip (PVX_TRACE=1): the library's marks on stderr, Python's around them."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pypevoc_amd
from tests.conftest import load_golden

g = load_golden("G7_perlman")
x = g["x"]
for it in range(6):
    if it == 5:
        os.environ["PVX_TRACE"] = "1"
    t0 = time.perf_counter()
    q = pypevoc_amd.PV(x, g["sr"], nfft=4096, hop=1024, npks=100, progress=False)
    t1 = time.perf_counter(); q.run_pv()
    t2 = time.perf_counter(); ss = q.toSinSum()
    t3 = time.perf_counter(); w = ss.synth(g["sr"], 1024)
    t4 = time.perf_counter()
    if it >= 4:
        print("python: PV() %.1f us | run_pv %.1f | toSinSum %.1f | synth %.1f | all %.1f" % tuple(1e6 * v for v in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)), file=sys.stderr)
