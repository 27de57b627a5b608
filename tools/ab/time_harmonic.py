"""PVHarmonic.run_pv on config 2's signal a few times, host-side stamps on stderr:   PVX_TRACE=1 python tools/ab/time_harmonic.py [same]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pypevoc_amd
from bench import c2_signal
x = c2_signal(600).astype(np.float64)
F = (len(x) - 2048 + 511) // 512
same = len(sys.argv) > 1
h = None
for it in range(10):
    if h is None or not same:
        h = pypevoc_amd.PVHarmonic(x, 44100, nfft=2048, hop=512, npks=8, progress=False, precision=64)
        h.set_f0(np.full(F + 1, 220.0))
    t0 = time.perf_counter(); h.run_pv(); print("run_pv %.2f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
