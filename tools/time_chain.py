#!/usr/bin/env python3
"""Where the wall time of the config-3 round trip goes (fixture G7, nfft 4096, hop 1024, npks 100): Python-side
construction, the three library calls, and the same calls with a float32 signal.  python tools/time_chain.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pypevoc_amd
from tests.conftest import load_golden

g = load_golden("G7_perlman")


def best(fn, n=30):
    b = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); b = min(b, time.perf_counter() - t0)
    return b * 1e3, r


for name, x in (("float64 signal", g["x"]), ("float32 signal", g["x"].astype(np.float32))):
    mk = lambda: pypevoc_amd.PV(x, g["sr"], nfft=4096, hop=1024, npks=100, progress=False)
    t_init, p = best(mk)
    p.run_pv()
    t_run, _ = best(lambda: (setattr(p, "oldfft", np.zeros(p.nfft2)), p.run_pv()))
    t_sin, ss = best(lambda: p.toSinSum())
    t_syn, w = best(lambda: ss.synth(g["sr"], 1024))
    def chain():
        q = pypevoc_amd.PV(x, g["sr"], nfft=4096, hop=1024, npks=100, progress=False)
        q.run_pv()
        return q.toSinSum().synth(g["sr"], 1024)
    t_all, _ = best(chain)
    print("%s: PV() %.3f ms | run_pv %.3f | toSinSum %.3f | synth %.3f | whole chain incl. construction %.3f ms" % (name, t_init, t_run, t_sin, t_syn, t_all))

# the resynthesis kernel's workgroup size on this short signal (default: 512 threads per segment below 1024 segments)
x = g["x"]
q = pypevoc_amd.PV(x, g["sr"], nfft=4096, hop=1024, npks=100, progress=False); q.run_pv(); ss = q.toSinSum()
for nt in ("256", "512"):
    os.environ["PVX_SYNTH_THREADS"] = nt
    t, _ = best(lambda: ss.synth(g["sr"], 1024))
    print("synth with %s threads per segment: %.3f ms" % (nt, t))
os.environ.pop("PVX_SYNTH_THREADS", None)
