#!/usr/bin/env python3
"""Wall-clock of the whole chain through the Python drop-in API (host buffers, PCIe included):
run_pv -> toSinSum -> synth, on the BASELINE.md 60-s signal and on config 2 (600 s) and config 3."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import pypevoc_amd
from tests.conftest import load_golden

def chain(x, sr, nfft, hop, K, label, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        p = pypevoc_amd.PV(x, sr, nfft=nfft, hop=hop, npks=K, progress=False)
        p.run_pv(); t1 = time.perf_counter()
        ss = p.toSinSum(); pid, st, ln = ss.partial_table(); t2 = time.perf_counter()
        w = ss.synth(sr, hop); t3 = time.perf_counter()
        r = (t1 - t0, t2 - t1, t3 - t2)
        best = r if best is None or sum(r) < sum(best) else best
    F = p.nframes
    print("%s: F=%d partials=%d | run_pv %.4f s (%.0f frames/s) | toSinSum %.4f s (%.0f frames/s) | synth %.4f s (%.0f frames/s, %.0fx real time)" %
          (label, F, len(st), best[0], F / best[0], best[1], F / best[1], best[2], F / best[2], (len(w) / sr) / best[2]))

x = bench.c2_signal(60)
chain(x, 44100, 2048, 512, 8, "60 s, 44.1 kHz, nfft 2048, hop 512, K 8 (BASELINE.md row)")
x = bench.c2_signal(600)
chain(x, 44100, 2048, 512, 8, "config 2: 600 s")
g = load_golden("G7_perlman")
chain(g["x"], g["sr"], 4096, 1024, 100, "config 3: perlmanVn.wav nfft 4096 K 100")
