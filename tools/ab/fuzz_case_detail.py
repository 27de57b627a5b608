"""Frames of a fuzz case where a float32 mode takes another peak set than the oracle, with the float64 magnitudes of the
bins involved:   python tools/ab/fuzz_case_detail.py SEED:INDEX [mode ...]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz
from oracle import pvoracle
s, i = sys.argv[1].split(":")
c = fuzz.make_case(int(s), int(i))
o = pvoracle.analyze(c["x"], c["sr"], c["nfft"], c["hop"], c["K"], c["thr"])
F = len(o["t"]); S = fuzz.spectrogram(c["x"], c["nfft"], c["hop"], F)
print({k: c[k] for k in ("nfft", "hop", "K", "thr", "sr", "n", "kind", "f32in")})
for mode in [int(m) for m in sys.argv[2:]] or [5, 2, 0]:
    p = fuzz.run_hip(c, 32, mode)
    for fr in range(F):
        ob = o["binno"][fr][o["f"][fr] > 0].astype(int); hb = p.binno[fr][p.f[fr] > 0].astype(int)
        if not np.array_equal(ob, hb):
            print("mode %d frame %d: oracle bins %s (|X| %s)  hip bins %s (|X| %s)  frame max %.6g" % (
                mode, fr, ob[:6], ["%.9g" % S[fr, b] for b in ob[:6]], hb[:6], ["%.9g" % S[fr, b] for b in hb[:6]], S[fr].max()))
