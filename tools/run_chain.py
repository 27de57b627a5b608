#!/usr/bin/env python3
"""The whole path on BASELINE config 2's signal -- PV.run_pv -> toSinSum -> SinSum.synth, results resident -- a few times
(for rocprofv3 --pmc / --kernel-trace runs over the tracker and resynthesis kernels).
   python3 tools/run_chain.py [reps] [seconds] [precision]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pypevoc_amd  # noqa: E402
from bench import c2_signal  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
secs = int(sys.argv[2]) if len(sys.argv) > 2 else 600
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 32
x = c2_signal(secs)
for _ in range(reps):
    p = pypevoc_amd.PV(x, 44100, nfft=2048, hop=512, npks=8, progress=False, precision=prec)
    p.run_pv()
    ss = p.toSinSum()
    w = ss.synth(44100, 512)
print("chain: %d frames, %d samples out, %d reps" % (p.nframes, len(w), reps))
