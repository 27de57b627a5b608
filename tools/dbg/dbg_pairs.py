"""Frames on which the analysis of the 'pairs' test signal differs from the oracle: python tools/dbg/dbg_pairs.py nfft npks thr hop"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
import pypevoc_amd as amd
from oracle import pvoracle as oracle
nfft = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.005
hop = int(sys.argv[4]) if len(sys.argv) > 4 else nfft // 4
sr = 44100.0
n = 40000 * nfft // 2048
pairs = np.zeros(n + 64)
for j in range(16):
    pairs[nfft // 2 + 7 + j::nfft] += 0.5 ** j
    pairs[nfft // 2 + 47 + j::nfft] += 0.5 * 0.5 ** j
x = pairs[:n].astype(np.float32).astype(np.float64)
p = amd.PV(x, sr, nfft=nfft, hop=hop, npks=K, pkthresh=thr, progress=False, precision=32); p.run_pv()
o = oracle.analyze(x, sr, nfft, hop, K, thr)
print("mode", amd._lib.load().pvx_plan_get_fft_mode(p._plan.handle), "frames", p.nframes)
ob = np.asarray(o["binno"]); gb = np.asarray(p.binno)
nbad = 0
for fr in range(p.nframes):
    if not np.array_equal(gb[fr], ob[fr]):
        nbad += 1
        if nbad <= 6:
            print(fr, "gpu", gb[fr][gb[fr] > 0].astype(int)); print(fr, "ref", ob[fr][ob[fr] > 0].astype(int))
print("frames that differ:", nbad)
