#!/usr/bin/env python3
"""Accuracy of fft mode 5 (k_fused_team) next to mode 2 (k_fused_mw) against the oracle on the signals of
tests/test_hip_parity.py::test_team_kernel_against_oracle_and_itself.   python tools/ab_team.py [nfft]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pypevoc_amd as amd
from oracle import pvoracle as oracle
from tests.parity import compare_analysis, pv_result
nffts = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4096, 8192]
for nfft in nffts:
    rng = np.random.default_rng(78); sr = 44100.0; n = 40000 * nfft // 2048; t = np.arange(n) / sr
    noise = 0.1 * rng.standard_normal(n)
    harm = sum(0.3 / h * np.sin(2 * np.pi * 220 * h * t) for h in range(1, 9)) + 1e-3 * rng.standard_normal(n)
    gaps = harm.copy(); gaps[n // 7:n // 7 + 3 * nfft] = 0.0; gaps[n // 2:n // 2 + nfft + 100] = 0.0
    quant = np.round(harm * 50) / 50
    high = 0.2 * np.sin(2 * np.pi * 0.23 * sr * t) + 0.1 * np.sin(2 * np.pi * 0.249 * sr * t) + 0.02 * rng.standard_normal(n)
    for name, x in (("noise", noise), ("harm", harm), ("gaps", gaps), ("quant", quant), ("high", high)):
        x = x.astype(np.float32).astype(np.float64)
        for K, thr, hop in ((8, 0.005, nfft // 4), (1, 0.005, 333 * nfft // 2048), (3, 0.0, nfft // 4), (20, 0.3, nfft - 1), (64, 0.005, nfft // 8), (8, 0.005, nfft // 2), (40, 0.0005, nfft // 4)):
            o = oracle.analyze(x, sr, nfft, hop, K, thr)
            row = []
            for mode in (2, 5):
                os.environ["PVX_FFT_MODE"] = str(mode)
                p = amd.PV(x, sr, nfft=nfft, hop=hop, npks=K, pkthresh=thr, progress=False, precision=32); p.run_pv()
                c = compare_analysis(pv_result(p), o, nfft, hop, sr)
                row.append("m%d bad %d/%d ph %.2e rp %.2e f %.2e mag %.2e tm %.2e" % (mode, c["bad_peaks"], c["ref_peaks"], c["ph_norm"], c["realph_norm"], c["f_norm"], c["mag_norm"], c["totalmag_rel"]))
            print(nfft, name, K, thr, hop, "|", " | ".join(row)); sys.stdout.flush()
