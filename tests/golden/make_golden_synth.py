#!/usr/bin/env python3
"""Golden vectors for SinSum.synth's OTHER parameters (run in the build container only):

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_synth.py

  S1_synth_params   one analysis (harmonic vibrato with two silent stretches, so partials start and stop: attacks and
                    releases everywhere; nfft 1024, hop 256, npks 10) resynthesised by the reference with
                    (synthesis hop, edge, minframes) in
                        (256, 0.5, 3) (256, 2.0, 3) (256, 1.0, 1) (256, 1.0, 6) (300, 0.25, 2) (128, 1.5, 4) (256, 0.0, 3)
                    -- edge scales the raised-cosine attack / release (PVAnalysis.py:738-751, 1055-1056; 0: none), minframes
                    drops short partials (PVAnalysis.py:1061), a synthesis hop other than the analysis hop stretches time.
                    Keys: w_<hop>_<edge x 100>_<minframes>.  The other fixtures only use edge = 1.0, minframes = 3.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import f32exact, harmonic_vibrato, partial_table, ref_sinsum_synth, HERE  # noqa: E402  (sets up the reference import)

import numpy as np  # noqa: E402
from pypevoc import PV  # noqa: E402

CASES = ((256, 0.5, 3), (256, 2.0, 3), (256, 1.0, 1), (256, 1.0, 6), (300, 0.25, 2), (128, 1.5, 4), (256, 0.0, 3))


def main():
    sr, nfft, hop, npks = 22050, 1024, 256, 10
    x = harmonic_vibrato(sr, 1.2, f0=330.0, nharm=7, seed=41, noise=0.004)
    n = len(x)
    x[n // 5:n // 5 + 3000] = 0.0
    x[(3 * n) // 5:(3 * n) // 5 + 1500] *= 0.02
    x = f32exact(x)
    p = PV(x, sr, nfft=nfft, hop=hop, npks=npks, pkthresh=0.005, progress=False)
    p.run_pv()
    out = dict(x=x.astype(np.float32), sr=np.float64(sr), nfft=np.int64(nfft), hop=np.int64(p.hop), npks=np.int64(npks),
               pkthresh=np.float64(0.005), f=p.f, mag=p.mag, ph=p.ph, realph=p.realph, binno=p.binno, t=p.t,
               totalmag=np.array(p.totalmag), nframes=np.int64(p.nframes))
    ss = p.toSinSum()
    start, plen, slots = partial_table(p, ss)
    out.update(part_start=start, part_len=plen, part_slot=slots)
    for h, edge, mf in CASES:
        out["w_%d_%d_%d" % (h, int(round(edge * 100)), mf)] = ref_sinsum_synth(ss, sr, h, edge=edge, minframes=mf)
    path = os.path.join(HERE, "S1_synth_params.npz")
    np.savez_compressed(path, **out)
    print("S1_synth_params: F=%d partials=%d (lengths %d..%d) -> %.0f KB" % (p.nframes, len(start), plen.min(), plen.max(), os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
