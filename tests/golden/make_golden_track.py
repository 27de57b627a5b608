#!/usr/bin/env python3
"""Golden vectors for the tracker's jump limit (run in the build container only):

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_track.py

  T2_maxpitchjmp   SinSum.add_frame(fr, f, mag, ph, realph, maxpitchjmp) (PVAnalysis.py:871-957) over the frames of one analysis
                   -- the loop of PV.toSinSum (PVAnalysis.py:306-322), which itself never forwards its maxpitchjmp argument --
                   with maxpitchjmp in {0.05, 0.5, 1.5, 12.0} semitones: a gliding, vibrating harmonic signal whose per-frame
                   pitch steps straddle those limits, so the number of partials changes with each.  Keys per limit
                   (in hundredths of a semitone): start_<j>, len_<j>, slot_<j>.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import f32exact, HERE  # noqa: E402  (sets up the reference import)

import numpy as np  # noqa: E402
from pypevoc import PV  # noqa: E402
from pypevoc.PVAnalysis import SinSum  # noqa: E402

LIMITS = (0.05, 0.5, 1.5, 12.0)


def table(p, ss):
    start = np.array([pp.start_idx for pp in ss.partial], dtype=np.int32)
    plen = np.array([len(pp.f) for pp in ss.partial], dtype=np.int32)
    slots = []
    for pp in ss.partial:
        for j in range(len(pp.f)):
            fr = pp.start_idx + j
            hit = np.flatnonzero((p.f[fr] == pp.f[j]) & (p.mag[fr] == pp.mag[j]) & (p.ph[fr] == pp.ph[j]) & (p.realph[fr] == pp.realph[j]))
            assert len(hit) == 1, (fr, j, hit)
            slots.append(hit[0])
    return start, plen, np.array(slots, dtype=np.int16)


def main():
    sr, nfft, hop, npks = 22050, 1024, 256, 8
    rng = np.random.default_rng(52)
    n = int(1.5 * sr)
    t = np.arange(n) / float(sr)
    # a glide of two octaves per second with a wide, fast vibrato: frame-to-frame steps from a few cents to several semitones
    f0 = 220.0 * 2.0 ** (1.3 * t) * (1.0 + 0.06 * np.sin(2 * np.pi * 11.0 * t) * (t > 0.5))
    phase = 2 * np.pi * np.cumsum(f0) / sr
    x = sum(0.3 / h * np.sin(h * phase) for h in range(1, 6)) + 0.003 * rng.standard_normal(n)
    x = f32exact(x)
    p = PV(x, sr, nfft=nfft, hop=hop, npks=npks, pkthresh=0.005, progress=False)
    p.run_pv()
    out = dict(x=x.astype(np.float32), sr=np.float64(sr), nfft=np.int64(nfft), hop=np.int64(p.hop), npks=np.int64(npks),
               pkthresh=np.float64(0.005), f=p.f, mag=p.mag, ph=p.ph, realph=p.realph, binno=p.binno, nframes=np.int64(p.nframes))
    msg = []
    for lim in LIMITS:
        ss = SinSum(p.sr, nfft=p.nfft, hop=p.hop)
        for fr in range(p.nframes):
            ss.add_frame(fr, p.f[fr, :], p.mag[fr, :], p.ph[fr, :], realph=p.realph[fr, :], maxpitchjmp=lim)
        st, ln, sl = table(p, ss)
        key = "%d" % int(round(lim * 100))
        out["start_" + key] = st; out["len_" + key] = ln; out["slot_" + key] = sl
        msg.append("%g: %d partials" % (lim, len(st)))
    path = os.path.join(HERE, "T2_maxpitchjmp.npz")
    np.savez_compressed(path, **out)
    print("T2_maxpitchjmp: F=%d; %s -> %.0f KB" % (p.nframes, "; ".join(msg), os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
