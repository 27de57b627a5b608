#!/usr/bin/env python3
"""Golden vector T1 from the reference (run in the build container only):

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ties.py

  T1_tracker_ties   hand-built (f, mag) arrays through the reference's own PV.toSinSum / SinSum.add_frame
                    (PVAnalysis.py:299-322, 871-957) with EXACT ties:
      * previous partials of equal magnitude that are exactly equally far (in semitones) from a new peak:
        sorted(zip(pmag, pidx), reverse=True) (PVAnalysis.py:893) lets the higher PARTIAL INDEX win, wherever
        the two sit in the frame's slots (both slot orders occur);
      * new peaks of equal magnitude only where the outcome does not depend on their order: for equal
        magnitudes np.argsort(mag)[::-1] (PVAnalysis.py:873-875) is whatever numpy's unstable default sort does
        on the CPU at hand.  The script checks that: the table must not change when np.argsort is forced to
        kind="stable", and prints how often the two sorts disagree on this host.
    With e = 2**-6 and c a multiple of 64: a new peak at (1 - e*e) c is exactly 17.312 * e semitones from
    partials at (1 - e) c and (1 + e) c (both quotients are exact in float64).

  R1_regpartial_nofstep   RegPartial(istart, overlap=0.25, fstep=None).synth(sr, hop, edge) on the longest
                    partial of fixture G4 and with a time-stretching hop: the branch without frequency-slope
                    phase correction (PVAnalysis.py:710-713), which SinSum never takes (it always sets fstep).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import partial_table, HERE  # noqa: E402  (sets up the reference import)

import numpy as np  # noqa: E402
import pypevoc.PVAnalysis as PVA  # noqa: E402
from pypevoc import PV  # noqa: E402


def build():
    K, e = 20, 2.0 ** -6
    rows = []            # per frame: list of (f, mag)
    cs = [4096.0, 1024.0, 8192.0, 2048.0]
    lo = [(1 - e) * c for c in cs]
    hi = [(1 + e) * c for c in cs]
    mid = [(1 - e * e) * c for c in cs]
    filler = [(300.0 + 37.0 * i, 0.01 + 0.001 * i) for i in range(12)]          # distinct magnitudes, far away
    # frame 0: every pair starts with distinct magnitudes (creation order = magnitude order)
    rows.append([(lo[0], .50), (hi[0], .49), (lo[1], .48), (hi[1], .47), (hi[2], .46), (lo[2], .45), (lo[3], .44), (hi[3], .43)] + filler)
    # frame 1: the pairs continue with EQUAL magnitudes (order-independent: each is nearest to its own partial);
    # pair 0 and 3: the higher partial index sits in the LOWER slot, pair 1 and 2: in the higher slot
    rows.append([(hi[0], .30), (lo[0], .30), (lo[1], .29), (hi[1], .29), (lo[2], .28), (hi[2], .28), (hi[3], .27), (lo[3], .27)] + filler)
    # frame 2: one new peak exactly between the two partials of each pair
    rows.append([(mid[0], .20), (mid[1], .19), (mid[2], .18), (mid[3], .17)] + filler[:6])
    # frame 3: the pairs again (the partial that was not continued is gone: new partials appear), equal magnitudes
    rows.append([(lo[0], .30), (hi[0], .30), (hi[1], .29), (lo[1], .29)] + filler[:6])
    # frame 4: exactly between again, and two equal-magnitude newcomers far from everything (order-independent?
    # no: both would start partials and their numbering follows their order -- so give them distinct magnitudes)
    rows.append([(mid[0], .21), (mid[1], .20), (15000.0, .05), (16000.0, .06)] + filler[:6])
    rows.append([(mid[0], .21), (mid[1], .20), (15000.0, .05)] + filler[:3])
    F = len(rows)
    f = np.zeros((F, K)); mag = np.zeros((F, K)); ph = np.zeros((F, K)); realph = np.zeros((F, K))
    for i, r in enumerate(rows):
        for j, (ff, mm) in enumerate(r):
            f[i, j] = ff; mag[i, j] = mm
            ph[i, j] = 0.001 * (i * K + j + 1); realph[i, j] = ph[i, j] + 0.5
    return f, mag, ph, realph


def track(f, mag, ph, realph, sr, nfft, hop):
    p = PV(np.zeros(nfft + hop * f.shape[0] + 1), sr, nfft=nfft, hop=hop, npks=f.shape[1], progress=False)
    p.f, p.mag, p.ph, p.realph, p.nframes = f, mag, ph, realph, f.shape[0]
    ss = p.toSinSum()
    return partial_table(p, ss)


def main():
    sr, nfft, hop = 44100, 1024, 256
    f, mag, ph, realph = build()
    e = 2.0 ** -6
    for c in (4096.0, 1024.0, 8192.0, 2048.0):                  # the construction is exact
        fc = (1 - e * e) * c
        assert abs(PVA.dpitch2st((1 - e) * c, fc)) == abs(PVA.dpitch2st((1 + e) * c, fc)) < 0.5
    start, plen, slots = track(f, mag, ph, realph, sr, nfft, hop)
    real_argsort = np.argsort
    PVA.np.argsort = lambda a, *args, **kw: real_argsort(a, kind="stable")
    try:
        s2, l2, sl2 = track(f, mag, ph, realph, sr, nfft, hop)
    finally:
        PVA.np.argsort = real_argsort
    assert np.array_equal(start, s2) and np.array_equal(plen, l2) and np.array_equal(slots, sl2), \
        "the fixture depends on numpy's tie order"
    rng = np.random.default_rng(0)
    rowsK = [np.round(rng.random(8) * 6) / 6.0 for _ in range(2000)]
    dis = sum(not np.array_equal(np.argsort(r), np.argsort(r, kind="stable")) for r in rowsK)
    print("numpy %s: default argsort differs from kind='stable' on %d of 2000 8-element rows with ties" % (np.__version__, dis))
    path = os.path.join(HERE, "T1_tracker_ties.npz")
    np.savez_compressed(path, f=f, mag=mag, ph=ph, realph=realph, sr=np.float64(sr), nfft=np.int64(nfft), hop=np.int64(hop),
                        part_start=start, part_len=plen, part_slot=slots)
    print("T1_tracker_ties: F=%d K=%d partials=%d; start=%s len=%s" % (f.shape[0], f.shape[1], len(start), start.tolist(), plen.tolist()))

    # ---- R1: RegPartial without fstep
    g = np.load(os.path.join(HERE, "G4_harm8_vibrato.npz"))
    p = PV(np.zeros(int(g["nfft"]) + int(g["hop"]) * g["f"].shape[0] + 1), float(g["sr"]), nfft=int(g["nfft"]), hop=int(g["hop"]),
           npks=g["f"].shape[1], progress=False)
    p.f, p.mag, p.ph, p.realph, p.nframes = g["f"], g["mag"], g["ph"], g["realph"], g["f"].shape[0]
    ss = p.toSinSum()
    src = max(ss.partial, key=lambda pp: len(pp.f))
    npt = 40                                                     # the first 40 points keep the fixture small
    out = dict(sr=g["sr"], overlap=np.float64(src.overlap), start_idx=np.int64(src.start_idx),
               f=np.array(src.f[:npt]), mag=np.array(src.mag[:npt]), ph=np.array(src.ph[:npt]), realph=np.array(src.realph[:npt]))
    for hop_s, edge in ((512, 0.5), (700, 1.0)):
        part = PVA.RegPartial(src.start_idx, overlap=src.overlap, fstep=None)
        for a, b, c, d in zip(out["f"], out["mag"], out["ph"], out["realph"]):
            part.append_point(a, b, c, realph=d)
        sig, first = part.synth(float(g["sr"]), hop_s, edge=edge)
        out["sig_hop%d" % hop_s] = sig
        out["first_hop%d" % hop_s] = np.int64(first)
        out["edge_hop%d" % hop_s] = np.float64(edge)
    path = os.path.join(HERE, "R1_regpartial_nofstep.npz")
    np.savez_compressed(path, **out)
    print("R1_regpartial_nofstep: %d points -> %.0f KB" % (npt, os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
