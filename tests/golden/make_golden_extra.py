#!/usr/bin/env python3
"""More golden vectors from the reference (run in the build container only):

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_extra.py

  G12_hop_eighth      hop = nfft/8: in frame 0 and after silent frames the x/0 phase rule (PV.py:171, 190)
                      puts two unwrapping candidates of dphase2freq (PV.py:140-147) at exactly the same
                      distance from the bin centre -- pins which one the reference's float64 arithmetic keeps
  G13_blackman_thr01  a non-default window callable (wind=np.blackman) and pkthresh = 0.1
  G14_npks1           npks = 1 on a two-tone signal whose stronger tone changes half way
  G15_salience        PeakFinder.filter_by_salience with sal != 0 (PeakFinder.py:113-136; off the PV path, which passes
                      sal = 0): rows of G8's kind, rad in {1, 5}, sal in {-0.05, -0.5, 0.01}
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import f32exact, harmonic_vibrato, partial_table, ref_sinsum_synth, HERE  # noqa: E402  (sets up the reference import)

import numpy as np  # noqa: E402
from pypevoc import PV  # noqa: E402


def run(name, x, sr, nfft, hop, npks, pkthresh=0.005, wind=None, synth_hop=None):
    kw = dict(nfft=nfft, hop=hop, npks=npks, pkthresh=pkthresh, progress=False)
    if wind is not None:
        kw["wind"] = wind
    p = PV(x, sr, **kw)
    p.run_pv()
    out = dict(x=x.astype(np.float32), sr=np.float64(sr), nfft=np.int64(nfft), hop=np.int64(p.hop), npks=np.int64(npks),
               pkthresh=np.float64(pkthresh), f=p.f, mag=p.mag, ph=p.ph, realph=p.realph, binno=p.binno, t=p.t,
               totalmag=np.array(p.totalmag), nframes=np.int64(p.nframes))
    if wind is not None:
        out["win"] = np.asarray(p.win, dtype=np.float64)
    ss = p.toSinSum()
    start, plen, slots = partial_table(p, ss)
    out.update(part_start=start, part_len=plen, part_slot=slots)
    if synth_hop:
        out["w_hop%d" % synth_hop] = ref_sinsum_synth(ss, sr, synth_hop)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%s: F=%d K=%d partials=%d -> %.0f KB" % (name, p.nframes, npks, len(start), os.path.getsize(path) / 1024.0))


def main():
    sr = 44100
    rng = np.random.default_rng(12)
    n = 12000
    t = np.arange(n) / float(sr)
    x = 0.2 * np.sin(2 * np.pi * 997.0 * t) + 0.05 * rng.standard_normal(n)
    x[5000:7400] = 0.0                                       # silent frames and the frames after them
    run("G12_hop_eighth", f32exact(x), sr, 1024, 128, 20, synth_hop=128)
    x13 = f32exact(harmonic_vibrato(sr, 0.5, f0=440.0, nharm=6, seed=13, noise=0.01))
    run("G13_blackman_thr01", x13, sr, 2048, 512, 12, pkthresh=0.1, wind=np.blackman, synth_hop=512)
    t14 = np.arange(sr // 2) / float(sr)
    a = np.linspace(1.0, 0.0, len(t14))
    x14 = f32exact(0.3 * a * np.sin(2 * np.pi * 500.0 * t14) + 0.3 * (1 - a) * np.sin(2 * np.pi * 3000.0 * t14))
    run("G14_npks1", x14, sr, 1024, 256, 1, synth_hop=256)


def salience():
    from pypevoc.PeakFinder import PeakFinder
    rng = np.random.default_rng(2025)
    ys = f32exact(np.abs(rng.standard_normal((8, 512))) * np.exp(-np.arange(512) / 200.0))
    res = {}
    for rad in (1, 5):
        for sal in (-0.05, -0.5, 0.01):
            k = 12
            pos = -np.ones((len(ys), k), dtype=np.int32)
            keep = np.zeros((len(ys), k), dtype=np.int8)
            for i, y in enumerate(ys):
                pk = PeakFinder(y, npeaks=k, minrattomax=0.005)
                pk.filter_by_salience(rad=rad, sal=sal)
                n = len(pk._idx)
                pos[i, :n] = pk._idx
                keep[i, :n] = pk._keep
            tag = "r%d_s%s" % (rad, ("%g" % sal).replace(".", "p").replace("-", "m"))
            res["pos_" + tag] = pos
            res["keep_" + tag] = keep
    path = os.path.join(HERE, "G15_salience.npz")
    np.savez_compressed(path, ys=ys.astype(np.float32), **res)
    print("G15_salience: kept", {k: int(v.sum()) for k, v in res.items() if k.startswith("keep")})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "salience":
        salience()
    else:
        main()
        salience()
