#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run in the build container only (the reference never travels to the GPU box):

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is pinned (cases G1..G10):
  * PV.run_pv()            -> f, mag, ph, realph, binno, t, totalmag   (PVAnalysis.py:213-264)
  * PV.toSinSum()          -> partial table (start, len, slot per point) (PVAnalysis.py:299-322, 871-957)
  * SinSum.synth()         -> waveform                                   (PVAnalysis.py:1053-1070, 684-756)
  * PeakFinder(...)+filter_by_salience(rad=5) -> kept positions          (PeakFinder.py:35-74, 113-136, 155-194)

The reference resynthesiser is Python-2 code.  Two non-invasive shims make it run on
Python 3 / numpy 2 without touching /root/reference:
  1. `xrange = range` injected into the pypevoc.PVAnalysis module namespace
     (PVAnalysis.py:703 uses xrange);
  2. `ref_sinsum_synth` below restates the 12-line body of SinSum.synth
     (PVAnalysis.py:1053-1070) with edgsamp = int(edge*hop*dfr)  -- float sizes / slice
     indices were truncated by Python 2 + old numpy -- and calls the reference's own
     RegPartial.synth for all arithmetic.

Every fixture is data only: inputs + the reference's outputs on them.
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import warnings

import numpy as np

warnings.simplefilter("ignore")

import pypevoc.PVAnalysis as pva  # noqa: E402
from pypevoc import PV  # noqa: E402
from pypevoc.PeakFinder import PeakFinder  # noqa: E402

pva.xrange = range  # shim 1

HERE = os.path.dirname(os.path.abspath(__file__))


def ref_sinsum_synth(ss, sr, hop, edge=1.0, minframes=3):
    """Shim 2: SinSum.synth (PVAnalysis.py:1053-1070) with integer edge length."""
    hop = int(hop)
    dfr = ss.nfft / ss.hop / 2.0
    edgsamp = int(edge * hop * dfr)
    w = np.zeros((max(ss.end) + 2) * hop + 2 * edgsamp)
    for part in ss.partial:
        if len(part.f) >= minframes:
            wi, spl_st = part.synth(sr, hop, edge=edge)
            spl_st += edgsamp
            if spl_st >= 0:
                w[spl_st:spl_st + len(wi)] += wi
    return w[edgsamp:]


def f32exact(x):
    """Quantise to values exactly representable in float32 (kept as float64) so the fp32
    device path and the f64 reference see the same samples."""
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def partial_table(p, ss):
    """Partial table as (start, len, slots): point j of partial i is peak slot
    slots[off_i + j] of frame start_i + j in the (F, K) analysis arrays."""
    start = np.array([pp.start_idx for pp in ss.partial], dtype=np.int32)
    plen = np.array([len(pp.f) for pp in ss.partial], dtype=np.int32)
    slots = []
    for pp in ss.partial:
        for j in range(len(pp.f)):
            fr = pp.start_idx + j
            hit = np.flatnonzero((p.f[fr] == pp.f[j]) & (p.mag[fr] == pp.mag[j]) &
                                 (p.ph[fr] == pp.ph[j]) & (p.realph[fr] == pp.realph[j]))
            assert len(hit) == 1, (fr, j, hit)
            slots.append(hit[0])
    return start, plen, np.array(slots, dtype=np.int16)


def run_case(name, x, sr, nfft, hop, npks, pkthresh=0.005, synth_hops=(), track=True,
             x_store=None, extra=None, wave_dtype=np.float64):
    kw = dict(nfft=nfft, npks=npks, pkthresh=pkthresh, progress=False)
    if hop is not None:
        kw["hop"] = hop
    p = PV(x, sr, **kw)
    p.run_pv()
    out = dict(
        x=x if x_store is None else x_store, sr=np.float64(sr), nfft=np.int64(nfft),
        hop=np.int64(p.hop), npks=np.int64(npks), pkthresh=np.float64(pkthresh),
        f=p.f, mag=p.mag, ph=p.ph, realph=p.realph, binno=p.binno, t=p.t,
        totalmag=np.array(p.totalmag, dtype=np.float64), nframes=np.int64(p.nframes),
    )
    msg = "%s: F=%d K=%d" % (name, p.nframes, npks)
    if track and p.nframes > 0:
        ss = p.toSinSum()
        start, plen, slots = partial_table(p, ss)
        out.update(part_start=start, part_len=plen, part_slot=slots)
        msg += " partials=%d (>=3: %d)" % (len(start), int((plen >= 3).sum()))
        for h in synth_hops:
            w = ref_sinsum_synth(ss, sr, h)
            out["w_hop%d" % h] = w.astype(wave_dtype)
            msg += " w[h=%d]=%d" % (h, len(w))
    if extra:
        out.update(extra)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(msg, "-> %.0f KB" % (os.path.getsize(path) / 1024.0))
    return p


def harmonic_vibrato(sr, dur, f0=220.0, nharm=8, seed=1234, noise=0.001):
    """SURVEY.md section 8c G4 generator: 8-harmonic tone, 1 % / 5 Hz vibrato, amplitudes
    0.3/h, plus white noise."""
    n = int(sr * dur)
    t = np.arange(n) / float(sr)
    fi = f0 * (1.0 + 0.01 * np.sin(2 * np.pi * 5.0 * t))
    ph = 2 * np.pi * np.cumsum(fi) / sr
    x = np.zeros(n)
    for h in range(1, nharm + 1):
        x += 0.3 / h * np.sin(h * ph)
    x += noise * np.random.default_rng(seed).standard_normal(n)
    return x


def readme_signal(sr=44100):
    """README.md:26-63 signal; noise from RandomState(0) so it is reproducible."""
    rs = np.random.RandomState(0)
    vibfreq = 5.0
    hamp0 = 0.1 * np.array([1, .5, .3])
    hvib = 1.0 * np.array([.5, 0.1, .9])
    hph = np.array([0, np.pi / 2, np.pi])
    f0 = 500
    f0vib = 0.01
    dur = 1.0
    sig = np.zeros(int(sr * dur)) + 0.01 * (rs.rand(int(sr * dur)) - .5)
    t = np.arange(0, dur, 1. / sr)
    hvibsig = np.zeros((int(sr * dur), len(hamp0)))
    vibsig = np.sin(2 * np.pi * vibfreq * t)
    f0sig = f0 * (1 + f0vib * vibsig)
    for n, ha in enumerate(hamp0):
        hno = n + 1
        fsig = f0sig * hno
        phsig = np.cumsum(2 * np.pi * fsig / sr)
        hvibsig[:, n] = ha * (1 + hvib[n] * np.sin(2 * np.pi * vibfreq * t + hph[n]))
        sig += (hvibsig[:, n]) * np.sin(phsig)
    return sig


def main():
    sr = 44100

    # G1 -- the reference's own tests/test_pypevoc.py:4-16 signal, bit for bit (float64 input)
    t = np.arange(sr) / float(sr)
    xx = np.zeros(len(t))
    for ff, mm in zip([400., 1200.], [.1, .05]):
        xx += mm * np.sin(2.0 * np.pi * ff * 1.0 * t)
    p = run_case("G1_two_sines", xx, sr, 1024, 512, 20, synth_hops=(512,))
    ss = p.toSinSum()
    summ = [(ii, pp.start_idx, len(pp.f), float(np.mean(pp.f)), float(np.mean(pp.mag)))
            for ii, pp in enumerate(ss.partial) if np.mean(pp.mag) > 0.05 * 0.001]
    print("   G1 summary (tests/test_pypevoc.py prints this):", summ)

    # G2 / G3 -- README signal (BASELINE config 1) at default hop and at the metric hop
    sig = f32exact(readme_signal(sr))
    run_case("G2_readme_defaulthop", sig, sr, 2048, None, 3, synth_hops=(1024,),
             x_store=sig.astype(np.float32))
    run_case("G3_readme_hop512", sig, sr, 2048, 512, 3, synth_hops=(512,),
             x_store=sig.astype(np.float32))

    # G4 + G10 -- config-2 shape, 2 s; G10 = time-stretched resynthesis (hop 700 != 512)
    x4 = f32exact(harmonic_vibrato(sr, 2.0))
    run_case("G4_harm8_vibrato", x4, sr, 2048, 512, 8, synth_hops=(512, 700),
             x_store=x4.astype(np.float32))

    # G5 -- white noise: top-k / salience ties, tracker numbering, many partials
    x5 = f32exact(0.1 * np.random.default_rng(7).standard_normal(sr))
    run_case("G5a_noise_n1024_k20", x5, sr, 1024, 512, 20, synth_hops=(512,),
             x_store=x5.astype(np.float32))
    run_case("G5b_noise_n4096_k100", x5, sr, 4096, 1024, 100, synth_hops=(1024,),
             x_store=x5.astype(np.float32))
    # negative-threshold path of PeakFinder.findpos (npeaks > number of local maxima)
    run_case("G5c_noise_n512_k100", x5[:8192], sr, 512, 256, 100, synth_hops=(),
             x_store=x5[:8192].astype(np.float32))

    # G6 -- silence / post-silence frames (x/0 semantics, PF.py:69-70 threshold path)
    tt = np.arange(8192) / float(sr)
    s1k = 0.5 * np.sin(2 * np.pi * 1000.0 * tt)
    x6 = f32exact(np.concatenate([np.zeros(4096), s1k, np.zeros(4096), s1k]))
    run_case("G6_silence_gaps", x6, sr, 1024, 512, 4, synth_hops=(512,),
             x_store=x6.astype(np.float32))

    # G7 -- examples/perlmanVn.wav, the examples/WavResynth.py:18-36 round trip (config 3)
    from scipy.io import wavfile
    wsr, wav = wavfile.read("/root/reference/examples/perlmanVn.wav")
    assert wav.dtype == np.int16 and wav.ndim == 1
    sig7 = wav / float(np.iinfo(wav.dtype).max)
    run_case("G7_perlman", sig7, wsr, 4096, 1024, 100, synth_hops=(1024,), x_store=wav,
             extra=dict(x_scale=np.float64(1.0 / np.iinfo(wav.dtype).max)),
             wave_dtype=np.float32)

    # G9 -- chirp over the config-5 (nfft, hop) grid
    t9 = np.arange(2 * sr) / float(sr)
    x9 = f32exact(0.2 * np.sin(2 * np.pi * (100.0 * t9 + 2000.0 * t9 * t9)))
    np.savez_compressed(os.path.join(HERE, "G9_chirp_input.npz"), x=x9.astype(np.float32),
                        sr=np.float64(sr))
    for nfft in (512, 1024, 2048, 4096, 8192):
        for hop in (nfft // 4, nfft // 2):
            # the input is stored once; shorter clip for the small-nfft points keeps files small
            n9 = sr if nfft <= 1024 else 2 * sr
            run_case("G9_chirp_n%d_h%d" % (nfft, hop), x9[:n9], sr, nfft, hop, 8,
                     track=False, x_store=np.zeros(0), extra=dict(x_len=np.int64(n9)))

    # G11 -- odd geometry: hop not a multiple of 4, hop > nfft/2, nfft not a power of 4
    x11 = f32exact(harmonic_vibrato(sr, 0.5, f0=330.0, nharm=5, seed=99))
    run_case("G11_odd_hop300", x11, sr, 1024, 300, 6, synth_hops=(300, 256),
             x_store=x11.astype(np.float32))
    run_case("G11_odd_hop700", x11, sr, 1024, 700, 6, synth_hops=(700,),
             x_store=x11.astype(np.float32))

    # G8 -- PeakFinder in isolation
    rng = np.random.default_rng(2024)
    ys = f32exact(np.abs(rng.standard_normal((20, 1024))) * np.exp(-np.arange(1024) / 300.0))
    # plateaus and exact ties
    ys[3, 100:104] = ys[3, 100]
    ys[4, ::7] = 0.5
    ys[5, :] = np.round(ys[5, :] * 8) / 8
    res = {}
    for k in (1, 3, 8, 100):
        for thr in (0.005, 0.2):
            cnt = np.zeros(len(ys), dtype=np.int32)
            allpos = -np.ones((len(ys), k), dtype=np.int32)   # after findpos (ascending)
            keep = np.zeros((len(ys), k), dtype=np.int8)      # after filter_by_salience(rad=5)
            for i, y in enumerate(ys):
                pk = PeakFinder(y, npeaks=k, minrattomax=thr)
                pk.boundaries()
                pk.filter_by_salience(rad=5)
                n = len(pk._idx)
                cnt[i] = n
                allpos[i, :n] = pk._idx
                keep[i, :n] = pk._keep
                assert np.array_equal(pk.get_pos(), pk._idx[pk._keep])
            tag = "k%d_t%s" % (k, str(thr).replace(".", "p"))
            res["cnt_" + tag] = cnt
            res["pos_" + tag] = allpos
            res["keep_" + tag] = keep
    # tests/test_peak_finder.py:16-20 ramp, default arguments
    ramp = np.concatenate((np.linspace(0, 1, 10), np.linspace(.9, 1, 9)))
    pk = PeakFinder(ramp)
    res["ramp"] = ramp
    res["ramp_pos"] = np.asarray(pk.pos, dtype=np.int32)
    path = os.path.join(HERE, "G8_peakfinder.npz")
    np.savez_compressed(path, ys=ys.astype(np.float32), **res)
    print("G8_peakfinder: ramp pos", pk.pos, "-> %.0f KB" % (os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
