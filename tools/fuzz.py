#!/usr/bin/env python3
"""Randomised differential test: HIP analysis (every fft mode the plan supports, both precisions)
+ tracker + resynthesis against the CPU oracle over random signals and parameters.

  python tools/fuzz.py [seconds] [seed]        run cases seed:0, seed:1, ... for `seconds`
  python tools/fuzz.py --case SEED:INDEX       re-run one case verbosely
  PVX_FUZZ_LONG=1 ...                          cases of 300 .. 3000 frames instead of 1 .. 60 (waves with many rows, several
                                               tracker chunks, every flush cycle; the same checks)

float64 is held to the oracle strictly.  float32 is held to it on the WELL-CONDITIONED peaks
only: a peak whose bin, in this frame and in the previous one, is within 60 dB of that frame's
largest bin (the phase of a bin 60 dB down is already good to ~1e-4 rad in float32), in frames
whose selection is not decided by float32 rounding (see `robust_frame`)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pypevoc_amd
from pypevoc_amd import _lib
from oracle import pvoracle


def signal(rng, n, sr):
    t = np.arange(n) / sr
    kind = int(rng.integers(0, 5))
    x = np.zeros(n)
    if kind == 0:      # noise
        x = rng.uniform(0.01, 0.5) * rng.standard_normal(n)
    elif kind == 1:    # harmonic + noise
        f0 = rng.uniform(60, 2000)
        for h in range(1, int(rng.integers(2, 20))):
            if f0 * h < 0.45 * sr:
                x += rng.uniform(0.01, 0.5) / h * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6))
        x += rng.uniform(1e-5, 1e-2) * rng.standard_normal(n)
    elif kind == 2:    # chirps
        for _ in range(int(rng.integers(1, 5))):
            x += rng.uniform(0.05, 0.4) * np.sin(2 * np.pi * (rng.uniform(50, 0.2 * sr) * t + rng.uniform(-0.1, 0.1) * sr * t * t))
    elif kind == 3:    # bursts with exact silence in between
        f = rng.uniform(100, 0.3 * sr)
        x = 0.3 * np.sin(2 * np.pi * f * t)
        edges = np.sort(rng.integers(0, n, 6))
        for a, b in zip(edges[::2], edges[1::2]):
            x[a:b] = 0.0
    else:              # quantised (exact ties, plateaus)
        x = np.round(0.3 * rng.standard_normal(n) * 64) / 64
    return kind, x.astype(np.float32).astype(np.float64)


VERBOSE = False


LONG = bool(os.environ.get("PVX_FUZZ_LONG"))


def make_case(seed, idx, long=None):
    long = LONG if long is None else long
    rng = np.random.default_rng([seed, idx])
    nfft = int(rng.choice([128, 256, 512, 1000, 1024, 2048, 2048, 4096, 8192]))
    hop = int(rng.choice([nfft // 8, nfft // 4, nfft // 2, nfft // 3 + 1, nfft - 1]))
    K = int(rng.choice([1, 3, 8, 8, 20, 64, 100, 128, 70]))
    thr = float(rng.choice([0.0, 0.0005, 0.005, 0.005, 0.1]))
    sr = float(rng.choice([8000, 22050, 44100, 96000]))
    n = int(nfft + hop * (rng.integers(300, 3000) if long else rng.integers(1, 60)) + rng.integers(1, hop + 1))
    kind, x = signal(rng, n, sr)
    return dict(nfft=nfft, hop=hop, K=K, thr=thr, sr=sr, n=n, kind=kind, x=x,
                h2=int(rng.choice([hop, max(2, hop // 2), hop + 7])), f32in=bool(rng.random() < 0.5))


def spectrogram(x, nfft, hop, F):
    """|rfft| of the reference's frames in float64 (frame i starts at i*hop), F x (nfft/2+1)."""
    win = np.hanning(nfft)
    idx = np.arange(nfft)[None, :] + hop * np.arange(F)[:, None]
    return np.abs(np.fft.rfft(x[idx] * win, axis=1))


def run_hip(c, prec, mode):
    if mode is not None:
        os.environ["PVX_FFT_MODE"] = str(mode)
    try:
        xin = c["x"].astype(np.float32) if c["f32in"] else c["x"]
        p = pypevoc_amd.PV(xin, c["sr"], nfft=c["nfft"], hop=c["hop"], npks=c["K"], pkthresh=c["thr"],
                           progress=False, precision=prec)
        p.run_pv()
    finally:
        os.environ.pop("PVX_FFT_MODE", None)
    return p


def check64(p, o, c, S):
    """Every slot identical in bin and validity; values to float64 rounding of the spectra.

    Frames whose spectrum is FLAT (a single non-zero windowed sample: the Hann window is exactly 0
    at both ends, so this happens at every burst edge) have no maxima except those FFT rounding
    noise makes; numpy's FFT, the oracle's and rocFFT's noise differ, so they are skipped."""
    msgs = []
    smax = S.max(axis=1)
    live = (smax - S.min(axis=1)) > 1e-9 * smax
    same = (p.binno == o["binno"]).all(axis=1) & ((p.f > 0) == (o["f"] > 0)).all(axis=1)
    if not same[live].all():
        bad = np.nonzero(live & ~same)[0]
        msgs.append("peak sets differ in frames %s" % bad[:8].tolist())
        if VERBOSE:
            for i in bad[:3]:
                a = i * c["hop"]
                seg = c["x"][a:a + c["nfft"]]
                nz = np.nonzero(seg)[0]
                print(" frame %d: nonzero samples %d (first %s last %s)" % (i, len(nz), nz[:1], nz[-1:]))
                print("  oracle bins", o["binno"][i][:12], "f", o["f"][i][:6], "mag", o["mag"][i][:6])
                print("  hip    bins", p.binno[i][:12], "f", p.f[i][:6], "mag", p.mag[i][:6])
        return msgs
    v = (o["f"] > 0) & (live & same)[:, None]
    if not v.any():
        return msgs
    dt = c["hop"] / c["sr"]
    F, K = o["f"].shape
    b = o["binno"].astype(np.int64)
    rows = np.arange(F)[:, None]
    # conditioning of a bin's phase: (largest bin of the frame) / (this bin), this frame and the previous
    wc = smax[:, None] / np.maximum(S[rows, b], 1e-300)
    prev = np.maximum(rows - 1, 0)
    wp = np.where(smax[prev] > 0, smax[prev] / np.maximum(S[prev, b], 1e-300), 1.0)
    wp[0] = 1.0
    with np.errstate(invalid="ignore", divide="ignore"):
        dph = np.abs(p.ph - o["ph"])
        dph = np.minimum(dph, np.abs(dph - 2 * np.pi))                # +pi and -pi are the same phase
        ph = (dph / wc)[v].max()
    mg = (np.abs(p.mag - o["mag"]) / np.maximum(smax[:, None], 1e-300))[v].max()
    # + the rounding of (dphase + wfbin) itself: wfbin reaches 2*pi*nfft/2, one ulp there is ~4e-12 rad
    fn = (np.maximum(np.abs(p.f - o["f"]) * (2 * np.pi * dt) - 2e-11, 0.0) / (wc + wp))[v].max()
    if ph > 1e-13 or mg > 1e-13 or fn > 1e-13:
        msgs.append("normalised errors: phase %.3g mag %.3g dphase %.3g" % (ph, mg, fn))
    return msgs


def check32(p, o, c, S):
    """float32 against the float64 oracle on the well-conditioned part (module docstring)."""
    msgs = []
    nfft, hop, sr, K = c["nfft"], c["hop"], c["sr"], c["K"]
    dt = hop / sr
    F = len(o["t"])
    smax = S.max(axis=1)
    nbad = nchk = 0
    worst = dict(f=0.0, ph=0.0, mag=0.0)
    for i in range(F):
        rb = o["binno"][i][o["f"][i] > 0]
        gb = p.binno[i][p.f[i] > 0]
        if smax[i] <= 0:
            if len(gb):
                msgs.append("frame %d silent but %d peaks" % (i, len(gb)))
            continue
        if smax[i] - S[i].min() <= 1e-6 * smax[i]:                    # flat spectrum, see check64
            continue
        gset = {int(b): j for j, b in enumerate(p.binno[i]) if p.f[i][j] > 0}
        for j in range(K):
            if not o["f"][i][j] > 0:
                continue
            b = int(o["binno"][i][j])
            strong = S[i, b] >= 1e-3 * smax[i] and o["mag"][i][j] >= 30 * c["thr"] * smax[i]
            prev_ok = i > 0 and S[i - 1, b] >= 1e-3 * smax[i - 1] and smax[i - 1] >= 1e-2 * smax[i]
            prev_zero = i == 0 or smax[i - 1] == 0
            if not strong or not (prev_ok or prev_zero):
                continue
            nchk += 1
            if b not in gset:
                # the float32 run took another bin instead: not an error when that bin's float64 magnitude equals this
                # one's to float32 resolution (a chirp's plateau: |X| of neighbouring bins within 1e-8 of each other --
                # which of them is "the" maximum is decided by rounding in any float32 transform; case 91:9632)
                oset = set(int(q) for q in rb)
                if any(q not in oset and abs(S[i, q] - S[i, b]) <= 1e-6 * S[i, b] for q in gset):
                    continue
                nbad += 1
                continue
            g = gset[b]
            df = abs(p.f[i][g] - o["f"][i][j])
            alias = sr / hop
            # dphase2freq keeps the unwrapping candidate nearest the bin centre (PV.py:144-145); a peak whose
            # frequency sits half an alias from the centre is a coin toss between two candidates in ANY
            # arithmetic: an exact one-alias difference there is not an error
            if abs(df - alias) < 1e-3 * alias and abs(abs(o["f"][i][j] - b * sr / nfft) - alias / 2) < 1e-3 * alias:
                continue
            worst["f"] = max(worst["f"], df * 2 * np.pi * dt)
            dph = abs(p.ph[i][g] - o["ph"][i][j])
            worst["ph"] = max(worst["ph"], min(dph, abs(dph - 2 * np.pi)))       # +pi and -pi are the same phase
            worst["mag"] = max(worst["mag"], abs(p.mag[i][g] - o["mag"][i][j]) / o["mag"][i][j])
    # a strong peak can only go missing when more than K strong peaks compete (rank decided by rounding)
    if nbad > 0.01 * nchk + (1 if K < 100 else 3):
        msgs.append("%d of %d strong peaks missing" % (nbad, nchk))
    if worst["f"] > 2e-3 or worst["ph"] > 1e-3 or worst["mag"] > 1e-3:
        msgs.append("strong-peak errors %s" % worst)
    tm = np.asarray(p.totalmag); tr = np.asarray(o["totalmag"])
    if len(tm) and np.max(np.abs(tm - tr) / np.maximum(tr, 1e-300)) > 1e-5:
        msgs.append("totalmag")
    return msgs, nchk, nbad, worst


def run_case(seed, idx, verbose=False, long=None):
    c = make_case(seed, idx, long)
    nfft, hop, K, thr, sr, x = c["nfft"], c["hop"], c["K"], c["thr"], c["sr"], c["x"]
    o = pvoracle.analyze(x, sr, nfft, hop, K, thr)
    F = len(o["t"])
    S = spectrogram(x, nfft, hop, F) if F else None
    fails = []
    stats = dict(runs=0, chk=0, bad=0, f=0.0, ph=0.0, mag=0.0)
    tag = "case %d:%d nfft=%d hop=%d K=%d thr=%g sr=%g n=%d kind=%d" % (seed, idx, nfft, hop, K, thr, sr, c["n"], c["kind"])
    p64 = None
    for prec in (64, 32):
        # (fft modes 1 and 3, the witness kernels, exist in tests/libpvx_witness.so only: PVX_LIB=tests/libpvx_witness.so python tools/fuzz.py)
        wit = [1, 3] if "witness" in os.path.basename(os.environ.get("PVX_LIB", "")) else []
        modes = [None] if prec == 64 else [0] + ([2] if nfft in (2048, 4096, 8192) else []) + (wit + [4] if nfft in (512, 1024, 2048) else []) + ([5] if nfft in (4096, 8192) and K <= 128 else [])
        for mode in modes:
            p = run_hip(c, prec, mode)
            stats["runs"] += 1
            if p.nframes != F:
                fails.append("%s prec=%d mode=%s: nframes %d != %d" % (tag, prec, mode, p.nframes, F)); continue
            if F == 0:
                continue
            if prec == 64:
                p64 = p
                m = check64(p, o, c, S)
            else:
                m, nchk, nbad, worst = check32(p, o, c, S)
                stats["chk"] += nchk; stats["bad"] += nbad
                for k in ("f", "ph", "mag"):
                    stats[k] = max(stats[k], worst[k])
            for s in m:
                fails.append("%s prec=%d mode=%s: %s" % (tag, prec, mode, s))
    if p64 is not None and F and not fails:
        ss = p64.toSinSum()
        pid, st, ln = ss.partial_table()
        opid, ost, oln = pvoracle.track(p64.f, p64.mag)
        P = len(st)
        if not (np.array_equal(pid, opid) and np.array_equal(st, ost[:P]) and np.array_equal(ln, oln[:P])):
            fails.append("%s: tracker differs" % tag)
        elif P and (ln >= 3).any() and nfft / hop <= 16:
            w = ss.synth(sr, c["h2"])
            ow = pvoracle.synth(p64.f, p64.mag, p64.realph, opid, ost, oln, sr, nfft, hop, c["h2"])
            err = np.abs(w - ow).max() if w.shape == ow.shape else np.inf
            if not err <= 1e-9 * max(1.0, np.abs(ow).max()):
                fails.append("%s: synth h2=%d err %g" % (tag, c["h2"], err))
    # PVHarmonic (float64) on the same signal with a random f0 track, against pvo_harmonic
    if F and not fails:
        rng = np.random.default_rng([seed, idx, 7])
        base = float(rng.uniform(2.0, 40.0)) * sr / nfft               # 2..40 bins
        f0 = base * (1 + 0.02 * rng.standard_normal(F))
        f0[rng.random(F) < 0.15] = 0.0
        f0[rng.random(F) < 0.05] = np.nan
        ph = pypevoc_amd.PVHarmonic(x, sr, nfft=nfft, hop=hop, npks=K, progress=False, precision=64)
        ph.set_f0(f0)
        ph.run_pv()
        oh = pvoracle.harmonic(x, sr, f0, nfft, hop, K)
        stats["runs"] += 1
        live = (S.max(axis=1) - S.min(axis=1)) > 1e-9 * S.max(axis=1)
        # frames with a flat spectrum (single non-zero windowed sample) are skipped as in check64: there
        # an exactly-zero real or imaginary part -- NaN through the x/0 rule -- is FFT rounding noise
        same0 = np.array_equal((ph.f == 0)[live], (oh["f"] == 0)[live]) and \
            np.array_equal(np.isnan(ph.f)[live], np.isnan(oh["f"])[live])
        if not same0:
            fails.append("%s: harmonic zero/NaN pattern differs" % tag)
        else:
            # a harmonic whose re-centred bin round(h*f1*nfft/sr) sits within rounding of .5 may flip
            # to the neighbouring bin: detect through the magnitude and bound the rate
            fmaxh = np.maximum(S.max(axis=1, keepdims=True), 1e-300)
            v = (oh["f"] != 0) & ~np.isnan(oh["f"]) & live[:, None]
            flip = v & (np.abs(ph.mag - oh["mag"]) > 1e-9 * fmaxh)
            if flip.sum() > 1e-3 * max(v.sum(), 1) + 1:
                fails.append("%s: %d of %d harmonics on a different bin" % (tag, flip.sum(), v.sum()))
            good = v & ~flip & (oh["mag"] >= 1e-6 * fmaxh)
            if good.any():
                dph = np.abs(ph.ph - oh["ph"])[good].max()
                if dph > 1e-8:
                    fails.append("%s: harmonic phase error %g" % (tag, dph))
            tot = (S ** 2).sum(axis=1)
            # the residual sums the 3-bin energies of ALL harmonics up to Nyquist, re-centred on the measured frequency f1
            # of the first one (PV.py:466-469): when that first harmonic is at the rounding floor of its frame (a chirp
            # with nothing at f0), f1 is numerical noise and which bin a harmonic beyond npks lands on is a coin toss
            # in any arithmetic -- those frames are not compared (case 202:20873: one of 106 harmonics on the next bin)
            fin = np.isfinite(oh["residuals"]) & np.isfinite(ph.residuals) & live & ~flip.any(axis=1)
            fin &= oh["mag"][:, 0] >= 1e-6 * fmaxh[:, 0]
            if fin.any():
                e = np.abs(ph.residuals[fin] ** 2 - oh["residuals"][fin] ** 2) / np.maximum(tot[fin], 1e-300)
                if e.max() > 1e-10:
                    fails.append("%s: harmonic residual error %g" % (tag, e.max()))
    if verbose:
        print(tag, stats)
    return fails, stats


def main():
    _lib.load(); _lib.init()
    if len(sys.argv) > 2 and sys.argv[1] == "--case":
        global VERBOSE
        VERBOSE = True
        s, i = sys.argv[2].split(":")
        fails, _ = run_case(int(s), int(i), verbose=True)
        print("\n".join(fails) if fails else "ok")
        return 1 if fails else 0
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t_end = time.time() + budget
    idx = nfail = 0
    tot = dict(runs=0, chk=0, bad=0, f=0.0, ph=0.0, mag=0.0)
    t_note = time.time() + 60.0
    while time.time() < t_end:
        if time.time() > t_note:                                    # a line a minute: long runs are not silent
            print("... %d cases, %d failing so far" % (idx, nfail), flush=True)
            t_note = time.time() + 60.0
        fails, st = run_case(seed, idx)
        for k in ("runs", "chk", "bad"):
            tot[k] += st[k]
        for k in ("f", "ph", "mag"):
            tot[k] = max(tot[k], st[k])
        for f in fails:
            print("FAIL", f)
        nfail += bool(fails)
        idx += 1
    print("fuzz seed %d: %d cases (%d HIP analyses), %d failing cases; float32 strong peaks checked %d, missing %d, "
          "worst errors: dphase %.3g rad, phase %.3g rad, mag rel %.3g"
          % (seed, idx, tot["runs"], nfail, tot["chk"], tot["bad"], tot["f"], tot["ph"], tot["mag"]))
    return 1 if nfail else 0


if __name__ == "__main__":
    sys.exit(main())
