"""Pins the CPU oracle (oracle/pvoracle.c) against the reference's own outputs.

The golden vectors were produced by importing goiosunsw/PyPeVoc itself
(tests/golden/make_golden.py).  Tolerances are float64 round-off only: the oracle restates
the same arithmetic; the FFT (pocketfft in the reference) is the only different routine.
"""
import os

import numpy as np
import pytest

from .conftest import GOLDEN, golden_names, load_golden

ANALYSIS = golden_names()
TRACKED = [n for n in ANALYSIS if not n.startswith("G9_")]


def _same_peaks(g, o):
    assert np.array_equal(g["binno"], o["binno"])
    assert np.array_equal(g["f"] > 0, o["f"] > 0)


@pytest.mark.parametrize("name", ANALYSIS)
def test_analysis_matches_reference(oracle, name):
    g = load_golden(name)
    o = oracle.analyze(g["x"], g["sr"], g["nfft"], g["hop"], g["npks"], g["pkthresh"], win=g.get("win"))
    assert o["f"].shape == g["f"].shape == (g["nframes"], g["npks"])
    _same_peaks(g, o)
    # stated float64 tolerances: |df| <= 1e-9 Hz, rel mag 1e-12 (+ a floor for
    # peaks at the 1e-5 threshold of the spectrum), phases 1e-10 rad away from the x/0 frames
    np.testing.assert_allclose(o["f"], g["f"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(o["mag"], g["mag"], rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(o["ph"], g["ph"], rtol=0, atol=2e-9)
    np.testing.assert_allclose(o["realph"], g["realph"], rtol=0, atol=2e-9)
    np.testing.assert_allclose(o["t"], g["t"], rtol=0, atol=0)
    np.testing.assert_allclose(o["totalmag"], g["totalmag"], rtol=1e-12, atol=1e-18)


@pytest.mark.parametrize("name", TRACKED)
def test_tracker_matches_reference(oracle, name):
    g = load_golden(name)
    # run the tracker on the REFERENCE's analysis arrays: isolates toSinSum
    pid, st, ln = oracle.track(g["f"], g["mag"])
    assert np.array_equal(st, g["part_start"])
    assert np.array_equal(ln, g["part_len"])
    assert np.array_equal(oracle.part_slots(pid, st, ln), g["part_slot"])


def test_tracker_exact_ties_follow_reference(oracle):
    """Fixture T1 (tests/golden/make_golden_ties.py): previous partials of equal magnitude, exactly equally far
    from a new peak -> the higher partial index wins (PVAnalysis.py:893), whichever slot it sits in."""
    g = dict(np.load(os.path.join(GOLDEN, "T1_tracker_ties.npz")))
    pid, st, ln = oracle.track(g["f"], g["mag"])
    assert np.array_equal(st, g["part_start"]) and np.array_equal(ln, g["part_len"])
    assert np.array_equal(oracle.part_slots(pid, st, ln), g["part_slot"])


@pytest.mark.parametrize("name", TRACKED)
def test_synth_matches_reference(oracle, name):
    g = load_golden(name)
    hops = [int(k[5:]) for k in g if k.startswith("w_hop")]
    if not hops:
        pytest.skip("no waveform in this fixture")
    pid, st, ln = oracle.track(g["f"], g["mag"])
    for h in hops:
        w = oracle.synth(g["f"], g["mag"], g["realph"], pid, st, ln, g["sr"], g["nfft"], g["hop"], h)
        ref = g["w_hop%d" % h]
        assert w.shape == ref.shape
        tol = 1e-10 if ref.dtype == np.float64 else 2e-7   # G7 waveform is stored as float32
        np.testing.assert_allclose(w, ref, rtol=0, atol=tol * max(1.0, np.abs(ref).max()))


def _synth_param_cases(g):
    for k in g:
        if k.startswith("w_") and not k.startswith("w_hop"):
            h, e100, mf = (int(v) for v in k[2:].split("_"))
            yield k, h, e100 / 100.0, mf


def test_synth_parameters_match_reference(oracle):
    """SinSum.synth's other parameters (fixture S1: edge 0 .. 2, minframes 1 .. 6, synthesis hops 128 / 256 / 300 on an
    analysis whose partials start and stop): the oracle against the reference's waveforms."""
    g = load_golden("S1_synth_params")
    pid, st, ln = oracle.track(g["f"], g["mag"])
    assert np.array_equal(st, g["part_start"]) and np.array_equal(ln, g["part_len"])
    n = 0
    for k, h, edge, mf in _synth_param_cases(g):
        w = oracle.synth(g["f"], g["mag"], g["realph"], pid, st, ln, g["sr"], g["nfft"], g["hop"], h, edge=edge, minframes=mf)
        ref = g[k]
        assert w.shape == ref.shape, (k, w.shape, ref.shape)
        np.testing.assert_allclose(w, ref, rtol=0, atol=1e-10 * max(1.0, np.abs(ref).max()), err_msg=k)
        n += 1
    assert n == 7


def test_tracker_jump_limit_matches_reference(oracle):
    """SinSum.add_frame's maxpitchjmp (fixture T2: 0.05 / 0.5 / 1.5 / 12 semitones on a gliding, vibrating signal: 577 / 181 / 9 / 5
    partials): the oracle's table against the reference's, limit by limit."""
    g = load_golden("T2_maxpitchjmp")
    seen = []
    for key in sorted(k[6:] for k in g if k.startswith("start_")):
        pid, st, ln = oracle.track(g["f"], g["mag"], maxpitchjmp=int(key) / 100.0)
        assert np.array_equal(st, g["start_" + key]) and np.array_equal(ln, g["len_" + key]), key
        assert np.array_equal(oracle.part_slots(pid, st, ln), g["slot_" + key]), key
        seen.append(len(st))
    assert sorted(seen) == [5, 9, 181, 577]


def test_g1_known_answer(oracle):
    """The by-eye known answer of the reference's tests/test_pypevoc.py."""
    g = load_golden("G1_two_sines")
    o = oracle.analyze(g["x"], g["sr"], g["nfft"], g["hop"], g["npks"], g["pkthresh"])
    pid, st, ln = oracle.track(o["f"], o["mag"])
    rows = []
    for p in range(len(st)):
        fr = np.arange(st[p], st[p] + ln[p])
        sl = [np.flatnonzero(pid[i] == p)[0] for i in fr]
        rows.append((int(st[p]), int(ln[p]), o["f"][fr, sl].mean(), o["mag"][fr, sl].mean()))
    exp = [(0, 1, 419.897461, 0.099773), (0, 85, 1199.688926, 0.049980), (1, 84, 400.000007, 0.099773)]
    assert len(rows) == 3
    for r, e in zip(rows, exp):
        assert r[0] == e[0] and r[1] == e[1]
        assert abs(r[2] - e[2]) < 1e-6 and abs(r[3] - e[3]) < 1e-6


def test_peakfinder_matches_reference(oracle):
    g = np.load(os.path.join(GOLDEN, "G8_peakfinder.npz"))
    ys = g["ys"].astype(np.float64)
    for k in (1, 3, 8, 100):
        for thr in (0.005, 0.2):
            tag = "k%d_t%s" % (k, str(thr).replace(".", "p"))
            for i, y in enumerate(ys):
                pos, keep = oracle.peakfinder(y, npeaks=k, minrattomax=thr, rad=5)
                n = int(g["cnt_" + tag][i])
                assert len(pos) == n
                assert np.array_equal(pos, g["pos_" + tag][i, :n])
                assert np.array_equal(keep, g["keep_" + tag][i, :n].astype(bool))
    # tests/test_peak_finder.py:16-20
    pos, keep = oracle.peakfinder(g["ramp"], rad=None)
    assert np.array_equal(pos[keep], g["ramp_pos"]) and list(pos) == [9]


def test_nframes_formula(oracle):
    # PV.py:224-225 frame count, checked against the reference's frame counts in the fixtures
    for name in ANALYSIS:
        g = load_golden(name)
        assert oracle.nframes(len(g["x"]), g["nfft"], g["hop"]) == g["nframes"]
    assert oracle.nframes(1024, 1024, 512) == 0
    assert oracle.nframes(1025, 1024, 512) == 1
    assert oracle.nframes(100, 1024, 512) == 0


# ------------------------------------------------------------------ PVHarmonic
HARMONIC = golden_names(prefix="H", exclude=())


@pytest.mark.parametrize("name", HARMONIC)
def test_harmonic_matches_reference(oracle, name):
    """pvo_harmonic against PVHarmonic.run_pv of the reference (tests/golden/make_golden_harmonic.py):
    f0-guided bins, stale previous spectrum over skipped frames, zero padding, NaN residuals."""
    g = load_golden(name)
    o = oracle.harmonic(g["x"], g["sr"], g["f0_used"], g["nfft"], g["hop"], g["npks"], g["fmin"])
    assert o["f"].shape == g["f"].shape == (g["nframes"], g["npks"])
    assert np.array_equal(o["t"], g["t"])
    assert np.array_equal(np.isnan(o["residuals"]), np.isnan(g["residuals"]))
    assert np.array_equal(o["f"] == 0, g["f"] == 0)
    assert np.nanmax(np.abs(o["f"] - g["f"])) <= 1e-9
    assert np.abs(o["mag"] - g["mag"]).max() <= 1e-14
    assert np.abs(o["ph"] - g["ph"]).max() <= 1e-10
    # residual = sqrt(total - harmonic energy): compare the squares (cancellation)
    fin = np.isfinite(g["residuals"])
    assert np.abs(o["residuals"][fin] ** 2 - g["residuals"][fin] ** 2).max() <= 1e-13


def test_harmonic_short_f0_is_index_error(oracle):
    g = load_golden("H3_readme_f0const")
    with pytest.raises(IndexError):
        oracle.harmonic(g["x"], g["sr"], g["f0_used"][:10], g["nfft"], g["hop"], g["npks"])


# ------------------------------------------------------------------ windowed reductions
def _w1():
    g = dict(np.load(os.path.join(GOLDEN, "W1_windowed.npz")))
    g["x"] = g["x"].astype(np.float64)
    return g


def test_windowed_reductions_match_reference(oracle):
    """pvo_heterodyne / pvo_rms_frames against Heterodyne.heterodyne and SoundUtils.RMSWind of the
    reference (tests/golden/make_golden_harmonic.py, W1).  numpy sums pairwise, the oracle left to right:
    tolerance 1e-13 of the window's mean amplitude scale."""
    g = _w1()
    hetsig = np.exp(-2j * np.pi * np.cumsum(g["het_fvec"]))
    h, ic = oracle.heterodyne(g["x"], hetsig, np.hanning(1024), 256)
    assert np.array_equal(ic, g["het_icent"])
    assert np.abs(np.stack([h.real, h.imag], axis=1) - g["het"]).max() <= 1e-13
    h2, ic2 = oracle.heterodyne(g["x"], hetsig, np.ones(256), 100)
    assert np.array_equal(ic2, g["het_rect_icent"])
    assert np.abs(np.stack([h2.real, h2.imag], axis=1) - g["het_rect"]).max() <= 1e-13
    assert np.abs(oracle.rms_frames(g["x"], np.blackman(1024), 512) - g["rms"]).max() <= 1e-14
    assert np.abs(oracle.rms_frames(g["x"], np.hanning(1000), 333) - g["rms_odd"]).max() <= 1e-14


def test_funcwind_matches_reference(oracle):
    """pvo_funcwind against SoundUtils.FuncWind of the reference with np.sum / mean / max / min / std / var (W2,
    tests/golden/make_golden_harmonic.py funcwind): real signal x power 0 / 1 / 2 and an odd window / hop, complex signal for
    sum / mean / std / var.  numpy sums pairwise, the oracle left to right: 1e-13 absolute on O(0.1) amplitudes; max / min exact."""
    g = dict(np.load(os.path.join(GOLDEN, "W2_funcwind.npz")))
    x = g["x"].astype(np.float64)
    for name in ("sum", "mean", "max", "min", "std", "var"):
        tol = 0.0 if name in ("max", "min") else 1e-13
        for power in (0, 1, 2):
            got = oracle.funcwind(name, x, np.blackman(1024), 512, power)
            assert got.shape == g["%s_p%d" % (name, power)].shape and np.abs(got - g["%s_p%d" % (name, power)]).max() <= tol, (name, power)
        got = oracle.funcwind(name, x, np.hanning(1000), 333, 1)
        assert np.abs(got - g["%s_odd" % name]).max() <= tol, name
    xc = x * np.exp(2j * np.pi * np.arange(len(x)) * 1000.0 / float(g["sr"]))
    for name in ("sum", "mean"):
        got = oracle.funcwind(name, xc, np.blackman(1024), 256, 1)
        assert np.abs(np.stack([got.real, got.imag], axis=1) - g["c_%s" % name]).max() <= 1e-13, name
    for name in ("std", "var"):
        assert np.abs(oracle.funcwind(name, xc, np.blackman(1024), 256, 1) - g["c_%s" % name]).max() <= 1e-13, name
