"""ctypes front end of the CPU oracle (oracle/pvoracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the checker / reported baseline.  The product package (pypevoc_amd)
never imports this module.  Parity of the oracle itself is pinned against the reference's
own outputs in tests/test_oracle_golden.py (fixtures: tests/golden/make_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    so = os.path.join(_HERE, "libpvoracle.so")
    src = os.path.join(_HERE, "pvoracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpvoracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libpvoracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.pvo_nframes.restype = ctypes.c_int64
        L.pvo_nframes.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
        L.pvo_analyze.restype = ctypes.c_int64
        L.pvo_analyze.argtypes = [_dp, ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_double, _dp] + [_dp] * 7
        L.pvo_harmonic.restype = ctypes.c_int64
        L.pvo_harmonic.argtypes = [_dp, ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   _dp, _dp, ctypes.c_int64, ctypes.c_double] + [_dp] * 5
        L.pvo_heterodyne.restype = ctypes.c_int64
        L.pvo_heterodyne.argtypes = [_dp, _dp, ctypes.c_int64, _dp, ctypes.c_int, ctypes.c_int, _dp,
                                     ctypes.POINTER(ctypes.c_int64)]
        L.pvo_rms_frames.restype = ctypes.c_int64
        L.pvo_rms_frames.argtypes = [_dp, ctypes.c_int64, _dp, ctypes.c_int, ctypes.c_int, _dp]
        L.pvo_funcwind.restype = ctypes.c_int64
        L.pvo_funcwind.argtypes = [_dp, ctypes.c_int, ctypes.c_int64, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, _dp]
        L.pvo_stft_frame.restype = ctypes.c_int
        L.pvo_stft_frame.argtypes = [_dp, ctypes.c_int64, ctypes.c_int, _dp, _dp, _dp]
        L.pvo_peakfinder.restype = ctypes.c_int
        L.pvo_peakfinder.argtypes = [_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                     ctypes.c_int, _ip, _ip]
        L.pvo_track.restype = ctypes.c_int64
        L.pvo_track.argtypes = [_dp, _dp, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                _i32p, _i32p, _i32p]
        L.pvo_synth_len.restype = ctypes.c_int64
        L.pvo_synth_len.argtypes = [_i32p, _i32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_double]
        L.pvo_synth.restype = ctypes.c_int
        L.pvo_synth.argtypes = [_dp, _dp, _dp, _i32p, ctypes.c_int64, ctypes.c_int, _i32p, _i32p,
                                ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                ctypes.c_int, ctypes.c_double, ctypes.c_int, _dp, ctypes.c_int64]
        _LIB = L
    return _LIB


def _d(a):
    return a.ctypes.data_as(_dp)


def _i32(a):
    return a.ctypes.data_as(_i32p)


def nframes(nsamp, nfft, hop):
    return int(lib().pvo_nframes(int(nsamp), int(nfft), int(hop)))


def analyze(x, sr, nfft=1024, hop=None, npks=20, pkthresh=0.005, win=None):
    """run_pv on float64 data.  Returns dict(f, mag, ph, realph, binno, t, totalmag)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    if hop is None:
        hop = int(nfft / 2)
    hop = int(hop)
    win = np.hanning(nfft) if win is None else win
    win = np.ascontiguousarray(win, dtype=np.float64)
    F = nframes(len(x), nfft, hop)
    out = {k: np.zeros((F, npks)) for k in ("f", "mag", "ph", "realph", "binno")}
    out["t"] = np.zeros(F)
    out["totalmag"] = np.zeros(F)
    r = lib().pvo_analyze(_d(x), len(x), float(sr), int(nfft), hop, int(npks), float(pkthresh), _d(win),
                          _d(out["f"]), _d(out["mag"]), _d(out["ph"]), _d(out["realph"]),
                          _d(out["binno"]), _d(out["t"]), _d(out["totalmag"]))
    if r != F:
        raise RuntimeError("pvo_analyze failed: %d" % r)
    return out


def harmonic(x, sr, f0, nfft=1024, hop=None, npks=20, fmin=30.0, win=None):
    """PVHarmonic(x, sr, nfft, hop, npks).set_f0(f0); run_pv().  Returns dict(f, mag, ph, residuals, t)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    f0 = np.ascontiguousarray(f0, dtype=np.float64)
    if hop is None:
        hop = int(nfft / 2)
    hop = int(hop)
    win = np.hanning(nfft) if win is None else win
    win = np.ascontiguousarray(win, dtype=np.float64)
    F = nframes(len(x), nfft, hop)
    out = {k: np.zeros((F, npks)) for k in ("f", "mag", "ph")}
    out["residuals"] = np.zeros(F)
    out["t"] = np.zeros(F)
    r = lib().pvo_harmonic(_d(x), len(x), float(sr), int(nfft), hop, int(npks), _d(win), _d(f0), len(f0), float(fmin),
                           _d(out["f"]), _d(out["mag"]), _d(out["ph"]), _d(out["residuals"]), _d(out["t"]))
    if r == -3:
        raise IndexError("f0 is shorter than the number of frames")
    if r != F:
        raise RuntimeError("pvo_harmonic failed: %d" % r)
    return out


def heterodyne(x, hetsig, wind, hop):
    """Heterodyne.heterodyne(x, hetsig, wind, hop) -> (complex amplitudes, centre samples)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    h = np.ascontiguousarray(hetsig, dtype=np.complex128)
    wind = np.ascontiguousarray(wind, dtype=np.float64)
    nfr = nframes(len(x), len(wind), hop)
    out = np.zeros(nfr, dtype=np.complex128)
    ic = np.zeros(nfr, dtype=np.int64)
    r = lib().pvo_heterodyne(_d(x), h.view(np.float64).ctypes.data_as(_dp), len(x), _d(wind), len(wind), int(hop),
                             out.view(np.float64).ctypes.data_as(_dp), ic.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    if r != nfr:
        raise RuntimeError("pvo_heterodyne: %d frames, expected %d" % (r, nfr))
    return out, ic


def rms_frames(x, wind, hop):
    """SoundUtils.RMSWind values for window `wind` and step `hop`."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    wind = np.ascontiguousarray(wind, dtype=np.float64)
    nfr = nframes(len(x), len(wind), hop)
    out = np.zeros(nfr)
    r = lib().pvo_rms_frames(_d(x), len(x), _d(wind), len(wind), int(hop), _d(out))
    if r != nfr:
        raise RuntimeError("pvo_rms_frames: %d frames, expected %d" % (r, nfr))
    return out


FUNCWIND_OPS = {"sum": 0, "mean": 1, "max": 2, "min": 3, "std": 4, "var": 5}


def funcwind(func, x, wind, hop, power=1):
    """SoundUtils.FuncWind(np.<func>, x, nwind=len(wind), nhop=hop, power=power) values (SoundUtils.py:42-69)."""
    x = np.asarray(x)
    cpx = np.iscomplexobj(x)
    x = np.ascontiguousarray(x, dtype=np.complex128 if cpx else np.float64)
    wind = np.ascontiguousarray(wind, dtype=np.float64)
    divisor = float(sum(wind ** power)) if power > 0 else 1.0
    nfr = nframes(len(x), len(wind), hop)
    op = FUNCWIND_OPS[func]
    out = np.zeros(nfr, dtype=np.complex128 if (cpx and op <= 1) else np.float64)
    r = lib().pvo_funcwind(x.view(np.float64).ctypes.data_as(_dp), int(cpx), len(x), _d(wind), len(wind), int(hop), op, divisor,
                           out.view(np.float64).ctypes.data_as(_dp))
    if r != nfr:
        raise RuntimeError("pvo_funcwind: %d frames, expected %d" % (r, nfr))
    return out


def stft_frame(x, pos, nfft, win=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    win = np.hanning(nfft) if win is None else win
    win = np.ascontiguousarray(win, dtype=np.float64)
    re = np.zeros(nfft // 2)
    im = np.zeros(nfft // 2)
    if lib().pvo_stft_frame(_d(x), int(pos), int(nfft), _d(win), _d(re), _d(im)) != 0:
        raise RuntimeError("pvo_stft_frame failed")
    return re + 1j * im


def peakfinder(y, npeaks=None, minrattomax=None, minval=None, rad=5):
    """PeakFinder(y, npeaks=, minrattomax=, minval=) then filter_by_salience(rad) (rad=None: no
    filter).  Returns (positions after findpos, keep mask)."""
    y = np.ascontiguousarray(np.squeeze(y), dtype=np.float64)
    n = len(y)
    kind, val = 0, 0.0
    if minrattomax is not None:
        kind, val = 1, float(minrattomax)
    elif minval is not None:
        kind, val = 2, float(minval)
    idx = np.zeros(max(n, 1), dtype=np.int32)
    keep = np.zeros(max(n, 1), dtype=np.int32)
    c = lib().pvo_peakfinder(_d(y), n, int(npeaks) if npeaks else 0, kind, val,
                             -1 if rad is None else int(rad),
                             idx.ctypes.data_as(_ip), keep.ctypes.data_as(_ip))
    if c < 0:
        raise RuntimeError("pvo_peakfinder failed")
    return idx[:c].copy(), keep[:c].astype(bool)


def track(f, mag, maxpitchjmp=0.5):
    """toSinSum: returns (partial_id[F,K], part_start[P], part_len[P])."""
    f = np.ascontiguousarray(f, dtype=np.float64)
    mag = np.ascontiguousarray(mag, dtype=np.float64)
    F, K = f.shape
    pid = np.full((F, K), -1, dtype=np.int32)
    st = np.zeros(max(F * K, 1), dtype=np.int32)
    ln = np.zeros(max(F * K, 1), dtype=np.int32)
    P = lib().pvo_track(_d(f), _d(mag), F, K, float(maxpitchjmp), _i32(pid), _i32(st), _i32(ln))
    return pid, st[:P].copy(), ln[:P].copy()


def part_slots(pid, st, ln):
    """CSR slot list in the layout of the golden fixtures (part_slot)."""
    F, K = pid.shape
    slots = []
    for p in range(len(st)):
        for j in range(ln[p]):
            s = np.flatnonzero(pid[st[p] + j] == p)
            assert len(s) == 1
            slots.append(s[0])
    return np.array(slots, dtype=np.int16)


def synth(f, mag, realph, pid, st, ln, sr, nfft, hop_analysis, hop_synth=None, edge=1.0, minframes=3):
    """SinSum.synth(sr, hop_synth, edge, minframes)."""
    f = np.ascontiguousarray(f, dtype=np.float64)
    mag = np.ascontiguousarray(mag, dtype=np.float64)
    realph = np.ascontiguousarray(realph, dtype=np.float64)
    pid = np.ascontiguousarray(pid, dtype=np.int32)
    st = np.ascontiguousarray(st, dtype=np.int32)
    ln = np.ascontiguousarray(ln, dtype=np.int32)
    F, K = f.shape
    if hop_synth is None:
        hop_synth = hop_analysis
    hop_synth = int(hop_synth)
    n = lib().pvo_synth_len(_i32(st), _i32(ln), len(st), int(nfft), int(hop_analysis), hop_synth, float(edge))
    if n < 0:
        raise ValueError("max() arg is an empty sequence")
    w = np.zeros(n)
    r = lib().pvo_synth(_d(f), _d(mag), _d(realph), _i32(pid), F, K, _i32(st), _i32(ln), len(st),
                        float(sr), int(nfft), int(hop_analysis), hop_synth, float(edge), int(minframes),
                        _d(w), n)
    if r != 0:
        raise RuntimeError("pvo_synth failed: %d" % r)
    return w
