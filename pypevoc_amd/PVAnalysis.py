"""Drop-in mirror of pypevoc.PVAnalysis for the hot path PV.run_pv -> PV.toSinSum -> SinSum.synth.

Same class names, constructor arguments, attributes and array layouts as the reference
(goiosunsw/PyPeVoc, pypevoc/PVAnalysis.py); the bodies of the hot methods are calls into
libpvx_hip.so (HIP kernels for MI355X, C ABI in include/pvx.h).  There is no numpy fallback for
those bodies: without the library / a GPU they raise.

What is host Python here is only what the reference also does once per object (constants of
PV.__init__, PVAnalysis.py:84-121) and the list-like views over device results.
"""
import ctypes
import os
import sys

import numpy as np

from . import _lib

pi2 = 2.0 * np.pi                      # PVAnalysis.py:45


def dpitch2st_exact(f1, f2):
    """PVAnalysis.py:55-59."""
    return 12 * np.log2(float(f2) / f1)


def dpitch2st(f1, f2):
    """PVAnalysis.py:62-68: approximate semitone interval used by the tracker."""
    return 17.312 * (float(f2) / f1 - 1.0)


_MISSING = object()


class _Result(object):
    """A result array of PV.run_pv that lives in the plan's resident block (HBM) until somebody touches it:
    the first read copies that one array to the host and caches it; assigning to the attribute (the
    reference's attributes are plain, writable ndarray members) makes the host value authoritative and
    takes the object off the device-resident toSinSum / synth / calc_f0 chain."""

    def __init__(self, name, which):
        self.key = "_res_" + name
        self.which = which

    def __get__(self, obj, cls):
        if obj is None:
            return self
        v = obj.__dict__.get(self.key, _MISSING)
        if v is _MISSING:
            v = obj._fetch_result(self.which)
            obj.__dict__[self.key] = v
            if isinstance(v, np.ndarray):
                # the caller gets a plain writable ndarray like the reference's attribute: an in-place edit
                # (pv.mag[pv.f > 5000] = 0) must take effect in toSinSum / synth / calc_f0, so the fetched
                # copy is fingerprinted and PV._on_device() compares before it trusts the arrays in HBM
                obj.__dict__.setdefault("_res_prints", {})[self.key] = _fingerprint(v)
        return v

    def __set__(self, obj, v):
        obj.__dict__[self.key] = v
        obj.__dict__["_results_edited"] = True


try:
    from xxhash import xxh3_64_intdigest as _digest          # optional: ~10 GB/s
except ImportError:                                          # pragma: no cover
    import hashlib

    def _digest(buf):                                        # standard library: ~1 GB/s, 64 bits
        return hashlib.blake2b(buf, digest_size=8).digest()


def _fingerprint(a):
    """Content hash (64 bits) of a fetched result array: only arrays the caller has read exist on the host, and only they
    are hashed again when toSinSum / synth / calc_f0 decide whether the results in HBM still are the results.  The
    `totalmag` list is not tracked: nothing on the device-resident chain reads it."""
    return _digest(memoryview(np.ascontiguousarray(a)).cast("B"))


_NP_WINDOWS = (np.hanning, np.hamming, np.blackman, np.bartlett)
_WIN_CACHE = {}


class _Plan(object):
    """Owner of a pvx_plan handle (constants, tables and every device / pinned buffer of the host entry points).

    Plans are pooled per parameter set: the reference's API makes one PV object per signal, and a plan's
    creation plus the first-call allocation of its buffers costs more than analysing a short signal.  A plan
    serves one PV at a time (`owner`: it may hold that PV's resident results); a PV whose plan is wanted by
    another one first brings its results to the host (PV._release_resident)."""

    _pool = {}
    _POOL_MAX = 4

    @classmethod
    def acquire(cls, owner, sr, nfft, hop, npks, pkthresh, win, precision, max_rows=0):
        import weakref
        win = np.ascontiguousarray(win, dtype=np.float64)
        # the rows hint sizes the general path's workspace (and its rocFFT batch): pool per power-of-two bucket
        bucket = 0
        if max_rows and max_rows > 0:
            bucket = 64
            while bucket < max_rows and bucket < (1 << 16):
                bucket <<= 1
        # a plan's buffers, streams and events live on the device it was created under, and the library refuses to run it from
        # a thread bound to another one (pvx_require_plan_device): the bound device is part of the key, so that after
        # `_lib.init(1)` a PV of the same parameters gets a plan of device 1, not the pooled one of device 0
        key = (_lib.init(), float(sr), int(nfft), int(hop), int(npks), float(pkthresh), int(precision), win.tobytes(), bucket,
               os.environ.get("PVX_FFT_MODE"), os.environ.get("PVX_MAX_ROWS"), os.environ.get("PVX_FUSED_BLOCKS"), os.environ.get("PVX_FPW"),
               os.environ.get("PVX_NO_STFT"), os.environ.get("PVX_NO_STFT_PV"), os.environ.get("PVX_NO_PV_REV"), os.environ.get("PVX_PV_TEAM"))
        plans = cls._pool.setdefault(key, [])
        free = [pl for pl in plans if pl.owner is None or pl.owner() is None]
        if free:
            pl = free[0]
        elif len(plans) < cls._POOL_MAX:
            pl = cls(sr, nfft, hop, npks, pkthresh, win, precision, max_rows=bucket)
            plans.append(pl)
        else:
            pl = plans[0]                                    # the oldest: its owner keeps its results on the host
            prev = pl.owner() if pl.owner is not None else None
            if prev is not None:
                prev._release_resident()
                prev._plan = None
                # the plan's progress callback is re-installed by its next owner: the old owner's thunk must
                # neither stay registered (wrong nsamp / hop, freed thunk) nor look "already installed" later
                prev._progress_cb = None
                prev._progress_plan = None
        plans.remove(pl) if pl in plans else None
        plans.append(pl)                                     # most recently used last
        pl.owner = weakref.ref(owner)
        if len(cls._pool) > 16:                              # bound the number of parameter sets kept alive
            for k in list(cls._pool)[:-16]:
                if all(q.owner is None or q.owner() is None for q in cls._pool[k]):
                    del cls._pool[k]
        return pl

    owner = None

    def __init__(self, sr, nfft, hop, npks, pkthresh, win, precision, max_rows=0):
        lib = _lib.load()
        _lib.init()
        h = ctypes.c_void_p()
        win = np.ascontiguousarray(win, dtype=np.float64)
        _lib.check(lib.pvx_plan_create(ctypes.byref(h), float(sr), int(nfft), int(hop), int(npks),
                                       float(pkthresh), _lib.dptr(win), int(precision), int(max_rows)),
                   "pvx_plan_create")
        self.handle = h
        self._lib = lib

    def __del__(self):
        try:
            if self.handle:
                self._lib.pvx_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class PV(object):
    """Phase vocoder (pypevoc.PVAnalysis.PV, PVAnalysis.py:71-417) on an MI355X.

    Precision follows the samples unless it is given: a float64 signal (what the reference's users hand over) is analysed in
    the reference's own float64 arithmetic end to end (`precision=64`: |df| <= 1e-9 Hz, identical partial tables), a float32
    or int16 signal in float32 on the device (`precision=32`, 2.7 x the frames per second: results are float64 arrays of
    float32-FFT accuracy -- the same peak bins, and on the fixtures |df| <= 1e-3 Hz, |dmag|/mag <= 1e-5, |dph| <= 2e-5 rad,
    waveform <= 1e-4 max|w|; on ill-conditioned material the bounds are the normalised ones of DESIGN.md section 4).  So
    switching the import alone reproduces the reference; `precision=32` on float64 samples buys the speed.

    Everything else follows the reference: constructor arguments, attributes (`f, mag, ph, realph, binno, t, totalmag,
    nframes, win, wfact, fstep, dt, fbin, wfbin, oldfft, ...`), array layouts, `run_pv() / toSinSum() / calc_f0() /
    calc_harmonic_power() / calc_fft_frame() / calc_pv_frame()`.  The result arrays live in GPU memory until they are read;
    assigning or editing one takes effect in the later steps exactly as with the reference's ndarrays."""
    f = _Result("f", 0)
    mag = _Result("mag", 1)
    ph = _Result("ph", 2)
    realph = _Result("realph", 3)
    binno = _Result("binno", 4)
    t = _Result("t", 5)
    totalmag = _Result("totalmag", 6)

    def __init__(self, x, sr, nfft=1024, hop=None, npks=20,
                 pkthresh=0.005, wind=np.hanning, progress=True, precision=None):
        '''
        Phase vocoder object (PVAnalysis.py:72-131).
        Arguments:
            * sr   = Sampling rate
            * nfft = Number of points in FFT analysis window
            * hop  = Number of points between FFT windows
            * npks = Maximum number of peaks at each frame
            * pkthresh = Threshold of peak amplitude relative of maximum
        Extra (not in the reference):
            * precision = 64: float64 end to end (the reference's arithmetic); 32: float32 frames/spectra on the
                          device (outputs float64, tolerances in DESIGN.md); None: 64 for float64 samples, 32 for
                          float32 / int16 samples
        '''
        self._xdev = None
        self._resident = False          # run_pv results are in the plan's resident block
        self._results_edited = False
        self._dependents = []           # weak references to SinSum objects that read the resident block
        if _lib.is_device_array(x):
            # Extension: a signal that is already in GPU memory (torch tensor on the GPU, anything with
            # __cuda_array_interface__) is analysed in place -- no host copy, no PCIe transfer of the input
            self._xdev = _lib.DeviceSignal(x)
            if len(self._xdev.shape) != 1:
                raise ValueError("PV expects a 1-D signal")
            self.x = x
            self.nsamp = self._xdev.shape[0]
        else:
            self.x = np.array(x)                               # PVAnalysis.py:84
            self.nsamp = len(self.x)
        self.sr = sr
        self.nfft = nfft
        self.nfft2 = int(nfft / 2)
        if hop is None:
            self.hop = int(self.nfft / 2)
        else:
            self.hop = hop
        self.peakthresh = pkthresh
        self.npeaks = npks
        self.nframes = 0
        if precision is None:
            dt_ = self._xdev.dtype if self._xdev is not None else self.x.dtype
            precision = 32 if np.dtype(dt_) in (np.dtype(np.float32), np.dtype(np.int16)) else 64
        self.precision = precision

        # numpy's own windows are functions of nfft alone: their array and the sums below are kept per (function, nfft);
        # every PV still gets its own copy of the window (callers may edit pv.win)
        ck = (wind, nfft) if wind in _NP_WINDOWS else None
        cached = _WIN_CACHE.get(ck) if ck is not None else None
        if cached is None:
            win = wind(nfft)
            # PVAnalysis.py:98-99 use Python's sum(): left-to-right float64 additions, which is what a cumulative
            # sum does too (np.sum would add pairwise and round differently)
            w = np.asarray(win, dtype=np.float64)
            wsum = np.cumsum(w)[-1] if len(w) else 0
            wsum2 = np.cumsum(w ** 2)[-1] if len(w) else 0
            cached = (win, wsum, wsum2)
            if ck is not None:
                if len(_WIN_CACHE) > 32:
                    _WIN_CACHE.clear()
                _WIN_CACHE[ck] = cached
            self.win = win if ck is None else win.copy()
        else:
            self.win = cached[0].copy()
        self.wsum, self.wsum2 = cached[1], cached[2]
        self.wfact = np.sqrt(self.wsum2 * self.nfft) / 2.0     # PVAnalysis.py:102
        self.fstep = float(self.sr) / float(self.nfft)
        self.dt = float(self.hop) / float(self.sr)
        self.fbin = np.arange(float(nfft)) * self.fstep
        dthetabin = pi2 * self.fbin * self.dt
        self.wfbin = np.round(dthetabin / pi2) * pi2           # PVAnalysis.py:118
        self.oldfft = np.zeros(self.nfft2)                     # PVAnalysis.py:121

        self.t = []
        self.f = []
        self.ph = []
        self.mag = []
        # PVAnalysis.py:128-131 builds a console progress display that prints once per frame.  Here
        # the analysis is a handful of kernel launches: libpvx_hip reports after every launch chunk
        # (pvx_plan_set_progress) and the same "cur / max (pct%)" line is printed in samples.
        # Extension: `progress` may be a callable(frames_done, frames_total).
        self.progress = progress if callable(progress) else bool(progress)
        self._plan = None
        self._progress_cb = None

    # ------------------------------------------------------------------ device plumbing
    def _get_plan(self, rows=None):
        if self._plan is None:
            if int(self.hop) != self.hop:
                raise TypeError("hop must be an integer number of samples")
            want = 0
            if rows is not None:
                want = max(2, int(rows))
            self._plan = _Plan.acquire(self, self.sr, self.nfft, int(self.hop), self.npeaks, self.peakthresh,
                                       self.win, self.precision, max_rows=want)
        self._install_progress(self._plan)
        return self._plan

    def _install_progress(self, plan):
        if not self.progress:
            _lib.check(_lib.load().pvx_plan_set_progress(plan.handle, _lib.PROGRESS_FN(), None), "pvx_plan_set_progress")
            plan.progress_owner = None
            return
        if self._progress_cb is not None and getattr(self, "_progress_plan", None) is plan \
                and getattr(plan, "progress_owner", None) is self._progress_cb:
            return
        user = self.progress if callable(self.progress) else None
        hop, nsamp = int(self.hop), int(self.nsamp)

        def report(done, total, _):
            if user is not None:
                user(int(done), int(total))
                return
            cur = nsamp if done >= total else min(int(done) * hop, nsamp)      # PVAnalysis.py:249-254
            pct = cur / float(max(nsamp, 1)) * 100
            print('\r' + '%d / %d (%.2f%%)' % (cur, nsamp, pct), end=" ")      # ProgressDisplay.py:95-101
            sys.stdout.flush()

        self._progress_cb = _lib.PROGRESS_FN(report)                           # keep the thunk alive
        self._progress_plan = plan
        plan.progress_owner = self._progress_cb                                # the thunk lives as long as the plan uses it
        _lib.check(_lib.load().pvx_plan_set_progress(plan.handle, self._progress_cb, None), "pvx_plan_set_progress")

    def _signal(self):
        if self._xdev is not None:
            # the per-frame / streaming helpers work on host copies (they are CPU-side conveniences)
            import torch
            return _lib.as_signal(torch.as_tensor(self.x).cpu().numpy())
        if self.x.ndim != 1:
            raise ValueError("PV expects a 1-D signal")
        return _lib.as_signal(self.x)

    def _fetch_result(self, which):
        """One resident result array -> host (PV.py:256-264 layouts: (F, K) float64, t (F,), totalmag a list)."""
        if not self._resident:
            raise AttributeError("no analysis results yet (run_pv)")
        F, K = self.nframes, self.npeaks
        a = np.empty((F, K) if which < 5 else (F,))
        _lib.check(_lib.load().pvx_resident_fetch(self._plan.handle, which, _lib.dptr(a)), "pvx_resident_fetch")
        return list(a) if which == 6 else a

    def _on_device(self):
        """True while the results of run_pv are in HBM and nobody has replaced or edited them on the host.

        The reference's f / mag / ph / realph are plain ndarrays that toSinSum (PVAnalysis.py:319) and calc_f0
        (PVAnalysis.py:379) read at call time, so both an assignment (`pv.mag = ...`, _Result.__set__) and an
        in-place edit of an array the caller has fetched (`pv.mag[pv.f > 5000] = 0`) take the object off the
        device-resident chain: the host copies then are the results."""
        if not self._resident or self._results_edited:
            return False
        for key, fp in self.__dict__.get("_res_prints", {}).items():
            v = self.__dict__.get(key, _MISSING)
            if isinstance(v, np.ndarray) and _fingerprint(v) != fp:
                self._results_edited = True
                return False
        return True

    def _release_resident(self):
        """Before the plan's resident block is overwritten: bring what is still only there to the host."""
        if not self._resident:
            return
        for d in self._dependents:
            ss = d()
            if ss is not None:
                ss._detach_resident()
        self._dependents = []
        for name in ("f", "mag", "ph", "realph", "binno", "t", "totalmag"):
            getattr(self, name)
        self._resident = False

    def _run_pv_device(self, F, K, prev0):
        """run_pv for a device-resident signal: pvx_analyze_dev on torch's current stream."""
        lib = _lib.load()
        _lib.init()
        import torch
        dprev = torch.from_numpy(prev0).to("cuda") if prev0 is not None else None
        n = F * K
        plan = self._get_plan(rows=F + 1)

        def launch(o, stream):
            ptrs = [ctypes.c_void_p(o + i * n * 8) for i in range(5)]
            r = lib.pvx_analyze_dev(plan.handle, ctypes.c_void_p(self._xdev.ptr), self._xdev.dtype_code, self.nsamp, 1,
                                    self.nsamp, *ptrs, ctypes.c_void_p(o + 5 * n * 8), ctypes.c_void_p(o + 5 * n * 8 + F * 8),
                                    ctypes.c_void_p(dprev.data_ptr()) if dprev is not None else None, stream)
            _lib.check(r, "pvx_analyze_dev")

        blk = _lib.device_run((5 * n + 2 * F) * 8, launch)
        arrs = [blk[i * n:(i + 1) * n].reshape(F, K).copy() for i in range(5)]
        return arrs, blk[5 * n:5 * n + F].copy(), blk[5 * n + F:].copy()

    # ------------------------------------------------------------------ reference API
    def dphase2freq(self, dph, nbin):
        '''
        "Instantaneous frequency" for the phase difference dph between two consecutive frames
        (PVAnalysis.py:133-148).  Scalar helper for streaming users; run_pv evaluates the same
        three candidates per peak on the device.
        '''
        dphw = dph + self.wfbin[nbin] + pi2 * np.arange(-1, 2)
        freq = dphw / self.dt / pi2
        df = self.fbin[nbin] - freq
        ii = np.argmin(abs(df))
        return freq[ii], df[ii]

    def calc_fft_frame(self, pos):
        '''Windowed, normalised FFT of the frame at pos (PVAnalysis.py:150-158), length nfft.'''
        lib = _lib.load()
        x, dt = self._signal()
        nb = self.nfft // 2 + 1
        spec = np.zeros((1, nb, 2))
        posa = np.array([int(pos)], dtype=np.int64)
        _lib.check(lib.pvx_stft_frames(self._get_plan().handle, x.ctypes.data_as(ctypes.c_void_p), dt,
                                       len(x), posa.ctypes.data_as(_lib.c_int64_p), 1, _lib.dptr(spec)),
                   "pvx_stft_frames")
        half = spec[0, :, 0] + 1j * spec[0, :, 1]
        full = np.zeros(self.nfft, dtype=complex)
        full[:nb] = half
        # conjugate mirror of a real signal's spectrum
        full[nb:] = np.conj(half[1:self.nfft - nb + 1][::-1])
        return full

    def calc_pv_frame(self, pos):
        '''
        Peaks and frequencies of the frame at pos, based on the previous frame kept in
        self.oldfft (PVAnalysis.py:160-211).  Returns f, mag, ph, realph, binno, totalmag.
        '''
        lib = _lib.load()
        x, dt = self._signal()
        pos = int(pos)
        seg = x[pos:pos + self.nfft]
        if len(seg) < self.nfft:
            raise ValueError("operands could not be broadcast together: frame at %d leaves the signal" % pos)
        seg = np.concatenate([seg, np.zeros(1, dtype=seg.dtype)])   # one frame: nsamp = nfft + 1
        K = self.npeaks
        out = [np.zeros((1, K)) for _ in range(5)]
        t = np.zeros(1)
        tm = np.zeros(1)
        old = np.asarray(self.oldfft, dtype=complex)
        prev0 = np.ascontiguousarray(np.stack([old.real, old.imag], axis=1), dtype=np.float64)
        last = np.zeros((self.nfft2, 2))
        F = lib.pvx_analyze(self._get_plan().handle, seg.ctypes.data_as(ctypes.c_void_p), dt, len(seg), 1, len(seg),
                            *[_lib.dptr(a) for a in out], _lib.dptr(t), _lib.dptr(tm), _lib.dptr(prev0),
                            _lib.dptr(last))
        _lib.check(F, "pvx_analyze")
        f, mag, ph, realph, binno = [a[0] for a in out]
        n = int(np.count_nonzero(f > 0))
        self.oldfft = last[:, 0] + 1j * last[:, 1]              # PVAnalysis.py:209
        return (list(f[:n]), list(mag[:n]), list(ph[:n]), list(realph[:n]),
                [int(b) for b in binno[:n]], float(tm[0]))

    def run_pv(self):
        '''The analysis loop (PVAnalysis.py:213-264) as one call into the HIP library.'''
        lib = _lib.load()
        x, dt = (None, None) if self._xdev is not None else self._signal()
        K = self.npeaks
        F = int(lib.pvx_nframes(self.nsamp, self.nfft, int(self.hop)))
        if F == 0:
            # the reference ends with empty arrays (np.array([]))
            self.f = np.array([]); self.mag = np.array([]); self.ph = np.array([])
            self.realph = np.array([]); self.binno = np.array([]); self.t = np.array([])
            self.nframes = 0
            self.totalmag = []
            return
        last = np.zeros((self.nfft2, 2))
        old = np.asarray(self.oldfft)
        prev0 = None
        if np.any(old != 0):                                     # run_pv after manual calc_pv_frame calls
            oc = old.astype(complex)
            prev0 = np.ascontiguousarray(np.stack([oc.real, oc.imag], axis=1), dtype=np.float64)
        self._release_resident()
        if self._xdev is not None:
            (f, mag, ph, realph, binno), t, tm = self._run_pv_device(F, K, prev0)
            # PV.oldfft after the loop = spectrum of the last frame (PVAnalysis.py:209): one frame's worth
            # of samples comes back to the host for it
            import torch
            pos = (F - 1) * int(self.hop)
            tail = torch.as_tensor(self.x)[pos:pos + self.nfft].cpu().numpy()
            tail_sig, tail_dt = _lib.as_signal(tail)
            posa = np.zeros(1, dtype=np.int64)
            spec = np.zeros((1, self.nfft2 + 1, 2))
            _lib.check(lib.pvx_stft_frames(self._get_plan().handle, tail_sig.ctypes.data_as(ctypes.c_void_p), tail_dt,
                                           len(tail_sig), posa.ctypes.data_as(_lib.c_int64_p), 1, _lib.dptr(spec)),
                       "pvx_stft_frames")
            last = spec[0, :self.nfft2, :]
            self.f = f
            self.mag = mag
            self.ph = ph
            self.realph = realph
            self.binno = binno
            self.t = t
            self.totalmag = list(tm)                             # PVAnalysis.py:264 (a Python list)
        else:
            # The results stay in HBM (the plan's resident block): toSinSum, SinSum.synth, calc_f0 and
            # calc_harmonic_power run there; an attribute comes to the host when it is first read.
            plan = self._get_plan(rows=F + 1)
            r = lib.pvx_analyze_resident(plan.handle, x.ctypes.data_as(ctypes.c_void_p), dt, len(x), 1, len(x),
                                         _lib.dptr(prev0) if prev0 is not None else None, _lib.dptr(last))
            _lib.check(r, "pvx_analyze_resident")
            for name in ("f", "mag", "ph", "realph", "binno", "t", "totalmag"):
                self.__dict__.pop("_res_" + name, None)
            self.__dict__.pop("_res_prints", None)
            self._resident = True
        self._results_edited = self._xdev is not None
        self.nframes = F
        self.oldfft = last[:, 0] + 1j * last[:, 1]

    def calc_harmonic_power(self, f_threshold=0.01):
        """
        Harmonic power of the individual sine components (PVAnalysis.py:266-297), vectorised over
        frames.  For each valid peak j of a frame, the peaks c whose frequency is within
        `f_threshold` (relative) of an integer multiple of f_j form its harmonic set; nharmonics is
        their number.  NB the reference takes `valid_mag = self.mag[valid_idx]` (:278), i.e. the ROWS
        of mag numbered like the valid peak slots, so its hpower sums whole rows
        mag[slot_c, :]**2 over the harmonic set -- reproduced as is (IndexError like the reference
        when a valid slot index >= number of frames).
        """
        if self._on_device() and self.nframes > 0:
            # on the arrays where they are (k_desc.hip): only hpower and nharmonics cross PCIe
            F, K = self.nframes, self.npeaks
            hpower = np.empty((F, K)); nharm = np.empty((F, K))
            rc = _lib.load().pvx_harmonic_power_resident(self._plan.handle, float(f_threshold), _lib.dptr(hpower), _lib.dptr(nharm))
            if rc == _lib.PVX_ERR_SIZE:
                raise IndexError(_lib.load().pvx_last_error().decode())
            _lib.check(rc, "pvx_harmonic_power_resident")
            self.hpower = hpower
            self.nharmonics = nharm
            return
        ff = np.asarray(self.f, dtype=np.float64)
        mm = np.asarray(self.mag, dtype=np.float64)
        F, K = ff.shape
        hpower = np.zeros((F, K))
        nharm = np.zeros((F, K))
        valid = ff > 0
        if valid.any():
            top = int(np.max(np.nonzero(valid.any(axis=0))[0]))
            if top >= mm.shape[0]:
                raise IndexError("index %d is out of bounds for axis 0 with size %d" % (top, mm.shape[0]))
            rowpow = np.zeros(K)
            rowpow[: top + 1] = np.sum(mm[: top + 1] ** 2, axis=1)
            step = max(1, int(4e6 // max(K * K, 1)))                  # bound the (frames, K, K) temporaries
            with np.errstate(divide="ignore", invalid="ignore"):
                for a in range(0, F, step):
                    fa = ff[a:a + step]
                    va = valid[a:a + step]
                    ratio = fa[:, None, :] / fa[:, :, None]           # [frame, j, c] = f_c / f_j
                    hn = np.round(ratio)
                    hn[hn == 0] = 1
                    inh = np.abs(ratio / hn - 1)                      # |f_c / n / f_j - 1|
                    comp = (inh < f_threshold) & va[:, None, :] & va[:, :, None]
                    nharm[a:a + step] = comp.sum(axis=2)
                    hpower[a:a + step] = (comp * rowpow[None, None, :]).sum(axis=2)
        self.hpower = hpower
        self.nharmonics = nharm

    def toSinSum(self, maxpitchjmp=0.5):
        '''
        Convert to Sine sum (PVAnalysis.py:299-322).
        NB: as in the reference, `maxpitchjmp` is accepted but NOT forwarded (PVAnalysis.py:320-321
        call add_frame without it), so tracks are always built with the 0.5-semitone default.
        '''
        ss = SinSum(self.sr, nfft=self.nfft, hop=self.hop)
        ss._precision = int(self.precision)                      # (32: SinSum.synth's sample loop in float32, like the analysis)
        if self.nframes > 0:
            if self._on_device():
                ss._from_resident(self)                          # tracker on the arrays in HBM, table stays there
            else:
                ss._from_analysis(self.f, self.mag, self.ph, self.realph)
        return ss

    def get_time_vector(self):
        return self.t

    def get_sample_vector(self):
        return (self.t * self.sr).astype('int')

    def calc_f0(self, fmin=50, fmax=10000, thr=0.1):
        """
        Lowest-frequency strong peak per frame (PVAnalysis.py:371-391): one small kernel over the resident
        (F, K) arrays (only the F frequencies and indices come back), or vectorised numpy when the arrays
        were replaced on the host.
        """
        if self._on_device() and self.nframes > 0:
            F = self.nframes
            fm = np.empty(F)
            im = np.empty(F, dtype=np.int32)
            _lib.check(_lib.load().pvx_f0_resident(self._plan.handle, float(fmin), float(fmax), float(thr), _lib.dptr(fm),
                                                   im.ctypes.data_as(_lib.c_int32_p)), "pvx_f0_resident")
            self.fundamental_idx = im.astype('i')
            return fm
        ff = np.asarray(self.f)
        mm = np.asarray(self.mag)
        fm = np.zeros(ff.shape[0])
        im = np.zeros(ff.shape[0], dtype='i')
        if ff.size:
            maxmag = mm.max(axis=1, keepdims=True)
            ok = (ff > fmin) & (ff < fmax) & (mm > maxmag * thr)
            cand = np.where(ok, ff, np.inf)
            isel = np.argmin(cand, axis=1)
            has = ok.any(axis=1)
            rows = np.arange(ff.shape[0])
            fm[has] = ff[rows[has], isel[has]]
            im[has] = isel[has]
        self.fundamental_idx = im
        return fm

    @property
    def fundamental_frequency(self):
        try:
            return self.f[np.arange(self.f.shape[0]), self.fundamental_idx]
        except AttributeError:
            return self.calc_f0()

    @property
    def fundamental_magnitude(self):
        try:
            return self.mag[np.arange(self.f.shape[0]), self.fundamental_idx]
        except AttributeError:
            self.calc_f0()
            return self.mag[np.arange(self.f.shape[0]), self.fundamental_idx]

    @property
    def partial_sum_magnitude(self):
        return np.sqrt(np.sum(self.mag ** 2, axis=1))

    @property
    def partial_magnitude_ratio(self):
        return self.partial_sum_magnitude / self.totalmag


class PVHarmonic(PV):
    """Phase vocoder sampled at the multiples of a given fundamental (PVAnalysis.py:419-535):
    `set_f0(f0, t)` then `run_pv()` -> f, mag, ph ((F, npks): first npks harmonics), residuals (F,), t.
    Runs as window -> rocFFT -> one harmonic kernel in libpvx_hip (pvx_harmonic_analyze)."""

    def __init__(self, *args, **kwargs):
        self.fmin = 30.0                                        # PVAnalysis.py:421
        PV.__init__(self, *args, **kwargs)
        self._frame_plan = None

    def set_f0(self, f0, t=None):
        '''
        Assign a f0 vector to the search (PVAnalysis.py:424-440)
        Argument:
            * f0: f0 vector over time
            * t: if present, values of time corresponding to f0
                 otherwise, the time values correspond to the hop size
        '''
        tint = np.arange(round(self.hop + self.nfft / 2), len(self.x), self.hop) / float(self.sr)
        if t is None:
            self.f0 = f0
        else:
            self.f0 = np.interp(tint, t, f0)

    def _prev0(self):
        old = np.asarray(self.oldfft)
        if not np.any(old != 0):
            return None
        oc = old.astype(complex)
        return np.ascontiguousarray(np.stack([oc.real, oc.imag], axis=1), dtype=np.float64)

    def calc_pv_frame(self, pos, f0):
        '''
        Harmonics of f0 in the frame at pos, based on the previous frame kept in self.oldfft
        (PVAnalysis.py:442-491).  Returns f, mag, ph (ALL multiples of f0 below nfft/2 - 1), residual.
        '''
        lib = _lib.load()
        x, dt = self._signal()
        pos = int(pos)
        seg = x[pos:pos + self.nfft]
        if len(seg) < self.nfft:
            raise ValueError("operands could not be broadcast together: frame at %d leaves the signal" % pos)
        seg = np.concatenate([seg, np.zeros(1, dtype=seg.dtype)])    # one frame: nsamp = nfft + 1
        f0bin = f0 / self.sr * self.nfft
        nh = len(np.arange(f0bin, self.nfft2 - 1, f0bin)) if f0 > 0 else 0
        kmax = 2 * self.nfft2 + 2                                    # f0 >= half a bin: never more harmonics
        if self._frame_plan is None:
            self._frame_plan = _Plan(self.sr, self.nfft, int(self.hop), kmax, self.peakthresh, self.win,
                                     self.precision, max_rows=2)
        out = [np.zeros((1, kmax)) for _ in range(3)]
        res = np.zeros(1)
        last = np.zeros((self.nfft2, 2))
        prev0 = self._prev0()
        f0a = np.array([f0], dtype=np.float64)
        F = lib.pvx_harmonic_analyze(self._frame_plan.handle, seg.ctypes.data_as(ctypes.c_void_p), dt, len(seg),
                                     _lib.dptr(f0a), 1, float(self.fmin), *[_lib.dptr(a) for a in out],
                                     _lib.dptr(res), None, _lib.dptr(prev0) if prev0 is not None else None,
                                     _lib.dptr(last))
        _lib.check(F, "pvx_harmonic_analyze")
        self.oldfft = last[:, 0] + 1j * last[:, 1]                   # PVAnalysis.py:491
        f, mag, ph = [list(a[0, :nh]) for a in out]
        return f, mag, ph, float(res[0])

    def run_pv(self):
        '''The analysis loop (PVAnalysis.py:493-535) as one call into the HIP library.'''
        lib = _lib.load()
        x, dt = self._signal()
        K = self.npeaks
        F = int(lib.pvx_nframes(self.nsamp, self.nfft, int(self.hop)))
        if F == 0:
            self.f = np.array([]); self.mag = np.array([]); self.ph = np.array([])
            self.residuals = np.array([]); self.t = np.array([])
            self.nframes = 0
            return
        self._release_resident()
        f0 = np.ascontiguousarray(np.asarray(self.f0, dtype=np.float64).ravel())
        if len(f0) < F:
            # the reference indexes self.f0[int(curpos / hop)] (PVAnalysis.py:507)
            raise IndexError("index %d is out of bounds for axis 0 with size %d" % (len(f0), len(f0)))
        f = np.empty((F, K)); mag = np.empty((F, K)); ph = np.empty((F, K))
        res = np.empty(F); t = np.empty(F)
        last = np.zeros((self.nfft2, 2))
        prev0 = self._prev0()
        plan = self._get_plan(rows=F + 1)
        r = lib.pvx_harmonic_analyze(plan.handle, x.ctypes.data_as(ctypes.c_void_p), dt, len(x), _lib.dptr(f0), len(f0),
                                     float(self.fmin), _lib.dptr(f), _lib.dptr(mag), _lib.dptr(ph), _lib.dptr(res),
                                     _lib.dptr(t), _lib.dptr(prev0) if prev0 is not None else None, _lib.dptr(last))
        _lib.check(r, "pvx_harmonic_analyze")
        self.f = f
        self.mag = mag
        self.ph = ph
        self.residuals = res
        self.t = t
        self.nframes = F
        self.oldfft = last[:, 0] + 1j * last[:, 1]


class RegPartial(object):
    def __init__(self, istart, pdict=None, overlap=0.5, fstep=None):
        '''
        A quasi-sinusoidal partial with homogeneous sampling (PVAnalysis.py:585-614).
        '''
        self.start_idx = istart
        self.overlap = overlap
        self.fstep = fstep
        if pdict is None:
            self.f = []
            self.mag = []
            self.ph = []
            self.realph = []
        else:
            self.f = pdict['f']
            self.mag = pdict['mag']
            self.ph = pdict['ph']
            try:
                self.realph = pdict['realph']
            except KeyError:
                self.realph = pdict['ph']

    def append_point(self, f, mag, ph, realph=None):
        '''Add a single point to the end of partial (PVAnalysis.py:616-626).'''
        self.f.append(f)
        self.mag.append(mag)
        self.ph.append(ph)
        if realph is None:
            self.realph.append(ph)
        else:
            self.realph.append(realph)

    def get_freq_at_frame(self, fr):
        relidx = fr - self.start_idx
        if relidx >= 0:
            return self.f[relidx]
        else:
            return np.nan

    def get_mag_at_frame(self, fr):
        relidx = fr - self.start_idx
        if relidx >= 0:
            return self.mag[relidx]
        else:
            return np.nan

    def synth(self, sr, hop, intermediate=False, edge=.5):
        '''
        Phase-preserving resynthesis of this partial (PVAnalysis.py:684-756) on the device.
        Returns (signal, first sample index).  fstep=None: no frequency-slope phase correction
        (PVAnalysis.py:710-713); SinSum always sets fstep (PVAnalysis.py:824-825).
        '''
        if intermediate:
            raise NameError("name 'phsig' is not defined")      # what the reference raises (PVAnalysis.py:754)
        hop = int(hop)
        nfr = len(self.f)
        dfr = 1. / self.overlap / 2.
        edgsam = int(dfr * hop * edge)
        flags = 0
        if self.fstep is None:
            # only the overlap matters: any integer pair with hop_a / nfft == overlap
            from fractions import Fraction
            fr = Fraction(self.overlap).limit_denominator(1 << 20)
            hop_a, nfft = fr.numerator, fr.denominator
            flags = _lib.PVX_SYNTH_NO_PHCOR
            if hop_a <= 0 or hop_a / float(nfft) != self.overlap:
                raise NotImplementedError("overlap %r is not a ratio of integers" % (self.overlap,))
        else:
            # derive (nfft, hop_analysis) with hop_a/nfft == overlap and sr/nfft == fstep
            nfft = int(round(sr / float(self.fstep)))
            hop_a = int(round(self.overlap * nfft))
            if abs(hop_a / float(nfft) - self.overlap) > 1e-15 or abs(sr / float(nfft) - self.fstep) > 1e-9 * self.fstep:
                raise NotImplementedError("overlap/fstep do not correspond to integer nfft and hop")
        pad = (edgsam + hop - 1) // hop + 1                     # frames of head room so the attack is not clipped
        F = pad + nfr
        f = np.zeros((F, 1)); mag = np.zeros((F, 1)); rp = np.zeros((F, 1))
        pid = np.full((F, 1), -1, dtype=np.int32)
        f[pad:, 0] = self.f; mag[pad:, 0] = self.mag; rp[pad:, 0] = self.realph
        pid[pad:, 0] = 0
        st = np.array([pad], dtype=np.int32)
        ln = np.array([nfr], dtype=np.int32)
        w = _device_synth(f, mag, rp, pid, st, ln, sr, nfft, hop_a, hop, edge, 1, flags)
        a = pad * hop - edgsam
        return w[a:a + hop * nfr + 2 * edgsam].copy(), int((self.start_idx) * hop - edgsam)


def _device_synth(f, mag, realph, pid, st, ln, sr, nfft, hop_a, hop_s, edge, minframes, flags=0):
    lib = _lib.load()
    _lib.init()
    F, K = f.shape
    maxend = int((st.astype(np.int64) + ln - 1).max())
    n = lib.pvx_synth_len(maxend, int(nfft), int(hop_a), int(hop_s), float(edge))
    _lib.check(n, "pvx_synth_len")
    w = np.zeros(n)
    f = np.ascontiguousarray(f, dtype=np.float64)
    mag = np.ascontiguousarray(mag, dtype=np.float64)
    realph = np.ascontiguousarray(realph, dtype=np.float64)
    pid = np.ascontiguousarray(pid, dtype=np.int32)
    st = np.ascontiguousarray(st, dtype=np.int32)
    ln = np.ascontiguousarray(ln, dtype=np.int32)
    i32 = lambda a: a.ctypes.data_as(_lib.c_int32_p)
    _lib.check(lib.pvx_synth_flags(_lib.dptr(f), _lib.dptr(mag), _lib.dptr(realph), i32(pid), F, K, i32(st), i32(ln),
                                   len(st), float(sr), int(nfft), int(hop_a), int(hop_s), float(edge), int(minframes),
                                   _lib.dptr(w), n, int(flags)), "pvx_synth")
    return w


class _PartialList(list):
    """The `partial` list of a SinSum; remembers whether user code changed it."""
    pass


class SinSum(object):
    def __init__(self, sr, nfft=1024, hop=512):
        '''
        Sine sum object (PVAnalysis.py:797-817): a sound decomposed in a sum of quasi-sine waves.
        '''
        self._partial = []
        self._st = []
        self._end = []
        self.nfft = nfft
        self.hop = hop
        self.sr = sr
        # device-side table (set by PV.toSinSum): analysis arrays + partial ids
        self._tab = None
        self._materialised = True
        self._rpv = None            # the PV whose resident results (and resident partial table) this object reads
        self._precision = 64        # 32 for the SinSum of a precision-32 analysis: the resynthesis' sample loop in float32 (tolerance 1e-4 max|w|)
        self._rP = 0
        self._rmaxend = -1

    # ---- table built by the tracker kernels ------------------------------------------------
    def _from_resident(self, pv, maxpitchjmp=0.5):
        """PV.toSinSum on results that are still in HBM: the tracker runs there and its table stays there."""
        import weakref
        me = ctypes.c_int64(-1)
        P = _lib.load().pvx_track_resident(pv._plan.handle, float(maxpitchjmp), ctypes.byref(me))
        _lib.check(P, "pvx_track_resident")
        self._rpv, self._rP, self._rmaxend = pv, int(P), int(me.value)
        self._precision = int(getattr(pv, "precision", 64))
        self._tab = None
        self._materialised = False
        pv._dependents.append(weakref.ref(self))

    def _detach_resident(self):
        """Copy the resident table (and, through the PV's attributes, the arrays) to the host."""
        pv = self._rpv
        if pv is None:
            return
        F, K, P = pv.nframes, pv.npeaks, self._rP
        pid = np.empty((F, K), dtype=np.int32)
        st = np.empty(max(P, 1), dtype=np.int32)
        ln = np.empty(max(P, 1), dtype=np.int32)
        i32 = lambda a: a.ctypes.data_as(_lib.c_int32_p)
        _lib.check(_lib.load().pvx_resident_fetch_table(pv._plan.handle, i32(pid), i32(st), i32(ln)), "pvx_resident_fetch_table")
        self._rpv = None
        # the values this object was built from (toSinSum copies them at call time, PVAnalysis.py:319-321): the PV's
        # host copies while nobody has touched them, else the untouched arrays that are still in the plan's block
        stale = pv._resident and not pv._on_device()

        def arr(name, which):
            return pv._fetch_result(which) if stale else np.ascontiguousarray(getattr(pv, name), dtype=np.float64)

        self._tab = dict(f=arr("f", 0), mag=arr("mag", 1), ph=arr("ph", 2), realph=arr("realph", 3),
                         pid=pid, st=st[:P].copy(), ln=ln[:P].copy())
        self._materialised = False

    def _from_analysis(self, f, mag, ph, realph, maxpitchjmp=0.5):
        lib = _lib.load()
        _lib.init()
        f = np.ascontiguousarray(f, dtype=np.float64)
        mag = np.ascontiguousarray(mag, dtype=np.float64)
        F, K = f.shape
        pid = np.empty((F, K), dtype=np.int32)
        cap = F * K
        st = np.empty(cap, dtype=np.int32)
        ln = np.empty(cap, dtype=np.int32)
        i32 = lambda a: a.ctypes.data_as(_lib.c_int32_p)
        P = lib.pvx_track(_lib.dptr(f), _lib.dptr(mag), F, K, float(maxpitchjmp), i32(pid), i32(st), i32(ln), cap)
        _lib.check(P, "pvx_track")
        self._tab = dict(f=f, mag=mag, ph=np.ascontiguousarray(ph, dtype=np.float64),
                         realph=np.ascontiguousarray(realph, dtype=np.float64),
                         pid=pid, st=st[:P].copy(), ln=ln[:P].copy())
        self._materialised = False

    def _materialise(self):
        """Build the reference's Python objects (RegPartial lists, st, end) from the table."""
        if self._materialised:
            return
        self._detach_resident()
        tab = self._tab
        pid, st, ln = tab['pid'], tab['st'], tab['ln']
        P = len(st)
        F, K = pid.shape
        # CSR gather: order the valid nodes by (partial, frame)
        flat = pid.ravel()
        nodes = np.flatnonzero(flat >= 0)
        order = nodes[np.argsort(flat[nodes], kind='stable')]   # within a partial: ascending frame
        off = np.concatenate([[0], np.cumsum(ln.astype(np.int64))])
        vals = {k: tab[k].ravel()[order] for k in ('f', 'mag', 'ph', 'realph')}
        overlap = self.hop / float(self.nfft)                    # PVAnalysis.py:824
        fstep = self.sr / float(self.nfft)
        parts = []
        for p in range(P):
            a, b = off[p], off[p + 1]
            parts.append(RegPartial(int(st[p]), pdict=dict(f=vals['f'][a:b].tolist(), mag=vals['mag'][a:b].tolist(),
                                                            ph=vals['ph'][a:b].tolist(),
                                                            realph=vals['realph'][a:b].tolist()),
                                    overlap=overlap, fstep=fstep))
        self._partial = parts
        self._st = [int(v) for v in st]
        self._end = [int(v) for v in (st.astype(np.int64) + ln - 1)]
        self._materialised = True

    @property
    def partial(self):
        self._materialise()
        self._tab_dirty = True      # the caller may mutate the objects; synth then re-packs them
        return self._partial

    @partial.setter
    def partial(self, v):
        self._materialise()
        self._partial = v
        self._tab_dirty = True

    @property
    def st(self):
        self._materialise()
        return self._st

    @st.setter
    def st(self, v):
        self._materialise()
        self._st = v

    @property
    def end(self):
        self._materialise()
        return self._end

    @end.setter
    def end(self, v):
        self._materialise()
        self._end = v

    _tab_dirty = False

    # ---- reference API ----------------------------------------------------------------------
    def add_empty_partial(self, idx):
        '''Append an empty partial at frame idx (PVAnalysis.py:819-830).'''
        self._materialise()
        newpart = RegPartial(idx, overlap=self.hop / float(self.nfft), fstep=self.sr / float(self.nfft))
        self._partial.append(newpart)
        self._st.append(idx)
        self._end.append(idx)
        self._tab_dirty = True
        return newpart

    def get_partials_idx_ending_at_frame(self, fr):
        '''Index of the partials ending at fr (PVAnalysis.py:984-994).'''
        self._materialise()
        st = np.asarray(self._st)
        end = np.asarray(self._end)
        return np.flatnonzero((fr >= st) & (fr == end)) if len(st) else np.array([])

    def get_partials_idx_at_frame(self, fr):
        self._materialise()
        st = np.asarray(self._st)
        end = np.asarray(self._end)
        return np.flatnonzero((fr >= st) & (fr <= end)) if len(st) else np.array([])

    def get_partials_at_frame(self, fr):
        return [self._partial[i] for i in self.get_partials_idx_at_frame(fr)]

    def get_points_at_frame(self, fr):
        '''Placeholder in the reference too (PVAnalysis.py:996-1000).'''
        pass

    def add_frame(self, fr, f, mag, ph, realph=None, maxpitchjmp=0.5):
        '''
        Add the peaks of frame fr to the matching partials or start new ones (PVAnalysis.py:871-957).
        The matching itself runs in the tracker kernels on a two-row table: row 0 = the partials
        that end at fr-1 (in partial-index order), row 1 = the new peaks.
        '''
        lib = _lib.load()
        _lib.init()
        self._materialise()
        f = np.asarray(f, dtype=np.float64)
        mag = np.asarray(mag, dtype=np.float64)
        ph = np.asarray(ph, dtype=np.float64)
        rp = ph if realph is None else np.asarray(realph, dtype=np.float64)
        pidx = [int(i) for i in self.get_partials_idx_ending_at_frame(fr - 1)]
        K = max(len(f), len(pidx), 1)
        tf = np.zeros((2, K)); tm = np.zeros((2, K))
        for j, i in enumerate(pidx):
            tf[0, j] = self._partial[i].get_freq_at_frame(fr - 1)
            tm[0, j] = self._partial[i].get_mag_at_frame(fr - 1)
        tf[1, :len(f)] = f
        tm[1, :len(mag)] = mag
        pid = np.empty((2, K), dtype=np.int32)
        st = np.empty(2 * K, dtype=np.int32)
        ln = np.empty(2 * K, dtype=np.int32)
        i32 = lambda a: a.ctypes.data_as(_lib.c_int32_p)
        P = lib.pvx_track(_lib.dptr(tf), _lib.dptr(tm), 2, K, float(maxpitchjmp), i32(pid), i32(st), i32(ln), 2 * K)
        _lib.check(P, "pvx_track")
        # process the new peaks in the reference's order: extended partials keep their object,
        # new ones are created in creation order (= ascending new id)
        valid = [s for s in range(len(f)) if pid[1, s] >= 0]
        news = sorted([s for s in valid if st[pid[1, s]] == 1], key=lambda s: pid[1, s])
        for s in valid:
            q = pid[1, s]
            if st[q] == 0:                                       # continues the partial in row-0 slot
                j = int(np.flatnonzero(pid[0] == q)[0])
                idx = pidx[j]
                self._partial[idx].append_point(f[s], mag[s], ph[s], realph=rp[s])
                self._end[idx] = fr
        for s in news:
            part = self.add_empty_partial(fr)
            part.append_point(f[s], mag[s], ph[s], realph=rp[s])
            self._end[-1] = fr
        self._tab_dirty = True

    def _pack_partials(self):
        """(F, K) arrays + ids from the Python partial objects (after user edits)."""
        parts = self._partial
        if not parts:
            raise ValueError("max() arg is an empty sequence")
        st = np.array([p.start_idx for p in parts], dtype=np.int64)
        ln = np.array([len(p.f) for p in parts], dtype=np.int64)
        F = int(max(max(self._end), (st + ln - 1).max())) + 1
        occ = np.zeros(F, dtype=np.int64)
        for s, l in zip(st, ln):
            occ[s:s + l] += 1
        K = max(int(occ.max()), 1)
        f = np.zeros((F, K)); mag = np.zeros((F, K)); rp = np.zeros((F, K))
        pid = np.full((F, K), -1, dtype=np.int32)
        fill = np.zeros(F, dtype=np.int64)
        for i, p in enumerate(parts):
            for j in range(len(p.f)):
                fr = p.start_idx + j
                s = fill[fr]; fill[fr] += 1
                f[fr, s] = p.f[j]; mag[fr, s] = p.mag[j]; rp[fr, s] = p.realph[j]
                pid[fr, s] = i
        return f, mag, rp, pid, st.astype(np.int32), ln.astype(np.int32)

    def synth(self, sr, hop, edge=1.0, minframes=3, phase_preserve=True):
        '''
        Overlap-add resynthesis (PVAnalysis.py:1053-1070) on the device.  `hop` may differ from the
        analysis hop (time stretch); callers pass floats like `mypv.hop/1` (examples/WavResynth.py:36),
        which is truncated to int as Python 2 did.
        '''
        if not phase_preserve:
            # RegPartial.synth_no_phase (PVAnalysis.py:653-682) fails on current numpy
            # (np.ones(float)); mirror that instead of inventing behaviour.
            raise TypeError("'float' object cannot be interpreted as an integer")
        hop = int(hop)
        pv = self._rpv
        if pv is not None and not self._tab_dirty and pv._on_device():
            # analysis arrays and partial table are in HBM: only the waveform comes back
            lib = _lib.load()
            if self._rP == 0:
                raise ValueError("max() arg is an empty sequence")
            n = lib.pvx_synth_len(self._rmaxend, int(self.nfft), int(self.hop), hop, float(edge))
            _lib.check(n, "pvx_synth_len")
            w = _lib.result_empty(n)                          # page-locked: the waveform is DMA'd into the array itself
            _lib.check(lib.pvx_synth_resident(pv._plan.handle, float(sr), hop, float(edge), int(minframes), _lib.dptr(w), n),
                       "pvx_synth_resident")
            return w
        self._detach_resident()
        # (the SinSum of a precision-32 analysis resynthesises like its resident form does: float32 sample loop)
        flags = _lib.PVX_SYNTH_F32 if (self._precision == 32 and not os.environ.get("PVX_SYNTH_F64")) else 0
        if self._tab is not None and not self._tab_dirty:
            tab = self._tab
            if len(tab['st']) == 0:
                raise ValueError("max() arg is an empty sequence")
            return _device_synth(tab['f'], tab['mag'], tab['realph'], tab['pid'], tab['st'], tab['ln'],
                                 sr, self.nfft, int(self.hop), hop, edge, minframes, flags)
        self._materialise()
        f, mag, rp, pid, st, ln = self._pack_partials()
        return _device_synth(f, mag, rp, pid, st, ln, sr, self.nfft, int(self.hop), hop, edge, minframes, flags)

    def get_avfreq(self):
        return np.array([np.mean(xx.f) for xx in self.partial])

    def get_avmag(self):
        return np.array([np.mean(xx.mag) for xx in self.partial])

    def get_nframes(self):
        return max(self.end)

    # table access without building Python objects (large analyses)
    def partial_table(self):
        """(partial_id[F,K], part_start[P], part_len[P]) as produced by the tracker kernels."""
        self._detach_resident()
        if self._tab is None or self._tab_dirty:
            self._materialise()
            _, _, _, pid, st, ln = self._pack_partials()
            return pid, st, ln
        return self._tab['pid'], self._tab['st'], self._tab['ln']


def _unsupported(name, where):
    def method(self, *args, **kwargs):
        raise NotImplementedError(
            "%s (%s) is outside the accelerated PV.run_pv -> toSinSum -> synth path and is not mirrored by "
            "pypevoc_amd; use the reference class for it (see INTEGRATION.md, 'not mirrored')" % (name, where))
    method.__name__ = name.split(".")[-1]
    method.__doc__ = "Not mirrored: %s." % where
    return method


# helpers of the reference classes that the path never calls (plotting, the self-described slow
# add_point, summaries, and RegPartial methods that are broken on current numpy): a clear error, not an AttributeError
for _cls, _n, _w in ((SinSum, "add_point", "PVAnalysis.py:832-868"), (SinSum, "get_summary", "PVAnalysis.py:1078-1089"),
                     (SinSum, "get_part_data_around_freq", "PVAnalysis.py:1091-1111"), (SinSum, "plot_time_freq", "PVAnalysis.py:1002-1031"),
                     (SinSum, "plot_time_freq_mag", "PVAnalysis.py:1033-1051"), (RegPartial, "prepend_point", "PVAnalysis.py:628-651"),
                     (RegPartial, "synth_no_phase", "PVAnalysis.py:653-682"), (RegPartial, "get_rel_phase", "PVAnalysis.py:758-794"),
                     (PV, "plot_time_freq", "PVAnalysis.py:324-346"), (PV, "plot_time_mag", "PVAnalysis.py:348-369")):
    setattr(_cls, _n, _unsupported(_cls.__name__ + "." + _n, _w))

