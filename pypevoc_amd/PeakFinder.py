"""Drop-in mirror of pypevoc.PeakFinder.PeakFinder for the part the phase vocoder uses
(pypevoc/PeakFinder.py:35-74 ctor, 155-194 findpos, 113-136 filter_by_salience, 76-111 properties).

Peak selection and the salience filter run in libpvx_hip (one wave64 per row, k_peaks.hip); the
refinement / prominence / area helpers of the reference are outside the hot path and not mirrored.
"""
import numpy as np

from . import _lib


def find_peaks_rows(y, npeaks=None, minrattomax=None, minval=None, rad=None):
    """PeakFinder + optional filter_by_salience(rad) on every row of a 2-D array in one launch.
    Returns (pos[R, cap] int32 (-1 padded), keep[R, cap] bool, count[R])."""
    lib = _lib.load()
    _lib.init()
    y = np.ascontiguousarray(np.atleast_2d(y), dtype=np.float64)
    R, n = y.shape
    kind, val = 0, 0.0
    if minrattomax is not None:
        kind, val = 1, float(minrattomax)
    elif minval is not None:
        kind, val = 2, float(minval)
    npk = int(npeaks) if npeaks else 0
    cap = max(1, min(npk if npk > 0 else n, n))
    pos = np.empty((R, cap), dtype=np.int32)
    keep = np.empty((R, cap), dtype=np.int8)
    count = np.empty(R, dtype=np.int32)
    _lib.check(lib.pvx_peakfinder(_lib.dptr(y), R, n, npk, kind, val, -1 if rad is None else int(rad),
                                  pos.ctypes.data_as(_lib.c_int32_p), keep.ctypes.data_as(_lib.c_int8_p),
                                  count.ctypes.data_as(_lib.c_int32_p), cap), "pvx_peakfinder")
    return pos, keep.astype(bool), count


class PeakFinder(object):

    def __init__(self, y, x=None, npeaks=None, minrattomax=None, minval=None):
        """Creates the peak finder object from a numpy array (PeakFinder.py:35-74)

        Arguments:
            y:           the numpy array in which to find peaks
            npeaks:      maximum number of peaks to find
          Thresholds:
            minrattomax: ratio of minimum to maximum peak amplitude (has priority over minval)
            minval:      an absolute minimum value of peak
        """
        self.y = np.array(np.squeeze(y))
        if x is not None:
            self.x = np.array(np.squeeze(x))
        else:
            self.x = np.arange(len(self.y))
        self._idx = np.array([])
        self._val = np.array([])
        if minrattomax is None:
            self.minamp = minval
        else:
            self.minamp = self.y.max() * minrattomax
        self.sorttype = 0
        if not npeaks:
            self.npeaks = len(self.y)
        else:
            self.npeaks = npeaks
        if not self.minamp:
            self.minamp = np.min(self.y)
        self._thr = dict(minrattomax=minrattomax, minval=minval)
        self.findpos()

    @property
    def pos(self):
        return self._fine_pos[self._keep]

    @property
    def rough_pos(self):
        return self.x[self._idx[self._keep]]

    @property
    def all_pos(self):
        return self.x[self._idx]

    @property
    def val(self):
        return self._fine_val[self._keep]

    @property
    def all_val(self):
        return self._val

    @property
    def rough_val(self):
        return self._val[self._keep]

    @property
    def bounds(self):
        b = np.array(self._bounds)
        return self.x[b[self._keep, :]]

    def _run(self, rad):
        pos, keep, count = find_peaks_rows(self.y, npeaks=self.npeaks, rad=rad, **self._thr)
        n = int(count[0])
        return pos[0, :n].astype(np.int64), keep[0, :n]

    def findpos(self):
        """Finds the peaks positions (PeakFinder.py:155-194)."""
        idx, _ = self._run(None)
        self._idx = idx
        self._val = np.array([self.y[i] for i in self._idx])
        self._keep = np.ones(len(self._idx), dtype='bool')
        self._order = np.arange(len(self._idx))
        self._fine_pos = np.array([self.x[ii] for ii in self._idx])
        self._fine_val = self._val

    def filter_by_salience(self, rad=1, sal=0):
        ''' Filters the peaks by salience (PeakFinder.py:113-136): any peak that is lower than a
            neighbouring point within 'rad' is filtered out.'''
        if sal != 0:
            # Off the phase-vocoder path (PV always passes sal=0, PVAnalysis.py:177): the test `any(w + sal > y[p])`
            # of PeakFinder.py:129-134 on the <= npeaks selected positions, on the host.  (The window contains the peak
            # itself, so any sal > 0 drops every peak -- in the reference too.)
            y = np.asarray(self.y)
            for i, p in enumerate(self._idx):
                w = y[max(int(p) - rad, 1):min(int(p) + rad, len(y)) + 1]
                if np.any(w + sal > self._val[i]):
                    self._keep[i] = False
            return
        _, keep = self._run(int(rad))
        self._keep = np.logical_and(self._keep, keep)

    def find_boundaries(self, all=False):
        """Local minima on either side of each peak (PeakFinder.py:269-302).  The phase vocoder
        calls this and discards the result (PVAnalysis.py:176); kept for callers that read it."""
        pos = self._idx if all else self._idx[self._keep]
        y = self.y
        try:
            prevb = int(np.argmin(y[0:self._idx[0]]))
        except (IndexError, ValueError):
            prevb = 0
        bounds = []
        for i, thismax in enumerate(pos):
            if i < len(pos) - 1:
                nextb = int(np.argmin(y[thismax:pos[i + 1]])) + int(thismax)
            else:
                nextb = len(y) - 1
            bounds.append([prevb, nextb])
            prevb = nextb
        self._bounds = np.array(bounds)

    def boundaries(self):
        try:
            self.find_boundaries(all=True)
        except IndexError:
            self._bounds = np.array([])
        return self._bounds

    def get_pos(self):
        return self.pos


def _unsupported(name, where):
    def method(self, *args, **kwargs):
        raise NotImplementedError(
            "%s (%s) is outside the accelerated PV.run_pv -> toSinSum -> synth path and is not mirrored by "
            "pypevoc_amd; use the reference class for it (see INTEGRATION.md, 'not mirrored')" % (name, where))
    method.__name__ = name.split(".")[-1]
    method.__doc__ = "Not mirrored: %s." % where
    return method


# sub-sample refinement, areas, prominence and export helpers of the reference class: named here so that a caller
# gets a clear NotImplementedError instead of an AttributeError
for _n, _w in (("refine", "PeakFinder.py:331-372"), ("refine_opt", "PeakFinder.py:304-329"), ("refine_all", "PeakFinder.py:374-413"),
               ("get_areas", "PeakFinder.py:415-437"), ("find_prominence", "PeakFinder.py:196-221"),
               ("filter_by_prominence", "PeakFinder.py:138-153"), ("to_dict", "PeakFinder.py:439-473"), ("plot", "PeakFinder.py:223-267")):
    setattr(PeakFinder, _n, _unsupported("PeakFinder." + _n, _w))

