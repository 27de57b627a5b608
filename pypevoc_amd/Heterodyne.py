"""Drop-in for pypevoc.Heterodyne.heterodyne (pypevoc/Heterodyne.py:35-60): the windowed complex
demodulation runs as one HIP kernel (pvx_heterodyne, k_reduce.hip).  The Heterodyne / HeterodyneHarmonic
classes built on top of it in the reference are host-side orchestration and are out of scope."""
import ctypes

import numpy as np

from . import _lib


def heterodyne(x, hetsig, wind=None, hop=None):
    """
    Heterodyner: calculates the complex amplitude of a sine wave centered at f

    Arguments:
        x: signal
        hetsig: complex heterodyning signal, same length as x (exp(-2j*pi*cumsum(f/sr)))
        wind: window (array, defaults to 256 point rectangular)
        hop: samples between windows (required, as in the reference where None fails in range())
    Returns (2 * windowed mean of x*hetsig per frame, centre sample of each frame).
    """
    if wind is None:
        wind = np.ones(2 ** 8)
    if hop is None:
        raise TypeError("'NoneType' object cannot be interpreted as an integer")   # range(0, n, None)
    lib = _lib.load()
    _lib.init()
    x = np.ascontiguousarray(x, dtype=np.float64)
    h = np.ascontiguousarray(hetsig, dtype=np.complex128)
    if len(h) != len(x):
        raise ValueError("operands could not be broadcast together with shapes (%d,) (%d,)" % (len(x), len(h)))
    wind = np.ascontiguousarray(wind, dtype=np.float64)
    wlen = len(wind)
    nfr = int(lib.pvx_nframes(len(x), wlen, int(hop)))
    out = np.zeros(nfr, dtype=np.complex128)
    icent = np.zeros(nfr, dtype=np.int64)
    if nfr > 0:
        r = lib.pvx_heterodyne(_lib.dptr(x), h.view(np.float64).ctypes.data_as(_lib.c_double_p), len(x), _lib.dptr(wind),
                               wlen, int(hop), out.view(np.float64).ctypes.data_as(_lib.c_double_p),
                               icent.ctypes.data_as(_lib.c_int64_p))
        _lib.check(r, "pvx_heterodyne")
    return out, icent
