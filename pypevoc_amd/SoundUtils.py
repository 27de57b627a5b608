"""Drop-ins for the hop-strided windowed measures of pypevoc/SoundUtils.py that share the analysis
framing: RMSWind (:71-103), Heterodyn (:106-117), HeterodynWithF0Track (:120-138).  The per-frame
reductions run as HIP kernels (k_reduce.hip); building the heterodyning signal is host numpy as in the
reference.  FuncWind (:42-69) runs on the device for the named reducers np.sum / np.mean / np.max / np.min / np.std /
np.var (k_funcwind); an arbitrary Python callable has no device form and raises TypeError."""
import numpy as np

from . import _lib
from .Heterodyne import heterodyne


def _frame_times(nsam, sr, nwind, nhop):
    nfr = int(_lib.load().pvx_nframes(int(nsam), int(nwind), int(nhop)))
    ist = np.arange(nfr) * int(nhop)
    return (ist + ist + int(nwind)) / 2.0 / float(sr)                # SoundUtils.py:64, 98


_FW_OPS = {"sum": 0, "mean": 1, "max": 2, "amax": 2, "min": 3, "amin": 3, "std": 4, "var": 5}


def _funcwind_op(func):
    """PVX_FW_* of a numpy reducer (np.sum, np.mean, np.max / np.amax, np.min / np.amin, np.std, np.var; the builtins sum /
    max / min count as their numpy namesakes) or a TypeError: other callables would have to run in Python per frame."""
    if isinstance(func, str):
        name = func
    elif any(func is b for b in (sum, max, min)):                    # (identity: `in` would compare an array-like elementwise)
        name = func.__name__
    elif (getattr(func, "__module__", None) or "").split(".")[0] == "numpy" and getattr(np, getattr(func, "__name__", ""), None) is func:
        name = func.__name__
    else:
        name = None
    if name not in _FW_OPS:
        raise TypeError("FuncWind on the device takes np.sum, np.mean, np.max, np.min, np.std or np.var (got %r): "
                        "an arbitrary callable has no device form (INTEGRATION.md)" % (func,))
    return _FW_OPS[name]


def FuncWind(func, x, sr=1, nwind=1024, nhop=512, power=1, windfunc=np.blackman):
    '''
    Applies a function window by window to a time series
    (func: np.sum, np.mean, np.max, np.min, np.std or np.var)
    '''
    op = _funcwind_op(func)
    lib = _lib.load()
    _lib.init()
    x = np.asarray(x)
    cpx = np.iscomplexobj(x)
    x = np.ascontiguousarray(x, dtype=np.complex128 if cpx else np.float64)
    wind = np.ascontiguousarray(windfunc(nwind), dtype=np.float64)
    if power > 0:
        wsumpow = sum(wind ** power)                                 # SoundUtils.py:55-58
    else:
        wsumpow = 1.
    t = _frame_times(len(x), sr, nwind, nhop)
    cout = cpx and op in (0, 1)
    out = np.zeros(len(t), dtype=np.complex128 if cout else np.float64)
    if len(t):
        r = lib.pvx_funcwind(x.view(np.float64).ctypes.data_as(_lib.c_double_p), int(cpx), len(x), _lib.dptr(wind), int(nwind), int(nhop), op,
                             float(wsumpow), out.view(np.float64).ctypes.data_as(_lib.c_double_p))
        _lib.check(r, "pvx_funcwind")
    return out, t


def RMSWind(x, sr=1, nwind=1024, nhop=512, windfunc=np.blackman):
    '''
    Calculates the RMS amplitude amplitude of x, in frames of
    length nwind, and in steps of nhop. windfunc is used as
    windowing function.
    '''
    lib = _lib.load()
    _lib.init()
    x = np.ascontiguousarray(x, dtype=np.float64)
    wind = np.ascontiguousarray(windfunc(nwind), dtype=np.float64)
    t = _frame_times(len(x), sr, nwind, nhop)
    out = np.zeros(len(t))
    if len(t):
        _lib.check(lib.pvx_rms_frames(_lib.dptr(x), len(x), _lib.dptr(wind), int(nwind), int(nhop), _lib.dptr(out)),
                   "pvx_rms_frames")
    return out, t


def _het(x, sinsig, sr, nwind, nhop, windfunc):
    # FuncWind(np.sum, x*sinsig, power=1) * 2 = heterodyne(x, sinsig, windfunc(nwind), nhop)
    hamp, _ = heterodyne(x, sinsig, wind=windfunc(nwind), hop=nhop)
    return hamp, _frame_times(len(x), sr, nwind, nhop)


def Heterodyn(x, f, sr=1, nwind=1024, nhop=512, windfunc=np.blackman):
    '''
    Calculates the amplitude near frequency f in x
    '''
    sinsig = np.exp(2j * np.pi * np.arange(len(x)) * f / float(sr))
    return _het(x, sinsig, sr, nwind, nhop, windfunc)


def HeterodynWithF0Track(x, tf0, f0, sr=1, nwind=1024, nhop=512, windfunc=np.blackman):
    '''
    Calculates the amplitude near frequency f0 in x
    (f0 is time-varying, values given at tf0)
    '''
    tf0 = np.asarray(tf0)
    f0 = np.asarray(f0)
    valid_idx = np.logical_not(np.isnan(f0))
    tx = np.arange(len(x)) / float(sr)
    f0s = np.interp(tx, tf0[valid_idx], f0[valid_idx])
    phs = np.cumsum(2 * np.pi * f0s / sr)
    return _het(x, np.exp(1j * phs), sr, nwind, nhop, windfunc)
