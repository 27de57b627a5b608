"""ctypes binding of libpvx_hip.so (C ABI: include/pvx.h).

The library is the product: there is no Python/numpy fallback.  Importing this module only loads
the shared object; the first call that needs the device raises PvxError if no MI355X is usable.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PVX_LIB") or os.path.join(_HERE, "libpvx_hip.so")   # PVX_LIB: A/B builds of the same library

PVX_SYNTH_NO_PHCOR = 1
PVX_SYNTH_F32 = 2
PVX_ERR_SIZE = -6            # include/pvx.h
PVX_F32, PVX_F64, PVX_I16 = 0, 1, 2

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_int64_p = ctypes.POINTER(ctypes.c_int64)
c_int8_p = ctypes.POINTER(ctypes.c_int8)

# name -> (restype, argtypes); every symbol include/pvx.h declares
SIGNATURES = {
    "pvx_init": (ctypes.c_int, [ctypes.c_int]),
    "pvx_last_error": (ctypes.c_char_p, []),
    "pvx_version": (ctypes.c_int, []),
    "pvx_build_fingerprint": (ctypes.c_char_p, []),
    "pvx_device_name": (ctypes.c_char_p, []),
    "pvx_device": (ctypes.c_int, []),
    "pvx_host_alloc": (ctypes.c_void_p, [ctypes.c_size_t]),
    "pvx_host_free": (None, [ctypes.c_void_p]),
    "pvx_nframes": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "pvx_plan_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_double, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_double, c_double_p,
                                       ctypes.c_int, ctypes.c_int64]),
    "pvx_plan_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "pvx_plan_workspace_bytes": (ctypes.c_int64, [ctypes.c_void_p]),
    "pvx_plan_set_fft_mode": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "pvx_plan_get_fft_mode": (ctypes.c_int, [ctypes.c_void_p]),
    "pvx_plan_device": (ctypes.c_int, [ctypes.c_void_p]),
    "pvx_plan_last_kernels": (ctypes.c_char_p, [ctypes.c_void_p]),
    "pvx_plan_set_timing": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "pvx_plan_get_timing": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_int64_p]),
    "pvx_analyze_dev": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.c_int64] + [ctypes.c_void_p] * 8 +
                        [ctypes.c_void_p]),
    "pvx_analyze": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                     ctypes.c_int64, ctypes.c_int64] + [c_double_p] * 7 +
                    [c_double_p, c_double_p]),
    "pvx_stft_frames": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                       c_int64_p, ctypes.c_int64, c_double_p]),
    "pvx_peakfinder": (ctypes.c_int, [c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_double, ctypes.c_int, c_int32_p, c_int8_p, c_int32_p,
                                      ctypes.c_int]),
    "pvx_track": (ctypes.c_int64, [c_double_p, c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                   c_int32_p, c_int32_p, c_int32_p, ctypes.c_int64]),
    "pvx_track_dev": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                       ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_int64, ctypes.c_void_p]),
    "pvx_synth_len": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_double]),
    "pvx_synth": (ctypes.c_int, [c_double_p, c_double_p, c_double_p, c_int32_p, ctypes.c_int64, ctypes.c_int,
                                 c_int32_p, c_int32_p, ctypes.c_int64, ctypes.c_double, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, c_double_p,
                                 ctypes.c_int64]),
    "pvx_analyze_resident": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_int64, c_double_p, c_double_p]),
    "pvx_resident_fetch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_double_p]),
    "pvx_resident_ptr": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]),
    "pvx_track_resident": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_double, c_int64_p]),
    "pvx_resident_fetch_table": (ctypes.c_int, [ctypes.c_void_p, c_int32_p, c_int32_p, c_int32_p]),
    "pvx_synth_resident": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                          c_double_p, ctypes.c_int64]),
    "pvx_f0_resident": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_double_p, c_int32_p]),
    "pvx_harmonic_power_resident": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, c_double_p, c_double_p]),
    "pvx_synth_flags": (ctypes.c_int, [c_double_p, c_double_p, c_double_p, c_int32_p, ctypes.c_int64, ctypes.c_int,
                                       c_int32_p, c_int32_p, ctypes.c_int64, ctypes.c_double, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, c_double_p,
                                       ctypes.c_int64, ctypes.c_int]),
    "pvx_synth_dev_flags": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                                                   ctypes.c_void_p, ctypes.c_int64, ctypes.c_double,
                                                                   ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                                   ctypes.c_double, ctypes.c_int, ctypes.c_void_p,
                                                                   ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]),
    "pvx_synth_dev": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                                             ctypes.c_void_p, ctypes.c_int64, ctypes.c_double,
                                                             ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                             ctypes.c_double, ctypes.c_int, ctypes.c_void_p,
                                                             ctypes.c_int64, ctypes.c_void_p]),
    "pvx_harmonic_analyze": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                              c_double_p, ctypes.c_int64, ctypes.c_double] + [c_double_p] * 7),
    "pvx_harmonic_analyze_dev": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                                  c_double_p, ctypes.c_int64, ctypes.c_double] + [ctypes.c_void_p] * 7),
    "pvx_heterodyne": (ctypes.c_int64, [c_double_p, c_double_p, ctypes.c_int64, c_double_p, ctypes.c_int, ctypes.c_int,
                                        c_double_p, c_int64_p]),
    "pvx_heterodyne_dev": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, c_double_p, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pvx_rms_frames": (ctypes.c_int64, [c_double_p, ctypes.c_int64, c_double_p, ctypes.c_int, ctypes.c_int, c_double_p]),
    "pvx_funcwind": (ctypes.c_int64, [c_double_p, ctypes.c_int, ctypes.c_int64, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_double, c_double_p]),
    "pvx_funcwind_dev": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    "pvx_rms_frames_dev": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int64, c_double_p, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_void_p]),
    "pvx_plan_set_progress": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "pvx_plan_set_wire_format": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "pvx_plan_get_wire_format": (ctypes.c_int, [ctypes.c_void_p]),
    "pvx_wire_bytes": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int64]),
    "pvx_batch_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_double, c_double_p, ctypes.c_int, c_int32_p, ctypes.c_int, ctypes.c_int]),
    "pvx_batch_run": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]),
    "pvx_batch_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "pvx_analyze_batch": (ctypes.c_int64, [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_double_p,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, c_int32_p, ctypes.c_int]),
    "pvx_pack_rows_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 7),
    "pvx_analyze_dev_wire": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_void_p, ctypes.c_void_p]),
    "pvx_unpack_rows_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 8),
}


class BatchItem(ctypes.Structure):
    """pvx_batch_item of include/pvx.h."""
    _fields_ = [("x", ctypes.c_void_p), ("nsamp", ctypes.c_int64),
                ("f", c_double_p), ("mag", c_double_p), ("ph", c_double_p), ("realph", c_double_p), ("binno", c_double_p),
                ("t", c_double_p), ("totalmag", c_double_p),
                ("nframes", ctypes.c_int64), ("device", ctypes.c_int32), ("reserved", ctypes.c_int32)]


PROGRESS_FN = ctypes.CFUNCTYPE(None, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p)


class PvxError(RuntimeError):
    """A libpvx_hip call failed (message from pvx_last_error())."""


_LIB = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7; if
    libpvx_hip.so pulled in the system one first, a later `import torch` would bring a second runtime
    into the process and fail with "No HIP GPUs are available".  libpvx_hip.so binds to whichever
    libamdhip64.so.7 is already loaded, so when torch is installed (and not yet imported) its copy is
    loaded first.  PVX_SYSTEM_HIP=1 skips this."""
    import sys
    if os.environ.get("PVX_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass            # no torch, or an unusual layout: the system runtime is used


def _bind(path):
    """dlopen `path`, bind every entry point and check that it was built from the sources beside the package."""
    if not os.path.exists(path):
        raise ImportError(
            "pypevoc_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pypevoc_amd/csrc`.  There is no CPU fallback." % path)
    _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    want = source_fingerprint()
    have = lib.pvx_build_fingerprint().decode()
    if want is not None and have != want and not os.environ.get("PVX_ALLOW_STALE_LIB"):
        raise ImportError(
            "pypevoc_amd: %s was built from other sources, or with other compiler flags / another ARCH than the Makefile's "
            "defaults (fingerprint %s; pypevoc_amd/csrc with the default flags is %s) -- rebuild it with "
            "`make -C pypevoc_amd/csrc`, or load it as it is with PVX_ALLOW_STALE_LIB=1." % (path, have, want))
    return lib


def load():
    """Load libpvx_hip.so and bind every entry point.  Fails loudly if the library is missing."""
    global _LIB
    if _LIB is None:
        _LIB = _bind(LIB_PATH)
    return _LIB


def swap_library(lib_or_path):
    """TESTS ONLY: make another build of the same sources (tests/libpvx_witness.so: the product plus the witness kernels of
    fft modes 1 and 3) the library every call goes through; returns the handle it replaces, to be swapped back.  Pooled plans
    belong to the library that made them: the pool is emptied on every swap."""
    global _LIB, _INIT_DEVICE
    old = load()
    _LIB = _bind(lib_or_path) if isinstance(lib_or_path, str) else lib_or_path
    _INIT_DEVICE = None
    from . import PVAnalysis
    PVAnalysis._Plan._pool.clear()
    return old


def source_fingerprint():
    """What csrc/Makefile writes into build_sha.inc (the library's pvx_build_fingerprint()): sha256 over pypevoc_amd/csrc/*.hip
    and *.h in name order, include/pvx.h and the Makefile's default CXXFLAGS, first 16 hex digits; None when the sources are
    not beside the package."""
    import glob
    import hashlib
    import re
    d = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")), key=os.path.basename)
    if not files:
        return None
    files.append(os.path.join(_HERE, "..", "include", "pvx.h"))
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(d, "Makefile")) as fh:
        mk = fh.read()
    arch = re.search(r"^ARCH\s*\?=\s*(\S+)", mk, re.M).group(1)
    flags = re.search(r"^CXXFLAGS \?= ((?:.*\\\n)*.*)$", mk, re.M).group(1)
    flags = re.sub(r"\s*\\\n\s*", " ", flags).replace("$(ARCH)", arch).strip()
    h.update(flags.encode())
    return h.hexdigest()[:16]


def check(rc, what="pvx call"):
    if rc < 0:
        msg = load().pvx_last_error()
        raise PvxError("%s failed (status %d): %s" % (what, rc, msg.decode() if msg else ""))
    return rc


_INIT_DEVICE = None


def init(device=None):
    """Bind the library to a HIP device (default: the current one).  Raises PvxError without a GPU."""
    global _INIT_DEVICE
    lib = load()
    dev = -1 if device is None else int(device)
    if _INIT_DEVICE is not None and (device is None or _INIT_DEVICE == dev):
        return _INIT_DEVICE
    check(lib.pvx_init(dev), "pvx_init")
    _INIT_DEVICE = int(lib.pvx_device())        # the device the library actually bound (the current one for None)
    return _INIT_DEVICE


def device_name():
    return load().pvx_device_name().decode()


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def as_signal(x):
    """Signal buffer in one of the three sample types the kernels read directly."""
    x = np.asarray(x)
    if x.dtype == np.float32:
        return np.ascontiguousarray(x), PVX_F32
    if x.dtype == np.int16:
        return np.ascontiguousarray(x), PVX_I16
    return np.ascontiguousarray(x, dtype=np.float64), PVX_F64


def nframes_host(nsamp, nfft, hop):
    """Number of frames of run_pv (PVAnalysis.py:224-225, 249): positions 0, hop, ... < nsamp - nfft.
    Pure arithmetic, usable without the library (partitioning helpers)."""
    nsamp, nfft, hop = int(nsamp), int(nfft), int(hop)
    return (nsamp - nfft + hop - 1) // hop if nsamp > nfft else 0


class DeviceSignal(object):
    """A signal that already lives in GPU memory (anything with __cuda_array_interface__, e.g. a torch
    tensor on the GPU; float32, float64 or int16, C-contiguous).  The analysis reads it in place."""

    def __init__(self, obj):
        cai = obj.__cuda_array_interface__
        ts = cai["typestr"]
        if ts not in ("<f4", "<f8", "<i2"):
            raise TypeError("device signal must be float32, float64 or int16, got %s" % ts)
        if cai.get("strides") is not None:
            item = int(ts[2:])
            expect = []
            acc = item
            for n in reversed(cai["shape"]):
                expect.insert(0, acc)
                acc *= n
            if tuple(cai["strides"]) != tuple(expect):
                raise ValueError("device signal must be C-contiguous")
        self.obj = obj                       # keeps the memory alive
        self.ptr = int(cai["data"][0])
        self.shape = tuple(int(v) for v in cai["shape"])
        self.dtype_code = {"<f4": PVX_F32, "<f8": PVX_F64, "<i2": PVX_I16}[ts]
        self.dtype = np.dtype(ts)


def is_device_array(x):
    return hasattr(x, "__cuda_array_interface__") and not isinstance(x, np.ndarray)


def device_run(nbytes_out, launch):
    """Run `launch(out_ptr, stream_ptr)` with a float64 device output block of nbytes_out/8 elements
    on torch's current stream, return the block as a host numpy array."""
    import torch
    bound = init()
    if torch.cuda.current_device() != bound:
        raise PvxError("torch's current device is cuda:%d but libpvx_hip is bound to device %d (pypevoc_amd._lib.init(device))"
                       % (torch.cuda.current_device(), bound))
    out = torch.empty(nbytes_out // 8, dtype=torch.float64, device=torch.device("cuda", bound))
    stream = torch.cuda.current_stream()
    launch(out.data_ptr(), ctypes.c_void_p(stream.cuda_stream))
    stream.synchronize()
    return out.cpu().numpy()


# ---- result arrays in page-locked memory ------------------------------------------------------------
class _HostPool(object):
    """numpy arrays whose memory the DMA engine can write directly (pvx_host_alloc).  A buffer goes back to the
    pool when the last array / view on it is collected; sizes are bucketed to powers of two, the pool keeps at most
    _CAP (+ _BIG_CAP for the large buckets) bytes of idle buffers and hands out at most _LIVE_CAP bytes in all (beyond
    that: np.empty).
    Small results (64 KB .. 16 MB) are always page-locked.  LARGE ones (.. 1 GiB: the waveform of a 10-minute signal is
    212 MB) only from the second request of their bucket on: page-locking a quarter of a gigabyte costs tens of
    milliseconds, which a one-off call would never get back (its result arrives through the threaded ring of pvx_api.hip
    instead: ~30 GB/s, the page faults of the fresh array included), while a process that resynthesises signal after
    signal gets the link's 55 GB/s and no page faults from then on."""
    _CAP = 64 << 20
    _BIG_CAP = 1 << 30
    _LIVE_CAP = 3 << 30
    _MIN = 64 << 10
    _MAX = 16 << 20
    _BIG_MAX = 1 << 30

    def __init__(self):
        self.free = {}
        self.idle = 0
        self.idle_big = 0
        self.live = 0
        self.asked = {}

    def _release(self, ptr, size):
        self.live -= size
        big = size > self._MAX
        if (self.idle_big + size <= self._BIG_CAP) if big else (self.idle + size <= self._CAP):
            self.free.setdefault(size, []).append(ptr)
            if big:
                self.idle_big += size
            else:
                self.idle += size
        else:
            try:
                load().pvx_host_free(ctypes.c_void_p(ptr))
            except Exception:
                pass

    def empty(self, n, dtype=np.float64):
        import weakref
        nbytes = int(n) * np.dtype(dtype).itemsize
        if nbytes < self._MIN or nbytes > self._BIG_MAX:
            return np.empty(n, dtype=dtype)
        size = self._MIN
        while size < nbytes:
            size <<= 1
        big = size > self._MAX
        if big:
            self.asked[size] = self.asked.get(size, 0) + 1
            if self.asked[size] < 2 and not os.environ.get("PVX_PIN_LARGE_RESULTS"):
                return np.empty(n, dtype=dtype)
        lst = self.free.get(size)
        if lst:
            ptr = lst.pop()
            if big:
                self.idle_big -= size
            else:
                self.idle -= size
        else:
            if self.live + size > self._LIVE_CAP:
                return np.empty(n, dtype=dtype)
            ptr = load().pvx_host_alloc(size)
            if not ptr:
                return np.empty(n, dtype=dtype)
        self.live += size
        buf = (ctypes.c_char * size).from_address(ptr)
        weakref.finalize(buf, self._release, ptr, size).atexit = False
        return np.frombuffer(buf, dtype=dtype, count=int(n))


_HOST_POOL = _HostPool()


def result_empty(n, dtype=np.float64):
    """An uninitialised 1-D result array, page-locked when that pays (see _HostPool)."""
    return _HOST_POOL.empty(n, dtype)
