#!/usr/bin/env python3
"""Per-stage kernel time of one analysis pass (libpvx_hip's stage events): python tools/stage_times.py NFFT PRECISION [K]"""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pypevoc_amd import _lib
from bench import c2_signal
nfft = int(sys.argv[1]); prec = int(sys.argv[2]); K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lib = _lib.load(); _lib.init(0)
dev = torch.device("cuda", 0); s = torch.cuda.Stream(device=dev); sp = ctypes.c_void_p(s.cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(1)
for name, x in (("harmonic", torch.from_numpy(c2_signal(600)).to(dev)), ("noise", 0.1 * torch.randn(44100 * 600, device=dev, generator=g))):
    hop = nfft // 4; n = x.numel(); F = int(lib.pvx_nframes(n, nfft, hop))
    out = torch.zeros(5 * F * K + 2 * F, dtype=torch.float64, device=dev); b = out.data_ptr()
    ptrs = [b + i * F * K * 8 for i in range(5)] + [b + 5 * F * K * 8, b + 5 * F * K * 8 + F * 8]
    plan = ctypes.c_void_p(); win = np.hanning(nfft)
    _lib.check(lib.pvx_plan_create(ctypes.byref(plan), 44100.0, nfft, hop, K, 0.005, _lib.dptr(win), prec, 0), "plan")
    for _ in range(2): _lib.check(lib.pvx_analyze_dev(plan, x.data_ptr(), 0, n, 1, n, *ptrs, None, sp), "a")
    torch.cuda.synchronize()
    _lib.check(lib.pvx_plan_set_timing(plan, 1), "t")
    reps = 5
    for _ in range(reps): lib.pvx_analyze_dev(plan, x.data_ptr(), 0, n, 1, n, *ptrs, None, sp)
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 4)(); nl = (ctypes.c_int64 * 4)()
    _lib.check(lib.pvx_plan_get_timing(plan, ms, nl), "g")
    print(json.dumps(dict(nfft=nfft, precision=prec, input=name, frames=F, fft_mode=int(lib.pvx_plan_get_fft_mode(plan)),
                          ms_per_pass={k: round(ms[i] / reps, 4) for i, k in enumerate(("stft_or_frames", "rocfft", "phase_peaks", "fused")) if nl[i]})))
    lib.pvx_plan_destroy(plan)
