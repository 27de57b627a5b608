#!/usr/bin/env python3
"""Frames/s of the analysis stage over the nfft sweep of BASELINE config 5 (hop = nfft/4, npks 8, 10 min of 44.1 kHz
harmonic signal and of white noise), plan default fft mode.   python tools/ab_nfft.py [nfft,...]"""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pypevoc_amd import _lib
from bench import c2_signal
nffts = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512,1024,2048,4096,8192").split(",")]
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lib = _lib.load(); _lib.init(0)
dev = torch.device("cuda", 0); s = torch.cuda.Stream(device=dev); sp = ctypes.c_void_p(s.cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(1)
inputs = {"harmonic": torch.from_numpy(c2_signal(600)).to(dev), "noise": 0.1 * torch.randn(44100 * 600, device=dev, generator=g)}
g7 = os.path.join(ROOT, "tests", "golden", "G7_perlman.npz")
if os.path.exists(g7):
    xv = np.load(g7)["x"].astype(np.float32)
    inputs["violin"] = torch.from_numpy(np.tile(xv, 44100 * 600 // len(xv) + 1)[: 44100 * 600]).to(dev)
XD = 1 if os.environ.get("AB_F64IN") else 0               # AB_F64IN=1: float64 samples in HBM (what a reference user's arrays are), x_dtype PVX_F64
if XD:
    inputs = {k: v.double() for k, v in inputs.items()}
K = int(os.environ.get("AB_K", "8"))                   # peaks per frame (AB_K=100: BASELINE config 3's)
for nfft in nffts:
    hop = nfft // int(os.environ.get("AB_HOP_DIV", "4"))                 # (AB_HOP_DIV=2: hop = nfft/2, the reference's default)
    for name, x in inputs.items():
        n = x.numel(); F = int(lib.pvx_nframes(n, nfft, hop))
        out = torch.zeros(5 * F * K + 2 * F, dtype=torch.float64, device=dev); b = out.data_ptr()
        ptrs = [b + i * F * K * 8 for i in range(5)] + [b + 5 * F * K * 8, b + 5 * F * K * 8 + F * 8]
        plan = ctypes.c_void_p(); win = np.hanning(nfft)
        _lib.check(lib.pvx_plan_create(ctypes.byref(plan), 44100.0, nfft, hop, K, 0.005, _lib.dptr(win), prec, 0), "plan")
        for _ in range(2): _lib.check(lib.pvx_analyze_dev(plan, x.data_ptr(), XD, n, 1, n, *ptrs, None, sp), "a")
        torch.cuda.synchronize()
        import time
        t_r = time.perf_counter() + 0.15                      # clock ramp (bench.py): the card leaves its idle clocks
        while time.perf_counter() < t_r:
            for _ in range(16): lib.pvx_analyze_dev(plan, x.data_ptr(), XD, n, 1, n, *ptrs, None, sp)
            torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); reps = 10
        e0.record(s)
        for _ in range(reps): lib.pvx_analyze_dev(plan, x.data_ptr(), XD, n, 1, n, *ptrs, None, sp)
        e1.record(s); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        stages = None
        if os.environ.get("AB_STAGES"):                       # the general path's stages of one more call: frames / transform, rocFFT, peaks, one-launch kernel
            ms4 = (ctypes.c_double * 4)(); nl4 = (ctypes.c_int64 * 4)()
            lib.pvx_plan_set_timing(plan, 1)
            lib.pvx_analyze_dev(plan, x.data_ptr(), XD, n, 1, n, *ptrs, None, sp); torch.cuda.synchronize()
            lib.pvx_plan_get_timing(plan, ms4, nl4); lib.pvx_plan_set_timing(plan, 0)
            stages = dict(ms=[round(v, 4) for v in ms4], launches=[int(v) for v in nl4])
        print(json.dumps(dict(nfft=nfft, hop=hop, input=name, precision=prec, npks=K, stages=stages, fft_mode=int(lib.pvx_plan_get_fft_mode(plan)), ms=round(ms, 4),
                              Mframes_per_s=round(F / ms / 1e3, 1), checksum=float(out[: F * K].sum().item()))))
        sys.stdout.flush(); lib.pvx_plan_destroy(plan)
