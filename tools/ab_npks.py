#!/usr/bin/env python3
"""Frames/s of the analysis stage by npks (the reference's default is 20) on BASELINE config 2's signal, plan default fft mode.
   python tools/ab_npks.py [nfft] [npks,...]"""
import ctypes, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pypevoc_amd import _lib
from bench import c2_signal
nfft = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ks = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "4,8,12,16,20,24,32,48,64").split(",")]
lib = _lib.load(); _lib.init(0)
dev = torch.device("cuda", 0); s = torch.cuda.Stream(device=dev); sp = ctypes.c_void_p(s.cuda_stream)
x = torch.from_numpy(c2_signal(600)).to(dev)
hop = nfft // 4
for K in ks:
    n = x.numel(); F = int(lib.pvx_nframes(n, nfft, hop))
    out = torch.zeros(5 * F * K + 2 * F, dtype=torch.float64, device=dev); b = out.data_ptr()
    ptrs = [b + i * F * K * 8 for i in range(5)] + [b + 5 * F * K * 8, b + 5 * F * K * 8 + F * 8]
    plan = ctypes.c_void_p(); win = np.hanning(nfft)
    _lib.check(lib.pvx_plan_create(ctypes.byref(plan), 44100.0, nfft, hop, K, 0.005, _lib.dptr(win), 32, 0), "plan")
    for _ in range(2): _lib.check(lib.pvx_analyze_dev(plan, x.data_ptr(), 0, n, 1, n, *ptrs, None, sp), "a")
    torch.cuda.synchronize()
    t_r = time.perf_counter() + 0.15
    while time.perf_counter() < t_r:
        for _ in range(16): lib.pvx_analyze_dev(plan, x.data_ptr(), 0, n, 1, n, *ptrs, None, sp)
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); reps = 10
    e0.record(s)
    for _ in range(reps): lib.pvx_analyze_dev(plan, x.data_ptr(), 0, n, 1, n, *ptrs, None, sp)
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps(dict(nfft=nfft, npks=K, fft_mode=int(lib.pvx_plan_get_fft_mode(plan)), ms=round(ms, 4), Mframes_per_s=round(F / ms / 1e3, 1),
                          peaks_found=int((out[:F * K] > 0).sum().item()))), flush=True)
    lib.pvx_plan_destroy(plan)
