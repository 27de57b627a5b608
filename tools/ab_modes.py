#!/usr/bin/env python3
"""A/B of the fused analysis kernels (fft modes 1, 2, 3) on the bench workload (BASELINE config 2: 10 min of
44.1 kHz, nfft 2048, hop 512, npks 8) and on white noise: ms per pass, frames/s, and whether every output
array is bit-identical to mode 1's.   python tools/ab_modes.py [modes, e.g. 1,3] [seconds]"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pypevoc_amd import _lib  # noqa: E402
from bench import c2_signal  # noqa: E402

modes = [int(m) for m in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3").split(",")]
seconds = int(sys.argv[2]) if len(sys.argv) > 2 else 600
lib = _lib.load(); _lib.init(0)
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev)
sp = ctypes.c_void_p(s.cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(1)
inputs = {"c2_harmonic": torch.from_numpy(c2_signal(seconds)).to(dev),
          "white_noise": 0.1 * torch.randn(44100 * seconds, device=dev, generator=g)}
nfft, hop = 2048, 512
for K in (8, 20):
    for name, x in inputs.items():
        nsamp = x.numel()
        F = int(lib.pvx_nframes(nsamp, nfft, hop))
        ref = None
        for mode in modes:
            out = torch.zeros(5 * F * K + 2 * F, dtype=torch.float64, device=dev); b = out.data_ptr()
            ptrs = [b + i * F * K * 8 for i in range(5)] + [b + 5 * F * K * 8, b + 5 * F * K * 8 + F * 8]
            plan = ctypes.c_void_p(); win = np.hanning(nfft)
            _lib.check(lib.pvx_plan_create(ctypes.byref(plan), 44100.0, nfft, hop, K, 0.005, _lib.dptr(win), 32, 0), "plan")
            rc = lib.pvx_plan_set_fft_mode(plan, mode)
            if rc != 0:
                print(json.dumps(dict(input=name, K=K, mode=mode, error=lib.pvx_last_error().decode() if lib.pvx_last_error() else rc)))
                lib.pvx_plan_destroy(plan)
                continue
            for _ in range(3):
                _lib.check(lib.pvx_analyze_dev(plan, x.data_ptr(), 0, nsamp, 1, nsamp, *ptrs, None, sp), "analyze")
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record(s)
            for _ in range(n):
                lib.pvx_analyze_dev(plan, x.data_ptr(), 0, nsamp, 1, nsamp, *ptrs, None, sp)
            e1.record(s); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            same = None
            if ref is None:
                ref = out.clone()
            else:
                same = bool(torch.equal(out.view(torch.int64), ref.view(torch.int64)))
            print(json.dumps(dict(input=name, K=K, mode=mode, ms=round(ms, 4), Mframes_per_s=round(F / ms / 1e3, 1),
                                  peaks_per_frame=round(int((out[: F * K] > 0).sum().item()) / F, 2), bit_identical_to_first=same)))
            sys.stdout.flush()
            lib.pvx_plan_destroy(plan)
