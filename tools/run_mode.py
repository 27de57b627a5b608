#!/usr/bin/env python3
"""Run the analysis stage a few times in one fft mode on the bench workload (for rocprofv3 --pmc runs).
   python3 tools/run_mode.py MODE [harmonic|noise] [K] [reps] [precision]      (PVX_RUN_NFFT=4096: that nfft, hop = nfft/4)"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pypevoc_amd import _lib  # noqa: E402
from bench import c2_signal  # noqa: E402

mode = int(sys.argv[1])
kind = sys.argv[2] if len(sys.argv) > 2 else "harmonic"
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
precision = int(sys.argv[5]) if len(sys.argv) > 5 else 32
lib = _lib.load(); _lib.init(0)
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev)
sp = ctypes.c_void_p(s.cuda_stream)
if kind == "noise":
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = 0.1 * torch.randn(44100 * 600, device=dev, generator=g)
else:
    x = torch.from_numpy(c2_signal(600)).to(dev)
nfft = int(os.environ.get("PVX_RUN_NFFT", "2048")); hop = nfft // 4      # PVX_RUN_NFFT: another point of config 5's sweep
nsamp = x.numel()
F = int(lib.pvx_nframes(nsamp, nfft, hop))
out = torch.zeros(5 * F * K + 2 * F, dtype=torch.float64, device=dev); b = out.data_ptr()
ptrs = [b + i * F * K * 8 for i in range(5)] + [b + 5 * F * K * 8, b + 5 * F * K * 8 + F * 8]
plan = ctypes.c_void_p(); win = np.hanning(nfft)
_lib.check(lib.pvx_plan_create(ctypes.byref(plan), 44100.0, nfft, hop, K, 0.005, _lib.dptr(win), precision, 0), "plan")
if mode >= 0:
    _lib.check(lib.pvx_plan_set_fft_mode(plan, mode), "mode")
for _ in range(reps):
    _lib.check(lib.pvx_analyze_dev(plan, x.data_ptr(), 0, nsamp, 1, nsamp, *ptrs, None, sp), "analyze")
torch.cuda.synchronize()
print("mode %d %s K=%d: %d frames x %d reps" % (lib.pvx_plan_get_fft_mode(plan), kind, K, F, reps))
lib.pvx_plan_destroy(plan)
