#!/usr/bin/env python3
"""What a launch of the nfft-2048 analysis kernel costs beyond its rows: ms per launch over signal lengths, fitted as
a + b * (rows per wave) -- `a` is the fixed cost (dispatch gap between dependent launches, table set-up, first window, tail).
   python tools/launch_overhead.py [precision]"""
import ctypes, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pypevoc_amd import _lib
from bench import c2_signal
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lib = _lib.load(); _lib.init(0)
dev = torch.device("cuda", 0); s = torch.cuda.Stream(device=dev); sp = ctypes.c_void_p(s.cuda_stream)
nfft, hop, K = 2048, 512, 8
xfull = torch.from_numpy(np.tile(c2_signal(600), 4)).to(dev)
rows, ms = [], []
for secs in (75, 150, 300, 600, 1200, 2400):
    n = 44100 * secs
    x = xfull[:n]
    F = int(lib.pvx_nframes(n, nfft, hop))
    out = torch.zeros(5 * F * K + 2 * F, dtype=torch.float64, device=dev); b = out.data_ptr()
    ptrs = [b + i * F * K * 8 for i in range(5)] + [b + 5 * F * K * 8, b + 5 * F * K * 8 + F * 8]
    plan = ctypes.c_void_p(); win = np.hanning(nfft)
    _lib.check(lib.pvx_plan_create(ctypes.byref(plan), 44100.0, nfft, hop, K, 0.005, _lib.dptr(win), prec, 0), "plan")
    for _ in range(3): _lib.check(lib.pvx_analyze_dev(plan, x.data_ptr(), 0, n, 1, n, *ptrs, None, sp), "a")
    torch.cuda.synchronize()
    t_r = time.perf_counter() + 0.2
    while time.perf_counter() < t_r:
        for _ in range(16): lib.pvx_analyze_dev(plan, x.data_ptr(), 0, n, 1, n, *ptrs, None, sp)
        torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); reps = 20
        e0.record(s)
        for _ in range(reps): lib.pvx_analyze_dev(plan, x.data_ptr(), 0, n, 1, n, *ptrs, None, sp)
        e1.record(s); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    # one launch alone (events right around it, the stream idle before): no dependent-launch gap in it
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(s); lib.pvx_analyze_dev(plan, x.data_ptr(), 0, n, 1, n, *ptrs, None, sp); e1.record(s); torch.cuda.synchronize()
    rows.append((F + 1) / 3072.0); ms.append(best)
    print(json.dumps(dict(seconds=secs, frames=F, rows_per_wave=round((F + 1) / 3072.0, 2), ms_back_to_back=round(best, 4), ms_single=round(e0.elapsed_time(e1), 4),
                          Mframes_per_s=round(F / best / 1e3, 1))))
    lib.pvx_plan_destroy(plan)
A = np.vstack([np.ones(len(rows)), np.ceil(rows)]).T
a, b = np.linalg.lstsq(A, np.array(ms), rcond=None)[0]
print(json.dumps(dict(fit="ms = a + b * ceil(rows per wave)", a_us=round(a * 1e3, 2), b_us_per_row=round(b * 1e3, 3), precision=prec)))
