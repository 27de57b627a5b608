#!/usr/bin/env python3
"""k_stft_pv (one launch) against k_stft + k_phase_peaks (PVX_NO_STFT_PV=1): bitwise comparison of the result arrays
and time per pass, harmonic and noise input, in child processes (the switch is read when a plan is created).
   python tools/ab_stft_pv.py [nfft,...] [precision]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes, json, os, sys, hashlib
import numpy as np, torch
sys.path.insert(0, %r)
from pypevoc_amd import _lib
from bench import c2_signal
nfft, prec, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lib = _lib.load(); _lib.init(0)
dev = torch.device("cuda", 0); s = torch.cuda.Stream(device=dev); sp = ctypes.c_void_p(s.cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(1)
secs = 120
inputs = {"harmonic": torch.from_numpy(c2_signal(secs)).to(dev), "noise": 0.1 * torch.randn(44100 * secs, device=dev, generator=g),
          "harmonic_f64": torch.from_numpy(c2_signal(secs)).to(dev).double()}
hop = nfft // 4
for name, x in inputs.items():
    n = x.numel(); F = int(lib.pvx_nframes(n, nfft, hop))
    out = torch.zeros(5 * F * K + 2 * F, dtype=torch.float64, device=dev); b = out.data_ptr()
    ptrs = [b + i * F * K * 8 for i in range(5)] + [b + 5 * F * K * 8, b + 5 * F * K * 8 + F * 8]
    plan = ctypes.c_void_p(); win = np.hanning(nfft)
    _lib.check(lib.pvx_plan_create(ctypes.byref(plan), 44100.0, nfft, hop, K, 0.005, _lib.dptr(win), prec, 0), "plan")
    if prec == 32: lib.pvx_plan_set_fft_mode(plan, 0)
    dt = 1 if x.dtype == torch.float64 else 0
    for _ in range(2): _lib.check(lib.pvx_analyze_dev(plan, x.data_ptr(), dt, n, 1, n, *ptrs, None, sp), "a")
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); reps = 10
    e0.record(s)
    for _ in range(reps): lib.pvx_analyze_dev(plan, x.data_ptr(), dt, n, 1, n, *ptrs, None, sp)
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    h = out.cpu().numpy()
    print(json.dumps(dict(nfft=nfft, K=K, input=name, precision=prec, ms=round(ms, 4), Mframes_per_s=round(F / ms / 1e3, 1),
                          sha_peaks=hashlib.sha1(h[: 5 * F * K].tobytes()).hexdigest()[:12], sha_t=hashlib.sha1(h[5 * F * K: 5 * F * K + F].tobytes()).hexdigest()[:12],
                          totalmag_sum=float(h[5 * F * K + F:].sum()))))
    lib.pvx_plan_destroy(plan)
''' % ROOT
nffts = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512,1024,2048").split(",")]
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 64
Ks = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "8").split(",")]
bad = 0
for nfft in nffts:
    for K in Ks:
        res = {}
        for mode in ("fused", "two"):
            env = dict(os.environ)
            if mode == "two": env["PVX_NO_STFT_PV"] = "1"
            r = subprocess.run([sys.executable, "-c", CHILD, str(nfft), str(prec), str(K)], env=env, capture_output=True, text=True, timeout=600)
            if r.returncode != 0: print(mode, "FAILED", r.stderr[-2000:]); bad += 1; continue
            res[mode] = [json.loads(l) for l in r.stdout.strip().splitlines()]
        if len(res) == 2:
            for a, b in zip(res["fused"], res["two"]):
                same = a["sha_peaks"] == b["sha_peaks"] and a["sha_t"] == b["sha_t"] and a["totalmag_sum"] == b["totalmag_sum"]
                bad += 0 if same else 1
                print(json.dumps(dict(nfft=nfft, K=K, input=a["input"], precision=prec, identical=same, fused_ms=a["ms"], two_kernel_ms=b["ms"],
                                      fused_Mfps=a["Mframes_per_s"], two_kernel_Mfps=b["Mframes_per_s"])))
                sys.stdout.flush()
sys.exit(1 if bad else 0)
