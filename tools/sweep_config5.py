#!/usr/bin/env python3
"""BASELINE config 5: nfft in {512..8192} x hop in {nfft/4, nfft/2} on a 60-min 96 kHz signal
(345.6 M samples, 1.38 GB float32, resident in HBM), npks=8.  Prints one JSON line per point: frames/s, the fraction
of the 8 TB/s HBM roofline at the fused kernels' own algorithmic bytes (hop*4 in + (5 npks + 2)*8 out), and the
throughput against north_star's target (60 % of the roofline at SURVEY.md 8(d)'s three-kernel bytes).
   python tools/sweep_config5.py [seconds] [precision]"""
import ctypes, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pypevoc_amd import _lib

def signal(n, sr):
    # G4 generator evaluated in float32 chunks on the GPU (the 1.4 GB signal never exists on the host)
    dev = torch.device("cuda", 0)
    x = torch.empty(n, dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(1234)
    step = 1 << 24
    for a in range(0, n, step):
        b = min(n, a + step)
        t = torch.arange(a, b, device=dev, dtype=torch.float64) / sr
        ph = 2 * np.pi * 220.0 * (t - 0.01 / (2 * np.pi * 5.0) * torch.cos(2 * np.pi * 5.0 * t))
        s = torch.zeros(b - a, dtype=torch.float64, device=dev)
        for h in range(1, 9):
            s += 0.3 / h * torch.sin(h * ph)
        s += 0.001 * torch.randn(b - a, generator=g, device=dev, dtype=torch.float64)
        x[a:b] = s.to(torch.float32)
    return x

def main():
    sr, secs, K = 96000, int(sys.argv[1]) if len(sys.argv) > 1 else 3600, 8
    prec = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    lib = _lib.load(); _lib.init(0)
    dev = torch.device("cuda", 0)
    n = sr * secs
    x = signal(n, sr)
    stream = torch.cuda.current_stream(dev)
    for nfft in [int(v) for v in os.environ.get("PVX_SWEEP_NFFT", "512,1024,2048,4096,8192").split(",")]:      # PVX_SWEEP_NFFT: a subset
        for hop in (nfft // 4, nfft // 2):
            F = int(lib.pvx_nframes(n, nfft, hop))
            out = torch.empty(5 * F * K + 2 * F, dtype=torch.float64, device=dev)
            base = out.data_ptr()
            ptrs = [base + i * F * K * 8 for i in range(5)] + [base + 5 * F * K * 8, base + 5 * F * K * 8 + F * 8]
            plan = ctypes.c_void_p()
            win = np.hanning(nfft)
            _lib.check(lib.pvx_plan_create(ctypes.byref(plan), float(sr), nfft, hop, K, 0.005, _lib.dptr(win), prec, 0), "plan")
            def step():
                _lib.check(lib.pvx_analyze_dev(plan, x.data_ptr(), 0, n, 1, n, *ptrs, None, ctypes.c_void_p(stream.cuda_stream)), "analyze")
            step(); torch.cuda.synchronize()
            import time
            t_r = time.perf_counter() + 0.15                  # clock ramp (bench.py): the card leaves its idle clocks
            while time.perf_counter() < t_r:
                step(); step(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record(stream)
            for _ in range(reps): step()
            e1.record(stream); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            s_ = 4 if prec == 32 else 8
            contract = hop * 4 + 2 * nfft * s_ + 2 * (nfft // 2 + 1) * 2 * s_ + (K * 5 * 8 + 16)
            mode = int(lib.pvx_plan_get_fft_mode(plan))
            own = hop * 4 + (K * 5 * 8 + 16) if mode != 0 else contract
            kern = (lib.pvx_plan_last_kernels(plan) or b"").decode()
            if mode == 0 and prec == 64 and "k_pv_rev" in kern:             # rows walked downwards, the row on chip: the fused bytes
                own = hop * 4 + (K * 5 * 8 + 16)
            elif mode == 0 and prec == 64 and nfft in (512, 1024, 2048):    # k_stft_pv: spectrum rows written once, never read back
                own = hop * 4 + (nfft // 2) * 16 + (K * 5 * 8 + 16)
            elif mode == 0 and prec == 64 and nfft in (4096, 8192):         # k_stft_split + k_phase_peaks: written once, read once
                own = hop * 4 + 2 * (nfft // 2) * 16 + (K * 5 * 8 + 16)
            fps = F / (ms * 1e-3)
            print(json.dumps(dict(nfft=nfft, hop=hop, precision=prec, frames=F, ms=round(ms, 3), frames_per_s=round(fps, 1),
                                  fft_mode=mode, kernels=kern if mode == 0 else None, alg_bytes_per_frame=own, achieved_GBps=round(fps * own / 1e9, 1),
                                  frac_of_8TBps=round(fps * own / 8e12, 4), contract_bytes_per_frame=contract,
                                  throughput_vs_60pct_target=round(fps * contract / (0.6 * 8e12), 3))))
            sys.stdout.flush()
            lib.pvx_plan_destroy(plan)
            del out
            torch.cuda.empty_cache()

if __name__ == "__main__":
    main()
