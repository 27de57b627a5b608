#!/usr/bin/env python3
"""In-process sweep of launch parameters (PVX_MAX_ROWS, PVX_FPW, fft mode) on the bench workload.
Usage: python tools/tune.py "rows=6144,16384 fpw=1,2,4,8 mode=0" [seconds]"""
import ctypes
import itertools
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pypevoc_amd import _lib  # noqa: E402


def main():
    spec = sys.argv[1] if len(sys.argv) > 1 else "rows=6144 fpw=4"
    seconds = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    opts = dict(kv.split("=") for kv in spec.split())
    rows = [int(v) for v in opts.get("rows", "6144").split(",")]
    fpws = [int(v) for v in opts.get("fpw", "4").split(",")]
    modes = [int(v) for v in opts.get("mode", "0").split(",")]
    lib = _lib.load()
    _lib.init(0)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(bench.c2_signal(seconds)).to(dev)
    nsamp = x.numel()
    F = int(lib.pvx_nframes(nsamp, bench.NFFT, bench.HOP))
    K = bench.NPKS
    packed = torch.empty(5 * F * K + 2 * F, dtype=torch.float64, device=dev)
    base = packed.data_ptr()
    ptrs = [base + i * F * K * 8 for i in range(5)] + [base + 5 * F * K * 8, base + 5 * F * K * 8 + F * 8]
    win = np.hanning(bench.NFFT)
    stream = torch.cuda.current_stream(dev)
    ref = None
    for rnd in range(2):
        for r, fpw, mode in itertools.product(rows, fpws, modes):
            os.environ["PVX_MAX_ROWS"] = str(r)
            os.environ["PVX_FPW"] = str(fpw)
            plan = ctypes.c_void_p()
            _lib.check(lib.pvx_plan_create(ctypes.byref(plan), float(bench.SR), bench.NFFT, bench.HOP, K, 0.005,
                                           _lib.dptr(win), 32, 0), "plan")
            rc = lib.pvx_plan_set_fft_mode(plan, mode)
            if rc < 0:
                print("mode %d unavailable" % mode)
                lib.pvx_plan_destroy(plan)
                continue

            def step():
                _lib.check(lib.pvx_analyze_dev(plan, x.data_ptr(), 0, nsamp, 1, nsamp, *ptrs, None,
                                               ctypes.c_void_p(stream.cuda_stream)), "analyze")
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            out = packed.clone()
            if ref is None:
                ref = out
            same = bool(torch.equal(out, ref))
            maxdiff = float((out - ref).abs().max())
            lib.pvx_plan_set_timing(plan, 1)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record(stream)
            for _ in range(n):
                step()
            e1.record(stream)
            torch.cuda.synchronize()
            ms = (ctypes.c_double * 4)(); nl = (ctypes.c_int64 * 4)()
            lib.pvx_plan_get_timing(plan, ms, nl)
            lib.pvx_plan_set_timing(plan, 0)
            e2 = torch.cuda.Event(enable_timing=True); e3 = torch.cuda.Event(enable_timing=True)
            e2.record(stream)
            for _ in range(n):
                step()
            e3.record(stream)
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / n
            t2 = e2.elapsed_time(e3) / n
            print("round %d rows=%6d fpw=%d mode=%d: %.4f ms/step (%.4f without stage events) = %.1f Mframes/s | "
                  "frames %.4f fft %.4f peaks %.4f fused %.4f ms/step | same_as_first=%s maxdiff=%.3g" %
                  (rnd, r, fpw, mode, t, t2, F / t2 / 1e3, ms[0] / n, ms[1] / n, ms[2] / n, ms[3] / n, same, maxdiff))
            sys.stdout.flush()
            lib.pvx_plan_destroy(plan)


if __name__ == "__main__":
    main()
