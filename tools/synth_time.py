#!/usr/bin/env python3
"""The resynthesis alone, device to device, on BASELINE config 2's shape: analysis and tracker once, then pvx_synth_dev timed
with HIP events and the first frames' waveform checked against the oracle.  Signals: the harmonic C2 signal (bodies only),
white noise (short partials: attacks and releases everywhere) and a time-stretched harmonic run (synthesis hop 700: the
pieces of fsig change inside k_synth_bodies' runs unless the cuts are right).
   python3 tools/synth_time.py [seconds] [reps]"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import pypevoc_amd  # noqa: E402
from pypevoc_amd import _lib  # noqa: E402
from oracle import pvoracle  # noqa: E402
from bench import c2_signal  # noqa: E402

secs = int(sys.argv[1]) if len(sys.argv) > 1 else 600
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
SR, NFFT, HOP, K = 44100, 2048, 512, int(os.environ.get("SYNTH_TIME_NPKS", "8"))
lib = _lib.load()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev)
sp = ctypes.c_void_p(stream.cuda_stream)
FLAGS = _lib.PVX_SYNTH_F32 if os.environ.get("SYNTH_TIME_F32") else 0      # the float32 sample loop (what a precision-32 plan resynthesises with)
TOL = 1e-4 if FLAGS else 1e-9


def run(name, x, hop_s, check_frames=1500):
    p = pypevoc_amd.PV(x, SR, nfft=NFFT, hop=HOP, npks=K, progress=False, precision=32)
    p.run_pv()
    F = p.nframes
    f, mag, rp = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (p.f, p.mag, p.realph))
    nK = F * K
    pid_d = torch.empty(nK, dtype=torch.int32, device=dev)
    pst_d = torch.empty(nK, dtype=torch.int32, device=dev)
    pln_d = torch.empty(nK, dtype=torch.int32, device=dev)
    P = int(lib.pvx_track_dev(f.data_ptr(), mag.data_ptr(), F, K, 0.5, pid_d.data_ptr(), pst_d.data_ptr(), pln_d.data_ptr(), nK, sp))
    assert P > 0, P
    maxend = int((pst_d[:P].to(torch.int64) + pln_d[:P].to(torch.int64) - 1).max().item())
    wlen = int(lib.pvx_synth_len(maxend, NFFT, HOP, hop_s, 1.0))
    w_d = torch.empty(wlen, dtype=torch.float64, device=dev)

    def once():
        _lib.check(lib.pvx_synth_dev_flags(f.data_ptr(), mag.data_ptr(), rp.data_ptr(), pid_d.data_ptr(), F, K, pst_d.data_ptr(), pln_d.data_ptr(), P,
                                           float(SR), NFFT, HOP, hop_s, 1.0, 3, w_d.data_ptr(), wlen, sp, FLAGS), "pvx_synth_dev_flags")

    once()
    torch.cuda.synchronize(dev)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        once()
    e1.record(stream)
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / reps
    if os.environ.get("SYNTH_TIME_NOCHECK"):
        print(json.dumps(dict(signal=name, frames=F, partials=P, hop_s=hop_s, ms=round(ms, 4))), flush=True)
        return
    FC = min(F, check_frames)
    hf, hm, hr = p.f[:FC], p.mag[:FC], p.realph[:FC]
    cpid, cst, cln = pvoracle.track(hf, hm)
    ow = pvoracle.synth(hf, hm, hr, cpid, cst, cln, SR, NFFT, HOP, hop_s)
    ncmp = (FC - 8) * hop_s
    hw = w_d[:ncmp].cpu().numpy()
    err = float(np.abs(hw - ow[:ncmp]).max())
    print(json.dumps(dict(signal=name, frames=F, partials=P, hop_s=hop_s, samples=wlen, ms=round(ms, 4), frames_per_s=round(F / ms * 1e3, 1),
                          max_abs_err=err, peak=float(np.abs(ow[:ncmp]).max()), sample_loop="f32" if FLAGS else "f64", ok=bool(err <= TOL * max(1.0 if not FLAGS else 0.0, float(np.abs(ow[:ncmp]).max()))))), flush=True)


x = c2_signal(secs)
run("harmonic", x, 512)
if os.environ.get("SYNTH_TIME_ONLY") == "harmonic":
    sys.exit(0)
run("harmonic_stretch700", x[: len(x) // 4], 700)
run("white_noise", (0.1 * np.random.default_rng(7).standard_normal(len(x) // 2)).astype(np.float32), 512)
