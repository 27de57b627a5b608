#!/usr/bin/env python3
"""Timings of the rows SURVEY.md 8(f) marks "next" (N1-N4) on BASELINE config 2's signal (10 min @ 44.1 kHz, nfft 2048,
hop 512, npks 8), through the Python mirrors (host arrays in / out unless stated), each line with the numpy restatement
of the reference's loop timed on a bounded sample beside it.  One JSON line per row:
   python tools/bench_next_rows.py [seconds]          (kernel durations: run it under rocprofv3 --kernel-trace --stats)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pypevoc_amd  # noqa: E402
from pypevoc_amd import Heterodyne, SoundUtils  # noqa: E402
from bench import c2_signal  # noqa: E402

secs = int(sys.argv[1]) if len(sys.argv) > 1 else 600
SR, NFFT, HOP, K = 44100, 2048, 512, 8
x = c2_signal(secs).astype(np.float64)
n = len(x)


def best(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, r


def line(**kw):
    print(json.dumps(kw))
    sys.stdout.flush()


# ---- N1: the batch entry point (PVBatch): 16 signals of 30 s in one call, host in / host out
xb = np.stack([c2_signal(30, SR, seed=100 + b, f0=110.0 * 2 ** (b / 16.0)) for b in range(16)]).astype(np.float32)
b = pypevoc_amd.PVBatch(xb, SR, nfft=NFFT, hop=HOP, npks=K)
ms, _ = best(lambda: b.run_pv(), 5)
Fb = int(b.nframes) * xb.shape[0]
line(row="N1 PVBatch.run_pv", shape="16 x 30 s, host float32 in (85 MB), host arrays out (14 MB)", ms=round(ms, 3), frames=Fb, frames_per_s=round(Fb / ms * 1e3, 1),
     host_bytes_in=int(xb.nbytes))
del b

# ---- N2: f0 and descriptors on the resident (F, K) arrays
p = pypevoc_amd.PV(x, SR, nfft=NFFT, hop=HOP, npks=K, progress=False)
p.run_pv()
F = int(p.nframes)
ms, f0 = best(lambda: p.calc_f0(fmin=50, fmax=10000, thr=0.1))
line(row="N2 PV.calc_f0", shape="resident (F, K) arrays, F floats + F ints back", ms=round(ms, 3), frames=F, frames_per_s=round(F / ms * 1e3, 1),
     alg_bytes_per_frame=K * 16 + 12)
ms, hp = best(lambda: p.calc_harmonic_power())
line(row="N2 PV.calc_harmonic_power", shape="resident (F, K) arrays", ms=round(ms, 3), frames=F, frames_per_s=round(F / ms * 1e3, 1))
# the reference's loop restated in numpy on a bounded sample of the frames (PVAnalysis.py:371-391)
ff = np.asarray(p.f)[:20000]
mm = np.asarray(p.mag)[:20000]
t0 = time.perf_counter()
fm = np.zeros(len(ff))
for i in range(len(ff)):
    ok = (ff[i] > 50) & (ff[i] < 10000) & (mm[i] > mm[i].max() * 0.1)
    idx = np.flatnonzero(ok)
    if len(idx):
        fm[i] = ff[i][idx].min()
cpu_ms = (time.perf_counter() - t0) * 1e3
line(row="N2 calc_f0 cpu", shape="the reference's per-frame loop restated in numpy, first 20000 frames, one thread", ms=round(cpu_ms, 1), frames=len(ff),
     frames_per_s=round(len(ff) / cpu_ms * 1e3, 1))

# ---- N3: PVHarmonic on the f0 track just found (host signal in, host arrays out)
f0t = np.where(f0 > 0, f0, 220.0)
h = pypevoc_amd.PVHarmonic(x, SR, nfft=NFFT, hop=HOP, npks=K, progress=False, precision=64)     # (the constructor copies x like the reference's: not timed)
h.set_f0(np.concatenate([f0t, f0t[-1:]]))
ms, _ = best(lambda: h.run_pv(), 5)
line(row="N3 PVHarmonic.run_pv", shape="float64, host signal in (212 MB), (F, K) arrays out", ms=round(ms, 3), frames=int(h.nframes), frames_per_s=round(h.nframes / ms * 1e3, 1),
     alg_bytes_per_frame=HOP * 8 + (NFFT // 2) * 16 * 2 + K * 24 + 16)
del h

# ---- N4: hop-strided windowed reductions with the analysis framing
ms, (rms, t) = best(lambda: SoundUtils.RMSWind(x, SR, NFFT, HOP))
Fw = len(t)
line(row="N4 SoundUtils.RMSWind", shape="float64 host signal in (212 MB), F values out", ms=round(ms, 3), frames=Fw, frames_per_s=round(Fw / ms * 1e3, 1),
     alg_bytes_per_frame=HOP * 8 + 8)
sinsig = np.exp(-2j * np.pi * np.arange(n) * 220.0 / SR)
ms, (ha, ic) = best(lambda: Heterodyne.heterodyne(x, sinsig, wind=np.blackman(NFFT), hop=HOP), 3)
line(row="N4 Heterodyne.heterodyne", shape="float64 signal + complex128 heterodyning signal in (635 MB), F complex out", ms=round(ms, 3), frames=len(ha),
     frames_per_s=round(len(ha) / ms * 1e3, 1), alg_bytes_per_frame=HOP * 24 + 16)
# the reference's loops restated in numpy on a bounded sample (SoundUtils.py:71-103, Heterodyne.py:35-60)
nfr_c = 20000
w = np.blackman(NFFT)
t0 = time.perf_counter()
r = np.zeros(nfr_c)
for i in range(nfr_c):
    seg = x[i * HOP:i * HOP + NFFT]
    r[i] = np.sqrt(np.sum((seg * w) ** 2) / np.sum(w ** 2))
cpu_ms = (time.perf_counter() - t0) * 1e3
line(row="N4 RMSWind cpu", shape="the reference's per-frame loop restated in numpy, first 20000 frames, one thread", ms=round(cpu_ms, 1), frames=nfr_c,
     frames_per_s=round(nfr_c / cpu_ms * 1e3, 1))
t0 = time.perf_counter()
hc = np.zeros(nfr_c, dtype=complex)
for i in range(nfr_c):
    sl = slice(i * HOP, i * HOP + NFFT)
    hc[i] = 2 * np.sum(x[sl] * sinsig[sl] * w) / np.sum(w)
cpu_ms = (time.perf_counter() - t0) * 1e3
line(row="N4 heterodyne cpu", shape="the reference's per-frame loop restated in numpy, first 20000 frames, one thread", ms=round(cpu_ms, 1), frames=nfr_c,
     frames_per_s=round(nfr_c / cpu_ms * 1e3, 1))
