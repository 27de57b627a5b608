#!/usr/bin/env python3
"""pvx_batch_run (include/pvx.h; pypevoc_amd.PVMany) on BASELINE config 4's shard shape from HOST buffers: 128 signals x 30 s
@ 48 kHz float32, nfft 2048, hop 512, npks 8 -- results into the caller's arrays, so the rate is PCIe-inclusive (results are
2.7x the input).  Beside it the same signals through one pvx_analyze call (PVBatch: equal lengths, one buffer) and a ragged
batch.  Workers per device 1 / 2 / 3; the device list repeats device 0 when asked (`--slots N`) to show what several queues
on one card give.
   python3 tools/batch_time.py [--signals 128] [--seconds 30] [--reps 3]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pypevoc_amd  # noqa: E402
from pypevoc_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--signals", type=int, default=128)
ap.add_argument("--seconds", type=float, default=30.0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--slots", type=int, default=1)
a = ap.parse_args()
SR, NFFT, HOP, K = 48000, 2048, 512, 8
n = int(a.seconds * SR)
rng = np.random.default_rng(5)
t = np.arange(n) / SR
x = np.empty((a.signals, n), dtype=np.float32)
for b in range(a.signals):
    f0 = 110.0 * 2 ** (b / 1024.0 * 3)
    ph = 2 * np.pi * f0 * t
    x[b] = (sum(0.3 / h * np.sin(h * ph) for h in range(1, 9)) + 0.001 * rng.standard_normal(n)).astype(np.float32)
F = _lib.nframes_host(n, NFFT, HOP)


def best(fn):
    fn()
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


pb = pypevoc_amd.PVBatch(x, SR, nfft=NFFT, hop=HOP, npks=K, precision=32)
s = best(pb.run_pv)
print(json.dumps(dict(path="pvx_analyze nsig=%d (one buffer)" % a.signals, frames=a.signals * F, s=round(s, 4), frames_per_s=round(a.signals * F / s, 1))), flush=True)
sigs = [x[b] for b in range(a.signals)]
for w in (1, 2, 3):
    many = pypevoc_amd.PVMany(SR, nfft=NFFT, hop=HOP, npks=K, devices=[0] * a.slots, workers_per_device=w)
    s = best(lambda: many.run(sigs))
    r = many.run(sigs)
    ok = all(np.array_equal(r[b]["f"], pb.f[b]) for b in (0, a.signals // 2, a.signals - 1))
    print(json.dumps(dict(path="pvx_batch_run", slots=a.slots, workers_per_device=w, frames=a.signals * F, s=round(s, 4),
                          frames_per_s=round(a.signals * F / s, 1), equal_to_one_buffer=bool(ok))), flush=True)
    many.close()
rag = [x[b][: int(n * (0.2 + 0.8 * ((b * 37) % a.signals) / a.signals))] for b in range(a.signals)]
fr = sum(_lib.nframes_host(len(v), NFFT, HOP) for v in rag)
many = pypevoc_amd.PVMany(SR, nfft=NFFT, hop=HOP, npks=K, devices=[0] * a.slots, workers_per_device=2)
s = best(lambda: many.run(rag))
print(json.dumps(dict(path="pvx_batch_run ragged (6..30 s)", slots=a.slots, workers_per_device=2, frames=fr, s=round(s, 4), frames_per_s=round(fr / s, 1))), flush=True)
