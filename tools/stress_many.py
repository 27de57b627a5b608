"""pvx_batch_run under load: 155 ragged signals, device lists of 1 .. 4 entries (device 0 repeated) x 1 .. 3 workers each,
18 runs on kept handles -- every result compared bit for bit with the single-signal PV(...).run_pv().
   python tools/stress_many.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import pypevoc_amd
rng = np.random.default_rng(3)
sr, nfft, hop, K = 22050, 1024, 256, 6
lens = [int(v) for v in rng.integers(900, 400000, 150)] + [1500000, 2500000, 700, 1024, 1025]
sigs = []
for i, n in enumerate(lens):
    t = np.arange(n) / sr
    sigs.append((0.3 * np.sin(2 * np.pi * (200 + 3 * i) * t) + 0.02 * rng.standard_normal(n)).astype(np.float32))
ref = []
for x in sigs:
    p = pypevoc_amd.PV(x, sr, nfft=nfft, hop=hop, npks=K, progress=False)
    p.run_pv(); ref.append((p.f.copy(), p.mag.copy(), p.realph.copy()))
bad = 0
t0 = time.time()
for rep in range(6):
    many = pypevoc_amd.PVMany(sr, nfft=nfft, hop=hop, npks=K, devices=[0] * (1 + rep % 4), workers_per_device=1 + rep % 3)
    for inner in range(3):
        res = many.run(sigs)
        for r, (f, m, rp) in zip(res, ref):
            if r["nframes"] == 0:
                continue
            if not (np.array_equal(r["f"], f) and np.array_equal(r["mag"], m) and np.array_equal(r["realph"], rp)):
                bad += 1
    many.close()
print("stress: %d signals x 18 runs, mismatches %d, %.1f s" % (len(sigs), bad, time.time() - t0))
