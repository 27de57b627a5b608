"""Does a large pageable host -> device transfer slow down from call to call?   python tools/ab/time_h2d.py"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pypevoc_amd
from pypevoc_amd import SoundUtils
from bench import c2_signal
x = c2_signal(600).astype(np.float64)
for it in range(8):
    t0 = time.perf_counter(); SoundUtils.RMSWind(x, 44100, 2048, 512); print("RMSWind float64 212 MB: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
x32 = x.astype(np.float32)
for it in range(8):
    p = pypevoc_amd.PV(x32, 44100, nfft=2048, hop=512, npks=8, progress=False)
    t0 = time.perf_counter(); p.run_pv(); print("PV.run_pv float32 106 MB: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
import torch
xt = torch.from_numpy(x)
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); y = xt.to("cuda"); torch.cuda.synchronize(); print("torch .to(cuda) 212 MB: %.2f ms" % ((time.perf_counter() - t0) * 1e3)); del y
