#!/usr/bin/env python3
"""The tracker alone on BASELINE config 2's analysis arrays (harmonic and white noise): wall time of pvx_track_dev (it returns
the partial count) and the table against the oracle's.   python3 tools/track_time.py [seconds]"""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pypevoc_amd  # noqa: E402
from pypevoc_amd import _lib  # noqa: E402
from oracle import pvoracle  # noqa: E402
from bench import c2_signal  # noqa: E402
secs = int(sys.argv[1]) if len(sys.argv) > 1 else 600
SR, NFFT, HOP, K = 44100, 2048, 512, int(os.environ.get("TRACK_TIME_NPKS", "8"))
lib = _lib.load()
dev = torch.device("cuda:0")
sp = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def run(name, x):
    p = pypevoc_amd.PV(x, SR, nfft=NFFT, hop=HOP, npks=K, progress=False, precision=32)
    p.run_pv()
    F = p.nframes
    f, mag = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (p.f, p.mag))
    nK = F * K
    pid_d = torch.empty(nK, dtype=torch.int32, device=dev); pst_d = torch.empty(nK, dtype=torch.int32, device=dev); pln_d = torch.empty(nK, dtype=torch.int32, device=dev)
    once = lambda: int(lib.pvx_track_dev(f.data_ptr(), mag.data_ptr(), F, K, 0.5, pid_d.data_ptr(), pst_d.data_ptr(), pln_d.data_ptr(), nK, sp))
    P = once()
    tt = []
    for _ in range(20):
        torch.cuda.synchronize(dev); t0 = time.perf_counter(); once(); tt.append(time.perf_counter() - t0)
    opid, ost, oln = pvoracle.track(p.f, p.mag)
    same = P == len(ost) and np.array_equal(pid_d.cpu().numpy().reshape(F, K), opid) and np.array_equal(pst_d[:P].cpu().numpy(), ost) and np.array_equal(pln_d[:P].cpu().numpy(), oln)
    print(json.dumps(dict(signal=name, frames=F, partials=P, ms_min=round(min(tt) * 1e3, 4), ms_median=round(sorted(tt)[10] * 1e3, 4), table_identical=bool(same))), flush=True)


x = c2_signal(secs)
run("harmonic", x)
run("white_noise", (0.1 * np.random.default_rng(7).standard_normal(len(x) // 2)).astype(np.float32))
