"""Wall time of BASELINE config 2 through the whole path (host signal in, host waveform out), the previous round's
objects released outside the clock.   python tools/time_c2_chain.py"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pypevoc_amd, bench
x = bench.c2_signal()
p = ss = w = None
for it in range(5):
    p = ss = w = None; gc.collect()
    p = pypevoc_amd.PV(x, bench.SR, nfft=2048, hop=512, npks=8, progress=False)
    t0 = time.perf_counter(); p.run_pv(); t1 = time.perf_counter(); ss = p.toSinSum(); t2 = time.perf_counter(); w = ss.synth(bench.SR, 512); t3 = time.perf_counter()
    print("run_pv %.3f toSinSum %.3f synth %.3f ms  total %.3f" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t3-t0)*1e3))
