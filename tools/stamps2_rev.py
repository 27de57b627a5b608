"""Where a wave of k_fused_rev (nfft 2048) spends its time: s_memtime stamps of an instrumented build (round-5 sources).
   bash tools/ab/buildstamp2_rev.sh && PVX_ALLOW_STALE_LIB=1 PVX_LIB=tools/ab/libpvx_st.so python tools/stamps2_rev.py [noise|violin]"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PVX_STAMPS"] = "1"
import bench
from pypevoc_amd import _lib
lib = _lib.load(); _lib.init(0)
dev = torch.device("cuda", 0)
what = sys.argv[1] if len(sys.argv) > 1 else "harmonic"
if what == "noise": x = 0.1 * torch.randn(44100 * 600, device=dev)
elif what == "violin":
    xv = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G7_perlman.npz"))["x"].astype(np.float32)
    x = torch.from_numpy(np.tile(xv, 44100 * 600 // len(xv) + 1)[: 44100 * 600]).to(dev)
else: x = torch.from_numpy(bench.c2_signal(600)).to(dev)
nsamp = x.numel()
F = int(lib.pvx_nframes(nsamp, 2048, 512)); K = 8
packed = torch.empty(5 * F * K + 2 * F, dtype=torch.float64, device=dev); base = packed.data_ptr()
ptrs = [base + i * F * K * 8 for i in range(5)] + [base + 5 * F * K * 8, base + 5 * F * K * 8 + F * 8]
plan = ctypes.c_void_p(); win = np.hanning(2048)
_lib.check(lib.pvx_plan_create(ctypes.byref(plan), 44100.0, 2048, 512, K, 0.005, _lib.dptr(win), 32, 0), "plan")
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.pvx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
for _ in range(3):
    lib.pvx_analyze_dev(plan, x.data_ptr(), 0, nsamp, 1, nsamp, *ptrs, None, None)
torch.cuda.synchronize()
raw.pvx_debug_stamps(plan, None, 1)
n = 5
for _ in range(n):
    lib.pvx_analyze_dev(plan, x.data_ptr(), 0, nsamp, 1, nsamp, *ptrs, None, None)
NS = 18
out = (ctypes.c_uint64 * NS)()
raw.pvx_debug_stamps(plan, out, 0)
names = ["0 loop top, row bookkeeping, window reads + multiply", "1 slide + sample loads issued", "2 stage 1 (radix-16)",
         "3 twiddle reads + products + exchange write + reads issued", "4 exchange reads landed + stage 2", "5 natural-order store + join twiddles, drained",
         "6 join + untangle (both halves) + the row's tail", "7 stash / previous-spectrum pick-up (+ flush every 8th frame)", "8 candidate scan + list",
         "9 fetch, rank, salience, staging", "10 loop bottom", "11 (after the loop)", "12 (of 0) loop top + window reads landed", "13 (of 0) the row's samples have arrived (vmcnt(0))", "14 (of 0) back edge: the top of the loop body", "15 (of 0) row bookkeeping + explicit vmcnt(0): samples in registers", "16 (PVX_STAMP_LAT=1) slide loads issued", "17 (PVX_STAMP_LAT=1) ... and landed: their raw latency"]
tot = sum(out)
frames = n * (F + 1 + 256)       # (+ the row under a workgroup's range that its wave 0 transforms)
for i in range(NS):
    print("%-62s %8.1f ticks/frame  %5.1f %%" % (names[i], out[i] / frames, 100.0 * out[i] / max(tot, 1)))
print("total %.1f s_memtime ticks per frame and wave (100 MHz ticks; three waves share a SIMD); input: %s" % (tot / frames, what))
