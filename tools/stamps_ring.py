import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PVX_STAMPS"] = "1"; os.environ.setdefault("PVX_FFT_MODE", "3")
import bench
from pypevoc_amd import _lib
lib = _lib.load(); _lib.init(0)
dev = torch.device("cuda", 0)
noise = len(sys.argv) > 1 and sys.argv[1] == "noise"       # python tools/stamps.py noise: dense-candidate input
x = (0.1 * torch.randn(44100 * 600, device=dev)) if noise else torch.from_numpy(bench.c2_signal(600)).to(dev); nsamp = x.numel()
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
out = (ctypes.c_uint64 * 16)()
raw.pvx_debug_stamps(plan, out, 0)
names = ["0 loads+window", "1 dft16 #1", "2 twiddle+exchange", "3 dft16 #2", "4 tw2+stage3+Zwrite", "5 untangle+mags", "6 reductions", "7 peak_select", "8 salience+staging", "9 flush", "10 pre-load (addr, loop top, swap)", "11 prefetch issue", "12 barrier 1 (spectra complete)", "13 barrier 2 (peaks done)", "14", "15"]
tot = sum(out)
frames = n * (F + 256)
for i in range(16):
    print("%-36s %8.1f cycles/frame  %5.1f %%" % (names[i], out[i] / frames, 100.0 * out[i] / max(tot, 1)))
print("total %.1f cycles/frame (s_memtime ticks)" % (tot / frames))
