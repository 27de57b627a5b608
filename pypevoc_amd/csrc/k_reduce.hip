// k_reduce.hip -- hop-strided windowed reductions with the analysis framing:
//   heterodyne(x, hetsig, wind, hop)   pypevoc/Heterodyne.py:35-60
//       out[i] = 2 * sum_j x[i*hop+j] * hetsig[i*hop+j] * wind[j] / sum(wind)      (complex)
//   RMSWind(x, sr, nwind, nhop, windfunc)   pypevoc/SoundUtils.py:71-103
//       out[i] = sqrt(sum_j (x[i*hop+j] * wind[j])**2 / sum(wind**2))
//   (SoundUtils.Heterodyn / HeterodynWithF0Track, :106-138, are the first with a generated hetsig)
// Frames: i*hop for i*hop < n - wlen (both loops), i.e. ceil((n - wlen) / hop) of them.
//
// One wave64 per frame, float64 throughout: lanes stride over the window with coalesced 8/16-byte
// loads, partial sums are reduced with DPP (wave_sum).  HBM-bound streaming: 8 (x) + 16 (hetsig)
// bytes per sample per frame; with hop = wlen/2 the second read of a sample comes from L2 because
// neighbouring frames are neighbouring waves.  The window (wlen doubles) stays in L2.
#include "pvx_wave.h"

using namespace pvxw;

namespace {

__global__ __launch_bounds__(256) void k_heterodyne(ReduceParams p) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int64_t fr = (int64_t)blockIdx.x * nw + wid;
    if (fr >= p.nfr) return;
    const int64_t pos = fr * (int64_t)p.hop;
    const double* x = p.x + pos;
    const double2* h = (const double2*)p.hetsig + pos;
    double sr = 0.0, si = 0.0;
    for (int j = lane; j < p.wlen; j += 64) {
        const double xv = x[j], w = p.wind[j];
        const double2 hv = h[j];
        // (x * hetsig) * wind, as the reference orders it (Heterodyne.py:54-57)
        sr += (xv * hv.x) * w;
        si += (xv * hv.y) * w;
    }
    sr = wave_sum(sr);
    si = wave_sum(si);
    if (lane == 0) {
        p.out[2 * fr] = sr / p.norm * 2.0;                            // Heterodyne.py:58, 60
        p.out[2 * fr + 1] = si / p.norm * 2.0;
        if (p.icent) p.icent[fr] = pos + p.wlen / 2;                  // Heterodyne.py:59
    }
}

__global__ __launch_bounds__(256) void k_rms_frames(ReduceParams p) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int64_t fr = (int64_t)blockIdx.x * nw + wid;
    if (fr >= p.nfr) return;
    const int64_t pos = fr * (int64_t)p.hop;
    const double* x = p.x + pos;
    double s = 0.0;
    for (int j = lane; j < p.wlen; j += 64) {
        const double xw = x[j] * p.wind[j];
        s += xw * xw / p.norm;                                        // SoundUtils.py:97
    }
    s = wave_sum(s);
    if (lane == 0) p.out[fr] = sqrt(s);                               // SoundUtils.py:103
}

// FuncWind(func, x, sr, nwind, nhop, power, windfunc) for the named reducers (pypevoc/SoundUtils.py:42-69):
//   out[i] = func(x[i*hop : i*hop+wlen] * wind) / norm,   func in {np.sum, np.mean, np.max, np.min, np.std, np.var}
// CPX: x is complex128 (Heterodyn's x * sinsig, SoundUtils.py:112); sum / mean are then complex, std / var real
// (numpy: mean(abs(xw - mean(xw))**2)); max / min of complex frames are refused by the entry point.
// std / var in two passes like numpy's (_var: the mean first, then the mean of the squared deviations); the second
// pass re-reads the frame's samples (L1 / L2).
template <bool CPX>
__global__ __launch_bounds__(256) void k_funcwind(ReduceParams p, int func) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int64_t fr = (int64_t)blockIdx.x * nw + wid;
    if (fr >= p.nfr) return;
    const int64_t pos = fr * (int64_t)p.hop;
    const double* x = p.x + (CPX ? 2 : 1) * pos;
    auto prod = [&](int j, double& re, double& im) {
        const double w = p.wind[j];
        if constexpr (CPX) { const double2 v = ((const double2*)x)[j]; re = v.x * w; im = v.y * w; }
        else { re = x[j] * w; im = 0.0; }
    };
    double sr = 0.0, si = 0.0, mx = -INFINITY, mn = INFINITY;
    bool nan = false;                                                // np.max / np.min propagate NaN
    for (int j = lane; j < p.wlen; j += 64) {
        double re, im;
        prod(j, re, im);
        sr += re; si += im;
        mx = fmax(mx, re); mn = fmin(mn, re);
        nan = nan || (re != re);
    }
    double o0 = 0.0, o1 = 0.0;
    if (func == PVX_FW_MAX || func == PVX_FW_MIN) {
        o0 = (func == PVX_FW_MAX) ? wave_max(mx) : wave_min(mn);
        if (__ballot(nan) != 0ull) o0 = NAN;
    } else {
        sr = wave_sum(sr);
        if constexpr (CPX) si = wave_sum(si);
        if (func == PVX_FW_SUM) { o0 = sr; o1 = si; }
        else {
            const double mr = sr / (double)p.wlen, mi = si / (double)p.wlen;
            if (func == PVX_FW_MEAN) { o0 = mr; o1 = mi; }
            else {
                double q = 0.0;
                for (int j = lane; j < p.wlen; j += 64) {
                    double re, im;
                    prod(j, re, im);
                    const double dr = re - mr, di = im - mi;
                    q += CPX ? dr * dr + di * di : dr * dr;
                }
                q = wave_sum(q) / (double)p.wlen;
                o0 = (func == PVX_FW_STD) ? sqrt(q) : q;
            }
        }
    }
    if (lane == 0) {
        const bool cout = CPX && (func == PVX_FW_SUM || func == PVX_FW_MEAN);
        if (cout) { p.out[2 * fr] = o0 / p.norm; p.out[2 * fr + 1] = o1 / p.norm; }
        else p.out[fr] = o0 / p.norm;
    }
}

}  // namespace

int pvx_launch_funcwind(const ReduceParams& p, int func, bool cpx, hipStream_t s) {
    if (p.nfr <= 0) return PVX_OK;
    const unsigned nb = (unsigned)((p.nfr + 3) / 4);
    if (cpx) hipLaunchKernelGGL(k_funcwind<true>, dim3(nb), dim3(256), 0, s, p, func);
    else hipLaunchKernelGGL(k_funcwind<false>, dim3(nb), dim3(256), 0, s, p, func);
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}

int pvx_launch_reduce(const ReduceParams& p, int mode, hipStream_t s) {
    if (p.nfr <= 0) return PVX_OK;
    const unsigned nb = (unsigned)((p.nfr + 3) / 4);
    if (mode == 0) hipLaunchKernelGGL(k_heterodyne, dim3(nb), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_rms_frames, dim3(nb), dim3(256), 0, s, p);
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}
