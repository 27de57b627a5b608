// k_harmonic.hip -- PVHarmonic: phase vocoder sampled at the multiples of a given fundamental.
// Replaces PVHarmonic.calc_pv_frame / run_pv (pypevoc/PVAnalysis.py:442-535) on the spectra the
// general path already produces (k_frames -> rocFFT): instead of PeakFinder, frame fr looks at the
// bins round(h*f0bin), h = 1, 2, ... (PV.py:463-464), re-centred on multiples of the measured first
// harmonic once that is known (PV.py:465-470), and reports per harmonic the instantaneous frequency
// (dphase2freq, PV.py:133-148), the 3-bin magnitude (PV.py:481-485) and the phase, plus the residual
// sqrt(total energy - harmonic energy) (PV.py:490).
//
// State: the reference only updates `oldfft` on frames it analyses (f0 > 0 and not NaN, PV.py:509,
// 491), so the "previous spectrum" of a frame is that of the last VALID frame -- `prevrow`, computed
// on the host from f0: a row of this launch's workspace, the `carry` row saved from an earlier
// launch, or zero / `prev0` if there is none.
//
// One wave64 per frame.  HBM-bound on the one pass over the row for the total energy
// ((nfft/2) * sizeof(complex) bytes per valid frame); the harmonic bins are a few scattered reads of
// the same row (L2 hits).  The first harmonic is computed redundantly by every lane, then lanes take
// harmonics h = lane, lane + 64, ...
#include "pvx_wave.h"

using namespace pvxw;

namespace {

// where the previous spectrum of a frame lives
enum { PREV_ROW = 0, PREV_F64 = 1, PREV_ZERO = 2 };

template <typename T>
__device__ __forceinline__ PeakOut harm_bin(const T* cur, const T* prv, int pmode, const double* prev0,
                                            int nbin, int N2, const PeakConst& pc, T& s3_out) {
    const T re = cur[2 * nbin], im = cur[2 * nbin + 1];
    T pr = (T)0, pi = (T)0;
    if (pmode == PREV_F64) { pr = (T)prev0[2 * nbin]; pi = (T)prev0[2 * nbin + 1]; }
    else if (pmode == PREV_ROW) { pr = prv[2 * nbin]; pi = prv[2 * nbin + 1]; }
    // PV.py:481-483: famp[max(nbin-1, 1) : min(nbin+1, nfft2) + 1] ** 2 summed left to right
    const int imin = nbin - 1 > 1 ? nbin - 1 : 1;
    int imax = nbin + 1 < N2 ? nbin + 1 : N2;
    if (imax > N2 - 1) imax = N2 - 1;
    T s3 = (T)0;
    for (int j = imin; j <= imax; j++) {
        const T a = cur[2 * j], c = cur[2 * j + 1];
        if constexpr (sizeof(T) == 8) { const T m = hypot(a, c); s3 = s3 + m * m; }   // abs() then **2
        else s3 = s3 + (a * a + c * c);
    }
    s3_out = s3;
    return peak_math<T>(nbin, re, im, pr, pi, s3, pc);
}

template <typename T>
__global__ __launch_bounds__(256) void k_harmonic_rows(HarmParams p) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int64_t li = (int64_t)blockIdx.x * nw + wid;
    if (li >= p.nfr) return;
    const int64_t fr = p.fr_begin + li;
    const int K = p.K, N2 = p.N2;
    double* of = p.f + fr * K;
    double* om = p.mag + fr * K;
    double* op = p.ph + fr * K;
    for (int j = lane; j < K; j += 64) { of[j] = 0.0; om[j] = 0.0; op[j] = 0.0; }     // PV.py:504-506
    if (lane == 0 && p.t) p.t[fr] = ((double)(fr * (int64_t)p.hop) + p.nfft / 2.0) / p.sr;   // PV.py:525
    const double f0v = p.f0[fr];
    if (!(f0v > 0.0)) {                                               // PV.py:509 (NaN fails too)
        if (lane == 0) p.residual[fr] = __builtin_nan("");            // PV.py:507
        return;
    }
    const T* cur = (const T*)p.spec + (size_t)(p.ws_off + li) * p.ldo * 2;
    const int pw = p.prevrow[fr];
    int pmode = PREV_ROW;
    const T* prv = (const T*)p.carry;                                 // pw == -2
    if (pw >= 0) prv = (const T*)p.spec + (size_t)pw * p.ldo * 2;
    else if (pw == -1) pmode = p.prev0 != nullptr ? PREV_F64 : PREV_ZERO;

    // total energy np.sum(famp**2) (PV.py:490)
    double lsum = 0.0;
    for (int k = lane; k < N2; k += 64) {
        const T a = cur[2 * k], c = cur[2 * k + 1];
        if constexpr (sizeof(T) == 8) { const double m = hypot(a, c); lsum += m * m; }
        else lsum += (double)(a * a + c * c);
    }
    const double tot = wave_sum(lsum);

    PeakConst pc;
    pc.fstep = p.fstep; pc.dt = p.dt; pc.nfft = p.nfft; pc.hop = p.hop; pc.wfbin = p.wfbin;
    const double f0bin = f0v / p.sr * (double)p.nfft;                 // PV.py:462
    const double stop = (double)(N2 - 1);
    // len(np.arange(f0bin, nfft2 - 1, f0bin)) = ceil((stop - start) / step)
    double nhd = ceil((stop - f0bin) / f0bin);
    if (!(nhd > 0.0)) nhd = 0.0;
    const int nh = nhd > 2.0 * N2 + 2.0 ? 2 * N2 + 2 : (int)nhd;      // host rejects f0 below half a bin

    double cum = 0.0;
    if (nh > 0) {
        // first harmonic, by every lane (uniform): its measured frequency re-centres the others
        T s3;
        const int nbin0 = (int)rint(f0bin);                           // np.round: half to even
        const PeakOut o0 = harm_bin<T>(cur, prv, pmode, p.prev0, nbin0, N2, pc, s3);
        const double f1 = o0.nanph ? __builtin_nan("") : o0.freq;
        const bool recentre = f1 > p.fmin;                            // PV.py:466 (False for NaN)
        for (int h0 = 0; h0 < nh; h0 += 64) {
            const int h = h0 + lane;
            double e = 0.0;
            if (h < nh) {
                int nbin = (int)rint(f0bin + (double)h * f0bin);      // arange element, then np.round
                if (h > 0 && recentre) {
                    const double corrbin = f1 / p.sr * (double)p.nfft * (double)(h + 1);   // PV.py:467
                    if (corrbin < stop) nbin = (int)rint(corrbin);    // int(round(corrbin)), PV.py:468-469
                }
                T s3h;
                const PeakOut o = harm_bin<T>(cur, prv, pmode, p.prev0, nbin, N2, pc, s3h);
                e = (double)s3h;
                if (h < K) {                                          // PV.py:512-516
                    of[h] = o.nanph ? __builtin_nan("") : o.freq;
                    om[h] = o.mag;
                    op[h] = o.thisph;
                }
            }
            cum += wave_sum(e);
        }
    }
    if (lane == 0) p.residual[fr] = sqrt(tot - cum);                  // PV.py:490 (NaN when negative)
}

}  // namespace

int pvx_launch_harmonic(const HarmParams& p, int precision, hipStream_t s) {
    if (p.nfr <= 0) return PVX_OK;
    const unsigned nb = (unsigned)((p.nfr + 3) / 4);
    if (precision == 64) hipLaunchKernelGGL(k_harmonic_rows<double>, dim3(nb), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_harmonic_rows<float>, dim3(nb), dim3(256), 0, s, p);
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}
