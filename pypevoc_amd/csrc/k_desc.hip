// k_desc.hip -- small kernels around the analysis result arrays:
//   k_f0             PV.calc_f0 (pypevoc/PVAnalysis.py:371-391): lowest-frequency strong peak of every frame
//   k_hpower_*       PV.calc_harmonic_power (PVAnalysis.py:266-297), including its row-indexing quirk (:278)
//   k_fill_t         frame times t = (pos + nfft/2) / sr (PVAnalysis.py:247) for chunked host input
//   k_spec_to_prev   last spectrum of a chunk -> the float64 [nfft/2][2] "oldfft" of the next one (PVAnalysis.py:209)
// Run on the (F, K) arrays where they already are -- in HBM -- so that a caller who wants the fundamental
// track does not pull five (F, K) arrays over PCIe first.
#include <math.h>

#include "pvx_internal.h"

namespace {

// one thread per frame: K is small and the rows are contiguous
__global__ __launch_bounds__(256) void k_f0(const double* __restrict__ f, const double* __restrict__ mag, int64_t F, int K,
                                            double fmin, double fmax, double thr, double* __restrict__ fm, int32_t* __restrict__ im) {
    const int64_t fr = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (fr >= F) return;
    const double* ff = f + fr * K;
    const double* mm = mag + fr * K;
    double maxmag = mm[0];                                            // np.max (PVAnalysis.py:381)
    for (int k = 1; k < K; k++) maxmag = (mm[k] > maxmag || maxmag != maxmag) ? mm[k] : maxmag;
    const double lim = maxmag * thr;
    double best = 0.0;
    int bi = 0;
    bool has = false;
    for (int k = 0; k < K; k++) {                                      // np.argmin of ff[in0]: first minimum (:386)
        if (ff[k] > fmin && ff[k] < fmax && mm[k] > lim && (!has || ff[k] < best)) { best = ff[k]; bi = k; has = true; }
    }
    fm[fr] = has ? best : 0.0;
    im[fr] = has ? bi : 0;
}

// rowpow[k] = sum(mag[k, :]**2) for k < min(K, F): the reference indexes the ROWS of mag with the valid peak
// slots (valid_mag = self.mag[valid_idx], PVAnalysis.py:278); top = highest slot that is valid in any frame
__global__ __launch_bounds__(64) void k_hpower_rows(const double* __restrict__ f, const double* __restrict__ mag, int64_t F, int K,
                                                    double* __restrict__ rowpow, int32_t* __restrict__ top) {
    const int k = blockIdx.x;                                          // slot / row index
    const int lane = threadIdx.x;
    if (k < F) {
        double s = 0.0;
        for (int c = lane; c < K; c += 64) { const double v = mag[(int64_t)k * K + c]; s += v * v; }
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) rowpow[k] = s;
    } else if (lane == 0) rowpow[k] = 0.0;
    bool any = false;
    for (int64_t fr = lane; fr < F; fr += 64) any = any || (f[fr * K + k] > 0.0);
    if (__ballot(any) != 0ull && lane == 0) atomicMax(top, k);
}

// one wave per frame: lane j owns peak j (K <= 64 per pass), loops over the candidates c
__global__ __launch_bounds__(256) void k_hpower(const double* __restrict__ f, int64_t F, int K, double f_threshold,
                                                const double* __restrict__ rowpow, double* __restrict__ hpower,
                                                double* __restrict__ nharm) {
    const int lane = threadIdx.x & 63;
    const int64_t fr = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (fr >= F) return;
    const double* ff = f + fr * K;
    for (int j = lane; j < K; j += 64) {
        const double fj = ff[j];
        double hp = 0.0, nh = 0.0;
        if (fj > 0.0) {
            for (int c = 0; c < K; c++) {
                const double fc = ff[c];
                if (!(fc > 0.0)) continue;
                const double ratio = fc / fj;
                double hn = nearbyint(ratio);                          // np.round: half to even (:283)
                if (hn == 0.0) hn = 1.0;
                const double inh = fabs(fc / hn / fj - 1.0);            // :285
                if (inh < f_threshold) { hp += rowpow[c]; nh += 1.0; }
            }
        }
        hpower[fr * K + j] = hp;
        nharm[fr * K + j] = nh;
    }
}

__global__ __launch_bounds__(256) void k_fill_t(double* __restrict__ t, int64_t F, int64_t nsig, int hop, int nfft, double sr) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= F * nsig) return;
    const int64_t fr = i % F;
    t[i] = ((double)(fr * (int64_t)hop) + nfft / 2.0) / sr;            // PVAnalysis.py:247
}

template <typename T> __global__ __launch_bounds__(256) void k_spec_to_prev(double* __restrict__ dst, const T* __restrict__ src, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}

}  // namespace

int pvx_launch_f0(const double* f, const double* mag, int64_t F, int K, double fmin, double fmax, double thr, double* fm,
                  int32_t* im, hipStream_t s) {
    if (F <= 0) return PVX_OK;
    hipLaunchKernelGGL(k_f0, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, s, f, mag, F, K, fmin, fmax, thr, fm, im);
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}

int pvx_launch_hpower_rows(const double* f, const double* mag, int64_t F, int K, double* rowpow, int32_t* top, hipStream_t s) {
    if (F <= 0) return PVX_OK;
    PVX_HIP_CHECK(hipMemsetAsync(top, 0xff, 4, s));                    // -1
    hipLaunchKernelGGL(k_hpower_rows, dim3((unsigned)K), dim3(64), 0, s, f, mag, F, K, rowpow, top);
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}

int pvx_launch_hpower(const double* f, int64_t F, int K, double f_threshold, const double* rowpow, double* hpower,
                      double* nharm, hipStream_t s) {
    if (F <= 0) return PVX_OK;
    hipLaunchKernelGGL(k_hpower, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, s, f, F, K, f_threshold, rowpow, hpower, nharm);
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}

int pvx_launch_fill_t(double* t, int64_t F, int64_t nsig, int hop, int nfft, double sr, hipStream_t s) {
    if (F <= 0 || !t) return PVX_OK;
    hipLaunchKernelGGL(k_fill_t, dim3((unsigned)((F * nsig + 255) / 256)), dim3(256), 0, s, t, F, nsig, hop, nfft, sr);
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}

int pvx_launch_spec_to_prev(double* dst, const void* src, int n, int src_is_float, hipStream_t s) {
    if (n <= 0) return PVX_OK;
    if (src_is_float) hipLaunchKernelGGL((k_spec_to_prev<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, (const float*)src, n);
    else hipLaunchKernelGGL((k_spec_to_prev<double>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, (const double*)src, n);
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}
