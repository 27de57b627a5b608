// k_peaks.hip -- spectrum -> sinusoidal peaks.  Replaces, for every frame of a launch:
//   PV.calc_pv_frame        pypevoc/PVAnalysis.py:160-211  (abs, peak loop, 3-bin energy, realph)
//   PV.dphase2freq          pypevoc/PVAnalysis.py:133-148  (phase difference -> frequency)
//   PeakFinder.__init__     pypevoc/PeakFinder.py:35-74    (thresholds)
//   PeakFinder.findpos      pypevoc/PeakFinder.py:155-194  (top-npeaks interior local maxima)
//   PeakFinder.filter_by_salience  pypevoc/PeakFinder.py:113-136
//   (PeakFinder.boundaries, PeakFinder.py:269-302/475-484, is called by the reference but its result
//    is discarded -- PV.py:176 -- so it has no counterpart here)
//
// Roofline: HBM.  One wave64 per frame: the half spectrum (nfft/2 complex) is streamed once with
// 16-byte-per-lane loads, magnitudes go to LDS, everything else (min/max/energy wave reductions,
// local-maximum scores, top-K by repeated wave arg-max, salience test) runs out of LDS/registers.
// Only the <= K selected bins are re-read (current and previous row; L2 hits, the wave that owns
// frame fr-1 is the same wave or its neighbour).  Algorithmic bytes/frame = (nfft/2)*sizeof(complex)
// read + (5K+2)*8 written.
//
// The per-peak arithmetic (atan2, frequency candidates, energy, realph) is float64 in both
// precisions: it is <= K values per frame and it is what the reference's float64 outputs hold.
#include <float.h>
#include <math.h>

#include "pvx_internal.h"

namespace {

constexpr double kPi = 3.141592653589793238462643383279502884;
constexpr double kPi2 = 2.0 * kPi;   // PV.py:45

__device__ inline void wave_sync() {
    // LDS hand-off between lanes of ONE wave: order the accesses, no cross-wave barrier.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

template <typename T> __device__ inline T neg_inf();
template <> __device__ inline float neg_inf<float>() { return -INFINITY; }
template <> __device__ inline double neg_inf<double>() { return -(double)INFINITY; }

template <typename T> __device__ inline T wave_max(T v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { T u = __shfl_xor(v, o); v = u > v ? u : v; }
    return v;
}
template <typename T> __device__ inline T wave_min(T v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { T u = __shfl_xor(v, o); v = u < v ? u : v; }
    return v;
}
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// arg-max with "first index wins" ties (np.argmax, PF.py:173/183)
template <typename T> __device__ inline void wave_argmax(T& v, int& i) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        T u = __shfl_xor(v, o);
        int j = __shfl_xor(i, o);
        if (u > v || (u == v && j < i)) { v = u; i = j; }
    }
}

// ---------------------------------------------------------------------------------------------
// PeakFinder core on one row held in LDS.
//   y[n]      : the row (T)
//   score[n]  : scratch (T)
//   sel[cap], srt[cap] : int scratch, cap >= min(npeaks, n)
// Returns (wave-uniform) the number of positions; srt[0..count) = positions ascending (PF.py:189).
// miny/maxy must be the wave-reduced extremes of y.
template <typename T>
__device__ inline int peak_select(const T* y, T* score, int* sel, int* srt, int n, int npeaks,
                                  double minamp_in, bool have_minamp, T miny, int lane) {
    if (n < 3) return 0;
    // PF.py:69-70: "if not self.minamp: self.minamp = np.min(self.y)"
    double minamp = (have_minamp && minamp_in != 0.0) ? minamp_in : (double)miny;
    const double th = minamp - (double)miny;                       // PF.py:174
    const T NEG = neg_inf<T>();
    // PF.py:166-167: interior local maxima, score = y - miny, 0 elsewhere
    T bv = NEG;
    int bi = 0x7fffffff;
    for (int k = lane; k < n; k += 64) {
        T s = NEG;
        if (k >= 1 && k <= n - 2) {
            T a = y[k - 1], b = y[k], c = y[k + 1];
            s = (a < b && b >= c) ? (T)(b - miny) : (T)0;
        }
        score[k] = s;
        if (s > bv) { bv = s; bi = k; }
    }
    int nsel = 0;
    while (nsel < npeaks) {                                        // PF.py:177-187
        T v = bv;
        int i = bi;
        wave_argmax(v, i);
        if (!((double)v > th)) break;
        if (lane == 0) sel[nsel] = i;
        nsel++;
        if ((i & 63) == lane) {                                    // owner retires it and rescans its bins
            score[i] = NEG;                                        // pkmskamp[b] = th - 1
            bv = NEG;
            bi = 0x7fffffff;
            for (int k = lane; k < n; k += 64) {
                T s = score[k];
                if (s > bv) { bv = s; bi = k; }
            }
        }
    }
    wave_sync();
    // np.sort(pos): rank by counting (positions are distinct)
    for (int e = lane; e < nsel; e += 64) {
        int mine = sel[e], r = 0;
        for (int j = 0; j < nsel; j++) r += (sel[j] < mine) ? 1 : 0;
        srt[r] = mine;
    }
    wave_sync();
    return nsel;
}

// filter_by_salience(rad), sal = 0 (PF.py:126-134): keep unless any y in
// [max(p-rad,1), min(p+rad,n)] (clipped to the array) exceeds y[p]
template <typename T> __device__ inline bool salient(const T* y, int n, int p, int rad) {
    if (rad < 0) return true;
    T v = y[p];
    int lo = p - rad > 1 ? p - rad : 1;
    int hi = p + rad < n ? p + rad : n;
    if (hi > n - 1) hi = n - 1;
    bool keep = true;
    for (int j = lo; j <= hi; j++) keep = keep && !(y[j] > v);
    return keep;
}

// ---------------------------------------------------------------------------------------------
template <typename T> struct Vec16;   // 16-byte global access
template <> struct Vec16<float> { using type = float4; static constexpr int CPV = 2; };
template <> struct Vec16<double> { using type = double2; static constexpr int CPV = 1; };

template <typename T>
__global__ __launch_bounds__(256) void k_phase_peaks(PeaksParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int N2 = p.N2, K = p.K;
    const int n2pad = (N2 + 3) & ~3;
    const int kpad = (K + 3) & ~3;
    // per-wave LDS carve: y[n2pad] T | score[n2pad] T | sel[kpad] int | srt[kpad] int
    const size_t per_wave = (size_t)n2pad * sizeof(T) * 2 + (size_t)kpad * sizeof(int) * 2;
    unsigned char* base = smem + per_wave * wid;
    T* y = (T*)base;
    T* score = y + n2pad;
    int* sel = (int*)(score + n2pad);
    int* srt = sel + kpad;

    const int64_t wave_global = (int64_t)blockIdx.x * nwaves + wid;
    for (int it = 0; it < p.frames_per_wave; ++it) {
        const int64_t rel = wave_global * p.frames_per_wave + it;   // row within the launch
        if (rel >= p.nrows) break;                                  // wave-uniform
        const int64_t g = p.R0 + rel;
        const int64_t b = g / (p.F + 1);
        const int64_t q = g - b * (p.F + 1);
        if (q == 0) continue;                                       // zero row of a signal
        const int64_t fr = q - 1;
        const int64_t orow = b * p.F + fr;                          // output row
        const T* cur = (const T*)p.spec + (size_t)(rel + 1) * p.ldo * 2;
        const T* prv = (const T*)p.spec + (size_t)rel * p.ldo * 2;
        const bool use_prev0 = (p.prev0 != nullptr) && (orow == 0);

        wave_sync();   // previous iteration's readers are done with y/score/sel/srt
        // ---- pass 1: stream the row, |X| -> LDS, extremes and energy (PV.py:173, 210; PF.py:60,164)
        T lmax = neg_inf<T>(), lmin = -neg_inf<T>();
        double lsum = 0.0;
        {
            using V = typename Vec16<T>::type;
            constexpr int CPV = Vec16<T>::CPV;
            const int nvec = N2 / CPV;
            const V* cv = (const V*)cur;
#pragma unroll 4
            for (int i = lane; i < nvec; i += 64) {
                V v = cv[i];
                if constexpr (CPV == 2) {
                    T m0 = sqrtf(v.x * v.x + v.y * v.y);
                    T m1 = sqrtf(v.z * v.z + v.w * v.w);
                    *(float2*)(y + 2 * i) = make_float2(m0, m1);
                    lmax = fmaxf(lmax, fmaxf(m0, m1));
                    lmin = fminf(lmin, fminf(m0, m1));
                    lsum += (double)m0 * (double)m0 + (double)m1 * (double)m1;
                } else {
                    T m0 = hypot(v.x, v.y);                         // np.abs of complex128
                    y[i] = m0;
                    lmax = m0 > lmax ? m0 : lmax;
                    lmin = m0 < lmin ? m0 : lmin;
                    lsum += m0 * m0;
                }
            }
            for (int k = nvec * CPV + lane; k < N2; k += 64) {      // odd tail (CPV == 2, N2 odd)
                T re = cur[2 * k], im = cur[2 * k + 1];
                T m0 = (T)sqrt((double)re * re + (double)im * im);
                y[k] = m0;
                lmax = m0 > lmax ? m0 : lmax;
                lmin = m0 < lmin ? m0 : lmin;
                lsum += (double)m0 * (double)m0;
            }
        }
        const T maxy = wave_max(lmax);
        const T miny = wave_min(lmin);
        const double tot = wave_sum(lsum);
        wave_sync();

        // ---- peak picking (PV.py:175-178)
        const double minamp = (double)maxy * p.thr;                 // PF.py:60
        const int nsel = peak_select<T>(y, score, sel, srt, N2, K, minamp, true, miny, lane);

        // ---- per-peak phase vocoder arithmetic, ascending bin order (PV.py:187-207)
        double* of = p.f + orow * K;
        double* om = p.mag + orow * K;
        double* op = p.ph + orow * K;
        double* orp = p.realph + orow * K;
        double* ob = p.binno + orow * K;
        int nout = 0;
        for (int basei = 0; basei < nsel; basei += 64) {
            const int e = basei + lane;
            bool valid = e < nsel;
            int nbin = 0;
            double freq = 0.0, dfb = 0.0, thisph = 0.0, mg = 0.0;
            if (valid) {
                nbin = srt[e];
                valid = salient<T>(y, N2, nbin, p.rad);
            }
            if (valid) {
                const double re = (double)cur[2 * nbin], im = (double)cur[2 * nbin + 1];
                double pr, pi;
                if (use_prev0) { pr = p.prev0[2 * nbin]; pi = p.prev0[2 * nbin + 1]; }
                else { pr = (double)prv[2 * nbin]; pi = (double)prv[2 * nbin + 1]; }
                thisph = atan2(im, re);                              // PV.py:188
                double dph;
                if (pr == 0.0 && pi == 0.0) {
                    // numpy: (a+bj)/(0+0j) = (a/0) + (b/0)j -> +-inf +-inf j, NaN when a or b is 0;
                    // angle() then is +-pi/4, +-3pi/4 by quadrant (PV.py:171, 190; frame 0 and any
                    // frame that follows an all-zero one)
                    if (re == 0.0 || im == 0.0 || re != re || im != im) dph = NAN;
                    else dph = (re > 0.0) ? (im > 0.0 ? kPi / 4 : -kPi / 4) : (im > 0.0 ? 3 * kPi / 4 : -3 * kPi / 4);
                } else {
                    // angle(fx/old) = angle(fx * conj(old))
                    dph = atan2(im * pr - re * pi, re * pr + im * pi);
                }
                if (dph != dph) {
                    valid = false;                                   // NaN: `freq > 0` is False (PV.py:193)
                } else {
                    // PV.py:140-147: three unwrapping candidates, the one nearest the bin centre
                    const double fb = (double)nbin * p.fstep;        // PV.py:114
                    const double w0 = dph + p.wfbin[nbin];
                    double bestabs = 0.0;
#pragma unroll
                    for (int m = -1; m <= 1; m++) {
                        double dphw = w0 + kPi2 * (double)m;
                        double fq = dphw / p.dt / kPi2;
                        double df = fb - fq;
                        double a = fabs(df);
                        if (m == -1 || a < bestabs) { freq = fq; dfb = df; bestabs = a; }
                    }
                    valid = freq > 0.0;                              // PV.py:193
                }
                if (valid) {
                    // PV.py:197-199: 3-bin energy, bin 0 excluded
                    const int imin = nbin - 1 > 1 ? nbin - 1 : 1;
                    int imax = nbin + 1 < N2 ? nbin + 1 : N2;
                    if (imax > N2 - 1) imax = N2 - 1;
                    double s = 0.0;
                    for (int j = imin; j <= imax; j++) {
                        double a = (double)cur[2 * j], c = (double)cur[2 * j + 1];
                        s = s + (a * a + c * c);
                    }
                    mg = sqrt(s);
                }
            }
            const unsigned long long bal = __ballot(valid);
            if (valid) {
                const int o = nout + __popcll(bal & ((1ull << lane) - 1ull));
                ob[o] = (double)nbin;
                of[o] = freq;
                om[o] = mg;
                op[o] = thisph;
                orp[o] = thisph + kPi * dfb / p.fstep;               // PV.py:207
            }
            nout += __popcll(bal);
        }
        for (int j = nout + lane; j < K; j += 64) {                  // zero padding, PV.py:226-239
            ob[j] = 0.0; of[j] = 0.0; om[j] = 0.0; op[j] = 0.0; orp[j] = 0.0;
        }
        if (lane == 0) {
            if (p.totalmag) p.totalmag[orow] = sqrt(tot);            // PV.py:210
            if (p.t) p.t[orow] = ((double)(fr * (int64_t)p.hop) + p.nfft / 2.0) / p.sr;   // PV.py:247
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Standalone PeakFinder on rows of float64 (the drop-in for `PeakFinder(y, ...)` +
// filter_by_salience): one wave per row.
__global__ __launch_bounds__(256) void k_peak_rows(PeakRowsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int n = p.n;
    const int npk = (p.npeaks <= 0 || p.npeaks > n) ? n : p.npeaks;    // PF.py:64-67 (at most n-2 exist)
    const int npad = (n + 3) & ~3;
    const int kpad = (npk + 3) & ~3;
    const size_t per_wave = (size_t)npad * sizeof(double) * 2 + (size_t)kpad * sizeof(int) * 2;
    unsigned char* base = smem + per_wave * wid;
    double* y = (double*)base;
    double* score = y + npad;
    int* sel = (int*)(score + npad);
    int* srt = sel + kpad;
    const int64_t row = (int64_t)blockIdx.x * nwaves + wid;
    if (row >= p.nrows) return;
    const double* src = p.y + row * (int64_t)n;
    double lmax = -(double)INFINITY, lmin = (double)INFINITY;
    for (int k = lane; k < n; k += 64) {
        double v = src[k];
        y[k] = v;
        lmax = v > lmax ? v : lmax;
        lmin = v < lmin ? v : lmin;
    }
    const double maxy = wave_max(lmax), miny = wave_min(lmin);
    wave_sync();
    double minamp = 0.0;
    bool have = false;
    if (p.thr_kind == 1) { minamp = maxy * p.thr_val; have = true; }   // PF.py:60
    else if (p.thr_kind == 2) { minamp = p.thr_val; have = true; }     // PF.py:58
    const int nsel = peak_select<double>(y, score, sel, srt, n, npk, minamp, have, miny, lane);
    int32_t* pos = p.pos + row * (int64_t)p.cap;
    int8_t* keep = p.keep + row * (int64_t)p.cap;
    for (int e = lane; e < nsel && e < p.cap; e += 64) {
        int q = srt[e];
        pos[e] = q;
        keep[e] = salient<double>(y, n, q, p.rad) ? 1 : 0;
    }
    if (lane == 0) p.count[row] = nsel;
}

}  // namespace

size_t pvx_phase_peaks_lds_bytes(int N2, int K, int precision, int waves) {
    const size_t ts = precision == 32 ? 4 : 8;
    const size_t n2pad = (size_t)((N2 + 3) & ~3), kpad = (size_t)((K + 3) & ~3);
    return (n2pad * ts * 2 + kpad * sizeof(int) * 2) * (size_t)waves;
}

static constexpr size_t kMaxLds = 160 * 1024;

int pvx_launch_phase_peaks(const PeaksParams& p, int precision, hipStream_t s) {
    if (p.nrows <= 0) return PVX_OK;
    int waves = 4;
    while (waves > 1 && pvx_phase_peaks_lds_bytes(p.N2, p.K, precision, waves) > kMaxLds / 2) waves >>= 1;
    const size_t lds = pvx_phase_peaks_lds_bytes(p.N2, p.K, precision, waves);
    if (lds > kMaxLds) {
        pvx_set_error("nfft=%d with npks=%d needs %zu bytes of LDS per wave (limit %zu)", p.nfft, p.K, lds, kMaxLds);
        return PVX_ERR_UNSUPPORTED;
    }
    const int64_t per_block = (int64_t)waves * p.frames_per_wave;
    const int64_t nblocks = (p.nrows + per_block - 1) / per_block;
    if (nblocks > 0x7fffffffLL) { pvx_set_error("too many rows in one launch"); return PVX_ERR_INVALID; }
    dim3 grid((unsigned)nblocks), block(64 * waves);
    if (precision == 32) {
        if (lds > 64 * 1024)
            PVX_HIP_CHECK(hipFuncSetAttribute((const void*)k_phase_peaks<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_phase_peaks<float>, grid, block, lds, s, p);
    } else {
        if (lds > 64 * 1024)
            PVX_HIP_CHECK(hipFuncSetAttribute((const void*)k_phase_peaks<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_phase_peaks<double>, grid, block, lds, s, p);
    }
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}

int pvx_launch_peak_rows(const PeakRowsParams& p, hipStream_t s) {
    if (p.nrows <= 0) return PVX_OK;
    const int n = p.n;
    const int npk = (p.npeaks <= 0 || p.npeaks > n) ? n : p.npeaks;
    int waves = 4;
    auto bytes = [&](int w) { return pvx_phase_peaks_lds_bytes(n, npk, 64, w); };
    while (waves > 1 && bytes(waves) > kMaxLds / 2) waves >>= 1;
    const size_t lds = bytes(waves);
    if (lds > kMaxLds) {
        pvx_set_error("PeakFinder row of %d samples does not fit in LDS (%zu bytes)", n, lds);
        return PVX_ERR_UNSUPPORTED;
    }
    const int64_t nblocks = (p.nrows + waves - 1) / waves;
    if (nblocks > 0x7fffffffLL) { pvx_set_error("too many rows"); return PVX_ERR_INVALID; }
    if (lds > 64 * 1024)
        PVX_HIP_CHECK(hipFuncSetAttribute((const void*)k_peak_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_peak_rows, dim3((unsigned)nblocks), dim3(64 * waves), lds, s, p);
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}
