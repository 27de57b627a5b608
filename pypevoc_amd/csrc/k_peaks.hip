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
// 16-byte-per-lane loads and only |X|^2 is kept, in LDS.  Everything else runs out of LDS/registers:
//   - min / max / energy: DPP row reductions + 4 readlanes (no LDS crossbar traffic);
//   - candidates = interior local maxima above the threshold, compacted into an LDS list with
//     ballot + mbcnt (list order = ascending bin, so no sort is ever needed);
//   - top-npeaks of the candidates: nothing to do when there are <= npeaks of them (the common
//     case), otherwise an exact radix select on the float bits driven by ballots/popcounts only;
//   - salience test: 2*rad+1 LDS reads per selected peak.
// The per-peak phase-vocoder arithmetic is batched: a wave first selects the peaks of G frames, then
// all 64 lanes work on (frame, peak) pairs at once, re-reading only the <= K selected bins of the
// current and previous rows (L2 hits).  Algorithmic bytes/frame = (nfft/2)*sizeof(complex) read +
// (5K+2)*8 written.
//
// precision=32: angles are computed in float32 (the float32 FFT already limits them to ~1e-7 rad)
// and assembled into float64 outputs around the exactly known bin centre; precision=64: the
// reference's operation order in float64 for the per-peak arithmetic.
// The peak SEARCH runs on |X|^2 at both precisions (re*re + im*im, three instructions per bin where hypot() is
// about fifty): every test it makes -- local maximum, threshold, ranking, salience -- is monotone in |X|, so it
// selects the bins np.abs() would unless two compared magnitudes agree to the last bit or two, where the
// reference's own outcome hangs on its libm's hypot() rounding (no two implementations agree there either).
#include "pvx_wave.h"

using namespace pvxw;

namespace {

// ---------------------------------------------------------------------------------------------
template <typename T> struct Vec16;   // 16-byte global access
template <> struct Vec16<float> { using type = float4; static constexpr int CPV = 2; };
template <> struct Vec16<double> { using type = double2; static constexpr int CPV = 1; };

constexpr int GMAX = 8;               // frames a wave batches before the per-peak pass

// LDS per wave (bytes), shared with the host-side sizing below
__host__ __device__ __forceinline__ size_t peaks_lds_per_wave(int N2, int K, size_t ts, bool cand = false) {
    const size_t n2pad = (size_t)((N2 + 3) & ~3);
    const size_t cap = (n2pad / 2 + 4 + 1) & ~(size_t)1;
    const size_t kpad = (size_t)((K + 3) & ~3);
    if (cand) {
        // candidate variant (the split transform left the row's candidates): cs[cap] T | ci[cap] u16 | sel | lst | cnt | tot
        size_t b = cap * ts + cap * 2 + kpad * 4 + (size_t)GMAX * kpad * 4 + GMAX * 4;
        b = (b + 7) & ~(size_t)7;
        b += GMAX * 8;
        return (b + 15) & ~(size_t)15;
    }
    // y[n2pad] T | ci[cap] u16 | sel[kpad] int | lst[GMAX][kpad] int | cnt[GMAX] int | tot[GMAX] double
    // (no score list and 16-bit bins: 10 bytes of LDS per bin at float64 instead of 14, i.e. 8 waves per CU instead of
    // 4 at nfft 4096 -- what this latency-bound kernel's speed hangs on there)
    size_t b = n2pad * ts + cap * 2 + kpad * 4 + (size_t)GMAX * kpad * 4 + GMAX * 4;
    b = (b + 7) & ~(size_t)7;
    b += GMAX * 8;
    return (b + 15) & ~(size_t)15;
}

// |X|^2 of a spectrum row in global memory, indexed like the LDS row (the candidate variant's salience test and its
// rare "fewer maxima than npks under a negative threshold" search read the few bins they need from the row itself)
template <typename T> struct RowMag {
    const T* cur;
    __device__ __forceinline__ T operator[](int k) const { const T re = cur[2 * k], im = cur[2 * k + 1]; return re * re + im * im; }
};

// CAND: the transform kernel (k_stft_split) left every row's candidates -- bins and |X|^2 of the interior local maxima
// above the threshold, max / min / sum of |X|^2 -- so phase A loads a few hundred bytes per row instead of streaming
// nfft/2 complex bins: same candidates, same selection (the energy is summed in another order: totalmag to round-off).
// (A/B: -D overrides) vector loads of a row in flight per lane; bins per piece of the candidate scan
#ifndef PVX_PEAKS_UNROLL
#define PVX_PEAKS_UNROLL 8
#endif
#ifndef PVX_PEAKS_PIECE
#define PVX_PEAKS_PIECE 512
#endif
template <typename T, bool CAND>
__global__ __launch_bounds__(256) void k_phase_peaks(PeaksParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: row bookkeeping on the scalar unit
    const int nwaves = blockDim.x >> 6;
    const int N2 = p.N2, K = p.K;
    const int n2pad = (N2 + 3) & ~3;
    const int cap = (n2pad / 2 + 4 + 1) & ~1;
    const int kpad = (K + 3) & ~3;
    unsigned char* base = smem + peaks_lds_per_wave(N2, K, sizeof(T), CAND) * wid;
    T* y = (T*)base;                                                   // CAND: the candidates' scores (cs)
    unsigned short* ci = (unsigned short*)(y + (CAND ? cap : n2pad));
    int* sel = (int*)(ci + cap);
    int* lst = sel + kpad;                       // [GMAX][kpad]
    int* cntv = lst + GMAX * kpad;               // [GMAX]
    double* totv = (double*)(((uintptr_t)(cntv + GMAX) + 7) & ~(uintptr_t)7);   // [GMAX]

    const int G = p.frames_per_wave < GMAX ? p.frames_per_wave : GMAX;
    const int64_t rel0 = ((int64_t)blockIdx.x * nwaves + wid) * G;           // first row of this wave
    if (rel0 >= p.nrows) return;
    const int ng = (p.nrows - rel0 < G) ? (int)(p.nrows - rel0) : G;

    // ================= phase A: per frame, stream |X| into LDS and select the peak bins
    for (int g = 0; g < ng; ++g) {
        const int64_t rel = rel0 + g;
        const int64_t gr = p.R0 + rel;
        const int64_t q = gr % (p.F + 1);
        if (q == 0) {                                                // zero row of a signal: no output
            if (lane == 0) cntv[g] = -1;
            continue;
        }
        const T* cur = (const T*)p.spec + (size_t)(rel + 1) * p.ldo * 2;
        wave_sync();                                                 // y / cs / ci / sel free again
        if constexpr (CAND) {
            const double* st = p.cand_stats + (size_t)(rel + 1) * 4;
            const T maxv = (T)st[0], minv = (T)st[1];
            const double tot = st[2];
            const int C = (int)st[3];
            double minamp;
            if constexpr (sizeof(T) == 4) minamp = (double)sqrtf(maxv) * p.thr;          // PF.py:60
            else minamp = sqrt((double)maxv) * p.thr;
            const double th = (minamp != 0.0) ? minamp * minamp - (double)minv : 0.0;
            const size_t rb = (size_t)(rel + 1) * p.cand_cap;
            const T* gy = (const T*)p.cand_y + rb;
            const unsigned short* gb = p.cand_bin + rb;
            T* cs = y;
            for (int c = lane; c < C; c += 64) { ci[c] = gb[c]; cs[c] = (T)(gy[c] - minv); }
            wave_sync();
            const RowMag<T> ym{cur};
            int nsel = 0;
            if (N2 >= 3) nsel = peak_pick<T, 0, true, true>(ym, cs, ci, sel, N2, K, C, th, lane, minv);
            int nk = 0;
            for (int e0 = 0; e0 < nsel; e0 += 64) {
                const int e = e0 + lane;
                int pb = 0;
                bool keep = false;
                if (e < nsel) { pb = sel[e]; keep = salient<T>(ym, N2, pb, p.rad); }
                const unsigned long long bal = __ballot(keep);
                if (keep) lst[g * kpad + nk + lane_prefix(bal)] = pb;
                nk += __popcll(bal);
            }
            if (lane == 0) { cntv[g] = nk; totv[g] = tot; }
            continue;
        }
        // ---- stream the row: |X| -> LDS, extremes and energy (PV.py:173, 210; PF.py:60, 164)
        T lmax = (T)-INFINITY, lmin = (T)INFINITY;
        double lsum = 0.0;
        {
            using V = typename Vec16<T>::type;
            constexpr int CPV = Vec16<T>::CPV;
            const int nvec = N2 / CPV;
            const V* cv = (const V*)cur;
#pragma unroll PVX_PEAKS_UNROLL
            for (int i = lane; i < nvec; i += 64) {
                // (non-temporal: the row streams through once -- +2 % at float64 / nfft 4096, +8 % where nfft 8192 runs without candidates)
                typedef T evt __attribute__((ext_vector_type(16 / sizeof(T))));
                const evt ev = __builtin_nontemporal_load((const evt*)&cv[i]);
                V v;
                __builtin_memcpy(&v, &ev, 16);
                if constexpr (CPV == 2) {
                    // float32: the peak search runs on |X|^2 -- every test it makes is monotone in |X|
                    const float e0 = v.x * v.x + v.y * v.y, e1 = v.z * v.z + v.w * v.w;
                    *(float2*)(y + 2 * i) = make_float2(e0, e1);
                    lmax = fmaxf(lmax, fmaxf(e0, e1));
                    lmin = fminf(lmin, fminf(e0, e1));
                    lsum += (double)e0 + (double)e1;
                } else {
                    const T m0 = v.x * v.x + v.y * v.y;              // float64: |X|^2 too (see the header)
                    y[i] = m0;
                    lmax = m0 > lmax ? m0 : lmax;
                    lmin = m0 < lmin ? m0 : lmin;
                    lsum += m0;
                }
            }
            for (int k = nvec * CPV + lane; k < N2; k += 64) {       // odd tail (CPV == 2, N2 odd)
                const T re = cur[2 * k], im = cur[2 * k + 1];
                const T m0 = re * re + im * im;                      // float32 only: |X|^2 like the vector body
                y[k] = m0;
                lmax = m0 > lmax ? m0 : lmax;
                lmin = m0 < lmin ? m0 : lmin;
                lsum += (double)m0;
            }
        }
        const T maxv = wave_max(lmax);
        const T minv = wave_min(lmin);
        const double tot = wave_sum(lsum);
        wave_sync();
        // ---- PeakFinder(famp, npeaks, minrattomax) + filter_by_salience(rad=5)  (PV.py:175-178)
        // the row holds |X|^2: |X| - miny > minamp - miny  <=>  |X|^2 - mine > minamp^2 - mine; minamp == 0
        // means minamp = miny (PF.py:69-70) and the threshold is then exactly 0 (see k_fused.hip)
        double minamp;
        if constexpr (sizeof(T) == 4) minamp = (double)sqrtf(maxv) * p.thr;          // PF.py:60
        else minamp = sqrt((double)maxv) * p.thr;
        const double th = (minamp != 0.0) ? minamp * minamp - (double)minv : 0.0;
        int nsel = 0;
        if (N2 >= 3) {
            // the candidate scan in pieces of 512 bins, each with all of its LDS reads in flight (the loop form is a dependent
            // round trip per 64 bins: 32 of them at nfft 4096), the rest of the row in the loop form
            int C = 0, kb = 0;
#pragma unroll 1
            for (; kb + PVX_PEAKS_PIECE <= N2; kb += PVX_PEAKS_PIECE) C += peak_scan<T, PVX_PEAKS_PIECE / 64, false>((const T*)y, kb, PVX_PEAKS_PIECE, N2, minv, th, (T*)nullptr, ci + C, lane);
            if (kb < N2) C += peak_scan<T, 0, false>((const T*)y, kb, N2 - kb, N2, minv, th, (T*)nullptr, ci + C, lane);
            wave_sync();
            nsel = peak_pick<T, 0, false, true>((const T*)y, (T*)nullptr, ci, sel, N2, K, C, th, lane, minv);
        }
        int nk = 0;
        for (int e0 = 0; e0 < nsel; e0 += 64) {
            const int e = e0 + lane;
            int pb = 0;
            bool keep = false;
            if (e < nsel) { pb = sel[e]; keep = salient<T>(y, N2, pb, p.rad); }
            const unsigned long long bal = __ballot(keep);
            if (keep) lst[g * kpad + nk + lane_prefix(bal)] = pb;
            nk += __popcll(bal);
        }
        if (lane == 0) { cntv[g] = nk; totv[g] = tot; }
    }
    wave_sync();

    // ================= phase B: per-peak phase vocoder arithmetic on (frame, peak) pairs
    // lanes_per_frame LPF = smallest power of two >= K (at most 64); 64/LPF frames per pass
    PeakConst pc;
    pc.fstep = p.fstep; pc.dt = p.dt; pc.nfft = p.nfft; pc.hop = p.hop; pc.wfbin = p.wfbin;
    int LPF = 1;
    while (LPF < K && LPF < 64) LPF <<= 1;
    const int fpp = 64 / LPF;
    const int gl = lane / LPF, e0 = lane - gl * LPF;
    const unsigned long long gmask = (LPF == 64 ? ~0ull : ((1ull << LPF) - 1ull)) << (gl * LPF);
    for (int gb = 0; gb < ng; gb += fpp) {
        const int g = gb + gl;
        const bool gvalid = g < ng;
        const int cnt = gvalid ? cntv[g] : -1;
        const int64_t rel = rel0 + (gvalid ? g : 0);
        const int64_t gr = p.R0 + rel;
        const int64_t b = gr / (p.F + 1);
        const int64_t fr = gr - b * (p.F + 1) - 1;
        const int64_t orow = b * p.F + fr;
        const T* cur = (const T*)p.spec + (size_t)(rel + 1) * p.ldo * 2;
        const T* prv = (const T*)p.spec + (size_t)rel * p.ldo * 2;
        const bool use_prev0 = (p.prev0 != nullptr) && (orow == 0);
        double* of = p.f + orow * K;
        double* om = p.mag + orow * K;
        double* op = p.ph + orow * K;
        double* orp = p.realph + orow * K;
        double* ob = p.binno + orow * K;
        int nout = 0;
        // all lanes run the same number of passes (ballots inside): wave-uniform bound
        for (int eb = 0; eb < K; eb += LPF) {
            const int e = eb + e0;
            bool valid = (cnt >= 0) && (e < cnt);
            int nbin = 0;
            double freq = 0.0, dfb = 0.0, thisph = 0.0, mg = 0.0;
            if (valid) {
                nbin = lst[g * kpad + e];
                const T re = cur[2 * nbin], im = cur[2 * nbin + 1];
                T pr, pi;
                if (use_prev0) { pr = (T)p.prev0[2 * nbin]; pi = (T)p.prev0[2 * nbin + 1]; }
                else { pr = prv[2 * nbin]; pi = prv[2 * nbin + 1]; }
                // PV.py:197-199: 3-bin energy, bin 0 excluded
                const int imin = nbin - 1 > 1 ? nbin - 1 : 1;
                int imax = nbin + 1 < N2 ? nbin + 1 : N2;
                if (imax > N2 - 1) imax = N2 - 1;
                T s3 = (T)0;
                for (int j = imin; j <= imax; j++) { const T a = cur[2 * j], c = cur[2 * j + 1]; s3 = s3 + (a * a + c * c); }
                const PeakOut o = peak_math<T>(nbin, re, im, pr, pi, s3, pc);
                freq = o.freq; dfb = o.dfb; thisph = o.thisph; mg = o.mag;
                valid = o.valid;
            }
            const unsigned long long bal = __ballot(valid) & gmask;
            if (valid) {
                const int o = nout + __popcll(bal & ((1ull << lane) - 1ull));
                ob[o] = (double)nbin;
                of[o] = freq;
                om[o] = mg;
                op[o] = thisph;
                orp[o] = thisph + kPi * dfb / p.fstep;                    // PV.py:207
            }
            nout += __popcll(bal);
        }
        if (cnt >= 0) {
            for (int j = nout + e0; j < K; j += LPF) {                    // zero padding, PV.py:226-239
                ob[j] = 0.0; of[j] = 0.0; om[j] = 0.0; op[j] = 0.0; orp[j] = 0.0;
            }
            if (e0 == 0) {
                if (p.totalmag) p.totalmag[orow] = sqrt(totv[g]);                                   // PV.py:210
                if (p.t) p.t[orow] = ((double)(fr * (int64_t)p.hop) + p.nfft / 2.0) / p.sr;         // PV.py:247
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Standalone PeakFinder on rows of float64 (the drop-in for `PeakFinder(y, ...)` +
// filter_by_salience): one wave per row.
__host__ __device__ __forceinline__ size_t rows_lds_per_wave(int n, int npk) {
    const size_t npad = (size_t)((n + 3) & ~3), cap = npad / 2 + 4, kpad = (size_t)((npk + 3) & ~3);
    return (npad * 8 + cap * 8 + cap * 4 + kpad * 4 + 15) & ~(size_t)15;
}

__global__ __launch_bounds__(256) void k_peak_rows(PeakRowsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int n = p.n;
    const int npk = (p.npeaks <= 0 || p.npeaks > n) ? n : p.npeaks;    // PF.py:64-67 (at most n-2 exist)
    const int npad = (n + 3) & ~3;
    const int cap = npad / 2 + 4;
    unsigned char* base = smem + rows_lds_per_wave(n, npk) * wid;
    double* y = (double*)base;
    double* cs = y + npad;
    int* ci = (int*)(cs + cap);
    int* sel = ci + cap;
    const int64_t row = (int64_t)blockIdx.x * nwaves + wid;
    if (row >= p.nrows) return;
    const double* src = p.y + row * (int64_t)n;
    double lmax = -(double)INFINITY, lmin = (double)INFINITY;
    for (int k = lane; k < n; k += 64) {
        const double v = src[k];
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
    const int nsel = peak_select<double>(y, cs, ci, sel, n, npk, minamp, have, miny, lane);
    int32_t* pos = p.pos + row * (int64_t)p.cap;
    int8_t* keep = p.keep + row * (int64_t)p.cap;
    for (int e = lane; e < nsel && e < p.cap; e += 64) {
        const int q = sel[e];
        pos[e] = q;
        keep[e] = salient<double>(y, n, q, p.rad) ? 1 : 0;
    }
    if (lane == 0) p.count[row] = nsel;
}

}  // namespace

size_t pvx_phase_peaks_lds_bytes(int N2, int K, int precision, int waves) {
    return peaks_lds_per_wave(N2, K, precision == 32 ? 4 : 8) * (size_t)waves;
}
static size_t peaks_lds_bytes_cand(int N2, int K, int precision, int waves) {
    return peaks_lds_per_wave(N2, K, precision == 32 ? 4 : 8, true) * (size_t)waves;
}

static constexpr size_t kMaxLds = 160 * 1024;

int pvx_launch_phase_peaks(const PeaksParams& pin, int precision, hipStream_t s) {
    if (pin.nrows <= 0) return PVX_OK;
    PeaksParams p = pin;
    if (p.frames_per_wave < 1) p.frames_per_wave = 1;
    if (p.frames_per_wave > GMAX) p.frames_per_wave = GMAX;
    const bool cand = p.cand_bin != nullptr && p.cand_y != nullptr && p.cand_stats != nullptr;
    int waves = 4;
    auto ldsb = [&](int w) { return cand ? peaks_lds_bytes_cand(p.N2, p.K, precision, w) : pvx_phase_peaks_lds_bytes(p.N2, p.K, precision, w); };
    while (waves > 1 && ldsb(waves) > kMaxLds / 2) waves >>= 1;
    const size_t lds = ldsb(waves);
    if (lds > kMaxLds) {
        pvx_set_error("nfft=%d with npks=%d needs %zu bytes of LDS per wave (limit %zu)", p.nfft, p.K, lds, kMaxLds);
        return PVX_ERR_UNSUPPORTED;
    }
    const int64_t per_block = (int64_t)waves * p.frames_per_wave;
    const int64_t nblocks = (p.nrows + per_block - 1) / per_block;
    if (nblocks > 0x7fffffffLL) { pvx_set_error("too many rows in one launch"); return PVX_ERR_INVALID; }
    dim3 grid((unsigned)nblocks), block(64 * waves);
    const void* fn = precision == 32 ? (cand ? (const void*)k_phase_peaks<float, true> : (const void*)k_phase_peaks<float, false>)
                                     : (cand ? (const void*)k_phase_peaks<double, true> : (const void*)k_phase_peaks<double, false>);
    if (lds > 64 * 1024) PVX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* args[] = {&p};
    PVX_HIP_CHECK(hipLaunchKernel(fn, grid, block, args, lds, s));
    return PVX_OK;
}

int pvx_launch_peak_rows(const PeakRowsParams& p, hipStream_t s) {
    if (p.nrows <= 0) return PVX_OK;
    const int n = p.n;
    const int npk = (p.npeaks <= 0 || p.npeaks > n) ? n : p.npeaks;
    int waves = 4;
    while (waves > 1 && rows_lds_per_wave(n, npk) * waves > kMaxLds / 2) waves >>= 1;
    const size_t lds = rows_lds_per_wave(n, npk) * waves;
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
