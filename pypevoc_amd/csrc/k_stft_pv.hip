// k_stft_pv.hip -- the general path's analysis stage in ONE kernel: k_stft's windowed FFT (k_stft.hip) and
// k_phase_peaks' peak search + phase-vocoder arithmetic (k_peaks.hip) on the same wave, at the reference's own
// precision (float64) or in plain float32:
//
//   PV.calc_fft_frame   pypevoc/PVAnalysis.py:150-158
//   PV.calc_pv_frame    pypevoc/PVAnalysis.py:160-211   (abs, PeakFinder + filter_by_salience, dphase2freq, 3-bin energy)
//   PV.run_pv           pypevoc/PVAnalysis.py:213-264   (frame loop, zero-padded packing)
//
// Why: as two kernels the half spectra (nfft/2 complex per frame: 847 MB at BASELINE config 2 in float64) are written
// by one kernel and streamed back by the next, and that second kernel is latency-bound (one wave per row: load, reduce,
// search, in turn).  Here the magnitudes a frame's peak search needs are in the wave's registers when the untangle
// produces the bins: they go to LDS, INTO the buffer the transform has just finished with (|X|^2 row + candidate lists
// fit where the exchange matrix was), so the kernel needs no more LDS per wave than k_stft -- 7 waves per CU at nfft
// 2048 -- and never reads a spectrum row back.  The spectrum rows are still written to the workspace
// (PVHarmonic, calc_fft_frame and the chunk carry read them there); the per-peak arithmetic needs the current and
// the PREVIOUS frame's spectrum at the <= K selected bins only, and fetches those few values from the workspace
// rows this same wave wrote (a wave owns a contiguous range of rows and computes the row before its first one
// itself: no wave ever reads what another wave wrote).  Results are bit-identical to k_stft + k_phase_peaks.
//
// Order of a frame: window * samples -> radix-R / LDS exchange / radix-R / P-lane DFT -> natural order in LDS ->
// untangle (bins to the workspace row, |X|^2 of both bins of a pair) -> squared magnitudes to LDS ->
// max / min / energy -> PeakFinder (pvx_wave.h: peak_select) -> salience -> selected bins staged.  Every GF staged
// frames: all 64 lanes do the per-peak arithmetic on (frame, peak) pairs (peak_math) and write the result rows.
#include <stdlib.h>

#include <type_traits>

#include "pvx_stft4.h"

using namespace pvxw;
using namespace pvxf;
using namespace pvxs;

namespace {

constexpr int GF = 8;                 // frames staged before the per-peak pass

// issue priorities of the row loop's phases (k_fused_rev.hip; -D overrides for A/B builds): transform, search, selection + staging
#ifndef PVX_PV_PRIO_T
#define PVX_PV_PRIO_T 2
#endif
#ifndef PVX_PV_PRIO_S
#define PVX_PV_PRIO_S 1
#endif
#ifndef PVX_PV_PRIO_C
#define PVX_PV_PRIO_C 0
#endif
// (nfft 512 at float64 loses 5 % with the phases ranked -- 448 -> 427 M frames/s: its transform is a fifth of the 2048 one's and
// the waves' phases interleave by themselves --, nfft 1024 / 2048 gain 5 / 7 %: ranked from R = 8 on)
#ifndef PVX_PV_PRIO_MINR
#define PVX_PV_PRIO_MINR 8
#endif
struct StftPvParams {
    StftParams s;                     // rows, input, tables, workspace
    PeaksParams p;                    // peak parameters and result arrays (p.spec / p.ldo = s.spec / s.ldo)
    int win_symmetric = 0;            // host knowledge: the window reads the same backwards
};

// per-wave LDS behind the transform buffer: lst[GF][kpad] int | cnt[GF] int | rel[GF] int64 | tot[GF] double
__host__ __device__ inline size_t pv_stage_bytes(int K) {
    const size_t kpad = (size_t)((K + 3) & ~3);
    size_t b = (size_t)GF * kpad * 4 + GF * 4;
    b = (b + 7) & ~(size_t)7;
    b += GF * 8 + GF * 8;
    return (b + 15) & ~(size_t)15;
}
// what the peak search keeps in the transform buffer: y[M] T | cs[cap] T | ci[cap] int | sel[kpad] int
template <int R, typename T> __host__ __device__ inline size_t pv_search_bytes(int K) {
    constexpr size_t M = StftGeo<R, T>::M, cap = M / 2 + 4;
    return M * sizeof(T) + cap * sizeof(T) + cap * 4 + (size_t)((K + 3) & ~3) * 4;
}
// the untangle's twiddles W_nfft^k, k < nfft/4, from the first octant of the table (k <= nfft/8) and its symmetry
// W^(nfft/4 - k) = (-Im, -Re) W^k (exact in the host's table, pvx_plan_create): 4 KB of LDS instead of 8 at nfft 2048,
// which is what lets a seventh wave's buffer in
// R = 16 (nfft 2048): the four-quarter transform of pvx_stft4.h -- t1 is [16][16] W_256^(l q), there is no t2, and the
// untangle's table becomes the lane-ordered join / untangle twiddles [2][4][64]: 28 KB of tables instead of 37 at float64
// SYM (nfft 2048 at float64, a window with w[n] = w[N-1-n] -- np.hanning and its kind, checked on the host): only the first
// half of the window is kept, the pairs of the upper half are the mirrored pairs read backwards.  8 KB less, which is what lets
// an EIGHTH wave's buffer in: two waves on every SIMD instead of 2 + 2 + 2 + 1 (the kernel's speed follows its waves:
// 4 / 5 / 6 / 7 waves per CU run 113 / 119 / 135 / 148 M frames/s on BASELINE config 2).
template <int R, typename T, bool SYM = false> struct PvGeo : StftGeo<R, T> {
    static constexpr bool X4 = (R == 16);
    static_assert(!SYM || X4, "the half window exists for the four-quarter layout");
    static constexpr int TW8N = X4 ? 512 : ((StftGeo<R, T>::HALF / 2 + 1 + 7) & ~7);
    static constexpr size_t OFF_T1 = SYM ? (size_t)(StftGeo<R, T>::N / 2) * sizeof(T) : StftGeo<R, T>::OFF_T1;
    static constexpr size_t OFF_T2 = X4 ? OFF_T1 + (size_t)256 * 2 * sizeof(T) : StftGeo<R, T>::OFF_T2;
    static constexpr size_t OFF_TW3 = X4 ? OFF_T2 : StftGeo<R, T>::OFF_TW3;
    static constexpr size_t OFF_BUF = OFF_TW3 + (size_t)TW8N * 2 * sizeof(T);                          // cx [NW][BUFC]
};
template <int R, typename T, bool SYM = false> __host__ __device__ inline size_t pv_total_lds(int nw, int K) {
    return PvGeo<R, T, SYM>::OFF_BUF + (size_t)nw * (StftGeo<R, T>::BUFC * 2 * sizeof(T) + pv_stage_bytes(K));
}

// a value this wave stored earlier in the kernel: read it where the store went (L2), not from a vector-L1 line that
// may predate it
__device__ __forceinline__ double ldw(const double* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ldw(const float* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// H > 0: hop = 128 H (nfft/4 or nfft/2).  A lane's samples of the next frame are then the ones it holds, moved up by H
// register pairs (sample 2l + 128 r + hop = 2l + 128 (r + H)): only the last H pairs are loaded -- hop*4 bytes per
// frame from HBM instead of nfft*4 (a wave walks its rows alone: by the time it comes back for the next frame the
// spectrum rows streaming through L2 have evicted the samples it shared with this one).
template <int R, typename T, typename InT, int H, bool SYM = false, bool DENSE = false>
__global__ __launch_bounds__(SYM ? 512 : 448) void k_stft_pv(StftPvParams a) {
    using G = PvGeo<R, T, SYM>;
    constexpr int M = G::M, P = G::P, PITCH = G::PITCH;
    constexpr bool X4 = G::X4;
    const StftParams& p = a.s;
    const PeaksParams& pk = a.p;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x >> 6;
    const int K = pk.K;
    const int kpad = (K + 3) & ~3;
    T* const winL = (T*)(smem + G::OFF_WIN);
    cx<T>* const t1L = (cx<T>*)(smem + G::OFF_T1);
    cx<T>* const t2L = (cx<T>*)(smem + G::OFF_T2);
    cx<T>* const tw3 = (cx<T>*)(smem + G::OFF_TW3);
    unsigned char* const wbase = smem + G::OFF_BUF + (size_t)wid * (G::BUFC * 2 * sizeof(T) + pv_stage_bytes(K));
    cx<T>* const dz = (cx<T>*)wbase;
    // the peak search's arrays live in dz between the untangle's reads and the next frame's exchange
    constexpr int CAP = M / 2 + 4;
    T* const y = (T*)wbase;
    T* const cs = y + M;
    int* const ci = (int*)(cs + CAP);
    int* const sel = ci + CAP;
    int* const lst = (int*)(wbase + G::BUFC * 2 * sizeof(T));            // [GF][kpad]
    int* const cntv = lst + GF * kpad;                                    // [GF]
    long long* const relv = (long long*)(((uintptr_t)(cntv + GF) + 7) & ~(uintptr_t)7);   // [GF]
    double* const totv = (double*)(relv + GF);                            // [GF]
    {
        const cx<T>* tab = (const cx<T>*)p.twiddle;
        constexpr int NMASK = G::N - 1;
        for (int i = threadIdx.x; i < (SYM ? G::N / 2 : G::N); i += blockDim.x) winL[i] = ((const T*)p.win)[i];
        if constexpr (X4) {
            for (int i = threadIdx.x; i < 256; i += blockDim.x) t1L[i] = tab[((G::N / 256) * (i & 15) * (i >> 4)) & NMASK];    // [q][l] W_256^(l q)
            for (int i = threadIdx.x; i < 512; i += blockDim.x) {   // [j][u][lane]: W_N^k1 (u = 0), W_1024^(u k1); k1 = lane + 64 j
                const int ln = i & 63, u = (i >> 6) & 3, k1 = ln + 64 * (i >> 8);
                tw3[i] = tab[(u == 0 ? k1 : 2 * u * k1) & NMASK];
            }
        } else {
        for (int i = threadIdx.x; i < R * 64; i += blockDim.x) t1L[i] = tab[(2 * (i & 63) * (i >> 6)) & NMASK];
        for (int i = threadIdx.x; i < 64; i += blockDim.x) t2L[i] = tab[((G::N / 64) * (i % P) * (i / P)) & NMASK];   // [t2][l1]
        for (int i = threadIdx.x; i <= G::HALF / 2; i += blockDim.x) tw3[i] = tab[i];
        }
    }
    __syncthreads();
    const int Q = lane / P, L1 = lane % P;
    T csg[G::LOGP > 0 ? G::LOGP : 1];
    cx<T> cw[G::LOGP > 0 ? G::LOGP : 1];
    {
        const cx<T>* tab = (const cx<T>*)p.twiddle;
        constexpr int NMASK = G::N - 1;
#pragma unroll
        for (int s = 0; s < G::LOGP; s++) {
            const int h = P >> (s + 1);
            const bool up = (L1 & h) != 0;
            csg[s] = up ? (T)-1 : (T)1;
            const cx<T> wv = tab[((G::N / (2 * h)) * (L1 % h)) & NMASK];
            cw[s] = up ? wv : mkc<T>((T)1, (T)0);
        }
    }
    int t1v = 0;
#pragma unroll
    for (int b = 0; b < G::LOGP; b++) if (L1 & (1 << b)) t1v |= 1 << (G::LOGP - 1 - b);

    // ---- rows of this wave: result rows rel in [rel0, rel1) = workspace rows rel0 + 1 .. rel1; workspace row rel0 is
    // the predecessor of its first one and is computed here too (spectrum only)
    const int64_t W = (int64_t)gridDim.x * nw;
    const int64_t w = (int64_t)blockIdx.x * nw + wid;
    const int64_t nrows = p.ws_rows - 1;
    const int64_t rel0 = nrows * w / W, rel1 = nrows * (w + 1) / W;
    if (rel0 >= rel1) return;
    const int64_t rows1 = p.F + 1;
    auto row_src = [&](int64_t j) -> const InT* {                     // samples of workspace row j; nullptr: zero row
        if (j > rel1) return nullptr;
        const int64_t g = p.R0 - 1 + j;
        if (g < 0 || g >= p.total_rows) return nullptr;
        const int64_t b = g / rows1, q = g - b * rows1;
        if (q == 0) return nullptr;
        return (const InT*)p.x + b * p.sig_stride + (q - 1) * (int64_t)p.hop;
    };
    // samples (2l + 128 r, + 1) of the next row, kept as float32 until the window multiply unless both the input and
    // the arithmetic are float64 (int16 and float32 samples are exact in float32: half the registers across the frame)
    using RawT = typename std::conditional<(sizeof(T) == 8 && sizeof(InT) == 8), double, float>::type;
    RawT raw[2 * R];
    const int lofs = X4 ? lofs4(lane) : 2 * lane;                     // sample offset of the lane's first pair (X4: pair 4 l + u of lane 16 u + l)
    auto prefetch_part = [&](const InT* src, int part) {
        if (src == nullptr) return;
        constexpr int PR = R / 4;
#pragma unroll
        for (int r = part * PR; r < (part + 1) * PR; r++) {
            const InT* q = src + lofs + 128 * r;
            raw[2 * r] = (RawT)q[0]; raw[2 * r + 1] = (RawT)q[1];
        }
    };

    PeakConst pc;
    pc.fstep = pk.fstep; pc.dt = pk.dt; pc.nfft = pk.nfft; pc.hop = pk.hop; pc.wfbin = pk.wfbin;
    int LPF = 1;
    while (LPF < K && LPF < 64) LPF <<= 1;
    const int fpp = 64 / LPF;

    // per-peak phase vocoder arithmetic on the (frame, peak) pairs of the staged frames [0, ng) (k_peaks.hip, phase B)
    auto flush = [&](int ng) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's spectrum rows are in L2
        wave_sync();
        // (the lane index goes through an opaque move here and in the peak search below: derived indices and addresses
        // are then computed where they are used instead of being hoisted out of the frame loop, where they would
        // occupy -- and spill -- registers across the transform)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if constexpr (DENSE) {
            // ---- DENSE: 8 < npks <= 32 at float64 (a separate instantiation: the others are the code they were) (the reference's default is 20): the staged frames' peaks taken 64 at a time ACROSS the frames instead of
            // 64 / pow2(npks) frames at a time -- at npks 20 a pass per two frames, on signals whose frames keep 8 peaks.  A pass of
            // this arithmetic (two float64 arctangents per peak) costs what it costs whether 16 or 64 of its lanes hold a peak.
            // Same peaks, same arithmetic, same places: bit-identical to the loop below (PVX_PV_NO_DENSE=1, GPU test).
            static_assert(GF == 8, "two 16-byte reads of the counts");
            const int4 ca = *(const int4*)cntv, cb = *((const int4*)cntv + 1);
            int cnts[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w}, offs[8], nprev[8];
            int total = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) { cnts[j] = j < ng ? cnts[j] : 0; offs[j] = total; total += cnts[j]; nprev[j] = 0; }     // (a staged frame's count is >= 0: zero rows stage nothing)
            for (int i0 = 0; i0 < total; i0 += 64) {
                const int i = i0 + ln;
                bool valid = i < total;
                int g = 0, off = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) { if (cnts[j] > 0 && offs[j] <= i) { g = j; off = offs[j]; } }
                const int e = i - off;
                const int64_t rel = (int64_t)relv[g];
                const int64_t gr = p.R0 + rel;
                const int64_t b = gr / rows1;
                const int64_t fr = gr - b * rows1 - 1;
                const int64_t orow = b * p.F + fr;
                const T* cur = (const T*)p.spec + (size_t)(rel + 1) * p.ldo * 2;
                const T* prv = (const T*)p.spec + (size_t)rel * p.ldo * 2;
                const bool use_prev0 = (pk.prev0 != nullptr) && (orow == 0);
                int nbin = 0;
                double freq = 0.0, dfb = 0.0, thisph = 0.0, mg = 0.0;
                if (valid) {
                    nbin = lst[g * kpad + e];
                    const T re = ldw(cur + 2 * nbin), im = ldw(cur + 2 * nbin + 1);
                    T pr, pi;
                    if (use_prev0) { pr = (T)pk.prev0[2 * nbin]; pi = (T)pk.prev0[2 * nbin + 1]; }
                    else { pr = ldw(prv + 2 * nbin); pi = ldw(prv + 2 * nbin + 1); }
                    const int imin = nbin - 1 > 1 ? nbin - 1 : 1;     // PV.py:197-199: 3-bin energy, bin 0 excluded
                    int imax = nbin + 1 < M ? nbin + 1 : M;
                    if (imax > M - 1) imax = M - 1;
                    T s3 = (T)0;
                    for (int j = imin; j <= imax; j++) { const T ar = ldw(cur + 2 * j), c = ldw(cur + 2 * j + 1); s3 = s3 + (ar * ar + c * c); }
                    const PeakOut o = peak_math<T>(nbin, re, im, pr, pi, s3, pc);
                    freq = o.freq; dfb = o.dfb; thisph = o.thisph; mg = o.mag;
                    valid = o.valid;
                }
                const unsigned long long ball = __ballot(valid);
                // the lanes of this pass that hold frame j's peaks: [max(offs[j], i0), min(offs[j] + cnts[j], i0 + 64)) - i0
                unsigned long long gm = 0ull;
                int before = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    int lo = offs[j] - i0, hi = offs[j] + cnts[j] - i0;
                    lo = lo < 0 ? 0 : lo; hi = hi > 64 ? 64 : hi;
                    const unsigned long long mj = hi > lo ? ((hi - lo >= 64 ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << lo) : 0ull;
                    if (j == g) { gm = mj; before = nprev[j]; }
                    nprev[j] += __popcll(ball & mj);
                }
                if (valid) {
                    const int o = before + __popcll(ball & gm & ((1ull << ln) - 1ull));
                    pk.binno[orow * K + o] = (double)nbin;
                    pk.f[orow * K + o] = freq;
                    pk.mag[orow * K + o] = mg;
                    pk.ph[orow * K + o] = thisph;
                    pk.realph[orow * K + o] = thisph + kPi * dfb / pk.fstep;     // PV.py:207
                }
            }
            {
                // zero padding (PV.py:226-239) and the frames' scalars: eight lanes per staged frame
                const int g2 = ln >> 3, c2 = ln & 7;
                int n2 = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) { if (j == g2) n2 = nprev[j]; }
                if (g2 < ng) {
                    const int64_t rel = (int64_t)relv[g2];
                    const int64_t gr = p.R0 + rel;
                    const int64_t b = gr / rows1;
                    const int64_t fr = gr - b * rows1 - 1;
                    const int64_t orow = b * p.F + fr;
                    for (int j = n2 + c2; j < K; j += 8) { pk.binno[orow * K + j] = 0.0; pk.f[orow * K + j] = 0.0; pk.mag[orow * K + j] = 0.0; pk.ph[orow * K + j] = 0.0; pk.realph[orow * K + j] = 0.0; }
                    if (c2 == 0) {
                        if (pk.totalmag) pk.totalmag[orow] = sqrt(totv[g2]);                                  // PV.py:210
                        if (pk.t) pk.t[orow] = ((double)(fr * (int64_t)pk.hop) + pk.nfft / 2.0) / pk.sr;     // PV.py:247
                    }
                }
            }
            wave_sync();
            return;
        }
        const int gl = ln / LPF, e0 = ln - gl * LPF;
        const unsigned long long gmask = (LPF == 64 ? ~0ull : ((1ull << LPF) - 1ull)) << (gl * LPF);
        for (int gb = 0; gb < ng; gb += fpp) {
            const int g = gb + gl;
            const bool gvalid = g < ng;
            const int cnt = gvalid ? cntv[g] : -1;
            const int64_t rel = gvalid ? (int64_t)relv[g] : rel0;
            const int64_t gr = p.R0 + rel;
            const int64_t b = gr / rows1;
            const int64_t fr = gr - b * rows1 - 1;
            const int64_t orow = b * p.F + fr;
            const T* cur = (const T*)p.spec + (size_t)(rel + 1) * p.ldo * 2;
            const T* prv = (const T*)p.spec + (size_t)rel * p.ldo * 2;
            const bool use_prev0 = (pk.prev0 != nullptr) && (orow == 0);
            double* of = pk.f + orow * K;
            double* om = pk.mag + orow * K;
            double* op = pk.ph + orow * K;
            double* orp = pk.realph + orow * K;
            double* ob = pk.binno + orow * K;
            int nout = 0;
            for (int eb = 0; eb < K; eb += LPF) {                     // wave-uniform bound (ballots inside)
                const int e = eb + e0;
                bool valid = (cnt >= 0) && (e < cnt);
                int nbin = 0;
                double freq = 0.0, dfb = 0.0, thisph = 0.0, mg = 0.0;
                if (valid) {
                    nbin = lst[g * kpad + e];
                    const T re = ldw(cur + 2 * nbin), im = ldw(cur + 2 * nbin + 1);
                    T pr, pi;
                    if (use_prev0) { pr = (T)pk.prev0[2 * nbin]; pi = (T)pk.prev0[2 * nbin + 1]; }
                    else { pr = ldw(prv + 2 * nbin); pi = ldw(prv + 2 * nbin + 1); }
                    // PV.py:197-199: 3-bin energy, bin 0 excluded
                    const int imin = nbin - 1 > 1 ? nbin - 1 : 1;
                    int imax = nbin + 1 < M ? nbin + 1 : M;
                    if (imax > M - 1) imax = M - 1;
                    T s3 = (T)0;
                    for (int j = imin; j <= imax; j++) { const T ar = ldw(cur + 2 * j), c = ldw(cur + 2 * j + 1); s3 = s3 + (ar * ar + c * c); }
                    const PeakOut o = peak_math<T>(nbin, re, im, pr, pi, s3, pc);
                    freq = o.freq; dfb = o.dfb; thisph = o.thisph; mg = o.mag;
                    valid = o.valid;
                }
                const unsigned long long bal = __ballot(valid) & gmask;
                if (valid) {
                    const int o = nout + __popcll(bal & ((1ull << ln) - 1ull));
                    ob[o] = (double)nbin;
                    of[o] = freq;
                    om[o] = mg;
                    op[o] = thisph;
                    orp[o] = thisph + kPi * dfb / pk.fstep;           // PV.py:207
                }
                nout += __popcll(bal);
            }
            if (cnt >= 0) {
                for (int j = nout + e0; j < K; j += LPF) {            // zero padding, PV.py:226-239
                    ob[j] = 0.0; of[j] = 0.0; om[j] = 0.0; op[j] = 0.0; orp[j] = 0.0;
                }
                if (e0 == 0) {
                    if (pk.totalmag) pk.totalmag[orow] = sqrt(totv[g]);                                  // PV.py:210
                    if (pk.t) pk.t[orow] = ((double)(fr * (int64_t)pk.hop) + pk.nfft / 2.0) / pk.sr;     // PV.py:247
                }
            }
        }
        wave_sync();
    };

    {
        const InT* s0 = row_src(rel0);
        prefetch_part(s0, 0); prefetch_part(s0, 1); prefetch_part(s0, 2); prefetch_part(s0, 3);
    }
    int ng = 0;
    for (int64_t j = rel0; j <= rel1; ++j) {
        cx<T>* out = (cx<T>*)p.spec + (size_t)j * p.ldo;
        const bool zero_row = row_src(j) == nullptr;
        const bool with_peaks = j > rel0;
        const InT* nsrc = row_src(j + 1);                             // (nullptr once the window has been slid)
        if constexpr (R >= PVX_PV_PRIO_MINR) __builtin_amdgcn_s_setprio(PVX_PV_PRIO_T);      // issue priority by phase (k_fused_rev.hip): the transform over the peak search
        cx<T> z[R];
        {
            // the lane's window pairs, ALL in flight before the first product (one at a time, each behind its own wait -- what the
            // compiler makes of the plain loop -- a frame starts with sixteen LDS round trips in a row)
            // (float64 samples at nfft 2048 hold their rows in register pairs: two batches of eight there, or a register spills)
            constexpr int WB = (sizeof(RawT) == 8 && sizeof(T) == 8 && R == 16) ? R / 2 : R;
#pragma unroll
            for (int r0 = 0; r0 < R; r0 += WB) {
                cx<T> wq[WB];
#pragma unroll
                for (int r = 0; r < WB; r++) {
                    // (SYM: the pair n, n + 1 of the upper half is w[N-1-n], w[N-2-n]: the pair at N-2-n, read backwards)
                    const bool up = SYM && (r0 + r) >= R / 2;
                    wq[r] = *(const cx<T>*)(winL + (up ? G::N - 2 - (lofs + 128 * (r0 + r)) : lofs + 128 * (r0 + r)));
                }
#pragma unroll
                for (int r = 0; r < WB; r++) asm volatile("" : "+v"(wq[r].x), "+v"(wq[r].y));
#pragma unroll
                for (int r = 0; r < WB; r++) {
                    const bool up = SYM && (r0 + r) >= R / 2;
                    z[r0 + r] = mkc<T>((T)raw[2 * (r0 + r)] * (up ? wq[r].y : wq[r].x), (T)raw[2 * (r0 + r) + 1] * (up ? wq[r].x : wq[r].y));
                    asm volatile("" : "+v"(z[r0 + r].x), "+v"(z[r0 + r].y));       // the multiplies stay above the next loads
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // rows j and j + 1 are consecutive frames of one signal: slide the window, fetch the hop's new samples
        const bool slide = (H > 0) && !zero_row && (nsrc != nullptr);
        if constexpr (H > 0) {
            if (slide) {
#pragma unroll
                for (int r = 0; r < R - H; r++) { raw[2 * r] = raw[2 * (r + H)]; raw[2 * r + 1] = raw[2 * (r + H) + 1]; }
#pragma unroll
                for (int r = R - H; r < R; r++) {
                    const InT* q = nsrc + lofs + 128 * r;
                    raw[2 * r] = (RawT)q[0]; raw[2 * r + 1] = (RawT)q[1];
                }
                nsrc = nullptr;                                       // nothing else to fetch for row j + 1
            }
        }
        prefetch_part(nsrc, 0);
        if (zero_row) {
            // the zero frame in front of every signal (PV.py:121): a spectrum row of zeros, no result row
            prefetch_part(nsrc, 1); prefetch_part(nsrc, 2); prefetch_part(nsrc, 3);
            for (int k = lane; k < M; k += 64) out[k] = mkc<T>((T)0, (T)0);
            continue;
        }
        constexpr bool LATE = sizeof(RawT) == 8 && R == 16;
        if constexpr (X4) {
            // ---- four 256-point transforms, then the radix-4 join inside the untangle (pvx_stft4.h): bins straight to the
            // workspace row, |X|^2 of every bin to LDS where the transform buffer was
            fft4_quartersT<T, (LATE ? 8 : 16)>(z, dz, t1L, lane, [&]() { prefetch_part(nsrc, 1); }, [&]() { if constexpr (!LATE) prefetch_part(nsrc, 2); },
                              [&]() { if constexpr (!LATE) prefetch_part(nsrc, 3); });
            int lu = lane;
            asm volatile("" : "+v"(lu));                             // (see flush: addresses derived here, not hoisted)
            wave_sync();
            Join4In<T> in;
            join4_read<T>(dz, lu, in);
            wave_sync();                                              // dz is free: the magnitude row goes there
            if constexpr (LATE) { __builtin_amdgcn_sched_barrier(0); prefetch_part(nsrc, 2); prefetch_part(nsrc, 3); __builtin_amdgcn_sched_barrier(0); }
            join4_emit<T>(in, tw3, lu, [&](int k, cx<T> x) {
#ifndef PVX_AB_NO_ROWSTORE      // tools/ab: timing-only build without the spectrum rows in HBM (the peaks then read stale rows)
                row_store(&out[k], x);
#else
                if (p.ldo < 0) out[k] = x;
#endif
                if (with_peaks) y[k] = x.x * x.x + x.y * x.y;
            });
        } else {
        dftT<R, T>(z);                                                // stage 1
        __builtin_amdgcn_sched_barrier(0);
        prefetch_part(nsrc, 1);
        cx<T> tq1[R];
        lds_gather<1, R, T>(tq1, t1L + lane, 64);
#pragma unroll
        for (int q2 = 0; q2 < R; q2++) dz[q2 * PITCH + lane] = (q2 > 0) ? cmulT(z[q2], tq1[q2]) : z[q2];
        wave_sync();
#pragma unroll
        for (int l2 = 0; l2 < R; l2++) z[l2] = dz[Q * PITCH + L1 + P * l2];
        // float64 samples: the second half of the next row is fetched after the transform (with all of it in flight
        // through stages 2 and 3 the kernel needs more than 256 registers)
        if constexpr (!LATE) prefetch_part(nsrc, 2);
        wave_sync();
        dftT<R, T>(z);                                                // stage 2
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!LATE) prefetch_part(nsrc, 3);
        cx<T> tq2[R];
        lds_gather<1, R, T>(tq2, t2L + L1, P);
#pragma unroll
        for (int t = 0; t < R; t++) {
            // twiddle W_64^(l1 t2), then stage 3: P-point DFT across the P lanes of a group (decimation in frequency)
            cx<T> v = (t > 0) ? cmulT(z[t], tq2[t]) : z[t];
            if constexpr (G::LOGP >= 1) {
                if constexpr (P >= 16) v = xstepT<8, true, T>(v, csg[G::LOGP - 4], cw[G::LOGP - 4]);
                if constexpr (P >= 8) v = xstepT<4, true, T>(v, csg[G::LOGP - 3], cw[G::LOGP - 3]);
                if constexpr (P >= 4) v = xstepT<2, true, T>(v, csg[G::LOGP - 2], cw[G::LOGP - 2]);
                v = xstepT<1, false, T>(v, csg[G::LOGP - 1], cw[G::LOGP - 1]);
            }
            dz[zpadT<R, T>(Q + R * t + G::R2 * t1v)] = v;
        }
        wave_sync();
        // ---- untangle straight to the workspace row: pairs (k, M-k), k = lane + 64 j2 (k_stft.hip); the magnitudes
        // of both bins stay in registers until every lane has read its part of dz
        constexpr int NPAIR = R / 2;
        int lu = lane;
        asm volatile("" : "+v"(lu));                                 // (see flush: addresses derived here, not hoisted)
        cx<T> za[NPAIR], zb[NPAIR];
#pragma unroll
        for (int j2 = 0; j2 < NPAIR; j2++) {
            const int k = lu + 64 * j2;
            za[j2] = dz[zpadT<R, T>(k)];
            zb[j2] = dz[zpadT<R, T>((M - k) & (M - 1))];
        }
        const cx<T> zc = dz[zpadT<R, T>(G::HALF)];
        wave_sync();                                                  // dz is free: the magnitude row goes there
        if constexpr (LATE) { __builtin_amdgcn_sched_barrier(0); prefetch_part(nsrc, 2); prefetch_part(nsrc, 3); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int j2 = 0; j2 < NPAIR; j2++) {
            const int k = lu + 64 * j2;
            const int km = (M - k) & (M - 1);
            const cx<T> S = mkc<T>(za[j2].x + zb[j2].x, za[j2].y - zb[j2].y);
            const cx<T> D = mkc<T>(za[j2].x - zb[j2].x, za[j2].y + zb[j2].y);
            const cx<T> O = mkc<T>((T)0.5 * D.y, (T)-0.5 * D.x);
            cx<T> wk;
            if (j2 < NPAIR / 2) wk = tw3[k];                         // k <= nfft/8
            else { const cx<T> e = tw3[G::HALF - k]; wk = mkc<T>(-e.y, -e.x); }
            const cx<T> Pk = cmulT(O, wk);
            const cx<T> x0 = mkc<T>(fmaT((T)0.5, S.x, Pk.x), fmaT((T)0.5, S.y, Pk.y));
            cx<T> x1 = mkc<T>(fmaT((T)0.5, S.x, -Pk.x), -fmaT((T)0.5, S.y, -Pk.y));
            int kk = km;
            if (j2 == 0 && lu == 0) { x1 = mkc<T>(zc.x, -zc.y); kk = G::HALF; }      // bin 0 pairs with itself; its slot takes bin M/2
            row_store(&out[k], x0);
            row_store(&out[kk], x1);
            if (with_peaks) {
                // the peak search runs on |X|^2: every test it makes (local maximum, threshold, ranking, salience) is
                // monotone in |X| (k_peaks.hip); plain products and one sum, the same value whichever kernel computes it
                y[k] = x0.x * x0.x + x0.y * x0.y;
                y[kk] = x1.x * x1.x + x1.y * x1.y;
            }
        }
        }
        wave_sync();
        if constexpr (R >= PVX_PV_PRIO_MINR) __builtin_amdgcn_s_setprio(PVX_PV_PRIO_S);
        if (!with_peaks) continue;

        // ---- extremes and energy of the row, in k_phase_peaks' order of summation (PV.py:173, 210; PF.py:60, 164)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        T lmax = (T)-INFINITY, lmin = (T)INFINITY;
        double lsum = 0.0;
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int i = 0; i < M / 128; i++) {
                const float2 v = *(const float2*)(y + 2 * (ln + 64 * i));
                lmax = fmaxf(lmax, fmaxf(v.x, v.y));
                lmin = fminf(lmin, fminf(v.x, v.y));
                lsum += (double)v.x + (double)v.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < M / 64; i++) {
                const T m0 = y[ln + 64 * i];
                lmax = m0 > lmax ? m0 : lmax;
                lmin = m0 < lmin ? m0 : lmin;
                lsum += m0;
            }
        }
        const T maxv = wave_max(lmax);
        const T minv = wave_min(lmin);
        const double tot = wave_sum(lsum);
        // ---- PeakFinder(famp, npeaks, minrattomax) + filter_by_salience(rad=5)  (PV.py:175-178)
        // PF.py:69-70, 174 on the squared row: |X| - miny > minamp - miny  <=>  |X|^2 - mine > minamp^2 - mine; minamp == 0
        // means minamp = miny there and the threshold is then exactly 0 (k_peaks.hip)
        double minamp;
        if constexpr (sizeof(T) == 4) minamp = (double)sqrtf(maxv) * pk.thr;        // PF.py:60
        else minamp = sqrt((double)maxv) * pk.thr;
        const double th = (minamp != 0.0) ? minamp * minamp - (double)minv : 0.0;
        // the candidate scan in pieces of at most 512 bins: all of a piece's LDS reads are in flight at once (peak_scan),
        // and a piece's values fit beside the prefetched samples of the next row
        constexpr int PIECE = M < 512 ? M : 512;
        int C = 0;
#pragma unroll 1
        for (int kb = 0; kb < M; kb += PIECE) C += peak_scan<T, PIECE / 64>(y, kb, PIECE, M, minv, th, cs + C, ci + C, ln);
        wave_sync();
        if constexpr (R >= PVX_PV_PRIO_MINR) __builtin_amdgcn_s_setprio(PVX_PV_PRIO_C);
        const int nsel = peak_pick<T>(y, cs, ci, sel, M, K, C, th, ln);
        int nk = 0;
        int rad = pk.rad;
        asm volatile("" : "+s"(rad));                                 // (like ln above: nothing derived from it is hoisted)
        for (int eb = 0; eb < nsel; eb += 64) {
            const int e = eb + ln;
            int pb = 0;
            bool keep = false;
            if (e < nsel) { pb = sel[e]; keep = salient<T>(y, M, pb, rad); }
            const unsigned long long bal = __ballot(keep);
            if (keep) lst[ng * kpad + nk + lane_prefix(bal)] = pb;
            nk += __popcll(bal);
        }
        if (lane == 0) { cntv[ng] = nk; relv[ng] = j - 1; totv[ng] = tot; }
        ng++;
        if (ng == GF) { flush(ng); ng = 0; }
        wave_sync();                                                  // y / cs / ci / sel are read: dz is free again
    }
    if (ng > 0) flush(ng);
}

template <int R, typename T> size_t pv_lds(int nw, int K, bool sym) {
    if constexpr (R == 16) { if (sym) return pv_total_lds<R, T, true>(nw, K); }
    return pv_total_lds<R, T>(nw, K);
}

template <int R, typename T, bool DENSE> int launch_stft_pv_rd(const StftPvParams& a, int x_dtype, hipStream_t s) {
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    const int K = a.p.K;
    // float64: LDS admits 7 waves per CU at nfft 2048 (npks <= 12); float32 (<= 176 registers: two waves per SIMD): two workgroups of 4
    int nw = (sizeof(T) == 8 && R == 16) ? 7 : 4;                      // (the others hold two waves per SIMD by registers)
    // a symmetric window at nfft 2048 / float64: half of it in LDS, an eighth wave (PvGeo)
    bool sym = false;
    if constexpr (sizeof(T) == 8 && R == 16) sym = a.win_symmetric != 0 && getenv("PVX_STFT_PV_NOSYM") == nullptr && pv_lds<R, T>(8, K, true) <= 160 * 1024;
    if (a.s.hop != 32 * R && a.s.hop != 64 * R) sym = false;           // (instantiated for the two sliding-window hops)
    if (sym) nw = 8;
    if (const char* e = getenv("PVX_STFT_PV_NW")) { const int v = atoi(e); if (v >= 1 && v <= (sym ? 8 : 7)) nw = v; }      // tests: other workgroups
    while (nw > 1 && pv_lds<R, T>(nw, K, sym) > 160 * 1024) nw--;
    const size_t lds = pv_lds<R, T>(nw, K, sym);
    // the sliding-window instantiations for the two usual hops
    const int H = (a.s.hop == 32 * R) ? R / 4 : (a.s.hop == 64 * R) ? R / 2 : 0;
    const void* fn = nullptr;
    switch (x_dtype) {
        case PVX_F32: fn = H == R / 4 ? (const void*)k_stft_pv<R, T, float, R / 4, false, DENSE> : H ? (const void*)k_stft_pv<R, T, float, R / 2, false, DENSE> : (const void*)k_stft_pv<R, T, float, 0, false, DENSE>; break;
        case PVX_F64:
            if constexpr (sizeof(T) == 8 && R == 16) {
                // float64 samples into the float64 transform at nfft 2048 with a hop that does not slide the window: the whole
                // next row would wait in 64 registers beside the transform -- pvx_stft_pv_takes() sends that plan through
                // k_stft + k_phase_peaks instead
                if (H == 0) { pvx_set_error("k_stft_pv does not take float64 samples at nfft 2048 with hop %d", a.s.hop); return PVX_ERR_UNSUPPORTED; }
                fn = H == R / 4 ? (const void*)k_stft_pv<R, T, double, R / 4, false, DENSE> : (const void*)k_stft_pv<R, T, double, R / 2, false, DENSE>;
            } else {
                fn = H == R / 4 ? (const void*)k_stft_pv<R, T, double, R / 4, false, DENSE> : H ? (const void*)k_stft_pv<R, T, double, R / 2, false, DENSE> : (const void*)k_stft_pv<R, T, double, 0, false, DENSE>;
            }
            break;
        case PVX_I16: fn = H == R / 4 ? (const void*)k_stft_pv<R, T, int16_t, R / 4, false, DENSE> : H ? (const void*)k_stft_pv<R, T, int16_t, R / 2, false, DENSE> : (const void*)k_stft_pv<R, T, int16_t, 0, false, DENSE>; break;
        default: pvx_set_error("bad x_dtype %d", x_dtype); return PVX_ERR_INVALID;
    }
    if constexpr (sizeof(T) == 8 && R == 16) {
        if (sym && H > 0) {
            switch (x_dtype) {
                case PVX_F32: fn = H == R / 4 ? (const void*)k_stft_pv<R, T, float, R / 4, true, DENSE> : (const void*)k_stft_pv<R, T, float, R / 2, true, DENSE>; break;
                case PVX_F64: fn = H == R / 4 ? (const void*)k_stft_pv<R, T, double, R / 4, true, DENSE> : (const void*)k_stft_pv<R, T, double, R / 2, true, DENSE>; break;
                default: fn = H == R / 4 ? (const void*)k_stft_pv<R, T, int16_t, R / 4, true, DENSE> : (const void*)k_stft_pv<R, T, int16_t, R / 2, true, DENSE>; break;
            }
        }
    }
    if (lds > 64 * 1024) PVX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((160 * 1024) / lds);                            // workgroups that fit a CU's LDS side by side
    if (per_cu < 1) per_cu = 1;
    if (per_cu * nw > 16) per_cu = 16 / nw > 0 ? 16 / nw : 1;
    { const int nb = pvx_resident_blocks(fn, 64 * nw, lds); if (nb >= 1 && nb < per_cu) per_cu = nb; }      // (registers: pvx_internal.h)
    const int64_t nrows = a.s.ws_rows - 1;
    int64_t nblocks = (int64_t)ncu * per_cu;
    if (const char* e = getenv("PVX_STFT_PV_BLOCKS")) { const long long v = atoll(e); if (v >= 1) nblocks = v; }   // tests: other grids
    const int64_t maxb = (nrows + nw - 1) / nw;                        // never more waves than rows
    if (nblocks > maxb) nblocks = maxb > 0 ? maxb : 1;
    dim3 grid((unsigned)nblocks), block(64 * nw);
    StftPvParams arg = a;
    void* args[] = {&arg};
    PVX_HIP_CHECK(hipLaunchKernel(fn, grid, block, args, lds, s));
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}

// float64 at 8 < npks <= 32: the instantiation whose per-peak pass runs across the staged frames (the kernel's DENSE); PVX_PV_NO_DENSE=1:
// the other one (A/B, tests)
template <int R, typename T> int launch_stft_pv_r(const StftPvParams& a, int x_dtype, hipStream_t s) {
    if constexpr (sizeof(T) == 8) {
        if (a.p.K > 8 && a.p.K <= 32 && getenv("PVX_PV_NO_DENSE") == nullptr) return launch_stft_pv_rd<R, T, true>(a, x_dtype, s);
    }
    return launch_stft_pv_rd<R, T, false>(a, x_dtype, s);
}

template <int R, typename T> bool pv_fits(int K) {
    return pv_search_bytes<R, T>(K) <= StftGeo<R, T>::BUFC * 2 * sizeof(T) && pv_total_lds<R, T>(1, K) <= 160 * 1024;
}

}  // namespace

// does the one-launch kernel take this call?  (pvx_stft_pv_supported says whether the plan's shape fits at all)
int pvx_stft_pv_takes(int nfft, int precision, int x_dtype, int hop) {
    if (nfft == 2048 && precision == 64 && x_dtype == PVX_F64 && hop != 512 && hop != 1024) return 0;
    return 1;
}

int pvx_stft_pv_supported(int nfft, int precision, int K) {
    if (precision != 64 && precision != 32) return 0;
    switch (nfft) {
        case 512: return precision == 64 ? pv_fits<4, double>(K) : pv_fits<4, float>(K);
        case 1024: return precision == 64 ? pv_fits<8, double>(K) : pv_fits<8, float>(K);
        case 2048: return precision == 64 ? pv_fits<16, double>(K) : pv_fits<16, float>(K);
        default: return 0;
    }
}

// result rows [R0, R0 + nrows) of the general path in one launch: spectra into spec rows [0, nrows], peaks into the
// result arrays of pp (pp.spec / pp.ldo are taken from the arguments)
int pvx_launch_stft_pv(const FrameParams& fp, const PeaksParams& pp, void* spec, int64_t ldo, const void* twiddle, int x_dtype,
                       int precision, hipStream_t s) {
    if (fp.ws_rows <= 1) return PVX_OK;
    StftPvParams a;
    a.s.x = fp.x; a.s.nsamp = fp.nsamp; a.s.sig_stride = fp.sig_stride; a.s.F = fp.F; a.s.R0 = fp.R0; a.s.ws_rows = fp.ws_rows;
    a.s.total_rows = fp.total_rows; a.s.hop = fp.hop; a.s.win = fp.win; a.s.twiddle = twiddle; a.s.spec = spec; a.s.ldo = ldo;
    a.p = pp; a.p.spec = spec; a.p.ldo = ldo;
    a.win_symmetric = fp.win_symmetric;
    if (precision == 64) {
        switch (fp.nfft) {
            case 512: return launch_stft_pv_r<4, double>(a, x_dtype, s);
            case 1024: return launch_stft_pv_r<8, double>(a, x_dtype, s);
            case 2048: return launch_stft_pv_r<16, double>(a, x_dtype, s);
            default: break;
        }
    } else {
        switch (fp.nfft) {
            case 512: return launch_stft_pv_r<4, float>(a, x_dtype, s);
            case 1024: return launch_stft_pv_r<8, float>(a, x_dtype, s);
            case 2048: return launch_stft_pv_r<16, float>(a, x_dtype, s);
            default: break;
        }
    }
    pvx_set_error("the fused STFT + peaks kernel does not handle nfft=%d", fp.nfft);
    return PVX_ERR_UNSUPPORTED;
}
