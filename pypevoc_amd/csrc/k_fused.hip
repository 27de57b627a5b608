// k_fused.hip -- the whole analysis stage of one frame inside one wave64, nothing but the input
// samples and the <= K peaks ever touching HBM.  For nfft = 2048, float32:
//
//   PV.calc_fft_frame   pypevoc/PVAnalysis.py:150-158   x[pos:pos+nfft]*win -> FFT -> /wfact
//   PV.calc_pv_frame    pypevoc/PVAnalysis.py:160-211   abs, PeakFinder, salience, phase difference,
//                                                        instantaneous frequency, 3-bin energy, realph
//   PV.run_pv           pypevoc/PVAnalysis.py:213-264   frame loop and zero-padded packing
//
// Why: the three-kernel form (k_frames -> rocFFT -> k_peaks) moves 35 KB per frame through HBM /
// Infinity Cache; here a frame costs hop*4 = 2 KB of input (the 4-fold overlap between frames is
// served by L2) and (5K+2)*8 B of output.  The kernel is then bound by VALU/LDS work, not by HBM.
//
// Real FFT of nfft = 2048 samples = complex FFT of M = 1024 points z[j] = xw[2j] + i xw[2j+1]
// followed by the untangle X[k] = E[k] + W_2048^k O[k].  The 1024-point FFT is laid out on the 64
// lanes x 16 registers of one wave as 1024 = 16 (registers) x 16 (registers, after one LDS
// exchange) x 4 (across the lanes of a quad, by DPP):
//   stage 1  lane l holds z[l + 64 r], r < 16: radix-16 DFT over r in registers, twiddle W_1024^(l q)
//   exchange Y[l][q] -> LDS rows [q][l] (row pitch 68 complex: conflict-free both ways)
//   stage 2  lane (q, l1) = 4q + l1 reads Y[l1 + 4 l2][q], l2 < 16: radix-16 DFT over l2, twiddle W_64^(l1 t2)
//   stage 3  4-point DFT over l1 across the quad with two DPP exchanges (quad_perm xor 2, xor 1)
//   result   lane (q, l1) holds Z[q + 16 t2 + 256 t1], t1 = bitrev2(l1)
// Z goes to LDS in natural order (8 complex of padding per 256 keeps the writes conflict-free), is
// untangled in place into X[0..1024), |X| goes to a second LDS array, and from there on the frame is
// handled exactly like k_peaks.hip does (same PeakFinder core, same per-peak arithmetic), except
// that the previous frame's spectrum is the LDS buffer the wave filled one iteration earlier.
//
// Work distribution: persistent-style.  The launch has about one wave per SIMD of the chip; wave w
// owns the contiguous global rows [w*R/W, (w+1)*R/W) and recomputes the spectrum of the row before
// its first one (1 extra FFT per wave, a few percent).  No inter-wave communication at all.
#include "pvx_wave.h"

using namespace pvxw;

namespace {

constexpr int FN = 2048;            // nfft handled by this kernel
constexpr int FM = 1024;            // complex FFT length
constexpr int EXP = 68;             // exchange row pitch (complex)
constexpr int BUFC = 16 * EXP;      // complex slots per spectrum buffer (>= 1024 + 32)
constexpr int GF = 8;               // frames staged before the per-peak pass

__device__ __host__ __forceinline__ int zpad(int k) { return k + 8 * (k >> 8); }

// complex multiply with explicit fused multiply-adds: the same rounding wherever it is inlined, so
// results do not depend on which wave computes a frame
__device__ __forceinline__ void cmul(float& a, float& c, float wr, float wi) {
    const float nr = __builtin_fmaf(a, wr, -(c * wi));
    c = __builtin_fmaf(a, wi, c * wr);
    a = nr;
}

template <int CTRL> __device__ __forceinline__ float dppf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// ---- radix-16 DFT in registers (radix-4 x radix-4), forward, natural order in and out
__device__ __forceinline__ void dft16(float (&xr)[16], float (&xi)[16]) {
    constexpr float C1 = 0.92387953251128673848f;   // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f;   // sin(pi/8)
    constexpr float H = 0.70710678118654752440f;    // sqrt(1/2)
    // W16^m = (cr[m], ci[m]) for m = n1*k2
    constexpr float cr[10] = {1.f, C1, H, S1, 0.f, 0.f, -H, 0.f, 0.f, -C1};
    constexpr float ci[10] = {0.f, -S1, -H, -C1, -1.f, 0.f, -H, 0.f, 0.f, S1};
    float tr[16], ti[16];
#pragma unroll
    for (int n1 = 0; n1 < 4; n1++) {
        const float ar = xr[n1], ai = xi[n1], br = xr[n1 + 4], bi = xi[n1 + 4];
        const float cr_ = xr[n1 + 8], ci_ = xi[n1 + 8], dr = xr[n1 + 12], di = xi[n1 + 12];
        const float Ar = ar + cr_, Ai = ai + ci_, Br = ar - cr_, Bi = ai - ci_;
        const float Cr = br + dr, Ci = bi + di, Dr = br - dr, Di = bi - di;
        float yr[4], yi[4];
        yr[0] = Ar + Cr; yi[0] = Ai + Ci;
        yr[2] = Ar - Cr; yi[2] = Ai - Ci;
        yr[1] = Br + Di; yi[1] = Bi - Dr;        // B - i D
        yr[3] = Br - Di; yi[3] = Bi + Dr;        // B + i D
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) {
            const int m = n1 * k2;
            if (m == 0) { tr[n1 * 4 + k2] = yr[k2]; ti[n1 * 4 + k2] = yi[k2]; }
            else if (m == 4) { tr[n1 * 4 + k2] = yi[k2]; ti[n1 * 4 + k2] = -yr[k2]; }     // * (-i)
            else {
                float a = yr[k2], c = yi[k2];
                cmul(a, c, cr[m], ci[m]);
                tr[n1 * 4 + k2] = a; ti[n1 * 4 + k2] = c;
            }
        }
    }
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) {
        const float ar = tr[k2], ai = ti[k2], br = tr[4 + k2], bi = ti[4 + k2];
        const float cr_ = tr[8 + k2], ci_ = ti[8 + k2], dr = tr[12 + k2], di = ti[12 + k2];
        const float Ar = ar + cr_, Ai = ai + ci_, Br = ar - cr_, Bi = ai - ci_;
        const float Cr = br + dr, Ci = bi + di, Dr = br - dr, Di = bi - di;
        xr[k2] = Ar + Cr;      xi[k2] = Ai + Ci;
        xr[k2 + 8] = Ar - Cr;  xi[k2 + 8] = Ai - Ci;
        xr[k2 + 4] = Br + Di;  xi[k2 + 4] = Bi - Dr;
        xr[k2 + 12] = Br - Di; xi[k2 + 12] = Bi + Dr;
    }
}

template <typename InT> __device__ __forceinline__ float ld1(const InT* p) { return (float)*p; }

template <typename InT, bool AL2>
__device__ __forceinline__ void load_raw(const InT* x, int lane, float (&ra)[16], float (&rb)[16]) {
    // lane l takes z[l + 64 r] = (x[2l + 128 r], x[2l + 128 r + 1]): 512 contiguous bytes per
    // wave-instruction.  Issued one frame ahead of its use (software prefetch): the loads of row
    // g+1 are in flight while row g is transformed and searched for peaks.
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const InT* p = x + 2 * lane + 128 * r;
        if constexpr (AL2 && sizeof(InT) == 4) {
            const float2 v = *(const float2*)p;
            ra[r] = v.x; rb[r] = v.y;
        } else {
            ra[r] = ld1(p); rb[r] = ld1(p + 1);
        }
    }
}

struct FusedLds {       // per-wave carve
    float2* bufA;       // [BUFC]
    float2* bufB;       // [BUFC]
    float* y;           // [1024]
    float* cs;          // [516]
    int* ci;            // [516]
    int* sel;           // [kpad]
    int* sbin;          // [GF][kpad]
    float* sval;        // [GF][kpad][5]   re, im, pr, pi, s3
    int* cnt;           // [GF]
    int* frm;           // [GF]  frame index within its signal
    long long* orow;    // [GF]
    double* tot;        // [GF]
};

__host__ __device__ __forceinline__ size_t fused_lds_per_wave(int K) {
    const size_t kpad = (size_t)((K + 3) & ~3);
    size_t b = (size_t)BUFC * 8 * 2 + 1024 * 4 + 516 * 4 + 516 * 4 + kpad * 4 + (size_t)GF * kpad * 4 +
               (size_t)GF * kpad * 5 * 4 + GF * 4 + GF * 4;
    b = (b + 7) & ~(size_t)7;
    b += GF * 8 + GF * 8;
    return (b + 15) & ~(size_t)15;
}
constexpr size_t kTw3Bytes = 520 * 8;      // W_2048^k, k <= 512 (+ padding)
constexpr size_t kWinBytes = 2048 * 4;     // window / wfact
constexpr size_t kTw2Bytes = 16 * 4 * 8;   // W_64^(l1 t2) as [t2][l1]
constexpr size_t kFusedShared = kTw3Bytes + kWinBytes + kTw2Bytes;

template <typename InT, bool AL2>
__global__ __launch_bounds__(128) void k_fused_pv2048(FusedParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int K = p.K;
    const int kpad = (K + 3) & ~3;
    float2* tw3 = (float2*)smem;                                  // shared by the block
    unsigned char* base = smem + kFusedShared + fused_lds_per_wave(K) * wid;
    FusedLds L;
    L.bufA = (float2*)base;
    L.bufB = L.bufA + BUFC;
    L.y = (float*)(L.bufB + BUFC);
    L.cs = L.y + 1024;
    L.ci = (int*)(L.cs + 516);
    L.sel = L.ci + 516;
    L.sbin = L.sel + kpad;
    L.sval = (float*)(L.sbin + GF * kpad);
    L.cnt = (int*)(L.sval + GF * kpad * 5);
    L.frm = L.cnt + GF;
    L.orow = (long long*)(((uintptr_t)(L.frm + GF) + 7) & ~(uintptr_t)7);
    L.tot = (double*)(L.orow + GF);

    float2* winl = (float2*)(smem + kTw3Bytes);                   // window as (w[2i], w[2i+1])
    float2* tw2l = (float2*)(smem + kTw3Bytes + kWinBytes);       // [t2][l1]
    const float2* tab = (const float2*)p.twiddle;                 // W_2048^j, j < 2048
    for (int k = threadIdx.x; k <= 512; k += blockDim.x) tw3[k] = tab[k];
    for (int k = threadIdx.x; k < 1024; k += blockDim.x) winl[k] = ((const float2*)p.win)[k];
    for (int k = threadIdx.x; k < 64; k += blockDim.x) tw2l[k] = tab[(32 * (k & 3) * (k >> 2)) & 2047];
    __syncthreads();

    // ---- lane constants
    const int Q = lane >> 2, L1 = lane & 3;
    float w0[16], w1[16], t1r[16], t1i[16], t2r[16], t2i[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const float2 wv = winl[lane + 64 * r];
        w0[r] = wv.x; w1[r] = wv.y;
        const float2 a = tab[(2 * lane * r) & 2047];              // W_1024^(l q)
        t1r[r] = a.x; t1i[r] = a.y;
        const float2 b = tw2l[r * 4 + L1];                        // W_64^(l1 t2)
        t2r[r] = b.x; t2i[r] = b.y;
    }
    // keep the lane constants in registers: without this the compiler re-loads the twiddles from
    // global memory every frame (a full L2 round trip on the critical path) instead of holding them
#pragma unroll
    for (int r = 0; r < 16; r++) {
        asm volatile("" : "+v"(w0[r]), "+v"(w1[r]), "+v"(t1r[r]), "+v"(t1i[r]), "+v"(t2r[r]), "+v"(t2i[r]));
    }
    const float sA = (L1 & 2) ? -1.f : 1.f;
    // stage 3 step 2: res = alpha*u + beta*p;  lanes 0..3: alpha = 1,-1,1,i   beta = 1,1,-i,1
    const float alr = (L1 == 0 || L1 == 2) ? 1.f : (L1 == 1 ? -1.f : 0.f);
    const float ali = (L1 == 3) ? 1.f : 0.f;
    const float ber = (L1 == 2) ? 0.f : 1.f;
    const float bei = (L1 == 2) ? -1.f : 0.f;
    const int t1v = (L1 == 1) ? 2 : (L1 == 2 ? 1 : L1);

    // ---- rows of this wave
    const int64_t W = (int64_t)gridDim.x * nwaves;
    const int64_t w = (int64_t)blockIdx.x * nwaves + wid;
    const int64_t r0 = p.total_rows * w / W, r1 = p.total_rows * (w + 1) / W;
    if (r0 >= r1) return;

    PeakConst pc;
    pc.fstep = p.fstep; pc.dt = p.dt; pc.nfft = FN; pc.hop = p.hop; pc.wfbin = p.wfbin;

    float2* cur = L.bufA;
    float2* prv = L.bufB;
    float ra[16], rb[16];                                         // raw samples of the next row (prefetched)
    // Rows are addressed as (signal b, row-in-signal q), advanced incrementally: a 64-bit division
    // per frame costs more than the whole peak search.
    auto prefetch = [&](int64_t gn, int64_t bn, int64_t qn) {     // issue the loads of global row gn = (bn, qn)
        if (gn < 0 || gn >= r1 || qn == 0) return;
        load_raw<InT, AL2>((const InT*)p.x + bn * p.sig_stride + (qn - 1) * (int64_t)p.hop, lane, ra, rb);
    };

    // spectrum of global row g into `dst` (zeros for a zero row); with_mag: also |X| -> y and the
    // wave-reduced max / min / energy
    auto spectrum = [&](int64_t g, int64_t b, int64_t q, float2* dst, bool with_mag, float& maxy, float& miny, double& tot) {
        // (b, q) of row g + 1
        const int64_t qn = (q == p.F) ? 0 : q + 1;
        const int64_t bn = (q == p.F) ? b + 1 : b;
        if (g < 0 || q == 0) {
#pragma unroll
            for (int j = 0; j < 17; j++) dst[lane + 64 * j] = make_float2(0.f, 0.f);
            wave_sync();
            prefetch(g + 1, bn, qn);
            return;
        }
        float xr[16], xi[16];
#pragma unroll
        for (int r = 0; r < 16; r++) { xr[r] = ra[r] * w0[r]; xi[r] = rb[r] * w1[r]; }
        prefetch(g + 1, bn, qn);
        dft16(xr, xi);                                            // stage 1
#pragma unroll
        for (int q2 = 0; q2 < 16; q2++) {
            float a = xr[q2], c = xi[q2];
            if (q2 > 0) cmul(a, c, t1r[q2], t1i[q2]);
            dst[q2 * EXP + lane] = make_float2(a, c);
        }
        wave_sync();
#pragma unroll
        for (int l2 = 0; l2 < 16; l2++) {
            const float2 v = dst[Q * EXP + L1 + 4 * l2];
            xr[l2] = v.x; xi[l2] = v.y;
        }
        wave_sync();
        dft16(xr, xi);                                            // stage 2
#pragma unroll
        for (int t2 = 0; t2 < 16; t2++) {
            float a = xr[t2], c = xi[t2];
            if (t2 > 0) cmul(a, c, t2r[t2], t2i[t2]);
            // stage 3: 4-point DFT across the quad
            float pr_ = dppf<0x4E>(a), pi_ = dppf<0x4E>(c);       // lane ^ 2
            const float ur = __builtin_fmaf(sA, a, pr_), ui = __builtin_fmaf(sA, c, pi_);
            pr_ = dppf<0xB1>(ur); pi_ = dppf<0xB1>(ui);           // lane ^ 1
            const float zr = __builtin_fmaf(alr, ur, __builtin_fmaf(-ali, ui, __builtin_fmaf(ber, pr_, -(bei * pi_))));
            const float zi = __builtin_fmaf(alr, ui, __builtin_fmaf(ali, ur, __builtin_fmaf(ber, pi_, bei * pr_)));
            dst[zpad(Q + 16 * t2 + 256 * t1v)] = make_float2(zr, zi);
        }
        wave_sync();
        // ---- untangle in place: pairs (k, 1024-k), k = lane + 64 j; bins 0 and 512 have no partner.
        // Phase 1 reads everything (the loads do not wait for the in-place stores of other pairs),
        // phase 2 computes and stores.
        float lmax = -INFINITY, lmin = INFINITY, ls0 = 0.f, ls1 = 0.f;
        float2 za[8], zb[8], wv8[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = lane + 64 * j;
            const int km = (FM - k) & (FM - 1);                   // k = 0: Z[1024] == Z[0]
            za[j] = dst[zpad(k)];
            zb[j] = dst[zpad(km)];
            wv8[j] = tw3[k];
        }
        const float2 zc = dst[zpad(512)];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = lane + 64 * j;
            const int km = (FM - k) & (FM - 1);
            const float er = 0.5f * (za[j].x + zb[j].x), ei = 0.5f * (za[j].y - zb[j].y);      // E = (Za + conj Zb)/2
            const float orr = 0.5f * (za[j].y + zb[j].y), oi = -0.5f * (za[j].x - zb[j].x);    // O = (Za - conj Zb)/(2i)
            float pr2 = orr, pi2_ = oi;
            cmul(pr2, pi2_, wv8[j].x, wv8[j].y);                                    // P = W^k O
            const float x0r = er + pr2, x0i = ei + pi2_;                            // X[k]
            float x1r = er - pr2, x1i = pi2_ - ei;                                  // X[1024-k] = conj(E - P)
            int kk = km;
            if (j == 0) {
                // lane 0: k = 0 pairs with itself and its "partner" result is not a bin; that slot
                // takes bin 512, which pairs with itself too: X[512] = conj(Z[512])
                if (lane == 0) { x1r = zc.x; x1i = -zc.y; kk = 512; }
            }
            const float e0 = __builtin_fmaf(x0r, x0r, x0i * x0i), e1 = __builtin_fmaf(x1r, x1r, x1i * x1i);
            dst[zpad(k)] = make_float2(x0r, x0i);
            dst[zpad(kk)] = make_float2(x1r, x1i);
            if (with_mag) {
                // v_sqrt_f32 (1 ulp) instead of the 15-instruction correctly rounded sequence
                const float m0 = __builtin_amdgcn_sqrtf(e0), m1 = __builtin_amdgcn_sqrtf(e1);
                L.y[k] = m0; L.y[kk] = m1;
                lmax = fmaxf(lmax, fmaxf(m0, m1)); lmin = fminf(lmin, fminf(m0, m1)); ls0 += e0; ls1 += e1;
            }
        }
        const double lsum = (double)ls0 + (double)ls1;
        if (with_mag) {
            maxy = wave_max(lmax);
            miny = wave_min(lmin);
            tot = wave_sum(lsum);
        }
        wave_sync();
    };

    // per-peak pass over the staged frames [0, ng)
    int LPF = 1;
    while (LPF < K && LPF < 64) LPF <<= 1;
    const int fpp = 64 / LPF;
    const int G = (fpp < GF) ? fpp : GF;                          // frames staged per pass
    const int gl = lane / LPF, e0 = lane - gl * LPF;
    const unsigned long long gmask = (LPF == 64 ? ~0ull : ((1ull << LPF) - 1ull)) << (gl * LPF);
    auto flush = [&](int ng) {
        wave_sync();
        const int g = gl;
        const bool gvalid = g < ng;
        const int cnt = gvalid ? L.cnt[g] : -1;
        const int64_t orow = gvalid ? (int64_t)L.orow[g] : 0;
        double* of = p.f + orow * K;
        double* om = p.mag + orow * K;
        double* op = p.ph + orow * K;
        double* orp = p.realph + orow * K;
        double* ob = p.binno + orow * K;
        int nout = 0;
        for (int eb = 0; eb < K; eb += LPF) {
            const int e = eb + e0;
            bool valid = (cnt >= 0) && (e < cnt);
            int nbin = 0;
            PeakOut o;
            o.freq = 0.0; o.dfb = 0.0; o.thisph = 0.0; o.mag = 0.0; o.valid = false;
            if (valid) {
                nbin = L.sbin[g * kpad + e];
                const float* sv = L.sval + (size_t)(g * kpad + e) * 5;
                o = peak_math<float>(nbin, sv[0], sv[1], sv[2], sv[3], sv[4], pc);
                valid = o.valid;
            }
            const unsigned long long bal = __ballot(valid) & gmask;
            if (valid) {
                const int oi = nout + __popcll(bal & ((1ull << lane) - 1ull));
                ob[oi] = (double)nbin;
                of[oi] = o.freq;
                om[oi] = o.mag;
                op[oi] = o.thisph;
                orp[oi] = o.thisph + kPi * o.dfb / p.fstep;       // PV.py:207
            }
            nout += __popcll(bal);
        }
        if (cnt >= 0) {
            for (int j = nout + e0; j < K; j += LPF) {            // zero padding, PV.py:226-239
                ob[j] = 0.0; of[j] = 0.0; om[j] = 0.0; op[j] = 0.0; orp[j] = 0.0;
            }
            if (e0 == 0) {
                const int64_t fr = L.frm[g];
                if (p.totalmag) p.totalmag[orow] = sqrt(L.tot[g]);                                   // PV.py:210
                if (p.t) p.t[orow] = ((double)(fr * (int64_t)p.hop) + FN / 2.0) / p.sr;              // PV.py:247
            }
        }
        wave_sync();
    };

    // ---- previous spectrum of the first row
    int64_t gb, gq;                                               // (b, q) of the row being processed
    {
        const int64_t g0 = r0 - 1;                                // may be -1: treated as a zero row
        if (g0 >= 0) { gb = g0 / (p.F + 1); gq = g0 - gb * (p.F + 1); }      // the only division
        else { gb = -1; gq = p.F; }                                         // so that g0 + 1 = (0, 0)
        prefetch(g0, gb, gq);
        float d0, d1;
        double d2;
        spectrum(g0, gb, gq, prv, false, d0, d1, d2);
    }
    int ng = 0;
    for (int64_t g = r0; g < r1; ++g) {
        if (gq == p.F) { gq = 0; gb += 1; } else { gq += 1; }
        const int64_t b = gb, q = gq;
        float maxy = 0.f, miny = 0.f;
        double tot = 0.0;
        spectrum(g, b, q, cur, true, maxy, miny, tot);
        if (q != 0) {
            const int64_t orow = b * p.F + (q - 1);
            // PeakFinder(famp, npeaks, minrattomax) + filter_by_salience(rad=5)  (PV.py:175-178)
            const double minamp = (double)maxy * p.thr;           // PF.py:60
            const int nsel = peak_select<float, FM>(L.y, L.cs, L.ci, L.sel, FM, K, minamp, true, miny, lane);
            const bool use_prev0 = (p.prev0 != nullptr) && (orow == 0);
            int nk = 0;
            for (int eb = 0; eb < nsel; eb += 64) {
                const int e = eb + lane;
                int pb = 0;
                bool keep = false;
                if (e < nsel) { pb = L.sel[e]; keep = salient<float>(L.y, FM, pb, p.rad); }
                const unsigned long long bal = __ballot(keep);
                if (keep) {
                    const int slot = ng * kpad + nk + lane_prefix(bal);
                    const float2 c = cur[zpad(pb)];
                    float2 pv;
                    if (use_prev0) pv = make_float2((float)p.prev0[2 * pb], (float)p.prev0[2 * pb + 1]);
                    else pv = prv[zpad(pb)];
                    // PV.py:197-199: 3-bin energy, bin 0 excluded.  A selected bin is an interior
                    // local maximum, 1 <= pb <= 1022: pb+1 is always a bin, pb-1 counts unless it is 0
                    const float2 vm = cur[zpad(pb - 1)], vp = cur[zpad(pb + 1)];
                    const float em = (pb > 1) ? __builtin_fmaf(vm.x, vm.x, vm.y * vm.y) : 0.f;
                    const float s3 = (em + __builtin_fmaf(c.x, c.x, c.y * c.y)) + __builtin_fmaf(vp.x, vp.x, vp.y * vp.y);
                    L.sbin[slot] = pb;
                    float* sv = L.sval + (size_t)slot * 5;
                    sv[0] = c.x; sv[1] = c.y; sv[2] = pv.x; sv[3] = pv.y; sv[4] = s3;
                }
                nk += __popcll(bal);
            }
            if (lane == 0) { L.cnt[ng] = nk; L.frm[ng] = (int)(q - 1); L.orow[ng] = orow; L.tot[ng] = tot; }
            ng++;
            if (ng == G) { flush(ng); ng = 0; }
        }
        if (p.spec_out != nullptr && g == p.spec_row) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const float2 v = cur[zpad(lane + 64 * j)];
                p.spec_out[2 * (lane + 64 * j)] = v.x;
                p.spec_out[2 * (lane + 64 * j) + 1] = v.y;
            }
        }
        float2* t = cur; cur = prv; prv = t;
    }
    if (ng > 0) flush(ng);
}

}  // namespace

int pvx_fused_supported(int nfft, int precision, int K) {
    if (nfft != FN || precision != 32) return 0;
    return (kFusedShared + fused_lds_per_wave(K) * 2 <= 160 * 1024 / 1) ? 1 : 0;
}

int pvx_launch_fused(const FusedParams& p, int x_dtype, hipStream_t s) {
    if (p.total_rows <= 0) return PVX_OK;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    const size_t per_wave = fused_lds_per_wave(p.K);
    int waves = 2;
    if (kFusedShared + per_wave * waves > 160 * 1024) waves = 1;
    const size_t lds = kFusedShared + per_wave * waves;
    if (lds > 160 * 1024) { pvx_set_error("npks=%d needs %zu bytes of LDS in the fused kernel", p.K, lds); return PVX_ERR_UNSUPPORTED; }
    int blocks_per_cu = (int)((160 * 1024) / lds);
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    if (blocks_per_cu * waves > 8) blocks_per_cu = 8 / waves;
    int64_t nblocks = (int64_t)ncu * blocks_per_cu;
    if (p.blocks_override > 0) nblocks = p.blocks_override;
    // never more waves than rows (each wave needs at least one row to be worth its halo FFT)
    const int64_t min_rows_per_wave = 4;
    const int64_t maxb = (p.total_rows / min_rows_per_wave + waves - 1) / waves;
    if (nblocks > maxb) nblocks = maxb > 0 ? maxb : 1;
    const bool al2 = (x_dtype == PVX_F32) && (p.hop % 2 == 0) && (p.sig_stride % 2 == 0) && (((uintptr_t)p.x) % 8 == 0);
    dim3 grid((unsigned)nblocks), block(64 * waves);
#define PVX_FUSED_LAUNCH(IT, AL)                                                                              \
    do {                                                                                                      \
        if (lds > 64 * 1024)                                                                                  \
            PVX_HIP_CHECK(hipFuncSetAttribute((const void*)k_fused_pv2048<IT, AL>,                            \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));         \
        hipLaunchKernelGGL((k_fused_pv2048<IT, AL>), grid, block, lds, s, p);                                 \
    } while (0)
    switch (x_dtype) {
        case PVX_F32: if (al2) PVX_FUSED_LAUNCH(float, true); else PVX_FUSED_LAUNCH(float, false); break;
        case PVX_F64: PVX_FUSED_LAUNCH(double, false); break;
        case PVX_I16: PVX_FUSED_LAUNCH(int16_t, false); break;
        default: pvx_set_error("bad x_dtype %d", x_dtype); return PVX_ERR_INVALID;
    }
#undef PVX_FUSED_LAUNCH
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}
