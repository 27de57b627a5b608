// pvx_stft.h -- the float64 / float32 single-wave STFT front end shared by k_stft.hip (spectrum rows only) and
// k_stft_pv.hip (spectrum rows + peaks): complex helpers in plain type T, the radix-R register DFT, the cross-lane
// DFT step, the LDS geometry and the kernel parameters.  Factorisation and index maps: see k_stft.hip / k_fused.hip.
#pragma once
#include "pvx_fft.h"

namespace pvxs {
using namespace pvxw;
using namespace pvxf;

template <typename T> struct cx { T x, y; };

// A spectrum row's store (k_stft.hip, k_stft_pv.hip): non-temporal.  The rows stream out -- of a row's nfft/2 bins the peak pass reads
// back the few kept ones, soon, from the L2 it went through -- and as ordinary stores they pushed everything else out of the caches on
// their way to HBM: float64 at nfft 1024 282 -> 324 M frames/s on the harmonic signal, nfft 2048 168 -> 176 (the recording 146 -> 157),
// nfft 512 512 -> 534 (profiles/r05_ab_steps.txt; PVX_ROWS_TEMPORAL=1 at build time: the plain store).
template <typename T> __device__ __forceinline__ void row_store(cx<T>* q, cx<T> v) {
#ifdef PVX_ROWS_TEMPORAL
    *q = v;
#else
    typedef T v2t __attribute__((ext_vector_type(2)));
    v2t w; w.x = v.x; w.y = v.y;
    __builtin_nontemporal_store(w, (v2t*)q);
#endif
}
template <typename T> __device__ __forceinline__ cx<T> mkc(T a, T b) { cx<T> r; r.x = a; r.y = b; return r; }
template <typename T> __device__ __forceinline__ cx<T> operator+(cx<T> a, cx<T> b) { return mkc<T>(a.x + b.x, a.y + b.y); }
template <typename T> __device__ __forceinline__ cx<T> operator-(cx<T> a, cx<T> b) { return mkc<T>(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double fmaT(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fmaT(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// z * w with explicit fused multiply-adds (the rounding of pvxf::cmul)
template <typename T> __device__ __forceinline__ cx<T> cmulT(cx<T> z, cx<T> w) {
    return mkc<T>(fmaT(z.x, w.x, -(z.y * w.y)), fmaT(z.x, w.y, z.y * w.x));
}
template <typename T> __device__ __forceinline__ cx<T> mniT(cx<T> z) { return mkc<T>(z.y, -z.x); }          // -i z
template <typename T> __device__ __forceinline__ cx<T> addmni(cx<T> b, cx<T> d) { return mkc<T>(b.x + d.y, b.y - d.x); }   // b - i d
template <typename T> __device__ __forceinline__ cx<T> addpi(cx<T> b, cx<T> d) { return mkc<T>(b.x - d.y, b.y + d.x); }    // b + i d

// radix-R DFT in registers, natural order in and out (pvx_fft.h's dft16 / radix-2 network in type T)
template <int R, typename T> __device__ __forceinline__ void dftT(cx<T> (&x)[R]) {
    constexpr T C1 = (T)0.92387953251128673848, S1 = (T)0.38268343236508978178, H = (T)0.70710678118654752440;
    if constexpr (R == 16) {
        const T cr[10] = {(T)1, C1, H, S1, (T)0, (T)0, -H, (T)0, (T)0, -C1};
        const T ci[10] = {(T)0, -S1, -H, -C1, (T)-1, (T)0, -H, (T)0, (T)0, S1};
        cx<T> t[16];
#pragma unroll
        for (int n1 = 0; n1 < 4; n1++) {
            const cx<T> a = x[n1], b = x[n1 + 4], c = x[n1 + 8], d = x[n1 + 12];
            const cx<T> A = a + c, B = a - c, C = b + d, D = b - d;
            cx<T> y[4];
            y[0] = A + C; y[2] = A - C; y[1] = addmni(B, D); y[3] = addpi(B, D);
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) {
                const int m = n1 * k2;
                if (m == 0) t[n1 * 4 + k2] = y[k2];
                else if (m == 4) t[n1 * 4 + k2] = mniT(y[k2]);
                else t[n1 * 4 + k2] = cmulT(y[k2], mkc<T>(cr[m], ci[m]));
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) {
            const cx<T> a = t[k2], b = t[4 + k2], c = t[8 + k2], d = t[12 + k2];
            const cx<T> A = a + c, B = a - c, C = b + d, D = b - d;
            x[k2] = A + C; x[k2 + 8] = A - C; x[k2 + 4] = addmni(B, D); x[k2 + 12] = addpi(B, D);
        }
    } else {
        // radix-2 decimation in frequency, R in {4, 8}: the only non-trivial twiddles are W_8^1 and W_8^3
#pragma unroll
        for (int h = R / 2; h >= 1; h >>= 1) {
#pragma unroll
            for (int blk = 0; blk < R; blk += 2 * h) {
#pragma unroll
                for (int i = 0; i < h; i++) {
                    const int a = blk + i, b = blk + i + h;
                    const cx<T> s = x[a] + x[b];
                    cx<T> d = x[a] - x[b];
                    const int tw = i * (8 / (2 * h)) % 8;              // W_2h^i = W_8^(i * 8/(2h)), 2h in {2, 4, 8}
                    if (tw == 1) d = cmulT(d, mkc<T>(H, -H));
                    else if (tw == 2) d = mniT(d);
                    else if (tw == 3) d = cmulT(d, mkc<T>(-H, -H));
                    x[a] = s; x[b] = d;
                }
            }
        }
        constexpr int bits = ilog2(R);
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int j = bitrev_c(i, bits);
            if (i < j) { const cx<T> t = x[i]; x[i] = x[j]; x[j] = t; }
        }
    }
}

// value of lane (l ^ H)
template <int H> __device__ __forceinline__ double lane_xorT(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const float lo = lane_xor<H>(__builtin_bit_cast(float, (int)b));
    const float hi = lane_xor<H>(__builtin_bit_cast(float, (int)(b >> 32)));
    return __builtin_bit_cast(double, ((long long)__builtin_bit_cast(int, hi) << 32) | (unsigned)__builtin_bit_cast(int, lo));
}
template <int H> __device__ __forceinline__ float lane_xorT(float v) { return lane_xor<H>(v); }

// one decimation-in-frequency step of the cross-lane DFT: a' = sg a + partner, then the lane's twiddle
template <int H, bool TW, typename T> __device__ __forceinline__ cx<T> xstepT(cx<T> a, T sg, cx<T> w) {
    const cx<T> q = mkc<T>(lane_xorT<H>(a.x), lane_xorT<H>(a.y));
    cx<T> r = mkc<T>(fmaT(sg, a.x, q.x), fmaT(sg, a.y, q.y));
    if constexpr (TW) r = cmulT(r, w);
    return r;
}

template <int R, typename T> struct StftGeo {
    static constexpr int M = 64 * R, N = 128 * R, P = 64 / R, LOGP = ilog2(P), LOGR = ilog2(R), R2 = R * R, HALF = M / 2;
    static constexpr int PITCH = 64 + P;                              // exchange row pitch (complex)
    // padding of the natural-order spectrum per R^2 bins so that the P lanes of a group (which write bins R^2 apart)
    // land in different banks: 16-byte elements -> 16 bank groups
    static constexpr int EPB = 64 / (int)(2 * sizeof(T) / 4);        // complex elements per 256-byte bank sweep
    static constexpr int ZP = (EPB / P > 0 ? EPB / P : 1);
    static constexpr int ZLEN = M + ZP * (P - 1);
    static constexpr int BUFRAW = (R * PITCH > ZLEN) ? R * PITCH : ZLEN;
    static constexpr int BUFC = ((BUFRAW + 63) / 64) * 64;
    static constexpr int TW3N = (HALF + 8) & ~7;
    static constexpr size_t OFF_WIN = 0;                                                     // T [N]  window / wfact
    static constexpr size_t OFF_T1 = OFF_WIN + (size_t)N * sizeof(T);                        // cx [R][64]  W_M^(l q)
    static constexpr size_t OFF_T2 = OFF_T1 + (size_t)R * 64 * 2 * sizeof(T);                // cx [R][P]   W_64^(l1 t2)
    static constexpr size_t OFF_TW3 = OFF_T2 + 64 * 2 * sizeof(T);                           // cx [TW3N]   W_nfft^k
    static constexpr size_t OFF_BUF = OFF_TW3 + (size_t)TW3N * 2 * sizeof(T);                // cx [NW][BUFC]
    __host__ __device__ static size_t total(int nw) { return OFF_BUF + (size_t)nw * BUFC * 2 * sizeof(T); }
};
// n strided LDS table entries of a lane, ALL in flight before the first use: read one at a time beside the stores or the
// arithmetic that use them -- what the compiler makes of the plain loops -- every entry is an LDS round trip of its own
// (measured on k_fused_rev / k_stft_pv: the longest stalls of the transform)
template <int N0, int N1, typename T> __device__ __forceinline__ void lds_gather(cx<T> (&dst)[N1], const cx<T>* base, int stride) {
#pragma unroll
    for (int q = N0; q < N1; q++) dst[q] = base[q * stride];
#pragma unroll
    for (int q = N0; q < N1; q++) asm volatile("" : "+v"(dst[q].x), "+v"(dst[q].y));
}

// ... in batches of B entries where the registers do not hold them all: use(q, entry) for q in [N0, N1)
template <int N0, int N1, int B, typename T, typename F> __device__ __forceinline__ void lds_gather_use(const cx<T>* base, int stride, F use) {
#pragma unroll
    for (int q0 = N0; q0 < N1; q0 += B) {
        cx<T> t[B];
#pragma unroll
        for (int i = 0; i < B; i++) if (q0 + i < N1) t[i] = base[(q0 + i) * stride];
#pragma unroll
        for (int i = 0; i < B; i++) if (q0 + i < N1) asm volatile("" : "+v"(t[i].x), "+v"(t[i].y));
#pragma unroll
        for (int i = 0; i < B; i++) if (q0 + i < N1) use(q0 + i, t[i]);
    }
}
// entries in flight at a time: all of them, except where 16 float64 values per lane already fill the registers
template <int R, typename T> constexpr int lds_batch() { return (sizeof(T) == 8 && R == 16) ? 4 : R; }

template <int R, typename T> __device__ __forceinline__ int zpadT(int k) { return k + StftGeo<R, T>::ZP * (k >> (2 * StftGeo<R, T>::LOGR)); }

struct StftParams {
    const void* x;            // input samples
    int64_t nsamp, sig_stride, F, R0, ws_rows, total_rows;
    int hop;
    const void* win;          // T [nfft]  window / wfact
    const void* twiddle;      // cx<T> [nfft]  W_nfft^j
    void* spec;               // cx<T> [ws_rows][ldo]
    int64_t ldo;
    // optional (k_stft_split): the row's candidate peaks for k_phase_peaks, which then does not stream the row again --
    // cand_bin / cand_y [ws_rows][cand_cap]: bins (ascending) and |X|^2 of the interior local maxima above the threshold
    // (PF.py:60, 69-70, 166-174 with pkthresh = cand_thr); cand_stats [ws_rows][4]: max |X|^2, min |X|^2, sum |X|^2, count
    void* cand_y = nullptr;
    unsigned short* cand_bin = nullptr;
    double* cand_stats = nullptr;
    int cand_cap = 0;
    double cand_thr = 0.0;
};

}  // namespace pvxs
