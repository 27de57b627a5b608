// pvx_fft.h -- in-register / cross-lane FFT building blocks shared by the fused analysis kernels
// (k_fused.hip: one wave per frame; k_fused_mw.hip: several waves per frame).
#pragma once

#include "pvx_wave.h"
#include "pvx_cplx.h"

namespace pvxf {

// frames staged before the per-peak pass: as many as fill the 64 lanes with (frame, peak) pairs
__host__ __device__ inline int staged_frames(int K, int gmax) {
    int lpf = 1;
    while (lpf < K && lpf < 64) lpf <<= 1;
    const int fpp = 64 / lpf;
    return fpp < gmax ? fpp : gmax;
}

// W_64^k = (kW64r[k], kW64i[k])
constexpr float kW64r[64] = {1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f, 0.000000000e+00f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f, -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f, -9.951847267e-01f, -1.000000000e+00f, -9.951847267e-01f, -9.807852804e-01f, -9.569403357e-01f, -9.238795325e-01f, -8.819212643e-01f, -8.314696123e-01f, -7.730104534e-01f, -7.071067812e-01f, -6.343932842e-01f, -5.555702330e-01f, -4.713967368e-01f, -3.826834324e-01f, -2.902846773e-01f, -1.950903220e-01f, -9.801714033e-02f, 0.000000000e+00f, 9.801714033e-02f, 1.950903220e-01f, 2.902846773e-01f, 3.826834324e-01f, 4.713967368e-01f, 5.555702330e-01f, 6.343932842e-01f, 7.071067812e-01f, 7.730104534e-01f, 8.314696123e-01f, 8.819212643e-01f, 9.238795325e-01f, 9.569403357e-01f, 9.807852804e-01f, 9.951847267e-01f};
constexpr float kW64i[64] = {0.000000000e+00f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f, -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f, -9.951847267e-01f, -1.000000000e+00f, -9.951847267e-01f, -9.807852804e-01f, -9.569403357e-01f, -9.238795325e-01f, -8.819212643e-01f, -8.314696123e-01f, -7.730104534e-01f, -7.071067812e-01f, -6.343932842e-01f, -5.555702330e-01f, -4.713967368e-01f, -3.826834324e-01f, -2.902846773e-01f, -1.950903220e-01f, -9.801714033e-02f, 0.000000000e+00f, 9.801714033e-02f, 1.950903220e-01f, 2.902846773e-01f, 3.826834324e-01f, 4.713967368e-01f, 5.555702330e-01f, 6.343932842e-01f, 7.071067812e-01f, 7.730104534e-01f, 8.314696123e-01f, 8.819212643e-01f, 9.238795325e-01f, 9.569403357e-01f, 9.807852804e-01f, 9.951847267e-01f, 1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f};

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }
constexpr int bitrev_c(int v, int bits) {
    int r = 0;
    for (int b = 0; b < bits; b++) if (v & (1 << b)) r |= 1 << (bits - 1 - b);
    return r;
}

// complex multiply with explicit fused multiply-adds: the same rounding wherever it is inlined, so
// results do not depend on which wave computes a frame
__device__ __forceinline__ void cmul(float& a, float& c, float wr, float wi) {
    const float nr = __builtin_fmaf(a, wr, -(c * wi));
    c = __builtin_fmaf(a, wi, c * wr);
    a = nr;
}

// full-wave DPP move: every lane is written (row/bank masks 0xf, the permutations used here have no
// invalid source lane), so there is no "old" value to preserve and none has to be materialised
template <int CTRL> __device__ __forceinline__ float dppf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// value of lane (l ^ H) for H in {1, 2, 4, 8}
template <int H> __device__ __forceinline__ float lane_xor(float v) {
    if constexpr (H == 1) return dppf<0xB1>(v);                       // quad_perm [1,0,3,2]
    else if constexpr (H == 2) return dppf<0x4E>(v);                  // quad_perm [2,3,0,1]
    else return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (H << 10) | 0x1f));
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

// ---- radix-R DFT in registers for any power of two R <= 64: unrolled radix-2 decimation in
// frequency with compile-time twiddles, natural order in and out (the bit reversal is a renaming)
template <int R> __device__ __forceinline__ void dft_regs(float (&xr)[R], float (&xi)[R]) {
    if constexpr (R == 16) {
        dft16(xr, xi);
    } else {
#pragma unroll
        for (int h = R / 2; h >= 1; h >>= 1) {
#pragma unroll
            for (int blk = 0; blk < R; blk += 2 * h) {
#pragma unroll
                for (int i = 0; i < h; i++) {
                    const int a = blk + i, b = blk + i + h;
                    const float sr = xr[a] + xr[b], si = xi[a] + xi[b];
                    float dr = xr[a] - xr[b], di = xi[a] - xi[b];
                    const int tw = i * (32 / h);                    // W_2h^i = W_64^(i * 64/(2h))
                    if (tw == 0) { }
                    else if (tw == 16) { const float t = dr; dr = di; di = -t; }      // * (-i)
                    else cmul(dr, di, kW64r[tw], kW64i[tw]);
                    xr[a] = sr; xi[a] = si; xr[b] = dr; xi[b] = di;
                }
            }
        }
        constexpr int bits = ilog2(R);
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int j = bitrev_c(i, bits);
            if (i < j) { float t = xr[i]; xr[i] = xr[j]; xr[j] = t; t = xi[i]; xi[i] = xi[j]; xi[j] = t; }
        }
    }
}

// ======== the same DFTs on complex register pairs (pvx_cplx.h): one packed instruction per complex
// add, two per complex multiply, multiply-by-(-i) folded into the adds' operand modifiers.  Same
// arithmetic in the same order as the scalar versions above (bit-identical results).
using pvxc::v2f;

template <int CTRL> __device__ __forceinline__ v2f dpp2(v2f v) { return pvxc::mk(dppf<CTRL>(v.x), dppf<CTRL>(v.y)); }
template <int H> __device__ __forceinline__ v2f lane_xor2(v2f v) { return pvxc::mk(lane_xor<H>(v.x), lane_xor<H>(v.y)); }

__device__ __forceinline__ void dft16(v2f (&x)[16]) {
    using namespace pvxc;
    constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f, H = 0.70710678118654752440f;
    constexpr float cr[10] = {1.f, C1, H, S1, 0.f, 0.f, -H, 0.f, 0.f, -C1};
    constexpr float ci[10] = {0.f, -S1, -H, -C1, -1.f, 0.f, -H, 0.f, 0.f, S1};
    v2f t[16];
#pragma unroll
    for (int n1 = 0; n1 < 4; n1++) {
        const v2f a = x[n1], b = x[n1 + 4], c = x[n1 + 8], d = x[n1 + 12];
        const v2f A = a + c, B = a - c, C = b + d, D = b - d;
        v2f y[4];
        y[0] = A + C; y[2] = A - C;
        y[1] = add_mni(B, D);                    // B - i D
        y[3] = add_pi(B, D);                     // B + i D
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) {
            const int m = n1 * k2;
            if (m == 0) t[n1 * 4 + k2] = y[k2];
            else if (m == 4) t[n1 * 4 + k2] = mni(y[k2]);
            else t[n1 * 4 + k2] = cmul_k(y[k2], mk(cr[m], ci[m]));
        }
    }
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) {
        const v2f a = t[k2], b = t[4 + k2], c = t[8 + k2], d = t[12 + k2];
        const v2f A = a + c, B = a - c, C = b + d, D = b - d;
        x[k2] = A + C; x[k2 + 8] = A - C;
        x[k2 + 4] = add_mni(B, D); x[k2 + 12] = add_pi(B, D);
    }
}

template <int R> __device__ __forceinline__ void dft_regs(v2f (&x)[R]) {
    using namespace pvxc;
    if constexpr (R == 16) {
        dft16(x);
    } else {
#pragma unroll
        for (int h = R / 2; h >= 1; h >>= 1) {
#pragma unroll
            for (int blk = 0; blk < R; blk += 2 * h) {
#pragma unroll
                for (int i = 0; i < h; i++) {
                    const int a = blk + i, b = blk + i + h;
                    const v2f s = x[a] + x[b];
                    v2f d = x[a] - x[b];
                    const int tw = i * (32 / h);
                    if (tw == 0) { }
                    else if (tw == 16) d = mni(d);
                    else d = cmul_k(d, mk(kW64r[tw], kW64i[tw]));
                    x[a] = s; x[b] = d;
                }
            }
        }
        constexpr int bits = ilog2(R);
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int j = bitrev_c(i, bits);
            if (i < j) { const v2f t = x[i]; x[i] = x[j]; x[j] = t; }
        }
    }
}

// One decimation-in-frequency step of the cross-lane DFT on four independent values at once: exchange
// with lane ^ H, a' = sg * a + partner (sg = -1 in the upper lane), then the lane's twiddle.  Written
// phase by phase over the four values so that the DPP exchanges (which need wait states after the VALU
// write of their source) and the dependent multiply-adds of different values interleave.
template <int H, bool TW> __device__ __forceinline__ void xstep4(v2f (&a)[4], float sg, v2f w) {
    v2f q[4];
#pragma unroll
    for (int j = 0; j < 4; j++) q[j] = lane_xor2<H>(a[j]);
#pragma unroll
    for (int j = 0; j < 4; j++) a[j] = pvxc::fma_s(sg, a[j], q[j]);
    if constexpr (TW) {
#pragma unroll
        for (int j = 0; j < 4; j++) a[j] = pvxc::cmul(a[j], w);
    }
}

template <typename InT> __device__ __forceinline__ float ld1(const InT* p) { return (float)*p; }

// ======== one wave per frame (k_fused.hip, k_fused_ring.hip): geometry and the sample loads
// ---- compile-time geometry for R registers per lane
template <int R> struct Geo {
    static constexpr int M = 64 * R;                 // complex FFT length = bins 0..M-1
    static constexpr int N = 128 * R;                // nfft
    static constexpr int P = 64 / R;                 // lanes per cross-lane DFT
    static constexpr int LOGP = ilog2(P);
    static constexpr int LOGR = ilog2(R);
    static constexpr int PITCH = 64 + P;             // exchange row pitch (complex)
    static constexpr int R2 = R * R;
    static constexpr int ZP = ((32 / P) - (R2 % 32) + 32) % 32;     // padding per R^2 spectrum bins
    static constexpr int ZLEN = M + ZP * (P - 1);
    static constexpr int BUFRAW = (R * PITCH > ZLEN) ? R * PITCH : ZLEN;
    static constexpr int BUFC = ((BUFRAW + 63) / 64) * 64;           // complex slots per spectrum buffer
    static constexpr int CAP = M / 2 + 4;            // candidate list capacity
    static constexpr int HALF = M / 2;
};
template <int R> __device__ __host__ __forceinline__ int zpad(int k) { return k + Geo<R>::ZP * (k >> (2 * Geo<R>::LOGR)); }

template <int R, typename InT, bool AL2>
__device__ __forceinline__ void load_raw(const InT* x, int lane, v2f (&raw)[R]) {
    // lane l takes z[l + 64 r] = (x[2l + 128 r], x[2l + 128 r + 1]): 512 contiguous bytes per
    // wave-instruction.  Issued one frame ahead of its use (software prefetch): the loads of row
    // g+1 are in flight while row g is transformed and searched for peaks.
#pragma unroll
    for (int r = 0; r < R; r++) {
        const InT* p = x + 2 * lane + 128 * r;
        if constexpr (AL2 && sizeof(InT) == 4) {
            raw[r] = *(const v2f*)p;
        } else {
            raw[r] = pvxc::mk(ld1(p), ld1(p + 1));
        }
    }
}



}  // namespace pvxf
