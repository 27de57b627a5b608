// pvx_stft4.h -- pvx_fft4.h's four-quarter transform in plain type T (float64 has no packed instructions): a 1024-point
// complex transform on one wave as four 256-point ones (one per 16-lane group: radix-16 registers, transpose through LDS
// inside the group, radix-16 registers; no cross-lane stage), the remaining radix-4 done by the lane that untangles the
// bins.  Index maps, twiddles and the k1 = 0 / k1 = 128 special case: pvx_fft4.h and tools/models/fft4_model.py.
//   PV.calc_fft_frame   pypevoc/PVAnalysis.py:150-158
#pragma once

#include "pvx_fft4.h"
#include "pvx_stft.h"

namespace pvxs {

template <typename T> __device__ __forceinline__ cx<T> cmulcT(cx<T> z, cx<T> w) {            // z * conj(w)
    return mkc<T>(fmaT(z.x, w.x, z.y * w.y), fmaT(z.x, -w.y, z.y * w.x));
}
template <typename T> __device__ __forceinline__ void dft4T(const cx<T> (&a)[4], cx<T> (&A)[4]) {
    const cx<T> e = a[0] + a[2], f = a[0] - a[2], g = a[1] + a[3], h = a[1] - a[3];
    A[0] = e + g; A[2] = e - g; A[1] = addmni(f, h); A[3] = addpi(f, h);
}
// (Za, Zb = Z[M-k], twiddle w = W_N^k) -> X[k], X[M-k]   (k_stft.hip's untangle: S = Za + conj Zb, D = Za - conj Zb,
// O = -i D / 2, P = O w; X[k] = S/2 + P, X[M-k] = conj(S/2 - P))
template <typename T> __device__ __forceinline__ void untangleT(cx<T> za, cx<T> zb, cx<T> w, cx<T>& x0, cx<T>& x1) {
    const cx<T> S = mkc<T>(za.x + zb.x, za.y - zb.y);
    const cx<T> D = mkc<T>(za.x - zb.x, za.y + zb.y);
    const cx<T> O = mkc<T>((T)0.5 * D.y, (T)-0.5 * D.x);
    const cx<T> Pk = cmulT(O, w);
    x0 = mkc<T>(fmaT((T)0.5, S.x, Pk.x), fmaT((T)0.5, S.y, Pk.y));
    x1 = mkc<T>(fmaT((T)0.5, S.x, -Pk.x), -fmaT((T)0.5, S.y, -Pk.y));
}

// stages 1 and 2 of the four quarters and the natural-order store (quarter u at u RP); t1: LDS table [16][16] W_256^(l q)
// TB: twiddles in flight at a time (16: all; 8 where the caller has no registers to spare)
template <typename T, int TB = 16, typename H1, typename H2, typename H3>
__device__ __forceinline__ void fft4_quartersT(cx<T> (&z)[16], cx<T>* dz, const cx<T>* t1, int lane, H1 hook1, H2 hook2, H3 hook3) {
    dftT<16, T>(z);                                                  // stage 1: radix-16 over r
    __builtin_amdgcn_sched_barrier(0);
    hook1();
    const int l = lane & 15, u = lane >> 4;
    cx<T>* const ew = dz + u * F4::EU + l;
    {
        // the lane's fifteen twiddles first, all in flight (a twiddle at a time between the stores, every one is an LDS round
        // trip behind the store before it)
#pragma unroll
        for (int q0 = 0; q0 < 16; q0 += TB) {
            cx<T> tw[TB];
#pragma unroll
            for (int q = (q0 == 0 ? 1 : 0); q < TB; q++) tw[q] = t1[(q0 + q) * 16 + l];
#pragma unroll
            for (int q = (q0 == 0 ? 1 : 0); q < TB; q++) asm volatile("" : "+v"(tw[q].x), "+v"(tw[q].y));
#pragma unroll
            for (int q = 0; q < TB; q++) ew[(q0 + q) * F4::EP] = (q0 + q > 0) ? cmulT(z[q0 + q], tw[q]) : z[q0 + q];
        }
    }
    wave_sync();
    const cx<T>* const er = dz + u * F4::EU + l * F4::EP;
#pragma unroll
    for (int l2 = 0; l2 < 16; l2++) z[l2] = er[l2];
    hook2();
    wave_sync();
    dftT<16, T>(z);                                                  // stage 2: radix-16 over l
    __builtin_amdgcn_sched_barrier(0);
    hook3();
    cx<T>* const nw = dz + u * F4::RP + l;
#pragma unroll
    for (int t = 0; t < 16; t++) nw[16 * t] = z[t];                  // E_u[l + 16 t]
}

// The values of the join: a[j][u] = E_u[k1], b[j][u] = E_u[256 - k1] (k1 = lane + 64 j), c[u] = E_u[128] -- read first, so
// that the caller can reuse the buffer (the |X|^2 row goes there) before the arithmetic
// (the k1 = 128 family is reduced to its four bins `spv` right away: 4 values live instead of the 4 it came from + temporaries)
template <typename T> struct Join4In { cx<T> a[2][4], b[2][4], spv[4]; };
template <typename T> __device__ __forceinline__ void join4_read(const cx<T>* dz, int lane, Join4In<T>& in) {
    constexpr T H = (T)0.70710678118654752440, C1 = (T)0.92387953251128673848, S1 = (T)0.38268343236508978178;
    {
        cx<T> c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) c[u] = dz[u * F4::RP + 128];
        const cx<T> c1 = cmulT(c[1], mkc<T>(H, -H)), c2 = mniT(c[2]), c3 = cmulT(c[3], mkc<T>(-H, -H));
        const cx<T> A = c[0] + c2, B = c[0] - c2, C = c1 + c3, D = c1 - c3;
        const cx<T> z0 = A + C, z2 = A - C, z1 = addmni(B, D), z3 = addpi(B, D);      // Z[128 + 256 u]
        untangleT(z0, z3, mkc<T>(C1, -S1), in.spv[0], in.spv[3]);                     // W_N^128 = W_16
        untangleT(z1, z2, mkc<T>(S1, -C1), in.spv[1], in.spv[2]);                     // W_N^(128 + 256) = W_16^3
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int k1 = lane + 64 * j, kb = (256 - k1) & 255;
#pragma unroll
        for (int u = 0; u < 4; u++) { in.a[j][u] = dz[u * F4::RP + k1]; in.b[j][u] = dz[u * F4::RP + kb]; }
    }
}

// Radix-4 join fused with the untangle: emit(bin, X[bin]) for the lane's 16 bins -- X[k1 + 256 t] and
// X[(256 - k1) + 256 (3 - t)] of k1 = lane + 64 j, lane 0's mirrored slots of j = 0 taking the k1 = 128 family.
// tw: LDS table [2][4][64]: entry (j 4 + u) 64 + lane = W_N^k1 (u = 0), W_M^(u k1) (M = 1024, N = 2048); read set by set.
template <typename T, typename EMIT>
__device__ __forceinline__ void join4_emit(Join4In<T>& in, const cx<T>* tw, int lane, EMIT emit) {
    constexpr T H = (T)0.70710678118654752440;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int k1 = lane + 64 * j;
        const int kb = (256 - k1) & 255;
#pragma unroll
        for (int u = 1; u < 4; u++) {
            const cx<T> w = tw[(j * 4 + u) * 64 + lane];
            in.a[j][u] = cmulT(in.a[j][u], w);
            in.b[j][u] = cmulcT(in.b[j][u], w);
        }
        cx<T> A[4], B[4], x0[4], x1[4];
        dft4T(in.a[j], A);
        dft4T(in.b[j], B);
        // pairs (A_t, B_((4 - t) mod 4)), untangle twiddle W_8^t W_N^k1
        const cx<T> wu = tw[(j * 4) * 64 + lane];
        untangleT(A[0], B[0], wu, x0[0], x1[0]);
        untangleT(A[1], B[3], cmulT(wu, mkc<T>(H, -H)), x0[1], x1[1]);
        untangleT(A[2], B[2], mniT(wu), x0[2], x1[2]);
        untangleT(A[3], B[1], cmulT(wu, mkc<T>(-H, -H)), x0[3], x1[3]);
        int kbb = kb;
        if (j == 0) {
            if (lane == 0) {
#pragma unroll
                for (int t = 0; t < 4; t++) x1[t] = in.spv[3 - t];
                kbb = 128;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            emit(k1 + 256 * t, x0[t]);
            emit(kbb + 256 * (3 - t), x1[t]);
        }
    }
}

}  // namespace pvxs
