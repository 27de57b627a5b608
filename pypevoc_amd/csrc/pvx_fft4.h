// pvx_fft4.h -- a 1024-point complex transform on ONE wave64 as four 256-point ones, joined inside the untangle pass
// (k_fused_rev.hip at nfft 2048, k_fused_team.hip at nfft 4096 / 8192).
//
//   PV.calc_fft_frame   pypevoc/PVAnalysis.py:150-158   (np.fft.fft of the windowed frame; bins 0 .. nfft/2 - 1 are used)
//
// z[j], j < 1024 (z[j] = (x w)[2j] + i (x w)[2j+1]).  Decimation in time by 4: lane = 16 u + l (u < 4, l < 16), register
// r holds z[4 (l + 16 r) + u] -- the lane's float2 sits at sample offset 8 l + 2 u + 128 r, so one load instruction of the
// wave still covers 512 contiguous bytes.  A 16-lane group then has a 256-point transform to itself and 256 = 16 x 16
// fits it exactly: radix-16 over r in registers, twiddle W_256^(l q), transpose through LDS inside the group, radix-16
// over l in registers -> E_u[q + 16 t] in lane (u, q), register t.  There is NO cross-lane butterfly stage (the 16 x 16
// x 4 form of k_fused.hip spends 64 DPP moves, 48 packed operations and ~40 hazard no-ops per frame on its 4-lane
// stage): the last radix-4 -- Z[k1 + 256 t] = sum_u W_4^(u t) W_1024^(u k1) E_u[k1] -- is done by the lane that
// untangles those bins anyway.  With A = DFT_4(E_u[k1] W_1024^(u k1)) and B = DFT_4(E_u[256 - k1] conj W_1024^(u k1)),
// (A_t, B_((4 - t) mod 4)) are the untangle pairs (Z[k], Z[M - k]) of k = k1 + 256 t: a lane reads the 8 values of k1
// and 256 - k1 and writes the 8 bins X[k1 + 256 t], X[(256 - k1) + 256 (3 - t)] back to the same 8 slots.  k1 = 0
// pairs with itself; the mirrored slots of its lane take the k1 = 128 family (bins 128 + 256 u), which pairs within
// itself.  Index maps and twiddles: tools/models/fft4_model.py (checked against numpy.fft).
// The same pass joins the 1024-point sub-transforms of a team of 4 waves (k_fused_team.hip, nfft 8192): quarter =
// a wave's region, index inside a quarter through `IA`.
#pragma once
#ifndef PVX_THIN_FROM
#define PVX_THIN_FROM 96    // candidates per row (segment) from which the list is thinned before it is written (peak_scan_block_thin)
#endif

#include "pvx_fft.h"

namespace pvxf {

struct F4 {                                  // LDS layout of a wave's buffer (complex slots)
    static constexpr int EP = 17;            // exchange matrix: pitch of a q row inside a 16-lane group
    static constexpr int EU = 272;           // exchange matrix: pitch of a group
    static constexpr int RP = 272;           // natural order: pitch of a 256-bin quarter (16 slots of padding: the four
                                             // groups of a store / load instruction fall on different banks)
    static constexpr int BUF = 1088;         // slots per wave
};
// bin k of a buffer in natural order; also valid across the regions of a team (4 RP = BUF)
__device__ __host__ __forceinline__ int xa4(int k) { return (k >> 8) * F4::RP + (k & 255); }
// sample offset (floats) of the lane's first pair
__device__ __forceinline__ int lofs4(int lane) { return 8 * (lane & 15) + 2 * (lane >> 4); }

struct IdentityIA { __device__ __forceinline__ int operator()(int i) const { return i; } };
struct Xa4IA { __device__ __forceinline__ int operator()(int i) const { return xa4(i); } };

// |X|^2 of bin k from a wave's spectrum in the natural-order quarter layout -- the expression join4_untangle uses for the
// row's maximum / minimum / energy, on the very values it stored: bit-identical to the |X|^2 row the other kernels keep.
// (k_fused_rev at nfft 2048 keeps no such row: 4.3 KB per wave, the difference between two and three waves per SIMD.)
// (as two plain instructions: left to itself the compiler pairs the bins of a 16-byte read into packed multiplies -- x^2 it
// does not need, two moves to line the operands up, and packed float32 issues at half rate: 8 issue slots per two bins
// instead of 4)
__device__ __forceinline__ float norm2(float2 v) {
    float t, e;
    asm("v_mul_f32 %0, %1, %1" : "=v"(t) : "v"(v.y));
    asm("v_fma_f32 %0, %1, %1, %2" : "=v"(e) : "v"(v.x), "v"(t));
    return e;
}
struct YofX4 {
    const float2* x;
    __device__ __forceinline__ float operator[](int k) const { return norm2(x[xa4(k)]); }
};

// peak_scan_block_thin (pvx_wave.h) for a 1024-bin row that exists only as the spectrum X (quarter layout): the candidate
// list (ascending bins) of the row's interior maxima above the threshold, thinned when there are more than 192 of them.
// Lane l owns FOUR consecutive bins of EVERY quarter, 256 j + 4 l + i: its 32 contiguous bytes per quarter are two 16-byte
// reads on a 32-byte lane stride (16 consecutive bins per lane would put every lane of an instruction on the same banks:
// their 128-byte blocks are half the bank array apart).  Neighbours across lanes come by DPP wave shifts (lane 0 / 63: the
// value from the quarter before / after, through a broadcast).  List order = quarter, lane, bin: two count bits per
// quarter (four consecutive bins hold at most two maxima).
// It is also where the row's maximum, minimum and energy are taken (maxe, miny, tot: every |X|^2 passes through here, the
// join forms none) and the threshold th they set (PF.py:60, 69-70; thr = PeakFinder's minrattomax).
template <typename CI, int MAXK = 16>
__device__ __forceinline__ int peak_scan_x4_thin(const float2* X, double thr, float& maxe, float& miny, double& tot, double& th,
                                                 CI* ci, int trash, int lane, int npeaks) {
    float v[16];                                                     // v[4 j + i] = |X[256 j + 4 lane + i]|^2
    {
        // (all eight reads in flight before the first |X|^2: a quarter at a time they are four round trips in a row)
        pvxc::v4f xa[4], xb[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            xa[j] = *(const pvxc::v4f*)(X + j * F4::RP + 4 * lane);
            xb[j] = *(const pvxc::v4f*)(X + j * F4::RP + 4 * lane + 2);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) asm volatile("" : "+v"(xa[j]), "+v"(xb[j]));
#pragma unroll
        for (int j = 0; j < 4; j++) {
            v[4 * j] = norm2(make_float2(xa[j].x, xa[j].y)); v[4 * j + 1] = norm2(make_float2(xa[j].z, xa[j].w));
            v[4 * j + 2] = norm2(make_float2(xb[j].x, xb[j].y)); v[4 * j + 3] = norm2(make_float2(xb[j].z, xb[j].w));
        }
    }
    {
        float lmax = v[0], lmin = v[0], ls0 = 0.f, ls1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            lmax = pvxw::max3f(lmax, v[i], v[i + 1]); lmin = pvxw::min3f(lmin, v[i], v[i + 1]);
            ls0 += v[i]; ls1 += v[i + 1];
        }
        pvxw::wave_max_min_sum_nn(lmax, lmin, (double)ls0 + (double)ls1, maxe, miny, tot);
        const float maxy = __builtin_amdgcn_sqrtf(maxe);
        const double minamp = (double)maxy * thr;                   // PF.py:60
        th = (minamp != 0.0) ? minamp * minamp - (double)miny : 0.0;
    }
    const float thf = __double2float_rd(th);                         // see peak_scan (pvx_wave.h)
    const int thb = thf < 0.f ? -1 : __float_as_int(thf);
    // rise[j][i], sign bit set: y[k-1] < y[k] at k = 256 j + 4 lane + i; rise[j][4] = the next lane's rise[j][0]
    int rise[4][5];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        // the bin to the left of the lane's first: the lane before's last (lane 0: lane 63's last of the quarter before; bin 0
        // has itself there and never rises)
        const int old = (j == 0) ? __float_as_int(v[0]) : __builtin_amdgcn_readlane(__float_as_int(v[4 * j - 1]), 63);
        const int left = __builtin_amdgcn_update_dpp(old, __float_as_int(v[4 * j + 3]), 0x138, 0xf, 0xf, false);   // wave_shr:1
        rise[j][0] = left - __float_as_int(v[4 * j]);
#pragma unroll
        for (int i = 1; i < 4; i++) rise[j][i] = __float_as_int(v[4 * j + i - 1]) - __float_as_int(v[4 * j + i]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int old = (j == 3) ? 0 : __builtin_amdgcn_readlane(rise[j + 1][0], 0);     // (bin 1023 is not interior: cleared below)
        rise[j][4] = __builtin_amdgcn_update_dpp(old, rise[j][0], 0x130, 0xf, 0xf, false);                          // wave_shl:1
    }
    unsigned m = 0;                                                  // bit 4 j + i
    float sc[16];                                                    // scores y - miny (>= 0)
#pragma unroll
    for (int j = 3; j >= 0; j--) {
#pragma unroll
        for (int i = 3; i >= 0; i--) {
            sc[4 * j + i] = v[4 * j + i] - miny;
            const int above = thb - __float_as_int(sc[4 * j + i]);   // sign bit set: score > thf
            const unsigned t = (unsigned)(rise[j][i] & ~rise[j][i + 1] & above);
            m = (m << 1) | (t >> 31);
        }
    }
    if (lane == 63) m &= ~(1u << 15);                                // bin 1023
    // list positions of the lane's candidates, per quarter
    unsigned pk01, pk23;
    auto count = [&](unsigned mm) -> int {
        int C = 0;
        unsigned pos[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int cnt = __popc((mm >> (4 * j)) & 15u);           // 0, 1 or 2
            const unsigned long long b0 = __ballot((cnt & 1) != 0), b1 = __ballot((cnt & 2) != 0);
            pos[j] = (unsigned)(C + pvxw::lane_prefix(b0) + 2 * pvxw::lane_prefix(b1));
            C += __popcll(b0) + 2 * __popcll(b1);
        }
        pk01 = pos[0] | (pos[1] << 16); pk23 = pos[2] | (pos[3] << 16);
        return C;
    };
    int C = count(m);
    if (C > PVX_THIN_FROM && npeaks <= MAXK) {                                 // wave-uniform; see peak_scan_block_thin
        float best = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i++) best = fmaxf(best, ((m >> i) & 1u) ? sc[i] : 0.f);
        const float T = pvxw::thin_bound<MAXK>(best, npeaks);
        unsigned keep = 0u;
#pragma unroll
        for (int i = 15; i >= 0; i--) keep = (keep << 1) | (sc[i] >= T ? 1u : 0u);
        m &= keep;
        C = count(m);
    }
    // one round per candidate of the busiest lane: every lane pops its lowest set bit (lanes that have run out write to
    // their trash slot)
    const int tr = trash + lane;
    while (__ballot(m != 0u) != 0ull) {                              // wave-uniform
        const bool has = m != 0u;
        const int b = __ffs((int)m) - 1;                             // 4 j + i
        const bool hi = (b & 8) != 0;
        const unsigned sh = (unsigned)(b & 4) << 2;                  // 16 (j & 1)
        const unsigned pk = hi ? pk23 : pk01;
        const int pos = (int)((pk >> sh) & 0xffffu);
        ci[has ? pos : tr] = (CI)(((b >> 2) << 8) + 4 * lane + (b & 3));
        const unsigned inc = has ? (1u << sh) : 0u;
        pk23 += hi ? inc : 0u;
        pk01 += hi ? 0u : inc;
        m &= m - 1u;
    }
    return C;
}

constexpr float kC16 = 0.92387953251128673848f;   // cos(pi/8)
constexpr float kS16 = 0.38268343236508978178f;   // sin(pi/8)
constexpr float kH8 = 0.70710678118654752440f;    // sqrt(1/2)

// (S = Za + conj Zb, O = -i/2 (Za - conj Zb) W_pair, twiddle w) -> X[k] = S/2 + O w, X[M-k] = conj(S/2 - O w)
__device__ __forceinline__ void untangle_so(v2f Sm, v2f O, v2f w, v2f& x0, v2f& x1) {
    const v2f khalf = pvxc::splat(0.5f);
    const v2f Pk = pvxc::cmul(O, w);
    x0 = __builtin_elementwise_fma(khalf, Sm, Pk);
    x1 = pvxc::fms_conj(khalf, Sm, Pk);
}

// The four pairs (A_t, B_((4-t) mod 4)), t < 4, with untangle twiddles W_8^t wu -> x0[t] = X[k1 + LQ t], x1[t] = X[(LQ - k1) + LQ (3-t)]
__device__ __forceinline__ void untangle4(const v2f (&A)[4], const v2f (&B)[4], v2f wu, v2f (&x0)[4], v2f (&x1)[4]) {
    const v2f kmih = pvxc::mk(0.5f, -0.5f), kmh = pvxc::splat(-0.5f);
    untangle_so(pvxc::add_conj(A[0], B[0]), pvxc::mul_swap(pvxc::sub_conj(A[0], B[0]), kmih), wu, x0[0], x1[0]);
    untangle_so(pvxc::add_conj(A[1], B[3]), pvxc::cmul_k(pvxc::mul_swap(pvxc::sub_conj(A[1], B[3]), kmih), pvxc::mk(kH8, -kH8)), wu, x0[1], x1[1]);
    untangle_so(pvxc::add_conj(A[2], B[2]), pvxc::sub_conj(A[2], B[2]) * kmh, wu, x0[2], x1[2]);
    untangle_so(pvxc::add_conj(A[3], B[1]), pvxc::cmul_k(pvxc::mul_swap(pvxc::sub_conj(A[3], B[1]), kmih), pvxc::mk(-kH8, -kH8)), wu, x0[3], x1[3]);
}

__device__ __forceinline__ void dft4(const v2f (&a)[4], v2f (&A)[4]) {
    const v2f e = a[0] + a[2], f = a[0] - a[2], g = a[1] + a[3], h = a[1] - a[3];
    A[0] = e + g; A[2] = e - g; A[1] = pvxc::add_mni(f, h); A[3] = pvxc::add_pi(f, h);
}

// The k1 = LQ/2 family: its four values c_u (already read) pair among themselves -> spv[u] = X[LQ/2 + LQ u]
__device__ __forceinline__ void special4(const v2f (&c)[4], v2f (&spv)[4]) {
    const v2f khalf = pvxc::splat(0.5f), kmih = pvxc::mk(0.5f, -0.5f);
    const v2f c1 = pvxc::cmul_k(c[1], pvxc::mk(kH8, -kH8)), c2 = pvxc::mni(c[2]), c3 = pvxc::cmul_k(c[3], pvxc::mk(-kH8, -kH8));
    const v2f A = c[0] + c2, B = c[0] - c2, C = c1 + c3, D = c1 - c3;
    const v2f z0 = A + C, z2 = A - C, z1 = pvxc::add_mni(B, D), z3 = pvxc::add_pi(B, D);   // Z[LQ/2 + LQ u]
    {   // (u = 0, 3): twiddle W_16
        const v2f Sm = pvxc::add_conj(z0, z3), Dd = pvxc::sub_conj(z0, z3);
        const v2f Pk = pvxc::cmul_k(pvxc::mul_swap(Dd, kmih), pvxc::mk(kC16, -kS16));
        spv[0] = __builtin_elementwise_fma(khalf, Sm, Pk);
        spv[3] = pvxc::fms_conj(khalf, Sm, Pk);
    }
    {   // (u = 1, 2): twiddle W_16^3
        const v2f Sm = pvxc::add_conj(z1, z2), Dd = pvxc::sub_conj(z1, z2);
        const v2f Pk = pvxc::cmul_k(pvxc::mul_swap(Dd, kmih), pvxc::mk(kS16, -kC16));
        spv[1] = __builtin_elementwise_fma(khalf, Sm, Pk);
        spv[2] = pvxc::fms_conj(khalf, Sm, Pk);
    }
}

// Radix-4 join fused with the untangle, in place.  xz: the buffer (quarter u at u QP, index i of a quarter at IA(i));
// LQ: points per quarter; T lanes (lt = this one) share the LQ/2 sets, NPS = LQ / (2 T) per lane; tw[j] = {W_N^k1,
// W_M^k1, W_M^(2 k1), W_M^(3 k1)} of k1 = lt + T j (M = 4 LQ, N = 2 M).  |X|^2 of every bin -> Ly (padded layout);
// lmax / lmin / ls0 / ls1 accumulate the lane's max, min and sums of |X|^2.
// WY = false: no |X|^2 row is written (k_fused_rev at nfft 2048: the peak search recomputes it from the spectrum, YofX4).
// WS = false (with WY = false): no |X|^2 at all -- the row's maximum / minimum / energy come from the peak search as well
// (peak_scan_x4_thin), which forms every |X|^2 anyway.
template <int LQ, int QP, int T, typename IA, bool WY = true, bool WS = true>
__device__ __forceinline__ void join4_untangle(v2f* xz, float* Ly, const v2f (&tw)[LQ / (2 * T)][4], int lt, IA ia,
                                               float& lmax, float& lmin, float& ls0, float& ls1) {
    constexpr int NPS = LQ / (2 * T);
    v2f spv[4];
    {
        v2f c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) c[u] = xz[u * QP + ia(LQ / 2)];
        special4(c, spv);
    }
#pragma unroll
    for (int j = 0; j < NPS; j++) {
        const int k1 = lt + T * j;
        const int kb = (LQ - k1) & (LQ - 1);
        const int sa = ia(k1);
        int sb = ia(kb);
        v2f a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { a[u] = xz[u * QP + sa]; b[u] = xz[u * QP + sb]; }
#pragma unroll
        for (int u = 1; u < 4; u++) {
            a[u] = pvxc::cmul(a[u], tw[j][u]);
            b[u] = pvxc::cmul_conj(b[u], tw[j][u]);
        }
        v2f A[4], B[4], x0[4], x1[4];
        dft4(a, A);
        dft4(b, B);
        untangle4(A, B, tw[j][0], x0, x1);
        int kbb = kb;                                               // bins of the mirrored slots: kbb + LQ (3 - t)
        if (j == 0) {
            if (lt == 0) {
#pragma unroll
                for (int t = 0; t < 4; t++) x1[t] = spv[3 - t];
                kbb = LQ / 2; sb = ia(LQ / 2);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            xz[t * QP + sa] = x0[t];                                // X[k1 + LQ t]
            xz[(3 - t) * QP + sb] = x1[t];                          // X[kbb + LQ (3 - t)]
            if constexpr (WY || WS) {
                const float e0 = __builtin_fmaf(x0[t].x, x0[t].x, x0[t].y * x0[t].y), e1 = __builtin_fmaf(x1[t].x, x1[t].x, x1[t].y * x1[t].y);
                if constexpr (WY) {
                    Ly[pvxw::ymap<1>(k1 + LQ * t)] = e0;
                    Ly[pvxw::ymap<1>(kbb + LQ * (3 - t))] = e1;
                }
                lmax = pvxw::max3f(lmax, e0, e1); lmin = pvxw::min3f(lmin, e0, e1); ls0 += e0; ls1 += e1;
            }
        }
    }
}

// The plain radix-4 join of a wave's four quarters, in place (a team of 4 waves joins the waves' 1024-point results in a
// second pass, join4_untangle over the regions): E[k1 + 256 t] = sum_u W_4^(u t) W_1024^(u k1) E_u[k1], k1 = lane + 64 j.
// tw: LDS table [4][3][64] of W_1024^(u k1), u = 1..3.
__device__ __forceinline__ void join4_plain(v2f* dz, const v2f* tw, int lane) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k1 = lane + 64 * j;
        v2f a[4], A[4];
#pragma unroll
        for (int u = 0; u < 4; u++) a[u] = dz[u * F4::RP + k1];
#pragma unroll
        for (int u = 1; u < 4; u++) a[u] = pvxc::cmul(a[u], tw[(j * 3 + (u - 1)) * 64 + lane]);
        dft4(a, A);
#pragma unroll
        for (int t = 0; t < 4; t++) dz[t * F4::RP + k1] = A[t];
    }
}

// Radix-8 join fused with the untangle for a team of TWO waves (nfft 4096): the eight 256-point quarters E_c, c = 2 u + s
// (wave s, quarter u), sit at slot (c & 1) BUF + (c >> 1) RP + k1; lane lt < 128 takes k1 = lt: the 16 values of k1 and
// 256 - k1 in, the 16 bins X[k1 + 256 t], X[(256 - k1) + 256 (7 - t)] out (slot t RP + k1: natural order across both
// regions), in place.  tw = {W_N^k1, W_M^k1, ..., W_M^(7 k1)} (M = 2048, N = 4096).  The k1 = 128 family (bins 128 + 256 u)
// goes to the mirrored slots of lane 0, which only wave 0 computes (`special`).
__device__ __forceinline__ void join8_untangle(v2f* xz, float* Ly, const v2f (&tw)[8], int lt, bool special,
                                               float& lmax, float& lmin, float& ls0, float& ls1) {
    const v2f khalf = pvxc::splat(0.5f), kmih = pvxc::mk(0.5f, -0.5f), kmh = pvxc::splat(-0.5f);
    constexpr float C32[4] = {0.98078528040323044913f, 0.83146961230254523708f, 0.55557023301960222474f, 0.19509032201612826785f};   // cos((1 + 2t) pi / 16)
    constexpr float S32[4] = {0.19509032201612826785f, 0.55557023301960222474f, 0.83146961230254523708f, 0.98078528040323044913f};   // sin((1 + 2t) pi / 16)
    // W_16^t, t < 8
    constexpr float W16r[8] = {1.f, kC16, kH8, kS16, 0.f, -kS16, -kH8, -kC16};
    constexpr float W16i[8] = {0.f, -kS16, -kH8, -kC16, -1.f, -kC16, -kH8, -kS16};
    auto slot = [](int c, int k) -> int { return (c & 1) * F4::BUF + (c >> 1) * F4::RP + k; };
    v2f spv[8];
    if (special) {                                                  // wave-uniform
        v2f c[8];
#pragma unroll
        for (int cc = 0; cc < 8; cc++) {
            const v2f v = xz[slot(cc, 128)];
            c[cc] = (cc == 0) ? v : (cc == 4) ? pvxc::mni(v) : pvxc::cmul_k(v, pvxc::mk(W16r[cc], W16i[cc]));     // W_M^(128 c) = W_16^c
        }
        dft_regs<8>(c);                                             // Z[128 + 256 u]
#pragma unroll
        for (int t = 0; t < 4; t++) {                               // pairs (t, 7 - t), twiddle W_N^(128 + 256 t) = W_32^(1 + 2 t)
            const v2f Sm = pvxc::add_conj(c[t], c[7 - t]), Dd = pvxc::sub_conj(c[t], c[7 - t]);
            const v2f Pk = pvxc::cmul_k(pvxc::mul_swap(Dd, kmih), pvxc::mk(C32[t], -S32[t]));
            spv[t] = __builtin_elementwise_fma(khalf, Sm, Pk);
            spv[7 - t] = pvxc::fms_conj(khalf, Sm, Pk);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 8; t++) spv[t] = pvxc::splat(0.f);
    }
    const int k1 = lt;
    const int kb = (256 - k1) & 255;
    v2f a[8], b[8];
#pragma unroll
    for (int c = 0; c < 8; c++) { a[c] = xz[slot(c, k1)]; b[c] = xz[slot(c, kb)]; }
#pragma unroll
    for (int c = 1; c < 8; c++) {
        a[c] = pvxc::cmul(a[c], tw[c]);
        b[c] = pvxc::cmul_conj(b[c], tw[c]);
    }
    dft_regs<8>(a);
    dft_regs<8>(b);
    v2f x0[8], x1[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
        // (A_t, B_((8 - t) mod 8)), twiddle W_16^t W_N^k1
        const v2f zb = b[(8 - t) & 7];
        const v2f Sm = pvxc::add_conj(a[t], zb), D = pvxc::sub_conj(a[t], zb);
        v2f O;
        if (t == 0) O = pvxc::mul_swap(D, kmih);
        else if (t == 4) O = D * kmh;
        else O = pvxc::cmul_k(pvxc::mul_swap(D, kmih), pvxc::mk(W16r[t], W16i[t]));
        untangle_so(Sm, O, tw[0], x0[t], x1[t]);
    }
    int kbb = kb, sbk = kb;
    if (lt == 0) {
#pragma unroll
        for (int t = 0; t < 8; t++) x1[t] = spv[7 - t];
        kbb = 128; sbk = 128;
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const float e0 = __builtin_fmaf(x0[t].x, x0[t].x, x0[t].y * x0[t].y), e1 = __builtin_fmaf(x1[t].x, x1[t].x, x1[t].y * x1[t].y);
        xz[t * F4::RP + k1] = x0[t];                                // X[k1 + 256 t]
        xz[(7 - t) * F4::RP + sbk] = x1[t];                         // X[kbb + 256 (7 - t)]
        Ly[pvxw::ymap<1>(k1 + 256 * t)] = e0;
        Ly[pvxw::ymap<1>(kbb + 256 * (7 - t))] = e1;
        lmax = pvxw::max3f(lmax, e0, e1); lmin = pvxw::min3f(lmin, e0, e1); ls0 += e0; ls1 += e1;
    }
}

// The four 256-point transforms of a wave, stages 1 and 2 (between them the transpose inside the 16-lane groups), and
// the natural-order store: quarter u of `dz`, E_u[k'] at u RP + k'.  z: the lane's 16 windowed values.
// t1: LDS table [16][16] W_256^(l q).  HOOK1 / HOOK2 run after stage 1 / during the transpose (the callers issue
// their sample prefetches there).
template <typename H1, typename H2, typename H3>
__device__ __forceinline__ void fft4_quarters(v2f (&z)[16], v2f* dz, const v2f* t1, int lane, H1 hook1, H2 hook2, H3 hook3) {
    using pvxw::wave_sync;
    const int l = lane & 15, u = lane >> 4;
#ifndef PVX_TW_EARLY
#define PVX_TW_EARLY 1
#endif
    v2f tw[16];
#pragma unroll
    for (int q = 1; q < PVX_TW_EARLY; q++) tw[q] = t1[q * 16 + l];
#pragma unroll
    for (int q = 1; q < PVX_TW_EARLY; q++) asm volatile("" : "+v"(tw[q]));
    dft_regs<16>(z);                                                // stage 1: radix-16 over r
    __builtin_amdgcn_sched_barrier(0);
    hook1();
    v2f* const ew = dz + u * F4::EU + l;
    // the sixteen twiddles of the lane first, ALL in flight before the first product (taken a pair at a time beside the
    // stores -- what the compiler makes of the plain loop -- every pair is an LDS round trip of its own behind the store of
    // the pair before: eight in a row, the longest stall of the transform); then two rows at a time, adjacent, so that the
    // accesses pair into ds_read2 / ds_write2
#pragma unroll
    for (int q = PVX_TW_EARLY; q < 16; q++) tw[q] = t1[q * 16 + l];
#pragma unroll
    for (int q = PVX_TW_EARLY; q < 16; q++) asm volatile("" : "+v"(tw[q]));
#pragma unroll
    for (int q2 = 0; q2 < 16; q2 += 2) {
        const v2f pa = (q2 > 0) ? pvxc::cmul(z[q2], tw[q2]) : z[q2], pb2 = pvxc::cmul(z[q2 + 1], tw[q2 + 1]);
        ew[q2 * F4::EP] = pa;
        ew[(q2 + 1) * F4::EP] = pb2;
    }
    wave_sync();
    const v2f* const er = dz + u * F4::EU + l * F4::EP;
#pragma unroll
    for (int l2 = 0; l2 < 16; l2++) z[l2] = er[l2];
    hook2();
    wave_sync();
    dft_regs<16>(z);                                                // stage 2: radix-16 over l
    __builtin_amdgcn_sched_barrier(0);
    hook3();
    v2f* const nw = dz + u * F4::RP + l;
#pragma unroll
    for (int t = 0; t < 16; t++) nw[16 * t] = z[t];                 // E_u[l + 16 t]
}

}  // namespace pvxf
