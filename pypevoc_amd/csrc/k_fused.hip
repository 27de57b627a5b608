// k_fused.hip -- the whole analysis stage of one frame inside one wave64: nothing but the input
// samples and the <= K peaks ever touches HBM.  float32, nfft = 128 R with R in {4, 8, 16}
// (nfft 512, 1024, 2048; the template also instantiates for R = 32 but is not used there):
//
//   PV.calc_fft_frame   pypevoc/PVAnalysis.py:150-158   x[pos:pos+nfft]*win -> FFT -> /wfact
//   PV.calc_pv_frame    pypevoc/PVAnalysis.py:160-211   abs, PeakFinder, salience, phase difference,
//                                                        instantaneous frequency, 3-bin energy, realph
//   PV.run_pv           pypevoc/PVAnalysis.py:213-264   frame loop and zero-padded packing
//
// Why: the three-kernel form (k_frames -> rocFFT -> k_peaks) moves 35 KB per frame (nfft 2048)
// through HBM / Infinity Cache; here a frame costs hop*4 bytes of input (the overlap between
// consecutive frames is served by L2: one wave walks consecutive frames) and (5K+2)*8 B of output.
// The kernel is then bound by instruction issue, not by HBM.
//
// Real FFT of nfft samples = complex FFT of M = nfft/2 = 64 R points z[j] = xw[2j] + i xw[2j+1]
// followed by the untangle X[k] = E[k] + W_nfft^k O[k].  The M-point FFT is laid out on the 64 lanes
// x R registers of one wave as M = R (registers) x R (registers, after one LDS exchange) x P (across
// P = 64/R neighbouring lanes, by DPP / swizzle):
//   stage 1  lane l holds z[l + 64 r], r < R: radix-R DFT over r in registers, twiddle W_M^(l q)
//   exchange Y[l][q] -> LDS rows [q][l] (row pitch 64 + P complex: conflict-free both ways)
//   stage 2  lane (q, l1) = P q + l1 reads Y[l1 + P l2][q], l2 < R: radix-R DFT over l2, twiddle W_64^(l1 t2)
//   stage 3  P-point DFT over l1 across the P lanes of a group: log2 P decimation-in-frequency steps,
//            each one lane exchange (xor h) + one lane-constant twiddle W_2h^(l1 mod h)
//   result   lane (q, l1) holds Z[q + R t2 + R^2 t1], t1 = bitrev(l1)
// (R = 16: 1024 = 16 x 16 x 4, the quad-level 4-point DFT is two DPP quad_perm exchange steps.)
// Complex values are even-aligned register pairs (pvx_cplx.h): one packed instruction per complex add,
// two per complex multiply, conjugations and multiplications by -i folded into operand modifiers.
// Z goes to LDS in natural order (a few complex of padding per R^2 keep the accesses
// spread over the banks), is untangled in place into X[0..M), |X|^2 goes to a second (bank-padded) LDS
// row.  Peak search (pvx_wave.h): every lane owns R consecutive bins, candidate flags are sign bits of
// integer subtractions on the float bit patterns, list positions a ballot prefix (peak_block_masks /
// peak_block_write); ranking / radix select when there are more candidates than npks (peak_pick); the
// salience test with 8 lanes per peak (salient_groups).  The per-peak arithmetic (peak_math) is the same
// as in k_peaks.hip; the previous frame's spectrum is the LDS buffer the wave filled one iteration
// earlier.  The index maps were validated in numpy for every R before this was written.
//
// Work distribution: persistent-style.  Wave w owns the contiguous global rows [w*Rows/W, (w+1)*Rows/W)
// and recomputes the spectrum of the row before its first one (one extra FFT per wave).  No
// inter-wave communication.  The samples of the next row are prefetched while a row is processed.
// All float32 arithmetic that must not depend on which wave computes a frame uses explicit fmaf.
#include "pvx_fft.h"

using namespace pvxw;
using namespace pvxf;

namespace {

constexpr int GF = 8;               // frames staged before the per-peak pass

struct FusedLds {       // per-wave carve
    float2* bufA;       // [BUFC]
    float2* bufB;       // [BUFC]
    float* y;           // [M + 4 R] |X|^2, padded layout ymap<1>
    int* ci;            // [CAP]
    int* sel;           // [kpad]
    int* sbin;          // [GF][kpad]
    float* sval;        // [GF][kpad][5]   re, im, pr, pi, s3
    int* cnt;           // [GF]
    int* frm;           // [GF]  frame index within its signal
    long long* orow;    // [GF]
    double* tot;        // [GF]
};

template <int R> __host__ __device__ inline size_t fused_lds_per_wave(int K) {
    using G = Geo<R>;
    const size_t kpad = (size_t)((K + 3) & ~3);
    const size_t gs = (size_t)staged_frames(K, GF);
    size_t b = (size_t)G::BUFC * 8 * 2 + (size_t)(G::M + 4 * R) * 4 + (size_t)(G::CAP + 64) * 4 + kpad * 4 +
               gs * kpad * 4 + gs * kpad * 5 * 4 + GF * 4 + GF * 4;
    b = (b + 7) & ~(size_t)7;
    b += GF * 8 + GF * 8;
    return (b + 15) & ~(size_t)15;
}
template <int R> __host__ __device__ inline size_t fused_lds_shared() {
    // W_nfft^k for k <= M/2 (padded) | W_64^(l1 t2) as [t2][l1] (64 entries)
    return (size_t)((Geo<R>::HALF + 8) & ~7) * 8 + 64 * 8;
}

template <int R, typename InT, bool AL2>
__global__ __launch_bounds__(128) void k_fused_pv(FusedParams p) {
    using G = Geo<R>;
    constexpr int M = G::M, P = G::P, PITCH = G::PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    // the wave index is wave-uniform but the compiler cannot know: readfirstlane moves it -- and with it
    // the whole row bookkeeping (loop counters, row addresses, branches) -- to the scalar unit
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = blockDim.x >> 6;
    const int K = p.K;
    const int kpad = (K + 3) & ~3;
    float2* tw3 = (float2*)smem;                                  // shared by the block
    float2* tw2l = (float2*)(smem + (size_t)((G::HALF + 8) & ~7) * 8);
    unsigned char* base = smem + fused_lds_shared<R>() + fused_lds_per_wave<R>(K) * wid;
    FusedLds L;
    L.bufA = (float2*)base;
    L.bufB = L.bufA + G::BUFC;
    L.y = (float*)(L.bufB + G::BUFC);
    L.ci = (int*)(L.y + M + 4 * R);                               // y is padded: ymap<1>
    L.sel = L.ci + G::CAP + 64;                                   // 64 trash slots after the candidate list
    L.sbin = L.sel + kpad;
    const int gs = staged_frames(K, GF);
    L.sval = (float*)(L.sbin + gs * kpad);
    L.cnt = (int*)(L.sval + gs * kpad * 5);
    L.frm = L.cnt + GF;
    L.orow = (long long*)(((uintptr_t)(L.frm + GF) + 7) & ~(uintptr_t)7);
    L.tot = (double*)(L.orow + GF);

    const float2* tab = (const float2*)p.twiddle;                 // W_nfft^j, j < nfft
    constexpr int NMASK = G::N - 1;
    for (int k = threadIdx.x; k <= G::HALF; k += blockDim.x) tw3[k] = tab[k];
    for (int k = threadIdx.x; k < 64; k += blockDim.x) tw2l[k] = tab[((G::N / 64) * (k % P) * (k / P)) & NMASK];
    __syncthreads();

    // ---- lane constants (complex values are register pairs, pvx_cplx.h)
    const int Q = lane / P, L1 = lane % P;
    v2f wv[R], t1[R], t2[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        wv[r] = ((const v2f*)p.win)[lane + 64 * r];               // (w[2l + 128 r], w[2l + 128 r + 1])
        t1[r] = ((const v2f*)tab)[(2 * lane * r) & NMASK];        // W_M^(l q)
        t2[r] = ((const v2f*)tw2l)[r * P + L1];                   // W_64^(l1 t2)
    }
    // cross-lane DFT constants: step with half-size h = P >> (s+1): sign and twiddle W_2h^(l1 mod h)
    float csg[G::LOGP > 0 ? G::LOGP : 1];
    v2f cw[G::LOGP > 0 ? G::LOGP : 1];
#pragma unroll
    for (int s = 0; s < G::LOGP; s++) {
        const int h = P >> (s + 1);
        const bool up = (L1 & h) != 0;
        csg[s] = up ? -1.f : 1.f;
        const float2 wvv = tab[((G::N / (2 * h)) * (L1 % h)) & NMASK];
        cw[s] = up ? pvxc::mk(wvv.x, wvv.y) : pvxc::mk(1.f, 0.f);
    }
    // keep the lane constants in registers: without this the compiler re-loads the twiddles from
    // global memory every frame (a full L2 round trip on the critical path) instead of holding them
#pragma unroll
    for (int r = 0; r < R; r++) {
        asm volatile("" : "+v"(wv[r]), "+v"(t1[r]), "+v"(t2[r]));
    }
    int t1v = 0;                                                  // t1 = bitrev(l1)
#pragma unroll
    for (int b = 0; b < G::LOGP; b++) if (L1 & (1 << b)) t1v |= 1 << (G::LOGP - 1 - b);

    // ---- rows of this wave
    const int64_t W = (int64_t)gridDim.x * nwaves;
    const int64_t w = (int64_t)blockIdx.x * nwaves + wid;
    const int64_t r0 = p.total_rows * w / W, r1 = p.total_rows * (w + 1) / W;
    if (r0 >= r1) return;

    PeakConst pc;
    pc.fstep = p.fstep; pc.dt = p.dt; pc.nfft = G::N; pc.hop = p.hop; pc.wfbin = p.wfbin;

    float2* cur = L.bufA;
    float2* prv = L.bufB;
    v2f raw[R];                                                   // raw samples of the next row (prefetched)
#pragma unroll
    for (int r = 0; r < R; r++) raw[r] = pvxc::splat(0.f);
    // Rows are addressed as (signal b, row-in-signal q), advanced incrementally: a 64-bit division
    // per frame costs more than the whole peak search.
    auto prefetch = [&](int64_t gn, int64_t bn, int64_t qn) {     // issue the loads of global row gn = (bn, qn)
        if (gn < 0 || gn >= r1 || qn == 0) return;
        load_raw<R, InT, AL2>((const InT*)p.x + bn * p.sig_stride + (qn - 1) * (int64_t)p.hop, lane, raw);
    };

    // spectrum of global row g into `dst` (zeros for a zero row); with_mag: also |X| -> y and the
    // wave-reduced max / min / energy
    // (maxe, mine = largest / smallest |X|^2 of the row)
    auto spectrum = [&](int64_t g, int64_t b, int64_t q, float2* dst, bool with_mag, float& maxe, float& mine, double& tot) {
        // (b, q) of row g + 1
        const int64_t qn = (q == p.F) ? 0 : q + 1;
        const int64_t bn = (q == p.F) ? b + 1 : b;
        // One prefetch site, after the multiplies, for both kinds of row: with the prefetch duplicated into
        // the zero-row branch the compiler hoists it above the branch and copies all of `raw` every frame.
        // (For a zero row `raw` is stale and z is garbage that nobody reads.)
        v2f z[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            z[r] = raw[r] * wv[r];
            asm volatile("" : "+v"(z[r]));                        // the multiply stays here: it must not sink below the loads
        }
        __builtin_amdgcn_sched_barrier(0);
        prefetch(g + 1, bn, qn);
        if (g < 0 || q == 0) {
#pragma unroll
            for (int j = 0; j < G::BUFC / 64; j++) dst[lane + 64 * j] = make_float2(0.f, 0.f);
            wave_sync();
            return;
        }
        dft_regs<R>(z);                                           // stage 1
        v2f* dz = (v2f*)dst;
#pragma unroll
        for (int q2 = 0; q2 < R; q2++) {
            dz[q2 * PITCH + lane] = (q2 > 0) ? pvxc::cmul(z[q2], t1[q2]) : z[q2];
        }
        wave_sync();
#pragma unroll
        for (int l2 = 0; l2 < R; l2++) z[l2] = dz[Q * PITCH + L1 + P * l2];
        wave_sync();
        dft_regs<R>(z);                                           // stage 2
#pragma unroll
        for (int t0 = 0; t0 < R; t0 += 4) {
            // twiddle W_64^(l1 t2), then stage 3: P-point DFT across P lanes, decimation in frequency
            // (lower lane a + b, upper lane (a - b) W_2h^(l1 mod h)), four values at a time
            v2f a[4];
#pragma unroll
            for (int j = 0; j < 4; j++) a[j] = (t0 + j > 0) ? pvxc::cmul(z[t0 + j], t2[t0 + j]) : z[t0 + j];
            if constexpr (G::LOGP >= 1) {
                if constexpr (P >= 16) xstep4<8, true>(a, csg[G::LOGP - 4], cw[G::LOGP - 4]);
                if constexpr (P >= 8) xstep4<4, true>(a, csg[G::LOGP - 3], cw[G::LOGP - 3]);
                if constexpr (P >= 4) xstep4<2, true>(a, csg[G::LOGP - 2], cw[G::LOGP - 2]);
                xstep4<1, false>(a, csg[G::LOGP - 1], cw[G::LOGP - 1]);           // h = 1: twiddle is 1
            }
#pragma unroll
            for (int j = 0; j < 4; j++) dz[zpad<R>(Q + R * (t0 + j) + G::R2 * t1v)] = a[j];
        }
        wave_sync();
        // ---- untangle in place: pairs (k, M-k), k = lane + 64 j; bins 0 and M/2 have no partner.
        // Phase 1 reads everything (the loads do not wait for the in-place stores of other pairs),
        // phase 2 computes and stores.
        //   S = Za + conj Zb, D = Za - conj Zb;  E = S/2, O = -i D/2, P = W^k O
        //   X[k] = E + P,  X[M-k] = conj(E - P)
        constexpr int NPAIR = R / 2;
        float lmax = -INFINITY, lmin = INFINITY, ls0 = 0.f, ls1 = 0.f;
        v2f za[NPAIR], zb[NPAIR], wv8[NPAIR];
#pragma unroll
        for (int j = 0; j < NPAIR; j++) {
            const int k = lane + 64 * j;
            const int km = (M - k) & (M - 1);                     // k = 0: Z[M] == Z[0]
            za[j] = dz[zpad<R>(k)];
            zb[j] = dz[zpad<R>(km)];
            wv8[j] = ((const v2f*)tw3)[k];
        }
        const v2f zc = dz[zpad<R>(G::HALF)];
        const v2f khalf = pvxc::splat(0.5f), kmih = pvxc::mk(0.5f, -0.5f);
#pragma unroll
        for (int j = 0; j < NPAIR; j++) {
            const int k = lane + 64 * j;
            const int km = (M - k) & (M - 1);
            const v2f S = pvxc::add_conj(za[j], zb[j]);
            const v2f D = pvxc::sub_conj(za[j], zb[j]);
            const v2f O = pvxc::mul_swap(D, kmih);                // (D.y/2, -D.x/2)
            const v2f Pk = pvxc::cmul(O, wv8[j]);                 // W^k O
            const v2f x0 = __builtin_elementwise_fma(khalf, S, Pk);               // X[k] = S/2 + P
            v2f x1 = pvxc::fms_conj(khalf, S, Pk);                                // X[M-k] = conj(S/2 - P)
            int kk = km;
            if (j == 0) {
                // lane 0: k = 0 pairs with itself and its "partner" result is not a bin; that slot
                // takes bin M/2, which pairs with itself too: X[M/2] = conj(Z[M/2])
                if (lane == 0) { x1 = pvxc::mk(zc.x, -zc.y); kk = G::HALF; }
            }
            const float e0 = __builtin_fmaf(x0.x, x0.x, x0.y * x0.y), e1 = __builtin_fmaf(x1.x, x1.x, x1.y * x1.y);
            dz[zpad<R>(k)] = x0;
            dz[zpad<R>(kk)] = x1;
            if (with_mag) {
                // the peak search runs on |X|^2: every test it makes (local maximum, threshold, ranking,
                // salience) is monotone in |X|, and the square roots are a quarter-rate instruction
                L.y[k + 4 * j] = e0; L.y[ymap<1>(kk)] = e1;       // padded row: ymap<1>(lane + 64 j) = k + 4 j
                lmax = fmaxf(lmax, fmaxf(e0, e1)); lmin = fminf(lmin, fminf(e0, e1)); ls0 += e0; ls1 += e1;
            }
        }
        const double lsum = (double)ls0 + (double)ls1;
        if (with_mag) {
            maxe = wave_max(lmax);
            mine = wave_min(lmin);
            tot = wave_sum(lsum);
        }
        wave_sync();
    };

    // per-peak pass over the staged frames [0, ng)
    int LPF = 1;
    while (LPF < K && LPF < 64) LPF <<= 1;
    const int G_ = gs;                                            // frames staged per pass
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
                if (p.t) p.t[orow] = ((double)(fr * (int64_t)p.hop) + G::N / 2.0) / p.sr;            // PV.py:247
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
        float maxe = 0.f, mine = 0.f;
        double tot = 0.0;
        spectrum(g, b, q, cur, true, maxe, mine, tot);
        if (q != 0) {
            const int64_t orow = b * p.F + (q - 1);
            // PeakFinder(famp, npeaks, minrattomax) + filter_by_salience(rad=5)  (PV.py:175-178)
            // v_sqrt_f32 (1 ulp) instead of the 15-instruction correctly rounded sequence
            const float maxy = __builtin_amdgcn_sqrtf(maxe);
            const double minamp = (double)maxy * p.thr;           // PF.py:60
            // PF.py:69-70, 174: a bin qualifies when |X| - miny > minamp - miny, i.e. |X| > minamp; on the
            // squared row: |X|^2 - mine > minamp^2 - mine.  minamp == 0 means minamp = miny there: the
            // threshold is then EXACTLY 0 (its sign selects the "zeros qualify too" rule of findpos)
            const double th = (minamp != 0.0) ? minamp * minamp - (double)mine : 0.0;
            const int nsel = peak_select_block<R, int>(L.y, L.ci, G::CAP, L.sel, K, th, mine, lane);
            const bool use_prev0 = (p.prev0 != nullptr) && (orow == 0);
            int nk = 0;
            for (int eb = 0; eb < nsel; eb += 64) {
                const int e = eb + lane;
                int pb = 0;
                if (e < nsel) pb = L.sel[e];
                const bool keep = (p.rad <= 8) ? salient_groups<1>(L.y, M, L.sel, eb, nsel, p.rad, lane)
                                               : ((e < nsel) && salient<float, 1>(L.y, M, pb, p.rad));
                const unsigned long long bal = __ballot(keep);
                if (keep) {
                    const int slot = ng * kpad + nk + lane_prefix(bal);
                    const float2 c = cur[zpad<R>(pb)];
                    float2 pv;
                    if (use_prev0) pv = make_float2((float)p.prev0[2 * pb], (float)p.prev0[2 * pb + 1]);
                    else pv = prv[zpad<R>(pb)];
                    // PV.py:197-199: 3-bin energy, bin 0 excluded.  A selected bin is an interior
                    // local maximum, 1 <= pb <= M-2: pb+1 is always a bin, pb-1 counts unless it is 0
                    const float2 vm = cur[zpad<R>(pb - 1)], vp = cur[zpad<R>(pb + 1)];
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
            if (ng == G_) { flush(ng); ng = 0; }
        }
        if (p.spec_out != nullptr && g == p.spec_row) {
#pragma unroll
            for (int j = 0; j < R; j++) {
                const float2 v = cur[zpad<R>(lane + 64 * j)];
                p.spec_out[2 * (lane + 64 * j)] = v.x;
                p.spec_out[2 * (lane + 64 * j) + 1] = v.y;
            }
        }
        float2* t = cur; cur = prv; prv = t;
    }
    if (ng > 0) flush(ng);
}

template <int R> int launch_fused_r(const FusedParams& p, int x_dtype, hipStream_t s) {
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    const size_t per_wave = fused_lds_per_wave<R>(p.K), shared = fused_lds_shared<R>();
    int waves = 2;
    if (shared + per_wave * waves > 160 * 1024) waves = 1;
    const size_t lds = shared + per_wave * waves;
    if (lds > 160 * 1024) { pvx_set_error("nfft=%d npks=%d needs %zu bytes of LDS in the fused kernel", Geo<R>::N, p.K, lds); return PVX_ERR_UNSUPPORTED; }
    const bool al2 = (x_dtype == PVX_F32) && (p.hop % 2 == 0) && (p.sig_stride % 2 == 0) && (((uintptr_t)p.x) % 8 == 0);
    const void* fn = nullptr;
    switch (x_dtype) {
        case PVX_F32: fn = al2 ? (const void*)k_fused_pv<R, float, true> : (const void*)k_fused_pv<R, float, false>; break;
        case PVX_I16: fn = (const void*)k_fused_pv<R, int16_t, false>; break;
        default: pvx_set_error("the fused kernels take float32 or int16 samples (x_dtype %d: float64 is narrowed before the launch)", x_dtype); return PVX_ERR_INVALID;
    }
    if (lds > 64 * 1024) PVX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // resident workgroups per CU: what LDS and registers admit (at most 2 waves per SIMD)
    int blocks_per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, fn, 64 * waves, lds) != hipSuccess || blocks_per_cu < 1)
        blocks_per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;
    if (blocks_per_cu * waves > 8) blocks_per_cu = 8 / waves;
    int64_t nblocks = (int64_t)ncu * blocks_per_cu;
    if (p.blocks_override > 0) nblocks = p.blocks_override;
    // never more waves than rows; a short signal spreads one row per wave (plus its halo FFT): latency counts there
    const int64_t min_rows_per_wave = 1;
    const int64_t maxb = (p.total_rows / min_rows_per_wave + waves - 1) / waves;
    if (nblocks > maxb) nblocks = maxb > 0 ? maxb : 1;
    dim3 grid((unsigned)nblocks), block(64 * waves);
    switch (x_dtype) {
        case PVX_F32:
            if (al2) hipLaunchKernelGGL((k_fused_pv<R, float, true>), grid, block, lds, s, p);
            else hipLaunchKernelGGL((k_fused_pv<R, float, false>), grid, block, lds, s, p);
            break;
        default: hipLaunchKernelGGL((k_fused_pv<R, int16_t, false>), grid, block, lds, s, p); break;
    }
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}

}  // namespace

int pvx_fused_supported(int nfft, int precision, int K) {
    if (precision != 32) return 0;
    switch (nfft) {
        case 512: return fused_lds_shared<4>() + fused_lds_per_wave<4>(K) <= 160 * 1024;
        case 1024: return fused_lds_shared<8>() + fused_lds_per_wave<8>(K) <= 160 * 1024;
        case 2048: return fused_lds_shared<16>() + fused_lds_per_wave<16>(K) <= 160 * 1024;
        // nfft = 4096 (R = 32) works but needs > 512 registers per lane (spills) and 52 KB of LDS per
        // wave: no faster than the rocFFT path, so it stays there until the multi-wave-per-frame form
        default: return 0;
    }
}

int pvx_launch_fused(const FusedParams& p, int nfft, int x_dtype, hipStream_t s) {
    if (p.total_rows <= 0) return PVX_OK;
    switch (nfft) {
        case 512: return launch_fused_r<4>(p, x_dtype, s);
        case 1024: return launch_fused_r<8>(p, x_dtype, s);
        case 2048: return launch_fused_r<16>(p, x_dtype, s);
        default: pvx_set_error("the fused kernel does not handle nfft=%d", nfft); return PVX_ERR_UNSUPPORTED;
    }
}
