// k_fused_mw.hip -- the fused analysis stage with SEVERAL waves per frame: a workgroup of W waves
// (T = 64 W lanes) transforms one frame at a time, R = 8 complex points per lane, nfft = 2 T R:
//      nfft 2048 = 2 waves, nfft 4096 = 4 waves, nfft 8192 = 8 waves.
// Same arithmetic and same reference lines as k_fused.hip (PV.calc_fft_frame / calc_pv_frame / run_pv,
// pypevoc/PVAnalysis.py:150-264); what changes is the shape: with 8 points per lane the kernel needs
// ~230 registers instead of ~380, so two waves fit on every SIMD and the issue slots that a lone wave
// leaves empty (one instruction per 4 cycles) get used.  LDS is per frame, not per wave.
//
// M = nfft/2 = T R complex points = R (registers) x R (registers, after the LDS exchange) x P (P = T/R
// lanes, cross-lane decimation-in-frequency steps: DPP / ds_swizzle inside a wave; for P = 64 the
// first step pairs lanes 32 apart with ds_bpermute):
//   stage 1  lane l < T holds z[l + T r]: radix-R DFT over r, twiddle W_M^(l q)        | block barrier
//   stage 2  lane (q, l1) = P q + l1 reads Y[l1 + P l2][q]: radix-R DFT, twiddle W_T^(l1 t2)
//   stage 3  P-point DFT over l1 -> Z[q + R t2 + R^2 bitrev(l1)] -> LDS                | block barrier
//   untangle pairs (k, M-k), k = l + T j, j < R/2; |X| -> LDS; per-wave max/min/energy | block barrier
//   peaks    every wave scans its M/W bins (8 consecutive bins per lane), counts are exchanged  | block barrier
//            and every wave writes its candidates into the one ascending list          | block barrier
//            wave 0 selects, applies the salience test, stages the peaks'
//            raw data and (every G frames) does the per-peak arithmetic + stores        | block barrier
// The index maps were validated in numpy (R = 8, W = 2, 4, 8) before this was written.
#include "pvx_fft.h"

using namespace pvxw;
using namespace pvxf;

namespace {

constexpr int GFM = 8;              // frames staged before the per-peak pass

// Workgroup barrier for LDS hand-offs: LDS traffic drained, then s_barrier.  Not __syncthreads(): that also
// waits for vmcnt(0), i.e. for the prefetched samples of the next frame, at every one of the ~8 barriers of
// a frame.
__device__ __forceinline__ void block_sync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

template <int R, int W> struct GeoMW {
    static constexpr int T = 64 * W;                 // lanes per frame
    static constexpr int M = T * R;                  // complex FFT length = bins 0..M-1
    static constexpr int N = 2 * M;                  // nfft
    static constexpr int P = T / R;                  // lanes per cross-lane DFT
    static constexpr int LOGP = ilog2(P);
    static constexpr int LOGR = ilog2(R);
    static constexpr int PITCH = T + (P <= 16 ? P : 0);              // exchange row pitch (complex)
    static constexpr int R2 = R * R;
    static constexpr int QH = (P >= 32) ? 1 : 32 / P;                // q values per half wave
    static constexpr int ZP = (QH - (R2 % 32) + 32) % 32;            // padding per R^2 spectrum bins
    static constexpr int ZLEN = M + ZP * (P - 1);
    static constexpr int BUFRAW = (R * PITCH > ZLEN) ? R * PITCH : ZLEN;
    static constexpr int BUFC = ((BUFRAW + T - 1) / T) * T;          // complex slots per spectrum buffer
    static constexpr int HALF = M / 2;
    static constexpr int SCAN = M / W;                               // bins scanned by one wave
    static constexpr int CAPW = SCAN / 2 + 4;                        // candidate segment capacity
    static constexpr int CAP = CAPW * W;
    static constexpr int TW3 = (HALF + 8) & ~7;
};
template <int R, int W> __device__ __host__ __forceinline__ int zpadm(int k) { return k + GeoMW<R, W>::ZP * (k >> (2 * GeoMW<R, W>::LOGR)); }

template <int R, int W> __host__ __device__ inline size_t mw_lds_bytes(int K) {
    using G = GeoMW<R, W>;
    const size_t kpad = (size_t)((K + 3) & ~3);
    size_t b = (size_t)G::TW3 * 8 + (size_t)G::T * 8            // tw3 | tw2l
             + (size_t)G::BUFC * 8 * 2                          // bufA | bufB
             + (size_t)G::M * 4                                 // y
             + (size_t)(G::CAP * 2 + G::T) * 4                  // cs | ci | trash
             + kpad * 4                                         // sel
             + (size_t)staged_frames(K, GFM) * kpad * 4 * 6        // sbin | sval
             + GFM * 4 * 2 + W * 4 * 2 + W * 4 * 2;             // cnt | frm | Cw | Nw | pmax | pmin
    b = (b + 7) & ~(size_t)7;
    b += GFM * 8 * 2 + W * 8;                                   // orow | tot | psum
    return (b + 15) & ~(size_t)15;
}

// value of lane (l ^ H) within the wave, H up to 32
template <int H> __device__ __forceinline__ float lane_xor_w(float v, int lane) {
    if constexpr (H <= 8) return lane_xor<H>(v);
    else if constexpr (H == 16) return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (16 << 10) | 0x1f));
    else return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane ^ 32) << 2), __builtin_bit_cast(int, v)));
}

template <int H> __device__ __forceinline__ v2f lane_xor_w2(v2f v, int lane) {
    return pvxc::mk(lane_xor_w<H>(v.x, lane), lane_xor_w<H>(v.y, lane));
}
// xstep4 of pvx_fft.h with the exchanges that reach across rows of 16 lanes (H = 16, 32)
template <int H, bool TW> __device__ __forceinline__ void xstep4w(v2f (&a)[4], float sg, v2f w, int lane) {
    v2f q[4];
#pragma unroll
    for (int j = 0; j < 4; j++) q[j] = lane_xor_w2<H>(a[j], lane);
#pragma unroll
    for (int j = 0; j < 4; j++) a[j] = pvxc::fma_s(sg, a[j], q[j]);
    if constexpr (TW) {
#pragma unroll
        for (int j = 0; j < 4; j++) a[j] = pvxc::cmul(a[j], w);
    }
}

template <int R, int W, typename InT, bool AL2>
__device__ __forceinline__ void load_raw_mw(const InT* x, int tid, v2f (&raw)[R]) {
    constexpr int T = 64 * W;
    // thread l takes z[l + T r] = (x[2l + 2T r], x[2l + 2T r + 1]): 512 contiguous bytes per wave-instruction
#pragma unroll
    for (int r = 0; r < R; r++) {
        const InT* p = x + 2 * tid + 2 * T * r;
        if constexpr (AL2 && sizeof(InT) == 4) raw[r] = *(const v2f*)p;
        else raw[r] = pvxc::mk(ld1(p), ld1(p + 1));
    }
}

template <int R, int W, typename InT, bool AL2>
__global__ __launch_bounds__(64 * W, 2) void k_fused_mw(FusedParams p) {
    using G = GeoMW<R, W>;
    constexpr int T = G::T, M = G::M, P = G::P, PITCH = G::PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int K = p.K;
    const int kpad = (K + 3) & ~3;
    // ---- LDS carve (per block = per frame in flight)
    float2* tw3 = (float2*)smem;
    float2* tw2l = tw3 + G::TW3;
    float2* bufA = tw2l + T;
    float2* bufB = bufA + G::BUFC;
    float* y = (float*)(bufB + G::BUFC);
    float* cs = y + M;
    int* ci = (int*)(cs + G::CAP);
    int* sel = ci + G::CAP + T;                                   // T trash slots after the candidate list
    int* sbin = sel + kpad;
    const int gs = staged_frames(K, GFM);
    float* sval = (float*)(sbin + gs * kpad);
    int* cntv = (int*)(sval + gs * kpad * 5);
    int* frmv = cntv + GFM;
    int* Cw = frmv + GFM;
    int* Nw = Cw + W;                                             // survivors of each wave's local selection
    float* pmax = (float*)(Nw + W);
    float* pmin = pmax + W;
    long long* orowv = (long long*)(((uintptr_t)(pmin + W) + 7) & ~(uintptr_t)7);
    double* totv = (double*)(orowv + GFM);
    double* psum = totv + GFM;

    const float2* tab = (const float2*)p.twiddle;                 // W_nfft^j, j < nfft
    constexpr int NMASK = G::N - 1;
    for (int k = tid; k <= G::HALF; k += T) tw3[k] = tab[k];
    for (int k = tid; k < T; k += T) tw2l[k] = tab[((G::N / T) * (k % P) * (k / P)) & NMASK];   // [t2][l1] W_T^(l1 t2)
    __syncthreads();

    // ---- thread constants (complex values are register pairs, pvx_cplx.h)
    const int Q = tid / P, L1 = tid % P;
    v2f wv[R], t1[R], t2[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        wv[r] = ((const v2f*)p.win)[tid + T * r];                 // (w[2l + 2T r], w[2l + 2T r + 1])
        t1[r] = ((const v2f*)tab)[(2 * tid * r) & NMASK];         // W_M^(l q)
        t2[r] = ((const v2f*)tw2l)[r * P + L1];                   // W_T^(l1 t2)
    }
    float csg[G::LOGP];                                           // cross-lane DFT: sign and twiddle per step
    v2f cw[G::LOGP];
#pragma unroll
    for (int s = 0; s < G::LOGP; s++) {
        const int h = P >> (s + 1);
        const bool up = (L1 & h) != 0;
        csg[s] = up ? -1.f : 1.f;
        const float2 wvv = tab[((G::N / (2 * h)) * (L1 % h)) & NMASK];    // W_2h^(l1 mod h)
        cw[s] = up ? pvxc::mk(wvv.x, wvv.y) : pvxc::mk(1.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        asm volatile("" : "+v"(wv[r]), "+v"(t1[r]), "+v"(t2[r]));
    }
    int t1v = 0;                                                  // t1 = bitrev(l1)
#pragma unroll
    for (int b = 0; b < G::LOGP; b++) if (L1 & (1 << b)) t1v |= 1 << (G::LOGP - 1 - b);

    // ---- rows of this workgroup
    const int64_t NB = gridDim.x;
    const int64_t r0 = p.total_rows * (int64_t)blockIdx.x / NB, r1 = p.total_rows * ((int64_t)blockIdx.x + 1) / NB;
    if (r0 >= r1) return;                                         // block-uniform

    // output pointers and per-peak constants are re-read from the kernel argument segment inside flush() (every
    // 8th frame) instead of occupying ~30 scalar registers across the frame loop (see k_fused_ring.hip)
    const FusedParams* const kargs = (const FusedParams*)__builtin_amdgcn_kernarg_segment_ptr();

    float2* cur = bufA;
    float2* prv = bufB;
    v2f raw[R];                                                   // raw samples of the next row (prefetched)
#pragma unroll
    for (int r = 0; r < R; r++) raw[r] = pvxc::splat(0.f);
    auto prefetch = [&](int64_t gn, int64_t bn, int64_t qn) {
        if (gn < 0 || gn >= r1 || qn == 0) return;
        load_raw_mw<R, W, InT, AL2>((const InT*)p.x + bn * p.sig_stride + (qn - 1) * (int64_t)p.hop, tid, raw);
    };

    // spectrum of global row g into `dst` (zeros for a zero row); with_mag: also |X| -> y and the
    // block-reduced max / min / energy.  Every thread of the block takes part (block barriers inside).
    // (maxe, mine = largest / smallest |X|^2 of the row)
    auto spectrum = [&](int64_t g, int64_t b, int64_t q, float2* dst, bool with_mag, float& maxe, float& mine, double& tot) {
        const int64_t qn = (q == p.F) ? 0 : q + 1;
        const int64_t bn = (q == p.F) ? b + 1 : b;
        // one prefetch site, after the multiplies, for both kinds of row (see k_fused.hip); for a zero row
        // `raw` is stale and z is garbage that nobody reads
        v2f z[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            z[r] = raw[r] * wv[r];
            asm volatile("" : "+v"(z[r]));                        // the multiply stays here: it must not sink below the loads
        }
        __builtin_amdgcn_sched_barrier(0);
        prefetch(g + 1, bn, qn);
        if (g < 0 || q == 0) {                                    // block-uniform
#pragma unroll
            for (int j = 0; j < G::BUFC / T; j++) dst[tid + T * j] = make_float2(0.f, 0.f);
            block_sync_lds();
            return;
        }
        dft_regs<R>(z);                                           // stage 1
        v2f* dz = (v2f*)dst;
#pragma unroll
        for (int q2 = 0; q2 < R; q2++) dz[q2 * PITCH + tid] = (q2 > 0) ? pvxc::cmul(z[q2], t1[q2]) : z[q2];
        block_sync_lds();
#pragma unroll
        for (int l2 = 0; l2 < R; l2++) z[l2] = dz[Q * PITCH + L1 + P * l2];
        block_sync_lds();
        dft_regs<R>(z);                                           // stage 2
#pragma unroll
        for (int t0 = 0; t0 < R; t0 += 4) {
            // twiddle W_T^(l1 t2), then stage 3: P-point DFT across P lanes, decimation in frequency:
            // lower lane a + b, upper lane (a - b) W_2h^(l1 mod h); the last step (h = 1) has twiddle 1
            v2f a[4];
#pragma unroll
            for (int j = 0; j < 4; j++) a[j] = (t0 + j > 0) ? pvxc::cmul(z[t0 + j], t2[t0 + j]) : z[t0 + j];
            if constexpr (P >= 64) xstep4w<32, true>(a, csg[G::LOGP - 6], cw[G::LOGP - 6], lane);
            if constexpr (P >= 32) xstep4w<16, true>(a, csg[G::LOGP - 5], cw[G::LOGP - 5], lane);
            if constexpr (P >= 16) xstep4<8, true>(a, csg[G::LOGP - 4], cw[G::LOGP - 4]);
            if constexpr (P >= 8) xstep4<4, true>(a, csg[G::LOGP - 3], cw[G::LOGP - 3]);
            if constexpr (P >= 4) xstep4<2, true>(a, csg[G::LOGP - 2], cw[G::LOGP - 2]);
            if constexpr (P >= 2) xstep4<1, false>(a, csg[G::LOGP - 1], cw[G::LOGP - 1]);
#pragma unroll
            for (int j = 0; j < 4; j++) dz[zpadm<R, W>(Q + R * (t0 + j) + G::R2 * t1v)] = a[j];
        }
        block_sync_lds();
        // ---- untangle in place: pairs (k, M-k), k = tid + T j; bins 0 and M/2 have no partner
        //   S = Za + conj Zb, D = Za - conj Zb;  E = S/2, O = -i D/2, P = W^k O
        //   X[k] = E + P,  X[M-k] = conj(E - P)
        constexpr int NPAIR = R / 2;
        float lmax = -INFINITY, lmin = INFINITY, ls0 = 0.f, ls1 = 0.f;
        v2f za[NPAIR], zb[NPAIR], wv8[NPAIR];
#pragma unroll
        for (int j = 0; j < NPAIR; j++) {
            const int k = tid + T * j;
            const int km = (M - k) & (M - 1);                     // k = 0: Z[M] == Z[0]
            za[j] = dz[zpadm<R, W>(k)];
            zb[j] = dz[zpadm<R, W>(km)];
            wv8[j] = ((const v2f*)tw3)[k];
        }
        const v2f zc = dz[zpadm<R, W>(G::HALF)];
        const v2f khalf = pvxc::splat(0.5f), kmih = pvxc::mk(0.5f, -0.5f);
#pragma unroll
        for (int j = 0; j < NPAIR; j++) {
            const int k = tid + T * j;
            const int km = (M - k) & (M - 1);
            const v2f S = pvxc::add_conj(za[j], zb[j]);
            const v2f D = pvxc::sub_conj(za[j], zb[j]);
            const v2f O = pvxc::mul_swap(D, kmih);                // (D.y/2, -D.x/2)
            const v2f Pk = pvxc::cmul(O, wv8[j]);                 // W^k O
            const v2f x0 = __builtin_elementwise_fma(khalf, S, Pk);               // X[k] = S/2 + P
            v2f x1 = pvxc::fms_conj(khalf, S, Pk);                                // X[M-k] = conj(S/2 - P)
            int kk = km;
            if (j == 0) {
                // thread 0: k = 0 pairs with itself; its partner slot takes bin M/2: X[M/2] = conj(Z[M/2])
                if (tid == 0) { x1 = pvxc::mk(zc.x, -zc.y); kk = G::HALF; }
            }
            const float e0 = __builtin_fmaf(x0.x, x0.x, x0.y * x0.y), e1 = __builtin_fmaf(x1.x, x1.x, x1.y * x1.y);
            dz[zpadm<R, W>(k)] = x0;
            dz[zpadm<R, W>(kk)] = x1;
            if (with_mag) {
                // the peak search runs on |X|^2 (every test it makes is monotone in |X|; v_sqrt_f32 is a
                // quarter-rate instruction)
                y[k] = e0; y[kk] = e1;
                lmax = fmaxf(lmax, fmaxf(e0, e1)); lmin = fminf(lmin, fminf(e0, e1)); ls0 += e0; ls1 += e1;
            }
        }
        if (with_mag) {
            const float wm = wave_max(lmax), wn = wave_min(lmin);
            const double wsum = wave_sum((double)ls0 + (double)ls1);
            if (lane == 0) { pmax[wid] = wm; pmin[wid] = wn; psum[wid] = wsum; }
        }
        block_sync_lds();
        if (with_mag) {
            float mx = pmax[0], mn = pmin[0];
            double sm = psum[0];
#pragma unroll
            for (int w = 1; w < W; w++) { mx = fmaxf(mx, pmax[w]); mn = fminf(mn, pmin[w]); sm += psum[w]; }
            maxe = mx; mine = mn; tot = sm;
        }
    };

    // per-peak pass over the staged frames [0, ng): wave 0 only
    int LPF = 1;
    while (LPF < K && LPF < 64) LPF <<= 1;
    const int G_ = gs;                                            // frames staged per pass
    auto flush = [&](int ng) {
        wave_sync();
        // (the lane's group and ballot mask per flush, not as registers held through the frame loop: k_fused_rev.hip)
        const int lnf = fresh_lane();
        const int gl = lnf / LPF, e0 = lnf - gl * LPF;
        const unsigned long long gmask = (LPF == 64 ? ~0ull : ((1ull << LPF) - 1ull)) << (gl * LPF);
        const FusedParams* q = kargs;
        asm volatile("" : "+s"(q));                                  // loads through q stay here
        PeakConst pc;
        pc.fstep = q->fstep; pc.dt = q->dt; pc.nfft = G::N; pc.hop = q->hop; pc.wfbin = q->wfbin;
        const int g = gl;
        const bool gvalid = g < ng;
        const int cnt = gvalid ? cntv[g] : -1;
        const int64_t orow = gvalid ? (int64_t)orowv[g] : 0;
        double* of = q->f + orow * K;
        double* om = q->mag + orow * K;
        double* op = q->ph + orow * K;
        double* orp = q->realph + orow * K;
        double* ob = q->binno + orow * K;
        int nout = 0;
        for (int eb = 0; eb < K; eb += LPF) {
            const int e = eb + e0;
            bool valid = (cnt >= 0) && (e < cnt);
            int nbin = 0;
            PeakOut o;
            o.freq = 0.0; o.dfb = 0.0; o.thisph = 0.0; o.mag = 0.0; o.valid = false;
            if (valid) {
                nbin = sbin[g * kpad + e];
                const float* sv = sval + (size_t)(g * kpad + e) * 5;
                o = peak_math<float>(nbin, sv[0], sv[1], sv[2], sv[3], sv[4], pc);
                valid = o.valid;
            }
            const unsigned long long bal = __ballot(valid) & gmask;
            if (valid) {
                const int oi = nout + __popcll(bal & ((1ull << lnf) - 1ull));
                ob[oi] = (double)nbin;
                of[oi] = o.freq;
                om[oi] = o.mag;
                op[oi] = o.thisph;
                orp[oi] = o.thisph + kPi * o.dfb / pc.fstep;      // PV.py:207
            }
            nout += __popcll(bal);
        }
        if (cnt >= 0) {
            for (int j = nout + e0; j < K; j += LPF) {            // zero padding, PV.py:226-239
                ob[j] = 0.0; of[j] = 0.0; om[j] = 0.0; op[j] = 0.0; orp[j] = 0.0;
            }
            if (e0 == 0) {
                const int64_t fr = frmv[g];
                if (q->totalmag) q->totalmag[orow] = sqrt(totv[g]);                                  // PV.py:210
                if (q->t) q->t[orow] = ((double)(fr * (int64_t)pc.hop) + G::N / 2.0) / q->sr;        // PV.py:247
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
    int ng = 0;                                                   // staged frames (meaningful in wave 0)
    for (int64_t g = r0; g < r1; ++g) {
        if (gq == p.F) { gq = 0; gb += 1; } else { gq += 1; }
        const int64_t b = gb, q = gq;
        float maxe = 0.f, mine = 0.f;
        double tot = 0.0;
        spectrum(g, b, q, cur, true, maxe, mine, tot);
        if (q != 0) {                                             // block-uniform
            const int64_t orow = b * p.F + (q - 1);
            // PeakFinder(famp, npeaks, minrattomax) + filter_by_salience(rad=5)  (PV.py:175-178)
            const double minamp = (double)__builtin_amdgcn_sqrtf(maxe) * p.thr;   // PF.py:60
            // PF.py:69-70, 174 on the squared row: |X|^2 - mine > minamp^2 - mine; minamp == 0 means
            // minamp = miny there, the threshold is then exactly 0 (see k_fused.hip)
            const double th = (minamp != 0.0) ? minamp * minamp - (double)mine : 0.0;
            // every wave scans its share of the bins (8 consecutive bins per lane); the waves' counts are
            // exchanged through LDS and each wave writes its candidates straight to their place in the
            // one ascending list -- nothing to merge
            unsigned cm; int cpos;
            const int C_w = peak_block_masks<R, 0>(y, wid * G::SCAN, M, mine, th, lane, cm, cpos);
            if (lane == 0) Cw[wid] = C_w;
            block_sync_lds();
            int cbase = 0, C = 0;
#pragma unroll
            for (int w = 0; w < W; w++) { const int c = Cw[w]; cbase += (w < wid) ? c : 0; C += c; }
            peak_block_write<R, int>(ci, wid * G::SCAN, lane, cm, cbase + cpos, G::CAP + tid);
            // Dense frames (noise-like: hundreds of candidates): every wave first reduces ITS segment of the list to
            // its npks best -- the npks best of the whole row are among them, and ties are broken the same way
            // (score, then bin) -- and wave 0 only selects among the W * npks survivors.  Before, wave 0 ran the
            // radix select over the whole list alone while the others waited at the barrier.
            int Csel = C;                                             // length of the list wave 0 selects from
            const int* lsel = ci;
            if (C > 64 && C > K) {                                    // block-uniform
                wave_sync();                                          // a wave reads its own segment only
                int* mine_out = (int*)cs + wid * G::CAPW;
                int n_w = C_w;
                if (C_w > K) n_w = peak_radix_regs<(G::CAPW + 63) / 64, 0, int, (W <= 4)>(y, ci + cbase, mine_out, K, C_w, mine, lane);
                else { for (int c = lane; c < C_w; c += 64) mine_out[c] = ci[cbase + c]; }
                if (lane == 0) Nw[wid] = n_w;
                block_sync_lds();
                if (wid == 0) {
                    // survivors, segment after segment (= ascending bins), back into ci
                    int off = 0;
#pragma unroll
                    for (int w = 0; w < W; w++) {
                        const int n = Nw[w];
                        const int* src = (const int*)cs + w * G::CAPW;
                        for (int c = lane; c < n; c += 64) ci[off + c] = src[c];
                        off += n;
                    }
                    Csel = off;
                    wave_sync();
                }
            } else {
                block_sync_lds();
            }
            if (wid == 0) {
                // at most M/2 candidates: M/128 list entries per lane, ranked / radix-selected in registers
                const int nsel = peak_pick_regs<M / 128, 0, int>(y, lsel, sel, M, K, Csel, th, mine, lane);
                const bool use_prev0 = (p.prev0 != nullptr) && (orow == 0);
                int nk = 0;
                for (int eb = 0; eb < nsel; eb += 64) {
                    const int e = eb + lane;
                    int pb = 0;
                    if (e < nsel) pb = sel[e];
                    const bool keep = (p.rad <= 8) ? salient_groups<0>(y, M, sel, eb, nsel, p.rad, lane)
                                                   : ((e < nsel) && salient<float>(y, M, pb, p.rad));
                    const unsigned long long bal = __ballot(keep);
                    if (keep) {
                        const int slot = ng * kpad + nk + lane_prefix(bal);
                        const float2 c = cur[zpadm<R, W>(pb)];
                        float2 pv;
                        if (use_prev0) pv = make_float2((float)p.prev0[2 * pb], (float)p.prev0[2 * pb + 1]);
                        else pv = prv[zpadm<R, W>(pb)];
                        // PV.py:197-199: 3-bin energy, bin 0 excluded (1 <= pb <= M-2)
                        const float2 vm = cur[zpadm<R, W>(pb - 1)], vp = cur[zpadm<R, W>(pb + 1)];
                        const float em = (pb > 1) ? __builtin_fmaf(vm.x, vm.x, vm.y * vm.y) : 0.f;
                        const float s3 = (em + __builtin_fmaf(c.x, c.x, c.y * c.y)) + __builtin_fmaf(vp.x, vp.x, vp.y * vp.y);
                        sbin[slot] = pb;
                        float* sv = sval + (size_t)slot * 5;
                        sv[0] = c.x; sv[1] = c.y; sv[2] = pv.x; sv[3] = pv.y; sv[4] = s3;
                    }
                    nk += __popcll(bal);
                }
                if (lane == 0) { cntv[ng] = nk; frmv[ng] = (int)(q - 1); orowv[ng] = orow; totv[ng] = tot; }
                ng++;
                if (ng == G_) { flush(ng); ng = 0; }
            }
        }
        if (p.spec_out != nullptr && g == p.spec_row) {
#pragma unroll
            for (int j = 0; j < R; j++) {
                const float2 v = cur[zpadm<R, W>(tid + T * j)];
                p.spec_out[2 * (tid + T * j)] = v.x;
                p.spec_out[2 * (tid + T * j) + 1] = v.y;
            }
        }
        block_sync_lds();                                          // wave 0 is done with cur / prv / y
        float2* t = cur; cur = prv; prv = t;
    }
    if (wid == 0 && ng > 0) flush(ng);
}

template <int R, int W> int launch_mw(const FusedParams& p, int x_dtype, hipStream_t s) {
    using G = GeoMW<R, W>;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    const size_t lds = mw_lds_bytes<R, W>(p.K);
    if (lds > 160 * 1024) { pvx_set_error("nfft=%d npks=%d needs %zu bytes of LDS in the multi-wave fused kernel", G::N, p.K, lds); return PVX_ERR_UNSUPPORTED; }
    const bool al2 = (x_dtype == PVX_F32) && (p.hop % 2 == 0) && (p.sig_stride % 2 == 0) && (((uintptr_t)p.x) % 8 == 0);
    const void* fn = nullptr;
    switch (x_dtype) {
        case PVX_F32: fn = al2 ? (const void*)k_fused_mw<R, W, float, true> : (const void*)k_fused_mw<R, W, float, false>; break;
        case PVX_I16: fn = (const void*)k_fused_mw<R, W, int16_t, false>; break;
        default: pvx_set_error("the fused kernels take float32 or int16 samples (x_dtype %d: float64 is narrowed before the launch)", x_dtype); return PVX_ERR_INVALID;
    }
    if (lds > 64 * 1024) PVX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int blocks_per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, fn, 64 * W, lds) != hipSuccess || blocks_per_cu < 1)
        blocks_per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;
    if (blocks_per_cu * W > 8) blocks_per_cu = (8 / W) > 0 ? 8 / W : 1;     // 2 waves per SIMD
    int64_t nblocks = (int64_t)ncu * blocks_per_cu;
    if (p.blocks_override > 0) nblocks = p.blocks_override;
    // a short signal spreads over the chip one row per workgroup (each then transforms two rows: its own and the
    // one before it): what counts there is the latency of the launch, not the redundant transform
    const int64_t min_rows = 1;
    const int64_t maxb = p.total_rows / min_rows;
    if (nblocks > maxb) nblocks = maxb > 0 ? maxb : 1;
    dim3 grid((unsigned)nblocks), block(64 * W);
    switch (x_dtype) {
        case PVX_F32:
            if (al2) hipLaunchKernelGGL((k_fused_mw<R, W, float, true>), grid, block, lds, s, p);
            else hipLaunchKernelGGL((k_fused_mw<R, W, float, false>), grid, block, lds, s, p);
            break;
        default: hipLaunchKernelGGL((k_fused_mw<R, W, int16_t, false>), grid, block, lds, s, p); break;
    }
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}

}  // namespace

int pvx_fused_mw_supported(int nfft, int precision, int K) {
    if (precision != 32) return 0;
    switch (nfft) {
        case 2048: return mw_lds_bytes<8, 2>(K) <= 160 * 1024;
        case 4096: return mw_lds_bytes<8, 4>(K) <= 160 * 1024;
        case 8192: return mw_lds_bytes<8, 8>(K) <= 160 * 1024;
        default: return 0;
    }
}

int pvx_launch_fused_mw(const FusedParams& p, int nfft, int x_dtype, hipStream_t s) {
    if (p.total_rows <= 0) return PVX_OK;
    switch (nfft) {
        case 2048: return launch_mw<8, 2>(p, x_dtype, s);
        case 4096: return launch_mw<8, 4>(p, x_dtype, s);
        case 8192: return launch_mw<8, 8>(p, x_dtype, s);
        default: pvx_set_error("the multi-wave fused kernel does not handle nfft=%d", nfft); return PVX_ERR_UNSUPPORTED;
    }
}
