// k_fused_ring.hip -- the fused analysis stage (window + FFT + untangle + peaks, one wave64 per frame as in
// k_fused.hip) arranged so that TWO waves fit on every SIMD at nfft 2048:
//
//   PV.calc_fft_frame   pypevoc/PVAnalysis.py:150-158
//   PV.calc_pv_frame    pypevoc/PVAnalysis.py:160-211
//   PV.run_pv           pypevoc/PVAnalysis.py:213-264
//
// k_fused_pv<16> needs 285 registers (96 of them lane constants: window, two twiddle sets) and two private
// spectrum buffers per wave (current + previous frame, 17 KB) -- one wave per SIMD, which issues a vector
// instruction every 4 cycles at best and has nothing to cover its LDS / DPP / scalar round trips with.
// Here a workgroup of NW waves walks NW CONSECUTIVE frames per iteration:
//   * spectra live in a ring of NW + 1 LDS slots shared by the workgroup: the "previous spectrum" of
//     wave w's frame is the slot wave w-1 fills in the same iteration (wave 0: the slot wave NW-1
//     filled one iteration earlier), so a wave owns ONE slot instead of two;
//   * the two twiddle sets are tables in LDS shared by the workgroup (conflict-free 8-byte reads, 30 per
//     frame) instead of 64 registers per lane; the window stays in registers;
//   * the candidate list holds 16-bit bin numbers.
// nfft 2048: 8 waves = one workgroup per CU, ~150 KB of LDS, <= 256 registers -> 2 waves per SIMD.
// No workgroup barriers in the loop: wave w only ever depends on wave w-1 (its spectrum of this iteration is
// w's "previous spectrum"; its reads of the slot w overwrites next must be over), so the hand-off is two
// monotone progress counters per wave in LDS, polled by the one wave that needs them.  (With two
// s_barriers per iteration every wave waited for the slowest of eight twice per frame: 19 % of the wave
// cycles on the bench signal, more on noise where the peak search time varies from frame to frame.)
// Ring hazards: wave w writes slot (w - it) mod (NW + 1) in iteration it; the same slot held wave w-1's row
// of iteration it-1 (wave 0: wave NW-1's row of iteration it-2), last read by wave w-1 in its peak phase
// of that iteration and by wave w itself.
// The arithmetic of a frame is the instruction sequence of k_fused.hip (same FFT factorisation, same
// fmaf placement, same peak search), so results are bit-identical to it and independent of the launch
// geometry.
#include <stdlib.h>

#include "pvx_fft.h"

using namespace pvxw;
using namespace pvxf;

namespace {

constexpr int GFR = 8;              // frames staged before the per-peak pass

typedef unsigned short u16;

template <int R, int NW, int NG = 1> struct RingGeo {
    using G = Geo<R>;
    static constexpr int GW = NW / NG;                               // waves per ring (NG independent rings per workgroup)
    static constexpr int NS = NG * (GW + 1);                         // ring slots of the workgroup
    static constexpr int TW3N = (G::HALF + 8) & ~7;
    // block-shared tables (bytes)
    static constexpr size_t OFF_T1 = 0;                              // v2f [R][64]   W_M^(l q)
    static constexpr size_t OFF_T2 = OFF_T1 + (size_t)R * 64 * 8;    // v2f [R][P]    W_64^(l1 t2)
    static constexpr size_t OFF_TW3 = OFF_T2 + 64 * 8;               // v2f [TW3N]    W_nfft^k
    static constexpr size_t OFF_PROG = OFF_TW3 + (size_t)TW3N * 8;   // int [2][NW]   progress counters (flag sync)
    static constexpr size_t OFF_RING = OFF_PROG + 2 * NW * 4 + (16 - (2 * NW * 4) % 16) % 16;   // float2 [NS][BUFC]
    static constexpr size_t OFF_WAVE = OFF_RING + (size_t)NS * G::BUFC * 8;
    __host__ __device__ static size_t per_wave(int K) {
        const size_t kpad = (size_t)((K + 3) & ~3);
        const size_t gs = (size_t)staged_frames(K, GFR);
        size_t b = GFR * 8 * 2                                       // orow | tot
                 + (size_t)(G::M + 4 * R) * 4                        // y (padded, ymap<1>)
                 + gs * kpad * 5 * 4                                 // sval
                 + kpad * 4 + gs * kpad * 4                          // sel | sbin
                 + GFR * 4 * 2                                       // cnt | frm
                 + (size_t)(G::CAP + 64) * 2;                        // ci (u16) + 64 trash slots
        return (b + 15) & ~(size_t)15;
    }
    __host__ __device__ static size_t total(int K) { return OFF_WAVE + per_wave(K) * NW; }
};

template <int R, int NW, int NG, typename InT, bool AL2>
__global__ __launch_bounds__(64 * NW) void k_fused_ring(FusedParams p) {
    using G = Geo<R>;
    using RG = RingGeo<R, NW, NG>;
    constexpr int M = G::M, P = G::P, PITCH = G::PITCH, GW = RG::GW, NS = GW + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int K = p.K;
    const int kpad = (K + 3) & ~3;
    const int gs = staged_frames(K, GFR);

    v2f* const t1L = (v2f*)(smem + RG::OFF_T1);
    v2f* const t2L = (v2f*)(smem + RG::OFF_T2);
    v2f* const tw3 = (v2f*)(smem + RG::OFF_TW3);
    float2* const ring = (float2*)(smem + RG::OFF_RING);
    int* const Pfft = (int*)(smem + RG::OFF_PROG);                  // [w]: iterations whose spectrum wave w has completed
    int* const Ppk = Pfft + NW;                                     // [w]: iterations whose ring reads wave w has completed
    // per-wave region: everything whose size is known at compile time first, so that those arrays are one base
    // register plus immediate offsets (each runtime offset costs a scalar register across the whole frame loop)
    unsigned char* wb = smem + RG::OFF_WAVE + RG::per_wave(K) * wid;
    long long* const Lorow = (long long*)wb;
    double* const Ltot = (double*)(Lorow + GFR);
    float* const Ly = (float*)(Ltot + GFR);
    int* const Lcnt = (int*)(Ly + M + 4 * R);
    int* const Lfrm = Lcnt + GFR;
    u16* const Lci = (u16*)(Lfrm + GFR);
    int* const Lsel = (int*)(Lci + G::CAP + 64);
    int* const Lsbin = Lsel + kpad;
    float* const Lsval = (float*)(Lsbin + gs * kpad);

    // ---- block-shared tables
    {
        const v2f* tab = (const v2f*)p.twiddle;                     // W_nfft^j, j < nfft
        constexpr int NMASK = G::N - 1;
        for (int i = threadIdx.x; i < R * 64; i += 64 * NW) {
            const int q = i >> 6, l = i & 63;
            t1L[i] = tab[(2 * l * q) & NMASK];                      // W_M^(l q)
        }
        for (int i = threadIdx.x; i < 64; i += 64 * NW) t2L[i] = tab[((G::N / 64) * (i % P) * (i / P)) & NMASK];   // [t2][l1]
        for (int i = threadIdx.x; i <= G::HALF; i += 64 * NW) tw3[i] = tab[i];
        if (threadIdx.x < 2 * NW) Pfft[threadIdx.x] = 0;
    }
    __syncthreads();

    // ---- lane constants
    const int Q = lane / P, L1 = lane % P;
    float csg[G::LOGP > 0 ? G::LOGP : 1];
    v2f cw[G::LOGP > 0 ? G::LOGP : 1];
    {
        const float2* tab = (const float2*)p.twiddle;
        constexpr int NMASK = G::N - 1;
#pragma unroll
        for (int s = 0; s < G::LOGP; s++) {
            const int h = P >> (s + 1);
            const bool up = (L1 & h) != 0;
            csg[s] = up ? -1.f : 1.f;
            const float2 wvv = tab[((G::N / (2 * h)) * (L1 % h)) & NMASK];
            cw[s] = up ? pvxc::mk(wvv.x, wvv.y) : pvxc::mk(1.f, 0.f);
        }
    }
    int t1v = 0;                                                    // bitrev(l1)
#pragma unroll
    for (int b = 0; b < G::LOGP; b++) if (L1 & (1 << b)) t1v |= 1 << (G::LOGP - 1 - b);
    // the window stays in registers (there is room below 256): read from LDS it costs 16 reads that all
    // waves of the workgroup issue at the same moment, right after the barrier
    v2f wv[R];
#pragma unroll
    for (int r = 0; r < R; r++) wv[r] = ((const v2f*)p.win)[lane + 64 * r];
#pragma unroll
    for (int r = 0; r < R; r++) asm volatile("" : "+v"(wv[r]));

    // ---- rows of this workgroup: [r0, r1) plus the halo row r0 - 1 (spectrum only)
    // row indices fit 32 bits (the launcher checks): 64-bit scalar arithmetic in the frame loop costs SGPR pairs
    // NG independent rings per workgroup: waves [grp GW, (grp + 1) GW) walk their own contiguous share of the
    // workgroup's rows over their own GW + 1 slots.  Shorter hand-off chains: a wave that meets a slow frame
    // (dense candidates) holds up three followers instead of seven (+5 % on noise; nothing on the bench signal --
    // nor does running the SIMD partners, which then belong to different rings, deliberately out of step).
    const int grp = wid / GW, wl = wid - grp * GW;
    const int NB = (int)gridDim.x;
    const int R0 = (int)(p.total_rows * (int64_t)blockIdx.x / NB), R1 = (int)(p.total_rows * ((int64_t)blockIdx.x + 1) / NB);
    const int r0 = R0 + (int)((int64_t)(R1 - R0) * grp / NG), r1 = R0 + (int)((int64_t)(R1 - R0) * (grp + 1) / NG);
    if (r0 >= r1) return;                                           // uniform over the ring's waves
    float2* const ringg = ring + (size_t)grp * NS * G::BUFC;
    const int first = r0 - 1;
    const int nit = (r1 - first + GW - 1) / GW;
    const int Fi = (int)p.F;
    const int rows1 = Fi + 1;                                       // rows per signal

    // The output pointers and the constants of the per-peak arithmetic are only needed once per staging buffer
    // (every 8th frame): kept in scalar registers across the frame loop they are ~30 of the ~100 SGPRs the
    // compiler has, and what does not fit is spilled to VGPR lanes (v_writelane / v_readlane in the loop).
    // flush() re-reads them from the kernel argument segment instead.
    const FusedParams* const kargs = (const FusedParams*)__builtin_amdgcn_kernarg_segment_ptr();

    v2f raw[R];
#pragma unroll
    for (int r = 0; r < R; r++) raw[r] = pvxc::splat(0.f);
    // samples of global row gn = (bn, qn): nullptr when there is nothing to load
    auto row_src = [&](int gn, int bn, int qn) -> const InT* {
        if (gn < 0 || gn >= r1 || qn == 0) return nullptr;
        return (const InT*)p.x + (int64_t)bn * p.sig_stride + (int64_t)(qn - 1) * p.hop;
    };
    // The 16 loads of the next row are issued in four groups spread over the transform: issued together they
    // block the wave for ~1500 cycles (all waves of the workgroup queue 64 KB at the texture addresser at once)
    auto prefetch_part = [&](const InT* src, int part) {
        if (src == nullptr) return;
        constexpr int PR = R / 4;
#pragma unroll
        for (int r = part * PR; r < (part + 1) * PR; r++) {
            const InT* q = src + 2 * lane + 128 * r;
            if constexpr (AL2 && sizeof(InT) == 4) raw[r] = *(const v2f*)q;
            else raw[r] = pvxc::mk(ld1(q), ld1(q + 1));
        }
    };

    // spectrum of this wave's row into `dst` (zeros for a zero row) + |X|^2 -> Ly, wave-reduced max/min/energy;
    // issues the prefetch of row (gn, bn, qn) once the raw samples have been consumed
    auto spectrum = [&](bool zero_row, float2* dst, const InT* nsrc, float& maxe, float& mine, double& tot) {
        v2f z[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            z[r] = raw[r] * wv[r];
            asm volatile("" : "+v"(z[r]));                          // the multiply stays above the loads
        }
        __builtin_amdgcn_sched_barrier(0);
        prefetch_part(nsrc, 0);
        if (zero_row) {
            prefetch_part(nsrc, 1); prefetch_part(nsrc, 2); prefetch_part(nsrc, 3);
#pragma unroll
            for (int j = 0; j < G::BUFC / 64; j++) dst[lane + 64 * j] = make_float2(0.f, 0.f);
            wave_sync();
            return;
        }
        dft_regs<R>(z);                                             // stage 1
        __builtin_amdgcn_sched_barrier(0);
        prefetch_part(nsrc, 1);
        v2f* dz = (v2f*)dst;
#pragma unroll
        for (int q2 = 0; q2 < R; q2++) dz[q2 * PITCH + lane] = (q2 > 0) ? pvxc::cmul(z[q2], t1L[q2 * 64 + lane]) : z[q2];
        wave_sync();
#pragma unroll
        for (int l2 = 0; l2 < R; l2++) z[l2] = dz[Q * PITCH + L1 + P * l2];
        prefetch_part(nsrc, 2);
        wave_sync();
        dft_regs<R>(z);                                             // stage 2
        __builtin_amdgcn_sched_barrier(0);
        prefetch_part(nsrc, 3);
#pragma unroll
        for (int t0 = 0; t0 < R; t0 += 4) {
            v2f a[4];
#pragma unroll
            for (int j = 0; j < 4; j++) a[j] = (t0 + j > 0) ? pvxc::cmul(z[t0 + j], t2L[(t0 + j) * P + L1]) : z[t0 + j];
            if constexpr (G::LOGP >= 1) {
                if constexpr (P >= 16) xstep4<8, true>(a, csg[G::LOGP - 4], cw[G::LOGP - 4]);
                if constexpr (P >= 8) xstep4<4, true>(a, csg[G::LOGP - 3], cw[G::LOGP - 3]);
                if constexpr (P >= 4) xstep4<2, true>(a, csg[G::LOGP - 2], cw[G::LOGP - 2]);
                xstep4<1, false>(a, csg[G::LOGP - 1], cw[G::LOGP - 1]);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) dz[zpad<R>(Q + R * (t0 + j) + G::R2 * t1v)] = a[j];
        }
        wave_sync();
        // ---- untangle in place (k_fused.hip): pairs (k, M-k), k = lane + 64 j
        constexpr int NPAIR = R / 2;
        float lmax = -INFINITY, lmin = INFINITY, ls0 = 0.f, ls1 = 0.f;
        v2f za[NPAIR], zb[NPAIR], wv8[NPAIR];
#pragma unroll
        for (int j = 0; j < NPAIR; j++) {
            const int k = lane + 64 * j;
            const int km = (M - k) & (M - 1);
            za[j] = dz[zpad<R>(k)];
            zb[j] = dz[zpad<R>(km)];
            wv8[j] = tw3[k];
        }
        const v2f zc = dz[zpad<R>(G::HALF)];
        const v2f khalf = pvxc::splat(0.5f), kmih = pvxc::mk(0.5f, -0.5f);
#pragma unroll
        for (int j = 0; j < NPAIR; j++) {
            const int k = lane + 64 * j;
            const int km = (M - k) & (M - 1);
            const v2f S = pvxc::add_conj(za[j], zb[j]);
            const v2f D = pvxc::sub_conj(za[j], zb[j]);
            const v2f O = pvxc::mul_swap(D, kmih);
            const v2f Pk = pvxc::cmul(O, wv8[j]);
            const v2f x0 = __builtin_elementwise_fma(khalf, S, Pk);
            v2f x1 = pvxc::fms_conj(khalf, S, Pk);
            int kk = km;
            if (j == 0) {
                if (lane == 0) { x1 = pvxc::mk(zc.x, -zc.y); kk = G::HALF; }
            }
            const float e0 = __builtin_fmaf(x0.x, x0.x, x0.y * x0.y), e1 = __builtin_fmaf(x1.x, x1.x, x1.y * x1.y);
            dz[zpad<R>(k)] = x0;
            dz[zpad<R>(kk)] = x1;
            Ly[k + 4 * j] = e0; Ly[ymap<1>(kk)] = e1;
            lmax = fmaxf(lmax, fmaxf(e0, e1)); lmin = fminf(lmin, fminf(e0, e1)); ls0 += e0; ls1 += e1;
        }
        const double lsum = (double)ls0 + (double)ls1;
        maxe = wave_max(lmax);
        mine = wave_min(lmin);
        tot = wave_sum(lsum);
        wave_sync();
    };

    // per-peak pass over this wave's staged frames [0, ng)
    int LPF = 1;
    while (LPF < K && LPF < 64) LPF <<= 1;
    const int gl = lane / LPF, e0 = lane - gl * LPF;
    const unsigned long long gmask = (LPF == 64 ? ~0ull : ((1ull << LPF) - 1ull)) << (gl * LPF);
    auto flush = [&](int ng) {
        wave_sync();
        const FusedParams* q = kargs;
        asm volatile("" : "+s"(q));                                  // loads through q stay here
        PeakConst pc;
        pc.fstep = q->fstep; pc.dt = q->dt; pc.nfft = G::N; pc.hop = q->hop; pc.wfbin = q->wfbin;
        const int g = gl;
        const bool gvalid = g < ng;
        const int cnt = gvalid ? Lcnt[g] : -1;
        const int64_t orow = gvalid ? (int64_t)Lorow[g] : 0;
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
                nbin = Lsbin[g * kpad + e];
                const float* sv = Lsval + (size_t)(g * kpad + e) * 5;
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
                orp[oi] = o.thisph + kPi * o.dfb / pc.fstep;          // PV.py:207
            }
            nout += __popcll(bal);
        }
        if (cnt >= 0) {
            for (int j = nout + e0; j < K; j += LPF) {                // zero padding, PV.py:226-239
                ob[j] = 0.0; of[j] = 0.0; om[j] = 0.0; op[j] = 0.0; orp[j] = 0.0;
            }
            if (e0 == 0) {
                const int64_t fr = Lfrm[g];
                if (q->totalmag) q->totalmag[orow] = sqrt(Ltot[g]);                                   // PV.py:210
                if (q->t) q->t[orow] = ((double)(fr * (int64_t)pc.hop) + G::N / 2.0) / q->sr;         // PV.py:247
            }
        }
        wave_sync();
    };

    // ---- (signal b, row-in-signal q) of this wave's first row g = first + wid; rows advance by NW
    int g = first + wl, gb, gq;
    if (g >= 0) { gb = g / rows1; gq = g - gb * rows1; }           // the only division
    else { gb = -1; gq = Fi; }                                      // "row -1": a zero row
    auto advance = [&](int& b, int& q) {
        q += GW;
        while (q > Fi) { q -= rows1; b += 1; }
    };
    if (g < r1) { const InT* s0 = row_src(g, gb, gq); prefetch_part(s0, 0); prefetch_part(s0, 1); prefetch_part(s0, 2); prefetch_part(s0, 3); }
    int slot = wl;                                                  // (g - first) mod NS
    int ng = 0;
    // Pairwise hand-off: wave w only ever depends on wave w-1 (wave 0: on wave NW-1 one iteration back),
    // through two monotone counters per wave in LDS.  LDS operations of one wave execute in order, so "data
    // stores, then counter store" needs no fence, and a reader that has seen the counter sees the data.
    // (the poll is bounded -- about a second -- so that a broken hand-off could only ever give wrong
    // numbers, never waves that do not finish)
    auto wait_ge = [&](const int* f, int need) {
        for (int spin = 0; spin < (1 << 23); ++spin) {
            if (__builtin_amdgcn_readfirstlane(*(const volatile int*)f) >= need) break;
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    };
    auto post = [&](int* f, int v) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) *(volatile int*)f = v;
    };
    const int wprev = grp * GW + ((wl == 0) ? GW - 1 : wl - 1);
    for (int it = 0; it < nit; ++it) {
        int bn = gb, qn = gq;
        advance(bn, qn);
        const bool active = g < r1;                                 // wave-uniform
        float2* cur = ringg + (size_t)slot * G::BUFC;
        float2* prv = ringg + (size_t)(slot == 0 ? NS - 1 : slot - 1) * G::BUFC;
        const bool zero_row = (g < 0) || (gq == 0);
        float maxe = 0.f, mine = 0.f;
        double tot = 0.0;
        // this wave's slot was read by wave w-1 (its own spectrum, iteration it-1; wave 0: wave NW-1, it-2)
        if (it > 0) wait_ge(Ppk + wprev, wl == 0 ? it - 1 : it);
        if (active) spectrum(zero_row, cur, row_src(g + GW, bn, qn), maxe, mine, tot);
        post(Pfft + wid, it + 1);
        if (active && !zero_row && g >= r0) {
            // the previous row's spectrum: wave w-1, this iteration (wave 0: wave NW-1, iteration it-1)
            wait_ge(Pfft + wprev, wl == 0 ? it : it + 1);
            const int64_t orow = (int64_t)gb * Fi + (gq - 1);
            // PeakFinder(famp, npeaks, minrattomax) + filter_by_salience(rad=5)  (PV.py:175-178); see k_fused.hip
            const float maxy = __builtin_amdgcn_sqrtf(maxe);
            const double minamp = (double)maxy * p.thr;             // PF.py:60
            const double th = (minamp != 0.0) ? minamp * minamp - (double)mine : 0.0;
            const bool use_prev0 = (p.prev0 != nullptr) && (orow == 0);
            int nk = 0;
            // candidate list (ascending bins) -> Lci
            const int C = peak_scan_block<R, u16>(Ly, mine, th, Lci, G::CAP, lane);
            wave_sync();
            if (C <= 64 && p.rad <= 5 && !(th < 0.0 && C < K)) {
                // ---- at most one candidate per lane (every frame of music): lane c owns candidate c and fetches
                // in ONE LDS round trip all that the rest of the frame needs of it -- its score, the 2*rad
                // neighbours of the salience test (PF.py:126-134; indices clamped into the window like `salient`),
                // the spectrum around it and the previous spectrum at it -- instead of one round trip each for
                // ranking, the selected-bin list, the salience test and the staging.  Same selection as
                // peak_pick_regs: rank by (score desc, bin asc), the npeaks best; then the salience filter.
                const bool has = lane < C;
                const int pb = has ? (int)Lci[lane] : 1;
                const int rad = p.rad;
                const int lo = pb - rad > 1 ? pb - rad : 1;
                int hi = pb + rad < M ? pb + rad : M;
                hi = hi > M - 1 ? M - 1 : hi;
                const float v = Ly[ymap<1>(pb)];
                float nb[10];
#pragma unroll
                for (int d = 1; d <= 5; d++) {
                    const int dd = d > rad ? rad : d;
                    int j0 = pb - dd, j1 = pb + dd;
                    j0 = j0 < lo ? lo : j0;
                    j1 = j1 > hi ? hi : j1;
                    nb[2 * d - 2] = Ly[ymap<1>(j0)];
                    nb[2 * d - 1] = Ly[ymap<1>(j1)];
                }
                const float2 c = cur[zpad<R>(pb)];
                const float2 vm = cur[zpad<R>(pb - 1)], vp = cur[zpad<R>(pb + 1)];
                float2 pv;
                if (use_prev0) pv = make_float2((float)p.prev0[2 * pb], (float)p.prev0[2 * pb + 1]);
                else pv = prv[zpad<R>(pb)];
                int bad = 0;
#pragma unroll
                for (int d = 0; d < 10; d++) bad |= (int)(nb[d] > v);
                bool take = has;
                if (C > K) {
                    const unsigned mykey = has ? __float_as_uint(v - mine) : 0u;     // scores >= 0: bits order like values
                    int rank = 0;
                    for (int j = 0; j < C; ++j) {
                        const unsigned kj = (unsigned)__builtin_amdgcn_readlane((int)mykey, j);
                        rank += (kj > mykey || (kj == mykey && j < lane)) ? 1 : 0;
                    }
                    take = has && (rank < K);
                }
                const bool keep = take && (rad < 0 || bad == 0);
                const unsigned long long bal = __ballot(keep);
                if (keep) {
                    const int sl = ng * kpad + lane_prefix(bal);
                    // PV.py:197-199: 3-bin energy, bin 0 excluded (1 <= pb <= M-2)
                    const float em = (pb > 1) ? __builtin_fmaf(vm.x, vm.x, vm.y * vm.y) : 0.f;
                    const float s3 = (em + __builtin_fmaf(c.x, c.x, c.y * c.y)) + __builtin_fmaf(vp.x, vp.x, vp.y * vp.y);
                    Lsbin[sl] = pb;
                    float* sv = Lsval + (size_t)sl * 5;
                    sv[0] = c.x; sv[1] = c.y; sv[2] = pv.x; sv[3] = pv.y; sv[4] = s3;
                }
                nk = __popcll(bal);
            } else {
            // radix select inlined: the call of the out-of-line version saves / restores ~100 scalar registers per frame
            const int nsel = peak_pick_regs<R / 2, 1, u16, true>(Ly, Lci, Lsel, M, K, C, th, mine, lane);
            for (int eb = 0; eb < nsel; eb += 64) {
                const int e = eb + lane;
                int pb = 0;
                if (e < nsel) pb = Lsel[e];
                const bool keep = (p.rad <= 8) ? salient_groups<1>(Ly, M, Lsel, eb, nsel, p.rad, lane)
                                               : ((e < nsel) && salient<float, 1>(Ly, M, pb, p.rad));
                const unsigned long long bal = __ballot(keep);
                if (keep) {
                    const int sl = ng * kpad + nk + lane_prefix(bal);
                    const float2 c = cur[zpad<R>(pb)];
                    float2 pv;
                    if (use_prev0) pv = make_float2((float)p.prev0[2 * pb], (float)p.prev0[2 * pb + 1]);
                    else pv = prv[zpad<R>(pb)];
                    // PV.py:197-199: 3-bin energy, bin 0 excluded (1 <= pb <= M-2)
                    const float2 vm = cur[zpad<R>(pb - 1)], vp = cur[zpad<R>(pb + 1)];
                    const float em = (pb > 1) ? __builtin_fmaf(vm.x, vm.x, vm.y * vm.y) : 0.f;
                    const float s3 = (em + __builtin_fmaf(c.x, c.x, c.y * c.y)) + __builtin_fmaf(vp.x, vp.x, vp.y * vp.y);
                    Lsbin[sl] = pb;
                    float* sv = Lsval + (size_t)sl * 5;
                    sv[0] = c.x; sv[1] = c.y; sv[2] = pv.x; sv[3] = pv.y; sv[4] = s3;
                }
                nk += __popcll(bal);
            }
            }
            if (lane == 0) { Lcnt[ng] = nk; Lfrm[ng] = (int)(gq - 1); Lorow[ng] = orow; Ltot[ng] = tot; }
            ng++;
            if (ng == gs) { flush(ng); ng = 0; }
        }
        if (active && p.spec_out != nullptr && g == p.spec_row) {
#pragma unroll
            for (int j = 0; j < R; j++) {
                const float2 v = cur[zpad<R>(lane + 64 * j)];
                p.spec_out[2 * (lane + 64 * j)] = v.x;
                p.spec_out[2 * (lane + 64 * j) + 1] = v.y;
            }
        }
        post(Ppk + wid, it + 1);                                    // this wave's reads of the ring slots are over
        g += GW; gb = bn; gq = qn;
        slot = (slot == 0) ? NS - 1 : slot - 1;                     // (slot + NW) mod (NW + 1)
    }
    if (ng > 0) flush(ng);
}

template <int R, int NW, int NG> int launch_ring(const FusedParams& p, int x_dtype, hipStream_t s) {
    using RG = RingGeo<R, NW, NG>;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    if (p.total_rows >= 0x7fffff00LL) { pvx_set_error("the ring kernel indexes rows in 32 bits (%lld rows)", (long long)p.total_rows); return PVX_ERR_UNSUPPORTED; }
    const size_t lds = RG::total(p.K);
    if (lds > 160 * 1024) { pvx_set_error("nfft=%d npks=%d needs %zu bytes of LDS in the ring kernel", Geo<R>::N, p.K, lds); return PVX_ERR_UNSUPPORTED; }
    const bool al2 = (x_dtype == PVX_F32) && (p.hop % 2 == 0) && (p.sig_stride % 2 == 0) && (((uintptr_t)p.x) % 8 == 0);
    const void* fn = nullptr;
    switch (x_dtype) {
        case PVX_F32: fn = al2 ? (const void*)k_fused_ring<R, NW, NG, float, true> : (const void*)k_fused_ring<R, NW, NG, float, false>; break;
        case PVX_I16: fn = (const void*)k_fused_ring<R, NW, NG, int16_t, false>; break;
        default: pvx_set_error("the fused kernels take float32 or int16 samples (x_dtype %d: float64 is narrowed before the launch)", x_dtype); return PVX_ERR_INVALID;
    }
    if (lds > 64 * 1024) PVX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int blocks_per_cu = (int)((160 * 1024) / lds);
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    if (blocks_per_cu * NW > 16) blocks_per_cu = (16 / NW) > 0 ? 16 / NW : 1;
    int64_t nblocks = (int64_t)ncu * blocks_per_cu;
    if (p.blocks_override > 0) nblocks = p.blocks_override;
    // a workgroup handles NW rows per iteration plus one halo row: at least two iterations' worth each
    const int64_t maxb = p.total_rows / (2 * NW);
    if (nblocks > maxb) nblocks = maxb > 0 ? maxb : 1;
    dim3 grid((unsigned)nblocks), block(64 * NW);
    switch (x_dtype) {
        case PVX_F32:
            if (al2) hipLaunchKernelGGL((k_fused_ring<R, NW, NG, float, true>), grid, block, lds, s, p);
            else hipLaunchKernelGGL((k_fused_ring<R, NW, NG, float, false>), grid, block, lds, s, p);
            break;
        default: hipLaunchKernelGGL((k_fused_ring<R, NW, NG, int16_t, false>), grid, block, lds, s, p); break;
    }
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}

}  // namespace

int pvx_fused_ring_supported(int nfft, int precision, int K) {
    if (precision != 32) return 0;
    switch (nfft) {
        case 2048: return RingGeo<16, 8, 1>::total(K) <= 160 * 1024;
        case 1024: return RingGeo<8, 12>::total(K) <= 160 * 1024;
        case 512: return RingGeo<4, 12>::total(K) <= 160 * 1024;
        default: return 0;
    }
}

int pvx_launch_fused_ring(const FusedParams& p, int nfft, int x_dtype, hipStream_t s) {
    if (p.total_rows <= 0) return PVX_OK;
    // nfft 2048: two rings of four waves per workgroup where the LDS holds the tenth slot next to the staging
    // (npks <= 64); PVX_RING_GROUPS=1 forces the single ring (A/B; same results)
    static const int groups = [] { const char* e = getenv("PVX_RING_GROUPS"); return e ? atoi(e) : 2; }();
    switch (nfft) {
        case 2048: return (groups == 2 && RingGeo<16, 8, 2>::total(p.K) <= 160 * 1024) ? launch_ring<16, 8, 2>(p, x_dtype, s)
                                                                                      : launch_ring<16, 8, 1>(p, x_dtype, s);
        case 1024: return launch_ring<8, 12, 1>(p, x_dtype, s);
        case 512: return launch_ring<4, 12, 1>(p, x_dtype, s);
        default: pvx_set_error("the ring kernel does not handle nfft=%d", nfft); return PVX_ERR_UNSUPPORTED;
    }
}
