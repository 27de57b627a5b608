// k_pv_team.hip -- the float64 analysis stage at nfft 4096 / 8192 in ONE launch with no spectrum row in HBM: a frame is transformed
// by a TEAM of S waves (k_stft.hip's split transform: S interleaved 2048-point real transforms, joined in LDS), the team walks a
// contiguous range of rows in DESCENDING order and the row at hand stays on chip (k_pv_rev.hip's scheme, one level up).
//
//   PV.calc_fft_frame   pypevoc/PVAnalysis.py:150-158   (float64, as the reference computes it)
//   PV.calc_pv_frame    pypevoc/PVAnalysis.py:160-211   (abs, PeakFinder + filter_by_salience, dphase2freq, 3-bin energy)
//   PV.run_pv           pypevoc/PVAnalysis.py:213-264   (frame loop, zero-padded packing)
//
// Before: k_stft_split wrote every row (32 / 64 KB) to a workspace and k_phase_peaks streamed it back (nfft 4096: 847 MB out and
// 847 MB in per 25 838 rows, two launches).  Here, per row: sub-transforms -> join -> untangle with the bins left in the waves'
// REGISTERS and |X|^2 in the LDS the transform occupied -> every wave scans its own segment of the row for candidate maxima ->
// wave 0 selects the npks best of the merged lists and tests their salience (pvx_wave.h, the general path's functions on the
// general path's row) -> the bins go from the registers to the same LDS bytes in natural order -> wave 0 reads the kept peaks'
// values and 3-bin energies there, and the values the frame above's staged peaks were waiting for (PV.py:171, 190).  Staging and
// the per-peak pass are k_pv_rev's (64 slots per team, the slots' float64 values in a per-team block of global memory: LDS is the
// regions' and the tables').  Eight workgroup barriers per row.  The rows a team may be given: the sliding-window hops (nfft/4,
// nfft/2), npks <= 64; anything else keeps the two-kernel path.
// The five result arrays are bit-identical to the two-kernel path's (same transform, same |X|^2, same search); totalmag sums the
// row in k_phase_peaks' order.
#include <stdlib.h>

#include <type_traits>

#include "pvx_stft.h"

using namespace pvxw;
using namespace pvxf;
using namespace pvxs;

namespace {

#ifndef PVX_TEAM_F32_TEAMS
#define PVX_TEAM_F32_TEAMS 2          // nfft 4096, float32 / int16 samples: teams per workgroup (3 fit the LDS, at 256 registers a wave: 69 of them in scratch)
#endif
constexpr int GFV = 8;                // frames staged before the per-peak pass
constexpr int kSlots = 64;            // staging slots: one per lane of the per-peak pass

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS: [WL: window T[N]] | t1 cx[R][64] | t2 cx[64] | per team: S regions cx[BUFC] | staging (bins, frame records, the search's result)
template <int R, int S, bool WL> struct TeamGeo {
    using T = double;
    using G1 = StftGeo<R, T>;
    static constexpr int M1 = G1::M, M = M1 * S, N = 2 * M, LOGM1 = ilog2(M1);
    static constexpr int BUFC = G1::BUFC;
    static constexpr int SEG = M / S, CAPW = SEG / 2 + 4, CAPM = M / 2 + 4;
    static constexpr size_t OFF_WIN = 0;
    static constexpr size_t OFF_T1 = OFF_WIN + (WL ? (size_t)N * sizeof(T) : 0);
    static constexpr size_t OFF_T2 = OFF_T1 + (size_t)R * 64 * 2 * sizeof(T);
    static constexpr size_t OFF_BUF = OFF_T2 + 64 * 2 * sizeof(T);
    static constexpr size_t REGIONS = (size_t)S * BUFC * 2 * sizeof(T);
    // inside the regions between the untangle and the dump: y [M] T | ciw [S][CAPW] u16 | cim [CAPM] u16 | part [S][4] f64 | sel [64] int
    static constexpr size_t Y_CIW = (size_t)M * sizeof(T);
    static constexpr size_t Y_CIM = Y_CIW + (((size_t)S * CAPW * 2 + 15) & ~(size_t)15);
    static constexpr size_t Y_PART = Y_CIM + (((size_t)CAPM * 2 + 15) & ~(size_t)15);
    static constexpr size_t Y_SEL = Y_PART + (size_t)S * 4 * 8;                              // sel [64] int | selw [S][64] int
    static_assert(Y_SEL + (size_t)(S + 1) * 64 * 4 <= REGIONS, "the peak search fits the regions");
    // behind the regions, per team: orow [GFV] i64 | tot [GFV] f64 | cnt, off, frm [GFV] int | bin [kSlots] int
    static constexpr size_t T_OROW = REGIONS;
    static constexpr size_t T_TOT = T_OROW + GFV * 8;
    static constexpr size_t T_CNT = T_TOT + GFV * 8;
    static constexpr size_t T_OFF = T_CNT + GFV * 4;
    static constexpr size_t T_FRM = T_OFF + GFV * 4;
    static constexpr size_t T_BIN = T_FRM + GFV * 4;
    static constexpr size_t PER_TEAM = (T_BIN + kSlots * 4 + 15) & ~(size_t)15;
    __host__ __device__ static constexpr size_t total(int teams) { return OFF_BUF + PER_TEAM * (size_t)teams; }
};

typedef __attribute__((address_space(1))) double gdouble;

template <int R, int S, typename InT, int H, int TEAMS>
__global__ __launch_bounds__(64 * S * TEAMS) void k_pv_team(PvRevParams p) {
    using T = double;
    constexpr bool WL = (S == 2) || (sizeof(InT) == 8);              // the window in LDS (else: its values of the lane's pairs in registers)
    using G = TeamGeo<R, S, WL>;
    using G1 = StftGeo<R, T>;
    constexpr int M1 = G::M1, M = G::M, P = G1::P, PITCH = G1::PITCH, SEG = G::SEG, CAPW = G::CAPW;
    constexpr int NMASK = G::N - 1;
    static_assert(S == 2 || S == 4, "radix of the join");
    static_assert(H > 0, "the sliding-window hops");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int team = wid / S, sub = wid - team * S;
    const int K = p.K;
    cx<T>* const t1L = (cx<T>*)(smem + G::OFF_T1);
    cx<T>* const t2L = (cx<T>*)(smem + G::OFF_T2);
    unsigned char* const tbase = smem + G::OFF_BUF + (size_t)team * G::PER_TEAM;
    cx<T>* const buf = (cx<T>*)tbase;                                // the team's S regions
    cx<T>* const dz = buf + (size_t)sub * G::BUFC;                   // this wave's region
    T* const yL = (T*)tbase;
    unsigned short* const ciW = (unsigned short*)(tbase + G::Y_CIW);
    unsigned short* const ciM = (unsigned short*)(tbase + G::Y_CIM);
    double* const partL = (double*)(tbase + G::Y_PART);
    int* const sel = (int*)(tbase + G::Y_SEL);
    long long* const Lorow = (long long*)(tbase + G::T_OROW);
    double* const Ltot = (double*)(tbase + G::T_TOT);
    int* const Lcnt = (int*)(tbase + G::T_CNT);
    int* const Loff = (int*)(tbase + G::T_OFF);
    int* const Lfrm = (int*)(tbase + G::T_FRM);
    int* const Lbin = (int*)(tbase + G::T_BIN);
    gdouble* const Gval = (gdouble*)p.stage + ((size_t)blockIdx.x * TEAMS + team) * (kSlots * 5);
    const cx<T>* const tab = (const cx<T>*)p.twiddle;                 // W_N^j
    const T* const winG = (const T*)p.win;
    T* const winL = (T*)(smem + G::OFF_WIN);

    // bin k of the joined row / of the natural-order spectrum in the team's regions
    auto ZA = [](int k) -> int { return (k >> G::LOGM1) * G::BUFC + zpadT<R, T>(k & (M1 - 1)); };

    // ---- rows of this team: [r0, r1) walked downwards, then row r0 - 1 (spectrum only); the teams of a workgroup keep step
    const int64_t TT = (int64_t)gridDim.x * TEAMS, tv = (int64_t)blockIdx.x * TEAMS + team, nrows = p.row_end - p.row_begin;
    const int r0 = (int)(p.row_begin + nrows * tv / TT), r1 = (int)(p.row_begin + nrows * (tv + 1) / TT);
    int trips = 0;                                                    // of the workgroup: its longest team's rows + 1
#pragma unroll
    for (int t2 = 0; t2 < TEAMS; t2++) {
        const int64_t tv2 = (int64_t)blockIdx.x * TEAMS + t2;
        const int n2 = (int)(nrows * (tv2 + 1) / TT - nrows * tv2 / TT);
        if (n2 > 0 && n2 + 1 > trips) trips = n2 + 1;
    }
    const bool idle_team = r0 >= r1;
    const int glast = r0 - 1;
    const int Fi = (int)p.F, rows1 = Fi + 1;
    const int hopi = p.hop;

    using RawT = typename std::conditional<(sizeof(InT) == 8), double, float>::type;
    RawT raw[2 * R];
#pragma unroll
    for (int r = 0; r < 2 * R; r++) raw[r] = (RawT)0;
    const int lofs = 2 * S * lane + 2 * sub;                          // element j = lane + 64 r of this wave: the sample pair at 2 (S j + sub)
    auto row_ptr = [&](int b, int q) -> const InT* { return (const InT*)p.x + (int64_t)b * p.sig_stride + (int64_t)(q - 1) * hopi; };
    auto load_pair = [&](const InT* src, int r) {
        const InT* q = src + lofs + 128 * S * r;
        raw[2 * r] = (RawT)q[0]; raw[2 * r + 1] = (RawT)q[1];
    };
    auto load_all = [&](const InT* src) {
#pragma unroll
        for (int r = 0; r < R; r++) load_pair(src, r);
    };

    int g = r1 - 1, gq = 0, orow = 0;
    const InT* csrc = (const InT*)p.x;
    {
        const int gb = g >= 0 ? g / rows1 : 0;
        gq = g - gb * rows1;
        if (g >= 0 && gq >= 1) { csrc = row_ptr(gb, gq); orow = gb * Fi + gq - 1; }
    }
    if (!idle_team && g >= 0 && gq >= 1) load_all(csrc);

    if constexpr (WL) for (int i = threadIdx.x; i < G::N; i += 64 * S * TEAMS) winL[i] = winG[i];
    for (int i = threadIdx.x; i < R * 64; i += 64 * S * TEAMS) t1L[i] = tab[(2 * S * (i & 63) * (i >> 6)) & NMASK];          // W_M1^(l q)
    for (int i = threadIdx.x; i < 64; i += 64 * S * TEAMS) t2L[i] = tab[((G::N / 64) * (i % P) * (i / P)) & NMASK];           // [t2][l1]
    T wn[WL ? 2 : 2 * R];                                             // (!WL) the window at the lane's pairs: the window does not slide
    if constexpr (!WL) {
#pragma unroll
        for (int r = 0; r < R; r++) { wn[2 * r] = winG[lofs + 128 * S * r]; wn[2 * r + 1] = winG[lofs + 128 * S * r + 1]; }
    }
    __syncthreads();

    const int Q = lane / P, L1 = lane % P;
    T csg[G1::LOGP > 0 ? G1::LOGP : 1];
    cx<T> cw[G1::LOGP > 0 ? G1::LOGP : 1];
#pragma unroll
    for (int st = 0; st < G1::LOGP; st++) {
        const int h = P >> (st + 1);
        const bool up = (L1 & h) != 0;
        csg[st] = up ? (T)-1 : (T)1;
        const cx<T> wv = tab[((G::N / (2 * h)) * (L1 % h)) & NMASK];
        cw[st] = up ? wv : mkc<T>((T)1, (T)0);
    }
    int t1v = 0;
#pragma unroll
    for (int b = 0; b < G1::LOGP; b++) if (L1 & (1 << b)) t1v |= 1 << (G1::LOGP - 1 - b);

    PeakConst pc;
    pc.fstep = p.fstep; pc.dt = p.dt; pc.nfft = G::N; pc.hop = p.hop; pc.wfbin = p.wfbin;

    // ---- per-peak phase-vocoder arithmetic on the staged frames [0, ng): lane l of wave 0 takes slot l (k_pv_rev.hip)
    auto flush = [&](int ng) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's staged values are in L2
        wave_sync();
        const int lnf = fresh_lane();
        static_assert(GFV == 8, "two 16-byte reads per table");
        const int4 oa = *(const int4*)Loff, ob4 = *((const int4*)Loff + 1), ca = *(const int4*)Lcnt, cb4 = *((const int4*)Lcnt + 1);
        const int offs[8] = {oa.x, oa.y, oa.z, oa.w, ob4.x, ob4.y, ob4.z, ob4.w}, cnts[8] = {ca.x, ca.y, ca.z, ca.w, cb4.x, cb4.y, cb4.z, cb4.w};
        int top = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { if (j == ng - 1) top = offs[j] + cnts[j]; }
        bool valid = lnf < top;
        int gi = 0, start = 0, cnt = cnts[0];
#pragma unroll
        for (int j = 1; j < 8; j++) { if (j < ng && offs[j] <= lnf) { gi = j; start = offs[j]; cnt = cnts[j]; } }
        const int64_t orw = (int64_t)Lorow[gi];
        int nbin = 0;
        PeakOut o;
        o.freq = 0.0; o.dfb = 0.0; o.thisph = 0.0; o.mag = 0.0; o.valid = false;
        if (valid) {
            nbin = Lbin[lnf];
            double v[5];
#pragma unroll
            for (int i = 0; i < 5; i++) v[i] = __hip_atomic_load((const double*)(Gval + lnf * 5 + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            o = peak_math<T>(nbin, v[0], v[1], v[2], v[3], v[4], pc);
            valid = o.valid;
        }
        const unsigned long long gm = (cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull)) << start;
        const unsigned long long ball = __ballot(valid);
        if (valid) {
            const int oi = __popcll(ball & gm & ((1ull << lnf) - 1ull));
            ((gdouble*)p.binno + orw * K)[oi] = (double)nbin;
            ((gdouble*)p.f + orw * K)[oi] = o.freq;
            ((gdouble*)p.mag + orw * K)[oi] = o.mag;
            ((gdouble*)p.ph + orw * K)[oi] = o.thisph;
            ((gdouble*)p.realph + orw * K)[oi] = o.thisph + kPi * o.dfb / pc.fstep;        // PV.py:207
        }
        {
            const int g2 = lnf >> 3, c2 = lnf & 7;
            int o2 = 0, n2 = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) { if (j == g2 && j < ng) { o2 = offs[j]; n2 = cnts[j]; } }
            if (g2 < ng) {
                const unsigned long long gm2 = (n2 >= 64 ? ~0ull : ((1ull << n2) - 1ull)) << o2;
                const int nout2 = __popcll(ball & gm2);
                const int64_t orow2 = (int64_t)Lorow[g2];
                gdouble* of2 = (gdouble*)p.f + orow2 * K; gdouble* om2 = (gdouble*)p.mag + orow2 * K; gdouble* op2 = (gdouble*)p.ph + orow2 * K;
                gdouble* orp2 = (gdouble*)p.realph + orow2 * K; gdouble* ob2 = (gdouble*)p.binno + orow2 * K;
                for (int j = nout2 + c2; j < K; j += 8) { ob2[j] = 0.0; of2[j] = 0.0; om2[j] = 0.0; op2[j] = 0.0; orp2[j] = 0.0; }
                if (c2 == 0) {
                    const int64_t fr = Lfrm[g2];
                    if (p.totalmag) ((gdouble*)p.totalmag)[orow2] = sqrt(Ltot[g2]);                                   // PV.py:210
                    if (p.t) ((gdouble*)p.t)[orow2] = ((double)(fr * (int64_t)pc.hop) + G::N / 2.0) / p.sr;           // PV.py:247
                }
            }
        }
        wave_sync();
    };

    int ng = 0, nst = 0;                                              // (wave 0 of the team) staged frames, staged slots
    bool pend = false, pend_prev0 = false, pz = false;
    int pend_off = 0, pend_nk = 0;
    const int spec_g = (p.spec_out != nullptr && p.spec_row >= 0 && p.spec_row < 0x7fffffffLL) ? (int)p.spec_row : -0x7fffffff;
    for (int it = 0; it < trips; ++it, --g) {
        const bool have = !idle_team && g >= glast;                   // (a team whose rows are done keeps step at the barriers)
        const int qn = gq >= 1 ? gq - 1 : Fi;
        const bool zero_row = (g < 0) || (gq == 0);
        const bool real = have && !zero_row;
        const bool with_peaks = real && g >= r0;
        const InT* nsrc = nullptr;
        int norow = orow - 1;
        if (have && g - 1 >= glast && g - 1 >= 0 && qn >= 1) {
            if (gq >= 2) nsrc = csrc - hopi;
            else { const int bn = (g - 1) / rows1; nsrc = row_ptr(bn, qn); norow = bn * Fi + qn - 1; }
        }
        if (have) {
            if (!zero_row && pz) load_all(csrc);                      // the first frame below a zero row: its window did not slide in
            pz = zero_row;
        }
        cx<T> xr[R];                                                  // this wave's bins of the row: pairs (k, M - k), k = lane + 64 (sub R/2 + j2)
        if (real) {
            cx<T> z[R];
            if constexpr (WL) {
                lds_gather_use<0, R, lds_batch<R, T>(), T>((const cx<T>*)(winL + lofs), 64 * S, [&](int r, cx<T> w) {
                    z[r] = mkc<T>((T)raw[2 * r] * w.x, (T)raw[2 * r + 1] * w.y);
                    asm volatile("" : "+v"(z[r].x), "+v"(z[r].y));   // the multiplies stay above the next loads
                });
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    z[r] = mkc<T>((T)raw[2 * r] * wn[2 * r], (T)raw[2 * r + 1] * wn[2 * r + 1]);
                    asm volatile("" : "+v"(z[r].x), "+v"(z[r].y));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                // the row below is (almost always) the previous frame of the same signal: the lane's pairs move up by H, the hop's
                // new samples come in below them (k_fused_rev.hip); when it is no such frame nobody uses what is loaded
                const InT* ns = (nsrc != nullptr) ? nsrc : csrc;
#pragma unroll
                for (int r = R - 1; r >= H; r--) { raw[2 * r] = raw[2 * (r - H)]; raw[2 * r + 1] = raw[2 * (r - H) + 1]; }
#pragma unroll
                for (int r = 0; r < H; r++) load_pair(ns, r);
            }
            dftT<R, T>(z);                                            // stage 1
            __builtin_amdgcn_sched_barrier(0);
            dz[lane] = z[0];
            lds_gather_use<1, R, lds_batch<R, T>(), T>(t1L + lane, 64, [&](int q2, cx<T> w) { dz[q2 * PITCH + lane] = cmulT(z[q2], w); });
            wave_sync();
#pragma unroll
            for (int l2 = 0; l2 < R; l2++) z[l2] = dz[Q * PITCH + L1 + P * l2];
            wave_sync();
            dftT<R, T>(z);                                            // stage 2
            __builtin_amdgcn_sched_barrier(0);
            cx<T> tq2[R];
            if constexpr (lds_batch<R, T>() == R) lds_gather<1, R, T>(tq2, t2L + L1, P);
#pragma unroll
            for (int t = 0; t < R; t++) {
                cx<T> a = (t > 0) ? cmulT(z[t], lds_batch<R, T>() == R ? tq2[t] : t2L[t * P + L1]) : z[t];
                if constexpr (G1::LOGP >= 1) {
                    if constexpr (P >= 16) a = xstepT<8, true, T>(a, csg[G1::LOGP - 4], cw[G1::LOGP - 4]);
                    if constexpr (P >= 8) a = xstepT<4, true, T>(a, csg[G1::LOGP - 3], cw[G1::LOGP - 3]);
                    if constexpr (P >= 4) a = xstepT<2, true, T>(a, csg[G1::LOGP - 2], cw[G1::LOGP - 2]);
                    a = xstepT<1, false, T>(a, csg[G1::LOGP - 1], cw[G1::LOGP - 1]);
                }
                dz[zpadT<R, T>(Q + R * t + G1::R2 * t1v)] = a;       // Z_sub[k1], natural order
            }
        }
        lds_barrier();                                                // (1) the team's S sub-transforms are in their regions
        if (real) {
            // ---- join, in place: slots k1 of the S regions hold Z_s[k1] and become Z[k1 + M1 q]; this wave's share of k1 (k_stft.hip)
            constexpr int JW = R / S;
#pragma unroll 4
            for (int jj = sub * JW; jj < (sub + 1) * JW; jj++) {
                const int k1 = lane + 64 * jj;
                const int idx = zpadT<R, T>(k1);
                cx<T> a[S];
#pragma unroll
                for (int s2 = 0; s2 < S; s2++) a[s2] = buf[(size_t)s2 * G::BUFC + idx];
#pragma unroll
                for (int s2 = 1; s2 < S; s2++) a[s2] = cmulT(a[s2], tab[(2 * s2 * k1) & NMASK]);          // W_M^(s k1)
                if constexpr (S == 2) {
                    buf[idx] = a[0] + a[1];
                    buf[(size_t)G::BUFC + idx] = a[0] - a[1];
                } else {
                    const cx<T> A = a[0] + a[2], B = a[0] - a[2], C = a[1] + a[3], D = a[1] - a[3];
                    buf[idx] = A + C;
                    buf[(size_t)G::BUFC + idx] = addmni(B, D);        // W_4^q: 1, -i, -1, i
                    buf[(size_t)2 * G::BUFC + idx] = A - C;
                    buf[(size_t)3 * G::BUFC + idx] = addpi(B, D);
                }
            }
        }
        lds_barrier();                                                // (2) Z is complete
        if (real) {
            // ---- untangle: this wave's chunk of the pairs (k, M - k); the bins stay in xr
            const cx<T> zc = buf[ZA(M / 2)];
            constexpr int NPAIR = R / 2, NBATCH = 2, PB = NPAIR / NBATCH;
#pragma unroll
            for (int bt = 0; bt < NBATCH; bt++) {
                cx<T> za[PB], zb[PB], tw[PB];
#pragma unroll
                for (int j3 = 0; j3 < PB; j3++) {
                    const int k = lane + 64 * (sub * NPAIR + bt * PB + j3);
                    za[j3] = buf[ZA(k)];
                    zb[j3] = buf[ZA((M - k) & (M - 1))];
                    tw[j3] = tab[k];                                  // W_N^k, k < M/2
                }
#pragma unroll
                for (int j3 = 0; j3 < PB; j3++) {
                    const int j2 = bt * PB + j3;
                    const cx<T> Sm = mkc<T>(za[j3].x + zb[j3].x, za[j3].y - zb[j3].y);
                    const cx<T> D = mkc<T>(za[j3].x - zb[j3].x, za[j3].y + zb[j3].y);
                    const cx<T> O = mkc<T>((T)0.5 * D.y, (T)-0.5 * D.x);
                    const cx<T> Pk = cmulT(O, tw[j3]);
                    const cx<T> x0 = mkc<T>(fmaT((T)0.5, Sm.x, Pk.x), fmaT((T)0.5, Sm.y, Pk.y));
                    cx<T> x1 = mkc<T>(fmaT((T)0.5, Sm.x, -Pk.x), -fmaT((T)0.5, Sm.y, -Pk.y));
                    if (sub == 0 && j2 == 0 && lane == 0) x1 = mkc<T>(zc.x, -zc.y);      // bin 0 pairs with itself; its slot takes bin M/2
                    xr[2 * j2] = x0; xr[2 * j2 + 1] = x1;
                }
            }
        }
        lds_barrier();                                                // (3) every wave has read its pairs: the regions are free
        if (with_peaks) {
            constexpr int NPAIR = R / 2;
            T lmax = (T)-INFINITY, lmin = (T)INFINITY;
            double lsum = 0.0;
#pragma unroll
            for (int j2 = 0; j2 < NPAIR; j2++) {
                const int k = lane + 64 * (sub * NPAIR + j2);
                int kk = (M - k) & (M - 1);
                if (sub == 0 && j2 == 0 && lane == 0) kk = M / 2;
                // |X|^2, k_peaks.hip's formula: plain products and one sum
                const T e0 = xr[2 * j2].x * xr[2 * j2].x + xr[2 * j2].y * xr[2 * j2].y, e1 = xr[2 * j2 + 1].x * xr[2 * j2 + 1].x + xr[2 * j2 + 1].y * xr[2 * j2 + 1].y;
                yL[k] = e0;
                yL[kk] = e1;
                lmax = e0 > lmax ? e0 : lmax; lmax = e1 > lmax ? e1 : lmax;
                lmin = e0 < lmin ? e0 : lmin; lmin = e1 < lmin ? e1 : lmin;
                lsum += (double)e0 + (double)e1;
            }
            const double wmx = (double)wave_max(lmax), wmn = (double)wave_min(lmin), wsm = wave_sum(lsum);
            if (lane == 0) { partL[sub * 4] = wmx; partL[sub * 4 + 1] = wmn; partL[sub * 4 + 2] = wsm; }
        }
        lds_barrier();                                                // (4) the row of |X|^2 and the waves' extremes
        double th = 0.0, mn = 0.0;
        if (with_peaks) {
            double mx = partL[0];
            mn = partL[1];
#pragma unroll
            for (int w2 = 1; w2 < S; w2++) {
                mx = partL[w2 * 4] > mx ? partL[w2 * 4] : mx;
                mn = partL[w2 * 4 + 1] < mn ? partL[w2 * 4 + 1] : mn;
            }
            const double minamp = sqrt(mx) * p.thr;                   // PF.py:60
            th = (minamp != 0.0) ? minamp * minamp - mn : 0.0;
            // this wave's segment of the row, in pieces of 512 bins with a piece's reads in flight (k_stft.hip)
            unsigned short* const ciL = ciW + (size_t)sub * CAPW;
            int C_w = 0;
            constexpr int PIECE = 512;
#pragma unroll 1
            for (int kb = 0; kb < SEG; kb += PIECE) C_w += peak_scan<T, PIECE / 64, false>((const T*)yL, sub * SEG + kb, PIECE, M, (T)mn, th, (T*)nullptr, ciL + C_w, lane);
            // a segment with more candidates than npks keeps its npks best: whatever the row's selection takes is among them (the same
            // order -- score, then bin -- decides both), so the merged list wave 0 ranks holds at most S npks entries, and the dense
            // frames' selection is spread over the team
            if (C_w > K) {
                wave_sync();
                int* const selw = sel + 64;                            // (behind wave 0's sel: S x 64 ints)
                const int n_w = peak_pick<T, 0, false, false>((const T*)yL, (T*)nullptr, ciL, selw + sub * 64, M, K, C_w, th, lane, (T)mn);
                if (lane < n_w) ciL[lane] = (unsigned short)selw[sub * 64 + lane];
                C_w = n_w;
            }
            if (lane == 0) partL[sub * 4 + 3] = (double)C_w;
        }
        lds_barrier();                                                // (5) the waves' candidate lists and counts
        bool keep = false;
        int pb = 0;
        double tot = 0.0;
        if (with_peaks && sub == 0) {
            // ---- wave 0: the merged list (ascending bins), the npks best of it, their salience -- the general path's functions
            int C = 0;
#pragma unroll
            for (int w2 = 0; w2 < S; w2++) {
                const int c = (int)partL[w2 * 4 + 3];
                for (int i = lane; i < c; i += 64) ciM[C + i] = ciW[(size_t)w2 * CAPW + i];
                C += c;
            }
            // the row's energy (PV.py:210): the waves' sums, in k_stft_split's order
            tot = partL[2];
#pragma unroll
            for (int w2 = 1; w2 < S; w2++) tot += partL[w2 * 4 + 2];
            wave_sync();
            const int nsel = peak_pick<T, 0, false, false>((const T*)yL, (T*)nullptr, ciM, sel, M, K, C, th, lane, (T)mn);     // C <= S K <= 256 unless no segment was thinned; <= K <= 64 selected
            if (lane < nsel) { pb = sel[lane]; keep = salient<T>((const T*)yL, M, pb, p.rad); }
        }
        lds_barrier();                                                // (6) the last test on |X|^2 is made: the bins may land on it
        if (real) {
            constexpr int NPAIR = R / 2;
#pragma unroll
            for (int j2 = 0; j2 < NPAIR; j2++) {
                const int k = lane + 64 * (sub * NPAIR + j2);
                int kk = (M - k) & (M - 1);
                if (sub == 0 && j2 == 0 && lane == 0) kk = M / 2;
                buf[ZA(k)] = xr[2 * j2];
                buf[ZA(kk)] = xr[2 * j2 + 1];
            }
        }
        lds_barrier();                                                // (7) the row's spectrum, natural order, in the regions
        if (have && sub == 0) {
            if (pend) {
                // ---- the frame above (staged last) takes its previous spectrum from this row
                if (lane < pend_nk) {
                    const int sl = pend_off + lane;
                    const int nbin = Lbin[sl];
                    cx<T> pv = mkc<T>((T)0, (T)0);
                    if (zero_row) { if (pend_prev0) pv = mkc<T>((T)p.prev0[2 * nbin], (T)p.prev0[2 * nbin + 1]); }
                    else pv = buf[ZA(nbin)];
                    Gval[sl * 5 + 2] = pv.x; Gval[sl * 5 + 3] = pv.y;
                }
                pend = false;
            }
            if (with_peaks) {
                if (ng == GFV || nst + K > kSlots) { flush(ng); ng = 0; nst = 0; }
                const unsigned long long bal = __ballot(keep);
                if (keep) {
                    const int sl = nst + lane_prefix(bal);
                    const cx<T> c = buf[ZA(pb)], vm = buf[ZA(pb - 1)], vp = buf[ZA(pb + 1)];
                    // PV.py:197-199: 3-bin energy, bin 0 excluded (1 <= pb <= M-2); |X|^2 by the untangle's expression
                    T s3 = (T)0;
                    if (pb > 1) s3 = s3 + (vm.x * vm.x + vm.y * vm.y);
                    s3 = s3 + (c.x * c.x + c.y * c.y);
                    s3 = s3 + (vp.x * vp.x + vp.y * vp.y);
                    Lbin[sl] = pb;
                    Gval[sl * 5 + 0] = c.x; Gval[sl * 5 + 1] = c.y; Gval[sl * 5 + 4] = s3;
                }
                const int nk = __popcll(bal);
                if (lane == 0) { Lcnt[ng] = nk; Loff[ng] = nst; Lfrm[ng] = gq - 1; Lorow[ng] = (long long)orow; Ltot[ng] = tot; }
                pend = true; pend_prev0 = (p.prev0 != nullptr) && (orow == 0);
                pend_off = nst; pend_nk = nk;
                nst += nk; ng++;
            }
            if (g == spec_g && !zero_row) {
                gdouble* const so = (gdouble*)p.spec_out;
                for (int k = lane; k < M; k += 64) {
                    const cx<T> v = buf[ZA(k)];
                    so[2 * k] = v.x;
                    so[2 * k + 1] = v.y;
                }
            }
        }
        lds_barrier();                                                // (8) the regions are free for the row below
        if (have) {
            gq = qn;
            if (nsrc != nullptr) { csrc = nsrc; orow = norow; }
        }
    }
    if (sub == 0 && ng > 0) flush(ng);
}

template <int R, int S, int TEAMS, typename InT> const void* pick_fn(int H) {
    return H == R / 4 ? (const void*)k_pv_team<R, S, InT, R / 4, TEAMS> : (const void*)k_pv_team<R, S, InT, R / 2, TEAMS>;
}

template <int R, int S, int TEAMS, typename InT> int launch_team_t(const PvRevParams& p, int H, hipStream_t s) {
    constexpr bool WL = (S == 2) || (sizeof(InT) == 8);
    using G = TeamGeo<R, S, WL>;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    constexpr size_t lds = G::total(TEAMS);
    static_assert(lds <= 160 * 1024, "a workgroup fits the CU's LDS");
    const void* fn = pick_fn<R, S, TEAMS, InT>(H);
    if (lds > 64 * 1024) PVX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t nblocks = ncu;                                             // one workgroup per CU
    if (p.blocks_override > 0) nblocks = p.blocks_override;
    const int64_t maxb = (p.row_end - p.row_begin + TEAMS - 1) / TEAMS;   // never more teams than rows
    if (nblocks > maxb) nblocks = maxb > 0 ? maxb : 1;
    if (p.stage == nullptr || p.stage_bytes < (size_t)nblocks * TEAMS * kSlots * 5 * 8) {
        pvx_set_error("k_pv_team: the staging block holds %zu bytes, %zu needed", p.stage_bytes, (size_t)nblocks * TEAMS * kSlots * 5 * 8);
        return PVX_ERR_INVALID;
    }
    dim3 grid((unsigned)nblocks), block(64 * S * TEAMS);
    PvRevParams arg = p;
    void* args[] = {&arg};
    PVX_HIP_CHECK(hipLaunchKernel(fn, grid, block, args, lds, s));
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}

template <int R, int S> int launch_team(const PvRevParams& p, int x_dtype, hipStream_t s) {
    const int nfft = 128 * R * S;
    const int H = (p.hop == nfft / 4) ? R / 4 : (p.hop == nfft / 2) ? R / 2 : 0;
    if (H == 0) { pvx_set_error("k_pv_team takes the sliding-window hops (nfft/4, nfft/2), not %d", p.hop); return PVX_ERR_UNSUPPORTED; }
    // teams per workgroup by LDS: nfft 4096 three (float32 / int16 samples) or two (float64 samples: 512 registers), nfft 8192 one
    switch (x_dtype) {
        case PVX_F32: return launch_team_t<R, S, (S == 2 ? PVX_TEAM_F32_TEAMS : 1), float>(p, H, s);
        case PVX_I16: return launch_team_t<R, S, (S == 2 ? PVX_TEAM_F32_TEAMS : 1), int16_t>(p, H, s);
        case PVX_F64: return launch_team_t<R, S, (S == 2 ? 2 : 1), double>(p, H, s);
        default: pvx_set_error("bad x_dtype %d", x_dtype); return PVX_ERR_INVALID;
    }
}

}  // namespace

// nfft 4096 / 8192 at precision 64, npks <= 64, hop nfft/4 or nfft/2
int pvx_pv_team_supported(int nfft, int precision, int K, int hop) {
    if (precision != 64 || K < 1 || K > kSlots) return 0;
    if (nfft != 4096 && nfft != 8192) return 0;
    return hop == nfft / 4 || hop == nfft / 2;
}
size_t pvx_pv_team_stage_bytes(int nfft) {
    if (nfft != 4096 && nfft != 8192) return 0;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    if (ncu < 256) ncu = 256;
    return (size_t)ncu * 3 * kSlots * 5 * 8;
}
int pvx_launch_pv_team(const PvRevParams& p, int nfft, int x_dtype, hipStream_t s) {
    if (p.total_rows <= 0 || p.row_end <= p.row_begin) return PVX_OK;
    if (p.row_begin < 0 || p.row_end > p.total_rows || p.total_rows >= 0x7fffff00LL) { pvx_set_error("k_pv_team: rows [%lld, %lld) of %lld", (long long)p.row_begin, (long long)p.row_end, (long long)p.total_rows); return PVX_ERR_INVALID; }
    if (p.K > kSlots) { pvx_set_error("k_pv_team stages at most %d peaks per frame (npks = %d)", kSlots, p.K); return PVX_ERR_UNSUPPORTED; }
    switch (nfft) {
        case 4096: return launch_team<16, 2>(p, x_dtype, s);
        case 8192: return launch_team<16, 4>(p, x_dtype, s);
        default: break;
    }
    pvx_set_error("the float64 team kernel does not handle nfft=%d", nfft);
    return PVX_ERR_UNSUPPORTED;
}
