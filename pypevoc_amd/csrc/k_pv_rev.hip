// k_pv_rev.hip -- the float64 analysis stage (window + FFT + untangle + peaks + phase-vocoder arithmetic) in ONE launch
// with NO spectrum row in HBM: every wave walks a contiguous range of rows in DESCENDING order, the spectrum of the row
// at hand stays on chip.
//
//   PV.calc_fft_frame   pypevoc/PVAnalysis.py:150-158   (float64, as the reference computes it)
//   PV.calc_pv_frame    pypevoc/PVAnalysis.py:160-211   (abs, PeakFinder + filter_by_salience, dphase2freq, 3-bin energy)
//   PV.run_pv           pypevoc/PVAnalysis.py:213-264   (frame loop, zero-padded packing)
//
// k_stft_pv.hip (the kernel this one replaces for npks <= 64) wrote every spectrum row to a workspace -- 16 KB per frame
// at nfft 2048, 940 MB per BASELINE-config-2 launch against 17 MB of results -- only to read the <= npks kept bins of the
// current and the previous row back.  A frame's peaks need the PREVIOUS frame's spectrum at their bins (PV.py:171, 190).
// Walking the rows downwards (k_fused_rev.hip, the float32 kernel) that is free: the wave finds the peaks of frame j on
// |X_j|^2, keeps their X_j values and 3-bin energies in a few staging slots, and the transform it runs next leaves
// X_(j-1), where the staged peaks pick their previous values up.
//
// What is different from the float32 kernel is the room: a float64 row is 16 bytes per bin and the peak search wants the
// |X|^2 row (8 bytes per bin) beside its candidate lists, which is exactly what the transform's LDS buffer holds -- there
// is no second buffer to have (nfft 2048: 17 KB per wave, eight waves and the tables fill the CU's 160 KB).  So the
// untangle leaves the row's bins IN THE LANE'S REGISTERS (R complex values: the search that follows is light on
// registers), writes |X|^2 to LDS where the transform's exchange matrix was, the search runs there as in k_stft_pv, and
// once the last test on |X|^2 is made the bins go to the same LDS bytes in natural order: the kept peaks' values and the
// 3-bin energies are read there (|X|^2 recomputed by the untangle's own expression: the same bits), and the frame
// above's staged peaks take their previous-spectrum values.  One LDS write pass per frame instead of a 16 KB row through
// HBM.  Staging: 64 slots per wave, a frame's kept peaks behind the previous frame's (the dense form of k_fused_rev); the
// per-peak pass runs when eight frames wait or the next frame might not fit.  The slots' bins and the frames' records are
// in LDS; the five float64 values of a slot are in LDS too at nfft 512 / 1024, and at nfft 2048 -- where 256 bytes are
// all the CU has left -- in a per-wave block of global memory that stays in the L2 the wave wrote it to (40 bytes per
// kept peak instead of a row).
// Per-frame arithmetic is k_stft_pv's, expression for expression: results are bit-identical to it (GPU test) and do not
// depend on the launch geometry.
#include <stdlib.h>

#include <type_traits>

#include "pvx_stft4.h"

using namespace pvxw;
using namespace pvxf;
using namespace pvxs;

namespace {

constexpr int GFV = 8;                // frames staged before the per-peak pass
constexpr int kSlots = 64;            // staging slots: one per lane of the per-peak pass

// issue priorities of the row loop's phases (k_fused_rev.hip / k_stft_pv.hip): transform, search, selection + staging
#ifndef PVX_PVR_PRIO_T
#define PVX_PVR_PRIO_T 2
#endif
#ifndef PVX_PVR_PRIO_S
#define PVX_PVR_PRIO_S 1
#endif
#ifndef PVX_PVR_PRIO_C
#define PVX_PVR_PRIO_C 0
#endif
#ifndef PVX_PVR_PRIO_MINR
#define PVX_PVR_PRIO_MINR 8
#endif
// tools/ab knobs: waves per CU at nfft 1024 (12 = three per SIMD: the staged values then live in the global block and the wave has 168
// registers; 8 = two per SIMD, values in LDS) for float32 / int16 and for float64 samples, bins per piece of the candidate scan
#ifndef PVX_PVR_NW1024
#define PVX_PVR_NW1024 12
#endif
#ifndef PVX_PVR_NW512
#define PVX_PVR_NW512 16              // (four waves per SIMD at 128 registers, 20 of them in scratch: +5 .. 8 % over twelve, profiles/r06_ab_steps.txt)
#endif
#ifndef PVX_PVR_PIECE
#define PVX_PVR_PIECE 512
#endif

// LDS: the block's tables as k_stft_pv lays them out (PvGeo there), then per wave the transform buffer and the staging
template <int R, bool SYM, bool LVP = true> struct RvGeo : StftGeo<R, double> {
    using T = double;
    using B = StftGeo<R, double>;
    static constexpr bool X4 = (R == 16);
    static_assert(!SYM || X4, "the half window exists for the four-quarter layout");
    static constexpr bool LV = !X4 && LVP;                           // the staged values live in LDS (else: global staging block)
    static constexpr int TW8N = X4 ? 512 : ((B::HALF / 2 + 1 + 7) & ~7);
    static constexpr size_t OFF_T1 = SYM ? (size_t)(B::N / 2) * sizeof(T) : B::OFF_T1;
    static constexpr size_t OFF_T2 = X4 ? OFF_T1 + (size_t)256 * 2 * sizeof(T) : B::OFF_T2;
    static constexpr size_t OFF_TW3 = X4 ? OFF_T2 : B::OFF_TW3;
    // (the twelve-wave form of nfft 1024: the lanes' cross-lane twiddles cw [LOGP][64] wait in LDS instead of in registers across the loop
    // -- as loop invariants of a 168-register kernel they were spilled to scratch and reloaded from there in every frame)
    static constexpr bool CWL = !X4 && !LVP;
    static constexpr size_t OFF_CW = OFF_TW3 + (size_t)TW8N * 2 * sizeof(T);
    static constexpr size_t OFF_BUF = OFF_CW + (CWL ? (size_t)(B::LOGP > 0 ? B::LOGP : 1) * 64 * 2 * sizeof(T) : 0);
    // per wave: dz [BUFC] cx | orow [GFV] i64 | tot [GFV] f64 | cnt, off, frm [GFV] int | bin [kSlots] int | (LV) val [kSlots][5] f64
    static constexpr size_t W_OROW = (size_t)B::BUFC * 2 * sizeof(T);
    static constexpr size_t W_TOT = W_OROW + GFV * 8;
    static constexpr size_t W_CNT = W_TOT + GFV * 8;
    static constexpr size_t W_OFF = W_CNT + GFV * 4;
    static constexpr size_t W_FRM = W_OFF + GFV * 4;
    static constexpr size_t W_BIN = W_FRM + GFV * 4;
    static constexpr size_t W_VAL = W_BIN + kSlots * 4;
    static constexpr size_t PER_WAVE = (W_VAL + (LV ? (size_t)kSlots * 5 * 8 : 0) + 15) & ~(size_t)15;
    __host__ __device__ static constexpr size_t total(int nw) { return OFF_BUF + PER_WAVE * (size_t)nw; }
    // the search's arrays inside dz: y [M] T | cs [CAP] T | ci [CAP] int | sel [64] int
    static constexpr int CAP = B::M / 2 + 4;
    static_assert((size_t)B::M * 8 + (size_t)CAP * 12 + 64 * 4 <= W_OROW, "the peak search fits the transform buffer");
};

typedef __attribute__((address_space(1))) double gdouble;

template <int R, typename InT, int H, bool SYM, int NW>
__global__ __launch_bounds__(64 * NW) void k_pv_rev(PvRevParams p) {
    using T = double;
    // nfft 1024 with twelve waves per CU (three per SIMD: 168 registers) keeps the staged values in the global block like nfft 2048 --
    // twelve waves' staging does not fit the LDS beside their buffers -- and scans in pieces of 256 bins (fewer values in flight)
    using G = RvGeo<R, SYM, !((R == 8 && NW > 8) || (R == 4 && NW > 12))>;
    constexpr int M = G::M, P = G::P, PITCH = G::PITCH, CAP = G::CAP;
    constexpr bool X4 = G::X4, LV = G::LV;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int K = p.K;
    T* const winL = (T*)(smem + G::OFF_WIN);
    cx<T>* const t1L = (cx<T>*)(smem + G::OFF_T1);
    cx<T>* const t2L = (cx<T>*)(smem + G::OFF_T2);
    cx<T>* const tw3 = (cx<T>*)(smem + G::OFF_TW3);
    unsigned char* const wbase = smem + G::OFF_BUF + (size_t)wid * G::PER_WAVE;
    cx<T>* const dz = (cx<T>*)wbase;
    T* const y = (T*)wbase;
    T* const cs = y + M;
    int* const ci = (int*)(cs + CAP);
    int* const sel = ci + CAP;
    long long* const Lorow = (long long*)(wbase + G::W_OROW);
    double* const Ltot = (double*)(wbase + G::W_TOT);
    int* const Lcnt = (int*)(wbase + G::W_CNT);
    int* const Loff = (int*)(wbase + G::W_OFF);
    int* const Lfrm = (int*)(wbase + G::W_FRM);
    int* const Lbin = (int*)(wbase + G::W_BIN);
    double* const Lval = (double*)(wbase + G::W_VAL);                // (LV)
    // (!LV) this wave's block of the global staging: [kSlots][5] float64
    gdouble* const Gval = LV ? nullptr : (gdouble*)p.stage + ((size_t)blockIdx.x * NW + wid) * (kSlots * 5);

    auto XA = [](int k) -> int { if constexpr (X4) return xa4(k); else return zpadT<R, T>(k); };     // slot of bin k in dz

    // ---- rows of this wave: [r0, r1) walked downwards, then row r0 - 1 (spectrum only: the previous one of its last frame)
    // (the launch covers global rows [row_begin, row_end): all of them, or one piece of a call that reports its progress)
    const int64_t Wn = (int64_t)gridDim.x * NW, wv = (int64_t)blockIdx.x * NW + wid, nrows = p.row_end - p.row_begin;
    const int r0 = (int)(p.row_begin + nrows * wv / Wn), r1 = (int)(p.row_begin + nrows * (wv + 1) / Wn);
    const bool idle_wave = r0 >= r1;
    const int glast = r0 - 1;
    const int Fi = (int)p.F, rows1 = Fi + 1;
    const int hopi = p.hop;

    using RawT = typename std::conditional<(sizeof(InT) == 8), double, float>::type;
    RawT raw[2 * R];
#pragma unroll
    for (int r = 0; r < 2 * R; r++) raw[r] = (RawT)0;
    const int lofs = X4 ? lofs4(lane) : 2 * lane;                     // sample offset of the lane's first pair
    auto row_ptr = [&](int b, int q) -> const InT* { return (const InT*)p.x + (int64_t)b * p.sig_stride + (int64_t)(q - 1) * hopi; };
    auto load_pair = [&](const InT* src, int r) {
        const InT* q = src + lofs + 128 * r;
        raw[2 * r] = (RawT)q[0]; raw[2 * r + 1] = (RawT)q[1];
    };
    auto prefetch_part = [&](const InT* src, int part) {
        if (src == nullptr) return;
        constexpr int PR = R / 4;
#pragma unroll
        for (int r = part * PR; r < (part + 1) * PR; r++) load_pair(src, r);
    };

    // the row at hand: g (global), gq (row in its signal; 0 = the zero row) and -- valid while gq >= 1 -- its samples and output row
    int g = r1 - 1, gq = 0, orow = 0;
    const InT* csrc = (const InT*)p.x;
    {
        const int gb = g >= 0 ? g / rows1 : 0;
        gq = g - gb * rows1;
        if (g >= 0 && gq >= 1) { csrc = row_ptr(gb, gq); orow = gb * Fi + gq - 1; }
    }
    // (the first row's samples fly while the workgroup fills its tables)
    if (!idle_wave && g >= 0 && gq >= 1) { prefetch_part(csrc, 0); prefetch_part(csrc, 1); prefetch_part(csrc, 2); prefetch_part(csrc, 3); }

    {
        const cx<T>* tab = (const cx<T>*)p.twiddle;
        constexpr int NMASK = G::N - 1;
        for (int i = threadIdx.x; i < (SYM ? G::N / 2 : G::N); i += 64 * NW) winL[i] = ((const T*)p.win)[i];
        if constexpr (X4) {
            for (int i = threadIdx.x; i < 256; i += 64 * NW) t1L[i] = tab[((G::N / 256) * (i & 15) * (i >> 4)) & NMASK];    // [q][l] W_256^(l q)
            for (int i = threadIdx.x; i < 512; i += 64 * NW) {      // [j][u][lane]: W_N^k1 (u = 0), W_1024^(u k1); k1 = lane + 64 j
                const int ln = i & 63, u = (i >> 6) & 3, k1 = ln + 64 * (i >> 8);
                tw3[i] = tab[(u == 0 ? k1 : 2 * u * k1) & NMASK];
            }
        } else {
            for (int i = threadIdx.x; i < R * 64; i += 64 * NW) t1L[i] = tab[(2 * (i & 63) * (i >> 6)) & NMASK];
            for (int i = threadIdx.x; i < 64; i += 64 * NW) t2L[i] = tab[((G::N / 64) * (i % P) * (i / P)) & NMASK];   // [t2][l1]
            for (int i = threadIdx.x; i <= G::HALF / 2; i += 64 * NW) tw3[i] = tab[i];
            if constexpr (G::CWL) {
                cx<T>* const cwT = (cx<T>*)(smem + G::OFF_CW);
                for (int i = threadIdx.x; i < G::LOGP * 64; i += 64 * NW) {
                    const int st = i >> 6, l1 = (i & 63) % P, h = P >> (st + 1);
                    cwT[i] = (l1 & h) ? tab[((G::N / (2 * h)) * (l1 % h)) & NMASK] : mkc<T>((T)1, (T)0);
                }
            }
        }
    }
    __syncthreads();
    if (idle_wave) return;

    const int Q = lane / P, L1 = lane % P;
    T csg[G::LOGP > 0 ? G::LOGP : 1];
    cx<T> cw[G::LOGP > 0 ? G::LOGP : 1];
    cx<T>* const cwL = (cx<T>*)(smem + G::OFF_CW);                    // (CWL) [LOGP][64]
    if constexpr (!G::CWL) {
        const cx<T>* tab = (const cx<T>*)p.twiddle;
        constexpr int NMASK = G::N - 1;
#pragma unroll
        for (int s = 0; s < G::LOGP; s++) {
            const int h = P >> (s + 1);
            const bool up = (L1 & h) != 0;
            csg[s] = up ? (T)-1 : (T)1;
            const cx<T> wvv = tab[((G::N / (2 * h)) * (L1 % h)) & NMASK];
            cw[s] = up ? wvv : mkc<T>((T)1, (T)0);
        }
    }
    int t1v = 0;
#pragma unroll
    for (int b = 0; b < G::LOGP; b++) if (L1 & (1 << b)) t1v |= 1 << (G::LOGP - 1 - b);

    PeakConst pc;
    pc.fstep = p.fstep; pc.dt = p.dt; pc.nfft = G::N; pc.hop = p.hop; pc.wfbin = p.wfbin;

    auto sval_put = [&](int sl, int i, double v) {
        if constexpr (LV) Lval[sl * 5 + i] = v; else Gval[sl * 5 + i] = v;
    };

    // ---- per-peak phase-vocoder arithmetic on the staged frames [0, ng): lane l takes slot l
    auto flush = [&](int ng) {
        if constexpr (!LV) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's staged values are in L2
        wave_sync();
        const int lnf = fresh_lane();
        static_assert(GFV == 8, "two 16-byte reads per table");
        const int4 oa = *(const int4*)Loff, ob4 = *((const int4*)Loff + 1), ca = *(const int4*)Lcnt, cb4 = *((const int4*)Lcnt + 1);
        const int offs[8] = {oa.x, oa.y, oa.z, oa.w, ob4.x, ob4.y, ob4.z, ob4.w}, cnts[8] = {ca.x, ca.y, ca.z, ca.w, cb4.x, cb4.y, cb4.z, cb4.w};
        int top = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { if (j == ng - 1) top = offs[j] + cnts[j]; }
        bool valid = lnf < top;
        int gi = 0, start = 0, cnt = cnts[0];                          // the frame of this lane's slot: the last one that starts at or before it
#pragma unroll
        for (int j = 1; j < 8; j++) { if (j < ng && offs[j] <= lnf) { gi = j; start = offs[j]; cnt = cnts[j]; } }
        const int64_t orw = (int64_t)Lorow[gi];
        gdouble* of = (gdouble*)p.f + orw * K;
        gdouble* om = (gdouble*)p.mag + orw * K;
        gdouble* op = (gdouble*)p.ph + orw * K;
        gdouble* orp = (gdouble*)p.realph + orw * K;
        gdouble* ob = (gdouble*)p.binno + orw * K;
        int nbin = 0;
        PeakOut o;
        o.freq = 0.0; o.dfb = 0.0; o.thisph = 0.0; o.mag = 0.0; o.valid = false;
        if (valid) {
            nbin = Lbin[lnf];
            double v[5];
            if constexpr (LV) {
#pragma unroll
                for (int i = 0; i < 5; i++) v[i] = Lval[lnf * 5 + i];
            } else {
#pragma unroll
                for (int i = 0; i < 5; i++) v[i] = __hip_atomic_load((const double*)(Gval + lnf * 5 + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            o = peak_math<T>(nbin, v[0], v[1], v[2], v[3], v[4], pc);
            valid = o.valid;
        }
        const unsigned long long gm = (cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull)) << start;
        const unsigned long long ball = __ballot(valid);
        if (valid) {
            const int oi = __popcll(ball & gm & ((1ull << lnf) - 1ull));
            ob[oi] = (double)nbin;
            of[oi] = o.freq;
            om[oi] = o.mag;
            op[oi] = o.thisph;
            orp[oi] = o.thisph + kPi * o.dfb / pc.fstep;               // PV.py:207
        }
        {
            // zero padding (PV.py:226-239) and the frames' scalars: eight lanes per staged frame
            const int g2 = lnf >> 3, c2 = lnf & 7;
            int o2 = 0, n2 = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) { if (j == g2) { o2 = offs[j]; n2 = cnts[j]; } }
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

    int ng = 0, nst = 0;                                              // staged frames, staged slots
    bool pend = false, pend_prev0 = false, pz = false;               // the frame staged last waits for its previous spectrum (the caller's prev0) | the row above was a zero row
    int pend_off = 0, pend_nk = 0;
    const int spec_g = (p.spec_out != nullptr && p.spec_row >= 0 && p.spec_row < 0x7fffffffLL) ? (int)p.spec_row : -0x7fffffff;
    for (; g >= glast; --g) {
        const int qn = gq >= 1 ? gq - 1 : Fi;                        // row g - 1 in its signal (below a zero row: the last row of the signal before)
        const bool zero_row = (g < 0) || (gq == 0);
        const bool with_peaks = !zero_row && g >= r0;
        // samples and output row of row g - 1 (nullptr: nothing to load -- a zero row, or below the wave's range)
        const InT* nsrc = nullptr;
        int norow = orow - 1;
        if (g - 1 >= glast && g - 1 >= 0 && qn >= 1) {
            if (gq >= 2) nsrc = csrc - hopi;
            else { const int bn = (g - 1) / rows1; nsrc = row_ptr(bn, qn); norow = bn * Fi + qn - 1; }
        }
        if constexpr (H > 0) {
            // the first frame below a zero row: its window did not slide in
            if (!zero_row && pz) { prefetch_part(csrc, 0); prefetch_part(csrc, 1); prefetch_part(csrc, 2); prefetch_part(csrc, 3); }
            pz = zero_row;
        }
        cx<T> xr[R];                                                  // the row's bins of this lane (see the dump below for which)
        T maxv = (T)0, minv = (T)0;
        double tot = 0.0;
        int nsel = 0;
        if (!zero_row) {
            if constexpr (R >= PVX_PVR_PRIO_MINR) __builtin_amdgcn_s_setprio(PVX_PVR_PRIO_T);
            cx<T> z[R];
            {
                // the lane's window pairs, all in flight before the first product (k_stft_pv.hip)
                constexpr int WB = (sizeof(RawT) == 8 && R == 16) ? R / 2 : R;
#pragma unroll
                for (int q0 = 0; q0 < R; q0 += WB) {
                    cx<T> wq[WB];
#pragma unroll
                    for (int r = 0; r < WB; r++) {
                        // (SYM: the pair n, n + 1 of the upper half is w[N-1-n], w[N-2-n]: the pair at N-2-n, read backwards)
                        const bool up = SYM && (q0 + r) >= R / 2;
                        wq[r] = *(const cx<T>*)(winL + (up ? G::N - 2 - (lofs + 128 * (q0 + r)) : lofs + 128 * (q0 + r)));
                    }
#pragma unroll
                    for (int r = 0; r < WB; r++) asm volatile("" : "+v"(wq[r].x), "+v"(wq[r].y));
#pragma unroll
                    for (int r = 0; r < WB; r++) {
                        const bool up = SYM && (q0 + r) >= R / 2;
                        z[q0 + r] = mkc<T>((T)raw[2 * (q0 + r)] * (up ? wq[r].y : wq[r].x), (T)raw[2 * (q0 + r) + 1] * (up ? wq[r].x : wq[r].y));
                        asm volatile("" : "+v"(z[q0 + r].x), "+v"(z[q0 + r].y));       // the multiplies stay above the next loads
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const InT* nfull = nsrc;                                  // what the four prefetch parts fetch (nothing once the window has slid)
            if constexpr (H > 0) {
                // the row below is (almost always) the previous frame of the same signal: the lane's pairs move up by H and the
                // hop's new samples come in below them -- on every row, without a branch (k_fused_rev.hip); when the row below is
                // no such frame the loads read the row at hand again and nobody uses them
                const InT* ns = (nsrc != nullptr) ? nsrc : csrc;
#pragma unroll
                for (int r = R - 1; r >= H; r--) { raw[2 * r] = raw[2 * (r - H)]; raw[2 * r + 1] = raw[2 * (r - H) + 1]; }
#pragma unroll
                for (int r = 0; r < H; r++) load_pair(ns, r);
                nfull = nullptr;
            }
            prefetch_part(nfull, 0);
            if constexpr (X4) {
                // ---- four 256-point transforms, then the radix-4 join inside the untangle (pvx_stft4.h)
                fft4_quartersT<T, (sizeof(RawT) == 8 ? 8 : 16)>(z, dz, t1L, lane, [&]() { prefetch_part(nfull, 1); }, [&]() { prefetch_part(nfull, 2); },
                                                             [&]() { prefetch_part(nfull, 3); });
                int lu = lane;
                asm volatile("" : "+v"(lu));
                wave_sync();
                Join4In<T> in;
                join4_read<T>(dz, lu, in);
                wave_sync();                                          // dz is free: the magnitude row goes there
                int ei = 0;
                join4_emit<T>(in, tw3, lu, [&](int k, cx<T> x) {
                    xr[ei++] = x;
                    if (with_peaks) y[k] = x.x * x.x + x.y * x.y;
                });
            } else {
                dftT<R, T>(z);                                        // stage 1
                __builtin_amdgcn_sched_barrier(0);
                prefetch_part(nfull, 1);
                cx<T> tq1[R];
                lds_gather<1, R, T>(tq1, t1L + lane, 64);
#pragma unroll
                for (int q2 = 0; q2 < R; q2++) dz[q2 * PITCH + lane] = (q2 > 0) ? cmulT(z[q2], tq1[q2]) : z[q2];
                wave_sync();
#pragma unroll
                for (int l2 = 0; l2 < R; l2++) z[l2] = dz[Q * PITCH + L1 + P * l2];
                prefetch_part(nfull, 2);
                wave_sync();
                dftT<R, T>(z);                                        // stage 2
                __builtin_amdgcn_sched_barrier(0);
                prefetch_part(nfull, 3);
                cx<T> tq2[R];
                lds_gather<1, R, T>(tq2, t2L + L1, P);
                if constexpr (G::CWL) {
#pragma unroll
                    for (int st = 0; st < G::LOGP; st++) {
                        cw[st] = cwL[st * 64 + lane];
                        csg[st] = (L1 & (P >> (st + 1))) ? (T)-1 : (T)1;
                    }
                }
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
                // ---- untangle: pairs (k, M-k), k = lane + 64 j2 (k_stft.hip); the bins stay in xr
                constexpr int NPAIR = R / 2;
                int lu = lane;
                asm volatile("" : "+v"(lu));
                cx<T> za[NPAIR], zb[NPAIR];
#pragma unroll
                for (int j2 = 0; j2 < NPAIR; j2++) {
                    const int k = lu + 64 * j2;
                    za[j2] = dz[zpadT<R, T>(k)];
                    zb[j2] = dz[zpadT<R, T>((M - k) & (M - 1))];
                }
                const cx<T> zc = dz[zpadT<R, T>(G::HALF)];
                wave_sync();                                          // dz is free: the magnitude row goes there
#pragma unroll
                for (int j2 = 0; j2 < NPAIR; j2++) {
                    const int k = lu + 64 * j2;
                    const int km = (M - k) & (M - 1);
                    const cx<T> S = mkc<T>(za[j2].x + zb[j2].x, za[j2].y - zb[j2].y);
                    const cx<T> D = mkc<T>(za[j2].x - zb[j2].x, za[j2].y + zb[j2].y);
                    const cx<T> O = mkc<T>((T)0.5 * D.y, (T)-0.5 * D.x);
                    cx<T> wk;
                    if (j2 < NPAIR / 2) wk = tw3[k];                 // k <= nfft/8
                    else { const cx<T> e = tw3[G::HALF - k]; wk = mkc<T>(-e.y, -e.x); }
                    const cx<T> Pk = cmulT(O, wk);
                    const cx<T> x0 = mkc<T>(fmaT((T)0.5, S.x, Pk.x), fmaT((T)0.5, S.y, Pk.y));
                    cx<T> x1 = mkc<T>(fmaT((T)0.5, S.x, -Pk.x), -fmaT((T)0.5, S.y, -Pk.y));
                    int kk = km;
                    if (j2 == 0 && lu == 0) { x1 = mkc<T>(zc.x, -zc.y); kk = G::HALF; }      // bin 0 pairs with itself; its slot takes bin M/2
                    xr[2 * j2] = x0; xr[2 * j2 + 1] = x1;
                    if (with_peaks) {
                        // the peak search runs on |X|^2: every test it makes is monotone in |X| (k_peaks.hip)
                        y[k] = x0.x * x0.x + x0.y * x0.y;
                        y[kk] = x1.x * x1.x + x1.y * x1.y;
                    }
                }
            }
            wave_sync();
            if constexpr (R >= PVX_PVR_PRIO_MINR) __builtin_amdgcn_s_setprio(PVX_PVR_PRIO_S);
        } else {
            // the zero frame in front of every signal (PV.py:121): nothing to transform; the row below (the last frame of the
            // signal before) loads its whole window
            if constexpr (H == 0) { prefetch_part(nsrc, 0); prefetch_part(nsrc, 1); prefetch_part(nsrc, 2); prefetch_part(nsrc, 3); }
        }

        bool keep = false;
        int pb = 0;
        if (with_peaks) {
            // ---- extremes and energy of the row, in k_phase_peaks' order of summation (PV.py:173, 210; PF.py:60, 164)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            T lmax = (T)-INFINITY, lmin = (T)INFINITY;
            double lsum = 0.0;
#pragma unroll
            for (int i = 0; i < M / 64; i++) {
                const T m0 = y[ln + 64 * i];
                lmax = m0 > lmax ? m0 : lmax;
                lmin = m0 < lmin ? m0 : lmin;
                lsum += m0;
            }
            maxv = wave_max(lmax);
            minv = wave_min(lmin);
            tot = wave_sum(lsum);
            // ---- PeakFinder(famp, npeaks, minrattomax) + filter_by_salience(rad=5)  (PV.py:175-178), on the squared row (k_stft_pv.hip)
            const double minamp = sqrt((double)maxv) * p.thr;        // PF.py:60
            const double th = (minamp != 0.0) ? minamp * minamp - (double)minv : 0.0;
            constexpr int PIECE = (R == 8 && NW > 8) ? 256 : (M < PVX_PVR_PIECE ? M : PVX_PVR_PIECE);
            int C = 0;
#pragma unroll 1
            for (int kb = 0; kb < M; kb += PIECE) C += peak_scan<T, PIECE / 64>(y, kb, PIECE, M, minv, th, cs + C, ci + C, ln);
            wave_sync();
            if constexpr (R >= PVX_PVR_PRIO_MINR) __builtin_amdgcn_s_setprio(PVX_PVR_PRIO_C);
            nsel = peak_pick<T>(y, cs, ci, sel, M, K, C, th, ln);    // <= K <= 64: one selected peak per lane
            int rad = p.rad;
            asm volatile("" : "+s"(rad));
            if (ln < nsel) { pb = sel[ln]; keep = salient<T>(y, M, pb, rad); }
            wave_sync();                                              // the last read of |X|^2: the bins may land on it
        }
        if (!zero_row) {
            // ---- the row's bins, from the lanes' registers to dz in natural order
            int lu = lane;
            asm volatile("" : "+v"(lu));
            if constexpr (X4) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int k1 = lu + 64 * j;
                    const int kbb = (j == 0 && lu == 0) ? 128 : ((256 - k1) & 255);
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        dz[xa4(k1 + 256 * t)] = xr[j * 8 + 2 * t];
                        dz[xa4(kbb + 256 * (3 - t))] = xr[j * 8 + 2 * t + 1];
                    }
                }
            } else {
#pragma unroll
                for (int j2 = 0; j2 < R / 2; j2++) {
                    const int k = lu + 64 * j2;
                    const int kk = (j2 == 0 && lu == 0) ? G::HALF : ((M - k) & (M - 1));
                    dz[zpadT<R, T>(k)] = xr[2 * j2];
                    dz[zpadT<R, T>(kk)] = xr[2 * j2 + 1];
                }
            }
            wave_sync();
        }
        if (pend) {
            // ---- the frame above (staged last) takes its previous spectrum from this row
            if (lane < pend_nk) {
                const int sl = pend_off + lane;
                const int nbin = Lbin[sl];
                cx<T> pv = mkc<T>((T)0, (T)0);
                if (zero_row) { if (pend_prev0) pv = mkc<T>((T)p.prev0[2 * nbin], (T)p.prev0[2 * nbin + 1]); }
                else pv = dz[XA(nbin)];
                sval_put(sl, 2, pv.x); sval_put(sl, 3, pv.y);
            }
            pend = false;
        }
        if (with_peaks) {
            if (ng == GFV || nst + K > kSlots) { flush(ng); ng = 0; nst = 0; }
            const unsigned long long bal = __ballot(keep);
            if (keep) {
                const int sl = nst + lane_prefix(bal);
                const cx<T> c = dz[XA(pb)], vm = dz[XA(pb - 1)], vp = dz[XA(pb + 1)];
                // PV.py:197-199: 3-bin energy, bin 0 excluded (1 <= pb <= M-2); |X|^2 by the untangle's expression
                T s3 = (T)0;
                if (pb > 1) s3 = s3 + (vm.x * vm.x + vm.y * vm.y);
                s3 = s3 + (c.x * c.x + c.y * c.y);
                s3 = s3 + (vp.x * vp.x + vp.y * vp.y);
                Lbin[sl] = pb;
                sval_put(sl, 0, c.x); sval_put(sl, 1, c.y); sval_put(sl, 4, s3);
            }
            const int nk = __popcll(bal);
            if (lane == 0) { Lcnt[ng] = nk; Loff[ng] = nst; Lfrm[ng] = gq - 1; Lorow[ng] = (long long)orow; Ltot[ng] = tot; }
            pend = true; pend_prev0 = (p.prev0 != nullptr) && (orow == 0);
            pend_off = nst; pend_nk = nk;
            nst += nk; ng++;
        }
        if (g == spec_g && !zero_row) {
            gdouble* const so = (gdouble*)p.spec_out;
#pragma unroll
            for (int j = 0; j < R; j++) {
                const cx<T> v = dz[XA(lane + 64 * j)];
                so[2 * (lane + 64 * j)] = v.x;
                so[2 * (lane + 64 * j) + 1] = v.y;
            }
        }
        wave_sync();                                                  // dz is read: free for the row below
        gq = qn;
        if (nsrc != nullptr) { csrc = nsrc; orow = norow; }
    }
    if (ng > 0) flush(ng);
}

template <int R, bool SYM, int NW> int launch_rv(const PvRevParams& p, int x_dtype, hipStream_t s) {
    using G = RvGeo<R, SYM, !((R == 8 && NW > 8) || (R == 4 && NW > 12))>;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    if (p.total_rows >= 0x7fffff00LL) { pvx_set_error("the float64 analysis kernel indexes rows in 32 bits (%lld rows)", (long long)p.total_rows); return PVX_ERR_UNSUPPORTED; }
    const size_t lds = G::total(NW);
    static_assert(G::total(NW) <= 160 * 1024, "a workgroup fits the CU's LDS");
    const int H = (p.hop == 32 * R) ? R / 4 : (p.hop == 64 * R) ? R / 2 : 0;
    const void* fn = nullptr;
#define PVX_RV_PICK(INT) (H == R / 4 ? (const void*)k_pv_rev<R, INT, R / 4, SYM, NW> : H ? (const void*)k_pv_rev<R, INT, R / 2, SYM, NW> : (const void*)k_pv_rev<R, INT, 0, SYM, NW>)
#define PVX_RV_PICK_H(INT) (H == R / 4 ? (const void*)k_pv_rev<R, INT, R / 4, SYM, NW> : (const void*)k_pv_rev<R, INT, R / 2, SYM, NW>)
    if constexpr (SYM) {
        if (H == 0) { pvx_set_error("the half-window form takes the sliding-window hops only"); return PVX_ERR_UNSUPPORTED; }
        switch (x_dtype) {
            case PVX_F32: fn = PVX_RV_PICK_H(float); break;
            case PVX_F64: fn = PVX_RV_PICK_H(double); break;
            case PVX_I16: fn = PVX_RV_PICK_H(int16_t); break;
            default: pvx_set_error("bad x_dtype %d", x_dtype); return PVX_ERR_INVALID;
        }
    } else if constexpr (R == 8 && NW > 8) {
        // three waves per SIMD at the sliding-window hops (any other hop holds a whole next row in flight: two per SIMD)
        if (H == 0) return launch_rv<R, false, 8>(p, x_dtype, s);
        switch (x_dtype) {
            case PVX_F32: fn = PVX_RV_PICK_H(float); break;
            case PVX_F64: fn = PVX_RV_PICK_H(double); break;       // (22 registers in scratch -- the next row's sample pairs -- and still +12 .. 20 % over eight waves)
            case PVX_I16: fn = PVX_RV_PICK_H(int16_t); break;
            default: pvx_set_error("bad x_dtype %d", x_dtype); return PVX_ERR_INVALID;
        }
    } else {
        switch (x_dtype) {
            case PVX_F32: fn = PVX_RV_PICK(float); break;
            case PVX_F64:
                if constexpr (R == 16) {
                    // (float64 samples at nfft 2048 with a hop that does not slide the window: pvx_pv_rev_takes() sends that plan elsewhere)
                    if (H == 0) { pvx_set_error("k_pv_rev does not take float64 samples at nfft 2048 with hop %d", p.hop); return PVX_ERR_UNSUPPORTED; }
                    fn = PVX_RV_PICK_H(double);
                } else fn = PVX_RV_PICK(double);
                break;
            case PVX_I16: fn = PVX_RV_PICK(int16_t); break;
            default: pvx_set_error("bad x_dtype %d", x_dtype); return PVX_ERR_INVALID;
        }
    }
#undef PVX_RV_PICK
#undef PVX_RV_PICK_H
    if (lds > 64 * 1024) PVX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t nblocks = ncu;                                             // one workgroup per CU: its waves share the tables
    if (p.blocks_override > 0) nblocks = p.blocks_override;
    const int64_t maxb = (p.row_end - p.row_begin + NW - 1) / NW;     // never more waves than rows
    if (nblocks > maxb) nblocks = maxb > 0 ? maxb : 1;
    if (!G::LV && (p.stage == nullptr || p.stage_bytes < (size_t)nblocks * NW * kSlots * 5 * 8)) {
        pvx_set_error("k_pv_rev: the staging block holds %zu bytes, %zu needed", p.stage_bytes, (size_t)nblocks * NW * kSlots * 5 * 8);
        return PVX_ERR_INVALID;
    }
    dim3 grid((unsigned)nblocks), block(64 * NW);
    PvRevParams arg = p;
    void* args[] = {&arg};
    PVX_HIP_CHECK(hipLaunchKernel(fn, grid, block, args, lds, s));
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}

}  // namespace

// nfft 512 / 1024 / 2048 at precision 64, npks <= 64 (a frame's kept peaks fit the 64 staging slots)
int pvx_pv_rev_supported(int nfft, int precision, int K) {
    if (precision != 64 || K < 1 || K > kSlots) return 0;
    return nfft == 512 || nfft == 1024 || nfft == 2048;
}
int pvx_pv_rev_takes(int nfft, int x_dtype, int hop) {
    if (nfft == 2048 && x_dtype == PVX_F64 && hop != 512 && hop != 1024) return 0;
    return 1;
}
// bytes of global staging a launch may need (nfft 2048: the kept peaks' values; 0 elsewhere)
size_t pvx_pv_rev_stage_bytes(int nfft) {
    if (nfft != 2048 && nfft != 1024 && !(nfft == 512 && PVX_PVR_NW512 > 12)) return 0;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    if (ncu < 256) ncu = 256;
    return (size_t)ncu * 16 * kSlots * 5 * 8;
}

int pvx_launch_pv_rev(const PvRevParams& p, int nfft, int x_dtype, hipStream_t s) {
    if (p.total_rows <= 0 || p.row_end <= p.row_begin) return PVX_OK;
    if (p.row_begin < 0 || p.row_end > p.total_rows) { pvx_set_error("k_pv_rev: rows [%lld, %lld) of %lld", (long long)p.row_begin, (long long)p.row_end, (long long)p.total_rows); return PVX_ERR_INVALID; }
    if (p.K > kSlots) { pvx_set_error("k_pv_rev stages at most %d peaks per frame (npks = %d)", kSlots, p.K); return PVX_ERR_UNSUPPORTED; }
    switch (nfft) {
        case 512: return launch_rv<4, false, PVX_PVR_NW512>(p, x_dtype, s);
        // nfft 1024: twelve waves per CU at the sliding-window hops (+12 .. 20 % over eight: profiles/r06_ab_steps.txt); other hops stay at eight
        case 1024: return launch_rv<8, false, PVX_PVR_NW1024>(p, x_dtype, s);
        case 2048: {
            // a symmetric window keeps its first half in LDS: an eighth wave per CU (k_stft_pv.hip)
            const bool sym = p.win_symmetric != 0 && (p.hop == 512 || p.hop == 1024) && getenv("PVX_STFT_PV_NOSYM") == nullptr;
            return sym ? launch_rv<16, true, 8>(p, x_dtype, s) : launch_rv<16, false, 7>(p, x_dtype, s);
        }
        default: break;
    }
    pvx_set_error("the float64 descending-order kernel does not handle nfft=%d", nfft);
    return PVX_ERR_UNSUPPORTED;
}
