// k_fused_team.hip -- the fused analysis stage (window + FFT + untangle + peaks) for nfft 4096 / 8192 in the shape of
// k_fused_rev.hip: a TEAM of S waves per frame (S = 2: nfft 4096, S = 4: nfft 8192), every wave doing what a wave of
// k_fused_rev<16> does -- a 1024-point complex transform in registers + LDS, the peak search over 1024 bins, the staging
// and the per-peak arithmetic of ITS OWN peaks -- and three workgroup barriers per frame where the waves meet.
//
//   PV.calc_fft_frame   pypevoc/PVAnalysis.py:150-158
//   PV.calc_pv_frame    pypevoc/PVAnalysis.py:160-211      (PeakFinder: pypevoc/PeakFinder.py:155-194, 113-136)
//   PV.run_pv           pypevoc/PVAnalysis.py:213-264
//
// The transform.  z[j] = (x w)[2j] + i (x w)[2j+1], j < M = nfft/2 = S L, L = 1024.  Wave s takes the sub-sequence
// z[S i + s] (decimation in time) and transforms it as k_fused_rev<16> does (pvx_fft4.h): four 256-point transforms, one
// per 16-lane group, radix-16 registers -> LDS transpose inside the group -> radix-16 registers, no cross-lane stage.
// After barrier B1 the join and the real-FFT untangle run as ONE pass, in place, the pairs of the radix-J join being
// exactly the untangle pairs (Z[k], conj Z[M-k]) (pvx_fft4.h):
//   S = 2: the 8 quarters of the two waves are joined by a radix-8 pass (join8_untangle): a lane reads the 16 values of
//          k1 and 256 - k1 and writes 16 bins back to the same slots -- the LDS traffic of k_fused_rev's untangle;
//   S = 4: every wave first joins its own four quarters (join4_plain, in place), the team's radix-4 join then runs inside
//          the untangle (join4_untangle over the waves' regions).
// The peaks.  After B2 (row complete, waves' max / min / energy exchanged) every wave scans ITS 1024 bins, owns its
// candidates (one per lane; dense segments are first thinned / reduced to their npks best, k_fused_rev's code) and
// fetches in one LDS round trip all it needs of them; the waves exchange their candidates' keys (B3) and every wave
// ranks its own against all (score desc, bin asc: PeakFinder's repeated arg-max), applies the salience filter and
// stages its kept peaks.  Rows are walked in DESCENDING order over the one spectrum buffer (k_fused_rev.hip): the
// previous-spectrum values of a frame's peaks are picked up one row later.  Nothing reads the spectrum or the |X|^2
// row after B3, so the next row's transform may overwrite them: three barriers per frame, all waves equally loaded,
// no wave waits for another one's serial section (k_fused_mw.hip: ~9 barriers and wave 0 alone after the scan).
// Every staged frame is flushed by all waves together (one more barrier per 8 frames): a wave does the per-peak
// arithmetic of its own peaks, the waves exchange how many of theirs were emitted (freq > 0, PV.py:193) and write
// them behind those of the waves below them -- rows stay left-packed in ascending bin order (PV.py:226-239).
// npks <= 64: one candidate per lane (PPL = 1).  64 < npks <= 128 (BASELINE config 3: npks = 100): TWO per lane (PPL = 2) --
// lane l owns entries l and l + 64 of its wave's list, everything a candidate carries through B3 exists twice, the key rows
// are 128 long, the per-peak pass takes a wave's staged peaks in two rounds.  Larger npks: the general path (k_fused_mw.hip,
// which took them until round 6, is a witness kernel).
#include <stdlib.h>

#include "pvx_fft4.h"

using namespace pvxw;
using namespace pvxf;

namespace {

constexpr int GFT = 8;              // frames staged before the per-peak pass

typedef unsigned short u16;
// a kept peak's results: non-temporal stores (written once, read by nobody in the launch: +1.3 % at nfft 2048 / npks 8 for leaving the caches
// to the samples and the hand-over; -DPVX_RESULTS_TEMPORAL=1: plain stores).  The zero padding of a row stays with plain stores: scattered
// 8-byte non-temporal writes cost nfft 1024 at npks 20 (one peak, nineteen zeros per row) 12 %.
#ifdef PVX_RESULTS_TEMPORAL
#define PVX_RST(ptr, idx, val) ((ptr)[idx] = (val))
#else
#define PVX_RST(ptr, idx, val) __builtin_nontemporal_store((double)(val), &(ptr)[idx])
#endif

// LDS hand-off between the waves of the team: own LDS traffic drained, then the workgroup barrier (not
// __syncthreads(): that would also wait for the prefetched samples, vmcnt(0))
__device__ __forceinline__ void team_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// issue priorities of the frame loop's phases (see the loop; -D overrides for A/B builds)
#ifndef PVX_TEAM_PRIO_T
#define PVX_TEAM_PRIO_T 2
#endif
#ifndef PVX_TEAM_PRIO_S
#define PVX_TEAM_PRIO_S 1
#endif
#ifndef PVX_TEAM_PRIO_C
#define PVX_TEAM_PRIO_C 0
#endif
template <int S, int PPL = 1, bool DENSE = false> struct TeamGeo {
    static constexpr int L = 1024;                       // complex points per wave
    static constexpr int M = L * S;                      // bins 0..M-1
    static constexpr int N = 2 * M;                      // nfft
    static constexpr int T = 64 * S;                     // lanes per team
    static constexpr int REG = F4::BUF;                  // complex slots per wave region (4 quarters of 272)
    static constexpr int YLEN = M + (M >> 6) * 4;        // |X|^2 row, padded layout (ymap<1>)
    static constexpr int CAPW = L / 2 + 4;               // candidate list capacity per wave
    static constexpr int CW = 64 * PPL;                  // candidates a wave carries through B3 (PPL per lane)
    static constexpr int TWN = 512 * S;                  // join / untangle twiddles of the team, lane-ordered (global table)
    static constexpr int J4N = (S == 4) ? 4 * 3 * 64 : 0;   // S = 4: twiddles of the in-wave radix-4 join
    // block-shared (bytes)
    static constexpr size_t OFF_T1 = 0;                                  // v2f [16][16]  W_256^(l q)
    static constexpr size_t OFF_J4 = OFF_T1 + 256 * 8;                   // v2f [4][3][64] W_1024^(u k1) (S = 4)
    static constexpr size_t OFF_X = OFF_J4 + (size_t)J4N * 8;            // float2 [S][REG]
    static constexpr size_t OFF_Y = OFF_X + (size_t)S * REG * 8;         // float [YLEN]
    static constexpr size_t OFF_KEYS = (OFF_Y + (size_t)YLEN * 4 + 15) & ~(size_t)15;     // u64 [S][CW] (score, tie-break); before the keys are written: the scan's trash slots
    static constexpr size_t OFF_PSUM = OFF_KEYS + (size_t)S * CW * 8;    // double [S]
    static constexpr size_t OFF_MISC = OFF_PSUM + (size_t)S * 8;         // int nw[S] | float pmax[S] | float pmin[S] | int val[S][GFT]
    static constexpr size_t OFF_WAVE = (OFF_MISC + (size_t)S * 4 * (3 + GFT + (DENSE ? 1 : 0)) + 15) & ~(size_t)15;       // (DENSE: + int nst[S])
    // PPL = 2 stages one frame at a time (npks > 64) and flushes it before the next scan: the candidate list and the staged
    // values are never alive together and share their bytes -- what lets two teams of nfft 8192 into a CU's LDS at npks 128
    __host__ __device__ static size_t ci_or_sval(int K) {
        const size_t kpad = (size_t)((K + 3) & ~3), ci = ((size_t)CAPW * 2 + 15) & ~(size_t)15, sval = (size_t)staged_frames(K, GFT) * kpad * 5 * 4;
        return ci > sval ? ci : ((sval + 15) & ~(size_t)15);
    }
    __host__ __device__ static size_t per_wave(int K) {
        const size_t kpad = (size_t)((K + 3) & ~3);
        const size_t gs = (size_t)staged_frames(K, GFT);
        // DENSE (8 < npks <= 24, PPL = 1): 64 slots per wave, a frame's kept peaks behind the previous frame's (k_fused_rev.hip), + off[GFT]
        const size_t slots = DENSE ? (size_t)64 : gs * kpad;
        size_t b = GFT * 8 + GFT * 4 * 2 + (DENSE ? GFT * 4 : 0)         // tot | orow | cnt | (off)
                 + kpad * 4 + slots * 4;                                 // sel | sbin
        if (PPL == 2) b += ci_or_sval(K);
        else b += (size_t)CAPW * 2 + slots * 5 * 4;                      // ci (u16) | sval
        return (b + 15) & ~(size_t)15;
    }
    __host__ __device__ static size_t total(int K) { return OFF_WAVE + per_wave(K) * S; }
};

template <int S, typename InT, bool AL2, int H, int PPL = 1, bool DENSE = false>
__global__ __launch_bounds__(64 * S, 2) void k_fused_team(FusedParams p) {
    static_assert(!DENSE || PPL == 1, "dense staging is for one candidate per lane");
    using TG = TeamGeo<S, PPL, DENSE>;
    constexpr int CW = TG::CW;
    constexpr int R = 16, L = TG::L, M = TG::M, T = TG::T, REG = TG::REG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lt = (int)threadIdx.x;                                // lane within the team
    const int K = p.K;
    const int kpad = (K + 3) & ~3;
    const int gs = staged_frames(K, GFT);

    v2f* const t1L = (v2f*)(smem + TG::OFF_T1);
    v2f* const j4L = (v2f*)(smem + TG::OFF_J4);
    float2* const X = (float2*)(smem + TG::OFF_X);
    float* const Ly = (float*)(smem + TG::OFF_Y);
    unsigned long long* const Lkeys = (unsigned long long*)(smem + TG::OFF_KEYS);
    double* const Lpsum = (double*)(smem + TG::OFF_PSUM);
    int* const Lnw = (int*)(smem + TG::OFF_MISC);
    float* const Lpmax = (float*)(Lnw + S);
    float* const Lpmin = Lpmax + S;
    int* const Lval = (int*)(Lpmin + S);
    int* const Lnst = Lval + S * GFT;                               // (DENSE) slots every wave has staged since the last per-peak pass
    // per-wave region (see k_fused_rev.hip for the opaque offset)
    unsigned wboff = (unsigned)(TG::OFF_WAVE + TG::per_wave(K) * wid);
    asm volatile("" : "+s"(wboff));
    unsigned char* wb = smem + wboff;
    double* const Ltot = (double*)wb;
    int* const Lorow = (int*)(Ltot + GFT);                          // output row (the launcher checks that rows fit 32 bits)
    int* const Lcnt = Lorow + GFT;
    int* const Lsoff = Lcnt + GFT;                                  // (DENSE) a staged frame's first slot
    u16* const Lci = (u16*)(Lsoff + (DENSE ? GFT : 0));
    // PPL = 1: ci | sel | sbin | sval.  PPL = 2: (ci or sval) | sel | sbin  (see per_wave)
    int* const Lsel = (PPL == 2) ? (int*)((unsigned char*)Lci + TG::ci_or_sval(K)) : (int*)(Lci + TG::CAPW);
    int* const Lsbin = Lsel + kpad;
    float* const Lsval = (PPL == 2) ? (float*)Lci : (float*)(Lsbin + (DENSE ? 64 : gs * kpad));
    const int trash0 = (int)(((const unsigned char*)(Lkeys + wid * CW) - (const unsigned char*)Lci) / 2);    // in u16 slots from Lci
    unsigned xoff = (unsigned)(TG::OFF_X + (size_t)REG * 8 * wid);
    asm volatile("" : "+s"(xoff));
    float2* const cur = (float2*)(smem + xoff);                     // this wave's region: exchange matrix, then E_s

    constexpr int NMASK = TG::N - 1;
    // ---- block-shared tables
    {
        const v2f* tab = (const v2f*)p.twiddle;                     // W_nfft^j, j < nfft; then the team table
        for (int i = lt; i < 256; i += T) t1L[i] = tab[((TG::N / 256) * (i & 15) * (i >> 4)) & NMASK];          // [q][l] W_256^(l q)
        if constexpr (S == 4) {
            for (int i = lt; i < TG::J4N; i += T) {                 // [j][u-1][lane]: W_1024^(u k1), k1 = lane + 64 j
                const int ln = i & 63, u = (i >> 6) % 3 + 1, k1 = ln + 64 * (i / 192);
                j4L[i] = tab[((TG::N / 1024) * u * k1) & NMASK];
            }
        }
    }
    __syncthreads();
    const v2f* const twg = (const v2f*)p.twiddle + TG::N;           // the team's join / untangle twiddles, lane-ordered

    // the wave's share of the window: pairs j = S (4 l + u + 64 r) + wid of lane 16 u + l (pvx_fft4.h)
    v2f wv[R];
#pragma unroll
    for (int r = 0; r < R; r++) wv[r] = ((const v2f*)p.win)[S * (lofs4(lane) / 2 + 64 * r) + wid];
#pragma unroll
    for (int r = 0; r < R; r++) asm volatile("" : "+v"(wv[r]));

    // ---- rows of this team: [r0, r1), walked downwards, then row r0 - 1 (spectrum only)
    const int NB = (int)gridDim.x;
    const int r0 = (int)(p.total_rows * (int64_t)blockIdx.x / NB), r1 = (int)(p.total_rows * ((int64_t)blockIdx.x + 1) / NB);
    if (r0 >= r1) return;                                           // block-uniform
    const int Fi = (int)p.F;
    const int rows1 = Fi + 1;                                       // rows per signal

    // (in the CONSTANT address space: scalar loads.  As a generic pointer these were flat loads, and the result stores below
    // flat stores: with flat operations pending the compiler can only wait with vmcnt(0), and it did so at the start of
    // every frame's peak search -- a wait for the sample prefetch of the NEXT frame, a quarter of the wave's time)
    typedef const __attribute__((address_space(4))) FusedParams* kargs_t;
    typedef __attribute__((address_space(1))) double gdouble;
    const kargs_t kargs = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();

    v2f raw[R];
#pragma unroll
    for (int r = 0; r < R; r++) raw[r] = pvxc::splat(0.f);
    auto row_src = [&](int gn, int bn, int qn) -> const InT* {
        if (gn < r0 - 1 || gn < 0 || qn == 0) return nullptr;
        return (const InT*)p.x + (int64_t)bn * p.sig_stride + (int64_t)(qn - 1) * p.hop;
    };
    auto load_pair = [&](const InT* src, int r) {
        const InT* q = src + S * lofs4(lane) + 2 * wid + 128 * S * r;
        if constexpr (AL2 && sizeof(InT) == 4) raw[r] = *(const v2f*)q;
        else raw[r] = pvxc::mk(ld1(q), ld1(q + 1));
    };
    auto prefetch_part = [&](const InT* src, int part) {
        if (src == nullptr) return;
        constexpr int PR = R / 4;
#pragma unroll
        for (int r = part * PR; r < (part + 1) * PR; r++) load_pair(src, r);
    };

    // spectrum of the team's row into X (zeros for a zero row) + |X|^2 -> Ly, team-reduced max / min / energy.
    // Barriers B1 and B2 are in here; fetches this wave's samples of the row below (nsrc) on the way.
    auto spectrum = [&](bool zero_row, const InT* nsrc, float& maxe, float& mine, double& tot) {
        v2f z[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            z[r] = raw[r] * wv[r];
            asm volatile("" : "+v"(z[r]));                          // the multiply stays above the loads
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (H > 0) {
            // (k_fused_rev.hip) the row below is the previous frame: the lane's pairs move up by H registers, the hop's
            // new ones come in below -- on every row, without a branch
            const InT* ns = (nsrc != nullptr) ? nsrc : (const InT*)p.x;
#pragma unroll
            for (int r = R - 1; r >= H; r--) raw[r] = raw[r - H];
#pragma unroll
            for (int r = 0; r < H; r++) load_pair(ns, r);
            nsrc = nullptr;
        }
        prefetch_part(nsrc, 0);
        if (zero_row) {
            prefetch_part(nsrc, 1); prefetch_part(nsrc, 2); prefetch_part(nsrc, 3);
#pragma unroll
            for (int j = 0; j < REG / 64; j++) cur[lane + 64 * j] = make_float2(0.f, 0.f);
            team_sync();
            return;
        }
        v2f* dz = (v2f*)cur;
        // the wave's four 256-point transforms -> quarter u of its region, natural order (pvx_fft4.h)
        fft4_quarters(z, dz, t1L, lane, [&]() { prefetch_part(nsrc, 1); }, [&]() { prefetch_part(nsrc, 2); }, [&]() { prefetch_part(nsrc, 3); });
        v2f* const xz = (v2f*)X;
        float lmax = -INFINITY, lmin = INFINITY, ls0 = 0.f, ls1 = 0.f;
        if constexpr (S == 2) {
            // the lane's join / untangle twiddles: fetched before the barrier (from L2), their latency under its wait
            __builtin_amdgcn_sched_barrier(0);
            v2f twv[8];
#pragma unroll
            for (int c = 0; c < 8; c++) twv[c] = twg[c * T + lt];
            team_sync();                                            // ---- B1: all eight quarters are in place
            join8_untangle(xz, Ly, twv, lt, wid == 0, lmax, lmin, ls0, ls1);
        } else {
            wave_sync();
            join4_plain(dz, j4L, lane);                             // the wave's 1024-point result, in place
            __builtin_amdgcn_sched_barrier(0);
            v2f twv[2][4];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int u = 0; u < 4; u++) twv[j][u] = twg[(j * 4 + u) * T + lt];
            team_sync();                                            // ---- B1: every E_s is in place
            join4_untangle<1024, F4::BUF, 256>(xz, Ly, twv, lt, Xa4IA(), lmax, lmin, ls0, ls1);
        }
        {
            float wm, wn;
            double wsum;
            wave_max_min_sum_nn(lmax, lmin, (double)ls0 + (double)ls1, wm, wn, wsum);
            if (lane == 0) { Lpmax[wid] = wm; Lpmin[wid] = wn; Lpsum[wid] = wsum; }
        }
        team_sync();                                                // ---- B2: X, |X|^2 and the partial reductions
        float mx = Lpmax[0], mn = Lpmin[0];
        double sm = Lpsum[0];
#pragma unroll
        for (int w = 1; w < S; w++) { mx = fmaxf(mx, Lpmax[w]); mn = fminf(mn, Lpmin[w]); sm += Lpsum[w]; }
        maxe = mx; mine = mn; tot = sm;
    };

    // per-peak pass over the staged frames [0, ng): every wave its own peaks; PPL entries per lane
    int LPF = 1;
    while (LPF < K && LPF < 64) LPF <<= 1;
    auto flush = [&](int ng) {
        wave_sync();
        // (the lane's group and ballot mask, worked out per flush: as loop invariants they are registers held through the frame
        // loop, which has none to spare -- k_fused_rev.hip)
        const int lnf = fresh_lane();
        const int gl = lnf / LPF, e0 = lnf - gl * LPF;
        const unsigned long long gmask = (LPF == 64 ? ~0ull : ((1ull << LPF) - 1ull)) << (gl * LPF);
        kargs_t q = kargs;
        asm volatile("" : "+s"(q));                                  // loads through q stay here
        PeakConst pc;
        pc.fstep = q->fstep; pc.dt = q->dt; pc.nfft = TG::N; pc.hop = q->hop; pc.wfbin = q->wfbin;
        if constexpr (DENSE) {
            // ---- this wave's staged peaks sit back to back in its 64 slots (k_fused_rev.hip's dense pass); the waves of the team exchange
            // per FRAME how many each emitted and write theirs behind those of the waves below
            static_assert(GFT == 8, "two 16-byte reads per table");
            const int4 oa = *(const int4*)Lsoff, ob4 = *((const int4*)Lsoff + 1), ca = *(const int4*)Lcnt, cb4 = *((const int4*)Lcnt + 1);
            const int offs[8] = {oa.x, oa.y, oa.z, oa.w, ob4.x, ob4.y, ob4.z, ob4.w}, cnts[8] = {ca.x, ca.y, ca.z, ca.w, cb4.x, cb4.y, cb4.z, cb4.w};
            int top = 0, cnt = 0, g = 0, start = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) { if (j == ng - 1) top = offs[j] + cnts[j]; }
            cnt = cnts[0];
#pragma unroll
            for (int j = 1; j < 8; j++) { if (j < ng && offs[j] <= lnf) { g = j; start = offs[j]; cnt = cnts[j]; } }
            bool valid = lnf < top;
            int nbin = 0;
            PeakOut o;
            o.freq = 0.0; o.dfb = 0.0; o.thisph = 0.0; o.mag = 0.0; o.valid = false;
            if (valid) {
                nbin = Lsbin[lnf];
                const float* sv = Lsval + (size_t)lnf * 5;
                o = peak_math<float, true>(nbin, sv[0], sv[1], sv[2], sv[3], sv[4], pc);
                valid = o.valid;
            }
            const unsigned long long ball = __ballot(valid);
            const unsigned long long gm = (cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull)) << start;
            // eight lanes per staged frame: frame g2, column offset c2
            const int g2 = lnf >> 3, c2 = lnf & 7;
            int o2 = 0, n2 = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) { if (j == g2 && j < ng) { o2 = offs[j]; n2 = cnts[j]; } }      // (frames from ng on: slots nobody wrote this batch -- a shift by their "offset" is undefined)
            const unsigned long long gm2 = (n2 >= 64 ? ~0ull : ((1ull << n2) - 1ull)) << (o2 & 63);
            if (c2 == 0) Lval[wid * GFT + g2] = (g2 < ng) ? __popcll(ball & gm2) : 0;
            team_sync();                                             // the waves' emitted counts, frame by frame
            if (valid) {
                int off = 0;
#pragma unroll
                for (int w = 0; w < S; w++) off += (w < wid) ? Lval[w * GFT + g] : 0;
                const int64_t orow = (int64_t)Lorow[g];
                const int oi = off + __popcll(ball & gm & ((1ull << lnf) - 1ull));
                PVX_RST(((gdouble*)q->binno + orow * K), oi, (double)nbin);
                PVX_RST(((gdouble*)q->f + orow * K), oi, o.freq);
                PVX_RST(((gdouble*)q->mag + orow * K), oi, o.mag);
                PVX_RST(((gdouble*)q->ph + orow * K), oi, o.thisph);
                PVX_RST(((gdouble*)q->realph + orow * K), oi, o.thisph + kPi * o.dfb / pc.fstep);      // PV.py:207
            }
            if (g2 < ng && (wid == S - 1 || (wid == 0 && c2 == 0))) {
                const int64_t orow2 = (int64_t)Lorow[g2];
                if (wid == S - 1) {
                    int tot2 = 0;
#pragma unroll
                    for (int w = 0; w < S; w++) tot2 += Lval[w * GFT + g2];
                    gdouble* of2 = (gdouble*)q->f + orow2 * K; gdouble* om2 = (gdouble*)q->mag + orow2 * K; gdouble* op2 = (gdouble*)q->ph + orow2 * K;
                    gdouble* orp2 = (gdouble*)q->realph + orow2 * K; gdouble* ob2 = (gdouble*)q->binno + orow2 * K;
                    for (int j = tot2 + c2; j < K; j += 8) { ob2[j] = 0.0; of2[j] = 0.0; om2[j] = 0.0; op2[j] = 0.0; orp2[j] = 0.0; }     // zero padding, PV.py:226-239
                }
                if (wid == 0 && c2 == 0) {
                    const int64_t fr = orow2 % (int64_t)Fi;                                               // frame within its signal
                    if (q->totalmag) ((gdouble*)q->totalmag)[orow2] = sqrt(Ltot[g2]);                                  // PV.py:210
                    if (q->t) ((gdouble*)q->t)[orow2] = ((double)(fr * (int64_t)pc.hop) + TG::N / 2.0) / q->sr;        // PV.py:247
                }
            }
            if (lnf == 0) Lnst[wid] = 0;
            wave_sync();
            return;
        }
        const int g = gl;
        const bool gvalid = g < ng;
        const int cnt = gvalid ? Lcnt[g] : -1;
        const int64_t orow = gvalid ? (int64_t)Lorow[g] : 0;
        // (PPL = 2: npks > 64, LPF = 64, one staged frame: the wave's own peaks e0 and e0 + 64, in two rounds)
        bool valid[PPL];
        int nbin[PPL];
        PeakOut o[PPL];
        unsigned long long bal[PPL];
        int nval = 0;
#pragma unroll
        for (int h = 0; h < PPL; h++) {
            const int e = e0 + 64 * h;
            valid[h] = (cnt >= 0) && (e < cnt);
            nbin[h] = 0;
            o[h].freq = 0.0; o[h].dfb = 0.0; o[h].thisph = 0.0; o[h].mag = 0.0; o[h].valid = false;
            if (valid[h]) {
                nbin[h] = Lsbin[g * kpad + e];
                const float* sv = Lsval + (size_t)(g * kpad + e) * 5;
                o[h] = peak_math<float, true>(nbin[h], sv[0], sv[1], sv[2], sv[3], sv[4], pc);
                valid[h] = o[h].valid;
            }
            bal[h] = __ballot(valid[h]) & gmask;
            nval += __popcll(bal[h]);
        }
        if (gvalid && e0 == 0) Lval[wid * GFT + g] = nval;
        team_sync();                                                 // the waves' emitted counts
        int off = 0, tot = 0;
        if (gvalid) {
#pragma unroll
            for (int w = 0; w < S; w++) { const int v = Lval[w * GFT + g]; off += (w < wid) ? v : 0; tot += v; }
        }
        gdouble* of = (gdouble*)q->f + orow * K;
        gdouble* om = (gdouble*)q->mag + orow * K;
        gdouble* op = (gdouble*)q->ph + orow * K;
        gdouble* orp = (gdouble*)q->realph + orow * K;
        gdouble* ob = (gdouble*)q->binno + orow * K;
#pragma unroll
        for (int h = 0; h < PPL; h++) {
            if (valid[h]) {
                const int oi = off + __popcll(bal[h] & ((1ull << lnf) - 1ull));
                PVX_RST(ob, oi, (double)nbin[h]);
                PVX_RST(of, oi, o[h].freq);
                PVX_RST(om, oi, o[h].mag);
                PVX_RST(op, oi, o[h].thisph);
                PVX_RST(orp, oi, o[h].thisph + kPi * o[h].dfb / pc.fstep);    // PV.py:207
            }
            off += __popcll(bal[h]);
        }
        if (gvalid) {
            if (wid == S - 1) {
                for (int j = tot + e0; j < K; j += LPF) {             // zero padding, PV.py:226-239
                    ob[j] = 0.0; of[j] = 0.0; om[j] = 0.0; op[j] = 0.0; orp[j] = 0.0;
                }
            }
            if (wid == 0 && e0 == 0) {
                const int64_t fr = orow % (int64_t)Fi;                                                // frame within its signal
                if (q->totalmag) ((gdouble*)q->totalmag)[orow] = sqrt(Ltot[g]);                                   // PV.py:210
                if (q->t) ((gdouble*)q->t)[orow] = ((double)(fr * (int64_t)pc.hop) + TG::N / 2.0) / q->sr;        // PV.py:247
            }
        }
        wave_sync();
    };

    // ---- (signal b, row-in-signal q) of the team's first row g = r1 - 1; rows go down by one
    int g = r1 - 1, gb = g / rows1, gq = g - gb * rows1;            // the only division
    { const InT* s0 = row_src(g, gb, gq); prefetch_part(s0, 0); prefetch_part(s0, 1); prefetch_part(s0, 2); prefetch_part(s0, 3); }
    int ng = 0, nst = 0;                                            // (nst, DENSE: slots this wave has staged since the last per-peak pass)
    bool pend = false, pend_prev0 = false;                          // the frame staged last still waits for its previous spectrum
    int pend_nk = 0, own[PPL];                                      // the lane's staged peaks: slot << 16 | bin, -1: none (one register each across the transform)
#pragma unroll
    for (int h = 0; h < PPL; h++) own[h] = -1;
    bool prev_zero = false;                                         // (H > 0) the row above was a zero row
    for (; g >= r0 - 1; --g) {
        int bn = gb, qn = gq - 1;                                   // (b, q) of row g - 1
        if (qn < 0) { qn = Fi; bn -= 1; }
        const bool zero_row = (g < 0) || (gq == 0);
        if constexpr (H > 0) {
            if (!zero_row && prev_zero) { const InT* s1 = row_src(g, gb, gq); prefetch_part(s1, 0); prefetch_part(s1, 1); prefetch_part(s1, 2); prefetch_part(s1, 3); }
            prev_zero = zero_row;
        }
        float maxe = 0.f, mine = 0.f;
        double tot = 0.0;
        // issue priority by phase, as in k_fused_rev.hip (the teams of a CU are in different phases): transform and join 2,
        // pick-up / flush / candidate scan 1, ranking and staging 0
        __builtin_amdgcn_s_setprio(PVX_TEAM_PRIO_T);
        spectrum(zero_row, row_src(g - 1, bn, qn), maxe, mine, tot);
        __builtin_amdgcn_s_setprio(PVX_TEAM_PRIO_S);
        if (pend) {
            // ---- the frame above (staged as group ng - 1) takes its previous spectrum from this row
            if (!pend_prev0) {
                if constexpr (PPL == 1) {
                    if (own[0] >= 0) {
                        const float2 pv = X[xa4(own[0] & 0xffff)];
                        Lsval[(size_t)(own[0] >> 16) * 5 + 2] = pv.x;
                        Lsval[(size_t)(own[0] >> 16) * 5 + 3] = pv.y;
                    }
                } else {
                    // (two per lane: the staged bins are read back from LDS -- nothing of them is held in registers across the
                    // transform, which has none to spare)
                    const int ln = fresh_lane();
#pragma unroll
                    for (int h = 0; h < PPL; h++) {
                        const int e = ln + 64 * h;
                        if (e < pend_nk) {
                            const int sl = (ng - 1) * kpad + e;
                            const float2 pv = X[xa4(Lsbin[sl])];
                            Lsval[(size_t)sl * 5 + 2] = pv.x;
                            Lsval[(size_t)sl * 5 + 3] = pv.y;
                        }
                    }
                }
            } else
            for (int e = fresh_lane(); e < pend_nk; e += 64) {      // (rare: the first frame of a streamed call)
                const int sl = (DENSE ? nst - pend_nk : (ng - 1) * kpad) + e;
                const int nbin = Lsbin[sl];
                Lsval[(size_t)sl * 5 + 2] = (float)p.prev0[2 * nbin];
                Lsval[(size_t)sl * 5 + 3] = (float)p.prev0[2 * nbin + 1];
            }
            pend = false;
            if constexpr (DENSE) {
                // (team-uniform: every wave reads all the waves' slot counts, written before the barriers of this row's transform)
                int need = 0;
#pragma unroll
                for (int w = 0; w < S; w++) { const int v_ = Lnst[w]; need = v_ > need ? v_ : need; }
                if (ng == GFT || need + K > 64) { flush(ng); ng = 0; nst = 0; }
            } else
            if (ng == gs) { flush(ng); ng = 0; }                    // team-uniform
        }
        const bool real = !zero_row && g >= r0;                     // team-uniform
        bool has[PPL], slow = false;
        int pb[PPL], bad[PPL], n_w = 0;
        unsigned mykey[PPL];
        float2 c[PPL], vm[PPL], vp[PPL];
#pragma unroll
        for (int h = 0; h < PPL; h++) { has[h] = false; pb[h] = 1; bad[h] = 0; mykey[h] = 0u; c[h] = make_float2(0.f, 0.f); vm[h] = c[h]; vp[h] = c[h]; }
        double th = 0.0;
        if (real) {
            // PeakFinder(famp, npeaks, minrattomax) + filter_by_salience(rad=5)  (PV.py:175-178); see k_fused.hip
            const float maxy = __builtin_amdgcn_sqrtf(maxe);
            const double minamp = (double)maxy * p.thr;             // PF.py:60
            th = (minamp != 0.0) ? minamp * minamp - (double)mine : 0.0;
            // this wave's candidates (ascending bins) -> Lci
            // (non-candidates go to a trash slot per lane: the wave's key row, which is written after the scan)
            const int C_w = peak_scan_seg_thin<R, u16>(Ly, L * wid, M, mine, th, Lci, trash0, lane, K);
            wave_sync();
            n_w = C_w;
            if (C_w > CW) {
                // a dense segment: its npks best first (the row's npks best are among the segments' npks best, same tie rule)
                n_w = peak_pick_regs<R / 2, 1, u16, true>(Ly, Lci, Lsel, M, K, C_w, th, mine, lane);
#pragma unroll
                for (int h = 0; h < PPL; h++) { has[h] = lane + 64 * h < n_w; pb[h] = has[h] ? Lsel[lane + 64 * h] : 1; }
            } else {
#pragma unroll
                for (int h = 0; h < PPL; h++) { has[h] = lane + 64 * h < n_w; pb[h] = has[h] ? (int)Lci[lane + 64 * h] : 1; }
            }
            // ---- lane c owns candidates c (and c + 64): one LDS round trip for their scores, the 2*rad neighbours of the
            // salience test (PF.py:126-134), the spectrum around them (k_fused_rev.hip)
            const int rad = p.rad;
            float v[PPL], nb[PPL][10];
#pragma unroll
            for (int h = 0; h < PPL; h++) {
                const int lo = pb[h] - rad > 1 ? pb[h] - rad : 1;
                int hi = pb[h] + rad < M ? pb[h] + rad : M;
                hi = hi > M - 1 ? M - 1 : hi;
                v[h] = Ly[ymap<1>(pb[h])];
#pragma unroll
                for (int d = 1; d <= 5; d++) {
                    const int dd = d > rad ? rad : d;
                    int j0 = pb[h] - dd, j1 = pb[h] + dd;
                    j0 = j0 < lo ? lo : j0;
                    j1 = j1 > hi ? hi : j1;
                    nb[h][2 * d - 2] = Ly[ymap<1>(j0)];
                    nb[h][2 * d - 1] = Ly[ymap<1>(j1)];
                }
                c[h] = X[xa4(pb[h])];
                vm[h] = X[xa4(pb[h] - 1)]; vp[h] = X[xa4(pb[h] + 1)];
            }
#pragma unroll
            for (int h = 0; h < PPL; h++) {
                if constexpr (PPL == 2) {
                    // (two candidates per lane: the 3-bin energy now, the same sums in the same order -- four registers less through B3 and the ranking)
                    const float em = (pb[h] > 1) ? __builtin_fmaf(vm[h].x, vm[h].x, vm[h].y * vm[h].y) : 0.f;
                    vm[h].x = (em + __builtin_fmaf(c[h].x, c[h].x, c[h].y * c[h].y)) + __builtin_fmaf(vp[h].x, vp[h].x, vp[h].y * vp[h].y);
                }
#pragma unroll
                for (int d = 0; d < 10; d++) bad[h] |= (int)(nb[h][d] > v[h]);
                mykey[h] = has[h] ? __float_as_uint(v[h] - mine) : 0u;      // scores >= 0: bits order like values
                Lkeys[wid * CW + lane + 64 * h] = ((unsigned long long)mykey[h] << 32) | (unsigned)(CW * (S - 1 - wid) + CW - 1 - lane - 64 * h);
            }
            if (lane == 0) Lnw[wid] = n_w;
        }
        if (p.spec_out != nullptr && g == p.spec_row) {
#pragma unroll
            for (int j = 0; j < M / T; j++) {
                const float2 v = X[xa4(lt + T * j)];
                p.spec_out[2 * (lt + T * j)] = v.x;
                p.spec_out[2 * (lt + T * j) + 1] = v.y;
            }
        }
        team_sync();                                                // ---- B3: keys exchanged; X and |X|^2 are free for the row below
        __builtin_amdgcn_s_setprio(PVX_TEAM_PRIO_C);
        if (real) {
            int ctot = 0;
#pragma unroll
            for (int w = 0; w < S; w++) ctot += Lnw[w];
            slow = (th < 0.0) && (ctot < K);                        // zeros of pkmskamp qualify too (PF.py:166-187): wave 0, below
            bool take[PPL];
#pragma unroll
            for (int h = 0; h < PPL; h++) take[h] = has[h];
            if (ctot > K) {
                // rank of this lane's candidate among all the team's: larger score, or equal score and lower bin -- one
                // compare of (score, CW (S - 1 - wave) + CW - 1 - entry): unique keys, four list entries per trip (the entries
                // behind a wave's list hold score 0: below every candidate's)
                // (the keys are read where they are: a 16-byte LDS read at a wave-uniform address hands two of them to all
                // lanes -- a compare and an add per entry instead of a `v_readlane` broadcast and three more instructions)
                unsigned long long my64[PPL];
                int rank[PPL];
#pragma unroll
                for (int h = 0; h < PPL; h++) { my64[h] = ((unsigned long long)mykey[h] << 32) | (unsigned)(CW * (S - 1 - wid) + CW - 1 - lane - 64 * h); rank[h] = 0; }
#pragma unroll
                for (int w = 0; w < S; w++) {
                    const int n = Lnw[w];
                    const unsigned long long* kw = Lkeys + w * CW;
                    for (int j = 0; j < n; j += 4) {
                        const ulonglong2 ka = *(const ulonglong2*)(kw + j), kb = *(const ulonglong2*)(kw + j + 2);
#pragma unroll
                        for (int h = 0; h < PPL; h++)
                            rank[h] += (ka.x > my64[h] ? 1 : 0) + (ka.y > my64[h] ? 1 : 0) + (kb.x > my64[h] ? 1 : 0) + (kb.y > my64[h] ? 1 : 0);
                    }
                }
#pragma unroll
                for (int h = 0; h < PPL; h++) take[h] = has[h] && (rank[h] < K);
            }
            int nk = 0;
#pragma unroll
            for (int h = 0; h < PPL; h++) own[h] = -1;
            const int64_t orow = (int64_t)gb * Fi + (gq - 1);
            if (!slow) {
#pragma unroll
                for (int h = 0; h < PPL; h++) {
                    const bool keep = take[h] && (p.rad < 0 || bad[h] == 0);
                    const unsigned long long bal = __ballot(keep);
                    if (keep) {
                        const int sl = (DENSE ? nst : ng * kpad) + nk + lane_prefix(bal);       // entries l before entries l + 64: ascending bins
                        // PV.py:197-199: 3-bin energy, bin 0 excluded (1 <= pb <= M-2)
                        float s3;
                        if constexpr (PPL == 1) {
                            const float em = (pb[h] > 1) ? __builtin_fmaf(vm[h].x, vm[h].x, vm[h].y * vm[h].y) : 0.f;
                            s3 = (em + __builtin_fmaf(c[h].x, c[h].x, c[h].y * c[h].y)) + __builtin_fmaf(vp[h].x, vp[h].x, vp[h].y * vp[h].y);
                        } else s3 = vm[h].x;                         // (formed before B3, below)
                        Lsbin[sl] = pb[h];
                        float* sv = Lsval + (size_t)sl * 5;
                        sv[0] = c[h].x; sv[1] = c[h].y; sv[4] = s3;  // sv[2], sv[3]: the previous spectrum, one row later
                        if constexpr (PPL == 1) own[h] = (sl << 16) | pb[h];
                    }
                    nk += __popcll(bal);
                }
            } else {
                // fewer maxima than npks under a negative threshold: all maxima, then the first non-maximum interior
                // bins (peak_pick_regs' first branch, on the whole row) -- wave 0 alone, then one more barrier
                if (wid == 0) {
                    const int ln = fresh_lane();
                    const int nsel = peak_pick_regs<R / 2, 1, u16, true>(Ly, Lci, Lsel, M, K, ctot, th, mine, ln);
                    // (PPL = 2: the selected bins are read before anything is staged -- the staged values share the list's bytes,
                    // not Lsel's, but the order keeps this branch independent of the layout)
                    int sb2[PPL];
                    bool keep[PPL];
#pragma unroll
                    for (int h = 0; h < PPL; h++) {
                        const int e = ln + 64 * h;
                        sb2[h] = 1;
                        if (e < nsel) sb2[h] = Lsel[e];
                        keep[h] = (p.rad <= 8) ? salient_groups<1>(Ly, M, Lsel, 64 * h, nsel, p.rad, ln)
                                               : ((e < nsel) && salient<float, 1>(Ly, M, sb2[h], p.rad));
                    }
#pragma unroll
                    for (int h = 0; h < PPL; h++) {
                        const unsigned long long bal = __ballot(keep[h]);
                        if (keep[h]) {
                            const int sl = (DENSE ? nst : ng * kpad) + nk + lane_prefix(bal);
                            const float2 cc = X[xa4(sb2[h])];
                            const float2 cm = X[xa4(sb2[h] - 1)], cp = X[xa4(sb2[h] + 1)];
                            const float em = (sb2[h] > 1) ? __builtin_fmaf(cm.x, cm.x, cm.y * cm.y) : 0.f;
                            const float s3 = (em + __builtin_fmaf(cc.x, cc.x, cc.y * cc.y)) + __builtin_fmaf(cp.x, cp.x, cp.y * cp.y);
                            Lsbin[sl] = sb2[h];
                            float* sv = Lsval + (size_t)sl * 5;
                            sv[0] = cc.x; sv[1] = cc.y; sv[4] = s3;
                            if constexpr (PPL == 1) own[h] = (sl << 16) | sb2[h];
                        }
                        nk += __popcll(bal);
                    }
                }
                team_sync();
            }
            if (lane == 0) { Lcnt[ng] = nk; Lorow[ng] = (int)orow; Ltot[ng] = tot; if constexpr (DENSE) { Lsoff[ng] = nst; Lnst[wid] = nst + nk; } }
            if constexpr (DENSE) nst += nk;
            ng++;
            pend = true; pend_nk = nk; pend_prev0 = (p.prev0 != nullptr) && (orow == 0);
        }
        gb = bn; gq = qn;
    }
    if (ng > 0) flush(ng);
}

template <int S, int PPL, bool DENSE = false> int launch_team(const FusedParams& p, int x_dtype, hipStream_t s) {
    using TG = TeamGeo<S, PPL, DENSE>;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    if (p.total_rows >= 0x7fffff00LL) { pvx_set_error("the fused kernel indexes rows in 32 bits (%lld rows)", (long long)p.total_rows); return PVX_ERR_UNSUPPORTED; }
    if (p.K > 64 * PPL || (PPL == 2 && p.K <= 64) || p.rad > 5) { pvx_set_error("the team kernel takes npks <= 128 and rad <= 5 (npks=%d rad=%d)", p.K, p.rad); return PVX_ERR_UNSUPPORTED; }
    const size_t lds = TG::total(p.K);
    if (lds > 160 * 1024) { pvx_set_error("nfft=%d npks=%d needs %zu bytes of LDS in the team kernel", TG::N, p.K, lds); return PVX_ERR_UNSUPPORTED; }
    constexpr int R = 16;
    const bool al2 = (x_dtype == PVX_F32) && (p.hop % 2 == 0) && (p.sig_stride % 2 == 0) && (((uintptr_t)p.x) % 8 == 0);
    const int H = (p.hop == 32 * R * S) ? R / 4 : (p.hop == 64 * R * S) ? R / 2 : 0;
    const void* fn = nullptr;
#define PVX_TEAM_PICK(INT, AL) (H == R / 4 ? (const void*)k_fused_team<S, INT, AL, R / 4, PPL, DENSE> : H ? (const void*)k_fused_team<S, INT, AL, R / 2, PPL, DENSE> : (const void*)k_fused_team<S, INT, AL, 0, PPL, DENSE>)
    switch (x_dtype) {
        // (two candidates per lane: without the 8-byte sample loads' instantiations -- a row's new samples are a quarter of its loads)
        case PVX_F32: if constexpr (PPL == 1 && !DENSE) { fn = al2 ? PVX_TEAM_PICK(float, true) : PVX_TEAM_PICK(float, false); } else { fn = PVX_TEAM_PICK(float, false); } break;
        case PVX_I16: fn = PVX_TEAM_PICK(int16_t, false); break;
        default: pvx_set_error("the fused kernels take float32 or int16 samples (x_dtype %d: float64 is narrowed before the launch)", x_dtype); return PVX_ERR_INVALID;
    }
#undef PVX_TEAM_PICK
    if (lds > 64 * 1024) PVX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int blocks_per_cu = (int)((160 * 1024) / lds);
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    if (blocks_per_cu * S > 8) blocks_per_cu = 8 / S;               // two waves per SIMD
    int64_t nblocks = (int64_t)ncu * blocks_per_cu;
    if (p.blocks_override > 0) nblocks = p.blocks_override;
    if (nblocks > p.total_rows) nblocks = p.total_rows > 0 ? p.total_rows : 1;      // never more teams than rows
    dim3 grid((unsigned)nblocks), block(64 * S);
    FusedParams arg = p;
    void* args[] = {&arg};
    PVX_HIP_CHECK(hipLaunchKernel(fn, grid, block, args, lds, s));
    return PVX_OK;
}

}  // namespace

// lane-ordered twiddles of the team's join / untangle pass, appended to the plan's W_nfft^j table (N = nfft, M = N/2):
//   nfft 4096 (S = 2, 128 lanes, k1 = lt):            entry c * 128 + lt = W_N^k1 (c = 0), W_M^(c k1) (c = 1..7)
//   nfft 8192 (S = 4, 256 lanes, k1 = lt + 256 j):    entry (j * 4 + u) * 256 + lt = W_N^k1 (u = 0), W_M^(u k1) (u = 1..3)
int pvx_fused_team_table_len(int nfft) { return (nfft == 4096 || nfft == 8192) ? 512 * (nfft / 2048) : 0; }
void pvx_fused_team_table(int nfft, const float* tw /* [nfft][2] */, float* out) {
    const int S = nfft / 2048, T = 64 * S;
    const int nj = (S == 2) ? 1 : 2, nc = (S == 2) ? 8 : 4;
    for (int j = 0; j < nj; j++)
        for (int c = 0; c < nc; c++)
            for (int lt = 0; lt < T; lt++) {
                const int k1 = lt + T * j;
                const int idx = (c == 0 ? k1 : 2 * c * k1) & (nfft - 1);
                out[2 * ((j * nc + c) * T + lt)] = tw[2 * idx];
                out[2 * ((j * nc + c) * T + lt) + 1] = tw[2 * idx + 1];
            }
}

int pvx_fused_team_supported(int nfft, int precision, int K) {
    if (precision != 32 || K > 128) return 0;
    switch (nfft) {
        case 4096: return (K > 64 ? TeamGeo<2, 2>::total(K) : TeamGeo<2>::total(K)) <= 160 * 1024;
        case 8192: return (K > 64 ? TeamGeo<4, 2>::total(K) : TeamGeo<4>::total(K)) <= 160 * 1024;
        default: return 0;
    }
}

int pvx_launch_fused_team(const FusedParams& p, int nfft, int x_dtype, hipStream_t s) {
    if (p.total_rows <= 0) return PVX_OK;
    const bool dense = p.K > 8 && p.K <= 24 && getenv("PVX_TEAM_NO_DENSE") == nullptr;
    switch (nfft) {
        // (8 < npks <= 24: the dense staging of k_fused_rev.hip for the teams; PVX_TEAM_NO_DENSE=1: the strided one -- A/B, tests)
        case 4096: return p.K > 64 ? launch_team<2, 2>(p, x_dtype, s) : (dense ? launch_team<2, 1, true>(p, x_dtype, s) : launch_team<2, 1>(p, x_dtype, s));
        case 8192: return p.K > 64 ? launch_team<4, 2>(p, x_dtype, s) : (dense ? launch_team<4, 1, true>(p, x_dtype, s) : launch_team<4, 1>(p, x_dtype, s));
        default: pvx_set_error("the team kernel does not handle nfft=%d", nfft); return PVX_ERR_UNSUPPORTED;
    }
}
