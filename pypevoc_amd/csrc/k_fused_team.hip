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
// z[S i + s] (decimation in time), transforms it with k_fused_rev's stages (radix-16 registers, LDS exchange, radix-16
// registers, 4-lane DPP stage) and leaves E_s[k1], k1 < L, in natural order in region s of the team's spectrum buffer.
// After barrier B1 the join and the real-FFT untangle run as ONE pass, in place: with A = DFT_S(E_s[k1] W_M^(s k1)) and
// B = DFT_S(E_s[L-k1] conj(W_M^(s k1))) the pairs (A_t, B_((S-t) mod S)) are exactly the untangle pairs
// (Z[k], conj Z[M-k]) of k = k1 + L t, so a lane that reads the 2 S values of k1 and L - k1 writes the 2 S bins
// X[k1 + L t], X[(L - k1) + L (S-1-t)] back to the same 2 S slots: the same LDS traffic per lane as k_fused_rev's
// untangle, no extra pass for the join.  (k1 = 0 pairs with itself: its lane's mirrored slots take the k1 = L/2 family.)
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
// npks <= 64 (one staged peak per lane); larger npks stay with k_fused_mw.hip.
#include <stdlib.h>

#include "pvx_fft.h"

using namespace pvxw;
using namespace pvxf;

namespace {

constexpr int GFT = 8;              // frames staged before the per-peak pass

typedef unsigned short u16;

// LDS hand-off between the waves of the team: own LDS traffic drained, then the workgroup barrier (not
// __syncthreads(): that would also wait for the prefetched samples, vmcnt(0))
__device__ __forceinline__ void team_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

template <int S> struct TeamGeo {
    using G = Geo<16>;
    static constexpr int L = G::M;                       // 1024 complex points per wave
    static constexpr int M = L * S;                      // bins 0..M-1
    static constexpr int N = 2 * M;                      // nfft
    static constexpr int T = 64 * S;                     // lanes per team
    static constexpr int NPS = 512 / T;                  // (k1, L-k1) sets per lane in the join
    static constexpr int REG = G::BUFC;                  // complex slots per wave region
    static constexpr int YLEN = M + (M >> 6) * 4;        // |X|^2 row, padded layout (ymap<1>)
    static constexpr int CAPW = L / 2 + 4;               // candidate list capacity per wave
    static constexpr int TWN = NPS * S * T;              // join / untangle twiddles, lane-ordered
    // block-shared (bytes)
    static constexpr size_t OFF_T1 = 0;                                  // v2f [16][64]  W_L^(l q)
    static constexpr size_t OFF_T2 = OFF_T1 + 16 * 64 * 8;               // v2f [16][4]   W_64^(l1 t2)
    static constexpr size_t OFF_X = OFF_T2 + 64 * 8;                     // float2 [S][REG]
    static constexpr size_t OFF_Y = OFF_X + (size_t)S * REG * 8;         // float [YLEN]
    static constexpr size_t OFF_KEYS = OFF_Y + (size_t)YLEN * 4;         // u32 [S][64]; before the keys are written: the scan's trash slots
    static constexpr size_t OFF_PSUM = OFF_KEYS + (size_t)S * 64 * 4;    // double [S]
    static constexpr size_t OFF_MISC = OFF_PSUM + (size_t)S * 8;         // int nw[S] | float pmax[S] | float pmin[S] | int val[S][GFT]
    static constexpr size_t OFF_TW = (OFF_MISC + (size_t)S * 4 * (3 + GFT) + 15) & ~(size_t)15;
    __host__ __device__ static size_t off_wave(bool twl) { return OFF_TW + (twl ? (size_t)TWN * 8 : 0); }
    __host__ __device__ static size_t per_wave(int K) {
        const size_t kpad = (size_t)((K + 3) & ~3);
        const size_t gs = (size_t)staged_frames(K, GFT);
        size_t b = GFT * 8 + GFT * 4 * 2                                 // tot | orow | cnt
                 + (size_t)CAPW * 2                                      // ci (u16)
                 + kpad * 4 + gs * kpad * 4                              // sel | sbin
                 + gs * kpad * 5 * 4;                                    // sval
        return (b + 15) & ~(size_t)15;
    }
    __host__ __device__ static size_t total(int K, bool twl) { return off_wave(twl) + per_wave(K) * S; }
};

// bin k of the team's spectrum buffer: region k / L, k_fused_rev's padded natural order inside it
template <int S> __device__ __forceinline__ int xa(int k) { return (k >> 10) * TeamGeo<S>::REG + zpad<16>(k & 1023); }

template <int S, typename InT, bool AL2, int H, bool TWL>
__global__ __launch_bounds__(64 * S, 2) void k_fused_team(FusedParams p) {
    using G = Geo<16>;
    using TG = TeamGeo<S>;
    constexpr int R = 16, P = G::P, PITCH = G::PITCH, L = TG::L, M = TG::M, T = TG::T, NPS = TG::NPS, REG = TG::REG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lt = (int)threadIdx.x;                                // lane within the team
    const int K = p.K;
    const int kpad = (K + 3) & ~3;
    const int gs = staged_frames(K, GFT);

    v2f* const t1L = (v2f*)(smem + TG::OFF_T1);
    v2f* const t2L = (v2f*)(smem + TG::OFF_T2);
    float2* const X = (float2*)(smem + TG::OFF_X);
    float* const Ly = (float*)(smem + TG::OFF_Y);
    unsigned* const Lkeys = (unsigned*)(smem + TG::OFF_KEYS);
    double* const Lpsum = (double*)(smem + TG::OFF_PSUM);
    int* const Lnw = (int*)(smem + TG::OFF_MISC);
    float* const Lpmax = (float*)(Lnw + S);
    float* const Lpmin = Lpmax + S;
    int* const Lval = (int*)(Lpmin + S);
    const v2f* const twL = (const v2f*)(smem + TG::OFF_TW);
    // per-wave region (see k_fused_rev.hip for the opaque offset)
    unsigned wboff = (unsigned)(TG::off_wave(TWL) + TG::per_wave(K) * wid);
    asm volatile("" : "+s"(wboff));
    unsigned char* wb = smem + wboff;
    double* const Ltot = (double*)wb;
    int* const Lorow = (int*)(Ltot + GFT);                          // output row (the launcher checks that rows fit 32 bits)
    int* const Lcnt = Lorow + GFT;
    u16* const Lci = (u16*)(Lcnt + GFT);
    int* const Lsel = (int*)(Lci + TG::CAPW);
    int* const Lsbin = Lsel + kpad;
    float* const Lsval = (float*)(Lsbin + gs * kpad);
    const int trash0 = (int)(((const unsigned char*)(Lkeys + wid * 64) - (const unsigned char*)Lci) / 2);    // in u16 slots from Lci
    unsigned xoff = (unsigned)(TG::OFF_X + (size_t)REG * 8 * wid);
    asm volatile("" : "+s"(xoff));
    float2* const cur = (float2*)(smem + xoff);                     // this wave's region: exchange matrix, then E_s

    constexpr int NMASK = TG::N - 1;
    // ---- block-shared tables
    {
        const v2f* tab = (const v2f*)p.twiddle;                     // W_nfft^j, j < nfft; then the team table
        for (int i = lt; i < R * 64; i += T) {
            const int q = i >> 6, l = i & 63;
            t1L[i] = tab[(2 * S * l * q) & NMASK];                  // W_L^(l q)
        }
        for (int i = lt; i < 64; i += T) t2L[i] = tab[((TG::N / 64) * (i % P) * (i / P)) & NMASK];   // [t2][l1]
        if constexpr (TWL) {
            v2f* tw = (v2f*)(smem + TG::OFF_TW);
            for (int i = lt; i < TG::TWN; i += T) tw[i] = tab[TG::N + i];
        }
    }
    __syncthreads();
    const v2f* const twg = TWL ? twL : (const v2f*)p.twiddle + TG::N;      // [NPS][S][T]: W_N^k1, W_M^(s k1)

    // ---- lane constants of the 1024-point transform (k_fused_rev.hip)
    const int Q = lane / P, L1 = lane % P;
    float csg[G::LOGP];
    v2f cw[G::LOGP];
    {
        const float2* tab = (const float2*)p.twiddle;
#pragma unroll
        for (int s = 0; s < G::LOGP; s++) {
            const int h = P >> (s + 1);
            const bool up = (L1 & h) != 0;
            csg[s] = up ? -1.f : 1.f;
            const float2 wvv = tab[((TG::N / (2 * h)) * (L1 % h)) & NMASK];
            cw[s] = up ? pvxc::mk(wvv.x, wvv.y) : pvxc::mk(1.f, 0.f);
        }
    }
    int t1v = 0;                                                    // bitrev(l1)
#pragma unroll
    for (int b = 0; b < G::LOGP; b++) if (L1 & (1 << b)) t1v |= 1 << (G::LOGP - 1 - b);
    // the wave's share of the window: pairs j = S (lane + 64 r) + wid
    v2f wv[R];
#pragma unroll
    for (int r = 0; r < R; r++) wv[r] = ((const v2f*)p.win)[S * (lane + 64 * r) + wid];
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
        const InT* q = src + 2 * S * lane + 2 * wid + 128 * S * r;
        if constexpr (AL2 && sizeof(InT) == 4) raw[r] = *(const v2f*)q;
        else raw[r] = pvxc::mk(ld1(q), ld1(q + 1));
    };
    auto prefetch_part = [&](const InT* src, int part) {
        if (src == nullptr) return;
        constexpr int PR = R / 4;
#pragma unroll
        for (int r = part * PR; r < (part + 1) * PR; r++) load_pair(src, r);
    };

    const v2f khalf = pvxc::splat(0.5f), kmih = pvxc::mk(0.5f, -0.5f), kmh = pvxc::splat(-0.5f);
    constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f, HQ = 0.70710678118654752440f;
    // (Za, conj-partner Zb, twiddle W_N^k) -> X[k], X[M-k]   (k_fused.hip's untangle)
    auto untangle_o = [&](v2f Sm, v2f O, v2f w, v2f& x0, v2f& x1) {
        const v2f Pk = pvxc::cmul(O, w);
        x0 = __builtin_elementwise_fma(khalf, Sm, Pk);
        x1 = pvxc::fms_conj(khalf, Sm, Pk);
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
        dft_regs<R>(z);                                             // stage 1
        __builtin_amdgcn_sched_barrier(0);
        prefetch_part(nsrc, 1);
        v2f* dz = (v2f*)cur;
#pragma unroll
        for (int q2 = 0; q2 < R; q2 += 2) {
            const v2f ta = t1L[q2 * 64 + lane], tb = t1L[(q2 + 1) * 64 + lane];
            const v2f pa = (q2 > 0) ? pvxc::cmul(z[q2], ta) : z[q2], pb2 = pvxc::cmul(z[q2 + 1], tb);
            dz[q2 * PITCH + lane] = pa;
            dz[(q2 + 1) * PITCH + lane] = pb2;
        }
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
            xstep4<2, true>(a, csg[G::LOGP - 2], cw[G::LOGP - 2]);
            xstep4<1, false>(a, csg[G::LOGP - 1], cw[G::LOGP - 1]);
#pragma unroll
            for (int j = 0; j < 4; j++) dz[zpad<R>(Q + R * (t0 + j) + G::R2 * t1v)] = a[j];
        }
        // the lane's join / untangle twiddles: fetched before the barrier (from L2 / LDS), their latency under its wait
        __builtin_amdgcn_sched_barrier(0);                          // (not above the transform's last stage: no registers there)
        v2f twv[NPS][S];
#pragma unroll
        for (int j = 0; j < NPS; j++)
#pragma unroll
            for (int s = 0; s < S; s++) twv[j][s] = twg[(j * S + s) * T + lt];
        team_sync();                                                // ---- B1: every E_s is in place
        // ---- join + untangle, in place
        v2f* const xz = (v2f*)X;
        float lmax = -INFINITY, lmin = INFINITY, ls0 = 0.f, ls1 = 0.f;
        // the k1 = L/2 family (bins L/2 + L u): its S values pair among themselves; every lane computes it (a few
        // instructions on wave-uniform constants), lane 0 of the team stores it in the slots its k1 = 0 set leaves free
        v2f spv[S];
        {
            v2f c[S];
#pragma unroll
            for (int s = 0; s < S; s++) c[s] = xz[s * REG + zpad<R>(L / 2)];
            if constexpr (S == 2) {
                const v2f z0 = pvxc::add_mni(c[0], c[1]), z1 = pvxc::add_pi(c[0], c[1]);      // Z[L/2], Z[3L/2]
                const v2f Sm = pvxc::add_conj(z0, z1), D = pvxc::sub_conj(z0, z1);
                const v2f O = pvxc::mul_swap(D, kmih);
                const v2f Pk = pvxc::cmul_k(O, pvxc::mk(HQ, -HQ));                            // W_N^(L/2) = W_8
                spv[0] = __builtin_elementwise_fma(khalf, Sm, Pk);
                spv[1] = pvxc::fms_conj(khalf, Sm, Pk);
            } else {
                const v2f c1 = pvxc::cmul_k(c[1], pvxc::mk(HQ, -HQ)), c2 = pvxc::mni(c[2]), c3 = pvxc::cmul_k(c[3], pvxc::mk(-HQ, -HQ));
                const v2f A = c[0] + c2, B = c[0] - c2, C = c1 + c3, D = c1 - c3;
                const v2f z0 = A + C, z2 = A - C, z1 = pvxc::add_mni(B, D), z3 = pvxc::add_pi(B, D);   // Z[L/2 + L u]
                {   // (u = 0, 3): W_N^(L/2) = W_16
                    const v2f Sm = pvxc::add_conj(z0, z3), Dd = pvxc::sub_conj(z0, z3);
                    const v2f Pk = pvxc::cmul_k(pvxc::mul_swap(Dd, kmih), pvxc::mk(C1, -S1));
                    spv[0] = __builtin_elementwise_fma(khalf, Sm, Pk);
                    spv[3] = pvxc::fms_conj(khalf, Sm, Pk);
                }
                {   // (u = 1, 2): W_N^(L/2 + L) = W_16^3
                    const v2f Sm = pvxc::add_conj(z1, z2), Dd = pvxc::sub_conj(z1, z2);
                    const v2f Pk = pvxc::cmul_k(pvxc::mul_swap(Dd, kmih), pvxc::mk(S1, -C1));
                    spv[1] = __builtin_elementwise_fma(khalf, Sm, Pk);
                    spv[2] = pvxc::fms_conj(khalf, Sm, Pk);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NPS; j++) {
            const int k1 = lt + T * j;
            const int kb = (L - k1) & (L - 1);
            const int sa = zpad<R>(k1);
            int sb = zpad<R>(kb);
            v2f a[S], b[S];
#pragma unroll
            for (int s = 0; s < S; s++) { a[s] = xz[s * REG + sa]; b[s] = xz[s * REG + sb]; }
            const v2f wu = twv[j][0];                               // W_N^k1
#pragma unroll
            for (int s = 1; s < S; s++) {
                const v2f wj = twv[j][s];                           // W_M^(s k1)
                a[s] = pvxc::cmul(a[s], wj);
                b[s] = pvxc::cmul_conj(b[s], wj);
            }
            v2f x0[S], x1[S];
            if constexpr (S == 2) {
                const v2f A0 = a[0] + a[1], A1 = a[0] - a[1], B0 = b[0] + b[1], B1 = b[0] - b[1];
                // t = 0: (A0, B0), W_N^k1;   t = 1: (A1, B1), W_N^(k1 + L) = -i W_N^k1
                untangle_o(pvxc::add_conj(A0, B0), pvxc::mul_swap(pvxc::sub_conj(A0, B0), kmih), wu, x0[0], x1[0]);
                untangle_o(pvxc::add_conj(A1, B1), pvxc::sub_conj(A1, B1) * kmh, wu, x0[1], x1[1]);
            } else {
                v2f A[4], B[4];
                {
                    const v2f e = a[0] + a[2], f = a[0] - a[2], g = a[1] + a[3], h = a[1] - a[3];
                    A[0] = e + g; A[2] = e - g; A[1] = pvxc::add_mni(f, h); A[3] = pvxc::add_pi(f, h);
                }
                {
                    const v2f e = b[0] + b[2], f = b[0] - b[2], g = b[1] + b[3], h = b[1] - b[3];
                    B[0] = e + g; B[2] = e - g; B[1] = pvxc::add_mni(f, h); B[3] = pvxc::add_pi(f, h);
                }
                // t: (A_t, B_((4 - t) mod 4)), W_N^(k1 + L t) = W_8^t W_N^k1
                untangle_o(pvxc::add_conj(A[0], B[0]), pvxc::mul_swap(pvxc::sub_conj(A[0], B[0]), kmih), wu, x0[0], x1[0]);
                untangle_o(pvxc::add_conj(A[1], B[3]), pvxc::cmul_k(pvxc::mul_swap(pvxc::sub_conj(A[1], B[3]), kmih), pvxc::mk(HQ, -HQ)), wu, x0[1], x1[1]);
                untangle_o(pvxc::add_conj(A[2], B[2]), pvxc::sub_conj(A[2], B[2]) * kmh, wu, x0[2], x1[2]);
                untangle_o(pvxc::add_conj(A[3], B[1]), pvxc::cmul_k(pvxc::mul_swap(pvxc::sub_conj(A[3], B[1]), kmih), pvxc::mk(-HQ, -HQ)), wu, x0[3], x1[3]);
            }
            int kbb = kb;                                           // bins of the mirrored slots: kbb + L (S-1-t)
            if (j == 0) {
                if (lt == 0) {
                    // k1 = 0: x1[t] would be X[M - L t] (bin M, or a duplicate of x0[S-t]); the slots take bins L/2 + L u
#pragma unroll
                    for (int t = 0; t < S; t++) x1[t] = spv[S - 1 - t];
                    kbb = L / 2; sb = zpad<R>(L / 2);
                }
            }
#pragma unroll
            for (int t = 0; t < S; t++) {
                const float e0 = __builtin_fmaf(x0[t].x, x0[t].x, x0[t].y * x0[t].y), e1 = __builtin_fmaf(x1[t].x, x1[t].x, x1[t].y * x1[t].y);
                xz[t * REG + sa] = x0[t];                           // X[k1 + L t]
                xz[(S - 1 - t) * REG + sb] = x1[t];                 // X[kbb + L (S-1-t)]
                Ly[ymap<1>(k1 + L * t)] = e0;
                Ly[ymap<1>(kbb + L * (S - 1 - t))] = e1;
                lmax = fmaxf(lmax, fmaxf(e0, e1)); lmin = fminf(lmin, fminf(e0, e1)); ls0 += e0; ls1 += e1;
            }
        }
        {
            const float wm = wave_max(lmax), wn = wave_min(lmin);
            const double wsum = wave_sum((double)ls0 + (double)ls1);
            if (lane == 0) { Lpmax[wid] = wm; Lpmin[wid] = wn; Lpsum[wid] = wsum; }
        }
        team_sync();                                                // ---- B2: X, |X|^2 and the partial reductions
        float mx = Lpmax[0], mn = Lpmin[0];
        double sm = Lpsum[0];
#pragma unroll
        for (int w = 1; w < S; w++) { mx = fmaxf(mx, Lpmax[w]); mn = fminf(mn, Lpmin[w]); sm += Lpsum[w]; }
        maxe = mx; mine = mn; tot = sm;
    };

    // per-peak pass over the staged frames [0, ng): every wave its own peaks; K <= 64: one entry per lane
    int LPF = 1;
    while (LPF < K && LPF < 64) LPF <<= 1;
    const int gl = lane / LPF, e0 = lane - gl * LPF;
    const unsigned long long gmask = (LPF == 64 ? ~0ull : ((1ull << LPF) - 1ull)) << (gl * LPF);
    auto flush = [&](int ng) {
        wave_sync();
        kargs_t q = kargs;
        asm volatile("" : "+s"(q));                                  // loads through q stay here
        PeakConst pc;
        pc.fstep = q->fstep; pc.dt = q->dt; pc.nfft = TG::N; pc.hop = q->hop; pc.wfbin = q->wfbin;
        const int g = gl;
        const bool gvalid = g < ng;
        const int cnt = gvalid ? Lcnt[g] : -1;
        const int64_t orow = gvalid ? (int64_t)Lorow[g] : 0;
        bool valid = (cnt >= 0) && (e0 < cnt);
        int nbin = 0;
        PeakOut o;
        o.freq = 0.0; o.dfb = 0.0; o.thisph = 0.0; o.mag = 0.0; o.valid = false;
        if (valid) {
            nbin = Lsbin[g * kpad + e0];
            const float* sv = Lsval + (size_t)(g * kpad + e0) * 5;
            o = peak_math<float>(nbin, sv[0], sv[1], sv[2], sv[3], sv[4], pc);
            valid = o.valid;
        }
        const unsigned long long bal = __ballot(valid) & gmask;
        if (gvalid && e0 == 0) Lval[wid * GFT + g] = __popcll(bal);
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
        if (valid) {
            const int oi = off + __popcll(bal & ((1ull << lane) - 1ull));
            ob[oi] = (double)nbin;
            of[oi] = o.freq;
            om[oi] = o.mag;
            op[oi] = o.thisph;
            orp[oi] = o.thisph + kPi * o.dfb / pc.fstep;              // PV.py:207
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
    int ng = 0;
    bool pend = false, pend_prev0 = false;                          // the frame staged last still waits for its previous spectrum
    int pend_nk = 0, own_sl = -1, own_pb = 1;
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
        spectrum(zero_row, row_src(g - 1, bn, qn), maxe, mine, tot);
        if (pend) {
            // ---- the frame above (staged as group ng - 1) takes its previous spectrum from this row
            if (!pend_prev0) {
                if (own_sl >= 0) {
                    const float2 pv = X[xa<S>(own_pb)];
                    Lsval[(size_t)own_sl * 5 + 2] = pv.x;
                    Lsval[(size_t)own_sl * 5 + 3] = pv.y;
                }
            } else
            for (int e = lane; e < pend_nk; e += 64) {
                const int sl = (ng - 1) * kpad + e;
                const int nbin = Lsbin[sl];
                Lsval[(size_t)sl * 5 + 2] = (float)p.prev0[2 * nbin];
                Lsval[(size_t)sl * 5 + 3] = (float)p.prev0[2 * nbin + 1];
            }
            pend = false;
            if (ng == gs) { flush(ng); ng = 0; }                    // team-uniform
        }
        const bool real = !zero_row && g >= r0;                     // team-uniform
        bool has = false, slow = false;
        int pb = 1, bad = 0, n_w = 0;
        unsigned mykey = 0u;
        float2 c = make_float2(0.f, 0.f), vm = c, vp = c;
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
            if (C_w > 64) {
                // a dense segment: its npks best first (the row's npks best are among the segments' npks best, same tie rule)
                n_w = peak_pick_regs<R / 2, 1, u16, true>(Ly, Lci, Lsel, M, K, C_w, th, mine, lane);
                has = lane < n_w;
                pb = has ? Lsel[lane] : 1;
            } else {
                has = lane < n_w;
                pb = has ? (int)Lci[lane] : 1;
            }
            // ---- lane c owns candidate c: one LDS round trip for its score, the 2*rad neighbours of the salience test
            // (PF.py:126-134), the spectrum around it (k_fused_rev.hip)
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
            c = X[xa<S>(pb)];
            vm = X[xa<S>(pb - 1)]; vp = X[xa<S>(pb + 1)];
#pragma unroll
            for (int d = 0; d < 10; d++) bad |= (int)(nb[d] > v);
            mykey = has ? __float_as_uint(v - mine) : 0u;           // scores >= 0: bits order like values
            Lkeys[wid * 64 + lane] = mykey;
            if (lane == 0) Lnw[wid] = n_w;
        }
        if (p.spec_out != nullptr && g == p.spec_row) {
#pragma unroll
            for (int j = 0; j < M / T; j++) {
                const float2 v = X[xa<S>(lt + T * j)];
                p.spec_out[2 * (lt + T * j)] = v.x;
                p.spec_out[2 * (lt + T * j) + 1] = v.y;
            }
        }
        team_sync();                                                // ---- B3: keys exchanged; X and |X|^2 are free for the row below
        if (real) {
            int ctot = 0;
#pragma unroll
            for (int w = 0; w < S; w++) ctot += Lnw[w];
            slow = (th < 0.0) && (ctot < K);                        // zeros of pkmskamp qualify too (PF.py:166-187): wave 0, below
            bool take = has;
            if (ctot > K) {
                // rank of this lane's candidate among all the team's: larger score, or equal score and lower bin
                int rank = 0;
#pragma unroll
                for (int w = 0; w < S; w++) {
                    const int n = Lnw[w];
                    if (w == wid) {
                        for (int j = 0; j < n; ++j) {
                            const unsigned kj = (unsigned)__builtin_amdgcn_readlane((int)mykey, j);
                            rank += (kj > mykey || (kj == mykey && j < lane)) ? 1 : 0;
                        }
                    } else {
                        const unsigned ko = Lkeys[w * 64 + lane];
                        for (int j = 0; j < n; ++j) {
                            const unsigned kj = (unsigned)__builtin_amdgcn_readlane((int)ko, j);
                            rank += (kj > mykey || (kj == mykey && w < wid)) ? 1 : 0;
                        }
                    }
                }
                take = has && (rank < K);
            }
            int nk = 0;
            own_sl = -1;
            const int64_t orow = (int64_t)gb * Fi + (gq - 1);
            if (!slow) {
                const bool keep = take && (p.rad < 0 || bad == 0);
                const unsigned long long bal = __ballot(keep);
                if (keep) {
                    const int sl = ng * kpad + lane_prefix(bal);
                    // PV.py:197-199: 3-bin energy, bin 0 excluded (1 <= pb <= M-2)
                    const float em = (pb > 1) ? __builtin_fmaf(vm.x, vm.x, vm.y * vm.y) : 0.f;
                    const float s3 = (em + __builtin_fmaf(c.x, c.x, c.y * c.y)) + __builtin_fmaf(vp.x, vp.x, vp.y * vp.y);
                    Lsbin[sl] = pb;
                    float* sv = Lsval + (size_t)sl * 5;
                    sv[0] = c.x; sv[1] = c.y; sv[4] = s3;           // sv[2], sv[3]: the previous spectrum, one row later
                    own_sl = sl; own_pb = pb;
                }
                nk = __popcll(bal);
            } else {
                // fewer maxima than npks under a negative threshold: all maxima, then the first non-maximum interior
                // bins (peak_pick_regs' first branch, on the whole row) -- wave 0 alone, then one more barrier
                if (wid == 0) {
                    const int nsel = peak_pick_regs<R / 2, 1, u16, true>(Ly, Lci, Lsel, M, K, ctot, th, mine, lane);
                    const int e = lane;
                    int sb2 = 1;
                    if (e < nsel) sb2 = Lsel[e];
                    const bool keep = (p.rad <= 8) ? salient_groups<1>(Ly, M, Lsel, 0, nsel, p.rad, lane)
                                                   : ((e < nsel) && salient<float, 1>(Ly, M, sb2, p.rad));
                    const unsigned long long bal = __ballot(keep);
                    if (keep) {
                        const int sl = ng * kpad + lane_prefix(bal);
                        const float2 cc = X[xa<S>(sb2)];
                        const float2 cm = X[xa<S>(sb2 - 1)], cp = X[xa<S>(sb2 + 1)];
                        const float em = (sb2 > 1) ? __builtin_fmaf(cm.x, cm.x, cm.y * cm.y) : 0.f;
                        const float s3 = (em + __builtin_fmaf(cc.x, cc.x, cc.y * cc.y)) + __builtin_fmaf(cp.x, cp.x, cp.y * cp.y);
                        Lsbin[sl] = sb2;
                        float* sv = Lsval + (size_t)sl * 5;
                        sv[0] = cc.x; sv[1] = cc.y; sv[4] = s3;
                        own_sl = sl; own_pb = sb2;
                    }
                    nk = __popcll(bal);
                }
                team_sync();
            }
            if (lane == 0) { Lcnt[ng] = nk; Lorow[ng] = (int)orow; Ltot[ng] = tot; }
            ng++;
            pend = true; pend_nk = nk; pend_prev0 = (p.prev0 != nullptr) && (orow == 0);
        }
        gb = bn; gq = qn;
    }
    if (ng > 0) flush(ng);
}

template <int S, bool TWL> int launch_team(const FusedParams& p, int x_dtype, hipStream_t s) {
    using TG = TeamGeo<S>;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    if (p.total_rows >= 0x7fffff00LL) { pvx_set_error("the fused kernel indexes rows in 32 bits (%lld rows)", (long long)p.total_rows); return PVX_ERR_UNSUPPORTED; }
    if (p.K > 64 || p.rad > 5) { pvx_set_error("the team kernel takes npks <= 64 and rad <= 5 (npks=%d rad=%d)", p.K, p.rad); return PVX_ERR_UNSUPPORTED; }
    const size_t lds = TG::total(p.K, TWL);
    if (lds > 160 * 1024) { pvx_set_error("nfft=%d npks=%d needs %zu bytes of LDS in the team kernel", TG::N, p.K, lds); return PVX_ERR_UNSUPPORTED; }
    constexpr int R = 16;
    const bool al2 = (x_dtype == PVX_F32) && (p.hop % 2 == 0) && (p.sig_stride % 2 == 0) && (((uintptr_t)p.x) % 8 == 0);
    const int H = (p.hop == 32 * R * S) ? R / 4 : (p.hop == 64 * R * S) ? R / 2 : 0;
    const void* fn = nullptr;
#define PVX_TEAM_PICK(INT, AL) (H == R / 4 ? (const void*)k_fused_team<S, INT, AL, R / 4, TWL> : H ? (const void*)k_fused_team<S, INT, AL, R / 2, TWL> : (const void*)k_fused_team<S, INT, AL, 0, TWL>)
    switch (x_dtype) {
        case PVX_F32: fn = al2 ? PVX_TEAM_PICK(float, true) : PVX_TEAM_PICK(float, false); break;
        case PVX_F64: fn = PVX_TEAM_PICK(double, false); break;
        case PVX_I16: fn = PVX_TEAM_PICK(int16_t, false); break;
        default: pvx_set_error("bad x_dtype %d", x_dtype); return PVX_ERR_INVALID;
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

// lane-ordered twiddles of the join / untangle pass, appended to the plan's W_nfft^j table: entry (j S + s) T + lt is
// W_N^k1 (s = 0) or W_M^(s k1) = W_N^(2 s k1), k1 = lt + T j
int pvx_fused_team_table_len(int nfft) { return (nfft == 4096 || nfft == 8192) ? 512 * (nfft / 4096) * 2 : 0; }
void pvx_fused_team_table(int nfft, const float* tw /* [nfft][2] */, float* out) {
    const int S = nfft / 2048, T = 64 * S, NPS = 512 / T;
    for (int j = 0; j < NPS; j++)
        for (int s = 0; s < S; s++)
            for (int lt = 0; lt < T; lt++) {
                const int k1 = lt + T * j;
                const int idx = (s == 0 ? k1 : 2 * s * k1) & (nfft - 1);
                out[2 * ((j * S + s) * T + lt)] = tw[2 * idx];
                out[2 * ((j * S + s) * T + lt) + 1] = tw[2 * idx + 1];
            }
}

int pvx_fused_team_supported(int nfft, int precision, int K) {
    if (precision != 32 || K > 64) return 0;
    switch (nfft) {
        case 4096: return TeamGeo<2>::total(K, false) <= 160 * 1024;
        case 8192: return TeamGeo<4>::total(K, false) <= 160 * 1024;
        default: return 0;
    }
}

int pvx_launch_fused_team(const FusedParams& p, int nfft, int x_dtype, hipStream_t s) {
    if (p.total_rows <= 0) return PVX_OK;
    switch (nfft) {
        case 4096: return getenv("PVX_TEAM_TWL") ? launch_team<2, true>(p, x_dtype, s) : launch_team<2, false>(p, x_dtype, s);
        case 8192: return launch_team<4, false>(p, x_dtype, s);
        default: pvx_set_error("the team kernel does not handle nfft=%d", nfft); return PVX_ERR_UNSUPPORTED;
    }
}
