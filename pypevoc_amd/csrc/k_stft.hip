// k_stft.hip -- windowed real FFT of every frame in ONE kernel, at the reference's own precision (float64):
//
//   PV.calc_fft_frame   pypevoc/PVAnalysis.py:150-158   fft(x[pos:pos+nfft] * win) / wfact, np.fft on float64
//
// for nfft = 128 R, R in {4, 8, 16} (nfft 512, 1024, 2048).  It writes the half spectrum rows (bins 0 .. nfft/2-1,
// complex T) into the general path's workspace, where k_phase_peaks (k_peaks.hip) and k_harmonic_rows
// (k_harmonic.hip) read them -- replacing k_frames + rocFFT's two kernels: one launch instead of three per
// workspace chunk, hop*4 B in and nfft/2 * 16 B out per frame instead of ~5x that through HBM / Infinity Cache,
// and no windowed-frame buffer at all.
//
// Same factorisation as the float32 kernels (k_fused.hip): the real FFT of nfft samples is a complex FFT of
// M = nfft/2 = 64 R points on z[j] = xw[2j] + i xw[2j+1] plus the untangle; M = R (registers) x R (registers,
// after one LDS exchange) x P (P = 64/R lanes, DPP / swizzle steps).  One wave64 per frame, plain float64
// arithmetic (no packed instructions exist for it), explicit fma in the complex multiplies so that a frame's
// spectrum does not depend on which wave computes it.  The untangle writes straight to global memory (bins k
// and M-k: two coalesced 1 KiB stores per step), so a wave needs ONE LDS buffer; window and twiddles are LDS
// tables shared by the workgroup, which keeps the kernel under 256 registers: 6 waves per CU.
// Rows are dealt to waves round-robin (consecutive waves = consecutive frames: their 75 % input overlap is served
// by L1/L2); the samples of a wave's next row are prefetched in four groups spread over the transform.
#include <type_traits>

#include "pvx_stft.h"
#include "pvx_wave.h"

using namespace pvxw;
using namespace pvxf;
using namespace pvxs;

namespace {

template <int R, typename T, typename InT>
__global__ __launch_bounds__(384) void k_stft(StftParams p) {
    using G = StftGeo<R, T>;
    constexpr int M = G::M, P = G::P, PITCH = G::PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x >> 6;
    T* const winL = (T*)(smem + G::OFF_WIN);
    cx<T>* const t1L = (cx<T>*)(smem + G::OFF_T1);
    cx<T>* const t2L = (cx<T>*)(smem + G::OFF_T2);
    cx<T>* const tw3 = (cx<T>*)(smem + G::OFF_TW3);
    cx<T>* const dz = (cx<T>*)(smem + G::OFF_BUF) + (size_t)wid * G::BUFC;
    {
        const cx<T>* tab = (const cx<T>*)p.twiddle;
        constexpr int NMASK = G::N - 1;
        for (int i = threadIdx.x; i < G::N; i += blockDim.x) winL[i] = ((const T*)p.win)[i];
        for (int i = threadIdx.x; i < R * 64; i += blockDim.x) t1L[i] = tab[(2 * (i & 63) * (i >> 6)) & NMASK];
        for (int i = threadIdx.x; i < 64; i += blockDim.x) t2L[i] = tab[((G::N / 64) * (i % P) * (i / P)) & NMASK];   // [t2][l1]
        for (int i = threadIdx.x; i <= G::HALF; i += blockDim.x) tw3[i] = tab[i];
    }
    __syncthreads();
    const int Q = lane / P, L1 = lane % P;
    T csg[G::LOGP > 0 ? G::LOGP : 1];
    cx<T> cw[G::LOGP > 0 ? G::LOGP : 1];
    {
        const cx<T>* tab = (const cx<T>*)p.twiddle;
        constexpr int NMASK = G::N - 1;
#pragma unroll
        for (int s = 0; s < G::LOGP; s++) {
            const int h = P >> (s + 1);
            const bool up = (L1 & h) != 0;
            csg[s] = up ? (T)-1 : (T)1;
            const cx<T> wv = tab[((G::N / (2 * h)) * (L1 % h)) & NMASK];
            cw[s] = up ? wv : mkc<T>((T)1, (T)0);
        }
    }
    int t1v = 0;
#pragma unroll
    for (int b = 0; b < G::LOGP; b++) if (L1 & (1 << b)) t1v |= 1 << (G::LOGP - 1 - b);

    // workspace row j holds global row R0 - 1 + j (pvx_internal.h): rows are dealt round-robin to the waves
    const int64_t W = (int64_t)gridDim.x * nw;
    const int64_t w = (int64_t)blockIdx.x * nw + wid;
    const int64_t rows1 = p.F + 1;
    auto row_src = [&](int64_t j) -> const InT* {                     // samples of workspace row j; nullptr: zero row
        if (j >= p.ws_rows) return nullptr;
        const int64_t g = p.R0 - 1 + j;
        if (g < 0 || g >= p.total_rows) return nullptr;
        const int64_t b = g / rows1, q = g - b * rows1;
        if (q == 0) return nullptr;
        return (const InT*)p.x + b * p.sig_stride + (q - 1) * (int64_t)p.hop;
    };
    T raw[2 * R];                                                     // samples (2l + 128 r, + 1) of the next row
    auto prefetch_part = [&](const InT* src, int part) {
        if (src == nullptr) return;
        constexpr int PR = R / 4;
#pragma unroll
        for (int r = part * PR; r < (part + 1) * PR; r++) {
            const InT* q = src + 2 * lane + 128 * r;
            raw[2 * r] = (T)q[0]; raw[2 * r + 1] = (T)q[1];
        }
    };
    {
        const InT* s0 = row_src(w);
        prefetch_part(s0, 0); prefetch_part(s0, 1); prefetch_part(s0, 2); prefetch_part(s0, 3);
    }
    for (int64_t j = w; j < p.ws_rows; j += W) {
        cx<T>* out = (cx<T>*)p.spec + (size_t)j * p.ldo;
        const bool zero_row = row_src(j) == nullptr;
        const InT* nsrc = row_src(j + W);
        cx<T> z[R];
        // the lane's window pairs, as many in flight as the registers hold (pvx_stft.h: lds_gather_use)
        lds_gather_use<0, R, lds_batch<R, T>(), T>((const cx<T>*)(winL + 2 * lane), 64, [&](int r, cx<T> w) {
            z[r] = mkc<T>(raw[2 * r] * w.x, raw[2 * r + 1] * w.y);
            asm volatile("" : "+v"(z[r].x), "+v"(z[r].y));           // the multiplies stay above the next loads
        });
        __builtin_amdgcn_sched_barrier(0);
        prefetch_part(nsrc, 0);
        if (zero_row) {
            prefetch_part(nsrc, 1); prefetch_part(nsrc, 2); prefetch_part(nsrc, 3);
            for (int k = lane; k < M; k += 64) out[k] = mkc<T>((T)0, (T)0);
            continue;
        }
        dftT<R, T>(z);                                                // stage 1
        __builtin_amdgcn_sched_barrier(0);
        prefetch_part(nsrc, 1);
        dz[lane] = z[0];
        lds_gather_use<1, R, lds_batch<R, T>(), T>(t1L + lane, 64, [&](int q2, cx<T> w) { dz[q2 * PITCH + lane] = cmulT(z[q2], w); });
        wave_sync();
#pragma unroll
        for (int l2 = 0; l2 < R; l2++) z[l2] = dz[Q * PITCH + L1 + P * l2];
        prefetch_part(nsrc, 2);
        wave_sync();
        dftT<R, T>(z);                                                // stage 2
        __builtin_amdgcn_sched_barrier(0);
        prefetch_part(nsrc, 3);
        cx<T> tq2[R];
        if constexpr (lds_batch<R, T>() == R) lds_gather<1, R, T>(tq2, t2L + L1, P);
#pragma unroll
        for (int t = 0; t < R; t++) {
            // twiddle W_64^(l1 t2), then stage 3: P-point DFT across the P lanes of a group (decimation in frequency)
            cx<T> a = (t > 0) ? cmulT(z[t], lds_batch<R, T>() == R ? tq2[t] : t2L[t * P + L1]) : z[t];
            if constexpr (G::LOGP >= 1) {
                if constexpr (P >= 16) a = xstepT<8, true, T>(a, csg[G::LOGP - 4], cw[G::LOGP - 4]);
                if constexpr (P >= 8) a = xstepT<4, true, T>(a, csg[G::LOGP - 3], cw[G::LOGP - 3]);
                if constexpr (P >= 4) a = xstepT<2, true, T>(a, csg[G::LOGP - 2], cw[G::LOGP - 2]);
                a = xstepT<1, false, T>(a, csg[G::LOGP - 1], cw[G::LOGP - 1]);
            }
            dz[zpadT<R, T>(Q + R * t + G::R2 * t1v)] = a;
        }
        wave_sync();
        // ---- untangle straight to global memory: pairs (k, M-k), k = lane + 64 j2
        //   S = Za + conj Zb, D = Za - conj Zb;  E = S/2, O = -i D/2, Pk = W^k O;  X[k] = E + Pk, X[M-k] = conj(E - Pk)
        constexpr int NPAIR = R / 2;
        cx<T> za[NPAIR], zb[NPAIR];
#pragma unroll
        for (int j2 = 0; j2 < NPAIR; j2++) {
            const int k = lane + 64 * j2;
            za[j2] = dz[zpadT<R, T>(k)];
            zb[j2] = dz[zpadT<R, T>((M - k) & (M - 1))];
        }
        const cx<T> zc = dz[zpadT<R, T>(G::HALF)];
#pragma unroll
        for (int j2 = 0; j2 < NPAIR; j2++) {
            const int k = lane + 64 * j2;
            const int km = (M - k) & (M - 1);
            const cx<T> S = mkc<T>(za[j2].x + zb[j2].x, za[j2].y - zb[j2].y);
            const cx<T> D = mkc<T>(za[j2].x - zb[j2].x, za[j2].y + zb[j2].y);
            const cx<T> O = mkc<T>((T)0.5 * D.y, (T)-0.5 * D.x);
            const cx<T> Pk = cmulT(O, tw3[k]);
            const cx<T> x0 = mkc<T>(fmaT((T)0.5, S.x, Pk.x), fmaT((T)0.5, S.y, Pk.y));
            cx<T> x1 = mkc<T>(fmaT((T)0.5, S.x, -Pk.x), -fmaT((T)0.5, S.y, -Pk.y));
            int kk = km;
            if (j2 == 0 && lane == 0) { x1 = mkc<T>(zc.x, -zc.y); kk = G::HALF; }      // bin 0 pairs with itself; its slot takes bin M/2
            row_store(&out[k], x0);
            row_store(&out[kk], x1);
        }
        wave_sync();                                                  // dz is free again
    }
}

// ---- nfft = 128 R S: the M = 64 R S point transform as S interleaved M1 = 64 R point ones (decimation in time) ------
// nfft 4096 (R = 16, S = 2) does not fit one wave's registers as a single R x R x P factorisation (R = 32 is 64
// complex values per lane before any temporary).  A TEAM of S waves takes a frame: wave s runs sub-sequence z[S j + s],
// j < M1, through exactly the stages of k_stft above and leaves its natural-order result in its own LDS region; after a
// barrier one in-place radix-S pass with the twiddles W_M^(s k1) joins them (Z[k1 + M1 q] = sum_s W_S^(s q) W_M^(s k1)
// Z_s[k1]; every wave a quarter or half of the k1), and after another each wave untangles its share of the bin pairs
// straight to global memory.  nfft 4096: a workgroup of three teams (6 waves per CU, window table in LDS); nfft 8192: one
// team (4 waves per CU).  One wave per frame with all S regions to itself held 4 / 2 waves per CU.  Window, join and untangle twiddles come
// from global memory (L2-resident tables), the sub-transform tables stay in LDS.
// WIDE: float64 samples into the float64 transform -- the next row's samples wait in 2 R register pairs instead of 2 R
// registers, so the window stays in LDS also at S = 4 (64 KB beside the 87 KB of regions and tables) and S = 2 runs two
// teams per workgroup (512 registers per wave) instead of three: no register of these kernels lives in scratch.
template <int R, int S, typename T, bool WIDE = false> struct SplitGeo {
    using G1 = StftGeo<R, T>;
    static constexpr int M1 = G1::M, M = M1 * S, N = 2 * M, LOGM1 = ilog2(M1);
    static constexpr int BUFC = G1::BUFC;                                                    // one region (complex)
    // S = 2: three teams (6 waves, 256 registers each) with the window in LDS; S = 4: one team (4 waves, up to 512
    // registers each: the window values of the next row are prefetched with its samples)
    static constexpr bool WL = (S == 2) || WIDE;
    static constexpr int TEAMS = (S == 2) ? (WIDE ? 2 : 3) : 1;
    static constexpr size_t OFF_WIN = 0;                                                     // T [N]  window / wfact (WL)
    static constexpr size_t OFF_T1 = OFF_WIN + (WL ? (size_t)N * sizeof(T) : 0);             // cx [R][64]  W_M1^(l q)
    static constexpr size_t OFF_T2 = OFF_T1 + (size_t)R * 64 * 2 * sizeof(T);                // cx [R][P]   W_64^(l1 t2)
    static constexpr size_t OFF_BUF = OFF_T2 + 64 * 2 * sizeof(T);                           // cx [teams][S][BUFC]
    __host__ __device__ static size_t total(int teams) { return OFF_BUF + (size_t)teams * S * BUFC * 2 * sizeof(T); }
};

// workgroup barrier for LDS hand-offs: LDS traffic drained, then s_barrier -- not __syncthreads(), which would also
// wait for the prefetched samples of the next row
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifndef PVX_SPLIT_SCAN_PIECE
#define PVX_SPLIT_SCAN_PIECE 512
#endif
template <int R, int S, typename T, typename InT, bool CAND>
__global__ __launch_bounds__((S == 2 && !(sizeof(T) == 8 && sizeof(InT) == 8)) ? 384 : 256) void k_stft_split(StftParams p) {
    using G = SplitGeo<R, S, T, (sizeof(T) == 8 && sizeof(InT) == 8)>;
    using G1 = StftGeo<R, T>;
    constexpr int M1 = G::M1, M = G::M, P = G1::P, PITCH = G1::PITCH;
    constexpr int NMASK = G::N - 1;
    static_assert(S == 2 || S == 4, "radix of the join");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int teams = (int)(blockDim.x >> 6) / S;                     // frames a workgroup has in flight
    const int team = wid / S, sub = wid - team * S;                   // this wave: sub-sequence `sub` of its team's frame
    cx<T>* const t1L = (cx<T>*)(smem + G::OFF_T1);
    cx<T>* const t2L = (cx<T>*)(smem + G::OFF_T2);
    cx<T>* const buf = (cx<T>*)(smem + G::OFF_BUF) + (size_t)team * S * G::BUFC;     // the team's S regions
    cx<T>* const dz = buf + (size_t)sub * G::BUFC;                                    // this wave's region
    const cx<T>* const tab = (const cx<T>*)p.twiddle;                 // W_N^j
    const T* const winG = (const T*)p.win;
    T* const winL = (T*)(smem + G::OFF_WIN);
    if constexpr (G::WL) for (int i = threadIdx.x; i < G::N; i += blockDim.x) winL[i] = winG[i];
    for (int i = threadIdx.x; i < R * 64; i += blockDim.x) t1L[i] = tab[(2 * S * (i & 63) * (i >> 6)) & NMASK];          // W_M1^(l q)
    for (int i = threadIdx.x; i < 64; i += blockDim.x) t2L[i] = tab[((G::N / 64) * (i % P) * (i / P)) & NMASK];           // [t2][l1]
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

    // rows are dealt round-robin to the teams of the grid; every wave of a workgroup makes the same number of trips (the
    // barriers below are workgroup barriers), a team without a row on its last trip just keeps step
    const int64_t TT = (int64_t)gridDim.x * teams;
    const int64_t t0 = (int64_t)blockIdx.x * teams + team;
    const int64_t rows1 = p.F + 1;
    auto row_src = [&](int64_t j) -> const InT* {                     // samples of workspace row j; nullptr: zero row
        if (j >= p.ws_rows) return nullptr;
        const int64_t g = p.R0 - 1 + j;
        if (g < 0 || g >= p.total_rows) return nullptr;
        const int64_t b = g / rows1, q = g - b * rows1;
        if (q == 0) return nullptr;
        return (const InT*)p.x + b * p.sig_stride + (q - 1) * (int64_t)p.hop;
    };
    // samples and window values of this wave's sub-sequence of the next row: element j = l + 64 r is the sample pair at
    // 2 (S j + sub) = 2 S l + 128 S r + 2 sub
    using RawT = typename std::conditional<(sizeof(T) == 8 && sizeof(InT) == 8), double, float>::type;   // (see k_stft_pv.hip)
    RawT raw[2 * R];
    T wn[G::WL ? 2 : 2 * R];
    auto prefetch_part = [&](const InT* src, int part) {
        if (src == nullptr) return;
        constexpr int PR = R / 4;
#pragma unroll
        for (int r = part * PR; r < (part + 1) * PR; r++) {
            const int o = 2 * S * lane + 128 * S * r + 2 * sub;
            raw[2 * r] = (RawT)src[o]; raw[2 * r + 1] = (RawT)src[o + 1];
            if constexpr (!G::WL) { wn[2 * r] = winG[o]; wn[2 * r + 1] = winG[o + 1]; }
        }
    };
    {
        const InT* s0 = row_src(t0);
        prefetch_part(s0, 0); prefetch_part(s0, 1); prefetch_part(s0, 2); prefetch_part(s0, 3);
    }
    const int64_t trips = (p.ws_rows - (int64_t)blockIdx.x * teams + TT - 1) / TT;      // of the workgroup's first team: the most
    for (int64_t it = 0; it < trips; ++it) {
        const int64_t j = t0 + it * TT;
        const bool have = j < p.ws_rows;
        cx<T>* out = (cx<T>*)p.spec + (size_t)(have ? j : 0) * p.ldo;
        const InT* src = have ? row_src(j) : nullptr;
        const InT* nrow = row_src(j + TT);
        const bool real = src != nullptr;
        if (real) {
            cx<T> z[R];
            if constexpr (G::WL) {
                // (window in LDS: the lane's pairs, as many in flight as the registers hold)
                lds_gather_use<0, R, lds_batch<R, T>(), T>((const cx<T>*)(winL + 2 * S * lane + 2 * sub), 64 * S, [&](int r, cx<T> w) {
                    z[r] = mkc<T>((T)raw[2 * r] * w.x, (T)raw[2 * r + 1] * w.y);
                    asm volatile("" : "+v"(z[r].x), "+v"(z[r].y));   // the multiplies stay above the next loads
                });
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    z[r] = mkc<T>((T)raw[2 * r] * wn[2 * r], (T)raw[2 * r + 1] * wn[2 * r + 1]);
                    asm volatile("" : "+v"(z[r].x), "+v"(z[r].y));   // the multiplies stay above the next loads
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            prefetch_part(nrow, 0);
            dftT<R, T>(z);                                            // stage 1
            __builtin_amdgcn_sched_barrier(0);
            prefetch_part(nrow, 1);
            dz[lane] = z[0];
            lds_gather_use<1, R, lds_batch<R, T>(), T>(t1L + lane, 64, [&](int q2, cx<T> w) { dz[q2 * PITCH + lane] = cmulT(z[q2], w); });
            wave_sync();
#pragma unroll
            for (int l2 = 0; l2 < R; l2++) z[l2] = dz[Q * PITCH + L1 + P * l2];
            prefetch_part(nrow, 2);
            wave_sync();
            dftT<R, T>(z);                                            // stage 2
            __builtin_amdgcn_sched_barrier(0);
            prefetch_part(nrow, 3);
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
        } else {
            // a zero row (or no row on this trip): nothing of it was prefetched
            prefetch_part(nrow, 0); prefetch_part(nrow, 1); prefetch_part(nrow, 2); prefetch_part(nrow, 3);
        }
        lds_barrier();                                                // the team's S sub-transforms are in their regions
        if (real) {
            // ---- join, in place: slots k1 of the S regions hold Z_s[k1] and become Z[k1 + M1 q]; this wave's share of k1
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
        lds_barrier();                                                // Z is complete
        T ey[CAND ? R : 1];                                           // (CAND) |X|^2 of this wave's bins
        if (have && !real) {
            // the zero frame in front of every signal (PV.py:121): this wave's share of a row of zeros
            for (int k = lane + 64 * sub; k < M; k += 64 * S) out[k] = mkc<T>((T)0, (T)0);
        } else if (real) {
            // ---- untangle straight to global memory: this wave's chunk of the pairs (k, M-k), k = lane + 64 j2 (see k_stft)
            auto zat = [&](int k) -> cx<T> { return buf[(size_t)(k >> G::LOGM1) * G::BUFC + zpadT<R, T>(k & (M1 - 1))]; };
            const cx<T> zc = zat(M / 2);
            constexpr int NPAIR = R / 2;
            const int ch = sub;
            // (with the candidates: in two batches of pairs -- the |X|^2 values stay in registers until every wave has
            // read its pairs, and with all of a wave's pairs in flight at once beside them the float64-input kernels spill)
            constexpr int NBATCH = CAND ? 2 : 1, PB = NPAIR / NBATCH;
#pragma unroll
            for (int bt = 0; bt < NBATCH; bt++) {
            cx<T> za[PB], zb[PB], tw[PB];
#pragma unroll
            for (int j3 = 0; j3 < PB; j3++) {
                const int j2 = bt * PB + j3;
                const int k = lane + 64 * (ch * NPAIR + j2);
                za[j3] = zat(k);
                zb[j3] = zat((M - k) & (M - 1));
                tw[j3] = tab[k];                                      // W_N^k, k < M/2
            }
#pragma unroll
            for (int j3 = 0; j3 < PB; j3++) {
                const int j2 = bt * PB + j3;
                const int k = lane + 64 * (ch * NPAIR + j2);
                const int km = (M - k) & (M - 1);
                const cx<T> Sm = mkc<T>(za[j3].x + zb[j3].x, za[j3].y - zb[j3].y);
                const cx<T> D = mkc<T>(za[j3].x - zb[j3].x, za[j3].y + zb[j3].y);
                const cx<T> O = mkc<T>((T)0.5 * D.y, (T)-0.5 * D.x);
                const cx<T> Pk = cmulT(O, tw[j3]);
                const cx<T> x0 = mkc<T>(fmaT((T)0.5, Sm.x, Pk.x), fmaT((T)0.5, Sm.y, Pk.y));
                cx<T> x1 = mkc<T>(fmaT((T)0.5, Sm.x, -Pk.x), -fmaT((T)0.5, Sm.y, -Pk.y));
                int kk = km;
                if (ch == 0 && j2 == 0 && lane == 0) { x1 = mkc<T>(zc.x, -zc.y); kk = M / 2; }      // bin 0 pairs with itself; its slot takes bin M/2
                row_store(&out[k], x0);
                row_store(&out[kk], x1);
                if constexpr (CAND) {
                    // |X|^2 of both bins, k_phase_peaks' formula (k_peaks.hip): plain products and one sum
                    ey[2 * j2] = x0.x * x0.x + x0.y * x0.y;
                    ey[2 * j2 + 1] = x1.x * x1.x + x1.y * x1.y;
                }
            }
            }
        }
        if constexpr (CAND) {
            // ---- the row's candidate peaks, while its magnitudes are on chip (the peak kernel then reads a few hundred
            // bytes of this row instead of streaming its nfft/2 complex bins back from HBM): |X|^2 -> LDS where Z was,
            // extremes and energy over the team, every wave scans its M/S bins (PF.py:166-174 with the threshold of
            // PF.py:60, 69-70 exactly as k_phase_peaks forms it), the lists go out in ascending bin order
            constexpr int SEG = M / S, CAPW = SEG / 2 + 4, NPAIR = R / 2;
            lds_barrier();                                            // every wave has read its pairs: the regions are free
            T* const yL = (T*)buf;
            unsigned short* const ciL = (unsigned short*)(yL + M) + (size_t)sub * CAPW;
            double* const partL = (double*)(((uintptr_t)((unsigned short*)(yL + M) + (size_t)S * CAPW) + 15) & ~(uintptr_t)15);    // [S][4]
            if (real) {
                T lmax = (T)-INFINITY, lmin = (T)INFINITY;
                double lsum = 0.0;
#pragma unroll
                for (int j2 = 0; j2 < NPAIR; j2++) {
                    const int k = lane + 64 * (sub * NPAIR + j2);
                    int kk = (M - k) & (M - 1);
                    if (sub == 0 && j2 == 0 && lane == 0) kk = M / 2;
                    const T e0 = ey[2 * j2], e1 = ey[2 * j2 + 1];
                    yL[k] = e0;
                    yL[kk] = e1;
                    lmax = e0 > lmax ? e0 : lmax; lmax = e1 > lmax ? e1 : lmax;
                    lmin = e0 < lmin ? e0 : lmin; lmin = e1 < lmin ? e1 : lmin;
                    lsum += (double)e0 + (double)e1;
                }
                const double wmx = (double)wave_max(lmax), wmn = (double)wave_min(lmin), wsm = wave_sum(lsum);
                if (lane == 0) { partL[sub * 4] = wmx; partL[sub * 4 + 1] = wmn; partL[sub * 4 + 2] = wsm; }
            }
            lds_barrier();                                            // the row of |X|^2 and the waves' partial extremes
            int C_w = 0;
            double mx = 0.0, mn = 0.0, sm = 0.0;
            if (real) {
                mx = partL[0]; mn = partL[1]; sm = partL[2];
#pragma unroll
                for (int w2 = 1; w2 < S; w2++) {
                    mx = partL[w2 * 4] > mx ? partL[w2 * 4] : mx;
                    mn = partL[w2 * 4 + 1] < mn ? partL[w2 * 4 + 1] : mn;
                    sm += partL[w2 * 4 + 2];
                }
                double minamp;
                if constexpr (sizeof(T) == 4) minamp = (double)sqrtf((float)mx) * p.cand_thr;       // PF.py:60 (k_phase_peaks)
                else minamp = sqrt(mx) * p.cand_thr;
                const double th = (minamp != 0.0) ? minamp * minamp - mn : 0.0;
                // (S = 4: up to 512 registers per wave, all of the scan's LDS reads in flight at once; S = 2 has 256: two pieces of
                // 512 bins, each with its reads in flight together -- the loop form is a dependent LDS round trip per 64 bins)
                if constexpr (S == 4) C_w = pvxw::peak_scan<T, SEG / 64, false>((const T*)yL, sub * SEG, SEG, M, (T)mn, th, (T*)nullptr, ciL, lane);
                else {
                    constexpr int PIECE = PVX_SPLIT_SCAN_PIECE;
#pragma unroll 1
                    for (int kb = 0; kb < SEG; kb += PIECE) C_w += pvxw::peak_scan<T, PIECE / 64, false>((const T*)yL, sub * SEG + kb, PIECE, M, (T)mn, th, (T*)nullptr, ciL + C_w, lane);
                }
                if (lane == 0) partL[sub * 4 + 3] = (double)C_w;
            }
            lds_barrier();                                            // the waves' candidate counts (and this wave's list)
            if (real) {
                int off = 0, total = 0;
#pragma unroll
                for (int w2 = 0; w2 < S; w2++) { const int c = (int)partL[w2 * 4 + 3]; off += (w2 < sub) ? c : 0; total += c; }
                const size_t rb = (size_t)j * p.cand_cap;
                T* const gy = (T*)p.cand_y + rb;
                unsigned short* const gb = p.cand_bin + rb;
                for (int i = lane; i < C_w; i += 64) {
                    const int b = (int)ciL[i];
                    gb[off + i] = (unsigned short)b;
                    gy[off + i] = yL[b];
                }
                if (sub == 0 && lane == 0) {
                    double* st = p.cand_stats + (size_t)j * 4;
                    st[0] = mx; st[1] = mn; st[2] = sm; st[3] = (double)total;
                }
            }
        }
        lds_barrier();                                                // the regions are free again
    }
}

template <int R, int S, typename T, bool WIDE> int launch_stft_split_g(const StftParams& p, int x_dtype, hipStream_t s);
template <int R, int S, typename T> int launch_stft_split(const StftParams& p, int x_dtype, hipStream_t s) {
    if (sizeof(T) == 8 && x_dtype == PVX_F64) return launch_stft_split_g<R, S, T, sizeof(T) == 8>(p, x_dtype, s);
    return launch_stft_split_g<R, S, T, false>(p, x_dtype, s);
}
template <int R, int S, typename T, bool WIDE> int launch_stft_split_g(const StftParams& p, int x_dtype, hipStream_t s) {
    using G = SplitGeo<R, S, T, WIDE>;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    int teams = G::TEAMS;
    while (teams > 1 && G::total(teams) > 160 * 1024) teams--;
    const size_t lds = G::total(teams);
    if (lds > 160 * 1024) { pvx_set_error("nfft=%d needs %zu bytes of LDS in the split STFT kernel", G::N, lds); return PVX_ERR_UNSUPPORTED; }
    const void* fn = nullptr;
    const bool cand = p.cand_bin != nullptr && p.cand_y != nullptr && p.cand_stats != nullptr && p.cand_cap >= G::M / 2 + 4;
    switch (x_dtype) {
        case PVX_F32: fn = cand ? (const void*)k_stft_split<R, S, T, float, true> : (const void*)k_stft_split<R, S, T, float, false>; break;
        case PVX_F64: fn = cand ? (const void*)k_stft_split<R, S, T, double, true> : (const void*)k_stft_split<R, S, T, double, false>; break;
        case PVX_I16: fn = cand ? (const void*)k_stft_split<R, S, T, int16_t, true> : (const void*)k_stft_split<R, S, T, int16_t, false>; break;
        default: pvx_set_error("bad x_dtype %d", x_dtype); return PVX_ERR_INVALID;
    }
    if (lds > 64 * 1024) PVX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu < 1) per_cu = 1;
    if (per_cu * teams * S > 8) per_cu = 8 / (teams * S) > 0 ? 8 / (teams * S) : 1;
    { const int nb = pvx_resident_blocks(fn, 64 * teams * S, lds); if (nb >= 1 && nb < per_cu) per_cu = nb; }      // (registers: pvx_internal.h)
    int64_t nblocks = (int64_t)ncu * per_cu;
    const int64_t maxb = (p.ws_rows + teams - 1) / teams;
    if (nblocks > maxb) nblocks = maxb > 0 ? maxb : 1;
    dim3 grid((unsigned)nblocks), block(64 * teams * S);
    StftParams arg = p;
    void* args[] = {&arg};
    PVX_HIP_CHECK(hipLaunchKernel(fn, grid, block, args, lds, s));
    return PVX_OK;
}

template <int R, typename T> int launch_stft_r(const StftParams& p, int x_dtype, hipStream_t s) {
    using G = StftGeo<R, T>;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    int nw = 6;
    while (nw > 1 && G::total(nw) > 160 * 1024) nw--;
    const size_t lds = G::total(nw);
    const void* fn = nullptr;
    switch (x_dtype) {
        case PVX_F32: fn = (const void*)k_stft<R, T, float>; break;
        case PVX_F64: fn = (const void*)k_stft<R, T, double>; break;
        case PVX_I16: fn = (const void*)k_stft<R, T, int16_t>; break;
        default: pvx_set_error("bad x_dtype %d", x_dtype); return PVX_ERR_INVALID;
    }
    if (lds > 64 * 1024) PVX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((160 * 1024) / lds);                            // workgroups that fit a CU's LDS side by side
    if (per_cu < 1) per_cu = 1;
    if (per_cu * nw > 16) per_cu = 16 / nw > 0 ? 16 / nw : 1;
    { const int nb = pvx_resident_blocks(fn, 64 * nw, lds); if (nb >= 1 && nb < per_cu) per_cu = nb; }      // (registers: pvx_internal.h)
    int64_t nblocks = (int64_t)ncu * per_cu;
    const int64_t maxb = (p.ws_rows + nw - 1) / nw;
    if (nblocks > maxb) nblocks = maxb > 0 ? maxb : 1;
    dim3 grid((unsigned)nblocks), block(64 * nw);
    switch (x_dtype) {
        case PVX_F32: hipLaunchKernelGGL((k_stft<R, T, float>), grid, block, lds, s, p); break;
        case PVX_F64: hipLaunchKernelGGL((k_stft<R, T, double>), grid, block, lds, s, p); break;
        default: hipLaunchKernelGGL((k_stft<R, T, int16_t>), grid, block, lds, s, p); break;
    }
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}

}  // namespace

int pvx_stft_supported(int nfft, int precision) {
    return (precision == 64 || precision == 32) && (nfft == 512 || nfft == 1024 || nfft == 2048 || nfft == 4096 || nfft == 8192);
}

// spectra of workspace rows [0, ws_rows) (global rows R0-1 ..) -> spec [ws_rows][ldo] complex T (T by `precision`;
// window and twiddle table in T)
int pvx_launch_stft(const FrameParams& fp, void* spec, int64_t ldo, const void* twiddle, int x_dtype, int precision, hipStream_t s) {
    if (fp.ws_rows <= 0) return PVX_OK;
    StftParams p;
    p.x = fp.x; p.nsamp = fp.nsamp; p.sig_stride = fp.sig_stride; p.F = fp.F; p.R0 = fp.R0; p.ws_rows = fp.ws_rows;
    p.total_rows = fp.total_rows; p.hop = fp.hop; p.win = fp.win; p.twiddle = twiddle; p.spec = spec; p.ldo = ldo;
    p.cand_y = fp.cand_y; p.cand_bin = fp.cand_bin; p.cand_stats = fp.cand_stats; p.cand_cap = fp.cand_cap; p.cand_thr = fp.cand_thr;
    if (precision == 64) {
        switch (fp.nfft) {
            case 512: return launch_stft_r<4, double>(p, x_dtype, s);
            case 1024: return launch_stft_r<8, double>(p, x_dtype, s);
            case 2048: return launch_stft_r<16, double>(p, x_dtype, s);
            case 4096: return launch_stft_split<16, 2, double>(p, x_dtype, s);
            case 8192: return launch_stft_split<16, 4, double>(p, x_dtype, s);
            default: break;
        }
    } else {
        // float32: the fall-back front end when the fused kernels cannot take the plan (npks beyond their LDS) and for
        // PVHarmonic -- plain (unpacked) arithmetic, still one launch and no frame buffer
        switch (fp.nfft) {
            case 512: return launch_stft_r<4, float>(p, x_dtype, s);
            case 1024: return launch_stft_r<8, float>(p, x_dtype, s);
            case 2048: return launch_stft_r<16, float>(p, x_dtype, s);
            case 4096: return launch_stft_split<16, 2, float>(p, x_dtype, s);
            case 8192: return launch_stft_split<16, 4, float>(p, x_dtype, s);
            default: break;
        }
    }
    pvx_set_error("the fused STFT kernel does not handle nfft=%d", fp.nfft);
    return PVX_ERR_UNSUPPORTED;
}
