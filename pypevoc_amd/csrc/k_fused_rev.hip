// k_fused_rev.hip -- the fused analysis stage (window + FFT + untangle + peaks, one wave64 per frame, the instruction
// sequence of k_fused.hip / k_fused_ring.hip) with every wave on its own: a wave walks a contiguous range of rows in
// DESCENDING order over ONE private spectrum buffer.
//
//   PV.calc_fft_frame   pypevoc/PVAnalysis.py:150-158
//   PV.calc_pv_frame    pypevoc/PVAnalysis.py:160-211
//   PV.run_pv           pypevoc/PVAnalysis.py:213-264
//
// A frame's peaks need the PREVIOUS frame's spectrum at the peak bins (PV.py:171, 190).  Walking forwards that means two
// buffers per wave (k_fused.hip: one wave per SIMD at nfft 2048) or a ring of buffers shared by a workgroup whose waves
// hand their spectra to each other (k_fused_ring.hip).  Walking backwards it is free: a wave finds the peaks of frame j
// while X_j is in its buffer and stages bin, X_j[bin] and the 3-bin energy; the transform it runs next leaves X_(j-1) in
// the same buffer, the staged peaks of frame j pick their previous-spectrum values up there, and every 8 frames all 64
// lanes do the per-peak arithmetic.  So: one 8.7 KB buffer per wave, two waves per SIMD at nfft 2048, no hand-off
// between waves, no flags, no waiting for a neighbour that met a dense frame -- and since a wave's next frame now
// overlaps its current one, its samples are the ones the lane already holds moved up by hop/128 register pairs: 4
// loads per frame instead of 16 at hop = nfft/4 (H > 0 instantiations; any other hop loads the whole window).
// Per-frame arithmetic is k_fused.hip's, so results are bit-identical to it and independent of the launch geometry.
#include <stdlib.h>

#include "pvx_fft4.h"

using namespace pvxw;
using namespace pvxf;

namespace {

constexpr int GFR = 8;              // frames staged before the per-peak pass
// issue priorities of the frame loop's phases (see the loop; -D overrides for A/B builds)
#ifndef PVX_PRIO_T
#define PVX_PRIO_T 2
#endif
#ifndef PVX_PRIO_S
#define PVX_PRIO_S 1
#endif
#ifndef PVX_PRIO_C
#define PVX_PRIO_C 0
#endif

typedef unsigned short u16;
// a kept peak's results: non-temporal stores (written once, read by nobody in the launch: +1.3 % at nfft 2048 / npks 8 for leaving the caches
// to the samples and the hand-over; -DPVX_RESULTS_TEMPORAL=1: plain stores).  The zero padding of a row stays with plain stores: scattered
// 8-byte non-temporal writes cost nfft 1024 at npks 20 (one peak, nineteen zeros per row) 12 %.
#ifdef PVX_RESULTS_TEMPORAL
#define PVX_RST(ptr, idx, val) ((ptr)[idx] = (val))
#else
#define PVX_RST(ptr, idx, val) __builtin_nontemporal_store((double)(val), &(ptr)[idx])
#endif
constexpr int kDense = 64;               // slots of the dense staging: one per lane of the per-peak pass

// FusedParams::wire: the result rows leave in the wire format of the multi-GPU gather (k_wire.hip: f float64 | mag float32 | ph float32 |
// binno uint16 | totalmag float64, no realph -- the receiver rebuilds it with the expression above) instead of the reference's five
// float64 arrays: p.f / p.mag / p.ph / p.binno / p.totalmag then point at the sections of the wire block.  A rank that only feeds the
// gather writes 18 bytes per slot instead of 40 and needs no packing pass behind the analysis (which cannot run beside it: twelve
// waves of 168 registers leave a CU no register for another kernel's wave -- the pass cost a gathered step 23 us of 125).
typedef const __attribute__((address_space(4))) FusedParams* wire_kargs_t;
// (wire format 2, FusedParams::wire == 2: the float32 value the frequency is computed from instead of the frequency -- 14 bytes per slot)
__device__ __forceinline__ void emit_wire(wire_kargs_t q, int64_t i, int nbin, const PeakOut& o) {
    typedef __attribute__((address_space(1))) double gd;
    typedef __attribute__((address_space(1))) float gf;
    typedef __attribute__((address_space(1))) unsigned short gu;
    if (q->wire == 2) ((gf*)q->f)[i] = o.wu;
    else ((gd*)q->f)[i] = o.freq;
    ((gf*)q->mag)[i] = (float)o.mag;                                 // (exact: float32 values widened at precision 32)
    ((gf*)q->ph)[i] = (float)o.thisph;
    ((gu*)q->binno)[i] = (unsigned short)nbin;
}
__device__ __forceinline__ void pad_wire(wire_kargs_t q, int64_t i) {
    typedef __attribute__((address_space(1))) double gd;
    typedef __attribute__((address_space(1))) float gf;
    typedef __attribute__((address_space(1))) unsigned short gu;
    if (q->wire == 2) ((gf*)q->f)[i] = 0.f;
    else ((gd*)q->f)[i] = 0.0;
    ((gf*)q->mag)[i] = 0.f; ((gf*)q->ph)[i] = 0.f; ((gu*)q->binno)[i] = 0;
}

template <int R> struct RevGeo {
    using G = Geo<R>;
    static constexpr bool X4 = (R == 16);                            // nfft 2048: four 256-point transforms per wave (pvx_fft4.h)
    static constexpr int TW3N = X4 ? 512 : ((G::HALF + 8) & ~7);
    // block-shared tables (bytes)
    static constexpr size_t OFF_T1 = 0;                              // v2f [R][64] W_M^(l q)   | X4: [16][16] W_256^(l q)
    static constexpr size_t OFF_T2 = OFF_T1 + (X4 ? (size_t)256 * 8 : (size_t)R * 64 * 8);    // v2f [R][P] W_64^(l1 t2) | X4: none
    static constexpr size_t OFF_TW3 = OFF_T2 + (X4 ? 0 : 64 * 8);    // v2f [TW3N] W_nfft^k     | X4: [2][4][64] join / untangle twiddles of the lane
    static constexpr size_t OFF_WIN = OFF_TW3 + (size_t)TW3N * 8;    // X4: v2f [R][64] the window, lane-ordered | else: none (registers)
    static constexpr size_t OFF_FLAG = OFF_WIN + (X4 ? (size_t)R * 64 * 8 : 0);     // int [16]: wave w has left its first spectrum in the stash
    static constexpr size_t OFF_WAVE = OFF_FLAG + 64;
    // dense: the kept peaks of up to GFR frames staged back to back in kDense slots (8 < npks <= 24, see the kernel's DENSE)
    __host__ __device__ static size_t per_wave(int K, bool dense = false) {
        const size_t kpad = (size_t)((K + 3) & ~3);
        const size_t gs = (size_t)staged_frames(K, GFR);
        const size_t slots = dense ? (size_t)kDense : gs * kpad;
        size_t b = (size_t)G::BUFC * 8                               // the wave's spectrum buffer
                 + GFR * 8 * 2                                       // orow | tot
                 + (X4 ? 0 : (size_t)(G::M + 4 * R) * 4)             // y (padded, ymap<1>) | X4: none, |X|^2 is recomputed (YofX4)
                 + slots * 5 * 4                                     // sval
                 + kpad * 4 + slots * 4                              // sel | sbin
                 + (dense ? GFR * 4 : 0)                             // (dense) off: a frame's first slot
                 + GFR * 4 * 2                                       // cnt | frm
                 + ((size_t)(G::CAP + 64) * 2 > 512 ? (size_t)(G::CAP + 64) * 2 : 512);      // ci (u16) + 64 trash slots; then the 64 ranking keys (u64)
        return (b + 15) & ~(size_t)15;
    }
    __host__ __device__ static size_t total(int K, int nw, bool dense = false) { return OFF_WAVE + per_wave(K, dense) * nw; }
};

// DENSE (8 < npks <= 24; the reference's default npks is 20): the per-peak pass (flush) costs what half a frame costs whether 8 or
// 64 of its lanes hold a peak, and with a frame's peaks staged at a stride of npks it ran once per 64 / pow2(npks) frames --
// every second frame at npks 20, on signals whose frames keep 8 peaks.  Dense: a frame's kept peaks are staged behind the
// previous frame's, the pass runs when the next frame might not fit (staged + npks > 64) or GFR frames wait: once per 6 frames
// there.  The running slot count rides in bits 8.. of the loop's state word.
template <int R, int NW, typename InT, bool AL2, int H, bool DENSE = false>
__global__ __launch_bounds__(64 * NW) void k_fused_rev(FusedParams p) {
    using G = Geo<R>;
    using RG = RevGeo<R>;
    constexpr int M = G::M, P = G::P, PITCH = G::PITCH;
    constexpr bool X4 = RG::X4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int K = p.K;
    const int kpad = (K + 3) & ~3;
    const int gs = staged_frames(K, GFR);

    v2f* const t1L = (v2f*)(smem + RG::OFF_T1);
    v2f* const t2L = (v2f*)(smem + RG::OFF_T2);
    v2f* const tw3 = (v2f*)(smem + RG::OFF_TW3);
    // per-wave region: everything whose size is known at compile time first, so that those arrays are one base
    // register plus immediate offsets (each runtime offset costs a scalar register across the whole frame loop)
    // (the region's offset goes through an opaque scalar move: as a visible constant the compiler folds it into the
    // immediate offsets of every access -- 16-bit fields -- and then cannot pair the 64-bit accesses of the exchange and
    // the natural-order pass into ds_write2 / ds_read2, whose two offsets have 8 bits each: 15 more LDS instructions per
    // frame than k_fused_ring.hip, whose slot address is a run-time value anyway)
    unsigned wboff = (unsigned)(RG::OFF_WAVE + RG::per_wave(K, DENSE) * wid);
    asm volatile("" : "+s"(wboff));
    unsigned char* wb = smem + wboff;
    float2* const cur = (float2*)wb;                                // X of the row at hand
    long long* const Lorow = (long long*)(cur + G::BUFC);
    double* const Ltot = (double*)(Lorow + GFR);
    float* const Ly = (float*)(Ltot + GFR);                          // (X4: no such row)
    int* const Lcnt = (int*)(Ly + (X4 ? 0 : M + 4 * R));
    int* const Lfrm = Lcnt + GFR;
    int* const Lsoff = Lfrm + GFR;                                  // (DENSE) [GFR]: a staged frame's first slot (16-byte aligned like Lcnt)
    u16* const Lci = (u16*)(Lsoff + (DENSE ? GFR : 0));
    int* const Lsel = (int*)((unsigned char*)Lci + ((size_t)(G::CAP + 64) * 2 > 512 ? (size_t)(G::CAP + 64) * 2 : 512));
    int* const Lsbin = Lsel + kpad;
    float* const Lsval = (float*)(Lsbin + (DENSE ? kDense : gs * kpad));

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
    // (nfft 2048: from LDS, lane-ordered -- 16 conflict-free reads per frame; its 32 registers are what stands between
    // two and three waves per SIMD there)
    v2f wv[X4 ? 1 : R];
    const v2f* const winL = (const v2f*)(smem + RG::OFF_WIN) + 2 * lane;      // two register pairs per 16-byte read
    if constexpr (!X4) {
#pragma unroll
        for (int r = 0; r < R; r++) wv[r] = ((const v2f*)p.win)[lane + 64 * r];
#pragma unroll
        for (int r = 0; r < R; r++) asm volatile("" : "+v"(wv[r]));
    }

    auto XA = [](int k) -> int { if constexpr (X4) return xa4(k); else return zpad<R>(k); };      // slot of bin k in `cur`

    // ---- rows of this wave: [r0, r1), walked downwards, then row r0 - 1 (spectrum only)
    // row indices fit 32 bits (the launcher checks): 64-bit scalar arithmetic in the frame loop costs SGPR pairs
    // the workgroup's share of the rows, then its waves': the n mod NW waves that take one row more are waves 0, 1, ... --
    // consecutive waves sit on different SIMDs, so no SIMD carries two long waves while another carries none (a wave has
    // ~25 rows on BASELINE config 2: one row is 4 % of a SIMD's work; +2 % there over an even split by wave index)
    const int NB = (int)gridDim.x;
    const int R0 = (int)(p.total_rows * (int64_t)blockIdx.x / NB), R1 = (int)(p.total_rows * ((int64_t)blockIdx.x + 1) / NB);
    // (with the hand-over below only wave 0 still transforms the row under its range: it is charged one row for it, so
    // that no wave of the workgroup runs one transform longer than the others)
    const bool chained = p.stash != nullptr;
    const int nwg = R1 - R0, units = nwg + ((chained && nwg >= NW) ? 1 : 0), base = units / NW, extra = units - base * NW;
    const int dec = (chained && nwg >= NW) ? 1 : 0;                 // wave 0's share, in rows, is one less than in units
    const int r0 = R0 + wid * base + (wid < extra ? wid : extra) - (wid > 0 ? dec : 0), r1 = R0 + (wid + 1) * base + (wid + 1 < extra ? wid + 1 : extra) - dec;
    const bool idle_wave = r0 >= r1;                                // (no rows: it still helps to fill the tables below)
    const int Fi = (int)p.F;
    const int rows1 = Fi + 1;                                       // rows per signal
    // ---- the row below a wave's range (the previous spectrum of its last frame) is the FIRST row of the wave below it in the
    // workgroup: that wave leaves its spectrum in global memory (8 KB, once per launch) and raises a flag in LDS when the
    // stores have landed; this wave, 16 or 25 frames later, picks its last frame's few bins up there instead of running one
    // more transform (6 % of a wave's transforms at nfft 2048 -- and the frames a wave has are 16 or 17, not 17 or 18).
    // Same CU, so the same L2; the flag follows `s_waitcnt vmcnt(0)` behind the stores.  Not across workgroups (their wave
    // 0 computes the row as before), not into a zero row (nothing to fetch).
    const bool chain_out = p.stash != nullptr && wid + 1 < NW && r1 < R1 && ((r1 - 1) % rows1) != 0;
    const bool chain_in = p.stash != nullptr && wid > 0 && r0 > R0 && ((r0 - 1) % rows1) != 0;       // (r0 == R0: wave 0 was left without rows)
    const int glast = chain_in ? r0 : r0 - 1;                       // last row this wave transforms

    // (see k_fused_ring.hip: the flush-only kernel arguments are re-read from the kernel argument segment)
    // (in the CONSTANT address space: scalar loads.  As a generic pointer these were flat loads, and the result stores below
    // flat stores: with flat operations pending the compiler can only wait with vmcnt(0), and it did so at the start of
    // every frame's peak search -- a wait for the sample prefetch of the NEXT frame, a quarter of the wave's time)
    typedef const __attribute__((address_space(4))) FusedParams* kargs_t;
    typedef __attribute__((address_space(1))) double gdouble;
    const kargs_t kargs = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();

    v2f raw[R];
#pragma unroll
    for (int r = 0; r < R; r++) raw[r] = pvxc::splat(0.f);
    // samples of row q >= 1 of signal b, from the kernel argument segment: the frame loop carries the pointer of the row at
    // hand and steps it down by the hop (the base pointer, the signal stride and the 64-bit products stay out of the loop's
    // registers); only a wave's first row and the last row of the signal below a zero row come through here
    auto row_ptr = [&](int b, int q) -> const InT* {
        kargs_t a = kargs;
        asm volatile("" : "+s"(a));
        return (const InT*)a->x + (int64_t)b * a->sig_stride + (int64_t)(q - 1) * a->hop;
    };
    auto load_pair = [&](const InT* src, int r) {
        const InT* q = src + (X4 ? lofs4(lane) : 2 * lane) + 128 * r;      // X4: pair 4 l + u + 64 r of lane 16 u + l
        if constexpr (AL2 && sizeof(InT) == 4) raw[r] = *(const v2f*)q;
        else raw[r] = pvxc::mk(ld1(q), ld1(q + 1));
    };
    auto prefetch_part = [&](const InT* src, int part) {
        if (src == nullptr) return;
        constexpr int PR = R / 4;
#pragma unroll
        for (int r = part * PR; r < (part + 1) * PR; r++) load_pair(src, r);
    };

    // spectrum of this wave's row into `cur` (zeros for a zero row) + |X|^2 -> Ly, wave-reduced max/min/energy;
    // fetches the samples of the row below (nsrc) once the raw samples have been consumed
    auto spectrum = [&](bool zero_row, const InT* nsrc, const InT* safe_src, float& maxe, float& mine, double& tot) {
        float2* const dst = cur;
        v2f z[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            if constexpr (!X4) {
                z[r] = raw[r] * wv[r];
                asm volatile("" : "+v"(z[r]));                      // the multiply stays above the loads
            }
        }
        if constexpr (X4) {
            // the window from LDS: the register pairs (r, r + 1) of a lane sit side by side, so eight 16-byte reads, ALL in flight
            // before the first multiply (read one pair at a time into one temporary -- what the compiler made of the plain loop --
            // the frame pays sixteen LDS round trips one after the other here: the wave's longest stall)
            pvxc::v4f wq[R / 2];
#pragma unroll
            for (int m = 0; m < R / 2; m++) wq[m] = *(const pvxc::v4f*)(winL + 128 * m);
#pragma unroll
            for (int m = 0; m < R / 2; m++) asm volatile("" : "+v"(wq[m]));     // (the reads stay together, above the multiplies)
#pragma unroll
            for (int m = 0; m < R / 2; m++) {
                z[2 * m] = raw[2 * m] * pvxc::mk(wq[m].x, wq[m].y);
                z[2 * m + 1] = raw[2 * m + 1] * pvxc::mk(wq[m].z, wq[m].w);
                asm volatile("" : "+v"(z[2 * m]), "+v"(z[2 * m + 1]));          // the multiplies stay above the loads
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (H > 0) {
            // The row below is (almost always) the previous frame of the same signal: the lane's samples move up by H
            // pairs and the hop's new samples come in below them.  Done on EVERY row, without a branch -- with one, the
            // compiler parks the loaded pairs in temporaries and copies them into place at the end of the branch, i.e.
            // waits for HBM right there, once per frame.  When the row below is not such a frame (a zero row, the end of
            // the wave's range) the loads read the start of the signal and nobody uses them: the next real row
            // reloads its whole window (`full`, below).
            const InT* ns = (nsrc != nullptr) ? nsrc : safe_src;
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
            for (int j = 0; j < G::BUFC / 64; j++) dst[lane + 64 * j] = make_float2(0.f, 0.f);
            wave_sync();
            return;
        }
        float lmax = -INFINITY, lmin = INFINITY, ls0 = 0.f, ls1 = 0.f;
        v2f* dz = (v2f*)dst;
        if constexpr (X4) {
            // four 256-point transforms (one per 16-lane group), then the radix-4 join inside the untangle pass (pvx_fft4.h)
            fft4_quarters(z, dz, t1L, lane, [&]() { prefetch_part(nsrc, 1); }, [&]() { prefetch_part(nsrc, 2); }, [&]() { prefetch_part(nsrc, 3); });
            v2f tw[2][4];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int u = 0; u < 4; u++) tw[j][u] = tw3[(j * 4 + u) * 64 + lane];
            wave_sync();
            join4_untangle<256, F4::RP, 64, IdentityIA, false, false>(dz, nullptr, tw, lane, IdentityIA(), lmax, lmin, ls0, ls1);
        } else {
        dft_regs<R>(z);                                             // stage 1
        __builtin_amdgcn_sched_barrier(0);
        prefetch_part(nsrc, 1);
        // the lane's twiddles first, all in flight (a pair at a time they are R / 2 LDS round trips in a row behind the stores);
        // then two rows at a time: adjacent so that the accesses pair into ds_read2st64 / ds_write2
        v2f tq[R];
#pragma unroll
        for (int q = 1; q < R; q++) tq[q] = t1L[q * 64 + lane];
#pragma unroll
        for (int q = 1; q < R; q++) asm volatile("" : "+v"(tq[q]));
#pragma unroll
        for (int q2 = 0; q2 < R; q2 += 2) {
            const v2f pa = (q2 > 0) ? pvxc::cmul(z[q2], tq[q2]) : z[q2], pb2 = pvxc::cmul(z[q2 + 1], tq[q2 + 1]);
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
        v2f t2q[R];                                                  // (likewise: the stage's twiddles before its cross-lane steps)
#pragma unroll
        for (int t = 1; t < R; t++) t2q[t] = t2L[t * P + L1];
#pragma unroll
        for (int t = 1; t < R; t++) asm volatile("" : "+v"(t2q[t]));
#pragma unroll
        for (int t0 = 0; t0 < R; t0 += 4) {
            v2f a[4];
#pragma unroll
            for (int j = 0; j < 4; j++) a[j] = (t0 + j > 0) ? pvxc::cmul(z[t0 + j], t2q[t0 + j]) : z[t0 + j];
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
        }
        if constexpr (!X4) {                                        // (X4: the peak search takes them, peak_scan_x4_thin)
            const double lsum = (double)ls0 + (double)ls1;
            wave_max_min_sum_nn(lmax, lmin, lsum, maxe, mine, tot);
        }
        wave_sync();
    };

    // per-peak pass over this wave's staged frames [0, ng)
    int LPF = 1;
    while (LPF < K && LPF < 64) LPF <<= 1;
    auto flush = [&](int gbeg, int ng) {                           // groups [gbeg, ng)
        wave_sync();
        if constexpr (DENSE) {
            // ---- the staged peaks of frames [gbeg, ng) sit back to back: lane l takes slot base + l
            const int lnf = fresh_lane();
            kargs_t q = kargs;
            asm volatile("" : "+s"(q));
            PeakConst pc;
            pc.fstep = q->fstep; pc.dt = q->dt; pc.nfft = G::N; pc.hop = q->hop; pc.wfbin = q->wfbin;
            // (the frames' first slots and counts in ONE LDS round trip, then selects: a loop over the frames with a read each is a
            // chain of dependent round trips -- at nfft 1024 longer than the transform it sits beside)
            static_assert(GFR == 8, "two 16-byte reads per table");
            const int4 oa = *(const int4*)Lsoff, ob4 = *((const int4*)Lsoff + 1), ca = *(const int4*)Lcnt, cb4 = *((const int4*)Lcnt + 1);
            const int offs[8] = {oa.x, oa.y, oa.z, oa.w, ob4.x, ob4.y, ob4.z, ob4.w}, cnts[8] = {ca.x, ca.y, ca.z, ca.w, cb4.x, cb4.y, cb4.z, cb4.w};
            int base = 0, top = 0, cnt = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) { if (j == gbeg) { base = offs[j]; cnt = cnts[j]; } if (j == ng - 1) top = offs[j] + cnts[j]; }
            const int ent = base + lnf;
            bool valid = ent < top;
            int g = gbeg, start = base;                              // the frame of this lane's slot: the last one that starts at or before it
#pragma unroll
            for (int j = 1; j < 8; j++) { if (j > gbeg && j < ng && offs[j] <= ent) { g = j; start = offs[j]; cnt = cnts[j]; } }
            const int64_t orow = (int64_t)Lorow[g];
            gdouble* of = (gdouble*)q->f + orow * K;
            gdouble* om = (gdouble*)q->mag + orow * K;
            gdouble* op = (gdouble*)q->ph + orow * K;
            gdouble* orp = (gdouble*)q->realph + orow * K;
            gdouble* ob = (gdouble*)q->binno + orow * K;
            int nbin = 0;
            PeakOut o;
            o.freq = 0.0; o.dfb = 0.0; o.thisph = 0.0; o.mag = 0.0; o.wu = 0.f; o.valid = false;
            if (valid) {
                nbin = Lsbin[ent];
                const float* sv = Lsval + (size_t)ent * 5;
                o = peak_math<float, true>(nbin, sv[0], sv[1], sv[2], sv[3], sv[4], pc);
                valid = o.valid;
            }
            // the lanes of this frame: [start - base, start - base + cnt)
            const unsigned long long gm = (cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull)) << (start - base);
            const unsigned long long ball = __ballot(valid);
            const bool wire = q->wire != 0;                           // the rows leave in the gather's wire format (see emit_wire)
            if (valid) {
                const int oi = __popcll(ball & gm & ((1ull << lnf) - 1ull));
                if (wire) emit_wire(q, orow * K + oi, nbin, o);
                else {
                PVX_RST(ob, oi, (double)nbin);
                PVX_RST(of, oi, o.freq);
                PVX_RST(om, oi, o.mag);
                PVX_RST(op, oi, o.thisph);
                PVX_RST(orp, oi, o.thisph + kPi * o.dfb / pc.fstep);          // PV.py:207
                }
            }
            {
                // zero padding (PV.py:226-239) and the frames' scalars: eight lanes per staged frame, lane c of a frame takes the
                // columns nout + c, nout + c + 8, ... (a frame's own lanes may be one or two: they would pad for npks rounds)
                const int g2 = gbeg + (lnf >> 3), c2 = lnf & 7;
                int o2 = 0, n2 = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) { if (j == g2) { o2 = offs[j]; n2 = cnts[j]; } }
                if (g2 < ng) {
                    const unsigned long long gm2 = (n2 >= 64 ? ~0ull : ((1ull << n2) - 1ull)) << (o2 - base);
                    const int nout2 = __popcll(ball & gm2);
                    const int64_t orow2 = (int64_t)Lorow[g2];
                    gdouble* of2 = (gdouble*)q->f + orow2 * K; gdouble* om2 = (gdouble*)q->mag + orow2 * K; gdouble* op2 = (gdouble*)q->ph + orow2 * K;
                    gdouble* orp2 = (gdouble*)q->realph + orow2 * K; gdouble* ob2 = (gdouble*)q->binno + orow2 * K;
                    if (wire) { for (int j = nout2 + c2; j < K; j += 8) pad_wire(q, orow2 * K + j); }
                    else
                    for (int j = nout2 + c2; j < K; j += 8) { ob2[j] = 0.0; of2[j] = 0.0; om2[j] = 0.0; op2[j] = 0.0; orp2[j] = 0.0; }
                    if (c2 == 0) {
                        const int64_t fr = Lfrm[g2];
                        if (q->totalmag) ((gdouble*)q->totalmag)[orow2] = sqrt(Ltot[g2]);                                  // PV.py:210
                        if (q->t) ((gdouble*)q->t)[orow2] = ((double)(fr * (int64_t)pc.hop) + G::N / 2.0) / q->sr;        // PV.py:247
                    }
                }
            }
            wave_sync();
            return;
        }
        // (the lane's group and its ballot mask are worked out here, once per 8 frames: as loop invariants they are four
        // registers held through the frame loop, and at three waves per SIMD the loop has none to spare)
        const int lnf = fresh_lane();
        const int gl = lnf / LPF, e0 = lnf - gl * LPF;
        const unsigned long long gmask = (LPF == 64 ? ~0ull : ((1ull << LPF) - 1ull)) << (gl * LPF);
        kargs_t q = kargs;
        asm volatile("" : "+s"(q));                                  // loads through q stay here
        const bool wire = q->wire != 0;
        PeakConst pc;
        pc.fstep = q->fstep; pc.dt = q->dt; pc.nfft = G::N; pc.hop = q->hop; pc.wfbin = q->wfbin;
        const int g = gbeg + gl;
        const bool gvalid = g < ng;
        const int cnt = gvalid ? Lcnt[g] : -1;
        const int64_t orow = gvalid ? (int64_t)Lorow[g] : 0;
        gdouble* of = (gdouble*)q->f + orow * K;
        gdouble* om = (gdouble*)q->mag + orow * K;
        gdouble* op = (gdouble*)q->ph + orow * K;
        gdouble* orp = (gdouble*)q->realph + orow * K;
        gdouble* ob = (gdouble*)q->binno + orow * K;
        int nout = 0;
        for (int eb = 0; eb < K; eb += LPF) {
            const int e = eb + e0;
            bool valid = (cnt >= 0) && (e < cnt);
            int nbin = 0;
            PeakOut o;
            o.freq = 0.0; o.dfb = 0.0; o.thisph = 0.0; o.mag = 0.0; o.wu = 0.f; o.valid = false;
            if (valid) {
                nbin = Lsbin[g * kpad + e];
                const float* sv = Lsval + (size_t)(g * kpad + e) * 5;
                o = peak_math<float, true>(nbin, sv[0], sv[1], sv[2], sv[3], sv[4], pc);
                valid = o.valid;
            }
            const unsigned long long bal = __ballot(valid) & gmask;
            if (valid) {
                const int oi = nout + __popcll(bal & ((1ull << lnf) - 1ull));
                if (wire) emit_wire(q, orow * K + oi, nbin, o);
                else {
                PVX_RST(ob, oi, (double)nbin);
                PVX_RST(of, oi, o.freq);
                PVX_RST(om, oi, o.mag);
                PVX_RST(op, oi, o.thisph);
                PVX_RST(orp, oi, o.thisph + kPi * o.dfb / pc.fstep);          // PV.py:207
                }
            }
            nout += __popcll(bal);
        }
        if (cnt >= 0) {
            if (wire) { for (int j = nout + e0; j < K; j += LPF) pad_wire(q, orow * K + j); }
            else
            for (int j = nout + e0; j < K; j += LPF) {                // zero padding, PV.py:226-239
                ob[j] = 0.0; of[j] = 0.0; om[j] = 0.0; op[j] = 0.0; orp[j] = 0.0;
            }
            if (e0 == 0) {
                const int64_t fr = Lfrm[g];
                if (q->totalmag) ((gdouble*)q->totalmag)[orow] = sqrt(Ltot[g]);                                   // PV.py:210
                if (q->t) ((gdouble*)q->t)[orow] = ((double)(fr * (int64_t)pc.hop) + G::N / 2.0) / q->sr;         // PV.py:247
            }
        }
        wave_sync();
    };

    // ---- (signal b, row-in-signal q) of this wave's first row g = r1 - 1; rows go down by one
    // the row at hand: g (global), gq (row in its signal; 0 = the zero row), and -- valid while gq >= 1 -- its samples `csrc`
    // and its output row `orow`.  Rows go down by one: inside a signal the pointer steps down by the hop.
    int g = r1 - 1, gq;
    int orow = 0;
    const InT* csrc = (const InT*)p.x;                               // (always some address inside the input: the unused loads of the sliding window read there)
    {
        const int gb = g / rows1;                                    // (a division per wave, and one per signal boundary it crosses)
        gq = g - gb * rows1;
        if (g >= 0 && gq >= 1) { csrc = row_ptr(gb, gq); orow = gb * Fi + gq - 1; }
    }
    const int hopi = p.hop;
    if (!idle_wave && g >= glast && g >= 0 && gq >= 1) { prefetch_part(csrc, 0); prefetch_part(csrc, 1); prefetch_part(csrc, 2); prefetch_part(csrc, 3); }
    // (the first row's samples are on their way from HBM while the workgroup fills its tables: two latencies side by side --
    // every launch pays them once per wave, and BASELINE config 2 is one launch of 17 rows per wave)
    // ---- block-shared tables
    {
        const v2f* tab = (const v2f*)p.twiddle;                     // W_nfft^j, j < nfft
        constexpr int NMASK = G::N - 1;
        if constexpr (X4) {
            for (int i = threadIdx.x; i < 256; i += 64 * NW) t1L[i] = tab[((G::N / 256) * (i & 15) * (i >> 4)) & NMASK];     // [q][l] W_256^(l q)
            for (int i = threadIdx.x; i < 512; i += 64 * NW) {      // [j][u][lane]: W_N^k1 (u = 0), W_1024^(u k1); k1 = lane + 64 j
                const int ln = i & 63, u = (i >> 6) & 3, k1 = ln + 64 * (i >> 8);
                tw3[i] = tab[(u == 0 ? k1 : 2 * u * k1) & NMASK];
            }
            v2f* const wl = (v2f*)(smem + RG::OFF_WIN);             // [r / 2][lane][r & 1]: the pair 4 l + u + 64 r of lane 16 u + l
            for (int i = threadIdx.x; i < R * 64; i += 64 * NW) {
                const int r = ((i >> 7) << 1) | (i & 1), ln = (i >> 1) & 63;
                wl[i] = ((const v2f*)p.win)[lofs4(ln) / 2 + 64 * r];
            }
        } else {
            for (int i = threadIdx.x; i < R * 64; i += 64 * NW) {
                const int q = i >> 6, l = i & 63;
                t1L[i] = tab[(2 * l * q) & NMASK];                  // W_M^(l q)
            }
            for (int i = threadIdx.x; i < 64; i += 64 * NW) t2L[i] = tab[((G::N / 64) * (i % P) * (i / P)) & NMASK];   // [t2][l1]
            for (int i = threadIdx.x; i <= G::HALF; i += 64 * NW) tw3[i] = tab[i];
        }
    }
    int* const Lflag = (int*)(smem + RG::OFF_FLAG);
    if (threadIdx.x < 16) Lflag[threadIdx.x] = 0;
    __syncthreads();
    if (idle_wave) return;

    int ng = 0;
    // the loop's yes / no state in ONE scalar register (as separate bools each is a register, some a register pair, carried
    // through the frame loop -- and at three waves per SIMD the loop spills what it carries):
    //   ST_PEND   the frame staged last still waits for its previous spectrum      ST_PREV0  ... which is the caller's prev0
    //   ST_OWN    ... and every kept peak is remembered by the lane that staged it  ST_PZ     (H > 0) the row above was a zero row
    //   ST_STASH  two bits: 0 first spectrum to be written to the stash, 1 written, flag due, 2 done
    constexpr unsigned ST_PEND = 1u, ST_PREV0 = 2u, ST_OWN = 4u, ST_PZ = 8u, ST_STASH = 16u;
    constexpr unsigned ST_NST = 256u;                               // (DENSE) bits 8..: slots staged since the last per-peak pass
    unsigned st = chain_out ? 0u : 2u * ST_STASH;
    int pend_nk = 0, own_sl = -1, own_pb = 1;
    // rows the optional arguments name, as 32-bit row numbers (no such row: a number no row has): the pointers themselves are
    // re-read from the kernel argument segment where they are used, once per launch
    const int prev0_g = (p.prev0 != nullptr) ? 1 : -0x7fffffff;      // output row 0 = global row 1
    const int spec_g = (p.spec_out != nullptr && p.spec_row >= -1 && p.spec_row < 0x7fffffffLL) ? (int)p.spec_row : -0x7fffffff;
    for (; g >= glast; --g) {
        const int qn = gq >= 1 ? gq - 1 : Fi;                       // row g - 1 in its signal (below a zero row: the last row of the signal before)
        const bool zero_row = (g < 0) || (gq == 0);
        // samples and output row of row g - 1 (nullptr: nothing to load -- a zero row, or below the wave's range)
        const InT* nsrc = nullptr;
        int norow = orow - 1;
        if (g - 1 >= glast && g - 1 >= 0 && qn >= 1) {
            if (gq >= 2) nsrc = csrc - hopi;
            else { const int bn = (g - 1) / rows1; nsrc = row_ptr(bn, qn); norow = bn * Fi + qn - 1; }
        }
        if constexpr (H > 0) {
            // the first frame below a zero row: its window did not slide in
            if (!zero_row && (st & ST_PZ)) { prefetch_part(csrc, 0); prefetch_part(csrc, 1); prefetch_part(csrc, 2); prefetch_part(csrc, 3); }
            st = zero_row ? (st | ST_PZ) : (st & ~ST_PZ);
        }
        float maxe = 0.f, mine = 0.f;
        double tot = 0.0;
        // ---- issue priority by phase (s_setprio): transform 2 > previous-spectrum pick-up, flush and candidate scan 1 > the
        // candidates' fetch / ranking / staging 0.  The three waves of a SIMD run the same program; left at equal priority they
        // take turns instruction by instruction, every wave's dense phase (the transform: packed arithmetic, 4 cycles an
        // instruction) stretched by its neighbours' and every latency-bound phase (the peak search: short dependent steps between
        // LDS round trips and wave reductions) queueing for issue slots behind them.  With the phases ranked, a wave in its
        // transform runs it at the pipe's rate and the other two fit their searches into its LDS waits: the arbitration of
        // MI355X_MICROARCH.md, "Two waves per SIMD", items 2-4, used between phases of one program instead of between roles.
        // 441 -> 506 M frames/s at config 2, noise 373 -> 425, violin 373 -> 420; two levels (transform over everything else)
        // give 486, the search ranked ABOVE the transform 462; splitting the transform (stages over join) loses what the third
        // level gains (profiles/r05_ab_steps.txt).
        __builtin_amdgcn_s_setprio(PVX_PRIO_T);
        spectrum(zero_row, nsrc, csrc, maxe, mine, tot);
        __builtin_amdgcn_s_setprio(PVX_PRIO_S);
        if ((st & (3u * ST_STASH)) == ST_STASH) {
            // (one row after the stores: they have long landed, the wait is for form)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) *(volatile int*)(Lflag + wid) = 1;
            st += ST_STASH;
        } else if ((st & (3u * ST_STASH)) == 0u) {
            // (four bins per lane at a time: sixteen at once are 32 registers nobody has here)
            float2* const so = (float2*)kargs->stash + ((size_t)(blockIdx.x * NW + wid) * M + fresh_lane());
#pragma unroll
            for (int j = 0; j < R; j += 4) {
                float2 t[4];
#pragma unroll
                for (int u = 0; u < 4; u++) t[u] = cur[XA(lane + 64 * (j + u))];
#pragma unroll
                for (int u = 0; u < 4; u++) so[64 * (j + u)] = t[u];
                __builtin_amdgcn_sched_barrier(0);
            }
            st += ST_STASH;
        }
        if (st & ST_PEND) {
            // ---- the frame above (staged as group ng - 1) takes its previous spectrum from this row
            if ((st & (ST_OWN | ST_PREV0)) == ST_OWN) {
                // (the usual case: the lane that staged a peak kept its slot and bin -- one LDS round trip)
                if (own_sl >= 0) {
                    const float2 pv = cur[XA(own_pb)];
                    Lsval[(size_t)own_sl * 5 + 2] = pv.x;
                    Lsval[(size_t)own_sl * 5 + 3] = pv.y;
                }
            } else
            for (int e = lane; e < pend_nk; e += 64) {
                const int sl = (DENSE ? (int)(st / ST_NST) - pend_nk : (ng - 1) * kpad) + e;
                const int nbin = Lsbin[sl];
                float2 pv;
                if (st & ST_PREV0) { const double* pz = kargs->prev0; pv = make_float2((float)pz[2 * nbin], (float)pz[2 * nbin + 1]); }
                else pv = cur[XA(nbin)];
                Lsval[(size_t)sl * 5 + 2] = pv.x;
                Lsval[(size_t)sl * 5 + 3] = pv.y;
            }
            st &= ~ST_PEND;
            if constexpr (DENSE) {
                if (ng == GFR || (int)(st / ST_NST) + K > kDense) { flush(0, ng); ng = 0; st &= ST_NST - 1u; }
            } else
            if (ng == gs) { flush(0, ng); ng = 0; }
        }
        if (!zero_row && g >= r0) {
            bool own = false;                                       // every kept peak is remembered by the lane that staged it
            own_sl = -1;
            // PeakFinder(famp, npeaks, minrattomax) + filter_by_salience(rad=5)  (PV.py:175-178); see k_fused.hip
            double th = 0.0;
            int nk = 0;
            // candidate list (ascending bins) -> Lci
            int C;
            if constexpr (X4) C = peak_scan_x4_thin<u16, (DENSE ? 24 : 16)>(cur, p.thr, maxe, mine, tot, th, Lci, G::CAP, lane, K);
            else {
                const float maxy = __builtin_amdgcn_sqrtf(maxe);
                const double minamp = (double)maxy * p.thr;         // PF.py:60
                th = (minamp != 0.0) ? minamp * minamp - (double)mine : 0.0;
                C = peak_scan_block_thin<R, u16, (DENSE ? 24 : 16)>(Ly, mine, th, Lci, G::CAP, lane, K);
            }
            wave_sync();
            __builtin_amdgcn_s_setprio(PVX_PRIO_C);
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
                const float2 c = cur[XA(pb)];
                const float2 vm = cur[XA(pb - 1)], vp = cur[XA(pb + 1)];
                float v;
                float nb[10];
                if constexpr (X4) {
                    float2 nx[10];
#pragma unroll
                    for (int d = 1; d <= 5; d++) {
                        const int dd = d > rad ? rad : d;
                        int j0 = pb - dd, j1 = pb + dd;
                        j0 = j0 < lo ? lo : j0;
                        j1 = j1 > hi ? hi : j1;
                        nx[2 * d - 2] = cur[xa4(j0)];
                        nx[2 * d - 1] = cur[xa4(j1)];
                    }
                    v = norm2(c);
#pragma unroll
                    for (int d = 0; d < 10; d++) nb[d] = norm2(nx[d]);
                } else {
                    v = Ly[ymap<1>(pb)];
#pragma unroll
                    for (int d = 1; d <= 5; d++) {
                        const int dd = d > rad ? rad : d;
                        int j0 = pb - dd, j1 = pb + dd;
                        j0 = j0 < lo ? lo : j0;
                        j1 = j1 > hi ? hi : j1;
                        nb[2 * d - 2] = Ly[ymap<1>(j0)];
                        nb[2 * d - 1] = Ly[ymap<1>(j1)];
                    }
                }
                int bad = 0;
#pragma unroll
                for (int d = 0; d < 10; d++) bad |= (int)(nb[d] > v);
                bool take = has;
                if (C > K) {
                    // keys (score, 63 - lane): unique, ties go to the lower bin; scores >= 0: bits order like values
                    const unsigned mykey = has ? __float_as_uint(v - mine) : 0u;
                    const unsigned long long my64 = ((unsigned long long)mykey << 32) | (unsigned)(63 - lane);
                    // (the keys go through LDS, where the candidate list was -- every lane has read its entry, and a wave's LDS
                    // operations execute in order --: a 16-byte read at a wave-uniform address hands two keys to all lanes, so
                    // an entry costs a compare and an add; broadcast by `v_readlane` it cost five vector instructions, 100+
                    // per frame on a signal with a few more candidates than peaks wanted)
                    unsigned long long* const Lk = (unsigned long long*)Lci;
                    Lk[lane] = my64;                                   // (lanes from C on hold score 0: below every candidate's)
                    int rank = 0;
                    for (int j = 0; j < C; j += 4) {
                        const ulonglong2 ka = *(const ulonglong2*)(Lk + j), kb = *(const ulonglong2*)(Lk + j + 2);
                        rank += (ka.x > my64 ? 1 : 0) + (ka.y > my64 ? 1 : 0) + (kb.x > my64 ? 1 : 0) + (kb.y > my64 ? 1 : 0);
                    }
                    take = has && (rank < K);
                }
                const bool keep = take && (rad < 0 || bad == 0);
                const unsigned long long bal = __ballot(keep);
                if (keep) {
                    const int sl = (DENSE ? (int)(st / ST_NST) : ng * kpad) + lane_prefix(bal);
                    // PV.py:197-199: 3-bin energy, bin 0 excluded (1 <= pb <= M-2)
                    const float em = (pb > 1) ? __builtin_fmaf(vm.x, vm.x, vm.y * vm.y) : 0.f;
                    const float s3 = (em + __builtin_fmaf(c.x, c.x, c.y * c.y)) + __builtin_fmaf(vp.x, vp.x, vp.y * vp.y);
                    Lsbin[sl] = pb;
                    float* sv = Lsval + (size_t)sl * 5;
                    sv[0] = c.x; sv[1] = c.y; sv[4] = s3;                  // sv[2], sv[3]: the previous spectrum, one row later
                    own_sl = sl; own_pb = pb;
                }
                own = true;
                nk = __popcll(bal);
            } else {
            // radix select inlined: the call of the out-of-line version saves / restores ~100 scalar registers per frame
            int nsel;
            if constexpr (X4) nsel = peak_pick_regs<R / 2, 0, u16, true>(YofX4{cur}, Lci, Lsel, M, K, C, th, mine, lane);
            else nsel = peak_pick_regs<R / 2, 1, u16, true>(Ly, Lci, Lsel, M, K, C, th, mine, lane);
            for (int eb = 0; eb < nsel; eb += 64) {
                const int e = eb + lane;
                int pb = 0;
                if (e < nsel) pb = Lsel[e];
                bool keep;
                if constexpr (X4) keep = (p.rad <= 8) ? salient_groups<0>(YofX4{cur}, M, Lsel, eb, nsel, p.rad, lane)
                                                      : ((e < nsel) && salient<float, 0>(YofX4{cur}, M, pb, p.rad));
                else keep = (p.rad <= 8) ? salient_groups<1>(Ly, M, Lsel, eb, nsel, p.rad, lane)
                                         : ((e < nsel) && salient<float, 1>(Ly, M, pb, p.rad));
                const unsigned long long bal = __ballot(keep);
                if (keep) {
                    const int sl = (DENSE ? (int)(st / ST_NST) : ng * kpad) + nk + lane_prefix(bal);
                    const float2 c = cur[XA(pb)];
                    // PV.py:197-199: 3-bin energy, bin 0 excluded (1 <= pb <= M-2)
                    const float2 vm = cur[XA(pb - 1)], vp = cur[XA(pb + 1)];
                    const float em = (pb > 1) ? __builtin_fmaf(vm.x, vm.x, vm.y * vm.y) : 0.f;
                    const float s3 = (em + __builtin_fmaf(c.x, c.x, c.y * c.y)) + __builtin_fmaf(vp.x, vp.x, vp.y * vp.y);
                    Lsbin[sl] = pb;
                    float* sv = Lsval + (size_t)sl * 5;
                    sv[0] = c.x; sv[1] = c.y; sv[4] = s3;
                }
                nk += __popcll(bal);
            }
            }
            if (lane == 0) { Lcnt[ng] = nk; Lfrm[ng] = (int)(gq - 1); Lorow[ng] = (long long)orow; Ltot[ng] = tot; if constexpr (DENSE) Lsoff[ng] = (int)(st / ST_NST); }
            ng++;
            pend_nk = nk;
            if constexpr (DENSE) st += ST_NST * (unsigned)nk;
            st = (st & ~(ST_PREV0 | ST_OWN)) | ST_PEND | (g == prev0_g ? ST_PREV0 : 0u) | (own ? ST_OWN : 0u);
        }
        if (g == spec_g) {
            float* const so = kargs->spec_out;
#pragma unroll
            for (int j = 0; j < R; j++) {
                const float2 v = cur[XA(lane + 64 * j)];
                so[2 * (lane + 64 * j)] = v.x;
                so[2 * (lane + 64 * j) + 1] = v.y;
            }
        }
        wave_sync();                                                // cur / Ly / lists are read: free for the row below
        gq = qn;
        if (nsrc != nullptr) { csrc = nsrc; orow = norow; }
    }
    if ((st & (3u * ST_STASH)) == ST_STASH) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) *(volatile int*)(Lflag + wid) = 1;
    }
    // ---- tail.  With the hand-over, the last frame's previous spectrum comes from the stash of the wave below: the loads are
    // issued, the groups before the last are flushed while they fly, then the last group follows.
    int f1 = ng;
    unsigned long long pv64 = 0ull;
    bool late = false;
    if (chain_in && (st & ST_PEND)) {
        while (*(volatile int*)(Lflag + wid - 1) == 0) __builtin_amdgcn_s_sleep(4);
        asm volatile("" ::: "memory");                              // nothing of the stash is read before the flag
        // (same CU as the writer: same L2; this launch has read nothing of the stash before, so no older copy sits in the L1)
        const float2* stp = (const float2*)kargs->stash + (size_t)(blockIdx.x * NW + wid - 1) * M;
        if (st & ST_OWN) {
            if (own_sl >= 0) pv64 = __builtin_nontemporal_load((const unsigned long long*)(stp + own_pb));
            late = true; f1 = ng - 1;
        } else
        for (int e = lane; e < pend_nk; e += 64) {
            const int sl = (DENSE ? (int)(st / ST_NST) - pend_nk : (ng - 1) * kpad) + e;
            unsigned long long v;
            asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(stp + Lsbin[sl]) : "memory");
            Lsval[(size_t)sl * 5 + 2] = __uint_as_float((unsigned)v);
            Lsval[(size_t)sl * 5 + 3] = __uint_as_float((unsigned)(v >> 32));
        }
    }
    int f0 = 0;
    for (int pass = 0; pass < 2; pass++) {                          // (one copy of the flush code for both parts)
        if (pass == 1) {
            if (!late) break;
            if (own_sl >= 0) {
                Lsval[(size_t)own_sl * 5 + 2] = __uint_as_float((unsigned)pv64);
                Lsval[(size_t)own_sl * 5 + 3] = __uint_as_float((unsigned)(pv64 >> 32));
            }
            f0 = ng - 1; f1 = ng;
        }
        if (f1 > f0) flush(f0, f1);
    }
}

template <int R, int NW, bool DENSE = false> int launch_rev(const FusedParams& p, int x_dtype, hipStream_t s, size_t* stash_need = nullptr) {
    using RG = RevGeo<R>;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    if (p.total_rows >= 0x7fffff00LL) { pvx_set_error("the fused kernel indexes rows in 32 bits (%lld rows)", (long long)p.total_rows); return PVX_ERR_UNSUPPORTED; }
    const size_t lds = RG::total(p.K, NW, DENSE);
    if (lds > 160 * 1024) { pvx_set_error("nfft=%d npks=%d needs %zu bytes of LDS in the fused kernel", Geo<R>::N, p.K, lds); return PVX_ERR_UNSUPPORTED; }
    const bool al2 = (x_dtype == PVX_F32) && (p.hop % 2 == 0) && (p.sig_stride % 2 == 0) && (((uintptr_t)p.x) % 8 == 0);
    const int H = (p.hop == 32 * R) ? R / 4 : (p.hop == 64 * R) ? R / 2 : 0;
    const void* fn = nullptr;
#define PVX_REV_PICK(INT, AL) (H == R / 4 ? (const void*)k_fused_rev<R, NW, INT, AL, R / 4, DENSE> : H ? (const void*)k_fused_rev<R, NW, INT, AL, R / 2, DENSE> : (const void*)k_fused_rev<R, NW, INT, AL, 0, DENSE>)
    switch (x_dtype) {
        // (the dense staging's instantiations do without the 8-byte sample loads: a row's new samples are a quarter of its loads)
        case PVX_F32: if constexpr (DENSE) { fn = PVX_REV_PICK(float, false); } else { fn = al2 ? PVX_REV_PICK(float, true) : PVX_REV_PICK(float, false); } break;
        case PVX_I16: fn = PVX_REV_PICK(int16_t, false); break;
        default: pvx_set_error("the fused kernels take float32 or int16 samples (x_dtype %d: float64 is narrowed before the launch)", x_dtype); return PVX_ERR_INVALID;
    }
#undef PVX_REV_PICK
    if (lds > 64 * 1024) PVX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int blocks_per_cu = (int)((160 * 1024) / lds);
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    if (blocks_per_cu * NW > 16) blocks_per_cu = (16 / NW) > 0 ? 16 / NW : 1;
    int64_t nblocks = (int64_t)ncu * blocks_per_cu;
    if (p.blocks_override > 0) nblocks = p.blocks_override;
    const int64_t maxb = (p.total_rows + NW - 1) / NW;              // never more waves than rows
    if (nblocks > maxb) nblocks = maxb > 0 ? maxb : 1;
    dim3 grid((unsigned)nblocks), block(64 * NW);
    FusedParams arg = p;
    if (stash_need) { *stash_need = (size_t)nblocks * NW * Geo<R>::M * sizeof(float2); return PVX_OK; }
    if (arg.stash_bytes < (size_t)nblocks * NW * Geo<R>::M * sizeof(float2) || getenv("PVX_REV_NO_CHAIN") != nullptr) arg.stash = nullptr;
    void* args[] = {&arg};
    PVX_HIP_CHECK(hipLaunchKernel(fn, grid, block, args, lds, s));
    return PVX_OK;
}

}  // namespace

#ifndef PVX_REV_NW1024
#define PVX_REV_NW1024 12
#endif
// nfft 512: SIXTEEN waves per CU (four per SIMD; 127 registers, nothing in scratch): +9 .. 13 % over twelve (profiles/r06_ab_steps.txt); nfft 1024
// at sixteen needs 11 registers in scratch and runs the same as twelve
#ifndef PVX_REV_NW512
#define PVX_REV_NW512 16
#endif
int pvx_fused_rev_supported(int nfft, int precision, int K) {
    if (precision != 32) return 0;
    switch (nfft) {
        case 2048: return RevGeo<16>::total(K, 8) <= 160 * 1024;
        case 1024: return RevGeo<8>::total(K, PVX_REV_NW1024) <= 160 * 1024;
        case 512: return RevGeo<4>::total(K, 12) <= 160 * 1024;
        default: return 0;
    }
}

static int launch_fused_rev(const FusedParams& p, int nfft, int x_dtype, hipStream_t s, size_t* stash_need);
int pvx_launch_fused_rev(const FusedParams& p, int nfft, int x_dtype, hipStream_t s) { return launch_fused_rev(p, nfft, x_dtype, s, nullptr); }
size_t pvx_fused_rev_stash_bytes(const FusedParams& p, int nfft) {
    size_t need = 0;
    if (launch_fused_rev(p, nfft, PVX_F32, nullptr, &need) != PVX_OK) return 0;
    return need;
}
static int launch_fused_rev(const FusedParams& p, int nfft, int x_dtype, hipStream_t s, size_t* stash_need) {
    if (stash_need) *stash_need = 0;
    if (p.total_rows <= 0) return PVX_OK;
    // 8 < npks <= 24: the kept peaks staged densely (see the kernel's DENSE; from 25 on two full frames are all that fit either way, and the
    // dense pass is the dearer one: -3 % on a recording at npks 32); PVX_REV_NO_DENSE=1: the strided staging (A/B, tests)
    const bool dense = p.K > 8 && p.K <= 24 && getenv("PVX_REV_NO_DENSE") == nullptr;
    if (dense) {
        switch (nfft) {
            case 2048: if (getenv("PVX_REV_NW8") == nullptr) return launch_rev<16, 12, true>(p, x_dtype, s, stash_need); break;
            case 1024: return launch_rev<8, PVX_REV_NW1024, true>(p, x_dtype, s, stash_need);
            case 512: return launch_rev<4, PVX_REV_NW512, true>(p, x_dtype, s, stash_need);     // (8 < npks <= 24: sixteen waves' staging always fits)
            default: break;
        }
    }
    switch (nfft) {
        // three waves per SIMD when the waves' buffers fit (npks up to ~90), else two
        case 2048: return (RevGeo<16>::total(p.K, 12) <= 160 * 1024 && getenv("PVX_REV_NW8") == nullptr) ? launch_rev<16, 12>(p, x_dtype, s, stash_need) : launch_rev<16, 8>(p, x_dtype, s, stash_need);
        case 1024: return launch_rev<8, PVX_REV_NW1024>(p, x_dtype, s, stash_need);
        case 512: return RevGeo<4>::total(p.K, PVX_REV_NW512) <= 160 * 1024 ? launch_rev<4, PVX_REV_NW512>(p, x_dtype, s, stash_need) : launch_rev<4, 12>(p, x_dtype, s, stash_need);
        default: pvx_set_error("the fused kernel does not handle nfft=%d", nfft); return PVX_ERR_UNSUPPORTED;
    }
}
