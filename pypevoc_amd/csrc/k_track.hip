// k_track.hip -- sinusoidal track building.  Replaces PV.toSinSum (pypevoc/PVAnalysis.py:299-322)
// = SinSum.add_frame for every frame (PVAnalysis.py:871-957) with add_empty_partial (819-830),
// get_partials_idx_ending_at_frame (984-994), RegPartial.append_point (616-626), dpitch2st (62-68).
//
// The reference is a sequential loop over frames whose only carried state is "which partials
// ended at frame fr-1" -- and that set is exactly the set of valid peaks (f > 0 and mag > 0) of
// frame fr-1.  So the greedy assignment of frame fr depends on rows fr-1 and fr only and all
// frames are linked in parallel:
//   k_track_links       one wave64 per frame: greedy nearest-in-semitones assignment, new peaks in descending
//                       magnitude against previous peaks in descending magnitude; a workgroup holds a CHUNK of
//                       consecutive frames and leaves every peak's root (the first point of its partial) as far
//                       as the chunk knows it: final, or the peak of the chunk before that it continues
//   k_track_boundaries  one workgroup: exclusive scan of "partials created per frame" (creation-order numbering)
//                       and pointer jumping over the chunks' last frames only -- log2(#chunks) rounds in LDS
//   k_assign_chunked    every peak: at most one more hop to its root, then partial_id / part_start / part_len
//                       (length written by the partial's last point)
// Three launches whatever F is (the first version chained 3 + log2 F launches; for a 240-frame signal their launch
// gaps were most of the tracker's time) -- two for rows of at most 8 peaks: k_track_links_g8 (eight lanes per frame), then
// k_assign_bounds, whose every workgroup works the boundary step out for itself in LDS before it assigns its nodes.
// k_scan_counts / k_root_* / k_assign_ids remain for tables whose chunk boundaries do not fit one workgroup's LDS.
// Tiny, latency-bound integer work (<= K^2 compares per frame); no roofline claim.
//
// Exact ties.
//  * Previous partials are tried in the order sorted(zip(pmag, pidx), reverse=True) (PVAnalysis.py:893):
//    magnitude descending, then PARTIAL INDEX descending.  That order only ever decides between two unused
//    previous partials that are EXACTLY equally far (in semitones) from the new peak and have exactly equal
//    magnitudes -- and a partial's index is not known while frames are linked in parallel.  k_track_links
//    detects that situation (one extra ballot per new peak) and raises a flag; the host then re-runs the
//    whole table with k_track_sequential, the reference's loop on one wave with the partial indices at
//    hand.  Never taken on analysis output (two peaks of one frame cannot have bit-equal frequencies
//    ratios), pinned by fixture T1 (hand-built arrays through the reference).
//  * New peaks are processed in np.argsort(mag)[::-1] order (PVAnalysis.py:873-875).  For equal magnitudes
//    that order is not a property of the reference: numpy's default argsort is an unstable sort whose tie
//    order depends on the CPU it runs on (x86-simd-sort networks with AVX-512 / AVX2, insertion sort
//    elsewhere; on the AVX-512 host of this build 57 % of 8-element rows with ties come out differently from
//    kind="stable").  Here, as in the oracle: the stable order reversed, i.e. higher slot first.
#include <math.h>
#include <stdlib.h>

#include "pvx_internal.h"
#include "pvx_wave.h"

namespace {

__device__ inline void wave_sync_t() { pvxw::wave_sync(); }     // LDS hand-off inside one wave (pvx_wave.h)

// arg-min with first-index ties (np.argmin, PVAnalysis.py:920)
__device__ inline void wave_argmin(double& v, int& i) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        double u = __shfl_xor(v, o);
        int j = __shfl_xor(i, o);
        if (u < v || (u == v && j < i)) { v = u; i = j; }
    }
}

// tools/ubench/track_phases.hip builds this file with PVX_TRACK_STAMPS: phase boundaries as s_memtime stamps behind
// newbase[F] (the microbenchmark allocates 16 more entries)
#ifdef PVX_TRACK_STAMPS
#define PVX_STAMP(cond, slot) do { if (cond) p.newbase[p.F + 1 + (slot)] = (int64_t)clock64(); } while (0)
#else
#define PVX_STAMP(cond, slot) do { } while (0)
#endif

constexpr int kAmbBit = 1 << 30;   // newcount[fr] carries the frame's "exact double tie" flag in this bit
constexpr int kHasBit = 1 << 29;   // ... and "the frame has a valid peak" (max(SinSum.end), PV.py:1059) in this one

// LDS of one wave (= one frame): 4 double rows and 7 int rows of kp = K rounded up to even.  Rows 4..6 of the ints are
// the frame's result, read by the chunk step: linkL (slot in frame fr-1 | -1 new partial | -2 empty), nrkL (rank
// among the frame's new partials), succL (by slot of frame fr-1: continued); row 7 is the chunk step's root row.
struct WaveLds {
    double* d[4];
    int *a, *b, *linkL, *nrkL, *succL, *used, *rootL;
    static __host__ __device__ size_t bytes(int kp) { return (size_t)kp * 8 * 4 + (size_t)kp * 4 * 7; }
    __device__ WaveLds(unsigned char* smem, int kp, int w) {
        unsigned char* base = smem + bytes(kp) * w;
        d[0] = (double*)base; d[1] = d[0] + kp; d[2] = d[1] + kp; d[3] = d[2] + kp;
        a = (int*)(d[3] + kp); b = a + kp; linkL = b + kp; nrkL = linkL + kp; succL = nrkL + kp; used = succL + kp; rootL = used + kp;
    }
};

// ---- one frame, any K: the assignment loop (PVAnalysis.py:903-957) through LDS ---------------------------------
__device__ void links_frame_lds(const TrackParams& p, const WaveLds& L, int64_t fr, int lane, int& nnew_out, bool& amb_out, bool& has_out) {
    const int K = p.K;
    double *cm = L.d[0], *cf = L.d[1], *pm = L.d[2], *pf = L.d[3];
    int *corder = L.a, *porder = L.b, *linkL = L.linkL, *nrkL = L.nrkL, *succL = L.succL, *used = L.used;
    const double* fc = p.f + fr * K;
    const double* mc = p.mag + fr * K;
    for (int s = lane; s < K; s += 64) {
        cf[s] = fc[s]; cm[s] = mc[s];
        if (fr > 0) { pf[s] = p.f[(fr - 1) * K + s]; pm[s] = p.mag[(fr - 1) * K + s]; }
        else { pf[s] = 0.0; pm[s] = 0.0; }
        linkL[s] = -2;
        nrkL[s] = -1;
        succL[s] = 0;
    }
    pvxw::wave_sync();
    // descending-magnitude ranks of the valid entries (PVAnalysis.py:874-876, 891-893)
    int nc = 0, np = 0;
    for (int s0 = 0; s0 < K; s0 += 64) {
        const int s = s0 + lane;
        bool vc = false, vp = false;
        if (s < K) {
            vc = cf[s] > 0.0 && cm[s] > 0.0;
            vp = pf[s] > 0.0 && pm[s] > 0.0;
            if (vc) {
                int r = 0;
                for (int j = 0; j < K; j++)
                    if (cf[j] > 0.0 && cm[j] > 0.0 && (cm[j] > cm[s] || (cm[j] == cm[s] && j > s))) r++;
                corder[r] = s;
            }
            if (vp) {
                int r = 0;
                for (int j = 0; j < K; j++)
                    if (pf[j] > 0.0 && pm[j] > 0.0 && (pm[j] > pm[s] || (pm[j] == pm[s] && j > s))) r++;
                porder[r] = s;
            }
        }
        nc += __popcll(__ballot(vc));
        np += __popcll(__ballot(vp));
    }
    pvxw::wave_sync();
    const int npl = (np + 63) / 64;                                  // previous peaks per lane: i = lane*npl + j
    for (int j = 0; j < npl; j++) { const int i = lane * npl + j; if (i < np) used[i] = 0; }
    pvxw::wave_sync();
    int nnew = 0;
    bool amb_any = false;
    for (int c = 0; c < nc; c++) {                                   // PVAnalysis.py:903
        const int s = corder[c];
        const double fcur = cf[s];
        double best = INFINITY;
        int bj = -1;
        for (int j = 0; j < npl; j++) {
            const int i = lane * npl + j;
            if (i < np && !used[i]) {
                const double st = fabs(17.312 * (fcur / pf[porder[i]] - 1.0));   // dpitch2st, PVAnalysis.py:62-68, 914
                if (st < best) { best = st; bj = j; }
            }
        }
        const double m = pvxw::wave_min(best);
        // lanes hold ascending index ranges and keep their first minimum: the lowest lane at the minimum has the
        // first index (np.argmin, PVAnalysis.py:920)
        const unsigned long long bal = __ballot(bj >= 0 && best == m);
        const bool hit = bal != 0ull && m < p.maxjmp;                // PVAnalysis.py:923
        if (hit) {
            const int wl = __ffsll((long long)bal) - 1;
            const int wj = __builtin_amdgcn_readlane(bj, wl);
            const int myo = bj >= 0 ? porder[lane * npl + bj] : 0;
            const double wm = pvxw::rl_d(pm[myo], wl);
            const int wo = __builtin_amdgcn_readlane(myo, wl);
            // another unused previous partial exactly as near AND exactly as strong as the winner: the reference
            // would let the partial index decide (see the header)
            bool amb = false;
            for (int j = 0; j < npl; j++) {
                const int i = lane * npl + j;
                if (i < np && !used[i] && !(lane == wl && j == wj)) {
                    const int o = porder[i];
                    const double st = fabs(17.312 * (fcur / pf[o] - 1.0));
                    amb = amb || (st == m && pm[o] == wm);
                }
            }
            amb_any = amb_any || (__ballot(amb) != 0ull);
            if (lane == wl) used[lane * npl + wj] = 1;
            if (lane == 0) { linkL[s] = wo; succL[wo] = 1; }
        } else {
            if (lane == 0) { linkL[s] = -1; nrkL[s] = nnew; }        // add_empty_partial
            nnew++;
        }
        pvxw::wave_sync();
    }
    nnew_out = nnew; amb_out = amb_any; has_out = nc > 0;
}

// ---- one frame, K <= 64 NPL, with nothing but registers inside the loops ---------------------------------------
// One wave alone on its SIMD issues an instruction every ~4.5 cycles, so a K = 100 frame costs what its instruction
// count says: a first version spent 135 k cycles ranking through dependent LDS reads, 1 100 cycles per assignment
// step, and waited for a global store acknowledgement in every step (82 us per frame).  Here: ranks by readlane
// broadcast, the previous peaks of a lane (NPL consecutive ranks), the frame's new peaks (rank = lane + 64 q) and the
// "unused" flags in registers, the minimum by v_min_f64 over DPP with row broadcasts, the first index by ballots;
// the exact-tie test only runs when two distances are bit-equal; results reach global memory once, after the loop.
template <int CTRL, int ROWS> __device__ __forceinline__ double dpp_keep(double v) {   // rows not in ROWS keep v
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp((int)b, (int)b, CTRL, ROWS, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), CTRL, ROWS, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double min_nn(double a, double b) {     // neither is a NaN: no canonicalising v_max first
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double wave_min_pos(double v) {         // v >= 0 or +inf everywhere; uniform result
    v = min_nn(v, pvxw::dpp_d<0xB1>(v));
    v = min_nn(v, pvxw::dpp_d<0x4E>(v));
    v = min_nn(v, pvxw::dpp_d<0x141>(v));
    v = min_nn(v, pvxw::dpp_d<0x140>(v));
    v = min_nn(v, dpp_keep<0x142, 0xa>(v));                        // row_bcast:15 -> rows 1, 3
    v = min_nn(v, dpp_keep<0x143, 0xc>(v));                        // row_bcast:31 -> rows 2, 3
    return pvxw::rl_d(v, 63);
}

template <int NPL>
__device__ __forceinline__ void links_frame_reg(const TrackParams& p, const WaveLds& L, int64_t fr, int lane, int& nnew_out, bool& amb_out, bool& has_out) {
    const int K = p.K;
    double *sf = L.d[0], *pfs = L.d[1], *pms = L.d[2];               // by rank: new peaks' frequency; previous f, mag
    int *sslot = L.a, *pslot = L.b, *linkL = L.linkL, *nrkL = L.nrkL, *succL = L.succL;   // rank -> slot
    // lane holds slots lane + 64 q
    double cfv[NPL], pfv[NPL], kc[NPL], kpv[NPL], pmv[NPL];
    bool vc[NPL], vp[NPL];
#pragma unroll
    for (int q = 0; q < NPL; q++) {
        const int s = lane + 64 * q;
        double cm = 0.0;
        cfv[q] = 0.0; pfv[q] = 0.0; pmv[q] = 0.0;
        if (s < K) {
            cfv[q] = p.f[fr * K + s]; cm = p.mag[fr * K + s];
            if (fr > 0) { pfv[q] = p.f[(fr - 1) * K + s]; pmv[q] = p.mag[(fr - 1) * K + s]; }
            linkL[s] = -2; nrkL[s] = -1; succL[s] = 0;
        }
        vc[q] = cfv[q] > 0.0 && cm > 0.0;
        vp[q] = pfv[q] > 0.0 && pmv[q] > 0.0;
        kc[q] = vc[q] ? cm : -1.0;                                       // invalid entries rank below every valid one
        kpv[q] = vp[q] ? pmv[q] : -1.0;
    }
    // descending-magnitude ranks of the valid entries, ties: higher slot first (PVAnalysis.py:874-876, 891-893)
    int rc[NPL], rp[NPL];
#pragma unroll
    for (int q = 0; q < NPL; q++) { rc[q] = 0; rp[q] = 0; }
#pragma unroll
    for (int qj = 0; qj < NPL; qj++) {
        const int jn = K - 64 * qj < 64 ? K - 64 * qj : 64;
        for (int j = 0; j < jn; j++) {
            const double a = pvxw::rl_d(kc[qj], j), b = pvxw::rl_d(kpv[qj], j);
#pragma unroll
            for (int q = 0; q < NPL; q++) {
                // slot j + 64 qj against this lane's slot lane + 64 q: "greater, or equal and a higher slot"
                if (qj > q) { rc[q] += a >= kc[q]; rp[q] += b >= kpv[q]; }
                else if (qj < q) { rc[q] += a > kc[q]; rp[q] += b > kpv[q]; }
                else { rc[q] += (a > kc[q]) | ((a == kc[q]) & (j > lane)); rp[q] += (b > kpv[q]) | ((b == kpv[q]) & (j > lane)); }
            }
        }
    }
    int nc = 0, np = 0;
#pragma unroll
    for (int q = 0; q < NPL; q++) {
        const int s = lane + 64 * q;
        if (vc[q]) { sf[rc[q]] = cfv[q]; sslot[rc[q]] = s; }
        if (vp[q]) { pfs[rp[q]] = pfv[q]; pms[rp[q]] = pmv[q]; pslot[rp[q]] = s; }
        nc += __popcll(__ballot(vc[q]));
        np += __popcll(__ballot(vp[q]));
    }
    pvxw::wave_sync();
    PVX_STAMP(fr == 1 && lane == 0, 1);
    // previous peaks of this lane: ranks lane*NPL + j (ascending index ranges per lane: the lowest lane at the minimum
    // holds the first index, np.argmin, PVAnalysis.py:920); new peaks: rank lane + 64 q, broadcast by readlane
    double pfr[NPL], pmr[NPL], cfr[NPL];
    int por[NPL], csr[NPL];
    bool un[NPL];
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        const int i = lane * NPL + j;
        un[j] = i < np;
        pfr[j] = un[j] ? pfs[i] : 1.0;
        pmr[j] = un[j] ? pms[i] : 0.0;
        por[j] = un[j] ? pslot[i] : 0;
        const int c = lane + 64 * j;
        cfr[j] = c < nc ? sf[c] : 0.0;
        csr[j] = c < nc ? sslot[c] : 0;
    }
    int nnew = 0;
    bool amb_any = false;
#pragma unroll
    for (int q = 0; q < NPL; q++) {
        const int cn = nc - 64 * q < 64 ? nc - 64 * q : 64;
        for (int cc = 0; cc < cn; cc++) {                             // PVAnalysis.py:903
            const double fcur = pvxw::rl_d(cfr[q], cc);
            const int s = __builtin_amdgcn_readlane(csr[q], cc);
            double sm[NPL];
            double best = INFINITY;
#pragma unroll
            for (int j = 0; j < NPL; j++) {
                const double st = fabs(17.312 * (fcur / pfr[j] - 1.0));      // dpitch2st, PVAnalysis.py:62-68, 914
                sm[j] = un[j] ? st : INFINITY;
                best = min_nn(best, sm[j]);
            }
            const double m = wave_min_pos(best);
            if (m < p.maxjmp) {                                       // PVAnalysis.py:923 (m finite: some unused peak has it)
                unsigned long long e[NPL], any = 0ull;
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < NPL; j++) { e[j] = __ballot(sm[j] == m); any |= e[j]; cnt += __popcll(e[j]); }
                const int wl = __ffsll((long long)any) - 1;
                int wj = NPL - 1;
#pragma unroll
                for (int j = NPL - 2; j >= 0; j--) wj = ((e[j] >> wl) & 1ull) ? j : wj;
                int wo = __builtin_amdgcn_readlane(por[NPL - 1], wl);
#pragma unroll
                for (int j = NPL - 2; j >= 0; j--) { const int o = __builtin_amdgcn_readlane(por[j], wl); wo = (wj == j) ? o : wo; }
                if (cnt > 1) {
                    // two unused previous peaks exactly as near: if they are also exactly as strong as the winner
                    // the reference lets the partial index decide (see the header)
                    double wm = pvxw::rl_d(pmr[NPL - 1], wl);
#pragma unroll
                    for (int j = NPL - 2; j >= 0; j--) { const double o = pvxw::rl_d(pmr[j], wl); wm = (wj == j) ? o : wm; }
                    int same = 0;
#pragma unroll
                    for (int j = 0; j < NPL; j++) same += __popcll(e[j] & __ballot(pmr[j] == wm));
                    amb_any = amb_any || same > 1;
                }
#pragma unroll
                for (int j = 0; j < NPL; j++) un[j] = un[j] && !(lane == wl && wj == j);
                if (lane == 0) { linkL[s] = wo; succL[wo] = 1; }
            } else {
                if (lane == 0) { linkL[s] = -1; nrkL[s] = nnew; }     // add_empty_partial
                nnew++;
            }
        }
    }
    nnew_out = nnew; amb_out = amb_any; has_out = nc > 0;
}

// One wave per frame (p.fpw frames one after the other when rows are short), a workgroup per chunk of
// p.chunk = waves * fpw consecutive frames, one LDS block (WaveLds) per frame of the chunk.  NPL > 0: K <= 64 NPL.
template <int NPL>
__global__ __launch_bounds__(1024) void k_track_links(TrackParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int K = p.K, CL = p.chunk, fpw = p.fpw;
    const int kp = (K + 1) & ~1;
    const int64_t fb = (int64_t)blockIdx.x * CL;
    PVX_STAMP(fb + wid == 1 && lane == 0, 0);
    for (int j = 0; j < fpw; j++) {
        const int t = wid * fpw + j;
        const int64_t fr = fb + t;
        if (fr >= p.F) break;
        const WaveLds L(smem, kp, t);
        int nnew = 0;
        bool amb_any = false, has = false;
        if constexpr (NPL > 0) links_frame_reg<NPL>(p, L, fr, lane, nnew, amb_any, has);
        else links_frame_lds(p, L, fr, lane, nnew, amb_any, has);
        if (lane == 0) p.newcount[fr] = nnew | (amb_any ? kAmbBit : 0) | (has ? kHasBit : 0);
    }
    __syncthreads();
    PVX_STAMP(fb + wid == 1 && lane == 0, 2);
    // the chunk's roots, frame after frame (one wave: the steps are dependent): a new partial's root is its own
    // node, a continued peak takes its predecessor's -- the chunk's first frame points into the frame before it
    if (wid == 0) {
        for (int t = 0; t < CL && fb + t < p.F; t++) {
            const WaveLds Lt(smem, kp, t);
            const int* rprev = t > 0 ? WaveLds(smem, kp, t - 1).rootL : nullptr;
            const int64_t base = (fb + t) * K;
            for (int s = lane; s < K; s += 64) {
                const int l = Lt.linkL[s];
                Lt.rootL[s] = l == -2 ? -1 : (l == -1 ? (int)(base + s) : (t == 0 ? (int)(base - K + l) : rprev[l]));
            }
            pvxw::wave_sync();
        }
    }
    __syncthreads();
    // every frame defines its own link and root rows and the succ row of the frame before it (no memset)
    for (int j = 0; j < fpw; j++) {
        const int t = wid * fpw + j;
        const int64_t fr = fb + t;
        if (fr >= p.F) break;
        const WaveLds L(smem, kp, t);
        for (int s = lane; s < K; s += 64) {
            const int l = L.linkL[s];                                // slot in frame fr-1 | -1 new partial | -2 empty
            p.link[fr * K + s] = l >= 0 ? l : (l == -2 ? -1 : -(L.nrkL[s] + 2));   // the table's code (pvx_internal.h)
            p.root[fr * K + s] = L.rootL[s];
            if (fr > 0) p.succ[(fr - 1) * K + s] = (unsigned char)L.succL[s];
            if (fr == p.F - 1) p.succ[fr * K + s] = 0;
        }
    }
    PVX_STAMP(fb + wid == 1 && lane == 0, 3);
}

// ---- npks <= 8: a frame per LANE --------------------------------------------------------------------------------
// A frame of 8 peaks keeps 8 of a wave's 64 lanes busy in k_track_links and still pays every wave-wide step of the loop
// (1 200 wave instructions per frame: the tracker took longer than the analysis it follows).  With rows this short a lane
// does a whole frame by itself -- both rows in registers, every loop unrolled over the 8 slots, no cross-lane step at all:
// 64 frames per wave in ~2 500 instructions.  fc / fp goes through the previous peak's reciprocal (one division per previous
// peak instead of one per pair: estimate, exact residual, one correction -- within one ulp of the reference's division, equal
// to it almost always); a frame where a comparison is decided by less than that error is flagged and the table rebuilt by the
// sequential kernel, which divides.
// A workgroup is a chunk of 256 consecutive frames: it leaves the frames' links, the chunk-local creation ranks
// (newbase[fr] = new partials of the chunk's earlier frames; the chunk's total in chunktot[c]) and every node's root as far
// as the chunk knows it (pointer doubling in LDS, 8 rounds at most).
constexpr int KL = 8;        // slots per frame in registers
constexpr int CLL = 256;     // frames per workgroup = chunk
__global__ __launch_bounds__(CLL) void k_track_links_lane(TrackParams p) {
    __shared__ int R[CLL * KL];
    __shared__ int wtot[CLL / 64], wamb[CLL / 64], wlast[CLL / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, K = p.K;
    const int64_t fb = (int64_t)blockIdx.x * CLL, fr = fb + tid;
    const bool live = fr < p.F;
    double cf[KL], cm[KL], pf[KL], pm[KL];
#pragma unroll
    for (int s = 0; s < KL; s++) {
        cf[s] = cm[s] = pf[s] = pm[s] = 0.0;
        if (live && s < K) {
            cf[s] = p.f[fr * K + s]; cm[s] = p.mag[fr * K + s];
            if (fr > 0) { pf[s] = p.f[(fr - 1) * K + s]; pm[s] = p.mag[(fr - 1) * K + s]; }
        }
    }
    // rows of more than KL slots: this kernel is right as long as the slots from KL on hold no valid peak (TrackParams::wide)
    // (a workgroup that meets such a frame says so and leaves: the launch's result will not be used)
    if (K > KL) {
        bool widefr = false;
        if (live) for (int s = KL; s < K; s++) widefr = widefr || (p.f[fr * K + s] > 0.0 && p.mag[fr * K + s] > 0.0);
        if (__syncthreads_or(widefr)) {
            if (tid == 0) { *p.wide = p.gen; *p.wide_dev = p.gen; }
            return;
        }
    }
    // valid entries (PVAnalysis.py:874-876, 887) and their descending-magnitude ranks, ties: higher slot first
    // (np.argsort(mag)[::-1] / sorted(zip(pmag, pidx), reverse=True): see the header)
    bool vc[KL], vp[KL];
    int rc[KL], rp[KL], nc = 0;
#pragma unroll
    for (int s = 0; s < KL; s++) { vc[s] = cf[s] > 0.0 && cm[s] > 0.0; vp[s] = pf[s] > 0.0 && pm[s] > 0.0; nc += vc[s]; }
#pragma unroll
    for (int s = 0; s < KL; s++) {
        int a = 0, b = 0;
#pragma unroll
        for (int j = 0; j < KL; j++) {
            if (j == s) continue;
            a += vc[j] && (j > s ? cm[j] >= cm[s] : cm[j] > cm[s]);
            b += vp[j] && (j > s ? pm[j] >= pm[s] : pm[j] > pm[s]);
        }
        rc[s] = vc[s] ? a : KL;
        rp[s] = vp[s] ? b : KL;
    }
    // the new peaks by rank (frequency, slot); the previous peaks stay in their slots, with their reciprocals and ranks
    double cfs[KL], rpf[KL];
    int csl[KL];
    bool amb_any = false;
#pragma unroll
    for (int c = 0; c < KL; c++) {
        cfs[c] = 0.0; csl[c] = 0;
#pragma unroll
        for (int s = 0; s < KL; s++)
            if (rc[s] == c) { cfs[c] = cf[s]; csl[c] = s; }
        const double pfc = vp[c] ? pf[c] : 1.0;
        rpf[c] = 1.0 / pfc;
        // (a frequency whose reciprocal or quotients could leave the normal range: let the exact loop build the table)
        if (vp[c] && !(pfc > 1e-290 && pfc < 1e290)) amb_any = true;
    }
    // the assignment loop (PVAnalysis.py:903-957)
    int link[KL], nrk[KL];
#pragma unroll
    for (int s = 0; s < KL; s++) { link[s] = -2; nrk[s] = -1; }
    unsigned used = 0u;
    int nnew = 0;
#pragma unroll
    for (int c = 0; c < KL; c++) {
        if (c < nc) {
            const double fcur = cfs[c];
            double st[KL], best = INFINITY, wm = 0.0;
            int wo = -1, wr = KL;
#pragma unroll
            for (int i = 0; i < KL; i++) {
                // fcur / pf[i] through the previous peak's reciprocal: estimate, exact residual, one correction step.  That is the
                // correctly rounded quotient whenever q0 is a faithful estimate; RN(fcur RN(1 / pf)) can be 1.5 ulp off, so for rare
                // pairs q may differ from the reference's division by one ulp -- which only matters where a comparison below is
                // decided by less than that: those frames go to the exact sequential kernel (`near`, below).
                // Then dpitch2st, PVAnalysis.py:62-68, 914
                const double q0 = fcur * rpf[i];
                const double q = __builtin_fma(__builtin_fma(-q0, pf[i], fcur), rpf[i], q0);
                st[i] = (vp[i] && !((used >> i) & 1u)) ? fabs(17.312 * (q - 1.0)) : INFINITY;
                // the first minimum in the order of the previous partials (np.argmin over the sorted list, PVAnalysis.py:893, 920)
                if (st[i] < best || (st[i] == best && rp[i] < wr)) { best = st[i]; wm = pm[i]; wo = i; wr = rp[i]; }
            }
            // a one-ulp error of a quotient q (<= q 2^-52) moves its st by <= 17.312 q 2^-52: where the threshold test or the choice of
            // the nearest partial hangs on less than eps (sixteen times that at q ~ 1; st >= 0.5 sets no link at the default jump),
            // the frame is left to k_track_sequential, which divides (the table is then the reference's whatever the rounding was)
            if (wo >= 0 && best < INFINITY) {                       // (best at infinity: every previous peak is taken -- a new partial, nothing hangs on rounding)
                const double eps = 0x1p-46 * (1.0 + best);
                bool near = fabs(best - p.maxjmp) <= eps;
                if (best < p.maxjmp) {
#pragma unroll
                    for (int i = 0; i < KL; i++) near = near || (i != wo && st[i] - best <= eps);     // (unavailable ones are at infinity)
                }
                amb_any = amb_any || near;
            }
            if (wo >= 0 && best < p.maxjmp) {                       // PVAnalysis.py:923
                // another unused previous partial exactly as near AND exactly as strong as the winner: the reference would
                // let the partial index decide (see the header)
                int same = 0;
#pragma unroll
                for (int i = 0; i < KL; i++) same += (st[i] == best && pm[i] == wm);
                amb_any = amb_any || same > 1;
                used |= 1u << wo;
#pragma unroll
                for (int s = 0; s < KL; s++) if (csl[c] == s) link[s] = wo;
            } else {
#pragma unroll
                for (int s = 0; s < KL; s++) if (csl[c] == s) { link[s] = -1; nrk[s] = nnew; }     // add_empty_partial
                nnew++;
            }
        }
    }
    const unsigned succ = used;                                      // by slot of frame fr-1: continued
    // ---- the frame's rows of the table's workspace
    if (live) {
#pragma unroll
        for (int s = 0; s < KL; s++) {
            if (s < K) {
                p.link[fr * K + s] = link[s] >= 0 ? link[s] : (link[s] == -2 ? -1 : -(nrk[s] + 2));   // the table's code (pvx_internal.h)
                if (fr > 0) p.succ[(fr - 1) * K + s] = (unsigned char)((succ >> s) & 1u);
                if (fr == p.F - 1) p.succ[fr * K + s] = 0;
            }
        }
        for (int s = KL; s < K; s++) {                              // (empty slots of a wide row)
            p.link[fr * K + s] = -1;
            if (fr > 0) p.succ[(fr - 1) * K + s] = 0;
            if (fr == p.F - 1) p.succ[fr * K + s] = 0;
            p.root[fr * K + s] = -1;
        }
        p.newcount[fr] = nnew | (amb_any ? kAmbBit : 0) | (nc > 0 ? kHasBit : 0);
    }
    // ---- creation ranks inside the chunk, the chunk's totals
    const int mine = live ? nnew : 0;
    int inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    const unsigned long long bamb = __ballot(live && amb_any), bhas = __ballot(live && nc > 0);
    if (lane == 63) wtot[wid] = inc;
    if (lane == 0) { wamb[wid] = bamb != 0ull; wlast[wid] = bhas ? (int)(fb + wid * 64 + 63 - __builtin_clzll(bhas)) : -1; }
    // ---- roots: a new partial's root is its own node, a continued peak starts at its predecessor's node; pointer doubling
    // until every node of the chunk names a root or a node of the frame before the chunk.  In LDS a node is its index in
    // the chunk (frame * 8 + slot), a node of the frame before the chunk kOut + its slot, an empty slot -1.
    constexpr int kOut = 1 << 20;
#pragma unroll
    for (int s = 0; s < KL; s++)
        R[tid * KL + s] = (!live || s >= K || link[s] == -2) ? -1 : (link[s] == -1 ? tid * KL + s : (tid == 0 ? kOut + link[s] : (tid - 1) * KL + link[s]));
    __syncthreads();
    if (live) {
        int64_t before = 0;
        for (int w = 0; w < wid; w++) before += wtot[w];
        p.newbase[fr] = before + inc - mine;
    }
    if (tid == 0) {
        int tot = 0, amb = 0, last = -1;
        for (int w = 0; w < CLL / 64; w++) { tot += wtot[w]; amb |= wamb[w]; last = wlast[w] > last ? wlast[w] : last; }
        p.chunktot[blockIdx.x] = tot | (amb ? kAmbBit : 0);
        p.chunklast[blockIdx.x] = last;
    }
    for (int round = 0; round < 8; round++) {
        int moved = 0;
#pragma unroll
        for (int s = 0; s < KL; s++) {
            const int r = R[tid * KL + s];
            if (r >= 0 && r < kOut && r != tid * KL + s) {        // a node of this chunk that is not me: where does it point?
                const int rr = R[r];
                if (rr != r) { R[tid * KL + s] = rr; moved = 1; }
            }
        }
        if (!__syncthreads_or(moved)) break;
    }
    if (live) {
#pragma unroll
        for (int s = 0; s < KL; s++) {
            if (s < K) {
                const int r = R[tid * KL + s];
                p.root[fr * K + s] = r < 0 ? -1 : (r >= kOut ? (int)((fb - 1) * K + (r - kOut)) : (int)((fb + (r >> 3)) * K + (r & 7)));
            }
        }
    }
}

// The scan over the chunks' totals and the roots of the chunks' LAST frames, as k_track_boundaries does for k_track_links:
// one workgroup; chunkbase[c] = new partials before chunk c.
// (a workgroup of 1 024 threads; rb: NCH K ints of LDS)
// cb: where the scan of the chunks' totals goes ([NCH + 1]: p.chunkbase, or LDS); to_global: the result words and the resolved
// rows leave for memory (k_track_boundaries_lane; k_assign_bounds' first workgroup writes the words only).  false: TrackParams::wide
__device__ __forceinline__ bool boundaries_lane_body(const TrackParams& p, int32_t* rb, int64_t* cb, bool words, bool store_roots) {
    // (one workgroup alone on the chip: every dependent round trip to memory is 2 us of the launch, so the word that says whether
    // k_track_links_lane has given up (TrackParams::wide), the chunks' totals and their last rows are all asked for at once)
    const unsigned widev = p.wide != nullptr ? *(volatile unsigned*)p.wide_dev : 0u;
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    __shared__ int amb_s, last_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int K = p.K;
    const int64_t F = p.F;
    const int CL = p.chunk;                                       // frames per chunk: 256, or 128 (k_track_links_g8<128>)
    const int NCH = (int)((F + CL - 1) / CL);
    const int items = NCH * K;
    auto last_frame = [&](int c) { const int64_t e = ((int64_t)(c + 1)) * CL; return (e < F ? e : F) - 1; };
    const int raw0 = tid < NCH ? p.chunktot[tid] : 0, last0 = tid < NCH ? p.chunklast[tid] : -1;
    for (int w = tid; w < items; w += 1024) { const int c = (int)((unsigned)w / (unsigned)K); rb[w] = p.root[last_frame(c) * K + (w - c * K)]; }
    if (p.wide != nullptr && widev == p.gen) return false;        // (the same for every thread)
    if (tid == 0) { carry_s = 0; amb_s = 0; last_s = -1; }
    __syncthreads();
    bool amb = false;
    int last = -1;
    for (int base = 0; base < NCH; base += 1024) {
        const int c = base + tid;
        const int raw = base == 0 ? raw0 : (c < NCH ? p.chunktot[c] : 0);
        amb = amb || (raw & kAmbBit);
        if (c < NCH) { const int l = base == 0 ? last0 : p.chunklast[c]; last = l > last ? l : last; }
        const long long v = raw & (kAmbBit - 1);
        long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        long long run = carry_s + inc - v;
        for (int w = 0; w < wid; w++) run += wsum[w];
        if (c < NCH) cb[c] = run;
        __syncthreads();
        if (tid == 1023) carry_s = run + v;
        __syncthreads();
    }
    if (amb) amb_s = 1;
    if (last >= 0) atomicMax(&last_s, last);
    __syncthreads();
    if (tid == 0) {
        cb[NCH] = carry_s;
        if (words) { *p.npartials = carry_s; *p.ambiguous = amb_s; *p.maxend = last_s; }
    }
    // pointer doubling over the chunks' last frames.  An item is -1 (empty slot), a root that is final (>= 0: a node inside the
    // item's own chunk, or one that is not on a chunk's last frame), or -2 - j: "whatever item j says" (j: the last-frame node of an
    // earlier chunk the root so far names) -- the divisions that find j happen once, the rounds are two LDS reads and a write
    // (eight rounds of three divisions per item were 6 of this kernel's 12 us at BASELINE config 2)
    for (int w = tid; w < items; w += 1024) {
        const int v = rb[w];
        if (v < 0) continue;
        const int fv = (int)((unsigned)v / (unsigned)K), cv = fv / CL;    // the frame and chunk v names (CL a power of two)
        if (cv >= (int)((unsigned)w / (unsigned)K) || ((fv + 1) & (CL - 1)) != 0) continue;
        rb[w] = -2 - (cv * K + (v - fv * K));
    }
    __syncthreads();
    for (int r = 0; r < 40; r++) {                                // (a chain of n hops is done after log2 n + 1 rounds)
        int moved = 0;
        for (int w = tid; w < items; w += 1024) {
            const int x = rb[w];
            if (x <= -2) { rb[w] = rb[-2 - x]; moved = 1; }       // (whichever of item j's values this round: both are on the way)
        }
        if (!__syncthreads_or(moved)) break;
    }
    if (store_roots)
        for (int w = tid; w < items; w += 1024) { const int c = (int)((unsigned)w / (unsigned)K); p.root[last_frame(c) * K + (w - c * K)] = rb[w]; }
    return true;
}

__global__ __launch_bounds__(1024) void k_track_boundaries_lane(TrackParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    (void)boundaries_lane_body(p, (int32_t*)smem, p.chunkbase, true, true);     // [NCH][K]
}

// The boundary step and k_assign_chunked in ONE launch: every workgroup works the boundary step out for itself, in LDS -- a few
// thousand items, the same few microseconds for all of them side by side -- and then assigns its 1 024 nodes from what it holds:
// no launch of one workgroup that the whole chip waits for (13 us at BASELINE config 2), no dependent launch behind it.  While the
// chunks' last-frame rows are few (PVX_TRACK_NO_FUSE_ASSIGN=1: the two launches).
__global__ __launch_bounds__(1024) void k_assign_bounds(TrackParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int K = p.K, CL = p.chunk;
    const int NCH = (int)((p.F + CL - 1) / CL);
    int32_t* rb = (int32_t*)smem;                                 // [NCH][K]: the chunks' last-frame roots, resolved
    int64_t* cb = (int64_t*)(smem + (((size_t)NCH * K * 4 + 15) & ~(size_t)15));   // [NCH + 1]: new partials before each chunk
    // (what the first of a thread's nodes needs from memory is asked for before the boundary step, not after it: its root, and --
    // when that root lies in the node's own chunk, the usual case -- the root's creation rank and frame count)
    // (n < 2^31, pvx_launch_track: 32-bit quotients and a shift for the chunk -- four 64-bit divisions per node were most of this loop)
    const int64_t n = p.F * (int64_t)K;
    const int cls = 31 - __builtin_clz((unsigned)CL);               // CL = 1 << cls
    const unsigned uK = (unsigned)K;
    const int64_t i0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    int32_t r0 = -1, lk0 = 0;
    int64_t nb0 = 0;
    unsigned char sc0 = 0;
    bool pre = false;
    if (i0 < n) {
        r0 = p.root[i0];
        sc0 = p.succ[i0];
        // (r0 < n: when k_track_links_g8 gave a wide table up -- TrackParams::wide, known only inside the boundary step -- the roots are whatever the workspace held)
        const unsigned c0 = ((unsigned)i0 / uK) >> cls;
        if (r0 >= 0 && (int64_t)r0 < n && (unsigned)r0 >= (c0 << cls) * uK) { nb0 = p.newbase[(unsigned)r0 / uK]; lk0 = p.link[r0]; pre = true; }
    }
    if (!boundaries_lane_body(p, rb, cb, blockIdx.x == 0, false)) return;
    __syncthreads();
    // (at most a workgroup per CU, each taking its share of the nodes: the boundary step is worked out once per CU, not once per
    // 1 024 nodes -- 404 workgroups doing it, two to a CU, took 27 us where 202 take 10)
    for (int64_t i = i0; i < n; i += (int64_t)gridDim.x * 1024) {
        const bool first = i == i0;
        int32_t r = first ? r0 : p.root[i];
        if (r < 0) { p.partial_id[i] = -1; continue; }
        const unsigned ifr = (unsigned)i / uK, c = ifr >> cls;
        if ((unsigned)r < (c << cls) * uK) r = rb[(c - 1u) * uK + ((unsigned)r - ((c << cls) - 1u) * uK)];   // a node of the frame before the chunk: that chunk's last row
        const unsigned rfr = (unsigned)r / uK;
        const int64_t nbv = (first && pre) ? nb0 : p.newbase[rfr];
        const int32_t lkv = (first && pre) ? lk0 : p.link[r];
        const int64_t pid = nbv + cb[rfr >> cls] + (-(lkv + 2));                                // creation order, PVAnalysis.py:826
        p.partial_id[i] = (int32_t)pid;
        if (pid < p.cap) {
            if (r == (int32_t)i) p.part_start[pid] = (int32_t)rfr;
            // the last point of a partial (no peak of the next frame continues it) knows the length
            if (!(first ? sc0 : p.succ[i])) p.part_len[pid] = (int32_t)(ifr - rfr + 1u);
        }
    }
}

// ---- npks <= 8: EIGHT LANES per frame ---------------------------------------------------------------------------
// k_track_links_lane's tables from eight lanes per frame instead of one: lane l of a group holds slot l of the frame and of the
// frame before it, the assignment loop's eight steps each cost one quotient per lane and two minima over the group (v_min over
// three DPP steps: quad_perm, quad_perm, row_half_mirror) instead of eight quotients and sixteen selects in one lane.  A frame per
// lane left the chip with one wave per SIMD (202 workgroups of 256 frames at BASELINE config 2: a lone wave issues an
// instruction every 4.5 cycles, a v_cndmask every 19 -- 1 600 of its 5 400 instructions per 64 frames); here a workgroup of 1 024
// threads takes its chunk of CL = 256 frames in two passes of 128 (four waves per SIMD) -- or, while the boundary step's rows of
// twice as many chunks fit its LDS, a chunk of 128 frames in one --, the rows are read and the links written lane-contiguous.  The arithmetic -- quotient through the reciprocal, the `near` margins, the exact-tie test -- is
// k_track_links_lane's, and so are the flags that send a table to k_track_sequential.  The chunk step (creation ranks, roots by
// pointer doubling in LDS) follows on the frames' results left in LDS.
constexpr int G8T = 1024;                      // threads per workgroup
constexpr int G8F = G8T / 8;                   // frames per pass
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
template <int CL>
__global__ __launch_bounds__(G8T) void k_track_links_g8(TrackParams p) {
    __shared__ int R[CL * KL];
    __shared__ int Lnn[CL];                                              // per frame of the chunk: new partials | kAmbBit | kHasBit
    __shared__ __attribute__((aligned(16))) double Lrow[CL / G8F][G8F][3 * KL];   // per pass and frame: magnitudes' keys now | before; new frequencies by rank
    __shared__ int wtot[CL / 64], wamb[CL / 64], wlast[CL / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, K = p.K;
    const int l = tid & 7, g = tid >> 3, sh = lane & ~7;
    const int64_t fb = (int64_t)blockIdx.x * CL;
    constexpr int kOut = 1 << 20;
    // rows of more than KL slots: as k_track_links_lane (TrackParams::wide)
    if (K > KL) {
        bool widefr = false;
        for (int ps = 0; ps < CL / G8F; ps++) {
            const int64_t fr = fb + ps * G8F + g;
            if (fr < p.F) for (int s2 = KL + l; s2 < K; s2 += 8) widefr = widefr || (p.f[fr * K + s2] > 0.0 && p.mag[fr * K + s2] > 0.0);
        }
        if (__syncthreads_or(widefr)) {
            if (tid == 0) { *p.wide = p.gen; *p.wide_dev = p.gen; }
            return;
        }
    }
    // The chunk's NP = CL / 128 passes run SIDE BY SIDE: pass p is frame p * 128 + g of the chunk, and every step of the assignment
    // loop is written without a branch, so that the passes' dependent chains (quotient -> distance -> three DPP minima -> ballots)
    // interleave in one basic block -- a wave alone is a chain of ~60 dependent instructions per step, ~8 cycles apart, and four waves
    // per SIMD do not fill that.
    constexpr int NP = CL / G8F;
    double cf[NP], cm[NP], pf[NP], pm[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ps++) {
        const int64_t fr = fb + ps * G8F + g;
        cf[ps] = cm[ps] = pf[ps] = pm[ps] = 0.0;
        if (fr < p.F && l < K) {
            cf[ps] = p.f[fr * K + l]; cm[ps] = p.mag[fr * K + l];
            if (fr > 0) { pf[ps] = p.f[(fr - 1) * K + l]; pm[ps] = p.mag[(fr - 1) * K + l]; }
        }
    }
    // valid entries (PVAnalysis.py:874-876, 887) and their descending-magnitude ranks, ties: higher slot first (see the header)
    bool vc[NP], vp[NP];
    double kc[NP], kpv[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ps++) {
        vc[ps] = cf[ps] > 0.0 && cm[ps] > 0.0; vp[ps] = pf[ps] > 0.0 && pm[ps] > 0.0;
        kc[ps] = vc[ps] ? cm[ps] : -1.0; kpv[ps] = vp[ps] ? pm[ps] : -1.0;     // invalid entries rank below every valid one
        double* row = &Lrow[ps][g][0];
        row[l] = kc[ps]; row[KL + l] = kpv[ps];
    }
    pvxw::wave_sync();
    int rc[NP], rp[NP], nc[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ps++) {
        const double* row = &Lrow[ps][g][0];
        int ra = 0, rb = 0;
#pragma unroll
        for (int j = 0; j < KL; j++) {
            const double cj = row[j], pj = row[KL + j];
            ra += (cj > kc[ps]) | ((cj == kc[ps]) & (j > l));
            rb += (pj > kpv[ps]) | ((pj == kpv[ps]) & (j > l));
        }
        rc[ps] = vc[ps] ? ra : KL; rp[ps] = vp[ps] ? rb : KL;
        nc[ps] = __popc((unsigned)(__ballot(vc[ps]) >> sh) & 0xffu);
    }
#pragma unroll
    for (int ps = 0; ps < NP; ps++)
        if (vc[ps]) Lrow[ps][g][2 * KL + rc[ps]] = cf[ps];                // the new peaks' frequencies by rank
    pvxw::wave_sync();
    double cfs[NP][KL], rpf[NP];
    bool amb[NP], used[NP];
    int link[NP], nrk[NP], nnew[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ps++) {
#pragma unroll
        for (int c = 0; c < KL; c++) cfs[ps][c] = Lrow[ps][g][2 * KL + c];   // (entries from nc on: not used)
        const double pfc = vp[ps] ? pf[ps] : 1.0;
        rpf[ps] = 1.0 / pfc;
        // (a frequency whose reciprocal or quotients could leave the normal range: let the exact loop build the table)
        amb[ps] = vp[ps] && !(pfc > 1e-290 && pfc < 1e290);
        link[ps] = -2; nrk[ps] = -1; nnew[ps] = 0; used[ps] = false;
    }
    // the assignment loop (PVAnalysis.py:903-957): this lane is previous peak l, and the new peak of rank rc
#pragma unroll
    for (int c = 0; c < KL; c++) {
        bool tie[NP];
        unsigned gw_[NP];
        double st_[NP], best_[NP];
#pragma unroll
        for (int ps = 0; ps < NP; ps++) {
            const bool act = c < nc[ps];                                  // (the same for the eight lanes of a frame)
            const double fcur = cfs[ps][c];
            // fcur / pf through the previous peak's reciprocal, then dpitch2st: k_track_links_lane's expressions
            const double q0 = fcur * rpf[ps];
            const double q = __builtin_fma(__builtin_fma(-q0, pf[ps], fcur), rpf[ps], q0);
            const bool avail = vp[ps] && !used[ps];
            const double st = avail ? fabs(17.312 * (q - 1.0)) : INFINITY;
            double best = st;
            best = min_nn(best, pvxw::dpp_d<0xB1>(best));
            best = min_nn(best, pvxw::dpp_d<0x4E>(best));
            best = min_nn(best, pvxw::dpp_d<0x141>(best));
            // the first minimum in the order of the previous partials (np.argmin over the sorted list, PVAnalysis.py:893, 920)
            const bool cand = avail && st == best;
            int wr = cand ? rp[ps] : KL;
            wr = min(wr, dpp_i<0xB1>(wr));
            wr = min(wr, dpp_i<0x4E>(wr));
            wr = min(wr, dpp_i<0x141>(wr));
            const bool win = cand && rp[ps] == wr;
            const unsigned gw = (unsigned)(__ballot(win) >> sh) & 0xffu;
            const bool some = gw != 0u;                                   // an unused previous peak exists (best is finite)
            const bool hit = some && best < p.maxjmp;                     // PVAnalysis.py:923
            // where the threshold test or the choice of the nearest partial hangs on less than the quotients' rounding: see
            // k_track_links_lane
            const double eps = 0x1p-46 * (1.0 + best);
            const bool n2 = hit && !win && (st - best <= eps);            // (unavailable ones are at infinity)
            const bool near = some && (fabs(best - p.maxjmp) <= eps || ((unsigned)(__ballot(n2) >> sh) & 0xffu) != 0u);
            amb[ps] = amb[ps] || (act && near);
            const unsigned ge = (unsigned)(__ballot(st == best) >> sh) & 0xffu;
            tie[ps] = act && hit && __popc(ge) > 1;
            gw_[ps] = gw; st_[ps] = st; best_[ps] = best;
            used[ps] = used[ps] || (act && hit && win);
            const bool turn = act && rc[ps] == c;
            link[ps] = turn ? (hit ? __ffs((int)gw) - 1 : -1) : link[ps];
            nrk[ps] = (turn && !hit) ? nnew[ps] : nrk[ps];                // add_empty_partial
            nnew[ps] += (act && !hit) ? 1 : 0;
        }
        bool anytie = false;
#pragma unroll
        for (int ps = 0; ps < NP; ps++) anytie = anytie || tie[ps];
        if (__ballot(anytie) != 0ull) {
            // another unused previous partial exactly as near as the winner: if it is also exactly as strong the reference would
            // let the partial index decide (see the header)
#pragma unroll
            for (int ps = 0; ps < NP; ps++) {
                const int wl = gw_[ps] ? __ffs((int)gw_[ps]) - 1 : 0;
                const double wm = Lrow[ps][g][KL + wl];                   // (the winner is a valid peak: its key is its magnitude)
                const unsigned gs = (unsigned)(__ballot(tie[ps] && st_[ps] == best_[ps] && pm[ps] == wm) >> sh) & 0xffu;
                amb[ps] = amb[ps] || (tie[ps] && __popc(gs) > 1);
            }
        }
    }
#pragma unroll
    for (int ps = 0; ps < NP; ps++) {
        const int fl = ps * G8F + g;                                      // frame of the chunk
        const int64_t fr = fb + fl;
        const bool live = fr < p.F;
        // ---- the frame's rows of the table's workspace
        if (live) {
            if (l < K) {
                p.link[fr * K + l] = link[ps] >= 0 ? link[ps] : (link[ps] == -2 ? -1 : -(nrk[ps] + 2));   // the table's code (pvx_internal.h)
                if (fr > 0) p.succ[(fr - 1) * K + l] = (unsigned char)used[ps];      // by slot of frame fr-1: continued
                if (fr == p.F - 1) p.succ[fr * K + l] = 0;
            }
            for (int s2 = KL + l; s2 < K; s2 += 8) {                      // (empty slots of a wide row)
                p.link[fr * K + s2] = -1;
                if (fr > 0) p.succ[(fr - 1) * K + s2] = 0;
                if (fr == p.F - 1) p.succ[fr * K + s2] = 0;
                p.root[fr * K + s2] = -1;
            }
        }
        const bool amb_any = ((unsigned)(__ballot(amb[ps]) >> sh) & 0xffu) != 0u;
        if (l == 0) {
            const int v = live ? (nnew[ps] | (amb_any ? kAmbBit : 0) | (nc[ps] > 0 ? kHasBit : 0)) : 0;
            Lnn[fl] = v;
            if (live) p.newcount[fr] = v;
        }
        // roots: a new partial's root is its own node, a continued peak starts at its predecessor's node (k_track_links_lane)
        R[fl * KL + l] = (!live || l >= K || link[ps] == -2) ? -1 : (link[ps] == -1 ? fl * KL + l : (fl == 0 ? kOut + link[ps] : (fl - 1) * KL + link[ps]));
    }
    __syncthreads();
    // ---- creation ranks inside the chunk, the chunk's totals: thread t < CL is frame t
    const int64_t fr2 = fb + tid;
    const bool live2 = tid < CL && fr2 < p.F;
    const int v2 = live2 ? Lnn[tid] : 0;
    const int mine = v2 & 0xff;
    int inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    const unsigned long long bamb = __ballot((v2 & kAmbBit) != 0), bhas = __ballot((v2 & kHasBit) != 0);
    if (tid < CL) {
        if (lane == 63) wtot[wid] = inc;
        if (lane == 0) { wamb[wid] = bamb != 0ull; wlast[wid] = bhas ? (int)(fb + wid * 64 + 63 - __builtin_clzll(bhas)) : -1; }
    }
    __syncthreads();
    if (live2) {
        int64_t before = 0;
        for (int w = 0; w < wid; w++) before += wtot[w];
        p.newbase[fr2] = before + inc - mine;
    }
    if (tid == 0) {
        int tot = 0, amb = 0, last = -1;
        for (int w = 0; w < CL / 64; w++) { tot += wtot[w]; amb |= wamb[w]; last = wlast[w] > last ? wlast[w] : last; }
        p.chunktot[blockIdx.x] = tot | (amb ? kAmbBit : 0);
        p.chunklast[blockIdx.x] = last;
    }
    // pointer doubling until every node of the chunk names a root or a node of the frame before the chunk
    for (int round = 0; round < 8; round++) {
        int moved = 0;
#pragma unroll
        for (int n = tid; n < CL * KL; n += G8T) {
            const int r = R[n];
            if (r >= 0 && r < kOut && r != n) {                           // a node of this chunk that is not me: where does it point?
                const int rr = R[r];
                if (rr != r) { R[n] = rr; moved = 1; }
            }
        }
        if (!__syncthreads_or(moved)) break;
    }
#pragma unroll
    for (int n = tid; n < CL * KL; n += G8T) {
        const int fl = n >> 3, s2 = n & 7;
        if (fb + fl < p.F && s2 < K) {
            const int r = R[n];
            p.root[(fb + fl) * K + s2] = r < 0 ? -1 : (r >= kOut ? (int)((fb - 1) * K + (r - kOut)) : (int)((fb + (r >> 3)) * K + (r & 7)));
        }
    }
}

// The reference's loop as it stands (PVAnalysis.py:871-957), one wave, frames in order, partial indices at
// hand: previous partials ordered by (magnitude, partial index) descending.  Only launched when
// k_track_links met an exact double tie (see the header); O(F K) steps of one wave.
// LDS: cm cf pm pf doubles [K] | corder porder used ppid(prev partial id by slot) cpid ints [K]
__global__ __launch_bounds__(64) void k_track_sequential(TrackParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int K = p.K;
    const int kp = (K + 1) & ~1;
    double* cm = (double*)smem;
    double* cf = cm + kp;
    double* pm = cf + kp;
    double* pf = pm + kp;
    int* corder = (int*)(pf + kp);
    int* porder = corder + kp;
    int* used = porder + kp;
    int* ppid = used + kp;
    int* cpid = ppid + kp;
    long long P = 0, lastfr = -1;
    for (int s = lane; s < K; s += 64) { pf[s] = 0.0; pm[s] = 0.0; ppid[s] = -1; }
    wave_sync_t();
    for (int64_t fr = 0; fr < p.F; fr++) {
        const double* fc = p.f + fr * K;
        const double* mc = p.mag + fr * K;
        for (int s = lane; s < K; s += 64) { cf[s] = fc[s]; cm[s] = mc[s]; cpid[s] = -1; }
        wave_sync_t();
        int nc = 0, np = 0;
        for (int s0 = 0; s0 < K; s0 += 64) {
            const int s = s0 + lane;
            bool vc = false, vp = false;
            if (s < K) {
                vc = cf[s] > 0.0 && cm[s] > 0.0;
                vp = pf[s] > 0.0 && pm[s] > 0.0;
                if (vc) {
                    int r = 0;
                    for (int j = 0; j < K; j++)
                        if (cf[j] > 0.0 && cm[j] > 0.0 && (cm[j] > cm[s] || (cm[j] == cm[s] && j > s))) r++;
                    corder[r] = s;
                }
                if (vp) {
                    int r = 0;                                     // (mag, partial index) descending, PVAnalysis.py:893
                    for (int j = 0; j < K; j++)
                        if (pf[j] > 0.0 && pm[j] > 0.0 && (pm[j] > pm[s] || (pm[j] == pm[s] && ppid[j] > ppid[s]))) r++;
                    porder[r] = s;
                    used[r] = 0;
                }
            }
            nc += __popcll(__ballot(vc));
            np += __popcll(__ballot(vp));
        }
        wave_sync_t();
        for (int c = 0; c < nc; c++) {
            const int s = corder[c];
            const double fcur = cf[s];
            double best = INFINITY;
            int bi = 0x7fffffff;
            for (int i = lane; i < np; i += 64) {
                if (!used[i]) {
                    double st = fabs(17.312 * (fcur / pf[porder[i]] - 1.0));
                    if (st < best) { best = st; bi = i; }
                }
            }
            wave_argmin(best, bi);
            const bool hit = (bi != 0x7fffffff) && (best < p.maxjmp);
            if (lane == 0) {
                int32_t id;
                if (hit) { id = ppid[porder[bi]]; used[bi] = 1; }
                else { id = (int32_t)P; if (P < p.cap) { p.part_start[P] = (int32_t)fr; p.part_len[P] = 0; } }
                cpid[s] = id;
                if (id < p.cap) p.part_len[id] += 1;
            }
            if (!hit) P++;
            wave_sync_t();
        }
        for (int s = lane; s < K; s += 64) {
            p.partial_id[fr * K + s] = cpid[s];
            pf[s] = cf[s]; pm[s] = cm[s]; ppid[s] = cpid[s];
        }
        if (nc > 0) lastfr = (long long)fr;
        wave_sync_t();
    }
    if (lane == 0) { *p.npartials = P; *p.maxend = lastfr; }
}

// exclusive scan of newcount[F] (flags masked; "exact tie" OR-ed, "has a peak" -> last such frame) with a 1024-thread
// workgroup, 4 consecutive frames per thread and tile (one coalesced pass per 4096 frames).  out(i, value)
template <typename Out>
__device__ __forceinline__ void scan_counts(const TrackParams& p, long long* wsum, long long* carry_s, int* amb_s, int* last_s, Out out) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) { *carry_s = 0; *amb_s = 0; *last_s = -1; }
    __syncthreads();
    int last = -1;
    bool amb = false;
    for (int64_t base = 0; base < p.F; base += 4096) {
        const int64_t i0 = base + 4 * tid;
        int c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int raw = (i0 + u < p.F) ? p.newcount[i0 + u] : 0;
            amb = amb || (raw & kAmbBit);
            if (raw & kHasBit) last = (int)(i0 + u);
            c[u] = raw & (kHasBit - 1);
        }
        const long long sum = (long long)c[0] + c[1] + c[2] + c[3];
        long long inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            long long u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        long long run = *carry_s + inc - sum;
        for (int w = 0; w < wid; w++) run += wsum[w];
#pragma unroll
        for (int u = 0; u < 4; u++) { if (i0 + u < p.F) out(i0 + u, run); run += c[u]; }
        __syncthreads();
        if (tid == 1023) *carry_s = run;
        __syncthreads();
    }
    if (amb) *amb_s = 1;
    if (last >= 0) atomicMax(last_s, last);
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_scan_counts(TrackParams p) {
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    __shared__ int amb_s, last_s;
    scan_counts(p, wsum, &carry_s, &amb_s, &last_s, [&](int64_t i, long long v) { p.newbase[i] = v; });
    if (threadIdx.x == 0) { p.newbase[p.F] = carry_s; *p.npartials = carry_s; *p.ambiguous = amb_s; *p.maxend = last_s; }
}

// The scan, and the roots of the chunks' LAST frames: rb[c][s] starts as k_track_links left it -- final, or a node of
// the last frame of chunk c-1.  Pointer jumping between those rows only (in LDS, log2(#chunks) rounds; in place is
// safe: any value read is an ancestor), then back to p.root: every other node is now at most one hop from its root.
__global__ __launch_bounds__(1024) void k_track_boundaries(TrackParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    __shared__ int amb_s, last_s;
    int32_t* rb = (int32_t*)smem;                                // [NCH][K]
    const int tid = threadIdx.x;
    const int K = p.K, CL = p.chunk;
    const int64_t F = p.F;
    const int NCH = (int)((F + CL - 1) / CL);
    const int items = NCH * K;
    PVX_STAMP(tid == 0, 8);
    // 16 waves share this CU's issue slots (an instruction of a wave every ~18 cycles): no integer division below.
    // v / K through a float reciprocal (exact after one fix for v < 2^24, else the division); CL is a power of two.
    const float invK = 1.0f / (float)K;
    const bool small = F * (int64_t)K < (1 << 24);
    auto div_k = [&](int v) {
        if (!small) return v / K;
        int q = (int)((float)v * invK);
        if (q * K > v) q--; else if ((q + 1) * K <= v) q++;
        return q;
    };
    const int lg = 31 - __builtin_clz((unsigned)CL);
    auto last_frame = [&](int c) { const int64_t e = ((int64_t)(c + 1)) << lg; return (e < F ? e : F) - 1; };
    for (int w = tid; w < items; w += 1024) { const int c = div_k(w); rb[w] = p.root[last_frame(c) * K + (w - c * K)]; }
    scan_counts(p, wsum, &carry_s, &amb_s, &last_s, [&](int64_t i, long long v) { p.newbase[i] = v; });     // (barriers inside)
    // the three words the host waits for (they may live in page-locked host memory: single plain stores).  maxend =
    // max(SinSum.end) (PV.py:1059) = the last frame that holds a valid peak: every valid peak is a point of a partial
    if (tid == 0) { p.newbase[F] = carry_s; *p.npartials = carry_s; *p.ambiguous = amb_s; *p.maxend = last_s; }
    PVX_STAMP(tid == 0, 9);
    for (int r = 0; (1 << r) < NCH; r++) {
        int moved = 0;
        for (int w = tid; w < items; w += 1024) {
            const int v = rb[w];
            if (v < 0) continue;                                  // empty slot
            const int fv = div_k(v), cv = fv >> lg;               // the frame and chunk v names
            if (cv >= div_k(w)) continue;                         // a root inside this item's own chunk: final
            if (((fv + 1) & (CL - 1)) != 0) continue;             // a root that is not on a chunk's last frame: final
            const int wv = cv * K + (v - fv * K);                 // that last-frame node's item
            const int nv = rb[wv];
            if (nv != v) { rb[w] = nv; moved = 1; }
        }
        if (!__syncthreads_or(moved)) break;
    }
    for (int w = tid; w < items; w += 1024) { const int c = div_k(w); p.root[last_frame(c) * K + (w - c * K)] = rb[w]; }
    PVX_STAMP(tid == 0, 10);
}

__global__ __launch_bounds__(256) void k_assign_chunked(TrackParams p) {
    if (p.wide != nullptr && *(volatile unsigned*)p.wide_dev == p.gen) return;  // (as k_track_boundaries_lane)
    const int64_t n = p.F * (int64_t)p.K;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int32_t r = p.root[i];
    if (r < 0) { p.partial_id[i] = -1; return; }
    // (n < 2^31: 32-bit quotients; the chunk length is a power of two)
    const unsigned uK = (unsigned)p.K;
    const int cls = 31 - __builtin_clz((unsigned)p.chunk);
    const unsigned ifr = (unsigned)i / uK;
    if ((unsigned)r < ((ifr >> cls) << cls) * uK) r = p.root[r];         // before the chunk: a last-frame node, final by now
    const unsigned rfr = (unsigned)r / uK;
    const int64_t pid = p.newbase[rfr] + (p.chunkbase ? p.chunkbase[rfr >> cls] : 0) + (-(p.link[r] + 2));   // creation order, PVAnalysis.py:826
    p.partial_id[i] = (int32_t)pid;
    const bool last = !p.succ[i];
    if (pid < p.cap) {
        if (r == (int32_t)i) p.part_start[pid] = (int32_t)rfr;
        // the last point of a partial (no peak of the next frame continues it) knows the length:
        // one plain store per partial instead of one contended atomic per point
        if (last) p.part_len[pid] = (int32_t)(ifr - rfr + 1u);
    }
}

__global__ __launch_bounds__(256) void k_root_init(TrackParams p) {
    const int64_t n = p.F * (int64_t)p.K;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int l = p.link[i];
    int32_t r;
    if (l == -1) r = -1;
    else if (l <= -2) r = (int32_t)i;
    else r = (int32_t)(i - (i % p.K) - p.K + l);                     // (fr-1)*K + slot
    p.root[i] = r;
}

// one pointer-jumping round: root <- root[root]; in place is safe (any value read is an ancestor)
__global__ __launch_bounds__(256) void k_root_jump(TrackParams p) {
    const int64_t n = p.F * (int64_t)p.K;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t r = p.root[i];
    if (r < 0 || r == (int32_t)i) return;
    const int32_t rr = __hip_atomic_load(&p.root[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (rr != r) __hip_atomic_store(&p.root[i], rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void k_assign_ids(TrackParams p) {
    const int64_t n = p.F * (int64_t)p.K;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t r = p.root[i];
    if (r < 0) { p.partial_id[i] = -1; return; }
    const int64_t rfr = r / p.K;
    const int64_t pid = p.newbase[rfr] + (-(p.link[r] + 2));         // creation order, PVAnalysis.py:826
    p.partial_id[i] = (int32_t)pid;
    if (pid < p.cap) {
        if (r == (int32_t)i) p.part_start[pid] = (int32_t)rfr;
        // the last point of a partial (no peak of the next frame continues it) knows the length:
        // one plain store per partial instead of one contended atomic per point
        if (!p.succ[i]) p.part_len[pid] = (int32_t)(i / p.K - rfr + 1);
    }
}

}  // namespace

int pvx_launch_track(const TrackParams& p_in, hipStream_t s) {
    TrackParams p = p_in;
    if (p.F <= 0) return PVX_OK;
    const int64_t n = p.F * (int64_t)p.K;
    if (n >= 0x7fffffffLL) { pvx_set_error("F*K = %lld does not fit the 32-bit node index", (long long)n); return PVX_ERR_UNSUPPORTED; }
    const unsigned nb = (unsigned)((n + 255) / 256);
    // the wave-per-frame kernels' geometry
    const int kp = (p.K + 1) & ~1;
    const size_t per_wave = WaveLds::bytes(kp);
    if (per_wave > 160 * 1024) { pvx_set_error("npks=%d too large for the tracker", p.K); return PVX_ERR_UNSUPPORTED; }
    // frames per workgroup = chunk of the root step.  A frame with many peaks is a long serial chain that wants its
    // SIMD to itself (4 waves per workgroup: measured 74 k cycles per K = 100 frame against 98 k at 8); short chains
    // pack 16 to keep the number of chunk boundaries down.  Always within 64 KB of LDS, never below 1.
    int waves = p.K > 32 ? 4 : 16;
    while (waves > 1 && per_wave * waves > 64 * 1024) waves >>= 1;
    if (const char* e = getenv("PVX_TRACK_CHUNK")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) waves = v < waves ? v : waves; }
    // short rows on a long table: a wave takes 4 frames in turn, so a chunk is 64 frames and there are 4 x fewer
    // boundary rows for k_track_boundaries' single workgroup to jump over
    int fpw = (p.K <= 16 && p.F >= 4096 && waves == 16 && per_wave * 64 <= 64 * 1024) ? 4 : 1;
    if (const char* e = getenv("PVX_TRACK_FPW")) { const int v = atoi(e); if ((v == 1 || v == 2 || v == 4) && per_wave * waves * v <= 64 * 1024) fpw = v; }
    {
        // a frame per lane (k_track_links_lane): rows of npks <= 8 -- and wider rows (<= 64) in the hope that no frame has a
        // valid peak beyond slot 7 (a tone under the reference's default npks = 20): the kernel says so in *p.wide if one
        // has, and the caller, who waits for the result words anyway, then calls again without the word (track_on).
        // While the last-frame rows of the chunks fit one workgroup's LDS.
        // (chunks of 128 frames for k_track_links_g8 while the boundary step has the LDS for twice the rows; PVX_TRACK_CHUNK256=1: A/B)
        const bool g8 = getenv("PVX_TRACK_LANE_FRAME") == nullptr;
        // (... and while every workgroup still has a CU to itself: 0.043 -> 0.038 ms on 25 836 frames of noise; two workgroups on some CUs: no gain)
        const bool c128 = g8 && (p.F + 127) / 128 <= 256 && (size_t)((p.F + 127) / 128) * p.K * 4 <= 150 * 1024 && !getenv("PVX_TRACK_CHUNK256");
        const int CLx = c128 ? 128 : CLL;
        const int64_t nchl = (p.F + CLx - 1) / CLx;
        const size_t blds = (size_t)nchl * p.K * 4;
        const bool hope = p.K > KL;
        const bool fits = blds <= 150 * 1024 && (!hope || (p.K <= 64 && p.wide != nullptr && p.wide_dev != nullptr && p.gen != 0 && !getenv("PVX_TRACK_NO_LANE")));
        if (fits && p.chunkbase && !getenv("PVX_TRACK_CHUNK") && !getenv("PVX_TRACK_GENERIC") && !getenv("PVX_TRACK_LARGE") &&
            !getenv("PVX_TRACK_FPW") && !getenv("PVX_TRACK_WAVES")) {
            p.chunk = CLx;
            // (PVX_TRACK_LANE_FRAME=1: the frame-per-lane kernel, the same tables -- tests and A/B runs)
            if (!g8) hipLaunchKernelGGL(k_track_links_lane, dim3((unsigned)nchl), dim3(CLL), 0, s, p);
            else if (c128) hipLaunchKernelGGL(k_track_links_g8<128>, dim3((unsigned)nchl), dim3(G8T), 0, s, p);
            else hipLaunchKernelGGL(k_track_links_g8<256>, dim3((unsigned)nchl), dim3(G8T), 0, s, p);
            if (nchl * p.K <= 4096 && !getenv("PVX_TRACK_NO_FUSE_ASSIGN")) {
                const size_t flds = (((size_t)nchl * p.K * 4 + 15) & ~(size_t)15) + (size_t)(nchl + 1) * 8;
                int64_t ab = (n + 1023) / 1024;
                if (const char* e = getenv("PVX_TRACK_ASSIGN_BLOCKS")) { const long long v = atoll(e); if (v >= 1 && v < ab) ab = v; }   // tests, A/B
                else if (ab > 256) ab = 256;
                hipLaunchKernelGGL(k_assign_bounds, dim3((unsigned)ab), dim3(1024), flds, s, p);
                PVX_HIP_CHECK(hipGetLastError());
                return PVX_OK;
            }
            if (blds > 48 * 1024)
                PVX_HIP_CHECK(hipFuncSetAttribute((const void*)k_track_boundaries_lane, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds));
            hipLaunchKernelGGL(k_track_boundaries_lane, dim3(1), dim3(1024), blds, s, p);
            hipLaunchKernelGGL(k_assign_chunked, dim3(nb), dim3(256), 0, s, p);
            PVX_HIP_CHECK(hipGetLastError());
            return PVX_OK;
        }
        p.chunkbase = nullptr;
    }
    p.fpw = fpw;
    p.chunk = waves * fpw;
    const dim3 grid((unsigned)((p.F + p.chunk - 1) / p.chunk)), block(64 * waves);
    const size_t lds = per_wave * p.chunk;
#define PVX_LINKS(NPL)                                                                                                     \
    do {                                                                                                                   \
        if (lds > 48 * 1024)                                                                                               \
            PVX_HIP_CHECK(hipFuncSetAttribute((const void*)k_track_links<NPL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(k_track_links<NPL>, grid, block, lds, s, p);                                                    \
    } while (0)
    const bool generic = p.K > 256 || getenv("PVX_TRACK_GENERIC");
    if (generic) PVX_LINKS(0);
    else if (p.K <= 64) PVX_LINKS(1);
    else if (p.K <= 128) PVX_LINKS(2);
    else PVX_LINKS(4);
#undef PVX_LINKS
    const int64_t nch = (p.F + p.chunk - 1) / p.chunk;
    const size_t blds = (size_t)nch * p.K * 4;
    if (blds <= 150 * 1024 && !getenv("PVX_TRACK_LARGE")) {
        if (blds > 48 * 1024)
            PVX_HIP_CHECK(hipFuncSetAttribute((const void*)k_track_boundaries, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds));
        hipLaunchKernelGGL(k_track_boundaries, dim3(1), dim3(1024), blds, s, p);
        hipLaunchKernelGGL(k_assign_chunked, dim3(nb), dim3(256), 0, s, p);
        PVX_HIP_CHECK(hipGetLastError());
        return PVX_OK;
    }
    // the chunk boundaries do not fit one workgroup: roots by pointer jumping over all nodes, one launch per round
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, p);
    hipLaunchKernelGGL(k_root_init, dim3(nb), dim3(256), 0, s, p);
    int rounds = 1;
    while ((1LL << rounds) < p.F) rounds++;
    for (int r = 0; r < rounds; r++) hipLaunchKernelGGL(k_root_jump, dim3(nb), dim3(256), 0, s, p);
    hipLaunchKernelGGL(k_assign_ids, dim3(nb), dim3(256), 0, s, p);
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}

int pvx_launch_track_sequential(const TrackParams& p, hipStream_t s) {
    if (p.F <= 0) return PVX_OK;
    const int kp = (p.K + 1) & ~1;
    const size_t lds = (size_t)kp * 8 * 4 + (size_t)kp * 4 * 5;
    if (lds > 64 * 1024) { pvx_set_error("npks=%d too large for the tracker", p.K); return PVX_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(k_track_sequential, dim3(1), dim3(64), lds, s, p);
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}
