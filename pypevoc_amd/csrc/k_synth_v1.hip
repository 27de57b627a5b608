// k_synth.hip -- overlap-add resynthesis.  Replaces SinSum.synth (pypevoc/PVAnalysis.py:1053-1070)
// and RegPartial.synth (PVAnalysis.py:684-756) for every partial at once.
//
// The reference synthesises partial after partial and adds each one into the output
// (PVAnalysis.py:1060-1069).  Every hop-long body segment of a partial depends only on a few
// neighbouring points of that partial, and partial body segment ii lands exactly on output samples
// [(start+ii)*h, (start+ii+1)*h).  So the sum is restated as a GATHER: one workgroup owns one output
// segment of h samples and adds up
//   - the body segments of the (<= K) peaks of analysis frame `seg`,
//   - the raised-cosine attacks of partials that start within the next ceil(E/h) frames,
//   - the releases of partials that ended within the previous ceil(E/h) frames,
// in LDS accumulators, then stores the segment once (coalesced).  No atomics; the order of the
// additions is fixed (frame, then slot), so the output is reproducible run to run.
//
// fsig and msig (np.interp of the partial's f / mag, PVAnalysis.py:701-702) are piecewise linear with
// breakpoints one hop apart, so inside a segment each is two linear pieces and the cumulative phase
// (PVAnalysis.py:705-708) is a quadratic in the sample index: every sample is evaluated independently
// (no scan, no barrier), the per-contribution constants are derived once by one thread each.
// Bound: f64 VALU (one cos per partial sample); HBM traffic is 8*h bytes written per segment plus
// a few hundred bytes of table reads.
// All arithmetic is float64 (phase arguments reach 1e3..1e4 rad).
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "pvx_internal.h"

namespace {

constexpr double kPi = 3.141592653589793238462643383279502884;
constexpr double kPi2 = 2.0 * kPi;
constexpr int WMAX = 40;   // longest window of partial points (needs ceil(dfr + .5) + 6 <= WMAX)
constexpr int NBMAX = 128; // contributions whose windows are gathered together (fewer if LDS is short)

// cos for the phase arguments of this kernel (1e3 .. 1e5 rad, sometimes far more): x = n pi + r by a two-term
// Cody-Waite reduction with fused multiply-adds (n * pi is exact inside the fma, so r carries < 1 ulp of error for
// |n| < 2^30), then cos r = 1 - 2 sin^2(r/2) with the fdlibm sine polynomial on |r/2| <= pi/4.  Absolute error
// < 4e-16 in a fifth of the instructions of the library routine, which stays for arguments beyond the reduction's
// range (kept out of line: its Payne-Hanek tables would otherwise set the kernel's register count).
__device__ __attribute__((noinline)) double cos_far(double x) { return cos(x); }
constexpr double kNear = 5.0e8;       // |x| below this: the two-term reduction holds (n < 2^28)
__device__ __forceinline__ double fma3(double a, double b, double c) {
    // a*b + c with c left in place: the compiler's two-address v_fmac form copies every polynomial constant first
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <bool FAR> __device__ __forceinline__ double fcos(double x) {
    if constexpr (FAR) return cos_far(x);
    const double n = rint(x * 3.18309886183790691216e-01);                 // 1/pi
    double r = __builtin_fma(-n, 3.14159265358979311600e+00, x);
    r = __builtin_fma(-n, 1.22464679914735320717e-16, r);
    const double hr = 0.5 * r, z = hr * hr;
    double ps = fma3(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma3(z, ps, 2.75573137070700676789e-06);
    ps = fma3(z, ps, -1.98412698298579493134e-04);
    ps = fma3(z, ps, 8.33333333332248946124e-03);
    ps = fma3(z, ps, -1.66666666666666324348e-01);
    const double sn = __builtin_fma(hr * z, ps, hr);                       // sin(r/2)
    const double v = __builtin_fma(-2.0 * sn, sn, 1.0);
    return ((int)n & 1) ? -v : v;
}

// np.interp(x, xp, fp) with xp[j] = h * (off + j), j < nfr (PVAnalysis.py:701-702); fp through the window
__device__ inline double interp_w(double x, double h, double off, int nfr, const double* fp, int j0) {
    if (nfr == 1) return fp[0 - j0];
    const double xlast = h * (off + (double)(nfr - 1));
    const double xfirst = h * (off + 0.0);
    if (x > xlast) return fp[nfr - 1 - j0];
    if (x < xfirst) return fp[0 - j0];
    int j = (int)floor(x / h - off);
    if (j < 0) j = 0;
    if (j > nfr - 1) j = nfr - 1;
    // settle on xp[j] <= x < xp[j+1] with the same xp values numpy compares against
    while (j > 0 && x < h * (off + (double)j)) j--;
    while (j < nfr - 1 && x >= h * (off + (double)(j + 1))) j++;
    if (j == nfr - 1) return fp[j - j0];
    const double xj = h * (off + (double)j);
    if (xj == x) return fp[j - j0];
    const double xj1 = h * (off + (double)(j + 1));
    const double slope = (fp[j + 1 - j0] - fp[j - j0]) / (xj1 - xj);
    return slope * (x - xj) + fp[j - j0];
}

// np.interp restricted to a run of fewer than h consecutive sample positions x0, x0+1, ...: the
// breakpoints xp[j] = h*(off+j) are h apart, so the run meets at most one of them and two linear
// pieces (found once per contribution, with one division each) cover it.  Same slope and same
// evaluation formula as interp_w / numpy, hence the same values.
struct Piece2 { double b1, xa, fa, sa, xb, fb, sb; };

__device__ inline void piece_of(int j, double h, double off, int nfr, const double* fp, int j0, double& xj, double& fj, double& sj) {
    if (j < 0) { xj = 0.0; fj = fp[0 - j0]; sj = 0.0; return; }                     // left of xp[0]: fp[0]
    if (j >= nfr - 1) { xj = 0.0; fj = fp[nfr - 1 - j0]; sj = 0.0; return; }        // at / right of xp[last]
    xj = h * (off + (double)j);
    const double xj1 = h * (off + (double)(j + 1));
    fj = fp[j - j0];
    sj = (fp[j + 1 - j0] - fp[j - j0]) / (xj1 - xj);
}

__device__ inline Piece2 make_piece2(double x0, double h, double off, int nfr, const double* fp, int j0) {
    Piece2 q;
    int j;
    if (nfr == 1 || x0 < h * (off + 0.0)) j = -1;
    else {
        j = (int)floor(x0 / h - off);
        if (j < 0) j = 0;
        if (j > nfr - 1) j = nfr - 1;
        while (j > 0 && x0 < h * (off + (double)j)) j--;
        while (j < nfr - 1 && x0 >= h * (off + (double)(j + 1))) j++;
    }
    if (nfr == 1) j = nfr;                                       // constant everywhere
    piece_of(j, h, off, nfr, fp, j0, q.xa, q.fa, q.sa);
    piece_of(j + 1, h, off, nfr, fp, j0, q.xb, q.fb, q.sb);
    q.b1 = (j + 1 <= nfr - 1 && nfr > 1) ? h * (off + (double)(j + 1)) : INFINITY;
    return q;
}

// closed-form parameters of one contribution (see step 5 of the kernel)
struct CParam {
    int kind, fmb, mmb, far;                 // far: some phase argument may leave the fast cosine's range
    long long o0;
    double ph0, step, amp, cfr;
    double fa0, fsa, fb0, fsb, smb, tmb;     // fsig pieces and the sum / triangular number at the break
    double ma0, msa, mb0, msb;               // msig pieces
};

// sum_{q=0}^{m-1} fsig(nbase + q) for the two-piece linear fsig of a contribution
__device__ inline double prefix_sum(const CParam& c, int m) {
    const double tm = 0.5 * (double)m * (double)(m - 1);
    if (m <= c.fmb) return c.fa0 * (double)m + c.fsa * tm;
    return c.smb + c.fb0 * (double)(m - c.fmb) + c.fsb * (tm - c.tmb);
}
// tools/ubench/synth_phases.hip builds this file with PVX_SYNTH_STAMPS: s_memtime stamps of one workgroup in slot_of[]
#ifdef PVX_SYNTH_STAMPS
#define PVX_STAMP(slot) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) ((long long*)p.slot_of)[slot] = (long long)clock64(); } while (0)
#else
#define PVX_STAMP(slot) do { } while (0)
#endif

constexpr int HTMAX = 512; // pid -> contribution hash table of a batch: the power of two >= 4 * batch (short probe chains), 2^c_htbits entries
constexpr int TSMAX = 256; // threads across the samples of a segment (fewer in a smaller workgroup); NT / TS groups share the contributions

// LDS (dynamic): acc [G][h] | wf wm wr [NB][WL] doubles | prm [NB] | wslots [NB][WL] | cb_* 7 x [NB] | hkey hval [2^htbits] ints
__host__ __device__ inline int synth_htbits(int nb) {
    int b = 4;
    while ((1 << b) < 4 * nb && (1 << b) < HTMAX) b++;
    return b;
}
__host__ __device__ inline size_t synth_lds_bytes(int h, int nb, int wl, int groups) {
    const int HT = 1 << synth_htbits(nb);
    return (size_t)h * 8 * groups + (size_t)nb * wl * 8 * 3 + (size_t)nb * sizeof(CParam) + (size_t)nb * wl * 4 + (size_t)nb * 4 * 7 + (size_t)HT * 8;
}

// Waves per SIMD the register allocation aims at: the kernel waits on its set-up loads (one wave of a workgroup busy
// for most of its time), so occupancy is worth more than registers -- five waves (96 registers) in the two-wave
// workgroups of a long waveform, four in the larger ones (measured on config 2: 2 waves 1.35 ms, 4: 0.82, 5 with
// two-wave workgroups and batches of 16: 0.54, 6: 0.55, 8: 0.60 -- spilling by then).
#ifndef PVX_SYNTH_WAVES
#define PVX_SYNTH_WAVES(NT) ((NT) <= 128 ? 5 : 4)
#endif
template <int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(PVX_SYNTH_WAVES(NT), PVX_SYNTH_WAVES(NT)))) void k_synth_ola(SynthParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int wcnt[NT / 64];
    __shared__ int qnext;

#ifndef PVX_SYNTH_SP
#define PVX_SYNTH_SP 1
#endif
    constexpr int SP = PVX_SYNTH_SP; // samples per thread held in registers while a batch is added
    constexpr int TS = NT < TSMAX ? NT : TSMAX;
    constexpr int G = NT / TS;       // groups: group g adds contributions g, g + G, ... to its own accumulators
    const int h = p.hop_s, K = p.K, tid = threadIdx.x;
    const int NB = p.nbatch;
    const int64_t seg = (int64_t)blockIdx.x + p.seg0;  // output samples [seg*h, seg*h + h)
    const double dh = p.c_dh;                          // (double)h
    const double fstep = p.c_fstep;                    // sr / nfft, PVAnalysis.py:825
    const double dfr = p.c_dfr;                        // 1 / (hop_a / nfft) / 2, PVAnalysis.py:687, 824
    const int edgsam = p.c_edgsam;                     // (int)(dfr * h * edge), PVAnalysis.py:740
    const int64_t edgsamp = p.c_edgsamp;               // (int)(edge * h * (nfft / hop_a / 2.)), PVAnalysis.py:1055-1056 (integer, Python 2)
    const int EF = p.c_EF;                             // frames an edge can reach
    const int WB = p.c_WB;                             // ceil(dfr + .5) + 2: points needed behind the node
    const int WL = p.c_WL;                             // window length WB + 4 (<= WMAX, host-checked)

    double* acc = (double*)smem;                       // [G][h] accumulators; row 0 is the output
    double* wf = acc + (size_t)G * h;                  // windows of partial points, j in [j0, j0 + wn)
    double* wm = wf + (size_t)NB * WL;
    double* wr = wm + (size_t)NB * WL;
    CParam* prm = (CParam*)(wr + (size_t)NB * WL);
    int* wslots = (int*)(prm + NB);
    int* cb_pid = wslots + (size_t)NB * WL;
    int* cb_st = cb_pid + NB;
    int* cb_nfr = cb_st + NB;
    int* cb_ii = cb_nfr + NB;
    int* cb_kind = cb_ii + NB;
    int* cb_j0 = cb_kind + NB;
    int* cb_wn = cb_j0 + NB;
    int* hkey = cb_wn + NB;
    const int htbits = p.c_htbits, HT = 1 << htbits;
    int* hval = hkey + HT;

    PVX_STAMP(0);
    for (int m = tid; m < G * h; m += NT) acc[m] = 0.0;

    // contributions: kind 0 = body of a peak of frame seg; 1 = attack of a partial starting at
    // frame seg+1 .. seg+EF; 2 = release of a partial whose last frame is seg-EF .. seg-1.
    // Candidates q = (frame - fr_lo) * K + slot are examined NT at a time; the valid ones are taken
    // in order, NB per round, and the windows of partial points of a whole batch are gathered
    // together: four global round trips per batch instead of four per contribution (on a short signal
    // this kernel is bound by those dependent loads, so a batch is as large as LDS allows).
    const int64_t fr_lo = seg - EF;
    const int NC = (2 * EF + 1) * K;
    for (int qbase = 0; qbase < NC;) {
        // ---- round step 1: examine candidates qbase + [0, NT)
        __syncthreads();
        bool valid = false;
        int c_pid_ = 0, c_st_ = 0, c_nfr_ = 0, c_ii_ = 0, c_kind_ = 0;
        const int q = qbase + tid;
        if (q < NC) {
            const int64_t fr = fr_lo + q / K;
            const int sl = q % K;
            if (fr >= 0 && fr < p.F) {
                const int pid = p.partial_id[fr * K + sl];
                if (pid >= 0) {
                    const int st = p.part_start[pid], nfr = p.part_len[pid];
                    const int kind = (fr == seg) ? 0 : (fr > seg ? 1 : 2);
                    const int ii = (int)(fr - st);                    // index of this point in its partial
                    valid = !(nfr < p.minframes || nfr < 1)           // PVAnalysis.py:1061
                            && !((int64_t)st * h - edgsam + edgsamp < 0)   // PVAnalysis.py:1067
                            && !(kind == 1 && ii != 0) && !(kind == 2 && ii != nfr - 1);
                    c_pid_ = pid; c_st_ = st; c_nfr_ = nfr; c_ii_ = ii; c_kind_ = kind;
                }
            }
        }
        // ---- step 2: ordered compaction of the valid candidates
        const unsigned long long bal = __ballot(valid);
        const int lane_ = tid & 63, wid_ = tid >> 6;
        if (lane_ == 0) wcnt[wid_] = __popcll(bal);
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; w++) { if (w < wid_) woff += wcnt[w]; total += wcnt[w]; }
        const int pos = woff + __popcll(bal & ((1ull << lane_) - 1ull));
        if (tid == 0) qnext = qbase + NT;
        __syncthreads();
        if (valid && pos < NB) {
            cb_pid[pos] = c_pid_; cb_st[pos] = c_st_; cb_nfr[pos] = c_nfr_; cb_ii[pos] = c_ii_; cb_kind[pos] = c_kind_;
            const int j0 = (c_ii_ - WB > 0) ? c_ii_ - WB : 0;
            int j1 = c_ii_ + 3;
            if (j1 > c_nfr_ - 1) j1 = c_nfr_ - 1;
            cb_j0[pos] = j0; cb_wn[pos] = j1 - j0 + 1;                // <= WB + 4 = WL
            if (pos == NB - 1) qnext = q + 1;                         // the rest is re-examined next round
        }
        __syncthreads();
        const int nb = total < NB ? total : NB;
        PVX_STAMP(1);
        // ---- step 3: slots of the window points of the whole batch.  A partial sits in one slot per frame, somewhere
        // among K: instead of searching K slots per window point, the rows of partial_id the windows can reach
        // (frames seg-EF-WB .. seg+EF+3) are read once and every entry looks its partial up in a hash table of the batch
        for (int i = tid; i < HT; i += NT) hkey[i] = -1;
        __syncthreads();
        for (int bb = tid; bb < nb; bb += NT) {
            const int pid = cb_pid[bb];                               // unique within a segment's candidates
            unsigned hh = ((unsigned)pid * 2654435761u) >> (32 - htbits);
            while (atomicCAS(&hkey[hh], -1, pid) != -1) hh = (hh + 1) & (HT - 1);
            hval[hh] = bb;
        }
        __syncthreads();
        {
            const int64_t fr_min = seg - EF - WB;
            const int nrows = 2 * EF + WB + 4;
            for (int idx = tid; idx < nrows * K; idx += NT) {
                const int row = idx / K, s2 = idx - row * K;
                const int64_t f2 = fr_min + row;
                if (f2 < 0 || f2 >= p.F) continue;
                const int pid = p.partial_id[f2 * K + s2];
                if (pid < 0) continue;
                unsigned hh = ((unsigned)pid * 2654435761u) >> (32 - htbits);
                int k;
                while ((k = hkey[hh]) != -1 && k != pid) hh = (hh + 1) & (HT - 1);
                if (k == pid) {
                    const int bb = hval[hh];
                    const int64_t d = f2 - ((int64_t)cb_st[bb] + cb_j0[bb]);
                    if (d >= 0 && d < cb_wn[bb]) wslots[bb * WL + (int)d] = s2;
                }
            }
        }
        __syncthreads();
        PVX_STAMP(2);
        // ---- step 4: their values
        for (int idx = tid; idx < nb * WL; idx += NT) {
            const int bb = idx / WL, d = idx - bb * WL;
            if (d < cb_wn[bb]) {
                const int64_t node = ((int64_t)cb_st[bb] + cb_j0[bb] + d) * K + wslots[idx];
                wf[idx] = p.f[node];
                wm[idx] = p.mag[node];
                wr[idx] = p.realph[node];
            }
        }
        __syncthreads();
        PVX_STAMP(3);
        const int qn = qnext;
        // ---- step 5: the closed-form parameters of the contributions.  fsig and msig are piecewise linear (two pieces per
        // hop), so the phase prefix sum of PVAnalysis.py:705-708 is a quadratic in the sample index: no scan, no barrier,
        // every sample independent.  5a: the interpolations a contribution needs (np.interp at three or four positions,
        // the two pieces of fsig and of msig) are independent of each other -- eight lanes per contribution take one
        // each (same function, different arguments) and leave their results in the contribution's record; 5b: thread b
        // combines them.  (One thread per contribution doing all of it in turn was 30 % of a workgroup's time.)
        for (int t = tid; t < nb * 8; t += NT) {
            const int bb = t >> 3, k = t & 7;
            const int nfr = cb_nfr[bb], ii = cb_ii[bb], kind = cb_kind[bb], j0 = cb_j0[bb];
            const double* pf = wf + (size_t)bb * WL;
            const double* pm = wm + (size_t)bb * WL;
            const double offf = p.c_offf, offm = dfr;                     // dfr + .5, dfr: PVAnalysis.py:701-702
            const double nbase = dh * (double)ii;
            CParam* c = prm + bb;
            if (k < 4) {
                // k = 0, 1, 2: fsig at nbase, nbase + h, nbase + 2 h (PVAnalysis.py:711-718, 724-729) -> smb, tmb, step for now;
                // k = 3: the edge's amplitude msig[0] (attack) / msig[hop * nfr] (release) -> amp
                const bool need = (k == 3) ? (kind != 0) : ((kind != 1) && (k < 2 || (kind == 0 && ii < nfr - 1)));
                if (need) {
                    const double x = (k == 3) ? ((kind == 1) ? 0.0 : dh * (double)nfr) : nbase + (double)k * dh;
                    const double v = interp_w(x, dh, k == 3 ? offm : offf, nfr, k == 3 ? pm : pf, j0);
                    if (k == 0) c->smb = v; else if (k == 1) c->tmb = v; else if (k == 2) c->step = v; else c->amp = v;
                }
            } else if (k < 6 && kind != 1) {
                // fsig(nbase + q) = fa0 + fsa q for q < fmb, fb0 + fsb q beyond; msig likewise
                const bool isf = (k == 4);
                const Piece2 q = make_piece2(nbase, dh, isf ? offf : offm, nfr, isf ? pf : pm, j0);
                const double a0 = q.sa * (nbase - q.xa) + q.fa, b0 = q.sb * (nbase - q.xb) + q.fb;
                const double d = ceil(q.b1 - nbase);
                const int mb = d < 0.0 ? 0 : (d > dh ? h : (int)d);
                if (isf) { c->fa0 = a0; c->fsa = q.sa; c->fb0 = b0; c->fsb = q.sb; c->fmb = mb; }
                else { c->ma0 = a0; c->msa = q.sa; c->mb0 = b0; c->msb = q.sb; c->mmb = mb; }
            }
        }
        __syncthreads();
        for (int bb = tid; bb < nb; bb += NT) {
            const int st = cb_st[bb], nfr = cb_nfr[bb], ii = cb_ii[bb], kind = cb_kind[bb], j0 = cb_j0[bb];
            const double* pf = wf + (size_t)bb * WL;
            const double* pr = wr + (size_t)bb * WL;
            CParam c = prm[bb];
            c.kind = kind;
            if (kind == 1) {
                // attack, PVAnalysis.py:742-745: output index st*h - edgsam + j, j < edgsam; amp = msig[0] (5a)
                c.o0 = (long long)st * h - edgsam;
                c.cfr = pf[0 - j0] * 1.0 / p.sr;
                c.ph0 = pr[0 - j0];
            } else {
                const double fs0 = c.smb, fs1 = c.tmb, fs2 = c.step;      // (5a)
                // phase corrections, PVAnalysis.py:711-718
                const double phcor = p.no_phcor ? 0.0 : kPi * (fs1 - fs0) / fstep / 2.;          // PVAnalysis.py:710-715
                c.ph0 = pr[ii - j0] + phcor;                              // PVAnalysis.py:721
                const double tm = 0.5 * (double)c.fmb * (double)(c.fmb - 1);
                c.smb = c.fa0 * (double)c.fmb + c.fsa * tm;               // sum of the first fmb terms
                c.tmb = tm;
                // ph[h-1] + ph0 (before the discontinuity ramp): prefix over q = 0 .. h-2
                const double lastph = kPi2 * (prefix_sum(c, h - 1) / p.sr) + c.ph0;
                c.step = 0.0;
                if (kind == 2) {
                    // release, PVAnalysis.py:748-751: output index (st+nfr)*h + j, j < edgsam; amp = msig[hop*(ii+1)] (5a)
                    c.o0 = ((long long)st + nfr) * h;
                    c.cfr = pf[nfr - 1 - j0] * 1.0 / p.sr;
                    c.ph0 = lastph;
                } else if (ii < nfr - 1) {
                    // discontinuity ramp towards the next point, PVAnalysis.py:724-729
                    const double phcornext = p.no_phcor ? 0.0 : kPi * (fs2 - fs1) / fstep / 2.;
                    const double phend = lastph + kPi2 * fs1 / p.sr;
                    const double arg = pr[ii + 1 - j0] + phcornext - phend + kPi;
                    double md = fmod(arg, kPi2);                          // np.mod: sign of the divisor
                    if (md != 0.0 && md < 0.0) md += kPi2;
                    c.step = (md - kPi) / dh;                             // np.linspace(0, dph, h+1)[:-1]
                }
            }
            // can a phase argument of this contribution leave the fast cosine's range?  (a bound, not the maximum)
            double bound;
            if (kind == 1 || kind == 2) bound = fabs(c.ph0) + kPi2 * ((double)edgsam + 1.0) * fabs(c.cfr);
            else bound = fabs(c.ph0) + dh * fabs(c.step)
                         + (kPi2 / p.sr) * (fabs(c.smb) + dh * (fabs(c.fa0) + fabs(c.fb0) + dh * (fabs(c.fsa) + fabs(c.fsb))));
            c.far = !(bound < kNear);
            if (kind == 0) {
                // the per-sample form: 2 pi / sr folded into the two-piece phase polynomial (the sample loop is this
                // kernel's bound; the folding moves the phase by a few ulp of ~1e4 rad, 1e-12 of the waveform)
                const double sc = p.c_sc;                                 // 2 pi / sr
                c.fa0 *= sc; c.fsa *= sc; c.fb0 *= sc; c.fsb *= sc; c.smb *= sc;
            }
            prm[bb] = c;
        }
        __syncthreads();
        PVX_STAMP(4);
        // ---- step 6: the samples.  Thread (g, tl) adds contributions g, g + G, ... (in candidate order) to samples
        // tl, tl + TS, ... held in registers: the parameters of a contribution are read once for SP samples
        {
            const int g = tid / TS, tl = tid - g * TS;
            double* accg = acc + (size_t)g * h;
            for (int m0 = tl; m0 < h; m0 += TS * SP) {
                double a_[SP];
#pragma unroll
                for (int u = 0; u < SP; u++) a_[u] = (m0 + u * TS < h) ? accg[m0 + u * TS] : 0.0;
                auto add = [&](const CParam& c, auto far_tag) {
                    constexpr bool FAR = decltype(far_tag)::value;
#pragma unroll
                    for (int u = 0; u < SP; u++) {
                        const int m = m0 + u * TS;
                        if (m >= h) continue;
                        const double dm = (double)m;
                        if (c.kind == 0) {
                            // phase = ph0 + step m + [m <= fmb: fa0 m + fsa T(m) | smb + fb0 (m - fmb) + fsb (T(m) - T(fmb))],
                            // T(m) = m (m - 1) / 2, coefficients pre-scaled by 2 pi / sr; both pieces, then one select
                            const double tm = 0.5 * dm * (double)(m - 1);
                            const double pa = __builtin_fma(c.fsa, tm, c.fa0 * dm);
                            const double pb = __builtin_fma(c.fsb, tm - c.tmb, __builtin_fma(c.fb0, (double)(m - c.fmb), c.smb));
                            const double ph_m = (m <= c.fmb ? pa : pb) + __builtin_fma(c.step, dm, c.ph0);
                            const bool ma = m < c.mmb;
                            const double ms = __builtin_fma(ma ? c.msa : c.msb, dm, ma ? c.ma0 : c.mb0);
                            a_[u] = __builtin_fma(ms, fcos<FAR>(ph_m), a_[u]);    // PVAnalysis.py:734-736
                        } else {
                            const long long j = seg * (long long)h + m - c.o0;
                            if (j >= 0 && j < edgsam) {
                                const double cw = fcos<false>(kPi * (double)j / (double)edgsam);
                                if (c.kind == 1) {
                                    // flipud(realph[0] - 2 pi cumsum(f0/sr)): element j uses the (edgsam-j)-term sum
                                    a_[u] += (c.amp * (1 - cw) / 2.) * fcos<FAR>(c.ph0 - kPi2 * ((double)(edgsam - j) * c.cfr));
                                } else {
                                    a_[u] += (c.amp * (1 + cw) / 2.) * fcos<FAR>(c.ph0 + kPi2 * ((double)(j + 1) * c.cfr));
                                }
                            }
                        }
                    }
                };
                for (int bb = g; bb < nb; bb += G) {
                    const CParam c = prm[bb];
                    if (c.far) add(c, std::true_type{});
                    else add(c, std::false_type{});
                }
#pragma unroll
                for (int u = 0; u < SP; u++) if (m0 + u * TS < h) accg[m0 + u * TS] = a_[u];
            }
        }
        qbase = qn;
        PVX_STAMP(5);
    }
    __syncthreads();
    for (int m = tid; m < h; m += NT) {
        const int64_t o = seg * (int64_t)h + m;
        double v = acc[m];
#pragma unroll
        for (int g = 1; g < G; g++) v += acc[(size_t)g * h + m];          // fixed order: reproducible run to run
        if (o < p.wlen) p.w[o] = v;
    }
    PVX_STAMP(6);
}

}  // namespace

int pvx_launch_synth_v1(const SynthParams& p_in, hipStream_t s) {
    SynthParams p = p_in;
    if (p.wlen <= 0) return PVX_OK;
    const int h = p.hop_s;
    const double dfr = 1. / (p.hop_a / (double)p.nfft) / 2.;
    if ((int)ceil(dfr + 0.5) + 6 > WMAX) {
        pvx_set_error("nfft/hop = %g is too large for the resynthesis window (dfr=%g)", (double)p.nfft / p.hop_a, dfr);
        return PVX_ERR_UNSUPPORTED;
    }
    if ((size_t)h * sizeof(double) > 120 * 1024) { pvx_set_error("synthesis hop %d too large (LDS)", h); return PVX_ERR_UNSUPPORTED; }
    const int64_t nseg = (p.wlen + h - 1) / h;
    if (nseg > 0x7fffffffLL) { pvx_set_error("too many output segments"); return PVX_ERR_INVALID; }
    // batch: every candidate of a segment in one round if LDS allows (two workgroups per CU when the grid is long)
    const int WL = (int)ceil(dfr + 0.5) + 6;
    const int edgsam = (int)(dfr * h * p.edge);
    const int EF = edgsam > 0 ? (edgsam + h - 1) / h : 0;
    const int64_t NC = (int64_t)(2 * EF + 1) * p.K;
    int nb = NC < NBMAX ? (int)NC : NBMAX;
    // a short signal leaves most of the chip idle at 256 threads per segment: more waves share a segment's contributions
    int nt = 256;
    // (512 = two groups: f64 issue is already saturated by two waves per SIMD, and 1024 threads would cap the kernel
    // at 128 registers, which the four-sample loop body does not fit)
    if (nseg < 1024 && NC >= 16) nt = 512;
    // a long waveform: workgroups of two waves -- a segment's set-up (candidates, windows, parameters: dependent loads and
    // serial float64 on a handful of lanes, three quarters of a workgroup's time) keeps ONE wave busy, so smaller
    // workgroups mean more set-ups in flight per CU (7 x 2 waves at config 2's shape: 0.82 -> 0.69 ms)
    else if (nseg >= 2048) nt = 128;
    if (const char* e = getenv("PVX_SYNTH_THREADS")) { const int v = atoi(e); if (v == 64 || v == 128 || v == 256 || v == 512) nt = v; }
    while (nt > 256 && synth_lds_bytes(h, 4, WL, nt / TSMAX) > 150 * 1024) nt >>= 1;
    const int groups = nt > TSMAX ? nt / TSMAX : 1;
    const size_t budget = (nseg > 512 ? 72 : 150) * 1024;
    while (nb > 4 && synth_lds_bytes(h, nb, WL, groups) > budget) nb >>= 1;
    // two-wave workgroups: batches of max(16, 2 npks) -- a segment's live contributions (<= npks bodies and the odd edge; a
    // fuller segment takes another round) instead of all (2 EF + 1) npks candidates: 11 KB of LDS per workgroup instead of
    // 23, ten workgroups per CU
    if (nt == 128) { const int cap = p.K * 2 > 16 ? p.K * 2 : 16; if (nb > cap) nb = cap; }
    if (const char* e = getenv("PVX_SYNTH_NB")) { const int v = atoi(e); if (v >= 4 && v <= NBMAX && v < nb) nb = v; }     // tests: more rounds
    if (nb < 1) nb = 1;
    p.nbatch = nb;
    p.c_htbits = synth_htbits(nb);
    {
        const double overlap = p.hop_a / (double)p.nfft;              // PVAnalysis.py:824
        p.c_dh = (double)h;
        p.c_fstep = p.sr / (double)p.nfft;                            // PVAnalysis.py:825
        p.c_dfr = 1. / overlap / 2.;                                  // PVAnalysis.py:687
        p.c_offf = p.c_dfr + .5;
        p.c_sc = kPi2 / p.sr;
        p.c_edgsam = (int)(p.c_dfr * h * p.edge);                     // PVAnalysis.py:740
        const double dfr_s = (double)p.nfft / (double)p.hop_a / 2.;   // PVAnalysis.py:1055
        p.c_edgsamp = (int64_t)(p.edge * h * dfr_s);                  // PVAnalysis.py:1056
        p.c_EF = p.c_edgsam > 0 ? (p.c_edgsam + h - 1) / h : 0;
        p.c_WB = (int)ceil(p.c_dfr + 0.5) + 2;
        p.c_WL = p.c_WB + 4;
    }
    // a slice of the segments (p.seg_count > 0: pvx_synth_resident launches the waveform in slices whose DMA to the host
    // runs under the next slice's kernel); the geometry above is that of the whole waveform either way
    if (p.seg0 < 0 || p.seg0 > nseg) { pvx_set_error("bad segment slice"); return PVX_ERR_INVALID; }
    const int64_t nlaunch = (p.seg_count > 0 && p.seg0 + p.seg_count < nseg) ? p.seg_count : nseg - p.seg0;
    if (nlaunch <= 0) return PVX_OK;
    const size_t lds = synth_lds_bytes(h, nb, WL, groups);
    if (lds > 158 * 1024) { pvx_set_error("synthesis hop %d too large (LDS)", h); return PVX_ERR_UNSUPPORTED; }
#define PVX_SYNTH(NT_)                                                                                                      \
    do {                                                                                                                    \
        if (lds > 48 * 1024)                                                                                                \
            PVX_HIP_CHECK(hipFuncSetAttribute((const void*)k_synth_ola<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(k_synth_ola<NT_>, dim3((unsigned)nlaunch), dim3(NT_), lds, s, p);                                \
    } while (0)
    if (nt == 512) PVX_SYNTH(512);
    else if (nt == 128) PVX_SYNTH(128);
    else if (nt == 64) PVX_SYNTH(64);
    else PVX_SYNTH(256);
#undef PVX_SYNTH
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}
