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

#include "pvx_internal.h"

namespace {

constexpr double kPi = 3.141592653589793238462643383279502884;
constexpr double kPi2 = 2.0 * kPi;
constexpr int NT = 256;
constexpr int WMAX = 40;   // window of partial points kept in LDS (needs ceil(dfr + .5) + 6 <= WMAX)
constexpr int NBATCH = 12; // contributions whose windows are gathered together

struct Win {               // a window of one partial's points, j in [j0, j0 + n)
    double f[WMAX], m[WMAX], r[WMAX];
    int j0, n;
};

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
    int kind, fmb, mmb, pad_;
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

__global__ __launch_bounds__(NT) void k_synth_ola(SynthParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* acc = (double*)smem;                       // [h] output accumulators
    __shared__ Win wins[NBATCH];
    __shared__ int wslots[NBATCH][WMAX];
    __shared__ int cb_pid[NBATCH], cb_st[NBATCH], cb_nfr[NBATCH], cb_ii[NBATCH], cb_kind[NBATCH], cb_j0[NBATCH], cb_wn[NBATCH];
    __shared__ int wcnt[NT / 64];
    __shared__ int qnext;
    __shared__ CParam prm[NBATCH];

    const int h = p.hop_s, K = p.K, tid = threadIdx.x;
    const int64_t seg = blockIdx.x;                    // output samples [seg*h, seg*h + h)
    const double dh = (double)h;
    const double overlap = p.hop_a / (double)p.nfft;   // PVAnalysis.py:824
    const double fstep = p.sr / (double)p.nfft;        // PVAnalysis.py:825
    const double dfr = 1. / overlap / 2.;              // PVAnalysis.py:687
    const int edgsam = (int)(dfr * h * p.edge);        // PVAnalysis.py:740
    const double dfr_s = (double)p.nfft / (double)p.hop_a / 2.;      // PVAnalysis.py:1055
    const int64_t edgsamp = (int64_t)(p.edge * h * dfr_s);           // PVAnalysis.py:1056 (integer, Python 2)
    const int EF = edgsam > 0 ? (edgsam + h - 1) / h : 0;            // frames an edge can reach
    const int WB = (int)ceil(dfr + 0.5) + 2;                         // points needed behind the node

    for (int m = tid; m < h; m += NT) acc[m] = 0.0;

    // contributions: kind 0 = body of a peak of frame seg; 1 = attack of a partial starting at
    // frame seg+1 .. seg+EF; 2 = release of a partial whose last frame is seg-EF .. seg-1.
    // Candidates q = (frame - fr_lo) * K + slot are examined 256 at a time; the valid ones are taken
    // in order, NBATCH per round, and the windows of partial points of a whole batch are gathered
    // together: four global round trips per batch instead of four per contribution (this kernel is
    // bound by those dependent loads, not by arithmetic).
    const int64_t fr_lo = seg - EF;
    const int NC = (2 * EF + 1) * K;
    const int WL = WB + 4;                                            // window length (<= WMAX, host-checked)
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
        if (valid && pos < NBATCH) {
            cb_pid[pos] = c_pid_; cb_st[pos] = c_st_; cb_nfr[pos] = c_nfr_; cb_ii[pos] = c_ii_; cb_kind[pos] = c_kind_;
            const int j0 = (c_ii_ - WB > 0) ? c_ii_ - WB : 0;
            int j1 = c_ii_ + 3;
            if (j1 > c_nfr_ - 1) j1 = c_nfr_ - 1;
            cb_j0[pos] = j0; cb_wn[pos] = j1 - j0 + 1;                // <= WB + 4 <= WMAX (checked on the host)
            if (pos == NBATCH - 1) qnext = q + 1;                     // the rest is re-examined next round
        }
        __syncthreads();
        const int nb = total < NBATCH ? total : NBATCH;
        // ---- step 3: slots of the window points of the whole batch
        for (int idx = tid; idx < nb * WL * K; idx += NT) {
            const int bb = idx / (WL * K), rem = idx - bb * (WL * K);
            const int d = rem / K, s2 = rem - d * K;
            if (d < cb_wn[bb]) {
                const int64_t f2 = (int64_t)cb_st[bb] + cb_j0[bb] + d;
                if (p.partial_id[f2 * K + s2] == cb_pid[bb]) wslots[bb][d] = s2;
            }
        }
        __syncthreads();
        // ---- step 4: their values
        for (int idx = tid; idx < nb * WL; idx += NT) {
            const int bb = idx / WL, d = idx - bb * WL;
            if (d < cb_wn[bb]) {
                const int64_t node = ((int64_t)cb_st[bb] + cb_j0[bb] + d) * K + wslots[bb][d];
                wins[bb].f[d] = p.f[node];
                wins[bb].m[d] = p.mag[node];
                wins[bb].r[d] = p.realph[node];
            }
        }
        __syncthreads();
        const int qn = qnext;
        // ---- step 5: thread b derives the closed-form parameters of contribution b.  fsig and msig
        // are piecewise linear (two pieces per hop), so the phase prefix sum of PVAnalysis.py:705-708 is
        // a quadratic in the sample index: no scan, no barrier, every sample independent.
        if (tid < nb) {
            const int bb = tid;
            const int st = cb_st[bb], nfr = cb_nfr[bb], ii = cb_ii[bb], kind = cb_kind[bb], j0 = cb_j0[bb];
            const double* pf = wins[bb].f;
            const double* pm = wins[bb].m;
            const double* pr = wins[bb].r;
            const double offf = dfr + .5, offm = dfr;                     // PVAnalysis.py:701-702
            CParam c;
            c.kind = kind;
            if (kind == 1) {
                // attack, PVAnalysis.py:742-745: output index st*h - edgsam + j, j < edgsam
                c.o0 = (long long)st * h - edgsam;
                c.amp = interp_w(0.0, dh, offm, nfr, pm, j0);             // msig[0]
                c.cfr = pf[0 - j0] * 1.0 / p.sr;
                c.ph0 = pr[0 - j0];
            } else {
                const double nbase = dh * (double)ii;
                // phase corrections, PVAnalysis.py:711-718
                const double fs0 = interp_w(nbase, dh, offf, nfr, pf, j0);
                const double fs1 = interp_w(nbase + dh, dh, offf, nfr, pf, j0);
                const double phcor = p.no_phcor ? 0.0 : kPi * (fs1 - fs0) / fstep / 2.;          // PVAnalysis.py:710-715
                c.ph0 = pr[ii - j0] + phcor;                              // PVAnalysis.py:721
                // fsig(nbase + q) = fa0 + fsa q for q < fmb, fb0 + fsb q beyond; msig likewise
                const Piece2 qf = make_piece2(nbase, dh, offf, nfr, pf, j0);
                const Piece2 qm = make_piece2(nbase, dh, offm, nfr, pm, j0);
                c.fa0 = qf.sa * (nbase - qf.xa) + qf.fa; c.fsa = qf.sa;
                c.fb0 = qf.sb * (nbase - qf.xb) + qf.fb; c.fsb = qf.sb;
                c.ma0 = qm.sa * (nbase - qm.xa) + qm.fa; c.msa = qm.sa;
                c.mb0 = qm.sb * (nbase - qm.xb) + qm.fb; c.msb = qm.sb;
                double d = ceil(qf.b1 - nbase);
                c.fmb = d < 0.0 ? 0 : (d > dh ? h : (int)d);
                d = ceil(qm.b1 - nbase);
                c.mmb = d < 0.0 ? 0 : (d > dh ? h : (int)d);
                const double tm = 0.5 * (double)c.fmb * (double)(c.fmb - 1);
                c.smb = c.fa0 * (double)c.fmb + c.fsa * tm;               // sum of the first fmb terms
                c.tmb = tm;
                // ph[h-1] + ph0 (before the discontinuity ramp): prefix over q = 0 .. h-2
                const double lastph = kPi2 * (prefix_sum(c, h - 1) / p.sr) + c.ph0;
                c.step = 0.0;
                if (kind == 2) {
                    // release, PVAnalysis.py:748-751: output index (st+nfr)*h + j, j < edgsam
                    c.o0 = ((long long)st + nfr) * h;
                    c.amp = interp_w(dh * (double)nfr, dh, offm, nfr, pm, j0);        // msig[hop*(ii+1)]
                    c.cfr = pf[nfr - 1 - j0] * 1.0 / p.sr;
                    c.ph0 = lastph;
                } else if (ii < nfr - 1) {
                    // discontinuity ramp towards the next point, PVAnalysis.py:724-729
                    const double fs2 = interp_w(nbase + 2.0 * dh, dh, offf, nfr, pf, j0);
                    const double phcornext = p.no_phcor ? 0.0 : kPi * (fs2 - fs1) / fstep / 2.;
                    const double phend = lastph + kPi2 * fs1 / p.sr;
                    const double arg = pr[ii + 1 - j0] + phcornext - phend + kPi;
                    double md = fmod(arg, kPi2);                          // np.mod: sign of the divisor
                    if (md != 0.0 && md < 0.0) md += kPi2;
                    c.step = (md - kPi) / dh;                             // np.linspace(0, dph, h+1)[:-1]
                }
            }
            prm[bb] = c;
        }
        __syncthreads();
        // ---- step 6: every thread adds the batch's contributions to its own samples, in candidate order
        for (int m = tid; m < h; m += NT) {
            double a_ = acc[m];
            const long long osamp = seg * (long long)h + m;
            for (int bb = 0; bb < nb; ++bb) {
                const CParam& c = prm[bb];
                if (c.kind == 0) {
                    const double ph_m = kPi2 * (prefix_sum(c, m) / p.sr) + c.ph0 + ((double)m * c.step + 0.0);
                    const double ms = (m < c.mmb) ? (c.ma0 + c.msa * (double)m) : (c.mb0 + c.msb * (double)m);
                    a_ += ms * cos(ph_m);                                 // PVAnalysis.py:734-736
                } else {
                    const long long j = osamp - c.o0;
                    if (j >= 0 && j < edgsam) {
                        const double cw = cos(kPi * (double)j / (double)edgsam);
                        if (c.kind == 1) {
                            // flipud(realph[0] - 2 pi cumsum(f0/sr)): element j uses the (edgsam-j)-term sum
                            a_ += (c.amp * (1 - cw) / 2.) * cos(c.ph0 - kPi2 * ((double)(edgsam - j) * c.cfr));
                        } else {
                            a_ += (c.amp * (1 + cw) / 2.) * cos(c.ph0 + kPi2 * ((double)(j + 1) * c.cfr));
                        }
                    }
                }
            }
            acc[m] = a_;
        }
        qbase = qn;
    }
    __syncthreads();
    for (int m = tid; m < h; m += NT) {
        const int64_t o = seg * (int64_t)h + m;
        if (o < p.wlen) p.w[o] = acc[m];
    }
}

}  // namespace

int pvx_launch_synth(const SynthParams& p, hipStream_t s) {
    if (p.wlen <= 0) return PVX_OK;
    const int h = p.hop_s;
    const double dfr = 1. / (p.hop_a / (double)p.nfft) / 2.;
    if ((int)ceil(dfr + 0.5) + 6 > WMAX) {
        pvx_set_error("nfft/hop = %g is too large for the resynthesis window (dfr=%g)", (double)p.nfft / p.hop_a, dfr);
        return PVX_ERR_UNSUPPORTED;
    }
    const size_t lds = (size_t)h * sizeof(double);
    if (lds > 128 * 1024) { pvx_set_error("synthesis hop %d too large (LDS)", h); return PVX_ERR_UNSUPPORTED; }
    const int64_t nseg = (p.wlen + h - 1) / h;
    if (nseg > 0x7fffffffLL) { pvx_set_error("too many output segments"); return PVX_ERR_INVALID; }
    if (lds > 48 * 1024)
        PVX_HIP_CHECK(hipFuncSetAttribute((const void*)k_synth_ola, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_synth_ola, dim3((unsigned)nseg), dim3(NT), lds, s, p);
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}
