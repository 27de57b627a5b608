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
// Bound: f64 VALU (two interpolations, a block prefix sum and one cos per partial sample);
// HBM traffic is 8*h bytes written per segment plus a few hundred bytes of table reads.
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

__device__ inline double eval_piece2(const Piece2& q, double x) {
    return (x < q.b1) ? (q.sa * (x - q.xa) + q.fa) : (q.sb * (x - q.xb) + q.fb);
}

// block-wide inclusive prefix sum of one double per thread (NT = 256 threads); returns the
// inclusive value, *total = sum over the block.  Uses sc[4].
__device__ inline double block_scan(double v, double* sc, double* total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    __syncthreads();
    if (lane == 63) sc[wid] = inc;
    __syncthreads();
    double off = 0.0, tot = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
        if (w < wid) off += sc[w];
        tot += sc[w];
    }
    *total = tot;
    return inc + off;
}

__global__ __launch_bounds__(NT) void k_synth_ola(SynthParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* acc = (double*)smem;                       // [h] output accumulators
    __shared__ Win wins[NBATCH];
    __shared__ double sc[NT / 64];
    __shared__ int wslots[NBATCH][WMAX];
    __shared__ int cb_pid[NBATCH], cb_st[NBATCH], cb_nfr[NBATCH], cb_ii[NBATCH], cb_kind[NBATCH], cb_j0[NBATCH], cb_wn[NBATCH];
    __shared__ int wcnt[NT / 64];
    __shared__ int qnext;

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
        // ---- step 5: the contributions of the batch, in candidate order
        for (int bb = 0; bb < nb; ++bb) {
            const int st = cb_st[bb], nfr = cb_nfr[bb], ii = cb_ii[bb], kind = cb_kind[bb], j0 = cb_j0[bb];
            const Win& win = wins[bb];
            const double* pf = win.f;
            const double* pm = win.m;
            const double* pr = win.r;
            const double offf = dfr + .5, offm = dfr;                     // PVAnalysis.py:701-702

            if (kind == 1) {
                // ---- attack, PVAnalysis.py:742-745: output index st*h - edgsam + j, j < edgsam
                const int64_t o0 = (int64_t)st * h - edgsam;
                const double m0 = interp_w(0.0, dh, offm, nfr, pm, j0);   // msig[0]
                const double c = pf[0 - j0] * 1.0 / p.sr;
                for (int m = tid; m < h; m += NT) {
                    const int64_t j = seg * (int64_t)h + m - o0;
                    if (j >= 0 && j < edgsam) {
                        const double a = m0 * (1 - cos(kPi * (double)j / (double)edgsam)) / 2.;
                        // flipud(realph[0] - 2 pi cumsum(f0/sr)): element j uses the (edgsam-j)-term sum
                        const double phb = pr[0 - j0] - kPi2 * ((double)(edgsam - j) * c);
                        acc[m] += a * cos(phb);
                    }
                }
                continue;
            }

            // ---- body segment ii (kind 0) or the final phase of the last segment (kind 2)
            // PVAnalysis.py:705-708: ph[m] = 2 pi * sum_{q<m} fsig[h*ii + q] / sr, m = 0..h-1
            const double nbase = dh * (double)ii;
            double carry = 0.0;
            // phase corrections, PVAnalysis.py:711-718
            const double fs0 = interp_w(nbase, dh, offf, nfr, pf, j0);
            const double fs1 = interp_w(nbase + dh, dh, offf, nfr, pf, j0);
            const double phcor = kPi * (fs1 - fs0) / fstep / 2.;
            const double ph0 = pr[ii - j0] + phcor;                       // PVAnalysis.py:721
            // fsig over samples nbase .. nbase+h-2 and msig over nbase .. nbase+h-1 (two pieces each)
            const Piece2 qf = make_piece2(nbase, dh, offf, nfr, pf, j0);
            const Piece2 qm = make_piece2(nbase, dh, offm, nfr, pm, j0);
            double lastph = 0.0;       // ph[h-1] + ph0 (before the discontinuity ramp)
            // Prefix sums of the per-sample phase increments.  The ramp needs ph[h-1] (phend) before
            // any sample can be finalised; for h <= CH*NT the prefix of every sample stays in
            // registers (one scan sweep), longer hops scan twice.
            constexpr int CH = 4;
            const int nch = (h + NT - 1) / NT;
            const bool one_sweep = nch <= CH;
            double php[CH];
            {
                double run = 0.0;
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    php[c] = 0.0;
                    if (c < nch) {
                        const int m = c * NT + tid;
                        double term = 0.0;
                        if (m >= 1 && m < h) term = eval_piece2(qf, nbase + (double)(m - 1)) / p.sr;
                        double tot;
                        const double inc = block_scan(term, sc, &tot);
                        php[c] = run + inc;
                        run += tot;
                    }
                }
                for (int c0 = CH * NT; c0 < h; c0 += NT) {            // only when !one_sweep
                    const int m = c0 + tid;
                    double term = 0.0;
                    if (m >= 1 && m < h) term = eval_piece2(qf, nbase + (double)(m - 1)) / p.sr;
                    double tot;
                    (void)block_scan(term, sc, &tot);
                    run += tot;
                }
                lastph = kPi2 * run + ph0;
            }
            if (kind == 2) {
                // ---- release, PVAnalysis.py:748-751: output index (st+nfr)*h + j, j < edgsam
                const int64_t o0 = ((int64_t)st + nfr) * h;
                const double mend = interp_w(dh * (double)nfr, dh, offm, nfr, pm, j0);   // msig[hop*(ii+1)]
                const double c = pf[nfr - 1 - j0] * 1.0 / p.sr;
                for (int m = tid; m < h; m += NT) {
                    const int64_t j = seg * (int64_t)h + m - o0;
                    if (j >= 0 && j < edgsam) {
                        const double a = mend * (1 + cos(kPi * (double)j / (double)edgsam)) / 2.;
                        acc[m] += a * cos(lastph + kPi2 * ((double)(j + 1) * c));
                    }
                }
                continue;
            }
            // kind 0: discontinuity ramp towards the next point, PVAnalysis.py:724-729
            double step = 0.0;
            if (ii < nfr - 1) {
                const double fs2 = interp_w(nbase + 2.0 * dh, dh, offf, nfr, pf, j0);
                const double phcornext = kPi * (fs2 - fs1) / fstep / 2.;
                const double phend = lastph + kPi2 * fs1 / p.sr;
                const double arg = pr[ii + 1 - j0] + phcornext - phend + kPi;
                double md = fmod(arg, kPi2);                              // np.mod: sign of the divisor
                if (md != 0.0 && md < 0.0) md += kPi2;
                const double dph = md - kPi;
                step = dph / dh;                                          // np.linspace(0, dph, h+1)[:-1]
            }
            if (one_sweep) {
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    const int m = c * NT + tid;
                    if (c < nch && m < h) {
                        const double ph_m = kPi2 * php[c] + ph0 + ((double)m * step + 0.0);
                        const double ms = eval_piece2(qm, nbase + (double)m);
                        acc[m] += ms * cos(ph_m);                         // PVAnalysis.py:734-736
                    }
                }
            } else {
                for (int c0 = 0; c0 < h; c0 += NT) {
                    const int m = c0 + tid;
                    double term = 0.0;
                    if (m >= 1 && m < h) term = eval_piece2(qf, nbase + (double)(m - 1)) / p.sr;
                    double tot;
                    const double inc = block_scan(term, sc, &tot);
                    if (m < h) {
                        const double ph_m = kPi2 * (carry + inc) + ph0 + ((double)m * step + 0.0);
                        const double ms = eval_piece2(qm, nbase + (double)m);
                        acc[m] += ms * cos(ph_m);                         // PVAnalysis.py:734-736
                    }
                    carry += tot;
                }
            }
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
