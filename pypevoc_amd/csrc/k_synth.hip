// k_synth.hip -- overlap-add resynthesis.  Replaces SinSum.synth (pypevoc/PVAnalysis.py:1053-1070)
// and RegPartial.synth (PVAnalysis.py:684-756) for every partial at once.
//
// The reference synthesises partial after partial and adds each one into the output
// (PVAnalysis.py:1060-1069).  Every hop-long body segment of a partial depends only on a few
// neighbouring points of that partial, and partial body segment ii lands exactly on output samples
// [(start+ii)*h, (start+ii+1)*h).  So the sum is restated as a GATHER over output segments of h samples:
// segment `seg` is the sum of
//   - the body segments of the (<= K) peaks of analysis frame `seg`,
//   - the raised-cosine attacks of partials that start within the next ceil(E/h) frames,
//   - the releases of partials that ended within the previous ceil(E/h) frames.
// No atomics; the order of the additions is fixed (kind, then node index), so the output is reproducible.
//
// The kernels:
//   k_synth_alloc    one thread per partial: where the points of a partial that will sound live in the
//                    partial-major copy of the analysis arrays (wave prefix sum + one atomic per wave)
//   k_synth_scatter  one thread per (frame, slot) node: f / mag / realph of every partial, contiguous
//                    (rows of more than 16 peaks only: k_synth_params_direct reads the analysis rows where they are)
//   k_synth_params[_direct]
//                    one thread per node: the closed form of that node's contribution(s) -- fsig and msig
//                    (np.interp of the partial's f / mag, PVAnalysis.py:701-702) are piecewise linear with
//                    breakpoints one hop apart, so inside a segment each is two linear pieces and the cumulative
//                    phase (PVAnalysis.py:705-708) is a quadratic in the sample index; attack / release are pure
//                    sinusoids under a raised cosine.  One 128-byte record per body, 64 bytes per edge, and one bit
//                    per node and kind (the wave's ballot) saying which records exist.
//   k_synth_bodies<R, S>
//                    one thread per RUN of R consecutive output samples: walks the set bits of its segment,
//                    and for each contribution seeds exp(i phase) and exp(i phase increment) exactly (two sincos)
//                    at the run's first sample, then advances both by complex rotations -- the phase is
//                    quadratic, so its increment is linear and ITS increment constant: z *= w, w += w d, ten
//                    instructions per sample instead of a forty-instruction cosine.  The recurrence is re-seeded
//                    every run, so its error stays below R^2 * 1e-16 (S = double; S = float, plans at precision
//                    32: R^2 / 2 * 6e-8, two bodies' recurrences side by side).  R sums live in registers across all
//                    contributions and are stored once.  A segment's attacks and releases are added here as well.
//   k_synth_extras<R, XB>
//                    bodies whose pieces change inside a run (float rounding of a non-dyadic nfft / hop), and the
//                    attacks / releases as a launch of their own when those exist.
// Bound: VALU issue (sample loop).  HBM: 8*h bytes written per segment, 128 B per body record written and read
// once (cache-resident between the two launches).
// The closed forms and the seeds are float64 (phase arguments reach 1e3..1e4 rad).
#include <math.h>
#include <stdlib.h>

#include <map>
#include <mutex>

#include <type_traits>

#include "pvx_internal.h"

namespace {

constexpr double kPi = 3.141592653589793238462643383279502884;
constexpr double kPi2 = 2.0 * kPi;

__device__ __forceinline__ double fma3(double a, double b, double c) {
    // a*b + c with c left in place: the compiler's two-address v_fmac form copies every polynomial constant first
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// sin and cos of one argument: x = n pi/2 + r by a two-term Cody-Waite reduction with fused multiply-adds -- n * pio2_hi is
// exact inside the fma and x - n pio2_hi = r + n pio2_lo is of order one, so r carries < 1 ulp of error as long as n itself
// is exact (|x| < 2^50: phase arguments of this kernel reach 1e4, with time stretching 1e6) --, the fdlibm kernel
// polynomials on |r| <= pi/4, then the quadrant (n mod 4 taken in float64: n may exceed an int).  Absolute error < 3e-16.
// Beyond 2^50 rad an argument's own ulp exceeds 0.1 rad: its cosine says nothing about the signal; the reduction's
// remainder is then forced into range so that the result stays a bounded number (no library call in these kernels: a call
// makes every live sum of a thread a callee-saved register).
__device__ __forceinline__ void fsincos(double x, double& s, double& c) {
    const double n = rint(x * 6.36619772367581382433e-01);                 // 2/pi
    double r = __builtin_fma(-n, 1.57079632679489655800e+00, x);
    r = __builtin_fma(-n, 6.12323399573676603587e-17, r);
    r = fabs(r) <= 1.0 ? r : 0.0;
    const double z = r * r;
    double ps = fma3(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma3(z, ps, 2.75573137070700676789e-06);
    ps = fma3(z, ps, -1.98412698298579493134e-04);
    ps = fma3(z, ps, 8.33333333332248946124e-03);
    ps = fma3(z, ps, -1.66666666666666324348e-01);
    const double sn = __builtin_fma(r * z, ps, r);
    double pc = fma3(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma3(z, pc, -2.75573143513906633035e-07);
    pc = fma3(z, pc, 2.48015872894767294178e-05);
    pc = fma3(z, pc, -1.38888888888741095749e-03);
    pc = fma3(z, pc, 4.16666666666666019037e-02);
    const double cs = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    const int q = (int)__builtin_fma(-4.0, rint(n * 0.25), n);              // n mod 4 in {-2 .. 2}
    const double a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}
// The float32 form for the precision-32 sample loop (k_synth_bodies<R, float>): the argument and its reduction in float64 as above --
// phases reach 1e4 rad, a float32 argument would carry 1e-3 rad of error --, the polynomials on |r| <= pi/4 in float32 (fdlibm's
// k_sinf / k_cosf coefficients: absolute error < 1e-7).  Eight float64 and sixteen float32 instructions instead of thirty-five float64.
__device__ __forceinline__ void fsincos_f(double x, float& s, float& c) {
    const double n = rint(x * 6.36619772367581382433e-01);                 // 2/pi
    double r = __builtin_fma(-n, 1.57079632679489655800e+00, x);
    r = __builtin_fma(-n, 6.12323399573676603587e-17, r);
    r = fabs(r) <= 1.0 ? r : 0.0;
    const float rf = (float)r, z = rf * rf;
    float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(z, ps, -1.6666654611e-1f);
    const float sn = __builtin_fmaf(rf * z, ps, rf);
    float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(z, pc, 4.166664568298827e-2f);
    const float cs = __builtin_fmaf(z * z, pc, __builtin_fmaf(z, -0.5f, 1.0f));
    const int q = (int)__builtin_fma(-4.0, rint(n * 0.25), n);              // n mod 4 in {-2 .. 2}
    const float a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}
// a / b for a launch-wide b with rb = 1 / b correctly rounded (computed on the host): the quotient estimate, its exact
// residual and one correction -- the correctly rounded quotient (Markstein), i.e. what the reference's division gives, in
// three instructions instead of the ~35 of the general routine (no scaling or special cases needed: b is a hop, a sample
// rate or a bin width, a a frequency or a phase)
__device__ __forceinline__ double div_const(double a, double b, double rb) {
    const double q = a * rb;
    return __builtin_fma(__builtin_fma(-q, b, a), rb, q);
}
// the slope of a piece of np.interp: the breakpoints are h apart (exactly, unless the products h (off + j) round)
__device__ __forceinline__ double slope_of(double num, double den, double h, double rh) {
    return den == h ? div_const(num, h, rh) : num / den;
}

// exp(i x) - 1.  x is the change of a partial's phase increment from one sample to the next -- 2 pi / sr times a frequency
// slope per sample: 1e-3 at the very most -- so the series do it (|x| < 2^-6: next terms x^10 / 10!, x^9 / 9!, below 1e-22);
// anything larger: (-2 sin^2(x/2), 2 sin(x/2) cos(x/2))
__device__ __forceinline__ void expm1i(double x, double& re, double& im) {
    if (fabs(x) < 0.015625) {
        const double z = x * x;
        re = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, 1.0 / 40320.0, -1.0 / 720.0), 1.0 / 24.0), -0.5);
        im = x * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0), 1.0);
    } else {
        double s2, c2;
        fsincos(0.5 * x, s2, c2);
        re = -2.0 * s2 * s2;
        im = 2.0 * s2 * c2;
    }
}

// np.interp(x, xp, fp) with xp[j] = h * (off + j), j < nfr (PVAnalysis.py:701-702); rh = 1 / h.  fp(j): point j of the partial
// (an accessor: the points come from the partial-major copy, or straight from the analysis rows staged in LDS)
template <class A> __device__ inline double interp_w(double x, double h, double rh, double off, int nfr, const A& fp) {
    if (nfr == 1) return fp(0);
    const double xlast = h * (off + (double)(nfr - 1));
    const double xfirst = h * (off + 0.0);
    if (x > xlast) return fp(nfr - 1);
    if (x < xfirst) return fp(0);
    int j = (int)floor(x * rh - off);                          // (a guess: settled below)
    if (j < 0) j = 0;
    if (j > nfr - 1) j = nfr - 1;
    // settle on xp[j] <= x < xp[j+1] with the same xp values numpy compares against
    while (j > 0 && x < h * (off + (double)j)) j--;
    while (j < nfr - 1 && x >= h * (off + (double)(j + 1))) j++;
    if (j == nfr - 1) return fp(j);
    const double xj = h * (off + (double)j);
    if (xj == x) return fp(j);
    const double xj1 = h * (off + (double)(j + 1));
    const double slope = slope_of(fp(j + 1) - fp(j), xj1 - xj, h, rh);
    return slope * (x - xj) + fp(j);
}

// np.interp restricted to a run of fewer than h consecutive sample positions x0, x0+1, ...: the
// breakpoints xp[j] = h*(off+j) are h apart, so the run meets at most one of them and two linear
// pieces (found once per contribution, with one division each) cover it.  Same slope and same
// evaluation formula as interp_w / numpy, hence the same values.
struct Piece2 { double b1, xa, fa, sa, xb, fb, sb; };

template <class A> __device__ inline void piece_of(int j, double h, double rh, double off, int nfr, const A& fp, double& xj, double& fj, double& sj) {
    if (j < 0) { xj = 0.0; fj = fp(0); sj = 0.0; return; }                          // left of xp[0]: fp(0)
    if (j >= nfr - 1) { xj = 0.0; fj = fp(nfr - 1); sj = 0.0; return; }             // at / right of xp[last]
    xj = h * (off + (double)j);
    const double xj1 = h * (off + (double)(j + 1));
    fj = fp(j);
    sj = slope_of(fp(j + 1) - fp(j), xj1 - xj, h, rh);
}

template <class A> __device__ inline Piece2 make_piece2(double x0, double h, double rh, double off, int nfr, const A& fp) {
    Piece2 q;
    int j;
    if (nfr == 1 || x0 < h * (off + 0.0)) j = -1;
    else {
        j = (int)floor(x0 * rh - off);
        if (j < 0) j = 0;
        if (j > nfr - 1) j = nfr - 1;
        while (j > 0 && x0 < h * (off + (double)j)) j--;
        while (j < nfr - 1 && x0 >= h * (off + (double)(j + 1))) j++;
    }
    if (nfr == 1) j = nfr;                                       // constant everywhere
    piece_of(j, h, rh, off, nfr, fp, q.xa, q.fa, q.sa);
    piece_of(j + 1, h, rh, off, nfr, fp, q.xb, q.fb, q.sb);
    q.b1 = (j + 1 <= nfr - 1 && nfr > 1) ? h * (off + (double)(j + 1)) : INFINITY;
    return q;
}

// ---- records ---------------------------------------------------------------------------------------------------------
// body of node (frame, slot), sample m of its segment (0 <= m < h):
//   phase(m) = ph0 + step m + [m <= fmb: fa0 m + fsa T(m) | smb + fb0 (m - fmb) + fsb (T(m) - tmb)],  T(m) = m (m - 1) / 2
//   (coefficients pre-scaled by 2 pi / sr), amplitude(m) = m < mmb ? ma0 + msa m : mb0 + msb m       (PVAnalysis.py:734-736)
// so phase(m+1) - phase(m) = step + (m < fmb ? fa0 + fsa m : fb0 + fsb m), and its own increment is fsa, then fsb.
// da / db = exp(i fsa) - 1, exp(i fsb) - 1.  tmb = T(fmb) is an exact product of small integers: the readers form it.
// One cache line per record: the records are the resynthesis' largest stream (written once, read once).
struct __attribute__((aligned(32))) BodyRec {
    double ph0, step, fa0, fsa, fb0, fsb, smb;
    double ma0, msa, mb0, msb;
    double dar, dai, dbr, dbi;
    int fmb, mmb;
};
static_assert(sizeof(BodyRec) == 128, "BodyRec layout");
constexpr int kLaneB = 80;                       // bytes per lane in the kernels' store staging: 64 + 16 (conflict-free 16-byte rows)
// attack: sample j of [0, edgsam) at output index o0 + j is ah (1 - cos(pi j / edgsam)) cos(ph0 - 2 pi (edgsam - j) cfr);
// release: ah (1 + cos(pi j / edgsam)) cos(ph0 + 2 pi (j + 1) cfr)                                  (PVAnalysis.py:740-751)
struct __attribute__((aligned(32))) EdgeRec {
    double ph0, cfr, ah, wr, wi;     // (wr, wi) = exp(i 2 pi cfr)
    long long o0;
    int pad0, pad1;
};
static_assert(sizeof(EdgeRec) == 64, "EdgeRec layout");

struct SynthK {
    const double *f, *mag, *realph;
    const int32_t *pid, *pst, *pln;
    int64_t F, P, N;
    int K, h, minframes, no_phcor, edgsam, EF, rps;
    int64_t edgsamp;
    double sr, dh, fstep, dfr, offf, sc, vr, vi;     // (vr, vi) = exp(i pi / edgsam)
    double rsr, rdh, rfstep;                         // 1 / sr, 1 / dh, 1 / fstep, correctly rounded (div_const)
    // workspace
    unsigned long long* cursor;
    int* segflag;                    // [output segments]: == gen where k_synth_extras has something to add (set by k_synth_params;
    int gen;                         //  gen is the call's number on this workspace: nothing has to be cleared between calls)
    int64_t nseg_all;
    long long* off;                  // [P]  first point of the partial in the partial-major arrays; -1: it does not sound
    double *cf, *cm, *cr;            // [N]  partial-major f / mag / realph
    BodyRec* body;                   // [(fb1 - fb0) K]
    EdgeRec *att, *rel;              // [(fx1 - fx0) K]
    unsigned long long *bbits, *xbits, *abits, *rbits;   // bodies for k_synth_bodies | for k_synth_extras | attacks | releases
    int c1, c2, R;                   // the cuts of k_synth_bodies' runs and their length (RunCuts)
    int64_t fx0, fx1, fb0, fb1;      // frames whose edges / bodies this slice holds
    int64_t seg0, nseg;              // output segments of the sample launch
    double* w;
    int64_t wlen;
};

// PVAnalysis.py:1061, 1067: partials that sound
__device__ __forceinline__ bool sounds(const SynthK& q, int st, int nfr) {
    return !(nfr < q.minframes || nfr < 1) && !((int64_t)st * q.h - q.edgsam + q.edgsamp < 0);
}

__global__ __launch_bounds__(256) void k_synth_alloc(SynthK q) {
    const int64_t pid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    long long len = 0;
    if (pid < q.P) {
        const int st = q.pst[pid], nfr = q.pln[pid];
        if (sounds(q, st, nfr)) len = nfr;
    }
    long long incl = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    const long long total = __shfl(incl, 63);
    long long base = 0;
    if (lane == 63 && total > 0) base = (long long)atomicAdd(q.cursor, (unsigned long long)total);
    base = __shfl(base, 63);
    if (pid < q.P) q.off[pid] = (len > 0 && base + incl <= q.N) ? base + incl - len : -1;   // (a table that claims more points than there are nodes is not followed)
}

__global__ __launch_bounds__(256) void k_synth_scatter(SynthK q) {
    const int64_t node = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (node >= q.N) return;
    const int pid = q.pid[node];
    if (pid < 0 || pid >= q.P) return;
    const long long off = q.off[pid];
    if (off < 0) return;
    const int64_t fr = node / q.K;
    const int64_t ii = fr - q.pst[pid];
    if (ii < 0 || ii >= q.pln[pid]) return;
    q.cf[off + ii] = q.f[node];
    q.cm[off + ii] = q.mag[node];
    q.cr[off + ii] = q.realph[node];
}

// The closed forms of node (fr, slot) -- point ii of a partial of nfr points starting at frame st, its points through the
// accessors pf / pm / pr (f, mag, realph) -- into the slice's records; what it found: body / irregular body / attack / release
struct NodeKinds { bool isb, isx, isa, isr; };
// store_body(c): what to do with the node's body record (its place is q.body[(fr - q.fb0) K + slot])
template <class AF, class AM, class AR, class SB>
__device__ __forceinline__ NodeKinds node_params(const SynthK& q, int64_t li, int64_t node, int64_t fr, int st, int nfr, int ii_, const AF& pf, const AM& pm, const AR& pr,
                                                 SB&& store_body) {
    bool isb = false, isx = false, isa = false, isr = false;
    const int ii = ii_, h = q.h;
    const double dh = q.dh;
    const double offf = q.offf, offm = q.dfr;                       // dfr + .5, dfr: PVAnalysis.py:701-702
    if (ii == 0) {
        // attack, PVAnalysis.py:742-745: output index st*h - edgsam + j, j < edgsam; amplitude msig[0]
        EdgeRec e;
        e.o0 = (long long)st * h - q.edgsam;
        e.cfr = div_const(pf(0) * 1.0, q.sr, q.rsr);
        e.ph0 = pr(0);
        e.ah = interp_w(0.0, dh, q.rdh, offm, nfr, pm) / 2.;
        fsincos(kPi2 * e.cfr, e.wi, e.wr);
        e.pad0 = e.pad1 = 0;
        q.att[li] = e;
        isa = true;
        for (int64_t sg = fr - q.EF; sg < fr; sg++) if (sg >= 0 && sg < q.nseg_all) q.segflag[sg] = q.gen;
    }
    const bool wantb = fr >= q.fb0 && fr < q.fb1;
    isr = (ii == nfr - 1);
    if (wantb || isr) {
        BodyRec c;
        const double nbase = dh * (double)ii;
        // fsig at nbase, nbase + h, nbase + 2 h (PVAnalysis.py:711-718, 724-729)
        const double fs1 = interp_w(nbase + dh, dh, q.rdh, offf, nfr, pf);
        // fsig(nbase + m) = fa0 + fsa m for m < fmb, fb0 + fsb m beyond; msig likewise
        {
            const Piece2 p2 = make_piece2(nbase, dh, q.rdh, offf, nfr, pf);
            c.fa0 = p2.sa * (nbase - p2.xa) + p2.fa; c.fsa = p2.sa;
            c.fb0 = p2.sb * (nbase - p2.xb) + p2.fb; c.fsb = p2.sb;
            const double d = ceil(p2.b1 - nbase);
            c.fmb = d < 0.0 ? 0 : (d > dh ? h : (int)d);
        }
        {
            const Piece2 p2 = make_piece2(nbase, dh, q.rdh, offm, nfr, pm);
            c.ma0 = p2.sa * (nbase - p2.xa) + p2.fa; c.msa = p2.sa;
            c.mb0 = p2.sb * (nbase - p2.xb) + p2.fb; c.msb = p2.sb;
            const double d = ceil(p2.b1 - nbase);
            c.mmb = d < 0.0 ? 0 : (d > dh ? h : (int)d);
        }
        // phase corrections, PVAnalysis.py:710-715
        const double fs0 = c.fa0;                                 // fsig(nbase): the first piece at m = 0, the same expression as np.interp's
        const double phcor = q.no_phcor ? 0.0 : div_const(kPi * (fs1 - fs0), q.fstep, q.rfstep) / 2.;
        c.ph0 = pr(ii) + phcor;                                   // PVAnalysis.py:721
        const double tmb = 0.5 * (double)c.fmb * (double)(c.fmb - 1);
        c.smb = c.fa0 * (double)c.fmb + c.fsa * tmb;              // sum of the first fmb terms
        // ph[h-1] + ph0 (before the discontinuity ramp): prefix over the h - 1 terms fsig(nbase + 0 .. h-2)
        double lastsum;
        {
            const int m = h - 1;
            const double tm = 0.5 * (double)m * (double)(m - 1);
            lastsum = (m <= c.fmb) ? c.fa0 * (double)m + c.fsa * tm : c.smb + c.fb0 * (double)(m - c.fmb) + c.fsb * (tm - tmb);
        }
        const double lastph = kPi2 * div_const(lastsum, q.sr, q.rsr) + c.ph0;
        c.step = 0.0;
        if (ii < nfr - 1) {
            // discontinuity ramp towards the next point, PVAnalysis.py:724-729
            const double fs2 = interp_w(nbase + 2.0 * dh, dh, q.rdh, offf, nfr, pf);
            const double phcornext = q.no_phcor ? 0.0 : div_const(kPi * (fs2 - fs1), q.fstep, q.rfstep) / 2.;
            const double phend = lastph + div_const(kPi2 * fs1, q.sr, q.rsr);
            const double arg = pr(ii + 1) + phcornext - phend + kPi;
            double md = fmod(arg, kPi2);                          // np.mod: sign of the divisor
            if (md != 0.0 && md < 0.0) md += kPi2;
            c.step = div_const(md - kPi, dh, q.rdh);              // np.linspace(0, dph, h+1)[:-1]
        }
        if (isr) {
            // release, PVAnalysis.py:748-751: output index (st+nfr)*h + j, j < edgsam; amplitude msig[hop*nfr]
            EdgeRec e;
            e.o0 = ((long long)st + nfr) * h;
            e.cfr = div_const(pf(nfr - 1) * 1.0, q.sr, q.rsr);
            e.ph0 = lastph;
            e.ah = interp_w(dh * (double)nfr, dh, q.rdh, offm, nfr, pm) / 2.;
            fsincos(kPi2 * e.cfr, e.wi, e.wr);
            e.pad0 = e.pad1 = 0;
            q.rel[li] = e;
            for (int64_t sg = fr + 1; sg <= fr + q.EF; sg++) if (sg < q.nseg_all) q.segflag[sg] = q.gen;
        }
        if (wantb) {
            // 2 pi / sr folded into the two-piece phase polynomial
            const double sc = q.sc;
            c.fa0 *= sc; c.fsa *= sc; c.fb0 *= sc; c.fsb *= sc; c.smb *= sc;
            // rotations of the increment: exp(i x) - 1 = (-2 sin^2(x/2), 2 sin(x/2) cos(x/2)) -- x is tiny
            expm1i(c.fsa, c.dar, c.dai);
            expm1i(c.fsb, c.dbr, c.dbi);
            store_body(c);
            // does every run of k_synth_bodies lie on one piece of fsig and one of msig?  (a change at position x
            // is harmless at 0, h, a cut, or a multiple of R past the cut before it)
            auto on_edge = [&](int x) {
                if (x <= 0 || x >= h || x == q.c1 || x == q.c2) return true;
                const int base = x > q.c2 ? q.c2 : (x > q.c1 ? q.c1 : 0);
                return ((x - base) & (q.R - 1)) == 0;                 // (R: 8, 16 or 32)
            };
            isb = on_edge(c.fmb) && on_edge(c.mmb);
            isx = !isb;
            if (isx && fr < q.nseg_all) q.segflag[fr] = q.gen;
        }
    }
    NodeKinds k;
    k.isb = isb; k.isx = isx; k.isa = isa; k.isr = isr;
    return k;
}

__global__ __launch_bounds__(256) void k_synth_params(SynthK q) {
    const int64_t li = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nloc = (q.fx1 - q.fx0) * q.K;
    NodeKinds kd;
    kd.isb = kd.isx = kd.isa = kd.isr = false;
    if (li < nloc) {
        const int64_t node = q.fx0 * q.K + li;
        const int64_t fr = q.fx0 + (int64_t)((unsigned)li / (unsigned)q.K);     // (a slice holds about a million nodes)
        const int pid = q.pid[node];
        if (pid >= 0 && pid < q.P) {
            const long long off = q.off[pid];
            const int st = q.pst[pid], nfr = q.pln[pid];
            const int64_t ii64 = fr - st;
            if (off >= 0 && ii64 >= 0 && ii64 < nfr) {
                const double* cf = q.cf + off;
                const double* cm = q.cm + off;
                const double* cr = q.cr + off;
                kd = node_params(q, li, node, fr, st, nfr, (int)ii64, [&](int j) { return cf[j]; }, [&](int j) { return cm[j]; }, [&](int j) { return cr[j]; },
                                 [&](const BodyRec& c) { q.body[(fr - q.fb0) * q.K + (node - fr * q.K)] = c; });
            }
        }
    }
    const unsigned long long bb = __ballot(kd.isb), bx = __ballot(kd.isx), ba = __ballot(kd.isa), br = __ballot(kd.isr);
    if ((threadIdx.x & 63) == 0) {
        const int64_t w = li >> 6;
        q.bbits[w] = bb; q.xbits[w] = bx; q.abits[w] = ba; q.rbits[w] = br;
    }
}

// The same without the partial-major copy, for rows of at most 16 peaks: a workgroup's 256 nodes are consecutive, so the
// frames their partials' points can sit in -- WB before the first, 3 behind the last -- are a short run of analysis rows:
// partial_id, f, mag and realph of those rows are staged in LDS (one coalesced pass), a node finds the slots of its
// partial's points by scanning the staged partial_id rows (a nibble per point in a 64-bit word) and reads the values where
// they are.  No k_synth_alloc / k_synth_scatter, no second copy of the analysis arrays.
constexpr int kDirectMaxK = 16, kDirectMaxWL = 16;
#ifndef PVX_PARAMS_WAVES
#define PVX_PARAMS_WAVES 7
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PVX_PARAMS_WAVES, PVX_PARAMS_WAVES))) void k_synth_params_direct(SynthK q, int WB, int rows_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    const int K = q.K, tid = threadIdx.x;
    const int64_t li0 = (int64_t)blockIdx.x * 256, li = li0 + tid;
    const int64_t nloc = (q.fx1 - q.fx0) * K;
    // frames of this workgroup's nodes and the rows staged around them
    const int64_t liB = li0 + 255 < nloc - 1 ? li0 + 255 : nloc - 1;          // the workgroup's last node (slice-local)
    // (a slice holds about a million nodes: 32-bit quotients -- a 64-bit division by a run-time K is ~100 instructions)
    int64_t frow0 = q.fx0 + (int64_t)((unsigned)li0 / (unsigned)K) - WB, frow1 = q.fx0 + (int64_t)((unsigned)liB / (unsigned)K) + 3 + 1;
    if (frow0 < 0) frow0 = 0;
    if (frow1 > q.F) frow1 = q.F;
    const int nrows = (int)(frow1 - frow0);                          // <= rows_cap (host)
    double* Lf = (double*)dsm;
    double* Lm = Lf + (size_t)rows_cap * K;
    double* Lr = Lm + (size_t)rows_cap * K;
    int* Lp = (int*)(Lr + (size_t)rows_cap * K);
    for (int i = tid; i < nrows * K; i += 256) {
        const int64_t g = frow0 * K + i;
        Lp[i] = q.pid[g]; Lf[i] = q.f[g]; Lm[i] = q.mag[g]; Lr[i] = q.realph[g];
    }
    __syncthreads();
#ifdef PVX_AB_PARAMS_EXIT0     // tools/ab: the launch and the staging of the rows alone
    if (q.F >= 0) return;
#endif
    NodeKinds kd;
    kd.isb = kd.isx = kd.isa = kd.isr = false;
    BodyRec crec;                                                     // this node's body record, if it has one
    bool has_body = false;
    if (li < nloc) {
        const int64_t node = q.fx0 * K + li;
        const int64_t fr = q.fx0 + (int64_t)((unsigned)li / (unsigned)K);
        const int pid = Lp[(int)(node - frow0 * K)];
        if (pid >= 0 && pid < q.P) {
            const int st = q.pst[pid], nfr = q.pln[pid];
            const int64_t ii64 = fr - st;
            if (sounds(q, st, nfr) && ii64 >= 0 && ii64 < nfr) {
                const int ii = (int)ii64;
                // the slots of points j0 .. j1 of the partial (what the closed forms can touch: k_synth_params' window)
                const int j0 = ii - WB > 0 ? ii - WB : 0, j1 = ii + 3 < nfr - 1 ? ii + 3 : nfr - 1;
                unsigned long long slots = 0ull;
                for (int j = j0; j <= j1; j++) {
                    const int64_t f2 = (int64_t)st + j;
                    unsigned key = 0x100u;                            // (different ? 256 : 0) + slot, minimum over the row
                    if (f2 >= frow0 && f2 < frow1) {
                        const int* row = Lp + (int)(f2 - frow0) * K;
#pragma unroll
                        for (int s2 = 0; s2 < kDirectMaxK; s2++) {
                            if (s2 < K) {
                                const unsigned d = (unsigned)(row[s2] ^ pid);
                                const unsigned k2 = ((d < 1u ? d : 1u) << 8) | (unsigned)s2;     // (integer minima: no compare + select)
                                key = k2 < key ? k2 : key;
                            }
                        }
                    }
                    slots |= (unsigned long long)(key & 15u) << (4 * (j - j0));
                }
                const int rbase = (int)((int64_t)st - frow0) * K;    // (row of point j: rbase / K + j)
                auto at = [&](const double* L, int j) {
                    int jj = j < j0 ? j0 : (j > j1 ? j1 : j);          // (never outside for a consistent table)
                    return L[rbase + jj * K + (int)((slots >> (4 * (jj - j0))) & 15ull)];
                };
                kd = node_params(q, li, node, fr, st, nfr, ii, [&](int j) { return at(Lf, j); }, [&](int j) { return at(Lm, j); }, [&](int j) { return at(Lr, j); },
                                 [&](const BodyRec& c) { crec = c; has_body = true; });
            }
        }
    }
    // ---- the wave's records are consecutive in memory (a node's record sits at node - fb0 K): they leave through LDS, half a
    // record at a time, so that a store instruction writes 64 contiguous bytes of 16 records instead of 16 bytes of 64 (one
    // 128-byte line per lane, eight instructions each)
    // (the staging rows take the place of the staged analysis rows: 20 KB per workgroup instead of 30 -- with the registers held to
    // seven waves per SIMD every workgroup of BASELINE config 2's 1 615 is resident at once, where five per CU made a second round)
    __syncthreads();
    {
#ifdef PVX_AB_PARAMS_NOSTORE   // tools/ab: without the records' way out
        const unsigned long long bw = __ballot(has_body && crec.ph0 == 12345.0);
#else
        const unsigned long long bw = __ballot(has_body);
#endif
        if (bw != 0ull) {                                             // wave-uniform
            unsigned char* stg = dsm + (size_t)(tid >> 6) * (64 * kLaneB);
            const int lane = tid & 63;
            unsigned char* const rec0 = (unsigned char*)(q.body + (li0 + (tid & ~63) + (q.fx0 - q.fb0) * K));
            int4 pc[8];
            __builtin_memcpy(pc, &crec, sizeof(BodyRec));
#pragma unroll
            for (int half = 0; half < 2; half++) {
#pragma unroll
                for (int i = 0; i < 4; i++) *(int4*)(stg + lane * kLaneB + 16 * i) = pc[4 * half + i];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int r = 16 * i + (lane >> 2);
                    const int4 v = *(const int4*)(stg + r * kLaneB + 16 * (lane & 3));
                    if ((bw >> r) & 1ull) *(int4*)(rec0 + (size_t)r * sizeof(BodyRec) + 64 * half + 16 * (lane & 3)) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    const unsigned long long bb = __ballot(kd.isb), bx = __ballot(kd.isx), ba = __ballot(kd.isa), br = __ballot(kd.isr);
    if ((tid & 63) == 0) {
        const int64_t w = li >> 6;
        q.bbits[w] = bb; q.xbits[w] = bx; q.abits[w] = ba; q.rbits[w] = br;
    }
}

// z *= w (complex)
#define PVX_CMUL(zr, zi, wr, wi)                                        \
    do {                                                                \
        const double t_ = __builtin_fma(zr, wr, -(zi * wi));            \
        zi = __builtin_fma(zr, wi, zi * wr);                            \
        zr = t_;                                                        \
    } while (0)
// w += w * d (d = exp(i x) - 1, |d| << 1: the rounding errors scale with |d|, not with |w|)
#define PVX_CROT(wr, wi, dr, di)                                        \
    do {                                                                \
        const double t_ = __builtin_fma(wr, dr, __builtin_fma(-wi, di, wr)); \
        wi = __builtin_fma(wr, di, __builtin_fma(wi, dr, wi));          \
        wr = t_;                                                        \
    } while (0)

// What the sample kernels need of a slice (kept apart from SynthK: every field is a scalar register pair for the whole kernel)
struct SampK {
    const BodyRec* body;                 // [(fb1 - fb0) K]
    const EdgeRec *att, *rel;            // [(fx1 - fx0) K]
    const unsigned long long *bbits, *xbits, *abits, *rbits;
    const int* segflag;                  // per output segment: == gen where k_synth_extras has something to add (set by k_synth_params)
    int gen;
    double* w;
    int64_t wlen, seg0, nseg, nthreads;
    int fx0, fx1, fb0, fb1;              // frames whose edges / bodies the slice holds
    int K, h, EF, edgsam, rps;
    int xcs;                             // k_synth_extras: log2 of its chunk of segments
    int tile;                            // k_synth_bodies: records per wave and LDS tile (>= kTile)
    int c1, c2, n0, n1;                  // k_synth_bodies: the cuts of a segment's runs (RunCuts)
    double vr, vi;                       // k_synth_extras / add_edges: exp(i pi / edgsam)
    int tail_first;                      // k_synth_bodies: its last tail_first workgroups (the waveform's end: the releases) are dispatched first
    int edges_inline;                    // k_synth_bodies adds a flagged segment's attacks / releases itself (no k_synth_extras<, false> launch before it)
};

// Waves per SIMD the register allocation aims at (R sums = 2 R registers + the recurrences' state; the compiler fills
// whatever it is given with samples in flight)
#ifndef PVX_SYNTH_WAVES
#define PVX_SYNTH_WAVES(R) ((R) <= 8 ? 5 : ((R) <= 16 ? 4 : 3))
#endif

// the set bits [n0, n1) of a bit array, in ascending order
template <class Fn> __device__ __forceinline__ void for_bits(const unsigned long long* bits, int n0, int n1, Fn&& fn) {
    for (int wd = n0 >> 6; wd <= ((n1 - 1) >> 6); wd++) {
        unsigned long long mb = bits[wd];
        if (wd == (n0 >> 6)) mb &= ~0ull << (n0 & 63);
        if (wd == ((n1 - 1) >> 6) && (n1 & 63)) mb &= (1ull << (n1 & 63)) - 1ull;
        while (mb) {
            const int li = (wd << 6) + __builtin_ctzll(mb);
            mb &= mb - 1ull;
            fn(li);
        }
    }
}

// the same with the words wdA and wdB of the array already in registers
template <class Fn> __device__ __forceinline__ void for_bits_pre(const unsigned long long* bits, int n0, int n1, int wdA, unsigned long long wA,
                                                                 int wdB, unsigned long long wB, Fn&& fn) {
    for (int wd = n0 >> 6; wd <= ((n1 - 1) >> 6); wd++) {
        unsigned long long mb = wd == wdA ? wA : (wd == wdB ? wB : bits[wd]);
        if (wd == (n0 >> 6)) mb &= ~0ull << (n0 & 63);
        if (wd == ((n1 - 1) >> 6) && (n1 & 63)) mb &= (1ull << (n1 & 63)) - 1ull;
        while (mb) {
            const int li = (wd << 6) + __builtin_ctzll(mb);
            mb &= mb - 1ull;
            fn(li);
        }
    }
}

// ---- the attacks and releases that touch the run [s, s + R) of output segment `seg`, added to a[] (float64 arithmetic whatever the
// sums' type: a partial has two edges, its bodies are many).  Attacks of partials starting at frames seg+1 .. seg+EF (kind 1), then
// releases of partials whose last frame is seg-EF .. seg-1 (kind 2).  The record of the NEXT edge is fetched before the current one
// is worked on: on a short waveform a workgroup is one wave with nothing else to hide a load behind.
template <int R, typename S> __device__ __forceinline__ void add_edges(const SampK& q, int seg, int s, S (&a)[R]) {
    const int h = q.h, K = q.K;
    int kind = 0, wd = 0, wend = -1, n0 = 0, n1 = 0;
    unsigned long long mb = 0ull;
    auto masked = [&](const unsigned long long* bits, int w) {
        unsigned long long m = bits[w];
        if (w == (n0 >> 6)) m &= ~0ull << (n0 & 63);
        if (w == ((n1 - 1) >> 6) && (n1 & 63)) m &= (1ull << (n1 & 63)) - 1ull;
        return m;
    };
    auto advance = [&]() -> int {                                     // the next edge's node (kind set), or -1
        for (;;) {
            if (mb) { const int b = __builtin_ctzll(mb); mb &= mb - 1ull; return (wd << 6) + b; }
            if (wd < wend) { wd++; mb = masked(kind == 1 ? q.abits : q.rbits, wd); continue; }
            if (kind >= 2) return -1;
            kind++;
            int f0 = kind == 1 ? seg + 1 : seg - q.EF, f1 = kind == 1 ? seg + q.EF + 1 : seg;
            if (f0 < q.fx0) f0 = q.fx0;
            if (f1 > q.fx1) f1 = q.fx1;
            wend = -1; wd = 0;
            if (f1 <= f0) continue;
            n0 = (f0 - q.fx0) * K; n1 = (f1 - q.fx0) * K;
            wd = n0 >> 6; wend = (n1 - 1) >> 6;
            mb = masked(kind == 1 ? q.abits : q.rbits, wd);
        }
    };
    // float32 sums: the edge's two rotations in float32 as well, seeded per run from the float64 closed forms like a body's (an edge is
    // 2 h samples of every partial that starts or stops: on noise five times the bodies' work; 0.088 -> 0.074 ms there).  NE edges side
    // by side, as k_synth_bodies' float32 loop takes its bodies: measured at NE = 2, 7 registers spilled and 0.079 ms -- one at a time.
#ifndef PVX_SYNTH_NE
#define PVX_SYNTH_NE 1
#endif
    constexpr int NE = sizeof(S) == 4 ? PVX_SYNTH_NE : 1;
    struct Edge { S zr, zi, ur, ui, ah, ahs, ewr, ewi; unsigned ulo, un; };
    auto fm = [](const S x_, const S y_, const S z_) -> S {
        if constexpr (sizeof(S) == 4) return __builtin_fmaf(x_, y_, z_); else return __builtin_fma(x_, y_, z_);
    };
    auto setup = [&](const int knd, const EdgeRec& er, Edge& E) {
        const EdgeRec* e = &er;
        E.zr = E.zi = E.ur = E.ui = E.ah = E.ahs = E.ewr = E.ewi = (S)0; E.ulo = 0u; E.un = 0u;     // (un = 0: no sample of the run sounds)
        const long long j0l = (long long)seg * h + s - e->o0;
        if (j0l + R <= 0 || j0l >= q.edgsam) return;
        const int j0 = (int)j0l;
        const double cfr = e->cfr;
        const double x = (knd == 1) ? e->ph0 - kPi2 * ((double)(q.edgsam - j0) * cfr)
                                    : e->ph0 + kPi2 * ((double)(j0 + 1) * cfr);
        if constexpr (sizeof(S) == 4) {
            fsincos_f(x, E.zi, E.zr);
            fsincos_f(kPi * (double)j0 / (double)q.edgsam, E.ui, E.ur);
        } else {
            fsincos(x, E.zi, E.zr);
            fsincos(kPi * (double)j0 / (double)q.edgsam, E.ui, E.ur);
        }
        E.ah = (S)e->ah; E.ahs = (knd == 1) ? -E.ah : E.ah;
        E.ewr = (S)e->wr; E.ewi = (S)e->wi;
        E.ulo = (unsigned)(-j0); E.un = (unsigned)q.edgsam;                 // sample k sounds when 0 <= j0 + k < edgsam
    };
    const S evr = (S)q.vr, evi = (S)q.vi;
    auto run = [&](auto nec, Edge (&E)[NE]) {
        constexpr int NX = decltype(nec)::value;
#pragma unroll
        for (int k = 0; k < R; k++) {
#pragma unroll
            for (int b = 0; b < NX; b++) {
                const S v = fm(fm(E[b].ahs, E[b].ur, E[b].ah), E[b].zr, a[k]);
                a[k] = ((unsigned)k - E[b].ulo < E[b].un) ? v : a[k];
            }
#pragma unroll
            for (int b = 0; b < NX; b++) {
                const S t_ = fm(E[b].zr, E[b].ewr, -(E[b].zi * E[b].ewi));    // z *= exp(i 2 pi cfr)
                E[b].zi = fm(E[b].zr, E[b].ewi, E[b].zi * E[b].ewr);
                E[b].zr = t_;
                const S u_ = fm(E[b].ur, evr, -(E[b].ui * evi));              // u *= exp(i pi / edgsam)
                E[b].ui = fm(E[b].ur, evi, E[b].ui * evr);
                E[b].ur = u_;
            }
        }
    };
    for (;;) {
        int li[NE], kd[NE];
        li[0] = advance(); kd[0] = kind;
        if (li[0] < 0) break;
#pragma unroll
        for (int b = 1; b < NE; b++) { li[b] = advance(); kd[b] = kind; }
        Edge E[NE];
        bool any = false, second = false;
#pragma unroll
        for (int b = 0; b < NE; b++) {
            E[b].zr = E[b].zi = E[b].ur = E[b].ui = E[b].ah = E[b].ahs = E[b].ewr = E[b].ewi = (S)0; E[b].ulo = 0u; E[b].un = 0u;
            if (li[b] >= 0) setup(kd[b], (kd[b] == 1 ? q.att : q.rel)[li[b]], E[b]);
            any = any || E[b].un != 0u;
            if (b > 0) second = second || E[b].un != 0u;
        }
        if (__ballot(any) == 0ull) continue;                                // (no lane's run is touched by these edges)
        if (NE == 1 || __ballot(second) == 0ull) run(std::integral_constant<int, 1>{}, E);
        else run(std::integral_constant<int, NE>{}, E);
    }
}

// seeds of a body at sample s on the pieces (fa, ma): z = exp(i phase(s)), w = exp(i (phase(s+1) - phase(s))), amplitude
#define PVX_BODY_PHASE(c, s, ds, ts, fa)                                                                            \
    double ph_, dl_;                                                                                                \
    if (fa) {                                                                                                       \
        const double fa0_ = (c)->fa0, fsa_ = (c)->fsa;                                                              \
        ph_ = __builtin_fma(fsa_, ts, fa0_ * ds);                                                                   \
        dl_ = __builtin_fma(fsa_, ds, fa0_);                                                                        \
    } else {                                                                                                        \
        const double fb0_ = (c)->fb0, fsb_ = (c)->fsb;                                                              \
        const double dfmb_ = (double)(c)->fmb;                                                                      \
        ph_ = __builtin_fma(fsb_, ts - 0.5 * dfmb_ * (dfmb_ - 1.0), __builtin_fma(fb0_, ds - dfmb_, (c)->smb));     \
        dl_ = __builtin_fma(fsb_, ds, fb0_);                                                                        \
    }                                                                                                               \
    {                                                                                                               \
        const double step_ = (c)->step;                                                                             \
        ph_ += __builtin_fma(step_, ds, (c)->ph0);                                                                  \
        dl_ += step_;                                                                                               \
    }
#define PVX_BODY_SEEDS(c, s, ds, ts, fa, ma)                                                                        \
    PVX_BODY_PHASE(c, s, ds, ts, fa)                                                                                \
    double zr, zi, wr, wi;                                                                                          \
    fsincos(ph_, zi, zr);                                                                                           \
    fsincos(dl_, wi, wr);                                                                                           \
    const double dr = (fa) ? (c)->dar : (c)->dbr, di = (fa) ? (c)->dai : (c)->dbi;                                  \
    const double dms = (ma) ? (c)->msa : (c)->msb;                                                                  \
    double ms = __builtin_fma(dms, ds, (ma) ? (c)->ma0 : (c)->mb0);

// ---- the bodies whose pieces change at the launch-wide cuts (all of them, but for float rounding of an odd dfr): one thread
// per run of R samples, R sums in registers over all bodies of the segment, ONE unrolled loop.  A workgroup's 256 runs are
// consecutive, so the records of its segments are consecutive too: they are staged through LDS a tile at a time (one
// coalesced round of loads per tile instead of a dependent round trip per contribution and thread), and the finished sums
// leave through LDS as well, so that a store instruction writes 64-byte pieces instead of 16 bytes per 256.
constexpr int kTile = 40;                        // least records per wave and LDS tile (40 x 128 B = 5 KB = the wave's store staging); SampK::tile
#ifndef PVX_BODIES_TB
#define PVX_BODIES_TB 64
#endif
constexpr int kBodiesTB = PVX_BODIES_TB;         // threads per workgroup of k_synth_bodies: its waves never meet, and one-wave workgroups
                                                 // start as soon as ANY wave slot is free (four-wave ones wait for four: 124 against 129 us)
// S = double: the reference's arithmetic (every waveform fixture within 1e-10).  S = float (plans at precision 32, PVX_SYNTH_F32): the seeds
// of a run from the float64 closed form as before -- the phase reaches 1e4 rad --, then the rotation recurrence, the amplitude
// ramp and the R sums in float32: the recurrence's rounding grows like R^2 / 2 x 6e-8 = 3e-5 rad over a run of 32, inside the
// stated precision-32 tolerance of 1e-4 max|w|, at half the vector-ALU cycles of the float64 loop and half the
// registers for the sums (four waves per SIMD instead of three).
#ifndef PVX_SYNTH_NB
#define PVX_SYNTH_NB 2                         // bodies whose recurrences the float32 loop runs side by side
#endif
#ifndef PVX_SYNTH_WAVES_F32
#define PVX_SYNTH_WAVES_F32(R) ((R) <= 8 ? 5 : 4)
#endif
template <int R, typename S = double>
__global__ __launch_bounds__(kBodiesTB) __attribute__((amdgpu_waves_per_eu((sizeof(S) == 4 ? PVX_SYNTH_WAVES_F32(R) : PVX_SYNTH_WAVES(R)), (sizeof(S) == 4 ? PVX_SYNTH_WAVES_F32(R) : PVX_SYNTH_WAVES(R))))) void k_synth_bodies(SampK q) {
    // everything is per wave (a wave's 64 runs are consecutive, so are the records of their segments): no workgroup barrier
    // (q.tile records per wave, at least kTile: all the records a wave's 64 runs can need -- a few segments' K slots -- so that
    // its lanes walk their segments' bodies together; with a tile shorter than that, the lanes of different segments take
    // turns, tile by tile, and half the wave idles: rows of 20 slots cost twice the time of rows of 8 for the same bodies)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
    __shared__ long long s_o[kBodiesTB];
    __shared__ int s_len[kBodiesTB];
    static_assert(kTile * sizeof(BodyRec) == 64 * kLaneB, "store staging = the least record tile");
    const int tid = threadIdx.x, lane = tid & 63, wbase = tid & ~63;
    unsigned char* lds = lds_dyn + (size_t)(tid >> 6) * q.tile * sizeof(BodyRec);
    // (the workgroups of the waveform's end have the releases to add on top of their bodies: dispatched last, they would be the launch's
    // tail -- 0.073 -> 0.064 ms on a time-stretched resynthesis, hop 700)
    const unsigned bx = blockIdx.x < (unsigned)q.tail_first ? gridDim.x - 1u - blockIdx.x : blockIdx.x - (unsigned)q.tail_first;
    const int64_t gid0 = (int64_t)bx * kBodiesTB, gid = gid0 + tid;
    const bool live = gid < q.nthreads;
#ifdef PVX_AB_SYNTH_EXIT0       // tools/ab: what 12 920 one-wave workgroups cost before they do anything
    if (q.wlen >= 0) return;
#endif
    const int64_t g = live ? gid : q.nthreads - 1;
    const bool small = q.nthreads < ((int64_t)1 << 31);              // (32-bit quotients then: a 64-bit division is ~100 instructions)
    const int64_t segl = small ? (int64_t)((unsigned)g / (unsigned)q.rps) : g / q.rps;
    const int run = (int)(g - segl * q.rps);
    const int seg = (int)(q.seg0 + segl);
    const int h = q.h, K = q.K;
    // the run's samples [s, s + len)
    int s, len;
    if (run < q.n0) { s = run * R; len = q.c1 - s; }
    else if (run < q.n0 + q.n1) { s = q.c1 + (run - q.n0) * R; len = q.c2 - s; }
    else { s = q.c2 + (run - q.n0 - q.n1) * R; len = h - s; }
    if (len > R) len = R;
    const double ds = (double)s, ts = 0.5 * ds * (double)(s - 1);
    S a[R];
    // where this thread's run goes, for the lanes that move it between memory and the wave's staging rows (four lanes per
    // run, 8 doubles at a time: an instruction then touches the 64 contiguous bytes of 16 runs)
    const bool flagged = live && q.segflag[seg] == q.gen;      // k_synth_extras has left this segment's attacks / releases in w
    s_o[tid] = (int64_t)seg * h + s;
    s_len[tid] = live ? (flagged ? -len : len) : 0;
    __builtin_amdgcn_wave_barrier();
    if (q.edges_inline) {
        // the segment's attacks and releases, by this kernel: a launch of their own in front of this one (k_synth_extras<, false>) cost
        // a dependent launch with almost nothing to do on a tone (0.142 -> 0.131 ms at BASELINE config 2) and a round trip of the flagged
        // segments through w (white noise, attacks and releases everywhere: 0.149 -> 0.083 ms)
#pragma unroll
        for (int k = 0; k < R; k++) a[k] = (S)0;
        if (flagged) add_edges<R, S>(q, seg, s, a);
    } else if (__ballot(flagged) != 0ull) {
        long long oo[4];
        int ll[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { oo[i] = s_o[wbase + 16 * i + (lane >> 2)]; ll[i] = s_len[wbase + 16 * i + (lane >> 2)]; }
#pragma unroll
        for (int c8 = 0; c8 < R / 8; c8++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int k = 8 * c8 + 2 * (lane & 3);
                const long long o = oo[i] + k;
                double2 v = make_double2(0.0, 0.0);
                if (ll[i] < 0) {
                    const bool v0 = k < -ll[i] && o < q.wlen, v1 = k + 1 < -ll[i] && o + 1 < q.wlen;
                    if (v0 && v1 && (o & 1) == 0) v = *(const double2*)(q.w + o);
                    else {
                        if (v0) v.x = q.w[o];
                        if (v1) v.y = q.w[o + 1];
                    }
                }
                *(double2*)(lds + (16 * i + (lane >> 2)) * kLaneB + 16 * (lane & 3)) = v;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const double2 v = *(const double2*)(lds + lane * kLaneB + 16 * j);
                a[8 * c8 + 2 * j] = (S)v.x; a[8 * c8 + 2 * j + 1] = (S)v.y;
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
#pragma unroll
        for (int k = 0; k < R; k++) a[k] = (S)0;
    }
    const bool mine = live && seg >= q.fb0 && seg < q.fb1;
    const int m0 = (seg - q.fx0) * K, m1 = m0 + K;                          // (a slice holds at most 2^20 + 2 EF K nodes)
    // (the words of the body bits this run will walk, fetched now: their latency passes under the record tiles' copy)
    const int wdA = m0 >> 6, wdB = (m1 - 1) >> 6;
    unsigned long long wA = 0ull, wB = 0ull;
    if (mine) { wA = q.bbits[wdA]; wB = q.bbits[wdB]; }
    // the wave's segments and their body nodes (slice-local indices; wave-uniform)
    {
        const int64_t gfirst = gid0 + wbase, glast = gfirst + 63 < q.nthreads - 1 ? gfirst + 63 : q.nthreads - 1;
        int seg_lo = (int)(q.seg0 + (small ? (int64_t)((unsigned)gfirst / (unsigned)q.rps) : gfirst / q.rps));
        int seg_hi = (int)(q.seg0 + (small ? (int64_t)((unsigned)glast / (unsigned)q.rps) : glast / q.rps)) + 1;
        if (gfirst >= q.nthreads) seg_hi = seg_lo;
        if (seg_lo < q.fb0) seg_lo = q.fb0;
        if (seg_hi > q.fb1) seg_hi = q.fb1;
        const int nlo = __builtin_amdgcn_readfirstlane((seg_lo - q.fx0) * K), nhi = __builtin_amdgcn_readfirstlane((seg_hi - q.fx0) * K);
        const BodyRec* rec = q.body - (q.fb0 - q.fx0) * K;
        for (int tile = nlo; tile < nhi; tile += q.tile) {
            const int cnt = nhi - tile < q.tile ? nhi - tile : q.tile;
            // (LDS operations of one wave execute in order: the copy below is behind the previous tile's reads, the
            // reads behind the copy)
            {
                // (five 16-byte pieces per lane in flight -- the least tile, 40 records, in one go: as a plain loop this was a
                // load, a wait and an LDS write per piece, five round trips to memory one after the other in every wave)
                const int4* src = (const int4*)(rec + tile);
                int4* dst = (int4*)lds;
                const int n16 = cnt * (int)(sizeof(BodyRec) / 16);
                for (int i0 = lane; i0 < n16; i0 += 5 * 64) {
                    const int i1 = i0 + 64 < n16 ? i0 + 64 : n16 - 1, i2 = i0 + 128 < n16 ? i0 + 128 : n16 - 1;
                    const int i3 = i0 + 192 < n16 ? i0 + 192 : n16 - 1, i4 = i0 + 256 < n16 ? i0 + 256 : n16 - 1;
                    const int4 v0 = src[i0], v1 = src[i1], v2 = src[i2], v3 = src[i3], v4 = src[i4];      // (past the end: the last piece again)
                    dst[i0] = v0; dst[i1] = v1; dst[i2] = v2; dst[i3] = v3; dst[i4] = v4;
                }
            }
            __builtin_amdgcn_wave_barrier();
            const int a0 = m0 > tile ? m0 : tile, a1 = m1 < tile + cnt ? m1 : tile + cnt;
            if (mine && a1 > a0) {
                {
                    // float32: TWO bodies side by side -- two independent recurrences for the scheduler to interleave (a wave's dependent
                    // instructions are ~8 cycles apart, and four waves per SIMD do not fill that: 0.127 -> 0.115 ms at BASELINE config 2) --,
                    // each sample's sum taking body A's term, then body B's: the additions and their order are the one-body loop's, bit for
                    // bit.  float64: one body at a time (a second body's state does not fit the three-wave register budget: 218 spilled).
                    constexpr int NB = sizeof(S) == 4 ? PVX_SYNTH_NB : 1;
                    // Seeds of a body at sample s: the float64 closed form (float32 loop: float32 polynomials, then float32 state)
                    auto seed = [&](const BodyRec* c, S (&sd)[8]) {
                        // a run lies on one piece of fsig and one of msig (sample fmb itself sits on both)
                        const bool fa = s < c->fmb, ma = s < c->mmb;
                        if constexpr (sizeof(S) == 4) {
                            PVX_BODY_PHASE(c, s, ds, ts, fa)
#ifdef PVX_AB_SYNTH_NOSEED      // tools/ab: timing-only builds (wrong waveform)
                            sd[0] = (float)ph_; sd[1] = 0.f; sd[2] = (float)dl_; sd[3] = 0.f;
#else
                            fsincos_f(ph_, sd[1], sd[0]);
                            fsincos_f(dl_, sd[3], sd[2]);
#endif
                            sd[4] = (float)(fa ? c->dar : c->dbr); sd[5] = (float)(fa ? c->dai : c->dbi);
                            sd[7] = (float)(ma ? c->msa : c->msb);
                            sd[6] = (float)__builtin_fma(ma ? c->msa : c->msb, ds, ma ? c->ma0 : c->mb0);
                        } else {
                            PVX_BODY_SEEDS(c, s, ds, ts, fa, ma)
                            sd[0] = zr; sd[1] = zi; sd[2] = wr; sd[3] = wi; sd[4] = dr; sd[5] = di; sd[6] = ms; sd[7] = dms;
                        }
                    };
                    auto fm = [](const S x, const S y, const S z) -> S {
                        if constexpr (sizeof(S) == 4) return __builtin_fmaf(x, y, z); else return __builtin_fma(x, y, z);
                    };
                    // the set bits [a0, a1) of the body bits, two at a time
                    int wd = a0 >> 6;
                    const int wlast = (a1 - 1) >> 6;
                    auto word = [&](const int w) {
                        unsigned long long m = w == wdA ? wA : (w == wdB ? wB : q.bbits[w]);
                        if (w == (a0 >> 6)) m &= ~0ull << (a0 & 63);
                        if (w == wlast && (a1 & 63)) m &= (1ull << (a1 & 63)) - 1ull;
                        return m;
                    };
                    unsigned long long mb = word(wd);
                    auto next = [&]() -> int {
                        for (;;) {
                            if (mb) { const int b = __builtin_ctzll(mb); mb &= mb - 1ull; return (wd << 6) + b; }
                            if (wd >= wlast) return -1;
                            wd++;
                            mb = word(wd);
                        }
                    };
                    // NBX bodies' recurrences through a run, sample by sample: body 0's term, then body 1's, ... into each sum
                    auto run = [&](auto nbc, S (&st)[NB][8]) {
                        constexpr int NBX = decltype(nbc)::value;
                        S zr[NBX], zi[NBX], wr[NBX], wi[NBX], ms[NBX];
#pragma unroll
                        for (int b = 0; b < NBX; b++) { zr[b] = st[b][0]; zi[b] = st[b][1]; wr[b] = st[b][2]; wi[b] = st[b][3]; ms[b] = st[b][6]; }
#ifdef PVX_AB_SYNTH_NOLOOP
#pragma unroll
                        for (int b = 0; b < NBX; b++) a[0] += ms[b] * zr[b] + wr[b] * st[b][4] + wi[b] * st[b][5] + st[b][7];
#else
#pragma unroll
                        for (int k = 0; k < R; k++) {
#pragma unroll
                            for (int b = 0; b < NBX; b++) a[k] = fm(ms[b], zr[b], a[k]);                          // PVAnalysis.py:734-736
#pragma unroll
                            for (int b = 0; b < NBX; b++) {
                                const S t_ = fm(zr[b], wr[b], -(zi[b] * wi[b]));                                 // z *= w
                                zi[b] = fm(zr[b], wi[b], zi[b] * wr[b]);
                                zr[b] = t_;
                                const S u_ = fm(wr[b], st[b][4], fm(-wi[b], st[b][5], wr[b]));                   // w += w d
                                wi[b] = fm(wr[b], st[b][5], fm(wi[b], st[b][4], wi[b]));
                                wr[b] = u_;
                                ms[b] += st[b][7];
                            }
                        }
#endif
                    };
                    for (;;) {
                        int li[NB];
                        li[0] = next();
                        if (li[0] < 0) break;
#pragma unroll
                        for (int b = 1; b < NB; b++) li[b] = next();
                        S st[NB][8];
                        int nmax = 1;                                             // bodies of the wave's busiest lane (the same for all lanes)
#pragma unroll
                        for (int b = 0; b < NB; b++) {
#pragma unroll
                            for (int j = 0; j < 8; j++) st[b][j] = (S)0;          // (no body: amplitude zero -- its terms are +0)
                            if (li[b] >= 0) seed((const BodyRec*)lds + (li[b] - tile), st[b]);
                            if (b > 0 && __ballot(li[b] >= 0) != 0ull) nmax = b + 1;
                        }
                        if (nmax == 1) run(std::integral_constant<int, 1>{}, st);
                        else if (NB == 2 || nmax == 2) { if constexpr (NB >= 2) run(std::integral_constant<int, 2>{}, st); }
                        else run(std::integral_constant<int, NB>{}, st);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // ---- the sums leave the same way (the runs' places are fetched again: twelve registers not kept through the loop above)
#ifdef PVX_AB_SYNTH_EXIT1       // tools/ab: without the sums' way out (transposes through LDS and stores)
    if (q.wlen >= 0) { if (a[0] == (S)12345) q.w[0] = 1.0; return; }
#endif
    long long oo[4];
    int ll[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { oo[i] = s_o[wbase + 16 * i + (lane >> 2)]; const int l_ = s_len[wbase + 16 * i + (lane >> 2)]; ll[i] = l_ < 0 ? -l_ : l_; }
#pragma unroll
    for (int c8 = 0; c8 < R / 8; c8++) {
#pragma unroll
        for (int j = 0; j < 4; j++) *(double2*)(lds + lane * kLaneB + 16 * j) = make_double2((double)a[8 * c8 + 2 * j], (double)a[8 * c8 + 2 * j + 1]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const double2 v = *(const double2*)(lds + (16 * i + (lane >> 2)) * kLaneB + 16 * (lane & 3));
            const int k = 8 * c8 + 2 * (lane & 3);
            const long long o = oo[i] + k;
#ifdef PVX_AB_SYNTH_NOSTORE      // tools/ab: timing-only build without the waveform stores
            const bool v0 = k < ll[i] && o < q.wlen && q.wlen < 0, v1 = k + 1 < ll[i] && o + 1 < q.wlen && q.wlen < 0;
#else
            const bool v0 = k < ll[i] && o < q.wlen, v1 = k + 1 < ll[i] && o + 1 < q.wlen;
#endif
            if (v0 && v1 && (o & 1) == 0) *(double2*)(q.w + o) = v;     // (non-temporal: -1.5 % at hop 512, +50 % at a hop of 700 samples)
            else {
                if (v0) q.w[o] = v.x;
                if (v1) q.w[o + 1] = v.y;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- everything else, added to what k_synth_bodies stored: attacks, releases, and the bodies whose pieces change inside a
// run.  One thread per run of R samples [s, s + R) of a segment; a thread without work returns at once.
#ifndef PVX_EXTRAS_WAVES
#define PVX_EXTRAS_WAVES 3
#endif
// XB = false: attacks and releases; XB = true: the irregular bodies (launched only where a contribution's own break can
// differ from the launch-wide one: pvx_launch_synth)
template <int R, bool XB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PVX_EXTRAS_WAVES, PVX_EXTRAS_WAVES))) void k_synth_extras(SampK q) {
    // The slice's segments in chunks of CS = 2^xcs (16 on a long waveform; fewer on a short one, so that its few segments
    // still spread over the chip); workgroup b takes chunks b, b + gridDim.x, ...  The flags of 256 / CS chunks are fetched
    // in one go (k_synth_params wrote them long ago: a launch without work costs one load), the flagged segments of a
    // chunk are listed in LDS and their runs dealt to the threads.
    __shared__ int s_list[256];
    __shared__ int s_n[256];
    const int tid_ = threadIdx.x;
    const int cs = q.xcs, CS = 1 << cs, NCF = 256 >> cs;              // chunk size, chunks per fetch
    const int64_t nchunks = (q.nseg + CS - 1) >> cs;
#pragma unroll 1
    for (int64_t c0 = blockIdx.x; c0 < nchunks; c0 += (int64_t)gridDim.x * NCF) {
    __syncthreads();
    {
        // thread (i, j): segment j of chunk c0 + i gridDim.x
        const int i = tid_ >> cs, j = tid_ & (CS - 1);
        const int64_t ch = c0 + (int64_t)i * gridDim.x, sl_ = (ch << cs) + j;
        const bool f = ch < nchunks && sl_ < q.nseg && q.segflag[q.seg0 + sl_] == q.gen;
        // (lanes CS i' .. CS i' + CS - 1 of a wave are one chunk: its flagged segments in ascending order)
        const unsigned long long bal = __ballot(f);
        const unsigned mcs = (unsigned)(bal >> ((tid_ & 63) & ~(CS - 1))) & ((1u << CS) - 1u);
        if (f) s_list[(i << cs) + __popc(mcs & ((1u << j) - 1u))] = (int)(q.seg0 + sl_);
        if (j == 0) s_n[i] = __popc(mcs);
    }
    __syncthreads();
#pragma unroll 1
    for (int ic = 0; ic < NCF; ic++) {
    const int total = s_n[ic] * q.rps;
#pragma unroll 1
    for (int t = tid_; t < total; t += 256) {
    const int segl = t / q.rps;
    const int run = t - segl * q.rps;
    const int seg = s_list[(ic << cs) + segl];
    const int h = q.h, K = q.K;
    const int s = run * R;
    const int len = h - s < R ? h - s : R;
    const double ds = (double)s, ts = 0.5 * ds * (double)(s - 1);
    const int64_t o = (int64_t)seg * h + s;
    double* dst = q.w + o;
    double a[R];
    // (the attack / release launch leaves EVERY run of a flagged segment, zeros included: k_synth_bodies starts from them)
#pragma unroll
    for (int k = 0; k < R; k++) a[k] = (XB && k < len && o + k < q.wlen) ? dst[k] : 0.0;
    if constexpr (XB) {
        // the bodies of frame seg whose pieces change inside a run
        int f0 = seg, f1 = seg + 1;
        if (f0 < q.fb0) f0 = q.fb0;
        if (f1 > q.fb1) f1 = q.fb1;
        if (f1 > f0) {
            const int n0 = (f0 - q.fx0) * K, n1 = (f1 - q.fx0) * K;
            for_bits(q.xbits, n0, n1, [&](const int li) {
                const BodyRec* c = q.body + (li - (q.fb0 - q.fx0) * K);
                // leading samples of the run on the first piece of fsig (samples m < fmb; m = fmb sits on both) / of msig
                int nfa = c->fmb - s, nma = c->mmb - s;
                nfa = nfa < 0 ? 0 : (nfa > len ? len : nfa);
                nma = nma < 0 ? 0 : (nma > len ? len : nma);
                const int k1 = nfa < nma ? nfa : nma, k2 = nfa < nma ? nma : nfa;
#pragma unroll 1
                for (int part = 0; part < 3; part++) {
                    // each pair of pieces is a quadratic of its own: it is followed from s and added where it holds
                    const int klo = part == 0 ? 0 : (part == 1 ? k1 : k2);
                    const int khi = part == 0 ? k1 : (part == 1 ? k2 : len);
                    if (klo >= khi) continue;
                    const bool fa = klo < nfa, ma = klo < nma;
                    PVX_BODY_SEEDS(c, s, ds, ts, fa, ma)
#pragma unroll
                    for (int k = 0; k < R; k++) {
                        const double v = __builtin_fma(ms, zr, a[k]);
                        a[k] = (k >= klo && k < khi) ? v : a[k];
                        PVX_CMUL(zr, zi, wr, wi);
                        PVX_CROT(wr, wi, dr, di);
                        ms += dms;
                    }
                }
            });
        }
    } else {
        add_edges<R, double>(q, seg, s, a);
    }
#pragma unroll
    for (int k = 0; k < R; k++)
        if (k < len && o + k < q.wlen) dst[k] = a[k];
    }
    }
    }
}

// ---- workspace ---------------------------------------------------------------------------------------------------------
constexpr int64_t kSliceNodes = (int64_t)1 << 20;   // nodes of one params / samples launch pair: bounds the records' memory

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
struct WsLayout { size_t segflag, cursor, off, cf, cm, cr, body, att, rel, bb, xb, ab, rb, total; int64_t slice_segs; };
WsLayout ws_layout(int64_t F, int K, int64_t P, int EF, int64_t nseg_all) {
    WsLayout L;
    const int64_t N = F * K;
    int64_t slice = kSliceNodes / K - 2 * EF;
    if (slice < 64) slice = 64;
    const int64_t fb = slice < F ? slice : F;                                   // body frames of a slice
    const int64_t fx = (slice + 2 * EF) < F ? (slice + 2 * EF) : F;             // with the edges' reach
    size_t o = 0;
    L.cursor = o; o += 256;
    L.off = o; o += up256((size_t)P * 8);
    L.cf = o; o += up256((size_t)N * 8);
    L.cm = o; o += up256((size_t)N * 8);
    L.cr = o; o += up256((size_t)N * 8);
    L.body = o; o += up256((size_t)fb * K * sizeof(BodyRec));
    L.att = o; o += up256((size_t)fx * K * sizeof(EdgeRec));
    L.rel = o; o += up256((size_t)fx * K * sizeof(EdgeRec));
    const size_t words = (size_t)(fx * K) / 64 + 8;
    L.bb = o; o += up256(words * 8);
    L.xb = o; o += up256(words * 8);
    L.ab = o; o += up256(words * 8);
    L.rb = o; o += up256(words * 8);
    L.segflag = o; o += up256((size_t)(nseg_all + 16) * 4);
    L.total = o;
    L.slice_segs = slice;
    return L;
}

// calls without a caller-owned workspace (pvx_synth_dev / pvx_synth): one grow-only buffer per stream -- calls on one
// stream are ordered by the stream, calls on different streams must not share
struct StreamWs { void* p = nullptr; size_t cap = 0; unsigned gen = 0; };
std::mutex g_ws_mu;
std::map<std::pair<int, hipStream_t>, StreamWs> g_ws;               // (device, stream): the null stream of two devices is two streams

}  // namespace

size_t pvx_synth_ws_bytes(int64_t F, int K, int64_t P, int nfft, int hop_a, int hop_s, double edge) {
    const double dfr = 1. / (hop_a / (double)nfft) / 2.;
    const int edgsam = (int)(dfr * hop_s * edge);
    const int EF = edgsam > 0 ? (edgsam + hop_s - 1) / hop_s : 0;
    return ws_layout(F, K, P, EF, F + 2 + EF + 2).total;     // (segments of the longest waveform these frames can give)
}

int pvx_launch_synth(const SynthParams& p, hipStream_t s) {
    if (p.wlen <= 0) return PVX_OK;
    const int h = p.hop_s;
    const int64_t nseg_all = (p.wlen + h - 1) / h;
    if (nseg_all > 0x7fffffffLL) { pvx_set_error("too many output segments"); return PVX_ERR_INVALID; }
    if ((int64_t)p.F * p.K > ((int64_t)1 << 40)) { pvx_set_error("too many analysis points"); return PVX_ERR_INVALID; }
    SynthK q;
    q.f = p.f; q.mag = p.mag; q.realph = p.realph; q.pid = p.partial_id; q.pst = p.part_start; q.pln = p.part_len;
    q.F = p.F; q.P = p.P; q.N = p.F * p.K; q.K = p.K; q.h = h; q.minframes = p.minframes; q.no_phcor = p.no_phcor;
    {
        const double overlap = p.hop_a / (double)p.nfft;              // PVAnalysis.py:824
        q.sr = p.sr;
        q.dh = (double)h;
        q.fstep = p.sr / (double)p.nfft;                              // PVAnalysis.py:825
        q.dfr = 1. / overlap / 2.;                                    // PVAnalysis.py:687
        q.offf = q.dfr + .5;
        q.sc = kPi2 / p.sr;
        q.rsr = 1.0 / q.sr; q.rdh = 1.0 / q.dh; q.rfstep = 1.0 / q.fstep;
        q.edgsam = (int)(q.dfr * h * p.edge);                         // PVAnalysis.py:740
        const double dfr_s = (double)p.nfft / (double)p.hop_a / 2.;   // PVAnalysis.py:1055
        q.edgsamp = (int64_t)(p.edge * h * dfr_s);                    // PVAnalysis.py:1056
        q.EF = q.edgsam > 0 ? (q.edgsam + h - 1) / h : 0;
        q.vr = q.edgsam > 0 ? cos(kPi / (double)q.edgsam) : 1.0;
        q.vi = q.edgsam > 0 ? sin(kPi / (double)q.edgsam) : 0.0;
    }
    const int run_env = [] { const char* e = getenv("PVX_SYNTH_RUN"); return e ? atoi(e) : 0; }();           // tests
    // samples per thread: 32 (two sincos per 32 samples); a short waveform takes runs of 16 or 8 so that more of the chip works
    int R = (nseg_all * ((h + 31) / 32) >= 256 * 64) ? 32 : ((nseg_all * ((h + 15) / 16) >= 256 * 64) ? 16 : 8);
    if (run_env == 8 || run_env == 16 || run_env == 32) R = run_env;
    SampK k;
    k.edges_inline = 0; k.tail_first = 0;
    bool irregular = false;
    {
        // where the pieces of fsig / msig change inside a segment: np.interp's breakpoints are h (dfr + .5 + j) and h (dfr + j)
        // (PVAnalysis.py:701-702), i.e. at ceil(h frac(dfr + .5)) and ceil(h frac(dfr)) for every contribution that has a
        // break at all -- launch-wide positions (0 or h: none).  Runs never straddle them, so a run follows ONE quadratic.
        // (A contribution whose own break differs -- float rounding of a non-dyadic dfr -- is handled in the kernel, slower.)
        auto brk = [&](double off) { const double fr = off - floor(off); return fr == 0.0 ? h : (int)ceil((double)h * fr); };
        int b1 = brk(q.offf), b2 = brk(q.dfr);
        if (b1 > b2) { const int t = b1; b1 = b2; b2 = t; }
        // can a contribution's own break differ from these?  Not when h (off + j) is exact for every j: off with few
        // fractional bits (dfr = 1, 2, 4, 1.5 ...: every power-of-two nfft / hop); otherwise the rounding of the product decides
        auto dyadic = [&](double off) { const double t = ldexp(off, 20); return t == floor(t) && off < 1024.0; };
        irregular = !(dyadic(q.offf) && dyadic(q.dfr) && h < (1 << 20));
        if (getenv("PVX_SYNTH_NO_CUTS")) { b1 = b2 = h; irregular = true; } // tests: the pieces change inside runs -> k_synth_extras<, true>
        k.c1 = b1; k.c2 = b2;
        k.n0 = (b1 + R - 1) / R; k.n1 = (b2 - b1 + R - 1) / R;
        k.rps = k.n0 + k.n1 + (h - b2 + R - 1) / R;
        q.c1 = b1; q.c2 = b2; q.R = R;
    }
    q.rps = k.rps;
    const int RX = (R == 8) ? 8 : 16;                                         // k_synth_extras' runs (8 on a very short waveform, like k_synth_bodies)

    if (nseg_all > p.F + 2 + q.EF + 2) { pvx_set_error("waveform of %lld samples is longer than %lld frames can give", (long long)p.wlen, (long long)p.F); return PVX_ERR_SIZE; }
    const WsLayout L = ws_layout(p.F, p.K, p.P, q.EF, p.F + 2 + q.EF + 2);
    char* base = (char*)p.ws;
    unsigned* genp = p.ws_gen;                                        // the workspace's call counter (0: never used since it was allocated)
    if (!base) {
        std::lock_guard<std::mutex> lk(g_ws_mu);
        int dev = 0;
        (void)hipGetDevice(&dev);
        StreamWs& w = g_ws[std::make_pair(dev, s)];
        if (w.cap < L.total) {
            if (w.p) (void)hipFree(w.p);                              // (synchronises with the device: no kernel still uses it)
            w.p = nullptr; w.cap = 0; w.gen = 0;
            if (hipMalloc(&w.p, L.total) != hipSuccess) { pvx_set_error("hipMalloc(%zu) of the resynthesis workspace failed", L.total); w.p = nullptr; return PVX_ERR_ALLOC; }
            w.cap = L.total;
        }
        base = (char*)w.p;
        genp = &w.gen;
    } else if (p.ws_bytes < L.total || !genp) {
        pvx_set_error("resynthesis workspace of %zu bytes, %zu needed (and its call counter)", p.ws_bytes, L.total);
        return PVX_ERR_SIZE;
    }
    // segment flags carry the number of the call that set them: a fresh workspace is cleared once, then never again
    if (!p.skip_prepare) {
        if (*genp == 0 || *genp >= 0x7ffffff0u) {
            PVX_HIP_CHECK(hipMemsetAsync(base + L.segflag, 0, (size_t)(p.F + 2 + q.EF + 2 + 16) * 4, s));
            *genp = 0;
        }
        ++*genp;
    }
    q.gen = (int)*genp;
    q.cursor = (unsigned long long*)(base + L.cursor);
    q.segflag = (int*)(base + L.segflag); q.nseg_all = nseg_all;
    q.off = (long long*)(base + L.off);
    q.cf = (double*)(base + L.cf); q.cm = (double*)(base + L.cm); q.cr = (double*)(base + L.cr);
    q.body = (BodyRec*)(base + L.body); q.att = (EdgeRec*)(base + L.att); q.rel = (EdgeRec*)(base + L.rel);
    q.bbits = (unsigned long long*)(base + L.bb); q.xbits = (unsigned long long*)(base + L.xb); q.abits = (unsigned long long*)(base + L.ab); q.rbits = (unsigned long long*)(base + L.rb);
    q.w = p.w; q.wlen = p.wlen;
    q.fx0 = q.fx1 = q.fb0 = q.fb1 = 0; q.seg0 = 0; q.nseg = 0;

    // rows of at most 16 peaks: the closed forms straight from the analysis rows (k_synth_params_direct); wider rows through
    // the partial-major copy (searching a partial's slot in rows of 100 peaks costs more than copying its points once)
    const int WB = (int)ceil(q.dfr + 0.5) + 2;                        // points a closed form can need behind its node
    const bool direct = p.K <= kDirectMaxK && WB + 4 <= kDirectMaxWL && getenv("PVX_SYNTH_CSR") == nullptr;
    const int rows_cap = 256 / p.K + 2 + WB + 4;
    const size_t direct_lds = (size_t)rows_cap * p.K * 28 > (size_t)4 * 64 * kLaneB ? (size_t)rows_cap * p.K * 28 : (size_t)4 * 64 * kLaneB;   // (the rows, then the records' store staging)
    if (!p.skip_prepare) {
        if (!direct) {
            PVX_HIP_CHECK(hipMemsetAsync(q.cursor, 0, 8, s));
            hipLaunchKernelGGL(k_synth_alloc, dim3((unsigned)((p.P + 255) / 256)), dim3(256), 0, s, q);
            hipLaunchKernelGGL(k_synth_scatter, dim3((unsigned)((q.N + 255) / 256)), dim3(256), 0, s, q);
        }
    }
    // a slice of the segments (p.seg_count > 0: pvx_synth_resident launches the waveform in slices whose DMA to the host
    // runs under the next slice's kernel)
    if (p.seg0 < 0 || p.seg0 > nseg_all) { pvx_set_error("bad segment slice"); return PVX_ERR_INVALID; }
    const int64_t seg_end = (p.seg_count > 0 && p.seg0 + p.seg_count < nseg_all) ? p.seg0 + p.seg_count : nseg_all;
    const int64_t slice_env = [] { const char* e = getenv("PVX_SYNTH_SLICE"); return e ? atoll(e) : 0LL; }();   // tests: more slices
    const int64_t slice = (slice_env >= 1 && slice_env < L.slice_segs) ? slice_env : L.slice_segs;
    for (int64_t s0 = p.seg0; s0 < seg_end; s0 += slice) {
        const int64_t s1 = s0 + slice < seg_end ? s0 + slice : seg_end;
        q.seg0 = s0; q.nseg = s1 - s0;
        q.fb0 = s0 < p.F ? s0 : p.F; q.fb1 = s1 < p.F ? s1 : p.F;
        q.fx0 = s0 - q.EF > 0 ? s0 - q.EF : 0; if (q.fx0 > p.F) q.fx0 = p.F;
        q.fx1 = s1 + q.EF < p.F ? s1 + q.EF : p.F;
        const int64_t nloc = (q.fx1 - q.fx0) * p.K;
        if (nloc > 0) {
            if (direct) hipLaunchKernelGGL(k_synth_params_direct, dim3((unsigned)((nloc + 255) / 256)), dim3(256), direct_lds, s, q, WB, rows_cap);
            else hipLaunchKernelGGL(k_synth_params, dim3((unsigned)((nloc + 255) / 256)), dim3(256), 0, s, q);
        }
        k.body = q.body; k.att = q.att; k.rel = q.rel; k.bbits = q.bbits; k.xbits = q.xbits; k.abits = q.abits; k.rbits = q.rbits;
        k.w = q.w; k.wlen = q.wlen; k.seg0 = s0; k.nthreads = q.nseg * k.rps;
        const int64_t grid_blocks = (k.nthreads + kBodiesTB - 1) / kBodiesTB;
        k.segflag = q.segflag; k.gen = q.gen;
        k.fx0 = (int)q.fx0; k.fx1 = (int)q.fx1; k.fb0 = (int)q.fb0; k.fb1 = (int)q.fb1;
        k.K = p.K; k.h = h; k.EF = q.EF; k.edgsam = q.edgsam; k.vr = q.vr; k.vi = q.vi;
        // attacks and releases (and irregular bodies) first, into the flagged segments of w; k_synth_bodies starts from them
        // (the other way round, the few threads with something to add would wait for the other XCDs' L2s to give up what
        // k_synth_bodies wrote: ~8 us with nothing to hide behind; here those waits disappear among the bodies' work)
        SampK kx = k;
        kx.rps = (h + RX - 1) / RX;
        kx.nthreads = q.nseg * kx.rps;
        kx.nseg = q.nseg;
        // (three workgroups per CU fill the chip at this kernel's registers; a workgroup takes 16 segments at a time, fewer
        // when that would leave most of the chip without a chunk)
        int xcs = 4;
        while (xcs > 0 && (q.nseg >> xcs) < 512) xcs--;
        kx.xcs = xcs;
        const int64_t nchunks = (q.nseg + (1 << xcs) - 1) >> xcs;
        const int64_t xgrid = nchunks < 768 ? nchunks : 768;
        // (without irregular bodies k_synth_bodies adds the attacks and releases itself: one launch less in the dependent chain;
        // PVX_SYNTH_EDGES_KERNEL=1 keeps the launch of their own, for A/B runs and the tests of that path)
        k.edges_inline = (!irregular && getenv("PVX_SYNTH_EDGES_KERNEL") == nullptr) ? 1 : 0;
        {
            const int64_t tf = k.edges_inline && !getenv("PVX_SYNTH_NO_TAIL_FIRST") ? ((int64_t)(q.EF + 1) * k.rps + kBodiesTB - 1) / kBodiesTB + 1 : 0;
            k.tail_first = (int)(tf < grid_blocks ? tf : grid_blocks);
        }
        if (!k.edges_inline) {
            if (RX == 8) hipLaunchKernelGGL((k_synth_extras<8, false>), dim3((unsigned)xgrid), dim3(256), 0, s, kx);
            else hipLaunchKernelGGL((k_synth_extras<16, false>), dim3((unsigned)xgrid), dim3(256), 0, s, kx);
        }
        if (irregular) {
            if (RX == 8) hipLaunchKernelGGL((k_synth_extras<8, true>), dim3((unsigned)xgrid), dim3(256), 0, s, kx);
            else hipLaunchKernelGGL((k_synth_extras<16, true>), dim3((unsigned)xgrid), dim3(256), 0, s, kx);
        }
        const dim3 grid((unsigned)grid_blocks);
        // a wave's 64 runs span (64 + rps - 1) / rps + 1 segments at most: the tile holds their K slots each (within 16 KB)
        {
            const int64_t segs = (64 + k.rps - 1) / k.rps + 1;
            int64_t need = segs * p.K;
            need = need < kTile ? kTile : (need > 128 ? 128 : ((need + 7) & ~(int64_t)7));
            k.tile = (int)need;
        }
        const size_t blds = (size_t)k.tile * sizeof(BodyRec) * (kBodiesTB / 64);
        if (p.f32_samples) {
            if (R == 8) hipLaunchKernelGGL((k_synth_bodies<8, float>), grid, dim3(kBodiesTB), blds, s, k);
            else if (R == 16) hipLaunchKernelGGL((k_synth_bodies<16, float>), grid, dim3(kBodiesTB), blds, s, k);
            else hipLaunchKernelGGL((k_synth_bodies<32, float>), grid, dim3(kBodiesTB), blds, s, k);
        } else {
        if (R == 8) hipLaunchKernelGGL(k_synth_bodies<8>, grid, dim3(kBodiesTB), blds, s, k);
        else if (R == 16) hipLaunchKernelGGL(k_synth_bodies<16>, grid, dim3(kBodiesTB), blds, s, k);
        else hipLaunchKernelGGL(k_synth_bodies<32>, grid, dim3(kBodiesTB), blds, s, k);
        }
    }
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}
