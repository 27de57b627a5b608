// pvx_wave.h -- wave64-level building blocks shared by the analysis kernels (k_peaks.hip,
// k_fused.hip): LDS hand-off inside one wave, ballot prefix, DPP reductions, the PeakFinder core
// (pypevoc/PeakFinder.py:155-194, 113-136) and the per-peak phase-vocoder arithmetic
// (pypevoc/PVAnalysis.py:133-148, 187-207).
#pragma once
#ifndef PVX_THIN_FROM
#define PVX_THIN_FROM 96    // candidates per row (segment) from which the list is thinned before it is written (peak_scan_block_thin)
#endif

#include <float.h>
#include <math.h>

#include "pvx_internal.h"

namespace pvxw {

constexpr double kPi = 3.141592653589793238462643383279502884;
constexpr double kPi2 = 2.0 * kPi;   // PV.py:45

__device__ __forceinline__ void wave_sync() {
    // LDS hand-off between lanes of ONE wave.  LDS operations of a wave execute in order; what is
    // needed is (a) that the compiler does not move LDS accesses across this point and (b) that the
    // LDS queue has drained.  Deliberately NOT a workgroup fence: that would also wait for every
    // outstanding global load/store (vmcnt(0)) and serialise the prefetch of the next frame.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// the lane index, computed on the spot (v_mbcnt): in rarely executed branches of a long frame loop, using the kernel's
// `lane` variable keeps it (and what the compiler derives from it) alive across the whole loop -- as spilled registers
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

__device__ __forceinline__ int lane_prefix(unsigned long long bal) {   // set bits of bal below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
}

// ---- DPP reductions: 4 in-row steps (quad xor 1, xor 2, half mirror, mirror), then one readlane
// per row of 16.  Result is wave-uniform.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    // all four permutations used below write every lane: no "old" value to materialise
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float rl_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ double rl_d(double v, int l) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// three-operand maximum / minimum (one instruction; the compiler only forms them now and then)
__device__ __forceinline__ float max3f(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float min3f(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

#define PVX_ROW_REDUCE(v, OP, DPP)            \
    v = OP(v, DPP<0xB1>(v));                  \
    v = OP(v, DPP<0x4E>(v));                  \
    v = OP(v, DPP<0x141>(v));                 \
    v = OP(v, DPP<0x140>(v));
__device__ __forceinline__ float wave_max(float v) {
    PVX_ROW_REDUCE(v, fmaxf, dpp_f)
    return fmaxf(fmaxf(rl_f(v, 0), rl_f(v, 16)), fmaxf(rl_f(v, 32), rl_f(v, 48)));
}
__device__ __forceinline__ float wave_min(float v) {
    PVX_ROW_REDUCE(v, fminf, dpp_f)
    return fminf(fminf(rl_f(v, 0), rl_f(v, 16)), fminf(rl_f(v, 32), rl_f(v, 48)));
}
__device__ __forceinline__ double dmax(double a, double b) { return a > b ? a : b; }
__device__ __forceinline__ double dmin(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double dadd(double a, double b) { return a + b; }
__device__ __forceinline__ double wave_max(double v) {
    PVX_ROW_REDUCE(v, dmax, dpp_d)
    return dmax(dmax(rl_d(v, 0), rl_d(v, 16)), dmax(rl_d(v, 32), rl_d(v, 48)));
}
__device__ __forceinline__ double wave_min(double v) {
    PVX_ROW_REDUCE(v, dmin, dpp_d)
    return dmin(dmin(rl_d(v, 0), rl_d(v, 16)), dmin(rl_d(v, 32), rl_d(v, 48)));
}
__device__ __forceinline__ double wave_sum(double v) {
    PVX_ROW_REDUCE(v, dadd, dpp_d)
    return (rl_d(v, 0) + rl_d(v, 16)) + (rl_d(v, 32) + rl_d(v, 48));
}

// The same reductions for the three quantities every spectrum row ends with -- max |X|^2, min |X|^2 (non-negative floats:
// their bit patterns order like the values, so the steps are integer maxima / minima that the compiler folds into ONE
// DPP-modified instruction each, v_max_u32_dpp, instead of v_mov_dpp + a canonicalising v_max + v_max) and the float64
// energy -- with the rows joined by two row-broadcast steps (rows 1, 3 take row 0 / 2's lane 15, rows 2, 3 take lane 31)
// and ONE readlane of lane 63 instead of four readlanes and a max3 per quantity.  Written as one function so that the
// three dependent chains interleave: each fills the others' DPP hazard slots (s_nop before) and issue stalls.
// The sum associates exactly as wave_sum does: xor 1, xor 2, half mirror, mirror, (row 0 + row 1) + (row 2 + row 3).
template <int CTRL> __device__ __forceinline__ unsigned dpp_u(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL, int ROWS> __device__ __forceinline__ unsigned dpp_keep_u(unsigned v) {      // rows not in ROWS keep v
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWS, 0xf, false);
}
template <int CTRL, int ROWS> __device__ __forceinline__ double dpp_keep_d(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp((int)b, (int)b, CTRL, ROWS, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), CTRL, ROWS, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ unsigned umax_(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned umin_(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ void wave_max_min_sum_nn(float lmax, float lmin, double lsum, float& mx, float& mn, double& sm) {
    unsigned a = __float_as_uint(lmax), b = __float_as_uint(lmin);
    double s = lsum;
#define PVX_STEP3(CTRL) a = umax_(a, dpp_u<CTRL>(a)); b = umin_(b, dpp_u<CTRL>(b)); s = s + dpp_d<CTRL>(s);
    PVX_STEP3(0xB1) PVX_STEP3(0x4E) PVX_STEP3(0x141) PVX_STEP3(0x140)
#undef PVX_STEP3
    // rows 1 and 3 take the row before them, then rows 2 and 3 the first half: lane 63 holds everything
    a = umax_(a, dpp_keep_u<0x142, 0xa>(a)); b = umin_(b, dpp_keep_u<0x142, 0xa>(b));
    const double s1 = dpp_keep_d<0x142, 0xa>(s);
    s = s + s1;                                                      // (rows 0 and 2 add themselves: never read)
    a = umax_(a, dpp_keep_u<0x143, 0xc>(a)); b = umin_(b, dpp_keep_u<0x143, 0xc>(b));
    const double s2 = dpp_keep_d<0x143, 0xc>(s);
    s = s + s2;
    mx = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)a, 63));
    mn = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)b, 63));
    sm = rl_d(s, 63);
}

template <typename T> struct Key;
template <> struct Key<float> {
    using type = unsigned int;
    static constexpr int TOP = 30;   // scores are >= 0: the sign bit is never set
    static __device__ __forceinline__ type of(float v) { return __float_as_uint(v); }
};
template <> struct Key<double> {
    using type = unsigned long long;
    static constexpr int TOP = 62;
    static __device__ __forceinline__ type of(double v) { return (unsigned long long)__double_as_longlong(v); }
};

// ---------------------------------------------------------------------------------------------
// PeakFinder core on one row held in LDS (PF.py:155-194 semantics, restated without the K-round
// arg-max loop).  pkmskamp (PF.py:166-167) is y-miny at interior local maxima and 0 elsewhere; the
// reference repeatedly takes the arg-max (first index on ties) while it exceeds th = minamp-miny and
// fewer than npeaks have been taken, then sorts the positions.  Equivalent statement used here:
//   candidates  = local maxima with score > th                    (score > 0 always for a maximum)
//   selected    = the npeaks best candidates by (score desc, index asc), all of them if fewer;
//   if th < 0 the zeros of pkmskamp qualify too: after ALL maxima, non-maximum interior bins are
//   taken in ascending index order until npeaks are selected.
// Output: out[0..count) ascending bin indices (wave-uniform count).
//   y[n] row; cs[cap]/ci[cap] candidate scratch, cap >= n/2 + 1.
// peak_scan: candidates among bins [kbase, kbase + nscan) of the row y[0..n), compacted in ascending
// bin order into cs/ci; returns their number (wave-uniform).
//   NIT > 0: nscan == 64*NIT is known at compile time (fused kernels): the scan is unrolled and split
//   into a read phase (all LDS loads in flight at once) and a compaction phase, one LDS round trip
//   instead of one per 64 bins.
//   Y: the row -- a pointer, or any object whose operator[](bin) returns the value the search runs on; CI: the type of
//   the list's bin entries (unsigned short halves the list); CS = false: no score list is kept (cs unused), the scores
//   are re-derived from the row where they are needed -- together they cut a wave's LDS from 14 to 10 bytes per bin,
//   which is what decides how many waves of k_phase_peaks a CU holds at nfft 4096 and up.
template <typename T, int NIT = 0, bool CS = true, typename Y, typename CI>
__device__ __forceinline__ int peak_scan(Y y, int kbase, int nscan, int n, T miny, double th, T* cs, CI* ci, int lane) {
    int C = 0;
    if constexpr (NIT > 0) {
        const float thf = __double2float_rd(th);
        T sv[NIT];
        unsigned long long bals[NIT];
        // read phase: unconditional (index-clamped) loads so that all of them are in flight together
        T ya[NIT], yb[NIT], yc[NIT];
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int k = kbase + i * 64 + lane;
            const int kc = k < n ? k : n - 1;
            ya[i] = y[kc > 0 ? kc - 1 : 0];
            yb[i] = y[kc];
            yc[i] = y[kc < n - 1 ? kc + 1 : n - 1];
        }
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int k = kbase + i * 64 + lane;
            const T s = (T)(yb[i] - miny);
            // (double)s > th  <=>  s > thf with thf = th rounded DOWN to float (the next float above
            // thf is already > th), so the float32 instantiation needs no float64 compare per bin
            bool above;
            if constexpr (sizeof(T) == 4) above = s > thf; else above = (double)s > th;
            const bool cand = ((int)(k >= 1) & (int)(k <= n - 2) & (int)(ya[i] < yb[i]) & (int)(yb[i] >= yc[i]) & (int)above) != 0;
            sv[i] = s;
            bals[i] = __ballot(cand);
        }
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const unsigned long long bal = bals[i];
            if (bal != 0ull) {                                       // wave-uniform
                if ((bal >> lane) & 1ull) { const int pos = C + lane_prefix(bal); if constexpr (CS) cs[pos] = sv[i]; ci[pos] = (CI)(kbase + i * 64 + lane); }
                C += __popcll(bal);
            }
        }
    } else {
        for (int k0 = 0; k0 < nscan; k0 += 64) {
            const int k = kbase + k0 + lane;
            bool cand = false;
            T s = (T)0;
            if (k0 + lane < nscan && k >= 1 && k <= n - 2) {
                const T a = y[k - 1], b = y[k], c = y[k + 1];
                s = (T)(b - miny);
                cand = (a < b) && (b >= c) && ((double)s > th);
            }
            const unsigned long long bal = __ballot(cand);
            if (cand) { const int pos = C + lane_prefix(bal); if constexpr (CS) cs[pos] = s; ci[pos] = (CI)k; }
            C += __popcll(bal);
        }
    }
    return C;
}

// Exact radix select on a candidate list of at most 64 * NCH entries with the keys in REGISTERS (lane owns entries
// lane + 64 j): two bits per round -- three trial values whose counts come from independent compares, ballots and
// scalar popcounts -- and the loop stops as soon as exactly npeaks keys are at or above the prefix.  The list-resident
// loop below (one bit per round, the entries beyond the first 128 re-read from LDS every round) is what remains for
// longer lists; on a recording (80-300 candidates per frame at nfft 2048) it was most of a float64 frame's time.
template <typename T, int NCH, bool CS = true, typename Y, typename CI>
__device__ __forceinline__ int peak_radix_list(Y y, const T* cs, const CI* ci, int* out, int npeaks, int C, int lane, T miny) {
    using K = Key<T>;
    using KT = typename K::type;
    KT key[NCH];
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        const int c = lane + 64 * j;
        T sc;
        if constexpr (CS) sc = cs[c < C ? c : 0]; else sc = (T)(y[(int)ci[c < C ? c : 0]] - miny);
        key[j] = (c < C) ? K::of(sc) : (KT)0;
    }
    KT prefix = 0;
    for (int bit = K::TOP; bit >= 0; bit -= 2) {                    // TOP is even: rounds take bits (bit, bit - 1); the last one bit 0 alone
        const int lo = bit >= 1 ? bit - 1 : 0;
        const KT t1 = prefix | ((KT)1 << lo);
        const KT t2 = bit >= 1 ? (prefix | ((KT)2 << lo)) : ~(KT)0;
        const KT t3 = bit >= 1 ? (prefix | ((KT)3 << lo)) : ~(KT)0;
        int c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            c1 += __popcll(__ballot(key[j] >= t1));
            c2 += __popcll(__ballot(key[j] >= t2));
            c3 += __popcll(__ballot(key[j] >= t3));
        }
        int cnt = -1;
        if (c3 >= npeaks) { prefix = t3; cnt = c3; }
        else if (c2 >= npeaks) { prefix = t2; cnt = c2; }
        else if (c1 >= npeaks) { prefix = t1; cnt = c1; }
        if (cnt == npeaks) break;                                    // exactly the keys >= prefix: nothing left to resolve
    }
    // prefix = key of the npeaks-th best (or a lower bound that exactly npeaks keys reach); strictly greater ones
    // all go, ties in list (= bin) order
    int ngt = 0;
#pragma unroll
    for (int j = 0; j < NCH; j++) ngt += __popcll(__ballot(key[j] > prefix));
    // (after an early stop every key >= prefix is wanted: treat them all as "ties" of the prefix class)
    int tc = 0, cnt = 0;
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        const int c = lane + 64 * j;
        const bool valid = c < C;
        const bool ge = valid && (key[j] >= prefix);
        const bool gt = valid && (key[j] > prefix);
        const bool tie = ge && !gt;
        const unsigned long long bt = __ballot(tie);
        const bool take = gt || (tie && (tc + lane_prefix(bt)) < npeaks - ngt);
        const unsigned long long bk = __ballot(take);
        if (take) out[cnt + lane_prefix(bk)] = (int)ci[c];
        tc += __popcll(bt);
        cnt += __popcll(bk);
    }
    wave_sync();
    return cnt;
}

// peak_pick: the npeaks best of the C candidates in cs/ci (list in ascending bin order) -> out[],
// ascending bins; returns the count (wave-uniform).  th < 0 additionally admits the zeros of pkmskamp.
// Optional padded layout of a magnitude row in LDS: YP = 1 keeps 4 spare floats after every 64 so that
// the per-lane block reads of peak_scan_block below are bank-conflict free.
template <int YP> __device__ __forceinline__ int ymap(int k) { return YP ? k + ((k >> 6) << 2) : k; }

//   BIG: also instantiate the register-resident select for lists of up to 2048 entries (rows of nfft >= 4096)
template <typename T, int YP = 0, bool CS = true, bool BIG = false, typename Y, typename CI>
__device__ __forceinline__ int peak_pick(Y y, T* cs, CI* ci, int* out, int n, int npeaks, int C, double th, int lane, T miny = (T)0) {
    auto score = [&](int c) -> T { if constexpr (CS) return cs[c]; else return (T)(y[ymap<YP>((int)ci[c])] - miny); };
    if (C <= npeaks) {
        if (th < 0.0 && C < npeaks) {
            // zeros of pkmskamp are above the (negative) threshold: all maxima, then the first
            // non-maximum interior bins, emitted together in ascending bin order
            const int need = npeaks - C;
            int zc = 0, cnt = 0;
            for (int k0 = 0; k0 < n && cnt < npeaks; k0 += 64) {
                const int k = k0 + lane;
                bool ismax = false, isz = false;
                if (k >= 1 && k <= n - 2) {
                    const T a = y[ymap<YP>(k - 1)], b = y[ymap<YP>(k)], c = y[ymap<YP>(k + 1)];
                    ismax = (a < b) && (b >= c);
                    isz = !ismax;
                }
                const unsigned long long bz = __ballot(isz);
                const bool take = ismax || (isz && (zc + lane_prefix(bz)) < need);
                const unsigned long long bt = __ballot(take);
                if (take) out[cnt + lane_prefix(bt)] = k;
                zc += __popcll(bz);
                cnt += __popcll(bt);
            }
            wave_sync();
            return cnt;
        }
        for (int c = lane; c < C; c += 64) out[c] = (int)ci[c];
        wave_sync();
        return C;
    }
    if (C <= 64) {
        // ---- a few more candidates than wanted (the usual case): every lane owns one candidate and
        // counts the candidates that beat it, broadcast one by one with readlane (no LDS round trip).
        // "beats" = larger score, or equal score and lower bin (np.argmax takes the first maximum).
        const T mys = (lane < C) ? score(lane) : (T)0;
        const int myi = (lane < C) ? (int)ci[lane] : 0;
        int rank = 0;
        for (int j = 0; j < C; ++j) {
            T sj;
            if constexpr (sizeof(T) == 4) sj = rl_f(mys, j); else sj = rl_d(mys, j);
            rank += (sj > mys || (sj == mys && j < lane)) ? 1 : 0;
        }
        const bool take = (lane < C) && (rank < npeaks);
        const unsigned long long bk = __ballot(take);
        if (take) out[lane_prefix(bk)] = myi;                        // list order = ascending bin
        wave_sync();
        return __popcll(bk);
    }
    // ---- C > 64: exact radix select of the npeaks-th largest score (bits of a non-negative
    // float order like unsigned integers).  Ballot + popcount only.
    if (C <= 128) return peak_radix_list<T, 2, CS>(y, cs, ci, out, npeaks, C, lane, miny);
    if (C <= 192) return peak_radix_list<T, 3, CS>(y, cs, ci, out, npeaks, C, lane, miny);
    if (C <= 320) return peak_radix_list<T, 5, CS>(y, cs, ci, out, npeaks, C, lane, miny);
    if (C <= 512) return peak_radix_list<T, 8, CS>(y, cs, ci, out, npeaks, C, lane, miny);
    if constexpr (BIG) {
        if (C <= 768) return peak_radix_list<T, 12, CS>(y, cs, ci, out, npeaks, C, lane, miny);
        if (C <= 1280) return peak_radix_list<T, 20, CS>(y, cs, ci, out, npeaks, C, lane, miny);
        if (C <= 2048) return peak_radix_list<T, 32, CS>(y, cs, ci, out, npeaks, C, lane, miny);
    }
    using K = Key<T>;
    using KT = typename K::type;
    const KT k0r = (lane < C) ? K::of(score(lane)) : (KT)0;            // first two per lane live in registers
    const KT k1r = (lane + 64 < C) ? K::of(score(lane + 64)) : (KT)0;
    KT prefix = 0;
    for (int bit = K::TOP; bit >= 0; --bit) {
        const KT trial = prefix | ((KT)1 << bit);
        int cnt = __popcll(__ballot(k0r >= trial)) + __popcll(__ballot(k1r >= trial));
        for (int c = lane + 128; c - lane < C; c += 64) {           // wave-uniform trip count
            const bool p = (c < C) && (K::of(score(c < C ? c : 0)) >= trial);
            cnt += __popcll(__ballot(p));
        }
        if (cnt >= npeaks) {
            prefix = trial;
            if (cnt == npeaks) break;                                // exactly the keys >= trial: nothing left to resolve
        }
    }
    // prefix = key of the npeaks-th best; strictly greater ones all go, ties in index order
    int ngt = __popcll(__ballot(k0r > prefix)) + __popcll(__ballot(k1r > prefix));
    for (int c = lane + 128; c - lane < C; c += 64) ngt += __popcll(__ballot((c < C) && (K::of(score(c < C ? c : 0)) > prefix)));
    const int need = npeaks - ngt;
    int tc = 0, cnt = 0;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        const KT key = (c < C) ? K::of(score(c < C ? c : 0)) : (KT)0;
        const bool tie = (c < C) && (key == prefix);
        const unsigned long long bt = __ballot(tie);
        const bool take = (c < C) && ((key > prefix) || (tie && (tc + lane_prefix(bt)) < need));
        const unsigned long long bk = __ballot(take);
        if (take) out[cnt + lane_prefix(bk)] = (int)ci[c];
        tc += __popcll(bt);
        cnt += __popcll(bk);
    }
    wave_sync();
    return cnt;
}

// threshold of PF.py:69-70, 174: th = minamp - miny with "if not self.minamp: minamp = min(y)"
template <typename T>
__device__ __forceinline__ double peak_threshold(double minamp_in, bool have_minamp, T miny) {
    const double minamp = (have_minamp && minamp_in != 0.0) ? minamp_in : (double)miny;
    return minamp - (double)miny;
}

//   NS > 0: n == NS is known at compile time (fused kernel)
// peak_select_th: a bin qualifies when y - miny > th; y may be any monotone function of the magnitudes
// as long as th and miny are in the same domain (the float32 kernels search |X|^2)
template <typename T, int NS = 0>
__device__ __forceinline__ int peak_select_th(const T* y, T* cs, int* ci, int* out, int n, int npeaks, double th, T miny, int lane) {
    if (n < 3) return 0;
    const int C = peak_scan<T, (NS + 63) / 64>(y, 0, n, n, miny, th, cs, ci, lane);
    wave_sync();
    return peak_pick<T>(y, cs, ci, out, n, npeaks, C, th, lane);
}
template <typename T, int NS = 0>
__device__ __forceinline__ int peak_select(const T* y, T* cs, int* ci, int* out, int n, int npeaks, double minamp_in,
                                  bool have_minamp, T miny, int lane) {
    return peak_select_th<T, NS>(y, cs, ci, out, n, npeaks, peak_threshold<T>(minamp_in, have_minamp, miny), miny, lane);
}

// peak_scan_block: the same candidate list as peak_scan for a float row of n = 64*R bins held in the
// padded layout (ymap<1>), with lane l owning the R CONSECUTIVE bins R*l .. R*l+R-1: R/4 16-byte LDS
// reads per lane instead of 3*R 4-byte ones, and everything after that is integer VALU work on the
// float bit patterns (magnitudes are >= +0, so they order like their bits):
//   rise_i  = sign(bits(y[k-1]) - bits(y[k]))            one subtract per bin boundary; "y[k] >= y[k+1]"
//                                                         of bin k is the complement of rise_{k+1}
//   above_i = sign(bits(thf) - bits(y[k] - miny))        the same float subtraction as peak_scan
//   cand_i  = rise_i & ~rise_{i+1} & above_i             one 3-input bit operation
// and the sign bits are funnel-shifted into a per-lane 16-bit mask.  No compare results travel through
// SGPRs and there is no branch: with one wave per SIMD those round trips and taken branches cost far
// more than the arithmetic.  List positions come from a ballot prefix over the per-lane counts
// (<= R/2 <= 8); every lane then stores its R bins unconditionally, non-candidates into a per-lane
// trash slot (ci[trash + lane]).  Only the bins are listed; scores are re-read where they are needed.
// Two halves so that several waves can scan segments of one row and write one list (k_fused_mw.hip):
//   peak_block_masks  lane owns bins kbase + R*lane .. + R-1 of the n-bin row: candidate bit mask `m`,
//                     list position of its first candidate within this wave `pos`; returns the wave's
//                     candidate count (wave-uniform)
//   peak_block_write  stores the lane's bins at base + pos.. (non-candidates into the trash slot `tr`)
template <int R, int YP>
__device__ __forceinline__ int peak_block_masks(const float* y, int kbase, int n, float miny, double th, int lane,
                                                unsigned& m_out, int& pos_out) {
    static_assert(R % 4 == 0 && R <= 16, "block scan handles 4, 8 or 16 bins per lane");
    const float thf = __double2float_rd(th);                         // see peak_scan
    // thf < 0: every score (>= 0) is above it; -1 keeps the integer subtraction below from wrapping
    const int thb = thf < 0.f ? -1 : __float_as_int(thf);
    const int k0 = kbase + R * lane;
    float v[R];
#pragma unroll
    for (int j = 0; j < R / 4; j++) {
        const float4 q = *(const float4*)(y + ymap<YP>(k0 + 4 * j));
        v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
    }
    // first lane of the row: "left" is bin 0 itself, so bin 0 never rises; last lane: "right" is bin n-1
    const float left = y[ymap<YP>(k0 > 0 ? k0 - 1 : 0)];
    const float right = y[ymap<YP>(k0 + R < n ? k0 + R : n - 1)];
    int rise[R + 1];                                                 // sign bit set: y[k-1] < y[k]
    rise[0] = __float_as_int(left) - __float_as_int(v[0]);
#pragma unroll
    for (int i = 1; i < R; i++) rise[i] = __float_as_int(v[i - 1]) - __float_as_int(v[i]);
    rise[R] = __float_as_int(v[R - 1]) - __float_as_int(right);
    unsigned m = 0;
#pragma unroll
    for (int i = R - 1; i >= 0; i--) {
        const int above = thb - __float_as_int(v[i] - miny);         // sign bit set: score > thf
        const unsigned t = (unsigned)(rise[i] & ~rise[i + 1] & above);
        m = (m << 1) | (t >> 31);
    }
    if (k0 + R == n) m &= ~(1u << (R - 1));                          // bin n-1 is not interior
    // exclusive prefix of the per-lane counts (<= 8: four bits)
    const int cnt = __popc(m);
    int pos = 0, C = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const unsigned long long bal = __ballot(((cnt >> b) & 1) != 0);
        pos += lane_prefix(bal) << b;
        C += __popcll(bal) << b;
    }
    m_out = m; pos_out = pos;
    return C;
}

template <int R, typename CI>
__device__ __forceinline__ void peak_block_write(CI* ci, int kbase, int lane, unsigned m, int pos, int tr) {
    const int k0 = kbase + R * lane;
    // one round per candidate of the busiest lane (1-3 on music, <= R/2): every lane pops its lowest
    // set bit; lanes that have run out write to their trash slot, so the body has no divergence
    while (__ballot(m != 0u) != 0ull) {                              // wave-uniform
        const bool has = m != 0u;
        const int i = __ffs((int)m) - 1;
        ci[has ? pos : tr] = (CI)(k0 + i);
        pos += has ? 1 : 0;
        m &= m - 1u;
    }
}

template <int R, typename CI>
__device__ __forceinline__ int peak_scan_block(const float* y, float miny, double th, CI* ci, int trash, int lane) {
    unsigned m; int pos;
    const int C = peak_block_masks<R, 1>(y, 0, 64 * R, miny, th, lane, m, pos);
    peak_block_write<R, CI>(ci, 0, lane, m, pos, trash + lane);
    return C;
}

// peak_scan_block for frames with MORE candidates than a wave has lanes (noise, most frames of a recording) when npeaks
// <= 16 of them are wanted: before the list is written, candidates that cannot be among the npeaks best are dropped, in
// registers.  The 16 quads of lanes each have a best candidate score -- 16 different candidates --, so T = the npeaks-th
// largest of those (every lane counts the quads that beat its own: 16 broadcasts) is reached by at least npeaks
// candidates: a lower bound of the npeaks-th largest score of the row, and everything the selection can take -- ties
// included, the scores being the very values it ranks (y - miny) -- is >= T.  On white noise 10 of 280 candidates survive
// on average, on a violin recording 37 of 119 (the smallest of 8 group maxima, cheaper, leaves 98 there: a spectrum that
// falls off has groups with tiny maxima); the list then fits one lane each and the frame takes the callers' short
// path.  The exact selection among the survivors is the callers' as before (same list order, same tie rule): the
// result does not depend on what T drops.
// The bound T of peak_scan_block_thin from the lanes' best candidate scores (0: no candidate).
// (Tried: the exact npeaks-th largest of the 64 lane bests -- one more broadcast per lane at or above T -- and thinning
// from 64 candidates on: both slower on a violin recording and on noise; a broadcast costs what ten candidates cost the
// radix select.)
// MAXK > 16 (k_fused_rev's instantiations for 8 < npks <= 24: the reference's default npks is 20): for 16 < npeaks <= MAXK the same
// bound from the 32 PAIRS of lanes -- 32 different candidates, twice the broadcasts; white noise at npks 20 keeps ~50 of 280
// candidates (+18 %; a recording 0 .. +5 %; at npks 32 the bound keeps too many to repay itself, and in the npks <= 8 kernels the
// longer loop body costs 1 %: hence a template parameter, 16 everywhere else).
template <int MAXK = 16>
__device__ __forceinline__ float thin_bound(float best, int npeaks) {
    float qb = fmaxf(best, dpp_f<0xB1>(best));                       // every lane: its pair's best
    int beaten = 0;                                                  // groups whose best beats this one's
    if (MAXK <= 16 || npeaks <= 16) {
        qb = fmaxf(qb, dpp_f<0x4E>(qb));                             // every lane: its quad's best
#pragma unroll
        for (int j = 0; j < 16; j++) beaten += (rl_f(qb, 4 * j) > qb) ? 1 : 0;
    } else {
#pragma unroll
        for (int j = 0; j < 32; j++) beaten += (rl_f(qb, 2 * j) > qb) ? 1 : 0;
    }
    return wave_min(beaten < npeaks ? qb : INFINITY);
}

template <int R, typename CI, int MAXK = 16>
__device__ __forceinline__ int peak_scan_block_thin(const float* y, float miny, double th, CI* ci, int trash, int lane, int npeaks) {
    static_assert(R % 4 == 0 && R <= 16, "block scan handles 4, 8 or 16 bins per lane");
    constexpr int n = 64 * R;
    const float thf = __double2float_rd(th);
    const int thb = thf < 0.f ? -1 : __float_as_int(thf);
    const int k0 = R * lane;
    float v[R];
#pragma unroll
    for (int j = 0; j < R / 4; j++) {
        const float4 q = *(const float4*)(y + ymap<1>(k0 + 4 * j));
        v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
    }
    const float left = y[ymap<1>(k0 > 0 ? k0 - 1 : 0)];
    const float right = y[ymap<1>(k0 + R < n ? k0 + R : n - 1)];
    int rise[R + 1];                                                 // sign bit set: y[k-1] < y[k]
    rise[0] = __float_as_int(left) - __float_as_int(v[0]);
#pragma unroll
    for (int i = 1; i < R; i++) rise[i] = __float_as_int(v[i - 1]) - __float_as_int(v[i]);
    rise[R] = __float_as_int(v[R - 1]) - __float_as_int(right);
    unsigned m = 0;
    float sc[R];                                                     // scores y - miny (>= 0)
#pragma unroll
    for (int i = R - 1; i >= 0; i--) {
        sc[i] = v[i] - miny;
        const int above = thb - __float_as_int(sc[i]);               // sign bit set: score > thf
        const unsigned t = (unsigned)(rise[i] & ~rise[i + 1] & above);
        m = (m << 1) | (t >> 31);
    }
    if (k0 + R == n) m &= ~(1u << (R - 1));                          // bin n-1 is not interior
    auto count = [&](unsigned mm, int& pos) -> int {
        const int cnt = __popc(mm);
        int C = 0;
        pos = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const unsigned long long bal = __ballot(((cnt >> b) & 1) != 0);
            pos += lane_prefix(bal) << b;
            C += __popcll(bal) << b;
        }
        return C;
    };
    int pos;
    int C = count(m, pos);
    // (th < 0 -- the threshold lies below the row's minimum, as on white noise -- changes nothing here: with more than 64
    // candidates the selection takes maxima only, and at least npeaks of them survive)
    // (its ~150 instructions pay from about a hundred candidates up, PVX_THIN_FROM: below that the callers' radix select over
    // two keys per lane is as cheap as this plus the ranking pass over the survivors.  While that pass broadcast its keys by
    // v_readlane the break-even was near 200 -- a violin recording's frames, ~120 candidates, ran 8 % slower thinned; with
    // the keys read from LDS they run 2 % faster at nfft 2048 / 4096, white noise at nfft 1024, ~140 candidates, 14 %)
    if (C > PVX_THIN_FROM && npeaks <= MAXK) {                                   // wave-uniform
        float best = 0.f;                                            // this lane's best candidate score (0: it has none)
#pragma unroll
        for (int i = 0; i < R; i++) best = fmaxf(best, ((m >> i) & 1u) ? sc[i] : 0.f);
        const float T = thin_bound<MAXK>(best, npeaks);
        unsigned keep = 0u;
#pragma unroll
        for (int i = R - 1; i >= 0; i--) keep = (keep << 1) | (sc[i] >= T ? 1u : 0u);
        m &= keep;
        C = count(m, pos);
    }
    peak_block_write<R, CI>(ci, 0, lane, m, pos, trash + lane);
    return C;
}

// peak_scan_block_thin for ONE SEGMENT of a row that several waves share (k_fused_team.hip): the wave's lanes own the
// 64 R consecutive bins from kbase on of the n-bin row y (padded layout); the list holds row bins.  The bound T is
// taken over the segment's own candidates: at least npeaks candidates of the segment reach it, so it is a lower bound
// of the npeaks-th largest score of the whole row too, and every wave may thin its segment on its own.
template <int R, typename CI>
__device__ __forceinline__ int peak_scan_seg_thin(const float* y, int kbase, int n, float miny, double th, CI* ci, int trash, int lane, int npeaks) {
    static_assert(R % 4 == 0 && R <= 16, "block scan handles 4, 8 or 16 bins per lane");
    const float thf = __double2float_rd(th);
    const int thb = thf < 0.f ? -1 : __float_as_int(thf);
    const int k0 = kbase + R * lane;
    float v[R];
#pragma unroll
    for (int j = 0; j < R / 4; j++) {
        const float4 q = *(const float4*)(y + ymap<1>(k0 + 4 * j));
        v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
    }
    const float left = y[ymap<1>(k0 > 0 ? k0 - 1 : 0)];
    const float right = y[ymap<1>(k0 + R < n ? k0 + R : n - 1)];
    int rise[R + 1];                                                 // sign bit set: y[k-1] < y[k]
    rise[0] = __float_as_int(left) - __float_as_int(v[0]);
#pragma unroll
    for (int i = 1; i < R; i++) rise[i] = __float_as_int(v[i - 1]) - __float_as_int(v[i]);
    rise[R] = __float_as_int(v[R - 1]) - __float_as_int(right);
    unsigned m = 0;
    float sc[R];                                                     // scores y - miny (>= 0)
#pragma unroll
    for (int i = R - 1; i >= 0; i--) {
        sc[i] = v[i] - miny;
        const int above = thb - __float_as_int(sc[i]);               // sign bit set: score > thf
        const unsigned t = (unsigned)(rise[i] & ~rise[i + 1] & above);
        m = (m << 1) | (t >> 31);
    }
    if (k0 + R == n) m &= ~(1u << (R - 1));                          // bin n-1 is not interior
    auto count = [&](unsigned mm, int& pos) -> int {
        const int cnt = __popc(mm);
        int C = 0;
        pos = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const unsigned long long bal = __ballot(((cnt >> b) & 1) != 0);
            pos += lane_prefix(bal) << b;
            C += __popcll(bal) << b;
        }
        return C;
    };
    int pos;
    int C = count(m, pos);
    if (C > PVX_THIN_FROM && npeaks <= 16) {                                   // wave-uniform (see peak_scan_block_thin)
        float best = 0.f;
#pragma unroll
        for (int i = 0; i < R; i++) best = fmaxf(best, ((m >> i) & 1u) ? sc[i] : 0.f);
        const float T = thin_bound(best, npeaks);
        unsigned keep = 0u;
#pragma unroll
        for (int i = R - 1; i >= 0; i--) keep = (keep << 1) | (sc[i] >= T ? 1u : 0u);
        m &= keep;
        C = count(m, pos);
    }
    peak_block_write<R, CI>(ci, kbase, lane, m, pos, trash + lane);
    return C;
}

// The dense-candidate branch of peak_pick_regs, kept out of line: it runs on noise-like frames only, and
// inlined its NCH-wide register arrays and unrolled loops weigh on the register allocation and code
// layout of the common path (measured: -4 % on harmonic input in the multi-wave kernels).
template <int NCH, int YP, typename CI, typename Y = const float*>
__device__ __forceinline__ int peak_radix_body(Y y, const CI* ci, int* out, int npeaks, int C,
                                               float miny, int lane) {
    int cb[NCH];
    unsigned key[NCH];
#pragma unroll
    for (int j = 0; j < NCH; j++) { const int c = lane + 64 * j; cb[j] = (int)ci[c < C ? c : 0]; }
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        const int c = lane + 64 * j;
        const float s = y[ymap<YP>(cb[j])] - miny;
        key[j] = (c < C) ? __float_as_uint(s) : 0u;                  // scores are >= 0: bits order like the values
    }
    // ---- exact radix select of the npeaks-th largest key, two bits per round (three trial values, their
    // counts from ballots + scalar popcounts: the compares of a round are independent, so the VALU ->
    // SGPR -> SALU latency is paid once per round, and half as many rounds means half as many taken loop
    // branches).  The loop stops as soon as exactly npeaks keys are at or above the prefix.
    unsigned prefix = 0;
    bool exact = false;
    for (int bit = 30; bit >= 0; bit -= 2) {
        const int lo = bit >= 1 ? bit - 1 : 0;
        const unsigned t1 = prefix | (1u << lo);
        const unsigned t2 = bit >= 1 ? (prefix | (2u << lo)) : 0xffffffffu;      // last round (bit 0 alone): one trial
        const unsigned t3 = bit >= 1 ? (prefix | (3u << lo)) : 0xffffffffu;
        int c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            c1 += __popcll(__ballot(key[j] >= t1));
            c2 += __popcll(__ballot(key[j] >= t2));
            c3 += __popcll(__ballot(key[j] >= t3));
        }
        int cnt = -1;
        if (c3 >= npeaks) { prefix = t3; cnt = c3; }
        else if (c2 >= npeaks) { prefix = t2; cnt = c2; }
        else if (c1 >= npeaks) { prefix = t1; cnt = c1; }
        if (cnt == npeaks) { exact = true; break; }                  // exactly the keys >= prefix: nothing left to resolve
    }
    int cnt = 0;
    if (exact) {
        // no ties to order: everything at or above the prefix goes, in list (= bin) order
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            const bool take = key[j] >= prefix;                      // key 0 ("no entry") is below any prefix >= 1
            const unsigned long long bk = __ballot(take);
            if (take) out[cnt + lane_prefix(bk)] = cb[j];
            cnt += __popcll(bk);
        }
        wave_sync();
        return cnt;
    }
    // prefix = key of the npeaks-th best; strictly greater ones all go, ties in list (= bin) order
    int ngt = 0;
#pragma unroll
    for (int j = 0; j < NCH; j++) ngt += __popcll(__ballot(key[j] > prefix));
    const int need = npeaks - ngt;
    int tc = 0;
#pragma unroll
    for (int j = 0; j < NCH; j++) {
        const bool valid = lane + 64 * j < C;
        const bool tie = valid && (key[j] == prefix);
        const unsigned long long bt = __ballot(tie);
        const bool take = valid && ((key[j] > prefix) || (tie && (tc + lane_prefix(bt)) < need));
        const unsigned long long bk = __ballot(take);
        if (take) out[cnt + lane_prefix(bk)] = cb[j];
        tc += __popcll(bt);
        cnt += __popcll(bk);
    }
    wave_sync();
    return cnt;
}

template <int NCH, int YP, typename CI, typename Y = const float*>
__device__ __attribute__((noinline)) int peak_radix_out(Y y, const CI* ci, int* out, int npeaks, int C,
                                                        float miny, int lane) {
    return peak_radix_body<NCH, YP, CI, Y>(y, ci, out, npeaks, C, miny, lane);
}
// out of line only where the arrays are big (NCH > 4: nfft >= 2048); the small kernels inline it -- a
// call makes them reserve stack / callee registers, which costs the 2-waves-per-SIMD variants ~4 %
template <int NCH, int YP, typename CI, bool INL = false, typename Y = const float*>
__device__ __forceinline__ int peak_radix_regs(Y y, const CI* ci, int* out, int npeaks, int C,
                                               float miny, int lane) {
    // the select's cost is its register arrays' width (3 compares + ballots per 64 list entries and round, whether
    // the entries exist or not): a recording's frames have 80-200 candidates, white noise ~280 of the 512 the list
    // can hold at nfft 2048 -- take the narrowest instantiation that covers C
    if constexpr (NCH > 2) { if (C <= 128) return peak_radix_body<2, YP, CI, Y>(y, ci, out, npeaks, C, miny, lane); }
    if constexpr (NCH > 3) { if (C <= 192) return peak_radix_body<3, YP, CI, Y>(y, ci, out, npeaks, C, miny, lane); }
    if constexpr (NCH > 5) { if (C <= 320) return peak_radix_body<5, YP, CI, Y>(y, ci, out, npeaks, C, miny, lane); }
    if constexpr (NCH <= 4 || INL) return peak_radix_body<NCH, YP, CI, Y>(y, ci, out, npeaks, C, miny, lane);
    else return peak_radix_out<NCH, YP, CI, Y>(y, ci, out, npeaks, C, miny, lane);
}

// peak_pick with the candidate scores and bins in REGISTERS: lane owns list entries c = lane + 64 j,
// j < NCH (NCH * 64 >= the list's capacity).  Same selection as peak_pick; the dense-candidate case
// (noise-like frames: hundreds of maxima above the threshold) is a radix select on register keys --
// 31 rounds of NCH compares + scalar popcounts, no LDS in the loop -- where the LDS-resident version
// paid a memory round trip per bit.
template <int NCH, int YP, typename CI, bool INL = false, typename Y = const float*>
__device__ __forceinline__ int peak_pick_regs(Y y, const CI* ci, int* out, int n, int npeaks, int C,
                                              double th, float miny, int lane) {
    if (C <= npeaks) {
        if (th < 0.0 && C < npeaks) {
            // zeros of pkmskamp are above the (negative) threshold: all maxima, then the first
            // non-maximum interior bins, emitted together in ascending bin order
            const int need = npeaks - C;
            int zc = 0, cnt = 0;
            for (int k0 = 0; k0 < n && cnt < npeaks; k0 += 64) {
                const int k = k0 + lane;
                bool ismax = false, isz = false;
                if (k >= 1 && k <= n - 2) {
                    const float a = y[ymap<YP>(k - 1)], b = y[ymap<YP>(k)], c = y[ymap<YP>(k + 1)];
                    ismax = (a < b) && (b >= c);
                    isz = !ismax;
                }
                const unsigned long long bz = __ballot(isz);
                const bool take = ismax || (isz && (zc + lane_prefix(bz)) < need);
                const unsigned long long bt = __ballot(take);
                if (take) out[cnt + lane_prefix(bt)] = k;
                zc += __popcll(bz);
                cnt += __popcll(bt);
            }
            wave_sync();
            return cnt;
        }
        for (int c = lane; c < C; c += 64) out[c] = (int)ci[c];
        wave_sync();
        return C;
    }
    // ---- more candidates than wanted: bins, then scores, of this lane's list entries (two LDS round
    // trips in all); key 0 marks "no entry" (a real entry's score can only be 0 when th < 0)
    if (C <= 64) {
        const int cb0 = (int)ci[lane < C ? lane : 0];
        const unsigned mine = (lane < C) ? __float_as_uint(y[ymap<YP>(cb0)] - miny) : 0u;   // scores >= 0: bits order like values
        // a few more candidates than wanted (the usual case): every lane owns one and counts the
        // candidates that beat it, broadcast one by one with readlane.  "beats" = larger score, or equal
        // score and lower bin (np.argmax takes the first maximum).
        // (one compare of the unique keys (score, 63 - lane), four list entries per trip: the lanes from C on hold score 0)
        const unsigned long long my64 = ((unsigned long long)mine << 32) | (unsigned)(63 - lane);
        int rank = 0;
        for (int j = 0; j < C; j += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const unsigned kj = (unsigned)__builtin_amdgcn_readlane((int)mine, j + u);
                const unsigned long long k64 = ((unsigned long long)kj << 32) | (unsigned)(63 - j - u);
                rank += (k64 > my64) ? 1 : 0;
            }
        }
        const bool take = (lane < C) && (rank < npeaks);
        const unsigned long long bk = __ballot(take);
        if (take) out[lane_prefix(bk)] = cb0;                        // list order = ascending bin
        wave_sync();
        return __popcll(bk);
    }
    return peak_radix_regs<NCH, YP, CI, INL, Y>(y, ci, out, npeaks, C, miny, lane);
}

// `th`: a bin qualifies when y - miny > th (the caller derives it from PF.py:60, 69-70; y may be any
// monotone function of the magnitudes as long as th and miny are in the same domain).
// ci must have 64 spare ints at [trash, trash + 64).
template <int R, typename CI>
__device__ __forceinline__ int peak_select_block(const float* y, CI* ci, int trash, int* out, int npeaks,
                                                 double th, float miny, int lane) {
    const int C = peak_scan_block<R, CI>(y, miny, th, ci, trash, lane);
    wave_sync();
    return peak_pick_regs<R / 2, 1, CI>(y, ci, out, 64 * R, npeaks, C, th, miny, lane);
}

// filter_by_salience(rad), sal = 0 (PF.py:126-134): keep unless any y in
// [max(p-rad,1), min(p+rad,n)] (clipped to the array) exceeds y[p]
template <typename T, int YP = 0, typename Y> __device__ __forceinline__ bool salient(Y y, int n, int p, int rad) {
    if (rad < 0) return true;
    const T v = y[ymap<YP>(p)];
    const int lo = p - rad > 1 ? p - rad : 1;
    int hi = p + rad < n ? p + rad : n;
    if (hi > n - 1) hi = n - 1;
    bool keep = true;
    if (rad <= 8) {
        // fixed trip count, no predicates: neighbours outside [lo, hi] are clamped INTO the window
        // (re-testing a window value cannot change an "any value larger" test), so the 17 loads
        // depend on nothing but p and are issued back to back; the compares follow
        T w[17];
#pragma unroll
        for (int d = -8; d <= 8; d++) {
            int j = p + (d < -rad ? -rad : (d > rad ? rad : d));
            j = j < lo ? lo : (j > hi ? hi : j);
            w[d + 8] = y[ymap<YP>(j)];
        }
        int bad = 0;
#pragma unroll
        for (int d = 0; d < 17; d++) bad |= (int)(w[d] > v);
        keep = (bad == 0);
    } else {
        for (int j = lo; j <= hi; j++) keep = keep && !(y[ymap<YP>(j)] > v);
    }
    return keep;
}

// filter_by_salience for up to 64 selected peaks with ALL lanes working: 8 lanes per peak, lane o of a
// group tests the two neighbours at distance min(o + 1, rad) (rad <= 8; indices clamped into the window
// like `salient` does), one ballot per pass of 8 peaks.  `pb`/`mine`: this lane's own peak (list entry
// eb + lane) and whether it exists; returns its keep flag.  sel[] = the selected bins.
template <int YP, typename Y = const float*>
__device__ __forceinline__ bool salient_groups(Y y, int n, const int* sel, int eb, int nsel, int rad, int lane) {
    const int e = eb + lane;
    if (rad < 0) return e < nsel;
    bool keep = false;
    const int left = nsel - eb;
    const int npass = ((left < 64 ? left : 64) + 7) >> 3;            // wave-uniform
    const int grp = lane >> 3;
    int d = (lane & 7) + 1;
    d = d > rad ? rad : d;
    for (int g = 0; g < npass; g++) {
        const int eq = eb + 8 * g + grp;
        const bool act = eq < nsel;
        const int pq = sel[act ? eq : eb];
        const int lo = pq - rad > 1 ? pq - rad : 1;
        int hi = pq + rad < n ? pq + rad : n;
        hi = hi > n - 1 ? n - 1 : hi;
        int j0 = pq - d, j1 = pq + d;
        j0 = j0 < lo ? lo : j0;                                      // lo <= pq <= hi always
        j1 = j1 > hi ? hi : j1;
        const float v = y[ymap<YP>(pq)], a = y[ymap<YP>(j0)], b = y[ymap<YP>(j1)];
        const bool bad = act && ((int)(a > v) | (int)(b > v)) != 0;
        const unsigned long long bal = __ballot(bad);
        const bool k = ((bal >> (8 * (lane & 7))) & 0xFFull) == 0ull;   // the group of peak 8 g + (lane & 7)
        keep = (grp == g) ? k : keep;
    }
    return keep && (e < nsel);
}

// ---------------------------------------------------------------------------------------------
// Per-peak phase-vocoder arithmetic (PV.py:187-207 with dphase2freq, PV.py:133-148).
// Inputs: bin, current (re, im), previous (pr, pi), s3 = 3-bin energy sum (PV.py:197-199).
// precision 32: float32 angles, float64 assembly around the exactly known bin centre;
// precision 64: the reference's operation order in float64.
struct PeakConst {
    double fstep, dt;
    int nfft, hop;
    const double* wfbin;
};

struct PeakOut {
    double freq, dfb, thisph, mag;
    float wu;         // precision 32: the float32 value freq is computed from (wire format 2, k_wire.hip): the unwrapped offset `best` of
                      // the ordinary case, or 8 + 3 quadrant + (m + 1) of a frame that follows a zero spectrum
    bool valid;
    bool nanph;       // the phase difference is NaN (x/0 with a zero real or imaginary part)
};

// wfbin[nbin] (PV.py:114-118) as the plan's table holds it, computed on the spot: the host's own float64 sequence
// (pvx_plan_create), so the same bits.  For the kernels whose per-peak pass would otherwise wait for a global load
// (`WFC`): a round trip of thousands of cycles per flush against ~25 float64 instructions.
__device__ __forceinline__ double wfbin_of(int nbin, const PeakConst& c) {
    const double fbin = (double)nbin * c.fstep;
    const double dthetabin = kPi2 * fbin * c.dt;
    return __builtin_nearbyint(dthetabin / kPi2) * kPi2;
}

template <typename T, bool WFC = false>
__device__ __forceinline__ PeakOut peak_math(int nbin, T re, T im, T pr, T pi, T s3, const PeakConst& c) {
    const double wfb = WFC ? wfbin_of(nbin, c) : c.wfbin[nbin];
    PeakOut o;
    bool nanph = false;
    if constexpr (sizeof(T) == 4) {
        const float tph = atan2f(im, re);                            // PV.py:188
        if (pr == 0.f && pi == 0.f) {
            // numpy: (a+bj)/(0+0j) = (a/0) + (b/0)j -> +-inf +-inf j, NaN when a or b is 0; angle()
            // is then +-pi/4, +-3pi/4 by quadrant (PV.py:171, 190: frame 0 and any frame that
            // follows an all-zero one).  That angle is the same float64 constant in the reference,
            // and with hop = nfft/8 it puts two unwrapping candidates at EXACTLY the same distance
            // from the bin centre, so the reference's own float64 rounding decides: follow its
            // arithmetic literally here (rare frames, cost irrelevant).
            nanph = (re == 0.f || im == 0.f || re != re || im != im);
            const double dphd = (re > 0.f) ? (im > 0.f ? kPi / 4 : -kPi / 4) : (im > 0.f ? 3 * kPi / 4 : -3 * kPi / 4);
            const int qi = (re > 0.f) ? (im > 0.f ? 0 : 1) : (im > 0.f ? 2 : 3);
            const double fb = (double)nbin * c.fstep;
            const double w0 = dphd + wfb;
            double bestabs = 0.0;
            int mi = -1;
            o.freq = 0.0; o.dfb = 0.0;
#pragma unroll
            for (int m = -1; m <= 1; m++) {
                const double fq = (w0 + kPi2 * (double)m) / c.dt / kPi2;
                const double df = fb - fq;
                const double a = fabs(df);
                if (m == -1 || a < bestabs) { o.freq = fq; o.dfb = df; bestabs = a; mi = m; }
            }
            o.wu = 8.f + (float)(qi * 3 + (mi + 1));
        } else {
            const float dph = atan2f(im * pr - re * pi, re * pr + im * pi);   // angle(fx * conj(old))
            nanph = dph != dph;
            // PV.py:140-147 in closed form: with cyc = nbin*hop/nfft (cycles the bin centre advances per
            // hop) and w = wfbin/2pi, candidate m has df*dt = cyc - w - dph/2pi - m; cyc - w is exact in
            // float64 and |cyc - w| <= 1/2
            const double cw = (double)nbin * (double)c.hop / (double)c.nfft - wfb / kPi2;
            const float u = (float)cw - dph * (float)(1.0 / kPi2);
            float best = u + 1.f, ab = fabsf(best);                      // m = -1
            if (fabsf(u) < ab) { best = u; ab = fabsf(u); }              // m = 0   (first minimum wins)
            if (fabsf(u - 1.f) < ab) { best = u - 1.f; }                 // m = +1
            const double fb = (double)nbin * c.fstep;
            o.freq = fb - (double)best / c.dt;
            o.wu = best;
            o.dfb = fb - o.freq;                                         // df = fbin - freq (PV.py:146), in the
                                                                         // reference's order: realph is then a
                                                                         // function of (nbin, freq, ph) alone,
                                                                         // which the result wire format relies on
        }
        o.thisph = (double)tph;
        o.mag = (double)sqrtf(s3);
    } else {
        const double dre = re, dim = im, dpr = pr, dpi = pi;
        o.wu = 0.f;
        o.thisph = atan2(dim, dre);                                  // PV.py:188
        double dph;
        if (dpr == 0.0 && dpi == 0.0) {
            nanph = (dre == 0.0 || dim == 0.0 || dre != dre || dim != dim);
            dph = (dre > 0.0) ? (dim > 0.0 ? kPi / 4 : -kPi / 4) : (dim > 0.0 ? 3 * kPi / 4 : -3 * kPi / 4);
        } else {
            dph = atan2(dim * dpr - dre * dpi, dre * dpr + dim * dpi);
            nanph = dph != dph;
        }
        // PV.py:140-147 literally: three unwrapping candidates, the one nearest the bin centre
        const double fb = (double)nbin * c.fstep;                    // PV.py:114
        const double w0 = dph + wfb;
        double bestabs = 0.0;
        o.freq = 0.0; o.dfb = 0.0;
#pragma unroll
        for (int m = -1; m <= 1; m++) {
            const double dphw = w0 + kPi2 * (double)m;
            const double fq = dphw / c.dt / kPi2;
            const double df = fb - fq;
            const double a = fabs(df);
            if (m == -1 || a < bestabs) { o.freq = fq; o.dfb = df; bestabs = a; }
        }
        o.mag = sqrt((double)s3);
    }
    o.nanph = nanph;
    o.valid = !nanph && (o.freq > 0.0);                              // PV.py:193 (NaN fails the test)
    return o;
}

}  // namespace pvxw
