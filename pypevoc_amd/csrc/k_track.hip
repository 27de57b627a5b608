// k_track.hip -- sinusoidal track building.  Replaces PV.toSinSum (pypevoc/PVAnalysis.py:299-322)
// = SinSum.add_frame for every frame (PVAnalysis.py:871-957) with add_empty_partial (819-830),
// get_partials_idx_ending_at_frame (984-994), RegPartial.append_point (616-626), dpitch2st (62-68).
//
// The reference is a sequential loop over frames whose only carried state is "which partials
// ended at frame fr-1" -- and that set is exactly the set of valid peaks (f > 0 and mag > 0) of
// frame fr-1.  So the greedy assignment of frame fr depends on rows fr-1 and fr only and all
// frames are linked in parallel:
//   k_track_links  one wave64 per frame: greedy nearest-in-semitones assignment, new peaks in
//                  descending magnitude against previous peaks in descending magnitude
//   k_scan_counts  exclusive scan of "partials created per frame" -> creation-order numbering
//   k_root_*       pointer jumping along the links: every peak learns the first point of its partial
//   k_assign_ids   partial_id / part_start / part_len (length written by the partial's last point)
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

#include "pvx_internal.h"

namespace {

__device__ inline void wave_sync_t() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// arg-min with first-index ties (np.argmin, PVAnalysis.py:920)
__device__ inline void wave_argmin(double& v, int& i) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        double u = __shfl_xor(v, o);
        int j = __shfl_xor(i, o);
        if (u < v || (u == v && j < i)) { v = u; i = j; }
    }
}

// LDS per wave: cm[K] cf[K] pm[K] pf[K] doubles | corder[K] porder[K] used[K] ints
__global__ __launch_bounds__(256) void k_track_links(TrackParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int K = p.K;
    const int kp = (K + 1) & ~1;
    const size_t per_wave = (size_t)kp * 8 * 4 + (size_t)kp * 4 * 3;
    unsigned char* base = smem + per_wave * wid;
    double* cm = (double*)base;
    double* cf = cm + kp;
    double* pm = cf + kp;
    double* pf = pm + kp;
    int* corder = (int*)(pf + kp);   // corder[r] = slot of the r-th new peak (descending mag)
    int* porder = corder + kp;       // porder[r] = slot (in frame fr-1) of the r-th previous peak
    int* used = porder + kp;
    const int64_t fr = (int64_t)blockIdx.x * nw + wid;
    if (fr >= p.F) return;
    const double* fc = p.f + fr * K;
    const double* mc = p.mag + fr * K;
    int32_t* link = p.link + fr * K;
    int32_t* nrk = p.newrank + fr * K;
    for (int s = lane; s < K; s += 64) {
        cf[s] = fc[s]; cm[s] = mc[s];
        if (fr > 0) { pf[s] = p.f[(fr - 1) * K + s]; pm[s] = p.mag[(fr - 1) * K + s]; }
        else { pf[s] = 0.0; pm[s] = 0.0; }
        link[s] = -2;
        nrk[s] = -1;
    }
    wave_sync_t();
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
                used[r] = 0;
            }
        }
        nc += __popcll(__ballot(vc));
        np += __popcll(__ballot(vp));
    }
    wave_sync_t();
    int nnew = 0;
    for (int c = 0; c < nc; c++) {                                   // PVAnalysis.py:903
        const int s = corder[c];
        const double fcur = cf[s];
        double best = INFINITY;
        int bi = 0x7fffffff;
        for (int i = lane; i < np; i += 64) {
            if (!used[i]) {
                double st = fabs(17.312 * (fcur / pf[porder[i]] - 1.0));  // dpitch2st, PVAnalysis.py:62-68, 914
                if (st < best) { best = st; bi = i; }
            }
        }
        wave_argmin(best, bi);
        const bool hit = (bi != 0x7fffffff) && (best < p.maxjmp);    // PVAnalysis.py:923
        // another unused previous partial exactly as near AND exactly as strong as the winner: the reference
        // would let the partial index decide (see the header)
        if (hit) {
            bool amb = false;
            const double wm = pm[porder[bi]];
            for (int i = lane; i < np; i += 64) {
                if (!used[i] && i != bi) {
                    const double st = fabs(17.312 * (fcur / pf[porder[i]] - 1.0));
                    amb = amb || (st == best && pm[porder[i]] == wm);
                }
            }
            if (__ballot(amb) != 0ull && lane == 0) *p.ambiguous = 1;
        }
        if (lane == 0) {
            if (hit) { link[s] = porder[bi]; used[bi] = 1; p.succ[(fr - 1) * K + porder[bi]] = 1; }
            else { link[s] = -1; nrk[s] = nnew; }                    // add_empty_partial
        }
        if (!hit) nnew++;
        wave_sync_t();
    }
    if (lane == 0) p.newcount[fr] = nnew;
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

// exclusive scan of newcount[F] -> newbase[F+1] (single workgroup, 1024 threads, chunked)
__global__ __launch_bounds__(1024) void k_scan_counts(TrackParams p) {
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < p.F; base += 1024) {
        const int64_t i = base + tid;
        long long v = (i < p.F) ? (long long)p.newcount[i] : 0;
        long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            long long u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        long long woff = 0;
        for (int w = 0; w < wid; w++) woff += wsum[w];
        const long long carry = carry_s;
        if (i < p.F) p.newbase[i] = carry + woff + inc - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (tid == 0) { p.newbase[p.F] = carry_s; *p.npartials = carry_s; }
}

__global__ __launch_bounds__(256) void k_root_init(TrackParams p) {
    const int64_t n = p.F * (int64_t)p.K;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int l = p.link[i];
    int32_t r;
    if (l == -2) r = -1;
    else if (l == -1) r = (int32_t)i;
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
    const int64_t pid = p.newbase[rfr] + p.newrank[r];               // creation order, PVAnalysis.py:826
    p.partial_id[i] = (int32_t)pid;
    if (pid < p.cap) {
        if (r == (int32_t)i) p.part_start[pid] = (int32_t)rfr;
        // the last point of a partial (no peak of the next frame continues it) knows the length:
        // one plain store per partial instead of one contended atomic per point
        if (!p.succ[i]) p.part_len[pid] = (int32_t)(i / p.K - rfr + 1);
    }
    // max(SinSum.end): one atomic per partial (its last point)
    if (!p.succ[i]) atomicMax((long long*)p.maxend, (long long)(i / p.K));
}

}  // namespace

int pvx_launch_track(const TrackParams& p, hipStream_t s) {
    if (p.F <= 0) return PVX_OK;
    const int64_t n = p.F * (int64_t)p.K;
    if (n >= 0x7fffffffLL) { pvx_set_error("F*K = %lld does not fit the 32-bit node index", (long long)n); return PVX_ERR_UNSUPPORTED; }
    const int kp = (p.K + 1) & ~1;
    int waves = 4;
    size_t per_wave = (size_t)kp * 8 * 4 + (size_t)kp * 4 * 3;
    while (waves > 1 && per_wave * waves > 64 * 1024) waves >>= 1;
    if (per_wave * waves > 64 * 1024) { pvx_set_error("npks=%d too large for the tracker", p.K); return PVX_ERR_UNSUPPORTED; }
    PVX_HIP_CHECK(hipMemsetAsync(p.succ, 0, (size_t)n, s));
    PVX_HIP_CHECK(hipMemsetAsync(p.ambiguous, 0, sizeof(int64_t), s));
    PVX_HIP_CHECK(hipMemsetAsync(p.maxend, 0xff, sizeof(int64_t), s));          // -1
    hipLaunchKernelGGL(k_track_links, dim3((unsigned)((p.F + waves - 1) / waves)), dim3(64 * waves), per_wave * waves, s, p);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, p);
    const unsigned nb = (unsigned)((n + 255) / 256);
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
