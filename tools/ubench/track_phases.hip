// Where the tracker's time goes on a short signal (F = 239 frames, K = 100 peaks, config 3's shape): s_memtime
// stamps at the phase boundaries of k_track_links (frame 1) and k_track_boundaries.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DPVX_TRACK_STAMPS -Iinclude -Ipypevoc_amd/csrc \
//         tools/ubench/track_phases.hip -o tools/ubench/track_phases && tools/ubench/track_phases
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <random>
#include "../../pypevoc_amd/csrc/k_track.hip"

void pvx_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

int main(int argc, char** argv) {
    const int64_t F = argc > 1 ? atoll(argv[1]) : 239;
    const int K = argc > 2 ? atoi(argv[2]) : 100;
    const size_t n = (size_t)F * K;
    std::vector<double> f(n), m(n);
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (int64_t fr = 0; fr < F; fr++)
        for (int s = 0; s < K; s++) {
            // K slowly drifting partials + a few dropouts, like analysis output of a sustained note
            f[fr * K + s] = 100.0 * (s + 1) * (1.0 + 0.002 * (u(rng) - 0.5));
            m[fr * K + s] = u(rng) < 0.05 ? 0.0 : 1.0 / (s + 1) * (1.0 + 0.1 * u(rng));
        }
    double *df, *dm; int32_t *pid, *pst, *pln; char* w;
    const size_t wsz = n * 4 * 3 + F * 4 + 8 + (F + 1 + 16) * 8 + 24 + n + 256;
    hipMalloc(&df, n * 8); hipMalloc(&dm, n * 8); hipMalloc(&pid, n * 4); hipMalloc(&pst, n * 4); hipMalloc(&pln, n * 4); hipMalloc(&w, wsz);
    hipMemcpy(df, f.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dm, m.data(), n * 8, hipMemcpyHostToDevice);
    TrackParams tp;
    tp.f = df; tp.mag = dm; tp.F = F; tp.K = K; tp.maxjmp = 0.5; tp.partial_id = pid; tp.part_start = pst; tp.part_len = pln; tp.cap = (int64_t)n;
    size_t off = 0;
    tp.link = (int32_t*)(w + off); off += n * 4; tp.root = (int32_t*)(w + off); off += n * 4;
    tp.newcount = (int32_t*)(w + off); off += F * 4; off = (off + 7) & ~(size_t)7;
    tp.newbase = (int64_t*)(w + off); off += (F + 1 + 16) * 8; tp.npartials = (int64_t*)(w + off); tp.ambiguous = tp.npartials + 1; tp.maxend = tp.npartials + 2; off += 24;
    tp.succ = (unsigned char*)(w + off);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; it++) {
        hipEventRecord(e0, nullptr);
        if (pvx_launch_track(tp, nullptr) != 0) return 1;
        hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<int64_t> st(16);
        hipMemcpy(st.data(), tp.newbase + F + 1, 16 * 8, hipMemcpyDeviceToHost);
        int64_t pa[3]; hipMemcpy(pa, tp.npartials, 24, hipMemcpyDeviceToHost);
        printf("launch %d: %.1f us, %lld partials | k_track_links (frame 1, s_memtime ticks): load+rank %lld, assignment loop %lld, chunk roots + write-out %lld | k_track_boundaries: load + scan %lld, jumps + write-back %lld\n",
               it, ms * 1e3, (long long)pa[0], (long long)(st[1] - st[0]), (long long)(st[2] - st[1]), (long long)(st[3] - st[2]),
               (long long)(st[9] - st[8]), (long long)(st[10] - st[9]));
    }
    return 0;
}
