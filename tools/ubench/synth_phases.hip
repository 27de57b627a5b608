// Where the resynthesis kernel's time goes on a short signal (F = 239 frames, K = 100 partials alive throughout,
// nfft 4096, hop 1024: config 3's shape): s_memtime stamps of the middle workgroup at its step boundaries.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DPVX_SYNTH_STAMPS -Iinclude -Ipypevoc_amd/csrc \
//         tools/ubench/synth_phases.hip -o tools/ubench/synth_phases && tools/ubench/synth_phases [threads [F K nfft hop]]
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../pypevoc_amd/csrc/k_synth.hip"

void pvx_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

int main(int argc, char** argv) {
    if (argc > 1) setenv("PVX_SYNTH_THREADS", argv[1], 1);
    // [threads] [F K nfft hop]: e.g. `synth_phases 256 51676 8 2048 512` = BASELINE config 2's shape
    const int64_t F = argc > 2 ? atoll(argv[2]) : 239;
    const int K = argc > 3 ? atoi(argv[3]) : 100, nfft = argc > 4 ? atoi(argv[4]) : 4096, hop = argc > 5 ? atoi(argv[5]) : 1024;
    const size_t n = (size_t)F * K;
    std::vector<double> f(n), m(n), r(n);
    std::vector<int32_t> pid(n), st(K, 0), ln(K, (int32_t)F);
    for (int64_t fr = 0; fr < F; fr++)
        for (int s = 0; s < K; s++) {
            f[fr * K + s] = 100.0 * (s + 1) * (1.0 + 1e-4 * fr);
            m[fr * K + s] = 1.0 / (s + 1);
            r[fr * K + s] = 6.283185307179586 * f[fr * K + s] * fr * hop / 44100.0;
            pid[fr * K + s] = s;
        }
    const int64_t wlen = (F + 2) * hop;
    double *df, *dm, *dr, *dw; int32_t *dp, *ds, *dl; long long* stamps;
    hipMalloc(&df, n * 8); hipMalloc(&dm, n * 8); hipMalloc(&dr, n * 8); hipMalloc(&dw, wlen * 8);
    hipMalloc(&dp, n * 4); hipMalloc(&ds, K * 4); hipMalloc(&dl, K * 4); hipMalloc(&stamps, 64 * 8);
    hipMemcpy(df, f.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dm, m.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(dr, r.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dp, pid.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(ds, st.data(), K * 4, hipMemcpyHostToDevice); hipMemcpy(dl, ln.data(), K * 4, hipMemcpyHostToDevice);
    SynthParams sp;
    sp.f = df; sp.mag = dm; sp.realph = dr; sp.partial_id = dp; sp.part_start = ds; sp.part_len = dl; sp.F = F; sp.P = K; sp.K = K;
    sp.sr = 44100.0; sp.edge = 0.5; sp.nfft = nfft; sp.hop_a = hop; sp.hop_s = hop; sp.minframes = 3; sp.w = dw; sp.wlen = wlen;
    sp.slot_of = (int32_t*)stamps; sp.no_phcor = 0; sp.nbatch = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 4; it++) {
        hipEventRecord(e0, nullptr);
        if (pvx_launch_synth(sp, nullptr) != 0) return 1;
        hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        long long t[8]; hipMemcpy(t, stamps, 64, hipMemcpyDeviceToHost);
        printf("launch %d: %.1f us | middle workgroup, s_memtime ticks: candidates+compaction %lld, slot search %lld, values %lld, parameters %lld, samples %lld, store %lld\n",
               it, ms * 1e3, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5]);
    }
    return 0;
}
