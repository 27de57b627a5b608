// f64_lat.hip -- issue interval of v_fma_f64 / v_mul_f64 in one wave when every instruction depends on the one D
// instructions earlier (D interleaved chains), at 1 / 2 / 3 / 4 waves per SIMD: what a recurrence like k_synth_bodies' costs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/f64_lat.hip -o tools/ubench/f64_lat && tools/ubench/f64_lat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

template <int D, int MUL> __global__ void k(int iters, long long* out, double* sink, double seed) {
    double c[8];
#pragma unroll
    for (int i = 0; i < 8; i++) c[i] = seed + i + threadIdx.x * 1e-9;
    const double m = 1.0000001, a = 1e-9;
    __builtin_amdgcn_s_barrier();
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
            if (MUL) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(c[u % D]) : "v"(m));
            else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(c[u % D]) : "v"(m), "v"(a));
        }
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += c[i];
    if (s == 123.456) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int D, int MUL> void run(const char* name) {
    long long* d_out; double* d_sink;
    hipMalloc(&d_out, 8 * 4096); hipMalloc(&d_sink, 8);
    printf("%-18s", name);
    for (int wps : {1, 2, 3, 4}) {
        const int iters = 200, nt = 256 * wps;   // wps waves on each SIMD of the CU
        hipLaunchKernelGGL((k<D, MUL>), dim3(256), dim3(nt), 0, 0, iters, d_out, d_sink, 1.0);
        hipLaunchKernelGGL((k<D, MUL>), dim3(256), dim3(nt), 0, 0, iters, d_out, d_sink, 1.0);
        hipDeviceSynchronize();
        std::vector<long long> h(256 * nt / 64);
        hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double cyc = (double)h[h.size() / 2] / (iters * 32.0);
        printf("  %dw: %6.2f /wave %6.2f /SIMD", wps, cyc, cyc / wps);
    }
    printf("\n");
    hipFree(d_out); hipFree(d_sink);
}

int main() {
    run<1, 0>("fma dist 1"); run<2, 0>("fma dist 2"); run<3, 0>("fma dist 3"); run<4, 0>("fma dist 4"); run<8, 0>("fma dist 8");
    run<1, 1>("mul dist 1"); run<2, 1>("mul dist 2"); run<4, 1>("mul dist 4");
    return 0;
}
