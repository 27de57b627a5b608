// fetch_calib.hip -- known-byte-count streaming kernels to calibrate rocprofv3's FETCH_SIZE /
// WRITE_SIZE on gfx950 for the access widths our kernels use (MI355X_MICROARCH.md, HBM section:
// FETCH_SIZE under-reports wide coalesced reads by 2x; other widths must be calibrated).
// Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib
// Each kernel streams a 1 GiB buffer once (read) and writes 1/16 of that (so reads dominate).
#include <hip/hip_runtime.h>
#include <stdio.h>

template <typename V> __global__ void k_read(const V* __restrict__ in, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += stride) {
        V v = in[i];
        const float* f = (const float*)&v;
        for (unsigned j = 0; j < sizeof(V) / 4; j++) acc += f[j];
    }
    if (acc == 123.456f) out[0] = acc;   // never true: keeps the loads alive, writes nothing
}
template <typename V> __global__ void k_write(V* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    V v;
    float* f = (float*)&v;
    for (unsigned j = 0; j < sizeof(V) / 4; j++) f[j] = (float)j;
    for (; i < n; i += stride) out[i] = v;
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    void *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 0, bytes);
    hipMemset(b, 0, bytes);
    hipDeviceSynchronize();
    const int blocks = 256 * 8, threads = 256;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_read<float>, dim3(blocks), dim3(threads), 0, 0, (const float*)a, (float*)b, bytes / 4);
        hipLaunchKernelGGL(k_read<float2>, dim3(blocks), dim3(threads), 0, 0, (const float2*)a, (float*)b, bytes / 8);
        hipLaunchKernelGGL(k_read<float4>, dim3(blocks), dim3(threads), 0, 0, (const float4*)a, (float*)b, bytes / 16);
        hipLaunchKernelGGL(k_write<float>, dim3(blocks), dim3(threads), 0, 0, (float*)b, bytes / 4);
        hipLaunchKernelGGL(k_write<float2>, dim3(blocks), dim3(threads), 0, 0, (float2*)b, bytes / 8);
        hipLaunchKernelGGL(k_write<float4>, dim3(blocks), dim3(threads), 0, 0, (float4*)b, bytes / 16);
        hipLaunchKernelGGL(k_write<double>, dim3(blocks), dim3(threads), 0, 0, (double*)b, bytes / 8);
    }
    hipDeviceSynchronize();
    printf("calibration done: each kernel moved %zu bytes\n", bytes);
    return 0;
}
