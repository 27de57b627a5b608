// Checks the one-instruction complex primitives of pypevoc_amd/csrc/pvx_cplx.h against scalar
// arithmetic, bit for bit.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I pypevoc_amd/csrc
//   tools/ab/cplx_prims_test.hip -o tools/ab/cplx_prims_test && tools/ab/cplx_prims_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "pvx_cplx.h"
using namespace pvxc;

__global__ void k(const float2* a, const float2* b, float2* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const v2f x = mk(a[i].x, a[i].y), y = mk(b[i].x, b[i].y);
    v2f r[11];
    r[0] = add_mni(x, y); r[1] = add_pi(x, y); r[2] = add_conj(x, y); r[3] = sub_conj(x, y);
    r[4] = mni(x); r[5] = conj(x); r[6] = cmul(x, y); r[7] = fma_s(0.5f, x, y);
    r[8] = cmul_k(x, mk(0.92387953f, -0.38268343f)); r[9] = mul_swap(x, mk(0.5f, -0.5f)); r[10] = fms_conj(splat(0.5f), x, y);
    for (int j = 0; j < 11; j++) out[i * 11 + j] = make_float2(r[j].x, r[j].y);
}

int main() {
    const int n = 4096;
    std::vector<float2> a(n), b(n), o(n * 11);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) - (1 << 23)) / (float)(1 << 20); };
    for (int i = 0; i < n; i++) { a[i] = make_float2(rnd(), rnd()); b[i] = make_float2(rnd(), rnd()); }
    a[0] = make_float2(0.f, 1.f); a[1] = make_float2(-0.f, -0.f); a[2] = make_float2(2.f, 0.f);   // signed zeros through mni / conj
    float2 *da, *db, *dout;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dout, n * 88);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
    hipMemcpy(o.data(), dout, n * 88, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) {
        const float xr = a[i].x, xi = a[i].y, yr = b[i].x, yi = b[i].y;
        const float kr = 0.92387953f, ki = -0.38268343f;
        const float e[11][2] = {{xr + yi, xi - yr}, {xr - yi, xi + yr}, {xr + yr, xi - yi}, {xr - yr, xi + yi},
                               {xi, -xr}, {xr, -xi},
                               {__builtin_fmaf(xr, yr, -(xi * yi)), __builtin_fmaf(xr, yi, xi * yr)},
                               {__builtin_fmaf(0.5f, xr, yr), __builtin_fmaf(0.5f, xi, yi)},
                               {__builtin_fmaf(xr, kr, -(xi * ki)), __builtin_fmaf(xr, ki, xi * kr)},
                               {0.5f * xi, -0.5f * xr},
                               {__builtin_fmaf(0.5f, xr, -yr), __builtin_fmaf(-0.5f, xi, yi)}};
        for (int j = 0; j < 11; j++) {
            const float2 g = o[i * 11 + j];
            if (memcmp(&g.x, &e[j][0], 4) || memcmp(&g.y, &e[j][1], 4)) {
                { if (bad < 10) printf("mismatch prim %d at %d: got (%g,%g) want (%g,%g)\n", j, i, g.x, g.y, e[j][0], e[j][1]); bad++; }
            }
        }
    }
    printf(bad ? "FAILED: %d mismatches\n" : "all 11 complex primitives match scalar arithmetic (%d)\n", bad ? bad : n);
    return bad != 0;
}
