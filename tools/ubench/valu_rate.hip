// valu_rate.hip -- issue cost of the vector instructions the fused kernels are made of, at 1 / 2 / 4
// waves per SIMD on gfx950: cycles (s_memtime) per wave-instruction per SIMD for streams of independent
// instructions.  hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP> __device__ __forceinline__ void body(v2f (&a)[16], v2f b, v2f c, double (&d)[8], int (&n)[16], float* lds, int lane) {
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ADDF(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
#define ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 15]));
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(n[i]) : "v"(n[(i + 5) & 15]));
#define DPP(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(n[i]) : "v"(n[(i + 5) & 15]));
#define FMA64(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]), "v"(d[(i + 2) & 7]));
#define ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i & 7]) : "v"(d[(i + 3) & 7]));
#define CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(n[(i + 1) & 15]) : );
#define LSHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(n[i]));
#define SUB3(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(n[(i + 1) & 15]), "v"(n[(i + 2) & 15]));
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i].x));
#define SWZ(i) asm volatile("ds_swizzle_b32 %0, %1 offset:0x101f" : "=v"(n[i]) : "v"(n[(i + 5) & 15]));
#define LDR64(i) asm volatile("ds_read_b64 %0, %1 offset:" #i "*512" : "=v"(a[i]) : "v"(lane * 8));
#define LDW64(i) asm volatile("ds_write_b64 %0, %1 offset:" #i "*512" : : "v"(lane * 8), "v"(a[i]));
#define LDR128(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*1024" : "=v"(*(float4*)&a[(i & 7) * 2]) : "v"(lane * 16));
#define RDL(i) { int s_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s_) : "v"(n[i])); asm volatile("" :: "s"(s_)); }
#define SADD(i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(n[0]));
#define MIX(i) if ((i) & 1) { PKFMA(i) } else { ADDU(i) }
    if constexpr (OP == 0) { REP16(FMA) }
    else if constexpr (OP == 1) { REP16(PKFMA) }
    else if constexpr (OP == 2) { REP16(PKADD) }
    else if constexpr (OP == 3) { REP16(PKMUL) }
    else if constexpr (OP == 4) { REP16(ADDF) }
    else if constexpr (OP == 5) { REP16(ADDU) }
    else if constexpr (OP == 6) { REP16(MOV) }
    else if constexpr (OP == 7) { REP16(DPP) }
    else if constexpr (OP == 8) { REP16(FMA64) }
    else if constexpr (OP == 9) { REP16(ADD64) }
    else if constexpr (OP == 10) { REP16(CNDM) }
    else if constexpr (OP == 11) { REP16(LSHL) }
    else if constexpr (OP == 12) { REP16(SUB3) }
    else if constexpr (OP == 13) { REP16(SQRT) }
    else if constexpr (OP == 14) { REP16(SWZ) asm volatile("s_waitcnt lgkmcnt(0)"); }
    else if constexpr (OP == 15) { REP16(LDR64) asm volatile("s_waitcnt lgkmcnt(0)"); }
    else if constexpr (OP == 16) { REP16(LDW64) asm volatile("s_waitcnt lgkmcnt(0)"); }
    else if constexpr (OP == 17) { REP16(LDR128) asm volatile("s_waitcnt lgkmcnt(0)"); }
    else if constexpr (OP == 18) { REP16(RDL) }
    else if constexpr (OP == 19) { REP16(MIX) }
}

template <int OP> __global__ void k(int iters, long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) float lds[16 * 256 + 64];
    const int lane = threadIdx.x & 63;
    v2f a[16]; double d[8]; int n[16];
    for (int i = 0; i < 16; i++) { a[i] = (v2f){(float)i + lane, 1.f}; n[i] = i * 7 + lane; }
    for (int i = 0; i < 8; i++) d[i] = 1.0 + i;
    v2f b = (v2f){1.0001f, 0.9999f}, c = (v2f){1e-7f, -1e-7f};
    for (int i = threadIdx.x; i < 16 * 256 + 64; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    long long q0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        body<OP>(a, b, c, d, n, lds, lane); body<OP>(a, b, c, d, n, lds, lane); body<OP>(a, b, c, d, n, lds, lane); body<OP>(a, b, c, d, n, lds, lane);
        body<OP>(a, b, c, d, n, lds, lane); body<OP>(a, b, c, d, n, lds, lane); body<OP>(a, b, c, d, n, lds, lane); body<OP>(a, b, c, d, n, lds, lane);
    }
    asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)");
    long long t1 = __builtin_amdgcn_s_memtime();
    long long q1 = __builtin_amdgcn_s_memrealtime();
    float s = 0; for (int i = 0; i < 16; i++) s += a[i].x + a[i].y + (float)n[i]; for (int i = 0; i < 8; i++) s += (float)d[i];
    if (s == 123.456f) sink[0] = s;
    if (lane == 0) { out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0; if (blockIdx.x == 0 && threadIdx.x == 0) out[gridDim.x * (blockDim.x / 64)] = q1 - q0; }
}

template <int OP> void run(const char* name, int ncu) {
    const int iters = 3000;
    long long* out; float* sink;
    hipMalloc(&out, sizeof(long long) * (ncu * 16 + 1)); hipMalloc(&sink, 4);
    printf("%-22s", name);
    for (int wps : {1, 2, 4}) {
        const int waves = 4 * wps;
        hipLaunchKernelGGL((k<OP>), dim3(ncu), dim3(64 * waves), 0, 0, iters, out, sink);
        hipLaunchKernelGGL((k<OP>), dim3(ncu), dim3(64 * waves), 0, 0, iters, out, sink);
        hipDeviceSynchronize();
        std::vector<long long> h(ncu * waves + 1);
        hipMemcpy(h.data(), out, sizeof(long long) * (ncu * waves + 1), hipMemcpyDeviceToHost);
        const double rt = (double)h[ncu * waves];                 // 100 MHz ticks, block 0 wave 0
        const double t00 = (double)h[0];
        std::sort(h.begin(), h.begin() + ncu * waves);
        const double med = (double)h[(ncu * waves) / 2];
        // cycles per wave-instruction as seen by one wave, and per SIMD (wps waves share it); clock of wave (0,0)
        printf("  %dw: %6.2f /wave %5.2f /SIMD @%4.2f GHz", wps, med / (iters * 128.0), med / (iters * 128.0) / wps, t00 / rt * 0.1);
    }
    printf("\n");
    hipFree(out); hipFree(sink);
}

int main() {
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    { long long* o; float* sk; hipMalloc(&o, 8 * (ncu * 16 + 1)); hipMalloc(&sk, 4);       // warm the clocks up
      for (int i = 0; i < 100; i++) hipLaunchKernelGGL((k<1>), dim3(ncu), dim3(1024), 0, 0, 3000, o, sk);
      hipDeviceSynchronize(); hipFree(o); hipFree(sk); }
    run<0>("v_fma_f32", ncu); run<1>("v_pk_fma_f32", ncu); run<2>("v_pk_add_f32", ncu); run<3>("v_pk_mul_f32", ncu);
    run<4>("v_add_f32", ncu); run<5>("v_add_u32", ncu); run<6>("v_mov_b32", ncu); run<7>("v_mov_b32_dpp", ncu);
    run<8>("v_fma_f64", ncu); run<9>("v_add_f64", ncu); run<10>("v_cndmask_b32", ncu); run<11>("v_lshlrev_b32", ncu);
    run<12>("v_and_or_b32", ncu); run<13>("v_sqrt_f32", ncu); run<14>("ds_swizzle_b32", ncu); run<15>("ds_read_b64", ncu);
    run<16>("ds_write_b64", ncu); run<17>("ds_read_b128", ncu); run<18>("v_readlane_b32", ncu); run<19>("pk_fma/add_u32 mix", ncu);
    return 0;
}
