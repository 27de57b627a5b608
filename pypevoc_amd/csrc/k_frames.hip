// k_frames.hip -- hop-strided framing + windowing (replaces PV.calc_fft_frame's
// `x[pos:pos+nfft] * win`, pypevoc/PVAnalysis.py:155-156, for every frame of a launch at once).
//
// Roofline: HBM.  Per frame: hop new input samples enter from HBM (the nfft/hop-fold overlap of
// consecutive frames is served by L2: the frames that share samples are written by neighbouring
// workgroups), nfft windowed samples leave.  Algorithmic bytes / frame = hop*s_in + nfft*s_out.
//
// One 256-thread workgroup per workspace row.  16-byte-per-lane loads/stores when the geometry
// allows (hop, stride and nfft multiples of 4 samples, float32 in and out), plain coalesced dword
// traffic otherwise.  1/wfact (PV.py:102, 157) is folded into the window on the host.
#include "pvx_internal.h"

namespace {

template <typename InT> __device__ inline double to_double(InT v) { return (double)v; }

// Global row -> (signal, frame); returns false for zero rows (row 0 of each signal, rows outside
// the row space).
__device__ inline bool row_to_frame(int64_t g, int64_t F, int64_t total_rows, int64_t& b, int64_t& fr) {
    if (g < 0 || g >= total_rows) return false;
    b = g / (F + 1);
    int64_t q = g - b * (F + 1);
    if (q == 0) return false;
    fr = q - 1;
    return true;
}

template <typename InT, typename T>
__global__ __launch_bounds__(256) void k_frames_generic(FrameParams p) {
    const int64_t j = blockIdx.x;                 // workspace row
    const int64_t g = p.R0 - 1 + j;               // global row
    T* out = (T*)p.frames + j * p.ldi;
    const T* win = (const T*)p.win;
    int64_t b, fr;
    if (!row_to_frame(g, p.F, p.total_rows, b, fr)) {
        for (int n = threadIdx.x; n < p.nfft; n += 256) out[n] = (T)0;
        return;
    }
    const InT* x = (const InT*)p.x + b * p.sig_stride + fr * (int64_t)p.hop;
    for (int n = threadIdx.x; n < p.nfft; n += 256) out[n] = (T)x[n] * win[n];
}

// float32 -> float32, everything 16-byte aligned
__global__ __launch_bounds__(256) void k_frames_f32x4(FrameParams p) {
    const int64_t j = blockIdx.x;
    const int64_t g = p.R0 - 1 + j;
    float4* out = (float4*)((float*)p.frames + j * p.ldi);
    const float4* win = (const float4*)p.win;
    const int nv = p.nfft >> 2;
    int64_t b, fr;
    if (!row_to_frame(g, p.F, p.total_rows, b, fr)) {
        for (int n = threadIdx.x; n < nv; n += 256) out[n] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float4* x = (const float4*)((const float*)p.x + b * p.sig_stride + fr * (int64_t)p.hop);
#pragma unroll 2
    for (int n = threadIdx.x; n < nv; n += 256) {
        float4 v = x[n];
        float4 w = win[n];
        out[n] = make_float4(v.x * w.x, v.y * w.y, v.z * w.z, v.w * w.w);
    }
}

}  // namespace

int pvx_launch_frames(const FrameParams& p, int x_dtype, int precision, hipStream_t s) {
    if (p.ws_rows <= 0) return PVX_OK;
    if (p.ws_rows > 0x7fffffffLL) { pvx_set_error("too many rows in one launch"); return PVX_ERR_INVALID; }
    dim3 grid((unsigned)p.ws_rows), block(256);
    const bool vec_ok = precision == 32 && x_dtype == PVX_F32 && (p.nfft % 4 == 0) && (p.hop % 4 == 0) &&
                        (p.sig_stride % 4 == 0) && (p.ldi % 4 == 0) && (((uintptr_t)p.x) % 16 == 0) &&
                        (((uintptr_t)p.frames) % 16 == 0) && (((uintptr_t)p.win) % 16 == 0);
    if (vec_ok) {
        hipLaunchKernelGGL(k_frames_f32x4, grid, block, 0, s, p);
    } else if (precision == 32) {
        switch (x_dtype) {
            case PVX_F32: hipLaunchKernelGGL((k_frames_generic<float, float>), grid, block, 0, s, p); break;
            case PVX_F64: hipLaunchKernelGGL((k_frames_generic<double, float>), grid, block, 0, s, p); break;
            case PVX_I16: hipLaunchKernelGGL((k_frames_generic<int16_t, float>), grid, block, 0, s, p); break;
            default: pvx_set_error("bad x_dtype %d", x_dtype); return PVX_ERR_INVALID;
        }
    } else {
        switch (x_dtype) {
            case PVX_F32: hipLaunchKernelGGL((k_frames_generic<float, double>), grid, block, 0, s, p); break;
            case PVX_F64: hipLaunchKernelGGL((k_frames_generic<double, double>), grid, block, 0, s, p); break;
            case PVX_I16: hipLaunchKernelGGL((k_frames_generic<int16_t, double>), grid, block, 0, s, p); break;
            default: pvx_set_error("bad x_dtype %d", x_dtype); return PVX_ERR_INVALID;
        }
    }
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}

// dst[i] = (float) src[i]: a device-resident float64 signal for the fused float32 kernels (pvx_api.hip, analyze_rows).
// Streaming, 32 bytes in and 16 out per lane and trip; bound by HBM (12 bytes per sample).
namespace {
__global__ __launch_bounds__(256) void k_narrow(const double* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const double2 a = ((const double2*)src)[2 * i], b = ((const double2*)src)[2 * i + 1];
        ((float4*)dst)[i] = make_float4((float)a.x, (float)a.y, (float)b.x, (float)b.y);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] = (float)src[(n4 << 2) + threadIdx.x];
}
__global__ __launch_bounds__(256) void k_narrow_any(const double* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = (float)src[i];
}
}  // namespace

int pvx_launch_narrow(const double* src, float* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return PVX_OK;
    const int64_t want = (n / 4 + 255) / 256;
    const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 256 * 16 ? 256 * 16 : want));
    if (((uintptr_t)src) % 16 == 0 && ((uintptr_t)dst) % 16 == 0) hipLaunchKernelGGL(k_narrow, dim3(grid), dim3(256), 0, s, src, dst, n);
    else hipLaunchKernelGGL(k_narrow_any, dim3(grid), dim3(256), 0, s, src, dst, n);
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}
