// k_wire.hip -- compact wire format for the multi-GPU result gather.
//
// The reference returns five float64 [F, K] arrays per signal (pypevoc/PVAnalysis.py:256-264), 40 B
// per peak slot.  Only part of that is information:
//   binno   is an integer bin index < nfft/2                       -> uint16
//   realph  = ph + pi*(fbin[binno] - f)/fstep (PV.py:146, 207)      -> recomputed by the receiver
//   mag, ph are float32 values widened to float64 at precision 32  -> float32 (exact)
// so a slot travels as f (f64) + mag + ph (f32 | f64) + binno (u16) = 18 B (precision 32) or 26 B
// (precision 64), and the receiver rebuilds the five arrays BIT-identically (the kernels compute
// realph with exactly the expression used here, see peak_math in pvx_wave.h).  The gather into one
// GPU is bound by its xGMI links (point to point, one link per peer), so bytes on the wire are what
// the multi-GPU step time is made of.
//
// Block layout for n = rows*K slots (all sections 8-byte aligned):
//   f[n] f64 | mag[n] T | ph[n] T | binno[n] u16 | pad | totalmag[rows] f64
// Pure streaming kernels, HBM-bound: 32+18 B/slot to pack, 18+40 B/slot to unpack.
#include "pvx_internal.h"

namespace {

constexpr double kPi = 3.141592653589793238462643383279502884;       // as pvx_wave.h

__host__ __device__ inline size_t al8(size_t v) { return (v + 7) & ~(size_t)7; }

template <typename T>
__global__ __launch_bounds__(256) void k_pack_rows(WireParams p) {
    const int64_t n = p.rows * (int64_t)p.K;
    unsigned char* w = (unsigned char*)p.wire;
    double* wf = (double*)w;
    T* wm = (T*)(w + al8((size_t)n * 8));
    T* wp = (T*)(w + al8((size_t)n * 8) + al8((size_t)n * sizeof(T)));
    unsigned short* wb = (unsigned short*)(w + al8((size_t)n * 8) + 2 * al8((size_t)n * sizeof(T)));
    double* wt = (double*)(w + al8((size_t)n * 8) + 2 * al8((size_t)n * sizeof(T)) + al8((size_t)n * 2));
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        wf[i] = p.f[i];
        wm[i] = (T)p.mag[i];
        wp[i] = (T)p.ph[i];
        wb[i] = (unsigned short)(int)p.binno[i];
        if (i < p.rows) wt[i] = p.totalmag[i];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_unpack_rows(WireParams p) {
    const int64_t n = p.rows * (int64_t)p.K;
    const unsigned char* w = (const unsigned char*)p.wire;
    const double* wf = (const double*)w;
    const T* wm = (const T*)(w + al8((size_t)n * 8));
    const T* wp = (const T*)(w + al8((size_t)n * 8) + al8((size_t)n * sizeof(T)));
    const unsigned short* wb = (const unsigned short*)(w + al8((size_t)n * 8) + 2 * al8((size_t)n * sizeof(T)));
    const double* wt = (const double*)(w + al8((size_t)n * 8) + 2 * al8((size_t)n * sizeof(T)) + al8((size_t)n * 2));
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double f = wf[i];
        const double ph = (double)wp[i];
        const int nbin = wb[i];
        p.of[i] = f;
        p.omag[i] = (double)wm[i];
        p.oph[i] = ph;
        p.obinno[i] = (double)nbin;
        // PV.py:146 + 207; an empty slot (f = 0, binno = 0, ph = 0) gives +0.0 like the zero padding
        p.orealph[i] = ph + kPi * ((double)nbin * p.fstep - f) / p.fstep;
        if (i < p.rows) p.ototalmag[i] = wt[i];
    }
}

}  // namespace

size_t pvx_wire_block_bytes(int64_t rows, int K, int precision) {
    const size_t n = (size_t)rows * (size_t)K, ts = precision == 64 ? 8 : 4;
    return al8(n * 8) + 2 * al8(n * ts) + al8(n * 2) + (size_t)rows * 8;
}

int pvx_launch_wire(const WireParams& p, bool pack, hipStream_t s) {
    const int64_t n = p.rows * (int64_t)p.K;
    if (n <= 0) return PVX_OK;
    int64_t nb = (n + 255) / 256;
    if (nb > 256 * 32) nb = 256 * 32;                                 // grid-stride beyond 32 blocks per CU
    if (pack) {
        if (p.precision == 64) hipLaunchKernelGGL(k_pack_rows<double>, dim3((unsigned)nb), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(k_pack_rows<float>, dim3((unsigned)nb), dim3(256), 0, s, p);
    } else {
        if (p.precision == 64) hipLaunchKernelGGL(k_unpack_rows<double>, dim3((unsigned)nb), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(k_unpack_rows<float>, dim3((unsigned)nb), dim3(256), 0, s, p);
    }
    PVX_HIP_CHECK(hipGetLastError());
    return PVX_OK;
}
