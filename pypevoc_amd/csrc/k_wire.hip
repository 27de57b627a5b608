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

// ---- format 2 (precision 32): 14 B per slot.  At precision 32 a peak's frequency is a float64 function of its bin and ONE float32
// value (peak_math, pvx_wave.h): f = nbin fstep - (double)u / dt with u the unwrapped phase offset in cycles -- or, for a frame that
// follows an all-zero spectrum, one of twelve float64 expressions of (quadrant of the peak's value, unwrapping candidate m), sent as
// u = 8 + 3 quadrant + (m + 1).  The block carries u instead of f and the receiver evaluates the kernels' own expressions:
//   u[n] f32 | mag[n] f32 | ph[n] f32 | binno[n] u16 | pad | totalmag[rows] f64            (an empty slot: all zeros, which decode to +0.0)
// k_fused_rev writes u itself (FusedParams::wire == 2); k_pack_rows2 finds it from a result block: the float32 nearest
// (nbin fstep - f) dt that gives f back bit for bit, else the one of the twelve that does -- and a NaN where none does (arrays that no
// precision-32 analysis of this library wrote), so that the receiver's f says so.
constexpr double kPi2 = 2.0 * kPi;
__device__ __forceinline__ double f_of_code(int code, int nbin, const WireParams& p) {
    const int qi = code / 3, m = code - 3 * qi - 1;
    const double dphd = qi == 0 ? kPi / 4 : (qi == 1 ? -kPi / 4 : (qi == 2 ? 3 * kPi / 4 : -3 * kPi / 4));
    const double w0 = dphd + p.wfbin[nbin];
    return (w0 + kPi2 * (double)m) / p.dt / kPi2;                      // peak_math<float>, the branch of a zero previous spectrum
}
struct Wire2 { float *u, *m, *ph; unsigned short* b; double* t; };
__host__ __device__ inline Wire2 wire2_sections(void* wire, int64_t n) {
    unsigned char* w = (unsigned char*)wire;
    Wire2 s;
    s.u = (float*)w;
    s.m = (float*)(w + al8((size_t)n * 4));
    s.ph = (float*)(w + 2 * al8((size_t)n * 4));
    s.b = (unsigned short*)(w + 3 * al8((size_t)n * 4));
    s.t = (double*)(w + 3 * al8((size_t)n * 4) + al8((size_t)n * 2));
    return s;
}

__global__ __launch_bounds__(256) void k_pack_rows2(WireParams p) {
    const int64_t n = p.rows * (int64_t)p.K;
    const Wire2 w = wire2_sections(p.wire, n);
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double f = p.f[i];
        const int nbin = (int)p.binno[i];
        const double fb = (double)nbin * p.fstep;
        float u = __builtin_nanf("");
        const float c0 = (float)((fb - f) * p.dt);
        bool found = false;
        float c = c0;
#pragma unroll 1
        for (int t = 0; t < 5 && !found; t++) {                       // c0, then its neighbours one and two float32 steps away
            c = t == 0 ? c0 : (t == 1 ? nextafterf(c0, INFINITY) : (t == 2 ? nextafterf(c0, -INFINITY)
                        : (t == 3 ? nextafterf(nextafterf(c0, INFINITY), INFINITY) : nextafterf(nextafterf(c0, -INFINITY), -INFINITY))));
            found = c < 7.f && fb - (double)c / p.dt == f;
        }
        if (found) u = c;
        else if (nbin >= 0) {
#pragma unroll 1
            for (int code = 0; code < 12; code++)
                if (f_of_code(code, nbin, p) == f) { u = 8.f + (float)code; break; }
        }
        w.u[i] = u;
        w.m[i] = (float)p.mag[i];
        w.ph[i] = (float)p.ph[i];
        w.b[i] = (unsigned short)nbin;
        if (i < p.rows) w.t[i] = p.totalmag[i];
    }
}

__global__ __launch_bounds__(256) void k_unpack_rows2(WireParams p) {
    const int64_t n = p.rows * (int64_t)p.K;
    const Wire2 w = wire2_sections(p.wire, n);
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float u = w.u[i];
        const double ph = (double)w.ph[i];
        const int nbin = w.b[i];
        const double fb = (double)nbin * p.fstep;
        double f;
        if (u >= 7.5f) f = f_of_code((int)(u - 8.f) < 11 ? (int)(u - 8.f) : 11, nbin, p);
        else f = fb - (double)u / p.dt;                               // peak_math<float>; an empty slot: 0 - 0 = +0.0
        p.of[i] = f;
        p.omag[i] = (double)w.m[i];
        p.oph[i] = ph;
        p.obinno[i] = (double)nbin;
        p.orealph[i] = ph + kPi * (fb - f) / p.fstep;                 // PV.py:146 + 207, as k_unpack_rows
        if (i < p.rows) p.ototalmag[i] = w.t[i];
    }
}

}  // namespace

size_t pvx_wire_block_bytes(int64_t rows, int K, int precision, int fmt) {
    const size_t n = (size_t)rows * (size_t)K, ts = precision == 64 ? 8 : 4;
    if (fmt == 2) return 3 * al8(n * 4) + al8(n * 2) + (size_t)rows * 8;
    return al8(n * 8) + 2 * al8(n * ts) + al8(n * 2) + (size_t)rows * 8;
}

int pvx_launch_wire(const WireParams& p, bool pack, hipStream_t s) {
    const int64_t n = p.rows * (int64_t)p.K;
    if (n <= 0) return PVX_OK;
    int64_t nb = (n + 255) / 256;
    if (nb > 256 * 32) nb = 256 * 32;                                 // grid-stride beyond 32 blocks per CU
    if (p.fmt == 2) {
        if (p.precision != 32 || !p.wfbin || !(p.dt > 0.0)) { pvx_set_error("wire format 2 is the precision-32 format"); return PVX_ERR_UNSUPPORTED; }
        if (pack) hipLaunchKernelGGL(k_pack_rows2, dim3((unsigned)nb), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(k_unpack_rows2, dim3((unsigned)nb), dim3(256), 0, s, p);
        PVX_HIP_CHECK(hipGetLastError());
        return PVX_OK;
    }
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
