// pvx_internal.h -- shared declarations of libpvx_hip (gfx950 only; no host fallback).
#pragma once

#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <stdint.h>

#include "pvx.h"

// ---- error plumbing -------------------------------------------------------------------------
void pvx_set_error(const char* fmt, ...);

#define PVX_HIP_CHECK(call)                                                                     \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            pvx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,     \
                          __LINE__);                                                            \
            return PVX_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)

#define PVX_FFT_CHECK(call)                                                                     \
    do {                                                                                        \
        rocfft_status s__ = (call);                                                             \
        if (s__ != rocfft_status_success) {                                                     \
            pvx_set_error("%s failed: rocfft_status %d (%s:%d)", #call, (int)s__, __FILE__,     \
                          __LINE__);                                                            \
            return PVX_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)

int pvx_require_device();  // PVX_OK or PVX_ERR_NO_DEVICE (sets the message)

// workgroups of `threads` threads and `lds` bytes of dynamic LDS of kernel `fn` that are RESIDENT on a CU together, registers included
// (hipOccupancyMaxActiveBlocksPerMultiprocessor, remembered per (device, kernel, threads, lds)); <= 0: unknown.  The general path's
// kernels size their grids as CUs x workgroups per CU: a grid sized by LDS alone launches workgroups that wait for a slot and
// then run by themselves -- nfft 512 at float64 asked for four workgroups of four waves per CU where registers admit three:
// 450 -> 505 M frames/s with the grid that fits.
int pvx_resident_blocks(const void* fn, int threads, size_t lds);

// ---- row space ------------------------------------------------------------------------------
// The analysis works on a global "row space": every signal owns F+1 consecutive rows, row 0 of a
// signal being an all-zero frame (the reference's initial oldfft = zeros, PV.py:121) and row q > 0
// being frame q-1.  A launch covers global rows [R0, R0+nrows); workspace row j holds global row
// R0-1+j, so every processed row finds its predecessor in workspace row j-1.

struct FrameParams {
    const void* x;        // input samples (InT)
    int64_t nsamp;        // samples per signal
    int64_t sig_stride;   // samples between signal starts
    int64_t F;            // frames per signal
    int64_t R0;           // first global row of this launch (workspace row 1)
    int64_t ws_rows;      // workspace rows to write (= nrows + 1)
    int64_t total_rows;   // nsig * (F + 1)
    int nfft, hop;
    const void* win;      // window * (1/wfact), T[nfft]
    int win_symmetric = 0; // w[n] == w[nfft-1-n] for every n (np.hanning and its kind): k_stft_pv may keep half of it
    void* frames;         // T [ws_rows][ldi]
    int64_t ldi;          // elements per workspace row
    // optional candidate output of the split transform (pvx_stft.h: StftParams)
    void* cand_y = nullptr;
    unsigned short* cand_bin = nullptr;
    double* cand_stats = nullptr;
    int cand_cap = 0;
    double cand_thr = 0.0;
};

struct PeaksParams {
    const void* spec;     // complex<T> [ws_rows][ldo]
    int64_t ldo;          // complex elements per row
    int64_t F, R0, nrows;
    int nfft, hop, N2, K, rad;
    double thr;           // pkthresh (PF minrattomax)
    double sr, fstep, dt;
    const double* wfbin;  // [N2] 2*pi*round_half_even(k*fstep*dt)  (PV.py:116-118)
    const double* prev0;  // optional [N2][2] spectrum preceding frame 0 of signal 0
    double *f, *mag, *ph, *realph, *binno, *t, *totalmag;
    int frames_per_wave;
    // optional: the rows' candidate peaks left by the split transform (row rel + 1 of the workspace = this launch's row rel)
    const void* cand_y = nullptr;
    const unsigned short* cand_bin = nullptr;
    const double* cand_stats = nullptr;
    int cand_cap = 0;
};

struct PeakRowsParams {   // standalone PeakFinder
    const double* y;      // [nrows][n]
    int64_t nrows;
    int n, npeaks, thr_kind, rad, cap;
    double thr_val;
    int32_t* pos;         // [nrows][cap]
    int8_t* keep;         // [nrows][cap]
    int32_t* count;       // [nrows]
};

struct FusedParams {      // k_fused.hip: window + FFT + untangle + peaks in one wave per frame
    const void* x;        // input samples
    int64_t sig_stride, F, total_rows;
    int hop, K, rad;
    double thr, sr, fstep, dt;
    const double* wfbin;
    const double* prev0;
    double *f, *mag, *ph, *realph, *binno, *t, *totalmag;
    const void* win;      // float[nfft] window / wfact
    const void* twiddle;  // float2[nfft] W_nfft^j
    float* spec_out;      // optional: half spectrum (nfft/2 complex) of global row spec_row
    int64_t spec_row;
    int64_t blocks_override;
    // k_fused_rev.hip: where a wave leaves the first spectrum it computes for the wave below it in its workgroup (whose last
    // frame needs it as its previous spectrum): float2 [workgroups x waves][nfft / 2], or NULL -- every wave then computes
    // the row below its range itself.  pvx_fused_rev_stash_bytes: what the launch of these parameters would use.
    void* stash;
    size_t stash_bytes;
    // k_fused_rev.hip: 1 = f / mag / ph / binno / totalmag point at the sections of a wire block (k_wire.hip; realph and t unused)
    int wire = 0;
};
size_t pvx_fused_rev_stash_bytes(const FusedParams& p, int nfft);
int pvx_fused_supported(int nfft, int precision, int K);
int pvx_launch_fused(const FusedParams& p, int nfft, int x_dtype, hipStream_t s);
int pvx_fused_mw_supported(int nfft, int precision, int K);     // k_fused_mw.hip: several waves per frame
int pvx_launch_fused_mw(const FusedParams& p, int nfft, int x_dtype, hipStream_t s);
int pvx_fused_ring_supported(int nfft, int precision, int K);   // k_fused_ring.hip: workgroup-shared spectrum ring
int pvx_launch_fused_ring(const FusedParams& p, int nfft, int x_dtype, hipStream_t s);
int pvx_fused_rev_supported(int nfft, int precision, int K);    // k_fused_rev.hip: independent waves, rows in descending order
int pvx_launch_fused_rev(const FusedParams& p, int nfft, int x_dtype, hipStream_t s);
int pvx_fused_team_supported(int nfft, int precision, int K);   // k_fused_team.hip: nfft 4096 / 8192 as teams of k_fused_rev-shaped waves
int pvx_launch_fused_team(const FusedParams& p, int nfft, int x_dtype, hipStream_t s);
int pvx_fused_team_table_len(int nfft);                          // complex entries it wants appended to the W_nfft^j table
void pvx_fused_team_table(int nfft, const float* tw, float* out);

// k_stft.hip: fused float64 STFT into the spectrum workspace
int pvx_stft_supported(int nfft, int precision);
struct FrameParams;

// launchers (defined in the .hip files)
int pvx_launch_frames(const FrameParams& p, int x_dtype, int precision, hipStream_t s);
int pvx_launch_narrow(const double* src, float* dst, int64_t n, hipStream_t s);   // dst[i] = (float) src[i]
int pvx_launch_stft(const FrameParams& fp, void* spec, int64_t ldo, const void* twiddle, int x_dtype, int precision, hipStream_t s);
int pvx_launch_phase_peaks(const PeaksParams& p, int precision, hipStream_t s);
// k_stft_pv.hip: k_stft + k_phase_peaks in one launch (nfft 512..2048 while the peak search fits the transform buffer)
int pvx_stft_pv_supported(int nfft, int precision, int K);
int pvx_stft_pv_takes(int nfft, int precision, int x_dtype, int hop);
int pvx_launch_stft_pv(const FrameParams& fp, const PeaksParams& pp, void* spec, int64_t ldo, const void* twiddle, int x_dtype,
                       int precision, hipStream_t s);
// k_pv_rev.hip: the float64 analysis in one launch with the spectrum rows kept on chip (rows walked downwards; nfft 512 .. 2048, npks <= 64)
struct PvRevParams {
    const void* x;        // input samples
    int64_t sig_stride, F, total_rows;
    int64_t row_begin, row_end;   // the global rows this launch analyses (all: 0, total_rows)
    int hop, K, rad;
    double thr, sr, fstep, dt;
    const double* wfbin;
    const double* prev0;
    double *f, *mag, *ph, *realph, *binno, *t, *totalmag;
    const void* win;      // double[nfft] window / wfact
    const void* twiddle;  // complex double [nfft] W_nfft^j
    double* spec_out;     // optional: half spectrum (nfft/2 complex float64) of global row spec_row
    int64_t spec_row;
    int64_t blocks_override;
    int win_symmetric;
    void* stage;          // nfft 2048: double [workgroups x waves][64][5], the kept peaks' values between the frames (pvx_pv_rev_stage_bytes)
    size_t stage_bytes;
};
int pvx_pv_rev_supported(int nfft, int precision, int K);
int pvx_pv_rev_takes(int nfft, int x_dtype, int hop);
size_t pvx_pv_rev_stage_bytes(int nfft);
int pvx_launch_pv_rev(const PvRevParams& p, int nfft, int x_dtype, hipStream_t s);
// k_pv_team.hip: the same at nfft 4096 / 8192 with a team of waves per frame (hop nfft/4 or nfft/2, npks <= 64)
int pvx_pv_team_supported(int nfft, int precision, int K, int hop);
size_t pvx_pv_team_stage_bytes(int nfft);
int pvx_launch_pv_team(const PvRevParams& p, int nfft, int x_dtype, hipStream_t s);
int pvx_launch_peak_rows(const PeakRowsParams& p, hipStream_t s);
size_t pvx_phase_peaks_lds_bytes(int N2, int K, int precision, int waves);

struct TrackParams {
    const double* f;      // [F][K]
    const double* mag;
    int64_t F;
    int K;
    double maxjmp;
    int32_t* partial_id;  // [F][K]
    int32_t* part_start;  // [cap]
    int32_t* part_len;    // [cap]
    int64_t cap;
    // workspace
    int32_t* link;        // [F][K]  >= 0: continues that slot of frame-1; -1: empty slot; <= -2: starts a partial,
                          //         -(link + 2) = its rank among the new partials of its frame (creation order)
    int32_t* newcount;    // [F]     partials created at each frame
    int64_t* newbase;     // [F+1]   exclusive scan of newcount
    int32_t* root;        // [F][K]  flattened index of the first point of the slot's partial
    unsigned char* succ;  // [F][K]  1 if a peak of the next frame continues this one
    int64_t* npartials;   // [1]
    int64_t* maxend;      // [1]  last frame that holds a point of any partial = max(SinSum.end) (PV.py:1059)
    int chunk, fpw;       // frames per k_track_links workgroup = chunk length of the root step (a power of two), and
                          // frames a wave takes in turn (set by pvx_launch_track)
    // k_track_links_lane (npks <= 8): per chunk of 256 frames the partials it creates (flag bits as newcount), the last
    // frame with a peak, and the scan of the totals; nullptr: newbase holds the scan over all frames
    int32_t* chunktot = nullptr;     // [ceil(F / 256)]
    int32_t* chunklast = nullptr;    // [ceil(F / 256)]
    int64_t* chunkbase = nullptr;    // [ceil(F / 256) + 1]
    // rows wider than 8 slots that hold at most 8 valid peaks (the reference's default npks is 20; a tone has a handful) go
    // through k_track_links_lane too: it works on slots 0 .. 7 and stores gen into *wide when a frame has a valid peak beyond;
    // the caller then launches again with wide = nullptr (the wave-per-frame kernels)
    unsigned* wide = nullptr;        // [1], read by the host with the result words (page-locked host memory when the caller has some)
    unsigned* wide_dev = nullptr;    // [1], the same in device memory: what the launch's later kernels look at
    unsigned gen = 0;                // the call's number (never 0)
    int64_t* ambiguous;   // [1]  set by k_track_links when the reference's (magnitude, partial index) order of
                          //      the previous partials would decide an assignment (k_track.hip header)
};
int pvx_launch_track(const TrackParams& p, hipStream_t s);
int pvx_launch_track_sequential(const TrackParams& p, hipStream_t s);   // exact loop on one wave, after a flag

struct SynthParams {
    const double *f, *mag, *realph;   // [F][K]
    const int32_t* partial_id;        // [F][K]
    const int32_t *part_start, *part_len;
    int64_t F, P;
    int K;
    double sr, edge;
    int nfft, hop_a, hop_s, minframes;
    double* w;
    int64_t wlen;
    int no_phcor;         // PVX_SYNTH_NO_PHCOR: fstep=None partials (PV.py:710-713)
    int f32_samples = 0;  // PVX_SYNTH_F32 (plans at precision 32): the bodies' sample loop in float32 (k_synth_bodies<R, float>)
    int64_t seg0 = 0;     // first output segment (hop) of this launch ...
    int64_t seg_count = 0;   // ... and how many (0: all from seg0 on)
    // device workspace of pvx_synth_ws_bytes() bytes (nullptr: a grow-only buffer per stream, owned by the library);
    // skip_prepare: the partial-major copy of the analysis arrays is already in `ws` (a later slice of the same waveform)
    void* ws = nullptr;
    size_t ws_bytes = 0;
    unsigned* ws_gen = nullptr;   // with ws: the owner's call counter for it, 0 after every (re)allocation (k_synth.hip: segment flags)
    int skip_prepare = 0;
};
int pvx_launch_synth(const SynthParams& p, hipStream_t s);
size_t pvx_synth_ws_bytes(int64_t F, int K, int64_t P, int nfft, int hop_a, int hop_s, double edge);

// PVHarmonic.run_pv (k_harmonic.hip): f0-guided bin sampling on the spectra of the general path
struct HarmParams {
    const void* spec;     // complex<T> workspace rows of this chunk [nrows+1][ldo]
    int64_t ldo;
    int64_t fr_begin, nfr;    // frames [fr_begin, fr_begin + nfr) of the signal are analysed by this launch
    int64_t ws_off;           // workspace row of frame fr_begin
    int nfft, hop, N2, K;
    double sr, fstep, dt, fmin;
    const double* wfbin;
    const double* prev0;      // optional [N2][2]: spectrum preceding the first valid frame
    const void* carry;        // complex<T>[N2]: spectrum of the last valid frame of the earlier chunks
    const double* f0;         // [F] fundamental per frame (<= 0 or NaN: frame skipped)
    const int32_t* prevrow;   // [F] workspace row of the previous VALID frame's spectrum; -1: there is
                              // none (zero spectrum or prev0); -2: it is in `carry`
    double *f, *mag, *ph, *residual, *t;
};
int pvx_launch_harmonic(const HarmParams& p, int precision, hipStream_t s);

// windowed reductions with the analysis framing (k_reduce.hip)
struct ReduceParams {
    const double* x;          // [n]
    const double* hetsig;     // [n][2] complex (heterodyne only)
    const double* wind;       // [wlen]
    int64_t nfr;
    int wlen, hop;
    double norm;              // sum(wind) | sum(wind**2)
    double* out;              // [nfr][2] | [nfr]
    int64_t* icent;           // [nfr] optional (heterodyne)
};
int pvx_launch_reduce(const ReduceParams& p, int mode, hipStream_t s);
// FuncWind's named reducers (func: pvx_funcwind_op); x complex128 when cpx (p.x then points at [n][2])
int pvx_launch_funcwind(const ReduceParams& p, int func, bool cpx, hipStream_t s);

// result wire format for the multi-GPU gather (k_wire.hip)
struct WireParams {
    int64_t rows;                     // frames (all signals of the shard)
    int K, precision;
    int fmt = 1;                      // 1: f as float64 (18 / 26 B per slot); 2 (precision 32): the float32 it is computed from (14 B)
    double fstep;
    double dt = 0.0;                  // fmt 2: hop / sr and the plan's wfbin table (what peak_math computes a frequency from)
    const double* wfbin = nullptr;
    void* wire;
    const double *f, *mag, *ph, *binno, *totalmag;                    // pack: inputs
    double *of, *omag, *oph, *orealph, *obinno, *ototalmag;           // unpack: outputs
};
size_t pvx_wire_block_bytes(int64_t rows, int K, int precision, int fmt = 1);
int pvx_launch_wire(const WireParams& p, bool pack, hipStream_t s);

// frame descriptors and helpers on the result arrays (k_desc.hip)
int pvx_launch_f0(const double* f, const double* mag, int64_t F, int K, double fmin, double fmax, double thr, double* fm,
                  int32_t* im, hipStream_t s);
int pvx_launch_hpower_rows(const double* f, const double* mag, int64_t F, int K, double* rowpow, int32_t* top, hipStream_t s);
int pvx_launch_hpower(const double* f, int64_t F, int K, double f_threshold, const double* rowpow, double* hpower,
                      double* nharm, hipStream_t s);
int pvx_launch_fill_t(double* t, int64_t F, int64_t nsig, int hop, int nfft, double sr, hipStream_t s);
int pvx_launch_spec_to_prev(double* dst, const void* src, int n, int src_is_float, hipStream_t s);
