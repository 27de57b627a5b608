/*
 * pvx.h -- C ABI of libpvx_hip.so, the MI355X (gfx950) implementation of the PyPeVoc
 * phase-vocoder hot path:  PV.run_pv() -> PV.toSinSum() -> SinSum.synth().
 *
 * The reference (goiosunsw/PyPeVoc, pure Python) has no FFI of its own; the boundary is its
 * Python class API.  Each entry point below replaces the body of the reference method cited
 * beside it (PV.py = pypevoc/PVAnalysis.py, PF.py = pypevoc/PeakFinder.py) and is what a
 * ctypes binding inside those methods would call -- see INTEGRATION.md for the stub.
 *
 * Conventions
 *   - plain C types only; every buffer is caller-allocated and caller-owned;
 *   - "host" entry points take host pointers, copy in/out and return after the device has
 *     finished; "_dev" entry points take device pointers (HBM-resident data) and a
 *     hipStream_t passed as void*, and only enqueue work on that stream;
 *   - return value 0 (or a non-negative count) on success, a negative pvx_status on error;
 *     pvx_last_error() returns a thread-local message for the last failure;
 *   - there is NO CPU fallback: without a usable HIP device every call fails with
 *     PVX_ERR_NO_DEVICE.
 *   - array layouts are the reference's: (F, K) float64 C-order, zero padded, valid peaks
 *     left-packed in ascending-bin order (PV.py:226-245, 256-264).
 */
#ifndef PVX_H
#define PVX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVX_VERSION 100

typedef enum {
    PVX_OK = 0,
    PVX_ERR_NO_DEVICE = -1,   /* no HIP device / runtime failure at init */
    PVX_ERR_INVALID = -2,     /* bad argument */
    PVX_ERR_HIP = -3,         /* a HIP or rocFFT call failed */
    PVX_ERR_ALLOC = -4,       /* device or host allocation failed */
    PVX_ERR_UNSUPPORTED = -5, /* valid request outside what the kernels handle */
    PVX_ERR_SIZE = -6         /* caller buffer too small / wrong length */
} pvx_status;

/* sample types accepted for the input signal x (PV.py:84 copies whatever the caller passed) */
typedef enum { PVX_F32 = 0, PVX_F64 = 1, PVX_I16 = 2 } pvx_dtype;

/* Every function declared below is an exported entry point of libpvx_hip.so, and nothing else is: the library is built with
 * -fvisibility=hidden, these declarations carry default visibility (tests/test_abi_cpu.py checks both directions). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ---- library ------------------------------------------------------------------------- */

/* Select and initialise the HIP device (idempotent per device).  device < 0: current device. */
int pvx_init(int device);
const char* pvx_last_error(void);
int pvx_version(void);
/* Fingerprint of the kernel sources the library was built from: the first 16 hex digits of the sha256 over
 * the .hip and .h files of pypevoc_amd/csrc in name order (pypevoc_amd/_lib.py refuses a library that does not match its sources;
 * bench.py quotes committed profiles only when they carry the same fingerprint). */
const char* pvx_build_fingerprint(void);
/* Name of the device the library is bound to ("" before pvx_init). */
const char* pvx_device_name(void);
/* HIP device index the library is bound to (-1 before pvx_init) */
int pvx_device(void);

/* Number of frames run_pv produces: pos = 0, hop, ... while pos < nsamp - nfft (PV.py:223-249). */
int64_t pvx_nframes(int64_t nsamp, int nfft, int hop);

/* ---- analysis: PV.__init__ constants + PV.run_pv (PV.py:72-131, 150-264) ------------- */

typedef struct pvx_plan pvx_plan; /* opaque: constants, window, rocFFT plan, device workspace */

/*
 * Create an analysis plan.  Replaces the constant set-up of PV.__init__ (PV.py:97-121) and owns
 * everything calc_fft_frame/calc_pv_frame need on the device.
 *   win        nfft window samples = wind(nfft) (PV.py:97); NULL = symmetric Hann (np.hanning)
 *   precision  32: float32 frames/spectra (per-peak arithmetic and outputs stay float64)
 *              64: float64 end to end
 *   max_rows   upper bound on frames processed per internal launch (0 = default); bounds the
 *              device workspace: about max_rows * nfft * 3 * sizeof(real) bytes
 */
int pvx_plan_create(pvx_plan** plan, double sr, int nfft, int hop, int npks, double pkthresh,
                    const double* win, int precision, int64_t max_rows);
int pvx_plan_destroy(pvx_plan* plan);
/* bytes of device workspace the plan holds */
int64_t pvx_plan_workspace_bytes(const pvx_plan* plan);
/*
 * FFT mode of the analysis stage:
 *   0  framing kernel -> rocFFT batched real FFT -> phase/peak kernel (any nfft, both precisions)
 *   1  fused kernel: window, in-register/LDS FFT and peak stage in one wave per frame, no
 *      intermediate arrays in HBM (nfft in {512, 1024, 2048}, precision = 32)
 *   2  (a witness kernel since round 6) fused kernel with several waves per frame (nfft in {2048, 4096, 8192}, precision = 32)
 *   3  mode 1's arithmetic (bit-identical results) with a workgroup of 8 (nfft 2048) or 12 (nfft 512, 1024)
 *      waves walking as many consecutive frames over a shared ring of spectra in LDS: two / three waves per
 *      SIMD (precision = 32; npks <= 120 at nfft 2048: the staging has to fit the LDS next to the ring)
 *   4  mode 1's arithmetic again (bit-identical results) with every wave on its own: a wave walks a contiguous range
 *      of rows DOWNWARDS over one spectrum buffer, the previous spectrum a frame's peaks need arrives one row later;
 *      sliding sample window at hop = nfft/4, nfft/2 (precision = 32, nfft in {512, 1024, 2048})
 * Mode 0 itself runs as one launch for nfft in {512, 1024, 2048} (window + FFT + peaks: k_pv_rev at float64 with npks <= 64,
 * no spectrum workspace; k_stft_pv otherwise), as fused STFT + phase/peak kernel for nfft 4096 / 8192, and through rocFFT otherwise.
 * A new plan uses 4 where it is supported, else 5, else 0 (environment PVX_FFT_MODE overrides); 1, 2 and 3 are witness kernels of the
 * tests: the product library answers PVX_ERR_UNSUPPORTED for them (tests/libpvx_witness.so carries them).
 */
int pvx_plan_set_fft_mode(pvx_plan* plan, int mode);
int pvx_plan_get_fft_mode(const pvx_plan* plan);
/* The HIP device that was current when the plan was created: its buffers, streams and events live there.  Every entry point
 * that takes a plan fails with PVX_ERR_INVALID when the calling thread is bound to another device (pvx_init(d), or a
 * pvx_batch worker's device) instead of running kernels on buffers of the wrong device. */
int pvx_plan_device(const pvx_plan* plan);
/* Which kernels the plan's last calls ran, as "analysis=<kernel>;tracker=<kernel>;synth=<kernel>" (a part is missing until its
 * call has run): k_fused_rev | k_fused_team | k_pv_rev | k_stft_pv | k_stft+k_phase_peaks | k_frames+rocfft+k_phase_peaks
 * for the analysis, k_synth_bodies<f64> | k_synth_bodies<f32> for the resynthesis.  For tests and benchmarks that must
 * know what they measured; the string lives in the plan and is valid until its next call. */
const char* pvx_plan_last_kernels(const pvx_plan* plan);
/*
 * Progress reporting: replaces Progress.update (pypevoc/ProgressDisplay.py:82-88), which the
 * reference calls once per frame (PV.py:250-254, 528).  The HOST entry points pvx_analyze and
 * pvx_harmonic_analyze call fn(frames_done, frames_total, user) after every launch chunk has
 * completed on the device and once when all results are back (done == total).  fn = NULL disables.
 * The asynchronous *_dev variants never call it.
 */
typedef void (*pvx_progress_fn)(int64_t frames_done, int64_t frames_total, void* user);
int pvx_plan_set_progress(pvx_plan* plan, pvx_progress_fn fn, void* user);

/*
 * Stage timing for bench.py's roofline line.  While enabled, hipEvents recorded on the launch
 * stream bracket every stage of every chunk.  pvx_plan_get_timing synchronises with those events
 * and returns, accumulated since the last call: ms[0] framing kernel, ms[1] rocFFT, ms[2]
 * phase/peak kernel, ms[3] fused kernel (fft modes 1-4, and mode 0's one-launch form); launches[i] = stage launches counted.
 */
int pvx_plan_set_timing(pvx_plan* plan, int enable);
int pvx_plan_get_timing(pvx_plan* plan, double* ms /*[4]*/, int64_t* launches /*[4]*/);

/*
 * run_pv on `nsig` equal-length signals resident in HBM (nsig = 1: the reference's single
 * signal).  Signal b starts at x + b * sig_stride samples and has nsamp samples.
 *   d_f, d_mag, d_ph, d_realph, d_binno : device float64 [nsig, F, npks]
 *   d_t, d_totalmag                     : device float64 [nsig, F]       (either may be NULL)
 *   d_prev0 : optional device float64 [nfft/2][2] (re, im) -- the spectrum `oldfft` holds before
 *             the first frame (PV.py:121, 209); NULL = zeros.  Used by the streaming
 *             calc_pv_frame mirror; only valid with nsig == 1.
 *   stream  : hipStream_t as void* (NULL = default stream).  Asynchronous.
 * Returns F (frames per signal) or a negative status.
 */
int64_t pvx_analyze_dev(pvx_plan* plan, const void* d_x, int x_dtype, int64_t nsamp,
                        int64_t nsig, int64_t sig_stride,
                        double* d_f, double* d_mag, double* d_ph, double* d_realph, double* d_binno,
                        double* d_t, double* d_totalmag, const double* d_prev0, void* stream);

/* Same with host buffers (copies x in, results out, synchronous).  prev0 may be NULL.
 * If last_spec is not NULL it receives the half spectrum of the last frame, float64 [nfft/2][2]
 * (the value PV.oldfft holds after run_pv, PV.py:209). */
int64_t pvx_analyze(pvx_plan* plan, const void* x, int x_dtype, int64_t nsamp,
                    int64_t nsig, int64_t sig_stride,
                    double* f, double* mag, double* ph, double* realph, double* binno,
                    double* t, double* totalmag, const double* prev0, double* last_spec);

/*
 * The host entry points keep their device buffers in the plan (grow-only; no hipMalloc per call), take the
 * input in chunks that fit PVX_MAX_DEVICE_BYTES (environment; default 2 GiB per buffer), double-buffered --
 * the copy of chunk i+1 and the results of chunk i-1 move while the kernels of chunk i run -- and carry the
 * last spectrum from chunk to chunk on the device (PV.py:209), so a signal larger than HBM runs and the
 * result does not depend on the chunking (bit for bit).  Calls of a few MB go through pinned staging and
 * synchronise once.
 *
 * Resident results: pvx_analyze_resident is pvx_analyze without the copy back -- the reference's arrays
 * stay in the plan, in HBM, in the layout of pvx_analyze_dev.  What follows in the reference's pipeline then
 * runs where the data is, on the plan's stream, and only what the caller asks for crosses PCIe:
 *   pvx_resident_fetch(plan, which, host)   one array; which = PVX_RES_F .. PVX_RES_TOTALMAG
 *   pvx_resident_ptr(plan, which)           its device address (valid until the plan's next analysis)
 *   pvx_track_resident                      PV.toSinSum (PV.py:299-322): returns the number of partials, the
 *                                           table stays resident; *max_end_frame = max(SinSum.end)
 *   pvx_resident_fetch_table                partial_id int32 [F, K], part_start / part_len int32 [P] (any may be NULL)
 *   pvx_synth_resident                      SinSum.synth (PV.py:1053-1070) from the resident arrays and table;
 *                                           w: host float64 [wlen = pvx_synth_len(max_end_frame, ...)]
 *   pvx_f0_resident                         PV.calc_f0 (PV.py:371-391): fm float64 [F], idx int32 [F]
 *   pvx_harmonic_power_resident             PV.calc_harmonic_power (PV.py:266-297, including the row indexing
 *                                           of :278; PVX_ERR_SIZE where the reference raises IndexError):
 *                                           hpower, nharmonics float64 [F, K]
 */
#define PVX_RES_F 0
#define PVX_RES_MAG 1
#define PVX_RES_PH 2
#define PVX_RES_REALPH 3
#define PVX_RES_BINNO 4
#define PVX_RES_T 5
#define PVX_RES_TOTALMAG 6
int64_t pvx_analyze_resident(pvx_plan* plan, const void* x, int x_dtype, int64_t nsamp, int64_t nsig,
                             int64_t sig_stride, const double* prev0, double* last_spec);
int pvx_resident_fetch(pvx_plan* plan, int which, double* host);
const double* pvx_resident_ptr(pvx_plan* plan, int which);
int64_t pvx_track_resident(pvx_plan* plan, double maxpitchjmp, int64_t* max_end_frame);
int pvx_resident_fetch_table(pvx_plan* plan, int32_t* partial_id, int32_t* part_start, int32_t* part_len);
int pvx_synth_resident(pvx_plan* plan, double sr, int hop_synth, double edge, int minframes, double* w, int64_t wlen);
/*
 * run_pv on many independent signals of any lengths, from host buffers, over one or more GPUs of THIS process -- the C-level
 * batch / multi-GPU entry (SURVEY.md 8(b) `pvx_analyze_batch`; the reference's equivalent is a Python loop over PV objects,
 * PV.py:213-264).  The path shards by signal and has no exchange step: the devices never talk to each
 * other, each worker (workers_per_device host threads per device, 0 = 4, each with its own plan and stream so one signal's
 * transfers run under another's kernels) takes the next signal off one queue, longest first, and its results go straight
 * into the caller's arrays.  Every signal's arrays are bit-identical to pvx_analyze of that signal alone.
 *   devices / ndev : the HIP devices to use (an entry may repeat); ndev = 0: the device of pvx_init
 *   items[i]       : x / nsamp in; f .. binno float64 [F, npks] and t, totalmag float64 [F] (either may be NULL) out, caller-
 *                    allocated with F = pvx_nframes(nsamp, nfft, hop); nframes = F or that signal's negative status;
 *                    device = the device that analysed it
 * pvx_batch_run returns the frames analysed (sum of F) or the first negative status (pvx_last_error names the signal); it
 * may be called again and again on one pvx_batch -- the plans and their buffers are kept -- but not concurrently.
 * pvx_analyze_batch = create + run + destroy.
 */
typedef struct pvx_batch pvx_batch;
typedef struct pvx_batch_item {
    const void* x;
    int64_t nsamp;
    double *f, *mag, *ph, *realph, *binno;
    double *t, *totalmag;
    int64_t nframes;
    int32_t device;
    int32_t reserved;
} pvx_batch_item;
int pvx_batch_create(pvx_batch** batch, double sr, int nfft, int hop, int npks, double pkthresh, const double* win /*nfft or NULL = hanning*/,
                     int precision, const int* devices, int ndev, int workers_per_device);
int64_t pvx_batch_run(pvx_batch* batch, int x_dtype, pvx_batch_item* items, int64_t nitems);
int pvx_batch_destroy(pvx_batch* batch);
int64_t pvx_analyze_batch(double sr, int nfft, int hop, int npks, double pkthresh, const double* win, int precision, int x_dtype,
                          pvx_batch_item* items, int64_t nitems, const int* devices, int ndev);

/* Page-locked host memory for result arrays.  pvx_synth_resident recognises such a destination: the waveform
 * (SinSum.synth's return value, PV.py:1070) is written straight into it by the DMA engine (with PVX_SYNTH_ZEROCOPY=1, up
 * to 4 MB, by the kernels' own stores); any other pointer goes through
 * the plan's pinned staging block and one more host copy.  NULL (and pvx_last_error) on failure. */
void* pvx_host_alloc(size_t bytes);
void pvx_host_free(void* p);
int pvx_f0_resident(pvx_plan* plan, double fmin, double fmax, double thr, double* fm, int32_t* idx);
int pvx_harmonic_power_resident(pvx_plan* plan, double f_threshold, double* hpower, double* nharmonics);

/* calc_fft_frame (PV.py:150-158): windowed, 1/wfact-normalised spectra of `nfr` frames starting
 * at sample positions pos[0..nfr); spec: host float64 [nfr][nfft/2+1][2] (bins 0..nfft/2; the
 * remaining bins of the reference's length-nfft result are the conjugate mirror). */
int pvx_stft_frames(pvx_plan* plan, const void* x, int x_dtype, int64_t nsamp,
                    const int64_t* pos, int64_t nfr, double* spec);

/* ---- PeakFinder (PF.py:35-74, 155-194, 113-136) -------------------------------------- */

/*
 * PeakFinder(y, npeaks=, minrattomax=, minval=) followed by filter_by_salience(rad) on `nrows`
 * independent rows y[r][0..n) (host float64).
 *   npeaks    <= 0: "not npeaks" -> n (PF.py:64-67)
 *   thr_kind  0: no threshold given (minamp = min(y)); 1: minrattomax = thr_val; 2: minval = thr_val
 *   rad       salience radius; < 0 skips filter_by_salience
 *   pos       int32 [nrows][cap]: findpos() positions, ascending (PF.py:189)
 *   keep      int8  [nrows][cap]: _keep after filter_by_salience
 *   count     int32 [nrows]
 *   cap       row capacity of pos/keep (>= min(npeaks, n) to hold everything)
 */
int pvx_peakfinder(const double* y, int64_t nrows, int n, int npeaks, int thr_kind, double thr_val,
                   int rad, int32_t* pos, int8_t* keep, int32_t* count, int cap);

/* ---- tracker: PV.toSinSum / SinSum.add_frame (PV.py:299-322, 871-957) ---------------- */

/*
 * Builds the partial table from the analysis arrays of one signal.
 *   f, mag            host float64 [F, K]
 *   maxpitchjmp       semitone threshold (the reference always uses 0.5: PV.py:320-321)
 *   partial_id        int32 [F, K]: partial index of every peak slot, -1 if the slot is empty
 *   part_start/len    int32 [cap]: first frame and number of points of each partial, in the
 *                     reference's creation order (PV.py:819-830)
 *   cap               capacity of part_start/part_len (F*K always suffices)
 * Returns the number of partials or a negative status.
 */
int64_t pvx_track(const double* f, const double* mag, int64_t F, int K, double maxpitchjmp,
                  int32_t* partial_id, int32_t* part_start, int32_t* part_len, int64_t cap);
int64_t pvx_track_dev(const double* d_f, const double* d_mag, int64_t F, int K, double maxpitchjmp,
                      int32_t* d_partial_id, int32_t* d_part_start, int32_t* d_part_len, int64_t cap,
                      void* stream);

/* ---- resynthesis: SinSum.synth / RegPartial.synth (PV.py:1053-1070, 684-756) ---------- */

/* Output length of SinSum.synth: (max(end)+2)*hop_synth + int(edge*hop_synth*nfft/hop_analysis/2). */
int64_t pvx_synth_len(int64_t max_end_frame, int nfft, int hop_analysis, int hop_synth, double edge);

/*
 * SinSum.synth(sr, hop_synth, edge, minframes, phase_preserve=True).
 *   f, mag, realph    host float64 [F, K] analysis arrays
 *   partial_id        int32 [F, K] from pvx_track;  part_start/part_len: int32 [P]
 *   w                 host float64 [wlen], wlen = pvx_synth_len(max(end), ...)
 */
int pvx_synth(const double* f, const double* mag, const double* realph, const int32_t* partial_id,
              int64_t F, int K, const int32_t* part_start, const int32_t* part_len, int64_t P,
              double sr, int nfft, int hop_analysis, int hop_synth, double edge, int minframes,
              double* w, int64_t wlen);
int pvx_synth_dev(const double* d_f, const double* d_mag, const double* d_realph,
                  const int32_t* d_partial_id, int64_t F, int K,
                  const int32_t* d_part_start, const int32_t* d_part_len, int64_t P,
                  double sr, int nfft, int hop_analysis, int hop_synth, double edge, int minframes,
                  double* d_w, int64_t wlen, void* stream);
/*
 * The same with flags.  PVX_SYNTH_NO_PHCOR: RegPartial.synth of a partial built with fstep=None
 * (PV.py:710-713): no frequency-slope phase correction (phcor = phcornext = 0); nfft / hop_analysis then
 * only carry the overlap (hop_analysis / nfft).
 */
#define PVX_SYNTH_NO_PHCOR 1
/* PVX_SYNTH_F32: the sample loop of the partials' bodies in float32 (what pvx_synth_resident does for a plan at precision 32):
 * the seeds of every run of 32 samples still come from the float64 closed form; waveform within 1e-4 max|w| of the float64 one
 * (measured: see DESIGN.md), at about half the time. */
#define PVX_SYNTH_F32 2
int pvx_synth_flags(const double* f, const double* mag, const double* realph, const int32_t* partial_id,
                    int64_t F, int K, const int32_t* part_start, const int32_t* part_len, int64_t P,
                    double sr, int nfft, int hop_analysis, int hop_synth, double edge, int minframes,
                    double* w, int64_t wlen, int flags);
int pvx_synth_dev_flags(const double* d_f, const double* d_mag, const double* d_realph,
                        const int32_t* d_partial_id, int64_t F, int K,
                        const int32_t* d_part_start, const int32_t* d_part_len, int64_t P,
                        double sr, int nfft, int hop_analysis, int hop_synth, double edge, int minframes,
                        double* d_w, int64_t wlen, void* stream, int flags);

/* ---- PVHarmonic.run_pv / calc_pv_frame (PV.py:419-535) --------------------------------------
 *
 * Phase vocoder sampled at the multiples of a caller-supplied fundamental instead of PeakFinder
 * peaks.  Single signal.  Uses the plan's general path (window -> rocFFT -> k_harmonic_rows), any
 * nfft, precision 32 or 64.
 *   f0, nf0        HOST float64 [nf0 >= F]: fundamental of every frame (PV.py:507 indexes it by frame
 *                  number; a shorter array is the reference's IndexError -> PVX_ERR_SIZE).  Frames
 *                  with f0 <= 0 or NaN are skipped: zero rows, residual NaN, and they do NOT become
 *                  the "previous spectrum" of later frames (PV.py:509, 491).
 *   fmin           PVHarmonic.fmin (30.0 in the reference, PV.py:421)
 *   f, mag, ph     float64 [F, npks]: first npks harmonics (frequency may be NaN where the reference's
 *                  is: x/0 with a zero component); residual, t: float64 [F]
 *   prev0          optional [nfft/2][2] spectrum preceding the first analysed frame; last_spec
 *                  (host variant) receives PVHarmonic.oldfft after the loop.
 * A valid f0 below half a bin is rejected (PVX_ERR_INVALID).  Returns F or a negative status.
 */
int64_t pvx_harmonic_analyze(pvx_plan* plan, const void* x, int x_dtype, int64_t nsamp,
                             const double* f0, int64_t nf0, double fmin,
                             double* f, double* mag, double* ph, double* residual, double* t,
                             const double* prev0, double* last_spec);
int64_t pvx_harmonic_analyze_dev(pvx_plan* plan, const void* d_x, int x_dtype, int64_t nsamp,
                                 const double* f0, int64_t nf0, double fmin,
                                 double* d_f, double* d_mag, double* d_ph, double* d_residual,
                                 double* d_t, const double* d_prev0, void* stream);

/* ---- windowed reductions with the analysis framing ---------------------
 *
 * Frames start at i*hop for i*hop < n - wlen: pvx_nframes(n, wlen, hop) of them.  float64.
 *
 * pvx_heterodyne: Heterodyne.heterodyne(x, hetsig, wind, hop) (pypevoc/Heterodyne.py:35-60):
 *   out[i] = 2 * sum_j x[i*hop+j] * hetsig[i*hop+j] * wind[j] / sum(wind)   complex, [nfr][2]
 *   icent[i] = i*hop + wlen/2 (optional).  hetsig: complex128 [n] as [n][2].
 *   (SoundUtils.Heterodyn / HeterodynWithF0Track, SoundUtils.py:106-138, are this with
 *    hetsig = exp(+-2j*pi*phase) built by the caller and wind = windfunc(nwind).)
 * pvx_rms_frames: SoundUtils.RMSWind (SoundUtils.py:71-103):
 *   out[i] = sqrt(sum_j (x[i*hop+j]*wind[j])**2 / sum(wind**2))
 * `wind` is a HOST array in both variants.  Return nfr or a negative status.
 */
int64_t pvx_heterodyne(const double* x, const double* hetsig, int64_t n, const double* wind,
                       int wlen, int hop, double* out, int64_t* icent);
int64_t pvx_heterodyne_dev(const double* d_x, const double* d_hetsig, int64_t n, const double* wind,
                           int wlen, int hop, double* d_out, int64_t* d_icent, void* stream);
int64_t pvx_rms_frames(const double* x, int64_t n, const double* wind, int wlen, int hop, double* out);
int64_t pvx_rms_frames_dev(const double* d_x, int64_t n, const double* wind, int wlen, int hop,
                           double* d_out, void* stream);

/* pvx_funcwind: SoundUtils.FuncWind(func, x, sr, nwind, nhop, power, windfunc) (pypevoc/SoundUtils.py:42-69)
 * for the named reducers: out[i] = func(x[i*hop : i*hop+wlen] * wind) / divisor, with
 * divisor = sum(wind**power) for power > 0, else 1 (SoundUtils.py:55-58; computed by the caller).
 * x_complex != 0: x is complex128 [n] given as [n][2] (Heterodyn passes x * sinsig, :112): SUM and MEAN then
 * write complex results [nfr][2], STD / VAR real ones (numpy: mean(abs(xw - mean(xw))**2)); MAX / MIN of
 * complex frames -> PVX_ERR_UNSUPPORTED.  An arbitrary Python callable has no device form: the Python mirror
 * raises TypeError for anything but these six.  Returns nfr or a negative status. */
typedef enum { PVX_FW_SUM = 0, PVX_FW_MEAN = 1, PVX_FW_MAX = 2, PVX_FW_MIN = 3, PVX_FW_STD = 4, PVX_FW_VAR = 5 } pvx_funcwind_op;
int64_t pvx_funcwind(const double* x, int x_complex, int64_t n, const double* wind, int wlen, int hop,
                     int func, double divisor, double* out);
int64_t pvx_funcwind_dev(const double* d_x, int x_complex, int64_t n, const double* wind, int wlen, int hop,
                         int func, double divisor, double* d_out, void* stream);

/* ---- multi-GPU result gather: compact wire format --------------------------------------
 *
 * The reference has no multi-device path; its results are the five float64 [F, K] arrays of
 * PV.run_pv (PV.py:256-264), 40 B per peak slot.  For the gather of sharded results to one GPU
 * (RCCL over xGMI, link-bound) a shard's rows are packed to 18 B (precision 32) or 26 B
 * (precision 64) per slot: f (f64), mag, ph (f32|f64), binno (u16) + totalmag (f64) per row;
 * realph = ph + pi*(fbin[binno] - f)/fstep (PV.py:146, 207) is recomputed by the receiver.
 * pvx_unpack_rows_dev(pvx_pack_rows_dev(x)) == x bit for bit for anything pvx_analyze* produced
 * with the same plan parameters.  `rows` = frames of all signals of the shard; all pointers are
 * device memory; launches are asynchronous on `stream`.
 */
/* Wire format of a plan: 1 (default) as above; 2, for plans at precision 32: 14 B per slot.  There a peak's frequency is a float64
 * function of its bin and ONE float32 value (the unwrapped phase offset of PV.py:140-147, or one of twelve cases after a silent
 * frame): the block carries that value instead of f and the receiver evaluates the kernels' own expression -- the same bits.  Every
 * function below follows the plan's format; pack of arrays that no precision-32 analysis wrote leaves NaN frequencies in format 2.
 * PVX_ERR_UNSUPPORTED for format 2 on a precision-64 plan. */
int pvx_plan_set_wire_format(pvx_plan* plan, int format);
int pvx_plan_get_wire_format(const pvx_plan* plan);
int64_t pvx_wire_bytes(const pvx_plan* plan, int64_t rows);
int pvx_pack_rows_dev(const pvx_plan* plan, int64_t rows, const double* d_f, const double* d_mag,
                      const double* d_ph, const double* d_binno, const double* d_totalmag,
                      void* d_wire, void* stream);
/* run_pv of nsig device-resident signals straight into a wire block of pvx_wire_bytes(plan, nsig * F) bytes: what a rank of a
 * multi-GPU job hands to the gather (bench.py).  The fused float32 kernel of nfft 512 / 1024 / 2048 writes the block itself -- 18 (14 in
 * format 2) bytes per slot instead of 40, and no packing pass behind the analysis --; any other plan analyses into a plan-owned block and packs
 * it.  pvx_unpack_rows_dev of the block gives the arrays pvx_analyze_dev would have written, bit for bit.  Returns F. */
int64_t pvx_analyze_dev_wire(pvx_plan* plan, const void* d_x, int x_dtype, int64_t nsamp, int64_t nsig, int64_t sig_stride,
                             void* d_wire, void* stream);
int pvx_unpack_rows_dev(const pvx_plan* plan, int64_t rows, const void* d_wire, double* d_f,
                        double* d_mag, double* d_ph, double* d_realph, double* d_binno,
                        double* d_totalmag, void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* PVX_H */
