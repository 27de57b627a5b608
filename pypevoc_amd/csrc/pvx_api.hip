// pvx_api.hip -- host side of the C ABI declared in include/pvx.h: device selection, analysis
// plans (constants of PV.__init__, pypevoc/PVAnalysis.py:72-131; rocFFT plan; device workspace),
// chunked launch sequence of run_pv (PV.py:213-264) and the host-buffer convenience wrappers.
//
// There is deliberately no CPU path in this library: every entry point needs a HIP device.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <tuple>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include <time.h>

#include "pvx_internal.h"

// float64 -> float32 while staging (RNE, what v_cvt_f32_f64 does on the device): AVX2 where the host has it
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static void narrow_avx2(const double* src, float* dst, size_t n) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m128 a = _mm256_cvtpd_ps(_mm256_loadu_pd(src + i)), b = _mm256_cvtpd_ps(_mm256_loadu_pd(src + i + 4));
        _mm256_storeu_ps(dst + i, _mm256_set_m128(b, a));
    }
    for (; i < n; i++) dst[i] = (float)src[i];
}
#endif
static void narrow_f64_f32(const double* src, float* dst, size_t n) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) { narrow_avx2(src, dst, n); return; }
#endif
    for (size_t i = 0; i < n; i++) dst[i] = (float)src[i];
}

// PVX_TRACE=1: host-side time stamps of the small-call paths on stderr (where a sub-millisecond round trip goes)
namespace {
struct HostTrace {
    bool on;
    const char* what;
    double t0, last;
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
    explicit HostTrace(const char* w) : on(getenv("PVX_TRACE") != nullptr), what(w), t0(0), last(0) { if (on) { t0 = last = now(); } }
    void mark(const char* label) { if (on) { const double t = now(); fprintf(stderr, "[pvx %s] %-28s +%7.1f us (%7.1f)\n", what, label, t - last, t - t0); last = t; } }
};
}  // namespace

// ---- errors ---------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void pvx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* pvx_last_error(void) { return g_err; }
extern "C" int pvx_version(void) { return PVX_VERSION; }

// ---- device ---------------------------------------------------------------------------------
static std::mutex g_mu;
static int g_device = -1;
static thread_local int t_device = -1;   // a worker thread of pvx_batch_run is bound to its own device (the process default stays g_device)
static bool g_fft_setup = false;
static char g_devname[256] = "";

extern "C" int pvx_init(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        pvx_set_error("no HIP device available (%s); libpvx_hip has no CPU fallback",
                      e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return PVX_ERR_NO_DEVICE;
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= n) { pvx_set_error("device %d out of range (%d devices)", device, n); return PVX_ERR_INVALID; }
    PVX_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    PVX_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    snprintf(g_devname, sizeof(g_devname), "%s (%s)", prop.name, prop.gcnArchName);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        pvx_set_error("device %d is %s; libpvx_hip carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
        return PVX_ERR_NO_DEVICE;
    }
    if (!g_fft_setup) {
        PVX_FFT_CHECK(rocfft_setup());
        g_fft_setup = true;
    }
    g_device = device;
    return PVX_OK;
}

extern "C" const char* pvx_device_name(void) { return g_devname; }
extern "C" int pvx_device(void) { return g_device; }

int pvx_require_device() {
    const int d = t_device >= 0 ? t_device : g_device;
    if (d >= 0) {
        // the calling thread may be new: bind it
        if (hipSetDevice(d) != hipSuccess) { pvx_set_error("hipSetDevice(%d) failed", d); return PVX_ERR_NO_DEVICE; }
        return PVX_OK;
    }
    return pvx_init(-1);
}

struct pvx_plan;
static int plan_device(const pvx_plan* p);
// entry points that take a plan: the thread is bound as above, and the plan must have been created under the device the
// call runs on -- its buffers, streams and events live there (a worker of pvx_batch_run creates its plan under its own
// device; anything else handing a plan of device a to a thread bound to device b is refused, not run)
int pvx_require_plan_device(const pvx_plan* p) {
    const int rc = pvx_require_device();
    if (rc != PVX_OK || !p) return rc;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { pvx_set_error("hipGetDevice failed"); return PVX_ERR_HIP; }
    if (cur != plan_device(p)) {
        pvx_set_error("the plan was created on device %d and is used under device %d (pvx_init / the batch worker's device): create the plan under the device that runs it",
                      plan_device(p), cur);
        return PVX_ERR_INVALID;
    }
    return PVX_OK;
}

extern "C" int64_t pvx_nframes(int64_t nsamp, int nfft, int hop) {
    // PV.py:223-249: pos = 0, hop, 2 hop, ... while pos < nsamp - nfft (strict)
    const int64_t maxpos = nsamp - nfft;
    if (maxpos <= 0 || hop <= 0 || nfft <= 0) return 0;
    return (maxpos + hop - 1) / hop;
}

// ---- plan -----------------------------------------------------------------------------------
struct pvx_plan {
    int device = -1;          // the HIP device current when the plan was created: every buffer / stream / event below lives there
    double sr = 0, pkthresh = 0, wfact = 0, fstep = 0, dt = 0;
    int nfft = 0, hop = 0, npks = 0, N2 = 0, precision = 32, fft_mode = 0;
    int64_t max_rows = 0;     // rows per launch (without the halo row)
    bool rows_from_env = false;   // ... as PVX_MAX_ROWS set them
    const char* last_analysis = nullptr;   // kernels of the last calls (pvx_plan_last_kernels)
    const char* last_synth = nullptr;
    std::string last_kernels;
    int64_t ldi = 0, ldo = 0; // workspace row pitches (elements / complex elements)
    std::vector<double> win;  // caller's window
    bool win_symmetric = false;  // win[n] == win[nfft-1-n] for every n
    void* d_win = nullptr;    // window / wfact in the working precision
    double* d_wfbin = nullptr;
    void* d_frames = nullptr; // [max_rows+1][ldi]
    void* d_spec = nullptr;   // [max_rows+1][ldo] complex
    void* d_cand_y = nullptr;              // the split transform's candidate peaks per workspace row (pvx_stft.h): |X|^2 [max_rows+1][cand_cap],
    unsigned short* d_cand_bin = nullptr;  // bins, and [max_rows+1][4] max / min / sum / count
    double* d_cand_stats = nullptr;
    int cand_cap = 0;
    void* d_work = nullptr;
    size_t work_bytes = 0;
    int64_t ws_bytes = 0;
    rocfft_plan fft = nullptr;
    rocfft_execution_info info = nullptr;
    bool rocfft_ready = false;   // frames/spectrum workspace + rocFFT plan are created on first use
    bool use_stft = false;       // float64, nfft 512..2048: k_stft writes the spectrum rows (no frame buffer, no rocFFT)
    bool use_stft_pv = false;    // ... and finds the peaks in the same launch (k_stft_pv.hip)
    bool use_pv_team = false;    // float64 at nfft 4096 / 8192, npks <= 64, hop nfft/4 or nfft/2: the same with a team of waves per frame (k_pv_team.hip)
    bool use_pv_rev = false;     // ... without writing a spectrum row: rows walked downwards, the row at hand on chip (k_pv_rev.hip; npks <= 64)
    double* d_lastspec = nullptr;   // k_pv_rev: [N2][2] the spectrum of the one row a call asks for (chunk carry, last_spec)
    bool last_from_rev = false;  // the last analyze_rows() left its requested spectrum row in d_lastspec
    void* d_pvstage = nullptr;   // k_pv_rev at nfft 2048: the kept peaks' values between the frames (PvRevParams::stage)
    size_t pvstage_cap = 0;
    int wire_fmt = 1;            // the gather's wire format (k_wire.hip): 1 = f as float64, 2 = the float32 it is computed from (pvx_plan_set_wire_format)
    double* d_wiretmp = nullptr; // pvx_analyze_dev_wire on plans whose kernels do not write the wire format themselves: the result block that is then packed
    size_t wiretmp_cap = 0;
    int64_t rocfft_rows = 0;     // rows of the rocFFT workspace (2 when only pvx_stft_frames uses it)
    bool rocfft_small = false;   // ... and its output then goes to d_rspec, not to the analysis' d_spec
    void* d_rspec = nullptr;     // rocFFT output when the main spectrum workspace belongs to k_stft
    void* d_twiddle64 = nullptr; // complex<T>[nfft] W_nfft^j for k_stft (T = the plan's precision)
    void* d_twiddle = nullptr;   // float2[2048] W_2048^j for the fused kernel
    float* d_specrow = nullptr;  // 1024 complex: spectrum of one requested row (fused mode)
    float* spec_host = nullptr;  // when set: the fused kernels write that row straight into this page-locked host block
    // PVHarmonic: per-frame f0 / previous-row tables and the carried spectrum of the last valid frame
    double* d_hf0 = nullptr;
    unsigned char* d_hx = nullptr;         // pvx_harmonic_analyze: the host signal's device copy (kept: allocating and freeing
    size_t hx_cap = 0;                     // hundreds of MB per call cost more than the analysis)
    int32_t* d_hprev = nullptr;
    void* d_carry = nullptr;
    int64_t harm_cap = 0;
    // progress callback of the host entry points
    pvx_progress_fn progress_fn = nullptr;
    void* progress_user = nullptr;
    bool progress_live = false;   // inside a host entry point: chunk completions are reported
    int64_t fused_blocks = 0;    // PVX_FUSED_BLOCKS override
    int frames_per_wave = 2;     // k_phase_peaks: frames a wave handles one after the other (latency floor of a chunked launch; PVX_FPW)
    // ---- host entry points: plan-owned, grow-only buffers (no hipMalloc / hipFree per call)
    hipStream_t s_host = nullptr;          // non-blocking stream of the host entry points
    hipStream_t s_copy = nullptr;          // second stream: DMA of finished waveform slices under the next slice's kernel
    hipEvent_t ev_done[2] = {nullptr, nullptr};
    void* d_in[2] = {nullptr, nullptr};    // input chunks (double-buffered: H2D of chunk i+1 under the kernels of chunk i)
    size_t in_cap[2] = {0, 0};
    double* d_out[2] = {nullptr, nullptr}; // result blocks of a chunk when the results stream back to the host
    size_t out_cap[2] = {0, 0};
    void* h_pin = nullptr;                 // pinned staging of small calls (input, then the result block)
    size_t pin_cap = 0;
    void* h_ring = nullptr;                // pinned ring of the threaded staging of large host transfers (kStageThreads x 2 slots)
    hipEvent_t ev_ring[16] = {};
    double* d_prev = nullptr;              // [N2][2] spectrum carried from one input chunk to the next (PV.py:209)
    // resident results (pvx_analyze_resident): the reference's arrays stay in HBM for the tracker, the
    // resynthesis and the frame descriptors; only what the caller fetches crosses PCIe
    double* d_res = nullptr;
    size_t res_cap = 0;
    int64_t res_F = 0, res_nsig = 0;
    bool res_valid = false;
    int32_t *d_pid = nullptr, *d_pst = nullptr, *d_pln = nullptr;     // resident partial table
    size_t trk_cap = 0;                    // entries of d_pid / d_pst / d_pln
    void* d_tws = nullptr;                 // tracker workspace
    size_t tws_cap = 0;
    int64_t res_P = -1, res_maxend = -1;
    double* d_w = nullptr;                 // resynthesised waveform
    size_t w_cap = 0;
    void* d_sws = nullptr;                 // resynthesis workspace (k_synth.hip)
    size_t sws_cap = 0;
    unsigned sws_gen = 0;                  // calls on this workspace since it was allocated
    void* d_stash = nullptr;               // k_fused_rev: first spectra of its waves, for the waves below them (FusedParams::stash)
    size_t stash_cap = 0;
    float* d_x32 = nullptr;                // a device-resident float64 signal narrowed for the fused float32 kernels
    size_t x32_cap = 0;
    void* d_desc = nullptr;                // descriptor outputs (f0 / harmonic power)
    size_t desc_cap = 0;
    // optional stage timing (bench): events[4*i..4*i+3] bracket the three stages of chunk i
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<std::pair<int, size_t>> ev_spans;   // (stage, index of the start event; end = next event)
};
static int plan_device(const pvx_plan* p) { return p->device; }
extern "C" int pvx_plan_device(const pvx_plan* plan) { return plan ? plan->device : PVX_ERR_INVALID; }
extern "C" const char* pvx_plan_last_kernels(const pvx_plan* plan) {
    if (!plan) return "";
    pvx_plan* p = const_cast<pvx_plan*>(plan);
    p->last_kernels.clear();
    if (p->last_analysis) { p->last_kernels += "analysis="; p->last_kernels += p->last_analysis; }
    if (p->last_synth) { if (!p->last_kernels.empty()) p->last_kernels += ";"; p->last_kernels += "synth="; p->last_kernels += p->last_synth; }
    return p->last_kernels.c_str();
}

static size_t real_size(int precision) { return precision == 32 ? 4 : 8; }

static void plan_free(pvx_plan* p) {
    if (!p) return;
    for (hipEvent_t e : p->ev_pool) (void)hipEventDestroy(e);
    if (p->info) rocfft_execution_info_destroy(p->info);
    if (p->fft) rocfft_plan_destroy(p->fft);
    if (p->d_win) (void)hipFree(p->d_win);
    if (p->d_wfbin) (void)hipFree(p->d_wfbin);
    if (p->d_frames) (void)hipFree(p->d_frames);
    if (p->d_spec) (void)hipFree(p->d_spec);
    if (p->d_cand_y) (void)hipFree(p->d_cand_y);
    if (p->d_cand_bin) (void)hipFree(p->d_cand_bin);
    if (p->d_cand_stats) (void)hipFree(p->d_cand_stats);
    if (p->d_work) (void)hipFree(p->d_work);
    if (p->d_rspec) (void)hipFree(p->d_rspec);
    if (p->d_twiddle64) (void)hipFree(p->d_twiddle64);
    if (p->d_twiddle) (void)hipFree(p->d_twiddle);
    if (p->d_specrow) (void)hipFree(p->d_specrow);
    if (p->d_stash) (void)hipFree(p->d_stash);
    if (p->d_hf0) (void)hipFree(p->d_hf0);
    if (p->d_hx) (void)hipFree(p->d_hx);
    if (p->d_hprev) (void)hipFree(p->d_hprev);
    if (p->d_carry) (void)hipFree(p->d_carry);
    if (p->d_lastspec) (void)hipFree(p->d_lastspec);
    if (p->d_pvstage) (void)hipFree(p->d_pvstage);
    if (p->d_wiretmp) (void)hipFree(p->d_wiretmp);
    for (int i = 0; i < 2; i++) {
        if (p->d_in[i]) (void)hipFree(p->d_in[i]);
        if (p->d_out[i]) (void)hipFree(p->d_out[i]);
        if (p->ev_done[i]) (void)hipEventDestroy(p->ev_done[i]);
    }
    if (p->h_pin) (void)hipHostFree(p->h_pin);
    if (p->h_ring) (void)hipHostFree(p->h_ring);
    for (int i = 0; i < 16; i++) if (p->ev_ring[i]) (void)hipEventDestroy(p->ev_ring[i]);
    if (p->d_prev) (void)hipFree(p->d_prev);
    if (p->d_res) (void)hipFree(p->d_res);
    if (p->d_pid) (void)hipFree(p->d_pid);
    if (p->d_pst) (void)hipFree(p->d_pst);
    if (p->d_pln) (void)hipFree(p->d_pln);
    if (p->d_tws) (void)hipFree(p->d_tws);
    if (p->d_w) (void)hipFree(p->d_w);
    if (p->d_sws) (void)hipFree(p->d_sws);
    if (p->d_x32) (void)hipFree(p->d_x32);
    if (p->d_desc) (void)hipFree(p->d_desc);
    if (p->s_host) (void)hipStreamDestroy(p->s_host);
    if (p->s_copy) (void)hipStreamDestroy(p->s_copy);
    delete p;
}

#include "build_sha.inc"
extern "C" const char* pvx_build_fingerprint(void) { return PVX_BUILD_SHA; }

extern "C" int pvx_plan_destroy(pvx_plan* plan) {
    plan_free(plan);
    return PVX_OK;
}

extern "C" int64_t pvx_plan_workspace_bytes(const pvx_plan* plan) { return plan ? plan->ws_bytes : 0; }

int pvx_resident_blocks(const void* fn, int threads, size_t lds) {
    static std::mutex mu;
    static std::map<std::tuple<int, const void*, int, size_t>, int> seen;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    const auto key = std::make_tuple(dev, fn, threads, lds);
    std::lock_guard<std::mutex> lk(mu);
    auto it = seen.find(key);
    if (it != seen.end()) return it->second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, lds) != hipSuccess) { (void)hipGetLastError(); nb = 0; }
    seen[key] = nb;
    return nb;
}

extern "C" int pvx_plan_set_fft_mode(pvx_plan* plan, int mode) {
    if (!plan) { pvx_set_error("null plan"); return PVX_ERR_INVALID; }
    if (mode < 0 || mode > 5) { pvx_set_error("unknown fft mode %d", mode); return PVX_ERR_INVALID; }
    if (mode == 5 && !pvx_fused_team_supported(plan->nfft, plan->precision, plan->npks)) {
        pvx_set_error("the team kernel handles nfft in {4096, 8192} at precision=32 with npks <= 128 (this plan: nfft=%d precision=%d npks=%d)", plan->nfft, plan->precision, plan->npks);
        return PVX_ERR_UNSUPPORTED;
    }
    if (mode == 4 && !pvx_fused_rev_supported(plan->nfft, plan->precision, plan->npks)) {
        pvx_set_error("the descending-order fused kernel handles nfft in {512, 1024, 2048} at precision=32 while npks leaves it enough LDS (this plan: nfft=%d precision=%d npks=%d)", plan->nfft, plan->precision, plan->npks);
        return PVX_ERR_UNSUPPORTED;
    }
    if (mode == 3 && !pvx_fused_ring_supported(plan->nfft, plan->precision, plan->npks)) {
        pvx_set_error("fft mode 3 (the ring kernel, a witness of the bit-identity tests: tests/libpvx_witness.so, not libpvx_hip.so) handles nfft in {512, 1024, 2048} at precision=32 while npks leaves it enough LDS (this plan: nfft=%d precision=%d npks=%d)", plan->nfft, plan->precision, plan->npks);
        return PVX_ERR_UNSUPPORTED;
    }
    if (mode == 2 && !pvx_fused_mw_supported(plan->nfft, plan->precision, plan->npks)) {
        pvx_set_error("fft mode 2 (several waves per frame, a witness since the general path took its cases: tests/libpvx_witness.so, not libpvx_hip.so) handles nfft in {2048, 4096, 8192} at precision=32 (this plan: nfft=%d precision=%d)", plan->nfft, plan->precision);
        return PVX_ERR_UNSUPPORTED;
    }
    if (mode == 1 && !pvx_fused_supported(plan->nfft, plan->precision, plan->npks)) {
        pvx_set_error("fft mode 1 (one wave per frame over two buffers, a witness of the bit-identity tests: tests/libpvx_witness.so, not libpvx_hip.so) handles nfft in {512, 1024, 2048} at precision=32 (this plan: nfft=%d precision=%d)", plan->nfft, plan->precision);
        return PVX_ERR_UNSUPPORTED;
    }
    plan->fft_mode = mode;
    return PVX_OK;
}

extern "C" int pvx_plan_get_fft_mode(const pvx_plan* plan) { return plan ? plan->fft_mode : PVX_ERR_INVALID; }

extern "C" int pvx_plan_create(pvx_plan** out, double sr, int nfft, int hop, int npks, double pkthresh,
                               const double* win, int precision, int64_t max_rows) {
    if (!out) { pvx_set_error("null plan pointer"); return PVX_ERR_INVALID; }
    *out = nullptr;
    int rc = pvx_require_device();
    if (rc != PVX_OK) return rc;
    if (nfft < 4 || hop <= 0 || npks <= 0 || !(sr > 0) || (precision != 32 && precision != 64)) {
        pvx_set_error("invalid analysis parameters (sr=%g nfft=%d hop=%d npks=%d precision=%d)", sr, nfft, hop, npks, precision);
        return PVX_ERR_INVALID;
    }
    const int64_t rows_hint = max_rows;
    pvx_plan* p = new pvx_plan();
    if (hipGetDevice(&p->device) != hipSuccess) { delete p; pvx_set_error("hipGetDevice failed"); return PVX_ERR_HIP; }
    p->sr = sr; p->nfft = nfft; p->hop = hop; p->npks = npks; p->pkthresh = pkthresh;
    p->precision = precision;
    p->N2 = nfft / 2;                                   // PV.py:88
    p->win.resize(nfft);
    if (win) {
        memcpy(p->win.data(), win, sizeof(double) * nfft);
    } else {                                            // np.hanning(nfft)
        for (int i = 0; i < nfft; i++) {
            const double n = (double)(1 - nfft + 2 * i);
            p->win[i] = 0.5 + 0.5 * cos(3.141592653589793238462643383279502884 * n / (double)(nfft - 1));
        }
    }
    double wsum2 = 0.0;                                 // PV.py:99 (Python sum: left to right)
    for (int i = 0; i < nfft; i++) wsum2 = wsum2 + p->win[i] * p->win[i];
    p->win_symmetric = true;
    for (int i = 0; i < nfft / 2; i++) p->win_symmetric = p->win_symmetric && (p->win[i] == p->win[nfft - 1 - i]);
    p->wfact = sqrt(wsum2 * nfft) / 2.0;                // PV.py:102
    p->fstep = sr / (double)nfft;                       // PV.py:105
    p->dt = (double)hop / sr;                           // PV.py:108
    if (!(p->wfact > 0)) { pvx_set_error("window has no energy"); plan_free(p); return PVX_ERR_INVALID; }

    const size_t rs = real_size(precision);
    // Launch size: the frame + spectrum workspace of one launch (~96 MiB by default) stays inside
    // the 256 MiB Infinity Cache, so rocFFT and the peak kernel read what the previous kernel just
    // wrote from on-die memory.  `max_rows` from the caller is a "rows needed" hint and only ever
    // shrinks the workspace; PVX_MAX_ROWS (environment) overrides the cap for tuning.
    int64_t cap_rows = (int64_t)(96.0 * 1024 * 1024 / ((double)nfft * rs * 2.0));
    if (cap_rows < 256) cap_rows = 256;
    if (cap_rows > 65536) cap_rows = 65536;
    if (const char* e = getenv("PVX_MAX_ROWS")) {
        const long long v = atoll(e);
        if (v >= 2) { cap_rows = (int64_t)v; p->rows_from_env = true; }
    }
    if (const char* e = getenv("PVX_FPW")) {
        const int v = atoi(e);
        if (v >= 1) p->frames_per_wave = v;
    }
    if (max_rows <= 0 || max_rows > cap_rows) max_rows = cap_rows;
    if (max_rows < 2) max_rows = 2;
    p->max_rows = max_rows;
    p->ldi = (nfft + 3) & ~3;
    p->ldo = ((nfft / 2 + 1) + 1) & ~1;                 // rocFFT writes nfft/2+1 bins; keep rows 16-byte aligned

    // device constants
    {
        std::vector<double> wf(p->N2 > 0 ? p->N2 : 1);
        const double pi2 = 2.0 * 3.141592653589793238462643383279502884;
        for (int k = 0; k < p->N2; k++) {
            const double fbin = (double)k * p->fstep;                   // PV.py:114
            const double dthetabin = pi2 * fbin * p->dt;                // PV.py:116
            wf[k] = nearbyint(dthetabin / pi2) * pi2;                   // PV.py:118 (np.round: half to even)
        }
        if (hipMalloc(&p->d_wfbin, sizeof(double) * wf.size()) != hipSuccess) { pvx_set_error("hipMalloc(wfbin) failed"); plan_free(p); return PVX_ERR_ALLOC; }
        if (hipMemcpy(p->d_wfbin, wf.data(), sizeof(double) * wf.size(), hipMemcpyHostToDevice) != hipSuccess) { pvx_set_error("hipMemcpy(wfbin) failed"); plan_free(p); return PVX_ERR_HIP; }
        // window with 1/wfact folded in (fft(x*win)/wfact, PV.py:156-157)
        if (hipMalloc(&p->d_win, rs * nfft) != hipSuccess) { pvx_set_error("hipMalloc(win) failed"); plan_free(p); return PVX_ERR_ALLOC; }
        hipError_t e;
        if (precision == 32) {
            std::vector<float> w(nfft);
            for (int i = 0; i < nfft; i++) w[i] = (float)(p->win[i] / p->wfact);
            e = hipMemcpy(p->d_win, w.data(), rs * nfft, hipMemcpyHostToDevice);
        } else {
            std::vector<double> w(nfft);
            for (int i = 0; i < nfft; i++) w[i] = p->win[i] / p->wfact;
            e = hipMemcpy(p->d_win, w.data(), rs * nfft, hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) { pvx_set_error("hipMemcpy(win) failed"); plan_free(p); return PVX_ERR_HIP; }
    }

    // float64 fused STFT kernel (k_stft.hip): its own twiddle table; the spectrum workspace is the only intermediate
    // array left, so a launch may cover far more rows than the Infinity-Cache-sized chunks of the three-kernel path
    if (pvx_stft_supported(nfft, precision) && !getenv("PVX_NO_STFT")) {
        std::vector<double> tw(2 * (size_t)nfft);
        const double pi = 3.141592653589793238462643383279502884;
        if (nfft % 8 == 0) {
            // W^j = cos - i sin from libm on the first octant only; the rest by the table's exact symmetries
            // W^(N/4 - j) = (-Im, -Re) W^j and W^(j + N/4) = -i W^j, so that a kernel may keep an eighth of the table and
            // derive the other entries (k_stft_pv.hip) and still use the very values the full-table kernels read
            const int q = nfft / 4, o = nfft / 8;
            for (int j = 0; j <= o; j++) { tw[2 * j] = cos(2.0 * pi * j / (double)nfft); tw[2 * j + 1] = -sin(2.0 * pi * j / (double)nfft); }
            tw[2 * o] = 0.70710678118654752440; tw[2 * o + 1] = -0.70710678118654752440;      // W^(N/8) is its own mirror image
            for (int j = o + 1; j <= q; j++) { tw[2 * j] = -tw[2 * (q - j) + 1]; tw[2 * j + 1] = -tw[2 * (q - j)]; }
            for (int j = q + 1; j < nfft; j++) { tw[2 * j] = tw[2 * (j - q) + 1]; tw[2 * j + 1] = -tw[2 * (j - q)]; }
        } else
        for (int j = 0; j < nfft; j++) { tw[2 * j] = cos(2.0 * pi * j / (double)nfft); tw[2 * j + 1] = -sin(2.0 * pi * j / (double)nfft); }
        std::vector<float> twf(tw.begin(), tw.end());
        const void* src = precision == 64 ? (const void*)tw.data() : (const void*)twf.data();
        if (hipMalloc(&p->d_twiddle64, tw.size() * rs) != hipSuccess) { pvx_set_error("hipMalloc(stft twiddle) failed"); plan_free(p); return PVX_ERR_ALLOC; }
        if (hipMemcpy(p->d_twiddle64, src, tw.size() * rs, hipMemcpyHostToDevice) != hipSuccess) { pvx_set_error("hipMemcpy(stft twiddle) failed"); plan_free(p); return PVX_ERR_HIP; }
        p->use_stft = true;
        p->use_stft_pv = pvx_stft_pv_supported(nfft, precision, npks) != 0 && !getenv("PVX_NO_STFT_PV");
        p->use_pv_rev = p->use_stft_pv && pvx_pv_rev_supported(nfft, precision, npks) != 0 && !getenv("PVX_NO_PV_REV");
        // (k_pv_team is a witness kernel -- tests/libpvx_witness.so, asked for with PVX_PV_TEAM=1: correct, one launch, and slower than
        // the two kernels it would replace, profiles/r06_ab_steps.txt; in the product library pvx_pv_team_supported() says no)
        p->use_pv_team = precision == 64 && (nfft == 4096 || nfft == 8192) && npks <= 64 && getenv("PVX_PV_TEAM") != nullptr;
        if (!getenv("PVX_MAX_ROWS")) {
            int64_t big = (int64_t)(((size_t)1 << 30) / ((size_t)p->ldo * 2 * rs));
            if (big > 262144) big = 262144;
            const int64_t want = (rows_hint > 0 && rows_hint < big) ? rows_hint : big;
            p->max_rows = want < 2 ? 2 : want;
        }
    }

    // fused kernel tables
    const bool can1 = pvx_fused_supported(nfft, precision, npks) != 0, can2 = pvx_fused_mw_supported(nfft, precision, npks) != 0;
    const bool can3 = pvx_fused_ring_supported(nfft, precision, npks) != 0, can4 = pvx_fused_rev_supported(nfft, precision, npks) != 0;
    const bool can5 = pvx_fused_team_supported(nfft, precision, npks) != 0;
    if (can1 || can2 || can3 || can4 || can5) {
        const size_t ntt = can5 ? (size_t)pvx_fused_team_table_len(nfft) : 0;       // k_fused_team's lane-ordered twiddles follow the table
        std::vector<float> tw(2 * ((size_t)nfft + ntt));
        const double pi = 3.141592653589793238462643383279502884;
        for (int j = 0; j < nfft; j++) { tw[2 * j] = (float)cos(2.0 * pi * j / (double)nfft); tw[2 * j + 1] = (float)(-sin(2.0 * pi * j / (double)nfft)); }
        if (ntt) pvx_fused_team_table(nfft, tw.data(), tw.data() + 2 * (size_t)nfft);
        if (hipMalloc(&p->d_twiddle, tw.size() * 4) != hipSuccess || hipMalloc((void**)&p->d_specrow, (size_t)nfft * 4) != hipSuccess) { pvx_set_error("hipMalloc(fused tables) failed"); plan_free(p); return PVX_ERR_ALLOC; }
        if (hipMemcpy(p->d_twiddle, tw.data(), tw.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { pvx_set_error("hipMemcpy(twiddle) failed"); plan_free(p); return PVX_ERR_HIP; }
        // default: one wave per frame where it exists (nfft <= 2048: independent waves walking their rows downwards
        // over one buffer each, k_fused_rev.hip, while npks leaves them enough LDS), several waves per frame above
        // (nfft 4096 / 8192: teams of such waves, k_fused_team.hip, while npks <= 128; the general path beyond -- and wherever
        // npks asks for more staging than these kernels' LDS holds.  Mode 2, k_fused_mw.hip, is a witness: only when asked for)
        p->fft_mode = can4 ? 4 : can3 ? 3 : can1 ? 1 : can5 ? 5 : 0;
        if (const char* e = getenv("PVX_FFT_MODE")) {
            const int m = atoi(e);
            if (m == 0 || (m == 1 && can1) || (m == 2 && can2) || (m == 3 && can3) || (m == 4 && can4) || (m == 5 && can5)) p->fft_mode = m;
        }
        if (const char* e = getenv("PVX_FUSED_BLOCKS")) { const long long v = atoll(e); if (v > 0) p->fused_blocks = v; }
    }
    *out = p;
    return PVX_OK;
}

// frames + spectrum workspace and the rocFFT plan (fft mode 0, calc_fft_frame): created on first use
// spectrum workspace of the k_stft path: [max_rows+1][ldo] complex
static int ensure_spec_ws(pvx_plan* p) {
    if (p->d_spec) return PVX_OK;
    const size_t sbytes = (size_t)(p->max_rows + 1) * p->ldo * 2 * real_size(p->precision);
    if (hipMalloc(&p->d_spec, sbytes) != hipSuccess) { pvx_set_error("hipMalloc of %.1f MiB spectrum workspace failed", sbytes / 1048576.0); p->d_spec = nullptr; return PVX_ERR_ALLOC; }
    p->ws_bytes += (int64_t)sbytes;
    return PVX_OK;
}
// nfft 8192 on the general path (k_stft_split -> k_phase_peaks): the transform leaves every row's candidate peaks, the peak
// kernel does not stream the 64 KB rows again (float64, config 2's signal: 0.79 -> 0.63 ms; PVX_NO_CAND=1: the two kernels
// as before).  At nfft 4096 (teams of two waves at 256 registers: the scan's reads cannot all be in flight) the scan costs
// the transform kernel what it saves the peak kernel: off unless PVX_CAND_4096=1.
static bool plan_wants_cand(const pvx_plan* p) {
    if (!p->use_stft || p->use_stft_pv || getenv("PVX_NO_CAND") != nullptr) return false;
    return p->nfft == 8192 || (p->nfft == 4096 && getenv("PVX_CAND_4096") != nullptr);
}
static int ensure_cand_ws(pvx_plan* p) {
    if (p->d_cand_bin) return PVX_OK;
    const int cap = p->N2 / 2 + 8;
    const size_t rows = (size_t)p->max_rows + 1, rs = real_size(p->precision);
    if (hipMalloc(&p->d_cand_y, rows * cap * rs) != hipSuccess || hipMalloc((void**)&p->d_cand_bin, rows * cap * 2) != hipSuccess ||
        hipMalloc((void**)&p->d_cand_stats, rows * 4 * sizeof(double)) != hipSuccess) {
        pvx_set_error("hipMalloc of the candidate workspace failed");
        if (p->d_cand_y) (void)hipFree(p->d_cand_y);
        if (p->d_cand_bin) (void)hipFree(p->d_cand_bin);
        if (p->d_cand_stats) (void)hipFree(p->d_cand_stats);
        p->d_cand_y = nullptr; p->d_cand_bin = nullptr; p->d_cand_stats = nullptr;
        return PVX_ERR_ALLOC;
    }
    p->cand_cap = cap;
    p->ws_bytes += (int64_t)(rows * cap * (rs + 2) + rows * 32);
    return PVX_OK;
}

// frames + spectrum workspace and the rocFFT plan (fft mode 0, calc_fft_frame): created on first use.
// With k_stft in charge of the analysis the rocFFT side only serves pvx_stft_frames: two rows, its own output.
static void release_rocfft(pvx_plan* p) {
    if (p->info) { rocfft_execution_info_destroy(p->info); p->info = nullptr; }
    if (p->fft) { rocfft_plan_destroy(p->fft); p->fft = nullptr; }
    if (p->d_frames) { (void)hipFree(p->d_frames); p->d_frames = nullptr; }
    if (p->d_rspec) { (void)hipFree(p->d_rspec); p->d_rspec = nullptr; }
    if (p->d_work) { (void)hipFree(p->d_work); p->d_work = nullptr; }
    if (!p->rocfft_small && !p->use_stft && p->d_spec) { (void)hipFree(p->d_spec); p->d_spec = nullptr; }
    p->rocfft_ready = false; p->rocfft_small = false; p->work_bytes = 0;
}

// full = the analysis itself goes through rocFFT (fft mode 0 without k_stft, PVHarmonic at float32): workspace of
// max_rows + 1 rows; otherwise only pvx_stft_frames uses it, one frame at a time: 3 rows, output of its own
static int ensure_rocfft(pvx_plan* p, bool full) {
    if (p->rocfft_ready && (!full || !p->rocfft_small)) return PVX_OK;
    if (p->rocfft_ready) release_rocfft(p);
    const int nfft = p->nfft, precision = p->precision;
    const size_t rs = real_size(precision);
    const bool small = !full || p->use_stft;
    const int64_t max_rows = small ? 2 : p->max_rows;
    const int64_t ws_rows = max_rows + 1;
    p->rocfft_rows = max_rows;
    const size_t fbytes = (size_t)ws_rows * p->ldi * rs, sbytes = (size_t)ws_rows * p->ldo * 2 * rs;
    void** specp = small ? &p->d_rspec : &p->d_spec;
    p->rocfft_small = small;
    if (hipMalloc(&p->d_frames, fbytes) != hipSuccess || hipMalloc(specp, sbytes) != hipSuccess) {
        pvx_set_error("hipMalloc of %.1f MiB analysis workspace failed", (fbytes + sbytes) / 1048576.0);
        if (p->d_frames) { (void)hipFree(p->d_frames); p->d_frames = nullptr; }
        if (*specp) { (void)hipFree(*specp); *specp = nullptr; }
        return PVX_ERR_ALLOC;
    }
    // rocFFT: batched 1-D real -> hermitian, one transform per workspace row (PV.py:157)
    rocfft_plan_description desc = nullptr;
    rocfft_status st = rocfft_plan_description_create(&desc);
    if (st == rocfft_status_success) {
        size_t istride = 1, ostride = 1;
        st = rocfft_plan_description_set_data_layout(desc, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved,
                                                     nullptr, nullptr, 1, &istride, (size_t)p->ldi, 1, &ostride, (size_t)p->ldo);
    }
    if (st == rocfft_status_success) {
        size_t len = (size_t)nfft;
        st = rocfft_plan_create(&p->fft, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                                precision == 32 ? rocfft_precision_single : rocfft_precision_double, 1, &len,
                                (size_t)ws_rows, desc);
    }
    if (desc) rocfft_plan_description_destroy(desc);
    if (st != rocfft_status_success) { pvx_set_error("rocfft_plan_create(nfft=%d, batch=%lld) failed: %d", nfft, (long long)ws_rows, (int)st); return PVX_ERR_HIP; }
    st = rocfft_plan_get_work_buffer_size(p->fft, &p->work_bytes);
    if (st == rocfft_status_success) st = rocfft_execution_info_create(&p->info);
    if (st != rocfft_status_success) { pvx_set_error("rocfft work buffer query failed: %d", (int)st); return PVX_ERR_HIP; }
    if (p->work_bytes) {
        if (hipMalloc(&p->d_work, p->work_bytes) != hipSuccess) { pvx_set_error("hipMalloc(rocfft work, %zu) failed", p->work_bytes); return PVX_ERR_ALLOC; }
        st = rocfft_execution_info_set_work_buffer(p->info, p->d_work, p->work_bytes);
        if (st != rocfft_status_success) { pvx_set_error("rocfft set_work_buffer failed: %d", (int)st); return PVX_ERR_HIP; }
    }
    p->ws_bytes += (int64_t)(fbytes + sbytes + p->work_bytes);
    p->rocfft_ready = true;
    return PVX_OK;
}

extern "C" int pvx_plan_set_progress(pvx_plan* plan, pvx_progress_fn fn, void* user) {
    if (!plan) { pvx_set_error("null plan"); return PVX_ERR_INVALID; }
    plan->progress_fn = fn;
    plan->progress_user = user;
    return PVX_OK;
}

// after a launch chunk: rows [0, rows_done) of total_rows are in flight; report once they are done
static int plan_progress(pvx_plan* p, hipStream_t s, int64_t rows_done, int64_t total_rows, int64_t nsig) {
    if (!p->progress_live || !p->progress_fn) return PVX_OK;
    PVX_HIP_CHECK(hipStreamSynchronize(s));
    // rows include one zero row per signal; report frames
    const int64_t per = total_rows / nsig;                            // F + 1
    const int64_t full = rows_done / per, rem = rows_done - full * per;
    const int64_t frames = full * (per - 1) + (rem > 0 ? rem - 1 : 0);
    if (frames < nsig * (per - 1)) p->progress_fn(frames, nsig * (per - 1), p->progress_user);   // the last report comes from the entry point
    return PVX_OK;
}

extern "C" int pvx_plan_set_timing(pvx_plan* plan, int enable) {
    if (!plan) { pvx_set_error("null plan"); return PVX_ERR_INVALID; }
    plan->timing = enable != 0;
    plan->ev_used = 0;
    plan->ev_spans.clear();
    return PVX_OK;
}

// stage >= 0: the event starts a span of that stage (which ends at the next recorded event);
// stage < 0: the event only ends the previous span
static int plan_event(pvx_plan* p, hipStream_t s, int stage) {
    if (!p->timing) return PVX_OK;
    if (p->ev_used == p->ev_pool.size()) {
        hipEvent_t e;
        PVX_HIP_CHECK(hipEventCreate(&e));
        p->ev_pool.push_back(e);
    }
    if (stage >= 0) p->ev_spans.push_back(std::make_pair(stage, p->ev_used));
    PVX_HIP_CHECK(hipEventRecord(p->ev_pool[p->ev_used++], s));
    return PVX_OK;
}

extern "C" int pvx_plan_get_timing(pvx_plan* plan, double* ms, int64_t* launches) {
    if (!plan || !ms || !launches) { pvx_set_error("null argument"); return PVX_ERR_INVALID; }
    for (int i = 0; i < 4; i++) { ms[i] = 0.0; launches[i] = 0; }
    if (plan->ev_used) PVX_HIP_CHECK(hipEventSynchronize(plan->ev_pool[plan->ev_used - 1]));
    for (const auto& sp : plan->ev_spans) {
        if (sp.second + 1 >= plan->ev_used || sp.first > 3) continue;
        float t = 0.f;
        PVX_HIP_CHECK(hipEventElapsedTime(&t, plan->ev_pool[sp.second], plan->ev_pool[sp.second + 1]));
        ms[sp.first] += (double)t;
        launches[sp.first] += 1;
    }
    plan->ev_used = 0;
    plan->ev_spans.clear();
    return PVX_OK;
}

// ---- run_pv ---------------------------------------------------------------------------------
template <typename T> static int grow_dev(T** p, size_t* cap, size_t need);
static int analyze_rows(pvx_plan* p, const void* d_x, int x_dtype, int64_t nsamp, int64_t nsig, int64_t sig_stride,
                        int64_t F, double* d_f, double* d_mag, double* d_ph, double* d_realph, double* d_binno,
                        double* d_t, double* d_totalmag, const double* d_prev0, hipStream_t s,
                        int64_t spec_row = -1, bool wire_out = false) {
    const int64_t total_rows = nsig * (F + 1);
    int rc;
    // wire_out: d_f / d_mag / d_ph / d_binno / d_totalmag are the sections of a wire block (pvx_analyze_dev_wire): only the kernels
    // that write it themselves take such a call
    if (wire_out && !(p->fft_mode == 4 && p->precision == 32)) return PVX_ERR_UNSUPPORTED;
    if (p->fft_mode >= 1 && p->fft_mode <= 5) {
        // one launch: window + FFT + peaks, no intermediate arrays (k_fused_rev.hip / k_fused_team.hip; the witnesses' modes 1 - 3)
        if (x_dtype == PVX_F64) {
            // float64 samples already in HBM (the host entry points narrow while they stage): the fused kernels' first step
            // is (float) x[n] -- done here in one pass, so that they exist for float32 and int16 samples only (a float64
            // sample pair per lane and row cost them registers they do not have)
            const size_t nel = (size_t)((nsig - 1) * sig_stride + nsamp);
            if ((rc = grow_dev(&p->d_x32, &p->x32_cap, nel * 4)) != PVX_OK) return rc;
            if ((rc = pvx_launch_narrow((const double*)d_x, p->d_x32, (int64_t)nel, s)) != PVX_OK) return rc;
            d_x = p->d_x32; x_dtype = PVX_F32;
        }
        FusedParams fp;
        fp.x = d_x; fp.sig_stride = sig_stride; fp.F = F; fp.total_rows = total_rows;
        fp.hop = p->hop; fp.K = p->npks; fp.rad = 5;                                     // PV.py:177
        fp.thr = p->pkthresh; fp.sr = p->sr; fp.fstep = p->fstep; fp.dt = p->dt;
        fp.wfbin = p->d_wfbin; fp.prev0 = d_prev0;
        fp.f = d_f; fp.mag = d_mag; fp.ph = d_ph; fp.realph = d_realph; fp.binno = d_binno;
        fp.t = d_t; fp.totalmag = d_totalmag; fp.win = p->d_win; fp.twiddle = p->d_twiddle;
        fp.spec_out = spec_row >= 0 ? (p->spec_host ? p->spec_host : p->d_specrow) : nullptr; fp.spec_row = spec_row;
        fp.blocks_override = p->fused_blocks;
        fp.stash = nullptr; fp.stash_bytes = 0;
        fp.wire = wire_out ? p->wire_fmt : 0;
        if (p->fft_mode == 4) {
            // k_fused_rev: the block where its waves hand a spectrum to the wave below them (sized once, for a full grid)
            if (!p->d_stash) {
                FusedParams q = fp;
                q.total_rows = (int64_t)1 << 30;
                const size_t need = pvx_fused_rev_stash_bytes(q, p->nfft);
                if (need > 0 && hipMalloc(&p->d_stash, need) == hipSuccess) p->stash_cap = need;
                else { p->d_stash = nullptr; p->stash_cap = 0; (void)hipGetLastError(); }     // (without it the waves compute that row themselves)
            }
            fp.stash = p->d_stash; fp.stash_bytes = p->stash_cap;
        }
        if ((rc = plan_event(p, s, 3)) != PVX_OK) return rc;
        rc = (p->fft_mode == 1) ? pvx_launch_fused(fp, p->nfft, x_dtype, s)
           : (p->fft_mode == 2) ? pvx_launch_fused_mw(fp, p->nfft, x_dtype, s)
           : (p->fft_mode == 3) ? pvx_launch_fused_ring(fp, p->nfft, x_dtype, s)
           : (p->fft_mode == 5) ? pvx_launch_fused_team(fp, p->nfft, x_dtype, s) : pvx_launch_fused_rev(fp, p->nfft, x_dtype, s);
        // (what the plan could not know when it chose the team kernel -- a salience radius beyond its fetch, a row count
        // beyond 32 bits -- goes to the general path below instead of failing the call)
        p->last_analysis = p->fft_mode == 1 ? "k_fused" : p->fft_mode == 2 ? "k_fused_mw" : p->fft_mode == 3 ? "k_fused_ring" : p->fft_mode == 5 ? "k_fused_team" : "k_fused_rev";
        if (rc == PVX_ERR_UNSUPPORTED && p->fft_mode == 5 && !wire_out) {
            p->fft_mode = 0;                                       // (the plan stays there: what carries a spectrum between calls asks the mode)
        } else {
            if (rc != PVX_OK) return rc;
            return plan_event(p, s, -1);
        }
    }
    p->last_from_rev = false;
    const bool take_rev = p->use_stft && p->use_pv_rev && pvx_pv_rev_takes(p->nfft, x_dtype, p->hop) && total_rows < 0x7fffff00LL;
    const bool take_team = p->use_stft && p->use_pv_team && pvx_pv_team_supported(p->nfft, p->precision, p->npks, p->hop) != 0 && total_rows < 0x7fffff00LL;
    if (take_rev || take_team) {
        // float64, npks <= 64: ONE launch over all rows, no spectrum workspace -- nfft 512 .. 2048: a wave per frame (k_pv_rev.hip);
        // nfft 4096 / 8192 at the sliding-window hops: a team of waves per frame (k_pv_team.hip)
        PvRevParams rp;
        rp.x = d_x; rp.sig_stride = sig_stride; rp.F = F; rp.total_rows = total_rows;
        rp.hop = p->hop; rp.K = p->npks; rp.rad = 5;                                     // PV.py:177
        rp.thr = p->pkthresh; rp.sr = p->sr; rp.fstep = p->fstep; rp.dt = p->dt;
        rp.wfbin = p->d_wfbin; rp.prev0 = d_prev0;
        rp.f = d_f; rp.mag = d_mag; rp.ph = d_ph; rp.realph = d_realph; rp.binno = d_binno;
        rp.t = d_t; rp.totalmag = d_totalmag; rp.win = p->d_win; rp.twiddle = p->d_twiddle64;
        rp.spec_out = nullptr; rp.spec_row = -1;
        if (spec_row >= 0) {
            if (!p->d_lastspec) PVX_HIP_CHECK(hipMalloc((void**)&p->d_lastspec, sizeof(double) * 2 * (size_t)(p->N2 > 0 ? p->N2 : 1)));
            rp.spec_out = p->d_lastspec; rp.spec_row = spec_row;
        }
        rp.blocks_override = 0;
        if (const char* e = getenv("PVX_PV_REV_BLOCKS")) { const long long v = atoll(e); if (v >= 1) rp.blocks_override = v; }   // tests: other grids
        rp.win_symmetric = p->win_symmetric ? 1 : 0;
        rp.stage = nullptr; rp.stage_bytes = 0;
        const size_t need = take_team ? pvx_pv_team_stage_bytes(p->nfft) : pvx_pv_rev_stage_bytes(p->nfft);
        if (need > 0) {
            if ((rc = grow_dev(&p->d_pvstage, &p->pvstage_cap, need)) != PVX_OK) return rc;
            rp.stage = p->d_pvstage; rp.stage_bytes = p->pvstage_cap;
        }
        // one launch -- unless the caller set PVX_MAX_ROWS: then pieces of that many rows, each reported (tests, progress displays)
        const int64_t piece = p->rows_from_env ? p->max_rows : total_rows;
        for (int64_t R0 = 0; R0 < total_rows; R0 += piece) {
            rp.row_begin = R0; rp.row_end = (total_rows - R0 < piece) ? total_rows : R0 + piece;
            if ((rc = plan_event(p, s, 3)) != PVX_OK) return rc;
            if ((rc = take_team ? pvx_launch_pv_team(rp, p->nfft, x_dtype, s) : pvx_launch_pv_rev(rp, p->nfft, x_dtype, s)) != PVX_OK) return rc;
            if ((rc = plan_event(p, s, -1)) != PVX_OK) return rc;
            if ((rc = plan_progress(p, s, rp.row_end, total_rows, nsig)) != PVX_OK) return rc;
        }
        p->last_from_rev = spec_row >= 0;
        p->last_analysis = take_team ? "k_pv_team" : "k_pv_rev";
        return PVX_OK;
    }
    if (p->use_stft) { if ((rc = ensure_spec_ws(p)) != PVX_OK) return rc; }
    else {
        if ((rc = ensure_rocfft(p, true)) != PVX_OK) return rc;
        PVX_FFT_CHECK(rocfft_execution_info_set_stream(p->info, s));
    }
    const bool cand = plan_wants_cand(p);
    if (cand && (rc = ensure_cand_ws(p)) != PVX_OK) return rc;
    for (int64_t R0 = 0; R0 < total_rows; R0 += p->max_rows) {
        const int64_t nrows = (total_rows - R0 < p->max_rows) ? (total_rows - R0) : p->max_rows;
        FrameParams fp;
        fp.x = d_x; fp.nsamp = nsamp; fp.sig_stride = sig_stride; fp.F = F; fp.R0 = R0;
        fp.ws_rows = nrows + 1; fp.total_rows = total_rows; fp.nfft = p->nfft; fp.hop = p->hop;
        fp.win = p->d_win; fp.frames = p->d_frames; fp.ldi = p->ldi; fp.win_symmetric = p->win_symmetric ? 1 : 0;
        PeaksParams pp;
        pp.spec = p->d_spec; pp.ldo = p->ldo; pp.F = F; pp.R0 = R0; pp.nrows = nrows;
        pp.nfft = p->nfft; pp.hop = p->hop; pp.N2 = p->N2; pp.K = p->npks; pp.rad = 5;   // PV.py:177
        pp.thr = p->pkthresh; pp.sr = p->sr; pp.fstep = p->fstep; pp.dt = p->dt;
        pp.wfbin = p->d_wfbin; pp.prev0 = d_prev0;
        pp.f = d_f; pp.mag = d_mag; pp.ph = d_ph; pp.realph = d_realph; pp.binno = d_binno;
        pp.t = d_t; pp.totalmag = d_totalmag; pp.frames_per_wave = p->frames_per_wave;
        if (cand) {
            fp.cand_y = p->d_cand_y; fp.cand_bin = p->d_cand_bin; fp.cand_stats = p->d_cand_stats; fp.cand_cap = p->cand_cap; fp.cand_thr = p->pkthresh;
            pp.cand_y = p->d_cand_y; pp.cand_bin = p->d_cand_bin; pp.cand_stats = p->d_cand_stats; pp.cand_cap = p->cand_cap;
        }
        if (p->use_stft && p->use_stft_pv && pvx_stft_pv_takes(p->nfft, p->precision, x_dtype, p->hop)) {
            // window + FFT + untangle + peaks of every row in one kernel (k_stft_pv.hip); the spectrum rows still land in
            // the workspace
            if ((rc = plan_event(p, s, 3)) != PVX_OK) return rc;
            if ((rc = pvx_launch_stft_pv(fp, pp, p->d_spec, p->ldo, p->d_twiddle64, x_dtype, p->precision, s)) != PVX_OK) return rc;
            if ((rc = plan_event(p, s, -1)) != PVX_OK) return rc;
            if ((rc = plan_progress(p, s, R0 + nrows, total_rows, nsig)) != PVX_OK) return rc;
            p->last_analysis = "k_stft_pv";
            continue;
        }
        if ((rc = plan_event(p, s, 0)) != PVX_OK) return rc;
        if (p->use_stft) {
            // window + FFT + untangle of every workspace row in one kernel (k_stft.hip)
            if ((rc = pvx_launch_stft(fp, p->d_spec, p->ldo, p->d_twiddle64, x_dtype, p->precision, s)) != PVX_OK) return rc;
        } else {
            if ((rc = pvx_launch_frames(fp, x_dtype, p->precision, s)) != PVX_OK) return rc;
            if ((rc = plan_event(p, s, 1)) != PVX_OK) return rc;
            void* in[1] = {p->d_frames};
            void* out[1] = {p->d_spec};
            PVX_FFT_CHECK(rocfft_execute(p->fft, in, out, p->info));
        }
        if ((rc = plan_event(p, s, 2)) != PVX_OK) return rc;
        if ((rc = pvx_launch_phase_peaks(pp, p->precision, s)) != PVX_OK) return rc;
        p->last_analysis = p->use_stft ? "k_stft+k_phase_peaks" : "k_frames+rocfft+k_phase_peaks";
        if ((rc = plan_event(p, s, -1)) != PVX_OK) return rc;
        if ((rc = plan_progress(p, s, R0 + nrows, total_rows, nsig)) != PVX_OK) return rc;
    }
    return PVX_OK;
}

static int check_analyze_args(pvx_plan* p, const void* x, int x_dtype, int64_t nsamp, int64_t nsig, int64_t sig_stride,
                              const void* prev0) {
    if (!p) { pvx_set_error("null plan"); return PVX_ERR_INVALID; }
    if (!x && nsamp > 0) { pvx_set_error("null signal"); return PVX_ERR_INVALID; }
    if (x_dtype != PVX_F32 && x_dtype != PVX_F64 && x_dtype != PVX_I16) { pvx_set_error("bad x_dtype %d", x_dtype); return PVX_ERR_INVALID; }
    if (nsamp < 0 || nsig < 1) { pvx_set_error("bad nsamp/nsig"); return PVX_ERR_INVALID; }
    if (nsig > 1 && sig_stride < nsamp) { pvx_set_error("sig_stride %lld < nsamp %lld", (long long)sig_stride, (long long)nsamp); return PVX_ERR_INVALID; }
    if (prev0 && nsig != 1) { pvx_set_error("prev0 is only valid with nsig == 1"); return PVX_ERR_INVALID; }
    return PVX_OK;
}

extern "C" int64_t pvx_analyze_dev(pvx_plan* p, const void* d_x, int x_dtype, int64_t nsamp, int64_t nsig,
                                   int64_t sig_stride, double* d_f, double* d_mag, double* d_ph, double* d_realph,
                                   double* d_binno, double* d_t, double* d_totalmag, const double* d_prev0,
                                   void* stream) {
    int rc = pvx_require_plan_device(p);
    if (rc != PVX_OK) return rc;
    rc = check_analyze_args(p, d_x, x_dtype, nsamp, nsig, sig_stride, d_prev0);
    if (rc != PVX_OK) return rc;
    const int64_t F = pvx_nframes(nsamp, p->nfft, p->hop);
    if (F == 0) return 0;
    if (!d_f || !d_mag || !d_ph || !d_realph || !d_binno) { pvx_set_error("null output array"); return PVX_ERR_INVALID; }
    rc = analyze_rows(p, d_x, x_dtype, nsamp, nsig, sig_stride, F, d_f, d_mag, d_ph, d_realph, d_binno, d_t,
                      d_totalmag, d_prev0, (hipStream_t)stream);
    return rc == PVX_OK ? F : rc;
}

static size_t dtype_size(int x_dtype) { return x_dtype == PVX_F32 ? 4 : (x_dtype == PVX_F64 ? 8 : 2); }

namespace {
struct DevBuf {   // RAII for the host-buffer wrappers
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { pvx_set_error("hipMalloc(%zu) failed", bytes); p = nullptr; return PVX_ERR_ALLOC; }
        return PVX_OK;
    }
};
}  // namespace

// grow-only device / pinned buffers owned by the plan
template <typename T> static int grow_dev(T** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return PVX_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    const size_t want = need + need / 4 + 256;
    if (hipMalloc((void**)p, want) != hipSuccess) {
        if (hipMalloc((void**)p, need ? need : 1) != hipSuccess) { pvx_set_error("hipMalloc(%zu) failed", need); *p = nullptr; return PVX_ERR_ALLOC; }
        *cap = need;
        return PVX_OK;
    }
    *cap = want;
    return PVX_OK;
}
static int grow_pin(pvx_plan* p, size_t need) {
    if (need <= p->pin_cap && p->h_pin) return PVX_OK;
    if (p->h_pin) { (void)hipHostFree(p->h_pin); p->h_pin = nullptr; p->pin_cap = 0; }
    if (hipHostMalloc(&p->h_pin, need, hipHostMallocDefault) != hipSuccess) { pvx_set_error("hipHostMalloc(%zu) failed", need); return PVX_ERR_ALLOC; }
    p->pin_cap = need;
    return PVX_OK;
}
static int host_stream(pvx_plan* p) {
    if (!p->s_host) PVX_HIP_CHECK(hipStreamCreateWithFlags(&p->s_host, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++)
        if (!p->ev_done[i]) PVX_HIP_CHECK(hipEventCreateWithFlags(&p->ev_done[i], hipEventDisableTiming));
    return PVX_OK;
}

// Large host <-> device transfers of PAGEABLE caller memory.  A plain hipMemcpy stages through the runtime's own pinned
// buffers on one thread (~20 GB/s host-side; measured on BASELINE config 2's 106 MB signal: 4.9 ms -> 2.7 ms with four threads on 2 MB pieces, 2.3 ms on 4 MB pieces; eight threads on 1 MB pieces: 3.4 ms; `tools/ab/stage_ab.sh`); here kStageThreads threads copy 4 MB pieces between the caller's array and a
// pinned ring (two slots per thread) while the DMA engine moves the other slots: the host copies run in parallel with each
// other and with the DMA, the link (PCIe Gen5 x16, ~55 GB/s) becomes the limit.  The DMAs are queued on `s`: work issued on
// `s` afterwards is ordered behind them; the calls return when every host-side copy is done (H2D) / every byte has
// arrived in the caller's array (D2H).
static const int kStageMaxThreads = 8;
static const int kStageThreads = [] { const char* e = getenv("PVX_STAGE_THREADS"); const int v = e ? atoi(e) : 0; return (v >= 1 && v <= kStageMaxThreads) ? v : 4; }();
static const size_t kStagePiece = [] { const char* e = getenv("PVX_STAGE_PIECE_MB"); const int v = e ? atoi(e) : 0; return (size_t)((v >= 1 && v <= 16) ? v : 4) << 20; }();
static const size_t kStageMin = (size_t)16 << 20;           // below this a plain copy is as good

struct StageRing { void** h_ring; hipEvent_t* ev_ring; };     // a plan's ring, or the process-wide one of the plan-less entry points
static int stage_ring(StageRing r) {
    if (!*r.h_ring && hipHostMalloc(r.h_ring, kStagePiece * 2 * kStageThreads, hipHostMallocDefault) != hipSuccess) {
        *r.h_ring = nullptr; pvx_set_error("hipHostMalloc(staging ring) failed"); return PVX_ERR_ALLOC;
    }
    for (int i = 0; i < 16; i++)
        if (!r.ev_ring[i]) PVX_HIP_CHECK(hipEventCreateWithFlags(&r.ev_ring[i], hipEventDisableTiming));
    return PVX_OK;
}
static int stage_ring(pvx_plan* p) { return stage_ring(StageRing{&p->h_ring, p->ev_ring}); }

// narrow: `host` holds float64 samples and the device gets them as float32 (`bytes` counts the float32 bytes): every
// precision-32 kernel's first step is (float) x[n], so the staging threads do it while they copy -- the same rounding, half
// the bytes over the link.
static int staged_copy(StageRing p_, void* dev, void* host, size_t bytes, bool to_device, hipStream_t s, bool narrow = false) {
    int rc = stage_ring(p_);
    if (rc != PVX_OK) return rc;
    struct { void* h_ring; hipEvent_t* ev_ring; } pp = {*p_.h_ring, p_.ev_ring}, *p = &pp;
    int devid = 0;
    (void)hipGetDevice(&devid);
    const size_t npieces = (bytes + kStagePiece - 1) / kStagePiece;
    int err[kStageMaxThreads] = {0};
    auto worker = [&](int t, int nthreads) {
        if (hipSetDevice(devid) != hipSuccess) { err[t] = 1; return; }
        size_t k = 0;
        for (size_t i = (size_t)t; i < npieces; i += nthreads, k++) {
            const int slot = 2 * t + (int)(k & 1);
            char* pin = (char*)p->h_ring + (size_t)slot * kStagePiece;
            const size_t o = i * kStagePiece, c = bytes - o < kStagePiece ? bytes - o : kStagePiece;
            if (to_device) {
                // the slot's previous DMA has read it -- of this call or of an earlier one on the same ring (nothing orders
                // the host against those but this event; on an event never recorded the wait returns at once)
                if (hipEventSynchronize(p->ev_ring[slot]) != hipSuccess) { err[t] = 1; return; }
                if (narrow) narrow_f64_f32((const double*)host + o / 4, (float*)pin, c / 4);
                else memcpy(pin, (const char*)host + o, c);
                if (hipMemcpyAsync((char*)dev + o, pin, c, hipMemcpyHostToDevice, s) != hipSuccess || hipEventRecord(p->ev_ring[slot], s) != hipSuccess) { err[t] = 1; return; }
            } else {
                // two DMAs of this thread in flight: piece k+1 lands while piece k is copied out
                if (k == 0) {
                    if (hipMemcpyAsync(pin, (const char*)dev + o, c, hipMemcpyDeviceToHost, s) != hipSuccess || hipEventRecord(p->ev_ring[slot], s) != hipSuccess) { err[t] = 1; return; }
                }
                const size_t in = i + nthreads;
                if (in < npieces) {
                    const int ns = 2 * t + (int)((k + 1) & 1);
                    const size_t no = in * kStagePiece, nc = bytes - no < kStagePiece ? bytes - no : kStagePiece;
                    if (hipMemcpyAsync((char*)p->h_ring + (size_t)ns * kStagePiece, (const char*)dev + no, nc, hipMemcpyDeviceToHost, s) != hipSuccess ||
                        hipEventRecord(p->ev_ring[ns], s) != hipSuccess) { err[t] = 1; return; }
                }
                if (hipEventSynchronize(p->ev_ring[slot]) != hipSuccess) { err[t] = 1; return; }
                memcpy((char*)host + o, pin, c);
            }
        }
    };
    // (a thread that cannot be created must not take the process down through the C ABI: the copy then runs on this thread)
    const bool one_thread = getenv("PVX_NO_STAGE_THREADS") != nullptr;
    int nthreads = one_thread ? 1 : kStageThreads;
    std::thread th[kStageMaxThreads];
    int started = 1;
    try {
        for (int t = 1; t < nthreads; t++) { th[t] = std::thread(worker, t, nthreads); started = t + 1; }
    } catch (...) {
        for (int t = 1; t < started; t++) th[t].join();
        if (started > 1) { (void)hipStreamSynchronize(s); pvx_set_error("staged host transfer: could not start its threads"); return PVX_ERR_HIP; }
        nthreads = 1;
    }
    worker(0, nthreads);
    for (int t = 1; t < (started > 1 ? nthreads : 1); t++) th[t].join();
    for (int t = 0; t < nthreads; t++)
        if (err[t]) { (void)hipStreamSynchronize(s); pvx_set_error("staged host transfer failed (%s)", hipGetErrorString(hipGetLastError())); return PVX_ERR_HIP; }
    return PVX_OK;
}

static int staged_copy(pvx_plan* p, void* dev, void* host, size_t bytes, bool to_device, hipStream_t s, bool narrow = false) {
    return staged_copy(StageRing{&p->h_ring, p->ev_ring}, dev, host, bytes, to_device, s, narrow);
}
// Transfers between the CALLER's arrays and device memory.  A hipMemcpy of a pageable array of 1 MB or more makes the
// runtime register (pin in place) that memory; when the caller frees the array -- result arrays: every call -- the next
// transfer of the process waits tens of milliseconds for the unmapping (measured: PVHarmonic.run_pv 8 -> 32 ms per
// call).  So nothing of the caller's above kDirectMax is ever handed to hipMemcpy: page-locked arrays go straight, large
// pageable ones through the threaded ring, the ones in between bounce through the process-wide ring's memory.
static const size_t kDirectMax = (size_t)512 << 10;
static const int kMaxDevices = 16;
struct DevRing { void* ring = nullptr; hipEvent_t ev[16] = {}; std::mutex mu; };   // one per device: its events belong to that device
static DevRing g_rings[kMaxDevices];
static bool host_is_pinned(const void* host) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, host) != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeHost;
}
static int user_copy(void* dev, void* host, size_t bytes, bool to_device) {
    if (bytes == 0) return PVX_OK;
    const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
    if (bytes <= kDirectMax || host_is_pinned(host)) {
        PVX_HIP_CHECK(to_device ? hipMemcpy(dev, host, bytes, kind) : hipMemcpy(host, dev, bytes, kind));
        return PVX_OK;
    }
    int devid = 0;
    (void)hipGetDevice(&devid);
    DevRing& dr = g_rings[devid >= 0 && devid < kMaxDevices ? devid : 0];
    std::lock_guard<std::mutex> lk(dr.mu);
    void*& g_ring = dr.ring;
    const StageRing ring{&dr.ring, dr.ev};
    if (bytes >= kStageMin && getenv("PVX_NO_STAGE_THREADS") == nullptr) {
        const int rc = staged_copy(ring, dev, host, bytes, to_device, nullptr);
        if (rc != PVX_OK) return rc;
        PVX_HIP_CHECK(hipStreamSynchronize(nullptr));
        return PVX_OK;
    }
    int rc = stage_ring(ring);
    if (rc != PVX_OK) return rc;
    const size_t cap = kStagePiece * 2 * kStageThreads;
    for (size_t o = 0; o < bytes; o += cap) {
        const size_t c = bytes - o < cap ? bytes - o : cap;
        if (to_device) {
            memcpy(g_ring, (const char*)host + o, c);
            PVX_HIP_CHECK(hipMemcpy((char*)dev + o, g_ring, c, kind));
        } else {
            PVX_HIP_CHECK(hipMemcpy(g_ring, (const char*)dev + o, c, kind));
            memcpy((char*)host + o, g_ring, c);
        }
    }
    return PVX_OK;
}
static int host_to_device(void* dev, const void* host, size_t bytes) { return user_copy(dev, (void*)host, bytes, true); }
static int device_to_host(void* host, const void* dev, size_t bytes) { return user_copy((void*)dev, host, bytes, false); }

struct HostOut { double *f, *mag, *ph, *realph, *binno, *t, *totalmag; };

// pointers into a packed result block of `rows` frames: f | mag | ph | realph | binno | t | totalmag
static HostOut block_ptrs(double* base, int64_t rows, int K) {
    const size_t n = (size_t)rows * K;
    HostOut o;
    o.f = base; o.mag = base + n; o.ph = base + 2 * n; o.realph = base + 3 * n; o.binno = base + 4 * n;
    o.t = base + 5 * n; o.totalmag = base + 5 * n + rows;
    return o;
}

static const size_t kSmallCall = (size_t)4 << 20;     // calls up to this size go through pinned staging, one sync
// ... and analyses up to the size where the threaded ring takes over (kStageMin): the plan's own pinned block, the DMA of
// piece i under the host copy of piece i+1, one result copy, one synchronisation -- and no lock shared with other plans
// (pvx_batch_run's workers; the process-wide bounce ring in between cost a 30-s signal 0.45 ms alone and serialised them)
static const size_t kSmallAnalyze = kStageMin;

// spectrum of the last row of the launch that just ran on `s` -> p->d_prev (float64 [N2][2])
static int carry_spectrum(pvx_plan* p, int64_t rows_in_call, hipStream_t s) {
    if (!p->d_prev) PVX_HIP_CHECK(hipMalloc((void**)&p->d_prev, sizeof(double) * 2 * (size_t)(p->N2 > 0 ? p->N2 : 1)));
    if (p->fft_mode != 0) return pvx_launch_spec_to_prev(p->d_prev, p->d_specrow, 2 * p->N2, 1, s);
    if (p->last_from_rev) return pvx_launch_spec_to_prev(p->d_prev, p->d_lastspec, 2 * p->N2, 0, s);
    const int64_t lastR0 = ((rows_in_call - 1) / p->max_rows) * p->max_rows;
    const int64_t wsrow = (rows_in_call - 1) - lastR0 + 1;
    const size_t rs = real_size(p->precision);
    return pvx_launch_spec_to_prev(p->d_prev, (const char*)p->d_spec + (size_t)wsrow * p->ldo * 2 * rs, 2 * p->N2, p->precision == 32, s);
}

// The host entry point of run_pv (PV.py:213-264).
//   keep = false: results stream back into the caller's arrays (`ho`);
//   keep = true : results stay in the plan's resident block (pvx_analyze_resident), nothing comes back.
// The input is taken in chunks of frames (one signal) or of whole signals (a batch) that fit
// PVX_MAX_DEVICE_BYTES, double-buffered: while the kernels of chunk i run on the plan's stream the host
// copies chunk i+1 in and chunk i-1's results out.  Between chunks of one signal the spectrum of the last
// frame is carried on the device as the next chunk's `oldfft` (PV.py:209), so the result is the same,
// bit for bit, as one launch over the whole signal -- and a signal larger than HBM runs.
static int64_t analyze_host(pvx_plan* p, const void* x, int x_dtype, int64_t nsamp, int64_t nsig, int64_t sig_stride,
                            const HostOut* ho, const double* prev0, double* last_spec, bool keep) {
    HostTrace tr("analyze");
    int rc = pvx_require_plan_device(p);
    if (rc != PVX_OK) return rc;
    rc = check_analyze_args(p, x, x_dtype, nsamp, nsig, sig_stride, prev0);
    if (rc != PVX_OK) return rc;
    const int64_t F = pvx_nframes(nsamp, p->nfft, p->hop);
    if (keep) { p->res_valid = false; p->res_P = -1; }         // only a resident run replaces the resident block
    if (F == 0) return 0;
    if (!keep && (!ho->f || !ho->mag || !ho->ph || !ho->realph || !ho->binno)) { pvx_set_error("null output array"); return PVX_ERR_INVALID; }
    if (nsig == 1) sig_stride = nsamp;
    if ((rc = host_stream(p)) != PVX_OK) return rc;
    hipStream_t s = p->s_host;
    const int K = p->npks;
    const size_t es = dtype_size(x_dtype);
    const size_t per_frame_out = (size_t)(5 * K + 2) * sizeof(double);
    size_t limit = (size_t)2 << 30;                                   // per buffer (input chunk + its result block)
    if (const char* e = getenv("PVX_MAX_DEVICE_BYTES")) { const long long v = atoll(e); if (v > 0) limit = (size_t)v; }
    if (keep) {
        if ((rc = grow_dev(&p->d_res, &p->res_cap, (size_t)nsig * F * per_frame_out)) != PVX_OK) return rc;
    }
    {
        if (!p->d_prev) PVX_HIP_CHECK(hipMalloc((void**)&p->d_prev, sizeof(double) * 2 * (size_t)(p->N2 > 0 ? p->N2 : 1)));
    }
    if (prev0) PVX_HIP_CHECK(hipMemcpyAsync(p->d_prev, prev0, sizeof(double) * 2 * p->N2, hipMemcpyHostToDevice, s));

    // ---- chunking: units are frames of the one signal, or whole signals of a batch
    const bool by_frames = nsig == 1;
    const int64_t units = by_frames ? F : nsig;
    int64_t per_chunk;
    if (by_frames) {
        const size_t fixed = (size_t)p->nfft * es;
        const size_t per = (size_t)p->hop * es + (keep ? 0 : per_frame_out);
        per_chunk = limit > fixed + per ? (int64_t)((limit - fixed) / per) : 1;
    } else {
        const size_t per = (size_t)sig_stride * es + (keep ? 0 : (size_t)F * per_frame_out);
        per_chunk = (int64_t)(limit / per);
        if (per_chunk < 1) {
            // one signal of the batch does not fit: run the signals one by one through the frame chunking
            if (last_spec || prev0) { pvx_set_error("prev0 / last_spec need nsig == 1"); return PVX_ERR_INVALID; }
            for (int64_t b = 0; b < nsig; b++) {
                HostOut hb;
                const size_t o1 = (size_t)b * F, ok = o1 * K;
                if (!keep) { hb.f = ho->f + ok; hb.mag = ho->mag + ok; hb.ph = ho->ph + ok; hb.realph = ho->realph + ok; hb.binno = ho->binno + ok;
                             hb.t = ho->t ? ho->t + o1 : nullptr; hb.totalmag = ho->totalmag ? ho->totalmag + o1 : nullptr; }
                if (keep) { pvx_set_error("a signal of this batch exceeds PVX_MAX_DEVICE_BYTES: resident results need nsig == 1 for it"); return PVX_ERR_UNSUPPORTED; }
                const int64_t r = analyze_host(p, (const char*)x + (size_t)b * sig_stride * es, x_dtype, nsamp, 1, nsamp, &hb, nullptr, nullptr, false);
                if (r < 0) return r;
            }
            return F;
        }
    }
    if (per_chunk < 1) per_chunk = 1;
    if (per_chunk > units) per_chunk = units;
    const int64_t nchunks = (units + per_chunk - 1) / per_chunk;
    const size_t total_in = (size_t)((nsig - 1) * sig_stride + nsamp) * es;
    const size_t total_out = (size_t)nsig * F * per_frame_out;
    const bool small = nchunks == 1 && total_in + (keep ? 0 : total_out) <= kSmallAnalyze;
    // a float64 signal analysed at precision 32: every kernel's first step is (float)x[n], so the host side narrows while it
    // stages (same rounding, half the bytes over PCIe, the aligned float loads on the device) -- on the small-call path and
    // in the staging threads of the large one
    const size_t spec_pin = (size_t)(p->N2 > 0 ? p->N2 : 1) * 16 + 256;       // the last spectrum lands behind the staged data
    const bool narrow = p->precision == 32 && x_dtype == PVX_F64;
    const int dev_dtype = narrow ? PVX_F32 : x_dtype;
    const size_t des = narrow ? 4 : es;

    auto chunk_geom = [&](int64_t c, int64_t& u0, int64_t& u1, size_t& in_off, size_t& in_bytes, int64_t& c_nsamp, int64_t& c_nsig, int64_t& c_frames) {
        u0 = c * per_chunk; u1 = u0 + per_chunk < units ? u0 + per_chunk : units;
        if (by_frames) {
            in_off = (size_t)u0 * p->hop * es;
            // frames u0 .. u1-1 need samples [u0 hop, (u1-1) hop + nfft]: nframes() is strict (PV.py:224-225), one more sample
            c_nsamp = (u1 - 1 - u0) * (int64_t)p->hop + p->nfft + 1;
            if ((int64_t)u0 * p->hop + c_nsamp > nsamp) c_nsamp = nsamp - u0 * (int64_t)p->hop;
            in_bytes = (size_t)c_nsamp * es;
            c_nsig = 1; c_frames = u1 - u0;
        } else {
            in_off = (size_t)u0 * sig_stride * es;
            c_nsig = u1 - u0; c_nsamp = nsamp; c_frames = (u1 - u0) * F;
            in_bytes = (size_t)((c_nsig - 1) * sig_stride + nsamp) * es;
        }
    };
    // results of chunk c (in d_out[c & 1]) -> the caller's arrays
    auto fetch = [&](int64_t c) -> int {
        int64_t u0, u1, c_nsamp, c_nsig, c_frames; size_t in_off, in_bytes;
        chunk_geom(c, u0, u1, in_off, in_bytes, c_nsamp, c_nsig, c_frames);
        PVX_HIP_CHECK(hipEventSynchronize(p->ev_done[c & 1]));
        const HostOut d = block_ptrs(p->d_out[c & 1], c_frames, K);
        const size_t r0 = by_frames ? (size_t)u0 : (size_t)u0 * F;
        const size_t nk = (size_t)c_frames * K * sizeof(double), n1 = (size_t)c_frames * sizeof(double);
        int rcc;
        if ((rcc = device_to_host(ho->f + r0 * K, d.f, nk)) != PVX_OK || (rcc = device_to_host(ho->mag + r0 * K, d.mag, nk)) != PVX_OK ||
            (rcc = device_to_host(ho->ph + r0 * K, d.ph, nk)) != PVX_OK || (rcc = device_to_host(ho->realph + r0 * K, d.realph, nk)) != PVX_OK ||
            (rcc = device_to_host(ho->binno + r0 * K, d.binno, nk)) != PVX_OK) return rcc;
        if (ho->totalmag && (rcc = device_to_host(ho->totalmag + r0, d.totalmag, n1)) != PVX_OK) return rcc;
        return PVX_OK;
    };

    p->progress_live = true;
    int64_t last_rows = 0;
    tr.mark("setup");
    for (int64_t c = 0; c < nchunks; c++) {
        const int b = (int)(c & 1);
        int64_t u0, u1, c_nsamp, c_nsig, c_frames; size_t in_off, in_bytes;
        chunk_geom(c, u0, u1, in_off, in_bytes, c_nsamp, c_nsig, c_frames);
        if (c >= 2) {
            // chunk c-2 used these buffers: its kernels are done (and its results fetched, below)
            if ((rc = (int)hipEventSynchronize(p->ev_done[b])) != 0) { pvx_set_error("hipEventSynchronize failed"); p->progress_live = false; return PVX_ERR_HIP; }
        }
        if ((rc = grow_dev(&p->d_in[b], &p->in_cap[b], in_bytes)) != PVX_OK) { p->progress_live = false; return rc; }
        double* ob = nullptr;
        if (!keep) {
            if ((rc = grow_dev(&p->d_out[b], &p->out_cap[b], (size_t)c_frames * per_frame_out)) != PVX_OK) { p->progress_live = false; return rc; }
            ob = p->d_out[b];
        }
        if (small) {
            if ((rc = grow_pin(p, ((total_in + 255) & ~(size_t)255) + (keep ? 0 : total_out) + spec_pin)) != PVX_OK) { p->progress_live = false; return rc; }
            // staged in pieces: the DMA of piece i runs under the host copy of piece i+1
            const size_t nel = in_bytes / es, piece = (size_t)64 << 10;
            for (size_t e0 = 0; e0 < nel; e0 += piece) {
                const size_t cnt = nel - e0 < piece ? nel - e0 : piece;
                if (narrow) {
                    narrow_f64_f32((const double*)x + e0, (float*)p->h_pin + e0, cnt);
                    tr.mark("  piece narrowed");
                } else {
                    memcpy((char*)p->h_pin + e0 * es, (const char*)x + e0 * es, cnt * es);
                }
                PVX_HIP_CHECK(hipMemcpyAsync((char*)p->d_in[b] + e0 * des, (char*)p->h_pin + e0 * des, cnt * des, hipMemcpyHostToDevice, s));
            }
        } else {
            // pageable: concurrent with the kernels of chunk c-1 on the plan's stream; large chunks through the threaded ring
            const bool threaded = getenv("PVX_NO_STAGE_THREADS") == nullptr;
            if (narrow || (threaded && in_bytes >= kStageMin)) {
                if ((rc = staged_copy(p, p->d_in[b], (void*)((const char*)x + in_off), narrow ? in_bytes / 2 : in_bytes, true, s, narrow)) != PVX_OK) { p->progress_live = false; return rc; }
            } else {
                if ((rc = host_to_device(p->d_in[b], (const char*)x + in_off, in_bytes)) != PVX_OK) { p->progress_live = false; return rc; }
            }
        }
        tr.mark("staged + H2D issued");
        HostOut d;
        if (keep) {
            const HostOut all = block_ptrs(p->d_res, nsig * F, K);
            const size_t r0 = by_frames ? (size_t)u0 : (size_t)u0 * F;
            d.f = all.f + r0 * K; d.mag = all.mag + r0 * K; d.ph = all.ph + r0 * K; d.realph = all.realph + r0 * K;
            d.binno = all.binno + r0 * K; d.t = nchunks == 1 ? all.t : nullptr; d.totalmag = all.totalmag + r0;
        } else {
            d = block_ptrs(ob, c_frames, K);
            d.t = nullptr;
        }
        const bool want_spec = (c + 1 < nchunks && by_frames) || (last_spec && c + 1 == nchunks);
        // small call, fused kernels: the last spectrum is written by the kernel itself into the pinned block (one
        // 16 KB burst over PCIe instead of a separate copy operation behind the kernel)
        float* spec_host = nullptr;
        if (small && last_spec && p->fft_mode != 0) {
            spec_host = (float*)((unsigned char*)p->h_pin + ((((total_in + 255) & ~(size_t)255) + (keep ? 0 : total_out) + 255) & ~(size_t)255));
            if ((size_t)((unsigned char*)spec_host - (unsigned char*)p->h_pin) + (size_t)p->N2 * 8 > p->pin_cap) spec_host = nullptr;
        }
        p->spec_host = spec_host;
        const double* dprev = (c > 0 && by_frames) || prev0 ? p->d_prev : nullptr;
        last_rows = c_nsig * (c_frames / c_nsig + 1);
        rc = analyze_rows(p, p->d_in[b], dev_dtype, c_nsamp, c_nsig, by_frames ? c_nsamp : sig_stride, c_frames / c_nsig,
                          d.f, d.mag, d.ph, d.realph, d.binno, d.t, d.totalmag, dprev, s, want_spec ? last_rows - 1 : -1);
        p->spec_host = nullptr;
        tr.mark("kernels issued");
        if (rc == PVX_OK && c + 1 < nchunks && by_frames) rc = carry_spectrum(p, last_rows, s);
        if (rc != PVX_OK) { p->progress_live = false; return rc; }
        PVX_HIP_CHECK(hipEventRecord(p->ev_done[b], s));
        if (!keep && !small && c >= 1) {
            if ((rc = fetch(c - 1)) != PVX_OK) { p->progress_live = false; return rc; }      // under the kernels of chunk c
        }
        if (p->progress_fn && nchunks > 1 && c + 1 < nchunks) {
            const int64_t done = by_frames ? u1 : u1 * F;
            p->progress_fn(done, nsig * F, p->progress_user);
        }
    }
    p->progress_live = false;
    // ---- tail: last chunk's results, frame times, the last spectrum
    if (keep) {
        const HostOut all = block_ptrs(p->d_res, nsig * F, K);
        if (nchunks > 1 && (rc = pvx_launch_fill_t(all.t, F, nsig, p->hop, p->nfft, p->sr, s)) != PVX_OK) return rc;   // else the kernels wrote it
        p->res_F = F; p->res_nsig = nsig; p->res_valid = true;
    } else if (small) {
        // one D2H of the whole block into pinned memory, one synchronisation, then plain memcpys
        double* hb = (double*)((char*)p->h_pin + ((total_in + 255) & ~(size_t)255));
        PVX_HIP_CHECK(hipMemcpyAsync(hb, p->d_out[0], total_out, hipMemcpyDeviceToHost, s));
        PVX_HIP_CHECK(hipStreamSynchronize(s));
        const HostOut h = block_ptrs(hb, nsig * F, K);
        const size_t nk = (size_t)nsig * F * K * sizeof(double), n1 = (size_t)nsig * F * sizeof(double);
        memcpy(ho->f, h.f, nk); memcpy(ho->mag, h.mag, nk); memcpy(ho->ph, h.ph, nk); memcpy(ho->realph, h.realph, nk);
        memcpy(ho->binno, h.binno, nk);
        if (ho->totalmag) memcpy(ho->totalmag, h.totalmag, n1);
    } else {
        if ((rc = fetch(nchunks - 1)) != PVX_OK) return rc;
    }
    if (!keep && ho->t) {
        for (int64_t b = 0; b < nsig; b++)
            for (int64_t fr = 0; fr < F; fr++) ho->t[b * F + fr] = ((double)(fr * (int64_t)p->hop) + p->nfft / 2.0) / p->sr;   // PV.py:247
    }
    // the spectrum of the last frame (PV.oldfft after the loop, PV.py:209): float [N2][2] from the fused kernels,
    // or the last chunk's last row of the general path's workspace
    const size_t rs = real_size(p->fft_mode != 0 ? 32 : p->precision);
    const void* d_last = nullptr;
    if (last_spec && p->fft_mode != 0) d_last = p->d_specrow;
    else if (last_spec && p->last_from_rev) d_last = p->d_lastspec;
    else if (last_spec) {
        const int64_t lastR0 = ((last_rows - 1) / p->max_rows) * p->max_rows;
        const int64_t wsrow = (last_rows - 1) - lastR0 + 1;
        d_last = (const char*)p->d_spec + (size_t)wsrow * p->ldo * 2 * rs;
    }
    std::vector<unsigned char> tmp;
    unsigned char* h_last = nullptr;
    bool by_kernel = false;
    if (d_last && small && p->fft_mode != 0) {
        unsigned char* q = (unsigned char*)p->h_pin + ((((total_in + 255) & ~(size_t)255) + (keep ? 0 : total_out) + 255) & ~(size_t)255);
        if ((size_t)(q - (unsigned char*)p->h_pin) + (size_t)p->N2 * 8 <= p->pin_cap) { h_last = q; by_kernel = true; }   // as decided before the launch
    }
    if (d_last && small && !by_kernel) {
        // small call: into the pinned block behind the staged data, under the one synchronisation below
        h_last = (unsigned char*)p->h_pin + ((((total_in + 255) & ~(size_t)255) + (keep ? 0 : total_out) + 255) & ~(size_t)255);
        if ((size_t)(h_last - (unsigned char*)p->h_pin) + (size_t)p->N2 * 2 * rs > p->pin_cap) h_last = nullptr;
    }
    if (h_last && !by_kernel) PVX_HIP_CHECK(hipMemcpyAsync(h_last, d_last, (size_t)p->N2 * 2 * rs, hipMemcpyDeviceToHost, s));
    tr.mark("tail issued");
    PVX_HIP_CHECK(hipStreamSynchronize(s));
    tr.mark("synchronised");
    if (d_last) {
        if (!h_last) {
            tmp.resize((size_t)p->N2 * 2 * rs);
            h_last = tmp.data();
            PVX_HIP_CHECK(hipMemcpy(h_last, d_last, tmp.size(), hipMemcpyDeviceToHost));
        }
        for (int i = 0; i < 2 * p->N2; i++)
            last_spec[i] = rs == 4 ? (double)((const float*)h_last)[i] : ((const double*)h_last)[i];
    }
    if (p->progress_fn) p->progress_fn(nsig * F, nsig * F, p->progress_user);
    return F;
}

extern "C" int64_t pvx_analyze(pvx_plan* p, const void* x, int x_dtype, int64_t nsamp, int64_t nsig, int64_t sig_stride,
                               double* f, double* mag, double* ph, double* realph, double* binno, double* t,
                               double* totalmag, const double* prev0, double* last_spec) {
    HostOut ho = {f, mag, ph, realph, binno, t, totalmag};
    return analyze_host(p, x, x_dtype, nsamp, nsig, sig_stride, &ho, prev0, last_spec, false);
}

// ---- many independent signals over the GPUs of this process (SURVEY.md 8(b) pvx_analyze_batch, 8(e)) ------------------
// The path shards by signal and has no exchange step: every device analyses whole signals and its results go straight to
// the caller's arrays, so the devices never talk to each other (no RCCL here: that is for callers that live in separate
// processes, pypevoc_amd/batch.py).  One queue, longest signal first; `workers` host threads per device, each with its own
// plan and stream, so one signal's transfers run under another's kernels; a worker takes the next signal when it is free,
// which balances ragged batches and unequal devices without a schedule.
static_assert(sizeof(pvx_batch_item) == 88, "pvx_batch_item is part of the C ABI");
struct pvx_batch {
    double sr = 0, pkthresh = 0;
    int nfft = 0, hop = 0, npks = 0, precision = 32, workers = 4;
    std::vector<double> win;
    std::vector<int> devices;
    std::vector<pvx_plan*> plans;          // [device slot][worker], created by the worker on its first signal
};

extern "C" int pvx_batch_create(pvx_batch** out, double sr, int nfft, int hop, int npks, double pkthresh, const double* win,
                                int precision, const int* devices, int ndev, int workers_per_device) {
    if (!out) { pvx_set_error("null batch pointer"); return PVX_ERR_INVALID; }
    *out = nullptr;
    if (nfft < 4 || hop <= 0 || npks <= 0 || !(sr > 0) || (precision != 32 && precision != 64) || ndev < 0 || (ndev > 0 && !devices) ||
        workers_per_device < 0 || workers_per_device > 8) {
        pvx_set_error("invalid batch parameters (sr=%g nfft=%d hop=%d npks=%d precision=%d ndev=%d workers=%d)", sr, nfft, hop, npks, precision, ndev, workers_per_device);
        return PVX_ERR_INVALID;
    }
    int rc = pvx_require_device();
    if (rc != PVX_OK) return rc;
    int ndevices = 0;
    PVX_HIP_CHECK(hipGetDeviceCount(&ndevices));
    pvx_batch* b = new pvx_batch();
    b->sr = sr; b->nfft = nfft; b->hop = hop; b->npks = npks; b->pkthresh = pkthresh; b->precision = precision;
    b->workers = workers_per_device > 0 ? workers_per_device : 4;
    if (win) b->win.assign(win, win + nfft);
    if (ndev == 0) b->devices.push_back(t_device >= 0 ? t_device : g_device);
    for (int i = 0; i < ndev; i++) {
        const int d = devices[i];
        if (d < 0 || d >= ndevices || d >= kMaxDevices) { pvx_set_error("device %d out of range (%d devices)", d, ndevices); delete b; return PVX_ERR_INVALID; }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            pvx_set_error("device %d is not a gfx950 (MI355X) device; libpvx_hip carries gfx950 code objects only", d);
            delete b;
            return PVX_ERR_NO_DEVICE;
        }
        b->devices.push_back(d);
    }
    b->plans.assign(b->devices.size() * (size_t)b->workers, nullptr);
    *out = b;
    return PVX_OK;
}

extern "C" int pvx_batch_destroy(pvx_batch* b) {
    if (!b) return PVX_OK;
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (size_t i = 0; i < b->plans.size(); i++)
        if (b->plans[i]) { (void)hipSetDevice(b->devices[i / b->workers]); plan_free(b->plans[i]); }
    if (have) (void)hipSetDevice(cur);
    delete b;
    return PVX_OK;
}

extern "C" int64_t pvx_batch_run(pvx_batch* b, int x_dtype, pvx_batch_item* items, int64_t nitems) {
    if (!b || nitems < 0 || (nitems > 0 && !items)) { pvx_set_error("invalid batch arguments"); return PVX_ERR_INVALID; }
    if (x_dtype != PVX_F32 && x_dtype != PVX_F64 && x_dtype != PVX_I16) { pvx_set_error("unknown sample type %d", x_dtype); return PVX_ERR_INVALID; }
    if (nitems == 0) return 0;
    // (no C++ exception may leave an extern "C" function: a bad_alloc of the two vectors, or of a worker's plan, is
    // PVX_ERR_ALLOC; threads already started are always joined)
    std::vector<int64_t> order;
    std::vector<std::thread> th;
    const int nw = (int)b->plans.size();
    try {
        order.resize((size_t)nitems);
        th.reserve((size_t)nw);
    } catch (...) { pvx_set_error("pvx_batch_run: out of host memory for %lld items", (long long)nitems); return PVX_ERR_ALLOC; }
    for (int64_t i = 0; i < nitems; i++) { order[(size_t)i] = i; items[i].nframes = 0; items[i].device = -1; }
    try {
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t c) { return items[a].nsamp > items[c].nsamp; });
    } catch (...) { pvx_set_error("pvx_batch_run: out of host memory for %lld items", (long long)nitems); return PVX_ERR_ALLOC; }
    std::atomic<int64_t> next(0);
    std::atomic<int> first_rc(PVX_OK);
    std::mutex err_mu;
    char err_text[512] = "";
    auto worker = [&](int w) {
        const int dev = b->devices[(size_t)(w / b->workers)];
        t_device = dev;
        auto fail = [&](int64_t item, int rc) {
            if (item >= 0) items[item].nframes = rc;
            int expect = PVX_OK;
            if (first_rc.compare_exchange_strong(expect, rc)) {
                std::lock_guard<std::mutex> lk(err_mu);
                snprintf(err_text, sizeof(err_text), "signal %lld on device %d: %s", (long long)item, dev, g_err);
            }
        };
        for (;;) {
            const int64_t q = next.fetch_add(1);
            if (q >= nitems) break;
            const int64_t i = order[(size_t)q];
            pvx_batch_item& it = items[i];
            try {
            it.device = dev;
            if (!b->plans[(size_t)w]) {
                const int rc = pvx_plan_create(&b->plans[(size_t)w], b->sr, b->nfft, b->hop, b->npks, b->pkthresh, b->win.empty() ? nullptr : b->win.data(), b->precision, 0);
                if (rc != PVX_OK) { fail(i, rc); continue; }
            }
            const int64_t F = pvx_analyze(b->plans[(size_t)w], it.x, x_dtype, it.nsamp, 1, it.nsamp, it.f, it.mag, it.ph, it.realph, it.binno, it.t, it.totalmag, nullptr, nullptr);
            if (F < 0) { fail(i, (int)F); continue; }
            it.nframes = F;
            } catch (...) { pvx_set_error("out of host memory"); fail(i, PVX_ERR_ALLOC); }
        }
        t_device = -1;
    };
    const int nthreads = (int64_t)nw < nitems ? nw : (int)nitems;
    // workers are dealt to the devices in turn (worker w -> slot w % ndev's next worker), so a batch smaller than the pool
    // still spreads over the devices
    const int nd = (int)b->devices.size();
    auto slot_of = [&](int k) { return (k % nd) * b->workers + k / nd; };
    const int saved = t_device;
    try {
        for (int k = 1; k < nthreads; k++) th.emplace_back(worker, slot_of(k));      // (th has its capacity: only thread creation can fail)
    } catch (...) { /* the threads that started and this one share the queue */ }
    worker(slot_of(0));
    t_device = saved;
    for (auto& t : th) t.join();
    (void)pvx_require_device();            // this thread served a device: back to the caller's
    if (first_rc.load() != PVX_OK) { pvx_set_error("%s", err_text); return first_rc.load(); }
    int64_t total = 0;
    for (int64_t i = 0; i < nitems; i++) total += items[i].nframes;
    return total;
}

extern "C" int64_t pvx_analyze_batch(double sr, int nfft, int hop, int npks, double pkthresh, const double* win, int precision,
                                     int x_dtype, pvx_batch_item* items, int64_t nitems, const int* devices, int ndev) {
    pvx_batch* b = nullptr;
    const int rc = pvx_batch_create(&b, sr, nfft, hop, npks, pkthresh, win, precision, devices, ndev, 0);
    if (rc != PVX_OK) return rc;
    const int64_t r = pvx_batch_run(b, x_dtype, items, nitems);
    pvx_batch_destroy(b);
    return r;
}

// ---- page-locked host arrays for results (the DMA engine writes them directly) ------------------------
extern "C" void* pvx_host_alloc(size_t bytes) {
    if (pvx_require_device() != PVX_OK) return nullptr;
    void* q = nullptr;
    if (hipHostMalloc(&q, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); pvx_set_error("hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return q;
}
extern "C" void pvx_host_free(void* q) {
    if (q) (void)hipHostFree(q);
}

// ---- resident results ---------------------------------------------------------------------------
extern "C" int64_t pvx_analyze_resident(pvx_plan* p, const void* x, int x_dtype, int64_t nsamp, int64_t nsig,
                                        int64_t sig_stride, const double* prev0, double* last_spec) {
    return analyze_host(p, x, x_dtype, nsamp, nsig, sig_stride, nullptr, prev0, last_spec, true);
}

static int need_resident(pvx_plan* p) {
    if (!p) { pvx_set_error("null plan"); return PVX_ERR_INVALID; }
    if (!p->res_valid) { pvx_set_error("the plan holds no resident results (pvx_analyze_resident first)"); return PVX_ERR_INVALID; }
    return pvx_require_plan_device(p);
}

extern "C" int pvx_resident_fetch(pvx_plan* p, int which, double* host) {
    int rc = need_resident(p);
    if (rc != PVX_OK) return rc;
    if (!host || which < 0 || which > 6) { pvx_set_error("bad fetch argument"); return PVX_ERR_INVALID; }
    const int64_t rows = p->res_nsig * p->res_F;
    const HostOut all = block_ptrs(p->d_res, rows, p->npks);
    const double* src[7] = {all.f, all.mag, all.ph, all.realph, all.binno, all.t, all.totalmag};
    const size_t bytes = (size_t)rows * (which < 5 ? p->npks : 1) * sizeof(double);
    PVX_HIP_CHECK(hipStreamSynchronize(p->s_host));
    return device_to_host(host, src[which], bytes);
}

extern "C" const double* pvx_resident_ptr(pvx_plan* p, int which) {
    if (!p || !p->res_valid || which < 0 || which > 6) return nullptr;
    const HostOut all = block_ptrs(p->d_res, p->res_nsig * p->res_F, p->npks);
    const double* src[7] = {all.f, all.mag, all.ph, all.realph, all.binno, all.t, all.totalmag};
    return src[which];
}

extern "C" int pvx_stft_frames(pvx_plan* p, const void* x, int x_dtype, int64_t nsamp, const int64_t* pos, int64_t nfr,
                               double* spec) {
    int rc = pvx_require_plan_device(p);
    if (rc != PVX_OK) return rc;
    if (!p || !x || !pos || !spec || nfr < 0) { pvx_set_error("bad argument"); return PVX_ERR_INVALID; }
    if (x_dtype != PVX_F32 && x_dtype != PVX_F64 && x_dtype != PVX_I16) { pvx_set_error("bad x_dtype %d", x_dtype); return PVX_ERR_INVALID; }
    if ((rc = ensure_rocfft(p, false)) != PVX_OK) return rc;
    const size_t es = dtype_size(x_dtype), rs = real_size(p->precision);
    for (int64_t i = 0; i < nfr; i++)
        if (pos[i] < 0 || pos[i] + p->nfft > nsamp) { pvx_set_error("frame %lld at %lld leaves the signal", (long long)i, (long long)pos[i]); return PVX_ERR_INVALID; }
    // Each requested frame is framed as its own 1-frame "signal" (stride = its position), so the
    // regular framing kernel can be used: signal b = samples [pos[b], pos[b]+nfft+1).
    DevBuf dx;
    if ((rc = dx.alloc((size_t)(p->nfft + 1) * es)) != PVX_OK) return rc;
    const int nb = p->nfft / 2 + 1;
    std::vector<unsigned char> row((size_t)nb * 2 * rs);
    PVX_FFT_CHECK(rocfft_execution_info_set_stream(p->info, nullptr));
    for (int64_t i = 0; i < nfr; i++) {
        // copy nfft samples (+1 so that nframes(nfft+1) == 1)
        PVX_HIP_CHECK(hipMemset(dx.p, 0, (size_t)(p->nfft + 1) * es));
        PVX_HIP_CHECK(hipMemcpy(dx.p, (const char*)x + (size_t)pos[i] * es, (size_t)p->nfft * es, hipMemcpyHostToDevice));
        FrameParams fp;
        fp.x = dx.p; fp.nsamp = p->nfft + 1; fp.sig_stride = p->nfft + 1; fp.F = 1; fp.R0 = 0; fp.ws_rows = 3;
        fp.total_rows = 2; fp.nfft = p->nfft; fp.hop = p->hop; fp.win = p->d_win; fp.frames = p->d_frames; fp.ldi = p->ldi;
        if (p->rocfft_rows < 2) { pvx_set_error("plan workspace too small"); return PVX_ERR_SIZE; }
        rc = pvx_launch_frames(fp, x_dtype, p->precision, nullptr);
        if (rc != PVX_OK) return rc;
        void* rspec = p->rocfft_small ? p->d_rspec : p->d_spec;
        void* in[1] = {p->d_frames};
        void* out[1] = {rspec};
        PVX_FFT_CHECK(rocfft_execute(p->fft, in, out, p->info));
        PVX_HIP_CHECK(hipStreamSynchronize(nullptr));
        // global row 1 (frame 0) sits in workspace row 2
        PVX_HIP_CHECK(hipMemcpy(row.data(), (char*)rspec + (size_t)2 * p->ldo * 2 * rs, row.size(), hipMemcpyDeviceToHost));
        double* o = spec + (size_t)i * nb * 2;
        for (int k = 0; k < 2 * nb; k++)
            o[k] = p->precision == 32 ? (double)((float*)row.data())[k] : ((double*)row.data())[k];
    }
    return PVX_OK;
}

// ---- PeakFinder -----------------------------------------------------------------------------
extern "C" int pvx_peakfinder(const double* y, int64_t nrows, int n, int npeaks, int thr_kind, double thr_val, int rad,
                              int32_t* pos, int8_t* keep, int32_t* count, int cap) {
    int rc = pvx_require_device();
    if (rc != PVX_OK) return rc;
    if (!y || !pos || !keep || !count || nrows < 0 || n < 1 || cap < 1 || thr_kind < 0 || thr_kind > 2) {
        pvx_set_error("bad PeakFinder argument");
        return PVX_ERR_INVALID;
    }
    if (nrows == 0) return PVX_OK;
    DevBuf dy, dpos, dkeep, dcount;
    const size_t yb = (size_t)nrows * n * sizeof(double);
    if ((rc = dy.alloc(yb)) != PVX_OK || (rc = dpos.alloc((size_t)nrows * cap * 4)) != PVX_OK ||
        (rc = dkeep.alloc((size_t)nrows * cap)) != PVX_OK || (rc = dcount.alloc((size_t)nrows * 4)) != PVX_OK)
        return rc;
    if ((rc = host_to_device(dy.p, y, yb)) != PVX_OK) return rc;
    PVX_HIP_CHECK(hipMemset(dpos.p, 0xff, (size_t)nrows * cap * 4));
    PVX_HIP_CHECK(hipMemset(dkeep.p, 0, (size_t)nrows * cap));
    PeakRowsParams pp;
    pp.y = (const double*)dy.p; pp.nrows = nrows; pp.n = n; pp.npeaks = npeaks; pp.thr_kind = thr_kind;
    pp.thr_val = thr_val; pp.rad = rad; pp.cap = cap;
    pp.pos = (int32_t*)dpos.p; pp.keep = (int8_t*)dkeep.p; pp.count = (int32_t*)dcount.p;
    rc = pvx_launch_peak_rows(pp, nullptr);
    if (rc != PVX_OK) return rc;
    PVX_HIP_CHECK(hipStreamSynchronize(nullptr));
    if ((rc = device_to_host(pos, dpos.p, (size_t)nrows * cap * 4)) != PVX_OK || (rc = device_to_host(keep, dkeep.p, (size_t)nrows * cap)) != PVX_OK ||
        (rc = device_to_host(count, dcount.p, (size_t)nrows * 4)) != PVX_OK) return rc;
    return PVX_OK;
}

// ---- tracker: PV.toSinSum (PV.py:299-322) ---------------------------------------------------
static size_t track_ws_bytes(int64_t F, int K) {
    const size_t n = (size_t)F * K;
    // link, root: int32 [F*K]; newcount int32 [F]; newbase int64 [F+1]; npartials, ambiguous, maxend int64; succ [F*K]
    size_t off = n * 4 * 2 + (size_t)F * 4;
    off = (off + 7) & ~(size_t)7;
    off += ((size_t)F + 1) * 8 + 32 + n;
    off = (off + 7) & ~(size_t)7;
    const size_t nch = (size_t)(F + 127) / 128;                  // k_track_links_g8 / _lane: chunkbase int64 [nch + 1], chunktot / chunklast int32 [nch] (chunks of 128 or 256 frames)
    return off + (nch + 1) * 8 + nch * 8;
}

// the tracker on device arrays with a caller-provided workspace of track_ws_bytes(F, K); returns P
static int64_t track_on(const double* d_f, const double* d_mag, int64_t F, int K, double maxpitchjmp,
                        int32_t* d_partial_id, int32_t* d_part_start, int32_t* d_part_len, int64_t cap, char* w,
                        hipStream_t s, int64_t* maxend, int64_t* pinned3 = nullptr) {
    const size_t n = (size_t)F * K;
    const size_t off_link = 0, off_root = off_link + n * 4, off_cnt = off_root + n * 4;
    const size_t off_base = (off_cnt + (size_t)F * 4 + 7) & ~(size_t)7;
    const size_t off_np = off_base + ((size_t)F + 1) * 8, off_succ = off_np + 32;
    TrackParams tp;
    tp.f = d_f; tp.mag = d_mag; tp.F = F; tp.K = K; tp.maxjmp = maxpitchjmp;
    tp.partial_id = d_partial_id; tp.part_start = d_part_start; tp.part_len = d_part_len; tp.cap = cap;
    tp.link = (int32_t*)(w + off_link); tp.root = (int32_t*)(w + off_root);
    tp.succ = (unsigned char*)(w + off_succ); tp.newcount = (int32_t*)(w + off_cnt); tp.newbase = (int64_t*)(w + off_base);
    {
        const size_t nch = (size_t)(F + 127) / 128, off_cb = (off_succ + n + 7) & ~(size_t)7;
        tp.chunkbase = (int64_t*)(w + off_cb); tp.chunktot = (int32_t*)(w + off_cb + (nch + 1) * 8); tp.chunklast = tp.chunktot + nch;
    }
    // { partials, exact double tie met, last frame with a point }: three single stores by the kernels.  In page-locked
    // host memory when the caller has some (the resident chain): the host reads them after the one synchronisation,
    // no copy operation behind the kernels.
    // (the fourth word: k_track_links_lane's report of a frame too wide for it -- the call's number, see TrackParams::wide)
    int64_t pa_[4] = {0, 0, -1, 0};
    volatile int64_t* pa = pinned3 ? pinned3 : pa_;
    tp.npartials = pinned3 ? pinned3 : (int64_t*)(w + off_np);
    tp.ambiguous = tp.npartials + 1;
    tp.maxend = tp.npartials + 2;
    static std::atomic<unsigned> calls{0};
    tp.wide = (unsigned*)(tp.npartials + 3);
    tp.wide_dev = (unsigned*)(w + off_np + 24) + 1;              // (the workspace's own copy of the words is device memory either way)
    do { tp.gen = ++calls; } while (tp.gen == 0);
    if (K > 8) {                                                   // (rows of at most 8 slots never use the words)
        if (pinned3) pinned3[3] = 0;
        PVX_HIP_CHECK(hipMemsetAsync(w + off_np + 24, 0, 8, s));
    }
    int rc = pvx_launch_track(tp, s);
    if (rc != PVX_OK) return rc;
    if (!pinned3) PVX_HIP_CHECK(hipMemcpyAsync(pa_, tp.npartials, 32, hipMemcpyDeviceToHost, s));
    PVX_HIP_CHECK(hipStreamSynchronize(s));
    if ((unsigned)pa[3] == tp.gen) {
        // a frame with a valid peak beyond slot 7: the table of the wave-per-frame kernels instead
        tp.wide = nullptr;
        if ((rc = pvx_launch_track(tp, s)) != PVX_OK) return rc;
        if (!pinned3) PVX_HIP_CHECK(hipMemcpyAsync(pa_, tp.npartials, 24, hipMemcpyDeviceToHost, s));
        PVX_HIP_CHECK(hipStreamSynchronize(s));
    }
    if (pa[1] != 0 && getenv("PVX_TRACK_FORBID_SEQUENTIAL")) {      // tests: which tables the frame-parallel kernels hand over
        pvx_set_error("the frame-parallel tracker kernels left this table to k_track_sequential");
        return PVX_ERR_UNSUPPORTED;
    }
    if (pa[1] != 0 || getenv("PVX_TRACK_SEQUENTIAL")) {
        // an exact double tie (k_track.hip): the reference's order of the previous partials decides; redo the
        // table with the sequential kernel, which has the partial indices at hand
        if ((rc = pvx_launch_track_sequential(tp, s)) != PVX_OK) return rc;
        if (!pinned3) PVX_HIP_CHECK(hipMemcpyAsync(pa_, tp.npartials, 24, hipMemcpyDeviceToHost, s));
        PVX_HIP_CHECK(hipStreamSynchronize(s));
    }
    if (maxend) *maxend = pa[2];
    if (pa[0] > cap) { pvx_set_error("%lld partials exceed the table capacity %lld", (long long)pa[0], (long long)cap); return PVX_ERR_SIZE; }
    return pa[0];
}

extern "C" int64_t pvx_track_dev(const double* d_f, const double* d_mag, int64_t F, int K, double maxpitchjmp,
                                 int32_t* d_partial_id, int32_t* d_part_start, int32_t* d_part_len, int64_t cap,
                                 void* stream) {
    int rc = pvx_require_device();
    if (rc != PVX_OK) return rc;
    if (F < 0 || K <= 0 || cap < 0) { pvx_set_error("bad tracker argument"); return PVX_ERR_INVALID; }
    if (F == 0) return 0;
    if (!d_f || !d_mag || !d_partial_id || !d_part_start || !d_part_len) { pvx_set_error("null tracker array"); return PVX_ERR_INVALID; }
    // the workspace of the plan-less tracker is kept (allocating and freeing it cost as much as the kernels); the call is
    // synchronous, one at a time through it
    // (one workspace per DEVICE: a thread bound to another device must not hand its kernels a buffer of this one; the result
    // words are page-locked memory every device can write: hipHostMallocPortable)
    struct DevWs { char* ws = nullptr; size_t cap = 0; };
    static std::mutex mu;
    static std::map<int, DevWs> wsd;
    static int64_t* pin3 = nullptr;         // { partials, exact double tie, last frame with a point } in page-locked memory:
    std::lock_guard<std::mutex> lk(mu);     // the kernel's three stores are what the host waits for, no copy behind them
    if (!pin3 && hipHostMalloc((void**)&pin3, 64, hipHostMallocPortable) != hipSuccess) { pin3 = nullptr; (void)hipGetLastError(); }
    int dev = 0;
    PVX_HIP_CHECK(hipGetDevice(&dev));
    DevWs* wp = nullptr;
    try { wp = &wsd[dev]; } catch (...) { pvx_set_error("out of memory for the tracker's per-device workspace record"); return PVX_ERR_ALLOC; }   // (no C++ exception leaves the C ABI)
    DevWs& w = *wp;
    if ((rc = grow_dev(&w.ws, &w.cap, track_ws_bytes(F, K))) != PVX_OK) return rc;
    return track_on(d_f, d_mag, F, K, maxpitchjmp, d_partial_id, d_part_start, d_part_len, cap, w.ws, (hipStream_t)stream, nullptr, pin3);
}

extern "C" int pvx_synth_dev_flags(const double* d_f, const double* d_mag, const double* d_realph, const int32_t* d_partial_id,
                             int64_t F, int K, const int32_t* d_part_start, const int32_t* d_part_len, int64_t P,
                             double sr, int nfft, int hop_analysis, int hop_synth, double edge, int minframes,
                             double* d_w, int64_t wlen, void* stream, int flags);
extern "C" int64_t pvx_synth_len(int64_t max_end_frame, int nfft, int hop_analysis, int hop_synth, double edge);
static int synth_slice(const double* d_f, const double* d_mag, const double* d_realph, const int32_t* d_partial_id, int64_t F, int K,
                       const int32_t* d_part_start, const int32_t* d_part_len, int64_t P, double sr, int nfft, int hop_analysis, int hop_synth,
                       double edge, int minframes, double* d_w, int64_t wlen, hipStream_t stream, int64_t seg0, int64_t seg_count, bool first,
                       void* ws, size_t ws_bytes, unsigned* ws_gen, int f32_samples);
// plans at precision 32 resynthesise with the float32 sample loop (k_synth_bodies<R, float>: the stated waveform tolerance there is
// 1e-4 max|w|); PVX_SYNTH_F64=1: the float64 loop whatever the plan's precision (tests, A/B)
static int plan_synth_f32(const pvx_plan* p);

// ---- the chain on resident results: toSinSum -> synth, descriptors -------------------------------
extern "C" int64_t pvx_track_resident(pvx_plan* p, double maxpitchjmp, int64_t* max_end_frame) {
    int rc = need_resident(p);
    if (rc != PVX_OK) return rc;
    if (p->res_nsig != 1) { pvx_set_error("the tracker works on one signal (resident results hold %lld)", (long long)p->res_nsig); return PVX_ERR_INVALID; }
    const int64_t F = p->res_F;
    const int K = p->npks;
    const size_t n = (size_t)F * K;
    if (n > p->trk_cap) {
        if (p->d_pid) (void)hipFree(p->d_pid);
        if (p->d_pst) (void)hipFree(p->d_pst);
        if (p->d_pln) (void)hipFree(p->d_pln);
        p->d_pid = p->d_pst = p->d_pln = nullptr; p->trk_cap = 0;
        if (hipMalloc((void**)&p->d_pid, n * 4) != hipSuccess || hipMalloc((void**)&p->d_pst, n * 4) != hipSuccess ||
            hipMalloc((void**)&p->d_pln, n * 4) != hipSuccess) { pvx_set_error("hipMalloc of the partial table failed"); return PVX_ERR_ALLOC; }
        p->trk_cap = n;
    }
    if ((rc = grow_dev(&p->d_tws, &p->tws_cap, track_ws_bytes(F, K))) != PVX_OK) return rc;
    const HostOut all = block_ptrs(p->d_res, F, K);
    int64_t maxend = -1;
    if ((rc = grow_pin(p, 64)) != PVX_OK) return rc;
    HostTrace tr("track");
    const int64_t P = track_on(all.f, all.mag, F, K, maxpitchjmp, p->d_pid, p->d_pst, p->d_pln, (int64_t)n, (char*)p->d_tws, p->s_host, &maxend,
                               (int64_t*)p->h_pin);
    tr.mark("launched + synchronised");
    if (P < 0) return P;
    p->res_P = P; p->res_maxend = maxend;
    if (max_end_frame) *max_end_frame = maxend;
    return P;
}

extern "C" int pvx_resident_fetch_table(pvx_plan* p, int32_t* partial_id, int32_t* part_start, int32_t* part_len) {
    int rc = need_resident(p);
    if (rc != PVX_OK) return rc;
    if (p->res_P < 0) { pvx_set_error("no resident partial table (pvx_track_resident first)"); return PVX_ERR_INVALID; }
    PVX_HIP_CHECK(hipStreamSynchronize(p->s_host));
    if (partial_id && (rc = device_to_host(partial_id, p->d_pid, (size_t)p->res_F * p->npks * 4)) != PVX_OK) return rc;
    if (part_start && p->res_P && (rc = device_to_host(part_start, p->d_pst, (size_t)p->res_P * 4)) != PVX_OK) return rc;
    if (part_len && p->res_P && (rc = device_to_host(part_len, p->d_pln, (size_t)p->res_P * 4)) != PVX_OK) return rc;
    return PVX_OK;
}

extern "C" int pvx_synth_resident(pvx_plan* p, double sr, int hop_synth, double edge, int minframes, double* w, int64_t wlen) {
    int rc = need_resident(p);
    if (rc != PVX_OK) return rc;
    if (p->res_P < 0) { pvx_set_error("no resident partial table (pvx_track_resident first)"); return PVX_ERR_INVALID; }
    if (!w || hop_synth <= 0 || !(sr > 0)) { pvx_set_error("bad resynthesis argument"); return PVX_ERR_INVALID; }
    if (p->res_P == 0) { pvx_set_error("max() arg is an empty sequence"); return PVX_ERR_INVALID; }          // PV.py:1059
    const int64_t need = pvx_synth_len(p->res_maxend, p->nfft, p->hop, hop_synth, edge);
    if (need < 0 || need != wlen) { pvx_set_error("output length %lld, expected %lld", (long long)wlen, (long long)need); return PVX_ERR_SIZE; }
    HostTrace tr("synth");
    {
        const void* before = p->d_sws;
        const size_t cap_before = p->sws_cap;
        if ((rc = grow_dev(&p->d_sws, &p->sws_cap, pvx_synth_ws_bytes(p->res_F, p->npks, p->res_P, p->nfft, p->hop, hop_synth, edge))) != PVX_OK) return rc;
        if (p->d_sws != before || p->sws_cap != cap_before) p->sws_gen = 0;           // a new buffer: its flags are cleared on first use
    }
    const HostOut all = block_ptrs(p->d_res, p->res_F, p->npks);
    const size_t bytes = (size_t)wlen * 8;
    hipPointerAttribute_t attr;
    const bool pinned = hipPointerGetAttributes(&attr, w) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();                        // an ordinary pointer is reported as an error
    // (off by default since the attacks / releases reach the waveform in a launch of their own and k_synth_bodies reads them
    // back: over PCIe that costs more than the DMA it saves -- config 3: 0.21 ms against 0.17 ms)
    static const bool zero_copy = getenv("PVX_SYNTH_ZEROCOPY") != nullptr;
    if (pinned && zero_copy && bytes <= kSmallCall) {
        // the caller's array is page-locked (pvx_host_alloc) and small: the kernel stores its segments straight into
        // it (posted writes over PCIe, spread over the kernel's run time as workgroups finish) -- no copy operation
        // behind the kernel at all
        rc = synth_slice(all.f, all.mag, all.realph, p->d_pid, p->res_F, p->npks, p->d_pst, p->d_pln, p->res_P, sr, p->nfft,
                         p->hop, hop_synth, edge, minframes, w, wlen, p->s_host, 0, 0, true, p->d_sws, p->sws_cap, &p->sws_gen, plan_synth_f32(p));
        if (rc != PVX_OK) return rc;
        tr.mark("kernel issued");
        PVX_HIP_CHECK(hipStreamSynchronize(p->s_host));
        tr.mark("here");
        return PVX_OK;
    }
    if ((rc = grow_dev(&p->d_w, &p->w_cap, (size_t)wlen * 8)) != PVX_OK) return rc;
    if (pinned && bytes >= kStageMin && getenv("PVX_NO_SYNTH_SLICES") == nullptr) {
        // a large waveform into page-locked memory: the segments are computed in slices and the DMA of a finished slice
        // (second stream) runs under the next slice's kernel -- the link, not kernel + link, is what the call costs
        if ((rc = stage_ring(p)) != PVX_OK) return rc;                   // (its events)
        if (!p->s_copy) PVX_HIP_CHECK(hipStreamCreateWithFlags(&p->s_copy, hipStreamNonBlocking));
        const int NS = 8;                                                // slices (ev_ring holds 16 events)
        const int64_t nseg = (wlen + hop_synth - 1) / hop_synth, per = (nseg + NS - 1) / NS;
        for (int i = 0; i < NS; i++) {
            const int64_t s0 = (int64_t)i * per;
            if (s0 >= nseg) break;
            const int64_t cnt = nseg - s0 < per ? nseg - s0 : per;
            rc = synth_slice(all.f, all.mag, all.realph, p->d_pid, p->res_F, p->npks, p->d_pst, p->d_pln, p->res_P, sr, p->nfft, p->hop, hop_synth,
                             edge, minframes, p->d_w, wlen, p->s_host, s0, cnt, i == 0, p->d_sws, p->sws_cap, &p->sws_gen, plan_synth_f32(p));
            if (rc != PVX_OK) return rc;
            PVX_HIP_CHECK(hipEventRecord(p->ev_ring[i], p->s_host));
            PVX_HIP_CHECK(hipStreamWaitEvent(p->s_copy, p->ev_ring[i], 0));
            const int64_t o = s0 * hop_synth, c = (wlen - o < cnt * hop_synth) ? wlen - o : cnt * hop_synth;
            PVX_HIP_CHECK(hipMemcpyAsync(w + o, p->d_w + o, (size_t)c * 8, hipMemcpyDeviceToHost, p->s_copy));
        }
        tr.mark("slices + copies issued");
        PVX_HIP_CHECK(hipStreamSynchronize(p->s_copy));
        PVX_HIP_CHECK(hipStreamSynchronize(p->s_host));
        tr.mark("here");
        return PVX_OK;
    }
    rc = synth_slice(all.f, all.mag, all.realph, p->d_pid, p->res_F, p->npks, p->d_pst, p->d_pln, p->res_P, sr, p->nfft,
                     p->hop, hop_synth, edge, minframes, p->d_w, wlen, p->s_host, 0, 0, true, p->d_sws, p->sws_cap, &p->sws_gen, plan_synth_f32(p));
    if (rc != PVX_OK) return rc;
    if (pinned) {
        // the caller's array is page-locked: the DMA lands in it, nothing to stage or copy
        PVX_HIP_CHECK(hipMemcpyAsync(w, p->d_w, bytes, hipMemcpyDeviceToHost, p->s_host));
        tr.mark("kernel + copy issued");
        PVX_HIP_CHECK(hipStreamSynchronize(p->s_host));
        tr.mark("here");
        return PVX_OK;
    }
    if (bytes <= kSmallCall) {
        // through pinned memory in pieces: the host copy of piece i runs under the DMA of piece i+1 (two events in turn)
        if ((rc = grow_pin(p, bytes)) != PVX_OK) return rc;
        const size_t piece = bytes > ((size_t)256 << 10) ? ((bytes + 3) / 4 + 255) & ~(size_t)255 : bytes;
        const int np_ = (int)((bytes + piece - 1) / piece);
        auto issue = [&](int i) -> int {
            const size_t o = (size_t)i * piece, c = bytes - o < piece ? bytes - o : piece;
            PVX_HIP_CHECK(hipMemcpyAsync((char*)p->h_pin + o, (const char*)p->d_w + o, c, hipMemcpyDeviceToHost, p->s_host));
            PVX_HIP_CHECK(hipEventRecord(p->ev_done[i & 1], p->s_host));
            return PVX_OK;
        };
        if ((rc = issue(0)) != PVX_OK) return rc;
        if (np_ > 1 && (rc = issue(1)) != PVX_OK) return rc;
        tr.mark("kernel + copies issued");
        for (int i = 0; i < np_; i++) {
            const size_t o = (size_t)i * piece, c = bytes - o < piece ? bytes - o : piece;
            PVX_HIP_CHECK(hipEventSynchronize(p->ev_done[i & 1]));
            if (i + 2 < np_ && (rc = issue(i + 2)) != PVX_OK) return rc;       // its event is free again
            memcpy((char*)w + o, (const char*)p->h_pin + o, c);
        }
        tr.mark("copied out");
    } else {
        const bool threaded = getenv("PVX_NO_STAGE_THREADS") == nullptr;
        if (threaded && bytes >= kStageMin) {
            // (the DMAs are queued behind the kernel on the same stream)
            if ((rc = staged_copy(p, p->d_w, w, bytes, false, p->s_host)) != PVX_OK) return rc;
            tr.mark("copied out (threaded ring)");
        } else {
            PVX_HIP_CHECK(hipStreamSynchronize(p->s_host));
            if ((rc = device_to_host(w, p->d_w, bytes)) != PVX_OK) return rc;
        }
    }
    return PVX_OK;
}

// PV.calc_f0 (PV.py:371-391) on the resident arrays: fm float64 [F], idx int32 [F] come back, nothing else moves
extern "C" int pvx_f0_resident(pvx_plan* p, double fmin, double fmax, double thr, double* fm, int32_t* idx) {
    int rc = need_resident(p);
    if (rc != PVX_OK) return rc;
    if (!fm || !idx) { pvx_set_error("null output array"); return PVX_ERR_INVALID; }
    const int64_t rows = p->res_nsig * p->res_F;
    if ((rc = grow_dev(&p->d_desc, &p->desc_cap, (size_t)rows * 12)) != PVX_OK) return rc;
    const HostOut all = block_ptrs(p->d_res, rows, p->npks);
    double* d_fm = (double*)p->d_desc;
    int32_t* d_im = (int32_t*)(d_fm + rows);
    if ((rc = pvx_launch_f0(all.f, all.mag, rows, p->npks, fmin, fmax, thr, d_fm, d_im, p->s_host)) != PVX_OK) return rc;
    PVX_HIP_CHECK(hipStreamSynchronize(p->s_host));
    if ((rc = device_to_host(fm, d_fm, (size_t)rows * 8)) != PVX_OK || (rc = device_to_host(idx, d_im, (size_t)rows * 4)) != PVX_OK) return rc;
    return PVX_OK;
}

// PV.calc_harmonic_power (PV.py:266-297) on the resident arrays: hpower, nharmonics float64 [F, K].
// Returns PVX_ERR_SIZE where the reference raises IndexError (a valid peak slot >= number of frames, PV.py:278).
extern "C" int pvx_harmonic_power_resident(pvx_plan* p, double f_threshold, double* hpower, double* nharm) {
    int rc = need_resident(p);
    if (rc != PVX_OK) return rc;
    if (!hpower || !nharm) { pvx_set_error("null output array"); return PVX_ERR_INVALID; }
    if (p->res_nsig != 1) { pvx_set_error("calc_harmonic_power works on one signal"); return PVX_ERR_INVALID; }
    const int64_t F = p->res_F;
    const int K = p->npks;
    const size_t n = (size_t)F * K;
    if ((rc = grow_dev(&p->d_desc, &p->desc_cap, n * 16 + (size_t)K * 8 + 16)) != PVX_OK) return rc;
    const HostOut all = block_ptrs(p->d_res, F, K);
    double* d_hp = (double*)p->d_desc;
    double* d_nh = d_hp + n;
    double* d_rp = d_nh + n;
    int32_t* d_top = (int32_t*)(d_rp + K);
    if ((rc = pvx_launch_hpower_rows(all.f, all.mag, F, K, d_rp, d_top, p->s_host)) != PVX_OK) return rc;
    int32_t top = -1;
    PVX_HIP_CHECK(hipMemcpyAsync(&top, d_top, 4, hipMemcpyDeviceToHost, p->s_host));
    PVX_HIP_CHECK(hipStreamSynchronize(p->s_host));
    if (top >= F) { pvx_set_error("index %d is out of bounds for axis 0 with size %lld", (int)top, (long long)F); return PVX_ERR_SIZE; }
    if ((rc = pvx_launch_hpower(all.f, F, K, f_threshold, d_rp, d_hp, d_nh, p->s_host)) != PVX_OK) return rc;
    PVX_HIP_CHECK(hipStreamSynchronize(p->s_host));
    if ((rc = device_to_host(hpower, d_hp, n * 8)) != PVX_OK || (rc = device_to_host(nharm, d_nh, n * 8)) != PVX_OK) return rc;
    return PVX_OK;
}

extern "C" int64_t pvx_track(const double* f, const double* mag, int64_t F, int K, double maxpitchjmp,
                             int32_t* partial_id, int32_t* part_start, int32_t* part_len, int64_t cap) {
    int rc = pvx_require_device();
    if (rc != PVX_OK) return rc;
    if (F < 0 || K <= 0 || cap < 0) { pvx_set_error("bad tracker argument"); return PVX_ERR_INVALID; }
    if (F == 0) return 0;
    if (!f || !mag || !partial_id || !part_start || !part_len) { pvx_set_error("null tracker array"); return PVX_ERR_INVALID; }
    const size_t n = (size_t)F * K;
    const int64_t dcap = (int64_t)n;           // always sufficient on the device side
    DevBuf df, dm, dpid, dst, dln;
    if ((rc = df.alloc(n * 8)) != PVX_OK || (rc = dm.alloc(n * 8)) != PVX_OK || (rc = dpid.alloc(n * 4)) != PVX_OK ||
        (rc = dst.alloc(n * 4)) != PVX_OK || (rc = dln.alloc(n * 4)) != PVX_OK)
        return rc;
    if ((rc = host_to_device(df.p, f, n * 8)) != PVX_OK || (rc = host_to_device(dm.p, mag, n * 8)) != PVX_OK) return rc;
    const int64_t P = pvx_track_dev((const double*)df.p, (const double*)dm.p, F, K, maxpitchjmp, (int32_t*)dpid.p,
                                    (int32_t*)dst.p, (int32_t*)dln.p, dcap, nullptr);
    if (P < 0) return P;
    if (P > cap) { pvx_set_error("%lld partials exceed the caller's capacity %lld", (long long)P, (long long)cap); return PVX_ERR_SIZE; }
    if ((rc = device_to_host(partial_id, dpid.p, n * 4)) != PVX_OK || (rc = device_to_host(part_start, dst.p, (size_t)P * 4)) != PVX_OK ||
        (rc = device_to_host(part_len, dln.p, (size_t)P * 4)) != PVX_OK) return rc;
    return P;
}

// ---- PVHarmonic.run_pv (PV.py:493-535): f0-guided analysis on the general path's spectra ------
static int harmonic_rows(pvx_plan* p, const void* d_x, int x_dtype, int64_t nsamp, int64_t F, const double* f0,
                         double fmin, double* d_f, double* d_mag, double* d_ph, double* d_res, double* d_t,
                         const double* d_prev0, hipStream_t s, bool* any_valid) {
    int rc;
    if (p->use_stft) { if ((rc = ensure_spec_ws(p)) != PVX_OK) return rc; }
    else if ((rc = ensure_rocfft(p, true)) != PVX_OK) return rc;
    const size_t rs = real_size(p->precision);
    // previous-valid-frame table (PV.py:509, 491: oldfft only moves on analysed frames)
    std::vector<int32_t> prow((size_t)F);
    int64_t last = -1;
    for (int64_t fr = 0; fr < F; fr++) {
        const int64_t R0 = ((fr + 1) / p->max_rows) * p->max_rows;      // chunk holding row fr+1
        if (last < 0) prow[fr] = -1;
        else if (last + 1 >= R0 - 1) prow[fr] = (int32_t)(last + 1 - R0 + 1);
        else prow[fr] = -2;
        const double v = f0[fr];
        if (v > 0.0) {
            if (v / p->sr * (double)p->nfft < 0.5) {
                pvx_set_error("f0[%lld] = %g Hz is below half a bin (%g Hz): the harmonic series is not resolvable", (long long)fr, v, 0.5 * p->fstep);
                return PVX_ERR_INVALID;
            }
            last = fr;
        }
    }
    *any_valid = last >= 0;
    if (F > p->harm_cap) {
        if (p->d_hf0) (void)hipFree(p->d_hf0);
        if (p->d_hprev) (void)hipFree(p->d_hprev);
        p->d_hf0 = nullptr; p->d_hprev = nullptr; p->harm_cap = 0;
        if (hipMalloc((void**)&p->d_hf0, (size_t)F * 8) != hipSuccess || hipMalloc((void**)&p->d_hprev, (size_t)F * 4) != hipSuccess) {
            pvx_set_error("hipMalloc of the f0 tables failed"); return PVX_ERR_ALLOC;
        }
        p->harm_cap = F;
    }
    if (!p->d_carry && hipMalloc(&p->d_carry, (size_t)p->ldo * 2 * rs) != hipSuccess) { pvx_set_error("hipMalloc(carry) failed"); return PVX_ERR_ALLOC; }
    PVX_HIP_CHECK(hipMemcpyAsync(p->d_hf0, f0, (size_t)F * 8, hipMemcpyHostToDevice, s));
    PVX_HIP_CHECK(hipMemcpyAsync(p->d_hprev, prow.data(), (size_t)F * 4, hipMemcpyHostToDevice, s));
    PVX_HIP_CHECK(hipStreamSynchronize(s));                              // prow is a local
    if (!p->use_stft) PVX_FFT_CHECK(rocfft_execution_info_set_stream(p->info, s));
    const int64_t total_rows = F + 1;
    for (int64_t R0 = 0; R0 < total_rows; R0 += p->max_rows) {
        const int64_t nrows = (total_rows - R0 < p->max_rows) ? (total_rows - R0) : p->max_rows;
        FrameParams fp;
        fp.x = d_x; fp.nsamp = nsamp; fp.sig_stride = nsamp; fp.F = F; fp.R0 = R0;
        fp.ws_rows = nrows + 1; fp.total_rows = total_rows; fp.nfft = p->nfft; fp.hop = p->hop;
        fp.win = p->d_win; fp.frames = p->d_frames; fp.ldi = p->ldi;
        if (p->use_stft) {
            if ((rc = pvx_launch_stft(fp, p->d_spec, p->ldo, p->d_twiddle64, x_dtype, p->precision, s)) != PVX_OK) return rc;
        } else {
            if ((rc = pvx_launch_frames(fp, x_dtype, p->precision, s)) != PVX_OK) return rc;
            void* in[1] = {p->d_frames};
            void* out[1] = {p->d_spec};
            PVX_FFT_CHECK(rocfft_execute(p->fft, in, out, p->info));
        }
        HarmParams hp;
        hp.spec = p->d_spec; hp.ldo = p->ldo;
        hp.fr_begin = (R0 > 1 ? R0 : 1) - 1;
        const int64_t fr_last = R0 + nrows - 2;
        hp.nfr = fr_last - hp.fr_begin + 1;
        hp.ws_off = hp.fr_begin + 1 - R0 + 1;
        hp.nfft = p->nfft; hp.hop = p->hop; hp.N2 = p->N2; hp.K = p->npks;
        hp.sr = p->sr; hp.fstep = p->fstep; hp.dt = p->dt; hp.fmin = fmin;
        hp.wfbin = p->d_wfbin; hp.prev0 = d_prev0; hp.carry = p->d_carry;
        hp.f0 = p->d_hf0; hp.prevrow = p->d_hprev;
        hp.f = d_f; hp.mag = d_mag; hp.ph = d_ph; hp.residual = d_res; hp.t = d_t;
        if ((rc = pvx_launch_harmonic(hp, p->precision, s)) != PVX_OK) return rc;
        // carry the spectrum of the last valid frame of this chunk to the later ones
        int64_t lv = fr_last;
        while (lv >= hp.fr_begin && !(f0[lv] > 0.0)) lv--;
        if (lv >= hp.fr_begin)
            PVX_HIP_CHECK(hipMemcpyAsync(p->d_carry, (const char*)p->d_spec + (size_t)(lv + 1 - R0 + 1) * p->ldo * 2 * rs,
                                         (size_t)p->ldo * 2 * rs, hipMemcpyDeviceToDevice, s));
        if ((rc = plan_progress(p, s, R0 + nrows, total_rows, 1)) != PVX_OK) return rc;
    }
    return PVX_OK;
}

static int check_harmonic_args(pvx_plan* p, const void* x, int x_dtype, int64_t nsamp, const double* f0, int64_t nf0, int64_t F) {
    int rc = check_analyze_args(p, x, x_dtype, nsamp, 1, nsamp, nullptr);
    if (rc != PVX_OK) return rc;
    if (F > 0 && (!f0 || nf0 < F)) {
        // the reference indexes f0[int(curpos/hop)] (PV.py:507) and raises IndexError
        pvx_set_error("index %lld is out of bounds for f0 with size %lld", (long long)(nf0 < 0 ? 0 : nf0), (long long)nf0);
        return PVX_ERR_SIZE;
    }
    return PVX_OK;
}

extern "C" int64_t pvx_harmonic_analyze_dev(pvx_plan* p, const void* d_x, int x_dtype, int64_t nsamp, const double* f0,
                                            int64_t nf0, double fmin, double* d_f, double* d_mag, double* d_ph,
                                            double* d_residual, double* d_t, const double* d_prev0, void* stream) {
    int rc = pvx_require_plan_device(p);
    if (rc != PVX_OK) return rc;
    if (!p) { pvx_set_error("null plan"); return PVX_ERR_INVALID; }
    const int64_t F = pvx_nframes(nsamp, p->nfft, p->hop);
    if ((rc = check_harmonic_args(p, d_x, x_dtype, nsamp, f0, nf0, F)) != PVX_OK) return rc;
    if (F == 0) return 0;
    if (!d_f || !d_mag || !d_ph || !d_residual) { pvx_set_error("null output array"); return PVX_ERR_INVALID; }
    bool any = false;
    rc = harmonic_rows(p, d_x, x_dtype, nsamp, F, f0, fmin, d_f, d_mag, d_ph, d_residual, d_t, d_prev0, (hipStream_t)stream, &any);
    return rc == PVX_OK ? F : rc;
}

extern "C" int64_t pvx_harmonic_analyze(pvx_plan* p, const void* x, int x_dtype, int64_t nsamp, const double* f0, int64_t nf0,
                                        double fmin, double* f, double* mag, double* ph, double* residual, double* t,
                                        const double* prev0, double* last_spec) {
    int rc = pvx_require_plan_device(p);
    if (rc != PVX_OK) return rc;
    if (!p) { pvx_set_error("null plan"); return PVX_ERR_INVALID; }
    const int64_t F = pvx_nframes(nsamp, p->nfft, p->hop);
    if ((rc = check_harmonic_args(p, x, x_dtype, nsamp, f0, nf0, F)) != PVX_OK) return rc;
    if (F == 0) return 0;
    if (!f || !mag || !ph || !residual) { pvx_set_error("null output array"); return PVX_ERR_INVALID; }
    const size_t xbytes = (size_t)nsamp * dtype_size(x_dtype);
    const size_t fk = (size_t)F * p->npks * sizeof(double), f1 = (size_t)F * sizeof(double);
    HostTrace tr("harmonic");
    DevBuf dout, dprev;
    if ((rc = grow_dev(&p->d_hx, &p->hx_cap, xbytes)) != PVX_OK) return rc;
    struct { void* p; } dx = {p->d_hx};
    if ((rc = dout.alloc(3 * fk + 2 * f1)) != PVX_OK) return rc;
    tr.mark("device buffers");
    if ((rc = host_to_device(dx.p, x, xbytes)) != PVX_OK) return rc;
    tr.mark("signal on the device");
    if (prev0) {
        if ((rc = dprev.alloc(sizeof(double) * 2 * p->N2)) != PVX_OK) return rc;
        PVX_HIP_CHECK(hipMemcpy(dprev.p, prev0, sizeof(double) * 2 * p->N2, hipMemcpyHostToDevice));
    }
    char* o = (char*)dout.p;
    double *d_f = (double*)o, *d_mag = (double*)(o + fk), *d_ph = (double*)(o + 2 * fk), *d_res = (double*)(o + 3 * fk),
           *d_t = (double*)(o + 3 * fk + f1);
    bool any = false;
    p->progress_live = true;
    rc = harmonic_rows(p, dx.p, x_dtype, nsamp, F, f0, fmin, d_f, d_mag, d_ph, d_res, d_t, (const double*)dprev.p, nullptr, &any);
    p->progress_live = false;
    if (rc != PVX_OK) return rc;
    PVX_HIP_CHECK(hipStreamSynchronize(nullptr));
    tr.mark("kernels");
    if ((rc = device_to_host(f, d_f, fk)) != PVX_OK || (rc = device_to_host(mag, d_mag, fk)) != PVX_OK || (rc = device_to_host(ph, d_ph, fk)) != PVX_OK ||
        (rc = device_to_host(residual, d_res, f1)) != PVX_OK) return rc;
    if (t && (rc = device_to_host(t, d_t, f1)) != PVX_OK) return rc;
    tr.mark("results on the host");
    if (last_spec) {
        // oldfft after the loop = spectrum of the last analysed frame (PV.py:491), else unchanged
        if (any) {
            const size_t rs = real_size(p->precision);
            std::vector<unsigned char> tmp((size_t)p->N2 * 2 * rs);
            PVX_HIP_CHECK(hipMemcpy(tmp.data(), p->d_carry, tmp.size(), hipMemcpyDeviceToHost));
            for (int i = 0; i < 2 * p->N2; i++)
                last_spec[i] = p->precision == 32 ? (double)((float*)tmp.data())[i] : ((double*)tmp.data())[i];
        } else {
            for (int i = 0; i < 2 * p->N2; i++) last_spec[i] = prev0 ? prev0[i] : 0.0;
        }
    }
    if (p->progress_fn) p->progress_fn(F, F, p->progress_user);
    return F;
}

// ---- windowed reductions with the analysis framing (Heterodyne.py:35-60, SoundUtils.py:71-103) ----
static int reduce_args(int64_t n, const double* wind, int wlen, int hop, double* norm, int power) {
    if (n < 0 || wlen <= 0 || hop <= 0 || !wind) { pvx_set_error("bad windowed-reduction argument"); return PVX_ERR_INVALID; }
    double s = 0.0;
    for (int i = 0; i < wlen; i++) s += power == 2 ? wind[i] * wind[i] : wind[i];
    *norm = s;
    return PVX_OK;
}

extern "C" int64_t pvx_heterodyne_dev(const double* d_x, const double* d_hetsig, int64_t n, const double* wind, int wlen, int hop,
                                      double* d_out, int64_t* d_icent, void* stream) {
    int rc = pvx_require_device();
    if (rc != PVX_OK) return rc;
    double norm;
    if ((rc = reduce_args(n, wind, wlen, hop, &norm, 1)) != PVX_OK) return rc;
    const int64_t nfr = pvx_nframes(n, wlen, hop);
    if (nfr == 0) return 0;
    if (!d_x || !d_hetsig || !d_out) { pvx_set_error("null heterodyne array"); return PVX_ERR_INVALID; }
    DevBuf dw;
    if ((rc = dw.alloc((size_t)wlen * 8)) != PVX_OK) return rc;
    PVX_HIP_CHECK(hipMemcpy(dw.p, wind, (size_t)wlen * 8, hipMemcpyHostToDevice));
    ReduceParams rp = {};
    rp.x = d_x; rp.hetsig = d_hetsig; rp.wind = (const double*)dw.p; rp.nfr = nfr; rp.wlen = wlen; rp.hop = hop;
    rp.norm = norm; rp.out = d_out; rp.icent = d_icent;
    if ((rc = pvx_launch_reduce(rp, 0, (hipStream_t)stream)) != PVX_OK) return rc;
    PVX_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));            // dw is a local
    return nfr;
}

extern "C" int64_t pvx_heterodyne(const double* x, const double* hetsig, int64_t n, const double* wind, int wlen, int hop,
                                  double* out, int64_t* icent) {
    int rc = pvx_require_device();
    if (rc != PVX_OK) return rc;
    double norm;
    if ((rc = reduce_args(n, wind, wlen, hop, &norm, 1)) != PVX_OK) return rc;
    const int64_t nfr = pvx_nframes(n, wlen, hop);
    if (nfr == 0) return 0;
    if (!x || !hetsig || !out) { pvx_set_error("null heterodyne array"); return PVX_ERR_INVALID; }
    DevBuf dx, dh, dout, dic;
    if ((rc = dx.alloc((size_t)n * 8)) != PVX_OK || (rc = dh.alloc((size_t)n * 16)) != PVX_OK ||
        (rc = dout.alloc((size_t)nfr * 16)) != PVX_OK || (rc = dic.alloc((size_t)nfr * 8)) != PVX_OK) return rc;
    if ((rc = host_to_device(dx.p, x, (size_t)n * 8)) != PVX_OK || (rc = host_to_device(dh.p, hetsig, (size_t)n * 16)) != PVX_OK) return rc;
    const int64_t r = pvx_heterodyne_dev((const double*)dx.p, (const double*)dh.p, n, wind, wlen, hop, (double*)dout.p,
                                         (int64_t*)dic.p, nullptr);
    if (r < 0) return r;
    if ((rc = device_to_host(out, dout.p, (size_t)nfr * 16)) != PVX_OK) return rc;
    if (icent && (rc = device_to_host(icent, dic.p, (size_t)nfr * 8)) != PVX_OK) return rc;
    return nfr;
}

extern "C" int64_t pvx_rms_frames_dev(const double* d_x, int64_t n, const double* wind, int wlen, int hop, double* d_out,
                                      void* stream) {
    int rc = pvx_require_device();
    if (rc != PVX_OK) return rc;
    double norm;
    if ((rc = reduce_args(n, wind, wlen, hop, &norm, 2)) != PVX_OK) return rc;
    const int64_t nfr = pvx_nframes(n, wlen, hop);
    if (nfr == 0) return 0;
    if (!d_x || !d_out) { pvx_set_error("null rms array"); return PVX_ERR_INVALID; }
    DevBuf dw;
    if ((rc = dw.alloc((size_t)wlen * 8)) != PVX_OK) return rc;
    PVX_HIP_CHECK(hipMemcpy(dw.p, wind, (size_t)wlen * 8, hipMemcpyHostToDevice));
    ReduceParams rp = {};
    rp.x = d_x; rp.wind = (const double*)dw.p; rp.nfr = nfr; rp.wlen = wlen; rp.hop = hop; rp.norm = norm; rp.out = d_out;
    if ((rc = pvx_launch_reduce(rp, 1, (hipStream_t)stream)) != PVX_OK) return rc;
    PVX_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return nfr;
}

extern "C" int64_t pvx_rms_frames(const double* x, int64_t n, const double* wind, int wlen, int hop, double* out) {
    int rc = pvx_require_device();
    if (rc != PVX_OK) return rc;
    double norm;
    if ((rc = reduce_args(n, wind, wlen, hop, &norm, 2)) != PVX_OK) return rc;
    const int64_t nfr = pvx_nframes(n, wlen, hop);
    if (nfr == 0) return 0;
    if (!x || !out) { pvx_set_error("null rms array"); return PVX_ERR_INVALID; }
    DevBuf dx, dout;
    if ((rc = dx.alloc((size_t)n * 8)) != PVX_OK || (rc = dout.alloc((size_t)nfr * 8)) != PVX_OK) return rc;
    if ((rc = host_to_device(dx.p, x, (size_t)n * 8)) != PVX_OK) return rc;
    const int64_t r = pvx_rms_frames_dev((const double*)dx.p, n, wind, wlen, hop, (double*)dout.p, nullptr);
    if (r < 0) return r;
    if ((rc = device_to_host(out, dout.p, (size_t)nfr * 8)) != PVX_OK) return rc;
    return nfr;
}

// FuncWind's named reducers (SoundUtils.py:42-69)
static int funcwind_args(int x_complex, int64_t n, const double* wind, int wlen, int hop, int func, double divisor) {
    if (n < 0 || wlen <= 0 || hop <= 0 || !wind) { pvx_set_error("bad windowed-reduction argument"); return PVX_ERR_INVALID; }
    if (func < PVX_FW_SUM || func > PVX_FW_VAR) { pvx_set_error("pvx_funcwind: unknown reducer %d", func); return PVX_ERR_INVALID; }
    if (!(divisor == divisor) || divisor == 0.0) { pvx_set_error("pvx_funcwind: divisor %g", divisor); return PVX_ERR_INVALID; }
    if (x_complex && (func == PVX_FW_MAX || func == PVX_FW_MIN)) { pvx_set_error("pvx_funcwind: max / min of complex frames"); return PVX_ERR_UNSUPPORTED; }
    return PVX_OK;
}

extern "C" int64_t pvx_funcwind_dev(const double* d_x, int x_complex, int64_t n, const double* wind, int wlen, int hop, int func, double divisor,
                                    double* d_out, void* stream) {
    int rc = pvx_require_device();
    if (rc != PVX_OK) return rc;
    if ((rc = funcwind_args(x_complex, n, wind, wlen, hop, func, divisor)) != PVX_OK) return rc;
    const int64_t nfr = pvx_nframes(n, wlen, hop);
    if (nfr == 0) return 0;
    if (!d_x || !d_out) { pvx_set_error("null funcwind array"); return PVX_ERR_INVALID; }
    DevBuf dw;
    if ((rc = dw.alloc((size_t)wlen * 8)) != PVX_OK) return rc;
    PVX_HIP_CHECK(hipMemcpy(dw.p, wind, (size_t)wlen * 8, hipMemcpyHostToDevice));
    ReduceParams rp = {};
    rp.x = d_x; rp.wind = (const double*)dw.p; rp.nfr = nfr; rp.wlen = wlen; rp.hop = hop; rp.norm = divisor; rp.out = d_out;
    if ((rc = pvx_launch_funcwind(rp, func, x_complex != 0, (hipStream_t)stream)) != PVX_OK) return rc;
    PVX_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));            // dw is a local
    return nfr;
}

extern "C" int64_t pvx_funcwind(const double* x, int x_complex, int64_t n, const double* wind, int wlen, int hop, int func, double divisor, double* out) {
    int rc = pvx_require_device();
    if (rc != PVX_OK) return rc;
    if ((rc = funcwind_args(x_complex, n, wind, wlen, hop, func, divisor)) != PVX_OK) return rc;
    const int64_t nfr = pvx_nframes(n, wlen, hop);
    if (nfr == 0) return 0;
    if (!x || !out) { pvx_set_error("null funcwind array"); return PVX_ERR_INVALID; }
    const size_t xb = (size_t)n * (x_complex ? 16 : 8);
    const size_t ob = (size_t)nfr * ((x_complex && (func == PVX_FW_SUM || func == PVX_FW_MEAN)) ? 16 : 8);
    DevBuf dx, dout;
    if ((rc = dx.alloc(xb)) != PVX_OK || (rc = dout.alloc(ob)) != PVX_OK) return rc;
    if ((rc = host_to_device(dx.p, x, xb)) != PVX_OK) return rc;
    const int64_t r = pvx_funcwind_dev((const double*)dx.p, x_complex, n, wind, wlen, hop, func, divisor, (double*)dout.p, nullptr);
    if (r < 0) return r;
    if ((rc = device_to_host(out, dout.p, ob)) != PVX_OK) return rc;
    return nfr;
}

// ---- result wire format for the multi-GPU gather (k_wire.hip) -------------------------------
extern "C" int pvx_plan_set_wire_format(pvx_plan* plan, int format) {
    if (!plan) { pvx_set_error("null plan"); return PVX_ERR_INVALID; }
    if (format != 1 && format != 2) { pvx_set_error("unknown wire format %d", format); return PVX_ERR_INVALID; }
    if (format == 2 && plan->precision != 32) {
        pvx_set_error("wire format 2 (14 bytes per slot) carries the float32 value a precision-32 analysis computes a frequency from; this plan is at precision %d", plan->precision);
        return PVX_ERR_UNSUPPORTED;
    }
    plan->wire_fmt = format;
    return PVX_OK;
}
extern "C" int pvx_plan_get_wire_format(const pvx_plan* plan) {
    if (!plan) { pvx_set_error("null plan"); return PVX_ERR_INVALID; }
    return plan->wire_fmt;
}

extern "C" int64_t pvx_wire_bytes(const pvx_plan* plan, int64_t rows) {
    if (!plan || rows < 0) { pvx_set_error("bad wire argument"); return PVX_ERR_INVALID; }
    if (plan->nfft / 2 > 65536) { pvx_set_error("wire format holds bin numbers in 16 bits (nfft <= 131072)"); return PVX_ERR_UNSUPPORTED; }
    return (int64_t)pvx_wire_block_bytes(rows, plan->npks, plan->precision, plan->wire_fmt);
}

extern "C" int pvx_pack_rows_dev(const pvx_plan* plan, int64_t rows, const double* d_f, const double* d_mag,
                                 const double* d_ph, const double* d_binno, const double* d_totalmag, void* d_wire,
                                 void* stream) {
    int rc = pvx_require_plan_device(plan);
    if (rc != PVX_OK) return rc;
    if (pvx_wire_bytes(plan, rows) < 0) return PVX_ERR_INVALID;
    if (rows == 0) return PVX_OK;
    if (!d_f || !d_mag || !d_ph || !d_binno || !d_totalmag || !d_wire) { pvx_set_error("null wire array"); return PVX_ERR_INVALID; }
    WireParams wp = {};
    wp.rows = rows; wp.K = plan->npks; wp.precision = plan->precision; wp.fstep = plan->fstep; wp.wire = d_wire;
    wp.fmt = plan->wire_fmt; wp.dt = plan->dt; wp.wfbin = plan->d_wfbin;
    wp.f = d_f; wp.mag = d_mag; wp.ph = d_ph; wp.binno = d_binno; wp.totalmag = d_totalmag;
    return pvx_launch_wire(wp, true, (hipStream_t)stream);
}

// run_pv straight into the wire format: what a rank of a multi-GPU job hands to the gather.  k_fused_rev (precision 32, nfft 512 ..
// 2048) writes the block itself; every other plan analyses into a plan-owned result block and packs it.
extern "C" int64_t pvx_analyze_dev_wire(pvx_plan* p, const void* d_x, int x_dtype, int64_t nsamp, int64_t nsig, int64_t sig_stride,
                                        void* d_wire, void* stream) {
    int rc = pvx_require_plan_device(p);
    if (rc != PVX_OK) return rc;
    rc = check_analyze_args(p, d_x, x_dtype, nsamp, nsig, sig_stride, nullptr);
    if (rc != PVX_OK) return rc;
    const int64_t F = pvx_nframes(nsamp, p->nfft, p->hop);
    if (F == 0) return 0;
    if (!d_wire) { pvx_set_error("null wire block"); return PVX_ERR_INVALID; }
    if (pvx_wire_bytes(p, nsig * F) < 0) return PVX_ERR_INVALID;
    const int64_t rows = nsig * F;
    const size_t n = (size_t)rows * (size_t)p->npks, ts = p->precision == 64 ? 8 : 4;
    auto al8 = [](size_t v) { return (v + 7) & ~(size_t)7; };
    unsigned char* w = (unsigned char*)d_wire;
    if (p->fft_mode == 4 && p->precision == 32 && getenv("PVX_NO_WIRE_OUT") == nullptr) {
        // the sections of the block (k_wire.hip) as the kernel's output arrays
        const size_t fw = p->wire_fmt == 2 ? 4 : 8;                   // (format 2: the float32 the frequency is computed from)
        double* wf = (double*)w;
        double* wm = (double*)(w + al8(n * fw));
        double* wp = (double*)(w + al8(n * fw) + al8(n * ts));
        double* wb = (double*)(w + al8(n * fw) + 2 * al8(n * ts));
        double* wt = (double*)(w + al8(n * fw) + 2 * al8(n * ts) + al8(n * 2));
        rc = analyze_rows(p, d_x, x_dtype, nsamp, nsig, sig_stride, F, wf, wm, wp, nullptr, wb, nullptr, wt, nullptr, (hipStream_t)stream, -1, true);
        return rc == PVX_OK ? F : rc;
    }
    const size_t per_frame_out = (size_t)(5 * p->npks + 2) * sizeof(double);
    if ((rc = grow_dev(&p->d_wiretmp, &p->wiretmp_cap, (size_t)rows * per_frame_out)) != PVX_OK) return rc;
    const HostOut o = block_ptrs(p->d_wiretmp, rows, p->npks);
    rc = analyze_rows(p, d_x, x_dtype, nsamp, nsig, sig_stride, F, o.f, o.mag, o.ph, o.realph, o.binno, nullptr, o.totalmag, nullptr, (hipStream_t)stream);
    if (rc != PVX_OK) return rc;
    rc = pvx_pack_rows_dev(p, rows, o.f, o.mag, o.ph, o.binno, o.totalmag, d_wire, stream);
    return rc == PVX_OK ? F : rc;
}

extern "C" int pvx_unpack_rows_dev(const pvx_plan* plan, int64_t rows, const void* d_wire, double* d_f, double* d_mag,
                                   double* d_ph, double* d_realph, double* d_binno, double* d_totalmag, void* stream) {
    int rc = pvx_require_plan_device(plan);
    if (rc != PVX_OK) return rc;
    if (pvx_wire_bytes(plan, rows) < 0) return PVX_ERR_INVALID;
    if (rows == 0) return PVX_OK;
    if (!d_f || !d_mag || !d_ph || !d_realph || !d_binno || !d_totalmag || !d_wire) { pvx_set_error("null wire array"); return PVX_ERR_INVALID; }
    WireParams wp = {};
    wp.rows = rows; wp.K = plan->npks; wp.precision = plan->precision; wp.fstep = plan->fstep; wp.wire = (void*)d_wire;
    wp.fmt = plan->wire_fmt; wp.dt = plan->dt; wp.wfbin = plan->d_wfbin;
    wp.of = d_f; wp.omag = d_mag; wp.oph = d_ph; wp.orealph = d_realph; wp.obinno = d_binno; wp.ototalmag = d_totalmag;
    return pvx_launch_wire(wp, false, (hipStream_t)stream);
}

// ---- resynthesis: SinSum.synth (PV.py:1053-1070) -------------------------------------------
extern "C" int64_t pvx_synth_len(int64_t max_end_frame, int nfft, int hop_analysis, int hop_synth, double edge) {
    if (nfft <= 0 || hop_analysis <= 0 || hop_synth <= 0 || max_end_frame < 0) return PVX_ERR_INVALID;
    const double dfr = (double)nfft / (double)hop_analysis / 2.;      // PV.py:1055
    const int64_t edgsamp = (int64_t)(edge * hop_synth * dfr);        // PV.py:1056, integer as under Python 2
    return (max_end_frame + 2) * (int64_t)hop_synth + 2 * edgsamp - edgsamp;   // len(w[edgsamp:]), PV.py:1059, 1070
}

extern "C" int pvx_synth_dev_flags(const double* d_f, const double* d_mag, const double* d_realph, const int32_t* d_partial_id,
                             int64_t F, int K, const int32_t* d_part_start, const int32_t* d_part_len, int64_t P,
                             double sr, int nfft, int hop_analysis, int hop_synth, double edge, int minframes,
                             double* d_w, int64_t wlen, void* stream, int flags) {
    int rc = pvx_require_device();
    if (rc != PVX_OK) return rc;
    if (F <= 0 || K <= 0 || P <= 0 || nfft <= 0 || hop_analysis <= 0 || hop_synth <= 0 || !(sr > 0) || wlen <= 0 || !(edge >= 0)) {
        pvx_set_error("bad resynthesis argument");
        return PVX_ERR_INVALID;
    }
    if (!d_f || !d_mag || !d_realph || !d_partial_id || !d_part_start || !d_part_len || !d_w) { pvx_set_error("null resynthesis array"); return PVX_ERR_INVALID; }
    SynthParams sp;
    sp.f = d_f; sp.mag = d_mag; sp.realph = d_realph; sp.partial_id = d_partial_id;
    sp.part_start = d_part_start; sp.part_len = d_part_len; sp.F = F; sp.P = P; sp.K = K;
    sp.sr = sr; sp.edge = edge; sp.nfft = nfft; sp.hop_a = hop_analysis; sp.hop_s = hop_synth; sp.minframes = minframes;
    sp.w = d_w; sp.wlen = wlen; sp.no_phcor = (flags & PVX_SYNTH_NO_PHCOR) ? 1 : 0;
    sp.f32_samples = (flags & PVX_SYNTH_F32) ? 1 : 0;
    return pvx_launch_synth(sp, (hipStream_t)stream);
}

// one slice of the waveform's segments (internal: pvx_synth_resident overlaps the slices' kernels with their copies)
static int synth_slice(const double* d_f, const double* d_mag, const double* d_realph, const int32_t* d_partial_id, int64_t F, int K,
                       const int32_t* d_part_start, const int32_t* d_part_len, int64_t P, double sr, int nfft, int hop_analysis, int hop_synth,
                       double edge, int minframes, double* d_w, int64_t wlen, hipStream_t stream, int64_t seg0, int64_t seg_count, bool first,
                       void* ws, size_t ws_bytes, unsigned* ws_gen, int f32_samples) {
    SynthParams sp;
    sp.f32_samples = f32_samples;
    sp.skip_prepare = first ? 0 : 1;          // the partial-major copy of the analysis arrays is made with the first slice
    sp.ws = ws; sp.ws_bytes = ws_bytes; sp.ws_gen = ws_gen;
    sp.f = d_f; sp.mag = d_mag; sp.realph = d_realph; sp.partial_id = d_partial_id;
    sp.part_start = d_part_start; sp.part_len = d_part_len; sp.F = F; sp.P = P; sp.K = K;
    sp.sr = sr; sp.edge = edge; sp.nfft = nfft; sp.hop_a = hop_analysis; sp.hop_s = hop_synth; sp.minframes = minframes;
    sp.w = d_w; sp.wlen = wlen; sp.no_phcor = 0; sp.seg0 = seg0; sp.seg_count = seg_count;
    return pvx_launch_synth(sp, stream);
}

static int plan_synth_f32(const pvx_plan* p) {
    const int f32 = (p->precision == 32 && getenv("PVX_SYNTH_F64") == nullptr) ? 1 : 0;
    const_cast<pvx_plan*>(p)->last_synth = f32 ? "k_synth_bodies<f32>" : "k_synth_bodies<f64>";
    return f32;
}

extern "C" int pvx_synth_dev(const double* d_f, const double* d_mag, const double* d_realph, const int32_t* d_partial_id,
                             int64_t F, int K, const int32_t* d_part_start, const int32_t* d_part_len, int64_t P,
                             double sr, int nfft, int hop_analysis, int hop_synth, double edge, int minframes,
                             double* d_w, int64_t wlen, void* stream) {
    return pvx_synth_dev_flags(d_f, d_mag, d_realph, d_partial_id, F, K, d_part_start, d_part_len, P, sr, nfft, hop_analysis,
                               hop_synth, edge, minframes, d_w, wlen, stream, 0);
}

extern "C" int pvx_synth_flags(const double* f, const double* mag, const double* realph, const int32_t* partial_id, int64_t F,
                         int K, const int32_t* part_start, const int32_t* part_len, int64_t P, double sr, int nfft,
                         int hop_analysis, int hop_synth, double edge, int minframes, double* w, int64_t wlen, int flags) {
    int rc = pvx_require_device();
    if (rc != PVX_OK) return rc;
    if (F <= 0 || K <= 0 || P <= 0 || !f || !mag || !realph || !partial_id || !part_start || !part_len || !w) {
        pvx_set_error("bad resynthesis argument");
        return PVX_ERR_INVALID;
    }
    int64_t maxend = 0;                                               // max(self.end), PV.py:1059
    for (int64_t i = 0; i < P; i++) { const int64_t e = (int64_t)part_start[i] + part_len[i] - 1; if (e > maxend) maxend = e; }
    const int64_t need = pvx_synth_len(maxend, nfft, hop_analysis, hop_synth, edge);
    if (need < 0 || need != wlen) { pvx_set_error("output length %lld, expected %lld", (long long)wlen, (long long)need); return PVX_ERR_SIZE; }
    const size_t n = (size_t)F * K;
    DevBuf df, dm, dr, dpid, dst, dln, dw;
    if ((rc = df.alloc(n * 8)) != PVX_OK || (rc = dm.alloc(n * 8)) != PVX_OK || (rc = dr.alloc(n * 8)) != PVX_OK ||
        (rc = dpid.alloc(n * 4)) != PVX_OK || (rc = dst.alloc((size_t)P * 4)) != PVX_OK ||
        (rc = dln.alloc((size_t)P * 4)) != PVX_OK || (rc = dw.alloc((size_t)wlen * 8)) != PVX_OK)
        return rc;
    if ((rc = host_to_device(df.p, f, n * 8)) != PVX_OK || (rc = host_to_device(dm.p, mag, n * 8)) != PVX_OK || (rc = host_to_device(dr.p, realph, n * 8)) != PVX_OK ||
        (rc = host_to_device(dpid.p, partial_id, n * 4)) != PVX_OK || (rc = host_to_device(dst.p, part_start, (size_t)P * 4)) != PVX_OK ||
        (rc = host_to_device(dln.p, part_len, (size_t)P * 4)) != PVX_OK) return rc;
    rc = pvx_synth_dev_flags((const double*)df.p, (const double*)dm.p, (const double*)dr.p, (const int32_t*)dpid.p, F, K,
                             (const int32_t*)dst.p, (const int32_t*)dln.p, P, sr, nfft, hop_analysis, hop_synth, edge, minframes,
                             (double*)dw.p, wlen, nullptr, flags);
    if (rc != PVX_OK) return rc;
    PVX_HIP_CHECK(hipStreamSynchronize(nullptr));
    if ((rc = device_to_host(w, dw.p, (size_t)wlen * 8)) != PVX_OK) return rc;
    return PVX_OK;
}

extern "C" int pvx_synth(const double* f, const double* mag, const double* realph, const int32_t* partial_id, int64_t F,
                         int K, const int32_t* part_start, const int32_t* part_len, int64_t P, double sr, int nfft,
                         int hop_analysis, int hop_synth, double edge, int minframes, double* w, int64_t wlen) {
    return pvx_synth_flags(f, mag, realph, partial_id, F, K, part_start, part_len, P, sr, nfft, hop_analysis, hop_synth, edge,
                           minframes, w, wlen, 0);
}
