// k_witness_stub.hip -- the product library's stand-in for the witness kernels.
//
// k_fused.hip (fft mode 1: one wave per frame over two spectrum buffers) and k_fused_ring.hip (fft mode 3: a workgroup's
// waves over a shared ring of spectra) are the independently written predecessors of k_fused_rev.hip.  They are kept as
// witnesses for the bit-identity tests (modes 1 = 3 = 4 at nfft 1024 / 512, 1 = 3 at 2048) and are built into
// tests/libpvx_witness.so only (`make -C pypevoc_amd/csrc witness`): libpvx_hip.so, the product, carries fft modes 0, 4
// and 5.  Here their entry points say so.
// k_fused_mw.hip (fft mode 2: several waves per frame, the first fused shape of nfft 4096 / 8192) took npks > 128 there until round 6;
// the general path (k_stft_split + k_phase_peaks) is within 2 % of it at nfft 2048 / 4096 and 20 % behind at 8192
// (profiles/r06_ab_steps.txt), so the product has one float32 shape per size class and the kernel is a witness of the general
// path's results at those npks.
// k_pv_team.hip (float64 at nfft 4096 / 8192 as ONE launch: a team of waves per frame, rows walked downwards, the row on chip) is a
// witness of another kind: built, bit-identical to the two-kernel path -- and slower than it (profiles/r06_ab_steps.txt), so the
// product keeps k_stft_split + k_phase_peaks there and the kernel lives in the witness library with its test.
#include "pvx_internal.h"

int pvx_fused_supported(int, int, int) { return 0; }
int pvx_fused_ring_supported(int, int, int) { return 0; }
int pvx_fused_mw_supported(int, int, int) { return 0; }

static int not_here(int mode) {
    pvx_set_error("fft mode %d is a witness kernel: it is built into tests/libpvx_witness.so (make -C pypevoc_amd/csrc witness), not into libpvx_hip.so", mode);
    return PVX_ERR_UNSUPPORTED;
}
int pvx_launch_fused(const FusedParams&, int, int, hipStream_t) { return not_here(1); }
int pvx_launch_fused_ring(const FusedParams&, int, int, hipStream_t) { return not_here(3); }
int pvx_launch_fused_mw(const FusedParams&, int, int, hipStream_t) { return not_here(2); }

int pvx_pv_team_supported(int, int, int, int) { return 0; }
size_t pvx_pv_team_stage_bytes(int) { return 0; }
int pvx_launch_pv_team(const PvRevParams&, int, int, hipStream_t) {
    pvx_set_error("k_pv_team is a witness kernel: it is built into tests/libpvx_witness.so (make -C pypevoc_amd/csrc witness), not into libpvx_hip.so");
    return PVX_ERR_UNSUPPORTED;
}
