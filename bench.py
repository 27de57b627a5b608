#!/usr/bin/env python3
"""bench.py -- STFT frames/s of the phase-vocoder analysis stage (PV.run_pv) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c4]

N > 1 without a torch.distributed environment: bench.py launches itself as N ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py ...`, as a child process, before anything
touches the GPU) and relays rank 0's JSON line and exit code.

Metric (BASELINE.json): STFT frames/sec at 44.1 kHz, nfft=2048, hop=512, npks=8.
A "step" is one full run_pv pass over the workload, with the signal(s) already resident in HBM:
  c2 (default): BASELINE config 2 -- one 10-minute 44.1 kHz mono signal per GPU (26 460 000 samples,
                F = 51 676 frames); N > 1 = weak scaling, one such signal per GPU.
  c4          : BASELINE config 4's per-GPU shard -- 128 independent 30-s 48 kHz signals per GPU in ONE call
                (F = 2 809 frames each; 1 024 signals over 8 GPUs), the unit north_star shards on.
  N > 1: plus the single result gather to rank 0 over RCCL/xGMI inside the step (asynchronous, double-buffered:
  it overlaps the next step's kernels; all gathers have completed and been unpacked before the clock stops).
value = frames processed by all ranks / wall time of the K timed steps (max over ranks).

Also on the same JSON line (rank 0):
  roofline     -- the dominant kernel.  bound "hbm": achieved = ITS algorithmic bytes per launch (DESIGN.md 3:
                  the fused kernels read hop*4 B and write (5*npks+2)*8 B per frame; the three-kernel path's
                  kernels have their own figures) / its mean launch duration, measured with HIP events that
                  libpvx_hip records on the launch stream; `traffic` = HBM bytes per launch measured with
                  rocprofv3 PMC (profiles/traffic_latest.json).  `issue`: the bound the fused kernels are
                  actually up against -- vector-instruction issue and LDS occupancy from the SQ counters in
                  profiles/r02_fused_sq.json, priced with the measured per-instruction costs
                  (profiles/r02_ubench_valu_issue_cost.txt).  `throughput_vs_60pct_target`: frames/s against
                  north_star's target (60 % of 8 TB/s at SURVEY.md 8(d)'s 35 168 B/frame = 1.365e8 frames/s).
  self_check   -- the timed output of the last step, all frames, against the oracle on the same signal
                  (N = 1, c2): aborts with a non-zero exit code when it is out of the stated float32 tolerances.
  f64          -- the same workload at the reference's own precision (float64 end to end, PVAnalysis.py:155-157).
  workloads    -- the same geometry on 10 min of white noise and on a tiled violin recording (fixture G7):
                  the headline signal is the best case for the peak search.
  chain        -- the rest of the path on the headline step's results, device to device: tracker (pvx_track_dev) and
                  resynthesis (pvx_synth_dev) as frames/s, both checked against the oracle on the same arrays.
  cpu_baseline -- the oracle (C port of the reference algorithm) timed on this host: all cores (threads over
                  frame ranges) and one thread; the Python reference's own figure from BASELINE.md beside it.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR, NFFT, HOP, NPKS = 44100, 2048, 512, 8
HBM_PEAK = 8.0e12                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TARGET_FPS = 0.6 * HBM_PEAK / 35168.0   # north_star: 60 % of the roofline at SURVEY.md 8(d)'s contract bytes
WORKLOADS = {"c2": dict(sr=44100, seconds=600, nsig=1), "c4": dict(sr=48000, seconds=30, nsig=128)}


def c2_signal(seconds=600, sr=44100, seed=1234, f0=220.0):
    """SURVEY.md 8(d) C2 / G4 generator: 8 harmonics, 1 % / 5 Hz vibrato, amplitudes 0.3/h, noise."""
    n = int(sr * seconds)
    t = np.arange(n, dtype=np.float64) / sr
    ph = 2 * np.pi * f0 * (t - 0.01 / (2 * np.pi * 5.0) * np.cos(2 * np.pi * 5.0 * t))
    x = np.zeros(n)
    for h in range(1, 9):
        x += 0.3 / h * np.sin(h * ph)
    x += 0.001 * np.random.default_rng(seed).standard_normal(n)
    return x.astype(np.float32)


def c4_shard(torch, dev, rank, nsig=128, seconds=30, sr=48000):
    """signal b = the G4 generator with f0 = 110 * 2**(3 b / 1024); this rank's 128 signals,
    generated on the device."""
    n = sr * seconds
    t = torch.arange(n, dtype=torch.float64, device=dev) / sr
    b = torch.arange(rank * nsig, (rank + 1) * nsig, dtype=torch.float64, device=dev)
    f0 = 110.0 * torch.pow(torch.tensor(2.0, dtype=torch.float64, device=dev), 3.0 * b / 1024.0)
    x = torch.empty((nsig, n), dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev)
    vib = t - 0.01 / (2 * np.pi * 5.0) * torch.cos(2 * np.pi * 5.0 * t)
    for i in range(nsig):
        ph = 2 * np.pi * f0[i] * vib
        acc = torch.zeros(n, dtype=torch.float64, device=dev)
        for h in range(1, 9):
            acc += 0.3 / h * torch.sin(h * ph)
        g.manual_seed(1234 + rank * nsig + i)
        acc += 0.001 * torch.randn(n, dtype=torch.float64, device=dev, generator=g)
        x[i] = acc.to(torch.float32)
    return x


def alg_bytes(nfft=NFFT, hop=HOP, npks=NPKS, s=4, c=8):
    """Algorithmic bytes per frame of each kernel.  `contract` = SURVEY.md 8(d)'s figure for the
    three-kernel STFT+phase stage; `fused` = what a fused kernel has to move: the hop's new samples in, the
    reference's result rows out."""
    out = npks * 5 * 8 + 16
    return dict(frames=hop * s + nfft * s, fft=nfft * s + (nfft // 2 + 1) * c, peaks=(nfft // 2 + 1) * c + out,
                fused=hop * s + out, contract=hop * s + 2 * nfft * s + 2 * (nfft // 2 + 1) * c + out)


def csrc_sha16():
    """Fingerprint of the kernel sources this library was built from (profiles taken with other sources are not quoted):
    what the loaded library says about itself -- pypevoc_amd/_lib.py has refused it if the sources beside it differ."""
    from pypevoc_amd import _lib
    try:
        return _lib.load().pvx_build_fingerprint().decode()
    except Exception:
        return _lib.source_fingerprint()


def committed_traffic(kernel):
    """HBM bytes per C2 launch of `kernel` from profiles/traffic_latest.json (rocprofv3 PMC, tools/prof_traffic.sh) -- a
    COMMITTED profile, not a measurement of this run: it is only quoted when it was taken with the kernel sources of
    this build.  Returns (bytes or None, provenance dict)."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        ent = json.load(open(path)).get(kernel)
    except Exception:
        ent = None
    if not isinstance(ent, dict):
        return None, dict(source="profiles/traffic_latest.json", note="no entry for %s" % kernel)
    prov = dict(source="profiles/traffic_latest.json (committed rocprofv3 PMC profile, not measured in this run)",
                symbol=ent.get("symbol"), profile_csrc_sha16=ent.get("csrc_sha16"), build_csrc_sha16=csrc_sha16(), tag=ent.get("tag"))
    if ent.get("csrc_sha16") != prov["build_csrc_sha16"]:
        prov["note"] = "profile predates the kernel sources of this build: not quoted"
        return None, prov
    return int(ent["bytes"]), prov


def free_port():
    """A rendezvous port nobody is listening on (bound to port 0 and released)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(args, argv):
    """--gpus N > 1 outside a torch.distributed environment: run N ranks as a child process."""
    import torch
    have = torch.cuda.device_count()            # does not initialise the GPU
    if have < args.gpus:
        sys.stderr.write("bench.py --gpus %d: this node shows %d GPU(s)\n" % (args.gpus, have))
        return 2
    port = os.environ.get("MASTER_PORT") or str(free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, universal_newlines=True)
    got_line = False
    for ln in p.stdout.splitlines():            # rank 0 prints the one JSON line; pass anything else to stderr
        got_line = got_line or ln.startswith("{")
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    if p.returncode != 0:                       # any rank failing fails the launcher (torch.distributed.run's exit code)
        sys.stderr.write("bench.py --gpus %d: the %d-rank child run exited with code %d\n" % (args.gpus, args.gpus, p.returncode))
        return p.returncode
    if not got_line:
        sys.stderr.write("bench.py --gpus %d: rank 0 printed no result line\n" % args.gpus)
        return 4
    return 0


def issue_bound(kernel_name, frames_per_launch, ms_per_launch):
    """Instruction-issue / LDS view of a fused kernel from the COMMITTED SQ counters (profiles/sq_latest.json: rocprofv3
    --pmc over tools/run_mode.py, per-launch means) scaled to this run's launch duration; only quoted when the profile
    was taken with the kernel sources of this build."""
    path = os.path.join(ROOT, "profiles", "sq_latest.json")
    try:
        prof = json.load(open(path))
    except Exception:
        return None
    if prof.get("_csrc_sha16") != csrc_sha16():
        return dict(kernel=kernel_name, source="profiles/sq_latest.json", note="profile predates the kernel sources of this build: not quoted",
                    profile_csrc_sha16=prof.get("_csrc_sha16"), build_csrc_sha16=csrc_sha16())
    ent = None
    for k, v in prof.items():
        if kernel_name in k and isinstance(v, dict):
            ent = v
    if not ent:
        return None
    c = {k: v["mean"] for k, v in ent.items()}
    scale = frames_per_launch / float(prof.get("_frames_per_launch", frames_per_launch))
    simd_cycles = 1024 * 2.4e9 * ms_per_launch * 1e-3                  # 256 CUs x 4 SIMDs at the 2.4 GHz peak clock
    valu = c.get("SQ_INSTS_VALU", 0.0) * scale
    # a wave64 vector instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md); packed-f32, DPP and
    # f64 instructions for ~4.4 (measured, profiles/r02_ubench_valu_issue_cost.txt)
    slow = (c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0)
            + c.get("SQ_INSTS_VALU_ADD_F32", 0) + c.get("SQ_INSTS_VALU_FMA_F32", 0) + c.get("SQ_INSTS_VALU_MUL_F32", 0)) * scale
    out = dict(kernel=kernel_name, source="profiles/sq_latest.json (committed rocprofv3 --pmc profile of this build's sources, scaled to this run's launch duration)",
               valu_insts_per_launch=int(valu), valu_insts_per_frame=round(valu / frames_per_launch, 1),
               valu_issue=dict(bound="valu_issue", unit="wave-instructions/s", achieved=valu / (ms_per_launch * 1e-3),
                               peak=1024 * 2.4e9 / 2.0, frac=round(valu * 2.0 / simd_cycles, 4),
                               frac_with_measured_costs=round((valu * 2.0 + slow * 2.4) / simd_cycles, 4),
                               note="frac: every instruction at 2 cycles per SIMD; frac_with_measured_costs: f32 "
                                    "arithmetic (packed in this kernel) and f64 at the measured 4.4 cycles"),
               lds_busy_frac=round(c.get("SQ_LDS_IDX_ACTIVE", 0) / max(c.get("SQ_BUSY_CU_CYCLES", 1), 1), 4),
               wave_time_split=dict(issuing=round(c.get("SQ_ACTIVE_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 3),
                                    waiting_on_memory_or_lds=round(c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 3),
                                    issue_stalled=round(c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 3)))
    return out


def cpu_baseline(x_host, sr, o_full, dt_single):
    """The oracle on this host: one thread (already timed on the full signal: o_full, dt_single) and all cores
    (threads over frame ranges; ctypes releases the GIL inside the C call)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pvoracle
    ncpu = os.cpu_count() or 1
    F = len(o_full["t"])
    nthr = max(1, min(ncpu, F // 64))
    xs = x_host.astype(np.float64)                                  # outside the clock, like the GPU side's resident input
    bounds = [F * i // nthr for i in range(nthr + 1)]

    def work(i):
        f0, f1 = bounds[i], bounds[i + 1]
        seg = xs[f0 * HOP: (f1 - 1) * HOP + NFFT + 1]               # a view: no copy
        return len(pvoracle.analyze(seg, sr, NFFT, HOP, NPKS)["t"])
    with ThreadPoolExecutor(nthr) as ex:
        list(ex.map(lambda i: pvoracle.analyze(xs[: NFFT + HOP + 1], sr, NFFT, HOP, NPKS), range(nthr)))   # threads started, library paged in
        t0 = time.perf_counter()
        done = 0
        for _ in range(3):                                          # three passes: ~0.1 s each on a 256-thread host
            done += sum(ex.map(work, range(nthr)))
        dt_all = time.perf_counter() - t0
    return dict(value=round(done / dt_all, 1), unit="frames/s", cores=nthr, kind="port",
                sample="the full N=1 workload three times (%d frames): oracle/pvoracle.c, a warm pool of %d threads over "
                       "contiguous frame ranges, float64 input prepared beforehand, %.2f s wall; host shows %d cores" % (done, nthr, dt_all, ncpu),
                single_thread=dict(value=round(F / dt_single, 1), cores=1, seconds=round(dt_single, 2)),
                reference_python=dict(value=3518.0, unit="frames/s", cores=1,
                                      host="Intel Xeon @ 2.10 GHz (build container; the Python reference does not "
                                           "travel to the GPU box)", source="BASELINE.md: PV.run_pv, 60 s slice"))

COMPACT_MAX = 1800      # bytes: the driver's record keeps a short tail of stdout; round 4's 20.8 KB line was not parsed


def compact_line(full, detail_path):
    """The ONE stdout line: the contract keys, `roofline`, `cpu_baseline`, `self_check` and scalars only for the extras.
    Everything else (`config5.points`, `workloads`, `other_nfft`, `issue`, notes) is in the detail file."""
    def g(d, *ks):
        for k in ks:
            d = d.get(k) if isinstance(d, dict) else None
        return d
    r = full.get("roofline") or {}
    c = full.get("cpu_baseline")
    sc = full.get("self_check")
    cfg = full["config"]
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    out["value_from_idle"] = full.get("value_from_idle")
    out["value_is"] = "after %g ms of untimed clock ramp" % g(full, "clock_warmup", "ms") if g(full, "clock_warmup", "ms") else "from idle"
    out["config"] = {"workload": cfg["workload_short"], "nfft": cfg["nfft"], "hop": cfg["hop"], "npks": cfg["npks"], "sr": cfg["sr"],
                     "frames_per_gpu": cfg["frames_per_gpu"], "signals_per_gpu": cfg["signals_per_gpu"], "parallelism": cfg["parallelism_short"]}
    out["roofline"] = ({k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic",
                                              "ms_per_launch", "alg_bytes_per_frame", "throughput_vs_60pct_target")} if r else None)
    if r and g(r, "issue", "valu_insts_per_frame"):
        out["roofline"]["valu_per_frame"] = g(r, "issue", "valu_insts_per_frame")
    out["cpu_baseline"] = (dict(value=c["value"], unit=c["unit"], cores=c["cores"], kind=c["kind"], single_thread=g(c, "single_thread", "value"),
                                sample="oracle/pvoracle.c, the N=1 workload x3, %d threads over frame ranges" % c["cores"]) if c else None)
    out["self_check"] = (dict(ok=sc["ok"], frames=sc["frames"], bad_peaks=sc["bad_peaks"], f_abs_Hz=float("%.3g" % sc["f_abs_Hz"])) if sc else None)
    out["per_rank_ms_per_step"] = full.get("per_rank_ms_per_step")
    if full.get("f64"):
        out["f64_value"] = g(full, "f64", "value")
        out["f64_ok"] = g(full, "f64", "self_check", "ok")
        out["f64_frac"] = g(full, "f64", "roofline", "frac")
        out["f64_traffic_over_algorithmic"] = g(full, "f64", "roofline", "traffic_over_algorithmic")
    if full.get("workloads"):
        out["noise_value"] = g(full, "workloads", "white_noise", "value")
        out["violin_value"] = g(full, "workloads", "violin_g7_tiled", "value")
        out["defaults_f32_value"] = g(full, "workloads", "reference_defaults", "f32", "value")
        out["defaults_f64_value"] = g(full, "workloads", "reference_defaults", "f64", "value")
    if full.get("other_nfft"):
        out["nfft4096_value"] = g(full, "other_nfft", "4096", "value")
        out["nfft8192_value"] = g(full, "other_nfft", "8192", "value")
    if full.get("chain"):
        out["chain_total_ms"] = g(full, "chain", "total_ms")
        out["chain_tracker_ms"] = g(full, "chain", "tracker", "ms")
        out["chain_resynthesis_ms"] = g(full, "chain", "resynthesis", "ms")
    pts = g(full, "config5", "points")
    if pts:
        out["c5_min_vs_target"] = min(p["vs_contract_target"] for p in pts)
        out["c5_all_ok"] = all(g(p, "self_check", "ok") is not False for p in pts)
        out["c5_2048_512_value"] = next((p["value"] for p in pts if (p["nfft"], p["hop"]) == (2048, 512)), None)
    if full.get("host_batch"):
        out["host_batch_value"] = g(full, "host_batch", "value")
    if "extras_ok" in full:
        out["extras_ok"] = full["extras_ok"]
    if full.get("gather"):
        out["gather"] = {k: full["gather"].get(k) for k in ("collective", "rccl_world", "wire_bytes_per_rank", "wire_bytes_per_slot", "result_bytes_per_rank", "exposed_ms_per_step", "host_issue_ms_per_step",
                                                            "valid_peaks_gathered", "checked_signals", "check_ok") if k in full["gather"]}
    out["detail"] = detail_path
    s = json.dumps(out, separators=(",", ":"))
    if len(s) > COMPACT_MAX:                       # never again a line the driver cannot parse: drop the optional scalars first
        for k in ("per_rank_ms_per_step", "host_batch_value", "violin_value", "nfft4096_value", "nfft8192_value", "chain_tracker_ms",
                  "chain_resynthesis_ms", "c5_2048_512_value", "f64_traffic_over_algorithmic", "f64_frac", "value_is"):
            if len(s) <= COMPACT_MAX:
                break
            if k == "per_rank_ms_per_step" and out["n_gpus"] > 1:
                continue
            out.pop(k, None)
            s = json.dumps(out, separators=(",", ":"))
    assert len(s) <= COMPACT_MAX, "bench.py: the stdout line is %d bytes" % len(s)
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--clock-warmup-ms", type=float, default=300.0,
                    help="untimed passes of the step for this long before the W warm-up steps (0: none): brings the card from idle to its running clocks")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", type=int, default=32, choices=(32, 64))
    ap.add_argument("--workload", default="c2", choices=tuple(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the f64 / noise / violin lines (N = 1 only anyway)")
    ap.add_argument("--no-host-batch", action="store_true", help="skip the pvx_batch_run line (host signals over a device list; N = 1 only)")
    ap.add_argument("--no-config5", action="store_true", help="skip the `config5` object (BASELINE config 5: the nfft x hop sweep at 96 kHz)")
    ap.add_argument("--c5-seconds", type=int, default=3600, help="length of config 5's 96 kHz signal (BASELINE: 60 min = 1.38 GB of float32 in HBM)")
    ap.add_argument("--fft-mode", type=int, default=-1, help="-1: plan default; 0: general path; 1 ... 5: fused kernels")
    ap.add_argument("--streams", type=int, default=1, choices=(1, 2),
                    help="2: consecutive steps alternate between two streams (hides the launch gap and the kernel tail; "
                         "per-kernel durations then overlap and no longer compare with rocprofv3's)")
    ap.add_argument("--detail", default=os.path.join("gpurun_out", "bench_detail.json"),
                    help="where the full record goes (config5 points, workloads, other_nfft, issue view, every oracle check); stdout carries one compact line")
    ap.add_argument("--seconds", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--shard-signals", type=int, default=0, help=argparse.SUPPRESS)    # tests: fewer signals per GPU than the workload's
    ap.add_argument("--comm-cus", type=int, default=0, help="gathered runs: compute units left without an analysis workgroup, for the collective's kernels (0: none)")
    ap.add_argument("--unpack-in-step", action="store_true", help="rank 0 unpacks every gathered block inside the step (default: the blocks stay in wire format)")
    ap.add_argument("--check-gathered", type=int, default=0,
                    help="with a gather (N > 1, or PVX_BENCH_FORCE_GATHER=1): compare the gathered + unpacked results of rank 0's first "
                         "that many signals with the oracle (non-zero exit code on a miss)")
    args = ap.parse_args()

    # PVX_BENCH_SELF_LAUNCH=1: take the launcher path even for one rank (tests: what `--gpus 8` does on an 8-GPU node)
    if (args.gpus > 1 or os.environ.get("PVX_BENCH_SELF_LAUNCH") == "1") and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))

    # (the pool's driver only supports dmabuf IPC: RCCL and cross-process tensor sharing need this before the runtime starts)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py --gpus %d inside a torch.distributed environment of %d ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        sys.exit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from pypevoc_amd import _lib
    lib = _lib.load()
    _lib.init(local_rank)

    # ---- workload, resident in HBM before the timed region
    wl = dict(WORKLOADS[args.workload])
    if args.seconds:
        wl["seconds"] = args.seconds
    if args.shard_signals:
        wl["nsig"] = args.shard_signals
    sr = wl["sr"]
    x_host = None
    if args.workload == "c2":
        x_host = c2_signal(wl["seconds"], sr, seed=1234 + rank, f0=220.0 * 2 ** (rank / 8.0))
        x = torch.from_numpy(x_host).to(dev).reshape(1, -1)
    else:
        x = c4_shard(torch, dev, rank, wl["nsig"], wl["seconds"], sr)
    nsig, nsamp = int(x.shape[0]), int(x.shape[1])
    F = int(lib.pvx_nframes(nsamp, NFFT, HOP))
    FT = F * nsig                                                   # frames per step on this rank
    K = NPKS
    from pypevoc_amd.batch import PipelinedGather, ResultWire

    win = np.hanning(NFFT)

    def make_plan(precision, mode):
        pl = ctypes.c_void_p()
        _lib.check(lib.pvx_plan_create(ctypes.byref(pl), float(sr), NFFT, HOP, K, 0.005, _lib.dptr(win), precision, 0), "pvx_plan_create")
        if mode >= 0:
            _lib.check(lib.pvx_plan_set_fft_mode(pl, mode), "pvx_plan_set_fft_mode")
        return pl

    # --comm-cus N (default 0): a gathered run may leave N compute units without an analysis workgroup -- k_fused_rev puts ONE workgroup of
    # twelve waves on every CU for the whole launch (504 of a SIMD's 512 vector registers), so a collective's kernels find no CU before
    # the analysis has finished.  Measured with the collective forced at world size 1 it does not pay (0.1199 ms per step with none,
    # 0.1241 with eight, 0.1319 with 32: profiles/r06_ab_steps.txt) -- what a gathered step waited for was the stream-level wait on the
    # slot's previous gather, see PipelinedGather.host_retire --; kept for an N > 1 run to try.
    gathered = world > 1 or os.environ.get("PVX_BENCH_FORCE_GATHER") == "1"
    if gathered and args.comm_cus > 0 and "PVX_FUSED_BLOCKS" not in os.environ:
        ncu = torch.cuda.get_device_properties(dev).multi_processor_count
        if ncu > 2 * args.comm_cus:
            os.environ["PVX_FUSED_BLOCKS"] = str(ncu - args.comm_cus)
    plan = make_plan(args.precision, args.fft_mode)
    # compute runs on an explicit stream, never on the legacy default stream: launches there synchronise
    # implicitly with other streams
    stream = torch.cuda.Stream(device=dev)
    plans, cstreams = [plan, plan], [stream, stream]
    plan_b = None
    if args.streams == 2:
        plan_b = make_plan(args.precision, args.fft_mode)
        plans = [plan, plan_b]
        cstreams = [stream, torch.cuda.Stream(device=dev)]

    # This rank's results in the reference's layout (five float64 [rows, K] arrays + totalmag + t,
    # PV.py:256-264).  With more than one rank every step ends in ONE gather to rank 0: the rows are
    # packed to the wire format (include/pvx.h: 14 B per slot at precision 32), gathered asynchronously (RCCL, its own
    # stream, double-buffered so that the gather of step i overlaps the kernels of step i+1) and
    # unpacked on rank 0 into the full [world, ...] result, bit-identical to what each rank computed.
    if gathered and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    # (precision 32: wire format 2, 14 B per slot -- the float32 a frequency is computed from instead of the float64 frequency, decoded
    # bit-exactly on the other side; PVX_BENCH_WIRE_FORMAT=1: the 18-byte format, for A/B runs)
    wire_format = 1
    if gathered and args.precision == 32 and os.environ.get("PVX_BENCH_WIRE_FORMAT", "2") == "2":
        for pl in ([plan, plan_b] if plan_b is not None else [plan]):
            _lib.check(lib.pvx_plan_set_wire_format(pl, 2), "pvx_plan_set_wire_format")
        wire_format = 2
    wire = ResultWire(plan, FT, K)
    nres = wire.result_numel() + FT
    res2 = [torch.zeros(nres, dtype=torch.float64, device=dev) for _ in range(2)]
    rp2 = [wire.result_ptrs(r.data_ptr()) for r in res2]           # f, mag, ph, realph, binno, totalmag
    tp2 = [r.data_ptr() + wire.result_numel() * 8 for r in res2]
    full = torch.zeros((world, wire.result_numel()), dtype=torch.float64, device=dev) if (gathered and rank == 0) else None

    def consume(step_no, blocks):                                  # rank 0, side stream
        s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for r, b in enumerate(blocks):
            wire.unpack(b.data_ptr(), full[r].data_ptr(), s)

    # The gathered blocks stay in the wire format on rank 0 (bit-exact, decoded when somebody reads them -- here after
    # the timed region, for the checks): unpacking world blocks per step on rank 0 would load the one rank every other
    # rank waits for with world x the work of a step's pack.  --unpack-in-step restores the unpack inside the step.
    # (four slots, retired by the host: the compute stream never waits for a collective -- see PipelinedGather.host_retire)
    pipe = PipelinedGather(wire.nbytes, torch.uint8, dev, dst=0, consume=consume if args.unpack_in_step else None, force=gathered,
                           depth=2 if args.unpack_in_step else 4, host_retire=not args.unpack_in_step)
    counter = [0]
    if gathered:
        # (the frame times of the two result blocks, once: the wire format does not carry them)
        for j in range(2):
            _lib.check(lib.pvx_analyze_dev(plan, x.data_ptr(), _lib.PVX_F32, nsamp, nsig, nsamp, rp2[j][0], rp2[j][1], rp2[j][2], rp2[j][3], rp2[j][4],
                                           tp2[j], rp2[j][5], None, ctypes.c_void_p(stream.cuda_stream)), "pvx_analyze_dev")
        torch.cuda.synchronize(dev)

    def step(single=False):
        i = counter[0]
        counter[0] += 1
        j = i % 2
        single = single or args.streams == 1
        cs = cstreams[0] if single else cstreams[j]
        if gathered:
            # the analysis writes the step's rows straight into the wire block the gather sends (pvx_analyze_dev_wire: k_fused_rev
            # stores 18 B per slot itself; a packing kernel behind it could not run beside the next step's analysis -- twelve waves
            # of 168 registers leave a CU none for another kernel -- and cost a gathered step 23 us of 125)
            with torch.cuda.stream(cs):
                buf = pipe.buffer(i)              # its previous gather has completed (and been unpacked): cs waits for it
                r = lib.pvx_analyze_dev_wire(plans[0] if single else plans[j], x.data_ptr(), _lib.PVX_F32, nsamp, nsig, nsamp,
                                             buf.data_ptr(), ctypes.c_void_p(cs.cuda_stream))
                _lib.check(r, "pvx_analyze_dev_wire")
                if not os.environ.get("PVX_BENCH_NO_SUBMIT"):     # (A/B: the step without its collective)
                    pipe.submit(i)                # asynchronous gather to rank 0 (RCCL's stream waits for cs here)
            return j
        rp = rp2[j]
        r = lib.pvx_analyze_dev(plans[0] if single else plans[j], x.data_ptr(), _lib.PVX_F32, nsamp, nsig, nsamp,
                                rp[0], rp[1], rp[2], rp[3], rp[4], tp2[j], rp[5], None, ctypes.c_void_p(cs.cuda_stream))
        _lib.check(r, "pvx_analyze_dev")
        return j

    def local_block(step_no, j):
        """gathered runs: this rank's rows of step `step_no` in the reference's layout, decoded from the wire block it sent (after a fence)"""
        wire.unpack(pipe.bufs[step_no % pipe.depth].data_ptr(), res2[j].data_ptr(), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        torch.cuda.synchronize(dev)

    def fence():
        pipe.drain()                              # every outstanding gather has been waited for (and unpacked)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    enq = [0.0]

    def timed(nsteps):
        """Exactly nsteps steps between two fences; returns (wall seconds, HIP-event ms, result block index of the last step)."""
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        fence()
        t0 = time.perf_counter()
        e0.record(cstreams[0])
        if args.streams == 2:
            cstreams[1].wait_event(e0)
        last = 0
        for _ in range(nsteps):
            last = step()
        enq[0] = time.perf_counter() - t0         # the host's part: launches (and collectives) issued, nothing waited for
        if args.streams == 2:
            cstreams[0].wait_stream(cstreams[1])
        e1.record(cstreams[0])
        fence()
        return time.perf_counter() - t0, e0.elapsed_time(e1), last

    # ---- from idle first: the W warm-up steps and the K timed steps exactly as the contract says, on a card that has
    # just been handed the workload (a step is 0.14 ms: W + K steps are 3 ms of work and the card needs tens of
    # milliseconds of continuous work to leave its idle clocks) -> `value_from_idle`
    idle = None
    if args.clock_warmup_ms > 0:
        for _ in range(args.warmup):
            step()
        idle = timed(args.steps)
    # ---- clock ramp, then the same W + K again -> `value`: untimed passes of the same step for --clock-warmup-ms
    # (a steady job's state; measured: 326 M frames/s straight from idle, 358 M over 200 steps, 369 M over 1000)
    ramp_steps = 0
    if args.clock_warmup_ms > 0:
        # (the analysis launch alone, into block 0: no pack, no collective -- the ranks of a multi-GPU run loop by wall
        # time here and must not disagree about how many gathers they have posted)
        t_ramp = time.perf_counter() + args.clock_warmup_ms * 1e-3
        rp0 = rp2[0]
        while time.perf_counter() < t_ramp:
            for _ in range(64):
                lib.pvx_analyze_dev(plans[0], x.data_ptr(), _lib.PVX_F32, nsamp, nsig, nsamp, rp0[0], rp0[1], rp0[2], rp0[3], rp0[4],
                                    tp2[0], rp0[5], None, ctypes.c_void_p(cstreams[0].cuda_stream))
            ramp_steps += 64
            torch.cuda.synchronize(dev)
            if ramp_steps >= 1 << 16:
                break
        fence()
    for _ in range(args.warmup):
        step()
    fence()
    # ---- timed region: exactly K steps
    elapsed_local, ev_ms, last = timed(args.steps)
    host_issue_ms = enq[0] / args.steps * 1e3
    if os.environ.get("PVX_BENCH_NO_SUBMIT"):                     # (A/B only: no collective was posted, nothing below applies)
        print(json.dumps(dict(ab="gathered step without its collective", ms_per_step=round(elapsed_local / args.steps * 1e3, 4), host_issue_ms_per_step=round(host_issue_ms, 4))))
        return 0
    if gathered:
        local_block(counter[0] - 1, last)         # what was timed is what is checked: the last timed step's rows, from the block it sent
    res = res2[last]
    # ---- the same K steps once more with libpvx_hip's stage events on the launch stream: per-kernel
    # launch durations for the roofline line (kept out of the timed region: the extra event records
    # cost a few percent)
    _lib.check(lib.pvx_plan_set_timing(plan, 1), "pvx_plan_set_timing")
    for _ in range(args.steps):
        last2 = step(single=True)                 # one stream, one plan: launches do not overlap here
    fence()
    if gathered:
        local_block(counter[0] - 1, last2)
    ms = (ctypes.c_double * 4)()
    nl = (ctypes.c_int64 * 4)()
    _lib.check(lib.pvx_plan_get_timing(plan, ms, nl), "pvx_plan_get_timing")
    _lib.check(lib.pvx_plan_set_timing(plan, 0), "pvx_plan_set_timing")
    fft_mode = int(lib.pvx_plan_get_fft_mode(plan))
    elapsed = elapsed_local
    elapsed_idle = idle[0] if idle else None
    per_rank_ms = [elapsed_local / args.steps * 1e3]
    if world > 1:
        tl = torch.tensor([elapsed_local, idle[0] if idle else 0.0], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(tl) for _ in range(world)]
        dist.all_gather(allt, tl)
        per_rank_ms = [float(t[0].item()) / args.steps * 1e3 for t in allt]
        elapsed = max(float(t[0].item()) for t in allt)
        if idle:
            elapsed_idle = max(float(t[1].item()) for t in allt)

    gather_info = None
    gather_check_failed = False
    if gathered and rank == 0:
        # the block rank 0 received from itself must be, bit for bit, what its kernels wrote
        torch.cuda.synchronize(dev)
        if not args.unpack_in_step:
            consume(counter[0] - 1, pipe.result(counter[0] - 1))      # decode the last step's gathered blocks now
            torch.cuda.synchronize(dev)
        if not torch.equal(full[0].view(torch.int64), res2[last2][: wire.result_numel()].view(torch.int64)):
            sys.exit("bench.py: the gathered + unpacked block of rank 0 differs from its local result")
        n_ok = int((full[:, : FT * K] > 0).sum().item())
        gather_info = dict(collective="one asynchronous RCCL gather per step to rank 0, double-buffered",
                           unpack=("inside the step, on a side stream of rank 0" if args.unpack_in_step else
                                   "on demand: the gathered blocks stay in the wire format on rank 0 (decoded after the timed region for the checks)"),
                           wire_format=wire_format, wire_bytes_per_slot=(14 if wire_format == 2 else (18 if args.precision == 32 else 26)),
                           rccl_world=int(dist.get_world_size()), wire_bytes_per_rank=int(wire.nbytes),
                           result_bytes_per_rank=int(wire.result_numel() * 8), valid_peaks_gathered=n_ok)
        if args.check_gathered > 0:
            # what arrived through pack -> RCCL gather -> unpack, against the oracle on the signals themselves
            from oracle import pvoracle
            from tests.parity import compare_analysis
            pvoracle.build()
            hfull = full[0].cpu().numpy()
            nchk = min(args.check_gathered, nsig)
            bad = []
            for b in range(nchk):
                ob = pvoracle.analyze(x[b].cpu().numpy().astype(np.float64), sr, NFFT, HOP, NPKS)
                got = {k: hfull[i * FT * K:(i + 1) * FT * K].reshape(FT, K)[b * F:(b + 1) * F] for i, k in enumerate(("f", "mag", "ph", "realph", "binno"))}
                got["totalmag"] = hfull[5 * FT * K: 5 * FT * K + FT][b * F:(b + 1) * F]
                c = compare_analysis(got, ob, NFFT, HOP, sr)
                if not (c["bad_peaks"] <= 1e-3 * max(c["ref_peaks"], 1) and c["ph_norm"] <= 2e-6 and c["realph_norm"] <= 2e-5 and c["f_norm"] <= 2e-5
                        and c["mag_norm"] <= 1e-6 and c["totalmag_rel"] <= 1e-6):
                    bad.append(b)
            gather_info["checked_signals"] = nchk
            gather_info["check_ok"] = not bad
            if bad:
                sys.stderr.write("bench.py: gathered results of signals %s differ from the oracle\n" % bad)
                gather_check_failed = True

    rc = 0
    if rank == 0:
        ab = alg_bytes()
        frames_total = FT * world * args.steps
        value = frames_total / elapsed
        # fft mode 0 with a stage-3 span: k_stft_pv.hip (general path, STFT + peaks in one launch, spectrum rows still written)
        kname = {0: "k_stft_pv", 1: "k_fused_pv", 2: "k_fused_mw", 3: "k_fused_ring", 4: "k_fused_rev", 5: "k_fused_team"}.get(fft_mode, "k_fused_pv")
        # general path: k_stft.hip writes the spectrum rows when it can (no frame buffer, no rocFFT launches)
        names = ["k_stft" if (nl[0] > 0 and nl[1] == 0) else "k_frames", "rocfft_r2c", "k_phase_peaks", kname]
        s_in = 4 if args.precision == 32 else 8
        abp = alg_bytes(s=s_in, c=2 * s_in)
        abk = [abp["frames"], abp["fft"], abp["peaks"],
               ab["fused"] if fft_mode else HOP * 4 + (NFFT // 2) * 2 * s_in + NPKS * 40 + 16]
        per = []
        for i in range(4):
            if nl[i]:
                dur = ms[i] * 1e-3 / nl[i]                         # mean launch duration [s]
                frames_per_launch = FT * args.steps / float(nl[i])
                ach = abk[i] * frames_per_launch / dur
                per.append(dict(kernel=names[i], ms_per_launch=dur * 1e3, launches=int(nl[i]),
                                alg_bytes_per_frame=abk[i], achieved_GBps=ach / 1e9))
        dom = max(per, key=lambda d: d["ms_per_launch"] * d["launches"]) if per else None
        traffic, traffic_prov = None, None
        if dom and args.workload == "c2" and not args.seconds:
            traffic, traffic_prov = committed_traffic(dom["kernel"] if args.precision == 32 else dom["kernel"] + "_f64")
        roofline = None
        if dom:
            fpl = FT * args.steps / dom["launches"]
            roofline = dict(bound="hbm", kernel=dom["kernel"], achieved=round(dom["achieved_GBps"], 1),
                            peak=HBM_PEAK / 1e9, unit="GB/s", frac=round(dom["achieved_GBps"] * 1e9 / HBM_PEAK, 4),
                            traffic=traffic, traffic_provenance=traffic_prov, ms_per_launch=round(dom["ms_per_launch"], 4),
                            alg_bytes_per_frame=dom["alg_bytes_per_frame"],
                            alg_bytes_per_launch=int(dom["alg_bytes_per_frame"] * fpl),
                            traffic_over_algorithmic=(round(traffic / (dom["alg_bytes_per_frame"] * fpl), 3) if traffic else None),
                            traffic_GBps=(round(traffic / (dom["ms_per_launch"] * 1e-3) / 1e9, 1) if traffic else None),
                            throughput_vs_60pct_target=round(value / world / TARGET_FPS, 3),
                            note="the fused kernels are not HBM-bound: `issue` prices the same launch against vector "
                                 "issue and LDS; throughput_vs_60pct_target = per-GPU frames/s over north_star's "
                                 "1.365e8 frames/s (60 % of 8 TB/s at the 35 168 B/frame of the three-kernel split)")
            if fft_mode in (1, 2, 3, 4, 5):
                roofline["issue"] = issue_bound(dom["kernel"], fpl, dom["ms_per_launch"])
        stage_s = sum(ms[i] for i in range(4)) * 1e-3 / args.steps
        stage = dict(fft_mode=fft_mode, ms_per_step_kernels=round(stage_s * 1e3, 4),
                     kernels=[{k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items()} for d in per],
                     step_ms_hip_events=round(ev_ms / args.steps, 4))

        extras_ok = world == 1 and args.workload == "c2" and not args.no_extras
        self_check = cpu = None
        checks_failed = []

        def check_block(h, o, F_, K_, nfft, hop, precision, what, well_conditioned=True, sr=sr):
            """A timed result block (host copy, the reference's layout) against the oracle's result on the same signal,
            every frame; the tolerances of tests/test_hip_parity.py.  `well_conditioned=False` (noise, recordings at float32:
            thousands of peaks a float32 ulp apart in magnitude): the normalised errors only, and the share of frames whose
            peak set differs is reported, bounded at 1 %."""
            from tests.parity import compare_analysis
            got = {k: h[i * F_ * K_:(i + 1) * F_ * K_].reshape(F_, K_) for i, k in enumerate(("f", "mag", "ph", "realph", "binno"))}
            got["totalmag"] = h[5 * F_ * K_: 5 * F_ * K_ + F_]
            c = compare_analysis(got, o, nfft, hop, sr)
            shares = None
            if precision == 32:
                ok = c["ph_norm"] <= 2e-6 and c["mag_norm"] <= 1e-6 and c["totalmag_rel"] <= 1e-6
                if well_conditioned:
                    ok = ok and (c["realph_norm"] <= 2e-5 and c["f_norm"] <= 2e-5 and c["bad_peaks"] <= 1e-3 * max(c["ref_peaks"], 1)
                                 and c["f_abs"] <= 1e-3 and c["mag_rel"] <= 1e-5 and c["ph_abs"] <= 2e-5)
                else:
                    # f and realph come from the phase difference to the previous frame's bin: over a weak previous bin they are
                    # ill-conditioned beyond what the frame-maximum normalisation sees -- 99.9 % of the peaks within the float32
                    # tolerance, every one within 100 x, at most 1 % of the frames with another peak set
                    from tests.parity import peak_error_shares
                    shares = peak_error_shares(got, o, hop, sr)
                    ok = ok and (shares["f"] >= 0.999 and shares["realph"] >= 0.999 and c["f_norm"] <= 2e-3 and c["realph_norm"] <= 2e-3
                                 and c["frames_diff"] <= 0.01 * max(c["frames"], 1))
            else:
                # (a handful of frames of 51 676 may differ on noise: two candidate magnitudes equal to the last bit, where the
                # reference's own choice hangs on its libm's rounding of abs())
                ok = (c["bad_peaks"] <= (0 if well_conditioned else 1e-4 * max(c["ref_peaks"], 1)) and c["f_abs"] <= 1e-9 and c["mag_rel"] <= 1e-12
                      and c["ph_abs"] <= 1e-10 and c["realph_abs"] <= 1e-10 and c["totalmag_rel"] <= 1e-12)
            ok = ok and bool(np.array_equal(h[5 * F_ * K_ + F_: 5 * F_ * K_ + 2 * F_], o["t"]))
            d = dict(ok=bool(ok), frames=int(c["frames"]), frames_with_other_peaks=int(c["frames_diff"]), ref_peaks=int(c["ref_peaks"]),
                     bad_peaks=int(c["bad_peaks"]), f_abs_Hz=float(c["f_abs"]), mag_rel=float(c["mag_rel"]), ph_abs_rad=float(c["ph_abs"]),
                     realph_abs_rad=float(c["realph_abs"]),
                     normalised=dict(f=float(c["f_norm"]), ph=float(c["ph_norm"]), realph=float(c["realph_norm"]), mag=float(c["mag_norm"]),
                                     totalmag_rel=float(c["totalmag_rel"]), share_of_peaks_within_f32_tolerance=shares,
                                     note="errors in units of a complex-bin perturbation relative to the frame's largest magnitude (tests/parity.py)"),
                     against="oracle/pvoracle.c on the same signal, all frames of the last timed pass")
            if not ok:
                checks_failed.append(what)
            return d

        o = None
        if world == 1 and args.workload == "c2" and not args.no_cpu_baseline:
            from oracle import pvoracle
            pvoracle.build()
            xs = x_host.astype(np.float64)
            t1 = time.perf_counter()
            o = pvoracle.analyze(xs, sr, NFFT, HOP, NPKS)
            dtc = time.perf_counter() - t1
            # ---- what was timed is what is checked: the result block of the last timed step, every frame
            h = res.cpu().numpy()
            self_check = check_block(np.concatenate([h[: 5 * FT * K + FT], h[nres - FT:]]), o, FT, K, NFFT, HOP, args.precision, "headline")
            cpu = cpu_baseline(x_host, sr, o, dtc)

        def quick(pl, xin, steps, nfft=NFFT, hop=HOP, ramp=True, K=K):
            """frames/s of `steps` passes over a [1, n] device signal on `pl` (events on the launch stream); returns the
            frame count, ms per pass and the result block of the last pass."""
            n = int(xin.numel())
            Fq = int(lib.pvx_nframes(n, nfft, hop))
            out = torch.zeros(5 * Fq * K + 2 * Fq, dtype=torch.float64, device=dev)
            b = out.data_ptr()
            ptrs = [b + i * Fq * K * 8 for i in range(5)] + [b + 5 * Fq * K * 8, b + 5 * Fq * K * 8 + Fq * 8]
            sp = ctypes.c_void_p(stream.cuda_stream)
            for _ in range(2):
                _lib.check(lib.pvx_analyze_dev(pl, xin.data_ptr(), _lib.PVX_F32, n, 1, n, *ptrs, None, sp), "pvx_analyze_dev")
            torch.cuda.synchronize(dev)
            if args.clock_warmup_ms > 0 and ramp:               # the plan / signal set-up above let the card idle
                t_r = time.perf_counter() + min(args.clock_warmup_ms, 100.0) * 1e-3
                while time.perf_counter() < t_r:
                    for _ in range(16):
                        lib.pvx_analyze_dev(pl, xin.data_ptr(), _lib.PVX_F32, n, 1, n, *ptrs, None, sp)
                    torch.cuda.synchronize(dev)
            a0 = torch.cuda.Event(enable_timing=True)
            a1 = torch.cuda.Event(enable_timing=True)
            a0.record(stream)
            for _ in range(steps):
                lib.pvx_analyze_dev(pl, xin.data_ptr(), _lib.PVX_F32, n, 1, n, *ptrs, None, sp)
            a1.record(stream)
            torch.cuda.synchronize(dev)
            msq = a0.elapsed_time(a1) / steps
            return Fq, msq, out

        def host_block(out, Fq, K=K):
            """quick()'s result block on the host in check_block's order: the five arrays, totalmag, t."""
            h = out.cpu().numpy()
            return np.concatenate([h[: 5 * Fq * K], h[5 * Fq * K + Fq:], h[5 * Fq * K: 5 * Fq * K + Fq]])

        f64 = workloads = other_nfft = chain = config5 = host_batch = None
        if extras_ok:
            from oracle import pvoracle
            pvoracle.build()
            checks = o is not None                                  # (--no-cpu-baseline also skips the oracle runs)
            # ---- the other material first (host copies for the oracle): white noise, a tiled violin recording
            g = torch.Generator(device=dev)
            g.manual_seed(1)
            xn = 0.1 * torch.randn(nsamp, device=dev, generator=g)
            on = pvoracle.analyze(xn.cpu().numpy().astype(np.float64), sr, NFFT, HOP, NPKS) if checks else None
            xv = ov = None
            g7 = os.path.join(ROOT, "tests", "golden", "G7_perlman.npz")
            if os.path.exists(g7):
                xvh = np.load(g7)["x"].astype(np.float32)
                xvh = np.tile(xvh, nsamp // len(xvh) + 1)[:nsamp]
                xv = torch.from_numpy(xvh).to(dev)
                ov = pvoracle.analyze(xvh.astype(np.float64), sr, NFFT, HOP, NPKS) if checks else None
            # ---- the reference's own precision on the same workload
            p64 = make_plan(64, -1)
            F64, ms64, out64 = quick(p64, x, args.steps)
            _lib.check(lib.pvx_plan_set_timing(p64, 1), "pvx_plan_set_timing")
            quick(p64, x, 3, ramp=False)                           # exactly 2 + 3 launches of each kernel are recorded
            m64 = (ctypes.c_double * 4)()
            n64 = (ctypes.c_int64 * 4)()
            _lib.check(lib.pvx_plan_get_timing(p64, m64, n64), "pvx_plan_get_timing")
            _lib.check(lib.pvx_plan_set_timing(p64, 0), "pvx_plan_set_timing")
            ab64 = alg_bytes(s=8, c=16)
            stft = n64[0] > 0 and n64[1] == 0                      # k_stft.hip wrote the spectra: no frame buffer, no rocFFT
            kern64 = (lib.pvx_plan_last_kernels(p64) or b"").decode()
            rev64 = "analysis=k_pv_rev" in kern64                  # rows walked downwards, no spectrum row in HBM (k_pv_rev.hip)
            # input samples are float32 in HBM.  k_stft: hop*4 in, (nfft/2)*16 out; framing kernel: hop*4 in, nfft*8 out
            # the one-launch kernels are priced at the FUSED bytes -- hop*4 in, the result row out (SURVEY 8(d)'s lower bound: 2 384 B
            # per frame here) --, whether they keep the row on chip (k_pv_rev) or stream it through a workspace (k_stft_pv: its
            # 16 KB per frame then show as traffic_over_algorithmic, not as "algorithmic" bytes)
            ab64k = [HOP * 4 + (NFFT // 2) * 16 if stft else HOP * 4 + NFFT * 8, ab64["fft"], (NFFT // 2) * 16 + NPKS * 40 + 16,
                     HOP * 4 + NPKS * 40 + 16]
            n64names = ["k_stft" if stft else "k_frames", "rocfft_r2c", "k_phase_peaks", "k_pv_rev" if rev64 else "k_stft_pv"]
            k64 = []
            for i in range(4):
                if n64[i]:
                    dur = m64[i] * 1e-3 / n64[i]
                    fpl = F64 * 5.0 / n64[i]                       # 2 warm-up + 3 timed passes were recorded
                    k64.append(dict(kernel=n64names[i], ms_per_launch=round(dur * 1e3, 4), launches=int(n64[i]),
                                    alg_bytes_per_frame=ab64k[i], achieved_GBps=round(ab64k[i] * fpl / dur / 1e9, 1),
                                    frac=round(ab64k[i] * fpl / dur / HBM_PEAK, 4)))
            stage64 = sum(ab64k[i] for i in range(4) if n64[i])    # what the kernels of this path have to move per frame
            tr64, tr64_prov = committed_traffic("k_pv_rev_f64" if rev64 else "k_stft_pv_f64")
            f64 = dict(value=round(F64 / ms64 * 1e3, 1), unit="frames/s", ms_per_step=round(ms64, 4), steps=args.steps, dtype="f64",
                       fft_mode=int(lib.pvx_plan_get_fft_mode(p64)), kernels=kern64,
                       contract_bytes_per_frame=ab64["contract"] - HOP * 4,
                       self_check=(check_block(host_block(out64, F64), o, F64, K, NFFT, HOP, 64, "f64") if checks else None),
                       roofline=dict(bound="hbm", stage_alg_bytes_per_frame=stage64, traffic=tr64, traffic_provenance=tr64_prov,
                                     achieved=round(stage64 * F64 / (ms64 * 1e-3) / 1e9, 1), peak=HBM_PEAK / 1e9, unit="GB/s",
                                     frac=round(stage64 * F64 / (ms64 * 1e-3) / HBM_PEAK, 4),
                                     traffic_over_algorithmic=(round(tr64 / float(stage64 * F64), 3) if tr64 else None), kernels=k64))
            Fn64, msn64, outn64 = quick(p64, xn, 5)
            f64["white_noise"] = dict(value=round(Fn64 / msn64 * 1e3, 1), unit="frames/s", ms_per_step=round(msn64, 4),
                                      self_check=(check_block(host_block(outn64, Fn64), on, Fn64, K, NFFT, HOP, 64, "f64 white noise", well_conditioned=False) if checks else None))
            lib.pvx_plan_destroy(p64)
            # ---- the same geometry on other material (the headline signal has ~10 candidate maxima per frame)
            workloads = {}
            Fn, msn, outn = quick(plan, xn, 10)
            workloads["white_noise"] = dict(value=round(Fn / msn * 1e3, 1), unit="frames/s", ms_per_step=round(msn, 4),
                                            peaks_per_frame=round(int((outn[: Fn * K] > 0).sum().item()) / Fn, 2),
                                            data="0.1 * N(0,1), 600 s at 44.1 kHz",
                                            self_check=(check_block(host_block(outn, Fn), on, Fn, K, NFFT, HOP, 32, "white noise", well_conditioned=False) if checks else None))
            if xv is not None:
                Fv, msv, outv = quick(plan, xv, 10)
                workloads["violin_g7_tiled"] = dict(value=round(Fv / msv * 1e3, 1), unit="frames/s", ms_per_step=round(msv, 4),
                                                    peaks_per_frame=round(int((outv[: Fv * K] > 0).sum().item()) / Fv, 2),
                                                    data="tests/golden/G7_perlman.npz (examples/perlmanVn.wav) tiled to 26.46 M samples",
                                                    self_check=(check_block(host_block(outv, Fv), ov, Fv, K, NFFT, HOP, 32, "violin", well_conditioned=False) if checks else None))
            # ---- the reference's own defaults on the same material: PV(x, sr) is nfft 1024, hop nfft/2, npks 20 (PV.py:72-73), float64
            # arithmetic for a float64 array -- both precisions, the harmonic signal (against the oracle) and the recording
            KD, ND, HD = 20, 1024, 512
            od = pvoracle.analyze(xs, sr, ND, HD, KD) if checks else None
            ref_defaults = dict(config="PV(x, sr): nfft %d, hop %d, npks %d" % (ND, HD, KD))
            for prec_d in (32, 64):
                pd = ctypes.c_void_p()
                _lib.check(lib.pvx_plan_create(ctypes.byref(pd), float(sr), ND, HD, KD, 0.005, _lib.dptr(np.hanning(ND)), prec_d, 0), "pvx_plan_create")
                Fd, msd, outd = quick(pd, x, 10, nfft=ND, hop=HD, K=KD)
                blk = dict(value=round(Fd / msd * 1e3, 1), unit="frames/s", ms_per_step=round(msd, 4), frames=Fd, fft_mode=int(lib.pvx_plan_get_fft_mode(pd)),
                           self_check=(check_block(host_block(outd, Fd, KD), od, Fd, KD, ND, HD, prec_d, "defaults f%d" % prec_d) if checks else None))
                if xv is not None:
                    Fdv, msdv, _ = quick(pd, xv, 5, nfft=ND, hop=HD, K=KD)
                    blk["violin_g7_tiled"] = dict(value=round(Fdv / msdv * 1e3, 1), unit="frames/s", ms_per_step=round(msdv, 4))
                ref_defaults["f%d" % prec_d] = blk
                lib.pvx_plan_destroy(pd)
            workloads["reference_defaults"] = ref_defaults
            # ---- the larger transforms of BASELINE config 5 on the same signal (hop = nfft/4): k_fused_team.hip
            other_nfft = {}
            for nf in (4096, 8192):
                hp = nf // 4
                pl = ctypes.c_void_p()
                wn = np.hanning(nf)
                _lib.check(lib.pvx_plan_create(ctypes.byref(pl), float(sr), nf, hp, K, 0.005, _lib.dptr(wn), 32, 0), "pvx_plan_create")
                Fo, mso, outo = quick(pl, x, 10, nfft=nf, hop=hp)
                Fon, mson, _ = quick(pl, xn, 10, nfft=nf, hop=hp)
                oo = pvoracle.analyze(xs, sr, nf, hp, K) if checks else None
                other_nfft[str(nf)] = dict(value=round(Fo / mso * 1e3, 1), unit="frames/s", ms_per_step=round(mso, 4), hop=hp, frames=Fo,
                                           fft_mode=int(lib.pvx_plan_get_fft_mode(pl)),
                                           white_noise=dict(value=round(Fon / mson * 1e3, 1), unit="frames/s", ms_per_step=round(mson, 4)),
                                           contract_target=round(0.6 * HBM_PEAK / alg_bytes(nfft=nf, hop=hp)["contract"], 1),
                                           self_check=(check_block(host_block(outo, Fo), oo, Fo, K, nf, hp, 32, "nfft %d" % nf) if checks else None))
                lib.pvx_plan_destroy(pl)
            # ---- the rest of the path on the headline step's results, device to device (tracker and
            # resynthesis as frames/s): PV.toSinSum (pvx_track_dev) and SinSum.synth (pvx_synth_dev) on the (F, K) arrays
            # the timed step left in HBM; the partial table against the oracle's tracker on the same arrays, the first
            # frames' waveform against the oracle's resynthesis of them
            if nsig == 1:
                nK = F * K
                rb = res.data_ptr()
                d_f, d_mag, d_rp = rb, rb + nK * 8, rb + 3 * nK * 8
                pid_d = torch.empty(nK, dtype=torch.int32, device=dev)
                pst_d = torch.empty(nK, dtype=torch.int32, device=dev)
                pln_d = torch.empty(nK, dtype=torch.int32, device=dev)
                sp = ctypes.c_void_p(stream.cuda_stream)

                def track_once():
                    P_ = lib.pvx_track_dev(d_f, d_mag, F, K, 0.5, pid_d.data_ptr(), pst_d.data_ptr(), pln_d.data_ptr(), nK, sp)
                    _lib.check(P_, "pvx_track_dev")
                    return int(P_)

                P = track_once()
                tt = []
                for _ in range(5):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    track_once()                                     # (returns the number of partials: synchronous)
                    tt.append(time.perf_counter() - t0)
                ms_trk = min(tt) * 1e3
                chain = dict(tracker=dict(value=round(F / ms_trk * 1e3, 1), unit="frames/s", ms=round(ms_trk, 4), partials=P,
                                          what="pvx_track_dev on the resident (F, K) arrays, table left in HBM; wall time of the call (it returns the partial count)"))
                if P > 0:
                    maxend = int((pst_d[:P].to(torch.int64) + pln_d[:P].to(torch.int64) - 1).max().item())
                    wlen = int(lib.pvx_synth_len(maxend, NFFT, HOP, HOP, 1.0))
                    w_d = torch.empty(wlen, dtype=torch.float64, device=dev)

                    # what SinSum.synth of this analysis runs: the float32 sample loop for a precision-32 plan (k_synth_bodies<R, float>,
                    # stated tolerance 1e-4 max|w|), the float64 one at precision 64 (1e-9 here, 1e-10 on the fixtures)
                    sflags = _lib.PVX_SYNTH_F32 if args.precision == 32 else 0

                    def synth_once():
                        _lib.check(lib.pvx_synth_dev_flags(d_f, d_mag, d_rp, pid_d.data_ptr(), F, K, pst_d.data_ptr(), pln_d.data_ptr(), P, float(sr), NFFT, HOP, HOP,
                                                           1.0, 3, w_d.data_ptr(), wlen, sp, sflags), "pvx_synth_dev_flags")

                    synth_once()
                    e0 = torch.cuda.Event(enable_timing=True)
                    e1 = torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for _ in range(5):
                        synth_once()
                    e1.record(stream)
                    torch.cuda.synchronize(dev)
                    ms_syn = e0.elapsed_time(e1) / 5
                    chain["resynthesis"] = dict(value=round(F / ms_syn * 1e3, 1), unit="frames/s", ms=round(ms_syn, 4), samples_out=wlen,
                                                samples_per_s=round(wlen / ms_syn * 1e3, 1), output_GBps=round(wlen * 8 / ms_syn / 1e6, 1),
                                                sample_loop="f32" if sflags else "f64",
                                                what="pvx_synth_dev_flags (k_synth.hip: params, extras, bodies), waveform left in HBM; HIP events over 5 launches")
                    # the whole path device to device: one analysis step, the tracker and the resynthesis back to back on the
                    # resident signal (wall clock around the three calls; the tracker's call returns the partial count)
                    rpl = rp2[last]

                    def ana_once():                                  # (the same stream as the tracker and the resynthesis)
                        _lib.check(lib.pvx_analyze_dev(plans[0], x.data_ptr(), _lib.PVX_F32, nsamp, nsig, nsamp, rpl[0], rpl[1], rpl[2], rpl[3], rpl[4],
                                                       tp2[last], rpl[5], None, sp), "pvx_analyze_dev")

                    tc = []
                    if args.clock_warmup_ms > 0:                     # (as for `value`: the oracle checks above let the card idle)
                        t_r = time.perf_counter() + min(args.clock_warmup_ms, 100.0) * 1e-3
                        while time.perf_counter() < t_r:
                            for _ in range(16):
                                ana_once()
                            torch.cuda.synchronize(dev)
                    for _ in range(8):
                        torch.cuda.synchronize(dev)
                        t0 = time.perf_counter()
                        ana_once()
                        track_once()
                        synth_once()
                        torch.cuda.synchronize(dev)
                        tc.append(time.perf_counter() - t0)
                    chain["total_ms"] = round(min(tc) * 1e3, 4)
                    chain["total"] = dict(value=round(F / min(tc), 1), unit="frames/s", ms=round(min(tc) * 1e3, 4),
                                          what="analysis step + pvx_track_dev + pvx_synth_dev on the resident signal, wall clock, best of 8 after 100 ms of untimed analysis passes",
                                          goal_ms=0.55)
                    if checks:
                        hres = res.cpu().numpy()
                        hf, hm, hr = (hres[i * nK:(i + 1) * nK].reshape(F, K) for i in (0, 1, 3))
                        opid, ost, oln = pvoracle.track(hf, hm)
                        same_tab = (P == len(ost) and np.array_equal(pid_d.cpu().numpy().reshape(F, K), opid) and
                                    np.array_equal(pst_d[:P].cpu().numpy(), ost) and np.array_equal(pln_d[:P].cpu().numpy(), oln))
                        chain["tracker"]["check"] = dict(ok=bool(same_tab), against="oracle/pvoracle.c tracker on the same (F, K) arrays: partial ids, starts and lengths identical")
                        # the oracle's resynthesis of the first frames (its own table of them: partials still alive at the
                        # cut end there, so the last frames' releases differ and are left out)
                        FC = min(F, 3000)
                        cpid, cst, cln = pvoracle.track(hf[:FC], hm[:FC])
                        ow = pvoracle.synth(hf[:FC], hm[:FC], hr[:FC], cpid, cst, cln, sr, NFFT, HOP, HOP)
                        ncmp = (FC - 8) * HOP
                        hw = w_d[:ncmp].cpu().numpy()
                        err = float(np.abs(hw - ow[:ncmp]).max())
                        wtol = 1e-4 * float(np.abs(ow[:ncmp]).max()) if sflags else 1e-9 * max(1.0, float(np.abs(ow[:ncmp]).max()))
                        okw = err <= wtol
                        chain["resynthesis"]["check"] = dict(ok=bool(okw), max_abs_err=err, tolerance=wtol, samples_compared=int(ncmp),
                                                             against="oracle/pvoracle.c resynthesis of the first %d frames of the same arrays" % FC)
                        if not (same_tab and okw):
                            sys.stderr.write("bench.py: the tracker / resynthesis of the headline results differ from the oracle: %s\n" % json.dumps(chain))
                            checks_failed.append("chain")
            # ---- BASELINE config 5: nfft {512 .. 8192} x hop {nfft/4, nfft/2} on a 60-min 96 kHz signal (345.6 M samples, 1.38 GB
            # of float32 generated in HBM), every point timed with HIP events on the launch stream and the leading 60 s of its
            # result checked against the oracle on the same samples
            if not args.no_config5:
                sr5, n5 = 96000, 96000 * args.c5_seconds
                x5 = torch.empty(n5, dtype=torch.float32, device=dev)
                g5 = torch.Generator(device=dev)
                g5.manual_seed(1234)
                for a5 in range(0, n5, 1 << 24):                      # the G4 generator (c2_signal) in float64 pieces on the device
                    b5 = min(n5, a5 + (1 << 24))
                    t5 = torch.arange(a5, b5, device=dev, dtype=torch.float64) / sr5
                    ph5 = 2 * np.pi * 220.0 * (t5 - 0.01 / (2 * np.pi * 5.0) * torch.cos(2 * np.pi * 5.0 * t5))
                    s5 = 0.001 * torch.randn(b5 - a5, generator=g5, device=dev, dtype=torch.float64)
                    for hh in range(1, 9):
                        s5 += 0.3 / hh * torch.sin(hh * ph5)
                    x5[a5:b5] = s5.to(torch.float32)
                del t5, ph5, s5
                n60 = min(n5, 60 * sr5)
                x60 = x5[:n60].cpu().numpy().astype(np.float64) if checks else None
                points = []
                for nf in (512, 1024, 2048, 4096, 8192):
                    for hp in (nf // 4, nf // 2):
                        pl = ctypes.c_void_p()
                        wn = np.hanning(nf)
                        _lib.check(lib.pvx_plan_create(ctypes.byref(pl), float(sr5), nf, hp, K, 0.005, _lib.dptr(wn), 32, 0), "pvx_plan_create")
                        F5, ms5, out5 = quick(pl, x5, 3, nfft=nf, hop=hp)
                        pt = dict(nfft=nf, hop=hp, frames=F5, ms_per_step=round(ms5, 4), value=round(F5 / ms5 * 1e3, 1), unit="frames/s",
                                  fft_mode=int(lib.pvx_plan_get_fft_mode(pl)),
                                  contract_target=round(0.6 * HBM_PEAK / alg_bytes(nfft=nf, hop=hp)["contract"], 1),
                                  frac_of_hbm_at_fused_bytes=round(F5 / ms5 * 1e3 * alg_bytes(nfft=nf, hop=hp)["fused"] / HBM_PEAK, 4))
                        pt["vs_contract_target"] = round(pt["value"] / pt["contract_target"], 3)
                        if checks:
                            o5 = pvoracle.analyze(x60, sr5, nf, hp, K)
                            F60 = len(o5["t"])
                            nk5 = F5 * K
                            h5 = torch.cat([out5[i * nk5: i * nk5 + F60 * K] for i in range(5)] +
                                           [out5[5 * nk5 + F5: 5 * nk5 + F5 + F60], out5[5 * nk5: 5 * nk5 + F60]]).cpu().numpy()
                            ck = check_block(h5, o5, F60, K, nf, hp, 32, "config5 %d/%d" % (nf, hp), sr=sr5)
                            ck["against"] = "oracle/pvoracle.c on the leading %d s of the same signal (%d frames)" % (n60 // sr5, F60)
                            pt["self_check"] = ck
                        points.append(pt)
                        lib.pvx_plan_destroy(pl)
                        del out5
                        torch.cuda.empty_cache()
                config5 = dict(signal="G4 generator at 96 kHz, %d s, %d samples of float32 resident in HBM, npks %d" % (args.c5_seconds, n5, K),
                               timing="HIP events on the launch stream over 3 launches per point, after 2 untimed launches and <= 100 ms of clock ramp",
                               points=points)
                del x5
                torch.cuda.empty_cache()
            # ---- the C-level batch entry (pvx_batch_run, include/pvx.h): ragged signals in pageable host memory over a
            # device list inside this process, results into per-signal host arrays -- PCIe-inclusive, so never `value`; every signal
            # compared bit for bit with its own PV(...).run_pv()
            host_batch = None
            if not args.no_host_batch:
                import pypevoc_amd
                rngb = np.random.default_rng(11)
                nb_sig, nb_s = 48, 30
                tb = np.arange(nb_s * 48000) / 48000.0
                sigs = []
                for b in range(nb_sig):
                    f0 = 110.0 * 2 ** (b / 48.0 * 3)
                    xb = sum(0.3 / hh * np.sin(2 * np.pi * f0 * hh * tb) for hh in range(1, 9)) + 0.001 * rngb.standard_normal(len(tb))
                    sigs.append(xb[: int(len(tb) * (0.25 + 0.75 * ((b * 29) % nb_sig) / nb_sig))].astype(np.float32))
                many = pypevoc_amd.PVMany(48000, nfft=NFFT, hop=HOP, npks=NPKS, devices=[local_rank], precision=32)
                many.run(sigs)
                tb_ = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    resb = many.run(sigs)
                    tb_.append(time.perf_counter() - t0)
                many.close()
                frb = sum(r["nframes"] for r in resb)
                same_b = True
                for i in (0, 7, nb_sig - 1):
                    pb_ = pypevoc_amd.PV(sigs[i], 48000, nfft=NFFT, hop=HOP, npks=NPKS, progress=False, precision=32)
                    pb_.run_pv()
                    same_b = same_b and all(np.array_equal(resb[i][k], getattr(pb_, k)) for k in ("f", "mag", "ph", "realph", "binno"))
                host_batch = dict(value=round(frb / min(tb_), 1), unit="frames/s", signals=nb_sig, frames=frb, seconds=round(min(tb_), 4),
                                  input_MB=round(sum(len(v) for v in sigs) * 4 / 1e6, 1), devices=[local_rank], workers_per_device=4, pcie_inclusive=True,
                                  identical_to_single_signal_calls=bool(same_b),
                                  what="pvx_batch_run: %d signals of 7.5 .. 30 s @ 48 kHz float32 in pageable host memory, results into per-signal host arrays; "
                                       "wall clock of the call, best of 3 (the kernels are ~5 %% of it: host copies and PCIe are the rest)" % nb_sig)
                if not same_b:
                    sys.stderr.write("bench.py: pvx_batch_run differs from the single-signal calls\n")
                    checks_failed.append("host_batch")
        if checks_failed or gather_check_failed:
            rc = 3

        desc = ("BASELINE config 2: one %d-s 44.1 kHz mono signal per GPU" % wl["seconds"]) if args.workload == "c2" else \
               ("BASELINE config 4 shard: %d x %d-s 48 kHz mono signals per GPU in one call" % (nsig, wl["seconds"]))
        line = {
            "metric": "STFT frames/sec (%s kHz, nfft=2048, hop=512)" % ("44.1" if sr == 44100 else "%g" % (sr / 1000.0)),
            "value": round(value, 1), "unit": "frames/s",
            "value_from_idle": (round(FT * world * args.steps / elapsed_idle, 1) if elapsed_idle else None),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == 32 else "f64", "data": "synthetic",
            "config": {"workload": "%s, nfft=2048, hop=512, npks=8, analysis only (PV.run_pv), F=%d frames/signal%s" %
                                   (desc, F, "; results gathered to rank 0 over RCCL" if world > 1 else ""),
                       "workload_short": ("BASELINE config 2: one %d-s 44.1 kHz mono signal per GPU, PV.run_pv" % wl["seconds"]) if args.workload == "c2" else
                                         ("BASELINE config 4 shard: %d x %d-s 48 kHz signals per GPU, PV.run_pv" % (nsig, wl["seconds"])),
                       "nfft": NFFT, "hop": HOP, "npks": NPKS, "sr": sr, "frames_per_gpu": FT, "signals_per_gpu": nsig,
                       "parallelism": ("independent signals sharded %d/GPU, one RCCL gather per step" % nsig) if world > 1 else "single GPU",
                       "parallelism_short": ("signals sharded %d/GPU, one RCCL gather per step" % nsig) if world > 1 else "single GPU",
                       "streams": args.streams},
            "roofline": roofline, "stage": stage, "cpu_baseline": cpu, "self_check": self_check,
            "per_rank_ms_per_step": [round(v, 4) for v in per_rank_ms],
            "clock_warmup": {"ms": args.clock_warmup_ms, "untimed_steps": ramp_steps,
                             "note": "`value_from_idle`: W warm-up + K timed steps right after the workload was made resident; then "
                                     "untimed passes of the same step for `ms` (the card needs tens of milliseconds of continuous work "
                                     "to leave its idle clocks), then W + K again: `value`.  --clock-warmup-ms 0: one run, from idle"},
        }
        if f64:
            line["f64"] = f64
        if workloads:
            line["workloads"] = workloads
        if other_nfft:
            line["other_nfft"] = other_nfft
        if chain:
            line["chain"] = chain
        if config5:
            line["config5"] = config5
        if host_batch:
            line["host_batch"] = host_batch
        if gather_info:
            gather_info["ms_per_step_kernels_only"] = stage["step_ms_hip_events"]
            gather_info["exposed_ms_per_step"] = round(max(0.0, elapsed / args.steps * 1e3 - stage["step_ms_hip_events"]), 4)
            gather_info["host_issue_ms_per_step"] = round(host_issue_ms, 4)
            gather_info["note"] = ("exposed = wall time per step minus the analysis kernels' time on the compute stream (launch gaps included); "
                                   "host_issue = the host's time to issue a step (analysis launch + the collective's enqueue): a step cannot be shorter")
            line["gather"] = gather_info
        try:
            ctypes.CDLL(None).fflush(None)        # RCCL's version banner (C stdio) goes out before the JSON line
        except Exception:
            pass
        if extras_ok:
            line["extras_ok"] = not checks_failed
        # the whole record goes to a file (and, pretty-printed, nowhere else); stdout gets ONE compact line the driver can parse
        detail_path = args.detail if os.path.isabs(args.detail) else os.path.join(ROOT, args.detail)
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as fh:
                json.dump(line, fh, indent=1)
                fh.write("\n")
        except OSError as e:
            sys.stderr.write("bench.py: could not write %s: %s\n" % (detail_path, e))
            detail_path = None
        print(compact_line(line, os.path.relpath(detail_path, ROOT) if detail_path else None))
        sys.stdout.flush()
        if rc:
            sys.stderr.write("bench.py: timed output outside the stated tolerances against the oracle: %s\n" % ", ".join(checks_failed))
    lib.pvx_plan_destroy(plan)
    if plan_b is not None:
        lib.pvx_plan_destroy(plan_b)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
