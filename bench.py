#!/usr/bin/env python3
"""bench.py -- STFT frames/s of the phase-vocoder analysis stage (PV.run_pv) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU)

Metric (BASELINE.json): STFT frames/sec at 44.1 kHz, nfft=2048, hop=512, npks=8.
A "step" is one full run_pv pass over the workload, with the signal(s) already resident in HBM:
  N = 1 : BASELINE config 2 -- one 10-minute 44.1 kHz mono signal (26 460 000 samples, F = 51 676).
  N > 1 : weak scaling -- one such signal per GPU (independent signals are the unit the path
          shards on), plus the single result gather to rank 0 over RCCL/xGMI inside the step
          (asynchronous, double-buffered: it overlaps the next step's kernels; all gathers have
          completed before the clock stops).
value = frames processed by all ranks / wall time of the K timed steps (max over ranks).

Also printed on the same JSON line:
  roofline     -- the dominant kernel of the stage: algorithmic bytes per launch / its mean launch
                  duration measured with HIP events recorded on the launch stream (libpvx_hip's
                  stage timing), against the 8 TB/s HBM3E peak.
  stage        -- the whole STFT+phase stage priced at SURVEY.md 8(d)'s 35 168 B/frame.
  cpu_baseline -- the oracle (C port of the reference algorithm, single thread) timed on this host
                  on the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR, NFFT, HOP, NPKS, SECONDS = 44100, 2048, 512, 8, 600
HBM_PEAK = 8.0e12                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def c2_signal(seconds=SECONDS, sr=SR, seed=1234, f0=220.0):
    """SURVEY.md 8(d) C2 / G4 generator: 8 harmonics, 1 % / 5 Hz vibrato, amplitudes 0.3/h, noise."""
    n = int(sr * seconds)
    t = np.arange(n, dtype=np.float64) / sr
    ph = 2 * np.pi * f0 * (t - 0.01 / (2 * np.pi * 5.0) * np.cos(2 * np.pi * 5.0 * t))
    x = np.zeros(n)
    for h in range(1, 9):
        x += 0.3 / h * np.sin(h * ph)
    x += 0.001 * np.random.default_rng(seed).standard_normal(n)
    return x.astype(np.float32)


def alg_bytes(nfft=NFFT, hop=HOP, npks=NPKS, s=4, c=8):
    """Algorithmic bytes per frame: per kernel and for the stage."""
    out = npks * 5 * 8 + 16
    return dict(frames=hop * s + nfft * s,
                fft=nfft * s + (nfft // 2 + 1) * c,
                peaks=(nfft // 2 + 1) * c + out,
                stage=hop * s + 2 * nfft * s + 2 * (nfft // 2 + 1) * c + out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fft-mode", type=int, default=-1, help="-1: plan default (fused where supported), 0: rocFFT path, 1: fused")
    ap.add_argument("--streams", type=int, default=1, choices=(1, 2),
                    help="2: consecutive steps alternate between two streams (hides the launch gap and the kernel tail, "
                         "+2.5 %%; per-kernel durations then overlap and no longer compare with rocprofv3's)")
    ap.add_argument("--seconds", type=int, default=SECONDS, help=argparse.SUPPRESS)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from pypevoc_amd import _lib
    lib = _lib.load()
    _lib.init(local_rank)

    # ---- workload, resident in HBM before the timed region
    x_host = c2_signal(args.seconds, seed=1234 + rank, f0=220.0 * 2 ** (rank / 8.0))
    x = torch.from_numpy(x_host).to(dev)
    nsamp = x.numel()
    F = int(lib.pvx_nframes(nsamp, NFFT, HOP))
    K = NPKS
    from pypevoc_amd.batch import PipelinedGather, ResultWire

    plan = ctypes.c_void_p()
    win = np.hanning(NFFT)
    _lib.check(lib.pvx_plan_create(ctypes.byref(plan), float(SR), NFFT, HOP, K, 0.005, _lib.dptr(win),
                                   args.precision, 0), "pvx_plan_create")
    if args.fft_mode >= 0:
        _lib.check(lib.pvx_plan_set_fft_mode(plan, args.fft_mode), "pvx_plan_set_fft_mode")
    # --streams 2: consecutive steps alternate between two streams (each with its own plan and result
    # block): a step is one persistent kernel whose waves finish at slightly different times, and the
    # ~5 us the GPU needs to start the next kernel of the SAME stream sits on top of that tail; with two
    # streams the waves of step i+1 move in as those of step i drain.  Off by default: overlapping
    # launches have no well-defined individual duration, and the roofline line wants one that agrees
    # with rocprofv3's.
    # compute runs on an explicit stream, never on the legacy default stream: launches there synchronise
    # implicitly with other streams (with a second stream alive, ~0.27 ms per step were lost to that)
    stream = torch.cuda.Stream(device=dev)
    plans, cstreams = [plan, plan], [stream, stream]
    plan_b = None
    if args.streams == 2:
        plan_b = ctypes.c_void_p()
        _lib.check(lib.pvx_plan_create(ctypes.byref(plan_b), float(SR), NFFT, HOP, K, 0.005, _lib.dptr(win),
                                       args.precision, 0), "pvx_plan_create")
        if args.fft_mode >= 0:
            _lib.check(lib.pvx_plan_set_fft_mode(plan_b, args.fft_mode), "pvx_plan_set_fft_mode")
        plans = [plan, plan_b]
        cstreams = [stream, torch.cuda.Stream(device=dev)]

    # This rank's results in the reference's layout (five float64 [F, K] arrays + totalmag + t,
    # PV.py:256-264).  With more than one rank every step ends in ONE gather to rank 0: the rows are
    # packed to the 18 B/slot wire format (include/pvx.h), gathered asynchronously (RCCL, its own
    # stream, double-buffered so that the gather of step i overlaps the kernels of step i+1) and
    # unpacked on rank 0 into the full [world, ...] result, bit-identical to what each rank computed.
    gathered = world > 1 or os.environ.get("PVX_BENCH_FORCE_GATHER") == "1"
    if gathered and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    wire = ResultWire(plan, F, K)
    # two result blocks: while step i+1 is analysed into one, the other is packed and gathered
    nres = wire.result_numel() + F
    res2 = [torch.zeros(nres, dtype=torch.float64, device=dev) for _ in range(2)]
    rp2 = [wire.result_ptrs(r.data_ptr()) for r in res2]           # f, mag, ph, realph, binno, totalmag
    tp2 = [r.data_ptr() + wire.result_numel() * 8 for r in res2]
    full = torch.zeros((world, wire.result_numel()), dtype=torch.float64, device=dev) if (gathered and rank == 0) else None

    def consume(step_no, blocks):                                  # rank 0, side stream
        s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for r, b in enumerate(blocks):
            wire.unpack(b.data_ptr(), full[r].data_ptr(), s)

    pipe = PipelinedGather(wire.nbytes, torch.uint8, dev, dst=0, consume=consume, force=gathered)
    # The analysis kernel is bound by instruction issue, packing by HBM: they overlap almost for free.
    # So the compute stream only ever runs the analysis; packing and the gather of step i go to a side
    # stream (RCCL then orders itself after that stream) while step i+1 is analysed.
    pack_stream = torch.cuda.Stream(device=dev) if gathered else None
    packed = [None, None]                                          # events: res2[j] has been packed

    counter = [0]

    def step(single=False):
        i = counter[0]
        counter[0] += 1
        j = i % 2
        single = single or args.streams == 1
        cs = cstreams[0] if single else cstreams[j]
        if packed[j] is not None:
            cs.wait_event(packed[j])              # the pack of step i-2 has read this block
        rp = rp2[j]
        r = lib.pvx_analyze_dev(plans[0] if single else plans[j], x.data_ptr(), _lib.PVX_F32, nsamp, 1, nsamp,
                                rp[0], rp[1], rp[2], rp[3], rp[4], tp2[j], rp[5], None, ctypes.c_void_p(cs.cuda_stream))
        _lib.check(r, "pvx_analyze_dev")
        if gathered:
            done = torch.cuda.Event()
            done.record(cs)
            with torch.cuda.stream(pack_stream):
                pack_stream.wait_event(done)
                buf = pipe.buffer(i)              # its previous gather has completed and been unpacked
                wire.pack(res2[j].data_ptr(), buf.data_ptr(), ctypes.c_void_p(pack_stream.cuda_stream))
                ev = torch.cuda.Event()
                ev.record(pack_stream)
                packed[j] = ev
                pipe.submit(i)                    # asynchronous gather to rank 0
    res = res2[0]

    def fence():
        if gathered:
            with torch.cuda.stream(pack_stream):
                pipe.drain()                      # every outstanding gather has been waited for and unpacked
        else:
            pipe.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    # ---- timed region: exactly K steps
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    e0.record(cstreams[0])
    if args.streams == 2:
        cstreams[1].wait_event(e0)
    for _ in range(args.steps):
        step()
    if args.streams == 2:
        cstreams[0].wait_stream(cstreams[1])
    e1.record(cstreams[0])
    fence()
    elapsed = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)
    # ---- the same K steps once more with libpvx_hip's stage events on the launch stream: per-kernel
    # launch durations for the roofline line (kept out of the timed region: the extra event records
    # cost a few percent)
    _lib.check(lib.pvx_plan_set_timing(plan, 1), "pvx_plan_set_timing")
    for _ in range(args.steps):
        step(single=True)                         # one stream, one plan: launches do not overlap here
    fence()
    ms = (ctypes.c_double * 4)()
    nl = (ctypes.c_int64 * 4)()
    _lib.check(lib.pvx_plan_get_timing(plan, ms, nl), "pvx_plan_get_timing")
    _lib.check(lib.pvx_plan_set_timing(plan, 0), "pvx_plan_set_timing")
    fft_mode = int(lib.pvx_plan_get_fft_mode(plan))
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    gather_info = None
    if gathered and rank == 0:
        # the block rank 0 received from itself must be, bit for bit, what its kernels wrote
        torch.cuda.synchronize(dev)
        if not torch.equal(full[0].view(torch.int64), res[: wire.result_numel()].view(torch.int64)):
            sys.exit("bench.py: the gathered + unpacked block of rank 0 differs from its local result")
        n_ok = int((full[:, : F * K] > 0).sum().item())
        gather_info = dict(collective="one asynchronous RCCL gather per step to rank 0, double-buffered",
                           wire_bytes_per_rank=int(wire.nbytes), result_bytes_per_rank=int(wire.result_numel() * 8),
                           valid_peaks_gathered=n_ok)

    if rank == 0:
        ab = alg_bytes()
        frames_total = F * world * args.steps
        value = frames_total / elapsed
        names = ["k_frames", "rocfft_r2c", "k_phase_peaks", "k_fused_pv"]
        # algorithmic bytes per frame of each kernel.  The fused kernel is priced at the
        # stage figure of SURVEY.md 8(d): it does the work of the whole STFT+phase stage.
        abk = [ab["frames"], ab["fft"], ab["peaks"], ab["stage"]]
        per = []
        for i in range(4):
            if nl[i]:
                dur = ms[i] * 1e-3 / nl[i]                         # mean launch duration [s]
                frames_per_launch = F * args.steps / float(nl[i])  # zero rows excluded
                ach = abk[i] * frames_per_launch / dur
                per.append(dict(kernel=names[i], ms_per_launch=dur * 1e3, launches=int(nl[i]),
                                alg_bytes_per_frame=abk[i], achieved_GBps=ach / 1e9))
        dom = max(per, key=lambda d: d["ms_per_launch"] * d["launches"]) if per else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if dom and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom["kernel"])
            except Exception:
                traffic = None
        roofline = None
        if dom:
            roofline = dict(bound="hbm", kernel=dom["kernel"], achieved=round(dom["achieved_GBps"], 1),
                            peak=HBM_PEAK / 1e9, unit="GB/s", frac=round(dom["achieved_GBps"] * 1e9 / HBM_PEAK, 4),
                            traffic=traffic, ms_per_launch=round(dom["ms_per_launch"], 4),
                            alg_bytes_per_launch=int(dom["alg_bytes_per_frame"] * F * args.steps / dom["launches"]),
                            note="algorithmic bytes = SURVEY.md 8(d) contract figure for the STFT+phase stage "
                                 "(the three-kernel split north_star describes); `traffic` = HBM bytes per launch "
                                 "from rocprofv3 PMC (profiles/).  The fused kernel moves 13x fewer bytes than "
                                 "that figure (its own floor is hop*4 + outputs = %d B/frame, `fused` below), which "
                                 "is how frac can exceed 1: the kernel is bound by VALU issue, not by HBM" % (HOP * 4 + NPKS * 40 + 16))
            if dom["kernel"] == "k_fused_pv":
                fb = HOP * 4 + NPKS * 40 + 16
                fa = fb * F * args.steps / dom["launches"] / (dom["ms_per_launch"] * 1e-3)
                roofline["fused"] = dict(alg_bytes_per_frame=fb, achieved=round(fa / 1e9, 1), frac=round(fa / HBM_PEAK, 4))
        stage_s = sum(ms[i] for i in range(4)) * 1e-3 / args.steps
        stage = dict(fft_mode=fft_mode, alg_bytes_per_frame=ab["stage"], ms_per_step_kernels=round(stage_s * 1e3, 4),
                     achieved_GBps=round(ab["stage"] * F / stage_s / 1e9, 1) if stage_s > 0 else None,
                     frac_of_8TBps=round(ab["stage"] * F / stage_s / HBM_PEAK, 4) if stage_s > 0 else None,
                     kernels=[{k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items()} for d in per],
                     step_ms_hip_events=round(ev_ms / args.steps, 4))
        cpu = None
        if not args.no_cpu_baseline and world == 1:                  # contract: rank 0 at N=1 only
            from oracle import pvoracle
            pvoracle.build()
            xs = x_host.astype(np.float64)
            t1 = time.perf_counter()
            o = pvoracle.analyze(xs, SR, NFFT, HOP, NPKS)
            dtc = time.perf_counter() - t1
            cpu = dict(value=round(len(o["t"]) / dtc, 1), unit="frames/s", cores=1, kind="port",
                       sample="the full N=1 workload once (%d frames, %.1f s of CPU time), oracle/pvoracle.c single thread, "
                              "host has %d cores" % (len(o["t"]), dtc, os.cpu_count() or 0))
        line = {
            "metric": "STFT frames/sec (44.1 kHz, nfft=2048, hop=512)", "value": round(value, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == 32 else "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: one %d-s 44.1 kHz mono signal per GPU, nfft=2048, hop=512, npks=8, "
                                   "analysis only (PV.run_pv), F=%d frames/signal%s" %
                                   (args.seconds, F, "; results gathered to rank 0 over RCCL" if world > 1 else ""),
                       "nfft": NFFT, "hop": HOP, "npks": NPKS, "sr": SR, "frames_per_gpu": F,
                       "parallelism": "signals sharded 1/GPU" if world > 1 else "single GPU",
                       "streams": args.streams},
            "roofline": roofline, "stage": stage, "cpu_baseline": cpu,
        }
        if gather_info:
            line["gather"] = gather_info
        try:
            ctypes.CDLL(None).fflush(None)        # RCCL's version banner (C stdio) goes out before the JSON line
        except Exception:
            pass
        print(json.dumps(line))
        sys.stdout.flush()
    lib.pvx_plan_destroy(plan)
    if plan_b is not None:
        lib.pvx_plan_destroy(plan_b)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
