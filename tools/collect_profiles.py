#!/usr/bin/env python3
"""Copy what tools/gpu_bench_round.sh left under gpurun_out/ into profiles/ (the tracked evidence bench.py and
DESIGN.md cite):   python tools/collect_profiles.py r02
  profiles/<tag>_fused_sq.json        SQ counters of the analysis kernel, per-launch means (tools/prof_sq.sh)
  profiles/traffic_latest.json       HBM bytes per launch of the analysis kernel: FETCH_SIZE x calibration + WRITE_SIZE
  profiles/<tag>_traffic_pmc.json     the raw FETCH_SIZE / WRITE_SIZE means (run + calibration kernels)
  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 3`
  profiles/<tag>_bench*.json          the bench lines of that pass"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
FRAMES = 51676

sq = json.load(open(os.path.join(G, "sq_%s" % tag, "summary.json")))
keep = {k: v for k, v in sq.items() if "k_fused" in k or "k_phase" in k or "k_frames" in k}
keep["_frames_per_launch"] = FRAMES
keep["_method"] = ("rocprofv3 --pmc <8 SQ counters per pass> -- python3 tools/run_mode.py -1 harmonic 8 4 (tools/prof_sq.sh): "
                   "BASELINE config 2 signal, plan default fft mode; values are means over the kernel's launches; "
                   "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, "
                   "SQ_BUSY_CU_CYCLES and SQ_LDS_IDX_ACTIVE cycles summed over CUs")
json.dump(keep, open(os.path.join(P, "%s_fused_sq.json" % tag), "w"), indent=1)

run = json.load(open(os.path.join(G, "traffic_%s" % tag, "run.json")))
cal = json.load(open(os.path.join(G, "traffic_%s" % tag, "cal.json")))
json.dump(dict(run={k: v for k, v in run.items() if "k_" in k}, calibration=cal), open(os.path.join(P, "%s_traffic_pmc.json" % tag), "w"), indent=1)
calk = [v for k, v in cal.items() if "k_read<HIP_vector_type<float, 2u>" in k][0]
factor = (1 << 30) / (calk["FETCH_SIZE"]["mean"] * 1024.0)              # 8-byte-per-lane reads of 1 GiB
wcal = [v for k, v in cal.items() if "k_write<double>" in k][0]
wfactor = (1 << 30) / (wcal["WRITE_SIZE"]["mean"] * 1024.0)
out = {}
detail = {}
for k, v in run.items():
    if "k_fused" not in k or "FETCH_SIZE" not in v:
        continue
    short = re.search(r"(k_[a-z_0-9]+)", k).group(1)
    rd = v["FETCH_SIZE"]["mean"] * 1024.0 * factor
    wr = v["WRITE_SIZE"]["mean"] * 1024.0 * wfactor
    out[short] = int(rd + wr)
    detail[short] = dict(kernel=k, frames_per_launch=FRAMES, FETCH_SIZE_KB_mean=v["FETCH_SIZE"]["mean"], fetch_calibration_factor=factor,
                         WRITE_SIZE_KB_mean=v["WRITE_SIZE"]["mean"], write_calibration_factor=wfactor, read_bytes=int(rd), write_bytes=int(wr),
                         bytes_per_frame=round((rd + wr) / FRAMES, 1))
old = {}
try:
    old = json.load(open(os.path.join(P, "traffic_latest.json")))
except Exception:
    pass
for k, v in old.items():
    if not k.startswith("_") and k not in out:
        out[k] = v
out["_detail"] = dict(kernels=detail, previous_round=old.get("_detail"),
                      method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `tools/run_mode.py -1 harmonic 8 10` "
                             "(tools/prof_traffic.sh); FETCH_SIZE calibrated on tools/fetch_calib (1 GiB streamed with 8-byte-per-lane loads "
                             "reports 1/2 of the bytes on gfx950), WRITE_SIZE reads exact",
                      source=["profiles/%s_traffic_pmc.json" % tag])
json.dump(out, open(os.path.join(P, "traffic_latest.json"), "w"), indent=1)
st = os.path.join(G, "stats_%s" % tag, "r_kernel_stats.csv")
if os.path.exists(st):
    shutil.copy(st, os.path.join(P, "%s_kernel_stats.csv" % tag))
for src, dst in (("bench_%sa.json" % tag, "%s_bench.json" % tag), ("bench_c4.json", "%s_bench_c4.json" % tag), ("bench_fg.json", "%s_bench_forced_gather.json" % tag)):
    s = os.path.join(G, src)
    if os.path.exists(s):
        lines = [ln for ln in open(s).read().splitlines() if ln.startswith("{")]
        if lines:
            open(os.path.join(P, dst), "w").write(lines[-1] + "\n")
for src, dst in (("configs_%s.jsonl" % tag, "%s_configs_3_4.jsonl" % tag), ("sweep5_%s.jsonl" % tag, "%s_config5_sweep.jsonl" % tag),
                 ("sweep5_%s_f64.jsonl" % tag, "%s_config5_sweep_f64.jsonl" % tag), ("ab_nfft_%s.jsonl" % tag, "%s_nfft_harmonic_vs_noise.jsonl" % tag)):
    s = os.path.join(G, src)
    if os.path.exists(s) and os.path.getsize(s):
        shutil.copy(s, os.path.join(P, dst))
t64 = os.path.join(G, "traffic_%s_f64" % tag, "run.json")
if os.path.exists(t64):
    r64 = json.load(open(t64))
    o64 = {}
    for k, v in r64.items():
        if ("k_stft" in k or "k_phase_peaks" in k) and "FETCH_SIZE" in v:
            short = re.search(r"(k_[a-z_0-9]+)", k).group(1)
            rd = v["FETCH_SIZE"]["mean"] * 1024.0 * factor
            wr = v["WRITE_SIZE"]["mean"] * 1024.0 * wfactor
            o64[short] = dict(kernel=k, read_bytes=int(rd), write_bytes=int(wr), bytes_per_frame=round((rd + wr) / FRAMES, 1),
                              FETCH_SIZE_KB_mean=v["FETCH_SIZE"]["mean"], WRITE_SIZE_KB_mean=v["WRITE_SIZE"]["mean"])
    json.dump(dict(kernels=o64, frames_per_launch=FRAMES, method="tools/prof_traffic.sh OUT -1 harmonic 8 64 (float64 plan), same calibration as traffic_latest.json"),
              open(os.path.join(P, "%s_traffic_f64.json" % tag), "w"), indent=1)
    print(json.dumps({k: v["bytes_per_frame"] for k, v in o64.items()}))
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}))
