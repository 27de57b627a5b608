#!/usr/bin/env python3
"""Copy what tools/gpu_bench_round.sh left under gpurun_out/ into profiles/ (the tracked evidence bench.py and DESIGN.md
cite):   python tools/collect_profiles.py r03
  profiles/sq_latest.json            SQ counters of every analysis / tracker / resynthesis kernel profiled this round, per-launch
                                     means (tools/prof_sq.sh), stamped with the fingerprint of the kernel sources (bench.py quotes
                                     it only when that matches the build it times); the same as profiles/<tag>_sq.json
  profiles/traffic_latest.json       flat: kernel -> {bytes, read_bytes, write_bytes, bytes_per_frame, symbol, csrc_sha16, tag}:
                                     HBM bytes per launch on BASELINE config 2 (FETCH_SIZE x calibration + WRITE_SIZE)
  profiles/<tag>_traffic_pmc.json    the raw FETCH_SIZE / WRITE_SIZE means (run + calibration kernels)
  profiles/<tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 3`
  profiles/<tag>_bench*.json, <tag>_configs_3_4.jsonl, <tag>_config5_sweep*.jsonl, <tag>_nfft_harmonic_vs_noise.jsonl"""
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha16  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
FRAMES = 51676
SHA = csrc_sha16()


def short_name(k):
    m = re.search(r"(k_[a-z_0-9]+)", k)
    return m.group(1) if m else k


# ---- SQ counters: every summary.json under gpurun_out/sq_<tag>*/
sq = {"_csrc_sha16": SHA, "_tag": tag, "_frames_per_launch": FRAMES,
      "_method": ("rocprofv3 --pmc <8 SQ counters per pass> -- python3 tools/run_mode.py -1 harmonic 8 4 [precision] | tools/synth_time.py "
                  "(tools/prof_sq.sh): BASELINE config 2 signal, plan default fft mode; values are means over the kernel's launches; "
                  "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_BUSY_CU_CYCLES and "
                  "SQ_LDS_IDX_ACTIVE cycles summed over CUs")}
for d in sorted(glob.glob(os.path.join(G, "sq_%s*" % tag))):
    f = os.path.join(d, "summary.json")
    if not os.path.exists(f):
        continue
    what = os.path.basename(d)[len("sq_%s" % tag):].strip("_") or "f32"
    for k, v in json.load(open(f)).items():
        if "k_" in k and "SQ_WAVES" in v:
            sq["%s | %s" % (what, k)] = v
json.dump(sq, open(os.path.join(P, "sq_latest.json"), "w"), indent=1)
shutil.copy(os.path.join(P, "sq_latest.json"), os.path.join(P, "%s_sq.json" % tag))

# ---- HBM traffic
out = {}
raw = {}
for d in sorted(glob.glob(os.path.join(G, "traffic_%s*" % tag))):
    rf, cf = os.path.join(d, "run.json"), os.path.join(d, "cal.json")
    if not (os.path.exists(rf) and os.path.exists(cf)):
        continue
    what = os.path.basename(d)[len("traffic_%s" % tag):].strip("_")
    run, cal = json.load(open(rf)), json.load(open(cf))
    raw[what or "f32"] = dict(run={k: v for k, v in run.items() if "k_" in k}, calibration=cal)
    calk = [v for k, v in cal.items() if "k_read<HIP_vector_type<float, 2u>" in k][0]
    factor = (1 << 30) / (calk["FETCH_SIZE"]["mean"] * 1024.0)              # 8-byte-per-lane reads of 1 GiB
    wcal = [v for k, v in cal.items() if "k_write<double>" in k][0]
    wfactor = (1 << 30) / (wcal["WRITE_SIZE"]["mean"] * 1024.0)
    for k, v in run.items():
        if "k_" not in k or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        rd = v["FETCH_SIZE"]["mean"] * 1024.0 * factor
        wr = v["WRITE_SIZE"]["mean"] * 1024.0 * wfactor
        name = short_name(k) + ("_" + what if what else "")
        out[name] = dict(bytes=int(rd + wr), read_bytes=int(rd), write_bytes=int(wr), bytes_per_frame=round((rd + wr) / FRAMES, 1),
                         frames_per_launch=FRAMES, symbol=k, csrc_sha16=SHA, tag=tag, fetch_calibration_factor=round(factor, 4),
                         write_calibration_factor=round(wfactor, 4), launches=int(v["FETCH_SIZE"]["n"]))
out["_method"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/prof_traffic.sh) over tools/run_mode.py / "
                  "tools/synth_time.py on BASELINE config 2; FETCH_SIZE calibrated on tools/fetch_calib (1 GiB streamed with 8-byte-per-lane "
                  "loads reports half of the bytes on gfx950), WRITE_SIZE reads exact; earlier rounds: git log -- profiles/traffic_latest.json")
json.dump(out, open(os.path.join(P, "traffic_latest.json"), "w"), indent=1)
json.dump(raw, open(os.path.join(P, "%s_traffic_pmc.json" % tag), "w"), indent=1)

st = os.path.join(G, "stats_%s" % tag, "r_kernel_stats.csv")
if os.path.exists(st):
    shutil.copy(st, os.path.join(P, "%s_kernel_stats.csv" % tag))
for src, dst in (("bench_%sa.json" % tag, "%s_bench.json" % tag), ("bench_c4.json", "%s_bench_c4.json" % tag), ("bench_fg.json", "%s_bench_forced_gather.json" % tag)):
    s = os.path.join(G, src)
    if os.path.exists(s):
        lines = [ln for ln in open(s).read().splitlines() if ln.startswith("{")]
        if lines:
            open(os.path.join(P, dst), "w").write(lines[-1] + "\n")
    # the full record bench.py wrote beside its compact stdout line (--detail): one line of JSON in profiles/
    sd = os.path.join(G, src.replace(".json", "_detail.json"))
    if os.path.exists(sd):
        open(os.path.join(P, dst.replace(".json", "_detail.json")), "w").write(json.dumps(json.load(open(sd))) + "\n")
for src, dst in (("configs_%s.jsonl" % tag, "%s_configs_3_4.jsonl" % tag), ("sweep5_%s.jsonl" % tag, "%s_config5_sweep.jsonl" % tag),
                 ("sweep5_%s_f64.jsonl" % tag, "%s_config5_sweep_f64.jsonl" % tag), ("ab_nfft_%s.jsonl" % tag, "%s_nfft_harmonic_vs_noise.jsonl" % tag)):
    s = os.path.join(G, src)
    if os.path.exists(s) and os.path.getsize(s):
        shutil.copy(s, os.path.join(P, dst))
# the "next" rows (tools/bench_next_rows.py under rocprofv3 --kernel-trace --stats): its lines + the durations of its kernels
s = os.path.join(G, "next_rows_%s.jsonl" % tag)
if os.path.exists(s) and os.path.getsize(s):
    lines = [ln for ln in open(s).read().splitlines() if ln.startswith("{")]
    ks = os.path.join(G, "next_rows_%s" % tag, "r_kernel_stats.csv")
    if os.path.exists(ks):
        import csv
        for r in csv.DictReader(open(ks)):
            if any(k in r["Name"] for k in ("k_harmonic_rows", "k_stft<", "k_heterodyne", "k_rms_frames", "k_hpower", "k_f0")):
                lines.append(json.dumps({"kernel": r["Name"].replace("(anonymous namespace)::", "").split("(")[0].strip(), "calls": int(r["Calls"]),
                                         "avg_us": round(float(r["AverageNs"]) / 1e3, 1)}))
    open(os.path.join(P, "%s_next_rows.jsonl" % tag), "w").write("\n".join(lines) + "\n")
print(json.dumps({k: v["bytes_per_frame"] for k, v in out.items() if isinstance(v, dict)}))
