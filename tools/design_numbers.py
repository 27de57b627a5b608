#!/usr/bin/env python3
"""Regenerate the measured-numbers block of DESIGN.md (between the NUMBERS markers) from the committed profiles, so that the
prose cannot drift from the evidence:   python tools/design_numbers.py [tag]        (default tag: r04)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
prev = "r%02d" % (int(tag[1:]) - 1)


def jl(path):
    out = []
    try:
        for ln in open(os.path.join(P, path)):
            ln = ln.strip()
            if ln.startswith("{"):
                out.append(json.loads(ln))
    except Exception:
        pass
    return out


def M(v):
    if v is None:
        return "–"
    return "%.1f M" % (v / 1e6) if v >= 1e6 else "%.1f k" % (v / 1e3)


L = []
b = (jl("%s_bench_detail.json" % tag) or jl("%s_bench.json" % tag) or [None])[-1]      # the full record (bench.py --detail); rounds 1-4: the stdout line
if b:
    r = b["roofline"]
    L.append("**`bench.py`, N = 1 (`profiles/%s_bench.json`; BASELINE config 2: 10 min @ 44.1 kHz, nfft 2048, hop 512, npks 8, F = 51 676).**" % tag)
    L.append("")
    L.append("| line | frames/s | notes |")
    L.append("|---|---|---|")
    L.append("| `value` (float32, `%s`) | **%s** | %.4f ms per step; from idle: %s; %.2f × north_star's contract-priced target |" %
             (r["kernel"], M(b["value"]), b["ms_per_step"], M(b.get("value_from_idle")), r["throughput_vs_60pct_target"]))
    tr = r.get("traffic")
    L.append("| `roofline` | frac **%.4f** of 8 TB/s | %.1f GB/s at %d algorithmic B/frame, %.4f ms per launch; measured HBM traffic %s |" %
             (r["frac"], r["achieved"], r["alg_bytes_per_frame"], r["ms_per_launch"],
              ("%.1f MB per launch = %.3f × algorithmic" % (tr / 1e6, r["traffic_over_algorithmic"])) if tr else "not quoted (profile of other sources)"))
    iss = r.get("issue") or {}
    if iss.get("valu_issue"):
        w = iss["wave_time_split"]
        L.append("| `roofline.issue` | vector issue %.2f (%.2f at measured costs) | %d VALU instructions per frame; LDS pipe %.0f %% busy; wave time %d %% issuing / %d %% waiting / %d %% issue-stalled |" %
                 (iss["valu_issue"]["frac"], iss["valu_issue"]["frac_with_measured_costs"], round(iss["valu_insts_per_launch"] / 51676.0),
                  100 * iss["lds_busy_frac"], round(100 * w["issuing"]), round(100 * w["waiting_on_memory_or_lds"]), round(100 * w["issue_stalled"])))
    sc = b.get("self_check") or {}
    L.append("| `self_check` | ok = %s | %d peaks, %d missed, \\|Δf\\| ≤ %.1e Hz, \\|Δmag\\|/mag ≤ %.1e |" % (sc.get("ok"), sc.get("ref_peaks", 0), sc.get("bad_peaks", 0), sc.get("f_abs_Hz", 0), sc.get("mag_rel", 0)))
    f = b.get("f64")
    if f:
        k = f["roofline"]["kernels"]
        L.append("| `f64` (`%s`) | %s | %.4f ms per step; self-check ok = %s (\\|Δf\\| ≤ %.1e Hz); white noise %s (ok = %s); stage frac %.3f of HBM at %d B/frame; traffic %s |" %
                 (k[-1]["kernel"] if k else "?", M(f["value"]), f["ms_per_step"], (f.get("self_check") or {}).get("ok"), (f.get("self_check") or {}).get("f_abs_Hz", 0),
                  M(f["white_noise"]["value"]), (f["white_noise"].get("self_check") or {}).get("ok"), f["roofline"]["frac"], f["roofline"]["stage_alg_bytes_per_frame"],
                  ("%.0f MB per launch" % (f["roofline"]["traffic"] / 1e6)) if f["roofline"].get("traffic") else "not quoted"))
    for name, w in (b.get("workloads") or {}).items():
        if name == "reference_defaults":
            a32, a64 = w.get("f32") or {}, w.get("f64") or {}
            L.append("| `workloads.reference_defaults` (%s) | f32 %s, f64 %s | the recording: %s / %s; self-checks ok = %s / %s (\\|Δf\\| ≤ %.1e / %.1e Hz) |" %
                     (w.get("config"), M(a32.get("value", 0)), M(a64.get("value", 0)), M((a32.get("violin_g7_tiled") or {}).get("value", 0)), M((a64.get("violin_g7_tiled") or {}).get("value", 0)),
                      (a32.get("self_check") or {}).get("ok"), (a64.get("self_check") or {}).get("ok"), (a32.get("self_check") or {}).get("f_abs_Hz", 0), (a64.get("self_check") or {}).get("f_abs_Hz", 0)))
            continue
        s = w.get("self_check") or {}
        sh = (s.get("normalised") or {}).get("share_of_peaks_within_f32_tolerance") or {}
        L.append("| `workloads.%s` | %s | %.2f peaks per frame; check ok = %s: %d of %d frames with another peak set, %.2f %% / %.2f %% of the peaks within the f / realph tolerance |" %
                 (name, M(w["value"]), w["peaks_per_frame"], s.get("ok"), s.get("frames_with_other_peaks", 0), s.get("frames", 0), 100 * sh.get("f", 0), 100 * sh.get("realph", 0)))
    for nf, w in (b.get("other_nfft") or {}).items():
        s = w.get("self_check") or {}
        L.append("| `other_nfft.%s` (fft mode %d) | %s | white noise %s; contract-priced target %s; self-check ok = %s (%d peaks, %d missed) |" %
                 (nf, w["fft_mode"], M(w["value"]), M(w["white_noise"]["value"]), M(w["contract_target"]), s.get("ok"), s.get("ref_peaks", 0), s.get("bad_peaks", 0)))
    ch = b.get("chain") or {}
    if ch.get("tracker"):
        t_, r_ = ch["tracker"], ch.get("resynthesis") or {}
        L.append("| `chain.tracker` (`k_track.hip`) | %s | %.3f ms (wall time of `pvx_track_dev`), %d partials; table identical to the oracle's: %s |" %
                 (M(t_["value"]), t_["ms"], t_["partials"], (t_.get("check") or {}).get("ok")))
        if r_:
            L.append("| `chain.resynthesis` (`k_synth.hip`) | %s | %.3f ms per call, %.1f M samples out (%.0f GB/s written); first 3000 frames against the oracle: ok = %s, \\|Δw\\| ≤ %.1e |" %
                     (M(r_["value"]), r_["ms"], r_["samples_out"] / 1e6, r_["output_GBps"], (r_.get("check") or {}).get("ok"), (r_.get("check") or {}).get("max_abs_err", 0)))
        if ch.get("total"):
            L.append("| `chain.total` | %s | **%.3f ms**: one analysis step + `pvx_track_dev` + `pvx_synth_dev` back to back on the resident signal, wall clock (goal %.2f ms) |" %
                     (M(ch["total"]["value"]), ch["total"]["ms"], ch["total"].get("goal_ms", 0.55)))
    hb = b.get("host_batch")
    if hb:
        L.append("| `host_batch` (`pvx_batch_run`) | %s | PCIe-inclusive: %d ragged host signals (%.0f MB in), %d workers on device %s, %.4f s; identical to the single-signal calls: %s |" %
                 (M(hb["value"]), hb["signals"], hb["input_MB"], hb["workers_per_device"], hb["devices"], hb["seconds"], hb["identical_to_single_signal_calls"]))
    c = b.get("cpu_baseline")
    if c:
        L.append("| `cpu_baseline` | %s on %d threads | one thread %s; the Python reference %s (BASELINE.md) |" %
                 (M(c["value"]), c["cores"], M(c["single_thread"]["value"]), M(c["reference_python"]["value"])))
    L.append("")

c5 = (b or {}).get("config5")
if c5:
    L.append("**BASELINE config 5 on the bench line (`config5`: %s; every point HIP-event timed and its leading 60 s checked against the oracle).**" % c5["signal"])
    L.append("")
    L.append("| nfft / hop | fft mode | frames/s | vs contract target | frac of HBM at the fused bytes | oracle check (peaks, missed, \\|Δf\\| Hz) |")
    L.append("|---|---|---|---|---|---|")
    for d in c5["points"]:
        sc = d.get("self_check") or {}
        L.append("| %d / %d | %d | %s | %.2f × | %.3f | ok = %s (%d, %d, %.1e) |" % (d["nfft"], d["hop"], d["fft_mode"], M(d["value"]), d["vs_contract_target"],
                                                                                   d["frac_of_hbm_at_fused_bytes"], sc.get("ok"), sc.get("ref_peaks", 0), sc.get("bad_peaks", 0), sc.get("f_abs_Hz", 0)))
    L.append("")
sw, swp = [], []          # (the bench line's `config5` table above holds these ten points, each checked: `profiles/<tag>_config5_sweep.jsonl` is the same sweep by tools/sweep_config5.py)
if sw:
    pm = {(d["nfft"], d["hop"]): d for d in swp}
    L.append("**BASELINE config 5 (60 min @ 96 kHz, npks 8; `profiles/%s_config5_sweep.jsonl`, previous round beside it).**" % tag)
    L.append("")
    L.append("| nfft / hop | fft mode | frames/s | round before | vs contract target | frac of HBM at the fused bytes |")
    L.append("|---|---|---|---|---|---|")
    for d in sw:
        o = pm.get((d["nfft"], d["hop"]))
        L.append("| %d / %d | %d | %s | %s | %.2f × | %.3f |" % (d["nfft"], d["hop"], d["fft_mode"], M(d["frames_per_s"]), M(o["frames_per_s"]) if o else "–",
                                                               d["throughput_vs_60pct_target"], d["frac_of_8TBps"]))
    L.append("")
s64 = jl("%s_config5_sweep_f64.jsonl" % tag)
if s64:
    p64 = {(d["nfft"], d["hop"]): d for d in jl("%s_config5_sweep_f64.jsonl" % prev)}
    L.append("**Config 5 at float64** (`profiles/%s_config5_sweep_f64.jsonl`, M frames/s; round 5 in brackets): " % tag +
             ", ".join("%d/%d %s%s" % (d["nfft"], d["hop"], M(d["frames_per_s"]).replace(" M", ""), (" (%s)" % M(p64[(d["nfft"], d["hop"])]["frames_per_s"]).replace(" M", "")) if (d["nfft"], d["hop"]) in p64 else "") for d in s64) + ".")
    L.append("")
hn = []       # (`profiles/<tag>_nfft_harmonic_vs_noise.jsonl`: the nfft sweep on three kinds of material; the bench table above carries the noise / recording lines)
if hn:
    L.append("**10 min @ 44.1 kHz, hop = nfft/4, harmonic signal vs white noise (`profiles/%s_nfft_harmonic_vs_noise.jsonl`):** " % tag +
             "; ".join("nfft %d %s %s" % (d["nfft"], d["input"], M(d["Mframes_per_s"] * 1e6)) for d in hn) + ".")
    L.append("")
cf = jl("%s_configs_3_4.jsonl" % tag)
for d in cf:
    if d["config"].startswith("3") and d.get("precision") == 32:
        L.append("**Config 3** (perlmanVn.wav, nfft 4096, npks 100, Python API, host in / host out): run_pv %.3f + toSinSum %.3f + synth %.3f = **%.3f ms** (%.0f × real time), waveform within %.1e of the reference's." %
                 (d["run_pv_ms"], d["toSinSum_ms"], d["synth_ms"], d["round_trip_ms"], d["realtime_factor"], d["waveform_max_abs_err_vs_reference"]))
    if d["config"].startswith("2"):
        L.append("**Config 2, whole path** (host signal in, host waveform out): run_pv %.2f + toSinSum %.2f + synth %.2f = **%.2f ms** (106 MB in, 212 MB out over PCIe)." %
                 (d["run_pv_ms"], d["toSinSum_ms"], d["synth_ms"], d["whole_ms"]))
    if d["config"].startswith("4"):
        L.append("**Config 4 shard** (128 × 30 s @ 48 kHz in one call): %s frames/s; packing the shard for the gather %.3f ms (%.1f MB on the wire for %.1f MB of results)." %
                 (M(d["frames_per_s"]), d["pack_ms"], d["wire_MB"], d["result_MB"]))
if cf:
    L.append("")
nr = jl("%s_next_rows.jsonl" % tag)
if nr:
    L.append("**The rows SURVEY 8(f) marks next (N1-N4)** on config 2's signal through the Python mirrors, host arrays in and out unless resident (`profiles/%s_next_rows.jsonl`, best of 5): " % tag +
             "; ".join("%s %.3f ms (%s)" % (d["row"], d["ms"], M(d["frames_per_s"])) for d in nr if "row" in d and " cpu" not in d["row"]) + ".")
    L.append("")
try:
    sq = json.load(open(os.path.join(P, "sq_latest.json")))
    rows = []
    for k, v in sq.items():
        if not isinstance(v, dict) or "SQ_INSTS_VALU" not in v:
            continue
        what, name = k.split(" | ", 1)
        short = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", name)
        g = lambda c: v[c]["mean"] if c in v else float("nan")
        nf = 4096 if "4096" in what else 8192 if "8192" in what else 2048
        F = (44100 * 600 - nf + nf // 4 - 1) // (nf // 4)
        per = F if ("fused" in name or "stft" in name or "phase" in name or "k_pv_rev" in name or (what == "chain" and ("k_track" in name or "k_synth" in name or "k_assign" in name))) else None
        if per is None or (what == "chain" and "k_fused_rev" in name):      # (the chain's analysis kernel is the f32 line's)
            continue
        rows.append("| %s | `%s` | %.0f | %.0f | %.0f | %.2f / %.2f / %.2f | %.2f |" % (
            what, short.group(1)[:44] if short else name[:44], g("SQ_INSTS_VALU") / per, g("SQ_INSTS_SALU") / per, g("SQ_INSTS_LDS") / per,
            g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
            g("SQ_LDS_IDX_ACTIVE") / max(g("SQ_BUSY_CU_CYCLES"), 1)))
    if rows:
        L.append("**SQ counters per frame (`profiles/sq_latest.json`, rocprofv3 --pmc over the C2 signal; all waves of a frame summed).**")
        L.append("")
        L.append("| run | kernel | VALU | SALU | LDS | wave time issuing / waiting / issue-stalled | LDS pipe busy |")
        L.append("|---|---|---|---|---|---|---|")
        L += rows
        L.append("")
except Exception as e:
    L.append("(no sq_latest.json: %s)" % e)
try:
    tr = json.load(open(os.path.join(P, "traffic_latest.json")))
    items = ["`%s` %.1f MB = %.0f B/frame" % (k.replace("_chain", ""), v["bytes"] / 1e6, v["bytes_per_frame"]) for k, v in tr.items() if isinstance(v, dict) and "bytes" in v and k != "k_fused_rev_chain"]
    if items:
        L.append("**HBM traffic per C2 launch (`traffic_latest.json`: FETCH_SIZE × calibration + WRITE_SIZE):** " + "; ".join(items) + ".")
        L.append("")
except Exception:
    pass

path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
a = s.index("<!-- NUMBERS:BEGIN")
a = s.index("\n", a) + 1
b2 = s.index("<!-- NUMBERS:END -->")
s = s[:a] + "\n".join(L) + "\n" + s[b2:]
open(path, "w").write(s)
print("DESIGN.md: %d lines of numbers from profiles/%s_*" % (len(L), tag))
