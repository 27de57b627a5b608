"""bench.py's stdout contract without a GPU: whatever the full record holds, the ONE stdout line stays small enough for the
driver's record (round 4: a 20.8 KB line was not parsed and the round had no recorded measurement)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _full(world=1):
    chk = dict(ok=True, frames=51676, frames_with_other_peaks=0, ref_peaks=413408, bad_peaks=0, f_abs_Hz=7.917957191239111e-06,
               mag_rel=2.5e-7, normalised=dict(note="x" * 200))
    pts = [dict(nfft=n, hop=h, value=4.5e8, vs_contract_target=3.3 - 0.1 * i, self_check=dict(chk), note="y" * 300)
           for i, (n, h) in enumerate((n, h) for n in (512, 1024, 2048, 4096, 8192) for h in (n // 4, n // 2))]
    return {
        "metric": "STFT frames/sec (44.1 kHz, nfft=2048, hop=512)", "value": 411565010.4, "unit": "frames/s", "value_from_idle": 363972652.1,
        "n_gpus": world, "steps": 20, "warmup": 5, "ms_per_step": 0.1256, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "long " * 60, "workload_short": "BASELINE config 2: one 600-s 44.1 kHz mono signal per GPU, PV.run_pv",
                   "nfft": 2048, "hop": 512, "npks": 8, "sr": 44100, "frames_per_gpu": 51676, "signals_per_gpu": 1, "parallelism": "p" * 100,
                   "parallelism_short": "single GPU", "streams": 1},
        "roofline": dict(bound="hbm", kernel="k_fused_rev", achieved=977.9, peak=8000.0, unit="GB/s", frac=0.1222, traffic=169114052,
                         traffic_provenance=dict(note="z" * 500), traffic_over_algorithmic=1.373, ms_per_launch=0.126, alg_bytes_per_frame=2384,
                         throughput_vs_60pct_target=3.015, note="n" * 400, issue=dict(valu_insts_per_frame=928.0, note="i" * 900)),
        "stage": dict(kernels=[dict(note="k" * 100)] * 4),
        "cpu_baseline": dict(value=604509.9, unit="frames/s", cores=256, kind="port", sample="s" * 300,
                             single_thread=dict(value=37035.5, cores=1, seconds=1.4), reference_python=dict(value=3518.0, host="h" * 100)),
        "self_check": chk, "per_rank_ms_per_step": [0.1256] * world, "clock_warmup": {"ms": 300.0, "untimed_steps": 2432, "note": "c" * 400},
        "f64": dict(value=1.44e8, self_check=dict(chk), roofline=dict(frac=0.339, kernels=[dict(note="q" * 200)])),
        "workloads": dict(white_noise=dict(value=3.5e8, self_check=dict(chk)), violin_g7_tiled=dict(value=3.5e8, self_check=dict(chk))),
        "other_nfft": {"4096": dict(value=1.78e8, self_check=dict(chk)), "8192": dict(value=7.3e7, self_check=dict(chk))},
        "chain": dict(total_ms=0.4039, tracker=dict(ms=0.0579, what="w" * 200), resynthesis=dict(ms=0.1662, what="w" * 200)),
        "config5": dict(signal="s" * 100, points=pts), "host_batch": dict(value=1.2e7, what="w" * 300), "extras_ok": True,
    }


def test_the_stdout_line_is_compact_and_carries_the_contract():
    full = _full()
    assert len(json.dumps(full)) > 10000
    s = bench.compact_line(full, "gpurun_out/bench_detail.json")
    assert len(s) <= bench.COMPACT_MAX < 2000 and "\n" not in s
    j = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "self_check", "value_from_idle"):
        assert k in j, k
    assert j["config"]["workload"].startswith("BASELINE config 2") and "model" not in j["config"]
    r = j["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r) and r["valu_per_frame"] == 928.0
    c = j["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["single_thread"] == 37035.5
    assert j["self_check"] == dict(ok=True, frames=51676, bad_peaks=0, f_abs_Hz=7.92e-06)
    assert j["c5_all_ok"] and abs(j["c5_min_vs_target"] - 2.4) < 1e-9 and j["c5_2048_512_value"] == 4.5e8 and j["extras_ok"]
    # nothing nested beyond the contract objects
    assert all(not isinstance(v, (dict, list)) or k in ("config", "roofline", "cpu_baseline", "self_check", "per_rank_ms_per_step") for k, v in j.items())


def test_the_stdout_line_of_a_multi_rank_run_and_of_a_bare_run():
    full = _full(world=8)
    for k in ("f64", "workloads", "other_nfft", "chain", "config5", "host_batch", "extras_ok", "cpu_baseline", "self_check"):
        full.pop(k)
    full["cpu_baseline"] = full["self_check"] = None
    full["gather"] = dict(collective="one asynchronous RCCL gather per step to rank 0, double-buffered", unpack="u" * 200, rccl_world=8,
                          wire_bytes_per_rank=54600000, result_bytes_per_rank=117900000, valid_peaks_gathered=123456, exposed_ms_per_step=0.01, note="n" * 100)
    j = json.loads(bench.compact_line(full, None))
    assert j["cpu_baseline"] is None and j["self_check"] is None and j["detail"] is None
    assert len(j["per_rank_ms_per_step"]) == 8 and j["gather"]["rccl_world"] == 8 and "unpack" not in j["gather"]
    # a failed config-5 point shows on the line
    full = _full()
    full["config5"]["points"][3]["self_check"]["ok"] = False
    assert json.loads(bench.compact_line(full, "d.json"))["c5_all_ok"] is False
