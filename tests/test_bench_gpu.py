"""bench.py end to end on the GPU box (short signal): the JSON contract, the oracle self-check of the timed output,
the RCCL gather path forced on one GPU, and the refusal of more ranks than GPUs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, universal_newlines=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    # the driver's record keeps a short tail of stdout: exactly ONE JSON line, and a small one (round 4's 20.8 KB line was lost)
    assert len(lines) <= 1 and all(len(ln) < 2000 for ln in lines), [len(ln) for ln in lines]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def _detail(j):
    """The full record bench.py wrote beside the compact stdout line."""
    assert j.get("detail"), j
    with open(os.path.join(ROOT, j["detail"])) as fh:
        return json.load(fh)


def test_bench_line_and_self_check():
    rc, j, err = _run(["--steps", "3", "--warmup", "1", "--seconds", "20", "--no-extras", "--detail", "gpurun_out/test_bench_detail_a.json"])
    assert rc == 0 and j is not None, err[-2000:]
    for k in ("metric", "value", "value_from_idle", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "self_check"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["dtype"] == "f32" and j["vs_baseline"] is None
    assert set(j["config"]) == {"workload", "nfft", "hop", "npks", "sr", "frames_per_gpu", "signals_per_gpu", "parallelism"}
    r = j["roofline"]
    assert r["bound"] == "hbm" and 0.0 < r["frac"] <= 1.0 and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3 and r["ms_per_launch"] > 0 and r["alg_bytes_per_frame"] == 2384
    assert j["self_check"]["ok"] and j["self_check"]["bad_peaks"] == 0 and j["self_check"]["frames"] == j["config"]["frames_per_gpu"]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["single_thread"] > 0 and c["value"] > 0
    # everything but scalars is in the detail file; the compact line is a projection of it
    d = _detail(j)
    assert d["value"] == j["value"] and d["roofline"]["frac"] == r["frac"] and d["cpu_baseline"]["single_thread"]["cores"] == 1
    assert all(not isinstance(v, (dict, list)) or k in ("config", "roofline", "cpu_baseline", "self_check", "per_rank_ms_per_step", "gather")
               for k, v in j.items()), j
    if (d["roofline"].get("issue") or {}).get("valu_issue"):            # quoted only when profiles/sq_latest.json is of these sources
        assert 0.0 < d["roofline"]["issue"]["valu_issue"]["frac"] <= 1.0


def test_bench_extras_are_all_checked_against_the_oracle():
    """The default line's extra objects on a short signal: float64, the other material, nfft 4096 / 8192, and the rest of
    the path (tracker + resynthesis on the headline results) -- every one of them carries an oracle check that passed."""
    rc, jc, err = _run(["--steps", "3", "--warmup", "1", "--seconds", "30", "--c5-seconds", "20", "--detail", "gpurun_out/test_bench_detail_b.json"])
    assert rc == 0 and jc is not None, err[-2000:]
    assert jc["extras_ok"] and jc["c5_all_ok"] and jc["f64_ok"] and jc["f64_value"] > 0 and jc["chain_total_ms"] > 0 and jc["c5_min_vs_target"] > 0
    j = _detail(jc)
    assert j["chain"]["total_ms"] == jc["chain_total_ms"] and j["f64"]["value"] == jc["f64_value"]
    assert j["f64"]["self_check"]["ok"] and j["f64"]["white_noise"]["self_check"]["ok"]
    assert all(w["self_check"]["ok"] for k, w in j["workloads"].items() if k != "reference_defaults")
    rd = j["workloads"]["reference_defaults"]                       # PV(x, sr): nfft 1024, hop 512, npks 20 at both precisions
    assert rd["f32"]["self_check"]["ok"] and rd["f64"]["self_check"]["ok"] and rd["f32"]["value"] > rd["f64"]["value"] > 0
    assert jc["defaults_f32_value"] == rd["f32"]["value"] and jc["defaults_f64_value"] == rd["f64"]["value"]
    assert all(w["self_check"]["ok"] for w in j["other_nfft"].values())
    c = j["chain"]
    assert c["tracker"]["check"]["ok"] and c["tracker"]["partials"] > 0 and c["tracker"]["value"] > 0
    # (the headline is precision 32: its resynthesis runs the float32 sample loop, stated tolerance 1e-4 max|w|; measured ~1e-5)
    assert c["resynthesis"]["sample_loop"] == "f32" and c["resynthesis"]["check"]["ok"] and c["resynthesis"]["value"] > 0
    assert c["resynthesis"]["check"]["max_abs_err"] <= c["resynthesis"]["check"]["tolerance"] <= 1e-4
    assert c["total"]["ms"] >= c["tracker"]["ms"] + c["resynthesis"]["ms"] and c["total_ms"] == c["total"]["ms"]
    # BASELINE config 5 (here on a 20-s signal): all ten (nfft, hop) points, each timed, priced against the contract
    # target and checked against the oracle
    pts = j["config5"]["points"]
    assert sorted((p["nfft"], p["hop"]) for p in pts) == sorted((n, h) for n in (512, 1024, 2048, 4096, 8192) for h in (n // 4, n // 2))
    for p in pts:
        assert p["self_check"]["ok"] and p["self_check"]["bad_peaks"] == 0 and p["self_check"]["ref_peaks"] > 0, p
        assert p["value"] > 0 and p["contract_target"] > 0 and p["fft_mode"] in (4, 5)
    # the C-level batch entry on host signals: PCIe-inclusive, never `value`; identical to the single-signal calls
    hb = j["host_batch"]
    assert hb["identical_to_single_signal_calls"] and hb["pcie_inclusive"] and hb["value"] > 0 and hb["frames"] > 0


def test_bench_forced_gather_exercises_rccl():
    rc, j, err = _run(["--steps", "4", "--warmup", "1", "--seconds", "20", "--no-extras", "--no-cpu-baseline"],
                      env={"PVX_BENCH_FORCE_GATHER": "1", "MASTER_PORT": "29641"})
    assert rc == 0 and j is not None, err[-2000:]
    g = j["gather"]
    assert g["rccl_world"] == 1 and g["valid_peaks_gathered"] > 0 and g["wire_bytes_per_rank"] < g["result_bytes_per_rank"]


def test_bench_self_launch_c4_gather_matches_oracle():
    """The path `bench.py --gpus 8 --workload c4` takes on an 8-GPU node, with the one GPU this box has: bench.py starts
    its ranks itself (python -m torch.distributed.run as a child process, rendezvous on a port the kernel handed out),
    every step ends in the RCCL gather of the packed result block (forced at world size 1), and what rank 0 unpacked is
    checked against the oracle on the signals themselves."""
    env = {"PVX_BENCH_SELF_LAUNCH": "1", "PVX_BENCH_FORCE_GATHER": "1"}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        os.environ.pop(k, None)
    rc, j, err = _run(["--gpus", "1", "--workload", "c4", "--shard-signals", "3", "--seconds", "5", "--steps", "3", "--warmup", "1",
                       "--check-gathered", "3"], env=env)
    assert rc == 0 and j is not None, err[-3000:]
    g = j["gather"]
    assert g["rccl_world"] == 1 and g["checked_signals"] == 3 and g["check_ok"] and g["valid_peaks_gathered"] > 0
    assert j["config"]["signals_per_gpu"] == 3 and j["n_gpus"] == 1 and len(j["per_rank_ms_per_step"]) == 1
    # a failing rank fails the launcher: an impossible workload argument inside the child
    rc, j, err = _run(["--gpus", "1", "--workload", "c4", "--shard-signals", "3", "--seconds", "5", "--steps", "1", "--warmup", "0", "--precision", "32",
                       "--fft-mode", "0"], env=env)       # the general path exists at every size: fine -> rc 0 (the launcher passes the line through)
    assert rc == 0 and j is not None, err[-2000:]
    rc, j, err = _run(["--gpus", "1", "--workload", "c4", "--shard-signals", "3", "--seconds", "5", "--steps", "1", "--warmup", "0", "--fft-mode", "5"], env=env)
    assert rc != 0 and j is None and "exited with code" in err     # mode 5 does not exist at nfft 2048: the child fails, so does the parent


def test_bench_refuses_more_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count()
    rc, j, err = _run(["--gpus", str(n + 1), "--steps", "1", "--warmup", "0"], timeout=120)
    assert rc != 0 and j is None and "GPU" in err
