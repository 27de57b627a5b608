#!/bin/bash
# The 1 / 2 / 4 / 8-GPU curve of bench.py on ONE node, both workloads (c2: one 10-min signal per GPU; c4: 128 x 30-s signals per
# GPU, BASELINE config 4), as one JSON file:
#   bash tools/scale.sh [OUT.json] [STEPS] [WARMUP]          (from the repo root; needs as many GPUs as the largest N it can run)
# Per N and workload: value (whole-job frames/s), ms_per_step (max over ranks), per_rank_ms_per_step, the gather's exposed
# time per step, wire bytes per rank and the RCCL world size, and the scaling efficiency against N x the N = 1 value.
# bench.py starts its own ranks (python -m torch.distributed.run, 127.0.0.1); an N beyond the node's GPUs is recorded as skipped.
set -u
cd "$(dirname "$0")/.." || exit 1
OUT=${1:-gpurun_out/scale.json}; STEPS=${2:-20}; WARMUP=${3:-5}
mkdir -p "$(dirname "$OUT")"
NG=$(python3 -c 'import torch; print(torch.cuda.device_count())')
TMP=$(mktemp -d)
for WL in c2 c4; do
  for N in 1 2 4 8; do
    if [ "$N" -gt "$NG" ]; then echo "{\"workload\": \"$WL\", \"n_gpus\": $N, \"skipped\": \"the node has $NG GPU(s)\"}" > "$TMP/$WL.$N.json"; continue; fi
    EXTRA="--no-extras --no-cpu-baseline"
    python3 bench.py --gpus "$N" --workload "$WL" --steps "$STEPS" --warmup "$WARMUP" $EXTRA > "$TMP/$WL.$N.out" 2> "$TMP/$WL.$N.err"
    RC=$?
    grep '^{' "$TMP/$WL.$N.out" | tail -1 > "$TMP/$WL.$N.json"
    if [ "$RC" -ne 0 ] || [ ! -s "$TMP/$WL.$N.json" ]; then
      python3 - "$WL" "$N" "$RC" "$TMP/$WL.$N.err" > "$TMP/$WL.$N.json" <<'PY'
import json, sys
print(json.dumps(dict(workload=sys.argv[1], n_gpus=int(sys.argv[2]), failed=int(sys.argv[3]), stderr_tail=open(sys.argv[4]).read()[-1500:])))
PY
    fi
  done
done
python3 - "$TMP" "$OUT" <<'PY'
import json, os, sys
tmp, out = sys.argv[1], sys.argv[2]
res = {}
for wl in ("c2", "c4"):
    rows, base = [], None
    for n in (1, 2, 4, 8):
        j = json.load(open(os.path.join(tmp, "%s.%d.json" % (wl, n))))
        if "value" not in j:
            rows.append(j)
            continue
        g = j.get("gather") or {}
        r = dict(n_gpus=n, value=j["value"], unit=j["unit"], ms_per_step=j["ms_per_step"], scaling=j.get("scaling"),
                 per_rank_ms_per_step=j.get("per_rank_ms_per_step"), gather_exposed_ms_per_step=g.get("exposed_ms_per_step"),
                 gather_collective=g.get("collective"), wire_bytes_per_rank=g.get("wire_bytes_per_rank"), rccl_world=g.get("rccl_world"),
                 config=j.get("config"))
        if n == 1:
            base = j["value"]
        if base:
            r["efficiency_vs_n_times_1gpu"] = round(j["value"] / (n * base), 4)
        rows.append(r)
    res[wl] = rows
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({wl: [(r.get("n_gpus"), r.get("value", r.get("skipped", r.get("failed")))) for r in rows] for wl, rows in res.items()}))
PY
rm -rf "$TMP"
