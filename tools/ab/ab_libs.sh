#!/bin/bash
# interleaved A/B of library builds on the nfft-2048 sweep point:   bash tools/ab/ab_libs.sh OUT.jsonl ROUNDS name1 name2 ...   (tools/ab/libpvx_NAME.so)
OUT=$1; ROUNDS=$2; shift 2
: > "$OUT"
for r in $(seq 1 "$ROUNDS"); do
  for n in "$@"; do
    PVX_ALLOW_STALE_LIB=1 PVX_LIB=tools/ab/libpvx_$n.so python tools/ab_nfft.py ${AB_NFFT:-2048} ${AB_PREC:-32} 2>/dev/null | sed "s/^{/{\"lib\": \"$n\", /" >> "$OUT"
  done
done
python - "$OUT" <<'PY'
import json, sys, collections
d = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    j = json.loads(ln); d[(j["lib"], j["nfft"], j["input"])].append((j["Mframes_per_s"], j["checksum"]))
for k, v in sorted(d.items(), key=lambda kv: (kv[0][1], kv[0][2], kv[0][0])):
    print("%-10s %5d %-9s  %s   checksum %r" % (k[0], k[1], k[2], " ".join("%6.1f" % a for a, _ in v), v[0][1]))
PY
