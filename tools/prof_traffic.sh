#!/bin/bash
# HBM traffic of the analysis kernels: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md, HBM section) over tools/run_mode.py, plus the same two passes over tools/fetch_calib
# (known byte counts: FETCH_SIZE under-reports wide reads by 2x on gfx950).
#   bash tools/prof_traffic.sh OUTDIR [MODE] [harmonic|noise] [K] [precision]     (on the GPU box, from the repo root)
set -u
OUT=$1; MODE=${2:--1}; KIND=${3:-harmonic}; K=${4:-8}; PREC=${5:-32}
PROG=${PVX_PROF_PROG:-tools/run_mode.py $MODE $KIND $K 10 $PREC}      # PVX_PROF_PROG: another program (e.g. "tools/run_chain.py 3")
export TMPDIR=/tmp
mkdir -p "$OUT"
[ -x tools/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d "$OUT/run_$C" -o r --output-format csv -- python3 $PROG > "$OUT/run_$C.log" 2>&1
  rocprofv3 --pmc $C -d "$OUT/cal_$C" -o r --output-format csv -- tools/fetch_calib > "$OUT/cal_$C.log" 2>&1
done
python3 tools/pmc_summary.py "$OUT/run_FETCH_SIZE" "$OUT/run_WRITE_SIZE" > "$OUT/run.json"
python3 tools/pmc_summary.py "$OUT/cal_FETCH_SIZE" "$OUT/cal_WRITE_SIZE" > "$OUT/cal.json"
