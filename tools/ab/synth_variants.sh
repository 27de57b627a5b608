#!/bin/bash
# per-kernel time of the resynthesis on config 2 for several builds of k_synth.hip (tools/ab/build_variant.sh):
#   bash tools/ab/synth_variants.sh base rep2 nostore ...     (on the GPU box, from the repo root; libs tools/ab/libpvx_NAME.so)
cd "$(dirname "$0")/../.." && ROOT=$PWD
export TMPDIR=/tmp SYNTH_TIME_NOCHECK=1 SYNTH_TIME_ONLY=harmonic SYNTH_TIME_F32=${SYNTH_TIME_F32-1}
for v in "$@"; do
  if [ "$v" = base ]; then unset PVX_LIB; else export PVX_LIB=$ROOT/tools/ab/libpvx_$v.so; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/sv_$v -o r --output-format csv -- python3 $ROOT/tools/synth_time.py 600 50 > $ROOT/gpurun_out/sv_$v.log 2>&1) || { echo "$v failed"; tail -3 gpurun_out/sv_$v.log; continue; }
  echo "== $v: $(grep -o '"ms": [0-9.]*' gpurun_out/sv_$v.log | head -1)"
  python3 tools/ab/ktimes.py gpurun_out/sv_$v/r_kernel_trace.csv
done
