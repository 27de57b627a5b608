#!/bin/bash
# total resynthesis time on the three signals of tools/synth_time.py for several builds:  bash tools/ab/synth_variants_all.sh base NAME ...
cd "$(dirname "$0")/../.." && ROOT=$PWD
export SYNTH_TIME_NOCHECK=1
for v in "$@"; do
  if [ "$v" = base ]; then unset PVX_LIB; else export PVX_LIB=$ROOT/tools/ab/libpvx_$v.so; fi
  echo "== $v: $(python3 tools/synth_time.py 600 300 2>&1 | grep -o '"signal": "[a-z_0-9]*"\|"ms": [0-9.]*' | tr '\n' ' ')"
done
