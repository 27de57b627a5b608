#!/bin/bash
# float64 analysis of BASELINE config 2 (k_stft_pv<16, double>) with and without its spectrum-row stores, interleaved rounds,
# each timed by rocprofv3's kernel trace: what would a kernel that kept the rows on chip gain at most?
#   bash tools/ab/build_variant.sh norow "-DPVX_AB_NO_ROWSTORE=1" k_stft_pv && bash tools/ab/f64_rowstore_ab.sh
cd "$(dirname "$0")/../.." && ROOT=$PWD
export TMPDIR=/tmp
for round in 1 2 3; do
  for v in base norow; do
    if [ "$v" = base ]; then unset PVX_LIB; else export PVX_LIB=$ROOT/tools/ab/libpvx_$v.so; fi
    (cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/f64ab_${v}_$round -o r --output-format csv -- python3 $ROOT/tools/run_mode.py -1 harmonic 8 40 64 > /dev/null 2>&1)
    echo "round $round $v: $(python3 tools/ab/ktimes.py gpurun_out/f64ab_${v}_$round/r_kernel_trace.csv 'k_stft_pv<[^>]*>' | tr -s ' ')"
  done
done
