# k_synth_ola's time on config 2 (whole waveform in one launch), library variants:   bash tools/ab/synab.sh
cd /tmp && export TMPDIR=/tmp
export PVX_NO_SYNTH_SLICES=1
for v in base synS2w4 synS2w5 synS4w4 synS4w5; do
  if [ $v = base ]; then unset PVX_LIB; else export PVX_LIB=/root/repo/tools/ab/libpvx_$v.so; fi
  rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/synab_$v -o r --output-format csv -- python3 /root/repo/tools/run_chain.py 4 > /root/repo/gpurun_out/synab_$v.log 2>&1 || exit 1
  echo "$v: $(grep k_synth_ola /root/repo/gpurun_out/synab_$v/r_kernel_stats.csv | cut -d, -f2-4)"
done
