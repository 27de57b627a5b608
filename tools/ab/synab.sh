# k_synth_ola's time on config 2 (whole waveform in one launch) for several workgroup sizes:   bash tools/ab/synab.sh
cd /tmp && export TMPDIR=/tmp
export PVX_NO_SYNTH_SLICES=1
for v in 128 64; do
  export PVX_SYNTH_THREADS=$v
  rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/synab_$v -o r --output-format csv -- python3 /root/repo/tools/run_chain.py 4 > /root/repo/gpurun_out/synab_$v.log 2>&1 || exit 1
  echo "$v threads: $(grep k_synth_ola /root/repo/gpurun_out/synab_$v/r_kernel_stats.csv | cut -d, -f2-4)"
done
