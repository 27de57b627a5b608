# k_synth_ola's time on config 2 (whole waveform in one launch):   bash tools/ab/synab.sh
cd /tmp && export TMPDIR=/tmp
export PVX_NO_SYNTH_SLICES=1
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/synab -o r --output-format csv -- python3 /root/repo/tools/run_chain.py 4 > /root/repo/gpurun_out/synab.log 2>&1 || exit 1
grep "k_synth_ola\|k_track\|k_fused\|k_assign" /root/repo/gpurun_out/synab/r_kernel_stats.csv | cut -d, -f1-4
