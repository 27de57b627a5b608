# per-kernel durations of a program on the GPU box: rocprofv3 --kernel-trace --stats; prints the stats csv's head
#   bash tools/prof_kernels.sh OUTDIR python3 tools/synth_time.py 600 3
OUT=$1; shift
mkdir -p /root/repo/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/$OUT -o r --output-format csv -- "$@" > /root/repo/gpurun_out/$OUT/run.log 2>&1
cd /root/repo
f=$(ls gpurun_out/$OUT/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cut -c1-200 "$f" | head -${PROF_TOP:-14}
rm -f gpurun_out/$OUT/*kernel_trace.csv gpurun_out/$OUT/*agent_info.csv
