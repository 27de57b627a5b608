# one GPU-box pass for a round: SQ counters, HBM traffic, kernel-trace stats, then the bench lines (which quote the
# traffic / SQ profiles just taken: collect_profiles.py runs on the box first), the other BASELINE configs.
# Outputs under gpurun_out/; afterwards, here: python tools/collect_profiles.py r05 && python tools/design_numbers.py r05
T=${1:-r06}
PART=${2:-ab}      # a: the counter / traffic / kernel-trace passes, collect, the bench line; b: the other bench lines and sweeps; ab: both
if [[ $PART == *a* ]]; then
bash tools/prof_sq.sh gpurun_out/sq_${T} -1; rm -rf gpurun_out/sq_${T}/g*/
bash tools/prof_sq.sh gpurun_out/sq_${T}_f64 -1 harmonic 8 64; rm -rf gpurun_out/sq_${T}_f64/g*/
PVX_RUN_NFFT=4096 bash tools/prof_sq.sh gpurun_out/sq_${T}_nfft4096 -1; rm -rf gpurun_out/sq_${T}_nfft4096/g*/
PVX_RUN_NFFT=8192 bash tools/prof_sq.sh gpurun_out/sq_${T}_nfft8192 -1; rm -rf gpurun_out/sq_${T}_nfft8192/g*/
# the rest of the path, device to device, whole launches: tools/synth_time.py (analysis, pvx_track_dev, pvx_synth_dev x 3 on the C2 signal)
SYNTH_TIME_F32=1 SYNTH_TIME_NOCHECK=1 SYNTH_TIME_ONLY=harmonic PVX_PROF_PROG="tools/synth_time.py 600 3" bash tools/prof_sq.sh gpurun_out/sq_${T}_chain -1; rm -rf gpurun_out/sq_${T}_chain/g*/
bash tools/prof_traffic.sh gpurun_out/traffic_${T}; rm -rf gpurun_out/traffic_${T}/*_SIZE/
bash tools/prof_traffic.sh gpurun_out/traffic_${T}_f64 -1 harmonic 8 64; rm -rf gpurun_out/traffic_${T}_f64/*_SIZE/
SYNTH_TIME_F32=1 SYNTH_TIME_NOCHECK=1 SYNTH_TIME_ONLY=harmonic PVX_PROF_PROG="tools/synth_time.py 600 3" bash tools/prof_traffic.sh gpurun_out/traffic_${T}_chain; rm -rf gpurun_out/traffic_${T}_chain/*_SIZE/
cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/stats_${T} -o r --output-format csv -- python3 /root/repo/bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --detail gpurun_out/bench_stats_detail.json > /root/repo/gpurun_out/stats_${T}.log 2>&1; cd /root/repo; ls gpurun_out/stats_${T}
python tools/collect_profiles.py ${T} > gpurun_out/collect_${T}.log 2>&1; echo "collect rc=$?"
python bench.py --detail gpurun_out/bench_${T}a_detail.json > gpurun_out/bench_${T}a.json 2> gpurun_out/bench_${T}a.err; echo "bench rc=$?"; tail -c 400 gpurun_out/bench_${T}a.err
fi
if [[ $PART == *b* ]]; then
PVX_BENCH_FORCE_GATHER=1 python bench.py --detail gpurun_out/bench_fg_detail.json --steps 5 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/bench_fg.json 2> gpurun_out/bench_fg.err; echo "forced gather rc=$?"
python bench.py --detail gpurun_out/bench_c4_detail.json --workload c4 --steps 5 --warmup 1 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "c4 rc=$?"; tail -2 gpurun_out/bench_c4.err
python tools/bench_configs.py > gpurun_out/configs_${T}.jsonl 2> gpurun_out/configs_${T}.err
python tools/sweep_config5.py > gpurun_out/sweep5_${T}.jsonl 2> gpurun_out/sweep5_${T}.err
python tools/sweep_config5.py 3600 64 > gpurun_out/sweep5_${T}_f64.jsonl 2>> gpurun_out/sweep5_${T}.err
python tools/ab_nfft.py 512,1024,2048,4096,8192 > gpurun_out/ab_nfft_${T}.jsonl 2>/dev/null
cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/next_rows_${T} -o r --output-format csv -- python3 /root/repo/tools/bench_next_rows.py > /root/repo/gpurun_out/next_rows_${T}.jsonl 2> /root/repo/gpurun_out/next_rows_${T}.err; cd /root/repo
fi
