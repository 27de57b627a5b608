# one GPU-box pass for a round: bench lines, SQ counters, HBM traffic, kernel-trace stats (outputs under gpurun_out/)
python bench.py > gpurun_out/bench_r02a.json 2> gpurun_out/bench_r02a.err; echo "bench rc=$?"; tail -c 600 gpurun_out/bench_r02a.err
python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/bench_g2.json 2> gpurun_out/bench_g2.err; echo "gpus2 rc=$?"; tail -2 gpurun_out/bench_g2.err
PVX_BENCH_FORCE_GATHER=1 python bench.py --steps 5 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/bench_fg.json 2> gpurun_out/bench_fg.err; echo "forced gather rc=$?"
python bench.py --workload c4 --steps 5 --warmup 1 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "c4 rc=$?"; tail -2 gpurun_out/bench_c4.err
bash tools/prof_sq.sh gpurun_out/sq_r02 -1; rm -rf gpurun_out/sq_r02/g*/
bash tools/prof_traffic.sh gpurun_out/traffic_r02; rm -rf gpurun_out/traffic_r02/*_SIZE/
cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/stats_r02 -o r --output-format csv -- python3 /root/repo/bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > /root/repo/gpurun_out/stats_r02.log 2>&1; cd /root/repo; ls gpurun_out/stats_r02
bash tools/prof_traffic.sh gpurun_out/traffic_r02_f64 -1 harmonic 8 64; rm -rf gpurun_out/traffic_r02_f64/*_SIZE/
python tools/bench_configs.py > gpurun_out/configs_r02.jsonl 2> gpurun_out/configs_r02.err
python tools/sweep_config5.py > gpurun_out/sweep5_r02.jsonl 2> gpurun_out/sweep5_r02.err
python tools/sweep_config5.py 3600 64 > gpurun_out/sweep5_r02_f64.jsonl 2>> gpurun_out/sweep5_r02.err
python tools/ab_nfft.py 512,1024,2048,4096,8192 > gpurun_out/ab_nfft_r02.jsonl 2>/dev/null
