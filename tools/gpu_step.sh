# one development step on the GPU box: the fused kernels' tests, then the nfft sweep timings
timeout -k 10 1100 python -m pytest tests/test_hip_parity.py -x -q -k "${2:-team_kernel or multiwave or fused_kernel or chunked or ring_kernel or full_size}" > gpurun_out/t1.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t1.log
python tools/ab_nfft.py ${1:-2048,4096,8192} > gpurun_out/ab_step.jsonl 2> gpurun_out/ab_step.err; cat gpurun_out/ab_step.jsonl
