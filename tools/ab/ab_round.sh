echo "== mode 1 (wave kernel)"; PVX_FFT_MODE=1 python tools/ab_nfft.py 512,1024,2048 | cut -c1-140
echo "== mode 3 (ring kernel)"; PVX_FFT_MODE=3 python tools/ab_nfft.py 512,1024,2048 | cut -c1-140
echo "== nfft 4096/8192 (mw)"; python tools/ab_nfft.py 4096,8192 | cut -c1-140
for r in 6145 20000 65536; do echo "== general f32 path, PVX_MAX_ROWS=$r"; PVX_MAX_ROWS=$r python bench.py --fft-mode 0 --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(j['value'], j['ms_per_step'], [(k['kernel'], k['ms_per_launch'], k['launches']) for k in j['stage']['kernels']])"; done
