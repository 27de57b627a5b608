# k_fused_team (fft mode 5) against k_fused_mw (mode 2) at nfft 4096 / 8192 over npks: tools/ab_nfft.py's signals, hop = nfft/4
for k in ${AB_KS:-8 64 100 128}; do for m in 5 2; do echo "== npks=$k mode=$m"; AB_K=$k PVX_FFT_MODE=$m python3 tools/ab_nfft.py 4096,8192 2>&1 | grep "^{" | cut -c1-200; done; done
