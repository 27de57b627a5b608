# float64 analysis against the rows per launch (PVX_MAX_ROWS; default: 96 MiB of workspace): tools/ab_nfft.py's harmonic signal
for nf in ${AB_NFFTS:-2048 4096 8192}; do
  for mr in ${AB_ROWS:-0 1535 3071 4095 8191 16383 32767 65535 262143}; do
    if [ "$mr" = 0 ]; then unset PVX_MAX_ROWS; else export PVX_MAX_ROWS=$mr; fi
    echo "max_rows=$mr $(python3 tools/ab_nfft.py $nf 64 2>/dev/null | grep harmonic | cut -c1-170)"
  done
done
