# round-3 baseline on a GPU box: nfft sweep timings, then SQ counters of k_fused_mw at nfft 4096 / 8192 (harmonic, noise)
python tools/ab_nfft.py 512,1024,2048,4096,8192 > gpurun_out/r03_ab_nfft_base.jsonl 2> gpurun_out/r03_ab_nfft_base.err; echo "ab rc=$?"
for N in 4096 8192; do for KIND in harmonic noise; do
  PVX_RUN_NFFT=$N bash tools/prof_sq.sh gpurun_out/r03_mw_sq_${N}_${KIND} -1 $KIND 8; rm -rf gpurun_out/r03_mw_sq_${N}_${KIND}/g*/
  echo "sq $N $KIND done"
done; done
