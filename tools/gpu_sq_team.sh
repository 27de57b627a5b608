for N in ${1:-4096 8192}; do for KIND in ${2:-harmonic}; do
  PVX_RUN_NFFT=$N bash tools/prof_sq.sh gpurun_out/r03_team_sq_${N}_${KIND} -1 $KIND 8; rm -rf gpurun_out/r03_team_sq_${N}_${KIND}/g*/
  echo "sq $N $KIND done"
done; done
