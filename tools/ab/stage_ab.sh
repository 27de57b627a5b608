# whole-path config 2 for several shapes of the threaded staging ring:   bash tools/ab/stage_ab.sh
for cfg in "4 4" "4 8" "4 16" "2 8" "3 8" "5 4"; do set -- $cfg; echo "threads $1 piece $2 MB"; PVX_STAGE_THREADS=$1 PVX_STAGE_PIECE_MB=$2 python tools/time_c2_chain.py 2>/dev/null | tail -2 || exit 1; done
