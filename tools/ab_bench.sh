#!/bin/bash
# A/B of two library builds on bench.py's workload lines (harmonic C2, white noise, violin), alternating, in one call.
#   bash tools/ab_bench.sh [before.so]      (default tools/ab/libpvx_before.so against the tree's library)
cd $GRAFT_REPO_ROOT
B=${1:-tools/ab/libpvx_before.so}
for i in 1 2; do
  for lib in "$B" ""; do
    PVX_LIB=$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
w=d.get('workloads',{})
print('%-28s C2 %.1f M | ' % ('$lib' or 'tree', d['value']/1e6) + ' | '.join('%s %.1f M' % (k, v['value']/1e6) for k,v in w.items() if isinstance(v,dict) and 'value' in v))"
  done
done
