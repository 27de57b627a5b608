#!/bin/bash
# the config-3 round trip under the kernel + memory-copy trace, and its last iterations as a timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/chain
python3 $R/tools/time_chain.py > $R/gpurun_out/chain/time_chain.txt 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/chain/trace -- python3 $R/tools/time_chain.py > /dev/null 2>&1
python3 $R/tools/chain_timeline.py $R/gpurun_out/chain/trace 60 > $R/gpurun_out/chain/timeline.txt 2>&1
cat $R/gpurun_out/chain/time_chain.txt; tail -45 $R/gpurun_out/chain/timeline.txt
