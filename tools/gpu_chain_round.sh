#!/bin/bash
# GPU suite, then the config-3 round trip: wall times and the traced timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/chain
python -m pytest tests -m gpu -x -q > gpurun_out/chain/pytest.txt 2>&1; tail -5 gpurun_out/chain/pytest.txt
bash tools/gpu_chain_trace.sh
