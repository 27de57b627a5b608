#!/bin/bash
# tracker variants on the GPU: phase stamps, chunk lengths, then the GPU suite and the config-3 trace
cd $GRAFT_REPO_ROOT
for c in 4 8 16; do echo "PVX_TRACK_CHUNK=$c"; PVX_TRACK_CHUNK=$c tools/ubench/track_phases | tail -1; done
tools/ubench/track_phases 4700 8 | tail -1
tools/ubench/track_phases 50000 8 | tail -1
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for c in 4 8; do echo "PVX_TRACK_CHUNK=$c"; PVX_TRACK_CHUNK=$c python3 tools/trace_chain.py 2>&1 | grep "track\]\|python:" | tail -2; done
python3 tools/bench_configs.py 2>&1 | head -2 | cut -c150-420
tools/ubench/track_phases 51676 8 | tail -1
PVX_TRACK_FPW=1 tools/ubench/track_phases 51676 8 | tail -1
python3 tools/bench_configs.py 2>/dev/null | sed -n 2p | cut -c150-700
