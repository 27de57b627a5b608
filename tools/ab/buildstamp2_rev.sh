#!/bin/bash
# Instrumented build of libpvx_hip.so for tools/stamps2_rev.py: s_memtime stamps inside k_fused_rev (nfft 2048, round-5 sources)
#   bash tools/ab/buildstamp2_rev.sh && PVX_ALLOW_STALE_LIB=1 PVX_LIB=tools/ab/libpvx_st.so python tools/stamps2_rev.py [noise|violin]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(cd "$HERE/../.." && pwd)"
ST="${TMPDIR:-/tmp}/pvx_stamp_build"; rm -rf "$ST" && mkdir -p "$ST" && cp "$ROOT"/pypevoc_amd/csrc/*.hip "$ROOT"/pypevoc_amd/csrc/*.h "$ROOT"/pypevoc_amd/csrc/build_sha.inc "$ST"/
python "$HERE/mkstamp2_rev.py" "$ROOT/pypevoc_amd/csrc/k_fused_rev.hip" "$ST/k_fused_rev.hip" "$ROOT/pypevoc_amd/csrc/pvx_api.hip" "$ST/pvx_api.hip"
cd "$ST"
for f in pvx_api k_fused_rev; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I"$ROOT/include" -I/opt/rocm/include -c $f.hip -o $f.o &
done
wait
OBJS=""; for f in "$ROOT"/pypevoc_amd/csrc/*.o; do b=$(basename $f); case $b in pvx_api.o|k_fused_rev.o|k_fused.o|k_fused_ring.o) ;; *) OBJS="$OBJS $f";; esac; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 pvx_api.o k_fused_rev.o $OBJS -shared -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib -o "$HERE/libpvx_st.so"
echo "built $HERE/libpvx_st.so"
