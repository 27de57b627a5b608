#!/bin/bash
# Instrumented build of libpvx_hip.so for tools/stamps_ring.py: s_memtime stamps inside k_fused_ring
# (a scratch copy of the sources is patched; the tree itself is not touched).
#   bash tools/ab/buildstamp_ring.sh && PVX_LIB=tools/ab/libpvx_st.so python tools/stamps_ring.py [noise]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
ST="${TMPDIR:-/tmp}/pvx_stamp_build"
rm -rf "$ST" && mkdir -p "$ST" && cp "$ROOT"/pypevoc_amd/csrc/*.hip "$ROOT"/pypevoc_amd/csrc/*.h "$ST"/
python "$HERE/mkstamp_ring.py" "$ROOT/pypevoc_amd/csrc/k_fused_ring.hip" "$ST/k_fused_ring.hip" "$ROOT/pypevoc_amd/csrc/pvx_api.hip" "$ST/pvx_api.hip"
cd "$ST"
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I"$ROOT/include" -c $f -o ${f%.hip}.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 *.o -shared -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib -o "$HERE/libpvx_st.so"
echo "built $HERE/libpvx_st.so"
