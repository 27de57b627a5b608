#!/bin/bash
# A/B build: recompile some sources with extra flags and link them with the tree's other objects.
#   bash tools/ab/build_variant.sh NAME "-DFLAG=1 ..." k_peaks [k_stft ...]   ->  tools/ab/libpvx_NAME.so
#   PVX_LIB=tools/ab/libpvx_NAME.so python tools/ab_nfft.py ...
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(cd "$HERE/../.." && pwd)"; C="$ROOT/pypevoc_amd/csrc"
NAME="$1"; FLAGS="$2"; shift 2
make -s -C "$C" -j4 >/dev/null
T="${TMPDIR:-/tmp}/pvx_variant_$NAME"; rm -rf "$T"; mkdir -p "$T"; cp "$C"/*.o "$T"/; rm -f "$T"/k_fused.o "$T"/k_fused_ring.o "$T"/k_pv_team.o "$T"/k_fused_mw.o    # (the witness kernels live in tests/libpvx_witness.so)
for f in "$@"; do
  X=""; [ "$f" = k_synth ] && X="-fno-slp-vectorize"          # (as the Makefile)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $X -I"$ROOT/include" -I/opt/rocm/include $FLAGS -c "$C/$f.hip" -o "$T/$f.o" || { echo "variant compile of $f failed"; exit 1; }
done
:
/opt/rocm/bin/hipcc --offload-arch=gfx950 "$T"/*.o -shared -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib -o "$HERE/libpvx_$NAME.so"
echo "built $HERE/libpvx_$NAME.so"
