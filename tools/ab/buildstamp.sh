set -e
rm -rf /tmp/st && mkdir -p /tmp/st && cp /root/repo/pypevoc_amd/csrc/*.hip /root/repo/pypevoc_amd/csrc/*.h /tmp/st/
python /tmp/mkstamp.py /root/repo/pypevoc_amd/csrc/k_fused.hip /tmp/st/k_fused.hip /root/repo/pypevoc_amd/csrc/pvx_api.hip /tmp/st/pvx_api.hip
cd /tmp/st
for f in pvx_api k_frames k_peaks k_fused k_fused_mw k_track k_synth k_wire k_harmonic k_reduce; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I/root/repo/include -c $f.hip -o $f.o 2>/dev/null & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 *.o -shared -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib -o /root/repo/tools/ab/libpvx_st.so
