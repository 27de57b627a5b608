#!/usr/bin/env python3
"""Registers / scratch of every kernel in a gfx950 code object: compiles SRC.hip to assembly in a temp dir and prints the
metadata (vgpr, agpr, sgpr spills, vgpr spills, scratch bytes, LDS).   python tools/kstats.py pypevoc_amd/csrc/k_fused_rev.hip [filter]"""
import os, re, subprocess, sys, tempfile
src = os.path.abspath(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
inc = os.path.join(os.path.dirname(src), "..", "..", "include")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off"] + (["-fno-slp-vectorize"] if os.path.basename(src) == "k_synth.hip" else []) + ["-I" + inc, "-I/opt/rocm/include",
                           "-I" + os.path.dirname(src), "--cuda-device-only", "-S", src, "-o", out], stderr=subprocess.DEVNULL)
    txt = open(out).read()
pat = re.compile(r'\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', re.S)
for m in pat.finditer(txt):
    a, name, priv, sg, sgs, vg, vgs = m.groups()
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace("(anonymous namespace)::", "").replace("(FusedParams)", "")
    if flt in dn:
        print("%-64s vgpr %3s agpr %3s scratch %4s B  vgpr-spill %3s  sgpr %3s sgpr-spill %3s" % (dn[:64], vg, a, priv, vgs, sg, sgs))
