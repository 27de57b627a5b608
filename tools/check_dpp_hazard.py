#!/usr/bin/env python3
"""Static check on the generated gfx950 ISA of the fused kernels and the tracker: no DPP move may read a register that
one of the inline-asm primitives of pvx_cplx.h wrote less than two wait states earlier.  (VALU write ->
DPP read needs two wait states on gfx9; the compiler inserts them for instructions it emitted itself,
but its hazard recogniser does not look inside inline asm.)  Exit status 1 if a candidate is found.

    python tools/check_dpp_hazard.py            # compiles the fused kernels to assembly
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pypevoc_amd", "csrc")


def regs(tok):
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(path):
    inasm = False
    recent = []           # (wait states elapsed since the write, registers written by an asm instruction)
    ndpp = nasm = 0
    bad = []
    for ln in open(path):
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if t.startswith(";;#ASMEND"):
            inasm = False
            continue
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_nop":
            n = int(t.split()[1]) + 1
            recent = [(a + n, r) for a, r in recent]
            continue
        if inasm:
            nasm += 1
            parts = t.split()
            if op.startswith("v_") and len(parts) > 1:           # s_waitcnt / s_barrier in asm write no VGPR
                recent.append((0, regs(parts[1].rstrip(","))))
            continue
        if op.startswith("v_mov_b32_dpp"):
            ndpp += 1
            src = regs(t.split()[2].rstrip(","))
            if any(a < 2 and (src & r) for a, r in recent):
                bad.append(t)
        recent = [(a + 1, r) for a, r in recent if a + 1 < 3]
    return ndpp, nasm, bad


def main():
    hipcc = os.environ.get("HIPCC", "hipcc")
    status = 0
    with tempfile.TemporaryDirectory() as td:
        for name in ("k_fused", "k_fused_mw", "k_fused_ring", "k_track"):
            out = os.path.join(td, name + ".s")
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                                   "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-o", out,
                                   os.path.join(CSRC, name + ".hip")], stderr=subprocess.DEVNULL)
            ndpp, nasm, bad = scan(out)
            print("%s: %d DPP moves, %d inline-asm instructions, %d asm -> DPP hazard candidates" % (name, ndpp, nasm, len(bad)))
            for b in bad[:10]:
                print("   ", b)
            if bad or ndpp == 0 or nasm == 0:
                status = 1
    return status


if __name__ == "__main__":
    sys.exit(main())
