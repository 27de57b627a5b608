#!/usr/bin/env python3
"""Basic-block view of one kernel in a hipcc -S listing: per block the instruction mix (float VALU, integer / logic VALU,
moves, v_readlane / v_writelane = spilled scalar registers, SALU, LDS, vector memory, waits), and the branch targets.
   python tools/isa_blocks.py LISTING.s KERNEL_SUBSTRING [--loop]      (--loop: only blocks inside the largest backward-branch span)"""
import re, sys
txt = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
only_loop = "--loop" in sys.argv
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and key in l and l.rstrip().endswith(tuple([":" + "", ""])) and ":" in l.split(";")[0])
end = next(i for i in range(start, len(txt)) if txt[i].strip().startswith("s_endpgm"))
body = txt[start + 1:end + 1]
def cat(op):
    if op in ("v_readlane_b32", "v_writelane_b32"): return "spill"
    if op.startswith(("v_mov", "v_accvgpr", "v_pk_mov")): return "mov"
    if op.startswith("v_cvt"): return "cvt"
    if op.startswith("v_"):
        if re.search(r"_f(16|32|64)", op) and not op.startswith(("v_cmp", "v_cndmask")): return "vf64" if "f64" in op else "vf32"
        return "vint"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith(("s_cbranch", "s_branch")): return "br"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"
blocks, cur = [], dict(name="entry", line=0, ins=[], tgt=[])
for i, l in enumerate(body):
    s = l.split(";")[0].strip()
    if not s: continue
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        blocks.append(cur); cur = dict(name=m.group(1), line=i, ins=[], tgt=[]); continue
    if s.startswith("."): continue
    op = s.split()[0]
    cur["ins"].append(op)
    if op.startswith(("s_cbranch", "s_branch")):
        cur["tgt"].append(s.split()[-1])
blocks.append(cur)
idx = {b["name"]: k for k, b in enumerate(blocks)}
# largest backward span
span = (0, 0)
for k, b in enumerate(blocks):
    for t in b["tgt"]:
        if t in idx and idx[t] <= k and k - idx[t] > span[1] - span[0]: span = (idx[t], k)
cats = ["vf32", "vf64", "vint", "mov", "cvt", "spill", "salu", "smem", "lds", "vmem", "wait", "nop", "br"]
print("kernel lines %d..%d, %d blocks; largest loop: blocks %d..%d (%s .. %s)" % (start, end, len(blocks), span[0], span[1], blocks[span[0]]["name"], blocks[span[1]]["name"]))
print("%-12s %5s " % ("block", "n") + " ".join("%5s" % c for c in cats) + "  targets")
tot = dict.fromkeys(cats, 0)
for k, b in enumerate(blocks):
    if only_loop and not (span[0] <= k <= span[1]): continue
    c = dict.fromkeys(cats, 0)
    for op in b["ins"]:
        cc = cat(op)
        if cc in c: c[cc] += 1
    for x in cats: tot[x] += c[x]
    if len(b["ins"]) >= (1 if only_loop else 8):
        print("%-12s %5d " % (b["name"], len(b["ins"])) + " ".join("%5d" % c[x] for x in cats) + "  " + ",".join(b["tgt"]) + ("   <loop>" if span[0] <= k <= span[1] else ""))
print("%-12s %5d " % ("total", sum(tot.values())) + " ".join("%5d" % tot[x] for x in cats))
