#!/usr/bin/env python3
"""Timeline of the last config-3 round trip in a rocprofv3 trace: every kernel and memory copy with its start offset,
duration and the idle gap before it.
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/chain -- python3 tools/time_chain.py
    python3 tools/chain_timeline.py gpurun_out/chain [n_events]"""
import csv, glob, os, sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ev = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:70]))
for path in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
ev.sort()
ev = ev[-n:]
t0 = ev[0][0]
prev = t0
for s, e, name in ev:
    print("%9.1f us  dur %8.1f us  gap %8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, name))
    prev = e
