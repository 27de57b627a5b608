#!/usr/bin/env python3
"""median / min duration of every k_synth / k_track kernel in a rocprofv3 kernel trace:  python tools/ab/ktimes.py TRACE.csv [name-regex]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"k_(synth|track|assign)\w*(<[\d, ]+>)?")
d = collections.defaultdict(list)
for r in rows:
    m = pat.search(r["Kernel_Name"])
    if m: d[(m.group(0), r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    v = sorted(v); print("   %-26s grid %-9s n %3d median %7.1f us  min %7.1f" % (k[0], k[1], len(v), v[len(v) // 2] / 1e3, v[0] / 1e3))
