#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv) per kernel.
Usage: pmc_summary.py <dir-or-csv> [...]  -> JSON {kernel: {counter: {mean, n}}}"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for arg in sys.argv[1:]:
        files = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True)
        for f in files:
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    name = row.get("Kernel_Name") or row.get("Kernel Name") or "?"
                    cn = row.get("Counter_Name") or row.get("Counter Name")
                    cv = row.get("Counter_Value") or row.get("Counter Value")
                    if cn is None or cv is None:
                        continue
                    acc[name][cn].append(float(cv))
    out = {}
    for k, d in acc.items():
        out[k] = {c: dict(mean=sum(v) / len(v), n=len(v), min=min(v), max=max(v)) for c, v in d.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
