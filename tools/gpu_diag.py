#!/usr/bin/env python3
"""Prints measured HIP-vs-golden errors for every fixture (run on the GPU box; used to set the
tolerances written in tests/test_hip_parity.py)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.conftest import golden_names, load_golden  # noqa: E402
from tests.parity import compare_analysis, pv_result  # noqa: E402
import pypevoc_amd  # noqa: E402
from oracle import pvoracle  # noqa: E402


def main():
    rows = []
    for prec in (32, 64):
        for name in golden_names():
            g = load_golden(name)
            t0 = time.time()
            kw = dict(wind=lambda n, _w=g.get("win"): _w) if "win" in g else {}   # fixtures with a custom window
            p = pypevoc_amd.PV(g["x"], g["sr"], nfft=g["nfft"], hop=g["hop"], npks=g["npks"], **kw,
                               pkthresh=g["pkthresh"], progress=False, precision=prec)
            p.run_pv()
            c = compare_analysis(pv_result(p), g, g["nfft"], g["hop"], g["sr"])
            c.update(name=name, precision=prec, secs=round(time.time() - t0, 3))
            if "part_start" in g:
                ss = p.toSinSum()
                pid, st, ln = ss.partial_table()
                c["partials"] = int(len(st))
                c["partials_ref"] = int(len(g["part_start"]))
                # tracker in isolation: golden analysis arrays in
                s2 = pypevoc_amd.SinSum(g["sr"], nfft=g["nfft"], hop=g["hop"])
                s2._from_analysis(g["f"], g["mag"], g["ph"], g["realph"])
                pid2, st2, ln2 = s2.partial_table()
                c["track_exact"] = bool(np.array_equal(st2, g["part_start"]) and np.array_equal(ln2, g["part_len"]) and
                                        np.array_equal(pvoracle.part_slots(pid2, st2, ln2), g["part_slot"]))
                for k in g:
                    if k.startswith("w_hop"):
                        h = int(k[5:])
                        w2 = s2.synth(g["sr"], h)
                        c["synth_iso_err_h%d" % h] = float(np.abs(w2 - g[k]).max())
                        w = ss.synth(g["sr"], h)
                        if w.shape == g[k].shape:
                            c["synth_e2e_err_h%d" % h] = float(np.abs(w - g[k]).max())
                        c["wmax_h%d" % h] = float(np.abs(g[k]).max())
            rows.append(c)
            print(json.dumps({k: (float(v) if isinstance(v, (np.floating,)) else v) for k, v in c.items()}))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
