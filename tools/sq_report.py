#!/usr/bin/env python3
"""Per-frame view of a tools/prof_sq.sh summary.   python tools/sq_report.py SUMMARY.json NFFT [kernel-substring]"""
import json, sys
d = json.load(open(sys.argv[1])); N = int(sys.argv[2]); sub = sys.argv[3] if len(sys.argv) > 3 else "fused"
F = (44100 * 600 - N + N // 4 - 1) // (N // 4)
for k, v in d.items():
    if sub not in k: continue
    g = lambda c: v[c]['mean'] if c in v else float('nan')
    print(k[:90]); wc = g('SQ_WAVE_CYCLES')
    print("  waves %d  wave-cycles/frame %.0f  busy_cycles %.0f" % (g('SQ_WAVES'), wc / F, g('SQ_BUSY_CYCLES')))
    print("  per frame: VALU %.0f (add %.0f mul %.0f fma %.0f trans %.0f int %.0f cvt %.0f f64 %.0f) SALU %.0f LDS %.0f VMEM rd %.0f wr %.0f branch %.0f SMEM %.0f" % (tuple(g(c) / F for c in
          ['SQ_INSTS_VALU', 'SQ_INSTS_VALU_ADD_F32', 'SQ_INSTS_VALU_MUL_F32', 'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_TRANS_F32', 'SQ_INSTS_VALU_INT32', 'SQ_INSTS_VALU_CVT']) + ((g('SQ_INSTS_VALU_FMA_F64') + g('SQ_INSTS_VALU_ADD_F64') + g('SQ_INSTS_VALU_MUL_F64')) / F,) +
          tuple(g(c) / F for c in ['SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_BRANCH', 'SQ_INSTS_SMEM'])))
    print("  of wave cycles: active any %.3f valu %.3f lds %.3f sca %.3f | wait any %.3f wait-inst any %.3f (lds %.3f)" % tuple(g(c) / wc for c in
          ['SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_SCA', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS']))
    print("  LDS: bank-conflict/idx-active %.3f  idx-active/busy-cu-cycles %.3f" % (g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'), g('SQ_LDS_IDX_ACTIVE') / g('SQ_BUSY_CU_CYCLES')))
