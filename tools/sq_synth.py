#!/usr/bin/env python3
"""Per-kernel view of a tools/prof_sq.sh summary for the resynthesis kernels (largest launch of each kernel only).
   python tools/sq_synth.py SUMMARY.json"""
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if 'k_synth' not in k and 'k_track' not in k and 'k_assign' not in k: continue
    g = lambda c, f='max': v[c][f] if c in v else float('nan')
    wc = g('SQ_WAVE_CYCLES')
    gui = g('GRBM_GUI_ACTIVE') / 8.0
    print(k.replace('(anonymous namespace)::', '')[:70])
    print("  waves %d  kernel cycles (GUI/8) %.0f  mean waves per SIMD %.2f" % (g('SQ_WAVES'), gui, 4.0 * wc / 1024 / gui))
    print("  wave-instr: VALU %.3g (f64 fma %.3g add %.3g mul %.3g int %.3g cvt %.3g) SALU %.3g VMEM rd %.3g wr %.3g LDS %.3g branch %.3g" % tuple(
        g(c) for c in ['SQ_INSTS_VALU', 'SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_INT32', 'SQ_INSTS_VALU_CVT', 'SQ_INSTS_SALU', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_LDS', 'SQ_INSTS_BRANCH']))
    print("  VALU busy share of the kernel's SIMD-cycles %.3f  (cycles per VALU instr %.2f)" % (4.0 * g('SQ_ACTIVE_INST_VALU') / 1024 / gui, 4.0 * g('SQ_ACTIVE_INST_VALU') / g('SQ_INSTS_VALU')))
    print("  of wave cycles: active any %.3f valu %.3f vmem %.3f lds %.3f | wait any %.3f wait-inst %.3f" % tuple(g(c) / wc for c in
          ['SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY']))
