#!/bin/bash
# SQ counter passes (rocprofv3 --pmc, <= 8 SQ counters per pass) over tools/run_mode.py.
#   bash tools/prof_sq.sh OUTDIR MODE [harmonic|noise] [K] [PRECISION]      (run on the GPU box, from the repo root)
#   PVX_PROF_PROG="tools/run_chain.py 2": profile that program instead of tools/run_mode.py (its arguments replace MODE ...)
set -u
OUT=$1; MODE=$2; KIND=${3:-harmonic}; K=${4:-8}; PREC=${5:-32}
PROG=${PVX_PROF_PROG:-tools/run_mode.py $MODE $KIND $K 4 $PREC}
export TMPDIR=/tmp
mkdir -p "$OUT"
G_A="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
G_B="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"
G_C="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_IFETCH"
G_D="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64"
G_E="SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_IOPS SQ_ACTIVE_INST_VALU2"
i=0
for G in "$G_A" "$G_B" "$G_C" "$G_D" "$G_E"; do
  i=$((i+1))
  rocprofv3 --pmc $G GRBM_GUI_ACTIVE -d "$OUT/g$i" -o r --output-format csv -- python3 $PROG > "$OUT/g$i.log" 2>&1
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.json"
