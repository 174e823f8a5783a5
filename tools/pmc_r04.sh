#!/bin/bash
# round-4 counter passes (128 pairs per launch), one group per pass: bash tools/pmc_r04.sh [tag]
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r04}
O=gpurun_out/prof_$TAG
mkdir -p $O
PB="python3 bench.py --no-cpu-baseline --no-single-pair --no-ref-threshold --no-pcie --sections main"
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_CYCLES_SALU" \
         "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR"; do
  D=$O/pmc_$(echo $G | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $D -o p -- $PB --steps 1 --warmup 0 --pairs 128 > $D.json 2> $D.err || { echo "pmc pass failed: $G"; tail -3 $D.err; }
done
python3 tools/pmc_summary.py $O/${TAG}_pmc_summary.json 128 $O/pmc_*
