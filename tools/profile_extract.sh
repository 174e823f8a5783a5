#!/bin/bash
# PMC passes of the extraction kernels (row f3), run on the GPU box from the repo root: bash tools/profile_extract.sh
# (rocprofv3 gets the program itself after `--`; counters in their own passes, kernel trace only)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/prof_extract
mkdir -p $O
B="python3 tools/extract_bench.py --resident --cpu-images 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $B > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
         "SQ_INSTS_VMEM_WR SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD"; do
  D=$O/pmc_$(echo $G | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $D -o p -- $B > $D.json 2> $D.err || { echo "pmc pass failed: $G"; tail -3 $D.err; }
done
python3 tools/pmc_summary.py $O/extract_pmc_summary.json 64 $O/pmc_*
