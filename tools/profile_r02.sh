#!/bin/bash
# round-2 profile set, run on the GPU box from the repo root: bash tools/profile_r02.sh
# (rocprofv3 gets the program itself after `--`, never a wrapper; counters in their own passes)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/prof_r02
mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-single-pair --no-ref-threshold"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $B --steps 10 --warmup 2 > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  D=$O/pmc_$(echo $G | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $D -o p -- $B --steps 1 --warmup 0 --pairs 128 > $D.json 2> $D.err || { echo "pmc pass failed: $G"; tail -3 $D.err; }
done
python3 tools/pmc_summary.py $O/r02_pmc_summary.json 128 $O/pmc_*
ls $O/stats
