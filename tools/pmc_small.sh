set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/prof_r03c
mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1 || true
B="python3 bench.py --no-cpu-baseline --no-single-pair --no-ref-threshold --sections main"
for G in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ" "SQC_DCACHE_INPUT_VALID_READB SQC_TC_STALL SQC_ICACHE_MISSES SQC_ICACHE_REQ" \
         "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES" "SQ_IFETCH SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_ANY" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
  D=$O/pmc_$(echo $G | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $D -o p -- $B --steps 1 --warmup 0 --pairs 128 > $D.json 2> $D.err || { echo "pmc pass failed: $G"; tail -3 $D.err; }
done
python3 tools/pmc_summary.py $O/r03c_pmc_summary.json 128 $O/pmc_*
