set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/prof_r03b
mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-single-pair --no-ref-threshold --sections main"
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" \
         "SQ_INSTS_VALU_FMA_F32 SQ_WAIT_INST_LDS SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_INSTS_VALU_FMA_F64"; do
  D=$O/pmc_$(echo $G | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $D -o p -- $B --steps 1 --warmup 0 --pairs 128 > $D.json 2> $D.err || { echo "pmc pass failed: $G"; tail -3 $D.err; }
done
python3 tools/pmc_summary.py $O/r03b_pmc_summary.json 128 $O/pmc_*
