set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/prof_dense
rm -rf $O; mkdir -p $O
for G in "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
         "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVES" "TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
  D=$O/pmc_$(echo $G | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $D -o p -- python3 tools/dense_probe.py > $D.log 2> $D.err || { echo "pmc pass failed: $G"; tail -3 $D.err; }
done
python3 tools/pmc_summary.py $O/dense_pmc_summary.json 256 $O/pmc_*
