#!/bin/bash
# the round's profile set, run on the GPU box from the repo root: bash tools/profile.sh [tag]   (tag = round, default r05;
# python tools/refresh_profiles.py <tag> then copies the summaries into profiles/<tag>_*)
# (rocprofv3 gets the program itself after `--`, never a wrapper; counters in their own passes, one group per pass)
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r05}
O=gpurun_out/prof_$TAG
mkdir -p $O
# --one-stream: every launch covers the whole batch, as in bench.py's live per-kernel table (mvs_batch_time_kernels); the default
# run sends two half batches down two streams, whose launches overlap each other (a per-launch duration is then not a property
# of the kernel).  stats_halves is the default command shape for comparison.
B="python3 bench.py --no-cpu-baseline --no-single-pair --no-pcie --one-stream"
# per-kernel times of the driver's command shape, one leg per CSV
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_main -o s -- $B --no-ref-threshold --sections main --steps 10 --warmup 2 > $O/bench_under_rocprof_main.json 2> $O/stats_main.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_halves -o s -- python3 bench.py --no-cpu-baseline --no-single-pair --no-pcie --no-ref-threshold --sections main --steps 10 --warmup 2 > $O/bench_under_rocprof_halves.json 2> $O/stats_halves.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_refthr -o s -- $B --sections main --steps 3 --warmup 1 > $O/bench_under_rocprof_refthr.json 2> $O/stats_refthr.err || exit 1
for S in sequence refine extract; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$S -o s -- $B --no-ref-threshold --sections main,$S --steps 2 --warmup 1 > $O/bench_under_rocprof_$S.json 2> $O/stats_$S.err || exit 1
done
# counters (128 pairs per launch), one group per pass; FETCH_SIZE and WRITE_SIZE each alone (MI355X_MICROARCH.md: TCC slots)
PB="python3 bench.py --no-cpu-baseline --no-single-pair --no-ref-threshold --no-pcie --one-stream --sections main"
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
         "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_CYCLES_SALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR"; do
  D=$O/pmc_$(echo $G | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $D -o p -- $PB --steps 1 --warmup 0 --pairs 128 > $D.json 2> $D.err || { echo "pmc pass failed: $G"; tail -3 $D.err; }
done
# the summary records the command the counters were collected under (whole-batch launches, --one-stream)
python3 tools/pmc_summary.py $O/${TAG}_pmc_summary.json 128 "$PB --steps 1 --warmup 0 --pairs 128" $O/pmc_*
ls $O
