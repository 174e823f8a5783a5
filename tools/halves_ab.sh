#!/bin/bash
# A/B on the GPU box: half batches on two streams (default) against every launch on one stream (bench.py --one-stream),
# alternating, + a kernel trace of the default for the timeline.  bash tools/halves_ab.sh [repeats]
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-single-pair --no-pcie --no-ref-threshold --sections main,sequence"
for ((i = 0; i < ${1:-2}; i++)); do
  for V in "" "--one-stream"; do
    $B $V > gpurun_out/hab.json 2> gpurun_out/hab.err || exit 1
    python3 - "${V:-two halves}" <<PY
import json, sys
d = json.loads(open("gpurun_out/hab.json").read().strip().splitlines()[-1])
print("%-14s %9.1f pairs/s  %.3f ms/step  %9.1f frames/s" % (sys.argv[1], d["value"], d["ms_per_step"], d["sequence"]["value"]), flush=True)
PY
  done
done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/halves_trace -o s -- python3 bench.py --no-cpu-baseline --no-single-pair --no-pcie --no-ref-threshold --sections main --steps 4 --warmup 1 > /dev/null 2> gpurun_out/halves_trace.err
ls gpurun_out/halves_trace
