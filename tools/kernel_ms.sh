#!/bin/bash
# per-kernel ms of whole-batch launches (bench.py's own HIP-event table), two repeats: bash tools/kernel_ms.sh [pattern]
for i in 1 2; do
python3 bench.py --no-cpu-baseline --no-single-pair --no-pcie --no-ref-threshold --sections main 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
pk = d['roofline']['per_kernel']
print(d['value'], d['ms_per_step'], ' '.join('%s=%.3f' % (k.split('<')[0].replace('ransac_', '').replace('_kernel', ''), v['ms']) for k, v in pk.items() if v['ms'] > 0.03))
"
done
