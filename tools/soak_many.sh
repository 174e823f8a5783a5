#!/bin/bash
# several soak universes in a row on the GPU box: bash tools/soak_many.sh FIRST_SEED COUNT OUT.jsonl
# (one process per seed, sequentially; stops at the first failure)
set -o pipefail
for ((s = $1; s < $1 + $2; s++)); do
  timeout -k 10 900 python3 tools/soak.py --cases 200 --seed $s >> $3 2>> $3.err || { echo "soak seed $s failed" | tee -a $3; exit 1; }
done
