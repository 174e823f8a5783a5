#!/usr/bin/env python3
"""Fold rocprofv3 --pmc CSV outputs (one counter group per pass, each in its own directory) into one JSON:
per kernel the per-dispatch average of every counter, plus the HBM bytes of the RANSAC stage computed as
MI355X_MICROARCH.md prescribes ((2 * FETCH_SIZE + WRITE_SIZE) KB on gfx950).
usage: python tools/pmc_summary.py OUT.json PAIRS_PER_LAUNCH "COMMAND" DIR [DIR ...]
COMMAND is the bench command the passes ran (recorded verbatim); the script refuses a command without --one-stream when the
pair count would split into half batches (a per-dispatch average would then mix 64- and 128-pair launches: ADVICE r4) and
checks that every stage kernel was dispatched equally often in every pass."""
import csv, glob, json, os, sys
from collections import defaultdict

out_path, pairs, command = sys.argv[1], int(sys.argv[2]), sys.argv[3]
if pairs >= 64 and "--one-stream" not in command:
    sys.exit("pmc_summary: %d pairs per launch without --one-stream run as two half batches: per-dispatch averages would be "
             "per half, not per launch" % pairs)
acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[4:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        for row in csv.DictReader(open(f)):
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = row["Kernel_Name"]
        for (disp, ctr), v in per_dispatch.items():
            acc[names[disp]][ctr].append(v)
kern = {k: {c: sum(v) / len(v) for c, v in ctrs.items()} for k, ctrs in acc.items()}
# the stage's own launches: not the instrumented replay (the <..., true> variants and the fused ransac_kernel of mvs_batch_stats)
# (ransac_finish_mfma_kernel<false, true> is the product's PILOT: its second argument is not the instrumentation switch)
def replay(k):
    if "ransac_kernel<" in k or "<true" in k:
        return True
    return ", true" in k and "ransac_finish_mfma_kernel<false, true>" not in k


stage = [k for k in kern if ("ransac_" in k or "pair_prepare" in k) and not replay(k)]
hbm = sum((2 * kern[k].get("FETCH_SIZE", 0.0) + kern[k].get("WRITE_SIZE", 0.0)) * 1024 for k in stage)
# every counter of a kernel comes from its own pass: the same number of dispatches everywhere, or a pass saw other launches
dispatches = {k: sorted({len(v) for v in ctrs.values()}) for k, ctrs in acc.items()}
uneven = {k: v for k, v in dispatches.items() if len(v) > 1 and k in stage}
if uneven:
    sys.exit("pmc_summary: dispatch counts differ between passes: %r" % uneven)
json.dump({"source": "rocprofv3 --pmc (one counter group per pass): " + command,
           "dispatches_per_pass": {k: dispatches[k][0] for k in stage},
           "pairs_per_launch": pairs, "units": "per-dispatch averages; FETCH_SIZE / WRITE_SIZE in KB",
           "ransac_stage_kernels": stage, "hbm_bytes_per_pair": hbm / pairs,
           "formula": "sum over the stage's kernels of (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 / pairs",
           "kernels": kern}, open(out_path, "w"), indent=1)
print(out_path, "hbm_bytes_per_pair", hbm / pairs)
