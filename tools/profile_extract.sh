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
python3 tools/extract_pmc_summary.py $O/extract_pmc_summary.json $(ls -d $O/pmc_* | grep -v "\.")
# A/B of describe_kernel's block order: the diagnostics library with MVS_ORB_FLAT_ORDER runs the (level, image,
# split) order of rounds 2-4, without it the XCD-aware order of the product; kernel times and fetched bytes of both
export MVS_USE_DEBUG_LIB=1
for ORDER in xcd flat; do
  if [ $ORDER = flat ]; then export MVS_ORB_FLAT_ORDER=1; else unset MVS_ORB_FLAT_ORDER; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ab_${ORDER}_stats -o s -- $B > $O/ab_${ORDER}_stats.json 2> $O/ab_${ORDER}_stats.err || exit 1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/ab_${ORDER}_fetch -o p -- $B > $O/ab_${ORDER}_fetch.json 2> $O/ab_${ORDER}_fetch.err || echo "fetch pass failed: $ORDER"
done
unset MVS_ORB_FLAT_ORDER MVS_USE_DEBUG_LIB
python3 - <<'PY'
import csv, json, glob, collections
out = {"what": "describe_kernel, 64 frames of 640x480 at 2000 features (tools/extract_bench.py): block order A/B on the same binary "
               "(libmvslam_hip_dbg.so); FETCH_SIZE in KB per launch as rocprofv3 reports it (MI355X_MICROARCH.md: x2 for bytes from HBM/MALL on gfx950)"}
for order in ("xcd", "flat"):
    st = [r for r in csv.DictReader(open("gpurun_out/prof_extract/ab_%s_stats/s_kernel_stats.csv" % order)) if "describe_kernel" in r["Name"]]
    fetch = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/prof_extract/ab_%s_fetch/*counter_collection.csv" % order):
        for r in csv.DictReader(open(f)):
            if "describe_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                fetch[r["Dispatch_Id"]].append(float(r["Counter_Value"]))
    per = [sum(v) for v in fetch.values()]
    out[order] = {"avg_us": float(st[0]["AverageNs"]) / 1e3 if st else None, "launches": int(st[0]["Calls"]) if st else 0,
                  "fetch_size_kb_per_launch": sum(per) / len(per) if per else None}
json.dump(out, open("gpurun_out/prof_extract/describe_block_order_ab.json", "w"), indent=1)
print(json.dumps(out))
PY
