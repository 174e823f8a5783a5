#!/usr/bin/env python3
"""Per-kernel averages of the extraction kernels' PMC passes (tools/profile_extract.sh):
python tools/extract_pmc_summary.py OUT.json PASS_DIR...     (each PASS_DIR holds one rocprofv3 --pmc pass)"""
import collections, csv, glob, json, re, sys

out_path, dirs = sys.argv[1], sys.argv[2:]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/*counter_collection.csv"):
        acc, names = collections.defaultdict(float), {}
        for r in csv.DictReader(open(f)):
            acc[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (disp, cn), v in acc.items():
            per[names[disp]][cn].append(v)
out = {"source": "rocprofv3 --kernel-trace --pmc <group> (one group per pass) -- python3 tools/extract_bench.py --resident --cpu-images 0 "
                 "(tools/profile_extract.sh): 64 frames of 640x480, 2000 features; per-dispatch averages per kernel; SQ_*CYCLES / SQ_WAIT* / "
                 "SQ_ACTIVE* in units of 4 clocks", "kernels": {}}
for k, c in per.items():
    m = re.search(r"(\w+_kernel)", k)
    if "anonymous" not in k or not m:
        continue
    e = {cn: round(sum(v) / len(v), 1) for cn, v in sorted(c.items())}
    e["dispatches_per_pass"] = len(next(iter(c.values())))
    out["kernels"][m.group(1)] = e
json.dump(out, open(out_path, "w"), indent=1)
for k, v in out["kernels"].items():
    w = max(v.get("SQ_WAVES", 1.0), 1.0)
    print("%-18s waves %9.0f  valu/wave %6.0f  lds/wave %5.0f  vmem_rd/wave %5.1f" % (k, w, v.get("SQ_INSTS_VALU", 0) / w, v.get("SQ_INSTS_LDS", 0) / w,
                                                                                    v.get("SQ_INSTS_VMEM_RD", 0) / w))
