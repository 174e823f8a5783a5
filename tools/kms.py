"""print the per-kernel milliseconds of a bench.py JSON line (tools: reading gpurun_out results)"""
import json
import sys

d = json.load(open(sys.argv[1]))
print(d["value"], d["unit"], d["ms_per_step"], "ms/step")
for k, e in d["roofline"]["per_kernel"].items():
    print("  %-48s %8.4f ms  vgpr %s" % (k, e.get("ms", float("nan")), e.get("vgprs")))
