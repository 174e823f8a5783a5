#!/usr/bin/env python3
"""Copy the summaries of tools/profile.sh (gpurun_out/prof_<tag>/) into profiles/ under the round's names (<tag>_*), rebuild
profiles/<tag>_ransac_hbm_traffic.json from the PMC summary, and print the per-kernel averages.
usage: python tools/refresh_profiles.py [tag]      (tag = round, default r05)"""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
g = "gpurun_out/prof_%s/" % tag
for leg in ("main", "halves", "refthr", "sequence", "refine", "extract"):
    shutil.copy(g + "stats_%s/s_kernel_stats.csv" % leg, "profiles/%s_rocprofv3_kernel_stats_%s.csv" % (tag, leg))
    shutil.copy(g + "bench_under_rocprof_%s.json" % leg, "profiles/bench_%s_under_rocprof_%s.json" % (tag, leg))
shutil.copy(g + "%s_pmc_summary.json" % tag, "profiles/%s_pmc_summary.json" % tag)
d = json.load(open("profiles/%s_pmc_summary.json" % tag))
stage = d["ransac_stage_kernels"]


def short(k):
    return k.replace("void mvs::", "").replace("mvs::", "").split("(")[0]


per = {short(k): {"FETCH_SIZE_KB": d["kernels"][k].get("FETCH_SIZE"), "WRITE_SIZE_KB": d["kernels"][k].get("WRITE_SIZE"),
                  "hbm_bytes_per_pair": (2 * d["kernels"][k].get("FETCH_SIZE", 0.0) + d["kernels"][k].get("WRITE_SIZE", 0.0)) * 1024
                  / d["pairs_per_launch"]} for k in stage}
json.dump({"source": d["source"], "kernels": " + ".join(sorted(per)), "pairs_per_launch": d["pairs_per_launch"],
           "hbm_bytes_per_pair": d["hbm_bytes_per_pair"], "formula": d["formula"], "per_kernel": per,
           "note": "bench.py reads hbm_bytes_per_pair from this file for roofline.traffic (recorded, not live: counters need "
                   "their own rocprofv3 passes)"}, open("profiles/%s_ransac_hbm_traffic.json" % tag, "w"), indent=1)
print("hbm bytes per pair", d["hbm_bytes_per_pair"])
for k, v in sorted(per.items(), key=lambda kv: -kv[1]["hbm_bytes_per_pair"]):
    print("   %-52s %10.0f B/pair" % (k, v["hbm_bytes_per_pair"]))
for r in csv.DictReader(open("profiles/%s_rocprofv3_kernel_stats_main.csv" % tag)):
    print("  %-72s %5s calls  %.4f ms avg" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e6))
