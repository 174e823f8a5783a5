#!/usr/bin/env python3
"""Copy the summaries of tools/profile_r04.sh (gpurun_out/prof_<tag>/) into profiles/ under their round-4 names, rebuild
profiles/r04_ransac_hbm_traffic.json from the PMC summary, and print the per-kernel averages.
usage: python tools/refresh_profiles_r04.py [tag]"""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
g = "gpurun_out/prof_%s/" % tag
for leg in ("main", "halves", "refthr", "sequence", "refine", "extract"):
    shutil.copy(g + "stats_%s/s_kernel_stats.csv" % leg, "profiles/r04_rocprofv3_kernel_stats_%s.csv" % leg)
    shutil.copy(g + "bench_under_rocprof_%s.json" % leg, "profiles/bench_r04_under_rocprof_%s.json" % leg)
shutil.copy(g + "%s_pmc_summary.json" % tag, "profiles/r04_pmc_summary.json")
d = json.load(open("profiles/r04_pmc_summary.json"))
stage = d["ransac_stage_kernels"]


def short(k):
    return k.replace("void mvs::", "").replace("mvs::", "").split("(")[0]


per = {short(k): {"FETCH_SIZE_KB": d["kernels"][k].get("FETCH_SIZE"), "WRITE_SIZE_KB": d["kernels"][k].get("WRITE_SIZE"),
                  "hbm_bytes_per_pair": (2 * d["kernels"][k].get("FETCH_SIZE", 0.0) + d["kernels"][k].get("WRITE_SIZE", 0.0)) * 1024
                  / d["pairs_per_launch"]} for k in stage}
json.dump({"source": d["source"], "kernels": " + ".join(sorted(per)), "pairs_per_launch": d["pairs_per_launch"],
           "hbm_bytes_per_pair": d["hbm_bytes_per_pair"], "formula": d["formula"], "per_kernel": per,
           "note": "bench.py reads hbm_bytes_per_pair from this file for roofline.traffic (recorded, not live: counters need "
                   "their own rocprofv3 passes)"}, open("profiles/r04_ransac_hbm_traffic.json", "w"), indent=1)
print("hbm bytes per pair", d["hbm_bytes_per_pair"])
for k, v in sorted(per.items(), key=lambda kv: -kv[1]["hbm_bytes_per_pair"]):
    print("   %-52s %10.0f B/pair" % (k, v["hbm_bytes_per_pair"]))
for r in csv.DictReader(open("profiles/r04_rocprofv3_kernel_stats_main.csv")):
    print("  %-72s %5s calls  %.4f ms avg" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e6))
