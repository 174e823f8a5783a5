#!/usr/bin/env python3
"""Copy the summaries of the last GPU run (gpurun_out/, written by bench.py --pcie, tools/profile_r02.sh and
tools/sequence_bench.py) into profiles/ under their round-2 names and print the figures the documents quote."""
import csv, json, os, shutil
g = "gpurun_out/"
pairs = [("bench_default.json", "bench_r02_default.json"), ("prof_r02/bench_under_rocprof.json", "bench_r02_under_rocprof.json"),
         ("prof_r02/stats/s_kernel_stats.csv", "r02_rocprofv3_kernel_stats.csv"), ("prof_r02/r02_pmc_summary.json", "r02_pmc_summary.json"),
         ("seq.json", "bench_r02_sequence_config5.json"), ("seq_refit.json", "bench_r02_sequence_config5_refit.json")]
for a, b in pairs:
    if os.path.exists(g + a):
        shutil.copy(g + a, "profiles/" + b)
d = json.load(open("profiles/r02_pmc_summary.json"))
old = json.load(open("profiles/r02_ransac_hbm_traffic.json"))
stage = d["ransac_stage_kernels"]
json.dump({"source": d["source"],
           "kernels": " + ".join(k.replace("void mvs::", "").replace("mvs::", "").split("(")[0] for k in stage) + " (default variant 1784)",
           "pairs_per_launch": d["pairs_per_launch"], "hbm_bytes_per_pair": d["hbm_bytes_per_pair"], "formula": d["formula"],
           "per_kernel_KB": {k: {c: d["kernels"][k].get(c) for c in ("FETCH_SIZE", "WRITE_SIZE")} for k in stage},
           "note": old["note"]}, open("profiles/r02_ransac_hbm_traffic.json", "w"), indent=1)
b = json.load(open("profiles/bench_r02_default.json"))
print("bench:", b["value"], "pairs/s", b["ms_per_step"], "ms/step  frac", b["roofline"]["frac"], "achieved", b["roofline"]["achieved"],
      "launch_ms", b["roofline"]["launch_ms"], b["kernel_ms"])
print("pcie:", b.get("pcie_inclusive_pairs_per_s"), "naive", b.get("pcie_inclusive_pairs_per_s_naive"), " ref threshold:", b["reference_threshold"],
      " single pair ms:", b["single_pair_ms"], " cpu:", b["cpu_baseline"]["value"], b["cpu_baseline"]["single_thread_pairs_per_s"])
u = json.load(open("profiles/bench_r02_under_rocprof.json"))
print("under rocprof:", u["value"], u["roofline"]["launch_ms"])
for r in csv.DictReader(open("profiles/r02_rocprofv3_kernel_stats.csv")):
    print("  %-64s %4s calls  %.3f ms avg" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e6))
for f in ("profiles/bench_r02_sequence_config5.json", "profiles/bench_r02_sequence_config5_refit.json"):
    s = json.load(open(f))
    print(f, s["value"], "frames/s", s["ms_per_sequence"], "ms", s["stage_ms"])
