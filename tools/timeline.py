#!/usr/bin/env python3
"""Timeline of one pipeline step from a rocprofv3 --kernel-trace CSV: kernel, queue, start, end, duration (us).
usage: python tools/timeline.py s_kernel_trace.csv [step index]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "match_mfma" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) - 2
s = idx[k]
e = next((i for i in idx if i > s + 1 and int(rows[i]["Start_Timestamp"]) > int(rows[s]["End_Timestamp"]) + 2_000_000), len(rows))
t0 = int(rows[s]["Start_Timestamp"])
for r in rows[s:min(e, s + 45)]:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-40s q%s %8.1f %8.1f  %7.1f us" % (r["Kernel_Name"].replace("void mvs::", "").replace("mvs::", "").split("(")[0][:40],
                                              r["Queue_Id"], a / 1e3, b / 1e3, (b - a) / 1e3))
