#!/usr/bin/env python3
"""Digest of hipcc's -Rpass-analysis=kernel-resource-usage remarks: {mangled kernel name: {vgprs, agprs, sgprs, scratch,
lds, occupancy}}.  Build-time helper of the Makefile in this directory; bench.py reads the result next to what the runtime
reports (mvs_kernel_info_get), so the register split of a kernel in the bench line is the build's, never typed in."""
import json
import re
import sys

KEYS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
        "Occupancy [waves/SIMD]": "occupancy_waves_per_simd", "LDS Size [bytes/block]": "static_lds_bytes",
        "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills"}


def main(out, files):
    table = {}
    cur = None
    for f in files:
        for line in open(f, errors="replace"):
            m = re.search(r"remark:\s+(?:\S+:\d+:\d+:\s+)?Function Name: (\S+)", line)
            if m:
                cur = table.setdefault(m.group(1), {})
                continue
            m = re.search(r"remark:\s+(?:\S+:\d+:\d+:\s+)?([A-Za-z][^:]*): (\d+) \[-Rpass-analysis", line)
            if m and cur is not None and m.group(1).strip() in KEYS:
                cur[KEYS[m.group(1).strip()]] = int(m.group(2))
    json.dump(table, open(out, "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
