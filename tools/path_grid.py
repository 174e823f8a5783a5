#!/usr/bin/env python3
"""Fused hypothesis-per-lane kernel against the pre-screened stage over (pairs, hypotheses): ms per launch (diagnostics build; the
split minimum forces the path: 100 = fused, 1 = pre-screened).  MVS_USE_DEBUG_LIB=1 python tools/path_grid.py"""
import ctypes as C
import json
import os
import sys

os.environ["MVS_USE_DEBUG_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvslam_amd import capi, synth  # noqa: E402

ctx = capi.Context(0)
lib = capi.lib()
out = []
for n in (2, 3, 8, 32, 128):
    d = synth.make_batch(0, n, n_kp=2000)
    b = capi.Batch(ctx, n, 2000, 32)
    b.upload(0, d["desc1"], d["kp1"], d["n1"], d["desc2"], d["kp2"], d["n2"], d["K"], d["global_index"])
    for H in (256, 1024, 4096, 16384, 50000):
        prm = capi.default_params(num_hypotheses=H, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=1e-2)
        row = dict(pairs=n, hypotheses=H)
        ref = None
        for name, split in (("fused_ms", 100), ("prescreened_ms", 1)):
            lib.mvs_debug_set_split_min_pairs(C.c_int(split))
            b.run(prm)
            b.sync()
            res = b.download(matches=False, mask=False, points=False)["results"].tobytes()
            ref = ref or res
            assert res == ref
            row[name] = round(b.time(prm, steps=20, warmup=3, per_kernel=False)[0] / 20, 4)
        out.append(row)
        print(json.dumps(row), flush=True)
    b.close()
lib.mvs_debug_set_split_min_pairs(C.c_int(3))
