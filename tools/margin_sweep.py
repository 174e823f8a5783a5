#!/usr/bin/env python3
"""Sweep of the dense phase's margin (points beyond M - B0 it covers for every hypothesis) on the bench batch, diagnostics build:
ms per 512-pair step and of the counting kernels, results byte-identical.  MVS_USE_DEBUG_LIB=1 python tools/margin_sweep.py"""
import ctypes as C
import json
import os
import sys

os.environ["MVS_USE_DEBUG_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvslam_amd import capi, synth  # noqa: E402

P = int(os.environ.get("PAIRS", "512"))
data = synth.make_batch(0, P, n_kp=2000)
ctx = capi.Context(0)
b = capi.Batch(ctx, P, 2000, 32)
b.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"], data["global_index"])
prm = capi.default_params(num_hypotheses=50000, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=1e-2)
lib = capi.lib()
ref = None
out = []
for rep in range(2):
  for pilot in [int(x) for x in os.environ.get("PILOTS", "1024").split(",")]:
    lib.mvs_debug_set_count_dense(C.c_int(100000 + pilot))
    for margin in [int(x) for x in os.environ.get("MARGINS", "32,64,96,128,192,256").split(",")]:
        lib.mvs_debug_set_count_dense(C.c_int(1000 + margin))
        b.run(prm)
        b.sync()
        res = b.download(matches=False, mask=False, points=False)["results"].tobytes()
        ref = ref or res
        assert res == ref, "margin %d changes the results" % margin
        step = b.time(prm, steps=10, warmup=2, per_kernel=False)[0] / 10
        t = {}
        for n, ms in b.time_kernels(prm, steps=3):
            t[n] = t.get(n, 0) + ms
        short = lambda k: k.split("<")[0].replace("ransac_", "").replace("_kernel", "") + ("_pilot" if k.endswith(", true>") else "")
        cnt = {short(k): round(v, 3) for k, v in t.items() if any(s in k for s in ("count_mfma", "finish", "survivors", "exact_list", "select"))}
        out.append(dict(pilot=pilot, margin=margin, ms_per_step=round(step, 3), counting_ms=round(sum(cnt.values()), 3), **cnt))
        print(json.dumps(out[-1]), flush=True)
lib.mvs_debug_set_count_dense(C.c_int(1000 + 64))
lib.mvs_debug_set_count_dense(C.c_int(100000 + 1024))
