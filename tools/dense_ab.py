#!/usr/bin/env python3
"""A/B of the dense matrix-core counting phase's shape (diagnostics build): threads per workgroup x batches per workgroup x
software pipeline, timed on the bench batch.  MVS_USE_DEBUG_LIB=1 python tools/dense_ab.py"""
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
names = {1: "product default", 10: "256 x 8, plain", 15: "512 x 4, plain", 16: "512 x 4, plain, 640-point chunks (2 workgroups per CU)",
         17: "512 x 4, plain, 672-point chunks", 18: "512 x 2, plain, 672-point chunks"}
ref = None
out = {}
for rep in range(2):
    for v, nm in names.items():
        lib.mvs_debug_set_count_dense(C.c_int(v))
        b.run(prm)
        b.sync()
        res = b.download(matches=False, mask=False, points=False)["results"].tobytes()
        ref = ref or res
        t = {}
        for n, ms in b.time_kernels(prm, steps=5):
            t[n] = t.get(n, 0) + ms
        dense = sum(ms for k, ms in t.items() if "count_mfma" in k)
        out.setdefault(nm, []).append(round(dense, 4))
        assert res == ref, "variant %d changes the results" % v
lib.mvs_debug_set_count_dense(C.c_int(1))
print(json.dumps(out))
