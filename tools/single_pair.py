#!/usr/bin/env python3
"""Per-kernel timing of ONE pair at a time (BASELINE configs[1])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvslam_amd import capi, synth
ctx = capi.Context(0)
for n in (1, 2, 8):
    d = synth.make_batch(0, n, n_kp=2000)
    b = capi.Batch(ctx, n, 2000, 32)
    b.upload(0, d["desc1"], d["kp1"], d["n1"], d["desc2"], d["kp2"], d["n2"], d["K"], d["global_index"])
    prm = capi.default_params(num_hypotheses=50000, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=1e-2)
    tot, k = b.time(prm, steps=20, warmup=3)
    print("pairs=%d  total %.3f ms/step  kernels(ms/step): %s" % (n, tot / 20, {a: round(v / 20, 4) for a, v in k.items()}))
    b.close()
ctx.close()
