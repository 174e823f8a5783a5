#!/usr/bin/env python3
"""VisualFeature::extract as the reference calls it: ONE 640x480 frame per call (vision/visual-feature.cpp:40-49), host
buffers in and out through mvs_extract; wall clock per call and the kernel time of the call's graph.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mvslam_amd import capi

rng = np.random.default_rng(7)
base = rng.integers(0, 256, size=(82, 109)).astype(np.uint8)
img = np.clip(np.kron(base, np.ones((6, 6), np.uint8))[:480, :640].astype(np.int32) + rng.integers(-6, 7, size=(480, 640)), 0, 255).astype(np.uint8)[None]
ctx = capi.Context(0)
out = {}
for nf in (500, 2000):                      # the reference's MAX_FEATURE_COUNT (visual-feature.cpp:9) and the matching configs' count
    prm = capi.default_orb_params(nfeatures=nf)
    pin = capi.pinned_empty(img.shape, np.uint8)
    pin[...] = img
    po = dict(kp=capi.pinned_empty((1, nf), capi.KEYPOINT_DTYPE), desc=capi.pinned_empty((1, nf, 32), np.uint8), n=capi.pinned_empty((1,), np.int32))
    for _ in range(5):
        ctx.extract(pin, prm, out=po)
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.extract(pin, prm, out=po)
    wall = (time.perf_counter() - t0) / 200 * 1e3
    out["nfeatures_%d" % nf] = {"wall_ms_per_call": round(wall, 4), "kernel_ms_per_call": round(ctx.extract_time(steps=50), 4),
                                "keypoints": int(po["n"][0])}
ctx.close()
print(json.dumps({"what": "one 640x480 frame per mvs_extract call, pinned host buffers", **out}))
