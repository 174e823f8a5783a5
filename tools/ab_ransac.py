#!/usr/bin/env python3
"""A/B timing of the co-compiled ransac_kernel variants, interleaved rounds in ONE process (cdna guide rule 24).
usage: python tools/ab_ransac.py [--pairs 128] [--rounds 5] [--variants 0,4,1,3,7,5]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MVS_USE_DEBUG_LIB"] = "1"   # the variant switch exists in the diagnostics build only
from mvslam_amd import capi, synth  # noqa: E402

NAMES = {0: "round-1 first version", 120: "fused: LDS point stream + in-place rotation + mask-fma + unscaled sqrt/div",
         632: "120 split into a solve and a scoring launch (round-1 default)", 376: "timing only: 120 without V rotations",
         760: "632 + sqrt-free convergence test in the solve", 1656: "632 + pruned point-per-lane scoring",
         1784: "760 + pruned point-per-lane scoring (default)", 3832: "1784 with the solve as A / V wavefront pairs",
         5880: "1784 before the 3x3 SVD of the solve moved to the unscaled sequences"}

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=128)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--hyp", type=int, default=50000)
ap.add_argument("--kp", type=int, default=2000)
ap.add_argument("--variants", default="0,120")
ap.add_argument("--max-error-sq", type=float, default=1e-2, help="<= 0: the reference formula 5e-2 / K00 / K11")
ap.add_argument("--no-check", action="store_true", help="do not compare results (timing-only experimental variants)")
args = ap.parse_args()
variants = [int(v) for v in args.variants.split(",")]

ctx = capi.Context(0)
data = synth.make_batch(0, args.pairs, n_kp=args.kp)
b = capi.Batch(ctx, args.pairs, args.kp, 32)
b.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"], data["global_index"])
prm = capi.default_params(num_hypotheses=args.hyp, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=args.max_error_sq)
lib = capi.lib()
ref = None
t = {v: [] for v in variants}
for rnd in range(args.rounds + 1):
    for v in variants:
        lib.mvs_debug_set_ransac_variant(C.c_int(v))
        _, k = b.time(prm, steps=1, warmup=0)
        if rnd:
            t[v].append(k["ransac"])
        res = b.download(matches=False, mask=False, points=False)["results"]
        if ref is None:
            ref = res.tobytes()
        assert args.no_check or res.tobytes() == ref, "variant %d changes the results" % v
for v in variants:
    a = np.array(t[v])
    print("variant %d %-18s ransac ms/launch: median %.3f  min %.3f  (%d pairs -> %.1f us/pair)"
          % (v, NAMES.get(v, "?"), np.median(a), a.min(), args.pairs, np.median(a) * 1e3 / args.pairs))
b.close()
ctx.close()
