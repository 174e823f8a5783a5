#!/usr/bin/env python3
"""BASELINE configs[4]: 1000-frame synthetic sequence on one MI355X: per frame match(prev, new) + two-view RANSAC +
triangulation (batched over all consecutive pairs) + on-device join + pnp_solve (batched over frames).  BA / the VO
state machine are not part of the path (GTSAM, out of scope).  Prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mvslam_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1000)
ap.add_argument("--kp", type=int, default=2000)
ap.add_argument("--hyp", type=int, default=50000)
ap.add_argument("--pnp-hyp", type=int, default=100)      # the reference's iterationsCount (pnp-solve.cpp:47)
ap.add_argument("--steps", type=int, default=3)
args = ap.parse_args()

t0 = time.time()
seq = synth.make_sequence(args.frames, n_kp=args.kp)
gen_s = time.time() - t0
ctx = capi.Context(0)
s = capi.Sequence(ctx, args.frames, args.kp, 32)
s.upload(0, seq["desc"], seq["kp"], seq["n_kp"], seq["K"])
prm = capi.default_params(num_hypotheses=args.hyp, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=1e-2)
pprm = capi.default_pnp_params(num_hypotheses=args.pnp_hyp, seed=7, reproj_error=2.0)
ms = s.time(prm, pprm, steps=args.steps, warmup=1) / args.steps
s.run(prm, pprm)
gp, gt = s.download_pairs(), s.download_tracks()
res, tr = gp["results"], gt["tracks"]
print(json.dumps({
    "metric": "frames/sec, 1000-frame synthetic sequence (match + two-view + PnP + triangulate per frame, no BA)",
    "value": round(args.frames / (ms * 1e-3), 1), "unit": "frames/s", "ms_per_sequence": round(ms, 2),
    "frames": args.frames, "keypoints": args.kp, "hypotheses": args.hyp, "pnp_hypotheses": args.pnp_hyp,
    "valid_pairs": int(res["valid"].sum()), "avg_matches": round(float(res["n_matches"].mean()), 1),
    "avg_points": round(float(res["n_points"].mean()), 1), "tracks_ok": int(tr["ok"].sum()),
    "avg_corr": round(float(tr["n_corr"].mean()), 1), "avg_pnp_inliers": round(float(tr["n_inliers"].mean()), 1),
    "data_generation_s": round(gen_s, 1)}))
