#!/usr/bin/env python3
"""Row f4: ImagePair::refine for a resident batch (config 3 shape: 512 pairs x 2000 keypoints).  The batch is run once
(match + RANSAC + triangulate), then `--steps` refinement passes are timed by wall clock around run + sync (inputs and
results stay in HBM).  The CPU oracle refines a sample of the same pairs beside it.  Prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mvslam_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=512)
ap.add_argument("--kp", type=int, default=2000)
ap.add_argument("--hyp", type=int, default=50000)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--cpu-pairs", type=int, default=32)
args = ap.parse_args()

data = synth.make_batch(0, args.pairs, n_kp=args.kp)
ctx = capi.Context(0)
b = capi.Batch(ctx, args.pairs, args.kp)
b.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"], data["global_index"])
prm = capi.default_params(num_hypotheses=args.hyp, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=1e-2)
b.run(prm)
b.sync()
rp = capi.default_refine_params()
b.refine(rp, 0.5)
b.sync()
t0 = time.perf_counter()
for _ in range(args.steps):
    b.refine(rp, 0.5)
b.sync()
ms = (time.perf_counter() - t0) * 1e3 / args.steps
out = b.download()
ref = b.download_refined(points=True)
b.close()
ctx.close()
res, rr = out["results"], ref["refined"]
ok = rr["ok"] == 1
npts = res["n_points"][ok].astype(np.int64)
its = rr["iterations"][ok].astype(np.int64)
# algorithmic work: per linear solve every point is linearised twice (Schur pass + step pass) and its candidate cost is
# evaluated once; + one covariance pass.  flop counts per point from the source (mul + add, 2 cameras):
#   linearise 2 x 262, Schur update 78 x 6 + 36 x 5 + 12 x 6 + 3x3 inverse 40 = 760, step 36 x 2 + 15 + cost 2 x 45
LIN, SCHUR, STEP, COST, COV = 524, 760, 90, 90, 1100
flops = float(np.sum(npts * (its * (2 * LIN + SCHUR + STEP + COST) + (LIN + SCHUR + COV))))
cpu = None
if args.cpu_pairs > 0:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle_lib as o
    K = synth.K_DEFAULT
    idx = [p for p in range(args.pairs) if ok[p]][:args.cpu_pairs]
    t0 = time.perf_counter()
    worst = 0.0
    for p in idx:
        n = int(res["n_points"][p])
        mt = out["matches"][p][out["point_idx"][p][:n]]
        p1 = data["kp1"][p][mt["trainIdx"]].astype(np.float64)
        p2 = data["kp2"][p][mt["queryIdx"]].astype(np.float64)
        cov = np.tile((np.eye(2) * 0.25).reshape(4), (n, 1))
        w = o.sfm_refine(p1, cov, p2, cov, K, res["R"][p], res["t"][p], out["points"][p][:n])
        worst = max(worst, np.abs(w["t"] - rr["t"][p]).max(), np.abs(w["R"] - rr["R"][p]).max())
    dt = time.perf_counter() - t0
    cpu = {"value": round(len(idx) / dt, 2), "unit": "pairs/s", "cores": 1, "kind": "port",
           "sample": "%d pairs of the same batch, single thread" % len(idx), "max_pose_diff_vs_gpu": float(worst)}
# reprojection RMS before / after (pixels), all valid pairs
print(json.dumps({
    "metric": "refined image-pairs/sec (ImagePair::refine = sfm_refine, batched on device)",
    "value": round(args.pairs / (ms * 1e-3), 1), "unit": "pairs/s", "ms_per_batch": round(ms, 3), "pairs": args.pairs,
    "refined_ok": int(ok.sum()), "mean_points": float(npts.mean()), "mean_linear_solves": float(its.mean()),
    "max_linear_solves": int(its.max()), "algorithmic_gflop_per_batch": round(flops / 1e9, 3),
    "achieved_tflops": round(flops / (ms * 1e-3) / 1e12, 3), "cpu_baseline": cpu}))
