#!/usr/bin/env python3
"""Randomised GPU-vs-oracle soak over every entry point, resident sequences included (more cases than the test-suite keeps).  Integer / index outputs
must agree bit for bit, floating point within the tolerances of the tests.  Prints one JSON summary line; exit code 1
on any mismatch.  usage: python tools/soak.py [--cases 200] [--seed 2026]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mvslam_amd import capi, synth
import oracle_lib as o
import test_gpu_parity as TG
import test_pnp as TP
import test_refine as TR
import test_orb as TO
import test_sequence as TS

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=200)
ap.add_argument("--seed", type=int, default=2026, help="another seed = another universe of cases")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
PAIR0 = 10_000 + (args.seed - 2026) * 1000   # synthetic pair ids of part 1
ctx = capi.Context(0)
bad = []
cnt = dict(pairs=0, pairs_ref_threshold=0, pairs_match_settings=0, run_points=0, match=0, ransac=0, pnp=0, sfm_refine=0, pnp_refine=0,
           extract=0, sequence_frames=0, single_shot=0)
t0 = time.time()

# 1. whole image pairs through the batch API: ragged keypoint counts, varying noise / outliers / hypothesis counts
n_pairs = args.cases
sizes = rng.integers(40, 700, size=n_pairs)
N = int(sizes.max())
b = capi.Batch(ctx, n_pairs, N)
data = []
for i in range(n_pairs):
    p = synth.make_pair(PAIR0 + i, n_kp=int(sizes[i]), noise_px=float(rng.choice([0.0, 0.3, 1.0])),
                        outlier_frac=float(rng.choice([0.0, 0.3, 0.6])))
    data.append(p)
pad = lambda a, w: np.concatenate([a, np.zeros((N - len(a),) + a.shape[1:], a.dtype)])
b.upload(0, np.stack([pad(p["desc1"], 0) for p in data]), np.stack([pad(p["kp1"], 0) for p in data]), sizes.astype(np.int32),
         np.stack([pad(p["desc2"], 0) for p in data]), np.stack([pad(p["kp2"], 0) for p in data]), sizes.astype(np.int32),
         np.stack([p["K"].reshape(9) for p in data]), np.arange(n_pairs, dtype=np.int64))
H = 768
prm = capi.default_params(num_hypotheses=H, sampler=capi.SAMPLER_PHILOX, seed=99, max_error_sq=1e-2)
b.run(prm)
b.sync()
out = b.download()
# 1b. the same pairs at the reference threshold 5e-2 / K00 / K11 (sfm-solve.cpp:311): tiny inlier sets, MANY hypotheses
# tie at the best count, so the residual tie-break of ransac_select_kernel (both its paths) decides the winner
prm0 = capi.default_params(num_hypotheses=H, sampler=capi.SAMPLER_PHILOX, seed=5)
b.run(prm0)
b.sync()
out0 = b.download(points=False)
# 1c. (round 5) the batch matcher (matrix cores once the batch has 512 workgroups) under another (ratio, max_dist): no limit
# takes the uncapped kernel, a limit the capped one (only keys below the cap are inserted); match lists byte for byte
ratio_c, md_c = float(rng.choice([0.5, 0.7, 0.9])), float(rng.choice([-1.0, 3.0, 25.0, 64.0]))
b.run(capi.default_params(num_hypotheses=64, sampler=capi.SAMPLER_PHILOX, seed=1, max_error_sq=1e-2, ratio=ratio_c, max_dist=md_c))
b.sync()
outc = b.download(mask=False, points=False)
b.close()
for i, p in enumerate(data):
    n = int(sizes[i])
    want = o.match_visual_features(p["desc1"][:n], p["desc2"][:n], ratio_c, md_c)
    M = int(outc["results"][i]["n_matches"])
    cnt["pairs_match_settings"] += 1
    if not (M == len(want) and outc["matches"][i][:M].tobytes() == want.tobytes()):
        bad.append(("pair_match_settings", i, ratio_c, md_c))
# 1d. (round 5) mvs_batch_run_points: a batch of sfm_solve calls on the point pairs of random scenes
n_pts = max(8, args.cases // 2)
m_pts = rng.integers(0, 900, size=n_pts)
m_pts[:3] = [0, 7, 8]
Np = int(m_pts.max())
uvs1, uvs2 = np.zeros((n_pts, Np, 2)), np.zeros((n_pts, Np, 2))
Kp = np.array([[525.0, 0, 320], [0, 525, 240], [0, 0, 1]])
for i in range(n_pts):
    if m_pts[i]:
        a, c = TG._scene(int(rng.integers(0, 1 << 30)), int(m_pts[i]), float(rng.choice([0.0, 1e-4, 1e-3])), outliers=float(rng.choice([0.0, 0.3, 0.6])))
        uvs1[i, :m_pts[i]] = a * 525 + np.array([320, 240.0])
        uvs2[i, :m_pts[i]] = c * 525 + np.array([320, 240.0])
bp = capi.Batch(ctx, n_pts, Np, 32)
gidx_p = rng.integers(0, 1 << 20, size=n_pts).astype(np.int64)
bp.upload_intrinsics(0, Kp, gidx_p, count=n_pts)
thr_p, H_p, seed_p = float(rng.choice([1e-2, 1e-3, 0.0])), int(rng.choice([300, 768, 2049])), int(rng.integers(0, 1 << 30))
prm_p = capi.default_params(num_hypotheses=H_p, sampler=capi.SAMPLER_PHILOX, seed=seed_p, max_error_sq=thr_p)
bp.run_points(prm_p, uvs1, uvs2, m_pts)
bp.sync()
outp = bp.download(matches=False)
bp.close()
for i in range(n_pts):
    m, r = int(m_pts[i]), outp["results"][i]
    cnt["run_points"] += 1
    if m < 8:
        ok = not r["valid"] and int(r["n_matches"]) == m
    else:
        want = o.sfm_solve(uvs1[i, :m], uvs2[i, :m], Kp, o.make_params(H_p, o.SAMPLER_PHILOX, seed_p + int(gidx_p[i]), thr_p))
        ok = bool(r["valid"]) == want["ok"] and int(r["best_hyp"]) == want["best_hyp"] and int(r["best_count"]) == want["best_count"] \
            and float(r["best_residual"]) == want["best_residual"] and np.array_equal(outp["mask"][i][:m], want["mask"][:m])
        if ok and want["ok"]:
            k = want["n_points"]
            ok = int(r["n_points"]) == k and np.array_equal(outp["point_idx"][i][:k], want["point_idx"]) and \
                np.abs(r["R"] - want["R"]).max() <= 1e-12 and np.abs(r["t"] - want["t"]).max() <= 1e-12
    if not ok:
        bad.append(("run_points", i, m, thr_p, H_p))
for i, p in enumerate(data):
    want = o.image_pair(p["desc1"], p["kp1"], p["desc2"], p["kp2"], p["K"], o.make_params(H, o.SAMPLER_PHILOX, 5 + i, 0.0), 0.7, 10.0)
    r = out0["results"][i]
    M = int(r["n_matches"])
    ok = (bool(r["valid"]) == want["valid"] and int(r["best_hyp"]) == want["best_hyp"] and int(r["best_count"]) == want["best_count"]
          and float(r["best_residual"]) == want["best_residual"] and np.array_equal(out0["mask"][i][:M], want["mask"]))
    cnt["pairs_ref_threshold"] += 1
    if not ok:
        bad.append(("pair_ref_threshold", i))
for i, p in enumerate(data):
    n = int(sizes[i])
    want = o.image_pair(p["desc1"], p["kp1"], p["desc2"], p["kp2"], p["K"], o.make_params(H, o.SAMPLER_PHILOX, 99 + i, 1e-2), 0.7, 10.0)
    r = out["results"][i]
    M = int(r["n_matches"])
    ok = (bool(r["valid"]) == want["valid"] and M == want["n_matches"] and int(r["best_hyp"]) == want["best_hyp"]
          and int(r["n_points"]) == want["n_points"] and int(r["best_count"]) == want["best_count"])
    if ok and M:
        ok &= np.array_equal(out["matches"][i][:M], want["matches"].astype(capi.MATCH_DTYPE))
        ok &= np.array_equal(out["mask"][i][:M], want["mask"])
    if ok and want["valid"]:
        k = want["n_points"]
        ok &= np.array_equal(out["point_idx"][i][:k], want["point_idx"])
        ok &= np.abs(r["R"] - want["R"]).max() <= 1e-12 and np.abs(r["t"] - want["t"]).max() <= 1e-12
        ok &= np.abs(out["points"][i][:k] - want["points"]).max() <= 1e-9 * max(1.0, np.abs(want["points"]).max())
    cnt["pairs"] += 1
    if not ok:
        bad.append(("pair", i))

# 2. matcher alone: odd sizes, ties, all descriptor widths
for i in range(args.cases // 4):
    nb = int(rng.choice([16, 32, 64]))
    nt, nq = int(rng.integers(2, 900)), int(rng.integers(1, 900))
    tr = rng.integers(0, 256, size=(nt, nb), dtype=np.uint8)
    qu = rng.integers(0, 256, size=(nq, nb), dtype=np.uint8)
    k = min(nt, nq) // 2
    qu[:k] = tr[rng.integers(0, nt, size=k)]
    qu[:k, 0] ^= rng.integers(0, 4, size=k).astype(np.uint8)
    ratio, md = float(rng.choice([0.5, 0.7, 0.95])), float(rng.choice([-1.0, 10.0, 80.0]))
    got, want = ctx.match_hamming(tr, qu, ratio, md), o.match_visual_features(tr, qu, ratio, md)
    cnt["match"] += 1
    if not np.array_equal(got, want.astype(capi.MATCH_DTYPE)):
        bad.append(("match", i))

# 2b. the RANSAC stage alone with per-hypothesis tables: every count and residual sum, winner, mask, F -- bit for bit
for i in range(max(args.cases, 200)):
    m = int(rng.integers(8, 1800))
    Hh = int(rng.choice([1, 17, 256, 700, 2049]))
    thr = float(rng.choice([1e-7, 1e-4, 1e-3, 1e-2]))
    p1, p2 = TG._scene(7000 + i, m, float(rng.choice([0.0, 1e-4, 1e-3])), outliers=float(rng.choice([0.0, 0.3, 0.7])))
    seed = int(rng.integers(0, 1 << 40))
    want = o.ransac_fundamental(p1, p2, thr, Hh, o.SAMPLER_PHILOX, seed=seed, per_hyp=True)
    got = ctx.ransac_fundamental(p1, p2, thr, Hh, capi.SAMPLER_PHILOX, seed=seed, per_hyp=True)
    cnt["ransac"] += 1
    if not (np.array_equal(got["count"], want["count"]) and got["residual"].tobytes() == want["residual"].tobytes()
            and got["best_hyp"] == want["best_hyp"] and got["best_count"] == want["best_count"]
            and got["best_residual"] == want["best_residual"] and np.array_equal(got["mask"], want["mask"])
            and got["F"].tobytes() == want["F"].tobytes() and got["ok"] == want["ok"]):
        bad.append(("ransac", i, m, Hh, thr))

# 3. pnp_solve
for i in range(args.cases // 4):
    # (round 5: up to 2400 points -- the kernel streams them through LDS in 768-point chunks -- and hypothesis counts on
    # both sides of 64 / 128 / 192, where the number of wavefronts that share a block's points changes)
    npts = int(rng.integers(8, 600)) if i % 3 else int(rng.integers(600, 2400))
    K, X, uv, R, t, _ = TP._scene(500 + i, npts, float(rng.choice([0.0, 0.01, 0.5])), int(rng.integers(0, 5)))
    Hh, seed = int(rng.choice([1, 40, 64, 65, 100, 128, 129, 192, 193, 300])), int(rng.integers(0, 1 << 30))
    got = ctx.pnp_solve(X, uv, K, capi.default_pnp_params(num_hypotheses=Hh, seed=seed, reproj_error=1.0))
    want = o.pnp_solve(X, uv, K, o.make_pnp_params(Hh, o.SAMPLER_PHILOX, seed, 1.0))
    cnt["pnp"] += 1
    ok = got["ok"] == want["ok"]
    if ok and want["ok"]:
        ok = got["best_hyp"] == want["best_hyp"] and np.array_equal(got["inliers"], want["inliers"]) and \
            np.array_equal(got["R"], want["R"]) and np.array_equal(got["t"], want["t"])
    if not ok:
        bad.append(("pnp", i))

# 4. refinement
for i in range(args.cases // 8):
    m = int(rng.integers(1, 1500))
    pb = TR.two_view_problem(900 + i, m, K=synth.K_DEFAULT, sig=0.5, baseline=0.3, depth=(2.0, 10.0))
    got = ctx.sfm_refine(pb["p1"], pb["cov"], pb["p2"], pb["cov"], pb["K"], pb["Rg"], pb["tg"], pb["Xg"])
    want = o.sfm_refine(pb["p1"], pb["cov"], pb["p2"], pb["cov"], pb["K"], pb["Rg"], pb["tg"], pb["Xg"])
    cnt["sfm_refine"] += 1
    # pose: since round 3 the oracle factors the reduced system with the textbook (division-form) Cholesky again while the
    # kernel multiplies by stored reciprocals (the checker must not follow the kernel's arithmetic), so the two
    # sides stop up to ~1e-9 apart where the cost is flat, at the same error to 1e-15 (seed 3101, case 15: 5.7e-10 on t at
    # equal iteration counts and errors 810.0757243827807 / ...805): 1e-8 like pnp_refine below, + the error itself
    if not (got["ok"] and want["ok"] and got["iterations"] == want["iterations"] and np.abs(got["R"] - want["R"]).max() < 1e-8
            and np.abs(got["t"] - want["t"]).max() < 1e-8 and abs(got["error"] - want["error"]) <= 1e-12 * max(1.0, want["error"])
            # points: relative to their size (depths up to 10; the two sides' sin / cos differ in the last bit and a weakly
            # constrained depth amplifies it: 1.1e-9 seen once in 155 000 cases, at equal errors and iteration counts)
            and np.abs(got["points"] - want["points"]).max() < 1e-8 * max(1.0, np.abs(want["points"]).max())
            and np.abs(got["pose_cov"] - want["pose_cov"]).max() <= 1e-7 * np.abs(want["pose_cov"]).max()):
        bad.append(("sfm_refine", i, m, got["ok"], want["ok"], got["iterations"], want["iterations"], np.abs(got["R"] - want["R"]).max(),
                    np.abs(got["t"] - want["t"]).max(), np.abs(got["points"] - want["points"]).max(), got["error"], want["error"]))
    pp = TR.pnp_problem(1300 + i, int(rng.integers(7, 800)))
    got = ctx.pnp_refine(pp["X"], pp["wcov"], pp["uv"], pp["icov"], pp["K"], pp["Rg"], pp["tg"])
    want = o.pnp_refine(pp["X"], pp["wcov"], pp["uv"], pp["icov"], pp["K"], pp["Rg"], pp["tg"])
    cnt["pnp_refine"] += 1
    # where the cost is flat (pose weakly constrained next to loose point priors) the two sides stop up to ~1e-9 apart
    # at the same error to 1e-15: tolerance 1e-8 on the pose, 1e-12 relative on the error
    if not (got["ok"] and want["ok"] and np.abs(got["R"] - want["R"]).max() < 1e-8 and np.abs(got["t"] - want["t"]).max() < 1e-8
            and abs(got["error"] - want["error"]) <= 1e-12 * max(1.0, want["error"])):
        bad.append(("pnp_refine", i, len(pp["X"]), got["ok"], want["ok"], got["iterations"], want["iterations"],
                    np.abs(got["R"] - want["R"]).max(), np.abs(got["t"] - want["t"]).max(), got["error"], want["error"]))

# 5. extraction: random sizes and parameters
# (round 5: one to twelve images per call -- the kernels' block orders hand image b to XCD b mod 8 --, a noise or a
# binary-block image among them now and then: thousands of corners with equal FAST scores for the radix select)
for i in range(args.cases // 8):
    h, w = int(rng.integers(64, 400)), int(rng.integers(64, 500))
    nimg = int(rng.integers(1, 13))
    imgs = [TO.textured(3000 + 16 * i + j, h, w) for j in range(nimg)]
    if i % 3 == 0:
        imgs[int(rng.integers(0, nimg))] = rng.integers(0, 200, size=(h, w)).astype(np.uint8)
    if i % 4 == 1:
        c = int(rng.integers(3, 7))
        imgs[int(rng.integers(0, nimg))] = np.kron(rng.integers(0, 2, size=(h // c + 1, w // c + 1)).astype(np.uint8) * 255,
                                                   np.ones((c, c), np.uint8))[:h, :w]
    imgs = np.stack(imgs)
    kw = dict(nfeatures=int(rng.integers(1, 900)), nlevels=int(rng.integers(1, 9)), fast_threshold=int(rng.choice([5, 20, 60])),
              edge_threshold=int(rng.choice([19, 31])))
    try:
        got = ctx.extract(imgs, capi.default_orb_params(**kw))
    except capi.MvsError as e:      # more than 16384 corners at one level of some image: reported, not a mismatch
        if e.status != capi.MVS_ERR_CAPACITY:
            raise
        cnt["extract_capacity"] = cnt.get("extract_capacity", 0) + 1
        continue
    for j in range(nimg):
        want = o.orb_extract(imgs[j], o.make_orb_params(**kw))
        n = int(got["n"][j])
        cnt["extract"] += 1
        if not (n == len(want["kp"]) and np.array_equal(got["kp"][j][:n], want["kp"].astype(capi.KEYPOINT_DTYPE))
                and np.array_equal(got["desc"][j][:n], want["desc"])):
            bad.append(("extract", i, j, h, w, kw))
# 6. resident sequences: random lengths / sizes / hypothesis counts; pair views, the join, the batched PnP and the
# trajectory fold against the composed oracle, bit for bit
for i in range(max(1, args.cases // 40)):
    F, N = int(rng.integers(3, 9)), int(rng.integers(150, 450))
    seq = synth.make_sequence(F, n_kp=N, n_map=int(rng.integers(3000, 7000)), noise_px=float(rng.choice([0.0, 0.3, 0.8])),
                              seed=int(rng.integers(1, 1 << 30)))
    prm = dict(H=int(rng.integers(200, 900)), seed=int(rng.integers(0, 1 << 20)), thr=1e-2)
    pprm = dict(H=int(rng.integers(50, 300)), seed=int(rng.integers(0, 1 << 20)), err=float(rng.choice([1.0, 2.0])))
    sq = capi.Sequence(ctx, F, N, 32)
    sq.upload(0, seq["desc"], seq["kp"], seq["n_kp"], seq["K"])
    sq.run(capi.default_params(num_hypotheses=prm["H"], sampler=capi.SAMPLER_PHILOX, seed=prm["seed"], max_error_sq=prm["thr"]),
           capi.default_pnp_params(num_hypotheses=pprm["H"], seed=pprm["seed"], reproj_error=pprm["err"]))
    gp, gt, tr = sq.download_pairs(), sq.download_tracks(), sq.download_trajectory()
    sq.close()
    pairs, tracks = TS.oracle_sequence(seq, prm, pprm)
    ok = True
    for k, ref in enumerate(pairs):
        r, M = gp["results"][k], ref["n_matches"]
        ok &= r["n_matches"] == M and gp["matches"][k][:M].tobytes() == ref["matches"].tobytes()
        ok &= bool(r["valid"]) == ref["ok"] and np.array_equal(gp["mask"][k][:M], ref["mask"])
        if ok and ref["ok"]:
            n = ref["n_points"]
            ok &= np.array_equal(gp["point_idx"][k][:n], ref["point_idx"]) and gp["points"][k][:n].tobytes() == ref["points"].tobytes()
    for q, ref in enumerate(tracks):
        t, nc = gt["tracks"][q], len(ref["X"])
        ok &= t["n_corr"] == nc and gt["corr_xyz"][q][:nc].tobytes() == ref["X"].tobytes() and gt["corr_uv"][q][:nc].tobytes() == ref["uv"].tobytes()
        ok &= bool(t["ok"]) == ref["ok"] and t["best_hyp"] == ref["best_hyp"]
        if ok and ref["ok"]:
            ni = len(ref["inliers"])
            ok &= t["n_inliers"] == ni and np.array_equal(gt["inlier_idx"][q][:ni], ref["inliers"])
            ok &= t["R"].tobytes() == ref["R"].tobytes() and t["t"].tobytes() == ref["t"].tobytes()
    res, trk = gp["results"], gt["tracks"]
    want = o.seq_chain(res["R"], res["t"], res["valid"], trk["R"], trk["t"], trk["ok"])
    for key in ("R", "t", "pair_scale", "track_scale"):
        ok &= tr[key].tobytes() == want[key].tobytes()
    cnt["sequence_frames"] += F
    if not ok:
        bad.append(("sequence", i, F, N, prm, pprm))
# 7. the single-shot entry points (one pair per call: the fused small-launch kernel path, the pinned staging arena):
# mvs_image_pair against the oracle's composition, and the two separate calls against the single pass
for i in range(args.cases // 20):
    n_kp = int(rng.integers(30, 600))
    p = synth.make_pair(50_000 + (args.seed - 2026) * 1000 + i, n_kp=n_kp, noise_px=float(rng.choice([0.0, 0.3, 1.0])),
                        outlier_frac=float(rng.choice([0.0, 0.3, 0.6])))
    Hs, sd = int(rng.integers(1, 700)), int(rng.integers(0, 1 << 20))
    thr = float(rng.choice([1e-2, 0.0]))
    prm1 = capi.default_params(num_hypotheses=Hs, sampler=capi.SAMPLER_PHILOX, seed=sd, max_error_sq=thr)
    got = ctx.image_pair(p["desc1"], p["kp1"], p["desc2"], p["kp2"], p["K"], prm1)
    ref = o.image_pair(p["desc1"], p["kp1"], p["desc2"], p["kp2"], p["K"], o.make_params(Hs, o.SAMPLER_PHILOX, sd, thr), 0.7, 10.0)
    M = ref["n_matches"]
    ok = got["ok"] == ref["ok"] and got["n_matches"] == M and got["matches"].tobytes() == ref["matches"].tobytes()
    ok = ok and got["best_hyp"] == ref["best_hyp"] and got["best_count"] == ref["best_count"] and np.array_equal(got["mask"], ref["mask"])
    if ok and ref["ok"]:
        ok = np.array_equal(got["point_idx"], ref["point_idx"]) and np.abs(got["R"] - ref["R"]).max() <= 1e-12 and \
            np.abs(got["t"] - ref["t"]).max() <= 1e-12 and \
            np.abs(got["points"] - ref["points"]).max() <= 1e-12 * max(1.0, np.abs(ref["points"]).max())
    if ok and M >= 8:
        mt = ctx.match_hamming(p["desc1"], p["desc2"], 0.7, 10.0)
        two = ctx.two_view(p["kp1"][mt["trainIdx"]].astype(np.float64), p["kp2"][mt["queryIdx"]].astype(np.float64), p["K"], prm1)
        ok = mt.tobytes() == got["matches"].tobytes() and two["ok"] == got["ok"] and two["best_hyp"] == got["best_hyp"]
        if ok and got["ok"]:
            ok = two["R"].tobytes() == got["R"].tobytes() and two["points"].tobytes() == got["points"].tobytes()
    cnt["single_shot"] += 1
    if not ok:
        bad.append(("single_shot", i, n_kp, Hs, sd, thr))
ctx.close()
print(json.dumps({"seed": args.seed, "cases": cnt, "mismatches": len(bad), "first_mismatches": [list(map(str, x)) for x in bad[:8]],
                  "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
