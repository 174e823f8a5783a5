"""Row f2: device-resident frame sequence = batched ImagePair over consecutive frames + on-device join + batched
pnp_solve.  The oracle side composes image_pair -> (numpy join) -> pnp_solve per frame."""
import numpy as np
import pytest

import helpers
import oracle_lib as o
from mvslam_amd import synth


def _pool(fn, n_items, threads):
    """fn(k) for k < n_items on `threads` threads (the oracle's ctypes calls release the GIL)"""
    import threading

    out, err = [None] * n_items, []

    def work(k0):
        try:
            for k in range(k0, n_items, threads):
                out[k] = fn(k)
        except Exception as e:
            err.append(e)

    if threads <= 1:
        work(0) if n_items else None
        threads = 1
    else:
        ths = [threading.Thread(target=work, args=(k,)) for k in range(min(threads, n_items))]
        [t.start() for t in ths]
        [t.join() for t in ths]
    if err:
        raise err[0]
    return out


def oracle_sequence(seq, prm, pprm, ratio=0.7, max_dist=10.0, threads=1):
    F = len(seq["n_kp"])
    K = seq["K"]

    def pair(k):
        op = o.make_params(prm["H"], o.SAMPLER_PHILOX, prm["seed"] + k, prm["thr"])
        a, b = seq["n_kp"][k], seq["n_kp"][k + 1]
        return o.image_pair(seq["desc"][k][:a], seq["kp"][k][:a], seq["desc"][k + 1][:b], seq["kp"][k + 1][:b], K,
                            op, ratio, max_dist)

    pairs = _pool(pair, F - 1, threads)

    def track(q):
        pa, pb = pairs[q], pairs[q + 1]
        X, uv = np.zeros((0, 3)), np.zeros((0, 2))
        if pa["ok"] and len(pa["point_idx"]) and len(pb["matches"]):
            # join: point j of pair q belongs to frame q+1's keypoint queryIdx[point_idx[j]]; pair q+1's matches, in
            # their order, pick it up by trainIdx (image-pair.cpp:158-167, visual-odometer.cpp:384-445)
            tbl = np.full(int(seq["n_kp"][q + 1]) + 1, -1, dtype=np.int64)
            tbl[pa["matches"]["queryIdx"][pa["point_idx"]]] = np.arange(len(pa["point_idx"]))
            j = tbl[pb["matches"]["trainIdx"]]
            hit = j >= 0
            X = pa["points"][j[hit]].reshape(-1, 3)
            uv = seq["kp"][q + 2][pb["matches"]["queryIdx"][hit]].astype(np.float64).reshape(-1, 2)
        r = dict(ok=False, inliers=np.zeros(0, np.int64), best_hyp=-1)
        if len(X) >= 7:
            r = o.pnp_solve(X, uv, K, o.make_pnp_params(pprm["H"], o.SAMPLER_PHILOX, pprm["seed"] + q, pprm["err"]))
        r["X"], r["uv"] = X, uv
        return r

    tracks = _pool(track, F - 2, threads)
    return pairs, tracks


def test_sequence_generator_is_consistent():
    s = synth.make_sequence(4, n_kp=300, n_map=3000)
    assert s["desc"].shape == (4, 300, 32) and s["kp"].dtype == np.float32
    s2 = synth.make_sequence(4, n_kp=300, n_map=3000)
    assert np.array_equal(s["desc"], s2["desc"]) and np.array_equal(s["kp"], s2["kp"])
    # consecutive frames share most of their map points: the matcher finds them
    m = o.match_visual_features(s["desc"][0], s["desc"][1], 0.7, 10.0)
    assert len(m) > 100


@pytest.mark.gpu
def test_gpu_sequence_matches_oracle(ctx):
    from mvslam_amd import capi

    F, N = 6, 500
    seq = synth.make_sequence(F, n_kp=N, n_map=6000, noise_px=0.3)
    prm = dict(H=600, seed=4242, thr=1e-2)
    pprm = dict(H=300, seed=99, err=2.0)
    s = capi.Sequence(ctx, F, N, 32)
    s.upload(0, seq["desc"], seq["kp"], seq["n_kp"], seq["K"])
    s.run(capi.default_params(num_hypotheses=prm["H"], sampler=capi.SAMPLER_PHILOX, seed=prm["seed"], max_error_sq=prm["thr"]),
          capi.default_pnp_params(num_hypotheses=pprm["H"], seed=pprm["seed"], reproj_error=pprm["err"]))
    gp, gt = s.download_pairs(), s.download_tracks()
    s.close()
    pairs, tracks = oracle_sequence(seq, prm, pprm)
    for k, ref in enumerate(pairs):                                   # the zero-copy pair views behave like a batch
        r = gp["results"][k]
        M = ref["n_matches"]
        assert r["n_matches"] == M and gp["matches"][k][:M].tobytes() == ref["matches"].tobytes()
        assert bool(r["valid"]) == ref["ok"] and np.array_equal(gp["mask"][k][:M], ref["mask"])
        if ref["ok"]:
            n = ref["n_points"]
            assert np.array_equal(gp["point_idx"][k][:n], ref["point_idx"])
            assert gp["points"][k][:n].tobytes() == ref["points"].tobytes()
    n_ok = 0
    for q, ref in enumerate(tracks):
        t = gt["tracks"][q]
        nc = len(ref["X"])
        assert t["n_corr"] == nc                                       # the join: same correspondences, same order
        assert gt["corr_xyz"][q][:nc].tobytes() == ref["X"].tobytes()
        assert gt["corr_uv"][q][:nc].tobytes() == ref["uv"].tobytes()
        assert bool(t["ok"]) == ref["ok"] and t["best_hyp"] == ref["best_hyp"]
        if ref["ok"]:
            n_ok += 1
            ni = len(ref["inliers"])
            assert t["n_inliers"] == ni and np.array_equal(gt["inlier_idx"][q][:ni], ref["inliers"])   # bit-exact
            assert helpers.rel_err(t["R"], ref["R"]) <= 1e-4 and helpers.rel_err(t["t"], ref["t"]) <= 1e-4
            assert t["R"].tobytes() == ref["R"].tobytes() and t["t"].tobytes() == ref["t"].tobytes()
    assert n_ok >= F - 3                                                # the tracks actually solve


@pytest.mark.gpu
def test_gpu_sequence_recovers_motion(ctx):
    """Ground truth: the PnP pose of frame q+2 in frame q's camera frame matches the synthetic trajectory up to the
    unknown two-view scale (|t| of pair q is normalised to 1)."""
    from mvslam_amd import capi

    F, N = 5, 800
    seq = synth.make_sequence(F, n_kp=N, n_map=8000, noise_px=0.1, step=0.15)
    s = capi.Sequence(ctx, F, N, 32)
    s.upload(0, seq["desc"], seq["kp"], seq["n_kp"], seq["K"])
    s.run(capi.default_params(num_hypotheses=4096, sampler=capi.SAMPLER_PHILOX, seed=1, max_error_sq=2e-3),
          capi.default_pnp_params(num_hypotheses=512, seed=2, reproj_error=1.0))
    gp, gt = s.download_pairs(), s.download_tracks()
    s.close()
    for q in range(F - 2):
        t = gt["tracks"][q]
        assert t["ok"] and t["n_inliers"] >= 30
        (R0, t0), (R1, t1), (R2, t2) = seq["poses"][q], seq["poses"][q + 1], seq["poses"][q + 2]
        scale = np.linalg.norm(R1 @ (-R0.T @ t0) + t1)                  # true baseline of pair q
        Rrel = R0 @ R2.T                                                 # frame q+2 camera in frame q coordinates
        crel = R0 @ (-R2.T @ t2) + t0
        assert np.abs(t["R"] - Rrel).max() < 0.02
        assert np.abs(t["t"] * scale - crel).max() < 0.05 * max(1.0, np.linalg.norm(crel))


def test_oracle_seq_chain_on_exact_poses():
    """scale propagation (visual-odometer.cpp:577-588) on noise-free inputs: pairs with unit baselines and tracks in their
    pair's scale reproduce the true trajectory up to the first baseline; a failed track falls back to the two-view pose"""
    seq = synth.make_sequence(8, n_kp=50, n_map=500, step=0.07, yaw_step=0.02)
    P = seq["poses"]                                     # world -> camera
    cam = [(R.T, -R.T @ t) for R, t in P]                # camera in world
    rel = lambda a, b: (cam[a][0].T @ cam[b][0], cam[a][0].T @ (cam[b][1] - cam[a][1]))   # pose of frame b in frame a
    base = [np.linalg.norm(rel(k, k + 1)[1]) * (1.0 + 0.3 * k) ** 0 for k in range(7)]
    pR = [rel(k, k + 1)[0] for k in range(7)]
    pt = [rel(k, k + 1)[1] / base[k] for k in range(7)]
    tR = [rel(q, q + 2)[0] for q in range(6)]
    tt = [rel(q, q + 2)[1] / base[q] for q in range(6)]
    ok = np.ones(6, np.int32)
    ok[3] = 0
    r = o.seq_chain(pR, pt, np.ones(7, np.int32), tR, tt, ok)
    for q in range(6):
        want = base[q + 1] / base[q] if ok[q] else 1.0
        assert abs(r["track_scale"][q] - want) < 1e-12
    # with the failed track the scale of pair 4 is assumed equal to pair 3's: in this constant-speed sequence that is true
    for k in range(8):
        R, t = rel(0, k)
        assert np.abs(r["R"][k] - R).max() < 1e-12 and np.abs(r["t"][k] * base[0] - t).max() < 1e-12
    assert abs(r["pair_scale"][5] - base[5] / base[0]) < 1e-12


@pytest.mark.gpu
def test_gpu_sequence_trajectory_matches_oracle_and_truth(ctx):
    from mvslam_amd import capi

    F, N = 12, 800
    seq = synth.make_sequence(F, n_kp=N, n_map=9000, noise_px=0.1, step=0.12)
    s = capi.Sequence(ctx, F, N, 32)
    s.upload(0, seq["desc"], seq["kp"], seq["n_kp"], seq["K"])
    s.run(capi.default_params(num_hypotheses=4096, sampler=capi.SAMPLER_PHILOX, seed=1, max_error_sq=2e-3),
          capi.default_pnp_params(num_hypotheses=512, seed=2, reproj_error=1.0))
    gp, gt, tr = s.download_pairs(), s.download_tracks(), s.download_trajectory()
    s.close()
    res, trk = gp["results"], gt["tracks"]
    want = o.seq_chain(res["R"], res["t"], res["valid"], trk["R"], trk["t"], trk["ok"])
    for k in ("R", "t", "pair_scale", "track_scale"):
        assert tr[k].tobytes() == want[k].tobytes(), k               # same fold, same order of operations: bit-exact
    assert np.all(trk["ok"] == 1)
    # ground truth: camera centres in frame 0, in units of the first baseline
    cam = [(R.T, -R.T @ t) for R, t in seq["poses"]]
    b0 = np.linalg.norm(cam[1][1] - cam[0][1])
    for k in range(F):
        c_true = cam[0][0].T @ (cam[k][1] - cam[0][1]) / b0
        R_true = cam[0][0].T @ cam[k][0]
        assert np.abs(tr["R"][k] - R_true).max() < 0.02
        assert np.linalg.norm(tr["t"][k] - c_true) < 0.05 * max(1.0, np.linalg.norm(c_true)), (k, tr["t"][k], c_true)
    # constant speed: every pair has the same baseline; each step's ratio is good to a few per cent, and the product
    # drifts like a random walk (monocular scale drift -- the reference has it too, visual-odometer.cpp:577-588)
    assert np.all(np.abs(tr["track_scale"] - 1.0) < 0.1) and np.all(np.abs(tr["pair_scale"] - 1.0) < 0.25)


@pytest.mark.gpu
def test_gpu_sequence_trajectory_with_failed_tracks(ctx):
    """a frame with almost no keypoints invalidates its two pairs and the tracks through it: the fold falls back to the
    two-view poses / the identity exactly as the oracle's fold does"""
    from mvslam_amd import capi

    F, N = 9, 500
    seq = synth.make_sequence(F, n_kp=N, n_map=6000, noise_px=0.2, step=0.1)
    n_kp = seq["n_kp"].copy()
    n_kp[4] = 5                                                  # frame 4 is (almost) blind
    s = capi.Sequence(ctx, F, N, 32)
    s.upload(0, seq["desc"], seq["kp"], n_kp, seq["K"])
    s.run(capi.default_params(num_hypotheses=1024, sampler=capi.SAMPLER_PHILOX, seed=3, max_error_sq=2e-3),
          capi.default_pnp_params(num_hypotheses=256, seed=4, reproj_error=1.0))
    gp, gt, tr = s.download_pairs(), s.download_tracks(), s.download_trajectory()
    s.close()
    res, trk = gp["results"], gt["tracks"]
    assert not res["valid"][3] and not res["valid"][4] and res["valid"][0]
    assert not trk["ok"][2] and not trk["ok"][3] and not trk["ok"][4] and trk["ok"][0]
    want = o.seq_chain(res["R"], res["t"], res["valid"], trk["R"], trk["t"], trk["ok"])
    for k in ("R", "t", "pair_scale", "track_scale"):
        assert tr[k].tobytes() == want[k].tobytes(), k
    assert np.all(tr["track_scale"][2:5] == 1.0)
    assert np.array_equal(tr["R"][5], tr["R"][4]) and np.array_equal(tr["t"][5], tr["t"][4])   # pair 4 invalid: identity step


@pytest.mark.gpu
def test_gpu_sequence_refit_is_the_single_shot_refit_and_tightens_the_scale(ctx):
    """mvs_pnp_params.refit = 1 inside mvs_seq_run: the refit cv::solvePnPRansac ends with (pnp-solve.cpp:53-64), batched
    over all tracks on the device.  Every track's pose equals mvs_pnp_solve(refit = 1) on the joined correspondences
    (same kernel, same inputs in the same order), the inlier sets are the RANSAC ones, the fold is the oracle's, and the
    per-step scale ratios of a constant-speed sequence get closer to 1 than with the 3-point poses (median error)."""
    from mvslam_amd import capi

    F, N = 10, 800
    seq = synth.make_sequence(F, n_kp=N, n_map=9000, noise_px=0.3, step=0.08)
    prm = capi.default_params(num_hypotheses=4096, sampler=capi.SAMPLER_PHILOX, seed=1, max_error_sq=2e-3)
    out = {}
    for refit in (0, 1):
        s = capi.Sequence(ctx, F, N, 32)
        s.upload(0, seq["desc"], seq["kp"], seq["n_kp"], seq["K"])
        s.run(prm, capi.default_pnp_params(num_hypotheses=256, seed=2, reproj_error=1.5, refit=refit))
        out[refit] = (s.download_pairs(), s.download_tracks(), s.download_trajectory())
        s.close()
    (gp0, gt0, tr0), (gp1, gt1, tr1) = out[0], out[1]
    assert gp0["results"].tobytes() == gp1["results"].tobytes()              # the pairs do not depend on the refit
    t0, t1 = gt0["tracks"], gt1["tracks"]
    assert np.all(t1["ok"] == 1) and np.array_equal(t0["n_inliers"], t1["n_inliers"]) and np.array_equal(t0["best_hyp"], t1["best_hyp"])
    assert np.array_equal(gt0["inlier_idx"], gt1["inlier_idx"])
    for q in range(F - 2):
        nc = int(t1[q]["n_corr"])
        one = ctx.pnp_solve(gt1["corr_xyz"][q][:nc], gt1["corr_uv"][q][:nc], seq["K"],
                            capi.default_pnp_params(num_hypotheses=256, seed=2 + q, reproj_error=1.5, refit=1))
        assert one["ok"] and one["best_hyp"] == t1[q]["best_hyp"]
        assert one["R"].tobytes() == t1[q]["R"].tobytes() and one["t"].tobytes() == t1[q]["t"].tobytes()
        assert np.abs(t1[q]["R"] @ t1[q]["R"].T - np.eye(3)).max() < 1e-12
    want = o.seq_chain(gp1["results"]["R"], gp1["results"]["t"], gp1["results"]["valid"], t1["R"], t1["t"], t1["ok"])
    for k in ("R", "t", "pair_scale", "track_scale"):
        assert tr1[k].tobytes() == want[k].tobytes(), k
    e0, e1 = np.abs(tr0["track_scale"] - 1.0), np.abs(tr1["track_scale"] - 1.0)
    assert np.median(e1) < np.median(e0), (e0, e1)    # what remains is the two-view triangulation noise of the map points
