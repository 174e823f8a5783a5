"""BASELINE.json configs at FULL size inside the GPU suite.

configs[2]: 512 independent pairs x 2000 keypoints x 50 000 hypotheses in one resident batch -- size-independent
properties over every pair AND oracle parity on every pair (round 4, VERDICT r3 #1a: the pre-screen decides over the whole
population, so the oracle is asked about the whole population; ~0.5 s per pair and thread = ~16 s on the box's 16 threads).
configs[4]: a 1000-frame sequence x 2000 keypoints, 50 000 two-view + 100 PnP hypotheses per frame -- properties over
every pair / track, the trajectory fold against the oracle's fold, and full oracle parity on all 999 pairs / 998 tracks.
The 8-GPU config (configs[3]) is these 512 pairs per rank: the rank-local part is what runs here; the gather is covered
by tests/test_dist_gloo.py and tests/test_gather.py.
"""
import threading

import numpy as np
import pytest

import oracle_lib as o
from mvslam_amd import synth

pytestmark = pytest.mark.gpu


def _threads(fn, items, n=16):
    out, err = [None] * len(items), []

    def work(k0):
        try:
            for k in range(k0, len(items), n):
                out[k] = fn(items[k])
        except Exception as e:   # surface oracle-side failures in the main thread
            err.append(e)

    ths = [threading.Thread(target=work, args=(k,)) for k in range(min(n, len(items)))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    if err:
        raise err[0]
    return out


_BATCH512 = {}


def _batch512():
    if "d" not in _BATCH512:
        _BATCH512["d"] = synth.make_batch(0, 512, n_kp=2000)
    return _BATCH512["d"]



def _check_eval_counters(st):
    """score_evals_executed is the sum of every executed-evaluation counter (ABI 3: five matrix-core / binary32 parts + the
    double-precision remainder, which is not exported on its own and must come out non-negative) -- ADVICE r4."""
    parts = (st["score_evals_executed_f32"] + st["score_evals_executed_mfma"] + st["score_evals_executed_mfma_finish"] +
             st["score_evals_executed_mfma_rest"] + st["score_evals_executed_mfma_pilot"])
    f64 = st["score_evals_executed"] - parts
    assert f64 >= 0, st
    if st["pairs_mode"][0] == 0 and st["pairs_mode"][2] == 0:
        assert f64 == 0, st          # every pair counted on the matrix cores: nothing left for the double-precision kernel
    if st["pairs_mode"][1] == 0:
        assert parts == 0 and f64 > 0, st
    return f64

def test_config3_batch512(ctx):
    from mvslam_amd import capi

    P, N, H, THR = 512, 2000, 50000, 1e-2
    prm = capi.default_params(num_hypotheses=H, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=THR)
    data = _batch512()
    b = capi.Batch(ctx, P, N, 32)
    b.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"],
             data["global_index"])
    b.run(prm)
    b.sync()
    out = b.download()
    b.run(prm)                      # determinism of the whole batch (the pruned scoring drops hypotheses in a
    b.sync()                        # timing-dependent order; the winners must not depend on it)
    out2 = b.download(matches=False, mask=True, points=True)
    b.close()
    res = out["results"]
    assert res.tobytes() == out2["results"].tobytes()
    assert res["valid"].all() and (res["n_matches"] > 1200).all() and (res["n_points"] > 600).all()
    Kinv = np.linalg.inv(synth.K_DEFAULT)
    for i in range(P):
        r = res[i]
        M, n = int(r["n_matches"]), int(r["n_points"])
        assert np.array_equal(out["mask"][i][:M], out2["mask"][i][:M])
        assert out["points"][i][:n].tobytes() == out2["points"][i][:n].tobytes()
        mt = out["matches"][i][:M]
        key = mt["distance"].astype(np.int64) * 65536 + mt["queryIdx"]
        assert (np.diff(key) > 0).all()                                          # sorted by (distance, queryIdx)
        assert out["mask"][i][:M].sum() == r["n_inliers"] == r["best_count"]
        idx = out["point_idx"][i][:n]
        assert (np.diff(idx) > 0).all() and out["mask"][i][idx].all()            # ordered subset of the inliers
        pts = out["points"][i][:n]
        assert (pts[:, 2] > 0).all() and ((r["R1to2"] @ pts.T).T[:, 2] + r["t1to2"][2] > 0).all()   # cheirality
        x1 = (Kinv @ np.c_[data["kp1"][i][mt["trainIdx"]].astype(float), np.ones(M)].T).T
        x2 = (Kinv @ np.c_[data["kp2"][i][mt["queryIdx"]].astype(float), np.ones(M)].T).T
        e = np.abs(np.einsum("ij,jk,ik->i", x2, r["F"], x1))
        inl = out["mask"][i][:M].astype(bool)
        assert (e[inl] < THR * (1 + 1e-9)).all() and (e[~inl] > THR * (1 - 1e-9)).all()
        assert abs(np.linalg.norm(r["t1to2"]) - 1.0) < 1e-9 and np.abs(r["R"] @ r["R"].T - np.eye(3)).max() < 1e-9
    # shard invariance: a pair gives the same record alone (fused small-launch path) as inside the batch
    b1 = capi.Batch(ctx, 1, N, 32)
    for i in (0, 257, 511):
        sl = slice(i, i + 1)
        b1.upload(0, data["desc1"][sl], data["kp1"][sl], data["n1"][sl], data["desc2"][sl], data["kp2"][sl],
                  data["n2"][sl], data["K"][sl], data["global_index"][sl])
        b1.run(prm)
        b1.sync()
        assert b1.download(matches=False, mask=False, points=False)["results"][0].tobytes() == res[i].tobytes()
    b1.close()
    # oracle parity on ALL 512 pairs (estimator-RANSAC.cpp:76-84: the winner the reference's rule picks among 50 000)
    sample = list(range(P))

    def oracle(i):
        return o.image_pair(data["desc1"][i], data["kp1"][i], data["desc2"][i], data["kp2"][i],
                            data["K"][i].reshape(3, 3),
                            o.make_params(H, o.SAMPLER_PHILOX, synth.SEED_BASE + int(data["global_index"][i]), THR), 0.7, 10.0)

    for i, ref in zip(sample, _threads(oracle, sample)):
        r = res[i]
        M, n = ref["n_matches"], ref["n_points"]
        assert r["n_matches"] == M and out["matches"][i][:M].tobytes() == ref["matches"].tobytes()
        assert ref["ok"] and r["best_hyp"] == ref["best_hyp"] and r["best_count"] == ref["best_count"]
        assert r["best_residual"] == ref["best_residual"]
        assert np.array_equal(out["mask"][i][:M], ref["mask"])                    # inlier set: bit-exact
        assert r["n_points"] == n and np.array_equal(out["point_idx"][i][:n], ref["point_idx"])
        assert np.abs(out["points"][i][:n] - ref["points"]).max() <= 1e-12 * max(1.0, np.abs(ref["points"]).max())
        assert np.abs(r["R"] - ref["R"]).max() <= 1e-12 and np.abs(r["t"] - ref["t"]).max() <= 1e-12


def test_low_inlier_pairs_through_the_multi_chunk_counting(ctx):
    """Pairs with many matches and few inliers: the dense matrix-core phase must look at more points than one LDS chunk holds
    (n1 = M - B0 + 32 > 768, several staging passes with partial counts carried in hyp_cnt), the finish walks long lists.
    Plus ragged small pairs in the same batch (M below one tile, below eight).  Oracle parity of winner, count, residual sum,
    mask on every pair."""
    from mvslam_amd import capi

    specs = [(2000, 0.6), (2000, 0.75), (1800, 0.5), (2000, 0.3), (40, 0.3), (24, 0.0), (9, 0.0), (300, 0.6)]
    P, N, H, THR = len(specs), 2000, 6144, 1e-2
    pairs = [synth.make_pair(7000 + i, n_kp=n, outlier_frac=f) for i, (n, f) in enumerate(specs)]
    pad = lambda a: np.concatenate([a, np.zeros((N - len(a),) + a.shape[1:], a.dtype)])
    b = capi.Batch(ctx, P, N, 32)
    sizes = np.array([n for n, _ in specs], dtype=np.int32)
    gidx = np.arange(7000, 7000 + P, dtype=np.int64)
    b.upload(0, np.stack([pad(p["desc1"]) for p in pairs]), np.stack([pad(p["kp1"]) for p in pairs]), sizes,
             np.stack([pad(p["desc2"]) for p in pairs]), np.stack([pad(p["kp2"]) for p in pairs]), sizes,
             np.stack([p["K"].reshape(9) for p in pairs]), gidx)
    prm = capi.default_params(num_hypotheses=H, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=THR)
    b.run(prm)
    b.sync()
    out = b.download()
    st = b.stats(prm)
    b.close()
    assert st["pairs_mode"][1] >= 4 and st["score_evals_executed_mfma"] > 0 and st["score_evals_executed_mfma_finish"] > 0
    _check_eval_counters(st)

    def ref(i):
        p = pairs[i]
        return o.image_pair(p["desc1"], p["kp1"], p["desc2"], p["kp2"], p["K"],
                            o.make_params(H, o.SAMPLER_PHILOX, synth.SEED_BASE + int(gidx[i]), THR), 0.7, 10.0)

    refs = _threads(ref, list(range(P)))
    multi_chunk = 0
    for i, r in enumerate(refs):
        g = out["results"][i]
        M = r["n_matches"]
        assert g["n_matches"] == M and bool(g["valid"]) == bool(r["ok"]), i
        if not r["ok"]:
            continue
        assert g["best_hyp"] == r["best_hyp"] and g["best_count"] == r["best_count"], (i, int(g["best_hyp"]), r["best_hyp"])
        assert g["best_residual"] == r["best_residual"], i
        assert np.array_equal(out["mask"][i][:M], r["mask"]), i
        multi_chunk += int(M - r["best_count"] + 32 > 768)
    assert multi_chunk >= 1      # at least one pair needed more than one chunk of points in the dense phase


def test_config3_batch512_reference_threshold(ctx):
    """configs[2] at SURVEY 8(d)'s own threshold: max_error_sq = 0 selects the reference formula 5e-2 / K00 / K11
    (sfm-solve.cpp:18-19,311).  Best counts are ~5-10, hundreds of hypotheses tie at the maximum and the winner is decided
    by the residual sum (estimator-RANSAC.cpp:76-84): the regime that exercises the tie paths of the selection at full
    size.  Determinism over two runs of the batch + oracle parity on all 512 pairs."""
    from mvslam_amd import capi

    P, N, H = 512, 2000, 50000
    prm = capi.default_params(num_hypotheses=H, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=0.0)
    data = _batch512()
    b = capi.Batch(ctx, P, N, 32)
    b.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"],
             data["global_index"])
    b.run(prm)
    b.sync()
    out = b.download()
    b.run(prm)
    b.sync()
    out2 = b.download()
    b.close()
    res = out["results"]
    assert res.tobytes() == out2["results"].tobytes()
    assert out["mask"].tobytes() == out2["mask"].tobytes() and out["points"].tobytes() == out2["points"].tobytes()
    assert out["point_idx"].tobytes() == out2["point_idx"].tobytes()
    thr = 5e-2 / 525.0 / 525.0
    assert (res["best_hyp"] >= 0).all() and (res["best_count"] >= 1).all() and (res["best_count"] < 200).all()
    Kinv = np.linalg.inv(synth.K_DEFAULT)
    for i in range(P):
        r = res[i]
        M = int(r["n_matches"])
        assert out["mask"][i][:M].sum() == r["n_inliers"] == r["best_count"]
        mt = out["matches"][i][:M]
        x1 = (Kinv @ np.c_[data["kp1"][i][mt["trainIdx"]].astype(float), np.ones(M)].T).T
        x2 = (Kinv @ np.c_[data["kp2"][i][mt["queryIdx"]].astype(float), np.ones(M)].T).T
        e = np.abs(np.einsum("ij,jk,ik->i", x2, r["F"], x1))
        inl = out["mask"][i][:M].astype(bool)
        assert (e[inl] < thr * (1 + 1e-6)).all() and (e[~inl] > thr * (1 - 1e-6)).all()
        assert bool(r["valid"]) == (r["n_points"] > 0)
    sample = list(range(P))

    def oracle(i):
        return o.image_pair(data["desc1"][i], data["kp1"][i], data["desc2"][i], data["kp2"][i],
                            data["K"][i].reshape(3, 3),
                            o.make_params(H, o.SAMPLER_PHILOX, synth.SEED_BASE + int(data["global_index"][i]), 0.0), 0.7, 10.0)

    n_valid = 0
    for i, ref in zip(sample, _threads(oracle, sample)):
        r = res[i]
        M = ref["n_matches"]
        assert r["n_matches"] == M and out["matches"][i][:M].tobytes() == ref["matches"].tobytes()
        assert bool(r["valid"]) == bool(ref["ok"])
        assert r["best_hyp"] == ref["best_hyp"] and r["best_count"] == ref["best_count"]
        assert r["best_residual"] == ref["best_residual"]                         # the tie-breaker, bit for bit
        assert np.array_equal(out["mask"][i][:M], ref["mask"])
        if ref["ok"]:
            n_valid += 1
            n = ref["n_points"]
            assert r["n_points"] == n and np.array_equal(out["point_idx"][i][:n], ref["point_idx"])
            assert np.abs(r["R"] - ref["R"]).max() <= 1e-12 and np.abs(r["t"] - ref["t"]).max() <= 1e-12
    print("config3 at the reference threshold: best_count min/median/max = %d / %d / %d, valid %d of 512 (oracle: %d of 512)"
          % (res["best_count"].min(), np.median(res["best_count"]), res["best_count"].max(), int(res["valid"].sum()), n_valid))


def test_config5_sequence_window_reference_threshold(ctx):
    """One window of configs[4] at the reference threshold (max_error_sq = 0 -> 5e-2 / K00 / K11): the first 64 frames of
    the 1000-frame sequence at full size per frame (2000 keypoints, 50 000 hypotheses), determinism over two runs and
    oracle parity of every pair of one 4-frame window."""
    from mvslam_amd import capi
    from test_sequence import oracle_sequence

    F, N, H, HP = 64, 2000, 50000, 100
    seq = synth.make_sequence(1000, n_kp=N)
    seq = dict(desc=seq["desc"][:F], kp=seq["kp"][:F], n_kp=seq["n_kp"][:F], K=seq["K"])
    s = capi.Sequence(ctx, F, N, 32)
    s.upload(0, seq["desc"], seq["kp"], seq["n_kp"], seq["K"])
    p2 = capi.default_params(num_hypotheses=H, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=0.0)
    pp = capi.default_pnp_params(num_hypotheses=HP, seed=7, reproj_error=2.0)
    s.run(p2, pp)
    gp = s.download_pairs()
    s.run(p2, pp)
    gp2 = s.download_pairs()
    s.close()
    res = gp["results"]
    assert res.tobytes() == gp2["results"].tobytes() and gp["mask"].tobytes() == gp2["mask"].tobytes()
    k0 = 30
    sub = dict(desc=seq["desc"][k0:k0 + 4], kp=seq["kp"][k0:k0 + 4], n_kp=seq["n_kp"][k0:k0 + 4], K=seq["K"])
    pairs, _ = oracle_sequence(sub, dict(H=H, seed=synth.SEED_BASE + k0, thr=0.0), dict(H=HP, seed=7 + k0, err=2.0))
    for j, ref in enumerate(pairs):
        k = k0 + j
        r, M = res[k], ref["n_matches"]
        assert r["n_matches"] == M and gp["matches"][k][:M].tobytes() == ref["matches"].tobytes()
        assert bool(r["valid"]) == ref["ok"] and np.array_equal(gp["mask"][k][:M], ref["mask"])
        assert r["best_hyp"] == ref["best_hyp"] and r["best_count"] == ref["best_count"]
        assert r["best_residual"] == ref["best_residual"]


def test_config5_sequence1000(ctx):
    from mvslam_amd import capi
    from test_sequence import oracle_sequence

    F, N, H, HP = 1000, 2000, 50000, 100
    prm = dict(H=H, seed=synth.SEED_BASE, thr=1e-2)
    pprm = dict(H=HP, seed=7, err=2.0)
    seq = synth.make_sequence(F, n_kp=N)
    s = capi.Sequence(ctx, F, N, 32)
    s.upload(0, seq["desc"], seq["kp"], seq["n_kp"], seq["K"])
    s.run(capi.default_params(num_hypotheses=H, sampler=capi.SAMPLER_PHILOX, seed=prm["seed"], max_error_sq=prm["thr"]),
          capi.default_pnp_params(num_hypotheses=HP, seed=pprm["seed"], reproj_error=pprm["err"]))
    gp, gt, tr = s.download_pairs(), s.download_tracks(), s.download_trajectory()
    s.close()
    res, trk = gp["results"], gt["tracks"]
    assert res["valid"].sum() >= F - 3 and trk["ok"].sum() >= F - 5
    assert (res["n_matches"][res["valid"] == 1] > 300).all()
    for k in range(F - 1):
        r = res[k]
        M, n = int(r["n_matches"]), int(r["n_points"])
        assert gp["mask"][k][:M].sum() == r["n_inliers"]
        idx = gp["point_idx"][k][:n]
        assert (np.diff(idx) > 0).all() and gp["mask"][k][idx].all()
        if r["valid"]:
            assert (gp["points"][k][:n][:, 2] > 0).all() and abs(np.linalg.norm(r["t1to2"]) - 1.0) < 1e-9
    for q in range(F - 2):
        t = trk[q]
        nc, ni = int(t["n_corr"]), int(t["n_inliers"])
        assert 0 <= ni <= nc <= N
        if t["ok"]:
            ii = gt["inlier_idx"][q][:ni]
            assert (np.diff(ii) > 0).all() and (ii < nc).all()
            assert np.abs(t["R"] @ t["R"].T - np.eye(3)).max() < 1e-9
    # the scale-propagation fold (visual-odometer.cpp:577-588) over all 1000 frames: the oracle's fold, same bits
    want = o.seq_chain(res["R"], res["t"], res["valid"], trk["R"], trk["t"], trk["ok"])
    for k in ("R", "t", "pair_scale", "track_scale"):
        assert tr[k].tobytes() == want[k].tobytes(), k
    # ground truth: the heading follows the synthetic yaw (0.005 rad per frame).  The translation SCALE is not asserted:
    # at this config's 0.05 m baseline and 0.5 px noise the per-step scale ratio of a 3-point PnP pose is only good to
    # tens of per cent (monocular scale drift; the reference's VO has it too and relies on BA, which is out of scope)
    cam = [(R.T, -R.T @ t) for R, t in seq["poses"]]
    rot_err = [float(np.abs(tr["R"][k] - cam[0][0].T @ cam[k][0]).max()) for k in (10, 100, 500, 999)]
    print("config5 heading error at frames 10/100/500/999:", rot_err, "median |track_scale - 1|:",
          float(np.median(np.abs(tr["track_scale"] - 1.0))))
    assert rot_err[0] < 0.05 and rot_err[1] < 0.1
    for k in range(F):
        assert np.abs(tr["R"][k] @ tr["R"][k].T - np.eye(3)).max() < 1e-6
    # full oracle parity on ALL 999 pairs and 998 tracks (round 4, VERDICT r3 #1a); pair k runs with seed + k, track q with
    # seed + q, exactly as oracle_sequence does for a window starting at 0
    pairs, tracks = oracle_sequence(seq, prm, pprm, threads=16)
    assert len(pairs) == F - 1 and len(tracks) == F - 2
    for k, ref in enumerate(pairs):
        r, M = res[k], ref["n_matches"]
        assert r["n_matches"] == M and gp["matches"][k][:M].tobytes() == ref["matches"].tobytes(), k
        assert bool(r["valid"]) == ref["ok"] and np.array_equal(gp["mask"][k][:M], ref["mask"]), k
        assert r["best_hyp"] == ref["best_hyp"] and r["best_count"] == ref["best_count"], k
        assert r["best_residual"] == ref["best_residual"], k
        if ref["ok"]:
            n = ref["n_points"]
            assert np.array_equal(gp["point_idx"][k][:n], ref["point_idx"]), k
            assert gp["points"][k][:n].tobytes() == ref["points"].tobytes(), k
    for q, ref in enumerate(tracks):
        t, nc = trk[q], len(ref["X"])
        assert t["n_corr"] == nc and gt["corr_xyz"][q][:nc].tobytes() == ref["X"].tobytes(), q
        assert gt["corr_uv"][q][:nc].tobytes() == ref["uv"].tobytes(), q
        assert bool(t["ok"]) == ref["ok"] and t["best_hyp"] == ref["best_hyp"], q
        if ref["ok"]:
            ni = len(ref["inliers"])
            assert t["n_inliers"] == ni and np.array_equal(gt["inlier_idx"][q][:ni], ref["inliers"]), q
            assert t["R"].tobytes() == ref["R"].tobytes() and t["t"].tobytes() == ref["t"].tobytes(), q


def test_high_outlier_pairs_at_full_size(ctx):
    """where the pre-screen prunes least -- 90 % wrong matches, M ~ 2000 matches, 50 000 hypotheses, no
    all-inlier sample among them (best counts ~70 at 1e-2, ~16 at 1e-3; at 2e-4 the matrix cores' error term exceeds the
    threshold and the probe sends every pair to double-precision counting).  Oracle parity of winner, count, residual
    sum, mask, points on every pair at the three thresholds (estimator-RANSAC.cpp:76-84)."""
    from mvslam_amd import capi

    P, N, H = 8, 2500, 50000
    data = synth.make_batch(7700, P, n_kp=N, outlier_frac=0.9)
    b = capi.Batch(ctx, P, N, 32)
    b.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"],
             data["global_index"])
    for thr in (1e-2, 1e-3, 2e-4):
        prm = capi.default_params(num_hypotheses=H, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=thr)
        b.run(prm)
        b.sync()
        out = b.download()
        st = b.stats(prm)
        assert (out["results"]["n_matches"] > 1900).all()
        _check_eval_counters(st)
        assert st["pairs_mode"][0] == 0 and st["pairs_mode"][1] + st["pairs_mode"][2] == P      # pre-screened, one way or the other
        if thr == 2e-4:
            assert st["pairs_mode"][2] == P        # the matrix cores' 2^-14 T term is beyond this threshold: double-precision counting

        def oracle(i):
            return o.image_pair(data["desc1"][i], data["kp1"][i], data["desc2"][i], data["kp2"][i], data["K"][i].reshape(3, 3),
                                o.make_params(H, o.SAMPLER_PHILOX, synth.SEED_BASE + int(data["global_index"][i]), thr), 0.7, 10.0)

        for i, ref in enumerate(_threads(oracle, list(range(P)))):
            r = out["results"][i]
            M, n = ref["n_matches"], ref["n_points"]
            assert r["n_matches"] == M and bool(r["valid"]) == bool(ref["ok"]), (thr, i)
            assert r["best_hyp"] == ref["best_hyp"] and r["best_count"] == ref["best_count"], (thr, i)
            assert r["best_residual"] == ref["best_residual"], (thr, i)
            assert np.array_equal(out["mask"][i][:M], ref["mask"]), (thr, i)
            if ref["ok"]:
                assert r["n_points"] == n and np.array_equal(out["point_idx"][i][:n], ref["point_idx"]), (thr, i)
                assert np.abs(r["R"] - ref["R"]).max() <= 1e-12 and np.abs(r["t"] - ref["t"]).max() <= 1e-12, (thr, i)
    b.close()
