"""pnp_solve (vision/pnp-solve.cpp:16-104): oracle KATs on CPU, bit-exact GPU parity."""
import numpy as np
import pytest

import helpers
import oracle_lib as o


def _cube_rig():
    """test/test-pnp.cpp:14-37: K = I, camera pose exp(1,0,0,0,0,0), the cube at (0.6, 0, 3)."""
    K = np.eye(3)
    Rc, tc = o.se3_exp(np.array([1, 0, 0, 0, 0, 0.0]))        # camera in world
    Rw, tw = o.se3_inverse(Rc, tc)                            # world -> camera
    X = helpers.rig_points("cube", (0.0, 0.0, 0.0), (0.6, 0.0, 3.0), 1.0)
    return K, X, o.project_points(K, Rw, tw, X)


def _scene(seed, n, noise_px, n_out):
    rng = np.random.default_rng(seed)
    K = np.array([[525.0, 0, 320], [0, 525, 240], [0, 0, 1]])
    R = o.rodrigues(rng.normal(size=3) * 0.08)
    t = np.array([0.2, -0.1, 0.3]) + rng.normal(size=3) * 0.05
    X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(4, 9, n)], axis=1)
    uv = o.project_points(K, R, t, X) + rng.normal(scale=noise_px, size=(n, 2))
    bad = rng.choice(n, size=n_out, replace=False)
    uv[bad] = np.stack([rng.uniform(0, 640, n_out), rng.uniform(0, 480, n_out)], axis=1)
    return K, X, uv, R, t, np.sort(bad)


# ------------------------------------------------------------------------------------------- CPU (oracle)
def test_oracle_pnp_solve_cube():
    """test/test-pnp.cpp:14-60: all 8 points are inliers, pose.ln() == (1,0,0,0,0,0) within 1e-3."""
    K, X, uv = _cube_rig()
    for prm in (o.make_pnp_params(100, o.SAMPLER_PHILOX, 0), o.make_pnp_params(1, o.SAMPLER_IDENTITY, 0)):
        r = o.pnp_solve(X, uv, K, prm)
        assert r["ok"] and r["inliers"].tolist() == list(range(8))
        assert np.abs(o.se3_ln(r["R"], r["t"]) - [1, 0, 0, 0, 0, 0]).max() < 1e-3


def test_oracle_p3p_against_ground_truth():
    rng = np.random.default_rng(2)
    worst = []
    for _ in range(300):
        R = o.rodrigues(rng.normal(size=3) * 0.5)
        t = rng.normal(size=3) + [0, 0, 5.0]
        X = rng.uniform(-1, 1, (3, 3))
        Pc = (R @ X.T).T + t
        f = Pc / np.linalg.norm(Pc, axis=1, keepdims=True)
        Rs, ts = o.p3p(f, X)
        assert 1 <= len(Rs) <= 4
        worst.append(min(np.abs(Rk - R).max() + np.abs(tk - t).max() for Rk, tk in zip(Rs, ts)))
        for Rk, tk in zip(Rs, ts):                                    # every returned solution is a rigid motion that
            assert np.abs(Rk @ Rk.T - np.eye(3)).max() < 1e-9          # reproduces the three bearings
            Q = (Rk @ X.T).T + tk
            assert np.abs(Q / np.linalg.norm(Q, axis=1, keepdims=True) - f).max() < 1e-6
    assert np.median(worst) < 1e-10 and max(worst) < 1e-3


def test_oracle_pnp_with_outliers_and_noise():
    K, X, uv, R, t, bad = _scene(5, 300, 0.01, 60)
    r = o.pnp_solve(X, uv, K, o.make_pnp_params(500, o.SAMPLER_PHILOX, 7))
    assert r["ok"] and len(r["inliers"]) >= 230 and not set(r["inliers"].tolist()) & set(bad.tolist())
    assert np.abs(r["Rw2c"] - R).max() < 1e-3 and np.abs(r["tw2c"] - t).max() < 5e-3
    Rc, tc = o.se3_inverse(o.so3_rectify(r["Rw2c"]), r["tw2c"])       # pose convention (pnp-solve.cpp:101)
    assert np.array_equal(Rc, r["R"]) and np.array_equal(tc, r["t"])


def test_oracle_pnp_preconditions_and_sampler():
    K, X, uv = _cube_rig()
    assert not o.pnp_solve(X[:6], uv[:6], K, o.make_pnp_params())["ok"]          # < 7 points (reference: assert)
    for n in (7, 8, 100, 2048):
        for h in range(50):
            idx = o.sample4(3, h, n)
            assert len(set(idx.tolist())) == 4 and idx.min() >= 0 and idx.max() < n
    assert o.sample4(1, 0, 9, o.SAMPLER_IDENTITY).tolist() == [0, 1, 2, 3]


def test_golden_pnp():
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pnp_small.npz"))
    rc = o.pnp_solve(g["cube_X"], g["cube_uv"], np.eye(3), o.make_pnp_params(100, o.SAMPLER_PHILOX, 0))
    assert rc["R"].tobytes() == g["cube_R"].tobytes() and rc["t"].tobytes() == g["cube_t"].tobytes()
    rs = o.pnp_solve(g["X"], g["uv"], g["K"], o.make_pnp_params(int(g["H"]), o.SAMPLER_PHILOX, int(g["seed"])))
    assert rs["best_hyp"] == int(g["best_hyp"]) and np.array_equal(rs["inliers"], g["inliers"])
    assert rs["R"].tobytes() == g["R"].tobytes() and rs["t"].tobytes() == g["t"].tobytes()
    for h in (0, 100, 255):
        assert o.sample4(int(g["seed"]), h, len(g["X"])).tolist() == g["samples"][h].tolist()


# ------------------------------------------------------------------------------------------- GPU parity
@pytest.mark.gpu
def test_gpu_pnp_against_golden(ctx):
    """Oracle-free: the committed expected outputs (tests/golden/pnp_small.npz)."""
    import os
    from mvslam_amd import capi

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pnp_small.npz"))
    got = ctx.pnp_solve(g["X"], g["uv"], g["K"], capi.default_pnp_params(num_hypotheses=int(g["H"]), seed=int(g["seed"])))
    assert got["ok"] and got["best_hyp"] == int(g["best_hyp"]) and np.array_equal(got["inliers"], g["inliers"])
    assert got["R"].tobytes() == g["R"].tobytes() and got["t"].tobytes() == g["t"].tobytes()


@pytest.mark.gpu
def test_gpu_pnp_solve_cube(ctx):
    from mvslam_amd import capi

    K, X, uv = _cube_rig()
    for kw in (dict(num_hypotheses=100, sampler=capi.SAMPLER_PHILOX), dict(num_hypotheses=1, sampler=capi.SAMPLER_IDENTITY)):
        got = ctx.pnp_solve(X, uv, K, capi.default_pnp_params(**kw))
        ref = o.pnp_solve(X, uv, K, o.make_pnp_params(kw["num_hypotheses"], kw["sampler"], 0))
        assert got["ok"] and got["inliers"].tolist() == list(range(8))
        assert np.abs(o.se3_ln(got["R"], got["t"]) - [1, 0, 0, 0, 0, 0]).max() < 1e-3
        assert got["R"].tobytes() == ref["R"].tobytes() and got["t"].tobytes() == ref["t"].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("n,H,noise,n_out", [(7, 64, 0.0, 0), (50, 100, 0.01, 10), (300, 1000, 0.01, 60),
                                               (2048, 4096, 0.02, 500), (400, 257, 0.5, 100),
                                               # round 5: the point stream goes through LDS in 768-point chunks and a block
                                               # with <= 128 hypotheses splits the points over its wavefronts (1, 2, 3, 4
                                               # hypothesis groups -> 4, 2, 1, 1 parts); 4096 = the keypoint capacity
                                               (769, 64, 0.01, 100), (1537, 100, 0.01, 300), (4096, 129, 0.02, 900),
                                               (3000, 200, 0.02, 700), (900, 40, 0.01, 0)])
def test_gpu_pnp_matches_oracle_bit_for_bit(ctx, n, H, noise, n_out):
    """inlier indices and the winning hypothesis bit-exact; pose bitwise (the path uses only + - * / sqrt)."""
    from mvslam_amd import capi

    K, X, uv, R, t, bad = _scene(n + H, n, noise, n_out)
    got = ctx.pnp_solve(X, uv, K, capi.default_pnp_params(num_hypotheses=H, sampler=capi.SAMPLER_PHILOX, seed=n * 1000 + 1))
    ref = o.pnp_solve(X, uv, K, o.make_pnp_params(H, o.SAMPLER_PHILOX, n * 1000 + 1))
    assert got["ok"] == ref["ok"] and got["best_hyp"] == ref["best_hyp"]
    assert np.array_equal(got["inliers"], ref["inliers"])
    if ref["ok"]:
        assert helpers.rel_err(got["R"], ref["R"]) <= 1e-4 and helpers.rel_err(got["t"], ref["t"]) <= 1e-4
        assert got["R"].tobytes() == ref["R"].tobytes() and got["t"].tobytes() == ref["t"].tobytes()


@pytest.mark.gpu
def test_gpu_pnp_errors(ctx):
    from mvslam_amd import capi

    K, X, uv = _cube_rig()
    with pytest.raises(capi.MvsError):
        ctx.pnp_solve(X[:6], uv[:6], K, capi.default_pnp_params())             # < 7 points
    Kx, Xx, uvx, *_ = _scene(1, 40, 0.0, 0)
    uvx[:] = np.random.default_rng(0).uniform(0, 480, uvx.shape)               # pure outliers: no model, no abort
    got = ctx.pnp_solve(Xx, uvx, Kx, capi.default_pnp_params(num_hypotheses=64, min_inliers=10))
    ref = o.pnp_solve(Xx, uvx, Kx, o.make_pnp_params(64, o.SAMPLER_PHILOX, 0, min_inliers=10))
    assert got["ok"] == ref["ok"] and not got["ok"]


@pytest.mark.gpu
def test_gpu_pnp_refit_is_the_reprojection_minimiser_over_the_inliers(ctx):
    """mvs_pnp_params.refit = 1: the refit cv::solvePnPRansac ends with (pnp-solve.cpp:53-64).  The inlier set is the
    RANSAC one; the pose is the minimiser of the reprojection error over it (checked against scipy on the same
    residuals) and is closer to the truth than the best 3-point hypothesis; the cube fixture stays exact."""
    from scipy.optimize import least_squares

    from mvslam_amd import capi

    K, X, uv = _cube_rig()
    r = ctx.pnp_solve(X, uv, K, capi.default_pnp_params(num_hypotheses=100, seed=0, refit=1))
    assert r["ok"] and r["inliers"].tolist() == list(range(8))
    assert np.abs(o.se3_ln(r["R"], r["t"]) - [1, 0, 0, 0, 0, 0]).max() < 1e-6          # test/test-pnp.cpp:14-60, 1e-3
    better = 0
    for seed in range(6):
        K, X, uv, R, t, bad = _scene(40 + seed, 400, 0.5, 80)
        Rc, tc = o.se3_inverse(R, t)                                                    # truth: camera in world
        base = ctx.pnp_solve(X, uv, K, capi.default_pnp_params(num_hypotheses=300, seed=seed, reproj_error=2.0))
        got = ctx.pnp_solve(X, uv, K, capi.default_pnp_params(num_hypotheses=300, seed=seed, reproj_error=2.0, refit=1))
        assert base["ok"] and got["ok"] and got["best_hyp"] == base["best_hyp"]
        assert np.array_equal(got["inliers"], base["inliers"])                          # the inlier set is not re-voted
        Xi, ui = X[got["inliers"]], uv[got["inliers"]]

        def res(x):
            Rw, tw = o.se3_inverse(o.rodrigues(x[:3]) @ got["R"], got["t"] + x[3:])
            return (o.project_points(K, Rw, tw, Xi) - ui).ravel()

        s = least_squares(res, np.zeros(6), method="lm", xtol=1e-14, ftol=1e-14)
        assert np.abs(s.x).max() < 1e-6, s.x                                            # already at the minimiser
        e_base = np.abs(base["t"] - tc).max() + np.abs(base["R"] - Rc).max()
        e_got = np.abs(got["t"] - tc).max() + np.abs(got["R"] - Rc).max()
        better += e_got < e_base
    assert better >= 5
