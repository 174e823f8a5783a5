"""Row f4 of SURVEY.md section 8: sfm_refine / pnp_refine (vision/sfm-refine.cpp, pnp-refine.cpp -> ba.cpp -> GTSAM).

CPU part: pins the ORACLE (oracle/mvs_refine_oracle.c) -- its minimiser against scipy.optimize.least_squares on the same
residual vector, its covariances against a finite-difference Hessian, and the reference's own known-answer test
(test/test-sfm.cpp:157-286 sfm_refine_L_shape: truth recovered within 0.025 from 5e-3 noise).
GPU part: the HIP kernel through the C ABI against the oracle (tolerance: trigonometric functions come from different
libraries on the two sides; everything else follows the same order of operations).
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation as Rot

import helpers as h
import oracle_lib as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def proj(K, R, t, X):
    """camera-in-world pose (R, t): pixel = K [R^T (X - t)]"""
    q = (X - t) @ R
    xn = q[:, :2] / q[:, 2:3]
    return np.stack([K[0, 0] * xn[:, 0] + K[0, 1] * xn[:, 1] + K[0, 2], K[1, 1] * xn[:, 1] + K[1, 2]], axis=1)


def two_view_problem(seed, m, K=None, sig=None, baseline=1.0, depth=(2.0, 4.0)):
    rng = np.random.default_rng(seed)
    K = np.eye(3) if K is None else K
    f = K[0, 0]
    sig = (5e-3 * f) if sig is None else sig
    X = np.stack([rng.uniform(-1, 1, m), rng.uniform(-1, 1, m), rng.uniform(*depth, m)], 1)
    R_true = Rot.from_rotvec(rng.normal(0, 0.03, 3)).as_matrix()
    t_true = np.array([baseline, 0.05 * baseline, -0.02 * baseline])
    p1 = proj(K, np.eye(3), np.zeros(3), X) + rng.normal(0, sig, (m, 2))
    p2 = proj(K, R_true, t_true, X) + rng.normal(0, sig, (m, 2))
    cov = np.tile((np.eye(2) * sig ** 2).reshape(4), (m, 1))
    Rg = R_true @ Rot.from_rotvec(rng.normal(0, 1e-2, 3)).as_matrix()
    tg = t_true + rng.normal(0, 5e-3, 3)
    Xg = X + rng.normal(0, 5e-3, X.shape)
    return dict(K=K, X=X, R_true=R_true, t_true=t_true, p1=p1, p2=p2, cov=cov, Rg=Rg, tg=tg, Xg=Xg, sig=sig)


def sfm_residuals(pb, prm_sig=(1e-5, 1e-2, 1e-2)):
    """the whitened residual vector ba.cpp's graph stands for, parametrised globally (for scipy)"""
    a_sig, p_sig, x_sig = prm_sig
    m = len(pb["Xg"])
    K, Rg, tg, Xg, sig = pb["K"], pb["Rg"], pb["tg"], pb["Xg"], pb["sig"]

    def resid(x):
        R0 = Rot.from_rotvec(x[0:3]).as_matrix()
        t0 = x[3:6]
        R1 = Rg @ Rot.from_rotvec(x[6:9]).as_matrix()
        t1 = x[9:12]
        P = x[12:].reshape(m, 3)
        return np.concatenate([
            Rot.from_matrix(R0).as_rotvec() / a_sig, t0 / a_sig,
            Rot.from_matrix(Rg.T @ R1).as_rotvec() / p_sig, Rg.T @ (t1 - tg) / p_sig,
            ((P - Xg) / x_sig).ravel(),
            ((proj(K, R0, t0, P) - pb["p1"]) / sig).ravel(), ((proj(K, R1, t1, P) - pb["p2"]) / sig).ravel()])

    x0 = np.concatenate([np.zeros(6), np.zeros(3), tg, Xg.ravel()])
    return resid, x0


def test_oracle_sfm_refine_is_the_minimiser():
    pb = two_view_problem(1, 12)
    res = o.sfm_refine(pb["p1"], pb["cov"], pb["p2"], pb["cov"], pb["K"], pb["Rg"], pb["tg"], pb["Xg"])
    assert res["ok"] and res["iterations"] < 20
    resid, x0 = sfm_residuals(pb)
    sol = least_squares(resid, x0, xtol=1e-15, ftol=1e-15, gtol=1e-15, method="trf", jac="3-point", x_scale="jac")
    R1 = pb["Rg"] @ Rot.from_rotvec(sol.x[6:9]).as_matrix()
    assert abs(0.5 * np.sum(sol.fun ** 2) - res["error"]) <= 1e-9 * res["error"]
    assert np.abs(R1 - res["R"]).max() < 1e-9
    assert np.abs(sol.x[9:12] - res["t"]).max() < 1e-9
    assert np.abs(sol.x[12:].reshape(-1, 3) - res["points"]).max() < 1e-9
    # cost actually decreased from the guess
    assert res["error"] < 0.5 * np.sum(resid(x0) ** 2)


def test_oracle_covariances_match_finite_difference_hessian():
    pb = two_view_problem(2, 10)
    m = 10
    res = o.sfm_refine(pb["p1"], pb["cov"], pb["p2"], pb["cov"], pb["K"], pb["Rg"], pb["tg"], pb["Xg"])
    K, Rg, tg, Xg, sig = pb["K"], pb["Rg"], pb["tg"], pb["Xg"], pb["sig"]

    def local(d):  # right perturbation (rotation, translation) at the estimate; camera 1 sits at its anchor
        R0 = Rot.from_rotvec(d[0:3]).as_matrix()
        t0 = d[3:6]
        R1 = res["R"] @ Rot.from_rotvec(d[6:9]).as_matrix()
        t1 = res["t"] + res["R"] @ d[9:12]
        P = res["points"] + d[12:].reshape(m, 3)
        return np.concatenate([
            Rot.from_matrix(R0).as_rotvec() / 1e-5, t0 / 1e-5, Rot.from_matrix(Rg.T @ R1).as_rotvec() / 1e-2,
            Rg.T @ (t1 - tg) / 1e-2, ((P - Xg) / 1e-2).ravel(), ((proj(K, R0, t0, P) - pb["p1"]) / sig).ravel(),
            ((proj(K, R1, t1, P) - pb["p2"]) / sig).ravel()])

    n, hh = 12 + 3 * m, 1e-6
    J = np.stack([(local(np.eye(n)[k] * hh) - local(-np.eye(n)[k] * hh)) / (2 * hh) for k in range(n)], 1)
    Cfull = np.linalg.inv(J.T @ J)
    assert np.abs(Cfull[6:12, 6:12] - res["pose_cov"]).max() <= 1e-6 * np.abs(Cfull[6:12, 6:12]).max()
    pc = np.stack([Cfull[12 + 3 * i:15 + 3 * i, 12 + 3 * i:15 + 3 * i] for i in range(m)])
    assert np.abs(pc - res["point_cov"]).max() <= 1e-6 * np.abs(pc).max()
    assert np.allclose(res["pose_cov"], res["pose_cov"].T, rtol=0, atol=1e-18)


def l_shape_refine_problem(seed):
    """test/test-sfm.cpp:157-249: K = I, camera 2 at x = 1, L-shaped rig, 5e-3 measurement noise with matching
    covariance, guess = truth perturbed by (5e-3 translation, 1e-2 rotation), points perturbed by 5e-3"""
    rng = np.random.default_rng(seed)
    rig = h.two_camera_rig("L", rpy=(0.0, 0.7, 1.5), translation=(0.6, 0.0, 3.0), scale=0.5)
    X = rig["X"]
    K = np.eye(3)
    sig = 5e-3
    p1 = rig["uv1"] + rng.normal(0, sig, rig["uv1"].shape)
    p2 = rig["uv2"] + rng.normal(0, sig, rig["uv2"].shape)
    cov = np.tile((np.eye(2) * sig ** 2).reshape(4), (len(X), 1))
    delta = np.concatenate([rng.normal(0, 5e-3, 3), rng.normal(0, 1e-2, 3)])
    Rd, td = o.se3_exp(delta)
    Rg, tg = o.se3_compose(Rd, td, np.eye(3), np.array([1.0, 0.0, 0.0]))   # exp(delta) * P1 * P2^-1
    Xg = X + rng.normal(0, 5e-3, X.shape)
    return dict(K=K, X=X, p1=p1, p2=p2, cov=cov, Rg=Rg, tg=tg, Xg=Xg)


def test_rig_fixture_is_the_reference_fixture():
    rig = h.two_camera_rig("L", rpy=(0.0, 0.7, 1.5), translation=(0.6, 0.0, 3.0), scale=0.5)
    X = rig["X"]
    assert np.allclose(rig["uv1"], X[:, :2] / X[:, 2:3])
    assert np.allclose(rig["uv2"], (X - [1, 0, 0])[:, :2] / X[:, 2:3])


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_oracle_reference_kat_sfm_refine_L_shape(seed):
    pb = l_shape_refine_problem(seed)
    res = o.sfm_refine(pb["p1"], pb["cov"], pb["p2"], pb["cov"], pb["K"], pb["Rg"], pb["tg"], pb["Xg"])
    assert res["ok"]
    se3 = o.se3_ln(res["R"], res["t"])
    assert np.abs(se3 - np.array([1.0, 0, 0, 0, 0, 0])).max() < 0.025      # test-sfm.cpp:159,276-278
    assert np.abs(res["points"] - pb["X"]).max() < 0.025                    # test-sfm.cpp:280-285


def pnp_problem(seed, m, K=None):
    rng = np.random.default_rng(seed)
    K = np.array([[525.0, 0, 320], [0, 525, 240], [0, 0, 1]]) if K is None else K
    X = np.stack([rng.uniform(-1, 1, m), rng.uniform(-1, 1, m), rng.uniform(3, 6, m)], 1)
    R_true = Rot.from_rotvec(rng.normal(0, 0.05, 3)).as_matrix()
    t_true = rng.normal(0, 0.1, 3)
    sig = 0.5
    uv = proj(K, R_true, t_true, X) + rng.normal(0, sig, (m, 2))
    icov = np.tile((np.eye(2) * sig ** 2).reshape(4), (m, 1))
    A = rng.normal(0, 1, (m, 3, 3))
    wcov = 1e-4 * (np.eye(3) + 0.2 * (A @ A.transpose(0, 2, 1)))          # full SPD covariances
    Xn = X + np.einsum("mij,mj->mi", np.linalg.cholesky(wcov), rng.normal(0, 1, (m, 3)))
    Rg = R_true @ Rot.from_rotvec(rng.normal(0, 5e-3, 3)).as_matrix()
    tg = t_true + rng.normal(0, 5e-3, 3)
    return dict(K=K, X=Xn, wcov=wcov, uv=uv, icov=icov, Rg=Rg, tg=tg, sig=sig, R_true=R_true, t_true=t_true)


def test_oracle_pnp_refine_is_the_minimiser():
    pb = pnp_problem(5, 20)
    m = 20
    res = o.pnp_refine(pb["X"], pb["wcov"], pb["uv"], pb["icov"], pb["K"], pb["Rg"], pb["tg"])
    assert res["ok"]
    Lw = np.linalg.cholesky(np.linalg.inv(pb["wcov"]))   # info = L L^T -> whitened residual L^T d

    def resid(x):
        R = pb["Rg"] @ Rot.from_rotvec(x[0:3]).as_matrix()
        t = x[3:6]
        P = x[6:].reshape(m, 3)
        return np.concatenate([
            Rot.from_matrix(pb["Rg"].T @ R).as_rotvec() / 1e-2, pb["Rg"].T @ (t - pb["tg"]) / 1e-2,
            np.einsum("mji,mj->mi", Lw, P - pb["X"]).ravel(), ((proj(pb["K"], R, t, P) - pb["uv"]) / pb["sig"]).ravel()])

    x0 = np.concatenate([np.zeros(3), pb["tg"], pb["X"].ravel()])
    sol = least_squares(resid, x0, xtol=1e-15, ftol=1e-15, gtol=1e-15, method="trf", jac="3-point", x_scale="jac")
    R1 = pb["Rg"] @ Rot.from_rotvec(sol.x[0:3]).as_matrix()
    assert abs(0.5 * np.sum(sol.fun ** 2) - res["error"]) <= 1e-9 * max(res["error"], 1.0)
    assert np.abs(R1 - res["R"]).max() < 1e-9 and np.abs(sol.x[3:6] - res["t"]).max() < 1e-9
    assert np.all(np.linalg.eigvalsh(res["pose_cov"]) > 0)


def test_oracle_cheirality_point_does_not_break_the_solve():
    pb = two_view_problem(7, 16)
    pb["Xg"][3, 2] = -1.0          # a guess behind both cameras: constant residual, zero Jacobian, pulled by its prior only
    res = o.sfm_refine(pb["p1"], pb["cov"], pb["p2"], pb["cov"], pb["K"], pb["Rg"], pb["tg"], pb["Xg"])
    assert res["ok"] and np.isfinite(res["error"])
    assert np.allclose(res["points"][3], pb["Xg"][3], atol=1e-9)
    assert np.abs(res["t"] - pb["t_true"]).max() < 0.05


def track_refine_problem(seed, m, n_new):
    """the shape of VisualOdometer::track_refine (front-end/visual-odometer.cpp:618-800): the last frame anchored at its
    own pose, the new frame regularised, tracked points with isotropic priors, n_new new points without any, and each
    frame missing some observations (every point keeps at least one; prior-less points keep both)"""
    rng = np.random.default_rng(seed)
    K = np.array([[525.0, 0, 320], [0, 525, 240], [0, 0, 1]])
    X = np.stack([rng.uniform(-2, 2, m), rng.uniform(-1.5, 1.5, m), rng.uniform(4, 9, m)], 1)
    Ra = Rot.from_rotvec([0.02, -0.1, 0.03]).as_matrix()
    ta = np.array([0.4, -0.1, 0.2])
    Rb = Ra @ Rot.from_rotvec(rng.normal(0, 0.02, 3)).as_matrix()
    tb = ta + np.array([0.3, 0.02, 0.05])
    sig = 0.5
    obs = [proj(K, Ra, ta, X) + rng.normal(0, sig, (m, 2)), proj(K, Rb, tb, X) + rng.normal(0, sig, (m, 2))]
    cov = np.tile((np.eye(2) * sig ** 2).reshape(4), (m, 1))
    has_prior = np.ones(m, bool)
    has_prior[rng.choice(m, n_new, replace=False)] = False
    valid = [np.ones(m, np.uint8), np.ones(m, np.uint8)]
    for i in rng.choice(np.nonzero(has_prior)[0], m // 5, replace=False):
        valid[int(rng.integers(0, 2))][i] = 0
    pcov = np.zeros((m, 9))
    pcov[has_prior] = (np.eye(3) * 1e-2 ** 2).reshape(9)
    Xg = X + rng.normal(0, 5e-3, X.shape)
    Rbg = Rb @ Rot.from_rotvec(rng.normal(0, 5e-3, 3)).as_matrix()
    tbg = tb + rng.normal(0, 5e-3, 3)
    poses = np.stack([np.concatenate([Ra.reshape(9), ta]), np.concatenate([Rbg.reshape(9), tbg])])
    var = np.stack([np.full(6, 1e-5), np.full(6, 1e-2)])     # the reference passes the stddev as the variance (:687-699)
    return dict(K=K, X=X, poses=poses, var=var, Xg=Xg, pcov=pcov, obs=obs, cov=[cov, cov], valid=valid, has_prior=has_prior,
                sig=sig, Rb=Rb, tb=tb)


def test_oracle_ba_refine_general_two_frame_problem_is_the_minimiser():
    pb = track_refine_problem(3, 24, 5)
    m = 24
    res = o.ba_refine(pb["K"], pb["poses"], pb["var"], pb["Xg"], pb["pcov"], pb["obs"], pb["cov"], pb["valid"])
    assert res["ok"]
    Rg = [pb["poses"][f][:9].reshape(3, 3) for f in range(2)]
    tg = [pb["poses"][f][9:] for f in range(2)]

    def resid(x):
        r = []
        P = x[12:].reshape(m, 3)
        for f in range(2):
            R = Rg[f] @ Rot.from_rotvec(x[6 * f:6 * f + 3]).as_matrix()
            t = x[6 * f + 3:6 * f + 6]
            sd = np.sqrt(pb["var"][f])
            r += [Rot.from_matrix(Rg[f].T @ R).as_rotvec() / sd[:3], Rg[f].T @ (t - tg[f]) / sd[3:]]
            e = (proj(pb["K"], R, t, P) - pb["obs"][f]) / pb["sig"]
            r.append((e * pb["valid"][f][:, None]).ravel())
        r.append((((P - pb["Xg"]) / 1e-2) * pb["has_prior"][:, None]).ravel())
        return np.concatenate(r)

    x0 = np.concatenate([np.zeros(3), tg[0], np.zeros(3), tg[1], pb["Xg"].ravel()])
    sol = least_squares(resid, x0, xtol=1e-15, ftol=1e-15, gtol=1e-15, method="trf", jac="3-point", x_scale="jac")
    assert abs(0.5 * np.sum(sol.fun ** 2) - res["error"]) <= 1e-9 * res["error"]
    for f in range(2):
        R = Rg[f] @ Rot.from_rotvec(sol.x[6 * f:6 * f + 3]).as_matrix()
        assert np.abs(R - res["R"][f]).max() < 1e-8 and np.abs(sol.x[6 * f + 3:6 * f + 6] - res["t"][f]).max() < 1e-8
    assert np.abs(sol.x[12:].reshape(m, 3) - res["points"]).max() < 1e-7
    # the anchored frame barely moves, the new frame gets close to the truth
    assert np.abs(res["t"][0] - tg[0]).max() < 1e-3 and np.abs(res["t"][1] - pb["tb"]).max() < 0.02
    # sfm_refine is the special case: camera 1 at the identity, sigma 1e-5 / 1e-2, all points with sigma 1e-2
    tv = two_view_problem(1, 12)
    a = o.sfm_refine(tv["p1"], tv["cov"], tv["p2"], tv["cov"], tv["K"], tv["Rg"], tv["tg"], tv["Xg"])
    poses = np.stack([np.concatenate([np.eye(3).reshape(9), np.zeros(3)]), np.concatenate([tv["Rg"].reshape(9), tv["tg"]])])
    b = o.ba_refine(tv["K"], poses, np.stack([np.full(6, 1e-10), np.full(6, 1e-4)]), tv["Xg"],
                    np.tile((np.eye(3) * 1e-4).reshape(9), (12, 1)), [tv["p1"], tv["p2"]], [tv["cov"], tv["cov"]], [None, None])
    assert np.abs(a["R"] - b["R"][1]).max() < 1e-12 and np.abs(a["t"] - b["t"][1]).max() < 1e-12
    assert np.abs(a["pose_cov"] - b["pose_cov"][1]).max() <= 1e-9 * np.abs(a["pose_cov"]).max()


def _golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "refine_small.npz"))


def test_oracle_reproduces_golden_refine_vectors():
    g = _golden()
    for tag in ("s", "l"):
        r = o.sfm_refine(g[tag + "_p1"], g[tag + "_cov"], g[tag + "_p2"], g[tag + "_cov"], g[tag + "_K"], g[tag + "_Rg"],
                         g[tag + "_tg"], g[tag + "_Xg"])
        assert r["ok"] and r["iterations"] == int(g[tag + "_out_iterations"])
        for k in ("R", "t", "points"):
            assert np.abs(r[k] - g[tag + "_out_" + k]).max() < 1e-12
        assert abs(r["error"] - float(g[tag + "_out_error"])) <= 1e-12 * r["error"]
        assert np.abs(r["pose_cov"] - g[tag + "_out_pose_cov"]).max() <= 1e-9 * np.abs(r["pose_cov"]).max()
    r = o.pnp_refine(g["p_X"], g["p_wcov"], g["p_uv"], g["p_icov"], g["p_K"], g["p_Rg"], g["p_tg"])
    assert r["ok"] and np.abs(r["R"] - g["p_out_R"]).max() < 1e-12 and np.abs(r["t"] - g["p_out_t"]).max() < 1e-12


def test_header_and_ctypes_layouts_agree():
    from mvslam_amd import capi

    probe = r'''
#include <stdio.h>
#include <stddef.h>
#include "mvslam_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(mvs_refine_params), sizeof(mvs_refine_result),
         offsetof(mvs_refine_params, anchor_sigma), offsetof(mvs_refine_params, point_sigma),
         offsetof(mvs_refine_result, R), offsetof(mvs_refine_result, pose_cov));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(probe)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "p"),
                               os.path.join(d, "p.c")])
        v = list(map(int, subprocess.check_output([os.path.join(d, "p")]).decode().split()))
    P = capi.RefineParams
    assert v[0] == C.sizeof(P) == C.sizeof(o.RefineParams)
    assert v[1] == capi.REFINE_DTYPE.itemsize
    assert v[2] == P.anchor_sigma.offset and v[3] == P.point_sigma.offset
    assert v[4] == capi.REFINE_DTYPE.fields["R"][1] and v[5] == capi.REFINE_DTYPE.fields["pose_cov"][1]


# ---------------------------------------------------------------------------------------------------------------------
# GPU: the HIP kernel through the C ABI against the oracle

def _close(a, b, rel, what):
    scale = max(np.abs(b).max(), 1e-300)
    err = np.abs(np.asarray(a) - np.asarray(b)).max() / scale
    assert err <= rel, "%s: relative error %.3e > %.1e" % (what, err, rel)


@pytest.mark.gpu
@pytest.mark.parametrize("m,seed,pix", [(1, 3, False), (12, 1, False), (300, 2, True), (1100, 4, True), (4096, 5, True)])
def test_gpu_sfm_refine_matches_oracle(ctx, m, seed, pix):
    K = np.array([[525.0, 0, 320], [0, 525, 240], [0, 0, 1]]) if pix else np.eye(3)
    pb = two_view_problem(seed, m, K=K, sig=0.5 if pix else None, baseline=0.3 if pix else 1.0,
                          depth=(2.0, 10.0) if pix else (2.0, 4.0))
    ref = o.sfm_refine(pb["p1"], pb["cov"], pb["p2"], pb["cov"], pb["K"], pb["Rg"], pb["tg"], pb["Xg"])
    got = ctx.sfm_refine(pb["p1"], pb["cov"], pb["p2"], pb["cov"], pb["K"], pb["Rg"], pb["tg"], pb["Xg"])
    assert got["ok"] == ref["ok"] is True
    assert got["iterations"] == ref["iterations"]
    assert abs(got["error"] - ref["error"]) <= 1e-10 * ref["error"]
    assert np.abs(got["R"] - ref["R"]).max() < 1e-10 and np.abs(got["t"] - ref["t"]).max() < 1e-10
    assert np.abs(got["points"] - ref["points"]).max() < 1e-9
    _close(got["pose_cov"], ref["pose_cov"], 1e-7, "pose_cov")
    _close(got["point_cov"], ref["point_cov"], 1e-7, "point_cov")


@pytest.mark.gpu
def test_gpu_refine_against_golden(ctx):
    """Oracle-free: the committed expected outputs (tests/golden/refine_small.npz)."""
    g = _golden()
    for tag in ("s", "l"):
        r = ctx.sfm_refine(g[tag + "_p1"], g[tag + "_cov"], g[tag + "_p2"], g[tag + "_cov"], g[tag + "_K"], g[tag + "_Rg"],
                           g[tag + "_tg"], g[tag + "_Xg"])
        assert r["ok"] and r["iterations"] == int(g[tag + "_out_iterations"])
        assert np.abs(r["R"] - g[tag + "_out_R"]).max() < 1e-10 and np.abs(r["t"] - g[tag + "_out_t"]).max() < 1e-10
        assert np.abs(r["points"] - g[tag + "_out_points"]).max() < 1e-9
        assert abs(r["error"] - float(g[tag + "_out_error"])) <= 1e-10 * float(g[tag + "_out_error"])
        _close(r["pose_cov"], g[tag + "_out_pose_cov"], 1e-7, "pose_cov")
        _close(r["point_cov"], g[tag + "_out_point_cov"], 1e-7, "point_cov")
    r = ctx.pnp_refine(g["p_X"], g["p_wcov"], g["p_uv"], g["p_icov"], g["p_K"], g["p_Rg"], g["p_tg"])
    assert r["ok"] and np.abs(r["R"] - g["p_out_R"]).max() < 1e-10 and np.abs(r["t"] - g["p_out_t"]).max() < 1e-10
    _close(r["pose_cov"], g["p_out_pose_cov"], 1e-7, "pose_cov")


@pytest.mark.gpu
def test_gpu_sfm_refine_identity_covariance_and_reference_kat(ctx):
    pb = l_shape_refine_problem(0)
    got = ctx.sfm_refine(pb["p1"], pb["cov"], pb["p2"], pb["cov"], pb["K"], pb["Rg"], pb["tg"], pb["Xg"])
    se3 = o.se3_ln(got["R"], got["t"])
    assert got["ok"] and np.abs(se3 - np.array([1.0, 0, 0, 0, 0, 0])).max() < 0.025
    assert np.abs(got["points"] - pb["X"]).max() < 0.025
    # NULL covariances = identity
    ref = o.sfm_refine(pb["p1"], None, pb["p2"], None, pb["K"], pb["Rg"], pb["tg"], pb["Xg"])
    got = ctx.sfm_refine(pb["p1"], None, pb["p2"], None, pb["K"], pb["Rg"], pb["tg"], pb["Xg"], point_cov=False)
    assert got["ok"] and np.abs(got["t"] - ref["t"]).max() < 1e-10 and got["point_cov"] is None


@pytest.mark.gpu
@pytest.mark.parametrize("m,seed", [(7, 5), (200, 6), (2048, 7)])
def test_gpu_pnp_refine_matches_oracle(ctx, m, seed):
    pb = pnp_problem(seed, m)
    ref = o.pnp_refine(pb["X"], pb["wcov"], pb["uv"], pb["icov"], pb["K"], pb["Rg"], pb["tg"])
    got = ctx.pnp_refine(pb["X"], pb["wcov"], pb["uv"], pb["icov"], pb["K"], pb["Rg"], pb["tg"])
    assert got["ok"] and ref["ok"] and got["iterations"] == ref["iterations"]
    assert abs(got["error"] - ref["error"]) <= 1e-10 * max(ref["error"], 1.0)
    assert np.abs(got["R"] - ref["R"]).max() < 1e-10 and np.abs(got["t"] - ref["t"]).max() < 1e-10
    _close(got["pose_cov"], ref["pose_cov"], 1e-7, "pose_cov")
    # the refined pose is closer to the truth than a 5e-3 guess on average; at least it must not be far
    assert np.abs(got["t"] - pb["t_true"]).max() < 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("m,n_new,seed", [(24, 5, 3), (400, 60, 4), (1500, 0, 5)])
def test_gpu_ba_refine_matches_oracle(ctx, m, n_new, seed):
    """mvs_ba_refine on track_refine-shaped problems: own anchor pose, prior-less points, missing observations"""
    pb = track_refine_problem(seed, m, n_new)
    want = o.ba_refine(pb["K"], pb["poses"], pb["var"], pb["Xg"], pb["pcov"], pb["obs"], pb["cov"], pb["valid"])
    got = ctx.ba_refine(pb["K"], pb["poses"], pb["var"], pb["Xg"], pb["pcov"], pb["obs"], pb["cov"], pb["valid"])
    assert got["ok"] and want["ok"] and got["iterations"] == want["iterations"]
    assert abs(got["error"] - want["error"]) <= 1e-10 * want["error"]
    assert np.abs(got["R"] - want["R"]).max() < 1e-9 and np.abs(got["t"] - want["t"]).max() < 1e-9
    assert np.abs(got["points"] - want["points"]).max() < 1e-8
    for f in range(2):
        _close(got["pose_cov"][f], want["pose_cov"][f], 1e-6, "pose_cov[%d]" % f)
    _close(got["point_cov"], want["point_cov"], 1e-6, "point_cov")
    # one frame, no point priors at all -> every point is unconstrained along its ray: no model, never a crash
    one = ctx.ba_refine(pb["K"], pb["poses"][1:], pb["var"][1:], pb["Xg"], None, pb["obs"][1:], pb["cov"][1:], [None])
    assert not one["ok"]


@pytest.mark.gpu
def test_gpu_refine_argument_errors(ctx):
    from mvslam_amd import capi

    pb = two_view_problem(1, 12)
    bad = capi.default_refine_params(point_sigma=0.0)
    with pytest.raises(capi.MvsError) as e:
        ctx.sfm_refine(pb["p1"], pb["cov"], pb["p2"], pb["cov"], pb["K"], pb["Rg"], pb["tg"], pb["Xg"], params=bad)
    assert e.value.status == capi.MVS_ERR_INVALID_ARG
    Kbad = np.eye(3)
    Kbad[2, 0] = 1e-3
    with pytest.raises(capi.MvsError) as e:
        ctx.sfm_refine(pb["p1"], pb["cov"], pb["p2"], pb["cov"], Kbad, pb["Rg"], pb["tg"], pb["Xg"])
    assert e.value.status == capi.MVS_ERR_BAD_INTRINSICS
    # a non-finite guess: no model, never a hang
    Xg = pb["Xg"].copy()
    Xg[0, 0] = np.nan
    got = ctx.sfm_refine(pb["p1"], pb["cov"], pb["p2"], pb["cov"], pb["K"], pb["Rg"], pb["tg"], Xg)
    assert not got["ok"]


@pytest.mark.gpu
def test_gpu_batch_refine_matches_oracle_and_improves_reprojection(ctx):
    """ImagePair::refine on the device for a whole batch, from the batch's own results; every keypoint weighted by
    its pyramid octave as VisualFeature::get_point_estimates does (visual-feature.cpp:192-207: stddev = 2^octave / 2)."""
    from mvslam_amd import capi, synth

    n_pairs, n_kp = 4, 600
    data = synth.make_batch(0, n_pairs, n_kp=n_kp)
    rng = np.random.default_rng(5)
    oct1 = rng.integers(0, 4, size=(n_pairs, n_kp)).astype(np.uint8)
    oct2 = rng.integers(0, 4, size=(n_pairs, n_kp)).astype(np.uint8)
    oct2[3] = 0                     # one pair with the default everywhere in the second image
    b = capi.Batch(ctx, n_pairs, n_kp)
    b.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"],
             data["global_index"])
    b.upload_octaves(0, oct1, None)
    b.upload_octaves(0, None, oct2[:3])      # pair 3's second image keeps the octave-0 default
    with pytest.raises(capi.MvsError) as e:
        b.upload_octaves(0, np.full((1, n_kp), 31, np.uint8), None)
    assert e.value.status == capi.MVS_ERR_INVALID_ARG
    prm = capi.default_params(num_hypotheses=2048, sampler=capi.SAMPLER_PHILOX, seed=11, max_error_sq=1e-2)
    b.run(prm)
    b.refine(sigma_px=0.5)
    b.sync()
    out = b.download()
    ref = b.download_refined(points=True, point_cov=True)
    b.close()
    K = synth.K_DEFAULT
    n_checked = 0
    for p in range(n_pairs):
        r = out["results"][p]
        if not r["valid"]:
            assert ref["refined"][p]["ok"] == 0
            continue
        n = int(r["n_points"])
        mt = out["matches"][p][out["point_idx"][p][:n]]
        p1 = data["kp1"][p][mt["trainIdx"]].astype(np.float64)
        p2 = data["kp2"][p][mt["queryIdx"]].astype(np.float64)
        s1 = 0.5 * 2.0 ** oct1[p][mt["trainIdx"]].astype(np.float64)
        s2 = 0.5 * 2.0 ** oct2[p][mt["queryIdx"]].astype(np.float64)
        cov1 = (s1 * s1)[:, None] * np.eye(2).reshape(1, 4)
        cov2 = (s2 * s2)[:, None] * np.eye(2).reshape(1, 4)
        want = o.sfm_refine(p1, cov1, p2, cov2, K, r["R"], r["t"], out["points"][p][:n])
        got = ref["refined"][p]
        assert got["ok"] == 1 and want["ok"]
        assert abs(got["error"] - want["error"]) <= 1e-9 * want["error"]
        assert np.abs(got["R"] - want["R"]).max() < 1e-9 and np.abs(got["t"] - want["t"]).max() < 1e-9
        assert np.abs(ref["points"][p][:n] - want["points"]).max() < 1e-8
        _close(got["pose_cov"], want["pose_cov"], 1e-6, "pose_cov")
        _close(ref["point_cov"][p][:n], want["point_cov"], 1e-6, "point_cov")
        # whitened reprojection RMS over both images does not get worse
        def rms(R, t, X):
            e1 = (proj(K, np.eye(3), np.zeros(3), X) - p1) / s1[:, None]
            e2 = (proj(K, R, t, X) - p2) / s2[:, None]
            return np.sqrt(np.mean(np.concatenate([e1, e2]) ** 2))
        assert rms(got["R"], got["t"], ref["points"][p][:n]) <= rms(r["R"], r["t"], out["points"][p][:n]) + 1e-12
        n_checked += 1
    assert n_checked >= 3
