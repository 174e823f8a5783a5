"""CPU tests of the oracle: the reference's own known-answer tests (the ones reachable without OpenCV/Eigen/GTSAM),
LAPACK / brute-force cross-checks, and the committed golden vectors.  These PIN the oracle before it is used to judge
the GPU path.  Reference test files are cited per test (paths relative to the reference tree)."""
import os

import numpy as np
import pytest

import helpers
import oracle_lib as o

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---------------------------------------------------------------- test/test-svd.cpp:10-68 (tolerance 1e-3)
def test_svd_identity_3x3():
    w, u, vt = o.svd(np.eye(3))
    assert np.abs(w - 1).max() < 1e-3 and np.abs(u - np.eye(3)).max() < 1e-3 and np.abs(vt.T - np.eye(3)).max() < 1e-3


def test_svd_homogeneous_2x3():
    """V.col(2) == (0.57735, -0.57735, 0.57735) WITH this sign: produced by OpenCV's fixed-seed RNG completion of
    the null space, so it pins the restated RNG + Gram-Schmidt fallback (m < n branch of cv::SVDecomp)."""
    A = np.array([[1.0, 1, 0], [0, 1, 1]])
    w, u, vt = o.svd(A)
    X = vt.T[:, 2]
    assert abs(X[0] - 0.57735) < 1e-3 and abs(X[1] + 0.57735) < 1e-3 and abs(X[2] - 0.57735) < 1e-3
    assert np.abs(A @ X).max() < 1e-3


def test_svd_random_4x5():
    rng = np.random.default_rng(3)
    A = rng.uniform(-1, 1, (4, 5))
    w, u, vt = o.svd(A)
    S = np.zeros((4, 5))
    S[:4, :4] = np.diag(w)
    assert np.abs(u @ S @ vt - A).max() < 1e-3
    assert np.abs(u @ S @ vt - A).max() < 1e-12


# ---------------------------------------------------------------- SVD contract vs LAPACK (SURVEY 8(c) KAT 2)
@pytest.mark.parametrize("n", [3, 4, 9])
def test_svd_matches_lapack(n):
    rng = np.random.default_rng(100 + n)
    for trial in range(20):
        if n == 9:
            B = rng.normal(size=(8, 9))
            A = B.T @ B          # the A^T A of the 8-point solver: rank 8, one null direction
        else:
            A = rng.normal(size=(n, n))
        w, u, vt = o.svd(A)
        wl = np.linalg.svd(A, compute_uv=False)
        assert np.abs(w - wl).max() <= 1e-12 * wl[0]                  # sigma to 1e-12 relative
        assert (np.diff(w) <= 0).all()                                # descending
        assert np.abs(u @ u.T - np.eye(n)).max() < 1e-12 and np.abs(vt @ vt.T - np.eye(n)).max() < 1e-12
        assert np.abs(u @ np.diag(w) @ vt - A).max() < 1e-12 * max(1.0, wl[0])
        if n == 9:
            assert np.abs(B @ vt[8]).max() < 1e-12 * np.abs(B).max() * 10   # null-vector residual


def test_svd_exact_zero_singular_value_completion():
    """An exactly zero column (the cube's essential matrix) goes through the random completion; U must stay
    orthonormal."""
    E = np.array([[0.0, 0, 0], [0, 0, -1], [0, 1, 0]])
    w, u, vt = o.svd(E)
    assert w[2] == 0.0 and abs(w[0] - 1) < 1e-15 and abs(w[1] - 1) < 1e-15
    assert np.abs(u @ u.T - np.eye(3)).max() < 1e-12
    assert np.abs(u @ np.diag(w) @ vt - E).max() < 1e-15


# ---------------------------------------------------------------- test/test-lie-group.cpp:22-132 (tolerance 0.01)
def test_lie_group_so3():
    roll, pitch, yaw = 0.1, -0.2, 0.3
    R = o.so3_from_rpy(roll, pitch, yaw)
    assert abs(np.arctan2(R[2, 1], R[2, 2]) - roll) < 0.01        # get_roll  (lie-group.hpp:98-101)
    assert abs(np.arcsin(-R[2, 0]) - pitch) < 0.01                # get_pitch
    assert abs(np.arctan2(R[1, 0], R[0, 0]) - yaw) < 0.01         # get_yaw
    Rinv = o.so3_rectify(R.T)
    assert np.abs(Rinv @ R - np.eye(3)).max() < 0.01
    assert np.abs(o.so3_ln(np.eye(3))).max() < 0.01
    assert np.abs(o.rodrigues(o.so3_ln(R)) - R).max() < 0.01
    assert np.abs(o.rodrigues(o.so3_ln(R)) - R).max() < 1e-12


def test_lie_group_se3():
    R = o.so3_from_rpy(0.1, -0.2, 0.3)
    t = np.array([1.0, 2.0, -3.0])
    Ri, ti = o.se3_inverse(R, t)
    Rc, tc = o.se3_compose(Ri, ti, R, t)
    assert np.abs(Rc - np.eye(3)).max() < 0.01 and np.abs(tc).max() < 0.01
    assert np.abs(o.se3_ln(np.eye(3), np.zeros(3))).max() < 0.01
    se3 = o.se3_ln(R, t)
    R2, t2 = o.se3_exp(se3)
    assert np.abs(R2 - R).max() < 0.01 and np.abs(t2 - t).max() < 0.01
    assert np.abs(R2 - R).max() < 1e-12 and np.abs(t2 - t).max() < 1e-12


def test_rectify_leaves_second_row_unnormalised():
    """SURVEY Q8 (lie-group.hpp:89-95)."""
    M = np.array([[2.0, 0, 0], [1, 3, 0], [0, 0, 1]])
    R = o.so3_rectify(M)
    assert np.allclose(R[0], [1, 0, 0]) and np.allclose(R[1], [0, 3, 0]) and np.allclose(R[2], [0, 0, 3])


# ---------------------------------------------------------------- test/test-camera.cpp:18-69 (tolerance 1e-7)
def test_camera_projection():
    K = np.array([[0.5, 0, 20], [0, 0.5, 10], [0, 0, 1.0]])
    assert np.abs(o.project_points(np.eye(3), np.eye(3), np.zeros(3), [[1.0, 1, 1]]) - [[1, 1]]).max() < 1e-7
    R, t = o.se3_exp(np.array([0, 0, 1, 0, 0, 0.5 * 3.14159265358]))
    uv = o.project_points(K, R, t, [[3.0, 2.0, 1.0]])
    assert abs(uv[0, 0] - (-0.50 + 20)) < 1e-7 and abs(uv[0, 1] - (0.75 + 10)) < 1e-7
    pts = np.array([[x, y, 1.0] for x in (-1, 0, 1) for y in (-1, 0, 1)])
    uv = o.project_points(K, np.eye(3), np.zeros(3), pts)
    assert np.abs(uv - (pts[:, :2] * 0.5 + [20, 10])).max() < 1e-7


def test_normalize_points_inverts_projection():
    K = np.array([[525.0, 0, 320], [0, 525, 240], [0, 0, 1]])
    uv = np.array([[320.0, 240], [0, 0], [639.5, 479.25]])
    xy = o.normalize_points(K, uv)
    assert np.abs(xy * 525 + [320, 240] - uv).max() < 1e-10
    assert np.abs(o.mat3_inverse(K) @ K - np.eye(3)).max() < 1e-12


# ---------------------------------------------------------------- matcher (visual-feature.cpp:51-80)
def _match_numpy(train, query, ratio, max_dist):
    """Independent brute-force restatement: bit counts via unpackbits, stable argsort for the 2-NN."""
    tb, qb = np.unpackbits(train, axis=1), np.unpackbits(query, axis=1)
    D = (qb[:, None, :] != tb[None, :, :]).sum(axis=2)
    out = []
    for q in range(len(query)):
        order = np.argsort(D[q], kind="stable")       # ties -> smaller train index first
        d0, d1 = float(np.float32(D[q, order[0]])), float(np.float32(D[q, order[1]]))
        if d0 < ratio * d1 and (max_dist < 0 or d0 <= max_dist):
            out.append((d0, q, int(order[0])))
    out.sort()
    return out


def test_match_against_bruteforce_numpy():
    rng = np.random.default_rng(9)
    train = rng.integers(0, 256, size=(70, 32), dtype=np.uint8)
    query = rng.integers(0, 256, size=(50, 32), dtype=np.uint8)
    for i in range(25):
        query[i] = train[2 * i]
        query[i, i % 32] ^= 3
    train[69] = train[4]
    for ratio, md in ((0.7, -1.0), (0.7, 1.0), (0.99, 200.0)):
        got = o.match_visual_features(train, query, ratio, md)
        ref = _match_numpy(train, query, ratio, md)
        assert [(float(m["distance"]), int(m["queryIdx"]), int(m["trainIdx"])) for m in got] == ref
        assert (got["imgIdx"] == 0).all()


def test_match_preconditions():
    d = np.zeros((4, 32), np.uint8)
    assert o.match_visual_features(d[:1], d) is None      # fewer than two train rows (reference: UB at :67)
    assert o.match_visual_features(d, d[:0]) is None      # invalid VisualFeature (reference: assert at :56)


# ---------------------------------------------------------------- sampler
def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    assert o.philox4x32_10([0] * 4, [0] * 2).tolist() == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert o.philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2).tolist() == [0x408F276D, 0x41C83B0E, 0xA20BC7C6,
                                                                           0x6D5451FD]
    assert o.philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]).tolist() == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_sampler_properties():
    for M in (8, 9, 10, 100, 2000, 4096):
        seen = np.zeros(M, dtype=np.int64)
        for h in range(400):
            idx = o.sample8(0xC0FFEE, h, M)
            assert len(set(idx.tolist())) == 8 and idx.min() >= 0 and idx.max() < M
            seen[idx] += 1
        if M == 8:
            assert (seen == 400).all()
        if M == 100:
            assert seen.min() > 5          # every index gets drawn: 3200 draws over 100 slots
    assert o.sample8(1, 2, 50).tolist() == o.sample8(1, 2, 50).tolist()
    assert o.sample8(1, 2, 50).tolist() != o.sample8(1, 3, 50).tolist()
    assert o.sample8(7, 0, 50, o.SAMPLER_IDENTITY).tolist() == list(range(8))   # reference behaviour


# ---------------------------------------------------------------- 8-point / RANSAC
def test_eight_point_recovers_analytic_essential():
    """SURVEY 8(c) KAT 1: L-shape rig, noise free: F is [t]x R up to scale and sign."""
    rig = helpers.two_camera_rig("L", rpy=(1.5, 0.7, 0.0), scale=0.5, se3_2to1=(1, 0.2, -0.1, 0.05, 0.1, -0.02))
    ok, F = o.find_fundamental_matrix(rig["uv1"], rig["uv2"])
    assert ok
    R12, t12 = rig["T1to2"]
    E = helpers.skew(t12) @ R12
    Fn, En = F / np.linalg.norm(F), E / np.linalg.norm(E)
    assert min(np.abs(Fn - En).max(), np.abs(Fn + En).max()) < 1e-9
    x1 = np.c_[rig["uv1"], np.ones(8)]
    x2 = np.c_[rig["uv2"], np.ones(8)]
    assert np.abs(np.einsum("ij,jk,ik->i", x2, F, x1)).max() < 1e-12
    assert abs(np.linalg.det(F)) < 1e-12                      # rank 2


def test_eight_point_degenerate_samples():
    p = np.tile([[0.3, -0.2]], (8, 1))
    ok, _ = o.find_fundamental_matrix(p, p)                   # reference: assert(scale > epsilon)
    assert not ok


def test_ransac_selection_rule_and_tables():
    import test_gpu_parity as T

    p1, p2 = T._scene(5, 90, 2e-4)
    thr, H = 1e-3, 200
    r = o.ransac_fundamental(p1, p2, thr, H, o.SAMPLER_PHILOX, seed=11, per_hyp=True)
    # sequential replacement (estimator-RANSAC.cpp:76-84) == lexicographic arg-best (count desc, residual asc, id asc)
    best, bc, br = -1, 0, np.finfo(float).max / 10
    for h in range(H):
        if r["count"][h] < 0:
            continue
        if r["count"][h] > bc or (r["count"][h] == bc and r["residual"][h] < br):
            best, bc, br = h, r["count"][h], r["residual"][h]
    assert (best, bc, br) == (r["best_hyp"], r["best_count"], r["best_residual"])
    # the table entry of the winner is what count_inliers gives for its F, and so is the mask
    n, res, mask = o.count_inliers(p1, p2, r["F"], thr)
    assert n == bc and res == br and np.array_equal(mask, r["mask"]) and mask.sum() == n
    # each hypothesis is the 8-point fit of its sample
    for h in (0, 17, best):
        idx = o.sample8(11, h, 90)
        ok, F = o.find_fundamental_matrix(p1[idx], p2[idx])
        n, res, _ = o.count_inliers(p1, p2, F, thr)
        assert ok and n == r["count"][h] and res == r["residual"][h]
    assert not o.ransac_fundamental(p1[:7], p2[:7], thr, 5)["ok"]          # < 8 pairs (:25-29)


# ---------------------------------------------------------------- test/test-sfm.cpp geometry
def test_sfm_solve_L_shape():
    """test-sfm.cpp:17-90 with the non-degenerate rig: pose.ln() == (1,0,0,0,0,0), points in order, tol 1e-3."""
    rig = helpers.two_camera_rig("L", rpy=(1.5, 0.7, 0.0), scale=0.5)
    r = o.sfm_solve(rig["uv1"], rig["uv2"], rig["K"], o.make_params(1, o.SAMPLER_IDENTITY))
    assert r["ok"] and r["n_points"] == 8 and r["point_idx"].tolist() == list(range(8))
    assert np.abs(o.se3_ln(r["R"], r["t"]) - [1, 0, 0, 0, 0, 0]).max() < 1e-3
    assert np.abs(r["points"] - rig["X"]).max() < 1e-3


def test_sfm_solve_cube_is_degenerate_for_8_point():
    """the cube + both camera centres lie on a ruled quadric; the normalised design matrix has
    rank 7, so the reference's cube test pins cv::findEssentialMat, not the 8-point estimator."""
    rig = helpers.two_camera_rig("cube")
    x1, x2 = rig["uv1"], rig["uv2"]
    A = np.stack([x2[:, 0] * x1[:, 0], x2[:, 0] * x1[:, 1], x2[:, 0], x2[:, 1] * x1[:, 0], x2[:, 1] * x1[:, 1],
                  x2[:, 1], x1[:, 0], x1[:, 1], np.ones(8)], axis=1)
    s = np.linalg.svd(A, compute_uv=False)
    assert s[6] > 1e-3 and s[7] < 1e-12


def test_sfm_triangulate_cube():
    """test-sfm.cpp:92-155."""
    rig = helpers.two_camera_rig("cube")
    pts, idx = o.sfm_triangulate(rig["uv1"], rig["uv2"], rig["K"], (np.eye(3), np.zeros(3)), rig["pose2in1"])
    assert idx.tolist() == list(range(8)) and np.abs(pts - rig["X"]).max() < 1e-3


def test_recover_pose_cube_from_analytic_E():
    """Decomposition + 4-candidate cheirality selection + pose convention of test-sfm.cpp:17-90."""
    rig = helpers.two_camera_rig("cube")
    R12, t12 = rig["T1to2"]
    ok, R, t, pts, idx = o.recover_pose_and_points(helpers.skew(t12) @ R12, rig["uv1"], rig["uv2"])
    assert ok and idx.tolist() == list(range(8))
    Rp, tp = o.se3_inverse(o.so3_rectify(R), t)
    assert np.abs(o.se3_ln(Rp, tp) - [1, 0, 0, 0, 0, 0]).max() < 1e-3
    assert np.abs(pts - rig["X"]).max() < 1e-3
    Ra, Rb, tt = o.decompose_essential(helpers.skew(t12) @ R12)
    for Rc in (Ra, Rb):
        assert abs(np.linalg.det(Rc) - 1) < 1e-12 and np.abs(Rc @ Rc.T - np.eye(3)).max() < 1e-12
    assert abs(np.linalg.norm(tt) - 1) < 1e-12


def test_project_essential_equal_singular_values():
    rng = np.random.default_rng(8)
    F = rng.normal(size=(3, 3))
    E = o.project_essential(F)
    s = np.linalg.svd(E, compute_uv=False)
    s0 = np.linalg.svd(F, compute_uv=False)
    assert abs(s[0] - s[1]) < 1e-12 and s[2] < 1e-12 and abs(s[0] - np.sqrt(s0[0] * s0[1])) < 1e-12   # :80


def test_triangulation_cheirality_and_mask():
    rig = helpers.two_camera_rig("cube")
    R12, t12 = rig["T1to2"]
    mask = np.array([1, 0, 1, 1, 0, 1, 1, 1], dtype=np.uint8)
    pts, idx = o.triangulate_points(R12, t12, rig["uv1"], rig["uv2"], mask)
    assert idx.tolist() == [0, 2, 3, 5, 6, 7]
    pts, idx = o.triangulate_points(R12, -t12, rig["uv1"], rig["uv2"])      # wrong sign: all behind a camera
    assert len(idx) == 0


# ---------------------------------------------------------------- golden vectors (tests/golden/make_golden.py)
def test_golden_match():
    g = np.load(os.path.join(GOLD, "match_small.npz"))
    for tag in "abc":
        ratio, md = g["params_" + tag]
        got = o.match_visual_features(g["train"], g["query"], float(ratio), float(md))
        assert got.tobytes() == g["matches_" + tag].tobytes()


def test_golden_ransac():
    g = np.load(os.path.join(GOLD, "ransac_small.npz"))
    r = o.ransac_fundamental(g["p1"], g["p2"], float(g["thr"]), int(g["H"]), o.SAMPLER_PHILOX, int(g["seed"]), True)
    assert np.array_equal(r["count"], g["count"]) and r["residual"].tobytes() == g["residual"].tobytes()
    assert r["F"].tobytes() == g["F"].tobytes() and np.array_equal(r["mask"], g["mask"])
    assert [r["best_hyp"], r["best_count"]] == g["best"].tolist() and r["best_residual"] == float(g["best_residual"])
    for h in (0, 5, 383):
        assert o.sample8(int(g["seed"]), h, len(g["p1"])).tolist() == g["samples"][h].tolist()


def test_golden_image_pair():
    g = np.load(os.path.join(GOLD, "image_pair_small.npz"))
    for i in range(2):
        prm = o.make_params(int(g["H"]), o.SAMPLER_PHILOX, int(g["seed"]) + int(g["global_index"][i]),
                            float(g["max_error_sq"]))
        r = o.image_pair(g["desc1"][i], g["kp1"][i], g["desc2"][i], g["kp2"][i], g["K"][i].reshape(3, 3), prm, 0.7,
                         10.0)
        sc = g["scalars_%d" % i]
        assert [int(r["ok"]), r["n_matches"], r["n_inliers"], r["n_points"], r["best_hyp"], r["best_count"]] == sc.tolist()
        for k in ("matches", "mask", "points", "point_idx", "R", "t", "F", "E", "R1to2", "t1to2"):
            assert r[k].tobytes() == g["%s_%d" % (k, i)].tobytes(), k


def test_golden_rig():
    g = np.load(os.path.join(GOLD, "rig_kat.npz"))
    r = o.sfm_solve(g["L_uv1"], g["L_uv2"], np.eye(3), o.make_params(1, o.SAMPLER_IDENTITY))
    assert r["R"].tobytes() == g["L_R"].tobytes() and r["points"].tobytes() == g["L_points"].tobytes()
    assert np.abs(r["points"] - g["L_X"]).max() < 1e-3


# ---------------------------------------------------------------- the contract's fused residual vs the reference's form
def test_unfused_reference_residual_is_a_different_rounding_but_the_same_decisions():
    """tests/contract_sensitivity.py's switch: the reference's `p2^T F p1` (estimator-RANSAC.cpp:114, separate mul / add)
    differs from the contract's fused residual in the last bits, and on a synthetic pair decides every match the same way
    (the full-size study: profiles/r02_contract_sensitivity.json)."""
    import contract_sensitivity as cs
    from mvslam_amd import synth

    p = synth.make_pair(3, n_kp=400)
    ref = o.image_pair(p["desc1"], p["kp1"], p["desc2"], p["kp2"], p["K"], o.make_params(300, o.SAMPLER_PHILOX, 9, 1e-2), 0.7, 10.0)
    Kinv = np.linalg.inv(p["K"])
    mt = ref["matches"]
    x1 = (Kinv @ np.c_[p["kp1"][mt["trainIdx"]].astype(float), np.ones(len(mt))].T).T[:, :2]
    x2 = (Kinv @ np.c_[p["kp2"][mt["queryIdx"]].astype(float), np.ones(len(mt))].T).T[:, :2]
    try:
        n0, r0, m0 = o.count_inliers(x1, x2, ref["F"], 1e-2)
        o.lib().orc_set_residual_form(1)
        n1, r1, m1 = o.count_inliers(x1, x2, ref["F"], 1e-2)
    finally:
        o.lib().orc_set_residual_form(0)
    assert n0 == n1 and np.array_equal(m0, m1)
    assert r0 != r1 and abs(r0 - r1) <= 1e-12 * abs(r0)      # another rounding of the same sum
    out = cs.run(2, 400, 1e-2, n_kp=300, threads=2)
    assert out["winners_changed"] == 0 and out["mask_bits_flipped"] == 0 and out["pairs_with_another_pose"] == 0
