"""Generates tests/golden/*.npz: small input/expected-output vectors for the hot path.

The reference itself cannot run in this container (no OpenCV/Eigen/GTSAM), so these vectors come
from the CPU oracle AFTER it has been pinned by the reference's own known answers (tests/test_oracle_kat.py:
test-svd, test-sfm cube/L-shape geometry, test-lie-group, test-camera).  They are data only -- inputs and expected
outputs -- and serve two purposes: (1) regression pin for the oracle, (2) oracle-free expected values for the
GPU tests.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers  # noqa: E402
import oracle_lib as o  # noqa: E402
from mvslam_amd import synth  # noqa: E402


def refine_vectors():
    """sfm_refine / pnp_refine (row f4): pixel-unit two-view problem, the reference's L-shape KAT, a PnP problem"""
    import test_refine as TR

    K = np.array([[525.0, 0, 320], [0, 525, 240], [0, 0, 1]])
    pb = TR.two_view_problem(11, 40, K=K, sig=0.5, baseline=0.3, depth=(2.0, 10.0))
    rs = o.sfm_refine(pb["p1"], pb["cov"], pb["p2"], pb["cov"], pb["K"], pb["Rg"], pb["tg"], pb["Xg"])
    pl = TR.l_shape_refine_problem(0)
    rl = o.sfm_refine(pl["p1"], pl["cov"], pl["p2"], pl["cov"], pl["K"], pl["Rg"], pl["tg"], pl["Xg"])
    pp = TR.pnp_problem(12, 30)
    rp = o.pnp_refine(pp["X"], pp["wcov"], pp["uv"], pp["icov"], pp["K"], pp["Rg"], pp["tg"])
    assert rs["ok"] and rl["ok"] and rp["ok"]
    d = {}
    for tag, prob, res in (("s", pb, rs), ("l", pl, rl)):
        for k in ("p1", "p2", "cov", "K", "Rg", "tg", "Xg"):
            d[tag + "_" + k] = prob[k]
        for k in ("R", "t", "pose_cov", "points", "point_cov", "error", "iterations"):
            d[tag + "_out_" + k] = res[k]
    for k in ("X", "wcov", "uv", "icov", "K", "Rg", "tg"):
        d["p_" + k] = pp[k]
    for k in ("R", "t", "pose_cov", "error", "iterations"):
        d["p_out_" + k] = rp[k]
    np.savez_compressed(os.path.join(HERE, "refine_small.npz"), **d)


def orb_vectors():
    """extraction (row f3): a small textured image through the oracle; the sampling pattern's hash"""
    import hashlib

    import test_orb as TO

    img = TO.textured(21, 160, 200)
    r = o.orb_extract(img, o.make_orb_params(nfeatures=200, nlevels=3))
    assert r["ok"] and len(r["kp"]) > 100
    np.savez_compressed(os.path.join(HERE, "orb_small.npz"), image=img, nfeatures=200, nlevels=3, kp=r["kp"], desc=r["desc"],
                        pattern_sha16=hashlib.sha256(o.orb_pattern().tobytes()).hexdigest()[:16])


def tsukuba_frames():
    """The reference's own test images (data/tsukuba/1..5.jpg, used by test/test-image-pair.cpp and
    test/test-visual-odometer.cpp), decoded to grayscale once so that the GPU box needs neither the reference tree nor a
    JPEG decoder.  Data, not source.  Needs /root/reference and PIL; run here only."""
    from PIL import Image

    ims = np.stack([np.asarray(Image.open("/root/reference/data/tsukuba/%d.jpg" % i).convert("L")) for i in (1, 2, 3, 4, 5)])
    K = np.array([[350.0, 0, 192], [0, 350, 144], [0, 0, 1]])   # data/tsukuba/camera.config: 350 350 0 192 144
    np.savez_compressed(os.path.join(HERE, "tsukuba_gray.npz"), images=ims, K=K)


def main():
    if sys.argv[1:] == ["refine"]:
        refine_vectors()
        print("refine vectors written to", HERE)
        return
    if sys.argv[1:] == ["orb"]:
        if not os.path.exists(os.path.join(HERE, "orb_small.npz")):   # test_orb reads the pattern hash at import
            np.savez_compressed(os.path.join(HERE, "orb_small.npz"), pattern_sha16="")
        orb_vectors()
        tsukuba_frames()
        print("orb vectors written to", HERE)
        return
    rng = np.random.default_rng(20261003)
    # ---- matcher: random descriptors with planted matches and constructed ties
    train = rng.integers(0, 256, size=(96, 32), dtype=np.uint8)
    query = rng.integers(0, 256, size=(80, 32), dtype=np.uint8)
    for i in range(40):
        query[i] = train[(5 * i) % 96]
        query[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
    train[90] = train[7]   # duplicate train rows -> distance ties
    query[70] = train[7]
    out = {}
    for tag, (ratio, md) in {"a": (0.7, -1.0), "b": (0.7, 10.0), "c": (0.9, 120.0)}.items():
        out["matches_" + tag] = o.match_visual_features(train, query, ratio, md)
        out["params_" + tag] = np.array([ratio, md])
    np.savez_compressed(os.path.join(HERE, "match_small.npz"), train=train, query=query, **out)

    # ---- RANSAC tables on a small synthetic scene
    import test_gpu_parity as T  # the scene generator only

    p1, p2 = T._scene(4242, 120, 2e-4)
    r = o.ransac_fundamental(p1, p2, 1.5e-3, 384, o.SAMPLER_PHILOX, seed=0x1234ABCD5678, per_hyp=True)
    np.savez_compressed(os.path.join(HERE, "ransac_small.npz"), p1=p1, p2=p2, thr=1.5e-3, H=384,
                        seed=np.uint64(0x1234ABCD5678), count=r["count"], residual=r["residual"], F=r["F"],
                        mask=r["mask"], best=np.array([r["best_hyp"], r["best_count"]]),
                        best_residual=r["best_residual"],
                        samples=np.stack([o.sample8(0x1234ABCD5678, h, 120) for h in range(384)]))

    # ---- two full image pairs (match -> RANSAC -> decomposition -> triangulation)
    d = synth.make_batch(900, 2, n_kp=256)
    res = []
    for i in range(2):
        prm = o.make_params(256, o.SAMPLER_PHILOX, 0x5EED0000 + 900 + i, 1e-2)
        res.append(o.image_pair(d["desc1"][i], d["kp1"][i], d["desc2"][i], d["kp2"][i], d["K"][i].reshape(3, 3), prm,
                                0.7, 10.0))
    pack = {k: d[k] for k in ("desc1", "kp1", "desc2", "kp2", "K", "global_index")}
    for i, r in enumerate(res):
        for k in ("matches", "mask", "points", "point_idx", "R", "t", "F", "E", "R1to2", "t1to2"):
            pack["%s_%d" % (k, i)] = r[k]
        pack["scalars_%d" % i] = np.array([r["ok"], r["n_matches"], r["n_inliers"], r["n_points"], r["best_hyp"],
                                           r["best_count"]], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "image_pair_small.npz"), H=256, seed=np.uint64(0x5EED0000),
                        max_error_sq=1e-2, **pack)

    # ---- reference test geometry (test/test-sfm.cpp, test/unit-test-helper.cpp)
    L = helpers.two_camera_rig("L", rpy=(1.5, 0.7, 0.0), scale=0.5)
    cube = helpers.two_camera_rig("cube")
    rl = o.sfm_solve(L["uv1"], L["uv2"], L["K"], o.make_params(1, o.SAMPLER_IDENTITY))
    tp, ti = o.sfm_triangulate(cube["uv1"], cube["uv2"], cube["K"], (np.eye(3), np.zeros(3)), cube["pose2in1"])
    np.savez_compressed(os.path.join(HERE, "rig_kat.npz"), L_uv1=L["uv1"], L_uv2=L["uv2"], L_X=L["X"], L_R=rl["R"],
                        L_t=rl["t"], L_points=rl["points"], L_F=rl["F"], cube_uv1=cube["uv1"], cube_uv2=cube["uv2"],
                        cube_X=cube["X"], cube_R12=cube["T1to2"][0], cube_t12=cube["T1to2"][1], cube_points=tp,
                        cube_idx=ti)
    # ---- pnp_solve (row f1): the reference's cube rig + a noisy scene with outliers
    import test_pnp as TP

    Kc, Xc, uvc = TP._cube_rig()
    rc = o.pnp_solve(Xc, uvc, Kc, o.make_pnp_params(100, o.SAMPLER_PHILOX, 0))
    Ks, Xs, uvs, Rs, ts, bad = TP._scene(77, 200, 0.01, 40)
    rs = o.pnp_solve(Xs, uvs, Ks, o.make_pnp_params(256, o.SAMPLER_PHILOX, 5))
    np.savez_compressed(os.path.join(HERE, "pnp_small.npz"), cube_X=Xc, cube_uv=uvc, cube_R=rc["R"], cube_t=rc["t"],
                        cube_inliers=rc["inliers"], K=Ks, X=Xs, uv=uvs, H=256, seed=np.uint64(5), R=rs["R"], t=rs["t"],
                        inliers=rs["inliers"], best_hyp=rs["best_hyp"],
                        samples=np.stack([o.sample4(5, h, 200) for h in range(256)]))
    refine_vectors()
    orb_vectors()
    tsukuba_frames()
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
