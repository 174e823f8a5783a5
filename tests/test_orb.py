"""Row f3 of SURVEY.md section 8: VisualFeature::extract (vision/visual-feature.cpp:40-49 = cv::ORB detect + compute).

cv::ORB and its learned pattern are OpenCV-internal: PARITY UNPINNED by the reference.  CPU part: every stage of the
oracle (oracle/mvs_orb_oracle.c) against an independent numpy restatement of its definition, plus behaviour (rotation
covariance of the steered descriptor, repeatability on the reference's tsukuba frames, the pose the reference's own
test-image-pair expects: (I, (1, 0, 0)) -- test/test-image-pair.cpp:38-45).  GPU part: the HIP kernels through the C ABI,
bit-exact against the oracle.
"""
import os
from fractions import Fraction

import numpy as np
import pytest

import oracle_lib as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
          (-3, 1), (-2, 2), (-1, 3)]
UMAX = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


def tsukuba():
    g = np.load(os.path.join(ROOT, "tests", "golden", "tsukuba_gray.npz"))
    return g["images"], g["K"]


def textured(seed, h, w):
    """a blocky random image with plenty of corners"""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(h // 6 + 2, w // 6 + 2)).astype(np.uint8)
    img = np.kron(base, np.ones((6, 6), dtype=np.uint8))[:h, :w].astype(np.int32)
    img += rng.integers(-6, 7, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def test_fast_score_matches_brute_force_definition():
    img = textured(3, 60, 70)
    got = o.orb_fast_scores(img, 20)
    I = img.astype(np.int32)
    want = np.zeros_like(img)
    for y in range(3, 57):
        for x in range(3, 67):
            ring = np.array([I[y + dy, x + dx] for dx, dy in CIRCLE]) - I[y, x]
            best = -1
            for t in range(254, -1, -1):   # the largest t at which 9 contiguous ring pixels are all brighter / darker by > t
                hi, lo = ring > t, ring < -t
                if any(all(hi[(k + j) % 16] for j in range(9)) or all(lo[(k + j) % 16] for j in range(9)) for k in range(16)):
                    best = t
                    break
            if best >= 20:
                want[y, x] = best
    assert (got > 0).sum() > 50
    assert np.array_equal(got, want)


def test_harris_moments_blur_resize_match_numpy():
    img = textured(5, 80, 96)
    I = img.astype(np.int64)
    # Harris: the 3x3 derivative stencil of cv::ORB's HarrisResponses over a 7x7 block
    for (x, y) in [(40, 40), (20, 55), (70, 30)]:
        a = b = c = 0
        for dy in range(-3, 4):
            for dx in range(-3, 4):
                yy, xx = y + dy, x + dx
                ix = (I[yy, xx + 1] - I[yy, xx - 1]) * 2 + (I[yy - 1, xx + 1] - I[yy - 1, xx - 1]) + (I[yy + 1, xx + 1] - I[yy + 1, xx - 1])
                iy = (I[yy + 1, xx] - I[yy - 1, xx]) * 2 + (I[yy + 1, xx - 1] - I[yy - 1, xx - 1]) + (I[yy + 1, xx + 1] - I[yy - 1, xx + 1])
                a, b, c = a + ix * ix, b + iy * iy, c + ix * iy
        s = np.float32(1.0) / np.float32(4 * 7 * 255)
        s4 = s * s * s * s
        fa, fb, fc = np.float32(a), np.float32(b), np.float32(c)
        want = ((fa * fb - fc * fc) - (np.float32(0.04) * (fa + fb)) * (fa + fb)) * s4
        assert o.orb_harris(img, x, y) == float(want)
        # intensity-centroid moments over the radius-15 disc
        m10 = sum(u * I[y + v, x + u] for v in range(-15, 16) for u in range(-UMAX[abs(v)], UMAX[abs(v)] + 1))
        m01 = sum(v * I[y + v, x + u] for v in range(-15, 16) for u in range(-UMAX[abs(v)], UMAX[abs(v)] + 1))
        assert o.orb_moments(img, x, y) == (m10, m01)
        ang = o.orb_fast_atan2(float(m01), float(m10))
        assert abs((ang - np.degrees(np.arctan2(m01, m10))) % 360.0) < 0.4 or abs((ang - np.degrees(np.arctan2(m01, m10))) % 360.0) > 359.6
    # blur: Q8 kernel, reflect-101
    k = np.array([18, 34, 49, 54, 49, 34, 18], dtype=np.int64)
    pad = np.pad(I, ((0, 0), (3, 3)), mode="reflect")
    rows = sum(k[j] * pad[:, j:j + I.shape[1]] for j in range(7))
    pad = np.pad(rows, ((3, 3), (0, 0)), mode="reflect")
    want = (sum(k[j] * pad[j:j + I.shape[0], :] for j in range(7)) + 32768) >> 16
    assert np.array_equal(o.orb_blur(img), want.astype(np.uint8))
    assert np.array_equal(o.orb_blur(np.full((20, 20), 77, np.uint8)), np.full((20, 20), 77, np.uint8))   # unit gain
    # resize: pixel-centre bilinear with 11-bit weights from exact rationals
    dw, dh = 80, 67
    got = o.orb_resize(img, dw, dh)
    sh, sw = img.shape
    for (dx, dy) in [(0, 0), (79, 66), (13, 40), (50, 7), (33, 33)]:
        fx = Fraction((2 * dx + 1) * sw - dw, 2 * dw)
        fy = Fraction((2 * dy + 1) * sh - dh, 2 * dh)
        sx, sy = int(fx // 1), int(fy // 1)
        wx, wy = int(((fx - sx) * 2048 + Fraction(1, 2)) // 1), int(((fy - sy) * 2048 + Fraction(1, 2)) // 1)
        x1, y1 = min(sx + 1, sw - 1), min(sy + 1, sh - 1)
        v = (I[sy, sx] * (2048 - wx) * (2048 - wy) + I[sy, x1] * wx * (2048 - wy) + I[y1, sx] * (2048 - wx) * wy
             + I[y1, x1] * wx * wy + (1 << 21)) >> 22
        assert got[dy, dx] == v
    assert np.array_equal(o.orb_resize(img, sw, sh), img)                                   # identity size
    assert np.abs(got.astype(int).mean() - img.astype(int).mean()) < 2.0


def test_layout_and_pattern():
    prm = o.make_orb_params()
    ok, lw, lh, nl, sc = o.orb_layout(640, 480, prm)
    assert ok and list(lw[:3]) == [640, 533, 444] and list(lh[:3]) == [480, 400, 333]
    assert nl.sum() == 500 and all(nl[i] >= nl[i + 1] for i in range(6))                   # cv::ORB's geometric quota
    assert np.allclose(sc, 1.2 ** np.arange(8))
    P = o.orb_pattern()
    assert P.shape == (256, 4) and P.min() >= -13 and P.max() <= 13
    assert not np.any((P[:, 0] == P[:, 2]) & (P[:, 1] == P[:, 3]))
    assert 4.5 < P.std() < 7.5 and abs(P.mean()) < 1.0                                     # ~ N(0, (31 / 5)^2), clipped
    assert len({tuple(r) for r in P}) > 250
    import hashlib
    assert hashlib.sha256(P.tobytes()).hexdigest()[:16] == PATTERN_SHA16                   # frozen: descriptors depend on it


PATTERN_SHA16 = None  # filled in below once, from the committed golden file


def _load_pattern_hash():
    global PATTERN_SHA16
    PATTERN_SHA16 = str(np.load(os.path.join(ROOT, "tests", "golden", "orb_small.npz"))["pattern_sha16"])


_load_pattern_hash()


def test_extract_order_quota_and_margins():
    imgs, _ = tsukuba()
    prm = o.make_orb_params()
    r = o.orb_extract(imgs[0], prm)
    kp = r["kp"]
    ok, lw, lh, nl, sc = o.orb_layout(imgs.shape[2], imgs.shape[1], prm)
    assert r["ok"] and 400 <= len(kp) <= 500
    for l in range(8):
        k = kp[kp["octave"] == l]
        assert len(k) <= nl[l]
        assert np.all(np.diff(k["response"]) <= 0)                                          # response descending per level
        x, y = k["x"] / np.float32(sc[l]), k["y"] / np.float32(sc[l])
        assert np.all(x >= 31 - 1e-3) and np.all(x <= lw[l] - 31) and np.all(y >= 31 - 1e-3) and np.all(y <= lh[l] - 31)
        assert np.allclose(k["size"], 31.0 * sc[l])
    assert np.all(np.diff(kp["octave"]) >= 0) and np.all(kp["class_id"] == -1)
    assert np.all((kp["angle"] >= 0) & (kp["angle"] < 360))
    again = o.orb_extract(imgs[0], prm)
    assert np.array_equal(again["desc"], r["desc"]) and np.array_equal(again["kp"], kp)


def test_descriptor_is_steered_rotation_by_90_degrees():
    """rotating the image by 90 degrees moves the keypoints with it and leaves the steered descriptors (almost) alone"""
    img = textured(9, 200, 200)
    prm = o.make_orb_params(nfeatures=300, nlevels=1)
    a = o.orb_extract(img, prm)
    b = o.orb_extract(np.ascontiguousarray(np.rot90(img)), prm)      # (x, y) -> (y, W - 1 - x)
    pa = {(int(k["x"]), int(k["y"])): i for i, k in enumerate(a["kp"])}
    hits, dist = 0, []
    for j, k in enumerate(b["kp"]):
        src = (199 - int(k["y"]), int(k["x"]))
        if src in pa:
            i = pa[src]
            hits += 1
            dist.append(int(np.unpackbits(a["desc"][i] ^ b["desc"][j]).sum()))
            d = (a["kp"]["angle"][i] - k["angle"] - 90.0) % 360.0
            assert min(d, 360.0 - d) < 1.0
    assert hits > 0.9 * len(a["kp"])
    assert np.median(dist) <= 8 and np.mean(dist) < 16                                       # of 256 bits; unrelated ~128


def test_tsukuba_pair_recovers_the_reference_pose():
    """test/test-image-pair.cpp:38-45: frames 1, 2 -> T_pair_to_base = (I, (1, 0, 0)) within 1e-3"""
    imgs, K = tsukuba()
    prm = o.make_orb_params()
    a, b = o.orb_extract(imgs[0], prm), o.orb_extract(imgs[1], prm)
    kp1 = np.stack([a["kp"]["x"], a["kp"]["y"]], 1).astype(np.float32)
    kp2 = np.stack([b["kp"]["x"], b["kp"]["y"]], 1).astype(np.float32)
    m = o.match_visual_features(a["desc"], b["desc"], 0.7, 50.0)
    assert len(m) > 100                                                                      # repeatability
    dxy = kp2[m["queryIdx"]] - kp1[m["trainIdx"]]
    assert np.mean(np.abs(dxy[:, 1]) < 1.5) > 0.9                                            # a rectified stereo pair
    r = o.image_pair(a["desc"], kp1, b["desc"], kp2, K, o.make_params(2000, o.SAMPLER_PHILOX, 1, 1e-3), 0.7, 50.0)
    assert r["valid"] and r["n_points"] > 60
    assert np.abs(r["t"] - np.array([1.0, 0, 0])).max() < 1e-3 and np.abs(r["R"] - np.eye(3)).max() < 1e-3


def test_golden_orb_vectors():
    g = np.load(os.path.join(ROOT, "tests", "golden", "orb_small.npz"))
    r = o.orb_extract(g["image"], o.make_orb_params(nfeatures=int(g["nfeatures"]), nlevels=int(g["nlevels"])))
    assert np.array_equal(r["kp"], g["kp"]) and np.array_equal(r["desc"], g["desc"])


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", ["tsukuba", "textured", "tiny", "flat"])
def test_gpu_extract_bit_exact(ctx, case):
    from mvslam_amd import capi

    if case == "tsukuba":
        imgs, _ = tsukuba()
        prm = dict(nfeatures=500)
    elif case == "textured":
        imgs = np.stack([textured(s, 480, 640) for s in (1, 2)])
        prm = dict(nfeatures=2000)
    elif case == "tiny":
        imgs = textured(4, 70, 90)[None]     # only level 0 is larger than twice the edge margin
        prm = dict(nfeatures=100)
    else:
        imgs = np.full((1, 120, 160), 90, np.uint8)
        prm = dict(nfeatures=50)
    got = ctx.extract(imgs, capi.default_orb_params(**prm))
    for i in range(len(imgs)):
        want = o.orb_extract(imgs[i], o.make_orb_params(**prm))
        n = int(got["n"][i])
        assert n == len(want["kp"])
        assert np.array_equal(got["kp"][i][:n], want["kp"].astype(capi.KEYPOINT_DTYPE))
        assert np.array_equal(got["desc"][i][:n], want["desc"])
    if case == "flat":
        assert got["n"][0] == 0
    if case == "textured":
        assert got["n"].min() > 1500


def test_small_feature_counts_never_exceed_nfeatures():
    """cv::ORB rounds every level's share of nfeatures; for small counts the shares add up to more than nfeatures
    (7 features over 8 levels: 2+1+1+1+1+1+1 = 8).  The outputs have room for nfeatures: a level takes what is left."""
    img = textured(3112, 317, 204)      # the soak case that found it (seed 2309)
    for nf in range(1, 26):
        r = o.orb_extract(img, o.make_orb_params(nfeatures=nf, nlevels=8, fast_threshold=60, edge_threshold=19))
        assert len(r["kp"]) <= nf


@pytest.mark.gpu
def test_gpu_extract_small_feature_counts(ctx):
    """the same clamp on the device side: two images per call, so that an overrun of the first image's rows would land
    in the second image's"""
    from mvslam_amd import capi

    imgs = np.stack([textured(3112, 317, 204), textured(3113, 317, 204)])
    for nf in (1, 2, 3, 5, 7, 8, 11, 16, 23):
        prm = dict(nfeatures=nf, nlevels=8, fast_threshold=60, edge_threshold=19)
        got = ctx.extract(imgs, capi.default_orb_params(**prm))
        for i in range(2):
            want = o.orb_extract(imgs[i], o.make_orb_params(**prm))
            n = int(got["n"][i])
            assert n == len(want["kp"]) <= nf
            assert np.array_equal(got["kp"][i][:n], want["kp"].astype(capi.KEYPOINT_DTYPE))
            assert np.array_equal(got["desc"][i][:n], want["desc"])


@pytest.mark.gpu
def test_gpu_extract_against_golden_and_argument_errors(ctx):
    from mvslam_amd import capi

    g = np.load(os.path.join(ROOT, "tests", "golden", "orb_small.npz"))
    got = ctx.extract(g["image"], capi.default_orb_params(nfeatures=int(g["nfeatures"]), nlevels=int(g["nlevels"])))
    n = int(got["n"][0])
    assert n == len(g["kp"]) and np.array_equal(got["kp"][0][:n], g["kp"].astype(capi.KEYPOINT_DTYPE))
    assert np.array_equal(got["desc"][0][:n], g["desc"])
    with pytest.raises(capi.MvsError) as e:
        ctx.extract(g["image"], capi.default_orb_params(fast_threshold=0))
    assert e.value.status == capi.MVS_ERR_INVALID_ARG
    with pytest.raises(capi.MvsError) as e:
        ctx.extract(g["image"], capi.default_orb_params(nlevels=40))
    assert e.value.status == capi.MVS_ERR_INVALID_ARG
    # round 5: a level's candidate list holds the non-maximum suppression's own bound, so a frame with far more than the
    # 16384 corners per level that rounds 2-4 could take (~200 000 here) is simply extracted -- and equals the oracle
    noise = np.random.default_rng(0).integers(0, 256, size=(1, 1100, 1400), dtype=np.uint8)
    prm = dict(nfeatures=100, nlevels=1, fast_threshold=5)
    big = ctx.extract(noise, capi.default_orb_params(**prm))
    want = o.orb_extract(noise[0], o.make_orb_params(**prm))
    nb = int(big["n"][0])
    assert nb == len(want["kp"]) == 100
    assert np.array_equal(big["kp"][0][:nb], want["kp"].astype(capi.KEYPOINT_DTYPE)) and np.array_equal(big["desc"][0][:nb], want["desc"])
    # only when 2 n_l exceeds what the selection holds in LDS (16384 keys) the list is capped there, and a fuller level is
    # reported, never silently truncated
    with pytest.raises(capi.MvsError) as e:
        ctx.extract(noise, capi.default_orb_params(nfeatures=20000, nlevels=1, fast_threshold=5))
    assert e.value.status == capi.MVS_ERR_CAPACITY
    # ... and the context is still usable afterwards
    again = ctx.extract(g["image"], capi.default_orb_params(nfeatures=int(g["nfeatures"]), nlevels=int(g["nlevels"])))
    assert int(again["n"][0]) == n


@pytest.mark.gpu
@pytest.mark.parametrize("shape,prm", [
    ((257, 333), dict(nfeatures=3000, nlevels=8, fast_threshold=7, edge_threshold=19)),
    ((203, 415), dict(nfeatures=700, nlevels=6, fast_threshold=20, edge_threshold=31)),
    ((480, 640), dict(nfeatures=1200, nlevels=8, fast_threshold=40, edge_threshold=25)),
])
def test_gpu_extract_batch_shapes_and_ties(ctx, shape, prm):
    """Round 5 kernels: eleven images per call (the block orders hand image b to XCD b mod 8: a count that is not a
    multiple of eight leaves blocks without an image), widths that are not multiples of four (unaligned dword windows
    in FAST / blur / resize / describe, dword stores that straddle rows), levels on either side of twice the edge
    margin, blur tiles on the border and inside, and images whose corners tie in FAST score by the thousand -- the
    radix select of retainBest(2 n_l) then has to walk the y and x bytes of the keys."""
    from mvslam_amd import capi

    h, w = shape
    rng = np.random.default_rng(h * 1000 + w)
    imgs = [textured(900 + s, h, w) for s in range(6)]
    yy, xx = np.mgrid[0:h, 0:w]
    c = 5 if h < 300 else 9     # (a level holds at most 16384 corners: MVS_ERR_CAPACITY, tested elsewhere)
    imgs.append((((yy // c + xx // c) & 1) * 255).astype(np.uint8))                    # checkerboard: every corner ties
    imgs.append((((yy // 9 + xx // 7) & 1) * 200 + 20).astype(np.uint8))
    imgs.append(np.full((h, w), 77, np.uint8))                                         # nothing to find
    imgs.append(rng.integers(0, 256 if h < 300 else 160, size=(h, w)).astype(np.uint8))   # noise: ~11 000 corners at level 0
    c = 3 if h < 300 else 8
    imgs.append(np.kron(rng.integers(0, 2, size=(h // c + 1, w // c + 1)).astype(np.uint8) * 255,
                        np.ones((c, c), np.uint8))[:h, :w])                            # binary blocks
    imgs = np.stack(imgs)
    assert len(imgs) == 11
    got = ctx.extract(imgs, capi.default_orb_params(**prm))
    total = 0
    for i in range(len(imgs)):
        want = o.orb_extract(imgs[i], o.make_orb_params(**prm))
        n = int(got["n"][i])
        assert n == len(want["kp"]), i
        assert np.array_equal(got["kp"][i][:n], want["kp"].astype(capi.KEYPOINT_DTYPE)), i
        assert np.array_equal(got["desc"][i][:n], want["desc"]), i
        total += n
    assert got["n"][8] == 0 and total > 3 * prm["nfeatures"]


# (hypotheses, sampler, max_error_sq [0 = the reference formula 5e-2 / K00 / K11, sfm-solve.cpp:311], max_dist)
TSUKUBA_PARAMS = {
    "build": (2000, 1, 1e-3, 50.0),                 # the build's sampler, a wide matcher gate
    "reference-defaults": (2000, 1, 0.0, 10.0),     # ImagePair::get_default_params (image-pair.cpp:22-23) + sfm-solve.cpp:311
    "reference-as-shipped": (1, 0, 0.0, 10.0),      # ... with the reference's single identity-sample iteration (sfm-solve.cpp:67)
}


@pytest.mark.gpu
@pytest.mark.parametrize("which", list(TSUKUBA_PARAMS))
def test_gpu_images_to_poses_tsukuba_sequence(ctx, which):
    """BASELINE configs[0] ('plumbing'): the reference's tsukuba frames through extraction -> matching -> two-view
    RANSAC -> refinement -> PnP, everything on the device; frames 1, 2 must give (I, (1, 0, 0)) within 1e-3 AFTER
    refinement, as test/test-image-pair.cpp:13,38-45 expects -- with the reference's default parameters too."""
    from mvslam_amd import capi

    H, sampler, thr, max_dist = TSUKUBA_PARAMS[which]
    imgs, K = tsukuba()
    F, N = len(imgs), 512
    s = capi.Sequence(ctx, F, N, 32)
    s.upload_images(0, imgs, K, capi.default_orb_params())
    prm = capi.default_params(num_hypotheses=H, sampler=sampler, seed=1, max_error_sq=thr, max_dist=max_dist)
    pprm = capi.default_pnp_params(num_hypotheses=100, seed=7, reproj_error=1.0)
    s.run(prm, pprm)      # run + sync
    pairs = s.download_pairs()
    tracks = s.download_tracks()
    s.refine_pairs(sigma_px=0.5)      # ImagePair::refine of every pair, as the reference's test-image-pair does
    refined = s.download_refined()["refined"]
    s.close()
    assert np.all(refined["ok"] == 1)
    for p in range(F - 1):            # test-image-pair.cpp:13,38-45 asserts this AFTER refinement, at 1e-3
        assert np.abs(refined["t"][p] - np.array([1.0, 0, 0])).max() < 1e-3, (p, refined["t"][p])
        assert np.abs(refined["R"][p] - np.eye(3)).max() < 1e-3
    res = pairs["results"]
    assert np.all(res["valid"] == 1)
    # frames are 1 px apart horizontally in a rectified rig: every consecutive pair is a pure x translation
    for p in range(F - 1):
        assert np.abs(res["t"][p] - np.array([1.0, 0, 0])).max() < 1e-3, (p, res["t"][p])
        assert np.abs(res["R"][p] - np.eye(3)).max() < 1e-3
    # the same pairs through the oracle, from the oracle's own extraction: bit-exact inputs give bit-exact results
    ex = [o.orb_extract(im, o.make_orb_params(nfeatures=N)) for im in imgs]
    for p in range(F - 1):
        a, b = ex[p], ex[p + 1]
        kp1 = np.stack([a["kp"]["x"], a["kp"]["y"]], 1).astype(np.float32)
        kp2 = np.stack([b["kp"]["x"], b["kp"]["y"]], 1).astype(np.float32)
        want = o.image_pair(a["desc"], kp1, b["desc"], kp2, K, o.make_params(H, sampler, 1 + p, thr), 0.7, max_dist)
        assert want["valid"] and want["n_matches"] == res["n_matches"][p] and want["n_points"] == res["n_points"][p]
        assert want["best_hyp"] == res["best_hyp"][p]
    if which == "build":
        assert np.all(tracks["tracks"]["ok"] == 1)


@pytest.mark.gpu
def test_gpu_tsukuba_visual_odometer_fixture(ctx):
    """The reference's own VO fixture (test/test-visual-odometer.cpp:60-107): the five tsukuba frames, pose of frame i
    = (I, (i, 0, 0)) within i * 1e-3 (check_similar_SE3: component-wise on the se3 logarithm).  Pins row f2's
    scale-propagation fold (visual-odometer.cpp:577-588) with reference-held data, on the device and in the oracle."""
    from mvslam_amd import capi

    imgs, K = tsukuba()
    F, N = len(imgs), 512
    s = capi.Sequence(ctx, F, N, 32)
    s.upload_images(0, imgs, K, capi.default_orb_params())
    prm = capi.default_params(num_hypotheses=2000, sampler=capi.SAMPLER_PHILOX, seed=1, max_error_sq=1e-3, max_dist=50.0)
    # pnp_solve with the reference's own values: 100 iterations, reprojectionError 0.05 px (pnp-solve.cpp:47-49) -- at
    # that gate only the (exact, integer-disparity) level-0 keypoints of this rendered sequence are inliers
    pprm = capi.default_pnp_params(num_hypotheses=100, seed=7, reproj_error=0.05)
    s.run(prm, pprm)
    gp, gt, tr = s.download_pairs(), s.download_tracks(), s.download_trajectory()
    s.close()
    res, trk = gp["results"], gt["tracks"]
    assert np.all(res["valid"] == 1) and np.all(trk["ok"] == 1)
    want = o.seq_chain(res["R"], res["t"], res["valid"], trk["R"], trk["t"], trk["ok"])      # the oracle's fold
    for k in ("R", "t", "pair_scale", "track_scale"):
        assert tr[k].tobytes() == want[k].tobytes(), k
    for traj in (tr, want):
        for i in range(1, F):
            xi = o.se3_ln(traj["R"][i], traj["t"][i])
            assert np.abs(xi - np.array([i, 0, 0, 0, 0, 0.0])).max() <= i * 1e-3, (i, xi)
