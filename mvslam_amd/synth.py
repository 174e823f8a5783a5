"""Synthetic image pairs for the two-view path.

One pair = two 640x480 views of random 3-D points: N keypoints per image with 256-bit
descriptors.  Deterministic per pair: numpy Generator(PCG64) seeded with 0x5EED0000 + pair_index.
This is input data generation only (host side, numpy); nothing here is on the measured path.
"""
import numpy as np

K_DEFAULT = np.array([[525.0, 0.0, 320.0], [0.0, 525.0, 240.0], [0.0, 0.0, 1.0]])
SEED_BASE = 0x5EED0000


def _rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def make_pair(pair_index, n_kp=2000, noise_px=0.5, outlier_frac=0.3, flip_p=0.02, common_frac=0.8, width=640,
              height=480, desc_bytes=32, K=K_DEFAULT, baseline=0.3):
    """Returns dict(desc1, kp1, desc2, kp2, K, R_1to2, t_1to2, n_common).

    image 1 = base frame = train (vf1); image 2 = pair frame = query (vf2).
    """
    rng = np.random.default_rng(SEED_BASE + int(pair_index))
    n_common = int(common_frac * n_kp)
    # camera 2 relative to camera 1: x2 = R x1 + t
    w = rng.normal(size=3)
    w *= rng.uniform(0.0, 0.1) / np.linalg.norm(w)
    R = _rodrigues(w)
    while True:
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        if abs(d[2]) <= 0.1:
            break
    t = baseline * d
    Kinv = np.linalg.inv(K)
    # world points: uniform pixel in image 1, depth U[2, 10]; keep those that land inside image 2
    m = 3 * n_kp
    uv = np.stack([rng.uniform(0, width, m), rng.uniform(0, height, m)], axis=1)
    depth = rng.uniform(2.0, 10.0, m)
    X = (Kinv @ np.concatenate([uv, np.ones((m, 1))], axis=1).T).T * depth[:, None]
    X2 = (R @ X.T).T + t
    p2 = (K @ X2.T).T
    uv2 = p2[:, :2] / p2[:, 2:3]
    n1 = uv + rng.normal(scale=noise_px, size=uv.shape) if noise_px > 0 else uv.copy()
    n2 = uv2 + rng.normal(scale=noise_px, size=uv2.shape) if noise_px > 0 else uv2.copy()
    ok = (X2[:, 2] > 0.1)
    for a in (n1, n2):
        ok &= (a[:, 0] >= 0) & (a[:, 0] < width) & (a[:, 1] >= 0) & (a[:, 1] < height)
    sel = np.nonzero(ok)[0][:n_common]
    n_common = len(sel)
    kp1 = np.empty((n_kp, 2), dtype=np.float32)
    kp2 = np.empty((n_kp, 2), dtype=np.float32)
    kp1[:n_common] = n1[sel]
    kp2[:n_common] = n2[sel]
    n_pad = n_kp - n_common
    kp1[n_common:] = np.stack([rng.uniform(0, width, n_pad), rng.uniform(0, height, n_pad)], axis=1)
    kp2[n_common:] = np.stack([rng.uniform(0, width, n_pad), rng.uniform(0, height, n_pad)], axis=1)
    # descriptors
    nbits = desc_bytes * 8
    bits1 = rng.integers(0, 2, size=(n_kp, nbits), dtype=np.uint8)
    bits2 = rng.integers(0, 2, size=(n_kp, nbits), dtype=np.uint8)
    partner = np.arange(n_common)
    n_out = int(outlier_frac * n_common)
    if n_out > 0:
        bad = rng.choice(n_common, size=n_out, replace=False)
        shift = rng.integers(1, n_common, size=n_out)
        partner[bad] = (bad + shift) % n_common  # a different keypoint's descriptor: geometric outlier
    flips = (rng.random((n_common, nbits)) < flip_p).astype(np.uint8)
    bits2[:n_common] = bits1[partner] ^ flips
    # image-2 keypoints in random order so that indices carry no information
    perm = rng.permutation(n_kp)
    kp2 = kp2[perm]
    bits2 = bits2[perm]
    return dict(desc1=np.packbits(bits1, axis=1, bitorder="little"), kp1=kp1,
                desc2=np.packbits(bits2, axis=1, bitorder="little"), kp2=kp2, K=K.copy(), R_1to2=R, t_1to2=t,
                n_common=n_common)


def make_batch(first, count, n_kp=2000, **kw):
    """Stacked arrays for pairs [first, first + count) in the layout mvs_batch_upload expects."""
    desc_bytes = kw.get("desc_bytes", 32)
    out = dict(
        desc1=np.empty((count, n_kp, desc_bytes), dtype=np.uint8), kp1=np.empty((count, n_kp, 2), dtype=np.float32),
        desc2=np.empty((count, n_kp, desc_bytes), dtype=np.uint8), kp2=np.empty((count, n_kp, 2), dtype=np.float32),
        n1=np.full(count, n_kp, dtype=np.int32), n2=np.full(count, n_kp, dtype=np.int32),
        K=np.empty((count, 9)), global_index=np.arange(first, first + count, dtype=np.int64))
    for i in range(count):
        p = make_pair(first + i, n_kp=n_kp, **kw)
        out["desc1"][i], out["kp1"][i], out["desc2"][i], out["kp2"][i] = p["desc1"], p["kp1"], p["desc2"], p["kp2"]
        out["K"][i] = p["K"].reshape(9)
    return out


def make_sequence(n_frames, n_kp=2000, n_map=20000, noise_px=0.5, flip_p=0.02, seed=0x5E9, width=640, height=480,
                  desc_bytes=32, K=K_DEFAULT, step=0.05, yaw_step=0.005):
    """BASELINE configs[4]: a camera moving `step` m per frame (forward-lateral, small yaw) over a persistent map of
    `n_map` points; every frame keeps up to 0.8 * n_kp visible map points (noisy projections, the point's descriptor
    with per-frame bit flips) and pads to n_kp with clutter.  Returns desc [F, n_kp, B], kp [F, n_kp, 2] float32,
    n_kp [F], K, and the ground-truth camera poses (R_w2c, t_w2c) per frame."""
    rng = np.random.default_rng(seed)
    nbits = desc_bytes * 8
    # the map lies in a corridor in front of the trajectory
    L = step * n_frames
    Xw = np.stack([rng.uniform(-6, 6 + 0.5 * L, n_map), rng.uniform(-3, 3, n_map), rng.uniform(2, 12 + L, n_map)], axis=1)
    map_bits = rng.integers(0, 2, size=(n_map, nbits), dtype=np.uint8)
    desc = np.empty((n_frames, n_kp, desc_bytes), dtype=np.uint8)
    kp = np.empty((n_frames, n_kp, 2), dtype=np.float32)
    poses = []
    n_vis_max = int(0.8 * n_kp)
    for f in range(n_frames):
        yaw = yaw_step * f
        R = np.array([[np.cos(yaw), 0, -np.sin(yaw)], [0, 1, 0], [np.sin(yaw), 0, np.cos(yaw)]])   # world -> camera
        c = np.array([0.5 * step * f, 0.0, step * f * 0.866])                                      # camera centre
        t = -R @ c
        Xc = (R @ Xw.T).T + t
        z = Xc[:, 2]
        uvp = (K @ Xc.T).T
        uv = uvp[:, :2] / np.where(z[:, None] > 0.1, uvp[:, 2:3], 1.0)
        vis = (z > 1.0) & (uv[:, 0] >= 1) & (uv[:, 0] < width - 1) & (uv[:, 1] >= 1) & (uv[:, 1] < height - 1)
        ids = np.nonzero(vis)[0]
        if len(ids) > n_vis_max:
            ids = ids[np.argsort((ids * 2654435761) % 1000003)[:n_vis_max]]     # a stable pseudo-random subset
        nv = len(ids)
        pts = uv[ids] + rng.normal(scale=noise_px, size=(nv, 2)) if noise_px > 0 else uv[ids]
        bits = map_bits[ids] ^ (rng.random((nv, nbits)) < flip_p).astype(np.uint8)
        fk = np.empty((n_kp, 2))
        fb = rng.integers(0, 2, size=(n_kp, nbits), dtype=np.uint8)
        fk[:nv] = np.clip(pts, 0, [width - 1e-3, height - 1e-3])
        fb[:nv] = bits
        fk[nv:] = np.stack([rng.uniform(0, width, n_kp - nv), rng.uniform(0, height, n_kp - nv)], axis=1)
        perm = rng.permutation(n_kp)
        kp[f] = fk[perm].astype(np.float32)
        desc[f] = np.packbits(fb[perm], axis=1, bitorder="little")
        poses.append((R, t))
    return dict(desc=desc, kp=kp, n_kp=np.full(n_frames, n_kp, dtype=np.int32), K=K.copy(), poses=poses)
