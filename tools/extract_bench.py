#!/usr/bin/env python3
"""Row f3: VisualFeature::extract for a batch of 640x480 frames (2000 features each, the keypoint count BASELINE's
matching configs assume).  Times upload + extraction + download through the C ABI by wall clock (`mvs_extract` is a
host-buffer entry point), kernel time comes from rocprofv3.  The CPU oracle extracts a sample of the same frames.
Prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mvslam_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=64)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--features", type=int, default=2000)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--cpu-images", type=int, default=16)
ap.add_argument("--resident", action="store_true", help="also time extraction straight into a resident sequence (no D2H)")
args = ap.parse_args()


def textured(seed, h, w):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(h // 6 + 2, w // 6 + 2)).astype(np.uint8)
    img = np.kron(base, np.ones((6, 6), dtype=np.uint8))[:h, :w].astype(np.int32)
    img += rng.integers(-6, 7, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


imgs = np.stack([textured(100 + i, args.height, args.width) for i in range(args.images)])
ctx = capi.Context(0)
prm = capi.default_orb_params(nfeatures=args.features)
out = ctx.extract(imgs, prm)
t0 = time.perf_counter()
for _ in range(args.steps):
    out = ctx.extract(imgs, prm)
ms = (time.perf_counter() - t0) * 1e3 / args.steps
# the same through pinned host buffers (mvs_host_alloc) reused across calls: the copies are DMA transfers and no fresh
# output pages are touched per call
pin_img = capi.pinned_empty(imgs.shape, np.uint8)
pin_img[...] = imgs
pin_out = dict(kp=capi.pinned_empty((args.images, args.features), capi.KEYPOINT_DTYPE),
               desc=capi.pinned_empty((args.images, args.features, 32), np.uint8), n=capi.pinned_empty((args.images,), np.int32))
ctx.extract(pin_img, prm, out=pin_out)
t0 = time.perf_counter()
for _ in range(args.steps):
    ctx.extract(pin_img, prm, out=pin_out)
pinned_ms = (time.perf_counter() - t0) * 1e3 / args.steps
assert np.array_equal(pin_out["desc"], out["desc"]) and np.array_equal(pin_out["n"], out["n"])
resident_ms = None
if args.resident:
    seq = capi.Sequence(ctx, args.images, args.features, 32)
    K = np.array([[525.0, 0, 320], [0, 525, 240], [0, 0, 1]])
    seq.upload_images(0, imgs, K, prm)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        seq.upload_images(0, imgs, None, prm)
    resident_ms = (time.perf_counter() - t0) * 1e3 / args.steps
    seq.close()
ctx.close()
# algorithmic bytes per image (u8 pixels, pyramid = 3.16 x level 0): resize read+write, FAST read + score write, NMS read,
# blur read + u16 write/read + write, descriptors: ~ 9 pyramid passes
pyr = sum(round(args.width / 1.2 ** l) * round(args.height / 1.2 ** l) for l in range(8))
alg_bytes = 9.0 * pyr
cpu = None
if args.cpu_images > 0:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle_lib as o
    t0 = time.perf_counter()
    same = True
    for i in range(args.cpu_images):
        w = o.orb_extract(imgs[i], o.make_orb_params(nfeatures=args.features))
        n = int(out["n"][i])
        same &= n == len(w["kp"]) and np.array_equal(out["desc"][i][:n], w["desc"])
    dt = time.perf_counter() - t0
    cpu = {"value": round(args.cpu_images / dt, 2), "unit": "images/s", "cores": 1, "kind": "port",
           "sample": "%d of the same frames, single thread" % args.cpu_images, "bit_exact_vs_gpu": bool(same)}
print(json.dumps({
    "metric": "extracted images/sec (VisualFeature::extract, %dx%d, %d features, host buffers in and out)"
              % (args.width, args.height, args.features),
    "value": round(args.images / (ms * 1e-3), 1), "unit": "images/s", "ms_per_batch": round(ms, 3), "images": args.images,
    "mean_keypoints": float(out["n"].mean()), "algorithmic_MB_per_image": round(alg_bytes / 1e6, 2),
    "pinned_ms_per_batch": round(pinned_ms, 3), "pinned_images_per_s": round(args.images / (pinned_ms * 1e-3), 1),
    "resident_ms_per_batch": None if resident_ms is None else round(resident_ms, 3),
    "graph": os.environ.get("MVS_NO_GRAPH") is None, "cpu_baseline": cpu}))
