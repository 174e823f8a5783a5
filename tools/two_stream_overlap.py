#!/usr/bin/env python3
"""Do the RANSAC kernels of two batches on two streams overlap on the chip?  Two contexts (two streams), half the pairs
each, run() enqueued alternately; throughput against one batch of all the pairs on one stream.
usage: python tools/two_stream_overlap.py [--pairs 512] [--steps 6]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvslam_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=512)
ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
prm = capi.default_params(num_hypotheses=50000, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=1e-2)


def make(ctx, first, n):
    d = synth.make_batch(first, n, n_kp=2000)
    b = capi.Batch(ctx, n, 2000, 32)
    b.upload(0, d["desc1"], d["kp1"], d["n1"], d["desc2"], d["kp2"], d["n2"], d["K"], d["global_index"])
    return b


c0, c1 = capi.Context(0), capi.Context(0)
one = make(c0, 0, a.pairs)
one.run(prm); one.sync()
t0 = time.perf_counter()
for _ in range(a.steps):
    one.run(prm)
one.sync()
t_one = (time.perf_counter() - t0) / a.steps
ref = one.download(matches=False, mask=False, points=False)["results"].tobytes()
one.close()
h = a.pairs // 2
A, B = make(c0, 0, h), make(c1, h, h)
A.run(prm); B.run(prm); A.sync(); B.sync()
t0 = time.perf_counter()
for _ in range(a.steps):
    A.run(prm)
    B.run(prm)
A.sync(); B.sync()
t_two = (time.perf_counter() - t0) / a.steps
got = A.download(matches=False, mask=False, points=False)["results"].tobytes() + B.download(matches=False, mask=False, points=False)["results"].tobytes()
print(json.dumps(dict(pairs=a.pairs, one_stream_ms=round(t_one * 1e3, 3), two_streams_ms=round(t_two * 1e3, 3), same_results=got == ref,
                      count_threads=os.environ.get("MVS_CNT_THREADS", "768"))))
