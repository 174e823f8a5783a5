#!/usr/bin/env python3
"""Where does the transfer-inclusive leg of bench.py lose its time?  Two contexts / batches alternate (as in bench.py); variants:
run only, upload + run, run + download, all three; plus the raw H2D / D2H rates of the box.  usage: python tools/pcie_probe.py"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvslam_amd import capi, synth

import torch
P, N = 512, 2000
data = synth.make_batch(0, P, n_kp=N)
streams = [torch.cuda.Stream(device=0), torch.cuda.Stream(device=0)]
prm = capi.default_params(num_hypotheses=50000, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=1e-2)
lanes = []
for _i in range(2):
    ctx = capi.Context(0, stream=streams[_i].cuda_stream)
    b = capi.Batch(ctx, P, N, 32)
    pin = {k: capi.pinned_empty(np.asarray(data[k]).shape, np.asarray(data[k]).dtype) for k in ("desc1", "kp1", "n1", "desc2", "kp2", "n2", "global_index")}
    pin["K"] = capi.pinned_empty((P, 9), np.float64)
    for k in pin:
        pin[k][...] = np.asarray(data[k]).reshape(pin[k].shape)
    out = (capi.pinned_empty((P,), capi.RESULT_DTYPE), capi.pinned_empty((P, N), capi.MATCH_DTYPE), capi.pinned_empty((P, N), np.uint8),
           capi.pinned_empty((P, N, 3), np.float64), capi.pinned_empty((P, N), np.int32))
    b.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"], data["global_index"])
    lanes.append((b, pin, out))


def loop(up, down, reps=10, one_lane=False, chain=False):
    run_done = [None, None]

    def submit(k):
        b, pin, out = lanes[k]
        b.sync()
        if up:
            b.upload_async(0, pin["desc1"], pin["kp1"], pin["n1"], pin["desc2"], pin["kp2"], pin["n2"], pin["K"], pin["global_index"])
        if chain and run_done[k ^ 1] is not None:
            streams[k].wait_event(run_done[k ^ 1])      # the lanes' runs alternate, the copies float
        b.run(prm)
        if chain:
            run_done[k] = torch.cuda.Event()
            run_done[k].record(streams[k])
        if down:
            b.download_async(0, P, *out)
    use = [0] if one_lane else [0, 1]
    for k in range(2):
        submit(use[k % len(use)])
    for l in use:
        lanes[l][0].sync()
    t0 = time.perf_counter()
    for k in range(reps):
        submit(use[k % len(use)])
    for l in use:
        lanes[l][0].sync()
    return (time.perf_counter() - t0) / reps * 1e3


CASES = (("run only, two lanes", False, False, False, False), ("run only, one lane", False, False, True, False),
         ("upload + run, two lanes", True, False, False, False), ("run + download, two lanes", False, True, False, False),
         ("upload + run + download, two lanes", True, True, False, False), ("upload + run + download, one lane", True, True, True, False),
         ("upload + run + download, two lanes, runs CHAINED by events", True, True, False, True),
         ("upload + run, two lanes, runs CHAINED", True, False, False, True), ("run only, two lanes, runs CHAINED", False, False, False, True))
if len(sys.argv) > 1:      # one case only (for a trace): python tools/pcie_probe.py 2
    CASES = CASES[int(sys.argv[1]):int(sys.argv[1]) + 1]
for name, up, down, one, chain in CASES:
    ms = loop(up, down, one_lane=one, chain=chain)
    print("%-62s %.3f ms per step = %.0f pairs/s" % (name, ms, P / ms * 1e3))
