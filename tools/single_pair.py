#!/usr/bin/env python3
"""Per-kernel timing of ONE pair at a time (BASELINE configs[1]): the fused hypothesis-per-lane kernel (what one or two pairs
run on) against the pre-screened stage forced onto the same launch (diagnostics build).
MVS_USE_DEBUG_LIB=1 python tools/single_pair.py"""
import ctypes as C
import json
import os
import sys

os.environ["MVS_USE_DEBUG_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvslam_amd import capi, synth  # noqa: E402

ctx = capi.Context(0)
lib = capi.lib()
out = []
for n in (1, 2, 8):
    d = synth.make_batch(0, n, n_kp=2000)
    b = capi.Batch(ctx, n, 2000, 32)
    b.upload(0, d["desc1"], d["kp1"], d["n1"], d["desc2"], d["kp2"], d["n2"], d["K"], d["global_index"])
    prm = capi.default_params(num_hypotheses=50000, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=1e-2)
    ref = None
    for name, split in (("fused", 100), ("prescreened", 1)):   # (100: no launch of this tool reaches it -> the fused kernel)
        lib.mvs_debug_set_split_min_pairs(C.c_int(split))
        b.run(prm)
        b.sync()
        res = b.download(matches=False, mask=False, points=False)["results"].tobytes()
        ref = ref or res
        tot, _ = b.time(prm, steps=30, warmup=5, per_kernel=False)
        kern = {}
        for nm, ms in b.time_kernels(prm, steps=10):
            kern[nm] = round(kern.get(nm, 0) + ms, 4)
        out.append(dict(pairs=n, path=name if n < 3 else name + " (default for 3+ pairs: prescreened)", ms_per_step=round(tot / 30, 4),
                        same_results=res == ref, kernels=kern))
    lib.mvs_debug_set_split_min_pairs(C.c_int(3))
    b.close()
ctx.close()
print(json.dumps(out))
