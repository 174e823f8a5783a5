#!/usr/bin/env python3
"""Guard of the pre-screen's probe: for every cell of bench.py's sensitivity grid, the stage as the probe
decides it against the same stage with every pair forced exact (round 2's exact-everything path) and forced pre-screened.
Diagnostics build (the mode switch exists there only).  No cell may be slower probe-decided than forced exact.

    MVS_USE_DEBUG_LIB=1 python tools/sensitivity_guard.py > profiles/r04_sensitivity_guard.json"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MVS_USE_DEBUG_LIB"] = "1"
from mvslam_amd import capi, synth  # noqa: E402


def main():
    pairs, kp, hyp = 64, 2000, 50000
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    ctx = capi.Context(0)
    lib = capi.lib()
    batch = capi.Batch(ctx, pairs, kp, 32)
    cells = []
    for outl in ((0.3, 0.9) if quick else (0.3, 0.5, 0.7, 0.9)):
        for noise in (0.5, 2.0):
            data = synth.make_batch(9000, pairs, n_kp=kp, noise_px=noise, outlier_frac=outl)
            batch.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"],
                         data["global_index"])
            for thr in (1e-2, 1e-3, 5e-4, 2e-4, 1e-4, 1e-5):
                prm = capi.default_params(sampler=capi.SAMPLER_PHILOX, min_inliers=8, ratio=0.7, max_dist=10.0, max_error_sq=thr,
                                          num_hypotheses=hyp, seed=synth.SEED_BASE)
                row = {"outlier_frac": outl, "noise_px": noise, "max_error_sq": thr}
                ref = None
                for name, force in (("probe", -1), ("forced_exact", 0), ("forced_prescreened", 1), ("forced_mode2", 2)):
                    lib.mvs_debug_set_prescreen_force(C.c_int(force))
                    tot, _ = batch.time(prm, steps=3, warmup=1, per_kernel=False)
                    row[name + "_ms"] = round(tot / 3, 3)
                    res = batch.download(matches=False, mask=False, points=False)["results"]
                    if ref is None:
                        ref = res.tobytes()
                    row[name + "_same_results"] = res.tobytes() == ref
                lib.mvs_debug_set_prescreen_force(C.c_int(-1))
                row["pairs_mode"] = batch.stats(prm)["pairs_mode"]
                row["probe_over_forced_exact"] = round(row["probe_ms"] / row["forced_exact_ms"], 3)
                row["best_forced_over_probe"] = round(min(row["forced_exact_ms"], row["forced_prescreened_ms"], row["forced_mode2_ms"]) / row["probe_ms"], 3)
                cells.append(row)
    batch.close()
    ctx.close()
    worst = max(cells, key=lambda c: c["probe_over_forced_exact"])
    ok = all(c["probe_over_forced_exact"] <= 1.05 and c["probe_same_results"] and c["forced_exact_same_results"] and
             c["forced_prescreened_same_results"] and c["forced_mode2_same_results"] for c in cells)
    print(json.dumps({"ok": ok, "worst_probe_over_forced_exact": worst["probe_over_forced_exact"], "worst_cell": worst,
                      "cells": cells}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
