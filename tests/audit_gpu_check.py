#!/usr/bin/env python3
"""Full-population device audit of the pre-screened RANSAC stage (VERDICT r3 #1b; run by tests/test_prescreen.py in its own
process with MVS_USE_DEBUG_LIB=1: the audit kernel exists in the diagnostics library only -- same kernels.hip, same
launch path as the product library, plus the hooks).

The reference's rule (estimator-RANSAC.cpp:76-84) decides over ALL hypotheses of a pair; the pre-screen decides which of
them are ever solved exactly.  Here every hypothesis of BASELINE configs[2] (512 pairs x 50 000 = 25.6 M) and of the first
128 pairs of configs[4]'s sequence is solved exactly once more ON THE DEVICE and scored exactly on every match, and
  phase 0 (records as the pre-screen wrote them): state byte 0 only for samples the exact path rejects; for every certified
          record and every match |r_i(F_J) - r~_i| <= the record's band (B); U >= c_J >= L;
  phase 1 (after the default stage): no dropped hypothesis has an exact count at or above the pair's bound (count_viol); every
          record marked exact is F_J bit for bit; every survivor's matrix-core upper count >= its exact count; state 0 <=> the
          exact path rejects; the pair's bound <= the largest exact count == the winner's count;
and the largest 9x9 Jacobi sweep count seen is reported (assumption A1 of DESIGN.md 4.3e: <= 30).
Prints one JSON line; exit code 0 = all checks passed."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MVS_USE_DEBUG_LIB"] = "1"
from mvslam_amd import capi, synth  # noqa: E402

NAMES = ["hypotheses", "state_viol", "count_viol", "upper_viol", "lower_viol", "band_viol", "worst_ratio_bits", "checked",
         "max_sweeps9", "exact_F_mismatch", "unsolved_survivors", "nan_residuals", "mode0_count_mismatch", "matches_checked",
         "rejected_samples", "sum_sweeps9"]


def audit(lib, b, prm, P, phase):
    c = (C.c_ulonglong * 16)()
    maxc, bound, mode = (np.zeros(P, dtype=np.int32) for _ in range(3))
    st = lib.mvs_debug_audit(b._h, C.byref(prm), C.c_int(P), C.c_int(phase), c, maxc.ctypes.data_as(C.POINTER(C.c_int32)),
                             bound.ctypes.data_as(C.POINTER(C.c_int32)), mode.ctypes.data_as(C.POINTER(C.c_int32)))
    assert st == 0, (st, capi.lib().mvs_last_error(b.ctx._h))
    d = {n: int(c[k]) for k, n in enumerate(NAMES)}
    d["worst_ratio"] = float(np.array([d.pop("worst_ratio_bits")], dtype=np.uint64).view(np.float64)[0])
    return d, maxc, bound, mode


def run_case(ctx, lib, name, data, P, N, H, thr):
    prm = capi.default_params(num_hypotheses=H, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=thr)
    b = capi.Batch(ctx, P, N, 32)
    b.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"], data["global_index"])
    b.run(prm)
    b.sync()
    res = b.download(matches=False, mask=False, points=False)["results"]
    t0 = time.time()
    a1, maxc, bound, mode = audit(lib, b, prm, P, 1)          # the stage's decisions (must come first: phase 0 rewrites records)
    a0, _, _, mode0 = audit(lib, b, prm, P, 0)                # the pre-screen's records
    dt = time.time() - t0
    b.close()
    live = res["n_matches"] >= 8
    out = dict(case=name, pairs=P, hypotheses=a1["hypotheses"], seconds=round(dt, 2),
               pairs_mode=[int((mode[live] == m).sum()) for m in (0, 1, 2)], phase0=a0, phase1=a1)
    # host side of phase 1: the bound never exceeds the largest exact count, which is the winner's count
    out["bound_above_max"] = int((bound[live] > maxc[live]).sum())
    out["best_count_mismatch"] = int((res["best_count"][live] != maxc[live]).sum())
    out["mode_changed"] = int((mode0 != mode).sum())
    ok = (a0["state_viol"] == 0 and a0["band_viol"] == 0 and a0["upper_viol"] == 0 and a0["lower_viol"] == 0 and
          a1["state_viol"] == 0 and a1["count_viol"] == 0 and a1["upper_viol"] == 0 and a1["exact_F_mismatch"] == 0 and
          a1["unsolved_survivors"] == 0 and a1["mode0_count_mismatch"] == 0 and out["bound_above_max"] == 0 and
          out["best_count_mismatch"] == 0 and out["mode_changed"] == 0 and a1["max_sweeps9"] <= 30 and
          a1["hypotheses"] == int(live.sum()) * H)
    out["ok"] = bool(ok)
    return out


def main():
    small = len(sys.argv) > 1 and sys.argv[1] == "small"
    ctx = capi.Context(0)
    lib = capi.lib()
    lib.mvs_debug_audit.restype = C.c_int
    cases = []
    if small:
        P, N, H = 6, 500, 2048
        cases.append(run_case(ctx, lib, "small batch @1e-2", synth.make_batch(0, P, n_kp=N), P, N, H, 1e-2))
        cases.append(run_case(ctx, lib, "small batch @reference threshold", synth.make_batch(0, P, n_kp=N), P, N, H, 0.0))
    else:
        P, N, H = 512, 2000, 50000
        data = synth.make_batch(0, P, n_kp=N)
        cases.append(run_case(ctx, lib, "configs[2] @1e-2", data, P, N, H, 1e-2))
        # the reference-threshold regime (no pre-screening: mode 0 everywhere) on a 64-pair slice: the pruned double-precision
        # counting and the selection get the same population-wide check
        sl = {k: (v[:64] if hasattr(v, "__len__") and len(v) == P else v) for k, v in data.items()}
        cases.append(run_case(ctx, lib, "configs[2][:64] @reference threshold", sl, 64, N, H, 0.0))
        # the first 128 pairs of configs[4]'s sequence: pair k = (frame k, frame k + 1), sampler key offset k
        F = 129
        seq = synth.make_sequence(1000, n_kp=N)
        sq = dict(desc1=seq["desc"][:F - 1], kp1=seq["kp"][:F - 1], n1=seq["n_kp"][:F - 1], desc2=seq["desc"][1:F],
                  kp2=seq["kp"][1:F], n2=seq["n_kp"][1:F], K=np.tile(np.asarray(seq["K"]).reshape(1, 9), (F - 1, 1)),
                  global_index=np.arange(F - 1, dtype=np.int64))
        cases.append(run_case(ctx, lib, "configs[4] pairs 0..127 @1e-2", sq, F - 1, N, H, 1e-2))
    # negative control: the same audit must TRIP when the stage and the audit disagree about the threshold (the stage ran at
    # 1e-2, the audit scores at 2e-2: dropped hypotheses then reach the bound, the winner's count differs)
    Pc, Nc, Hc = 4, 500, 2048
    dc = synth.make_batch(0, Pc, n_kp=Nc)
    bc = capi.Batch(ctx, Pc, Nc, 32)
    bc.upload(0, dc["desc1"], dc["kp1"], dc["n1"], dc["desc2"], dc["kp2"], dc["n2"], dc["K"], dc["global_index"])
    bc.run(capi.default_params(num_hypotheses=Hc, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=1e-2))
    bc.sync()
    wrong = capi.default_params(num_hypotheses=Hc, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=2e-2)
    ctrl, cm, cb, _ = audit(lib, bc, wrong, Pc, 1)
    bc.close()
    control = dict(count_viol=ctrl["count_viol"], bound_below_max=int((cm > cb).sum()))
    control_ok = control["count_viol"] > 0 and control["bound_below_max"] == Pc
    ctx.close()
    ok = all(c["ok"] for c in cases) and control_ok
    print(json.dumps(dict(ok=ok, negative_control=control, cases=cases)))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
