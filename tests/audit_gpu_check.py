#!/usr/bin/env python3
"""Full-population device audit of the pre-screened RANSAC stage (VERDICT r3 #1b), since round 5 of the PRODUCT BINARY
(VERDICT r4 #6): libmvslam_hip.so runs the stage; its device-resident state (mvs_batch_device_state: a read-only view) is
handed, in the same process, to the audit kernel of libmvslam_hip_dbg.so, which only replays every hypothesis exactly and
compares -- every byte the checks read was written by the product library's kernels.  Run by tests/test_prescreen.py in its
own process.

The reference's rule (estimator-RANSAC.cpp:76-84) decides over ALL hypotheses of a pair; the pre-screen decides which of
them are ever solved exactly.  Here every hypothesis of BASELINE configs[2] (512 pairs x 50 000 = 25.6 M) and of the first
128 pairs of configs[4]'s sequence is solved exactly once more ON THE DEVICE and scored exactly on every match, and
  phase 2 (records as the product's pre-screen wrote them and the stage left them -- every record still in the approximate
          state, i.e. all but the few the stage solved exactly afterwards): state byte 0 only for samples the exact path
          rejects; for every such record and every match |r_i(F_J) - r~_i| <= the record's band (B); U >= c_J >= L;
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
os.environ.pop("MVS_USE_DEBUG_LIB", None)      # the batch below belongs to the PRODUCT library
from mvslam_amd import capi, synth  # noqa: E402

NAMES = ["hypotheses", "state_viol", "count_viol", "upper_viol", "lower_viol", "band_viol", "worst_ratio_bits", "checked",
         "max_sweeps9", "exact_F_mismatch", "unsolved_survivors", "nan_residuals", "mode0_count_mismatch", "matches_checked",
         "rejected_samples", "sum_sweeps9"]


def audit(dbg, b, prm, P, phase):
    """dbg = (diagnostics library, its own context); b = a batch of the product library, synchronised"""
    dlib, dctx = dbg
    c = (C.c_ulonglong * 16)()
    maxc, bound, mode = (np.zeros(P, dtype=np.int32) for _ in range(3))
    state = b.device_state()
    st = dlib.mvs_debug_audit_state(dctx, state, C.c_size_t(len(state)), C.byref(prm), C.c_int(P), C.c_int(phase), c,
                                    maxc.ctypes.data_as(C.c_void_p), bound.ctypes.data_as(C.c_void_p),
                                    mode.ctypes.data_as(C.c_void_p))
    assert st == 0, (st, dlib.mvs_last_error(dctx))
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
    a1, maxc, bound, mode = audit(lib, b, prm, P, 1)          # the stage's decisions
    a0, _, _, mode0 = audit(lib, b, prm, P, 2)                # the product's approximate records, as the stage left them
    dt = time.time() - t0
    b.close()
    live = res["n_matches"] >= 8
    out = dict(case=name, pairs=P, hypotheses=a1["hypotheses"], seconds=round(dt, 2),
               pairs_mode=[int((mode[live] == m).sum()) for m in (0, 1, 2)], phase2=a0, phase1=a1)
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


def run_crafted(ctx, lib, path):
    """The matches tests/constants_gpu_check.py crafted ON the band edges of chosen hypotheses (coordinates half-way between two
    binary32 numbers; half of them land on the wrong side in plain binary32), through the PRODUCT binary: the point sets go
    in as a batch of sfm_solve calls (mvs_batch_run_points, K = identity: ideal coordinates pass through bit for bit), the
    product's stage runs on them, and the audit checks every hypothesis of every pair against what it left behind."""
    z = np.load(path)
    P, N, H, thr = int(z["pairs"]), int(z["capacity"]), int(z["hypotheses"]), float(z["thr"])
    prm = capi.default_params(num_hypotheses=H, sampler=capi.SAMPLER_PHILOX, seed=int(z["seed"]), max_error_sq=thr)
    b = capi.Batch(ctx, P, N, 32)
    b.upload_intrinsics(0, np.eye(3), z["global_index"], count=P)
    out = dict(case="crafted band-edge matches through mvs_batch_run_points", trials=int(z["trials"]), pairs=P, viol=0, checked=0,
               matches_checked=0, hypotheses=0, survivors_checked=0)
    for t in range(int(z["trials"])):
        sets = [z["pts_%d_%d" % (t, p)] for p in range(P)]
        m = np.array([len(q) for q in sets], dtype=np.int32)
        uv1, uv2 = np.zeros((P, N, 2)), np.zeros((P, N, 2))
        for p, q in enumerate(sets):
            uv1[p, :len(q)], uv2[p, :len(q)] = q[:, :2], q[:, 2:]
        b.run_points(prm, uv1, uv2, m)
        b.sync()
        res = b.download(matches=False, mask=False, points=False)["results"]
        assert (res["n_matches"] == m).all()
        a1, maxc, bound, mode = audit(lib, b, prm, P, 1)
        a2, _, _, _ = audit(lib, b, prm, P, 2)
        out["viol"] += sum(a1[k] for k in ("state_viol", "count_viol", "upper_viol", "exact_F_mismatch", "unsolved_survivors",
                                            "mode0_count_mismatch")) + \
            sum(a2[k] for k in ("state_viol", "band_viol", "upper_viol", "lower_viol")) + \
            int((bound > maxc).sum()) + int((res["best_count"] != maxc).sum())
        out["checked"] += a2["checked"]
        out["matches_checked"] += a2["matches_checked"]
        out["hypotheses"] += a1["hypotheses"]
        out["survivors_checked"] += a1["checked"]
        out["pairs_mode"] = [int((mode == k).sum()) for k in (0, 1, 2)]
    b.close()
    out["ok"] = out["viol"] == 0 and out["checked"] > 0.5 * out["hypotheses"] and out["hypotheses"] == int(z["trials"]) * P * H
    return out


def main():
    small = len(sys.argv) > 1 and sys.argv[1] == "small"
    crafted = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "crafted" else None
    ctx = capi.Context(0)
    assert capi.LIB_PATH.endswith("libmvslam_hip.so")          # the product binary runs the stage ...
    dlib = capi.dbg_lib()                                      # ... the diagnostics binary only audits what it left behind
    dctx = C.c_void_p()
    assert dlib.mvs_ctx_create(0, C.byref(dctx)) == 0
    lib = (dlib, dctx)
    cases = []
    if crafted:
        c = run_crafted(ctx, lib, crafted)
        ctx.close()
        dlib.mvs_ctx_destroy(dctx)
        print(json.dumps(dict(ok=c["ok"], stage_binary=os.path.basename(capi.LIB_PATH), audit_binary=os.path.basename(capi.DBG_LIB_PATH),
                              cases=[c])))
        return 0 if c["ok"] else 1
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
    dlib.mvs_ctx_destroy(dctx)
    ok = all(c["ok"] for c in cases) and control_ok
    print(json.dumps(dict(ok=ok, stage_binary=os.path.basename(capi.LIB_PATH), audit_binary=os.path.basename(capi.DBG_LIB_PATH),
                          negative_control=control, cases=cases)))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
