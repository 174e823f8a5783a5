#!/usr/bin/env python3
"""How far can the contract's FUSED epipolar residual move a result away from the reference's UNFUSED one?

VERDICT r1 ("parity is green, with a hole nobody can close from here"): the oracle and the kernels score a match with
r = |fma(u0, x1, fma(u1, y1, u2))|, u_j = fma(x2, F0j, fma(y2, F1j, F2j)); the reference evaluates
`p2.transpose() * F * p1` (estimator-RANSAC.cpp:114) with separate multiplies and adds.  A residual within a rounding
error of the threshold can fall on the other side.  This script runs the whole image-pair oracle twice per pair -- the
contract's form and the reference's form (orc_set_residual_form) -- on the bench workload and counts what changes: the
winning hypothesis, the inlier mask, the pose.  CPU only; test infrastructure (it touches nothing but oracle/).
usage: python tests/contract_sensitivity.py [--pairs 32] [--hyp 50000] [--max-error-sq 1e-2 | 0] [--jacobi]"""
import argparse, json, os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as o
from mvslam_amd import synth


def run(n_pairs, H, thr, n_kp=2000, threads=8, jacobi=False, first=0):
    """jacobi=False: only the epipolar residual changes form; True: the Jacobi SVD inner loops too (OpenCV's literal
    p += a*b, hypot, c*x + s*y, a += t*t without contraction) -- every place the contract fuses or rewrites."""
    data = synth.make_batch(first, n_pairs, n_kp=n_kp)
    out = {}
    for form in (0, 1):
        o.lib().orc_set_residual_form(form)   # process-global: set before the worker threads start
        o.lib().orc_set_jacobi_form(form if jacobi else 0)
        res = [None] * n_pairs

        def work(k0):
            for i in range(k0, n_pairs, threads):
                res[i] = o.image_pair(data["desc1"][i], data["kp1"][i], data["desc2"][i], data["kp2"][i],
                                      data["K"][i].reshape(3, 3),
                                      o.make_params(H, o.SAMPLER_PHILOX, synth.SEED_BASE + int(data["global_index"][i]), thr),
                                      0.7, 10.0)
        ths = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        out[form] = res
    o.lib().orc_set_residual_form(0)
    o.lib().orc_set_jacobi_form(0)
    winners = flips = bits = pose_changed = 0
    dpose = dF = 0.0
    changed_detail = []
    dcount = []
    for a, b in zip(out[0], out[1]):
        M = a["n_matches"]
        bits += M
        if a["best_hyp"] != b["best_hyp"]:
            winners += 1
        flips += int((a["mask"][:M] != b["mask"][:M]).sum())
        dcount.append(int(b["best_count"]) - int(a["best_count"]))
        if a["best_hyp"] == b["best_hyp"] and a["best_hyp"] >= 0:
            dF = max(dF, float(np.abs(a["F"] - b["F"]).max() / np.abs(a["F"]).max()))
        if a["ok"] and b["ok"]:
            d = max(float(np.abs(a["R"] - b["R"]).max()), float(np.abs(a["t"] - b["t"]).max()))
            if d > 1e-6:   # another candidate of the decomposition, or another set of surviving points
                pose_changed += 1
                changed_detail.append(dict(inliers=int(a["n_inliers"]), points=[int(a["n_points"]), int(b["n_points"])], difference=d))
            else:
                dpose = max(dpose, d)
        elif a["ok"] != b["ok"]:
            pose_changed += 1
            changed_detail.append(dict(inliers=int(a["n_inliers"]), ok=[bool(a["ok"]), bool(b["ok"])]))
    return dict(pairs=n_pairs, hypotheses=H, max_error_sq=thr, winners_changed=winners, mask_bits=bits, mask_bits_flipped=flips,
                best_count_delta_min=min(dcount), best_count_delta_max=max(dcount), pairs_with_another_pose=pose_changed,
                another_pose_detail=changed_detail[:8], max_pose_entry_difference_otherwise=dpose,
                max_relative_F_difference_same_winner=dF, jacobi_forms=bool(jacobi))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--hyp", type=int, default=50000)
    ap.add_argument("--kp", type=int, default=2000)
    ap.add_argument("--first", type=int, default=0, help="index of the first synthetic pair")
    ap.add_argument("--max-error-sq", type=float, default=1e-2, help="0 = the reference's 5e-2 / K00 / K11")
    ap.add_argument("--jacobi", action="store_true", help="also switch the Jacobi SVD inner loops to OpenCV's literal forms")
    a = ap.parse_args()
    print(json.dumps(dict(first_pair=a.first, **run(a.pairs, a.hyp, a.max_error_sq, a.kp, jacobi=a.jacobi, first=a.first))))
