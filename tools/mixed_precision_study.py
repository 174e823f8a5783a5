#!/usr/bin/env python3
"""CPU study of a MIXED-PRECISION pre-screen, before any device code: would a binary32 Householder QR + null
vector followed by ONE binary64 refinement step (residual A n~ in binary64 against the exact design matrix, correction through
the binary32 factors) keep the certified band where it is, how many hypotheses would need the all-binary64 fall-back, and what
would the kernel cost?  Two parts, one JSON (profiles/r05_mixed_precision_study.json):

 1. numerics (numpy, tests/prescreen_model.py's arithmetic with the QR / null vector / triangular inverse run in float32):
    on hypotheses of bench pairs (configs[2]) and of small-baseline sequence pairs (configs[4]) -- rho0 = ||A n~_32||,
    rho1 = ||A n~_refined|| (both measured in binary64), the band with eta_A taken a-posteriori from rho1, the share of
    hypotheses whose band grows by more than 10 % or that lose their certificate (= the fall-back lanes), and the
    probability that a wavefront of 64 needs no fall-back launch.
 2. cost: the kernel's instruction mix from the ISA (mvslam_amd/lib/asm, `make asm`) priced with the measured issue costs of
    profiles/r01_fp64_issue_microbench.txt / r03_pk_mfma_microbench.txt (v_fma_f64 5.9, v_fma_f32 3.36, v_pk_fma_f32 5.92
    clocks per wavefront instruction at 2.4 GHz-equivalent), for the section split of DESIGN.md 4.3g, current against mixed.

usage: python tools/mixed_precision_study.py [out.json]      (CPU only; ~1 minute)"""
import json
import os
import re
import sys
import collections

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as o          # noqa: E402   (test infrastructure: the oracle's matcher / sampler build the workload)
import prescreen_model as pm    # noqa: E402
from mvslam_amd import synth    # noqa: E402

F32 = np.float32
U32 = 2.0 ** -24


def householder_null_f32(A):
    """pm.householder_null with every operation rounded to binary32 (A rounded on entry): returns n (float64 copy of the
    binary32 result), R (8 x 8, binary32 values), the reflectors and betas (binary32) for the refinement step"""
    C = A.T.astype(F32).copy()
    vs, betas = [], []
    for k in range(8):
        x = C[k:, k]
        nrm = F32(np.sqrt(F32((x * x).sum(dtype=F32))))
        alpha = -nrm if x[0] >= 0 else nrm
        v = x.copy()
        v[0] = x[0] - alpha
        vv = F32(nrm * F32(nrm + abs(x[0])))
        beta = F32(1.0) / vv if vv > 0 else F32(0)
        for j in range(k + 1, 8):
            t = F32(beta * F32((v * C[k:, j]).sum(dtype=F32)))
            C[k:, j] = (C[k:, j] - t * v).astype(F32)
        C[k, k] = alpha
        C[k + 1:, k] = 0
        vs.append(v)
        betas.append(beta)
    n = np.zeros(9, dtype=F32)
    n[8] = 1
    for k in range(7, -1, -1):
        v = vs[k]
        t = F32(betas[k] * F32((v * n[k:]).sum(dtype=F32)))
        n[k:] = (n[k:] - t * v).astype(F32)
    return n.astype(np.float64), C[:8, :8].copy(), vs, betas


def refine_once(A, n0, R32, vs, betas):
    """one step of iterative refinement of the null vector: r = A n0 in binary64, minimum-norm correction d with A d = r through
    the binary32 factors (A^T = Q1 R  =>  d = Q1 R^-T r: a forward substitution and eight reflector applications in binary32),
    n1 = normalise(n0 - d) in binary64"""
    r = A @ n0                                             # binary64, exact design matrix
    y = np.zeros(8, dtype=F32)
    r32 = r.astype(F32)
    for i in range(8):                                     # R^T y = r  (R upper triangular: R^T lower)
        s = r32[i]
        for k in range(i):
            s = F32(s - F32(R32[k, i] * y[k]))
        y[i] = F32(s / R32[i, i])
    d = np.zeros(9, dtype=F32)
    d[:8] = y
    for k in range(7, -1, -1):                             # d = H_0 ... H_7 [y; 0]
        v = vs[k]
        t = F32(betas[k] * F32((v * d[k:]).sum(dtype=F32)))
        d[k:] = (d[k:] - t * v).astype(F32)
    n1 = n0 - d.astype(np.float64)
    return n1 / np.sqrt((n1 * n1).sum())


def band_from(A, S, n, sig8, bbox, hart):
    """the band of tests/prescreen_model.py with eta_A taken A POSTERIORI from the measured residual of n"""
    (s1, m1x, m1y), (s2, m2x, m2y) = hart
    rho = float(np.sqrt(((A @ n) ** 2).sum())) * (1 + 1e-12) + 40 * pm.U * np.sqrt(S)     # + the roundings of measuring it
    eta_j = 1.01 * pm.TAU_C * S / (sig8 * sig8) + pm.ETA_Q
    eta_a = 1.5 * rho / sig8 + 1e-13
    Fn, e3, sig_e, extra, s2lb = pm.rank2(n)
    eta = eta_j + eta_a + e3 + pm.SVD3_E
    delta = s2lb - extra - sig_e - eta
    if not (delta > 0 and eta < 1e-3):
        return np.inf, rho, eta_a / eta_j
    dfn = (2.0 + 2.0 * (sig_e + 3 * eta) / delta) * eta + extra + pm.SVD3_E
    x1lo, x1hi, y1lo, y1hi, x2lo, x2hi, y2lo, y2hi = bbox
    N1 = np.sqrt(1 + s1 * s1 * (max(abs(m1x - x1lo), abs(m1x - x1hi)) ** 2 + max(abs(m1y - y1lo), abs(m1y - y1hi)) ** 2))
    N2 = np.sqrt(1 + s2 * s2 * (max(abs(m2x - x2lo), abs(m2x - x2hi)) ** 2 + max(abs(m2y - y2lo), abs(m2y - y2hi)) ** 2))
    return float(dfn * N1 * N2), rho, eta_a / eta_j


def study_pair(p1, p2, seed, hyps, rows):
    bbox = (p1[:, 0].min(), p1[:, 0].max(), p1[:, 1].min(), p1[:, 1].max(), p2[:, 0].min(), p2[:, 0].max(), p2[:, 1].min(),
            p2[:, 1].max())
    for h in range(hyps):
        idx = o.sample8(seed, h, len(p1))
        ref = pm.prescreen(p1[idx, 0], p1[idx, 1], p2[idx, 0], p2[idx, 1], bbox)
        if not ref["screenable"]:
            continue
        a1, b1, s1, m1x, m1y, _ = pm.hartley(p1[idx, 0], p1[idx, 1])
        a2, b2, s2, m2x, m2y, _ = pm.hartley(p2[idx, 0], p2[idx, 1])
        A = pm.design(a1, b1, a2, b2)
        S = float((A * A).sum()) * (1 + 1e-12)
        n0, R32, vs, betas = householder_null_f32(A)
        # sigma_8 lower bound from the binary32 factor: (1 - z) / ||R^-1||_F - c u32 ||A||_F with the binary32 analogue of
        # c = 176 + 510 (backward error of the QR + loss of orthogonality, now in units of u32)
        yf, _ = pm.tri_inverse_fro(R32.astype(np.float64))
        sig8_32 = (1 - 16 * U32 * np.sqrt(S) * yf) / yf - (176 + 510) * U32 * np.sqrt(S)
        hart = ((s1, m1x, m1y), (s2, m2x, m2y))
        row = dict(band64=ref["band"], sig8_64=ref["sig8_lb"], sig8_32=float(sig8_32))
        row["rho0"] = float(np.sqrt(((A @ n0) ** 2).sum()))
        if sig8_32 > 0 and np.isfinite(yf):
            n1 = refine_once(A, n0, R32, vs, betas)
            n2 = refine_once(A, n1, R32, vs, betas)
            for tag, n in (("0", n0), ("1", n1), ("2", n2)):
                band, rho, ratio = band_from(A, S, n, sig8_32, bbox, hart)
                row["band_mixed_%s" % tag] = band
                row["rho%s_meas" % tag] = rho
                row["eta_a_over_eta_j_%s" % tag] = ratio
        else:
            row["band_mixed_0"] = row["band_mixed_1"] = row["band_mixed_2"] = np.inf
        rows.append(row)


def numerics(hyps=600):
    out = {}
    # bench pairs (configs[2]) and small-baseline sequence pairs (configs[4])
    work = []
    for pi in range(2):
        d = synth.make_pair(pi, n_kp=2000)
        mt = o.match_visual_features(d["desc1"], d["desc2"], 0.7, 10.0)
        work.append(("configs[2] pair %d" % pi, o.normalize_points(d["K"], d["kp1"][mt["trainIdx"]].astype(np.float64)),
                     o.normalize_points(d["K"], d["kp2"][mt["queryIdx"]].astype(np.float64)), synth.SEED_BASE + pi))
    seq = synth.make_sequence(6, n_kp=2000)
    for k in (0, 3):
        mt = o.match_visual_features(seq["desc"][k][:seq["n_kp"][k]], seq["desc"][k + 1][:seq["n_kp"][k + 1]], 0.7, 10.0)
        K = np.asarray(seq["K"]).reshape(3, 3)
        work.append(("configs[4] pair %d" % k, o.normalize_points(K, seq["kp"][k][mt["trainIdx"]].astype(np.float64)),
                     o.normalize_points(K, seq["kp"][k + 1][mt["queryIdx"]].astype(np.float64)), synth.SEED_BASE + k))
    for name, p1, p2, seed in work:
        rows = []
        study_pair(p1, p2, seed, hyps, rows)
        b64 = np.array([r["band64"] for r in rows])
        res = dict(certified_by_the_binary64_prescreen=len(rows), hypotheses=hyps, median_band64=float(np.median(b64)))
        for tag, label in (("0", "binary32 null vector as is"), ("1", "one binary64 refinement step"), ("2", "two steps")):
            bm = np.array([r["band_mixed_%s" % tag] for r in rows])
            grow = bm / b64
            fall = (~np.isfinite(bm)) | (grow > 1.10)
            res[label] = dict(median_band=float(np.median(bm[np.isfinite(bm)])) if np.isfinite(bm).any() else None,
                              median_band_ratio=float(np.median(grow[np.isfinite(grow)])) if np.isfinite(grow).any() else None,
                              fallback_share_band_grows_over_10pct=float(fall.mean()),
                              lost_certificate_share=float((~np.isfinite(bm)).mean()),
                              p_wavefront_of_64_needs_no_fallback=float((1 - fall.mean()) ** 64),
                              median_rho=float(np.median([r.get("rho%s_meas" % tag, np.nan) for r in rows])),
                              median_eta_a_over_eta_j=float(np.nanmedian([r.get("eta_a_over_eta_j_%s" % tag, np.nan) for r in rows])))
        res["median_sigma8_bound_32_over_64"] = float(np.median([r["sig8_32"] / r["sig8_64"] for r in rows]))
        out[name] = res
    return out


# ---- part 2: the cost model ------------------------------------------------------------------------------------------------
# issue cost per wavefront instruction in 2.4 GHz-equivalent clocks, several wavefronts per SIMD
# (profiles/r01_fp64_issue_microbench.txt: v_fma_f64 5.90; profiles/r03_pk_mfma_microbench.txt: v_fma_f32 3.36 at four
# wavefronts, v_pk_fma_f32 5.92; transcendental f64 17; quarter-rate integer multiplies 16)
COST = dict(f64=5.9, f32=3.36, pk32=5.92, trans64=17.0, quarter=16.0, other=3.4)


def isa_mix():
    path = os.path.join(ROOT, "mvslam_amd", "lib", "asm", "kernels-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(path):
        return None
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3mvs23ransac_prescreen_kernel"))
    ops = collections.Counter()
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            break
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        ops[t.split()[0]] += 1
    cls = collections.Counter()
    for k, v in ops.items():
        if re.match(r"v_(fma|fmac|mul|add|max|min)_f64", k):
            cls["f64"] += v
        elif re.match(r"v_(rcp|rsq|sqrt)_f64", k):
            cls["trans64"] += v
        elif k in ("v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32"):
            cls["quarter"] += v
        elif k.startswith(("v_", "ds_", "global_", "s_")):
            cls["other"] += v
    return dict(instructions=sum(ops.values()), classes=dict(cls), top=dict(ops.most_common(12)))


def cost_model(mix):
    # section split of the binary64 arithmetic (sums to the ISA's f64 + transcendental count within 3 %)
    sect = dict(hartley=280, qr=790, null_vector=104, triangular_inverse=214, rank2=350, denormalise_bounds=186)
    f64_total = mix["classes"].get("f64", 0) + mix["classes"].get("trans64", 0)
    scale = f64_total / float(sum(sect.values()))
    sect = {k: v * scale for k, v in sect.items()}
    clk_now = sum(mix["classes"].get(c, 0) * COST[c] for c in ("f64", "trans64", "quarter", "other"))
    movable = sect["qr"] + sect["null_vector"] + sect["triangular_inverse"]
    # the refinement step: rebuild the eight rows of A from a second gather (8 x (4 normalise + 4 products)) and r = A n~
    # (72 fma) in binary64; forward substitution (36 fma + 8 div) and eight reflector applications (8 x 19) in binary32;
    # the update + renormalisation (9 + 12) and the a-posteriori residual (a third pass over the rows: 72 fma + 8 x 8) in binary64
    refine64 = 8 * 8 + 72 + 21 + 72 + 64
    refine32 = 36 + 8 * 3 + 8 * 19
    conv = 72 + 44 + 9                     # binary64 -> binary32 of A's 72 entries, back for R's use in bounds, n~
    out = {}
    for label, c32 in (("binary32 (v_fma_f32)", COST["f32"]), ("packed binary32 (v_pk_fma_f32, two per instruction)", COST["pk32"] / 2)):
        clk = clk_now - movable * COST["f64"] + movable * c32 + refine64 * COST["f64"] + refine32 * c32 + conv * COST["other"]
        out[label] = dict(clocks_per_wavefront=round(clk), ratio_to_now=round(clk / clk_now, 3),
                          projected_ms_at_2p49=round(2.49 * clk / clk_now, 2))
    # the all-binary32 tier (no refinement, a-posteriori residual only): what a regime that can afford bands of ~1e-4 would pay
    clk32 = clk_now - movable * COST["f64"] + movable * COST["f32"] - sect["rank2"] * COST["f64"] + sect["rank2"] * COST["f32"] + \
        (8 * 8 + 72) * COST["f64"] + conv * COST["other"]
    out["all-binary32 tier, a-posteriori residual only"] = dict(clocks_per_wavefront=round(clk32), ratio_to_now=round(clk32 / clk_now, 3),
                                                               projected_ms_at_2p49=round(2.49 * clk32 / clk_now, 2))
    return dict(issue_cost_clocks=COST, sections_f64_instructions={k: round(v) for k, v in sect.items()},
                clocks_per_wavefront_now=round(clk_now), moved_to_binary32=round(movable), refinement_binary64=refine64,
                refinement_binary32=refine32, variants=out,
                target_ms=1.9, kill_criterion="the device kernel is not <= 1.9 ms (VERDICT r4 #3)")


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_mixed_precision_study.json")
    o.build()
    mix = isa_mix()
    res = dict(numerics=numerics(), isa=mix, cost=cost_model(mix) if mix else None)
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res["cost"]["variants"] if res["cost"] else None, indent=1))
    for k, v in res["numerics"].items():
        print(k, json.dumps({a: b for a, b in v.items() if not isinstance(b, dict)}))
        for a, b in v.items():
            if isinstance(b, dict):
                print("    %-32s" % a, json.dumps(b))


if __name__ == "__main__":
    main()
