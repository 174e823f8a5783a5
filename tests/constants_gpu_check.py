#!/usr/bin/env python3
"""Worst-case-construction checks of the pre-screen's hand-derived constants, THROUGH THE DEVICE PATH (VERDICT r3 #1c; run by
tests/test_constants.py in its own process with MVS_USE_DEBUG_LIB=1 -- the probes exist in the diagnostics library only).

Random soaks sit orders of magnitude inside every bound of DESIGN.md 4.3e and therefore cannot see a constant that is too
small (one was, for most of round 3).  Each check below constructs operands AT the boundary the constant has to cover and
compares the device with exact rational arithmetic:

 A  64 u N1' N2' (prescreen.hpp, (v)): the roundings of the de-normalisation and of the fused residual evaluation, in both
    paths' own code (prescreen_denormalise / denormalise_exact + epipolar_residual), on box-corner points with the transform
    means placed for the largest |x| + |m|, Fn with cancelling terms: |fl(r) - r_exact| <= 32 u N1' N2' per path.
 B  the compare-free indicator of the matrix-core counting at |a| = tu' (1 +- k ulp) and |a| = tl' (1 +- k ulp): every |a| < tu'
    is counted by the upper form, no |a| >= tl' by the lower form -- with tu', tl' from the kernels' own threshold code.
 C  e32 = 16 * 2^-24 T (vector kernel) and 2^-14 T (matrix cores, on top of it): matches constructed ON the epipolar bands
    r(F~, p) = (thr + band)(1 - delta) and (thr - band)(1 + delta), delta down to 1e-9, with coordinates half-way between two
    binary32 numbers (the largest input rounding), counted by ransac_count32 and by pilot + ransac_count_mfma +
    ransac_finish_mfma: U >= #{r_exact < thr + band} and L <= #{r_exact < thr - band} in exact arithmetic.
Prints one JSON line; exit code 0 = all checks passed."""
import ctypes as C
import json
import os
import sys
from fractions import Fraction as Fr

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MVS_USE_DEBUG_LIB"] = "1"
import oracle_lib as o  # noqa: E402
from mvslam_amd import capi, synth  # noqa: E402

U64 = 2.0 ** -53
PD = C.POINTER(C.c_double)
PF = C.POINTER(C.c_float)
PI = C.POINTER(C.c_int32)


# ---- A: de-normalisation + residual roundings against exact arithmetic -----------------------------------------------------
def check_rounding(ctx, lib, rng, n=6000):
    bx, by = 0.61, 0.46                                   # 640x480 at f = 525: the box of every synthetic pair
    rows = np.zeros((n, 19))
    for i in range(n):
        Fn = rng.normal(size=9)
        if i % 3 == 0:                                    # signs arranged so that all nine terms of the residual are positive
            Fn = np.abs(Fn)
        Fn /= np.linalg.norm(Fn)
        s1, s2 = rng.uniform(1.0, 40.0, 2)
        corner = rng.integers(0, 2, 4) * 2 - 1            # box corner of the match
        x1, y1, x2, y2 = corner * np.array([bx, by, bx, by])
        # means: opposite corner region (largest |x - m| and |x| + |m|) or next to the point (|x - m| tiny, |x| + |m| large)
        far = rng.random() < 0.5
        m = -corner * np.array([bx, by, bx, by]) * rng.uniform(0.5, 1.0, 4) if far else \
            np.array([x1, y1, x2, y2]) * (1 - rng.uniform(1e-6, 1e-2, 4))
        if i % 3 == 1:
            # cancellation: choose Fn[8] so that the exact residual (almost) vanishes while the terms stay O(N1' N2')
            a1, b1 = s1 * (x1 - m[0]), s1 * (y1 - m[1])
            a2, b2 = s2 * (x2 - m[2]), s2 * (y2 - m[3])
            ph = np.array([a2 * a1, a2 * b1, a2, b2 * a1, b2 * b1, b2, a1, b1, 1.0])
            Fn[8] = -float(ph[:8] @ Fn[:8])
            Fn /= max(1.0, np.linalg.norm(Fn))
        rows[i] = np.r_[Fn, s1, s2, m, x1, y1, x2, y2]
    out = np.zeros((n, 2))
    st = lib.mvs_debug_rounding_probe(ctx._h, rows.ctypes.data_as(PD), C.c_int(n), out.ctypes.data_as(PD))
    assert st == 0, st
    worst = [0.0, 0.0]
    for i in range(n):
        q = [Fr(float(v)) for v in rows[i]]
        Fn, s1, s2, m1x, m1y, m2x, m2y, x1, y1, x2, y2 = q[:9], *q[9:]
        a1, b1, a2, b2 = s1 * (x1 - m1x), s1 * (y1 - m1y), s2 * (x2 - m2x), s2 * (y2 - m2y)
        ph = [a2 * a1, a2 * b1, a2, b2 * a1, b2 * b1, b2, a1, b1, Fr(1)]
        r_exact = abs(sum(p * f for p, f in zip(ph, Fn)))
        n1p = float(1 + s1 * s1 * ((abs(x1) + abs(m1x)) ** 2 + (abs(y1) + abs(m1y)) ** 2)) ** 0.5
        n2p = float(1 + s2 * s2 * ((abs(x2) + abs(m2x)) ** 2 + (abs(y2) + abs(m2y)) ** 2)) ** 0.5
        for k in range(2):
            err = float(abs(Fr(float(out[i, k])) - r_exact))
            worst[k] = max(worst[k], err / (U64 * n1p * n2p))
    return dict(cases=n, worst_over_u_N1N2_prescreen=worst[0], worst_over_u_N1N2_exact=worst[1], budget_per_path=32.0,
                ok=bool(worst[0] <= 32.0 and worst[1] <= 32.0))


# ---- B: the indicator at the thresholds ---------------------------------------------------------------------------------------
def f32(x):
    return np.float32(x)


def check_indicator(ctx, lib, rng):
    tus = np.float32(np.r_[10.0 ** rng.uniform(-9, 1, 400), [1e-2, 1.8e-7, 1.0, 2.0 ** -30, 3.0]])
    Ts = np.float32(np.r_[10.0 ** rng.uniform(-3, 2, len(tus) - 5), [0.0, 1.0, 2.0 ** -10, 100.0, 0.5]])
    a, tu, tl, T, kind = [], [], [], [], []
    one = np.float32(1.0)
    for t, TT in zip(tus, Ts):
        # the kernels' own expressions, in numpy binary32 (one IEEE operation per step)
        tup = f32(f32(t + f32(f32(2.0 ** -14) * TT)) * f32(one + f32(2.0 ** -22)))
        lrec = f32(t * f32(0.7))                                   # some lower threshold of the same record
        tlp = f32(f32(lrec - f32(f32(2.0 ** -14) * TT)) * f32(one - f32(2.0 ** -22)))
        for base, tag in ((tup, "u"), (tlp, "l")):
            if not base > 0:
                vals = [f32(0.0), f32(1e-30), t]
            else:
                vals = [base]
                lo, hi = base, base
                for _ in range(6):
                    lo, hi = np.nextafter(lo, f32(0.0)), np.nextafter(hi, f32(np.inf))
                    vals += [lo, hi]
                vals += [f32(base * f32(1 - 2.0 ** -18)), f32(base * f32(1 + 2.0 ** -18)), f32(0.0), f32(base * f32(0.5)),
                         f32(base * f32(2.0)), f32(3e38)]
            for v in vals:
                a.append(v); tu.append(t); tl.append(lrec); T.append(TT); kind.append((tag, float(tup), float(tlp)))
    a, tu, tl, T = (np.ascontiguousarray(v, dtype=np.float32) for v in (a, tu, tl, T))
    n = len(a)
    iu, il, sc = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    st = lib.mvs_debug_indicator_probe(ctx._h, a.ctypes.data_as(PF), tu.ctypes.data_as(PF), tl.ctypes.data_as(PF),
                                       T.ctypes.data_as(PF), C.c_int(n), iu.ctypes.data_as(PF), il.ctypes.data_as(PF),
                                       sc.ctypes.data_as(PF))
    assert st == 0, st
    bad_u = bad_l = loose_u = loose_l = shape = 0
    # the dense phase's normalisation: q = s a is tested by its exponent, |q| < 2.  Sound iff s tu' < 2 in exact arithmetic
    # (every |a| < tu' then has |s a| < 2); tight iff s tu' is within a few ulps of 2
    bad_scale = sum(1 for i in range(n) if not Fr(float(sc[i])) * Fr(kind[i][1]) < 2)
    loose_scale = sum(1 for i in range(n) if kind[i][1] > 2.0 ** -50 and Fr(float(sc[i])) * Fr(kind[i][1]) < Fr(2) * (1 - Fr(1, 2 ** 20)))
    for i in range(n):
        _, tup, tlp = kind[i]
        av = abs(float(a[i]))
        shape += int(iu[i] not in (0.0, 1.0)) + int(il[i] not in (0.0, 1.0))     # an indicator is 0 or 1, same for +a and -a
        if av < tup and iu[i] != 1.0:
            bad_u += 1                      # SOUNDNESS: an |a| below tu' that the upper count misses
        if (tlp <= 0 or av >= tlp) and il[i] != 0.0:
            bad_l += 1                      # SOUNDNESS: an |a| at or above tl' that the lower count takes
        if av > tup * (1 + 2.0 ** -17) and iu[i] != 0.0:
            loose_u += 1                    # tightness only
        if tlp > 0 and av < tlp * (1 - 2.0 ** -17) and il[i] != 1.0:
            loose_l += 1
    return dict(values=n, upper_missed=bad_u, lower_overcounted=bad_l, not_an_indicator=shape, upper_loose=loose_u,
                lower_loose=loose_l, scale_unsound=bad_scale, scale_loose=loose_scale,
                ok=bool(bad_u == 0 and bad_l == 0 and shape == 0 and loose_u == 0 and loose_l == 0 and bad_scale == 0 and
                        loose_scale == 0))


# ---- C: matches on the band edges through the counting kernels ---------------------------------------------------------------
def midway32(x):
    """the double half-way (minus a hair) between two neighbouring binary32 numbers near x: the largest input rounding"""
    f = np.float32(x)
    g = np.nextafter(f, np.float32(np.inf))
    return float(f) + 0.4999 * (float(g) - float(f))


def exact_residual(F, p):
    x1, y1, x2, y2 = (Fr(float(v)) for v in p)
    f = [Fr(float(v)) for v in F]
    return abs(x2 * (f[0] * x1 + f[1] * y1 + f[2]) + y2 * (f[3] * x1 + f[4] * y1 + f[5]) + (f[6] * x1 + f[7] * y1 + f[8]))


def craft_point(F, target, box, rng, sign):
    """a match inside the box whose exact residual under F is `target` up to the rounding of y2"""
    for _ in range(200):
        x1 = midway32(rng.uniform(box[0] * 0.9, box[1] * 0.9))
        y1 = midway32(rng.uniform(box[2] * 0.9, box[3] * 0.9))
        x2 = midway32(rng.uniform(box[4] * 0.9, box[5] * 0.9))
        f = [Fr(float(v)) for v in F]
        X1, Y1, X2 = Fr(x1), Fr(y1), Fr(x2)
        # r = | x2 (F0 x1 + F1 y1 + F2) + y2 (F3 x1 + F4 y1 + F5) + (F6 x1 + F7 y1 + F8) |
        A = f[3] * X1 + f[4] * Y1 + f[5]
        Bc = X2 * (f[0] * X1 + f[1] * Y1 + f[2]) + (f[6] * X1 + f[7] * Y1 + f[8])
        if A == 0:
            continue
        y2 = float((sign * Fr(target) - Bc) / A)
        if box[6] * 0.98 < y2 < box[7] * 0.98:
            return np.array([x1, y1, x2, y2])
    return None


def check_band_edges(ctx, lib, rng):
    P, N, H, THR = 6, 1500, 2048, 1e-2
    data = synth.make_batch(40, P, n_kp=N)
    prm = capi.default_params(num_hypotheses=H, sampler=capi.SAMPLER_PHILOX, seed=synth.SEED_BASE, max_error_sq=THR)
    b = capi.Batch(ctx, P, N, 32)
    b.upload(0, data["desc1"], data["kp1"], data["n1"], data["desc2"], data["kp2"], data["n2"], data["K"], data["global_index"])
    b.run(prm)
    b.sync()
    Hp = ((H + 255) // 256) * 256
    rows = []
    ok = True
    saved = {}                                  # the crafted point sets, for the product binary's leg (audit_gpu_check.py crafted)
    for trial in range(4):                      # per pair: hypotheses inside and outside the pilot's range
        pts, keep, crafted, recs = [], np.full(P, -1, dtype=np.int32), [], []
        # pass 1: natural points, double-precision records of every hypothesis
        st = lib.mvs_debug_count_only(b._h, C.byref(prm), C.c_int(P), C.c_int(2), C.c_int(1), None, None, None, None)
        assert st == 0, st
        for p in range(P):
            buf = np.zeros((N, 4))
            m = C.c_int(0)
            assert lib.mvs_debug_get_points(b._h, C.c_int(p), C.byref(m), buf.ctypes.data_as(PD)) == 0
            M = m.value
            P4 = buf[:M].copy()
            rec, state = np.zeros((H, 10)), np.zeros(H, dtype=np.uint8)
            assert lib.mvs_debug_read_hyp_rec(b._h, C.c_int(p), C.c_int(H), rec.ctypes.data_as(PD),
                                              state.ctypes.data_as(C.POINTER(C.c_ubyte)), None, None) == 0
            lo, hi = P4.min(0), P4.max(0)
            box = [lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lo[3], hi[3]]
            extremes = set(np.r_[P4.argmin(0), P4.argmax(0)].tolist())
            slots = [i for i in list(range(0, 48)) + list(range(M - 48, M)) if i not in extremes]
            seed = synth.SEED_BASE + int(data["global_index"][p])
            cand = range(16 * trial, 256) if trial % 2 == 0 else range(256 + 64 * trial, H)
            h = next(hh for hh in cand if state[hh] == 1 and not set(o.sample8(seed, hh, M).tolist()) & set(slots))
            F, tub = rec[h, :9].copy(), float(rec[h, 9])
            band = tub - THR
            n_up = n_lo = 0
            for j, i in enumerate(slots):
                delta = 10.0 ** rng.uniform(-9, -7)
                upper = j % 2 == 0 or THR - band <= 0
                target = (THR + band) * (1 - delta) if upper else (THR - band) * (1 + delta)
                q = craft_point(F, target, box, rng, 1 if rng.random() < 0.5 else -1)
                if q is not None:
                    P4[i] = q
                    n_up += int(upper)
                    n_lo += int(not upper)
            assert lib.mvs_debug_set_points(b._h, C.c_int(p), C.c_int(M), np.ascontiguousarray(P4).ctypes.data_as(PD)) == 0
            pts.append(P4); keep[p] = h; crafted.append((n_up, n_lo)); recs.append((F, tub))
        # pass 2: the crafted points in place.  The records must not have moved (same sample, same box)
        st = lib.mvs_debug_count_only(b._h, C.byref(prm), C.c_int(P), C.c_int(2), C.c_int(1), None, None, None, None)
        assert st == 0, st
        for p in range(P):
            rec = np.zeros((H, 10))
            assert lib.mvs_debug_read_hyp_rec(b._h, C.c_int(p), C.c_int(H), rec.ctypes.data_as(PD), None, None, None) == 0
            assert rec[keep[p]].tobytes() == np.r_[recs[p][0], recs[p][1]].tobytes(), "the crafted points moved the record"
        res = {}
        for name, dense in (("count32", 0), ("mfma", 1)):
            cnt, bound, n1 = (np.zeros(P, dtype=np.int32) for _ in range(3))
            st = lib.mvs_debug_count_only(b._h, C.byref(prm), C.c_int(P), C.c_int(1), C.c_int(dense), keep.ctypes.data_as(PI),
                                          cnt.ctypes.data_as(PI), bound.ctypes.data_as(PI), n1.ctypes.data_as(PI))
            assert st == 0, st
            res[name] = (cnt.copy(), bound.copy(), n1.copy())
        for p in range(P):
            F, tub = recs[p]
            band = tub - THR
            r = [exact_residual(F, q) for q in pts[p]]
            need_u = sum(1 for v in r if v < Fr(THR) + Fr(band))                 # what an upper bound must reach
            cap_l = sum(1 for v in r if v < Fr(THR) - Fr(band))                  # what a lower bound may reach
            # the construction bites: evaluated in plain binary32 some of the crafted matches fall on the wrong side
            Ff, pf = F.astype(np.float32), pts[p].astype(np.float32)
            r32 = np.abs(pf[:, 2] * (Ff[0] * pf[:, 0] + Ff[1] * pf[:, 1] + Ff[2]) + pf[:, 3] * (Ff[3] * pf[:, 0] + Ff[4] * pf[:, 1] + Ff[5])
                         + (Ff[6] * pf[:, 0] + Ff[7] * pf[:, 1] + Ff[8])).astype(np.float64)
            rex = np.array([float(v) for v in r])
            wrong_side = int(((rex < THR + band) & (r32 >= THR + band)).sum() + ((rex >= THR - band) & (r32 < THR - band)).sum())
            row = dict(trial=trial, pair=p, hyp=int(keep[p]), M=len(r), band=band, crafted_upper=crafted[p][0],
                       crafted_lower=crafted[p][1], need_upper=need_u, cap_lower=cap_l, plain_f32_wrong_side=wrong_side)
            for name in ("count32", "mfma"):
                cnt, bound, n1 = res[name]
                row[name] = dict(U=int(cnt[p]), L=int(bound[p]), n1=int(n1[p]))
                if not (cnt[p] >= need_u and bound[p] <= cap_l):
                    ok = False
            rows.append(row)
        for p in range(P):
            saved["pts_%d_%d" % (trial, p)] = pts[p]
        saved["keep_%d" % trial] = keep.copy()
        # restore the natural points for the next trial
        b.run(prm)
        b.sync()
    b.close()
    if os.environ.get("MVS_CRAFTED_OUT"):
        np.savez(os.environ["MVS_CRAFTED_OUT"], trials=4, pairs=P, capacity=N, hypotheses=H, thr=THR, seed=synth.SEED_BASE,
                 global_index=np.asarray(data["global_index"][:P], dtype=np.int64), **saved)
    return dict(ok=bool(ok), cases=len(rows), crafted=int(sum(r["crafted_upper"] + r["crafted_lower"] for r in rows)),
                plain_f32_wrong_side=int(sum(r["plain_f32_wrong_side"] for r in rows)),
                min_upper_margin=int(min(min(r["count32"]["U"], r["mfma"]["U"]) - r["need_upper"] for r in rows)),
                min_lower_margin=int(min(r["cap_lower"] - max(r["count32"]["L"], r["mfma"]["L"]) for r in rows)),
                pilot_hyps=int(sum(r["hyp"] < 256 for r in rows)), rows=rows)


def main():
    rng = np.random.default_rng(2024)
    ctx = capi.Context(0)
    lib = capi.lib()
    for f in ("mvs_debug_rounding_probe", "mvs_debug_indicator_probe", "mvs_debug_count_only", "mvs_debug_get_points",
              "mvs_debug_set_points", "mvs_debug_read_hyp_rec"):
        getattr(lib, f).restype = C.c_int
    out = dict(rounding=check_rounding(ctx, lib, rng), indicator=check_indicator(ctx, lib, rng),
               band_edges=check_band_edges(ctx, lib, rng))
    ctx.close()
    out["ok"] = all(out[k]["ok"] for k in ("rounding", "indicator", "band_edges"))
    print(json.dumps(out))
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
