"""ctypes binding of the CPU oracle (oracle/liboracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package (mvslam_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

SAMPLER_IDENTITY = 0
SAMPLER_PHILOX = 1


class Match(C.Structure):
    _fields_ = [("queryIdx", C.c_int32), ("trainIdx", C.c_int32), ("imgIdx", C.c_int32), ("distance", C.c_float)]


MATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])


class Params(C.Structure):
    _fields_ = [
        ("max_error_sq", C.c_double),
        ("num_hypotheses", C.c_int32),
        ("sampler", C.c_int32),
        ("seed", C.c_uint64),
        ("min_inliers", C.c_int32),
    ]


class TwoViewResult(C.Structure):
    _fields_ = [
        ("valid", C.c_int32),
        ("n_matches", C.c_int32),
        ("n_inliers", C.c_int32),
        ("n_points", C.c_int32),
        ("best_hyp", C.c_int32),
        ("best_count", C.c_int32),
        ("best_residual", C.c_double),
        ("F", C.c_double * 9),
        ("E", C.c_double * 9),
        ("R1to2", C.c_double * 9),
        ("t1to2", C.c_double * 3),
        ("R", C.c_double * 9),
        ("t", C.c_double * 3),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "rotations9", "pairs9", "rotations3", "pairs3", "rotations4", "pairs4", "hypotheses", "score_evals")]


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("mvs_oracle.c", "mvs_refine_oracle.c", "mvs_orb_oracle.c", "mvs_oracle.h")]
    stale = not os.path.exists(_LIB_PATH) or any(
        os.path.exists(f) and os.path.getmtime(_LIB_PATH) < os.path.getmtime(f) for f in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_match_visual_features.restype = C.c_int
        _lib.orc_find_fundamental_matrix.restype = C.c_int
        _lib.orc_ransac_fundamental.restype = C.c_int
        _lib.orc_count_inliers.restype = C.c_int
        _lib.orc_triangulate_points.restype = C.c_int
        _lib.orc_recover_pose_and_points.restype = C.c_int
        _lib.orc_sfm_solve.restype = C.c_int
        _lib.orc_sfm_triangulate.restype = C.c_int
        _lib.orc_image_pair.restype = C.c_int
        _lib.orc_project_point.restype = C.c_int
    return _lib


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def make_params(num_hypotheses=1, sampler=SAMPLER_IDENTITY, seed=0, max_error_sq=0.0, min_inliers=8):
    return Params(float(max_error_sq), int(num_hypotheses), int(sampler), int(seed), int(min_inliers))


# ---- Lie group -------------------------------------------------------------
def so3_rectify(R):
    R = _f64(R, (3, 3)).copy()
    lib().orc_so3_rectify(_p(R))
    return R


def so3_from_rpy(roll, pitch, yaw):
    R = np.empty((3, 3))
    lib().orc_so3_from_rpy(C.c_double(roll), C.c_double(pitch), C.c_double(yaw), _p(R))
    return R


def so3_ln(R):
    w = np.empty(3)
    lib().orc_so3_ln(_p(_f64(R, (3, 3))), _p(w))
    return w


def rodrigues(v):
    R = np.empty((3, 3))
    lib().orc_rodrigues(_p(_f64(v, (3,))), _p(R))
    return R


def se3_inverse(R, t):
    Ro, to = np.empty((3, 3)), np.empty(3)
    lib().orc_se3_inverse(_p(_f64(R, (3, 3))), _p(_f64(t, (3,))), _p(Ro), _p(to))
    return Ro, to


def se3_compose(Ra, ta, Rb, tb):
    Ro, to = np.empty((3, 3)), np.empty(3)
    lib().orc_se3_compose(_p(_f64(Ra, (3, 3))), _p(_f64(ta, (3,))), _p(_f64(Rb, (3, 3))), _p(_f64(tb, (3,))),
                          _p(Ro), _p(to))
    return Ro, to


def se3_ln(R, t):
    out = np.empty(6)
    lib().orc_se3_ln(_p(_f64(R, (3, 3))), _p(_f64(t, (3,))), _p(out))
    return out


def se3_exp(se3):
    R, t = np.empty((3, 3)), np.empty(3)
    lib().orc_se3_exp(_p(_f64(se3, (6,))), _p(R), _p(t))
    return R, t


# ---- SVD ---------------------------------------------------------------------
def svd(A):
    """cv::SVDecomp(A, w, u, vt, MODIFY_A | FULL_UV) -> (w, u, vt)."""
    A = _f64(A)
    m, n = A.shape
    w, u, vt = np.empty(min(m, n)), np.empty((m, m)), np.empty((n, n))
    lib().orc_svd(_p(A), C.c_int(m), C.c_int(n), _p(w), _p(u), _p(vt))
    return w, u, vt


# ---- camera --------------------------------------------------------------------
def mat3_inverse(K):
    out = np.empty((3, 3))
    lib().orc_mat3_inverse(_p(_f64(K, (3, 3))), _p(out))
    return out


def normalize_points(K, uv):
    uv = _f64(uv).reshape(-1, 2)
    out = np.empty_like(uv)
    Kinv = mat3_inverse(K)
    lib().orc_normalize_points(_p(Kinv), _p(uv), C.c_int(len(uv)), _p(out))
    return out


def project_points(K, Rw2c, tw2c, X):
    X = _f64(X).reshape(-1, 3)
    out = np.empty((len(X), 2))
    K, Rw2c, tw2c = _f64(K, (3, 3)), _f64(Rw2c, (3, 3)), _f64(tw2c, (3,))
    for i in range(len(X)):
        uv = np.empty(2)
        ok = lib().orc_project_point(_p(K), _p(Rw2c), _p(tw2c), _p(X[i].copy()), _p(uv))
        assert ok, "point behind camera"
        out[i] = uv
    return out


# ---- matcher -------------------------------------------------------------------
def match_visual_features(train_desc, query_desc, ratio=0.7, max_dist=-1.0):
    train_desc = np.ascontiguousarray(train_desc, dtype=np.uint8)
    query_desc = np.ascontiguousarray(query_desc, dtype=np.uint8)
    nq = query_desc.shape[0]
    out = np.zeros(max(nq, 1), dtype=MATCH_DTYPE)
    n = lib().orc_match_visual_features(
        _p(train_desc, C.c_uint8), C.c_int(train_desc.shape[0]), _p(query_desc, C.c_uint8), C.c_int(nq),
        C.c_int(train_desc.shape[1] if train_desc.ndim == 2 else 0), C.c_double(ratio), C.c_double(max_dist),
        out.ctypes.data_as(C.POINTER(Match)))
    if n < 0:
        return None
    return out[:n].copy()


# ---- 8-point / RANSAC ------------------------------------------------------------
def find_fundamental_matrix(p1, p2):
    p1, p2 = _f64(p1, (8, 2)), _f64(p2, (8, 2))
    F = np.empty((3, 3))
    ok = lib().orc_find_fundamental_matrix(_p(p1), _p(p2), _p(F))
    return bool(ok), F


def philox4x32_10(ctr, key):
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.empty(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(ctr, C.c_uint32), _p(key, C.c_uint32), _p(out, C.c_uint32))
    return out


def sample8(seed, hyp, M, sampler=SAMPLER_PHILOX):
    idx = np.empty(8, dtype=np.int32)
    lib().orc_sample8(C.c_uint64(seed), C.c_uint32(hyp), C.c_int(M), C.c_int(sampler), _p(idx, C.c_int))
    return idx


def count_inliers(p1, p2, F, max_error_sq):
    p1, p2 = _f64(p1).reshape(-1, 2), _f64(p2).reshape(-1, 2)
    M = len(p1)
    mask = np.zeros(M, dtype=np.uint8)
    res = C.c_double(0)
    n = lib().orc_count_inliers(_p(p1), _p(p2), C.c_int(M), _p(_f64(F, (3, 3))), C.c_double(max_error_sq),
                                _p(mask, C.c_uint8), C.byref(res))
    return n, res.value, mask


def ransac_fundamental(p1, p2, max_error_sq, H, sampler=SAMPLER_PHILOX, seed=0, per_hyp=False):
    p1, p2 = _f64(p1).reshape(-1, 2), _f64(p2).reshape(-1, 2)
    M = len(p1)
    F = np.zeros((3, 3))
    mask = np.zeros(max(M, 1), dtype=np.uint8)
    bh, bc, br = C.c_int(-1), C.c_int(0), C.c_double(0)
    cnt = np.zeros(H, dtype=np.int32) if per_hyp else None
    res = np.zeros(H, dtype=np.float64) if per_hyp else None
    ok = lib().orc_ransac_fundamental(
        _p(p1), _p(p2), C.c_int(M), C.c_double(max_error_sq), C.c_int(H), C.c_int(sampler), C.c_uint64(seed),
        _p(F), _p(mask, C.c_uint8), C.byref(bh), C.byref(bc), C.byref(br),
        _p(cnt, C.c_int32) if per_hyp else None, _p(res) if per_hyp else None)
    out = dict(ok=bool(ok), F=F, mask=mask[:M], best_hyp=bh.value, best_count=bc.value, best_residual=br.value)
    if per_hyp:
        out["count"] = cnt
        out["residual"] = res
    return out


# ---- sfm-solve -------------------------------------------------------------------
def project_essential(F):
    E = np.empty((3, 3))
    lib().orc_project_essential(_p(_f64(F, (3, 3))), _p(E))
    return E


def decompose_essential(E):
    Ra, Rb, t = np.empty((3, 3)), np.empty((3, 3)), np.empty(3)
    lib().orc_decompose_essential(_p(_f64(E, (3, 3))), _p(Ra), _p(Rb), _p(t))
    return Ra, Rb, t


def triangulate_points(R, t, p1, p2, mask=None):
    p1, p2 = _f64(p1).reshape(-1, 2), _f64(p2).reshape(-1, 2)
    M = len(p1)
    pts = np.empty((max(M, 1), 3))
    idx = np.empty(max(M, 1), dtype=np.int64)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    n = lib().orc_triangulate_points(_p(_f64(R, (3, 3))), _p(_f64(t, (3,))), _p(p1), _p(p2), C.c_int(M),
                                     _p(m, C.c_uint8) if m is not None else None, _p(pts), _p(idx, C.c_int64))
    return pts[:n].copy(), idx[:n].copy()


def recover_pose_and_points(E, p1, p2, mask=None):
    p1, p2 = _f64(p1).reshape(-1, 2), _f64(p2).reshape(-1, 2)
    M = len(p1)
    pts = np.empty((max(M, 1), 3))
    idx = np.empty(max(M, 1), dtype=np.int64)
    m = np.ones(M, dtype=np.uint8) if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    R, t = np.zeros((3, 3)), np.zeros(3)
    n = C.c_int(0)
    ok = lib().orc_recover_pose_and_points(_p(_f64(E, (3, 3))), _p(p1), _p(p2), C.c_int(M), _p(m, C.c_uint8),
                                           _p(R), _p(t), _p(pts), _p(idx, C.c_int64), C.byref(n))
    return bool(ok), R, t, pts[:n.value].copy(), idx[:n.value].copy()


def _result_dict(res, mask, pts, idx, M):
    n = res.n_points
    return dict(
        valid=bool(res.valid), n_matches=res.n_matches, n_inliers=res.n_inliers, n_points=n,
        best_hyp=res.best_hyp, best_count=res.best_count, best_residual=res.best_residual,
        F=np.array(res.F).reshape(3, 3), E=np.array(res.E).reshape(3, 3),
        R1to2=np.array(res.R1to2).reshape(3, 3), t1to2=np.array(res.t1to2),
        R=np.array(res.R).reshape(3, 3), t=np.array(res.t),
        mask=mask[:M].copy(), points=pts[:n].copy(), point_idx=idx[:n].copy())


def sfm_solve(uv1, uv2, K, params):
    uv1, uv2 = _f64(uv1).reshape(-1, 2), _f64(uv2).reshape(-1, 2)
    M = len(uv1)
    res = TwoViewResult()
    mask = np.zeros(max(M, 1), dtype=np.uint8)
    pts = np.zeros((max(M, 1), 3))
    idx = np.zeros(max(M, 1), dtype=np.int64)
    ok = lib().orc_sfm_solve(_p(uv1), _p(uv2), C.c_int(M), _p(_f64(K, (3, 3))), C.byref(params), C.byref(res),
                             _p(mask, C.c_uint8), _p(pts), _p(idx, C.c_int64))
    out = _result_dict(res, mask, pts, idx, M)
    out["ok"] = bool(ok)
    return out


def sfm_triangulate(uv1, uv2, K, pose1, pose2):
    uv1, uv2 = _f64(uv1).reshape(-1, 2), _f64(uv2).reshape(-1, 2)
    M = len(uv1)
    pts = np.zeros((max(M, 1), 3))
    idx = np.zeros(max(M, 1), dtype=np.int64)
    n = lib().orc_sfm_triangulate(_p(uv1), _p(uv2), C.c_int(M), _p(_f64(K, (3, 3))),
                                  _p(_f64(pose1[0], (3, 3))), _p(_f64(pose1[1], (3,))),
                                  _p(_f64(pose2[0], (3, 3))), _p(_f64(pose2[1], (3,))), _p(pts),
                                  _p(idx, C.c_int64))
    return pts[:n].copy(), idx[:n].copy()


def image_pair(base_desc, base_kp, pair_desc, pair_kp, K, params, ratio=0.7, max_dist=10.0):
    base_desc = np.ascontiguousarray(base_desc, dtype=np.uint8)
    pair_desc = np.ascontiguousarray(pair_desc, dtype=np.uint8)
    base_kp = np.ascontiguousarray(base_kp, dtype=np.float32).reshape(-1, 2)
    pair_kp = np.ascontiguousarray(pair_kp, dtype=np.float32).reshape(-1, 2)
    nb, npair = base_desc.shape[0], pair_desc.shape[0]
    matches = np.zeros(max(npair, 1), dtype=MATCH_DTYPE)
    res = TwoViewResult()
    mask = np.zeros(max(npair, 1), dtype=np.uint8)
    pts = np.zeros((max(npair, 1), 3))
    idx = np.zeros(max(npair, 1), dtype=np.int64)
    ok = lib().orc_image_pair(
        _p(base_desc, C.c_uint8), _p(base_kp, C.c_float), C.c_int(nb), _p(pair_desc, C.c_uint8),
        _p(pair_kp, C.c_float), C.c_int(npair), C.c_int(base_desc.shape[1]), C.c_double(ratio),
        C.c_double(max_dist), _p(_f64(K, (3, 3))), C.byref(params), matches.ctypes.data_as(C.POINTER(Match)),
        C.byref(res), _p(mask, C.c_uint8), _p(pts), _p(idx, C.c_int64))
    M = res.n_matches
    out = _result_dict(res, mask, pts, idx, M)
    out["ok"] = bool(ok)
    out["matches"] = matches[:M].copy()
    return out


def counters_reset():
    lib().orc_counters_reset()


def counters_get():
    c = Counters()
    lib().orc_counters_get(C.byref(c))
    return {n: getattr(c, n) for n, _ in Counters._fields_}


# ---- pnp-solve (row f1) ----------------------------------------------------------
class PnpParams(C.Structure):
    _fields_ = [("num_hypotheses", C.c_int32), ("sampler", C.c_int32), ("seed", C.c_uint64),
                ("reproj_error", C.c_double), ("min_inliers", C.c_int32)]


def make_pnp_params(num_hypotheses=100, sampler=SAMPLER_PHILOX, seed=0, reproj_error=0.05, min_inliers=4):
    return PnpParams(int(num_hypotheses), int(sampler), int(seed), float(reproj_error), int(min_inliers))


def sample4(seed, hyp, n, sampler=SAMPLER_PHILOX):
    idx = np.empty(4, dtype=np.int32)
    lib().orc_sample4(C.c_uint64(seed), C.c_uint32(hyp), C.c_int(n), C.c_int(sampler), _p(idx, C.c_int))
    return idx


def p3p(f, X):
    f, X = _f64(f, (3, 3)), _f64(X, (3, 3))
    R, t = np.zeros((4, 3, 3)), np.zeros((4, 3))
    lib().orc_p3p.restype = C.c_int
    n = lib().orc_p3p(_p(f), _p(X), _p(R), _p(t))
    return R[:n].copy(), t[:n].copy()


def pnp_solve(world_xyz, image_uv, K, params):
    X, uv = _f64(world_xyz).reshape(-1, 3), _f64(image_uv).reshape(-1, 2)
    n = len(X)
    R, t, Rw, tw = np.zeros((3, 3)), np.zeros(3), np.zeros((3, 3)), np.zeros(3)
    idx = np.zeros(max(n, 1), dtype=np.int64)
    ni, bh = C.c_int(0), C.c_int(-1)
    lib().orc_pnp_solve.restype = C.c_int
    ok = lib().orc_pnp_solve(_p(X), _p(uv), C.c_int(n), _p(_f64(K, (3, 3))), C.byref(params), _p(R), _p(t),
                             _p(idx, C.c_int64), C.byref(ni), _p(Rw), _p(tw), C.byref(bh))
    return dict(ok=bool(ok), R=R, t=t, Rw2c=Rw, tw2c=tw, inliers=idx[:ni.value].copy(), best_hyp=bh.value)


def seq_chain(pair_R, pair_t, pair_valid, track_R, track_t, track_ok):
    pR, pt = _f64(pair_R).reshape(-1, 9), _f64(pair_t).reshape(-1, 3)
    tR, tt = _f64(track_R).reshape(-1, 9), _f64(track_t).reshape(-1, 3)
    F = len(pR) + 1
    pv = np.ascontiguousarray(pair_valid, dtype=np.int32)
    ok = np.ascontiguousarray(track_ok, dtype=np.int32)
    R, t, ps, ts = np.zeros((F, 3, 3)), np.zeros((F, 3)), np.zeros(F - 1), np.zeros(max(F - 2, 1))
    lib().orc_seq_chain(C.c_int(F), _p(pR), _p(pt), _p(pv, C.c_int32), _p(tR), _p(tt), _p(ok, C.c_int32), _p(R), _p(t), _p(ps),
                        _p(ts))
    return dict(R=R, t=t, pair_scale=ps, track_scale=ts[:F - 2])


# ---- sfm-refine / pnp-refine (row f4) ----------------------------------------------
class RefineParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("reserved", C.c_int32), ("lambda_initial", C.c_double),
                ("lambda_factor", C.c_double), ("lambda_upper", C.c_double), ("rel_tol", C.c_double),
                ("abs_tol", C.c_double), ("anchor_sigma", C.c_double * 2), ("pose_sigma", C.c_double * 2),
                ("point_sigma", C.c_double)]


def make_refine_params(**kw):
    p = RefineParams()
    lib().orc_refine_params_default(C.byref(p))
    for k, v in kw.items():
        if k in ("anchor_sigma", "pose_sigma"):
            getattr(p, k)[0], getattr(p, k)[1] = float(v[0]), float(v[1])
        else:
            setattr(p, k, v)
    return p


def _opt(a, shape):
    return (None, None) if a is None else (lambda x: (x, _p(x)))(_f64(a, shape))


def sfm_refine(p1, cov1, p2, cov2, K, R_guess, t_guess, points_guess, params=None):
    p1, p2, pg = _f64(p1).reshape(-1, 2), _f64(p2).reshape(-1, 2), _f64(points_guess).reshape(-1, 3)
    m = len(p1)
    params = params or make_refine_params()
    c1, c1p = _opt(cov1, (m, 4))
    c2, c2p = _opt(cov2, (m, 4))
    R, t, pc = np.zeros((3, 3)), np.zeros(3), np.zeros((6, 6))
    pts, ptc = np.zeros((m, 3)), np.zeros((m, 3, 3))
    err, it = C.c_double(0), C.c_int(0)
    lib().orc_sfm_refine.restype = C.c_int
    ok = lib().orc_sfm_refine(_p(p1), c1p, _p(p2), c2p, C.c_int(m), _p(_f64(K, (3, 3))), _p(_f64(R_guess, (3, 3))),
                              _p(_f64(t_guess, (3,))), _p(pg), C.byref(params), _p(R), _p(t), _p(pc), _p(pts), _p(ptc),
                              C.byref(err), C.byref(it))
    return dict(ok=bool(ok), R=R, t=t, pose_cov=pc, points=pts, point_cov=ptc, error=err.value, iterations=it.value)


def pnp_refine(world, world_cov, img, img_cov, K, R_guess, t_guess, params=None):
    X, uv = _f64(world).reshape(-1, 3), _f64(img).reshape(-1, 2)
    m = len(X)
    params = params or make_refine_params()
    wc = _f64(world_cov, (m, 9))
    ic, icp = _opt(img_cov, (m, 4))
    R, t, pc = np.zeros((3, 3)), np.zeros(3), np.zeros((6, 6))
    err, it = C.c_double(0), C.c_int(0)
    lib().orc_pnp_refine.restype = C.c_int
    ok = lib().orc_pnp_refine(_p(X), _p(wc), _p(uv), icp, C.c_int(m), _p(_f64(K, (3, 3))), _p(_f64(R_guess, (3, 3))),
                              _p(_f64(t_guess, (3,))), C.byref(params), _p(R), _p(t), _p(pc), C.byref(err), C.byref(it))
    return dict(ok=bool(ok), R=R, t=t, pose_cov=pc, error=err.value, iterations=it.value)


def ba_refine(K, frame_pose, frame_prior_var, points, point_prior_cov, obs, obs_cov, obs_valid, params=None):
    """frame_pose [F, 12], frame_prior_var [F, 6], obs / obs_cov / obs_valid: lists of length F (entries may be None)"""
    fp, fv, pg = _f64(frame_pose).reshape(-1, 12), _f64(frame_prior_var).reshape(-1, 6), _f64(points).reshape(-1, 3)
    F, m = len(fp), len(pg)
    params = params or make_refine_params()
    keep = []

    def arr(a, shape, t=C.c_double, dt=np.float64):
        if a is None:
            return None
        x = np.ascontiguousarray(a, dtype=dt).reshape(shape)
        keep.append(x)
        return x.ctypes.data_as(C.POINTER(t))
    P2 = C.POINTER(C.c_double) * 2
    PV = C.POINTER(C.c_uint8) * 2
    o_, c_, v_ = P2(), P2(), PV()
    for f in range(F):
        o_[f] = arr(obs[f], (m, 2))
        c_[f] = arr(obs_cov[f], (m, 4)) if obs_cov[f] is not None else None
        v_[f] = arr(obs_valid[f], (m,), C.c_uint8, np.uint8) if obs_valid[f] is not None else None
    R, t, pc = np.zeros((F, 3, 3)), np.zeros((F, 3)), np.zeros((F, 6, 6))
    pts, ptc = np.zeros((m, 3)), np.zeros((m, 3, 3))
    err, it = C.c_double(0), C.c_int(0)
    lib().orc_ba_refine.restype = C.c_int
    ok = lib().orc_ba_refine(C.c_int(F), C.c_int(m), _p(_f64(K, (3, 3))), _p(fp), _p(fv), _p(pg),
                             arr(point_prior_cov, (m, 9)), o_, c_, v_, C.byref(params), _p(R), _p(t), _p(pc), _p(pts),
                             _p(ptc), C.byref(err), C.byref(it))
    return dict(ok=bool(ok), R=R, t=t, pose_cov=pc, points=pts, point_cov=ptc, error=err.value, iterations=it.value)


# ---- ORB-style extraction (row f3) ---------------------------------------------------
class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("nlevels", C.c_int32), ("edge_threshold", C.c_int32),
                ("fast_threshold", C.c_int32)]


KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])


def make_orb_params(nfeatures=500, nlevels=8, edge_threshold=31, fast_threshold=20):
    return OrbParams(int(nfeatures), int(nlevels), int(edge_threshold), int(fast_threshold))


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def orb_pattern():
    P = np.zeros((256, 4), dtype=np.int8)
    lib().orc_orb_pattern(P.ctypes.data_as(C.c_void_p))
    return P


def orb_layout(w, h, params):
    L = params.nlevels
    lw, lh, nl = (np.zeros(L, dtype=np.int32) for _ in range(3))
    sc = np.zeros(L)
    lib().orc_orb_layout.restype = C.c_int
    ok = lib().orc_orb_layout(C.c_int(w), C.c_int(h), C.byref(params), _p(lw, C.c_int), _p(lh, C.c_int), _p(nl, C.c_int),
                              _p(sc))
    return bool(ok), lw, lh, nl, sc


def orb_resize(src, dw, dh):
    src = _u8(src)
    dst = np.zeros((dh, dw), dtype=np.uint8)
    lib().orc_orb_resize(_p(src, C.c_uint8), C.c_int(src.shape[1]), C.c_int(src.shape[0]), _p(dst, C.c_uint8),
                         C.c_int(dw), C.c_int(dh))
    return dst


def orb_fast_scores(img, threshold):
    img = _u8(img)
    out = np.zeros_like(img)
    lib().orc_orb_fast_scores(_p(img, C.c_uint8), C.c_int(img.shape[1]), C.c_int(img.shape[0]), C.c_int(threshold),
                              _p(out, C.c_uint8))
    return out


def orb_blur(img):
    img = _u8(img)
    out = np.zeros_like(img)
    lib().orc_orb_blur(_p(img, C.c_uint8), C.c_int(img.shape[1]), C.c_int(img.shape[0]), _p(out, C.c_uint8))
    return out


def orb_harris(img, x, y):
    img = _u8(img)
    lib().orc_orb_harris.restype = C.c_float
    return float(lib().orc_orb_harris(_p(img, C.c_uint8), C.c_int(img.shape[1]), C.c_int(x), C.c_int(y)))


def orb_moments(img, x, y):
    img = _u8(img)
    a, b = C.c_int(0), C.c_int(0)
    lib().orc_orb_moments(_p(img, C.c_uint8), C.c_int(img.shape[1]), C.c_int(x), C.c_int(y), C.byref(a), C.byref(b))
    return a.value, b.value


def orb_fast_atan2(y, x):
    lib().orc_orb_fast_atan2.restype = C.c_float
    return float(lib().orc_orb_fast_atan2(C.c_float(y), C.c_float(x)))


def orb_extract(img, params=None):
    img = _u8(img)
    params = params or make_orb_params()
    kp = np.zeros(params.nfeatures, dtype=KEYPOINT_DTYPE)
    desc = np.zeros((params.nfeatures, 32), dtype=np.uint8)
    n = C.c_int(0)
    lib().orc_orb_extract.restype = C.c_int
    ok = lib().orc_orb_extract(_p(img, C.c_uint8), C.c_int(img.shape[1]), C.c_int(img.shape[0]), C.byref(params),
                               kp.ctypes.data_as(C.c_void_p), _p(desc, C.c_uint8), C.byref(n))
    return dict(ok=bool(ok), kp=kp[:n.value].copy(), desc=desc[:n.value].copy())
