"""ctypes binding of libmvslam_hip.so (include/mvslam_hip.h).

Plumbing only: it forwards numpy buffers to the C ABI and never computes anything itself.
There is no CPU fallback -- if the HIP library is missing or no device is present the
calls raise.
"""
import ctypes as C
import os
import weakref

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libmvslam_hip.so")
# diagnostics build with the mvs_debug_* hooks (make -C mvslam_amd/csrc): never loaded by this package unless a tool asks
# for it explicitly with MVS_USE_DEBUG_LIB=1 (tools/ab_ransac.py) -- tests load it side by side through ctypes
DBG_LIB_PATH = os.path.join(_PKG, "lib", "libmvslam_hip_dbg.so")
if os.environ.get("MVS_USE_DEBUG_LIB") == "1":
    LIB_PATH = DBG_LIB_PATH

MVS_OK = 0
MVS_NO_MODEL = 1
MVS_ERR_INVALID_ARG, MVS_ERR_NO_DEVICE, MVS_ERR_HIP, MVS_ERR_CAPACITY, MVS_ERR_BAD_INTRINSICS = -1, -2, -3, -4, -5
SAMPLER_IDENTITY = 0
SAMPLER_PHILOX = 1

MATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])


class Params(C.Structure):
    _fields_ = [
        ("ratio", C.c_double),
        ("max_dist", C.c_double),
        ("max_error_sq", C.c_double),
        ("num_hypotheses", C.c_int32),
        ("sampler", C.c_int32),
        ("seed", C.c_uint64),
        ("min_inliers", C.c_int32),
        ("reserved", C.c_int32),
    ]


class PairResult(C.Structure):
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


RESULT_DTYPE = np.dtype([
    ("valid", "<i4"), ("n_matches", "<i4"), ("n_inliers", "<i4"), ("n_points", "<i4"), ("best_hyp", "<i4"),
    ("best_count", "<i4"), ("best_residual", "<f8"), ("F", "<f8", (3, 3)), ("E", "<f8", (3, 3)),
    ("R1to2", "<f8", (3, 3)), ("t1to2", "<f8", (3,)), ("R", "<f8", (3, 3)), ("t", "<f8", (3,))])
assert RESULT_DTYPE.itemsize == C.sizeof(PairResult)


class PnpParams(C.Structure):
    _fields_ = [("num_hypotheses", C.c_int32), ("sampler", C.c_int32), ("seed", C.c_uint64),
                ("reproj_error", C.c_double), ("min_inliers", C.c_int32), ("refit", C.c_int32)]


TRACK_DTYPE = np.dtype([("ok", "<i4"), ("n_corr", "<i4"), ("n_inliers", "<i4"), ("best_hyp", "<i4"),
                        ("R", "<f8", (3, 3)), ("t", "<f8", (3,))])


class RefineParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("reserved", C.c_int32), ("lambda_initial", C.c_double),
                ("lambda_factor", C.c_double), ("lambda_upper", C.c_double), ("rel_tol", C.c_double),
                ("abs_tol", C.c_double), ("anchor_sigma", C.c_double * 2), ("pose_sigma", C.c_double * 2),
                ("point_sigma", C.c_double)]


class BaProblem(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_points", C.c_int32), ("K", C.POINTER(C.c_double)),
                ("frame_pose", C.POINTER(C.c_double)), ("frame_prior_var", C.POINTER(C.c_double)),
                ("points", C.POINTER(C.c_double)), ("point_prior_cov", C.POINTER(C.c_double)),
                ("obs", C.POINTER(C.c_double) * 2), ("obs_cov", C.POINTER(C.c_double) * 2),
                ("obs_valid", C.POINTER(C.c_uint8) * 2)]


REFINE_DTYPE = np.dtype([("ok", "<i4"), ("iterations", "<i4"), ("error", "<f8"), ("R", "<f8", (3, 3)),
                         ("t", "<f8", (3,)), ("pose_cov", "<f8", (6, 6))])


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("nlevels", C.c_int32), ("edge_threshold", C.c_int32),
                ("fast_threshold", C.c_int32)]


KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])   # cv::KeyPoint


class WorkStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("hypotheses", "rotations9", "pairs9", "score_evals", "matches", "inliers",
                                              "score_evals_executed", "score_evals_executed_f32", "exact_solves",
                                              "prescreened")] + [("pairs_mode", C.c_int64 * 3),
                                                                 ("score_evals_executed_mfma", C.c_int64),
                                                                 ("score_evals_executed_mfma_finish", C.c_int64),
                                                                 ("max_sweeps9", C.c_int64), ("dense_points", C.c_int64),
                                                                 ("matches_mode1", C.c_int64), ("score_evals_executed_mfma_rest", C.c_int64),
                                                                 ("score_evals_executed_mfma_pilot", C.c_int64)]


class KernelInfo(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("symbol", C.c_char * 160)] + [(n, C.c_int32) for n in (
        "kernel_id", "threads_per_block", "num_regs", "static_lds_bytes", "dynamic_lds_bytes", "scratch_bytes_per_lane",
        "max_threads_per_block", "blocks_per_cu", "waves_per_simd", "reserved")]


EXPORTS = [
    "mvs_abi_version", "mvs_status_str", "mvs_last_error", "mvs_params_default", "mvs_ctx_create",
    "mvs_ctx_create_on_stream", "mvs_ctx_destroy", "mvs_ctx_stream", "mvs_match_hamming", "mvs_two_view",
    "mvs_triangulate", "mvs_recover_pose", "mvs_find_fundamental_matrix", "mvs_ransac_fundamental",
    "mvs_batch_create", "mvs_batch_destroy", "mvs_batch_upload", "mvs_batch_run", "mvs_batch_sync",
    "mvs_batch_time", "mvs_batch_download", "mvs_batch_stats", "mvs_batch_results_device",
    "mvs_batch_copy_results_device", "mvs_pnp_params_default", "mvs_pnp_solve", "mvs_seq_create", "mvs_seq_destroy",
    "mvs_seq_upload", "mvs_seq_run", "mvs_seq_sync", "mvs_seq_time", "mvs_seq_download_pairs", "mvs_seq_download_tracks",
    "mvs_refine_params_default", "mvs_sfm_refine", "mvs_pnp_refine", "mvs_batch_refine", "mvs_batch_download_refined",
    "mvs_orb_params_default", "mvs_extract", "mvs_seq_upload_images", "mvs_seq_refine_pairs", "mvs_seq_download_refined",
    "mvs_ba_refine", "mvs_seq_download_trajectory", "mvs_batch_upload_octaves", "mvs_seq_upload_octaves",
    "mvs_batch_upload_async", "mvs_batch_download_async", "mvs_host_alloc", "mvs_host_free", "mvs_image_pair",
    "mvs_batch_gather_results", "mvs_seq_time_stages", "mvs_batch_time_kernels", "mvs_kernel_info_get",
    "mvs_extract_time", "mvs_ctx_set_half_batches", "mvs_batch_device_state", "mvs_batch_run_points",
]


_dbg = None


def dbg_lib():
    """The diagnostics library, loaded SIDE BY SIDE with the product one (its own handle, its own copy of every symbol): the
    tests let the product binary run the stage and hand its device state to this library's audit (mvs_debug_audit_state)."""
    global _dbg
    if _dbg is None:
        if not os.path.exists(DBG_LIB_PATH):
            raise RuntimeError("libmvslam_hip_dbg.so is missing (%s): make -C mvslam_amd/csrc" % DBG_LIB_PATH)
        _dbg = C.CDLL(DBG_LIB_PATH)
        _dbg.mvs_last_error.restype = C.c_char_p
        _dbg.mvs_last_error.argtypes = [C.c_void_p]
        _dbg.mvs_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        _dbg.mvs_ctx_destroy.argtypes = [C.c_void_p]
        _dbg.mvs_debug_audit_state.restype = C.c_int
        _dbg.mvs_debug_audit_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p]
    return _dbg


class MvsError(RuntimeError):
    def __init__(self, status, what):
        super().__init__("%s: status %d (%s)" % (what, status, status_str(status)))
        self.status = status


_lib = None


def lib():
    """Load the HIP library; raises (loudly) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libmvslam_hip.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C mvslam_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.mvs_status_str.restype = C.c_char_p
        _lib.mvs_last_error.restype = C.c_char_p
        _lib.mvs_last_error.argtypes = [C.c_void_p]
        _lib.mvs_ctx_stream.restype = C.c_void_p
        _lib.mvs_ctx_set_half_batches.restype = C.c_int
        _lib.mvs_ctx_set_half_batches.argtypes = [C.c_void_p, C.c_int]
        _lib.mvs_ctx_stream.argtypes = [C.c_void_p]
        _lib.mvs_ctx_destroy.argtypes = [C.c_void_p]
        _lib.mvs_batch_destroy.argtypes = [C.c_void_p]
        _lib.mvs_seq_destroy.argtypes = [C.c_void_p]
    return _lib


def status_str(status):
    return lib().mvs_status_str(C.c_int(status)).decode()


def default_params(**kw):
    p = Params()
    lib().mvs_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def default_pnp_params(**kw):
    p = PnpParams()
    lib().mvs_pnp_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def default_orb_params(**kw):
    p = OrbParams()
    lib().mvs_orb_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def default_refine_params(**kw):
    p = RefineParams()
    lib().mvs_refine_params_default(C.byref(p))
    for k, v in kw.items():
        if k in ("anchor_sigma", "pose_sigma"):
            getattr(p, k)[0], getattr(p, k)[1] = float(v[0]), float(v[1])
        else:
            setattr(p, k, v)
    return p


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a if shape is None else a.reshape(shape)


def pinned_empty(shape, dtype):
    """numpy array over pinned (page-locked) host memory from mvs_host_alloc: the asynchronous transfers of
    Batch.upload_async / download_async are true DMA copies on such buffers.  Free with pinned_free()."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    p = C.c_void_p()
    st = lib().mvs_host_alloc(C.c_size_t(max(n, 1)), C.byref(p))
    if st != MVS_OK:
        raise MvsError(st, "mvs_host_alloc")
    buf = (C.c_char * max(n, 1)).from_address(p.value)
    a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    _PINNED[a.ctypes.data] = p
    return a


_PINNED = {}


def pinned_free(a):
    p = _PINNED.pop(a.ctypes.data, None)
    if p is not None:
        lib().mvs_host_free(p)


class Context:
    """One mvs_ctx: one HIP stream on one GPU."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        self._children = weakref.WeakSet()  # batches / sequences must be destroyed before their context
        st = lib().mvs_ctx_create_on_stream(C.c_int(device), C.c_void_p(stream), C.byref(self._h))
        if st != MVS_OK:
            raise MvsError(st, "mvs_ctx_create")

    def close(self):
        if self._h:
            for child in list(self._children):
                child.close()
            lib().mvs_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return lib().mvs_ctx_stream(self._h)

    def set_half_batches(self, enable):
        """Large batches as two halves on two streams (default) or every launch on the one stream."""
        self._check(lib().mvs_ctx_set_half_batches(self._h, C.c_int(1 if enable else 0)), "mvs_ctx_set_half_batches")

    def _check(self, st, what, allow_no_model=False):
        if st == MVS_OK or (allow_no_model and st == MVS_NO_MODEL):
            return st
        err = lib().mvs_last_error(self._h)
        raise MvsError(st, what + (" [" + err.decode() + "]" if err else ""))

    def kernel_info(self, max_kp=2000, desc_bytes=32):
        """the kernels of the two-view pipeline as launched for this batch shape: what the RUNTIME reports for the loaded
        code object (registers, LDS, scratch, occupancy), merged with the build's resource-usage digest
        (lib/kernel_resources.json: the VGPR / AGPR split) when that file is present.  {name: {...}}, launch order"""
        import json
        build = {}
        rpath = os.path.join(_PKG, "lib", "kernel_resources.json")
        if os.path.exists(rpath):
            build = json.load(open(rpath))
        out = {}
        i = 0
        while True:
            ki = KernelInfo()
            st = lib().mvs_kernel_info_get(self._h, C.c_int(i), C.c_int(max_kp), C.c_int(desc_bytes), C.byref(ki))
            if st == MVS_ERR_INVALID_ARG:
                break
            self._check(st, "mvs_kernel_info_get")
            d = {n: getattr(ki, n) for n, _ in KernelInfo._fields_[2:] if n != "reserved"}
            d["symbol"] = ki.symbol.decode()
            bres = build.get(d["symbol"])
            if bres:
                d["build"] = bres
            out[ki.name.decode()] = d
            i += 1
        return out

    # VisualFeature::match_visual_features(vf1 = train, vf2 = query, max_dist)
    def match_hamming(self, train_desc, query_desc, ratio=0.7, max_dist=-1.0):
        train_desc = np.ascontiguousarray(train_desc, dtype=np.uint8)
        query_desc = np.ascontiguousarray(query_desc, dtype=np.uint8)
        nq = int(query_desc.shape[0])
        out = np.zeros(max(nq, 1), dtype=MATCH_DTYPE)
        n = C.c_int(0)
        st = lib().mvs_match_hamming(
            self._h, _ptr(train_desc, C.c_uint8), C.c_int(int(train_desc.shape[0])), _ptr(query_desc, C.c_uint8),
            C.c_int(nq), C.c_int(int(train_desc.shape[1]) if train_desc.ndim == 2 else 0), C.c_double(ratio),
            C.c_double(max_dist), out.ctypes.data_as(C.c_void_p), C.byref(n))
        self._check(st, "mvs_match_hamming")
        return out[:n.value].copy()

    # ImagePair::ImagePair + reconstruct of one pair in one device pass
    def image_pair(self, base_desc, base_kp, pair_desc, pair_kp, K, params):
        base_desc = np.ascontiguousarray(base_desc, dtype=np.uint8)
        pair_desc = np.ascontiguousarray(pair_desc, dtype=np.uint8)
        base_kp = np.ascontiguousarray(base_kp, dtype=np.float32).reshape(-1, 2)
        pair_kp = np.ascontiguousarray(pair_kp, dtype=np.float32).reshape(-1, 2)
        n1, n2 = len(base_desc), len(pair_desc)
        res = PairResult()
        mt = np.zeros(max(n2, 1), dtype=MATCH_DTYPE)
        mask = np.zeros(max(n2, 1), dtype=np.uint8)
        pts = np.zeros((max(n2, 1), 3))
        idx = np.zeros(max(n2, 1), dtype=np.int64)
        st = lib().mvs_image_pair(self._h, _ptr(base_desc, C.c_uint8), _ptr(base_kp, C.c_float), C.c_int(n1),
                                  _ptr(pair_desc, C.c_uint8), _ptr(pair_kp, C.c_float), C.c_int(n2),
                                  C.c_int(int(base_desc.shape[1])), _ptr(_f64(K, (9,)), C.c_double), C.byref(params),
                                  C.byref(res), mt.ctypes.data_as(C.c_void_p), _ptr(mask, C.c_uint8),
                                  _ptr(pts, C.c_double), _ptr(idx, C.c_int64))
        self._check(st, "mvs_image_pair", allow_no_model=True)
        r = np.frombuffer(bytes(res), dtype=RESULT_DTYPE)[0]
        out = self._unpack(res, mask, pts, idx, int(r["n_matches"]))
        out["matches"] = mt[:int(r["n_matches"])].copy()
        out["ok"] = st == MVS_OK
        return out

    @staticmethod
    def _unpack(res, mask, pts, idx, m):
        r = np.frombuffer(bytes(res), dtype=RESULT_DTYPE)[0]
        out = {k: (r[k].copy() if isinstance(r[k], np.ndarray) else r[k].item()) for k in RESULT_DTYPE.names}
        out["valid"] = bool(out["valid"])
        n = out["n_points"] if out["valid"] else 0
        out["mask"] = None if mask is None else mask[:m].copy()
        out["points"] = pts[:n].copy()
        out["point_idx"] = idx[:n].copy()
        return out

    # sfm_solve(p1, p2, K, pose2in1, points, point_indexes)
    def two_view(self, uv1, uv2, K, params):
        uv1, uv2 = _f64(uv1).reshape(-1, 2), _f64(uv2).reshape(-1, 2)
        m = len(uv1)
        R, t = np.zeros(9), np.zeros(3)
        pts = np.zeros((max(m, 1), 3))
        idx = np.zeros(max(m, 1), dtype=np.int64)
        mask = np.zeros(max(m, 1), dtype=np.uint8)
        n = C.c_int(0)
        res = PairResult()
        st = lib().mvs_two_view(self._h, _ptr(uv1, C.c_double), _ptr(uv2, C.c_double), C.c_int(m),
                                _ptr(_f64(K, (9,)), C.c_double), C.byref(params), _ptr(R, C.c_double),
                                _ptr(t, C.c_double), _ptr(pts, C.c_double), _ptr(idx, C.c_int64), C.byref(n),
                                _ptr(mask, C.c_uint8), C.byref(res))
        self._check(st, "mvs_two_view", allow_no_model=True)
        out = self._unpack(res, mask, pts, idx, m)
        out["ok"] = st == MVS_OK
        return out

    # sfm_triangulate with T_1_to_2 composed by the caller
    def triangulate(self, uv1, uv2, K, R1to2, t1to2):
        uv1, uv2 = _f64(uv1).reshape(-1, 2), _f64(uv2).reshape(-1, 2)
        m = len(uv1)
        pts = np.zeros((max(m, 1), 3))
        idx = np.zeros(max(m, 1), dtype=np.int64)
        n = C.c_int(0)
        st = lib().mvs_triangulate(self._h, _ptr(uv1, C.c_double), _ptr(uv2, C.c_double), C.c_int(m),
                                   _ptr(_f64(K, (9,)), C.c_double), _ptr(_f64(R1to2, (9,)), C.c_double),
                                   _ptr(_f64(t1to2, (3,)), C.c_double), _ptr(pts, C.c_double), _ptr(idx, C.c_int64),
                                   C.byref(n))
        self._check(st, "mvs_triangulate")
        return pts[:n.value].copy(), idx[:n.value].copy()

    def recover_pose(self, E, uv1, uv2, K, mask=None):
        uv1, uv2 = _f64(uv1).reshape(-1, 2), _f64(uv2).reshape(-1, 2)
        m = len(uv1)
        R, t = np.zeros(9), np.zeros(3)
        pts = np.zeros((max(m, 1), 3))
        idx = np.zeros(max(m, 1), dtype=np.int64)
        n = C.c_int(0)
        res = PairResult()
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        st = lib().mvs_recover_pose(self._h, _ptr(_f64(E, (9,)), C.c_double), _ptr(uv1, C.c_double),
                                    _ptr(uv2, C.c_double), C.c_int(m), _ptr(_f64(K, (9,)), C.c_double),
                                    _ptr(mk, C.c_uint8), _ptr(R, C.c_double), _ptr(t, C.c_double),
                                    _ptr(pts, C.c_double), _ptr(idx, C.c_int64), C.byref(n), C.byref(res))
        self._check(st, "mvs_recover_pose", allow_no_model=True)
        out = self._unpack(res, None, pts, idx, m)
        out["ok"] = st == MVS_OK
        return out

    # pnp_solve(world_points, image_points, K, pose, inlier_point_indexes)
    def pnp_solve(self, world_xyz, image_uv, K, params):
        X, uv = _f64(world_xyz).reshape(-1, 3), _f64(image_uv).reshape(-1, 2)
        n = len(X)
        R, t = np.zeros((3, 3)), np.zeros(3)
        idx = np.zeros(max(n, 1), dtype=np.int64)
        ni, bh = C.c_int(0), C.c_int(-1)
        st = lib().mvs_pnp_solve(self._h, _ptr(X, C.c_double), _ptr(uv, C.c_double), C.c_int(n),
                                 _ptr(_f64(K, (9,)), C.c_double), C.byref(params), _ptr(R, C.c_double),
                                 _ptr(t, C.c_double), _ptr(idx, C.c_int64), C.byref(ni), C.byref(bh))
        self._check(st, "mvs_pnp_solve", allow_no_model=True)
        return dict(ok=st == MVS_OK, R=R, t=t, inliers=idx[:ni.value].copy(), best_hyp=bh.value)

    # VisualFeature::extract(image) for a stack of equally sized grayscale images [B, H, W]
    def extract(self, images, params=None, out=None):
        images = np.ascontiguousarray(images, dtype=np.uint8)
        if images.ndim == 2:
            images = images[None]
        B, H, W = images.shape
        params = params or default_orb_params()
        if out is None:
            kp = np.zeros((B, params.nfeatures), dtype=KEYPOINT_DTYPE)
            desc = np.zeros((B, params.nfeatures, 32), dtype=np.uint8)
            n = np.zeros(B, dtype=np.int32)
        else:   # caller-owned outputs (e.g. pinned_empty() arrays: the copies become DMA transfers)
            kp, desc, n = out["kp"], out["desc"], out["n"]
            assert kp.shape == (B, params.nfeatures) and desc.shape == (B, params.nfeatures, 32) and n.shape == (B,)
        st = lib().mvs_extract(self._h, _ptr(images, C.c_uint8), C.c_int(B), C.c_int(W), C.c_int(H), C.byref(params),
                               kp.ctypes.data_as(C.c_void_p), _ptr(desc, C.c_uint8), _ptr(n, C.c_int32))
        self._check(st, "mvs_extract")
        return dict(kp=kp, desc=desc, n=n)

    # sfm_refine(p1_estimates, p2_estimates, K, pose2in1_guess, pointsin1_guess, pose2in1_estimate, pointsin1_estimate, error)
    def extract_time(self, steps=10):
        """kernel ms of ONE replay of the last extraction's launches (HIP events on the ctx stream, no transfers)"""
        ms = C.c_float(0)
        self._check(lib().mvs_extract_time(self._h, C.c_int(steps), C.byref(ms)), "mvs_extract_time")
        return ms.value / steps

    def sfm_refine(self, p1, cov1, p2, cov2, K, R_guess, t_guess, points_guess, params=None, point_cov=True):
        p1, p2, pg = _f64(p1).reshape(-1, 2), _f64(p2).reshape(-1, 2), _f64(points_guess).reshape(-1, 3)
        m = len(p1)
        params = params or default_refine_params()
        c1 = None if cov1 is None else _f64(cov1, (m, 4))
        c2 = None if cov2 is None else _f64(cov2, (m, 4))
        res = np.zeros(1, dtype=REFINE_DTYPE)
        pts = np.zeros((m, 3))
        ptc = np.zeros((m, 3, 3)) if point_cov else None
        st = lib().mvs_sfm_refine(self._h, _ptr(p1, C.c_double), _ptr(c1, C.c_double), _ptr(p2, C.c_double),
                                  _ptr(c2, C.c_double), C.c_int(m), _ptr(_f64(K, (9,)), C.c_double),
                                  _ptr(_f64(R_guess, (9,)), C.c_double), _ptr(_f64(t_guess, (3,)), C.c_double),
                                  _ptr(pg, C.c_double), C.byref(params), res.ctypes.data_as(C.c_void_p),
                                  _ptr(pts, C.c_double), _ptr(ptc, C.c_double))
        self._check(st, "mvs_sfm_refine", allow_no_model=True)
        r = res[0]
        return dict(ok=st == MVS_OK, R=r["R"].copy(), t=r["t"].copy(), pose_cov=r["pose_cov"].copy(), points=pts,
                    point_cov=ptc, error=float(r["error"]), iterations=int(r["iterations"]))

    # pnp_refine(world_point_estimates, image_point_estimates, K, pose_guess, pose_estimate, error)
    def pnp_refine(self, world, world_cov, image, image_cov, K, R_guess, t_guess, params=None):
        X, uv = _f64(world).reshape(-1, 3), _f64(image).reshape(-1, 2)
        m = len(X)
        params = params or default_refine_params()
        wc = _f64(world_cov, (m, 9))
        ic = None if image_cov is None else _f64(image_cov, (m, 4))
        res = np.zeros(1, dtype=REFINE_DTYPE)
        st = lib().mvs_pnp_refine(self._h, _ptr(X, C.c_double), _ptr(wc, C.c_double), _ptr(uv, C.c_double),
                                  _ptr(ic, C.c_double), C.c_int(m), _ptr(_f64(K, (9,)), C.c_double),
                                  _ptr(_f64(R_guess, (9,)), C.c_double), _ptr(_f64(t_guess, (3,)), C.c_double),
                                  C.byref(params), res.ctypes.data_as(C.c_void_p))
        self._check(st, "mvs_pnp_refine", allow_no_model=True)
        r = res[0]
        return dict(ok=st == MVS_OK, R=r["R"].copy(), t=r["t"].copy(), pose_cov=r["pose_cov"].copy(),
                    error=float(r["error"]), iterations=int(r["iterations"]))

    # ba_frame_pose_and_point for one or two frames (sfm_refine, pnp_refine, VisualOdometer::track_refine)
    def ba_refine(self, K, frame_pose, frame_prior_var, points, point_prior_cov, obs, obs_cov, obs_valid, params=None):
        fp, fv, pg = _f64(frame_pose).reshape(-1, 12), _f64(frame_prior_var).reshape(-1, 6), _f64(points).reshape(-1, 3)
        F, m = len(fp), len(pg)
        params = params or default_refine_params()
        keep = [fp, fv, pg, _f64(K, (9,))]
        pb = BaProblem()
        pb.n_frames, pb.n_points = F, m
        pb.K, pb.frame_pose, pb.frame_prior_var, pb.points = (_ptr(keep[3], C.c_double), _ptr(fp, C.c_double),
                                                            _ptr(fv, C.c_double), _ptr(pg, C.c_double))
        if point_prior_cov is not None:
            keep.append(_f64(point_prior_cov, (m, 9)))
            pb.point_prior_cov = _ptr(keep[-1], C.c_double)
        for f in range(F):
            keep.append(_f64(obs[f], (m, 2)))
            pb.obs[f] = _ptr(keep[-1], C.c_double)
            if obs_cov[f] is not None:
                keep.append(_f64(obs_cov[f], (m, 4)))
                pb.obs_cov[f] = _ptr(keep[-1], C.c_double)
            if obs_valid[f] is not None:
                keep.append(np.ascontiguousarray(obs_valid[f], dtype=np.uint8).reshape(m))
                pb.obs_valid[f] = _ptr(keep[-1], C.c_uint8)
        res = np.zeros(F, dtype=REFINE_DTYPE)
        pts, ptc = np.zeros((m, 3)), np.zeros((m, 3, 3))
        st = lib().mvs_ba_refine(self._h, C.byref(pb), C.byref(params), res.ctypes.data_as(C.c_void_p), _ptr(pts, C.c_double),
                                 _ptr(ptc, C.c_double))
        self._check(st, "mvs_ba_refine", allow_no_model=True)
        return dict(ok=st == MVS_OK, R=res["R"].copy(), t=res["t"].copy(), pose_cov=res["pose_cov"].copy(), points=pts,
                    point_cov=ptc, error=float(res["error"][0]), iterations=int(res["iterations"][0]))

    def find_fundamental_matrix(self, p1, p2):
        p1, p2 = _f64(p1, (16,)), _f64(p2, (16,))
        F = np.zeros((3, 3))
        st = lib().mvs_find_fundamental_matrix(self._h, _ptr(p1, C.c_double), _ptr(p2, C.c_double),
                                               _ptr(F, C.c_double))
        self._check(st, "mvs_find_fundamental_matrix", allow_no_model=True)
        return st == MVS_OK, F

    def ransac_fundamental(self, p1, p2, max_error_sq, H, sampler=SAMPLER_PHILOX, seed=0, per_hyp=False):
        p1, p2 = _f64(p1).reshape(-1, 2), _f64(p2).reshape(-1, 2)
        m = len(p1)
        F = np.zeros((3, 3))
        mask = np.zeros(max(m, 1), dtype=np.uint8)
        bh, bc, br = C.c_int(-1), C.c_int(0), C.c_double(0)
        cnt = np.zeros(H, dtype=np.int32) if per_hyp else None
        res = np.zeros(H, dtype=np.float64) if per_hyp else None
        st = lib().mvs_ransac_fundamental(
            self._h, _ptr(p1, C.c_double), _ptr(p2, C.c_double), C.c_int(m), C.c_double(max_error_sq), C.c_int(H),
            C.c_int(sampler), C.c_uint64(seed), _ptr(F, C.c_double), _ptr(mask, C.c_uint8), C.byref(bh),
            C.byref(bc), C.byref(br), _ptr(cnt, C.c_int32), _ptr(res, C.c_double))
        self._check(st, "mvs_ransac_fundamental", allow_no_model=True)
        out = dict(ok=st == MVS_OK, F=F, mask=mask[:m], best_hyp=bh.value, best_count=bc.value,
                   best_residual=br.value)
        if per_hyp:
            out["count"], out["residual"] = cnt, res
        return out


class Batch:
    """Device-resident batch of image pairs (ImagePair ctor + reconstruct per pair)."""

    def __init__(self, ctx, n_pairs, max_kp, desc_bytes=32):
        self.ctx, self.n_pairs, self.max_kp, self.desc_bytes = ctx, n_pairs, max_kp, desc_bytes
        self._h = C.c_void_p()
        st = lib().mvs_batch_create(ctx._h, C.c_int(n_pairs), C.c_int(max_kp), C.c_int(desc_bytes), C.byref(self._h))
        ctx._check(st, "mvs_batch_create")
        ctx._children.add(self)

    def close(self):
        if self._h:
            if self.ctx._h:  # never touch a batch whose context is already gone
                lib().mvs_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, first, base_desc, base_kp, n_base, pair_desc, pair_kp, n_pair, K, global_index=None):
        count = len(n_base)
        N, D = self.max_kp, self.desc_bytes
        base_desc = np.ascontiguousarray(base_desc, dtype=np.uint8).reshape(count, N, D)
        pair_desc = np.ascontiguousarray(pair_desc, dtype=np.uint8).reshape(count, N, D)
        base_kp = np.ascontiguousarray(base_kp, dtype=np.float32).reshape(count, N, 2)
        pair_kp = np.ascontiguousarray(pair_kp, dtype=np.float32).reshape(count, N, 2)
        n_base = np.ascontiguousarray(n_base, dtype=np.int32)
        n_pair = np.ascontiguousarray(n_pair, dtype=np.int32)
        K = _f64(K)
        if K.size == 9:
            K = np.tile(K.reshape(1, 9), (count, 1))
        K = np.ascontiguousarray(K.reshape(count, 9))
        gi = None if global_index is None else np.ascontiguousarray(global_index, dtype=np.int64)
        st = lib().mvs_batch_upload(self._h, C.c_int(first), C.c_int(count), _ptr(base_desc, C.c_uint8),
                                    _ptr(base_kp, C.c_float), _ptr(n_base, C.c_int32), _ptr(pair_desc, C.c_uint8),
                                    _ptr(pair_kp, C.c_float), _ptr(n_pair, C.c_int32), _ptr(K, C.c_double),
                                    _ptr(gi, C.c_int64))
        self.ctx._check(st, "mvs_batch_upload")

    def upload_async(self, first, base_desc, base_kp, n_base, pair_desc, pair_kp, n_pair, K, global_index):
        """enqueue the upload; the arrays (ideally pinned_empty() ones, exactly typed and contiguous: they are NOT
        copied) must stay alive and unchanged until sync()"""
        count = len(n_base)
        for a, t in ((base_desc, np.uint8), (pair_desc, np.uint8), (base_kp, np.float32), (pair_kp, np.float32),
                     (n_base, np.int32), (n_pair, np.int32), (K, np.float64), (global_index, np.int64)):
            assert a.dtype == t and a.flags["C_CONTIGUOUS"]
        assert K.size == 9 * count
        st = lib().mvs_batch_upload_async(self._h, C.c_int(first), C.c_int(count), _ptr(base_desc, C.c_uint8),
                                          _ptr(base_kp, C.c_float), _ptr(n_base, C.c_int32), _ptr(pair_desc, C.c_uint8),
                                          _ptr(pair_kp, C.c_float), _ptr(n_pair, C.c_int32), _ptr(K, C.c_double),
                                          _ptr(global_index, C.c_int64))
        self.ctx._check(st, "mvs_batch_upload_async")

    def download_async(self, first, count, results, matches=None, mask=None, points=None, point_idx32=None):
        """enqueue the download into caller-owned (ideally pinned) arrays; valid after sync()"""
        st = lib().mvs_batch_download_async(
            self._h, C.c_int(first), C.c_int(count), results.ctypes.data_as(C.c_void_p),
            None if matches is None else matches.ctypes.data_as(C.c_void_p), _ptr(mask, C.c_uint8),
            _ptr(points, C.c_double), _ptr(point_idx32, C.c_int32))
        self.ctx._check(st, "mvs_batch_download_async")

    def run(self, params, n_active=None):
        st = lib().mvs_batch_run(self._h, C.byref(params), C.c_int(n_active or self.n_pairs))
        self.ctx._check(st, "mvs_batch_run")

    def upload_intrinsics(self, first, K, global_index=None, count=None):
        """only K (+ the sampler key offsets) of pairs [first, first + count): what run_points() needs resident"""
        K = _f64(K)
        count = count or (K.size // 9 if K.size > 9 else self.n_pairs - first)
        if K.size == 9:
            K = np.tile(K.reshape(1, 9), (count, 1))
        K = np.ascontiguousarray(K.reshape(count, 9))
        gi = None if global_index is None else np.ascontiguousarray(global_index, dtype=np.int64)
        st = lib().mvs_batch_upload(self._h, C.c_int(first), C.c_int(count), None, None, None, None, None, None,
                                    _ptr(K, C.c_double), _ptr(gi, C.c_int64))
        self.ctx._check(st, "mvs_batch_upload")

    def run_points(self, params, uv1, uv2, m):
        """a batch of sfm_solve calls on matched image points: uv1 / uv2 [count][<= max_kp][2], m [count] (mvs_batch_run_points)"""
        m = np.ascontiguousarray(m, dtype=np.int32)
        count, N = len(m), self.max_kp

        def pad(a):
            a = _f64(a)
            out = np.zeros((count, N, 2))
            out[:, :a.shape[1]] = a.reshape(count, -1, 2)
            return out

        u1, u2 = pad(uv1), pad(uv2)
        st = lib().mvs_batch_run_points(self._h, C.byref(params), C.c_int(count), _ptr(u1, C.c_double), _ptr(u2, C.c_double),
                                        _ptr(m, C.c_int32))
        self.ctx._check(st, "mvs_batch_run_points")

    def device_state(self):
        """opaque bytes of the batch's device-resident state (mvs_batch_device_state): for the diagnostics library's audit"""
        n = C.c_size_t(0)
        self.ctx._check(lib().mvs_batch_device_state(self._h, None, C.c_size_t(0), C.byref(n)), "mvs_batch_device_state")
        buf = (C.c_ubyte * n.value)()
        self.ctx._check(lib().mvs_batch_device_state(self._h, buf, n, C.byref(n)), "mvs_batch_device_state")
        return buf

    def sync(self):
        self.ctx._check(lib().mvs_batch_sync(self._h), "mvs_batch_sync")

    def time(self, params, steps, warmup, n_active=None, per_kernel=True):
        total = C.c_float(0)
        kern = (C.c_float * 5)()
        st = lib().mvs_batch_time(self._h, C.byref(params), C.c_int(n_active or self.n_pairs), C.c_int(warmup),
                                  C.c_int(steps), C.byref(total), kern if per_kernel else None)
        self.ctx._check(st, "mvs_batch_time")
        names = ("match", "match_compact", "ransac", "finalize")
        return total.value, {n: kern[i] for i, n in enumerate(names)}

    def time_kernels(self, params, steps, n_active=None):
        """mean ms of every kernel launch of one pipeline pass, in launch order: [(kernel name, ms)] (HIP events in front
        of every launch, on the launches' own stream; an instrumented replay outside any timed region)"""
        cap = 32
        kid = (C.c_int32 * cap)()
        ms = (C.c_float * cap)()
        n = C.c_int(0)
        st = lib().mvs_batch_time_kernels(self._h, C.byref(params), C.c_int(n_active or self.n_pairs), C.c_int(steps),
                                          C.c_int(cap), kid, ms, C.byref(n))
        self.ctx._check(st, "mvs_batch_time_kernels")
        names = {v["kernel_id"]: k for k, v in self.ctx.kernel_info(self.max_kp, self.desc_bytes).items()}
        return [(names.get(kid[k], "kernel_%d" % kid[k]), float(ms[k])) for k in range(n.value)]

    def stats(self, params, n_active=None):
        ws = WorkStats()
        st = lib().mvs_batch_stats(self._h, C.byref(params), C.c_int(n_active or self.n_pairs), C.byref(ws))
        self.ctx._check(st, "mvs_batch_stats")
        d = {n: getattr(ws, n) for n, _ in WorkStats._fields_ if n != "pairs_mode"}
        d["pairs_mode"] = [int(x) for x in ws.pairs_mode]
        return d

    def download(self, first=0, count=None, matches=True, mask=True, points=True):
        count = count or (self.n_pairs - first)
        N = self.max_kp
        res = np.zeros(count, dtype=RESULT_DTYPE)
        mt = np.zeros((count, N), dtype=MATCH_DTYPE) if matches else None
        mk = np.zeros((count, N), dtype=np.uint8) if mask else None
        pts = np.zeros((count, N, 3)) if points else None
        idx = np.zeros((count, N), dtype=np.int64) if points else None
        st = lib().mvs_batch_download(self._h, C.c_int(first), C.c_int(count), res.ctypes.data_as(C.c_void_p),
                                      None if mt is None else mt.ctypes.data_as(C.c_void_p), _ptr(mk, C.c_uint8),
                                      _ptr(pts, C.c_double), _ptr(idx, C.c_int64))
        self.ctx._check(st, "mvs_batch_download")
        return dict(results=res, matches=mt, mask=mk, points=pts, point_idx=idx)

    def upload_octaves(self, first, base_octave=None, pair_octave=None):
        """cv::KeyPoint::octave of the uploaded keypoints, [count][max_kp] uint8 per image (observation covariances
        of refine(): stddev = 2^octave * sigma_px)"""
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.uint8) for a in (base_octave, pair_octave)]
        count = next(a for a in arrs if a is not None).shape[0]
        for a in arrs:
            assert a is None or a.shape == (count, self.max_kp)
        st = lib().mvs_batch_upload_octaves(self._h, C.c_int(first), C.c_int(count), _ptr(arrs[0], C.c_uint8),
                                            _ptr(arrs[1], C.c_uint8))
        self.ctx._check(st, "mvs_batch_upload_octaves")

    def refine(self, params=None, sigma_px=0.5):
        """ImagePair::refine of every valid pair, on the device, from the batch's own results (asynchronous)"""
        params = params or default_refine_params()
        self.ctx._check(lib().mvs_batch_refine(self._h, C.byref(params), C.c_double(sigma_px)), "mvs_batch_refine")

    def download_refined(self, points=True, point_cov=False):
        P, N = self.n_pairs, self.max_kp
        res = np.zeros(P, dtype=REFINE_DTYPE)
        pts = np.zeros((P, N, 3)) if points else None
        pc = np.zeros((P, N, 3, 3)) if point_cov else None
        st = lib().mvs_batch_download_refined(self._h, res.ctypes.data_as(C.c_void_p), _ptr(pts, C.c_double),
                                              _ptr(pc, C.c_double))
        self.ctx._check(st, "mvs_batch_download_refined")
        return dict(refined=res, points=pts, point_cov=pc)

    def copy_results_device(self, dst_ptr, first=0, count=None):
        """async D2D copy of the fixed-size result records into caller-owned device memory (e.g. a torch tensor)"""
        st = lib().mvs_batch_copy_results_device(self._h, C.c_int(first), C.c_int(count or self.n_pairs - first),
                                                 C.c_void_p(dst_ptr))
        self.ctx._check(st, "mvs_batch_copy_results_device")

    def results_device(self):
        p = C.c_void_p()
        sz = C.c_size_t(0)
        self.ctx._check(lib().mvs_batch_results_device(self._h, C.byref(p), C.byref(sz)), "mvs_batch_results_device")
        return p.value, sz.value


class Sequence:
    """Device-resident frame sequence (row f2): pair k = frames (k, k+1); track q = pnp_solve of frame q+2 against the
    points pair q triangulated, joined on the device through pair q+1's matches."""

    def __init__(self, ctx, n_frames, max_kp, desc_bytes=32):
        self.ctx, self.n_frames, self.max_kp, self.desc_bytes = ctx, n_frames, max_kp, desc_bytes
        self._h = C.c_void_p()
        ctx._check(lib().mvs_seq_create(ctx._h, C.c_int(n_frames), C.c_int(max_kp), C.c_int(desc_bytes),
                                        C.byref(self._h)), "mvs_seq_create")
        ctx._children.add(self)

    def close(self):
        if self._h:
            if self.ctx._h:
                lib().mvs_seq_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, first, desc, kp, n_kp, K):
        count = len(n_kp)
        desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(count, self.max_kp, self.desc_bytes)
        kp = np.ascontiguousarray(kp, dtype=np.float32).reshape(count, self.max_kp, 2)
        n_kp = np.ascontiguousarray(n_kp, dtype=np.int32)
        st = lib().mvs_seq_upload(self._h, C.c_int(first), C.c_int(count), _ptr(desc, C.c_uint8), _ptr(kp, C.c_float),
                                  _ptr(n_kp, C.c_int32), _ptr(_f64(K, (9,)), C.c_double))
        self.ctx._check(st, "mvs_seq_upload")

    def run(self, params, pnp_params):
        self.ctx._check(lib().mvs_seq_run(self._h, C.byref(params), C.byref(pnp_params)), "mvs_seq_run")
        self.ctx._check(lib().mvs_seq_sync(self._h), "mvs_seq_sync")

    def time(self, params, pnp_params, steps, warmup):
        ms = C.c_float(0)
        st = lib().mvs_seq_time(self._h, C.byref(params), C.byref(pnp_params), C.c_int(warmup), C.c_int(steps), C.byref(ms))
        self.ctx._check(st, "mvs_seq_time")
        return ms.value

    def time_stages(self, params, pnp_params, steps):
        ms = (C.c_float * 4)()
        st = lib().mvs_seq_time_stages(self._h, C.byref(params), C.byref(pnp_params), C.c_int(steps), ms)
        self.ctx._check(st, "mvs_seq_time_stages")
        return {n: ms[i] / steps for i, n in enumerate(("pairs", "join", "pnp", "chain"))}

    def upload_images(self, first, images, K, params=None):
        """extract keypoints + descriptors of frames [first, first + len(images)) on the device, straight into the
        sequence's resident frame arrays (up to max_kp per frame)"""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        B, H, W = images.shape
        params = params or default_orb_params()
        st = lib().mvs_seq_upload_images(self._h, C.c_int(first), C.c_int(B), _ptr(images, C.c_uint8), C.c_int(W),
                                         C.c_int(H), C.byref(params), None if K is None else _ptr(_f64(K, (9,)), C.c_double))
        self.ctx._check(st, "mvs_seq_upload_images")

    def download_trajectory(self):
        """pose of every frame in frame 0 (pair 0's baseline = 1), pair and track scales (scale propagation, row f2)"""
        F = self.n_frames
        R, t, ps, ts = np.zeros((F, 3, 3)), np.zeros((F, 3)), np.zeros(F - 1), np.zeros(F - 2)
        st = lib().mvs_seq_download_trajectory(self._h, _ptr(R, C.c_double), _ptr(t, C.c_double), _ptr(ps, C.c_double),
                                               _ptr(ts, C.c_double))
        self.ctx._check(st, "mvs_seq_download_trajectory")
        return dict(R=R, t=t, pair_scale=ps, track_scale=ts)

    def upload_octaves(self, first, octave):
        octave = np.ascontiguousarray(octave, dtype=np.uint8)
        assert octave.ndim == 2 and octave.shape[1] == self.max_kp
        st = lib().mvs_seq_upload_octaves(self._h, C.c_int(first), C.c_int(octave.shape[0]), _ptr(octave, C.c_uint8))
        self.ctx._check(st, "mvs_seq_upload_octaves")

    def refine_pairs(self, params=None, sigma_px=0.5):
        params = params or default_refine_params()
        self.ctx._check(lib().mvs_seq_refine_pairs(self._h, C.byref(params), C.c_double(sigma_px)), "mvs_seq_refine_pairs")

    def download_refined(self, points=True, point_cov=False):
        P, N = self.n_frames - 1, self.max_kp
        res = np.zeros(P, dtype=REFINE_DTYPE)
        pts = np.zeros((P, N, 3)) if points else None
        pc = np.zeros((P, N, 3, 3)) if point_cov else None
        st = lib().mvs_seq_download_refined(self._h, res.ctypes.data_as(C.c_void_p), _ptr(pts, C.c_double), _ptr(pc, C.c_double))
        self.ctx._check(st, "mvs_seq_download_refined")
        return dict(refined=res, points=pts, point_cov=pc)

    def download_pairs(self):
        P, N = self.n_frames - 1, self.max_kp
        res = np.zeros(P, dtype=RESULT_DTYPE)
        mt = np.zeros((P, N), dtype=MATCH_DTYPE)
        mk = np.zeros((P, N), dtype=np.uint8)
        pts = np.zeros((P, N, 3))
        idx = np.zeros((P, N), dtype=np.int64)
        st = lib().mvs_seq_download_pairs(self._h, C.c_int(0), C.c_int(P), res.ctypes.data_as(C.c_void_p),
                                          mt.ctypes.data_as(C.c_void_p), _ptr(mk, C.c_uint8), _ptr(pts, C.c_double),
                                          _ptr(idx, C.c_int64))
        self.ctx._check(st, "mvs_seq_download_pairs")
        return dict(results=res, matches=mt, mask=mk, points=pts, point_idx=idx)

    def download_tracks(self):
        T, N = self.n_frames - 2, self.max_kp
        tr = np.zeros(T, dtype=TRACK_DTYPE)
        X = np.zeros((T, N, 3))
        uv = np.zeros((T, N, 2))
        inl = np.zeros((T, N), dtype=np.int64)
        st = lib().mvs_seq_download_tracks(self._h, C.c_int(0), C.c_int(T), tr.ctypes.data_as(C.c_void_p),
                                           _ptr(X, C.c_double), _ptr(uv, C.c_double), _ptr(inl, C.c_int64))
        self.ctx._check(st, "mvs_seq_download_tracks")
        return dict(tracks=tr, corr_xyz=X, corr_uv=uv, inlier_idx=inl)
