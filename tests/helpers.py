"""Shared fixtures of the reference's tests, restated as data generators (test/unit-test-helper.cpp:42-79)."""
import numpy as np

import oracle_lib as o

CUBE = np.array([[-1, -1, -1], [-1, -1, 1], [-1, 1, -1], [-1, 1, 1], [1, -1, -1], [1, -1, 1], [1, 1, -1], [1, 1, 1.0]])
L_SHAPE = np.array([[1, 0, 0], [0, 0, 0], [0, 2, 0], [1, 0, 3], [0, 0, 3], [0, 2, 3], [0.5, 0, 1.5], [0, 1, 1.5]])


def rig_points(kind, rpy, translation, scale):
    """get_rig_points(type, SO3(roll, pitch, yaw), t, scale): p = R * (scale * p) + t."""
    P = CUBE if kind == "cube" else L_SHAPE
    R = o.so3_from_rpy(*rpy)
    return (R @ (scale * P).T).T + np.asarray(translation, dtype=float)


def two_camera_rig(kind, rpy=(0.0, 0.0, 0.0), translation=(0.6, 0.0, 3.0), scale=1.0, se3_2to1=(1, 0, 0, 0, 0, 0)):
    """The geometry of test/test-sfm.cpp:17-42: K = I, camera 1 at the origin, camera 2 = exp(se3_2to1)."""
    K = np.eye(3)
    R21, t21 = o.se3_exp(np.asarray(se3_2to1, dtype=float))  # pose of camera 2 in camera 1
    R12, t12 = o.se3_inverse(R21, t21)                        # world (= camera 1) -> camera 2
    X = rig_points(kind, rpy, translation, scale)
    uv1 = o.project_points(K, np.eye(3), np.zeros(3), X)
    uv2 = o.project_points(K, R12, t12, X)
    return dict(K=K, X=X, uv1=uv1, uv2=uv2, pose2in1=(R21, t21), T1to2=(R12, t12))


def skew(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0.0]])


def rel_err(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    d = np.abs(a - b).max() if a.size else 0.0
    s = max(np.abs(b).max() if b.size else 0.0, 1e-300)
    return d / s
