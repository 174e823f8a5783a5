"""numpy model of the RANSAC pre-screen: the approximate fundamental matrix of a sample and the
a-priori / a-posteriori bound `band` on how far any point's epipolar residual under it can be from the residual under the
EXACT contract result (the one-sided Jacobi path of oracle/mvs_oracle.c = device_math.hpp eight_point).

Test infrastructure: it restates the arithmetic the device pre-screen performs (device_math.hpp `prescreen_hypothesis`) so
that the bound's constants can be exercised on the CPU against the oracle, hypothesis by hypothesis, including adversarial
samples.  The product path never imports this file.

Per sample (8 point pairs, Hartley-normalised exactly like the exact path):
  A (8 x 9)  ->  Householder QR of A^T (9 x 8)  ->  n~ = Q e_9 (unit null vector), R (8 x 8 upper triangular)
  rho = 1.8e-13 ||A||_F >= ||A n~||   a-priori residual of the Householder null vector (the model asserts it)
  sigma8_lb <= sigma_8(A)             from ||R^-1||_F (explicit triangular inverse, backward-stable solve)
  eta_J  = 1.01 tau' / sigma8_lb^2 + 4e-12,   tau' = 2e-12 ||A||_F^2      exact path's null vector vs the true one
  eta_A  = 1.5 rho / sigma8_lb + 1e-13                                     approximate null vector vs the true one
  rank-2 step: ONE verified singular triplet (sig, u, v) of G = reshape(n~): v any approximation of the smallest right singular
  vector, w = G v, sig = ||w||, u = w / sig, eps2 = ||G^T u - sig v||; X = G - w v^T; s2lb <= sigma_2(X) from X's invariants
  e, sig_e, extra = (eps2, sig, eps2) if eps2 < sig else (sig, 0, 0)   (+ 1e-12 of roundings each)
  eta    = eta_J + eta_A + e + 2e-11                                       (+ backward error of the exact path's 3x3 SVD)
  delta  = s2lb - extra - sig_e - eta      gap of the singular value that the rank-2 step removes
  dFn    = (2 + 2 (sig_e + 3 eta) / delta) eta + extra + 2e-11             Wedin's sin-theta theorem
  band   = dFn N1 N2 (1 + 1e-9) + 64 u N1' N2'      N = max over the pair's points of ||T p||, N' with absolute values
  e32    = 16 * 2^-24 * [X2 Y2 1] |F~| [X1 Y1 1]^T    single-precision evaluation of the residual (added to band there)
"""
import numpy as np

U = 2.0 ** -53
TAU_C = 2.0e-12        # >= 2.001 (8000 u + 8.01 u): backward error of <= 1080 rotations + of forming A^T A
ETA_Q = 4.0e-12        # loss of orthogonality of the accumulated V^T over <= 1080 rotations (3.3e4 u)
SVD3_E = 2.0e-11       # generous bound on the backward error of a 3x3 Jacobi SVD + recomposition (<= 90 rotations)
TRIP_E = 1.0e-12       # roundings of the verified singular triplet (~150 operations on |x| <= 1.01)
BAND_FRAC = 4.0        # a hypothesis is screened only if band <= BAND_FRAC * threshold (kPsBandFrac)


def hartley(px, py):
    """normalise8 of device_math.hpp / find_normalization_transform (fundamental-matrix.cpp:18-54)"""
    mx = px.sum() * 0.125
    my = py.sum() * 0.125
    dx, dy = px - mx, py - my
    sc = np.sqrt(dx * dx + dy * dy).sum() * 0.125
    ok = sc > 2.220446049250313e-16
    s = np.sqrt(2.0) / sc
    return dx * s, dy * s, s, mx, my, ok


def design(a1, b1, a2, b2):
    return np.stack([a2 * a1, a2 * b1, a2, b2 * a1, b2 * b1, b2, a1, b1, np.ones(8)], axis=1)


def householder_null(A):
    """QR of A^T (columns = rows of A) by Householder reflections, no pivoting.  Returns n (9,), R (8, 8)."""
    C = A.T.copy()                 # 9 x 8
    vs, betas = [], []
    for k in range(8):
        x = C[k:, k]
        nrm = np.sqrt((x * x).sum())
        alpha = -nrm if x[0] >= 0 else nrm
        v = x.copy()
        v[0] = x[0] - alpha        # same sign: no cancellation
        vv = nrm * (nrm + abs(x[0]))     # = v.v / 2
        beta = 1.0 / vv if vv > 0 else 0.0
        for j in range(k + 1, 8):
            t = beta * (v * C[k:, j]).sum()
            C[k:, j] -= t * v
        C[k, k] = alpha
        C[k + 1:, k] = 0.0
        vs.append(v)
        betas.append(beta)
    n = np.zeros(9)
    n[8] = 1.0
    for k in range(7, -1, -1):
        v = vs[k]
        t = betas[k] * (v * n[k:]).sum()
        n[k:] -= t * v
    return n, C[:8, :8]


def tri_inverse_fro(R):
    """||R^-1||_F of an upper-triangular 8 x 8 by explicit back substitution (inf for a zero pivot)"""
    n = R.shape[0]
    if np.any(np.diag(R) == 0.0):
        return np.inf, None
    Y = np.zeros_like(R)
    for j in range(n):
        Y[j, j] = 1.0 / R[j, j]
        for i in range(j - 1, -1, -1):
            Y[i, j] = -(R[i, i + 1:j + 1] * Y[i + 1:j + 1, j]).sum() / R[i, i]
    return np.sqrt((Y * Y).sum()), Y


def rank2(f, v=None):
    """the pre-screen's rank-2 step (prescreen.hpp prescreen_rank2).  v: any approximation of the right singular vector of
    the smallest singular value of G = reshape(f) (default: numpy's; the device takes it from the characteristic polynomial
    of G^T G) -- the bound is a-posteriori, so a bad v widens the band instead of breaking it.
    Returns (X, e, sig_e, extra, s2lb)."""
    G = f.reshape(3, 3)
    if v is None:
        v = np.linalg.eigh(G.T @ G)[1][:, 0]
    v = v / np.sqrt((v * v).sum())
    w = G @ v
    X = G - np.outer(w, v)
    sig = float(np.sqrt((w * w).sum()))
    if sig >= 2.0 ** -190:
        u = w / sig
        r2 = G.T @ u - sig * v
        eps2 = float(np.sqrt((r2 * r2).sum()))
    else:
        eps2 = np.inf
    if eps2 < sig:
        e, sig_e, extra = eps2 + TRIP_E, sig, eps2 + TRIP_E
    else:
        e, sig_e, extra = sig + TRIP_E, 0.0, TRIP_E
    q1 = float((X * X).sum())
    T = X.T @ X
    q2 = float((T * T).sum())
    Pm = 0.5 * (q1 * q1 - q2) - 4e-15
    disc = max(q1 * q1 - 4.0 * Pm, 0.0) + 1e-13
    s2q = 2.0 * Pm / (q1 * (1 + 1e-13) + np.sqrt(disc))
    s2lb = float(np.sqrt(s2q)) * (1 - 1e-14) if s2q > 0 else 0.0
    return X, e, sig_e, extra, s2lb


def denormalise(Fn, s1, m1x, m1y, s2, m2x, m2y):
    T1 = np.array([[s1, 0, -m1x * s1], [0, s1, -m1y * s1], [0, 0, 1.0]])
    T2 = np.array([[s2, 0, -m2x * s2], [0, s2, -m2y * s2], [0, 0, 1.0]])
    return T2.T @ Fn @ T1


def prescreen(x1, y1, x2, y2, bbox, v3=None):
    """x1, y1, x2, y2: the 8 sampled ideal-camera points.  bbox = (x1lo, x1hi, y1lo, y1hi, x2lo, x2hi, y2lo, y2hi) of ALL
    matches of the pair.  Returns dict(ok, screenable, F, band, ...)."""
    a1, b1, s1, m1x, m1y, ok1 = hartley(x1, y1)
    a2, b2, s2, m2x, m2y, ok2 = hartley(x2, y2)
    out = dict(ok=bool(ok1 and ok2), screenable=False, F=None, band=np.inf)
    if not out["ok"]:
        return out
    A = design(a1, b1, a2, b2)
    S = float((A * A).sum()) * (1 + 1e-12)
    n, R = householder_null(A)
    rho = 1.8e-13 * np.sqrt(S)     # a-priori bound of ||A n~|| for the Householder null vector (1548 u ||A||_F: beta within 28 u)
    assert float(np.sqrt(((A @ n) ** 2).sum())) <= rho
    yf, _ = tri_inverse_fro(R)
    rf = float(np.sqrt((R * R).sum()))
    if not np.isfinite(yf):
        return out
    z = 16.0 * U * rf * yf         # (the device's pivot reciprocals are within 3 u, not correctly rounded)
    if not (z < 0.5):
        return out
    sig8 = (1.0 - z) / yf * (1 - 1e-13) - 4e-14 * np.sqrt(S)
    if not (sig8 > 0):
        return out
    g = sig8 * sig8
    eta_j = 1.01 * TAU_C * S / g + ETA_Q
    eta_a = 1.5 * rho / sig8 + 1e-13
    Fn, e3, sig_e, extra, s2lb = rank2(n, v3)
    eta = eta_j + eta_a + e3 + SVD3_E
    delta = s2lb - extra - sig_e - eta
    out.update(sig8_lb=sig8, eta=eta, delta=delta, n=n, A=A, sig_e=sig_e, s2lb=s2lb)
    if not (delta > 0 and eta < 1e-3):
        return out
    dfn = (2.0 + 2.0 * (sig_e + 3 * eta) / delta) * eta + extra + SVD3_E
    x1lo, x1hi, y1lo, y1hi, x2lo, x2hi, y2lo, y2hi = bbox
    d1x = max(abs(m1x - x1lo), abs(m1x - x1hi))
    d1y = max(abs(m1y - y1lo), abs(m1y - y1hi))
    d2x = max(abs(m2x - x2lo), abs(m2x - x2hi))
    d2y = max(abs(m2y - y2lo), abs(m2y - y2hi))
    N1 = np.sqrt(1 + s1 * s1 * (d1x * d1x + d1y * d1y))
    N2 = np.sqrt(1 + s2 * s2 * (d2x * d2x + d2y * d2y))
    e1x = max(abs(x1lo), abs(x1hi)) + abs(m1x)
    e1y = max(abs(y1lo), abs(y1hi)) + abs(m1y)
    e2x = max(abs(x2lo), abs(x2hi)) + abs(m2x)
    e2y = max(abs(y2lo), abs(y2hi)) + abs(m2y)
    N1p = np.sqrt(1 + s1 * s1 * (e1x * e1x + e1y * e1y))
    N2p = np.sqrt(1 + s2 * s2 * (e2x * e2x + e2y * e2y))
    band = dfn * N1 * N2 * (1 + 1e-9) + 64 * U * N1p * N2p
    F = denormalise(Fn, s1, m1x, m1y, s2, m2x, m2y)
    # single-precision evaluation of the residual: <= 7 roundings per term as a nested fma chain (ransac_count32_kernel),
    # <= 14 in the matrix-core form (rounded monomial + an fmaf chain over ten k, ransac_count_mfma_kernel)
    X1, Y1 = max(abs(x1lo), abs(x1hi)), max(abs(y1lo), abs(y1hi))
    X2, Y2 = max(abs(x2lo), abs(x2hi)), max(abs(y2lo), abs(y2hi))
    T = np.array([X2, Y2, 1.0]) @ np.abs(F) @ np.array([X1, Y1, 1.0])
    e32 = 16.0 * 2.0 ** -24 * T * (1 + 1e-6) + 1e-30
    out.update(F=F, band=float(band), e32=float(e32), dfn=dfn, N=(N1, N2), Fn=Fn, screenable=bool(np.isfinite(band)))
    return out


def residuals(F, p1, p2):
    """|p2^T F p1| with homogeneous 1 for all points (M x 2 arrays)"""
    x1, y1, x2, y2 = p1[:, 0], p1[:, 1], p2[:, 0], p2[:, 1]
    u0 = x2 * F[0, 0] + y2 * F[1, 0] + F[2, 0]
    u1 = x2 * F[0, 1] + y2 * F[1, 1] + F[2, 1]
    u2 = x2 * F[0, 2] + y2 * F[1, 2] + F[2, 2]
    return np.abs(u0 * x1 + u1 * y1 + u2)
