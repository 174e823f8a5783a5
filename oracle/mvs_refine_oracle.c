/*
 * mvs_refine_oracle.c -- CPU ORACLE for row f4 of SURVEY.md section 8: two-view refinement
 * (vision/sfm-refine.cpp:20-139) and motion-only refinement (vision/pnp-refine.cpp:14-108), both of which the
 * reference forwards to ba_frame_pose_and_point (vision/ba.cpp:26-156) = GTSAM LevenbergMarquardtOptimizer +
 * Marginals.  TEST INFRASTRUCTURE ONLY (see mvs_oracle.h).
 *
 * GTSAM is a third-party dependency that is absent from the reference tree (version unpinned, README), so this is a
 * restatement of the NONLINEAR LEAST-SQUARES PROBLEM ba.cpp builds, not of GTSAM's elimination order:
 *   variables   camera poses x_f (gtsam::Pose3 = camera in world, p_w = R p_c + t; ba.cpp:59-60) and points p_i
 *   factors     PriorFactor<Pose3>(guess, diag sigma^2)                                  ba.cpp:65-71
 *               PriorFactor<Point3>(guess, covariance)                                   ba.cpp:86-92
 *               GenericProjectionFactor<Pose3, Point3, Cal3_S2>(uv, covariance, K)        ba.cpp:97-118
 *   cost        F = 1/2 sum of squared Mahalanobis residuals  (= optimizer.error(), ba.cpp:155)
 *   outputs     the minimiser, and the marginal covariances of the linearised graph at it (ba.cpp:127,141,152)
 * Conventions restated from GTSAM's published definitions: tangent order (rotation, translation), right
 * perturbation R <- R Exp(dw), t <- t + R dv; Pose3 prior error = (Log(R0^T R), R0^T (t - t0)) (first-order chart,
 * the pre-4.1 default); a point at or behind the camera contributes the constant residual (2 fx, 2 fx) with a zero
 * Jacobian (GenericProjectionFactor with throwCheirality = false).
 * The minimiser does not depend on GTSAM's damping schedule; this oracle iterates to a tighter tolerance than
 * GTSAM's defaults (relative/absolute error decrease 1e-5), so it agrees with the reference to within the
 * reference's own convergence slack.  PARITY PINNING: tests/test_refine.py checks the minimiser against
 * scipy.optimize.least_squares on the same residual vector, the covariances against a finite-difference Hessian,
 * and the reference's own sfm_refine_L_shape known-answer test (test/test-sfm.cpp:157-286, tolerance 0.025).
 *
 * Summation order (shared with the HIP kernel by specification): 256 partial sums, partial l takes points
 * l, l+256, ... in order; the partials are combined by a balanced binary tree inside each group of 64 and then as
 * (g0 + g1) + (g2 + g3).  sin/cos/acos come from libm here and from the device library on the GPU, so parity for
 * this row is by tolerance, not bit-exact.
 */
#include "mvs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NT 256
#define MAXC 12
#define NLOW(n) ((n) * ((n) + 1) / 2)
#define LIDX(r, c) ((r) * ((r) + 1) / 2 + (c)) /* r >= c */

/* fused building blocks: the HIP kernel uses the same ones in the same places (one v_fma_f64 each) */
static inline double fd2(double a0, double b0, double a1, double b1) { return fma(a1, b1, a0 * b0); }
static inline double fd3(double a0, double b0, double a1, double b1, double a2, double b2)
{
    return fma(a2, b2, fma(a1, b1, a0 * b0));
}
static inline double fx2(double a, double b, double c, double d) { return fma(a, b, -(c * d)); } /* a b - c d */

typedef struct {
    int F, m;
    double fx, fy, sk, cx, cy;
    double R0[2][9], t0[2][3];
    double w[2][6];         /* 1 / sigma^2 per tangent coordinate */
    const double *pts0;     /* m x 3 */
    const double *pinfo;    /* m x 6 (xx xy xz yy yz zz) */
    const double *obs[2];   /* m x 2 */
    const double *oinfo[2]; /* m x 3 (xx xy yy) */
} ba_problem;

static void m3mul(const double *A, const double *B, double *C)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
}
static void m3tmul(const double *A, const double *B, double *C) /* A^T B */
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = (A[i] * B[j] + A[3 + i] * B[3 + j]) + A[6 + i] * B[6 + j];
}

static void so3_exp(const double w[3], double R[9])
{
    double th2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
    double th = sqrt(th2), A, B;
    if (th < 1e-4) {
        A = 1.0 - th2 / 6.0;
        B = 0.5 - th2 / 24.0;
    } else {
        A = sin(th) / th;
        B = (1.0 - cos(th)) / th2;
    }
    double x = w[0], y = w[1], z = w[2];
    R[0] = 1.0 - B * (y * y + z * z);
    R[1] = B * (x * y) - A * z;
    R[2] = B * (x * z) + A * y;
    R[3] = B * (x * y) + A * z;
    R[4] = 1.0 - B * (x * x + z * z);
    R[5] = B * (y * z) - A * x;
    R[6] = B * (x * z) - A * y;
    R[7] = B * (y * z) + A * x;
    R[8] = 1.0 - B * (x * x + y * y);
}

static void so3_log(const double R[9], double w[3])
{
    double vx = 0.5 * (R[7] - R[5]), vy = 0.5 * (R[2] - R[6]), vz = 0.5 * (R[3] - R[1]);
    double s = sqrt((vx * vx + vy * vy) + vz * vz);           /* sin(theta) */
    double c = 0.5 * (((R[0] + R[4]) + R[8]) - 1.0);           /* cos(theta) */
    double th = atan2(s, c), k;
    if (s < 1e-4 && c > 0.0)
        k = 1.0 + (s * s) / 6.0;
    else
        k = th / s; /* theta near pi is outside the priors' reach; s = 0 there gives inf and the step is rejected */
    w[0] = k * vx;
    w[1] = k * vy;
    w[2] = k * vz;
}

/* inverse right Jacobian of SO(3): d Log(R Exp(dw)) = Jr^-1(w) dw */
static void so3_jrinv(const double w[3], double J[9])
{
    double th2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
    double th = sqrt(th2), g;
    if (th < 1e-4)
        g = 1.0 / 12.0 + th2 / 720.0;
    else
        g = 1.0 / th2 - (1.0 + cos(th)) / ((2.0 * th) * sin(th));
    double x = w[0], y = w[1], z = w[2];
    /* I + 1/2 [w]x + g [w]x^2,   [w]x^2 = w w^T - th2 I */
    J[0] = 1.0 + g * (x * x - th2);
    J[1] = g * (x * y) - 0.5 * z;
    J[2] = g * (x * z) + 0.5 * y;
    J[3] = g * (x * y) + 0.5 * z;
    J[4] = 1.0 + g * (y * y - th2);
    J[5] = g * (y * z) - 0.5 * x;
    J[6] = g * (x * z) - 0.5 * y;
    J[7] = g * (y * z) + 0.5 * x;
    J[8] = 1.0 + g * (z * z - th2);
}

static int sym3_inverse(const double a[6], double o[6])
{ /* a = xx xy xz yy yz zz */
    double c00 = fx2(a[3], a[5], a[4], a[4]);
    double c01 = fx2(a[2], a[4], a[1], a[5]);
    double c02 = fx2(a[1], a[4], a[2], a[3]);
    double det = fd3(a[0], c00, a[1], c01, a[2], c02);
    double id = 1.0 / det;
    o[0] = c00 * id;
    o[1] = c01 * id;
    o[2] = c02 * id;
    o[3] = fx2(a[0], a[5], a[2], a[2]) * id;
    o[4] = fx2(a[1], a[2], a[0], a[4]) * id;
    o[5] = fx2(a[0], a[3], a[1], a[1]) * id;
    return det != 0.0 && isfinite(id);
}

/* one observation: residual r, Jacobians Jc (2x6, pose) and Jp (2x3, point).  returns 0 on cheirality failure */
static int project_lin(const ba_problem *P, const double R[9], const double t[3], const double p[3], const double *uv,
                       double r[2], double Jc[12], double Jp[6], int want_jac)
{
    double d0 = p[0] - t[0], d1 = p[1] - t[1], d2 = p[2] - t[2];
    double q0 = fd3(R[0], d0, R[3], d1, R[6], d2);
    double q1 = fd3(R[1], d0, R[4], d1, R[7], d2);
    double q2 = fd3(R[2], d0, R[5], d1, R[8], d2);
    if (!(q2 > 0.0)) {
        r[0] = 2.0 * P->fx;
        r[1] = 2.0 * P->fx;
        if (want_jac) {
            memset(Jc, 0, 12 * sizeof(double));
            memset(Jp, 0, 6 * sizeof(double));
        }
        return 0;
    }
    double iz = 1.0 / q2, xn = q0 * iz, yn = q1 * iz;
    double un = fd2(P->fx, xn, P->sk, yn), vn = P->fy * yn;
    r[0] = (un + P->cx) - uv[0];
    r[1] = (vn + P->cy) - uv[1];
    if (!want_jac)
        return 1;
    double A[6] = {P->fx * iz, P->sk * iz, -(un * iz), 0.0, P->fy * iz, -(vn * iz)};
    for (int a = 0; a < 2; ++a) {
        const double *Aa = A + 3 * a;
        Jc[6 * a + 0] = fx2(Aa[1], q2, Aa[2], q1);
        Jc[6 * a + 1] = fx2(Aa[2], q0, Aa[0], q2);
        Jc[6 * a + 2] = fx2(Aa[0], q1, Aa[1], q0);
        Jc[6 * a + 3] = -Aa[0];
        Jc[6 * a + 4] = -Aa[1];
        Jc[6 * a + 5] = -Aa[2];
        for (int k = 0; k < 3; ++k)
            Jp[3 * a + k] = fd3(Aa[0], R[3 * k], Aa[1], R[3 * k + 1], Aa[2], R[3 * k + 2]);
    }
    return 1;
}

/* per-point Schur blocks at the current estimate */
typedef struct {
    double Hpp[6];        /* sym */
    double gp[3];
    double Hcp[MAXC][3];
    double Hcc[2][21];    /* per-frame sym 6x6 lower */
    double gc[MAXC];
    double cost;          /* sum of squared Mahalanobis residuals of this point's factors (not halved) */
} point_blocks;

static void point_linearize(const ba_problem *P, const double R[2][9], const double t[2][3], const double *pts, int i,
                            point_blocks *B)
{
    const double *p = pts + 3 * i, *p0 = P->pts0 + 3 * i, *L = P->pinfo + 6 * i;
    double d[3] = {p[0] - p0[0], p[1] - p0[1], p[2] - p0[2]};
    double Ld[3] = {fd3(L[0], d[0], L[1], d[1], L[2], d[2]), fd3(L[1], d[0], L[3], d[1], L[4], d[2]),
                    fd3(L[2], d[0], L[4], d[1], L[5], d[2])};
    memcpy(B->Hpp, L, 6 * sizeof(double));
    memcpy(B->gp, Ld, 3 * sizeof(double));
    B->cost = fd3(d[0], Ld[0], d[1], Ld[1], d[2], Ld[2]);
    for (int f = 0; f < P->F; ++f) {
        double r[2], Jc[12], Jp[6];
        project_lin(P, R[f], t[f], p, P->obs[f] + 2 * i, r, Jc, Jp, 1);
        const double *W = P->oinfo[f] + 3 * i;
        double wr0 = fd2(W[0], r[0], W[1], r[1]), wr1 = fd2(W[1], r[0], W[2], r[1]);
        B->cost = fma(r[1], wr1, fma(r[0], wr0, B->cost));
        double WJc[12], WJp[6];
        for (int k = 0; k < 6; ++k) {
            WJc[k] = fd2(W[0], Jc[k], W[1], Jc[6 + k]);
            WJc[6 + k] = fd2(W[1], Jc[k], W[2], Jc[6 + k]);
        }
        for (int k = 0; k < 3; ++k) {
            WJp[k] = fd2(W[0], Jp[k], W[1], Jp[3 + k]);
            WJp[3 + k] = fd2(W[1], Jp[k], W[2], Jp[3 + k]);
        }
        for (int a = 0; a < 6; ++a) {
            for (int b = 0; b <= a; ++b)
                B->Hcc[f][LIDX(a, b)] = fd2(Jc[a], WJc[b], Jc[6 + a], WJc[6 + b]);
            B->gc[6 * f + a] = fd2(Jc[a], wr0, Jc[6 + a], wr1);
            for (int k = 0; k < 3; ++k)
                B->Hcp[6 * f + a][k] = fd2(Jc[a], WJp[k], Jc[6 + a], WJp[3 + k]);
        }
        int s = 0;
        for (int a = 0; a < 3; ++a)
            for (int b = a; b < 3; ++b, ++s)
                B->Hpp[s] = fma(Jp[3 + a], WJp[3 + b], fma(Jp[a], WJp[b], B->Hpp[s]));
        for (int k = 0; k < 3; ++k)
            B->gp[k] = fma(Jp[3 + k], wr1, fma(Jp[k], wr0, B->gp[k]));
    }
}

static double point_cost(const ba_problem *P, const double R[2][9], const double t[2][3], const double *p, int i)
{
    const double *p0 = P->pts0 + 3 * i, *L = P->pinfo + 6 * i;
    double d[3] = {p[0] - p0[0], p[1] - p0[1], p[2] - p0[2]};
    double Ld[3] = {fd3(L[0], d[0], L[1], d[1], L[2], d[2]), fd3(L[1], d[0], L[3], d[1], L[4], d[2]),
                    fd3(L[2], d[0], L[4], d[1], L[5], d[2])};
    double c = fd3(d[0], Ld[0], d[1], Ld[1], d[2], Ld[2]);
    for (int f = 0; f < P->F; ++f) {
        double r[2];
        project_lin(P, R[f], t[f], p, P->obs[f] + 2 * i, r, NULL, NULL, 0);
        const double *W = P->oinfo[f] + 3 * i;
        double wr0 = fd2(W[0], r[0], W[1], r[1]), wr1 = fd2(W[1], r[0], W[2], r[1]);
        c = fma(r[1], wr1, fma(r[0], wr0, c));
    }
    return c;
}

static double reduce_nt(double *v)
{
    for (int g = 0; g < NT; g += 64)
        for (int s = 1; s < 64; s *= 2)
            for (int l = 0; l < 64; l += 2 * s)
                v[g + l] = v[g + l] + v[g + l + s];
    return (v[0] + v[64]) + (v[128] + v[192]);
}

/* pose prior: error e (6), Jacobian Je (6x6, block diagonal: Jr^-1 | R0^T R) */
static void pose_prior(const ba_problem *P, int f, const double R[9], const double t[3], double e[6], double Jw[9],
                       double Jv[9])
{
    double Re[9];
    m3tmul(P->R0[f], R, Re);
    so3_log(Re, e);
    double dt[3] = {t[0] - P->t0[f][0], t[1] - P->t0[f][1], t[2] - P->t0[f][2]};
    const double *R0 = P->R0[f];
    e[3] = (R0[0] * dt[0] + R0[3] * dt[1]) + R0[6] * dt[2];
    e[4] = (R0[1] * dt[0] + R0[4] * dt[1]) + R0[7] * dt[2];
    e[5] = (R0[2] * dt[0] + R0[5] * dt[1]) + R0[8] * dt[2];
    if (Jw) {
        so3_jrinv(e, Jw);
        memcpy(Jv, Re, sizeof(Re));
    }
}

static double prior_cost(const ba_problem *P, const double R[2][9], const double t[2][3])
{
    double c = 0.0;
    for (int f = 0; f < P->F; ++f) {
        double e[6];
        pose_prior(P, f, R[f], t[f], e, NULL, NULL);
        for (int k = 0; k < 6; ++k)
            c = c + (e[k] * e[k]) * P->w[f][k];
    }
    return c;
}

/* in-place Cholesky of the lower triangle (packed); returns 0 if not positive definite.  Textbook form: the diagonal
 * holds l_jj and every "divide by l_jj" is a division.  (The device kernel stores 1 / l_jj and multiplies -- a correctly
 * rounded f64 division is ~30 instructions there; round 2 had copied that form into this file, which made the checker
 * follow the implementation.  The two forms differ by an ulp here and there; the refinement tests compare with a stated
 * tolerance, 1e-10 on poses, not bitwise.) */
static int chol_packed(double *S, int n)
{
    for (int j = 0; j < n; ++j) {
        double d = S[LIDX(j, j)];
        for (int k = 0; k < j; ++k)
            d = fma(-S[LIDX(j, k)], S[LIDX(j, k)], d);
        if (!(d > 0.0) || !isfinite(d))
            return 0;
        const double l = sqrt(d);
        S[LIDX(j, j)] = l;
        for (int i = j + 1; i < n; ++i) {
            double v = S[LIDX(i, j)];
            for (int k = 0; k < j; ++k)
                v = fma(-S[LIDX(i, k)], S[LIDX(j, k)], v);
            S[LIDX(i, j)] = v / l;
        }
    }
    return 1;
}
static void chol_solve(const double *Lc, int n, double *b)
{
    for (int i = 0; i < n; ++i) {
        double v = b[i];
        for (int k = 0; k < i; ++k)
            v = fma(-Lc[LIDX(i, k)], b[k], v);
        b[i] = v / Lc[LIDX(i, i)];
    }
    for (int i = n - 1; i >= 0; --i) {
        double v = b[i];
        for (int k = i + 1; k < n; ++k)
            v = fma(-Lc[LIDX(k, i)], b[k], v);
        b[i] = v / Lc[LIDX(i, i)];
    }
}

/* reduced camera system at (R, t, pts) with damping lam: S (packed lower, nc x nc), b (nc), total cost */
static void build_schur(const ba_problem *P, const double R[2][9], const double t[2][3], const double *pts, double lam,
                        double *S, double *b, double *cost2)
{
    const int nc = 6 * P->F, nl = NLOW(nc);
    static __thread double acc[NLOW(MAXC) + MAXC + 1][NT];
    for (int k = 0; k < nl + nc + 1; ++k)
        memset(acc[k], 0, sizeof(acc[k]));
    for (int l = 0; l < NT; ++l) {
        for (int i = l; i < P->m; i += NT) {
            point_blocks B;
            point_linearize(P, R, t, pts, i, &B);
            double Hd[6] = {B.Hpp[0] + lam, B.Hpp[1], B.Hpp[2], B.Hpp[3] + lam, B.Hpp[4], B.Hpp[5] + lam}, Pi[6];
            sym3_inverse(Hd, Pi);
            double Y[MAXC][3];
            for (int a = 0; a < nc; ++a) {
                const double *h = B.Hcp[a];
                Y[a][0] = fd3(h[0], Pi[0], h[1], Pi[1], h[2], Pi[2]);
                Y[a][1] = fd3(h[0], Pi[1], h[1], Pi[3], h[2], Pi[4]);
                Y[a][2] = fd3(h[0], Pi[2], h[1], Pi[4], h[2], Pi[5]);
            }
            for (int f = 0; f < P->F; ++f) /* the frame's own blocks first, then the Schur correction */
                for (int a = 0; a < 6; ++a) {
                    for (int c = 0; c <= a; ++c)
                        acc[LIDX(6 * f + a, 6 * f + c)][l] = acc[LIDX(6 * f + a, 6 * f + c)][l] + B.Hcc[f][LIDX(a, c)];
                    acc[nl + 6 * f + a][l] = acc[nl + 6 * f + a][l] - B.gc[6 * f + a];
                }
            for (int a = 0; a < nc; ++a) {
                for (int c = 0; c <= a; ++c) {
                    double s = acc[LIDX(a, c)][l];
                    s = fma(-Y[a][0], B.Hcp[c][0], s);
                    s = fma(-Y[a][1], B.Hcp[c][1], s);
                    acc[LIDX(a, c)][l] = fma(-Y[a][2], B.Hcp[c][2], s);
                }
                double g = acc[nl + a][l];
                g = fma(Y[a][0], B.gp[0], g);
                g = fma(Y[a][1], B.gp[1], g);
                acc[nl + a][l] = fma(Y[a][2], B.gp[2], g);
            }
            acc[nl + nc][l] = acc[nl + nc][l] + B.cost;
        }
    }
    for (int k = 0; k < nl; ++k)
        S[k] = reduce_nt(acc[k]);
    for (int a = 0; a < nc; ++a)
        b[a] = reduce_nt(acc[nl + a]);
    double c2 = reduce_nt(acc[nl + nc]);
    /* pose priors */
    for (int f = 0; f < P->F; ++f) {
        double e[6], Jw[9], Jv[9];
        pose_prior(P, f, R[f], t[f], e, Jw, Jv);
        const double *w = P->w[f];
        for (int half = 0; half < 2; ++half) {
            const double *J = half ? Jv : Jw;
            const int o = 6 * f + 3 * half;
            for (int a = 0; a < 3; ++a) {
                for (int c = 0; c <= a; ++c) {
                    double h = (J[a] * w[3 * half] * J[c] + J[3 + a] * w[3 * half + 1] * J[3 + c]) +
                               J[6 + a] * w[3 * half + 2] * J[6 + c];
                    S[LIDX(o + a, o + c)] = S[LIDX(o + a, o + c)] + h;
                }
                double g = (J[a] * w[3 * half] * e[3 * half] + J[3 + a] * w[3 * half + 1] * e[3 * half + 1]) +
                           J[6 + a] * w[3 * half + 2] * e[3 * half + 2];
                b[o + a] = b[o + a] - g;
            }
        }
        for (int k = 0; k < 6; ++k)
            c2 = c2 + (e[k] * e[k]) * w[k];
    }
    for (int a = 0; a < nc; ++a)
        S[LIDX(a, a)] = S[LIDX(a, a)] + lam;
    *cost2 = c2;
}

static void apply_pose_step(const double R[9], const double t[3], const double *dc, double Rn[9], double tn[3])
{
    double E[9];
    so3_exp(dc, E);
    m3mul(R, E, Rn);
    tn[0] = t[0] + ((R[0] * dc[3] + R[1] * dc[4]) + R[2] * dc[5]);
    tn[1] = t[1] + ((R[3] * dc[3] + R[4] * dc[4]) + R[5] * dc[5]);
    tn[2] = t[2] + ((R[6] * dc[3] + R[7] * dc[4]) + R[8] * dc[5]);
}

/* points step and candidate cost; returns the candidate's total squared Mahalanobis sum (not halved) */
static double step_points(const ba_problem *P, const double R[2][9], const double t[2][3], const double *pts, double lam,
                          const double *dc, const double Rn[2][9], const double tn[2][3], double *pts_new)
{
    const int nc = 6 * P->F;
    static __thread double acc[NT];
    memset(acc, 0, sizeof(acc));
    for (int l = 0; l < NT; ++l) {
        for (int i = l; i < P->m; i += NT) {
            point_blocks B;
            point_linearize(P, R, t, pts, i, &B);
            double Hd[6] = {B.Hpp[0] + lam, B.Hpp[1], B.Hpp[2], B.Hpp[3] + lam, B.Hpp[4], B.Hpp[5] + lam}, Pi[6];
            sym3_inverse(Hd, Pi);
            double v[3] = {B.gp[0], B.gp[1], B.gp[2]};
            for (int a = 0; a < nc; ++a)
                for (int k = 0; k < 3; ++k)
                    v[k] = fma(B.Hcp[a][k], dc[a], v[k]);
            double *pn = pts_new + 3 * i;
            pn[0] = pts[3 * i + 0] - fd3(Pi[0], v[0], Pi[1], v[1], Pi[2], v[2]);
            pn[1] = pts[3 * i + 1] - fd3(Pi[1], v[0], Pi[3], v[1], Pi[4], v[2]);
            pn[2] = pts[3 * i + 2] - fd3(Pi[2], v[0], Pi[4], v[1], Pi[5], v[2]);
            acc[l] = acc[l] + point_cost(P, Rn, tn, pn, i);
        }
    }
    return reduce_nt(acc) + prior_cost(P, Rn, tn);
}

static double total_cost(const ba_problem *P, const double R[2][9], const double t[2][3], const double *pts)
{
    static __thread double acc[NT];
    memset(acc, 0, sizeof(acc));
    for (int l = 0; l < NT; ++l)
        for (int i = l; i < P->m; i += NT)
            acc[l] = acc[l] + point_cost(P, R, t, pts + 3 * i, i);
    return reduce_nt(acc) + prior_cost(P, R, t);
}

/* marginal covariances at the estimate: Sinv = S^-1 (full nc x nc, row-major), per-point 3x3 */
static int covariances(const ba_problem *P, const double R[2][9], const double t[2][3], const double *pts, double *Sinv,
                       double *point_cov)
{
    const int nc = 6 * P->F;
    double S[NLOW(MAXC)], b[MAXC], c2;
    build_schur(P, R, t, pts, 0.0, S, b, &c2);
    if (!chol_packed(S, nc))
        return 0;
    for (int j = 0; j < nc; ++j) {
        double e[MAXC] = {0};
        e[j] = 1.0;
        chol_solve(S, nc, e);
        for (int i = 0; i < nc; ++i)
            Sinv[i * nc + j] = e[i];
    }
    if (point_cov) {
        for (int i = 0; i < P->m; ++i) {
            point_blocks B;
            point_linearize(P, R, t, pts, i, &B);
            double Pi[6];
            sym3_inverse(B.Hpp, Pi);
            double Pf[9] = {Pi[0], Pi[1], Pi[2], Pi[1], Pi[3], Pi[4], Pi[2], Pi[4], Pi[5]};
            double G[MAXC][3]; /* Hcp P */
            for (int a = 0; a < nc; ++a)
                for (int k = 0; k < 3; ++k)
                    G[a][k] = fd3(B.Hcp[a][0], Pf[k], B.Hcp[a][1], Pf[3 + k], B.Hcp[a][2], Pf[6 + k]);
            double SG[MAXC][3];
            for (int a = 0; a < nc; ++a)
                for (int k = 0; k < 3; ++k) {
                    double s = 0.0;
                    for (int c = 0; c < nc; ++c)
                        s = fma(Sinv[a * nc + c], G[c][k], s);
                    SG[a][k] = s;
                }
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k) {
                    double s = 0.0;
                    for (int a = 0; a < nc; ++a)
                        s = fma(G[a][r], SG[a][k], s);
                    point_cov[9 * i + 3 * r + k] = Pf[3 * r + k] + s;
                }
        }
    }
    return 1;
}

void orc_refine_params_default(orc_refine_params *p)
{
    p->max_iterations = 100;
    p->reserved = 0;
    p->lambda_initial = 1e-5;
    p->lambda_factor = 10.0;
    p->lambda_upper = 1e5;
    p->rel_tol = 1e-12;
    p->abs_tol = 1e-12;
    p->anchor_sigma[0] = p->anchor_sigma[1] = 1e-5;
    p->pose_sigma[0] = p->pose_sigma[1] = 1e-2;
    p->point_sigma = 1e-2;
}

/* Levenberg-Marquardt on the problem; R/t/pts hold the guess on entry and the estimate on return */
static int ba_solve(const ba_problem *P, const orc_refine_params *prm, double R[2][9], double t[2][3], double *pts,
                    double *error, int *iterations)
{
    const int nc = 6 * P->F;
    double *pts_new = (double *)malloc(sizeof(double) * 3 * (size_t)(P->m > 0 ? P->m : 1));
    double lam = prm->lambda_initial;
    double cur = total_cost(P, R, t, pts);
    int it = 0, ok = isfinite(cur);
    while (ok && it < prm->max_iterations) {
        double S[NLOW(MAXC)], b[MAXC], c2;
        build_schur(P, R, t, pts, lam, S, b, &c2);
        int accepted = 0, solved = 0;
        double cand = 0.0;
        if (chol_packed(S, nc)) {
            solved = 1;
            chol_solve(S, nc, b);
            double Rn[2][9], tn[2][3];
            for (int f = 0; f < P->F; ++f)
                apply_pose_step(R[f], t[f], b + 6 * f, Rn[f], tn[f]);
            cand = step_points(P, R, t, pts, lam, b, Rn, tn, pts_new);
            if (cand <= cur) {
                accepted = 1;
                memcpy(R, Rn, sizeof(Rn));
                memcpy(t, tn, sizeof(tn));
                memcpy(pts, pts_new, sizeof(double) * 3 * (size_t)P->m);
            }
        }
        ++it;
        if (accepted) {
            double dec = 0.5 * (cur - cand);
            int done = dec <= prm->abs_tol || dec <= prm->rel_tol * (0.5 * cur);
            cur = cand;
            lam = lam / prm->lambda_factor;
            if (done)
                break;
        } else {
            /* a trial that lands within the tolerances ABOVE the current error: the estimate is at the minimum to
             * rounding; raising lambda would only replay the same comparison */
            double inc = 0.5 * (cand - cur);
            if (solved && (inc <= prm->abs_tol || inc <= prm->rel_tol * (0.5 * cur)))
                break;
            lam = lam * prm->lambda_factor;
            if (lam > prm->lambda_upper)
                break;
        }
    }
    free(pts_new);
    *error = 0.5 * cur;
    *iterations = it;
    return ok;
}

static void cov2_to_info(const double *cov, int m, double *info)
{
    for (int i = 0; i < m; ++i) {
        if (!cov) {
            info[3 * i] = 1.0, info[3 * i + 1] = 0.0, info[3 * i + 2] = 1.0;
            continue;
        }
        const double a = cov[4 * i], b = 0.5 * (cov[4 * i + 1] + cov[4 * i + 2]), d = cov[4 * i + 3];
        double id = 1.0 / (a * d - b * b);
        info[3 * i] = d * id, info[3 * i + 1] = -(b * id), info[3 * i + 2] = a * id;
    }
}

static void set_K(ba_problem *P, const double K[9])
{
    P->fx = K[0], P->sk = K[1], P->cx = K[2], P->fy = K[4], P->cy = K[5];
}

int orc_sfm_refine(const double *p1, const double *cov1, const double *p2, const double *cov2, int m, const double K[9],
                   const double R_guess[9], const double t_guess[3], const double *points_guess,
                   const orc_refine_params *prm, double R_out[9], double t_out[3], double pose_cov[36], double *points,
                   double *point_cov, double *error, int *iterations)
{
    if (m < 1)
        return 0;
    ba_problem P;
    memset(&P, 0, sizeof(P));
    P.F = 2, P.m = m;
    set_K(&P, K);
    static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(P.R0[0], I3, sizeof(I3)); /* camera 1 = origin (sfm-refine.cpp:55) */
    memcpy(P.R0[1], R_guess, sizeof(I3));
    memcpy(P.t0[1], t_guess, 3 * sizeof(double));
    for (int k = 0; k < 6; ++k) { /* the reference's first three diagonal entries meet GTSAM's rotation block */
        P.w[0][k] = 1.0 / (prm->anchor_sigma[k / 3] * prm->anchor_sigma[k / 3]);
        P.w[1][k] = 1.0 / (prm->pose_sigma[k / 3] * prm->pose_sigma[k / 3]);
    }
    double *pinfo = (double *)malloc(sizeof(double) * 6 * (size_t)m);
    double *o1 = (double *)malloc(sizeof(double) * 3 * (size_t)m), *o2 = (double *)malloc(sizeof(double) * 3 * (size_t)m);
    const double wp = 1.0 / (prm->point_sigma * prm->point_sigma);
    for (int i = 0; i < m; ++i) {
        double *L = pinfo + 6 * i;
        L[0] = L[3] = L[5] = wp;
        L[1] = L[2] = L[4] = 0.0;
    }
    cov2_to_info(cov1, m, o1);
    cov2_to_info(cov2, m, o2);
    P.pts0 = points_guess, P.pinfo = pinfo, P.obs[0] = p1, P.obs[1] = p2, P.oinfo[0] = o1, P.oinfo[1] = o2;
    double R[2][9], t[2][3];
    memcpy(R, P.R0, sizeof(R));
    memcpy(t, P.t0, sizeof(t));
    memcpy(points, points_guess, sizeof(double) * 3 * (size_t)m);
    int ok = ba_solve(&P, prm, R, t, points, error, iterations);
    if (ok) {
        double Sinv[144];
        ok = covariances(&P, R, t, points, Sinv, point_cov);
        if (ok && pose_cov)
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c)
                    pose_cov[6 * r + c] = Sinv[(6 + r) * 12 + (6 + c)];
    }
    memcpy(R_out, R[1], 9 * sizeof(double));
    memcpy(t_out, t[1], 3 * sizeof(double));
    free(pinfo), free(o1), free(o2);
    return ok;
}

int orc_pnp_refine(const double *world, const double *world_cov, const double *img, const double *img_cov, int m,
                   const double K[9], const double R_guess[9], const double t_guess[3], const orc_refine_params *prm,
                   double R_out[9], double t_out[3], double pose_cov[36], double *error, int *iterations)
{
    if (m < 1)
        return 0;
    ba_problem P;
    memset(&P, 0, sizeof(P));
    P.F = 1, P.m = m;
    set_K(&P, K);
    memcpy(P.R0[0], R_guess, 9 * sizeof(double));
    memcpy(P.t0[0], t_guess, 3 * sizeof(double));
    for (int k = 0; k < 6; ++k)
        P.w[0][k] = 1.0 / (prm->pose_sigma[k / 3] * prm->pose_sigma[k / 3]);
    double *pinfo = (double *)malloc(sizeof(double) * 6 * (size_t)m), *o1 = (double *)malloc(sizeof(double) * 3 * (size_t)m);
    double *pts = (double *)malloc(sizeof(double) * 3 * (size_t)m);
    int ok = 1;
    for (int i = 0; i < m; ++i) {
        const double *C = world_cov + 9 * i;
        double a[6] = {C[0], 0.5 * (C[1] + C[3]), 0.5 * (C[2] + C[6]), C[4], 0.5 * (C[5] + C[7]), C[8]};
        ok &= sym3_inverse(a, pinfo + 6 * i);
    }
    cov2_to_info(img_cov, m, o1);
    P.pts0 = world, P.pinfo = pinfo, P.obs[0] = img, P.oinfo[0] = o1;
    double R[2][9], t[2][3];
    memcpy(R, P.R0, sizeof(R));
    memcpy(t, P.t0, sizeof(t));
    memcpy(pts, world, sizeof(double) * 3 * (size_t)m);
    if (ok)
        ok = ba_solve(&P, prm, R, t, pts, error, iterations);
    if (ok) {
        double Sinv[36];
        ok = covariances(&P, R, t, pts, Sinv, NULL);
        if (ok && pose_cov)
            memcpy(pose_cov, Sinv, sizeof(Sinv));
    }
    memcpy(R_out, R[0], 9 * sizeof(double));
    memcpy(t_out, t[0], 3 * sizeof(double));
    free(pinfo), free(o1), free(pts);
    return ok;
}

/* ba_frame_pose_and_point for one or two frames in its general form (VisualOdometer::track_refine,
 * front-end/visual-odometer.cpp:618-800): every frame has its own guess and diagonal prior (variance <= 0: none), a
 * point covariance with a first entry <= 0 means "no prior", obs_valid marks which frame sees which point. */
int orc_ba_refine(int n_frames, int m, const double K[9], const double *frame_pose, const double *frame_prior_var,
                  const double *points_guess, const double *point_prior_cov, const double *const obs[2],
                  const double *const obs_cov[2], const uint8_t *const obs_valid[2], const orc_refine_params *prm,
                  double *R_out, double *t_out, double *pose_cov_out, double *points, double *point_cov, double *error,
                  int *iterations)
{
    if (m < 1 || (n_frames != 1 && n_frames != 2))
        return 0;
    ba_problem P;
    memset(&P, 0, sizeof(P));
    P.F = n_frames, P.m = m;
    set_K(&P, K);
    double *pinfo = (double *)calloc(6 * (size_t)m, sizeof(double));
    double *oi[2] = {NULL, NULL};
    for (int f = 0; f < n_frames; ++f) {
        memcpy(P.R0[f], frame_pose + 12 * f, 9 * sizeof(double));
        memcpy(P.t0[f], frame_pose + 12 * f + 9, 3 * sizeof(double));
        for (int k = 0; k < 6; ++k) {
            const double v = frame_prior_var[6 * f + k];
            P.w[f][k] = v > 0.0 ? 1.0 / v : 0.0;
        }
        oi[f] = (double *)malloc(sizeof(double) * 3 * (size_t)m);
        cov2_to_info(obs_cov[f], m, oi[f]);
        if (obs_valid[f])
            for (int i = 0; i < m; ++i)
                if (!obs_valid[f][i])
                    oi[f][3 * i] = oi[f][3 * i + 1] = oi[f][3 * i + 2] = 0.0;
        P.obs[f] = obs[f];
        P.oinfo[f] = oi[f];
    }
    if (point_prior_cov)
        for (int i = 0; i < m; ++i) {
            const double *C = point_prior_cov + 9 * i;
            if (C[0] > 0.0) {
                double a[6] = {C[0], 0.5 * (C[1] + C[3]), 0.5 * (C[2] + C[6]), C[4], 0.5 * (C[5] + C[7]), C[8]};
                sym3_inverse(a, pinfo + 6 * i);
            }
        }
    P.pts0 = points_guess, P.pinfo = pinfo;
    double R[2][9], t[2][3];
    memcpy(R, P.R0, sizeof(R));
    memcpy(t, P.t0, sizeof(t));
    memcpy(points, points_guess, sizeof(double) * 3 * (size_t)m);
    int ok = ba_solve(&P, prm, R, t, points, error, iterations);
    if (ok) {
        double Sinv[144];
        const int nc = 6 * n_frames;
        ok = covariances(&P, R, t, points, Sinv, point_cov);
        if (ok && pose_cov_out)
            for (int f = 0; f < n_frames; ++f)
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 6; ++c)
                        pose_cov_out[36 * f + 6 * r + c] = Sinv[(6 * f + r) * nc + (6 * f + c)];
    }
    for (int f = 0; f < n_frames; ++f) {
        memcpy(R_out + 9 * f, R[f], 9 * sizeof(double));
        memcpy(t_out + 3 * f, t[f], 3 * sizeof(double));
    }
    free(pinfo), free(oi[0]), free(oi[1]);
    return ok;
}
