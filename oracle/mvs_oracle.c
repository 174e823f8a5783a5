/*
 * mvs_oracle.c -- CPU ORACLE (test infrastructure; see mvs_oracle.h for the rules).
 *
 * Plain C restatement of the reference's two-view-geometry path.  Nothing here is
 * copied from the reference; every function cites the reference file:line whose
 * behaviour it restates.  Build: gcc -O2 -ffp-contract=off -mfma (oracle/Makefile).
 *
 * Arithmetic contract: every floating-point operation below is a single IEEE-754
 * binary64 operation in the order written.  fma() appears only where the contract
 * (DESIGN.md "arithmetic contract") says "fused"; everything else is separate
 * mul / add / sub / div / sqrt.  The HIP kernels implement the same contract
 * independently, which is what makes bit-exact GPU<->oracle comparison possible.
 */
#include "mvs_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* system-config.hpp:6-14 */
#define ORC_EPSILON DBL_EPSILON
#define ORC_TOLERANCE (DBL_EPSILON * 1000.0)
#define ORC_TAYLOR_THRESHOLD 1e-5
#define ORC_INFINITY (DBL_MAX / 10.0)

static __thread orc_counters g_cnt;

void orc_counters_reset(void) { memset(&g_cnt, 0, sizeof(g_cnt)); }
void orc_counters_get(orc_counters *out) { *out = g_cnt; }

/* ------------------------------------------------------------------------- */
/* small fixed-size helpers (Eigen fixed-size semantics: coefficient sums are  */
/* evaluated left to right, (a0*b0 + a1*b1) + a2*b2, no vectorisation:         */
/* SConstruct:86 EIGEN_DONT_VECTORIZE)                                         */
/* ------------------------------------------------------------------------- */
static double dot3(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

static void mat3_mul(const double *A, const double *B, double *C)
{
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            T[i * 3 + j] = (A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j]) + A[i * 3 + 2] * B[2 * 3 + j];
    memcpy(C, T, sizeof(T));
}

static void mat3_vec(const double *A, const double *v, double *o)
{
    double t[3];
    for (int i = 0; i < 3; ++i)
        t[i] = dot3(A + 3 * i, v);
    o[0] = t[0];
    o[1] = t[1];
    o[2] = t[2];
}

static void mat3_transpose(const double *A, double *T)
{
    double t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            t[j * 3 + i] = A[i * 3 + j];
    memcpy(T, t, sizeof(t));
}

/* Eigen fixed 3x3 determinant (bruteforce_det3_helper order) */
static double mat3_det(const double *m)
{
    double h0 = m[0] * (m[4] * m[8] - m[5] * m[7]);
    double h1 = m[1] * (m[3] * m[8] - m[5] * m[6]);
    double h2 = m[2] * (m[3] * m[7] - m[4] * m[6]);
    return (h0 - h1) + h2;
}

/* ------------------------------------------------------------------------- */
/* math/lie-group                                                              */
/* ------------------------------------------------------------------------- */

/* lie-group.hpp:84-96  Gram-Schmidt on the rows; row 1 is NOT re-normalised. */
void orc_so3_rectify(double R[9])
{
    double u0[3] = {R[0], R[1], R[2]};
    double n = sqrt((u0[0] * u0[0] + u0[1] * u0[1]) + u0[2] * u0[2]);
    u0[0] = u0[0] / n;
    u0[1] = u0[1] / n;
    u0[2] = u0[2] / n;
    double u1[3] = {R[3], R[4], R[5]};
    double d = dot3(u1, u0);
    u1[0] = u1[0] - d * u0[0];
    u1[1] = u1[1] - d * u0[1];
    u1[2] = u1[2] - d * u0[2];
    double u2[3];
    u2[0] = u0[1] * u1[2] - u0[2] * u1[1];
    u2[1] = u0[2] * u1[0] - u0[0] * u1[2];
    u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
    R[0] = u0[0]; R[1] = u0[1]; R[2] = u0[2];
    R[3] = u1[0]; R[4] = u1[1]; R[5] = u1[2];
    R[6] = u2[0]; R[7] = u2[1]; R[8] = u2[2];
}

/* lie-group.hpp:31-36 */
void orc_so3_from_matrix(const double M[9], double R[9])
{
    memmove(R, M, 9 * sizeof(double));
    orc_so3_rectify(R);
}

/* lie-group.hpp:41-56  (roll, pitch, yaw): R = Rz * Ry * Rx */
void orc_so3_from_rpy(double roll, double pitch, double yaw, double R[9])
{
    double Rx[9] = {1, 0, 0, 0, cos(roll), -sin(roll), 0, sin(roll), cos(roll)};
    double Ry[9] = {cos(pitch), 0, sin(pitch), 0, 1, 0, -sin(pitch), 0, cos(pitch)};
    double Rz[9] = {cos(yaw), -sin(yaw), 0, sin(yaw), cos(yaw), 0, 0, 0, 1};
    double T[9];
    mat3_mul(Rz, Ry, T);
    mat3_mul(T, Rx, R);
}

static void skew3(const double *v, double *K)
{ /* lie-group.cpp:5-13 */
    K[0] = 0;     K[1] = -v[2]; K[2] = v[1];
    K[3] = v[2];  K[4] = 0;     K[5] = -v[0];
    K[6] = -v[1]; K[7] = v[0];  K[8] = 0;
}

/* lie-group.cpp:15-32 (Taylor branch at theta < epsilon) */
void orc_rodrigues(const double v[3], double R[9])
{
    double theta = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    double A, B;
    if (theta < ORC_EPSILON) {
        A = 1.0 - (theta * theta) / 6.0;
        B = 0.5 - (theta * theta) / 24.0;
    } else {
        A = sin(theta) / theta;
        B = (1.0 - cos(theta)) / (theta * theta);
    }
    double K[9], BK[9], BKK[9];
    skew3(v, K);
    for (int i = 0; i < 9; ++i)
        BK[i] = B * K[i];
    mat3_mul(BK, K, BKK);
    for (int i = 0; i < 9; ++i) {
        double id = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
        R[i] = (id + A * K[i]) + BKK[i];
    }
}

/* lie-group.hpp:138-162 */
void orc_so3_ln(const double R[9], double w[3])
{
    double c = 0.5 * (((R[0] + R[4]) + R[8]) - 1.0);
    if (c < -1.0) c = -1.0;
    if (c > 1.0) c = 1.0;
    double theta = acos(c);
    double v[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    double A;
    if (theta < ORC_TAYLOR_THRESHOLD)
        A = (1.0 + (theta * theta) / 6.0) * 0.5;
    else
        A = 0.5 * theta / sin(theta);
    w[0] = v[0] * A;
    w[1] = v[1] * A;
    w[2] = v[2] * A;
}

/* lie-group.hpp:75-79,212-216:  RT = SO3(R^T) (rectified), t' = -(RT * t) */
void orc_se3_inverse(const double R[9], const double t[3], double Ro[9], double to[3])
{
    double RT[9], tt[3];
    mat3_transpose(R, RT);
    orc_so3_rectify(RT);
    mat3_vec(RT, t, tt);
    memcpy(Ro, RT, sizeof(RT));
    to[0] = -tt[0];
    to[1] = -tt[1];
    to[2] = -tt[2];
}

/* lie-group.hpp:118-122,229-234 */
void orc_se3_compose(const double Ra[9], const double ta[3], const double Rb[9], const double tb[3], double Ro[9],
                     double to[3])
{
    double R[9], rt[3];
    mat3_mul(Ra, Rb, R);
    orc_so3_rectify(R);
    mat3_vec(Ra, tb, rt);
    double t0 = rt[0] + ta[0], t1 = rt[1] + ta[1], t2 = rt[2] + ta[2];
    memcpy(Ro, R, sizeof(R));
    to[0] = t0;
    to[1] = t1;
    to[2] = t2;
}

/* lie-group.hpp:245-269 */
void orc_se3_ln(const double R[9], const double t[3], double se3[6])
{
    double w[3];
    orc_so3_ln(R, w);
    double theta = sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
    double G;
    if (theta < ORC_TAYLOR_THRESHOLD) {
        G = 1.0 / 12.0 + (theta * theta) / 720.0;
    } else {
        double A = sin(theta) / theta;
        double B = (1.0 - cos(theta)) / (theta * theta);
        G = (1.0 - 0.5 * A / B) / (theta * theta);
    }
    double K[9], GK[9], GKK[9], Vinv[9];
    skew3(w, K);
    for (int i = 0; i < 9; ++i)
        GK[i] = G * K[i];
    mat3_mul(GK, K, GKK);
    for (int i = 0; i < 9; ++i) {
        double id = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
        Vinv[i] = (id - 0.5 * K[i]) + GKK[i];
    }
    double u[3];
    mat3_vec(Vinv, t, u);
    se3[0] = u[0]; se3[1] = u[1]; se3[2] = u[2];
    se3[3] = w[0]; se3[4] = w[1]; se3[5] = w[2];
}

/* lie-group.hpp:275-299 */
void orc_se3_exp(const double se3[6], double R[9], double t[3])
{
    const double *u = se3, *w = se3 + 3;
    double theta = sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
    double B, C;
    if (theta < ORC_TAYLOR_THRESHOLD) {
        B = 0.5 - (theta * theta) / 24.0;
        C = 1.0 / 6.0 - (theta * theta) / 120.0;
    } else {
        double A = sin(theta) / theta;
        B = (1.0 - cos(theta)) / (theta * theta);
        C = (1.0 - A) / (theta * theta);
    }
    orc_rodrigues(w, R);
    double K[9], CK[9], CKK[9], V[9];
    skew3(w, K);
    for (int i = 0; i < 9; ++i)
        CK[i] = C * K[i];
    mat3_mul(CK, K, CKK);
    for (int i = 0; i < 9; ++i) {
        double id = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;
        V[i] = (id + B * K[i]) + CKK[i];
    }
    mat3_vec(V, u, t);
}

/* ------------------------------------------------------------------------- */
/* cv::SVDecomp restated (one-sided Jacobi, OpenCV modules/core/src/lapack.cpp)*/
/* ------------------------------------------------------------------------- */

/* OpenCV's RNG (multiply-with-carry), used only to complete U when a singular
 * value is (numerically) zero. */
static uint32_t cvrng_next(uint64_t *state)
{
    *state = (uint64_t)(uint32_t)(*state) * 4164903690ULL + (uint32_t)(*state >> 32);
    return (uint32_t)(*state);
}

/*
 * jacobi_svd: At is n rows x m cols (row stride m), the rows are orthogonalised.
 * Vt is n x n (may be NULL).  n1 rows of At are normalised into left vectors
 * (rows >= n are generated).  which: 9/3/4 counters only.
 *
 * Contract (fused ops marked F):
 *   W[i]  = chain sd = fma(t, t, sd), k ascending, sd0 = 0                 (F)
 *   p     = chain p  = fma(Ai[k], Aj[k], p), k ascending, p0 = 0           (F)
 *   skip when |p| <= eps * sqrt(a * b), eps = 10 * DBL_EPSILON
 *   p *= 2; beta = a - b; gamma = sqrt(fma(p, p, beta * beta))   [hypot restated, F]
 *   beta < 0 : s = sqrt(((gamma - beta) * 0.5) / gamma); c = p / (gamma * s * 2)
 *   else     : c = sqrt((gamma + beta) / (gamma * 2));   s = p / (gamma * c * 2)
 *   t0 = fma(c, Ai[k], s * Aj[k]); t1 = fma(c, Aj[k], -(s * Ai[k]))        (F)
 *   a = chain fma(t0, t0, a); b = chain fma(t1, t1, b), from 0             (F)
 *   Vt rows i, j get the same t0/t1 update.
 *   sweeps: at most max(m, 30), stop after a sweep without rotation.
 *   afterwards W[i] = sqrt(chain fma(t, t, sd)); selection sort, descending,
 *   strict '<', swapping rows of At and Vt.
 */
/* diagnostics for the divergence analysis in DESIGN.md section 4.3: when set, every 9x9 decomposition appends one
 * 64-bit word per sweep (bit = pair index in (i, j) order, set if the pair was rotated) followed by ~0 */
static __thread uint64_t *g_trace = NULL;
static __thread size_t g_trace_cap = 0, g_trace_len = 0;
void orc_debug_set_jacobi_trace(uint64_t *buf, size_t cap)
{
    g_trace = buf;
    g_trace_cap = cap;
    g_trace_len = 0;
}
size_t orc_debug_jacobi_trace_len(void) { return g_trace_len; }

/* Study switch (tests/contract_sensitivity.py), 0 everywhere else: 1 = the Jacobi inner loops in OpenCV's literal forms
 * (lapack.cpp JacobiSVDImpl_: p += Ai[k]*Aj[k]; gamma = hypot(p, beta); t0 = c*Ai[k] + s*Aj[k]; t1 = -s*Ai[k] + c*Aj[k];
 * a += t0*t0) compiled without contraction, instead of the contract's fused ones. */
static int g_jacobi_form = 0;
void orc_set_jacobi_form(int form) { g_jacobi_form = form; }

static void jacobi_svd(double *At, int m, int n, double *Wout, double *Vt, int n1, int which)
{
    const double eps = DBL_EPSILON * 10.0;
    const double minval = DBL_MIN;
    double W[16];
    int max_iter = m > 30 ? m : 30;
    int64_t rot = 0, pairs = 0;

    for (int i = 0; i < n; ++i) {
        double sd = 0.0;
        for (int k = 0; k < m; ++k) {
            double t = At[i * m + k];
            sd = g_jacobi_form ? sd + t * t : fma(t, t, sd);
        }
        W[i] = sd;
        if (Vt) {
            for (int k = 0; k < n; ++k)
                Vt[i * n + k] = 0.0;
            Vt[i * n + i] = 1.0;
        }
    }

    for (int iter = 0; iter < max_iter; ++iter) {
        int changed = 0;
        uint64_t sweep_bits = 0;
        int pair_idx = -1;
        for (int i = 0; i < n - 1; ++i)
            for (int j = i + 1; j < n; ++j) {
                double *Ai = At + i * m, *Aj = At + j * m;
                double a = W[i], b = W[j], p = 0.0;
                ++pairs;
                ++pair_idx;
                for (int k = 0; k < m; ++k)
                    p = g_jacobi_form ? p + Ai[k] * Aj[k] : fma(Ai[k], Aj[k], p);
                if (fabs(p) <= eps * sqrt(a * b))
                    continue;
                p *= 2.0;
                double beta = a - b;
                double gamma = g_jacobi_form ? hypot(p, beta) : sqrt(fma(p, p, beta * beta));
                double c, s;
                if (beta < 0.0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2.0);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2.0));
                    s = p / (gamma * c * 2.0);
                }
                a = 0.0;
                b = 0.0;
                for (int k = 0; k < m; ++k) {
                    double t0 = g_jacobi_form ? c * Ai[k] + s * Aj[k] : fma(c, Ai[k], s * Aj[k]);
                    double t1 = g_jacobi_form ? -s * Ai[k] + c * Aj[k] : fma(c, Aj[k], -(s * Ai[k]));
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a = g_jacobi_form ? a + t0 * t0 : fma(t0, t0, a);
                    b = g_jacobi_form ? b + t1 * t1 : fma(t1, t1, b);
                }
                W[i] = a;
                W[j] = b;
                changed = 1;
                ++rot;
                sweep_bits |= 1ull << (pair_idx & 63);
                if (Vt) {
                    double *Vi = Vt + i * n, *Vj = Vt + j * n;
                    for (int k = 0; k < n; ++k) {
                        double t0 = g_jacobi_form ? c * Vi[k] + s * Vj[k] : fma(c, Vi[k], s * Vj[k]);
                        double t1 = g_jacobi_form ? -s * Vi[k] + c * Vj[k] : fma(c, Vj[k], -(s * Vi[k]));
                        Vi[k] = t0;
                        Vj[k] = t1;
                    }
                }
            }
        if (which == 9 && g_trace && g_trace_len + 2 <= g_trace_cap)
            g_trace[g_trace_len++] = sweep_bits;
        if (!changed)
            break;
    }
    if (which == 9 && g_trace && g_trace_len + 1 <= g_trace_cap)
        g_trace[g_trace_len++] = ~0ull;
    if (which == 9) { g_cnt.rotations9 += rot; g_cnt.pairs9 += pairs; }
    else if (which == 3) { g_cnt.rotations3 += rot; g_cnt.pairs3 += pairs; }
    else if (which == 4) { g_cnt.rotations4 += rot; g_cnt.pairs4 += pairs; }

    for (int i = 0; i < n; ++i) {
        double sd = 0.0;
        for (int k = 0; k < m; ++k) {
            double t = At[i * m + k];
            sd = g_jacobi_form ? sd + t * t : fma(t, t, sd);
        }
        W[i] = sqrt(sd);
    }

    for (int i = 0; i < n - 1; ++i) {
        int j = i;
        for (int k = i + 1; k < n; ++k)
            if (W[j] < W[k])
                j = k;
        if (i != j) {
            double tw = W[i]; W[i] = W[j]; W[j] = tw;
            if (Vt) {
                for (int k = 0; k < m; ++k) {
                    double t = At[i * m + k]; At[i * m + k] = At[j * m + k]; At[j * m + k] = t;
                }
                for (int k = 0; k < n; ++k) {
                    double t = Vt[i * n + k]; Vt[i * n + k] = Vt[j * n + k]; Vt[j * n + k] = t;
                }
            }
        }
    }
    for (int i = 0; i < n; ++i)
        Wout[i] = W[i];
    if (!Vt)
        return;

    /* left vectors: normalise row i by 1/W[i]; rows with a (numerically) zero
     * singular value are replaced by a pseudo-random vector orthogonalised
     * against the previous rows (two Gram-Schmidt passes with 1-norm rescaling). */
    uint64_t rng = 0x12345678ULL;
    for (int i = 0; i < n1; ++i) {
        double sd = i < n ? W[i] : 0.0;
        for (int ii = 0; ii < 100 && sd <= minval; ++ii) {
            const double val0 = 1.0 / m;
            for (int k = 0; k < m; ++k)
                At[i * m + k] = (cvrng_next(&rng) & 256) != 0 ? val0 : -val0;
            for (int iter = 0; iter < 2; ++iter)
                for (int j = 0; j < i; ++j) {
                    sd = 0.0;
                    for (int k = 0; k < m; ++k)
                        sd += At[i * m + k] * At[j * m + k];
                    double asum = 0.0;
                    for (int k = 0; k < m; ++k) {
                        double t = At[i * m + k] - sd * At[j * m + k];
                        At[i * m + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100.0 ? 1.0 / asum : 0.0;
                    for (int k = 0; k < m; ++k)
                        At[i * m + k] *= asum;
                }
            sd = 0.0;
            for (int k = 0; k < m; ++k) {
                double t = At[i * m + k];
                sd += t * t;
            }
            sd = sqrt(sd);
        }
        double s = sd > minval ? 1.0 / sd : 0.0;
        for (int k = 0; k < m; ++k)
            At[i * m + k] *= s;
    }
}

/* math/svd.hpp:59-72: cv::SVDecomp(A, w, u, vt, MODIFY_A | FULL_UV).
 * m >= n: At = A^T (n x m), rows of Vt are right vectors, u = (normalised At)^T.
 * m <  n: roles swap (the rows of A are orthogonalised), u = Vt^T, vt = completed At. */
void orc_svd(const double *A, int m, int n, double *w, double *u, double *vt)
{
    int at = 0, M = m, N = n;
    if (m < n) {
        at = 1;
        M = n;
        N = m;
    }
    /* temp_u: M x M (urows = M for FULL_UV), first N rows are At; rest zero */
    double *tu = (double *)calloc((size_t)M * M, sizeof(double));
    double *tv = (double *)calloc((size_t)N * N, sizeof(double));
    double tw[16];
    if (!at) {
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < m; ++k)
                tu[i * M + k] = A[k * n + i];
    } else {
        for (int i = 0; i < m; ++i)
            for (int k = 0; k < n; ++k)
                tu[i * M + k] = A[i * n + k];
    }
    int which = (M == 9 && N == 9) ? 9 : (M == 3 && N == 3) ? 3 : (M == 4 && N == 4) ? 4 : 0;
    jacobi_svd(tu, M, N, tw, tv, M, which);
    for (int i = 0; i < N; ++i)
        w[i] = tw[i];
    if (!at) {
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j)
                u[i * m + j] = tu[j * M + i];
        memcpy(vt, tv, sizeof(double) * n * n);
    } else {
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j)
                u[i * m + j] = tv[j * N + i];
        memcpy(vt, tu, sizeof(double) * n * n);
    }
    free(tu);
    free(tv);
}

/* ------------------------------------------------------------------------- */
/* vision/camera.cpp                                                           */
/* ------------------------------------------------------------------------- */

/* camera.cpp:16  K.inverse(): fixed 3x3 cofactor inverse, result(r,c) = cof(c,r) / det */
void orc_mat3_inverse(const double K[9], double Kinv[9])
{
#define M_(r, c) K[((r) % 3) * 3 + ((c) % 3)]
#define COF(i, j) (M_(i + 1, j + 1) * M_(i + 2, j + 2) - M_(i + 1, j + 2) * M_(i + 2, j + 1))
    double c00 = COF(0, 0), c10 = COF(1, 0), c20 = COF(2, 0);
    double det = (c00 * K[0] + c10 * K[3]) + c20 * K[6];
    double invdet = 1.0 / det;
    double out[9];
    out[0] = c00 * invdet;
    out[1] = c10 * invdet;
    out[2] = c20 * invdet;
    out[3] = COF(0, 1) * invdet;
    out[4] = COF(1, 1) * invdet;
    out[5] = COF(2, 1) * invdet;
    out[6] = COF(0, 2) * invdet;
    out[7] = COF(1, 2) * invdet;
    out[8] = COF(2, 2) * invdet;
#undef COF
#undef M_
    memcpy(Kinv, out, sizeof(out));
}

/* camera.cpp:55-79  K_inv * (u, v, 1); the homogeneous coordinate is defined to be
 * exactly 1 (affine intrinsics), only (x, y) are stored. */
void orc_normalize_points(const double Kinv[9], const double *uv, int n, double *xy)
{
    for (int i = 0; i < n; ++i) {
        double u = uv[2 * i], v = uv[2 * i + 1];
        xy[2 * i] = (Kinv[0] * u + Kinv[1] * v) + Kinv[2];
        xy[2 * i + 1] = (Kinv[3] * u + Kinv[4] * v) + Kinv[5];
    }
}

/* camera.cpp:24-37  p_cam = P * X; uv = K * p_cam / z.  returns 0 when behind the camera. */
int orc_project_point(const double K[9], const double Rw2c[9], const double tw2c[3], const double X[3],
                      double uv[2])
{
    double pc[3], ph[3];
    mat3_vec(Rw2c, X, pc);
    pc[0] += tw2c[0];
    pc[1] += tw2c[1];
    pc[2] += tw2c[2];
    if (!(pc[2] > 0))
        return 0;
    mat3_vec(K, pc, ph);
    uv[0] = ph[0] / ph[2];
    uv[1] = ph[1] / ph[2];
    return 1;
}

/* ------------------------------------------------------------------------- */
/* vision/visual-feature.cpp:51-80                                             */
/* ------------------------------------------------------------------------- */
static int hamming(const uint8_t *a, const uint8_t *b, int nbytes)
{
    int d = 0, k = 0;
    for (; k + 8 <= nbytes; k += 8) {
        uint64_t x, y;
        memcpy(&x, a + k, 8);
        memcpy(&y, b + k, 8);
        d += __builtin_popcountll(x ^ y);
    }
    for (; k < nbytes; ++k)
        d += __builtin_popcount((unsigned)(a[k] ^ b[k]));
    return d;
}

static int match_less(const void *pa, const void *pb)
{
    const orc_match *a = (const orc_match *)pa, *b = (const orc_match *)pb;
    if (a->distance < b->distance) return -1;
    if (a->distance > b->distance) return 1;
    return (a->queryIdx > b->queryIdx) - (a->queryIdx < b->queryIdx);
}

/*
 * knnMatch(query = vf2, train = vf1, k = 2) + Lowe ratio + max_dist + sort.
 * 2-NN rule (OpenCV brute force): scan train rows in index order, insert with
 * strict '<' so equal distances keep the smaller train index first.
 * Output order is canonical: (distance, queryIdx) ascending (the reference's
 * std::partition + std::sort leave ties unspecified).
 */
int orc_match_visual_features(const uint8_t *train_desc, int n_train, const uint8_t *query_desc, int n_query,
                              int desc_bytes, double ratio, double max_dist, orc_match *out)
{
    if (n_train < 2 || n_query < 1 || desc_bytes < 1)
        return -1; /* visual-feature.cpp:56 assert / :67 UB */
    int n_out = 0;
    for (int q = 0; q < n_query; ++q) {
        int d0 = 0x7fffffff, d1 = 0x7fffffff, i0 = -1, i1 = -1;
        const uint8_t *qd = query_desc + (size_t)q * desc_bytes;
        for (int t = 0; t < n_train; ++t) {
            int d = hamming(qd, train_desc + (size_t)t * desc_bytes, desc_bytes);
            if (d < d1) {
                if (d < d0) {
                    d1 = d0; i1 = i0;
                    d0 = d;  i0 = t;
                } else {
                    d1 = d; i1 = t;
                }
            }
        }
        (void)i1;
        float f0 = (float)d0, f1 = (float)d1;
        int check1 = ((double)f0 < ratio * (double)f1);              /* visual-feature.cpp:67 */
        int check2 = (max_dist < 0) || ((double)f0 <= max_dist);     /* :68 */
        if (check1 && check2) {
            out[n_out].queryIdx = q;
            out[n_out].trainIdx = i0;
            out[n_out].imgIdx = 0;
            out[n_out].distance = f0;
            ++n_out;
        }
    }
    qsort(out, (size_t)n_out, sizeof(orc_match), match_less);
    return n_out;
}

/* ------------------------------------------------------------------------- */
/* vision/fundamental-matrix.cpp                                               */
/* ------------------------------------------------------------------------- */

/* fundamental-matrix.cpp:18-54.  p: 8 x (x, y) with homogeneous 1 (so the mean of
 * the third coordinate is exactly 1 and its centred value exactly 0).
 * scale = sqrt(2) / MEAN distance.  returns 0 if scale <= epsilon
 * (the reference asserts, :45). */
static int find_normalization_transform(const double *p, double *np, double *scale_out, double *mean_out)
{
    double mx = 0.0, my = 0.0;
    for (int i = 0; i < 8; ++i) {
        mx += p[2 * i];
        my += p[2 * i + 1];
    }
    mx *= 0.125;
    my *= 0.125;
    double scale = 0.0;
    for (int i = 0; i < 8; ++i) {
        double dx = p[2 * i] - mx, dy = p[2 * i + 1] - my;
        np[2 * i] = dx;
        np[2 * i + 1] = dy;
        scale += sqrt(dx * dx + dy * dy);
    }
    scale *= 0.125;
    if (!(scale > ORC_EPSILON))
        return 0;
    scale = 1.4142135623730951 / scale; /* sqrt(2.0) / scale */
    for (int i = 0; i < 16; ++i)
        np[i] *= scale;
    *scale_out = scale;
    mean_out[0] = mx;
    mean_out[1] = my;
    return 1;
}

/* 3x3 SVD, rank-2 enforcement, recomposition (fundamental-matrix.cpp:127-136).
 * F = u * diag(w0, w1, 0) * vt  ==  F_ij = (u_i0*w0)*vt_0j + (u_i1*w1)*vt_1j. */
static void enforce_rank2(double F[9])
{
    double w[3], u[9], vt[9];
    orc_svd(F, 3, 3, w, u, vt);
    for (int i = 0; i < 3; ++i) {
        double a = u[i * 3 + 0] * w[0], b = u[i * 3 + 1] * w[1];
        for (int j = 0; j < 3; ++j)
            F[i * 3 + j] = a * vt[0 * 3 + j] + b * vt[1 * 3 + j];
    }
}

/* fundamental-matrix.cpp:56-140 + 204-267 */
int orc_find_fundamental_matrix(const double p1[16], const double p2[16], double F[9])
{
    double n1[16], n2[16], s1, s2, m1[2], m2[2];
    if (!find_normalization_transform(p1, n1, &s1, m1))
        return 0;
    if (!find_normalization_transform(p2, n2, &s2, m2))
        return 0;

    double A[8][9];
    for (int i = 0; i < 8; ++i) { /* :78-87 */
        double x1 = n1[2 * i], y1 = n1[2 * i + 1], x2 = n2[2 * i], y2 = n2[2 * i + 1];
        A[i][0] = x2 * x1; A[i][1] = x2 * y1; A[i][2] = x2;
        A[i][3] = y2 * x1; A[i][4] = y2 * y1; A[i][5] = y2;
        A[i][6] = x1;      A[i][7] = y1;      A[i][8] = 1.0;
    }
    double AtA[81]; /* :104-111, sequential k, separate mul/add */
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 8; ++k)
                acc += A[k][i] * A[k][j];
            AtA[i * 9 + j] = acc;
        }
    double w9[9], u9[81], vt9[81];
    orc_svd(AtA, 9, 9, w9, u9, vt9); /* :115 */
    double Fn[9];
    for (int i = 0; i < 9; ++i)
        Fn[i] = vt9[8 * 9 + i]; /* :117-124 */
    enforce_rank2(Fn);

    /* :245  F = T2^T * Fn * T1 with T = [s 0 -m0*s; 0 s -m1*s; 0 0 1]; structural
     * zeros dropped (they contribute exact zeros). */
    double tx1 = -m1[0] * s1, ty1 = -m1[1] * s1, tx2 = -m2[0] * s2, ty2 = -m2[1] * s2;
    double G[9];
    for (int j = 0; j < 3; ++j) {
        G[0 * 3 + j] = s2 * Fn[0 * 3 + j];
        G[1 * 3 + j] = s2 * Fn[1 * 3 + j];
        G[2 * 3 + j] = (tx2 * Fn[0 * 3 + j] + ty2 * Fn[1 * 3 + j]) + Fn[2 * 3 + j];
    }
    for (int i = 0; i < 3; ++i) {
        F[i * 3 + 0] = G[i * 3 + 0] * s1;
        F[i * 3 + 1] = G[i * 3 + 1] * s1;
        F[i * 3 + 2] = (G[i * 3 + 0] * tx1 + G[i * 3 + 1] * ty1) + G[i * 3 + 2];
    }
    return 1;
}

/* ------------------------------------------------------------------------- */
/* sampler: Philox4x32-10 (Salmon et al., SC'11), counter = (hyp, block, 0, 0),  */
/* key = (lo32(seed), hi32(seed))                                              */
/* ------------------------------------------------------------------------- */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 8 distinct indices in [0, M): draw k picks slot (w_k * (M - k)) >> 32 among the
 * not-yet-chosen indices (ascending), i.e. a partial Fisher-Yates without the array. */
void orc_sample8(uint64_t seed, uint32_t hyp, int M, int sampler, int idx[8])
{
    if (sampler == ORC_SAMPLER_IDENTITY) {
        for (int k = 0; k < 8; ++k)
            idx[k] = k;
        return;
    }
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[8];
    uint32_t c0[4] = {hyp, 0, 0, 0}, c1[4] = {hyp, 1, 0, 0};
    orc_philox4x32_10(c0, key, w);
    orc_philox4x32_10(c1, key, w + 4);
    int sorted[8];
    for (int k = 0; k < 8; ++k) {
        uint32_t r = (uint32_t)(((uint64_t)w[k] * (uint32_t)(M - k)) >> 32);
        int pos = 0;
        for (int t = 0; t < k; ++t)
            if (r >= (uint32_t)sorted[t]) {
                ++r;
                pos = t + 1;
            }
        for (int t = k; t > pos; --t)
            sorted[t] = sorted[t - 1];
        sorted[pos] = (int)r;
        idx[k] = (int)r;
    }
}

/* ------------------------------------------------------------------------- */
/* vision/estimator-RANSAC.cpp                                                 */
/* ------------------------------------------------------------------------- */

/* estimator-RANSAC.cpp:100-129.  r = |p2^T F p1| (homogeneous 1), fused form:
 *   u_j = fma(x2, F0j, fma(y2, F1j, F2j));  r = |fma(u0, x1, fma(u1, y1, u2))|
 * inlier iff r < max_error_sq (strict); residual += r in index order. */
/* Study switch (tests/contract_sensitivity.py), 0 everywhere else: 1 = the residual as the reference's own expression
 * evaluates it -- `p2.transpose() * F * p1` (estimator-RANSAC.cpp:114) is (p2^T F) p1 with Eigen's coefficient-wise
 * left-to-right sums and no fused multiply-add (x86-64 build without -mfma, SConstruct:86 EIGEN_DONT_VECTORIZE).  It
 * measures how far the contract's fused form can move an inlier decision or a winner away from the reference's. */
static int g_residual_form = 0;
void orc_set_residual_form(int form) { g_residual_form = form; }

int orc_count_inliers(const double *p1, const double *p2, int M, const double F[9], double max_error_sq,
                      uint8_t *mask, double *residual)
{
    int count = 0;
    double res = 0.0;
    const int unfused = g_residual_form == 1;
    for (int i = 0; i < M; ++i) {
        double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        double r;
        if (unfused) {
            const double v0 = (x2 * F[0] + y2 * F[3]) + F[6], v1 = (x2 * F[1] + y2 * F[4]) + F[7],
                         v2 = (x2 * F[2] + y2 * F[5]) + F[8];
            r = fabs((v0 * x1 + v1 * y1) + v2);
        } else {
        double u0 = fma(x2, F[0], fma(y2, F[3], F[6]));
        double u1 = fma(x2, F[1], fma(y2, F[4], F[7]));
        double u2 = fma(x2, F[2], fma(y2, F[5], F[8]));
        r = fabs(fma(u0, x1, fma(u1, y1, u2)));
        }
        if (r < max_error_sq) {
            ++count;
            res += r;
            if (mask) mask[i] = 1;
        } else {
            if (mask) mask[i] = 0;
        }
    }
    g_cnt.score_evals += M;
    *residual = res;
    return count;
}

/* estimator-RANSAC.cpp:16-90 with a sampler and H iterations (the reference runs one
 * un-shuffled iteration, sfm-solve.cpp:67). */
int orc_ransac_fundamental(const double *p1, const double *p2, int M, double max_error_sq, int H, int sampler,
                           uint64_t seed, double F[9], uint8_t *mask, int *best_hyp, int *best_count,
                           double *best_residual, int32_t *per_hyp_count, double *per_hyp_residual)
{
    *best_hyp = -1;
    *best_count = 0;
    *best_residual = ORC_INFINITY;
    if (M < 8)
        return 0; /* :25-29 */
    double residual_best = ORC_INFINITY;
    int count_best = 0, hyp_best = -1;
    double Fbest[9] = {0};
    for (int h = 0; h < H; ++h) {
        int idx[8];
        double s1[16], s2[16], Fp[9];
        orc_sample8(seed, (uint32_t)h, M, sampler, idx);
        for (int j = 0; j < 8; ++j) {
            s1[2 * j] = p1[2 * idx[j]];
            s1[2 * j + 1] = p1[2 * idx[j] + 1];
            s2[2 * j] = p2[2 * idx[j]];
            s2[2 * j + 1] = p2[2 * idx[j] + 1];
        }
        ++g_cnt.hypotheses;
        if (!orc_find_fundamental_matrix(s1, s2, Fp)) { /* :58-62 */
            if (per_hyp_count) per_hyp_count[h] = -1;
            if (per_hyp_residual) per_hyp_residual[h] = 0.0;
            continue;
        }
        double residual;
        int count = orc_count_inliers(p1, p2, M, Fp, max_error_sq, NULL, &residual);
        if (per_hyp_count) per_hyp_count[h] = count;
        if (per_hyp_residual) per_hyp_residual[h] = residual;
        if ((count > count_best) || ((count == count_best) && (residual < residual_best))) { /* :76-84 */
            count_best = count;
            residual_best = residual;
            hyp_best = h;
            memcpy(Fbest, Fp, sizeof(Fbest));
        }
    }
    if (hyp_best >= 0) {
        double r;
        memcpy(F, Fbest, sizeof(Fbest));
        orc_count_inliers(p1, p2, M, F, max_error_sq, mask, &r);
    }
    *best_hyp = hyp_best;
    *best_count = count_best;
    *best_residual = residual_best;
    return count_best > 0; /* :89 */
}

/* ------------------------------------------------------------------------- */
/* vision/sfm-solve.cpp                                                        */
/* ------------------------------------------------------------------------- */

/* sfm-solve.cpp:74-84:  E = U * diag(v, v, 0) * V^T, v = sqrt(s0*s1)
 *   == E_ij = (U_i0*v)*V_j0 + (U_i1*v)*V_j1, with V_jk = vt_kj. */
void orc_project_essential(const double F[9], double E[9])
{
    double w[3], u[9], vt[9];
    orc_svd(F, 3, 3, w, u, vt);
    double v = sqrt(w[0] * w[1]);
    double out[9];
    for (int i = 0; i < 3; ++i) {
        double a = u[i * 3 + 0] * v, b = u[i * 3 + 1] * v;
        for (int j = 0; j < 3; ++j)
            out[i * 3 + j] = a * vt[0 * 3 + j] + b * vt[1 * 3 + j];
    }
    memcpy(E, out, sizeof(out));
}

/* sfm-solve.cpp:97-127.  With W = [0 -1 0; 1 0 0; 0 0 1], Z = [0 1 0; -1 0 0; 0 0 0]:
 *   Ra_ij = (U_i1*V_j0 + (-U_i0)*V_j1) + U_i2*V_j2
 *   Rb_ij = ((-U_i1)*V_j0 + U_i0*V_j1) + U_i2*V_j2
 *   S_ij  = (-U_i1)*U_j0 + U_i0*U_j1 ;  t = (-S_12, S_02, -S_01) */
void orc_decompose_essential(const double E[9], double Ra[9], double Rb[9], double t[3])
{
    double w[3], U[9], vt[9], V[9];
    orc_svd(E, 3, 3, w, U, vt);
    mat3_transpose(vt, V);
    if (mat3_det(U) < 0.0)
        for (int i = 0; i < 9; ++i) U[i] = -U[i];
    if (mat3_det(V) < 0.0)
        for (int i = 0; i < 9; ++i) V[i] = -V[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Ra[i * 3 + j] = (U[i * 3 + 1] * V[j * 3 + 0] + (-U[i * 3 + 0]) * V[j * 3 + 1]) + U[i * 3 + 2] * V[j * 3 + 2];
            Rb[i * 3 + j] = ((-U[i * 3 + 1]) * V[j * 3 + 0] + U[i * 3 + 0] * V[j * 3 + 1]) + U[i * 3 + 2] * V[j * 3 + 2];
        }
#define S_(i, j) ((-U[(i) * 3 + 1]) * U[(j) * 3 + 0] + U[(i) * 3 + 0] * U[(j) * 3 + 1])
    t[0] = -S_(1, 2);
    t[1] = S_(0, 2);
    t[2] = -S_(0, 1);
#undef S_
}

/* sfm-solve.cpp:134-227.  P1 = [I|0]; P2 = [rectify(R)|t] (SO3 ctor, lie-group.hpp:31-36);
 * the cheirality test in camera 2 uses the un-rectified R (:216).  The det(V) sign flip
 * (:196-199) is a no-op after division by X3 and is omitted; |X3| < tol skips the point
 * instead of asserting. */
int orc_triangulate_points(const double R[9], const double t[3], const double *p1, const double *p2, int M,
                           const uint8_t *mask, double *points, int64_t *idx)
{
    double Rr[9];
    orc_so3_from_matrix(R, Rr);
    int n = 0;
    for (int i = 0; i < M; ++i) {
        if (mask && mask[i] == 0)
            continue;
        double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        double A[16], w[4], u[16], vt[16];
        A[0] = -1.0; A[1] = 0.0;  A[2] = x1; A[3] = 0.0;
        A[4] = 0.0;  A[5] = -1.0; A[6] = y1; A[7] = 0.0;
        A[8] = x2 * Rr[6] - Rr[0];  A[9] = x2 * Rr[7] - Rr[1];  A[10] = x2 * Rr[8] - Rr[2];  A[11] = x2 * t[2] - t[0];
        A[12] = y2 * Rr[6] - Rr[3]; A[13] = y2 * Rr[7] - Rr[4]; A[14] = y2 * Rr[8] - Rr[5]; A[15] = y2 * t[2] - t[1];
        orc_svd(A, 4, 4, w, u, vt);
        const double *X = vt + 12; /* V.col(3) */
        if (fabs(X[3]) < ORC_TOLERANCE)
            continue;
        double scale = 1.0 / X[3];
        double pt[3] = {X[0] * scale, X[1] * scale, X[2] * scale};
        if (pt[2] < ORC_TOLERANCE)
            continue;
        double z2 = ((R[6] * pt[0] + R[7] * pt[1]) + R[8] * pt[2]) + t[2];
        if (z2 < ORC_TOLERANCE)
            continue;
        points[3 * n] = pt[0];
        points[3 * n + 1] = pt[1];
        points[3 * n + 2] = pt[2];
        idx[n] = i;
        ++n;
    }
    return n;
}

/* sfm-solve.cpp:232-284: candidates (Ra,t),(Ra,-t),(Rb,t),(Rb,-t); strictly more points wins */
int orc_recover_pose_and_points(const double E[9], const double *p1, const double *p2, int M,
                                const uint8_t *mask, double R[9], double t[3], double *points, int64_t *idx,
                                int *n_points)
{
    double Rc[2][9], tc[2][3];
    orc_decompose_essential(E, Rc[0], Rc[1], tc[0]);
    tc[1][0] = -tc[0][0];
    tc[1][1] = -tc[0][1];
    tc[1][2] = -tc[0][2];
    double *cp = (double *)malloc(sizeof(double) * 3 * (size_t)(M > 0 ? M : 1));
    int64_t *ci = (int64_t *)malloc(sizeof(int64_t) * (size_t)(M > 0 ? M : 1));
    int best = 0, success = 0;
    for (int r = 0; r < 2; ++r)
        for (int s = 0; s < 2; ++s) {
            int n = orc_triangulate_points(Rc[r], tc[s], p1, p2, M, mask, cp, ci);
            if (n > best) {
                success = 1;
                best = n;
                memcpy(points, cp, sizeof(double) * 3 * (size_t)n);
                memcpy(idx, ci, sizeof(int64_t) * (size_t)n);
                memcpy(R, Rc[r], sizeof(double) * 9);
                memcpy(t, tc[s], sizeof(double) * 3);
            }
        }
    free(cp);
    free(ci);
    *n_points = best;
    return success;
}

/* sfm-solve.cpp:285-368 (find_essential_matrix #else branch :64-90 inlined) */
int orc_sfm_solve(const double *uv1, const double *uv2, int M, const double K[9], const orc_params *prm,
                  orc_two_view_result *res, uint8_t *mask, double *points, int64_t *idx)
{
    memset(res, 0, sizeof(*res));
    res->n_matches = M;
    res->best_hyp = -1;
    if (M < 1)
        return 0;
    double Kinv[9];
    orc_mat3_inverse(K, Kinv);
    double *p1 = (double *)malloc(sizeof(double) * 2 * (size_t)M);
    double *p2 = (double *)malloc(sizeof(double) * 2 * (size_t)M);
    orc_normalize_points(Kinv, uv1, M, p1);
    orc_normalize_points(Kinv, uv2, M, p2);
    double max_error_sq = prm->max_error_sq > 0 ? prm->max_error_sq : 5e-2 / K[0] / K[4]; /* :311 */
    memset(mask, 0, (size_t)M);

    int ok = orc_ransac_fundamental(p1, p2, M, max_error_sq, prm->num_hypotheses, prm->sampler, prm->seed,
                                    res->F, mask, &res->best_hyp, &res->best_count, &res->best_residual, NULL,
                                    NULL);
    int ret = 0;
    if (res->best_hyp >= 0)
        orc_project_essential(res->F, res->E); /* :74-84, done even when compute() said false */
    if (ok) {
        int inliers = 0;
        for (int i = 0; i < M; ++i)
            inliers += mask[i] > 0 ? 1 : 0;
        res->n_inliers = inliers;
        if (inliers >= prm->min_inliers) { /* :330 */
            int n_points = 0;
            if (orc_recover_pose_and_points(res->E, p1, p2, M, mask, res->R1to2, res->t1to2, points, idx,
                                            &n_points)) {
                double Rr[9];
                orc_so3_from_matrix(res->R1to2, Rr);
                orc_se3_inverse(Rr, res->t1to2, res->R, res->t); /* :364 */
                res->n_points = n_points;
                res->valid = 1;
                ret = 1;
            }
        }
    }
    free(p1);
    free(p2);
    return ret;
}

/* sfm-solve.cpp:370-394.  pose1 / pose2 are SE3 objects (already rectified). */
int orc_sfm_triangulate(const double *uv1, const double *uv2, int M, const double K[9], const double R1[9],
                        const double t1[3], const double R2[9], const double t2[3], double *points, int64_t *idx)
{
    double Ri[9], ti[3], R12[9], t12[3], Kinv[9];
    orc_se3_inverse(R2, t2, Ri, ti);
    orc_se3_compose(Ri, ti, R1, t1, R12, t12);
    orc_mat3_inverse(K, Kinv);
    double *p1 = (double *)malloc(sizeof(double) * 2 * (size_t)(M > 0 ? M : 1));
    double *p2 = (double *)malloc(sizeof(double) * 2 * (size_t)(M > 0 ? M : 1));
    orc_normalize_points(Kinv, uv1, M, p1);
    orc_normalize_points(Kinv, uv2, M, p2);
    int n = orc_triangulate_points(R12, t12, p1, p2, M, NULL, points, idx);
    free(p1);
    free(p2);
    return n;
}

/* front-end/image-pair.cpp:30-71,116-174 (without refine()).  matches: base = train, pair = query. */
int orc_image_pair(const uint8_t *base_desc, const float *base_kp, int n_base, const uint8_t *pair_desc,
                   const float *pair_kp, int n_pair, int desc_bytes, double ratio, double max_dist,
                   const double K[9], const orc_params *prm, orc_match *matches, orc_two_view_result *res,
                   uint8_t *mask, double *points, int64_t *idx)
{
    memset(res, 0, sizeof(*res));
    res->best_hyp = -1;
    int M = orc_match_visual_features(base_desc, n_base, pair_desc, n_pair, desc_bytes, ratio, max_dist, matches);
    if (M < 0)
        return 0;
    double *uv1 = (double *)malloc(sizeof(double) * 2 * (size_t)(M > 0 ? M : 1));
    double *uv2 = (double *)malloc(sizeof(double) * 2 * (size_t)(M > 0 ? M : 1));
    for (int m = 0; m < M; ++m) { /* image-pair.cpp:123-140; visual-feature.cpp:179-190 float -> double */
        uv1[2 * m] = (double)base_kp[2 * matches[m].trainIdx];
        uv1[2 * m + 1] = (double)base_kp[2 * matches[m].trainIdx + 1];
        uv2[2 * m] = (double)pair_kp[2 * matches[m].queryIdx];
        uv2[2 * m + 1] = (double)pair_kp[2 * matches[m].queryIdx + 1];
    }
    int ok = orc_sfm_solve(uv1, uv2, M, K, prm, res, mask, points, idx);
    res->n_matches = M;
    free(uv1);
    free(uv2);
    return ok;
}

/* ------------------------------------------------------------------------- */
/* vision/pnp-solve.cpp (row f1): P3P-RANSAC, the build's own (see mvs_oracle.h) */
/* ------------------------------------------------------------------------- */
void orc_sample4(uint64_t seed, uint32_t hyp, int n, int sampler, int idx[4])
{
    if (sampler == ORC_SAMPLER_IDENTITY) {
        for (int k = 0; k < 4; ++k)
            idx[k] = k;
        return;
    }
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t w[4];
    uint32_t c0[4] = {hyp, 2, 0, 0}; /* block 2: blocks 0 and 1 belong to the 8-of-M sampler */
    orc_philox4x32_10(c0, key, w);
    int sorted[4];
    for (int k = 0; k < 4; ++k) {
        uint32_t r = (uint32_t)(((uint64_t)w[k] * (uint32_t)(n - k)) >> 32);
        int pos = 0;
        for (int t = 0; t < k; ++t)
            if (r >= (uint32_t)sorted[t]) {
                ++r;
                pos = t + 1;
            }
        for (int t = k; t > pos; --t)
            sorted[t] = sorted[t - 1];
        sorted[pos] = (int)r;
        idx[k] = (int)r;
    }
}

/* real roots of x^4 + b x^3 + c x^2 + d x + e (Ferrari; the resolvent cubic's positive root by 64 bisections).
 * Root order: quadratic factor with sg = +1 (larger, smaller), then sg = -1 (larger, smaller). */
static int quartic_real_roots(double b, double c, double d, double e, double roots[4])
{
    const double p = c - 3.0 * b * b / 8.0;
    const double q = d - b * c / 2.0 + b * b * b / 8.0;
    const double r = e - b * d / 4.0 + b * b * c / 16.0 - 3.0 * b * b * b * b / 256.0;
    const double sh = b / 4.0;
    int n = 0;
    if (q == 0.0) { /* biquadratic */
        const double disc = p * p - 4.0 * r;
        if (disc >= 0.0) {
            const double sd = sqrt(disc);
            const double y2a = (-p + sd) / 2.0, y2b = (-p - sd) / 2.0;
            if (y2a >= 0.0) {
                const double y = sqrt(y2a);
                roots[n++] = y - sh;
                roots[n++] = -y - sh;
            }
            if (y2b >= 0.0) {
                const double y = sqrt(y2b);
                roots[n++] = y - sh;
                roots[n++] = -y - sh;
            }
        }
        return n;
    }
    /* 8 m^3 + 8 p m^2 + (2 p^2 - 8 r) m - q^2 = 0 has a root in (0, hi): f(0) = -q^2 < 0 */
    const double c1 = 2.0 * p * p - 8.0 * r, c0 = q * q;
    double hi = fabs(p);
    const double h1 = fabs(c1 / 8.0), h2 = fabs(c0 / 8.0);
    if (h1 > hi) hi = h1;
    if (h2 > hi) hi = h2;
    hi = hi + 1.0;
    double lo = 0.0;
    for (int it = 0; it < 64; ++it) {
        const double mid = 0.5 * (lo + hi);
        const double fm = ((8.0 * mid + 8.0 * p) * mid + c1) * mid - c0;
        if (fm > 0.0)
            hi = mid;
        else
            lo = mid;
    }
    const double m = 0.5 * (lo + hi);
    const double s = sqrt(2.0 * m);
    const double t = q / (2.0 * s);
    for (int k = 0; k < 2; ++k) {
        const double sg = k == 0 ? 1.0 : -1.0;
        const double cc = p / 2.0 + m + sg * t; /* y^2 - sg s y + cc = 0 */
        const double disc = s * s - 4.0 * cc;
        if (disc >= 0.0) {
            const double sd = sqrt(disc);
            roots[n++] = (sg * s + sd) / 2.0 - sh;
            roots[n++] = (sg * s - sd) / 2.0 - sh;
        }
    }
    return n;
}

static void tri_frame(const double P[9], double Fm[9])
{ /* orthonormal frame of a triangle, columns e1, e2, e3 */
    double e1[3] = {P[3] - P[0], P[4] - P[1], P[5] - P[2]};
    const double n1 = sqrt((e1[0] * e1[0] + e1[1] * e1[1]) + e1[2] * e1[2]);
    e1[0] = e1[0] / n1; e1[1] = e1[1] / n1; e1[2] = e1[2] / n1;
    const double d[3] = {P[6] - P[0], P[7] - P[1], P[8] - P[2]};
    double e3[3] = {e1[1] * d[2] - e1[2] * d[1], e1[2] * d[0] - e1[0] * d[2], e1[0] * d[1] - e1[1] * d[0]};
    const double n3 = sqrt((e3[0] * e3[0] + e3[1] * e3[1]) + e3[2] * e3[2]);
    e3[0] = e3[0] / n3; e3[1] = e3[1] / n3; e3[2] = e3[2] / n3;
    const double e2[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
    for (int k = 0; k < 3; ++k) {
        Fm[k * 3 + 0] = e1[k];
        Fm[k * 3 + 1] = e2[k];
        Fm[k * 3 + 2] = e3[k];
    }
}

/* Grunert's P3P (coefficients as in Haralick et al., "Review and analysis of solutions of the three point
 * perspective pose estimation problem", 1994): distances s1, s2 = u s1, s3 = v s1, quartic in v. */
int orc_p3p(const double f[9], const double X[9], double R[4][9], double t[4][3])
{
    const double d12[3] = {X[0] - X[3], X[1] - X[4], X[2] - X[5]};
    const double d13[3] = {X[0] - X[6], X[1] - X[7], X[2] - X[8]};
    const double d23[3] = {X[3] - X[6], X[4] - X[7], X[5] - X[8]};
    const double a2 = dot3(d23, d23), b2 = dot3(d13, d13), c2 = dot3(d12, d12);
    const double ca = dot3(f + 3, f + 6), cb = dot3(f, f + 6), cg = dot3(f, f + 3);
    const double k1 = (a2 - c2) / b2, k2 = (a2 + c2) / b2, k3 = (b2 - c2) / b2, k4 = (b2 - a2) / b2;
    const double A4 = (k1 - 1.0) * (k1 - 1.0) - 4.0 * c2 / b2 * ca * ca;
    const double A3 = 4.0 * (k1 * (1.0 - k1) * cb - (1.0 - k2) * ca * cg + 2.0 * c2 / b2 * ca * ca * cb);
    const double A2 = 2.0 * (k1 * k1 - 1.0 + 2.0 * k1 * k1 * cb * cb + 2.0 * k3 * ca * ca - 4.0 * k2 * ca * cb * cg +
                             2.0 * k4 * cg * cg);
    const double A1 = 4.0 * (-k1 * (1.0 + k1) * cb + 2.0 * a2 / b2 * cg * cg * cb - (1.0 - k2) * ca * cg);
    const double A0 = (1.0 + k1) * (1.0 + k1) - 4.0 * a2 / b2 * cg * cg;
    double roots[4];
    const int nr = quartic_real_roots(A3 / A4, A2 / A4, A1 / A4, A0 / A4, roots);
    double Fw[9];
    tri_frame(X, Fw);
    int ns = 0;
    for (int k = 0; k < nr; ++k) {
        const double v = roots[k];
        if (!(v > 0.0))
            continue;
        const double u = ((-1.0 + k1) * v * v - 2.0 * k1 * cb * v + 1.0 + k1) / (2.0 * (cg - v * ca));
        if (!(u > 0.0))
            continue;
        const double s1 = sqrt(c2 / (1.0 + u * u - 2.0 * u * cg));
        const double s2 = u * s1, s3 = v * s1;
        const double Pc[9] = {s1 * f[0], s1 * f[1], s1 * f[2], s2 * f[3], s2 * f[4], s2 * f[5],
                              s3 * f[6], s3 * f[7], s3 * f[8]};
        double Fc[9];
        tri_frame(Pc, Fc);
        for (int i = 0; i < 3; ++i) /* R = Fc * Fw^T */
            for (int j = 0; j < 3; ++j)
                R[ns][i * 3 + j] = (Fc[i * 3 + 0] * Fw[j * 3 + 0] + Fc[i * 3 + 1] * Fw[j * 3 + 1]) + Fc[i * 3 + 2] * Fw[j * 3 + 2];
        for (int i = 0; i < 3; ++i)
            t[ns][i] = Pc[i] - dot3(R[ns] + 3 * i, X);
        ++ns;
    }
    return ns;
}

/* division-free reprojection test: lhs = fx^2 dx^2 + fy^2 dy^2 (squared pixel error times zc^2), rhs = err^2 zc^2 */
static int pnp_inlier(const double R[9], const double t[3], const double X[3], double xi, double yi, double fx2,
                      double fy2, double thr2, double *lhs_out, double *rhs_out)
{
    const double xc = fma(R[0], X[0], fma(R[1], X[1], fma(R[2], X[2], t[0])));
    const double yc = fma(R[3], X[0], fma(R[4], X[1], fma(R[5], X[2], t[1])));
    const double zc = fma(R[6], X[0], fma(R[7], X[1], fma(R[8], X[2], t[2])));
    const double dx = fma(-xi, zc, xc), dy = fma(-yi, zc, yc);
    const double lhs = fma(fy2, dy * dy, fx2 * (dx * dx));
    const double rhs = thr2 * (zc * zc);
    if (lhs_out) *lhs_out = lhs;
    if (rhs_out) *rhs_out = rhs;
    return (zc > 0.0) && (lhs <= rhs);
}

int orc_pnp_solve(const double *world_xyz, const double *image_uv, int n, const double K[9],
                  const orc_pnp_params *prm, double R[9], double t[3], int64_t *inlier_idx, int *n_inliers,
                  double Rw2c[9], double tw2c[3], int *best_hyp_out)
{
    *n_inliers = 0;
    if (best_hyp_out) *best_hyp_out = -1;
    if (n < 7) /* pnp-solve.cpp:13,22 PNP_MIN_POINT_COUNT (assert) */
        return 0;
    double Kinv[9];
    orc_mat3_inverse(K, Kinv);
    double *xy = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    double *fb = (double *)malloc(sizeof(double) * 3 * (size_t)n);
    orc_normalize_points(Kinv, image_uv, n, xy);
    for (int i = 0; i < n; ++i) { /* unit bearing of (x, y, 1) */
        const double x = xy[2 * i], y = xy[2 * i + 1];
        const double nn = sqrt((x * x + y * y) + 1.0);
        fb[3 * i] = x / nn;
        fb[3 * i + 1] = y / nn;
        fb[3 * i + 2] = 1.0 / nn;
    }
    const double fx2 = K[0] * K[0], fy2 = K[4] * K[4];
    const double thr2 = prm->reproj_error * prm->reproj_error;
    int best_cnt = -1, best_hyp = -1;
    double bR[9] = {0}, bt[3] = {0};
    for (int h = 0; h < prm->num_hypotheses; ++h) {
        int idx[4];
        orc_sample4(prm->seed, (uint32_t)h, n, prm->sampler, idx);
        double f3[9], X3[9];
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < 3; ++c) {
                f3[3 * k + c] = fb[3 * idx[k] + c];
                X3[3 * k + c] = world_xyz[3 * idx[k] + c];
            }
        double Rs[4][9], ts[4][3];
        const int ns = orc_p3p(f3, X3, Rs, ts);
        /* disambiguate with the 4th point: smallest squared pixel error, compared without dividing:
         * lhs_k / rhs_k < lhs_sel / rhs_sel  <=>  lhs_k * rhs_sel < lhs_sel * rhs_k  (rhs = err^2 zc^2 > 0) */
        int sel = -1;
        double sel_lhs = 0.0, sel_rhs = 1.0;
        for (int k = 0; k < ns; ++k) {
            double lhs, rhs;
            pnp_inlier(Rs[k], ts[k], world_xyz + 3 * idx[3], xy[2 * idx[3]], xy[2 * idx[3] + 1], fx2, fy2, thr2, &lhs, &rhs);
            if (!(rhs > 0.0))
                continue;
            if (sel < 0 || lhs * sel_rhs < sel_lhs * rhs) {
                sel = k;
                sel_lhs = lhs;
                sel_rhs = rhs;
            }
        }
        if (sel < 0)
            continue;
        int cnt = 0;
        for (int i = 0; i < n; ++i)
            cnt += pnp_inlier(Rs[sel], ts[sel], world_xyz + 3 * i, xy[2 * i], xy[2 * i + 1], fx2, fy2, thr2, NULL, NULL);
        if (cnt > best_cnt) { /* first hypothesis with the most inliers wins */
            best_cnt = cnt;
            best_hyp = h;
            memcpy(bR, Rs[sel], sizeof(bR));
            memcpy(bt, ts[sel], sizeof(bt));
        }
    }
    int ok = 0;
    if (best_hyp >= 0 && best_cnt >= prm->min_inliers) {
        int m = 0;
        for (int i = 0; i < n; ++i)
            if (pnp_inlier(bR, bt, world_xyz + 3 * i, xy[2 * i], xy[2 * i + 1], fx2, fy2, thr2, NULL, NULL))
                inlier_idx[m++] = i;
        *n_inliers = m;
        if (Rw2c) memcpy(Rw2c, bR, sizeof(bR));
        if (tw2c) memcpy(tw2c, bt, sizeof(bt));
        double Rr[9];
        orc_so3_from_matrix(bR, Rr);   /* SE3(SO3(R), t) */
        orc_se3_inverse(Rr, bt, R, t); /* .inverse(): camera in world (pnp-solve.cpp:101) */
        ok = 1;
    }
    if (best_hyp_out) *best_hyp_out = best_hyp;
    free(xy);
    free(fb);
    return ok;
}


/* ---- scale propagation and trajectory of a sequence (row f2; front-end/visual-odometer.cpp:422-445,577-588) ----
 * pair k: pose of frame k+1 in frame k, unit baseline; track q: pose of frame q+2 in frame q in pair q's scale.
 *   track_scale[q] = |(pair_q^-1 o track_q).t|, sigma_0 = 1, sigma_{q+1} = sigma_q track_scale[q]
 *   G_0 = I, G_1 = pair_0, G_{q+2} = G_q o (R_track_q, sigma_q t_track_q)
 * a failed track keeps the scale and uses pair q+1: G_{q+2} = G_{q+1} o (R, sigma t); an invalid pair is the identity. */
static void chain_compose(const double *A, const double *R, const double *t, double sig, double *out)
{
    const double s0 = sig * t[0], s1 = sig * t[1], s2 = sig * t[2];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            out[3 * i + j] = (A[3 * i] * R[j] + A[3 * i + 1] * R[3 + j]) + A[3 * i + 2] * R[6 + j];
        out[9 + i] = ((A[3 * i] * s0 + A[3 * i + 1] * s1) + A[3 * i + 2] * s2) + A[9 + i];
    }
}

void orc_seq_chain(int n_frames, const double *pair_R, const double *pair_t, const int32_t *pair_valid,
                   const double *track_R, const double *track_t, const int32_t *track_ok, double *traj_R, double *traj_t,
                   double *pair_scale, double *track_scale)
{
    static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Z3[3] = {0, 0, 0};
    double Ga[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, Gb[12], Gn[12];
    double sigma = 1.0;
    memcpy(traj_R, Ga, 9 * sizeof(double));
    memcpy(traj_t, Ga + 9, 3 * sizeof(double));
    pair_scale[0] = 1.0;
    chain_compose(Ga, pair_valid[0] ? pair_R : I3, pair_valid[0] ? pair_t : Z3, 1.0, Gb);
    memcpy(traj_R + 9, Gb, 9 * sizeof(double));
    memcpy(traj_t + 3, Gb + 9, 3 * sizeof(double));
    for (int q = 0; q + 2 < n_frames; ++q) {
        const double *Rp = pair_R + 9 * q, *tp = pair_t + 3 * q, *Rt = track_R + 9 * q, *tt = track_t + 3 * q;
        double scale = 1.0;
        if (track_ok[q] && pair_valid[q]) {
            const double d0 = tt[0] - tp[0], d1 = tt[1] - tp[1], d2 = tt[2] - tp[2];
            const double r0 = (Rp[0] * d0 + Rp[3] * d1) + Rp[6] * d2;
            const double r1 = (Rp[1] * d0 + Rp[4] * d1) + Rp[7] * d2;
            const double r2 = (Rp[2] * d0 + Rp[5] * d1) + Rp[8] * d2;
            scale = sqrt((r0 * r0 + r1 * r1) + r2 * r2);
            chain_compose(Ga, Rt, tt, sigma, Gn);
        } else {
            const int v = pair_valid[q + 1];
            chain_compose(Gb, v ? pair_R + 9 * (q + 1) : I3, v ? pair_t + 3 * (q + 1) : Z3, sigma, Gn);
        }
        track_scale[q] = scale;
        sigma = sigma * scale;
        pair_scale[q + 1] = sigma;
        memcpy(traj_R + 9 * (q + 2), Gn, 9 * sizeof(double));
        memcpy(traj_t + 3 * (q + 2), Gn + 9, 3 * sizeof(double));
        memcpy(Ga, Gb, sizeof(Ga));
        memcpy(Gb, Gn, sizeof(Gb));
    }
}
