// prescreen.hpp -- sound pre-screen of 8-point RANSAC hypotheses.
//
// The exact solve of a hypothesis (device_math.hpp eight_point: A^T A, one-sided Jacobi SVD of the 9x9, rank-2, de-
// normalisation) costs ~44 k instructions per lane; 97 % of the hypotheses of a contaminated match set cannot win.  Here a
// hypothesis gets
//   * an APPROXIMATE fundamental matrix F~ from a Householder QR of A^T (9x8, no pivoting, ~1 k instructions) followed
//     by the exact path's own rank-2 + de-normalisation code, and
//   * a rigorous bound `band` with   | fl(r_i(F_J)) - fl(r_i(F~)) | <= band   for EVERY match i of the pair, where F_J is
//     what the exact path would have produced for this sample (bit for bit) and r_i the epipolar residual of match i.
// So  #{i : r~_i < thr + band}  >=  count_J  >=  #{i : r~_i < thr - band}: upper and lower bounds of the exact inlier count
// without running the exact solve.  A hypothesis whose upper bound is below the lower bound of some other hypothesis of
// the pair cannot be the reference's winner (estimator-RANSAC.cpp:76-84: most inliers first) and is never solved exactly;
// everything that survives is solved exactly and scored exactly, so the selected hypothesis, its count, residual sum, F
// and mask are the exact path's, bit for bit.  A hypothesis for which no useful bound can be certified (ill-conditioned
// sample, small singular-value gap, band too wide for the threshold) is flagged and goes to the exact solve directly.
//
// The derivation of every constant below is in DESIGN.md 4.3e; tests/prescreen_model.py restates this file in numpy and
// tests/test_prescreen.py checks band against the oracle hypothesis by hypothesis (CPU: model, GPU: this code).
#pragma once
#include "device_math.hpp"

namespace mvs {

// bounding box of ALL matches of a pair in ideal-camera coordinates (pair_prepare_kernel)
struct PairBox {
    double x1lo, x1hi, y1lo, y1hi, x2lo, x2hi, y2lo, y2hi;
};

constexpr double kPsU = 0x1p-53;          // unit roundoff
constexpr double kPsTauC = 2.0e-12;       // >= 2.001 (8000 u + 8.01 u): <= 1080 Jacobi rotations + forming A^T A
constexpr double kPsEtaQ = 4.0e-12;       // loss of orthogonality of the accumulated V^T over <= 1080 rotations
constexpr double kPsSvd3 = 2.0e-11;       // backward error of the 3x3 Jacobi SVD + recomposition, both paths together
constexpr double kPsBandFrac = 0.125;     // screened only if band <= kPsBandFrac * thr
constexpr int kPsInvalid = 0, kPsApprox = 1, kPsNeedExact = 2, kPsExact = 3;   // per-hypothesis state byte (hyp_okf)

// Householder QR of A^T, in place.  c[j][0..8] = row j of A = column j of A^T.  After step k the strict upper triangle of R
// sits in c[j][k] (k < j), R_kk in rd[k], the reflector's vector in c[k][k..8] and 1 / (v.v / 2) in beta[k].
MVS_DEV void householder_qr_9x8(double (&c)[8][9], double (&rd)[8], double (&beta)[8])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double ss = 0.0;
#pragma unroll
        for (int i = k; i < 9; ++i)
            ss = dfma(c[k][i], c[k][i], ss);
        const double nrm = dsqrt(ss);
        const double x0 = c[k][k];
        const double ax0 = dabs(x0);
        const double alpha = x0 >= 0.0 ? -nrm : nrm;
        c[k][k] = x0 - alpha;                  // v0: same sign as x0, no cancellation
        const double vv = nrm * (nrm + ax0);   // = v.v / 2
        const double bk = vv > 0.0 ? 1.0 / vv : 0.0;
        rd[k] = alpha;
        beta[k] = bk;
#pragma unroll
        for (int j = k + 1; j < 8; ++j) {
            double d = 0.0;
#pragma unroll
            for (int i = k; i < 9; ++i)
                d = dfma(c[k][i], c[j][i], d);
            const double t = bk * d;
#pragma unroll
            for (int i = k; i < 9; ++i)
                c[j][i] = dfma(-t, c[k][i], c[j][i]);
        }
    }
}

// flag (kPs*), F~ and band for one sample.  nm / a1..b2: the exact path's own normalisation of the sample (same bits).
// w_out: singular values of reshape(n~) as the 3x3 Jacobi computed them (diagnostics).
template <int VAR>
MVS_DEV int prescreen_hypothesis(const double (&x1)[8], const double (&y1)[8], const double (&x2)[8], const double (&y2)[8],
                                 const PairBox &bx, double thr, double (&F)[9], double &band_out, bool &bad3)
{
    double a1[8], b1[8], a2[8], b2[8];
    EightNorm nm;
    bool ok = normalise8(x1, y1, a1, b1, nm.s1, nm.m1x, nm.m1y);
    ok = normalise8(x2, y2, a2, b2, nm.s2, nm.m2x, nm.m2y) && ok;
    band_out = 0.0;
    if (!ok) {   // reference: assert(scale > epsilon); the exact path rejects the sample too (same bits, same decision)
#pragma unroll
        for (int k = 0; k < 9; ++k)
            F[k] = 0.0;
        return kPsInvalid;
    }
    // design matrix rows (fundamental-matrix.cpp:78-87), the exact path's products
    double c[8][9];
    double S = 0.0;   // ||A||_F^2
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c[i][0] = a2[i] * a1[i]; c[i][1] = a2[i] * b1[i]; c[i][2] = a2[i];
        c[i][3] = b2[i] * a1[i]; c[i][4] = b2[i] * b1[i]; c[i][5] = b2[i];
        c[i][6] = a1[i];         c[i][7] = b1[i];         c[i][8] = 1.0;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            S = dfma(c[i][k], c[i][k], S);
    }
    S *= 1.0 + 1e-12;
    const double sqrtS = dsqrt(S) * (1.0 + 1e-12);
    double rd[8], beta[8];
    householder_qr_9x8(c, rd, beta);
    // n~ = H_0 H_1 ... H_7 e_8
    double n[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
        n[i] = i == 8 ? 1.0 : 0.0;
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        double d = 0.0;
#pragma unroll
        for (int i = k; i < 9; ++i)
            d = dfma(c[k][i], n[i], d);
        const double t = beta[k] * d;
#pragma unroll
        for (int i = k; i < 9; ++i)
            n[i] = dfma(-t, c[k][i], n[i]);
    }
    // ||R^-1||_F by explicit back substitution, column by column (R_ik = c[k][i] for i < k)
    double inv[8];
    bool piv_ok = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        piv_ok = piv_ok && (dabs(rd[i]) > 0x1p-500);
        inv[i] = 1.0 / rd[i];
    }
    double y2sum = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double y[8];
        y[j] = inv[j];
        y2sum = dfma(y[j], y[j], y2sum);
#pragma unroll
        for (int i = j - 1; i >= 0; --i) {
            double s = 0.0;
#pragma unroll
            for (int k = i + 1; k <= j; ++k)
                s = dfma(c[k][i], y[k], s);
            y[i] = -(s * inv[i]);
            y2sum = dfma(y[i], y[i], y2sum);
        }
    }
    const double yf = dsqrt(y2sum) * (1.0 + 1e-12);
    // a-posteriori residual of n~ against the ORIGINAL rows of A (recomputed: the QR overwrote them)
    double rho2 = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        double r = n[8];
        r = dfma(a1[i], n[6], r);
        r = dfma(b1[i], n[7], r);
        r = dfma(a2[i], n[2], r);
        r = dfma(b2[i], n[5], r);
        r = dfma(a2[i] * a1[i], n[0], r);
        r = dfma(a2[i] * b1[i], n[1], r);
        r = dfma(b2[i] * a1[i], n[3], r);
        r = dfma(b2[i] * b1[i], n[4], r);
        rho2 = dfma(r, r, rho2);
    }
    const double rho = dsqrt(rho2) * (1.0 + 1e-12);
    // sigma_8(A) >= (1 - z) / ||R^-1||_F (1 - 250 u) - 176 u ||A||_F
    const double z = 12.0 * kPsU * sqrtS * yf;
    const double sig8 = (1.0 - z) / yf * (1.0 - 1e-13) - 4e-14 * sqrtS;
    const double g = sig8 * sig8;
    const double eta_j = 1.01 * kPsTauC * S / g + kPsEtaQ;
    const double eta_a = 1.5 * (rho + 1.2e-15 * sqrtS) / sig8 + 1e-13;
    const double eta = (eta_j + eta_a + kPsSvd3) * (1.0 + 1e-12);
    // rank-2 + de-normalisation: the exact path's code on the approximate null vector
    double w[3];
    eight_point_back<VAR>(n, nm, F, bad3, w);
    const double delta = (w[1] - w[2]) - eta - kPsSvd3;
    const double dfn = (2.0 + 2.0 * (w[2] + 3.0 * eta) / delta) * eta + kPsSvd3;
    // N = max over the pair's points of || T p || (T = the sample's Hartley transform), N' with absolute values
    const double d1x = fmax(dabs(nm.m1x - bx.x1lo), dabs(nm.m1x - bx.x1hi));
    const double d1y = fmax(dabs(nm.m1y - bx.y1lo), dabs(nm.m1y - bx.y1hi));
    const double d2x = fmax(dabs(nm.m2x - bx.x2lo), dabs(nm.m2x - bx.x2hi));
    const double d2y = fmax(dabs(nm.m2y - bx.y2lo), dabs(nm.m2y - bx.y2hi));
    const double e1x = fmax(dabs(bx.x1lo), dabs(bx.x1hi)) + dabs(nm.m1x);
    const double e1y = fmax(dabs(bx.y1lo), dabs(bx.y1hi)) + dabs(nm.m1y);
    const double e2x = fmax(dabs(bx.x2lo), dabs(bx.x2hi)) + dabs(nm.m2x);
    const double e2y = fmax(dabs(bx.y2lo), dabs(bx.y2hi)) + dabs(nm.m2y);
    const double s1q = nm.s1 * nm.s1, s2q = nm.s2 * nm.s2;
    // N1^2 N2^2 and N1'^2 N2'^2 first, one square root each
    const double n12 = dfma(s1q, dfma(d1x, d1x, d1y * d1y), 1.0) * dfma(s2q, dfma(d2x, d2x, d2y * d2y), 1.0);
    const double n12p = dfma(s1q, dfma(e1x, e1x, e1y * e1y), 1.0) * dfma(s2q, dfma(e2x, e2x, e2y * e2y), 1.0);
    const double band = (dfn * dsqrt(n12) * (1.0 + 1e-9) + 64.0 * kPsU * dsqrt(n12p)) * (1.0 + 1e-9) + 1e-15 * thr;
    band_out = band;
    // every comparison is written so that a NaN anywhere lands in "needs the exact solve"
    const bool certified = piv_ok && (z < 0.5) && (sig8 > 0.0) && (delta > 0.0) && (eta < 1e-3) &&
                           (band <= kPsBandFrac * thr);
    return certified ? kPsApprox : kPsNeedExact;
}

}  // namespace mvs
