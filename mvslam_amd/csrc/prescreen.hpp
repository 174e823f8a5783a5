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
constexpr double kPsSvd3 = 2.0e-11;       // backward error of the exact path's 3x3 Jacobi SVD + recomposition (generous)
constexpr double kPsTrip = 1.0e-12;       // roundings of the verified singular triplet below (~150 operations on |x| <= 1.01)
constexpr double kPsBandFrac = 4.0;       // a hypothesis is certified only if band <= kPsBandFrac * thr.  A certificate can only
                                          // help (a certified hypothesis is counted and then dropped or solved exactly, an
                                          // uncertified one is solved exactly in any case), and the UPPER count bound prunes
                                          // whenever the points within thr + band of the hypothesis are fewer than the pair's
                                          // bound -- also with a band beyond the threshold, where the lower threshold thr - band
                                          // is clamped to 0 and the lower count bound is 0
constexpr double kPsProbeFrac = 0.125;    // ... but a PAIR is pre-screened only if a third of its probe gets bands within this
                                          // fraction of the threshold: with bands near the threshold nothing is pruned, every
                                          // hypothesis ends in the exact solve anyway and the pre-screen is pure overhead
constexpr double kPsProbeMfmaFrac = 1.25; // ... and counted on the matrix cores (mode 1) only if the band INCLUDING the split-bf16
                                          // term 2^-14 T = 64 e32 stays within this multiple of the threshold.  Beyond the threshold
                                          // the matrix cores' lower threshold tl' = thr - band - e32 - 2^-14 T is gone and their
                                          // upper counts take in every match within twice the threshold: only the pilot's bound
                                          // (vector kernel, no 2^-14 T) prunes, less and less -- such pairs take double-precision
                                          // counting (mode 2).  Measured on 64-pair batches, outlier share 0.3 .. 0.9, 0.5 / 2 px noise
                                          // (profiles/r04_sensitivity_guard.json; 2^-14 T is 3 .. 6e-4 there): thr 5e-4 mode 1 1.7-2.8
                                          // against mode 2 2.7-3.0 ms; 2e-4: 3.6-8.3 against 2.9-3.1; 1e-4: 5.8-9.5 against 3.0-3.2 and
                                          // 7.4 for solving everything exactly (round 3's probe chose mode 1 there)
constexpr int kPsInvalid = 0, kPsApprox = 1, kPsNeedExact = 2, kPsExact = 3;   // per-hypothesis state byte (hyp_okf)

// 1 / x for 2^-190 <= |x| <= 2^190 (div_fast's guarded range), correctly rounded like the compiler's division
MVS_DEV double recip_guarded(double x) { return div_fast(1.0, x); }

// Square root and reciprocal for BOUND arithmetic (round 4): results within 2^-49 of the true value instead of correctly
// rounded -- 7 and 5 instructions where the IEEE sequences of hipcc take ~20 and ~13.  They are used only where the
// derivation multiplies by an explicit slack factor (1 +- 1e-14 or more) or adds an absolute term in the safe direction, and
// never on the path that has to reproduce the exact solve's bits (the Hartley scales) or that feeds F~ itself.
MVS_DEV double sqrt_bound(double x)   // x finite and >= 2^-700 (callers' arguments are >= 1e-13 by construction)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    const double h = y * 0.5;
    const double r = dfma(-h, g, 0.5);
    g = dfma(g, r, g);
    const double d = dfma(-g, g, x);
    return dfma(d, h, g);   // second-order correction with the unrefined h: error ~2^-52 + 2^-26 * 2^-50
}
MVS_DEV double sqrt_bound0(double x)  // the same, 0 for x below 2^-700 (a NaN stays a NaN)
{
    const double g = sqrt_bound(x);
    return x < 0x1p-700 ? 0.0 : g;
}
MVS_DEV double rcp_bound(double x)    // 1 / x for normal x, two Newton steps on v_rcp_f64
{
    double r = __builtin_amdgcn_rcp(x);
    double e = dfma(-x, r, 1.0);
    r = dfma(r, e, r);
    e = dfma(-x, r, 1.0);
    return dfma(r, e, r);
}

// Hartley normalisation of a sample from its gathered points: means and scales only (the normalised coordinates are
// rebuilt point by point where they are needed).  The same operations as normalise8: the same bits.
MVS_DEV bool sample_norm(const double (&px)[8], const double (&py)[8], double &scale, double &mx, double &my, bool &tiny)
{
    mx = 0.0;
    my = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        mx += px[i];
        my += py[i];
    }
    mx *= 0.125;
    my *= 0.125;
    double sc = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double dx = px[i] - mx, dy = py[i] - my;
        const double q = dx * dx + dy * dy;
        tiny = tiny | ((q != 0.0) & !sqrt_fast_ok(q));   // bitwise: no branches in the normalisation
        sc += sqrt_fast(q);
    }
    sc *= 0.125;
    const bool ok = sc > kEps;
    scale = kSqrt2 / sc;
    return ok;
}

// The rank-2 step of the pre-screen.  G = reshape(n~) (row-major 3x3, ||G||_F = 1).  The exact path takes the 3x3 Jacobi
// SVD and drops the smallest singular value; all the bound needs of G is ONE verified singular triplet:
//   v     an approximate right singular vector of the smallest singular value, from the characteristic polynomial of
//         C = G^T G (Newton from 0: monotone from the left towards the smallest root) and the cross products of the rows of
//         C - lambda I; HOW it is obtained is immaterial, because what is used of it is checked:
//   w = G v, sig = ||w||, u = w / sig, eps2 = || G^T u - sig v ||: (sig, u, v) is an exact singular triplet of a matrix G'
//         with ||G' - G||_F <= eps2 + (roundings), and X = G - w v^T = G (I - v v^T) is within the same distance of T_2(G');
//   s2lb  a lower bound of the second singular value of X from its invariants (||X||_F^2 and ||X^T X||_F^2).
// If eps2 is not below sig (a noise-free sample: sig ~ 1e-16, u is rounding noise) the triplet (0, *, v) of X itself is used
// instead: ||X - G||_F = sig.  Returns e (the distance that enters eta), sigma_e (the triplet's singular value), extra
// (the distance of X from the rank-2 matrix of the perturbed G), s2lb.  DESIGN.md 4.3e (iv).
MVS_DEV void prescreen_rank2(const double (&g)[9], double (&X)[9], double &e_out, double &sige_out, double &extra_out,
                             double &s2lb_out, bool &ok_out)
{
    // C = G^T G
    const double c00 = dfma(g[6], g[6], dfma(g[3], g[3], g[0] * g[0]));
    const double c01 = dfma(g[6], g[7], dfma(g[3], g[4], g[0] * g[1]));
    const double c02 = dfma(g[6], g[8], dfma(g[3], g[5], g[0] * g[2]));
    const double c11 = dfma(g[7], g[7], dfma(g[4], g[4], g[1] * g[1]));
    const double c12 = dfma(g[7], g[8], dfma(g[4], g[5], g[1] * g[2]));
    const double c22 = dfma(g[8], g[8], dfma(g[5], g[5], g[2] * g[2]));
    // lambda^3 - k2 lambda^2 + k1 lambda - k0, k0 = det(G)^2
    const double k2 = (c00 + c11) + c22;
    const double k1 = dfma(c00, c11, -(c01 * c01)) + dfma(c00, c22, -(c02 * c02)) + dfma(c11, c22, -(c12 * c12));
    const double dg = dfma(g[0], dfma(g[4], g[8], -(g[5] * g[7])),
                           dfma(-g[1], dfma(g[3], g[8], -(g[5] * g[6])), g[2] * dfma(g[3], g[7], -(g[4] * g[6]))));
    const double k0 = dg * dg;
    const double m2k2 = -2.0 * k2;
    double lam = 0.0;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const double f = dfma(dfma(lam - k2, lam, k1), lam, -k0);
        const double fp = dfma(dfma(3.0, lam, m2k2), lam, k1);
        double r = __builtin_amdgcn_rcp(fp);
        r = r * dfma(-fp, r, 2.0);
        lam = dfma(-f, r, lam);
    }
    // null vector of C - lambda I: the largest of the three cross products of its rows
    const double m00 = c00 - lam, m11 = c11 - lam, m22 = c22 - lam;
    const double ax = dfma(c01, c12, -(c02 * m11)), ay = dfma(c02, c01, -(m00 * c12)), az = dfma(m00, m11, -(c01 * c01));
    const double bx = dfma(c01, m22, -(c02 * c12)), by = dfma(c02, c02, -(m00 * m22)), bz = dfma(m00, c12, -(c01 * c02));
    const double cx = dfma(m11, m22, -(c12 * c12)), cy = dfma(c12, c02, -(c01 * m22)), cz = dfma(c01, c12, -(m11 * c02));
    const double na = dfma(az, az, dfma(ay, ay, ax * ax));
    const double nb = dfma(bz, bz, dfma(by, by, bx * bx));
    const double nc = dfma(cz, cz, dfma(cy, cy, cx * cx));
    const bool ub = nb > na;
    double vx = ub ? bx : ax, vy = ub ? by : ay, vz = ub ? bz : az, nv = ub ? nb : na;
    const bool uc = nc > nv;
    vx = uc ? cx : vx; vy = uc ? cy : vy; vz = uc ? cz : vz; nv = uc ? nc : nv;
    bool ok = nv >= 0x1p-190;            // (also false for a NaN)
    double h;
    (void)sqrt_fast_nz_h(nv, h);
    h += h;                              // ~ 1 / ||v||
    vx *= h; vy *= h; vz *= h;
    // w = G v, X = G - w v^T
    const double w0 = dfma(g[2], vz, dfma(g[1], vy, g[0] * vx));
    const double w1 = dfma(g[5], vz, dfma(g[4], vy, g[3] * vx));
    const double w2 = dfma(g[8], vz, dfma(g[7], vy, g[6] * vx));
    X[0] = dfma(-w0, vx, g[0]); X[1] = dfma(-w0, vy, g[1]); X[2] = dfma(-w0, vz, g[2]);
    X[3] = dfma(-w1, vx, g[3]); X[4] = dfma(-w1, vy, g[4]); X[5] = dfma(-w1, vz, g[5]);
    X[6] = dfma(-w2, vx, g[6]); X[7] = dfma(-w2, vy, g[7]); X[8] = dfma(-w2, vz, g[8]);
    const double sg2 = dfma(w2, w2, dfma(w1, w1, w0 * w0));
    const double sig = dsqrt(sg2);
    // u = w / sig, eps2 = || G^T u - sig v ||  (garbage when sig is tiny: then the other branch is taken)
    const double rs = div_fast(1.0, fmax(sig, 0x1p-190));
    const double u0 = w0 * rs, u1 = w1 * rs, u2 = w2 * rs;
    const double e0 = dfma(g[6], u2, dfma(g[3], u1, dfma(g[0], u0, -(sig * vx))));
    const double e1 = dfma(g[7], u2, dfma(g[4], u1, dfma(g[1], u0, -(sig * vy))));
    const double e2 = dfma(g[8], u2, dfma(g[5], u1, dfma(g[2], u0, -(sig * vz))));
    const double eps2 = sqrt_bound0(dfma(e2, e2, dfma(e1, e1, e0 * e0))) * (1.0 + 1e-14);   // (kPsTrip covers the rest)
    const bool useA = (eps2 < sig) && (sig >= 0x1p-190);
    e_out = (useA ? eps2 : sig) + kPsTrip;
    sige_out = useA ? sig : 0.0;
    extra_out = useA ? eps2 + kPsTrip : kPsTrip;
    ok = ok && (sig == sig);
    // second singular value of X from q1 = s1^2 + s2^2 and q2 = s1^4 + s2^4:  P = s1^2 s2^2 = (q1^2 - q2) / 2,
    // s2^2 = 2 P / (q1 + sqrt(q1^2 - 4 P)); every rounding is pushed towards a smaller result
    const double q1 = dfma(X[8], X[8], dfma(X[7], X[7], dfma(X[6], X[6], dfma(X[5], X[5], dfma(X[4], X[4],
                      dfma(X[3], X[3], dfma(X[2], X[2], dfma(X[1], X[1], X[0] * X[0]))))))));
    const double t00 = dfma(X[6], X[6], dfma(X[3], X[3], X[0] * X[0]));
    const double t01 = dfma(X[6], X[7], dfma(X[3], X[4], X[0] * X[1]));
    const double t02 = dfma(X[6], X[8], dfma(X[3], X[5], X[0] * X[2]));
    const double t11 = dfma(X[7], X[7], dfma(X[4], X[4], X[1] * X[1]));
    const double t12 = dfma(X[7], X[8], dfma(X[4], X[5], X[1] * X[2]));
    const double t22 = dfma(X[8], X[8], dfma(X[5], X[5], X[2] * X[2]));
    const double off = dfma(t12, t12, dfma(t02, t02, t01 * t01));
    const double q2 = dfma(2.0, off, dfma(t22, t22, dfma(t11, t11, t00 * t00)));
    const double q1q = q1 * q1;
    const double Pm = 0.5 * (q1q - q2) - 4e-15;
    const double disc = fmax(dfma(-4.0, Pm, q1q), 0.0) + 1e-13;
    const double s2q = (Pm + Pm) / (q1 * (1.0 + 1e-13) + dsqrt(disc));
    s2lb_out = s2q > 0.0 ? dsqrt(s2q) * (1.0 - 1e-14) : 0.0;
    ok_out = ok && (q1 <= 1.5);
}

// F = T2^T Fn T1 with the sample's Hartley transforms (eight_point_back's second half)
MVS_DEV void prescreen_denormalise(const double (&Fn)[9], const EightNorm &nm, double (&F)[9])
{
    const double s1 = nm.s1, s2 = nm.s2;
    const double tx1 = -nm.m1x * s1, ty1 = -nm.m1y * s1, tx2 = -nm.m2x * s2, ty2 = -nm.m2y * s2;
    double G[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        G[0][j] = s2 * Fn[j];
        G[1][j] = s2 * Fn[3 + j];
        G[2][j] = dfma(tx2, Fn[j], dfma(ty2, Fn[3 + j], Fn[6 + j]));
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        F[i * 3 + 0] = G[i][0] * s1;
        F[i * 3 + 1] = G[i][1] * s1;
        F[i * 3 + 2] = dfma(G[i][0], tx1, dfma(G[i][1], ty1, G[i][2]));
    }
}

constexpr int kPsTri = 28;      // strict upper triangle of R
constexpr int kPsParked = 26;   // ... of which this many are parked in LDS, [kPsParked][64 lanes] doubles per wavefront (13 KB:
                                // twelve wavefronts per CU); the last two (R_57, R_67, produced last) stay in registers
MVS_DEV constexpr int ps_tri(int i, int k) { return k * (k - 1) / 2 + i; }   // (i, k), i < k  ->  0 .. 27

// flag (kPs*), F~ and band for one sample.  P: the pair's points [M][4]; idx: the sample; park: this lane's column of the
// wavefront's LDS block (stride 64 doubles).
//
// Left-looking Householder QR of A^T (9 x 8): column j (= row j of A, a function of sample point j alone) is rebuilt from
// a fresh gather of point j, the reflectors 0 .. j-1 are applied to it, its entries above the diagonal are final entries of
// R and are parked in LDS, the rest defines reflector j.  Neither A nor the normalised sample ever exists as a whole in
// registers: live state is the reflectors (44 doubles) + one column, and the kernel fits two wavefronts per SIMD.
template <int VAR>
MVS_DEV int prescreen_hypothesis(const double *P, int (&idx)[8], double *park, const PairBox &bx, double thr, double (&F)[9],
                                 double &band_out, double &e32_out, bool &bad3)
{
    EightNorm nm;
    bool ok, tiny = false;
    {
        double x1[8], y1[8], x2[8], y2[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double4 p = *reinterpret_cast<const double4 *>(P + (size_t)idx[k] * 4);
            x1[k] = p.x; y1[k] = p.y; x2[k] = p.z; y2[k] = p.w;
        }
        ok = sample_norm(x1, y1, nm.s1, nm.m1x, nm.m1y, tiny);
        ok = sample_norm(x2, y2, nm.s2, nm.m2x, nm.m2y, tiny) && ok;
    }
    band_out = 0.0;
    e32_out = 0.0;
    if (!ok && !tiny) {   // reference: assert(scale > epsilon); the exact path rejects the sample too (same bits, same decision)
#pragma unroll
        for (int k = 0; k < 9; ++k)
            F[k] = 0.0;
        return kPsInvalid;
    }
    double v[8][9];   // v[k][k..8]: reflector k (entries below k are never touched)
    double rlast[kPsTri - kPsParked];
    double rd[8], beta[8];   // rd: RECIPROCALS of R's diagonal
    double S = 0.0;   // ||A||_F^2
    bool piv_ok = !tiny;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double col[9];
        {
            // a fresh gather: the first one's values must not stay alive across the QR, and the scheduler must not hoist the
            // later columns above the earlier reflectors (it would rebuild the whole matrix in registers): the index is
            // made opaque AND tied to the previous reflector's scale
            // ... of the column before the previous one: the gather of column j is in flight while column j - 1 is reduced
            if (j < 2)
                asm volatile("" : "+v"(idx[j]));
            else
                asm volatile("" : "+v"(idx[j]) : "v"(beta[j > 1 ? j - 2 : 0]));
            const double4 p = *reinterpret_cast<const double4 *>(P + (size_t)idx[j] * 4);
            const double a1 = (p.x - nm.m1x) * nm.s1, b1 = (p.y - nm.m1y) * nm.s1;   // normalise8's own operations
            const double a2 = (p.z - nm.m2x) * nm.s2, b2 = (p.w - nm.m2y) * nm.s2;
            // design matrix row (fundamental-matrix.cpp:78-87), the exact path's products
            col[0] = a2 * a1; col[1] = a2 * b1; col[2] = a2;
            col[3] = b2 * a1; col[4] = b2 * b1; col[5] = b2;
            col[6] = a1;      col[7] = b1;      col[8] = 1.0;
            // ||row||^2 = (a2^2 + b2^2 + 1)(a1^2 + b1^2 + 1): an upper estimate is all the bound needs
            S = dfma(dfma(a2, a2, dfma(b2, b2, 1.0)), dfma(a1, a1, dfma(b1, b1, 1.0)), S);
        }
#pragma unroll
        for (int k = 0; k < j; ++k) {
            double d = 0.0, d1 = 0.0;   // two partial sums: half the length of the dependent chain
#pragma unroll
            for (int i = k; i < 9; i += 2) {
                d = dfma(v[k][i], col[i], d);
                if (i + 1 < 9)
                    d1 = dfma(v[k][i + 1], col[i + 1], d1);
            }
            d += d1;
            const double t = beta[k] * d;
#pragma unroll
            for (int i = k; i < 9; ++i)
                col[i] = dfma(-t, v[k][i], col[i]);
            if (ps_tri(k, j) < kPsParked)
                park[ps_tri(k, j) * 64] = col[k];   // R_kj
            else
                rlast[ps_tri(k, j) - kPsParked] = col[k];
        }
        double ss = 0.0;
#pragma unroll
        for (int i = j; i < 9; ++i)
            ss = dfma(col[i], col[i], ss);
        piv_ok = piv_ok && (ss >= 0x1p-190);    // a (nearly) dependent row: no certificate; also keeps every divisor inside
        double hn;                              // the range of the unscaled sequences
        const double nrm = sqrt_fast_nz_h(ss, hn);   // hn ~ 1 / (2 nrm), a by-product of the square root's own sequence
        const double x0 = col[j];
        const double alpha = x0 >= 0.0 ? -nrm : nrm;
        const double vv = nrm * (nrm + dabs(x0));    // = v.v / 2
        // 1 / R_jj for the triangular inverse below: 2 hn and one Newton step against nrm itself (relative error < 3 u
        // whatever the seed's last bits were) -- the pivots used to be divided out again there, eight reciprocal
        // sequences of eleven instructions each
        double iv = hn + hn;
        iv = dfma(iv, dfma(-nrm, iv, 1.0), iv);
        rd[j] = x0 >= 0.0 ? -iv : iv;                // = 1 / alpha
        beta[j] = rcp_bound(vv);                     // within 2^-49 of 1 / vv: (iii)'s eps_beta <= 28 u
        v[j][j] = x0 - alpha;                        // v0: same sign as x0, no cancellation
#pragma unroll
        for (int i = j + 1; i < 9; ++i)
            v[j][i] = col[i];
    }
    S *= 1.0 + 1e-12;
    piv_ok = piv_ok && (S <= 0x1p100);
    const double sqrtS = sqrt_bound(S) * (1.0 + 1e-12);   // (S >= 8: every row of A holds a 1)
    // n~ = H_0 H_1 ... H_7 e_8
    double n[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
        n[i] = i == 8 ? 1.0 : 0.0;
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        double d = 0.0, d1 = 0.0;
#pragma unroll
        for (int i = k; i < 9; i += 2) {
            d = dfma(v[k][i], n[i], d);
            if (i + 1 < 9)
                d1 = dfma(v[k][i + 1], n[i + 1], d1);
        }
        d += d1;
        const double t = beta[k] * d;
#pragma unroll
        for (int i = k; i < 9; ++i)
            n[i] = dfma(-t, v[k][i], n[i]);
    }
    // ||R^-1||_F by explicit back substitution, column by column; R's strict upper triangle comes back from LDS (the
    // reflectors are dead by now)
    double y2sum = 0.0;
    {
        double R[kPsTri];
#pragma unroll
        for (int q = 0; q < kPsTri; ++q)
            R[q] = q < kPsParked ? park[q * 64] : rlast[q - kPsParked];
        double inv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            inv[i] = rd[i];   // the reciprocal pivots, from the QR loop
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
                    s = dfma(R[ps_tri(i, k)], y[k], s);
                y[i] = -(s * inv[i]);
                y2sum = dfma(y[i], y[i], y2sum);
            }
        }
    }
    const double yf = sqrt_bound(y2sum) * (1.0 + 1e-12);  // (y2sum >= 1 / pivot^2 >= 2^-380; a zero pivot fails piv_ok)
    // residual of n~ against the rows of A -- a priori: with A^T + E = Q~ [R^; 0] (||E||_F <= 176 u ||A||_F, Q~ within 510 u of
    // orthogonal: DESIGN.md 4.3e (iii)) and n~ = the computed Q~ e_9 (eight reflector applications, <= 176 u of rounding),
    // A n~ = [R^T 0] Q~^T n~ - E^T n~ and Q~^T n~ = e_9 up to 1020 u + 176 u, so || A n~ || <= 1372 u ||A||_F + 176 u ||A||_F
    // = 1548 u ||A||_F = 1.72e-13 ||A||_F <= 1.8e-13 sqrtS.
    // (The first version measured it from a second gather of the sample: 208 flop and eight loads per hypothesis for a term
    // that is five orders of magnitude below eta_J.)
    const double rho = 1.8e-13 * sqrtS;
    // sigma_8(A) >= (1 - z) / ||R^-1||_F (1 - 510 u) - 176 u ||A||_F
    const double z = 16.0 * kPsU * sqrtS * yf;
    const double sig8 = (1.0 - z) * rcp_bound(yf) * (1.0 - 1e-13) - 4e-14 * sqrtS;   // (510 u + the reciprocal's 2^-49 < 1e-13)
    const double rg = rcp_bound(sig8);   // (NaN / inf for sig8 <= 0: the certificate is refused below)
    const double eta_j = 1.01 * kPsTauC * S * (rg * rg) + kPsEtaQ;   // (1.01 * 2.0e-12 is 13 % above 2.001 (8000 u + 8.01 u))
    const double eta_a = 1.5 * rho * rg + 1e-13;
    // rank-2 through one verified singular triplet of reshape(n~), de-normalisation
    double Fn[9], e3, sige, extra, s2lb;
    bool ok3;
    prescreen_rank2(n, Fn, e3, sige, extra, s2lb, ok3);
    prescreen_denormalise(Fn, nm, F);
    bad3 = false;
    const double eta = (eta_j + eta_a + e3 + kPsSvd3) * (1.0 + 1e-12);
    const double delta = ((s2lb - extra) - sige) - eta;
    const double dfn = (2.0 + 2.0 * (sige + 3.0 * eta) * rcp_bound(delta) * (1.0 + 1e-14)) * eta + extra + kPsSvd3;
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
    const double band = (dfn * sqrt_bound(n12) * (1.0 + 1e-9) + 64.0 * kPsU * sqrt_bound(n12p)) * (1.0 + 1e-9) + 1e-15 * thr;
    band_out = band;
    // single-precision counting: r32 = the residual evaluated in binary32 on F~ and the point rounded to binary32, either as
    // the nested fma chain of ransac_count32_kernel (a term p2_j F_jk p1_k passes at most 7 roundings: three inputs, four
    // fma) or as the matrix-core form of ransac_count_mfma_kernel (three inputs, the rounded monomial p2_j p1_k, and an fmaf
    // chain over the ten k: at most 14):
    // | r32 - r(F~, p) | <= 16 * 2^-24 * T,  T = sum |p2_j| |F_jk| |p1_k| <= [X2 Y2 1] |F~| [X1 Y1 1]^T over the pair's box
    {
        const double X1 = fmax(dabs(bx.x1lo), dabs(bx.x1hi)), Y1 = fmax(dabs(bx.y1lo), dabs(bx.y1hi));
        const double X2 = fmax(dabs(bx.x2lo), dabs(bx.x2hi)), Y2 = fmax(dabs(bx.y2lo), dabs(bx.y2hi));
        const double t0 = dfma(X2, dabs(F[0]), dfma(Y2, dabs(F[3]), dabs(F[6])));
        const double t1 = dfma(X2, dabs(F[1]), dfma(Y2, dabs(F[4]), dabs(F[7])));
        const double t2 = dfma(X2, dabs(F[2]), dfma(Y2, dabs(F[5]), dabs(F[8])));
        const double T = dfma(t0, X1, dfma(t1, Y1, t2));
        e32_out = 16.0 * 0x1p-24 * T * (1.0 + 1e-6) + 1e-30;
    }
    // every comparison is written so that a NaN anywhere lands in "needs the exact solve"; the caller adds the band test
    // of its counting precision (band, or band + e32, against kPsBandFrac * thr)
    const bool certified = piv_ok && ok3 && (z < 0.5) && (sig8 > 0.0) && (delta > 0.0) && (eta < 1e-3) && (band < 0x1p100);
    return certified ? kPsApprox : kPsNeedExact;
}

}  // namespace mvs
