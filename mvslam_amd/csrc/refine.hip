// refine.hip -- sfm_refine / pnp_refine (vision/sfm-refine.cpp:20-139, vision/pnp-refine.cpp:14-108; SURVEY
// section 8 row f4) as ONE batched Levenberg-Marquardt kernel.
//
// Both reference functions hand a factor graph to GTSAM through ba_frame_pose_and_point (vision/ba.cpp:26-156):
// diagonal priors on the camera poses, priors on the points, one pinhole projection factor per observation; they
// return the minimiser, the marginal covariances of the linearised graph and the final error.  GTSAM is third-party
// and absent from the reference tree, so this is the build's own solver of that least-squares problem
// (DESIGN.md section 4.7):
//   one workgroup (256 threads) per problem, the whole LM loop inside one launch;
//   points are strided over the threads; each thread eliminates its points (3x3 inverse) and accumulates its share
//   of the reduced camera system S (6F x 6F, F = 1 or 2 cameras) in registers;
//   S, the right-hand side and the cost are reduced in a FIXED order (per-thread partials in point order, butterfly
//   inside each wavefront, then (w0 + w1) + (w2 + w3) through LDS), so every thread of the group ends up with the same
//   bits, takes the same accept / reject decision and the result does not depend on scheduling;
//   every thread then factors S (Cholesky, 12x12 at most) redundantly -- cheaper than a broadcast;
//   the covariance pass re-linearises at the estimate with lambda = 0: pose covariance = block of S^-1, point
//   covariance = P + P Hcp^T S^-1 Hcp P.
// The CPU oracle (oracle/mvs_refine_oracle.c) follows the same summation order; sin / cos / atan2 come from different
// libraries on the two sides, so parity for this row is by tolerance.
#include "kernels.hpp"

namespace mvs {

namespace {

constexpr int kRefineThreads = 256;

__host__ __device__ constexpr int lidx(int r, int c) { return r * (r + 1) / 2 + c; }  // r >= c

struct Cam {
    double fx, fy, sk, cx, cy;
};

// fused building blocks (one v_fma_f64 each); the oracle uses the same ones in the same places
__device__ __forceinline__ double fd2(double a0, double b0, double a1, double b1) { return fma(a1, b1, a0 * b0); }
__device__ __forceinline__ double fd3(double a0, double b0, double a1, double b1, double a2, double b2)
{
    return fma(a2, b2, fma(a1, b1, a0 * b0));
}
__device__ __forceinline__ double fx2(double a, double b, double c, double d) { return fma(a, b, -(c * d)); }  // a b - c d

__device__ __forceinline__ void so3_exp(const double (&w)[3], double (&R)[9])
{
    const double th2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
    const double th = sqrt(th2);
    double A, B;
    if (th < 1e-4) {
        A = 1.0 - th2 / 6.0;
        B = 0.5 - th2 / 24.0;
    } else {
        A = sin(th) / th;
        B = (1.0 - cos(th)) / th2;
    }
    const double x = w[0], y = w[1], z = w[2];
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

__device__ __forceinline__ void so3_log(const double (&R)[9], double (&w)[6])
{
    const double vx = 0.5 * (R[7] - R[5]), vy = 0.5 * (R[2] - R[6]), vz = 0.5 * (R[3] - R[1]);
    const double s = sqrt((vx * vx + vy * vy) + vz * vz);
    const double c = 0.5 * (((R[0] + R[4]) + R[8]) - 1.0);
    const double th = atan2(s, c);
    double k;
    if (s < 1e-4 && c > 0.0)
        k = 1.0 + (s * s) / 6.0;
    else
        k = th / s;
    w[0] = k * vx;
    w[1] = k * vy;
    w[2] = k * vz;
}

__device__ __forceinline__ void so3_jrinv(const double (&w)[6], double (&J)[9])
{
    const double th2 = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
    const double th = sqrt(th2);
    double g;
    if (th < 1e-4)
        g = 1.0 / 12.0 + th2 / 720.0;
    else
        g = 1.0 / th2 - (1.0 + cos(th)) / ((2.0 * th) * sin(th));
    const double x = w[0], y = w[1], z = w[2];
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

__device__ __forceinline__ void sym3_inverse(const double (&a)[6], double (&o)[6])
{
    const double c00 = fx2(a[3], a[5], a[4], a[4]);
    const double c01 = fx2(a[2], a[4], a[1], a[5]);
    const double c02 = fx2(a[1], a[4], a[2], a[3]);
    const double det = fd3(a[0], c00, a[1], c01, a[2], c02);
    const double id = 1.0 / det;
    o[0] = c00 * id;
    o[1] = c01 * id;
    o[2] = c02 * id;
    o[3] = fx2(a[0], a[5], a[2], a[2]) * id;
    o[4] = fx2(a[1], a[2], a[0], a[4]) * id;
    o[5] = fx2(a[0], a[3], a[1], a[1]) * id;
}

// pose prior of one frame: error e = (Log(R0^T R), R0^T (t - t0)); Jw = Jr^-1(e_w), Jv = R0^T R
template <bool JAC>
__device__ __forceinline__ void pose_prior(const double (&R0)[9], const double (&t0)[3], const double (&R)[9],
                                           const double (&t)[3], double (&e)[6], double (&Jw)[9], double (&Jv)[9])
{
    double Re[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Re[3 * i + j] = (R0[i] * R[j] + R0[3 + i] * R[3 + j]) + R0[6 + i] * R[6 + j];
    so3_log(Re, e);
    const double d0 = t[0] - t0[0], d1 = t[1] - t0[1], d2 = t[2] - t0[2];
    e[3] = (R0[0] * d0 + R0[3] * d1) + R0[6] * d2;
    e[4] = (R0[1] * d0 + R0[4] * d1) + R0[7] * d2;
    e[5] = (R0[2] * d0 + R0[5] * d1) + R0[8] * d2;
    if (JAC) {
        so3_jrinv(e, Jw);
#pragma unroll
        for (int k = 0; k < 9; ++k)
            Jv[k] = Re[k];
    }
}

// residual of one observation (and its Jacobians): GenericProjectionFactor with throwCheirality = false
template <bool JAC>
__device__ __forceinline__ void project_lin(const Cam &cam, const double (&R)[9], const double (&t)[3],
                                            const double (&p)[3], double u, double v, double (&r)[2], double (&Jc)[12],
                                            double (&Jp)[6])
{
    const double d0 = p[0] - t[0], d1 = p[1] - t[1], d2 = p[2] - t[2];
    const double q0 = fd3(R[0], d0, R[3], d1, R[6], d2);
    const double q1 = fd3(R[1], d0, R[4], d1, R[7], d2);
    const double q2 = fd3(R[2], d0, R[5], d1, R[8], d2);
    if (!(q2 > 0.0)) {
        r[0] = 2.0 * cam.fx;
        r[1] = 2.0 * cam.fx;
        if (JAC) {
#pragma unroll
            for (int k = 0; k < 12; ++k)
                Jc[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k)
                Jp[k] = 0.0;
        }
        return;
    }
    const double iz = 1.0 / q2, xn = q0 * iz, yn = q1 * iz;
    const double un = fd2(cam.fx, xn, cam.sk, yn), vn = cam.fy * yn;
    r[0] = (un + cam.cx) - u;
    r[1] = (vn + cam.cy) - v;
    if (JAC) {
        const double A[6] = {cam.fx * iz, cam.sk * iz, -(un * iz), 0.0, cam.fy * iz, -(vn * iz)};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const double a0 = A[3 * a], a1 = A[3 * a + 1], a2 = A[3 * a + 2];
            Jc[6 * a + 0] = fx2(a1, q2, a2, q1);
            Jc[6 * a + 1] = fx2(a2, q0, a0, q2);
            Jc[6 * a + 2] = fx2(a0, q1, a1, q0);
            Jc[6 * a + 3] = -a0;
            Jc[6 * a + 4] = -a1;
            Jc[6 * a + 5] = -a2;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                Jp[3 * a + k] = fd3(a0, R[3 * k], a1, R[3 * k + 1], a2, R[3 * k + 2]);
        }
    }
}

template <int F>
struct Dims {
    static constexpr int NC = 6 * F;
    static constexpr int NL = NC * (NC + 1) / 2;
    static constexpr int NV = NL + NC + 1;  // S (packed lower), b, cost
};

// problem view of one workgroup
template <int F>
struct Prob {
    Cam cam;
    int m;
    const double *obs[F];
    const double *oinfo[F];
    const double *pts0;
    const double *pinfo;
    // LDS-resident copy of the inputs of points [0, n_res): constants [NCONST][CAP] and the current / trial estimate
    // [2][3][CAP] (slot `cur` is current)
    double *res_c, *res_p;
    int n_res, cur;
};

// How many points keep their inputs in LDS.  The kernel runs at one wavefront per SIMD, so nothing hides a global load:
// every visit of a point exposed one full load latency (~2 k clocks, a third of the kernel's cycles after the
// reductions stopped waiting on ds_bpermute).  A point's inputs are read in every pass (two accumulation passes, the
// update pass and the cost, per linear solve) but written only here and by the point's own thread, so they live in the
// 160 KB of LDS the workgroup has to itself: 25 (F = 2) or 20 (F = 1) doubles per point.  Points beyond the capacity
// keep the global path (a wave-uniform branch: the capacity is a multiple of 64).
template <int F>
struct Res {
    static constexpr int NCONST = 6 + 3 + 2 * F + 3 * F;
    static constexpr int CAP = F == 2 ? 768 : 960;
};

// ---- per-point inputs -------------------------------------------------------------------------------------------------
// Every loop over a thread's points goes through for_points(), which loads the point's inputs once into PtIn.
// (Tried in round 2 and dropped: the inputs of the wavefront's next 64 points prefetched global -> LDS by LDS-DMA into a
// two-buffer ring per wavefront, retired with a counted s_waitcnt -- same results, 0.636 vs 0.596 ms per 512 pairs: the
// 13 DMA statements per point with their M0 set-up, the LDS reads and 36 B of scratch cost more than the exposed load
// latency they hide; the kernel's waits were the ds_bpermute chains of the reductions, see wave_sum.)
template <int F>
struct PtIn {
    double L[6];       // point prior information (packed symmetric)
    double p0[3];      // point prior mean
    double ob[F][2];   // observations
    double W[F][3];    // observation information (packed symmetric 2x2)
    double p[3];       // current estimate
};

// body(i, q) for every point i of this thread (i = tid, tid + 256, ...)
template <int F, typename Body>
__device__ __forceinline__ void for_points(const Prob<F> &P, const double *pts, Body body)
{
    constexpr int CAP = Res<F>::CAP;
    for (int i = threadIdx.x; i < P.m; i += kRefineThreads) {
        PtIn<F> q;
        if (i < P.n_res) {   // wave-uniform
            const double *c = P.res_c + i, *pc = P.res_p + (size_t)P.cur * 3 * CAP + i;
#pragma unroll
            for (int k = 0; k < 6; ++k)
                q.L[k] = c[k * CAP];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                q.p0[k] = c[(6 + k) * CAP];
                q.p[k] = pc[k * CAP];
            }
#pragma unroll
            for (int f = 0; f < F; ++f) {
                q.ob[f][0] = c[(9 + 5 * f) * CAP];
                q.ob[f][1] = c[(10 + 5 * f) * CAP];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    q.W[f][k] = c[(11 + 5 * f + k) * CAP];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k)
                q.L[k] = P.pinfo[6 * (size_t)i + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                q.p0[k] = P.pts0[3 * (size_t)i + k];
                q.p[k] = pts[3 * (size_t)i + k];
            }
#pragma unroll
            for (int f = 0; f < F; ++f) {
                q.ob[f][0] = P.obs[f][2 * (size_t)i];
                q.ob[f][1] = P.obs[f][2 * (size_t)i + 1];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    q.W[f][k] = P.oinfo[f][3 * (size_t)i + k];
            }
        }
        body(i, q);
    }
}

// linearise point i at (R, t, p): Hpp (+ prior), gp, Hcp; when ACC, the frames' own Hcc / gc blocks and the cost go
// straight into the thread's accumulators
// [LO, HI): the slice of the accumulator vector {S packed, b, cost} this call adds to (build_schur accumulates the
// vector in two passes over the points: all 91 accumulators at once do not fit in the register file beside the
// linearisation temporaries and went to scratch memory)
template <int F, bool ACC, int LO = 0, int HI = Dims<F>::NV>
__device__ __forceinline__ void point_linearize(const Prob<F> &P, const double (&R)[F][9], const double (&t)[F][3],
                                                const double (&p)[3], const PtIn<F> &q, double (&Hpp)[6], double (&gp)[3],
                                                double (&Hcp)[6 * F][3], double (&acc)[Dims<F>::NV])
{
    constexpr int NL = Dims<F>::NL, NC = Dims<F>::NC;
    auto in = [](int idx) { return idx >= LO && idx < HI; };
    const double (&L)[6] = q.L;
    const double d0 = p[0] - q.p0[0], d1 = p[1] - q.p0[1], d2 = p[2] - q.p0[2];
    const double Ld0 = fd3(L[0], d0, L[1], d1, L[2], d2), Ld1 = fd3(L[1], d0, L[3], d1, L[4], d2),
                 Ld2 = fd3(L[2], d0, L[4], d1, L[5], d2);
#pragma unroll
    for (int k = 0; k < 6; ++k)
        Hpp[k] = L[k];
    gp[0] = Ld0, gp[1] = Ld1, gp[2] = Ld2;
    double cost = fd3(d0, Ld0, d1, Ld1, d2, Ld2);
#pragma unroll
    for (int f = 0; f < F; ++f) {
        double r[2], Jc[12], Jp[6];
        project_lin<true>(P.cam, R[f], t[f], p, q.ob[f][0], q.ob[f][1], r, Jc, Jp);
        const double W0 = q.W[f][0], W1 = q.W[f][1], W2 = q.W[f][2];
        const double wr0 = fd2(W0, r[0], W1, r[1]), wr1 = fd2(W1, r[0], W2, r[1]);
        cost = fma(r[1], wr1, fma(r[0], wr0, cost));
        double WJc[12], WJp[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            WJc[k] = fd2(W0, Jc[k], W1, Jc[6 + k]);
            WJc[6 + k] = fd2(W1, Jc[k], W2, Jc[6 + k]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            WJp[k] = fd2(W0, Jp[k], W1, Jp[3 + k]);
            WJp[3 + k] = fd2(W1, Jp[k], W2, Jp[3 + k]);
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            if (ACC) {
#pragma unroll
                for (int b = 0; b <= a; ++b)
                    if (in(lidx(6 * f + a, 6 * f + b)))
                        acc[lidx(6 * f + a, 6 * f + b)] =
                            acc[lidx(6 * f + a, 6 * f + b)] + fd2(Jc[a], WJc[b], Jc[6 + a], WJc[6 + b]);
                if (in(NL + 6 * f + a))
                    acc[NL + 6 * f + a] = acc[NL + 6 * f + a] - fd2(Jc[a], wr0, Jc[6 + a], wr1);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k)
                Hcp[6 * f + a][k] = fd2(Jc[a], WJp[k], Jc[6 + a], WJp[3 + k]);
        }
        Hpp[0] = fma(Jp[3], WJp[3], fma(Jp[0], WJp[0], Hpp[0]));
        Hpp[1] = fma(Jp[3], WJp[4], fma(Jp[0], WJp[1], Hpp[1]));
        Hpp[2] = fma(Jp[3], WJp[5], fma(Jp[0], WJp[2], Hpp[2]));
        Hpp[3] = fma(Jp[4], WJp[4], fma(Jp[1], WJp[1], Hpp[3]));
        Hpp[4] = fma(Jp[4], WJp[5], fma(Jp[1], WJp[2], Hpp[4]));
        Hpp[5] = fma(Jp[5], WJp[5], fma(Jp[2], WJp[2], Hpp[5]));
#pragma unroll
        for (int k = 0; k < 3; ++k)
            gp[k] = fma(Jp[3 + k], wr1, fma(Jp[k], wr0, gp[k]));
    }
    if (ACC && in(NL + NC))
        acc[NL + NC] = acc[NL + NC] + cost;
}

template <int F>
__device__ __forceinline__ double point_cost(const Prob<F> &P, const double (&R)[F][9], const double (&t)[F][3],
                                             const double (&p)[3], const PtIn<F> &q)
{
    const double (&L)[6] = q.L;
    const double d0 = p[0] - q.p0[0], d1 = p[1] - q.p0[1], d2 = p[2] - q.p0[2];
    const double Ld0 = fd3(L[0], d0, L[1], d1, L[2], d2), Ld1 = fd3(L[1], d0, L[3], d1, L[4], d2),
                 Ld2 = fd3(L[2], d0, L[4], d1, L[5], d2);
    double c = fd3(d0, Ld0, d1, Ld1, d2, Ld2);
#pragma unroll
    for (int f = 0; f < F; ++f) {
        double r[2], Jc[12], Jp[6];
        project_lin<false>(P.cam, R[f], t[f], p, q.ob[f][0], q.ob[f][1], r, Jc, Jp);
        const double W0 = q.W[f][0], W1 = q.W[f][1], W2 = q.W[f][2];
        const double wr0 = fd2(W0, r[0], W1, r[1]), wr1 = fd2(W1, r[0], W2, r[1]);
        c = fma(r[1], wr1, fma(r[0], wr0, c));
    }
    return c;
}

// x of the lane a DPP control selects (two 32-bit moves; no LDS round trip, unlike ds_bpermute)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double x, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
// sum over the wavefront, the xor butterfly's tree ((x0 + x1) + (x2 + x3)) + ... in every lane: steps 1 and 2 are quad
// permutes; after them a quad is uniform, so the lane a half-row mirror (i -> 7 - i) selects holds what lane i ^ 4 holds,
// likewise the row mirror for i ^ 8; rows are then uniform and steps 16 / 32 are (r0 + r1) + (r2 + r3) of the four row
// values.  Bit-identical to x = x + __shfl_xor(x, s) for s = 1 .. 32, which cost 12 ds_bpermute round trips per value
// (the kernel spent 47 % of its cycles waiting on them at one wavefront per SIMD).
__device__ __forceinline__ double wave_sum(double x)
{
    x = x + dpp_f64<0xb1>(x);    // quad_perm [1, 0, 3, 2]
    x = x + dpp_f64<0x4e>(x);    // quad_perm [2, 3, 0, 1]
    x = x + dpp_f64<0x141>(x);   // row_half_mirror
    x = x + dpp_f64<0x140>(x);   // row_mirror
    const double r0 = readlane_f64(x, 0), r1 = readlane_f64(x, 16), r2 = readlane_f64(x, 32), r3 = readlane_f64(x, 48);
    return (r0 + r1) + (r2 + r3);
}

// a value every lane holds with the same bits, moved to scalar registers (the pose is wave-uniform: 24 doubles per frame
// pair that would otherwise occupy vector registers in every lane of a kernel that has none to spare)
__device__ __forceinline__ double uniform_f64(double x)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

// fixed-order sum over the workgroup; every thread returns with the same totals
template <int NV>
__device__ __forceinline__ void block_reduce(double (&v)[NV], double *red)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double x = wave_sum(v[k]);
        if (lane == 0)
            red[wave * NV + k] = x;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k)
        v[k] = (red[k] + red[NV + k]) + (red[2 * NV + k] + red[3 * NV + k]);
    __syncthreads();
}

__device__ __forceinline__ double block_reduce1(double x, double *red)
{
    double v[1] = {x};
    block_reduce<1>(v, red);
    return v[0];
}

template <int F>
__device__ __forceinline__ double prior_cost(const RefineCfg &cfg, const double (&R0)[F][9], const double (&t0)[F][3],
                                             const double (&R)[F][9], const double (&t)[F][3])
{
    double c = 0.0;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        double e[6], Jw[9], Jv[9];
        pose_prior<false>(R0[f], t0[f], R[f], t[f], e, Jw, Jv);
#pragma unroll
        for (int k = 0; k < 6; ++k)
            c = c + (e[k] * e[k]) * cfg.w[f][k];
    }
    return c;
}

// in-place Cholesky of the packed lower triangle; false if not positive definite.  The diagonal holds 1 / l_jj (one
// division per column; the other 78 divisions of a factorisation + solve are multiplications by it, as in the oracle)
template <int N>
__device__ __forceinline__ bool chol_packed(double (&S)[N * (N + 1) / 2])
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double d = S[lidx(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k)
            d = fma(-S[lidx(j, k)], S[lidx(j, k)], d);
        ok = ok && (d > 0.0) && (d < __builtin_inf());
        const double inv = 1.0 / sqrt(d);
        S[lidx(j, j)] = inv;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double v = S[lidx(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k)
                v = fma(-S[lidx(i, k)], S[lidx(j, k)], v);
            S[lidx(i, j)] = v * inv;
        }
    }
    return ok;
}

template <int N>
__device__ __forceinline__ void chol_solve(const double (&Lc)[N * (N + 1) / 2], double (&b)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k)
            v = fma(-Lc[lidx(i, k)], b[k], v);
        b[i] = v * Lc[lidx(i, i)];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double v = b[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k)
            v = fma(-Lc[lidx(k, i)], b[k], v);
        b[i] = v * Lc[lidx(i, i)];
    }
}

// one pass over the thread's points adding the entries [LO, HI) of {S packed, b, cost} (per-thread partial sums)
template <int F, int LO, int HI>
__device__ __forceinline__ void schur_pass(const Prob<F> &P, const double (&R)[F][9], const double (&t)[F][3],
                                           const double *pts, double lam, double (&acc)[Dims<F>::NV])
{
    constexpr int NC = Dims<F>::NC, NL = Dims<F>::NL;
    auto in = [](int idx) { return idx >= LO && idx < HI; };
    for_points<F>(P, pts, [&](int, const PtIn<F> &q) {
        const double (&p)[3] = q.p;
        double Hpp[6], gp[3], Hcp[NC][3];
        point_linearize<F, true, LO, HI>(P, R, t, p, q, Hpp, gp, Hcp, acc);
        const double Hd[6] = {Hpp[0] + lam, Hpp[1], Hpp[2], Hpp[3] + lam, Hpp[4], Hpp[5] + lam};
        double Pi[6];
        sym3_inverse(Hd, Pi);
#pragma unroll
        for (int a = 0; a < NC; ++a) {   // row a of Y = Hcp P lives only for this row of the update
            const double y0 = fd3(Hcp[a][0], Pi[0], Hcp[a][1], Pi[1], Hcp[a][2], Pi[2]);
            const double y1 = fd3(Hcp[a][0], Pi[1], Hcp[a][1], Pi[3], Hcp[a][2], Pi[4]);
            const double y2 = fd3(Hcp[a][0], Pi[2], Hcp[a][1], Pi[4], Hcp[a][2], Pi[5]);
#pragma unroll
            for (int c = 0; c <= a; ++c)
                if (in(lidx(a, c)))
                    acc[lidx(a, c)] = fma(-y2, Hcp[c][2], fma(-y1, Hcp[c][1], fma(-y0, Hcp[c][0], acc[lidx(a, c)])));
            if (in(NL + a))
                acc[NL + a] = fma(y2, gp[2], fma(y1, gp[1], fma(y0, gp[0], acc[NL + a])));
        }
    });
}

// accumulate, reduce over the workgroup and park in LDS (red[4 NV + k]) the entries [LO, HI)
template <int F, int LO, int HI>
__device__ __forceinline__ void schur_slice(const Prob<F> &P, const double (&R)[F][9], const double (&t)[F][3],
                                            const double *pts, double lam, double (&acc)[Dims<F>::NV], double *red)
{
    constexpr int NV = Dims<F>::NV;
    schur_pass<F, LO, HI>(P, R, t, pts, lam, acc);
    double part[HI - LO];
#pragma unroll
    for (int k = 0; k < HI - LO; ++k)
        part[k] = acc[LO + k];
    block_reduce<HI - LO>(part, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < HI - LO; ++k)
            red[4 * NV + LO + k] = part[k];
    }
}

// reduced camera system at (R, t, pts) with damping lam.  On return acc = {S packed, b, cost} on every thread.
template <int F>
__device__ __forceinline__ void build_schur(const Prob<F> &P, const RefineCfg &cfg, const double (&R0)[F][9],
                                            const double (&t0)[F][3], const double (&R)[F][9], const double (&t)[F][3],
                                            const double *pts, double lam, double (&acc)[Dims<F>::NV], double *red)
{
    constexpr int NC = Dims<F>::NC, NL = Dims<F>::NL, NV = Dims<F>::NV;
#pragma unroll
    for (int k = 0; k < NV; ++k)
        acc[k] = 0.0;
    // several passes over the thread's points, each owning a slice of the accumulator vector (same per-point operations,
    // same point order per entry, same reduction tree: the sums have the same bits as a single pass); the totals of a
    // finished slice wait in LDS.  Two frames: rows 0..8 of S | rows 9..11, b, cost.
    if constexpr (F == 2) {
        schur_slice<F, 0, lidx(9, 0)>(P, R, t, pts, lam, acc, red);
        schur_slice<F, lidx(9, 0), NV>(P, R, t, pts, lam, acc, red);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NV; ++k)
            acc[k] = red[4 * NV + k];
        __syncthreads();   // the next user of `red` must not overtake the reads above
    } else {
        schur_pass<F, 0, NV>(P, R, t, pts, lam, acc);
        block_reduce<NV>(acc, red);
    }
#pragma unroll
    for (int f = 0; f < F; ++f) {
        double e[6], Jw[9], Jv[9];
        pose_prior<true>(R0[f], t0[f], R[f], t[f], e, Jw, Jv);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int o = 6 * f + 3 * half;
            const double w0 = cfg.w[f][3 * half], w1 = cfg.w[f][3 * half + 1], w2 = cfg.w[f][3 * half + 2];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double j0 = half ? Jv[a] : Jw[a], j1 = half ? Jv[3 + a] : Jw[3 + a], j2 = half ? Jv[6 + a] : Jw[6 + a];
#pragma unroll
                for (int c = 0; c <= a; ++c) {
                    const double k0 = half ? Jv[c] : Jw[c], k1 = half ? Jv[3 + c] : Jw[3 + c],
                                 k2 = half ? Jv[6 + c] : Jw[6 + c];
                    const double h = (j0 * w0 * k0 + j1 * w1 * k1) + j2 * w2 * k2;
                    acc[lidx(o + a, o + c)] = acc[lidx(o + a, o + c)] + h;
                }
                const double g = (j0 * w0 * e[3 * half] + j1 * w1 * e[3 * half + 1]) + j2 * w2 * e[3 * half + 2];
                acc[NL + o + a] = acc[NL + o + a] - g;
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k)
            acc[NL + NC] = acc[NL + NC] + (e[k] * e[k]) * cfg.w[f][k];
    }
#pragma unroll
    for (int a = 0; a < NC; ++a)
        acc[lidx(a, a)] = acc[lidx(a, a)] + lam;
}

template <int F>
__global__ __launch_bounds__(kRefineThreads) void refine_kernel(RefineDev d)
{
    constexpr int NC = Dims<F>::NC, NL = Dims<F>::NL, NV = Dims<F>::NV;
    __shared__ double red[5 * NV];   // 4 wavefront partials + the parked totals of build_schur's first pass
    __shared__ double Sinv[NC * NC];
    __shared__ double res_c[Res<F>::NCONST * Res<F>::CAP];
    __shared__ double res_p[2 * 3 * Res<F>::CAP];
    const int g = blockIdx.x;
    const RefineCfg &cfg = d.cfg;
    Prob<F> P;
    P.m = d.m[g];
    const size_t base = (size_t)g * d.stride;
    mvs_refine_result *out = d.out + g;
    if (P.m < 1 || P.m > d.stride) {
        if (threadIdx.x == 0) {
            out->ok = 0;
            out->iterations = 0;
            out->error = 0.0;
        }
        return;
    }
    {
        const double *K = d.K + 9 * (size_t)g;
        P.cam.fx = K[0], P.cam.sk = K[1], P.cam.cx = K[2], P.cam.fy = K[4], P.cam.cy = K[5];
    }
#pragma unroll
    for (int f = 0; f < F; ++f) {
        P.obs[f] = d.obs[f] + 2 * base;
        P.oinfo[f] = d.oinfo[f] + 3 * base;
    }
    P.pts0 = d.pts0 + 3 * base;
    P.pinfo = d.pinfo + 6 * base;
    double *pts = d.pts + 3 * base, *pts_new = d.pts_tmp + 3 * base;

    // prior means = guesses; the moving camera is frame F - 1, frame 0 of a two-view problem is the identity
    double R0[F][9], t0[F][3], R[F][9], t[F][3];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        if (d.pose0_all) {   // general two-frame problem: every frame has its own guess (= prior mean)
#pragma unroll
            for (int k = 0; k < 9; ++k)
                R0[f][k] = uniform_f64(d.pose0_all[12 * ((size_t)g * F + f) + k]);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                t0[f][k] = uniform_f64(d.pose0_all[12 * ((size_t)g * F + f) + 9 + k]);
        } else if (f == F - 1) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
                R0[f][k] = uniform_f64(d.pose0[12 * (size_t)g + k]);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                t0[f][k] = uniform_f64(d.pose0[12 * (size_t)g + 9 + k]);
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k)
                R0[f][k] = (k % 4 == 0) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                t0[f][k] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k)
            R[f][k] = R0[f][k];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            t[f][k] = t0[f][k];
    }
    P.res_c = res_c;
    P.res_p = res_p;
    P.n_res = min(P.m, Res<F>::CAP);
    P.cur = 0;
    for (int i = threadIdx.x; i < P.m; i += kRefineThreads) {
        double p0[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p0[k] = P.pts0[3 * (size_t)i + k];
            pts[3 * (size_t)i + k] = p0[k];
        }
        if (i < P.n_res) {   // the thread that owns point i is the only one that ever touches its LDS entries
            constexpr int CAP = Res<F>::CAP;
#pragma unroll
            for (int k = 0; k < 6; ++k)
                res_c[k * CAP + i] = P.pinfo[6 * (size_t)i + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                res_c[(6 + k) * CAP + i] = p0[k];
                res_p[k * CAP + i] = p0[k];
            }
#pragma unroll
            for (int f = 0; f < F; ++f) {
                res_c[(9 + 5 * f) * CAP + i] = P.obs[f][2 * (size_t)i];
                res_c[(10 + 5 * f) * CAP + i] = P.obs[f][2 * (size_t)i + 1];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    res_c[(11 + 5 * f + k) * CAP + i] = P.oinfo[f][3 * (size_t)i + k];
            }
        }
    }
    __syncthreads();

    // cost at the guess
    double cur;
    {
        double c = 0.0;
        for_points<F>(P, pts, [&](int, const PtIn<F> &q) { c = c + point_cost<F>(P, R, t, q.p, q); });
        cur = block_reduce1(c, red) + prior_cost<F>(cfg, R0, t0, R, t);
    }
    double lam = cfg.lambda_initial;
    int it = 0;
    const bool ok0 = cur < __builtin_inf() && cur == cur;
    while (ok0 && it < cfg.max_iterations) {
        double acc[NV];
        build_schur<F>(P, cfg, R0, t0, R, t, pts, lam, acc, red);
        double S[NL], b[NC];
#pragma unroll
        for (int k = 0; k < NL; ++k)
            S[k] = acc[k];
#pragma unroll
        for (int k = 0; k < NC; ++k)
            b[k] = acc[NL + k];
        bool accepted = false, solved = false;
        double cand = 0.0;
        double Rn[F][9], tn[F][3];
        if (chol_packed<NC>(S)) {   // uniform: every thread holds the same S
            solved = true;
            chol_solve<NC>(S, b);
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const double dw[3] = {b[6 * f], b[6 * f + 1], b[6 * f + 2]};
                double E[9];
                so3_exp(dw, E);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        Rn[f][3 * r + c] = (R[f][3 * r] * E[c] + R[f][3 * r + 1] * E[3 + c]) + R[f][3 * r + 2] * E[6 + c];
                    tn[f][r] = t[f][r] + ((R[f][3 * r] * b[6 * f + 3] + R[f][3 * r + 1] * b[6 * f + 4]) +
                                          R[f][3 * r + 2] * b[6 * f + 5]);
                }
            }
            double c = 0.0;
            for_points<F>(P, pts, [&](int i, const PtIn<F> &q) {
                const double (&p)[3] = q.p;
                double Hpp[6], gp[3], Hcp[NC][3], dummy[NV];
                point_linearize<F, false>(P, R, t, p, q, Hpp, gp, Hcp, dummy);
                const double Hd[6] = {Hpp[0] + lam, Hpp[1], Hpp[2], Hpp[3] + lam, Hpp[4], Hpp[5] + lam};
                double Pi[6];
                sym3_inverse(Hd, Pi);
                double v0 = gp[0], v1 = gp[1], v2 = gp[2];
#pragma unroll
                for (int a = 0; a < NC; ++a) {
                    v0 = fma(Hcp[a][0], b[a], v0);
                    v1 = fma(Hcp[a][1], b[a], v1);
                    v2 = fma(Hcp[a][2], b[a], v2);
                }
                const double pn[3] = {p[0] - fd3(Pi[0], v0, Pi[1], v1, Pi[2], v2),
                                      p[1] - fd3(Pi[1], v0, Pi[3], v1, Pi[4], v2),
                                      p[2] - fd3(Pi[2], v0, Pi[4], v1, Pi[5], v2)};
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    pts_new[3 * (size_t)i + k] = pn[k];
                if (i < P.n_res) {
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        res_p[((P.cur ^ 1) * 3 + k) * Res<F>::CAP + i] = pn[k];
                }
                c = c + point_cost<F>(P, Rn, tn, pn, q);
            });
            cand = block_reduce1(c, red) + prior_cost<F>(cfg, R0, t0, Rn, tn);
            accepted = cand <= cur;
        }
        ++it;
        if (accepted) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    R[f][k] = uniform_f64(Rn[f][k]);
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    t[f][k] = uniform_f64(tn[f][k]);
            }
            double *sw = pts;
            pts = pts_new;
            pts_new = sw;
            P.cur ^= 1;
            const double dec = 0.5 * (cur - cand);
            const bool done = dec <= cfg.abs_tol || dec <= cfg.rel_tol * (0.5 * cur);
            cur = cand;
            lam = lam / cfg.lambda_factor;
            if (done)
                break;
        } else {
            // a trial within the tolerances ABOVE the current error: at the minimum to rounding, stop
            const double inc = 0.5 * (cand - cur);
            if (solved && (inc <= cfg.abs_tol || inc <= cfg.rel_tol * (0.5 * cur)))
                break;
            lam = lam * cfg.lambda_factor;
            if (lam > cfg.lambda_upper)
                break;
        }
    }

    // marginal covariances at the estimate (lambda = 0)
    bool ok = ok0;
    if (ok) {
        double acc[NV];
        build_schur<F>(P, cfg, R0, t0, R, t, pts, 0.0, acc, red);
        double S[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k)
            S[k] = acc[k];
        ok = chol_packed<NC>(S);
        if (ok) {
            if (threadIdx.x < NC) {   // thread a solves for column a of S^-1
                double e[NC];
#pragma unroll
                for (int k = 0; k < NC; ++k)
                    e[k] = (k == (int)threadIdx.x) ? 1.0 : 0.0;
                chol_solve<NC>(S, e);
#pragma unroll
                for (int k = 0; k < NC; ++k)
                    Sinv[k * NC + threadIdx.x] = e[k];
            }
            __syncthreads();
            if (d.point_cov) {
                double *pc = d.point_cov + 9 * base;
                for_points<F>(P, pts, [&](int i, const PtIn<F> &q) {
                    const double (&p)[3] = q.p;
                    double Hpp[6], gp[3], Hcp[NC][3], dummy[NV];
                    point_linearize<F, false>(P, R, t, p, q, Hpp, gp, Hcp, dummy);
                    double Pi[6];
                    sym3_inverse(Hpp, Pi);
                    const double Pf[9] = {Pi[0], Pi[1], Pi[2], Pi[1], Pi[3], Pi[4], Pi[2], Pi[4], Pi[5]};
                    double G[NC][3];
#pragma unroll
                    for (int a = 0; a < NC; ++a)
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            G[a][k] = fd3(Hcp[a][0], Pf[k], Hcp[a][1], Pf[3 + k], Hcp[a][2], Pf[6 + k]);
                    double C[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k)
                        C[k] = 0.0;
#pragma unroll
                    for (int a = 0; a < NC; ++a) {
                        double sg[3];   // row a of S^-1 G
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            double s = 0.0;
#pragma unroll
                            for (int c = 0; c < NC; ++c)
                                s = fma(Sinv[a * NC + c], G[c][k], s);
                            sg[k] = s;
                        }
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int k = 0; k < 3; ++k)
                                C[3 * r + k] = fma(G[a][r], sg[k], C[3 * r + k]);
                    }
#pragma unroll
                    for (int k = 0; k < 9; ++k)
                        pc[9 * (size_t)i + k] = Pf[k] + C[k];
                });
            }
        }
    }

    // the estimate ends in d.pts whichever buffer the last accepted step wrote
    double *dst = d.pts + 3 * base;
    if (pts != dst) {
        for (int i = threadIdx.x; i < P.m; i += kRefineThreads) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                dst[3 * (size_t)i + k] = pts[3 * (size_t)i + k];
        }
    }
    if (threadIdx.x == 0) {
        out->ok = ok ? 1 : 0;
        out->iterations = it;
        out->error = 0.5 * cur;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            out->R[k] = R[F - 1][k];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            out->t[k] = t[F - 1][k];
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c)
                out->pose_cov[6 * r + c] = ok ? Sinv[(6 * (F - 1) + r) * NC + (6 * (F - 1) + c)] : 0.0;
        if (d.out_all) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
                mvs_refine_result *o = d.out_all + (size_t)g * F + f;
                o->ok = ok ? 1 : 0;
                o->iterations = it;
                o->error = 0.5 * cur;
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    o->R[k] = R[f][k];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    o->t[k] = t[f][k];
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 6; ++c)
                        o->pose_cov[6 * r + c] = ok ? Sinv[(6 * f + r) * NC + (6 * f + c)] : 0.0;
            }
        }
    }
}

// covariance -> information, one thread per point slot
// valid0 / valid1: optional per-observation flags (0 = frame f does not see the point: zero information);
// a 3x3 covariance whose first entry is <= 0 means "no prior on this point" (zero information)
__global__ void refine_prep_kernel(RefineDev d, const double *cov2_0, const double *cov2_1, const double *cov3, double iso3,
                                   double *oinfo0, double *oinfo1, double *pinfo, const uint8_t *valid0,
                                   const uint8_t *valid1)
{
    const int g = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.m[g] || i >= d.stride)
        return;
    const size_t s = (size_t)g * d.stride + i;
    for (int f = 0; f < d.n_frames; ++f) {
        const double *cv = f ? cov2_1 : cov2_0;
        const uint8_t *vd = f ? valid1 : valid0;
        double *o = (f ? oinfo1 : oinfo0) + 3 * s;
        if (vd && !vd[s]) {
            o[0] = 0.0, o[1] = 0.0, o[2] = 0.0;
        } else if (!cv) {
            o[0] = 1.0, o[1] = 0.0, o[2] = 1.0;
        } else {
            const double a = cv[4 * s], b = 0.5 * (cv[4 * s + 1] + cv[4 * s + 2]), dd = cv[4 * s + 3];
            const double id = 1.0 / (a * dd - b * b);
            o[0] = dd * id, o[1] = -(b * id), o[2] = a * id;
        }
    }
    double *L = pinfo + 6 * s;
    if (!cov3) {
        L[0] = L[3] = L[5] = iso3;
        L[1] = L[2] = L[4] = 0.0;
    } else {
        const double *C = cov3 + 9 * s;
        const double a[6] = {C[0], 0.5 * (C[1] + C[3]), 0.5 * (C[2] + C[6]), C[4], 0.5 * (C[5] + C[7]), C[8]};
        double o[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (C[0] > 0.0)
            sym3_inverse(a, o);
#pragma unroll
        for (int k = 0; k < 6; ++k)
            L[k] = o[k];
    }
}

// ImagePair::refine inputs (front-end/image-pair.cpp:176-209) of every pair, built from the batch's own results:
// point j of pair p -> match point_idx[j] -> keypoints (trainIdx in the base frame, queryIdx in the pair frame).
// Observation covariance = VisualFeature::get_point_estimates (vision/visual-feature.cpp:192-207):
// stddev = 2^octave * 0.5 px, i.e. information w_obs / 4^octave with w_obs = 1 / sigma_px^2 (exact scaling).
__global__ void refine_gather_kernel(BatchDev b, double w_obs, double w_pt, int stride, int32_t *m, double *pose0,
                                     double *obs0, double *obs1, double *oinfo0, double *oinfo1, double *pts0, double *pinfo)
{
    const int p = blockIdx.y;
    const mvs_pair_result &res = b.results[p];
    const int n = (res.valid && res.n_points <= stride) ? res.n_points : 0;
    if (blockIdx.x == 0 && threadIdx.x < 12)
        pose0[12 * (size_t)p + threadIdx.x] = threadIdx.x < 9 ? res.R[threadIdx.x] : res.t[threadIdx.x - 9];
    if (blockIdx.x == 0 && threadIdx.x == 0)
        m[p] = n;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    const size_t s = (size_t)p * stride + j, sb = (size_t)p * b.max_kp;
    const mvs_match mt = b.matches[sb + b.point_idx[sb + j]];
    obs0[2 * s] = (double)b.kp1[2 * (sb + mt.trainIdx)];
    obs0[2 * s + 1] = (double)b.kp1[2 * (sb + mt.trainIdx) + 1];
    obs1[2 * s] = (double)b.kp2[2 * (sb + mt.queryIdx)];
    obs1[2 * s + 1] = (double)b.kp2[2 * (sb + mt.queryIdx) + 1];
    const double w0 = ldexp(w_obs, -2 * (int)b.oct1[sb + mt.trainIdx]);
    const double w1 = ldexp(w_obs, -2 * (int)b.oct2[sb + mt.queryIdx]);
    oinfo0[3 * s] = w0, oinfo0[3 * s + 1] = 0.0, oinfo0[3 * s + 2] = w0;
    oinfo1[3 * s] = w1, oinfo1[3 * s + 1] = 0.0, oinfo1[3 * s + 2] = w1;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        pts0[3 * s + k] = b.points[3 * (sb + j) + k];
    pinfo[6 * s] = pinfo[6 * s + 3] = pinfo[6 * s + 5] = w_pt;
    pinfo[6 * s + 1] = pinfo[6 * s + 2] = pinfo[6 * s + 4] = 0.0;
}

// pnp_solve's refit over the inliers (cv::solvePnPRansac ends with one, pnp-solve.cpp:53-64), batched over the tracks of a
// sequence: problem q = the inliers of track q in inlier order, world points held by a stiff prior (information w_pt),
// unweighted pixels, guess = the RANSAC pose.  Same inputs in the same order as mvs_pnp_solve(refit = 1) stages on the host.
__global__ void pnp_refit_gather_kernel(PnpDev p, double w_pt, int32_t *m, double *pose0, double *obs0, double *oinfo0,
                                        double *pts0, double *pinfo)
{
    const int q = blockIdx.y;
    const PnpOut &o = p.out[q];
    const int n = (o.ok && o.n_inliers >= 4) ? o.n_inliers : 0;
    if (blockIdx.x == 0 && threadIdx.x < 12)
        pose0[12 * (size_t)q + threadIdx.x] = threadIdx.x < 9 ? o.R[threadIdx.x] : o.t[threadIdx.x - 9];
    if (blockIdx.x == 0 && threadIdx.x == 0)
        m[q] = n;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    const size_t base = (size_t)q * p.stride, s = base + j;
    const size_t src = base + p.inliers[s];
    obs0[2 * s] = p.uv[2 * src];
    obs0[2 * s + 1] = p.uv[2 * src + 1];
    oinfo0[3 * s] = 1.0, oinfo0[3 * s + 1] = 0.0, oinfo0[3 * s + 2] = 1.0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        pts0[3 * s + k] = p.X[3 * src + k];
    pinfo[6 * s] = pinfo[6 * s + 3] = pinfo[6 * s + 5] = w_pt;
    pinfo[6 * s + 1] = pinfo[6 * s + 2] = pinfo[6 * s + 4] = 0.0;
}

// a converged refit replaces the track's pose (camera in world, and its inverse world -> camera)
__global__ void pnp_refit_apply_kernel(PnpDev p, const mvs_refine_result *rr)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= p.n_problems)
        return;
    PnpOut &o = p.out[q];
    const mvs_refine_result &r = rr[q];
    if (!o.ok || o.n_inliers < 4 || !r.ok)
        return;
#pragma unroll
    for (int k = 0; k < 9; ++k)
        o.R[k] = r.R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        o.t[k] = r.t[k];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            o.Rw2c[3 * i + j] = r.R[3 * j + i];
        o.tw2c[i] = -((r.R[i] * r.t[0] + r.R[3 + i] * r.t[1]) + r.R[6 + i] * r.t[2]);
    }
}

}  // namespace

void launch_pnp_refit(const PnpDev &p, const RefineDev &d, double point_sigma, hipStream_t stream)
{
    if (p.n_problems <= 0)
        return;
    dim3 grid((p.stride + 255) / 256, p.n_problems);
    hipLaunchKernelGGL(pnp_refit_gather_kernel, grid, dim3(256), 0, stream, p, 1.0 / (point_sigma * point_sigma),
                       const_cast<int32_t *>(d.m), const_cast<double *>(d.pose0), const_cast<double *>(d.obs[0]),
                       const_cast<double *>(d.oinfo[0]), const_cast<double *>(d.pts0), const_cast<double *>(d.pinfo));
    launch_refine(d, stream);
    hipLaunchKernelGGL(pnp_refit_apply_kernel, dim3((p.n_problems + 255) / 256), dim3(256), 0, stream, p, d.out);
}

void launch_refine_prep(const RefineDev &d, const double *cov2_0, const double *cov2_1, const double *cov3, double iso3,
                        double *oinfo0, double *oinfo1, double *pinfo, const uint8_t *valid0, const uint8_t *valid1,
                        hipStream_t stream)
{
    if (d.n_problems <= 0)
        return;
    dim3 grid((d.stride + 255) / 256, d.n_problems);
    hipLaunchKernelGGL(refine_prep_kernel, grid, dim3(256), 0, stream, d, cov2_0, cov2_1, cov3, iso3, oinfo0, oinfo1, pinfo,
                       valid0, valid1);
}

void launch_refine(const RefineDev &d, hipStream_t stream)
{
    if (d.n_problems <= 0)
        return;
    if (d.n_frames == 2)
        hipLaunchKernelGGL(refine_kernel<2>, dim3(d.n_problems), dim3(kRefineThreads), 0, stream, d);
    else
        hipLaunchKernelGGL(refine_kernel<1>, dim3(d.n_problems), dim3(kRefineThreads), 0, stream, d);
}

void launch_refine_gather(const BatchDev &b, int n_active, double sigma_px, double point_sigma, int stride, int32_t *m,
                          double *pose0, double *obs0, double *obs1, double *oinfo0, double *oinfo1, double *pts0,
                          double *pinfo, hipStream_t stream)
{
    if (n_active <= 0)
        return;
    dim3 grid((stride + 255) / 256, n_active);
    hipLaunchKernelGGL(refine_gather_kernel, grid, dim3(256), 0, stream, b, 1.0 / (sigma_px * sigma_px),
                       1.0 / (point_sigma * point_sigma), stride, m, pose0, obs0, obs1, oinfo0, oinfo1, pts0, pinfo);
}

}  // namespace mvs
