// pnp.hip -- pnp_solve (vision/pnp-solve.cpp:16-104) as a batched P3P-RANSAC.
//
// The reference forwards to cv::solvePnPRansac(SOLVEPNP_P3P, 100 iterations, reprojection error 0.05, confidence
// 0.95) and inverts the pose.  OpenCV's RANSAC kernel, RNG and final EPnP refit are third-party and differ between
// the 3.x versions the reference admits, so this is the build's own algorithm, written with
// + - * / sqrt only so that it matches the CPU oracle bit for bit:
//   pnp_prep      K^-1 (u, v, 1) and unit bearings                                   thread per point
//   pnp_ransac    one hypothesis per LANE: sample 4 -> Grunert P3P on 3 (quartic by Ferrari, resolvent cubic by
//                 64 bisections) -> pick among <= 4 solutions with the 4th point -> division-free reprojection
//                 test on all n points (point stream staged in LDS, broadcast reads) -> workgroup arg-best
//   pnp_finalize  arg-best over workgroups (most inliers, then first hypothesis: cv::RANSACPointSetRegistrator's
//                 rule), ordered inlier list, pose = SE3(SO3(R), t).inverse()          (pnp-solve.cpp:99-101)
#include "kernels.hpp"

#include "device_math.hpp"

namespace mvs {

// 4 distinct indices in [0, n): Philox block 2 (blocks 0 / 1 belong to the 8-of-M sampler), same draw rule
__device__ __forceinline__ void sample4(uint64_t seed, uint32_t hyp, int n, int sampler, int (&idx)[4])
{
    if (sampler == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            idx[k] = k;
        return;
    }
    uint32_t w[4];
    philox4x32_10(hyp, 2u, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    int sorted[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        sorted[k] = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t r = __umulhi(w[k], (uint32_t)(n - k));
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < k && r >= (uint32_t)sorted[t])
                ++r;
        idx[k] = (int)r;
        int carry = (int)r;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t <= k) {
                const int cur = sorted[t];
                const bool sw = carry < cur;
                sorted[t] = sw ? carry : cur;
                carry = sw ? cur : carry;
            }
        }
    }
}

// real roots of x^4 + b x^3 + c x^2 + d x + e in fixed slots (valid flags), order: factor sg=+1 (larger, smaller),
// then sg=-1 (larger, smaller); biquadratic case: +-sqrt(y2a), +-sqrt(y2b)
__device__ __forceinline__ void quartic_real_roots(double b, double c, double d, double e, double (&roots)[4],
                                                   bool (&valid)[4])
{
    const double p = c - 3.0 * b * b / 8.0;
    const double q = d - b * c / 2.0 + b * b * b / 8.0;
    const double r = e - b * d / 4.0 + b * b * c / 16.0 - 3.0 * b * b * b * b / 256.0;
    const double sh = b / 4.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        roots[k] = 0.0;
        valid[k] = false;
    }
    if (q == 0.0) {
        const double disc = p * p - 4.0 * r;
        if (disc >= 0.0) {
            const double sd = dsqrt(disc);
            const double y2a = (-p + sd) / 2.0, y2b = (-p - sd) / 2.0;
            if (y2a >= 0.0) {
                const double y = dsqrt(y2a);
                roots[0] = y - sh;
                roots[1] = -y - sh;
                valid[0] = valid[1] = true;
            }
            if (y2b >= 0.0) {
                const double y = dsqrt(y2b);
                roots[2] = y - sh;
                roots[3] = -y - sh;
                valid[2] = valid[3] = true;
            }
        }
        return;
    }
    const double c1 = 2.0 * p * p - 8.0 * r, c0 = q * q;
    double hi = dabs(p);
    const double h1 = dabs(c1 / 8.0), h2 = dabs(c0 / 8.0);
    if (h1 > hi) hi = h1;
    if (h2 > hi) hi = h2;
    hi = hi + 1.0;
    double lo = 0.0;
    for (int it = 0; it < 64; ++it) {
        const double mid = 0.5 * (lo + hi);
        const double fm = ((8.0 * mid + 8.0 * p) * mid + c1) * mid - c0;
        const bool pos = fm > 0.0;
        hi = pos ? mid : hi;
        lo = pos ? lo : mid;
    }
    const double m = 0.5 * (lo + hi);
    const double s = dsqrt(2.0 * m);
    const double t = q / (2.0 * s);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double sg = k == 0 ? 1.0 : -1.0;
        const double cc = p / 2.0 + m + sg * t;
        const double disc = s * s - 4.0 * cc;
        if (disc >= 0.0) {
            const double sd = dsqrt(disc);
            roots[2 * k] = (sg * s + sd) / 2.0 - sh;
            roots[2 * k + 1] = (sg * s - sd) / 2.0 - sh;
            valid[2 * k] = valid[2 * k + 1] = true;
        }
    }
}

__device__ __forceinline__ double dot3r(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

// orthonormal frame of a triangle P (3 points, row-major), columns e1, e2, e3
__device__ __forceinline__ void tri_frame(const double (&P)[9], double (&Fm)[9])
{
    double e1[3] = {P[3] - P[0], P[4] - P[1], P[5] - P[2]};
    const double n1 = dsqrt((e1[0] * e1[0] + e1[1] * e1[1]) + e1[2] * e1[2]);
    e1[0] = e1[0] / n1; e1[1] = e1[1] / n1; e1[2] = e1[2] / n1;
    const double d[3] = {P[6] - P[0], P[7] - P[1], P[8] - P[2]};
    double e3[3] = {e1[1] * d[2] - e1[2] * d[1], e1[2] * d[0] - e1[0] * d[2], e1[0] * d[1] - e1[1] * d[0]};
    const double n3 = dsqrt((e3[0] * e3[0] + e3[1] * e3[1]) + e3[2] * e3[2]);
    e3[0] = e3[0] / n3; e3[1] = e3[1] / n3; e3[2] = e3[2] / n3;
    const double e2[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Fm[k * 3 + 0] = e1[k];
        Fm[k * 3 + 1] = e2[k];
        Fm[k * 3 + 2] = e3[k];
    }
}

// division-free reprojection test (squared pixel error times zc^2 against err^2 zc^2, and zc > 0)
__device__ __forceinline__ bool pnp_inlier(const double (&R)[9], const double (&t)[3], double X0, double X1, double X2,
                                           double xi, double yi, double fx2, double fy2, double thr2, double &lhs,
                                           double &rhs)
{
    const double xc = dfma(R[0], X0, dfma(R[1], X1, dfma(R[2], X2, t[0])));
    const double yc = dfma(R[3], X0, dfma(R[4], X1, dfma(R[5], X2, t[1])));
    const double zc = dfma(R[6], X0, dfma(R[7], X1, dfma(R[8], X2, t[2])));
    const double dx = dfma(-xi, zc, xc), dy = dfma(-yi, zc, yc);
    lhs = dfma(fy2, dy * dy, fx2 * (dx * dx));
    rhs = thr2 * (zc * zc);
    return (zc > 0.0) && (lhs <= rhs);
}

// Grunert's P3P on 3 bearings / world points + disambiguation by a 4th correspondence.  returns false if no solution.
__device__ __forceinline__ bool p3p_select(const double (&f)[9], const double (&X)[9], const double (&X4)[3], double x4,
                                           double y4, double fx2, double fy2, double thr2, double (&Rb)[9],
                                           double (&tb)[3])
{
    const double d12[3] = {X[0] - X[3], X[1] - X[4], X[2] - X[5]};
    const double d13[3] = {X[0] - X[6], X[1] - X[7], X[2] - X[8]};
    const double d23[3] = {X[3] - X[6], X[4] - X[7], X[5] - X[8]};
    const double a2 = dot3r(d23, d23), b2 = dot3r(d13, d13), c2 = dot3r(d12, d12);
    const double ca = dot3r(&f[3], &f[6]), cb = dot3r(&f[0], &f[6]), cg = dot3r(&f[0], &f[3]);
    const double k1 = (a2 - c2) / b2, k2 = (a2 + c2) / b2, k3 = (b2 - c2) / b2, k4 = (b2 - a2) / b2;
    const double A4 = (k1 - 1.0) * (k1 - 1.0) - 4.0 * c2 / b2 * ca * ca;
    const double A3 = 4.0 * (k1 * (1.0 - k1) * cb - (1.0 - k2) * ca * cg + 2.0 * c2 / b2 * ca * ca * cb);
    const double A2 = 2.0 * (k1 * k1 - 1.0 + 2.0 * k1 * k1 * cb * cb + 2.0 * k3 * ca * ca - 4.0 * k2 * ca * cb * cg +
                             2.0 * k4 * cg * cg);
    const double A1 = 4.0 * (-k1 * (1.0 + k1) * cb + 2.0 * a2 / b2 * cg * cg * cb - (1.0 - k2) * ca * cg);
    const double A0 = (1.0 + k1) * (1.0 + k1) - 4.0 * a2 / b2 * cg * cg;
    double roots[4];
    bool valid[4];
    quartic_real_roots(A3 / A4, A2 / A4, A1 / A4, A0 / A4, roots, valid);
    double Fw[9];
    tri_frame(X, Fw);
    bool have = false;
    double sel_lhs = 0.0, sel_rhs = 1.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double v = roots[k];
        bool ok = valid[k] && (v > 0.0);
        const double u = ((-1.0 + k1) * v * v - 2.0 * k1 * cb * v + 1.0 + k1) / (2.0 * (cg - v * ca));
        ok = ok && (u > 0.0);
        if (ok) {  // divergent, but each branch is ~150 instructions and runs for most lanes of at most 2-4 slots
            const double s1 = dsqrt(c2 / (1.0 + u * u - 2.0 * u * cg));
            const double s2 = u * s1, s3 = v * s1;
            const double Pc[9] = {s1 * f[0], s1 * f[1], s1 * f[2], s2 * f[3], s2 * f[4], s2 * f[5],
                                  s3 * f[6], s3 * f[7], s3 * f[8]};
            double Fc[9], Rk[9], tk[3];
            tri_frame(Pc, Fc);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    Rk[i * 3 + j] = (Fc[i * 3 + 0] * Fw[j * 3 + 0] + Fc[i * 3 + 1] * Fw[j * 3 + 1]) + Fc[i * 3 + 2] * Fw[j * 3 + 2];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                tk[i] = Pc[i] - dot3r(&Rk[3 * i], &X[0]);
            double lhs, rhs;
            pnp_inlier(Rk, tk, X4[0], X4[1], X4[2], x4, y4, fx2, fy2, thr2, lhs, rhs);
            // smallest squared pixel error on the 4th point, compared without dividing (rhs = err^2 zc^2 > 0)
            if ((rhs > 0.0) && (!have || lhs * sel_rhs < sel_lhs * rhs)) {
                have = true;
                sel_lhs = lhs;
                sel_rhs = rhs;
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    Rb[i] = Rk[i];
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    tb[i] = tk[i];
            }
        }
    }
    return have;
}

// grid (ceil(stride/256), Q): ideal-camera coordinates and unit bearings
__global__ __launch_bounds__(256) void pnp_prep_kernel(PnpDev p)
{
    const int q = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n[q])
        return;
    const size_t o = (size_t)q * p.stride + i;
    const double *Ki = p.Kinv + (size_t)q * 9;
    const double u = p.uv[2 * o], v = p.uv[2 * o + 1];
    const double x = (Ki[0] * u + Ki[1] * v) + Ki[2];
    const double y = (Ki[3] * u + Ki[4] * v) + Ki[5];
    const double nn = dsqrt((x * x + y * y) + 1.0);
    p.xy[2 * o] = x;
    p.xy[2 * o + 1] = y;
    p.fb[3 * o] = x / nn;
    p.fb[3 * o + 1] = y / nn;
    p.fb[3 * o + 2] = 1.0 / nn;
}

constexpr int kPnpChunk = 768;
// grid (ceil(H/256), Q), block 256.
// Round 5: (1) the point stream goes through LDS in chunks of 768 points (36 KB: four workgroups per CU) -- the 2048-point
// array of rounds 2-4 (96 KB) left ONE workgroup per CU, four rounds of workgroups for the sequence's 998 tracks of ~570
// points; (2) a block with fewer than 129 live
// hypotheses -- the reference's iterationsCount is 100 -- has idle wavefronts: they take the same hypotheses and a share of the
// POINTS (wavefront w: hypothesis group w mod groups, points part, part + split, ...), and the integer counts are added up
// in LDS.  Every hypothesis is still solved and scored by the same operations on the same numbers: same counts, same winner.
__global__ __launch_bounds__(256) void pnp_ransac_kernel(PnpDev p)
{
    __shared__ __attribute__((aligned(16))) double s_pts[kPnpChunk * 6];  // a chunk of the point stream: X0 X1 X2 x y pad
    __shared__ int s_part[4][64];
    __shared__ int s_cnt[4];
    __shared__ uint32_t s_hyp[4];
    __shared__ uint32_t s_win;
    const int tid = threadIdx.x, q = blockIdx.y, n = p.n[q];
    PnpRec *rec = p.rec + (size_t)q * p.max_groups + blockIdx.x;
    if (n < 7) {  // pnp-solve.cpp:13,22 (the reference asserts)
        if (tid == 0) {
            rec->count = -1;
            rec->hyp = 0xffffffffu;
        }
        return;
    }
    const size_t o = (size_t)q * p.stride;
    const double *X = p.X + 3 * o, *xy = p.xy + 2 * o, *fb = p.fb + 3 * o;
    const double *K = p.K + (size_t)q * 9;
    const double fx2 = K[0] * K[0], fy2 = K[4] * K[4], thr2 = p.thr2;
    // hypotheses of this block: groups of 64 (one per lane); 4 / groups wavefronts share a group's points
    const int lane = tid & 63, wave = tid >> 6;
    const int in_block = min(p.num_hypotheses - (int)blockIdx.x * 256, 256);
    const int groups = (in_block + 63) >> 6, split = 4 / groups;          // 1, 2, 3, 4 groups -> 4, 2, 1, 1 parts
    const int grp = wave % groups, part = wave / groups;                   // (3 groups: wavefront 3 has part 1 >= split: idle)
    const uint32_t h = blockIdx.x * 256 + grp * 64 + lane;
    const bool live = h < (uint32_t)p.num_hypotheses && part < split;
    const uint32_t hh = h < (uint32_t)p.num_hypotheses ? h : (uint32_t)(p.num_hypotheses - 1);
    const uint64_t seed = p.seed + (p.gidx ? (uint64_t)p.gidx[q] : 0ull);
    int idx[4];
    sample4(seed, hh, n, p.sampler, idx);
    double f3[9], X3[9], X4[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f3[3 * k + c] = fb[3 * idx[k] + c];
            X3[3 * k + c] = X[3 * idx[k] + c];
        }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        X4[c] = X[3 * idx[3] + c];
    double R[9], t[3];
    const bool have = p3p_select(f3, X3, X4, xy[2 * idx[3]], xy[2 * idx[3] + 1], fx2, fy2, thr2, R, t);
    int cnt = 0;
    const bool score = part < split && __any(have);
    for (int c0 = 0; c0 < n; c0 += kPnpChunk) {   // the point stream in LDS chunks (n is the same for the whole block)
        const int cn = min(n - c0, kPnpChunk);
        __syncthreads();
        for (int i = tid; i < cn; i += 256) {
            s_pts[6 * i + 0] = X[3 * (c0 + i)];
            s_pts[6 * i + 1] = X[3 * (c0 + i) + 1];
            s_pts[6 * i + 2] = X[3 * (c0 + i) + 2];
            s_pts[6 * i + 3] = xy[2 * (c0 + i)];
            s_pts[6 * i + 4] = xy[2 * (c0 + i) + 1];
            s_pts[6 * i + 5] = 0.0;
        }
        __syncthreads();
        if (score) {
#pragma unroll 4
            for (int i = part; i < cn; i += split) {
                const double *pt = &s_pts[6 * i];  // wave-uniform address: LDS broadcast
                double lhs, rhs;
                cnt += pnp_inlier(R, t, pt[0], pt[1], pt[2], pt[3], pt[4], fx2, fy2, thr2, lhs, rhs) ? 1 : 0;
            }
        }
    }
    s_part[wave][lane] = cnt;
    __syncthreads();
    if (part == 0) {
        cnt = 0;
        for (int k = 0; k < split; ++k)
            cnt += s_part[grp + k * groups][lane];
    }
    if (!have || !live || part != 0)
        cnt = -1;
    // workgroup arg-best: most inliers, then the smaller hypothesis id (first maximum of the sequential loop)
    int bc = cnt;
    uint32_t bh = h;
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        const int oc = __shfl_xor(bc, o2);
        const uint32_t oh = __shfl_xor(bh, o2);
        if (oc > bc || (oc == bc && oh < bh)) {
            bc = oc;
            bh = oh;
        }
    }
    if (lane == 0) {
        s_cnt[wave] = bc;
        s_hyp[wave] = bh;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (s_cnt[w] > bc || (s_cnt[w] == bc && s_hyp[w] < bh)) {
                bc = s_cnt[w];
                bh = s_hyp[w];
            }
        s_win = bh;
        rec->count = bc;
        rec->hyp = bh;
    }
    __syncthreads();
    if (h == s_win && part == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i)
            rec->R[i] = R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            rec->t[i] = t[i];
    }
}

// grid Q, block 256
__global__ __launch_bounds__(256) void pnp_finalize_kernel(PnpDev p)
{
    __shared__ int s_tot[4];
    __shared__ double s_R[9], s_t[3];
    __shared__ int s_ok;
    const int tid = threadIdx.x, q = blockIdx.x, n = p.n[q];
    const int G = (p.num_hypotheses + 255) / 256;
    const PnpRec *rec = p.rec + (size_t)q * p.max_groups;
    PnpOut *out = p.out + q;
    if (tid == 0) {
        int bc = -1, bg = -1;
        uint32_t bh = 0xffffffffu;
        if (n >= 7) {
            for (int g = 0; g < G; ++g) {
                const int c = rec[g].count;
                const uint32_t h = rec[g].hyp;
                if (c >= 0 && (c > bc || (c == bc && h < bh))) {
                    bc = c;
                    bh = h;
                    bg = g;
                }
            }
        }
        const bool ok = bg >= 0 && bc >= p.min_inliers;
        s_ok = ok ? 1 : 0;
        out->ok = ok ? 1 : 0;
        out->best_hyp = bg >= 0 ? (int)bh : -1;
        out->n_inliers = 0;
        if (ok) {
#pragma unroll
            for (int i = 0; i < 9; ++i)
                s_R[i] = rec[bg].R[i];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                s_t[i] = rec[bg].t[i];
        }
    }
    __syncthreads();
    if (!s_ok) {
        // rows of the inlier list behind n_inliers are cleared on every run (a download of the whole capacity is deterministic)
        int32_t *z = p.inliers + (size_t)q * p.stride;
        for (int i = tid; i < p.stride; i += 256)
            z[i] = 0;
        return;
    }
    double R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i)
        R[i] = s_R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        t[i] = s_t[i];
    const size_t o = (size_t)q * p.stride;
    const double *X = p.X + 3 * o, *xy = p.xy + 2 * o;
    const double *K = p.K + (size_t)q * 9;
    const double fx2 = K[0] * K[0], fy2 = K[4] * K[4];
    int32_t *inl = p.inliers + o;
    // ordered inlier list
    const int lane = tid & 63, w = tid >> 6;
    int basepos = 0;
    for (int start = 0; start < n; start += 256) {
        const int i = start + tid;
        bool flag = false;
        if (i < n) {
            double lhs, rhs;
            flag = pnp_inlier(R, t, X[3 * i], X[3 * i + 1], X[3 * i + 2], xy[2 * i], xy[2 * i + 1], fx2, fy2, p.thr2, lhs, rhs);
        }
        const unsigned long long bal = __ballot(flag);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0)
            s_tot[w] = __popcll(bal);
        __syncthreads();
        int off = basepos, tot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = s_tot[k];
            off += (k < w) ? v : 0;
            tot += v;
        }
        if (flag)
            inl[off + pre] = i;
        basepos += tot;
        __syncthreads();
    }
    for (int i = basepos + tid; i < p.stride; i += 256)
        inl[i] = 0;
    if (tid == 0) {
        out->n_inliers = basepos;
        // pose = SE3(SO3(R), t).inverse() (pnp-solve.cpp:99-101; lie-group.hpp:31-36,212-216)
        double Rr[3][3], RT[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                Rr[i][j] = R[i * 3 + j];
                out->Rw2c[i * 3 + j] = R[i * 3 + j];
            }
        rectify3(Rr);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                RT[i][j] = Rr[j][i];
        rectify3(RT);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            out->tw2c[i] = t[i];
            out->t[i] = -((RT[i][0] * t[0] + RT[i][1] * t[1]) + RT[i][2] * t[2]);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                out->R[i * 3 + j] = RT[i][j];
        }
    }
}

// grid n_tracks, block 256.  Track q: pair a = q (frames q, q+1), pair b = q + 1 (frames q+1, q+2).
// point j of pair a belongs to match m = point_idx[a][j], whose queryIdx is a keypoint index of frame q+1; every
// match m' of pair b whose trainIdx is that keypoint observes the same point in frame q+2 (queryIdx of m').
// Correspondences are emitted in the order of pair b's match list (visual-odometer.cpp:528-556 restated as a join).
__global__ __launch_bounds__(256) void seq_join_kernel(SeqJoinDev j)
{
    __shared__ int16_t s_tbl[kMaxKp];
    __shared__ int s_tot[4];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int a = q, b = q + 1;
    const size_t N = j.max_kp;
    const bool valid = j.results[a].valid != 0;
    const int npts = valid ? j.results[a].n_points : 0;
    const int Mb = min(j.M[b], j.max_kp);
    for (int i = tid; i < j.max_kp; i += 256)
        s_tbl[i] = -1;
    __syncthreads();
    for (int p = tid; p < npts; p += 256) {
        const int m = j.point_idx[a * N + p];
        s_tbl[j.matches[a * N + m].queryIdx] = (int16_t)p;  // queryIdx values are unique within one match list
    }
    __syncthreads();
    const float *kp2 = j.kp + (size_t)(q + 2) * N * 2;
    double *X = j.X + (size_t)q * j.stride * 3, *uv = j.uv + (size_t)q * j.stride * 2;
    int basepos = 0;
    for (int start = 0; start < Mb; start += 256) {
        const int m = start + tid;
        int p = -1, qi = 0;
        if (m < Mb) {
            const mvs_match mt = j.matches[b * N + m];
            p = s_tbl[mt.trainIdx];
            qi = mt.queryIdx;
        }
        const bool flag = p >= 0;
        const unsigned long long bal = __ballot(flag);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0)
            s_tot[w] = __popcll(bal);
        __syncthreads();
        int off = basepos, tot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = s_tot[k];
            off += (k < w) ? v : 0;
            tot += v;
        }
        const int pos = off + pre;
        if (flag && pos < j.stride) {
            const double *src = j.points + ((size_t)a * N + p) * 3;
            X[3 * pos] = src[0];
            X[3 * pos + 1] = src[1];
            X[3 * pos + 2] = src[2];
            uv[2 * pos] = (double)kp2[2 * qi];       // visual-feature.cpp:179-190 float -> double
            uv[2 * pos + 1] = (double)kp2[2 * qi + 1];
        }
        basepos += tot;
        __syncthreads();
    }
    if (tid == 0)
        j.n_corr[q] = min(basepos, j.stride);
}

void launch_seq_join(const SeqJoinDev &j, hipStream_t stream)
{
    hipLaunchKernelGGL(seq_join_kernel, dim3(j.n_tracks), dim3(256), 0, stream, j);
}

void launch_pnp(const PnpDev &p, hipStream_t stream)
{
    const int Q = p.n_problems;
    hipLaunchKernelGGL(pnp_prep_kernel, dim3((p.stride + 255) / 256, Q), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(pnp_ransac_kernel, dim3((p.num_hypotheses + 255) / 256, Q), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(pnp_finalize_kernel, dim3(Q), dim3(256), 0, stream, p);
}

// ---- scale propagation and trajectory of a sequence (row f2; front-end/visual-odometer.cpp:422-445,577-588) ------------
// Pair k gives the pose of frame k+1 in frame k with a UNIT baseline; track q gives the pose of frame q+2 in frame q in pair
// q's scale.  rel_q = pair_q^-1 o track_q is the pose of frame q+2 in frame q+1 in pair q's scale, so
//   scale_q = |rel_q.t| = baseline(pair q+1) / baseline(pair q)      (visual-odometer.cpp:583-587)
//   sigma_0 = 1, sigma_{q+1} = sigma_q * scale_q                    (everything in pair 0's baseline)
//   G_0 = I, G_1 = pair_0, G_{q+2} = G_q o (R_track_q, sigma_q t_track_q)           (the PnP pose is the frame pose)
// A failed track keeps the scale (scale_q = 1) and falls back to the two-view pose: G_{q+2} = G_{q+1} o (R, sigma t) of
// pair q+1; an invalid pair contributes the identity.  A sequential fold: the order of operations WITHIN an output element is
// part of the specification (the CPU oracle repeats it) -- but the twelve elements of one composition are independent of each
// other, so (round 5) the fold runs on sixteen lanes of one wavefront instead of one: quad i holds row i of the running
// rotation in its lanes 0..2 and t_i in lane 3, the three factors A_i0, A_i1, A_i2 every element of row i needs are quad
// broadcasts (DPP quad_perm, a register move: no LDS, no scalar round trip), and each lane evaluates its own element with
// exactly the operations of the one-lane loop: (a0 c0 + a1 c1) + a2 c2 (+ a3 for the translation lanes).  The dependent
// chain per frame is ~9 instructions instead of ~100: 0.46 -> 0.05 ms per 1000 frames, every trajectory byte unchanged.
template <int M>
__device__ __forceinline__ double quad_bcast(double v)
{
    constexpr int ctrl = M | (M << 2) | (M << 4) | (M << 6);   // quad_perm:[M,M,M,M]
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256) void seq_chain_kernel(SeqChainDev c)
{
    // Only the fold itself is sequential.  Per chunk of kChunk steps the workgroup first prepares, in parallel, everything
    // that does not depend on the running state -- which operand a step composes with (track pose on G_q, or the next
    // pair's pose on G_{q+1}) and its scale ratio (the sqrt) -- into LDS; sixteen lanes then fold the chunk out of LDS with
    // the next step's operands prefetched, and the workgroup writes the chunk's results back.  (Reading HBM inside the
    // dependent loop cost 2 us per frame.)
    constexpr int kChunk = 128;
    __shared__ double s_op[kChunk * 12];    // R (9), t (3) of the step's operand, unscaled
    __shared__ double s_scale[kChunk];
    __shared__ int s_sel[kChunk];           // 0: compose on G_q, 1: on G_{q+1}
    __shared__ double s_out[kChunk * 14];   // G_{q+2} (12), track_scale, sigma_{q+1}
    __shared__ double s_g[24];              // G_0, G_1 (R 9, t 3 each)
    const int F = c.n_frames, tid = threadIdx.x;
    auto compose = [](const double *A, const double *R, const double *t, double sig, double *out) {
        // out = A o (R, sig * t):  R_out = A.R R,  t_out = A.R (sig t) + A.t
        const double s0 = sig * t[0], s1 = sig * t[1], s2 = sig * t[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                out[3 * i + j] = (A[3 * i] * R[j] + A[3 * i + 1] * R[3 + j]) + A[3 * i + 2] * R[6 + j];
            out[9 + i] = ((A[3 * i] * s0 + A[3 * i + 1] * s1) + A[3 * i + 2] * s2) + A[9 + i];
        }
    };
    if (tid == 0) {
        const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Z3[3] = {0, 0, 0};
        double Ga[12], Gb[12];
#pragma unroll
        for (int k = 0; k < 12; ++k)
            Ga[k] = (k < 9 && k % 4 == 0) ? 1.0 : 0.0;
        const mvs_pair_result &p0 = c.results[0];
        compose(Ga, p0.valid ? p0.R : I3, p0.valid ? p0.t : Z3, 1.0, Gb);
        for (int k = 0; k < 9; ++k) {
            c.traj_R[k] = Ga[k];
            c.traj_R[9 + k] = Gb[k];
        }
        for (int k = 0; k < 3; ++k) {
            c.traj_t[k] = Ga[9 + k];
            c.traj_t[3 + k] = Gb[9 + k];
        }
        c.traj_sigma[0] = 1.0;
        for (int k = 0; k < 12; ++k) {
            s_g[k] = Ga[k];
            s_g[12 + k] = Gb[k];
        }
    }
    __syncthreads();
    // fold lanes: tid = 4 i + j; j < 3: element (i, j) of the rotation, j = 3: t_i; quad 3 idles (its results are not stored)
    const int fi = min(tid >> 2, 2), fj = tid & 3;
    const bool fold_lane = tid < 12, is_t = fj == 3;
    const int elem = is_t ? 9 + fi : 3 * fi + fj;                     // this lane's element of a pose (R 9, t 3)
    const int o0 = is_t ? 9 : fj, o1 = is_t ? 10 : 3 + fj, o2 = is_t ? 11 : 6 + fj;   // its operand column
    double ga = s_g[elem], gb = s_g[12 + elem];   // this lane's element of G_q and G_{q+1}
    double sigma = 1.0;
    const int T = F - 2;
    for (int q0 = 0; q0 < T; q0 += kChunk) {
        const int n = min(kChunk, T - q0);
        __syncthreads();
        for (int j = tid; j < n; j += 256) {   // state-independent part of step q = q0 + j
            const int q = q0 + j;
            const mvs_pair_result &pq = c.results[q], &pn = c.results[q + 1];
            const PnpOut &tr = c.tracks[q];
            const bool ok = c.n_corr[q] >= 7 && tr.ok && pq.valid;
            double scale = 1.0;
            double *op = s_op + 12 * j;
            if (ok) {
                // rel.t = R_pair^T (t_track - t_pair)
                const double d0 = tr.t[0] - pq.t[0], d1 = tr.t[1] - pq.t[1], d2 = tr.t[2] - pq.t[2];
                const double r0 = (pq.R[0] * d0 + pq.R[3] * d1) + pq.R[6] * d2;
                const double r1 = (pq.R[1] * d0 + pq.R[4] * d1) + pq.R[7] * d2;
                const double r2 = (pq.R[2] * d0 + pq.R[5] * d1) + pq.R[8] * d2;
                scale = sqrt((r0 * r0 + r1 * r1) + r2 * r2);
                for (int k = 0; k < 9; ++k) op[k] = tr.R[k];
                for (int k = 0; k < 3; ++k) op[9 + k] = tr.t[k];
            } else if (pn.valid) {
                for (int k = 0; k < 9; ++k) op[k] = pn.R[k];
                for (int k = 0; k < 3; ++k) op[9 + k] = pn.t[k];
            } else {
                for (int k = 0; k < 12; ++k) op[k] = (k < 9 && k % 4 == 0) ? 1.0 : 0.0;
            }
            s_scale[j] = scale;
            s_sel[j] = ok ? 0 : 1;
        }
        __syncthreads();
        if (tid < 64) {   // the whole first wavefront walks the loop (the DPP moves need their source lanes active)
            double c0 = s_op[o0], c1 = s_op[o1], c2 = s_op[o2], scale = s_scale[0];
            int sel = s_sel[0];
            for (int j = 0; j < n; ++j) {
                const int jn = min(j + 1, n - 1);
                // prefetch the next step's operands while this step's products are in flight
                const double n0 = s_op[12 * jn + o0], n1 = s_op[12 * jn + o1], n2 = s_op[12 * jn + o2], nscale = s_scale[jn];
                const int nsel = s_sel[jn];
                const double a = sel == 0 ? ga : gb;                   // this lane's element of the left factor
                const double a0 = quad_bcast<0>(a), a1 = quad_bcast<1>(a), a2 = quad_bcast<2>(a);
                const double f = is_t ? sigma : 1.0;                    // translation lanes: the operand is sigma * t (x * 1.0 == x)
                const double r = (a0 * (f * c0) + a1 * (f * c1)) + a2 * (f * c2);
                const double gn = is_t ? r + a : r;                     // ... + A.t_i on the translation lanes
                sigma = sigma * scale;
                if (fold_lane)
                    s_out[14 * j + elem] = gn;
                if (tid == 0) {
                    s_out[14 * j + 12] = scale;
                    s_out[14 * j + 13] = sigma;
                }
                ga = gb;
                gb = gn;
                c0 = n0; c1 = n1; c2 = n2; scale = nscale; sel = nsel;
            }
        }
        __syncthreads();
        for (int i = tid; i < n * 14; i += 256) {
            const int j = i / 14, k = i - j * 14, q = q0 + j;
            const double v = s_out[i];
            if (k < 9) c.traj_R[9 * (size_t)(q + 2) + k] = v;
            else if (k < 12) c.traj_t[3 * (size_t)(q + 2) + (k - 9)] = v;
            else if (k == 12) c.track_scale[q] = v;
            else c.traj_sigma[q + 1] = v;
        }
    }
}

void launch_seq_chain(const SeqChainDev &c, hipStream_t stream)
{
    hipLaunchKernelGGL(seq_chain_kernel, dim3(1), dim3(256), 0, stream, c);
}

}  // namespace mvs
