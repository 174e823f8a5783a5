// device_math.hpp -- per-lane fp64 building blocks of the two-view kernels (gfx950).
//
// Everything here runs one problem per LANE with the whole state in registers:
// all matrix loops are fully unrolled so that every array index is a compile-time
// constant (a runtime-indexed register array would be demoted to scratch memory).
//
// Arithmetic contract: IEEE binary64, one rounding per written
// operation, fused multiply-add only where dfma() is written.  This file is
// compiled with -ffp-contract=off.  The CPU oracle implements the same contract
// independently; GPU<->oracle comparisons of integer outputs are bit-exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MVS_DEV __device__ __forceinline__

namespace mvs {

constexpr double kEps = 2.220446049250313e-16;       // reference `epsilon` (system-config.hpp:8)
constexpr double kTol = kEps * 1000.0;                // reference `tolerance` (system-config.hpp:10)
constexpr double kJacobiEps = kEps * 10.0;            // cv::SVDecomp rotation threshold for double
constexpr double kMinVal = 2.2250738585072014e-308;   // DBL_MIN: "zero singular value" in cv::SVDecomp
constexpr double kSqrt2 = 1.4142135623730951;

MVS_DEV double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
MVS_DEV double dsqrt(double a) { return __builtin_sqrt(a); }  // correctly rounded (no fast-math)
MVS_DEV double dabs(double a) { return __builtin_fabs(a); }

// ---------------------------------------------------------------------------------
// One (i, j) step of the one-sided Jacobi SVD (cv::SVDecomp restated, lapack.cpp
// JacobiSVDImpl_): orthogonalise rows Ai, Aj of At, mirror the rotation on Vt.
// ---------------------------------------------------------------------------------
// (x, y) <- (c*x + s*y, c*y - s*x) with the contract's rounding: x' = fma(c, x, s*y), y' = fma(c, y, -(s*x)).
// Written as VOP3 fma that overwrites its own multiplicand: hipcc otherwise picks the 2-address v_fmac form,
// computes both results into temporaries and copies them back at the end of the predicated region
// (20 v_mov_b64 per rotation; every instruction costs a full issue slot at one wave per SIMD).
MVS_DEV void rotate_inplace(double &x, double &y, double c, double s)
{
    double t0, t1;
    asm("v_mul_f64 %0, %4, %3\n\t"
        "v_mul_f64 %1, %4, %2\n\t"
        "v_fma_f64 %3, %5, %3, -%1\n\t"
        "v_fma_f64 %2, %5, %2, %0"
        : "=&v"(t0), "=&v"(t1), "+v"(x), "+v"(y)
        : "v"(s), "v"(c));
}

// ---------------------------------------------------------------------------------
// Unscaled IEEE sqrt / division.  hipcc's correctly rounded f64 sqrt is
//   [scale x by 2^256 if x < 2^-767]  y = rsq(x); g = x*y; h = y/2; r = fma(-h,g,1/2); g = fma(g,r,g); h = fma(h,r,h);
//   d = fma(-g,g,x); g = fma(d,h,g); d = fma(-g,g,x); g = fma(d,h,g)  [unscale]  result = (x == 0 || x == inf) ? x : g
// and its division is the v_div_scale / v_rcp / 2 Newton steps / v_div_fmas / v_div_fixup sequence, where the scale
// and fixup instructions only act on operands near the ends of the exponent range.  The functions below are the
// SAME sequences without the scaling steps, so they return bit-identical (correctly rounded) results whenever the
// guards hold.  The Jacobi loop only RECORDS a violated guard (`bad`); the caller then recomputes the wave's
// hypotheses with the compiler's full sequences (never taken for Hartley-normalised samples, whose A^T A has
// O(1) entries and row norms >= ~1e-17), so the hot loop carries no extra control flow.
// Every instruction costs a full issue slot at one wave per SIMD: this removes ~8 of 18 (sqrt) and 3 of 11 (div).
// ---------------------------------------------------------------------------------
MVS_DEV bool sqrt_fast_ok(double x) { return !(x < 0x1p-767); }  // zero / NaN handled below; tiny -> slow path
MVS_DEV double sqrt_fast(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = dfma(-h, g, 0.5);
    g = dfma(g, r, g);
    h = dfma(h, r, h);
    double d = dfma(-g, g, x);
    g = dfma(d, h, g);
    d = dfma(-g, g, x);
    g = dfma(d, h, g);
    return (x == 0.0 || x == __builtin_inf()) ? x : g;
}
// sqrt_fast for operands known to be finite and non-zero (inside a guarded range): without the final 0 / inf select
// (v_cmp_class + two v_cndmask).  The result for such x is the same correctly rounded root.
MVS_DEV double sqrt_fast_nz(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = dfma(-h, g, 0.5);
    g = dfma(g, r, g);
    h = dfma(h, r, h);
    double d = dfma(-g, g, x);
    g = dfma(d, h, g);
    d = dfma(-g, g, x);
    g = dfma(d, h, g);
    return g;
}
// sqrt_fast_nz that also hands out its by-product h ~ 1 / (2 sqrt(x)): the refined half reciprocal root of the
// sequence (relative error ~2^-50 after the one quadratic refinement of the 2^-26 v_rsq_f64 seed)
MVS_DEV double sqrt_fast_nz_h(double x, double &h_out)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = dfma(-h, g, 0.5);
    g = dfma(g, r, g);
    h = dfma(h, r, h);
    double d = dfma(-g, g, x);
    g = dfma(d, h, g);
    d = dfma(-g, g, x);
    g = dfma(d, h, g);
    h_out = h;
    return g;
}
// n / d from a reciprocal estimate r0 ~ 1 / d good to ~2^-45 or better: ONE Newton step brings it to the last bit (what
// div_fast's two steps do from v_rcp_f64's 2^-26), then the same quotient / remainder / correction as div_fast.
// v_rcp_f64 and v_rsq_f64 cost three v_fma_f64 each (profiles/r02_trans_issue_microbench.txt): a rotation's two divisions
// take their estimates from the two square roots' by-products instead (jacobi_pair) -- no v_rcp_f64, one Newton step less.
// The whole pair step is compared bit for bit with the IEEE one on 2^25 row pairs (test_rotation_parameters_are_ieee_exact).
MVS_DEV double div_seeded(double n, double d, double r0)
{
    const double e = dfma(-d, r0, 1.0);
    const double r = dfma(r0, e, r0);
    const double q = n * r;
    const double rem = dfma(-d, q, n);
    return dfma(rem, r, q);
}
MVS_DEV double div_fast(double n, double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = dfma(-d, r, 1.0);
    r = dfma(r, e, r);
    e = dfma(-d, r, 1.0);
    r = dfma(r, e, r);
    const double q = n * r;
    const double rem = dfma(-d, q, n);
    return dfma(rem, r, q);
}

// CHEAP variant of the pair step (bit-identical decisions and rotations, fewer issue slots; at one wave per SIMD every
// instruction, vector or scalar, costs a full slot).
//
// (1) Convergence test without the square root.  The contract's decision is
//         rotate  <=>  !(|p| <= T),   T = fl(eps10 * fl(sqrt(ab))),   ab = fl(a * b),   eps10 = 10 * 2^-52.
//     T^2 = eps10^2 * ab * (1 + d) with |d| < 5e-16.  With q = fl(p * p), tau = 2^-900 and e* = eps10^2 * ab:
//         q > fl(ab * E2 * (1 + 2e-14) + tau)  =>  p^2 > T^2   (rotate)
//         q < fl(ab * E2 * (1 - 2e-14) - tau)  =>  p^2 < T^2   (skip)
//     For e* >= 2^-850 every quantity is a normal number with relative error < 4e-16 and tau / e* < 1e-15, so both
//     implications hold with a margin of 1e-14; for e* < 2^-850 the first threshold is >= tau >> T^2 (still implies
//     "rotate") and the second is negative (never taken); a q that underflows is < tau, below the first threshold, and
//     satisfies the second only when e* > tau >> q.  Everything else -- the band, NaN -- evaluates the contract's own
//     test, for the whole wavefront (one uniform branch, never taken in practice).  sqrt(ab) is not an input of the
//     rotation, so nothing else changes.
// (2) Range guard of the unscaled sqrt / div sequences as ONE running minimum instead of three compares folded into a
//     flag: the sequences need 2^-400 <= g2 <= 2^400 and |2p| >= 2^-200 (see FAST below).  g2 = fl(4p^2 + beta^2) >= 4p^2
//     and 4p^2 = 4q exactly, so q >= 2^-398 gives both lower bounds; and g2 <= (a + b)^2 (Cauchy-Schwarz), where a + b
//     is bounded by the squared Frobenius norm of the matrix, which rotations preserve: checked ONCE before the sweeps
//     (jacobi_svd_core: sum W <= 2^198).  Exec-masked lanes do not update the minimum, exactly like the flag.
// (3) (gamma - beta) * 0.5 for beta < 0 and gamma + beta otherwise are both (gamma + |beta|) times an exact power of
//     two: one add and one multiply by a selected constant replace two candidates and a 64-bit select (same for den).
constexpr double kJacobiE2Hi = (kJacobiEps * kJacobiEps) * (1.0 + 2e-14);
constexpr double kJacobiE2Lo = (kJacobiEps * kJacobiEps) * (1.0 - 2e-14);
constexpr double kJacobiTau = 0x1p-900;
constexpr double kGuardQMin = 0x1p-398;
constexpr double kGuardWSumMax = 0x1p198;

// v_min_f64 without the canonicalising v_max x, x that __builtin_fmin emits for each operand
MVS_DEV void running_min(double &m, double x) { asm("v_min_f64 %0, %0, %1" : "+v"(m) : "v"(x)); }

template <int M, int N, bool HAS_V, bool INPLACE, bool FAST, bool CHEAP = false>
MVS_DEV void jacobi_pair(double (&Ai)[M], double (&Aj)[M], double (&Vi)[N], double (&Vj)[N], double &Wi, double &Wj,
                         bool &changed, unsigned &rot, bool &bad, double &qmin)
{
    double a = Wi, b = Wj, p = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k)
        p = dfma(Ai[k], Aj[k], p);
    bool rotate;
    double q = 0.0;
    if (CHEAP) {
        const double ab = a * b;
        q = p * p;
        const bool hi = q > dfma(ab, kJacobiE2Hi, kJacobiTau);
        const bool lo = q < dfma(ab, kJacobiE2Lo, -kJacobiTau);
        rotate = hi;
        if (__builtin_expect(__any(!hi && !lo), 0))
            rotate = !(dabs(p) <= kJacobiEps * dsqrt(a * b));
    } else {
        const double ab = a * b;
        double sq_ab;
        if (FAST) {
            bad = bad || !sqrt_fast_ok(ab);
            sq_ab = sqrt_fast(ab);
        } else {
            sq_ab = dsqrt(ab);
        }
        rotate = !(dabs(p) <= kJacobiEps * sq_ab);
    }
    if (rotate) {
        p *= 2.0;
        const double beta = a - b;
        const double g2 = dfma(p, p, beta * beta);
        // beta < 0: s = sqrt(((gamma-beta)*0.5)/gamma), c = p/(gamma*s*2)
        // else    : c = sqrt((gamma+beta)/(gamma*2)),   s = p/(gamma*c*2)
        const bool neg = beta < 0.0;
        double x, y;
        if (FAST) {
            // with gamma and |p| in [2^-200, 2^200] every sqrt / div operand below is far from the ends of the
            // exponent range (num, den in [gamma/2, 2 gamma]; num/den in [1/2, 1]); otherwise flag the lane
            double num, den;
            if (CHEAP) {
                running_min(qmin, q);
            } else {
                bad = bad || !((g2 >= 0x1p-400) && (g2 <= 0x1p400) && (dabs(p) >= 0x1p-200));
            }
            // CHEAP: g2 >= 2^-400 and num / den in [1/2, 1] are guarded (a violated guard recomputes the wavefront with
            // the full sequences), so neither root needs sqrt_fast's zero / infinity select
#ifndef MVS_NO_SEEDED_DIV
            if (CHEAP) {
                // ((gamma - beta) / 2) / gamma for beta < 0 and (gamma + beta) / (2 gamma) otherwise are the SAME real
                // number (t / 2) / gamma, t = gamma + |beta| (the halvings / doublings are exact inside the guard, so
                // they commute with the roundings): no candidates, no selects.  Both divisions take their reciprocal
                // estimates from the by-products of the two square roots (div_seeded).
                double hg, hx;
                const double gamma = sqrt_fast_nz_h(g2, hg);           // hg ~ 1 / (2 gamma)
                const double t = gamma + dabs(beta);
                x = sqrt_fast_nz_h(div_seeded(t * 0.5, gamma, hg + hg), hx);   // hx ~ 1 / (2 x)
                y = div_seeded(p, gamma * x * 2.0, (hg * hx) * 2.0);
            } else
#endif
            {
            const double gamma = CHEAP ? sqrt_fast_nz(g2) : sqrt_fast(g2);
            if (CHEAP) {
                // gamma - beta == gamma + |beta| for beta < 0, and the halving / doubling are exact inside the guard:
                // one add and one multiply by a selected power of two replace two candidates and a 64-bit select
                const double t = gamma + dabs(beta);
                num = t * __hiloint2double(neg ? 0x3fe00000 : 0x3ff00000, 0);
                den = gamma * __hiloint2double(neg ? 0x3ff00000 : 0x40000000, 0);
            } else {
                num = neg ? (gamma - beta) * 0.5 : (gamma + beta);
                den = neg ? gamma : gamma * 2.0;
            }
            x = CHEAP ? sqrt_fast_nz(div_fast(num, den)) : sqrt_fast(div_fast(num, den));
            y = div_fast(p, gamma * x * 2.0);
            }
        } else {
            const double gamma = dsqrt(g2);
            const double num = neg ? (gamma - beta) * 0.5 : (gamma + beta);
            const double den = neg ? gamma : gamma * 2.0;
            x = dsqrt(num / den);
            y = p / (gamma * x * 2.0);
        }
        const double c = neg ? y : x;
        const double s = neg ? x : y;
        a = 0.0;
        b = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) {
            if (INPLACE) {
                rotate_inplace(Ai[k], Aj[k], c, s);
            } else {
                const double t0 = dfma(c, Ai[k], s * Aj[k]);
                const double t1 = dfma(c, Aj[k], -(s * Ai[k]));
                Ai[k] = t0;
                Aj[k] = t1;
            }
            a = dfma(Ai[k], Ai[k], a);
            b = dfma(Aj[k], Aj[k], b);
        }
        Wi = a;
        Wj = b;
        changed = true;
        ++rot;
        if (HAS_V) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                if (INPLACE) {
                    rotate_inplace(Vi[k], Vj[k], c, s);
                } else {
                    const double t0 = dfma(c, Vi[k], s * Vj[k]);
                    const double t1 = dfma(c, Vj[k], -(s * Vi[k]));
                    Vi[k] = t0;
                    Vj[k] = t1;
                }
            }
        }
    }
}

// Sweeps until a sweep without rotation (at most max(M, 30)); W ends as singular values.
// At: N rows of length M.  Vt: N x N, initialised to identity here.
template <int M, int N, bool INPLACE = false, bool FAST = false, bool HAS_V = true, bool CHEAP = false>
MVS_DEV void jacobi_svd_core(double (&At)[N][M], double (&Vt)[N][N], double (&W)[N], unsigned &rot, unsigned &pairs,
                             bool &bad)
{
    static_assert(!CHEAP || FAST, "the CHEAP pair step is a form of the FAST one");
    double qmin = 0x1p1000, wsum = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double sd = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k)
            sd = dfma(At[i][k], At[i][k], sd);
        W[i] = sd;
        wsum += sd;
#pragma unroll
        for (int k = 0; k < N; ++k)
            Vt[i][k] = (i == k) ? 1.0 : 0.0;
    }
    constexpr int kMaxIter = M > 30 ? M : 30;
    for (int iter = 0; iter < kMaxIter; ++iter) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < N - 1; ++i) {
#pragma unroll
            for (int j = i + 1; j < N; ++j)
                jacobi_pair<M, N, HAS_V, INPLACE, FAST, CHEAP>(At[i], At[j], Vt[i], Vt[j], W[i], W[j], changed, rot, bad,
                                                               qmin);
        }
        pairs += N * (N - 1) / 2;
        if (!changed)
            break;
    }
    if (CHEAP)
        bad = bad || !((qmin >= kGuardQMin) && (wsum <= kGuardWSumMax));
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double sd = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k)
            sd = dfma(At[i][k], At[i][k], sd);
        W[i] = dsqrt(sd);
    }
}

// Selection sort (descending, strict '<', first maximum wins) of W with a row tag:
// after the call tag[p] is the original row that cv::SVDecomp would have moved to position p.
template <int N>
MVS_DEV void sort_tags_desc(double (&W)[N], int (&tag)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i)
        tag[i] = i;
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
        double best = W[i];
        int bt = tag[i];
        int bj = i;
#pragma unroll
        for (int k = i + 1; k < N; ++k) {
            const bool g = best < W[k];
            best = g ? W[k] : best;
            bt = g ? tag[k] : bt;
            bj = g ? k : bj;
        }
#pragma unroll
        for (int k = i + 1; k < N; ++k) {
            const bool hit = (bj == k);
            W[k] = hit ? W[i] : W[k];
            tag[k] = hit ? tag[i] : tag[k];
        }
        W[i] = best;
        tag[i] = bt;
    }
}

template <int N>
MVS_DEV void select_row(const double (&Mx)[N][N], int row, double (&out)[N])
{
#pragma unroll
    for (int k = 0; k < N; ++k) {
        // start from a constant, not from Mx[0][k]: a select between two loads of the same
        // private array gets folded into one load through a selected POINTER, which would
        // pin the whole matrix in scratch memory.
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i)
            v = (row == i) ? Mx[i][k] : v;
        out[k] = v;
    }
}

// ---------------------------------------------------------------------------------
// A / V wavefront pair (ransac_solve_av_kernel).  The 9x9 solve of one hypothesis per lane needs A^T (81) + V^T (81)
// doubles = 324 registers: one wavefront per SIMD, a third of V^T parked in AGPRs, ~50 accumulator moves per rotation,
// and at one wave per SIMD every instruction of any kind costs a full issue slot.  Split by ROLE instead: the A-wave
// owns A^T, decides and computes every rotation (c, s) exactly as jacobi_pair does and applies it to A^T; the V-wave
// (same lanes = same hypotheses, same SIMD) owns V^T and applies the SAME (c, s) to it.  Each fits in 256 registers,
// so both are resident on the SIMD: two waves per SIMD, no AGPR traffic, and the V-wave's 36 fp64 instructions per
// rotation issue in the shadow of the A-wave's dependent sqrt / div chains.  Every rotation is the same operation on
// the same operands in the same order as in the single-wave form: bit-identical results.
//
// Channel (LDS, one per pair): a ring of kAvRing slots, slot = visit number mod kAvRing.  Per (i, j) visit the A-wave
// writes (c, s) of its rotating lanes, then the 64-bit mask of rotating lanes, then the slot's sequence number
// (release); the V-wave polls the sequence number (acquire), reads the mask and, where set, (c, s).  Both walk the
// same static (i, j) order; the V-wave derives the end of the loop from the masks (a sweep without any rotation).
// LDS operations of one wavefront execute in order, so data -> mask -> sequence needs no further fence than the
// compiler-level release / acquire.  Every spin is bounded: a stuck partner raises `abort`, both leave, the
// hypotheses are reported as failed -- the grid always drains.
constexpr int kAvRing = 8;
constexpr unsigned kAvSpinLimit = 1u << 22;

typedef double av_dbl2 __attribute__((ext_vector_type(2)));   // plain vector type: volatile LDS accesses (ds_*_b128)

struct AvChannel {
    av_dbl2 cs[kAvRing][64];
    unsigned long long mask[kAvRing];
    unsigned seq[kAvRing];
    unsigned cons;     // visits the V-wave has consumed (published every 4th visit)
    unsigned abort;
    unsigned fin_a, fin_v;
    int tag8[64];
    double f[9][64];
};

// LDS operations of one wavefront are executed in program order by the LDS unit, so "data, then mask, then sequence
// number" needs no s_waitcnt between the stores: plain volatile accesses plus a compiler barrier (an atomic release
// store would drain lgkmcnt before every sequence-number write: ~250 stalls of an LDS round trip per hypothesis)
MVS_DEV unsigned av_load(const unsigned *p)
{
    const unsigned v = *(const volatile unsigned *)p;
    asm volatile("" ::: "memory");
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
MVS_DEV void av_store(unsigned *p, unsigned v)
{
    asm volatile("" ::: "memory");
    *(volatile unsigned *)p = v;
}
// wait until *p - want >= 0 (wrap-safe); false = gave up (abort raised).  SLEEP: s_sleep argument between polls (x 64
// clocks): the V-wave has ~400 clocks of slack per visit, every poll costs a VALU slot (v_readfirstlane)
template <int SLEEP = 1>
MVS_DEV bool av_wait_ge(AvChannel &ch, const unsigned *p, unsigned want)
{
    unsigned spins = 0;
#pragma nounroll
    while ((int)(av_load(p) - want) < 0) {
        __builtin_amdgcn_s_sleep(SLEEP);
        if ((++spins & 63u) == 0 && (spins > kAvSpinLimit || av_load(&ch.abort))) {
            av_store(&ch.abort, 1u);
            return false;
        }
    }
    return true;
}

// The 36 (i, j) visits of a sweep are expanded by template recursion: every row index is a compile-time constant, so
// A^T / V^T stay in registers (with #pragma unroll the outer loop was not unrolled around the spin loops and the
// matrices were demoted to scratch memory).
struct AState {
    double W[9];
    double qmin;
    unsigned visit, cons_seen;
    bool active, alive, changed;
};

template <int I, int J>
MVS_DEV void av_A_visit(double (&At)[9][9], AState &st, AvChannel &ch, int lane)
{
    double (&Ai)[9] = At[I];
    double (&Aj)[9] = At[J];
    double a = st.W[I], b = st.W[J], p = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k)
        p = dfma(Ai[k], Aj[k], p);
    const double ab = a * b;
    const double q = p * p;
    const bool hi = q > dfma(ab, kJacobiE2Hi, kJacobiTau);
    const bool lo = q < dfma(ab, kJacobiE2Lo, -kJacobiTau);
    bool rotate = hi;
    if (__builtin_expect(__any(!hi && !lo), 0))
        rotate = !(dabs(p) <= kJacobiEps * dsqrt(a * b));
    rotate = rotate && st.active;
    const unsigned long long m = __ballot(rotate);
    const unsigned slot = st.visit & (kAvRing - 1);
    if (__builtin_expect(st.visit - st.cons_seen >= (unsigned)kAvRing, 0)) {   // ring full as far as this wave knows
        st.alive = st.alive && av_wait_ge(ch, &ch.cons, st.visit - kAvRing + 1);
        st.cons_seen = av_load(&ch.cons);
    }
    if (rotate) {
        p *= 2.0;
        const double beta = a - b;
        const double g2 = dfma(p, p, beta * beta);
        const bool neg = beta < 0.0;
        running_min(st.qmin, q);
        const double gamma = sqrt_fast_nz(g2);
        const double t = gamma + dabs(beta);
        const double num = t * __hiloint2double(neg ? 0x3fe00000 : 0x3ff00000, 0);
        const double den = gamma * __hiloint2double(neg ? 0x3ff00000 : 0x40000000, 0);
        const double x = sqrt_fast_nz(div_fast(num, den));
        const double y = div_fast(p, gamma * x * 2.0);
        const double c = neg ? y : x;
        const double s = neg ? x : y;
        const av_dbl2 csv = {c, s};
        *(volatile av_dbl2 *)&ch.cs[slot][lane] = csv;
        a = 0.0;
        b = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            rotate_inplace(Ai[k], Aj[k], c, s);
            a = dfma(Ai[k], Ai[k], a);
            b = dfma(Aj[k], Aj[k], b);
        }
        st.W[I] = a;
        st.W[J] = b;
        st.changed = true;
    }
    if (lane == 0) {
        *(volatile unsigned long long *)&ch.mask[slot] = m;
        av_store(&ch.seq[slot], st.visit + 1);
    }
    ++st.visit;
    if constexpr (J < 8)
        av_A_visit<I, J + 1>(At, st, ch, lane);
    else if constexpr (I < 7)
        av_A_visit<I + 1, I + 2>(At, st, ch, lane);
}

// A-wave: jacobi_svd_core<9, 9, INPLACE, FAST, /*HAS_V*/ false, CHEAP> with a wave-uniform sweep loop (a converged
// lane stays in the loop, inactive: it would not rotate again anyway -- its state no longer changes) and the channel
// writes.  W ends as the singular values.  Returns false if the partner was lost.
MVS_DEV bool jacobi_A_wave(double (&At)[9][9], double (&W)[9], AvChannel &ch, int lane, bool &bad)
{
    AState st;
    double wsum = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        double sd = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            sd = dfma(At[i][k], At[i][k], sd);
        st.W[i] = sd;
        wsum += sd;
    }
    st.qmin = 0x1p1000;
    st.visit = 0;
    st.cons_seen = 0;
    st.active = true;
    st.alive = true;
    for (int iter = 0; iter < 30; ++iter) {
        st.changed = false;
        av_A_visit<0, 1>(At, st, ch, lane);
        st.active = st.active && st.changed;
        if (!__any(st.active) || !st.alive)
            break;
    }
    bad = bad || !((st.qmin >= kGuardQMin) && (wsum <= kGuardWSumMax));
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        double sd = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            sd = dfma(At[i][k], At[i][k], sd);
        W[i] = dsqrt(sd);
    }
    return st.alive;
}

struct VState {
    unsigned visit;
    unsigned long long any;
    bool alive;
};

template <int I, int J>
MVS_DEV void av_V_visit(double (&Vt)[9][9], VState &st, AvChannel &ch, int lane)
{
    const unsigned slot = st.visit & (kAvRing - 1);
    st.alive = st.alive && av_wait_ge<1>(ch, &ch.seq[slot], st.visit + 1);
    const unsigned long long mv = *(const volatile unsigned long long *)&ch.mask[slot];
    // the builtin returns a SIGNED int: widen through unsigned, or bit 31 smears over the upper half of the mask
    const unsigned m_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(mv >> 32));
    const unsigned m_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mv);
    const unsigned long long m = ((unsigned long long)m_hi << 32) | (unsigned long long)m_lo;
    st.any |= m;
    if (st.alive && ((m >> lane) & 1ull)) {
        const av_dbl2 cs = *(const volatile av_dbl2 *)&ch.cs[slot][lane];
        const double c = cs.x, s = cs.y;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            rotate_inplace(Vt[I][k], Vt[J][k], c, s);
    }
    ++st.visit;
    if ((st.visit & 3u) == 0 && lane == 0)
        av_store(&ch.cons, st.visit);
    if constexpr (J < 8)
        av_V_visit<I, J + 1>(Vt, st, ch, lane);
    else if constexpr (I < 7)
        av_V_visit<I + 1, I + 2>(Vt, st, ch, lane);
}

// V-wave: follows the A-wave's rotations on V^T (initialised to the identity here), then hands back the row of V^T the
// A-wave names (the right singular vector of the smallest singular value).
MVS_DEV void jacobi_V_wave(AvChannel &ch, int lane)
{
    double Vt[9][9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int k = 0; k < 9; ++k)
            Vt[i][k] = (i == k) ? 1.0 : 0.0;
    VState st;
    st.visit = 0;
    st.alive = true;
    for (int iter = 0; iter < 30; ++iter) {
        st.any = 0;
        av_V_visit<0, 1>(Vt, st, ch, lane);
        if (st.any == 0 || !st.alive)
            break;
    }
    if (lane == 0)
        av_store(&ch.cons, st.visit + kAvRing);   // nothing left to wait for on the A side
    if (!(st.alive && av_wait_ge(ch, &ch.fin_a, 1u)))
        return;
    const int row = ch.tag8[lane];
    double f[9];
    select_row<9>(Vt, row, f);
#pragma unroll
    for (int k = 0; k < 9; ++k)
        ch.f[k][lane] = f[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0)
        av_store(&ch.fin_v, 1u);
}

// Null vector of a symmetric 9x9 matrix B (= A^T A): last row of vt of cv::SVDecomp(B).
// HAS_V = false: TIMING EXPERIMENT ONLY (V is never rotated, the result is meaningless)
template <bool INPLACE, bool FAST, bool HAS_V = true, bool CHEAP = false>
MVS_DEV void svd9_last_vt_row(double (&At)[9][9], double (&f)[9], unsigned &rot, unsigned &pairs, bool &bad)
{
    double Vt[9][9], W[9];
    int tag[9];
    jacobi_svd_core<9, 9, INPLACE, FAST, HAS_V, CHEAP>(At, Vt, W, rot, pairs, bad);
    sort_tags_desc<9>(W, tag);
    select_row<9>(Vt, tag[8], f);
}

// Last row of vt of cv::SVDecomp(A) for a 4x4 A given as At = A^T.
// GUARDED: the pair step of the RANSAC solve (unscaled sequences, sqrt-free test); a violated guard raises `bad` and
// the caller recomputes with GUARDED = false (the guards and their proofs do not depend on the matrix size).
template <bool GUARDED = false>
MVS_DEV void svd4_last_vt_row(double (&At)[4][4], double (&x)[4], unsigned &rot, unsigned &pairs, bool &bad)
{
    double Vt[4][4], W[4];
    int tag[4];
    jacobi_svd_core<4, 4, GUARDED, GUARDED, true, GUARDED>(At, Vt, W, rot, pairs, bad);
    sort_tags_desc<4>(W, tag);
    select_row<4>(Vt, tag[3], x);
}
MVS_DEV void svd4_last_vt_row(double (&At)[4][4], double (&x)[4], unsigned &rot, unsigned &pairs)
{
    bool bad = false;
    svd4_last_vt_row<false>(At, x, rot, pairs, bad);
}

// OpenCV's multiply-with-carry generator, used only when a singular value is <= DBL_MIN.
MVS_DEV uint32_t cvrng_next(uint64_t &state)
{
    state = (uint64_t)(uint32_t)state * 4164903690ULL + (uint32_t)(state >> 32);
    return (uint32_t)state;
}

// Full 3x3 SVD: A = U diag(w) Vt, exactly cv::SVDecomp(A, w, u, vt, MODIFY_A | FULL_UV).
// U[i][j] row-major (columns are the left vectors), Vt rows are the right vectors.
// FAST / CHEAP: the pair step of the 9x9 solve (same guards, same proofs: they do not depend on the size); a violated
// guard raises `bad` and the caller recomputes with the full sequences.
template <bool INPLACE = false, bool FAST = false, bool CHEAP = false>
MVS_DEV void svd3_full(const double (&A)[3][3], double (&w)[3], double (&U)[3][3], double (&Vt)[3][3], unsigned &rot,
                       unsigned &pairs, bool &bad3)
{
    double At[3][3], W[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            At[i][k] = A[k][i];
    jacobi_svd_core<3, 3, INPLACE, FAST, true, CHEAP>(At, Vt, W, rot, pairs, bad3);
    // selection sort with physical row swaps (N = 3: cheap)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int bj = i;
        double best = W[i];
#pragma unroll
        for (int k = i + 1; k < 3; ++k) {
            const bool g = best < W[k];
            best = g ? W[k] : best;
            bj = g ? k : bj;
        }
#pragma unroll
        for (int k = i + 1; k < 3; ++k) {
            const bool hit = (bj == k);
            const double tw = W[k];
            W[k] = hit ? W[i] : tw;
            W[i] = hit ? tw : W[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double ta = At[k][c], tv = Vt[k][c];
                At[k][c] = hit ? At[i][c] : ta;
                At[i][c] = hit ? ta : At[i][c];
                Vt[k][c] = hit ? Vt[i][c] : tv;
                Vt[i][c] = hit ? tv : Vt[i][c];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
        w[i] = W[i];
    // left vectors: rows of At scaled by 1/W; (numerically) zero singular values get a
    // pseudo-random row orthogonalised against the previous ones (cold path).
    uint64_t rng = 0x12345678ULL;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double sd = W[i];
        for (int ii = 0; ii < 100 && sd <= kMinVal; ++ii) {
            const double val0 = 1.0 / 3.0;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                At[i][k] = (cvrng_next(rng) & 256u) != 0 ? val0 : -val0;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (j < i) {
                        sd = 0.0;
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            sd += At[i][k] * At[j][k];
                        double asum = 0.0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const double t = At[i][k] - sd * At[j][k];
                            At[i][k] = t;
                            asum += dabs(t);
                        }
                        asum = asum > kJacobiEps * 100.0 ? 1.0 / asum : 0.0;
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            At[i][k] *= asum;
                    }
                }
            }
            sd = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                sd += At[i][k] * At[i][k];
            sd = dsqrt(sd);
        }
        const double s = sd > kMinVal ? 1.0 / sd : 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            At[i][k] *= s;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            U[i][j] = At[j][i];
}
MVS_DEV void svd3_full(const double (&A)[3][3], double (&w)[3], double (&U)[3][3], double (&Vt)[3][3], unsigned &rot,
                       unsigned &pairs)
{
    bool bad3 = false;
    svd3_full<false, false, false>(A, w, U, Vt, rot, pairs, bad3);
}

// ---------------------------------------------------------------------------------
// Philox4x32-10 and the 8-of-M sampler
// ---------------------------------------------------------------------------------
MVS_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                           uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 64-bit product each (v_mad_u64_u32) instead of a high and a low 32-bit multiply: full-width integer multiplies
        // issue at a quarter of the vector rate, and the two Philox calls of a sample were 80 of them
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// idx[k]: k-th draw = slot (w_k * (M - k)) >> 32 among the not yet chosen indices.
MVS_DEV void sample8(uint64_t seed, uint32_t hyp, int M, int sampler, int (&idx)[8])
{
    if (sampler == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            idx[k] = k;
        return;
    }
    uint32_t w[8];
    {
        uint32_t o[4];
        philox4x32_10(hyp, 0u, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        w[0] = o[0]; w[1] = o[1]; w[2] = o[2]; w[3] = o[3];
        philox4x32_10(hyp, 1u, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        w[4] = o[0]; w[5] = o[1]; w[6] = o[2]; w[7] = o[3];
    }
    // sorted[] kept ascending with static indices only
    int sorted[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        sorted[k] = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t r = __umulhi(w[k], (uint32_t)(M - k));
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < k && r >= (uint32_t)sorted[t])
                ++r;
        idx[k] = (int)r;
        // insert r: everything greater shifts up by one
        int carry = (int)r;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t <= k) {
                const int cur = sorted[t];
                const bool sw = carry < cur;
                sorted[t] = sw ? carry : cur;
                carry = sw ? cur : carry;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// find_fundamental_matrix (vision/fundamental-matrix.cpp:18-54,56-140,204-267)
// p1 / p2: the 8 sampled ideal-camera points (x, y), homogeneous 1.
// returns false for a degenerate sample (reference: assert(scale > epsilon), :45).
// ---------------------------------------------------------------------------------
MVS_DEV bool normalise8(const double (&px)[8], const double (&py)[8], double (&nx)[8], double (&ny)[8], double &scale,
                        double &mx, double &my)
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
        nx[i] = dx;
        ny[i] = dy;
        sc += dsqrt(dx * dx + dy * dy);
    }
    sc *= 0.125;
    const bool ok = sc > kEps;
    sc = kSqrt2 / sc;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        nx[i] *= sc;
        ny[i] *= sc;
    }
    scale = sc;
    return ok;
}

struct EightNorm {   // the two Hartley normalisations of one sample
    double s1, s2, m1x, m1y, m2x, m2y;
};

// front half of find_fundamental_matrix: normalise both sets (:18-54), design matrix (:78-87), A^T A (:104-111)
MVS_DEV bool eight_point_front(const double (&x1)[8], const double (&y1)[8], const double (&x2)[8], const double (&y2)[8],
                               double (&At)[9][9], EightNorm &nm)
{
    double a1[8], b1[8], a2[8], b2[8];
    bool ok = normalise8(x1, y1, a1, b1, nm.s1, nm.m1x, nm.m1y);
    ok = normalise8(x2, y2, a2, b2, nm.s2, nm.m2x, nm.m2y) && ok;
    // design matrix rows [x2x1, x2y1, x2, y2x1, y2y1, y2, x1, y1, 1]  (:78-87)
    double A[8][9];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        A[i][0] = a2[i] * a1[i]; A[i][1] = a2[i] * b1[i]; A[i][2] = a2[i];
        A[i][3] = b2[i] * a1[i]; A[i][4] = b2[i] * b1[i]; A[i][5] = b2[i];
        A[i][6] = a1[i];         A[i][7] = b1[i];         A[i][8] = 1.0;
    }
    // A^T A, sequential k, separate mul / add (:104-111); symmetric by construction
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#pragma unroll
        for (int j = i; j < 9; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                acc += A[k][i] * A[k][j];
            At[i][j] = acc;
            At[j][i] = acc;
        }
    }
    return ok;
}

// F = T2^T Fn T1, T = [s 0 -m0 s; 0 s -m1 s; 0 0 1] (fundamental-matrix.cpp:245): the exact path's de-normalisation, separate
// multiplications and additions in the reference's order
MVS_DEV void denormalise_exact(const double (&Fn)[3][3], const EightNorm &nm, double (&F)[9])
{
    const double s1 = nm.s1, s2 = nm.s2;
    const double tx1 = -nm.m1x * s1, ty1 = -nm.m1y * s1, tx2 = -nm.m2x * s2, ty2 = -nm.m2y * s2;
    double G[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        G[0][j] = s2 * Fn[0][j];
        G[1][j] = s2 * Fn[1][j];
        G[2][j] = (tx2 * Fn[0][j] + ty2 * Fn[1][j]) + Fn[2][j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        F[i * 3 + 0] = G[i][0] * s1;
        F[i * 3 + 1] = G[i][1] * s1;
        F[i * 3 + 2] = (G[i][0] * tx1 + G[i][1] * ty1) + G[i][2];
    }
}

// back half: rank-2 enforcement (:127-136) and de-normalisation (:245) of the null vector f
// VAR bit 1024 (with 32 and 128): the 3x3 SVD with the unscaled sequences too
// wout: the singular values of reshape(f) as the 3x3 Jacobi computed them (the pre-screen's gap test reads them)
template <int VAR = 0>
MVS_DEV void eight_point_back(const double (&f)[9], const EightNorm &nm, double (&F)[9], bool &bad, double (&wout)[3])
{
    double Fn[3][3];
    {
        double Fp[3][3] = {{f[0], f[1], f[2]}, {f[3], f[4], f[5]}, {f[6], f[7], f[8]}};
        double w[3], U[3][3], Vt[3][3];
        unsigned r3 = 0, p3 = 0;
        constexpr bool F3 = (VAR & (32 | 128 | 1024)) == (32 | 128 | 1024);
        svd3_full<F3 && (VAR & 16) != 0, F3, F3>(Fp, w, U, Vt, r3, p3, bad);
        wout[0] = w[0]; wout[1] = w[1]; wout[2] = w[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double a = U[i][0] * w[0], b = U[i][1] * w[1];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                Fn[i][j] = a * Vt[0][j] + b * Vt[1][j];
        }
    }
    denormalise_exact(Fn, nm, F);
}

template <int VAR = 0>
MVS_DEV void eight_point_back(const double (&f)[9], const EightNorm &nm, double (&F)[9], bool &bad)
{
    double w[3];
    eight_point_back<VAR>(f, nm, F, bad, w);
}

// VAR: 16 = in-place rotation; 32 = unscaled sqrt / div sequences (flag + recompute);
// 128 (with 32) = sqrt-free convergence test, range record instead of per-operand flags, selected power-of-two factors
template <int VAR>
MVS_DEV bool eight_point(const double (&x1)[8], const double (&y1)[8], const double (&x2)[8], const double (&y2)[8],
                         double (&F)[9], unsigned &rot9, unsigned &pairs9, bool &bad)
{
    double f[9];
    EightNorm nm;
    bool ok;
    {
        double At[9][9];
        ok = eight_point_front(x1, y1, x2, y2, At, nm);
        svd9_last_vt_row<(VAR & 16) != 0, (VAR & 32) != 0, (VAR & 256) == 0, (VAR & 128) != 0>(At, f, rot9, pairs9, bad);
    }
    eight_point_back<VAR>(f, nm, F, bad);
    return ok;
}

// |p2^T F p1| with homogeneous 1 (estimator-RANSAC.cpp:114-116), fused form of the contract.
MVS_DEV double epipolar_residual(const double (&F)[9], double x1, double y1, double x2, double y2)
{
    const double u0 = dfma(x2, F[0], dfma(y2, F[3], F[6]));
    const double u1 = dfma(x2, F[1], dfma(y2, F[4], F[7]));
    const double u2 = dfma(x2, F[2], dfma(y2, F[5], F[8]));
    return dabs(dfma(u0, x1, dfma(u1, y1, u2)));
}

// ---------------------------------------------------------------------------------
// small fixed-size helpers used by the finalize kernel (Eigen left-to-right sums)
// ---------------------------------------------------------------------------------
MVS_DEV double det3(const double (&m)[3][3])
{
    const double h0 = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]);
    const double h1 = m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]);
    const double h2 = m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    return (h0 - h1) + h2;
}

// SO3::rectify (math/lie-group.hpp:84-96): Gram-Schmidt on rows, row 1 not re-normalised.
MVS_DEV void rectify3(double (&R)[3][3])
{
    const double n = dsqrt((R[0][0] * R[0][0] + R[0][1] * R[0][1]) + R[0][2] * R[0][2]);
    const double u00 = R[0][0] / n, u01 = R[0][1] / n, u02 = R[0][2] / n;
    const double d = (R[1][0] * u00 + R[1][1] * u01) + R[1][2] * u02;
    const double u10 = R[1][0] - d * u00, u11 = R[1][1] - d * u01, u12 = R[1][2] - d * u02;
    R[0][0] = u00; R[0][1] = u01; R[0][2] = u02;
    R[1][0] = u10; R[1][1] = u11; R[1][2] = u12;
    R[2][0] = u01 * u12 - u02 * u11;
    R[2][1] = u02 * u10 - u00 * u12;
    R[2][2] = u00 * u11 - u01 * u10;
}

}  // namespace mvs
