// kernels_audit.hip -- the full-population audit kernel and the worst-case-construction probes.  NOT a translation unit of its
// own: included at the end of kernels.hip under -DMVS_DEBUG_HOOKS only (libmvslam_hip_dbg.so).  The audit replays every
// hypothesis with the product's own exact-solve device code (same source, compiled into this library) and compares with what a
// stage left in device memory -- since round 5 that stage is run by the PRODUCT binary (mvs_debug_audit_state).
#ifndef MVS_DEBUG_HOOKS
#error "kernels_audit.hip belongs to the diagnostics build"
#endif

// diagnostics: one Jacobi pair step on two rows of three elements, guarded (unscaled sequences, seeded divisions) against
// the compiler's IEEE sqrt / division, bit for bit.  rows: n x 6 doubles (row i, row j).  out[0] = steps whose rotated
// rows or norms differ, out[1] = steps compared (both rotate, guards hold), out[2] = steps where the decision differs
__global__ __launch_bounds__(256) void pairstep_check_kernel(const double *rows, int n, unsigned long long *out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    double A[2][2][3], V[2][2][3], W[2][2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            double sd = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                A[v][r][k] = rows[(size_t)i * 6 + r * 3 + k];
                V[v][r][k] = r == k ? 1.0 : 0.0;
                sd = dfma(A[v][r][k], A[v][r][k], sd);
            }
            W[v][r] = sd;
        }
    }
    {   // accuracy of the reciprocal estimate the divisions start from: max |1 - gamma * 2 h| as double bits in out[3]
        const double g2 = dfma(W[0][0], W[0][0], W[0][1] * W[0][1]) + 0x1p-300;
        double h;
        const double gamma = sqrt_fast_nz_h(g2, h);
        const double e = dabs(dfma(-gamma, h + h, 1.0));
        atomicMax(&out[3], (unsigned long long)__double_as_longlong(e));
    }
    bool ch0 = false, ch1 = false, bad0 = false, bad1 = false;
    unsigned r0 = 0, r1 = 0;
    double q0 = 0x1p1000, q1 = 0x1p1000;
    jacobi_pair<3, 3, true, true, true, true>(A[0][0], A[0][1], V[0][0], V[0][1], W[0][0], W[0][1], ch0, r0, bad0, q0);
    jacobi_pair<3, 3, true, false, false, false>(A[1][0], A[1][1], V[1][0], V[1][1], W[1][0], W[1][1], ch1, r1, bad1, q1);
    if (ch0 != ch1) {
        atomicAdd(&out[2], 1ull);
        return;
    }
    if (!ch0 || !(q0 >= kGuardQMin) || !(W[1][0] + W[1][1] <= kGuardWSumMax))
        return;
    atomicAdd(&out[1], 1ull);
    bool same = __double_as_longlong(W[0][0]) == __double_as_longlong(W[1][0]) &&
                __double_as_longlong(W[0][1]) == __double_as_longlong(W[1][1]);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            same = same && __double_as_longlong(A[0][r][k]) == __double_as_longlong(A[1][r][k]) &&
                   __double_as_longlong(V[0][r][k]) == __double_as_longlong(V[1][r][k]);
    if (!same)
        atomicAdd(&out[0], 1ull);
}

void launch_pairstep_check(const double *rows, int n, unsigned long long *out, hipStream_t stream)
{
    hipLaunchKernelGGL(pairstep_check_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, rows, n, out);
}

void launch_fastmath_check(const double *x, const double *y, int n, unsigned long long *out, hipStream_t stream)
{
    hipLaunchKernelGGL(fastmath_check_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, y, n, out);
}


// ---- full-population audit of the pre-screened stage (diagnostics build; tests/audit_gpu_check.py, VERDICT r3 #1b) -------
// Every hypothesis of every pair is solved EXACTLY once more (the arithmetic of ransac_exact_list_kernel) and scored exactly
// on every match (estimator-RANSAC.cpp:100-129), and what the stage decided about it is checked on the device:
//   PHASE 0  the records are the pre-screen's own (pair_prepare + ransac_prescreen just ran, nothing else): state byte 0 only
//            if the exact path rejects the sample, no approximate record for a rejected sample; for every certified record
//            and EVERY match |r_i(F_J) - r~_i| <= the band the record carries (B), with r~ evaluated as the vector counting
//            kernels do (mode 1: binary32 nested fma on the rounded point; mode 2: the contract's fused form on F~), and
//            U >= c_J >= L for the counts against the record's own thresholds.
//   PHASE 1  the batch has just run the whole default stage: a hypothesis the stage DROPPED (record still approximate, or a
//            mode-0 record whose count was pruned) has an exact count strictly below the pair's final bound (count_viol); a
//            record marked exact holds F_J bit for bit; a survivor's recorded (matrix-core) upper count is >= its exact
//            count; no approximate record reaches the bound without having been solved; state 0 <=> the exact path rejects
//            the sample; maxc[pair] = the largest exact count (the host compares it with the bound and with best_count).
// out[16]: 0 hypotheses audited, 1 state violations, 2 count_viol, 3 upper-bound violations, 4 lower-bound violations,
// 5 matches violating (B), 6 bits of the worst |r_J - r~| / band, 7 certified records (phase 0) / survivors (phase 1)
// checked, 8 largest 9x9 sweep count, 9 exact records that differ from F_J, 10 approximate records at or above the bound,
// 11 matches with a NaN exact residual (skipped in (B)), 12 mode-0 counts that differ from the exact count, 13 matches
// checked for (B), 14 rejected samples, 15 sum of the 9x9 sweep counts
template <int PHASE>
__global__ __launch_bounds__(256, 1) void audit_kernel(BatchDev b, RunParams rp, unsigned long long *out, int32_t *maxc)
{
    extern __shared__ __attribute__((aligned(16))) double s_apts[];
    const int pair = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int M = min(b.M[pair], b.max_kp);
    if (M < 8)
        return;
    const int H = rp.num_hypotheses;
    const uint32_t h = (uint32_t)g * kHypPerBlock + tid;
    const bool live = h < (uint32_t)H;
    const uint32_t hh = live ? h : (uint32_t)(H - 1);
    const uint64_t seed = rp.seed + (uint64_t)b.gidx[pair];
    const double *P = b.pts + (size_t)pair * b.max_kp * 4;
    {
        const double2 *src = reinterpret_cast<const double2 *>(P);
        double2 *dst = reinterpret_cast<double2 *>(s_apts);
        for (int i = tid; i < 2 * M; i += kHypPerBlock)
            dst[i] = src[i];
        __syncthreads();
    }
    double F[9];
    unsigned rot = 0, pairs = 0;
    bool bad = false;
    bool ok = solve_hypothesis<240 + 1024>(seed, hh, M, rp.sampler, P, F, rot, pairs, bad);
    if (__builtin_expect(__any(bad), 0)) {
        rot = 0;
        pairs = 0;
        ok = solve_hypothesis<(240 + 1024) & ~(32 | 128)>(seed, hh, M, rp.sampler, P, F, rot, pairs, bad);
    }
    const unsigned sweeps = pairs / 36u;
    const double thr = pair_max_error_sq(b, rp, pair);
    const int mode = b.mode[pair];
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    const size_t rec = (size_t)pair * Hp + h;
    const int state = live ? (int)b.hyp_okf[rec] : kPsInvalid;
    const double4 *L4 = reinterpret_cast<const double4 *>(s_apts);
    unsigned long long v_state = 0, v_count = 0, v_upper = 0, v_lower = 0, v_band = 0, n_cert = 0, v_F = 0, v_surv = 0, n_nan = 0,
                       v_m0 = 0, n_match = 0;
    double worst = 0.0;
    int cJ = 0;
    if (PHASE == 0) {
        if (live && mode != 0) {
            if (state == kPsInvalid && ok) ++v_state;
            if (state == kPsApprox && !ok) ++v_state;
            if (state == kPsApprox) {
                ++n_cert;
                const double *Fo = b.hyp_F + rec * kHypRec;
                int U = 0, L = 0;
                if (mode == 1) {
                    const float *fo = b.hyp_r32 + rec * kHypRec32;
                    float Ft[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k)
                        Ft[k] = fo[k];
                    const float tu = fo[9], tl = fo[10];
                    const double beta = (double)tu - thr;
                    for (int i = 0; i < M; ++i) {
                        const double4 p = L4[i];
                        const double rJ = epipolar_residual(F, p.x, p.y, p.z, p.w);
                        const float x1 = (float)p.x, y1 = (float)p.y, x2 = (float)p.z, y2 = (float)p.w;
                        // ransac_count32_kernel's chain (count_pair32)
                        const float u0 = __builtin_fmaf(x2, Ft[0], __builtin_fmaf(y2, Ft[3], Ft[6]));
                        const float u1 = __builtin_fmaf(x2, Ft[1], __builtin_fmaf(y2, Ft[4], Ft[7]));
                        const float u2 = __builtin_fmaf(x2, Ft[2], __builtin_fmaf(y2, Ft[5], Ft[8]));
                        const float r32 = __builtin_fabsf(__builtin_fmaf(u0, x1, __builtin_fmaf(u1, y1, u2)));
                        cJ += rJ < thr ? 1 : 0;
                        U += r32 < tu ? 1 : 0;
                        L += r32 < tl ? 1 : 0;
                        if (rJ != rJ) {
                            ++n_nan;
                        } else {
                            const double d = dabs(rJ - (double)r32);
                            ++n_match;
                            if (!(d <= beta)) ++v_band;
                            worst = fmax(worst, d / beta);
                        }
                    }
                } else {
                    double Ft[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k)
                        Ft[k] = Fo[k];
                    const double tu = Fo[9];
                    const double beta = tu - thr;
                    const double tl = thr - (tu - thr) * (1.0 + 1e-9) - 1e-15 * thr;   // ransac_count2_kernel's lower threshold
                    for (int i = 0; i < M; ++i) {
                        const double4 p = L4[i];
                        const double rJ = epipolar_residual(F, p.x, p.y, p.z, p.w);
                        const double rt = epipolar_residual(Ft, p.x, p.y, p.z, p.w);
                        cJ += rJ < thr ? 1 : 0;
                        U += rt < tu ? 1 : 0;
                        L += rt < tl ? 1 : 0;
                        if (rJ != rJ) {
                            ++n_nan;
                        } else {
                            const double d = dabs(rJ - rt);
                            ++n_match;
                            if (!(d <= beta)) ++v_band;
                            worst = fmax(worst, d / beta);
                        }
                    }
                }
                if (U < cJ) ++v_upper;
                if (L > cJ) ++v_lower;
            }
        }
    } else {
        for (int i = 0; i < M; ++i) {
            const double4 p = L4[i];
            const double rJ = epipolar_residual(F, p.x, p.y, p.z, p.w);
            cJ += rJ < thr ? 1 : 0;
        }
        if (live) {
            const int cnt = b.hyp_cnt[rec];
            const int bound = b.bound[pair];
            const double *Fo = b.hyp_F + rec * kHypRec;
            const bool counted = cnt >= 0 && cnt != 0x7fffffff;
            if ((state == kPsInvalid) != !ok) ++v_state;
            if (state == kPsNeedExact) ++v_state;   // nothing may still wait for its exact solve
            if (state == kPsExact) {
                bool same = true;
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    same = same && __double_as_longlong(Fo[k]) == __double_as_longlong(F[k]);
                if (!same) ++v_F;
            }
            if (mode == 0) {
                if (ok && counted && cnt != cJ) ++v_m0;                 // a count that survived the pruning is the exact count
                if (ok && !counted && !(cJ < bound)) ++v_count;         // pruned: cannot reach the bound
                if (ok && counted && cnt < bound && !(cJ < bound)) ++v_count;
            } else {
                if (state == kPsApprox) {
                    if (cnt >= bound) ++v_surv;                         // would have had to be solved exactly
                    if (!(cJ < bound)) ++v_count;                       // DROPPED although its exact count reaches the bound
                } else if (state == kPsExact && counted) {
                    ++n_cert;                                           // a survivor of the counting: its recorded upper count
                    if (cnt < cJ) ++v_upper;
                }
            }
        }
    }
    // reductions: wavefront first, one atomic per wavefront and non-zero counter
    auto wsum = [&](unsigned long long v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            v += __shfl_xor(v, o);
        return v;
    };
    const unsigned long long vals[16] = {live ? 1ull : 0ull, v_state, v_count, v_upper, v_lower, v_band, 0ull, n_cert, 0ull, v_F,
                                         v_surv, n_nan, v_m0, n_match, (live && !ok) ? 1ull : 0ull, live ? sweeps : 0u};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k == 6 || k == 8)
            continue;
        const unsigned long long t = wsum(vals[k]);
        if (lane == 0 && t)
            atomicAdd(&out[k], t);
    }
    unsigned sw = live ? sweeps : 0u;
    int mc = (live && ok) ? cJ : -1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sw = max(sw, (unsigned)__shfl_xor((int)sw, o));
        mc = max(mc, __shfl_xor(mc, o));
        worst = fmax(worst, __shfl_xor(worst, o));
    }
    if (lane == 0) {
        atomicMax(&out[8], (unsigned long long)sw);
        atomicMax(&out[6], (unsigned long long)__double_as_longlong(worst));   // non-negative doubles order like their bits
        if (PHASE == 1)
            atomicMax(&maxc[pair], mc);
    }
}

// ---- worst-case-construction probes (tests/test_constants.py, VERDICT r3 #1c) -------------------------------------------
// every record of pair p except hypothesis keep[p] becomes a rejected sample (keep[p] < 0: the pair is left alone)
__global__ __launch_bounds__(256) void keep_only_kernel(BatchDev b, const int32_t *keep)
{
    const int pair = blockIdx.y;
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    const size_t h = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int k = keep[pair];
    if (h < Hp && k >= 0 && h != (size_t)k)
        b.hyp_okf[(size_t)pair * Hp + h] = (uint8_t)kPsInvalid;
}
// the counting launches of the stage alone, on the pre-screen's own records (every pair forced into mode `pmode`), with the
// counting variant `dense` (0: ransac_count32 in one launch, 1: pilot + matrix-core dense phase + matrix-core finish, the
// product path); afterwards hyp_cnt[keep] = U, bound = the best lower bound
void launch_count_only(const BatchDev &b, const RunParams &rp, int n_active, int pmode, int dense, const int32_t *keep,
                       hipStream_t stream)
{
    launch_prescreen_only(b, rp, n_active, pmode, stream);
    if (keep)
        hipLaunchKernelGGL(keep_only_kernel, dim3(b.max_groups, n_active), dim3(256), 0, stream, b, keep);
    const int old = g_count_dense;
    g_count_dense = dense;
    launch_counting(b, rp, n_active, stream, nullptr, false);
    g_count_dense = old;
}
// the compare-free indicator of the matrix-core counting on caller-supplied accumulator values: ind_u[i] / ind_l[i] = what
// dense_count adds for accumulator a[i] under a record with thresholds (tu[i], tl[i]) and box term T[i]
__global__ __launch_bounds__(64) void indicator_probe_kernel(const float *a, const float *tu, const float *tl, const float *T,
                                                             int n, float *ind_u, float *ind_l, float *scale)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n)
        return;
    const f32x2 negH = {-0x1p100f, -0x1p100f};
    bool lpos;
    const float t2u = dense_t2_upper(tu[i], T[i], true), t2l = dense_t2_lower(tl[i], T[i], true, lpos);
    const f32x2 x = {a[i], -a[i]};
    const f32x2 sq = pk_mul(x, x);
    const f32x2 u = pk_ind(sq, negH, f32x2{t2u, t2u}), l = pk_ind(sq, negH, f32x2{t2l, t2l});
    ind_u[i] = u.x == u.y ? u.x : -1.f;
    ind_l[i] = l.x == l.y ? l.x : -1.f;
    scale[i] = dense_scale(tu[i], T[i], true);   // the dense phase's B-operand scale: s tu' must stay below 2
}
void launch_indicator_probe(const float *a, const float *tu, const float *tl, const float *T, int n, float *ind_u, float *ind_l,
                            float *scale, hipStream_t stream)
{
    hipLaunchKernelGGL(indicator_probe_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, a, tu, tl, T, n, ind_u, ind_l, scale);
}
// de-normalisation + fused residual exactly as the two paths run them: in[i] = Fn (9, row-major), s1, s2, m1x, m1y, m2x, m2y,
// x1, y1, x2, y2; out[i] = {residual under prescreen_denormalise(Fn), residual under denormalise_exact(Fn)}
__global__ __launch_bounds__(64) void rounding_probe_kernel(const double *in, int n, double *out)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n)
        return;
    const double *q = in + (size_t)i * 19;
    double Fn[9], Fn3[3][3], Fa[9], Fb[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        Fn[k] = q[k];
        Fn3[k / 3][k % 3] = q[k];
    }
    EightNorm nm;
    nm.s1 = q[9]; nm.s2 = q[10]; nm.m1x = q[11]; nm.m1y = q[12]; nm.m2x = q[13]; nm.m2y = q[14];
    prescreen_denormalise(Fn, nm, Fa);
    denormalise_exact(Fn3, nm, Fb);
    out[2 * i] = epipolar_residual(Fa, q[15], q[16], q[17], q[18]);
    out[2 * i + 1] = epipolar_residual(Fb, q[15], q[16], q[17], q[18]);
}
void launch_rounding_probe(const double *in, int n, double *out, hipStream_t stream)
{
    hipLaunchKernelGGL(rounding_probe_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, in, n, out);
}

hipError_t launch_audit(const BatchDev &b, const RunParams &rp, int n_active, int phase, unsigned long long *out, int32_t *maxc,
                        hipStream_t stream)
{
    const int G = (rp.num_hypotheses + kHypPerBlock - 1) / kHypPerBlock;
    const size_t lds = (size_t)b.max_kp * 4 * sizeof(double);
    const void *fn = phase == 0 ? reinterpret_cast<const void *>(audit_kernel<0>) : reinterpret_cast<const void *>(audit_kernel<1>);
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxKp * 32);
    if (e != hipSuccess)
        return e;
    if (phase == 0)
        hipLaunchKernelGGL(audit_kernel<0>, dim3(G, n_active), dim3(kHypPerBlock), lds, stream, b, rp, out, maxc);
    else
        hipLaunchKernelGGL(audit_kernel<1>, dim3(G, n_active), dim3(kHypPerBlock), lds, stream, b, rp, out, maxc);
    return hipGetLastError();
}
