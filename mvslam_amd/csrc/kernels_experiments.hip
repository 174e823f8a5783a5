// kernels_experiments.hip -- retired experiment kernels and device-side probes.  NOT a translation unit of its own: included
// by kernels.hip under -DMVS_DEBUG_HOOKS only (libmvslam_hip_dbg.so), after every device function of the product path and in
// front of the launch wrappers.  Nothing here is reachable from libmvslam_hip.so.  What each kernel was, and why it lost, is in
// docs/DESIGN_rounds_1_2.md and DESIGN.md 4.3; tools/ab_ransac.py and tests/prescreen_gpu_check.py still run them as A/B
// references (byte-identical results are asserted against the product path).
#ifndef MVS_DEBUG_HOOKS
#error "kernels_experiments.hip belongs to the diagnostics build"
#endif

template <int VAR>
__global__ __launch_bounds__(256, 1) void ransac_solve_kernel(BatchDev b, RunParams rp, int respect_mode)
{
    const int pair = blockIdx.y, tid = threadIdx.x;
    const int M = b.M[pair];
    if (M < 8)
        return;
    if (respect_mode && b.mode[pair] != 0)
        return;   // this pair's hypotheses are pre-screened (ransac_prescreen_kernel)
    solve_record<VAR>(b, rp, pair, M, blockIdx.x * blockDim.x + tid);   // any block size that divides 256 (the launch picks it)
}


// counting without per-hypothesis thresholds -- diagnostics build only (tools/ab_ransac.py)
constexpr int kScoreChunk = 1024;   // points staged per pass: 32 KB of LDS -> 4 workgroups per CU
__global__ __launch_bounds__(256) void ransac_score_kernel(BatchDev b, RunParams rp)
{
    const int pair = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int M = b.M[pair];
    WgBest *out = b.wgbest + (size_t)pair * b.max_groups + g;
    if (M < 8) {  // estimator-RANSAC.cpp:25-29
        if (tid == 0) {
            out->count = -1;
            out->hyp = 0xffffffffu;
            out->residual = 0.0;
        }
        return;
    }
    const int H = rp.num_hypotheses;
    const uint32_t h = (uint32_t)g * kHypPerBlock + tid;
    const bool live = h < (uint32_t)H;
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    const double *Fi = b.hyp_F + ((size_t)pair * Hp + h) * kHypRec;
    double F[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
        F[k] = Fi[k];
    const bool ok = b.hyp_okf[(size_t)pair * Hp + h] != 0;
    const double *P = b.pts + (size_t)pair * b.max_kp * 4;
    __shared__ __attribute__((aligned(16))) double s_pts[kScoreChunk * 4];
    const double thr = pair_max_error_sq(b, rp, pair);
    int cnt = 0;
    double res = 0.0;
    for (int c0 = 0; c0 < M; c0 += kScoreChunk) {
        const int n = min(kScoreChunk, M - c0);
        __syncthreads();
        const double2 *src = reinterpret_cast<const double2 *>(P + (size_t)c0 * 4);
        double2 *dst = reinterpret_cast<double2 *>(s_pts);
        for (int i = tid; i < 2 * n; i += kHypPerBlock)
            dst[i] = src[i];
        __syncthreads();
        const double4 *L4 = reinterpret_cast<const double4 *>(s_pts);
#pragma unroll 8
        for (int i = 0; i < n; ++i) {   // same order and the same operations as the fused kernel: same bits
            const double4 p = L4[i];
            const double r = epipolar_residual(F, p.x, p.y, p.z, p.w);
            const bool in = r < thr;
            cnt += in ? 1 : 0;
            res += in ? r : 0.0;   // NaN-safe (a NaN residual is no inlier and adds nothing, as in the reference)
        }
    }
    if (!ok || !live) {
        cnt = -1;
        res = 0.0;
    }
    if (b.hyp_count && live) {
        b.hyp_count[(size_t)pair * H + h] = cnt;
        b.hyp_residual[(size_t)pair * H + h] = res;
    }
    // workgroup arg-best
    Cand me{cnt, h, res};
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Cand other;
        other.cnt = __shfl_xor(me.cnt, o);
        other.hyp = __shfl_xor(me.hyp, o);
        other.res = __shfl_xor(me.res, o);
        if (cand_better(other, me))
            me = other;
    }
    __shared__ Cand s_c[4];
    __shared__ uint32_t s_win;
    if ((tid & 63) == 0)
        s_c[tid >> 6] = me;
    __syncthreads();
    if (tid == 0) {
        Cand best = s_c[0];
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2)
            if (cand_better(s_c[w2], best))
                best = s_c[w2];
        s_win = best.hyp;
        out->count = best.cnt;
        out->hyp = best.hyp;
        out->residual = best.res;
    }
    __syncthreads();
    if (h == s_win) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
            out->F[k] = F[k];
    }
}

// ---- A / V wavefront pairs (device_math.hpp: jacobi_A_wave / jacobi_V_wave) --------------------------------------------
// grid (G, P) as ransac_solve_kernel, but 512 threads: wavefronts 0..3 solve the 256 hypotheses of the group (A role),
// wavefronts 4..7 carry V^T of the same lanes (V role).  Wavefront w and w + 4 share a SIMD (wavefronts of a workgroup
// are dealt round-robin to the four SIMDs), each needs <= 256 registers, so the SIMD holds two waves instead of one and
// nothing lives in AGPRs.  Same F bits as ransac_solve_kernel (the rotations are the same operations in the same
// order); a violated fast-math guard or a lost partner falls back to the single-wave solve of that wavefront.
template <int VAR>
__global__ __launch_bounds__(512, 1) void ransac_solve_av_kernel(BatchDev b, RunParams rp)
{
    __shared__ AvChannel s_ch[4];
    const int pair = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const int M = b.M[pair];
    if (M < 8)
        return;
    const int wave = tid >> 6, lane = tid & 63, role = wave >> 2, pidx = wave & 3;
    if (tid < 4) {
        AvChannel &c = s_ch[tid];
#pragma unroll
        for (int k = 0; k < kAvRing; ++k)
            c.seq[k] = 0u;
        c.cons = 0u;
        c.abort = 0u;
        c.fin_a = 0u;
        c.fin_v = 0u;
    }
    __syncthreads();
    AvChannel &ch = s_ch[pidx];
    if (role == 1) {
        jacobi_V_wave(ch, lane);
        return;
    }
    const int H = rp.num_hypotheses;
    const int ta = pidx * 64 + lane;
    const uint32_t h = (uint32_t)g * kHypPerBlock + ta;
    const uint32_t hh = h < (uint32_t)H ? h : (uint32_t)(H - 1);
    const uint64_t seed = rp.seed + (uint64_t)b.gidx[pair];
    const double *P = b.pts + (size_t)pair * b.max_kp * 4;
    double F[9];
    bool bad = false, ok, alive;
    {
        int idx[8];
        sample8(seed, hh, M, rp.sampler, idx);
        double x1[8], y1[8], x2[8], y2[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double4 p = *reinterpret_cast<const double4 *>(P + (size_t)idx[k] * 4);
            x1[k] = p.x; y1[k] = p.y; x2[k] = p.z; y2[k] = p.w;
        }
        EightNorm nm;
        double f[9];
        {
            double At[9][9], W[9];
            ok = eight_point_front(x1, y1, x2, y2, At, nm);
            alive = jacobi_A_wave(At, W, ch, lane, bad);
            int tag[9];
            sort_tags_desc<9>(W, tag);
            ch.tag8[lane] = tag[8];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0)
            av_store(&ch.fin_a, 1u);
        alive = alive && av_wait_ge(ch, &ch.fin_v, 1u);
#pragma unroll
        for (int k = 0; k < 9; ++k)
            f[k] = ch.f[k][lane];
        bool bad3 = false;
        eight_point_back<0>(f, nm, F, bad3);
    }
    if (__builtin_expect(__any(bad) || !alive, 0)) {
        // a fast-math guard was violated (never for Hartley-normalised samples) or the partner was lost: this wavefront
        // recomputes its 64 hypotheses alone with the compiler's fully scaled sqrt / div (spills to scratch: cold)
        unsigned rot = 0, pairs = 0;
        bool bad2 = false;
        ok = solve_hypothesis<16>(seed, hh, M, rp.sampler, P, F, rot, pairs, bad2);
    }
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    double *Fo = b.hyp_F + ((size_t)pair * Hp + h) * kHypRec;
#pragma unroll
    for (int k = 0; k < 9; ++k)
        Fo[k] = F[k];
    Fo[9] = pair_max_error_sq(b, rp, pair);
    b.hyp_okf[(size_t)pair * Hp + h] = ok ? kPsExact : kPsInvalid;
    if (g == 0 && tid == 0)
        b.bound[pair] = 0;
}

// ---- pruned scoring: ransac_count_kernel + ransac_select_kernel ------------------------------------------------------
// The reference keeps the hypothesis with the most inliers, ties by the smaller residual sum, then by the smaller index
// (estimator-RANSAC.cpp:76-84).  A hypothesis whose count can no longer reach a count that SOME hypothesis of the pair
// has already achieved in full cannot be that winner, whatever its residual: it is dropped the moment
//     count so far + points not yet visited  <  bound          (strict: ties stay in)
// and the result is the same hypothesis, bit for bit, as scoring everything.  On the bench workload 97 % of the
// hypotheses are contaminated and die after ~1/3 of the points.
//
// Mapping (the opposite of the solve): LANES ARE POINTS.  A wavefront takes four hypotheses at a time; their F are
// wave-uniform (scalar loads of the 72-byte records the solve wrote, SGPR operands of v_fma_f64), each lane reads one
// point of the current 64-point block from LDS and evaluates it for the hypotheses still alive, v_cmp writes the
// inlier mask straight to an SGPR pair and s_bcnt1 counts it: 9 VALU instructions per 64 evaluations (the
// hypothesis-per-lane scoring loop needs 15), no cross-lane traffic, and the exit test is scalar code.  Four
// hypotheses in flight amortise the LDS read and keep the SALU / branch latency of the exit tests off the critical
// path (the first attempt in round 1 had one hypothesis in flight and was latency-bound).
// The bound is per pair: LDS copy per workgroup + one word in global memory (atomicMax, refreshed once per group
// with the load issued a group ahead).  Which hypotheses get dropped depends on timing; the winner does not.
// Residual sums are not accumulated here: ransac_select_kernel computes them, in the reference's index order, for the
// hypotheses that tie at the final maximum only.

__device__ __forceinline__ int count_block(const double (&F)[9], const double4 &p, double thr)
{
    const double r = epipolar_residual(F, p.x, p.y, p.z, p.w);
    return __popcll(__ballot(r < thr));   // NaN (padding lanes, degenerate F) compares false
}

// PPL = points per lane and block (1: 64-point blocks, 2: 128-point blocks).  With two points per lane the scalar work per
// (hypothesis, block) -- count add, exit test, slot skip: the scalar unit is shared by the four SIMDs and was ~70 % busy
// with 8 scalar instructions per 9 vector ones -- is amortised over 18 vector instructions; a dying hypothesis is noticed
// up to 64 points later.
template <int CNT_THREADS, int PPL, bool STATS = false>
__global__ __launch_bounds__(CNT_THREADS) void ransac_count_kernel(BatchDev b, RunParams rp, int wg_per_pair)
{
    // two planes of double2, [nblk * 64] each: (x1, y1) and (x2, y2), NaN padded.  A lane reads one element of each with
    // ds_read_b128 at a 16-byte lane stride = 1 KB contiguous per wavefront: conflict-free (the AoS form, 32-byte
    // stride, spent as many cycles in bank conflicts as the kernel was busy: profiles/r02_pmc_summary.json history)
    extern __shared__ __attribute__((aligned(16))) double s_cpts[];
    __shared__ int s_bound;
    const int pair = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int M = min(b.M[pair], b.max_kp);
    if (M < 8)
        return;
    const int H = rp.num_hypotheses;
    const size_t Hp = (size_t)b.max_groups * kHypPerBlock;
    constexpr int BW = 64 * PPL;                  // points per block
    const int nblk = (M + BW - 1) / BW;
    double2 *s_p1 = reinterpret_cast<double2 *>(s_cpts);
    double2 *s_p2 = s_p1 + nblk * BW;
    {
        const double4 *src = reinterpret_cast<const double4 *>(b.pts + (size_t)pair * b.max_kp * 4);
        const double qnan = __builtin_nan("");
        for (int i = tid; i < nblk * BW; i += CNT_THREADS) {
            const double4 p = i < M ? src[i] : make_double4(qnan, qnan, qnan, qnan);
            s_p1[i] = make_double2(p.x, p.y);
            s_p2[i] = make_double2(p.z, p.w);
        }
    }
    int *gbound = b.bound + pair;
    if (tid == 0)
        s_bound = __hip_atomic_load(gbound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const double thr = pair_max_error_sq(b, rp, pair);
    const double *Fp = b.hyp_F + (size_t)pair * Hp * kHypRec;
    const uint32_t *okp = reinterpret_cast<const uint32_t *>(b.hyp_okf + (size_t)pair * Hp);
    int32_t *cntp = b.hyp_cnt + (size_t)pair * Hp;
    const double2 *L1 = s_p1 + lane, *L2 = s_p2 + lane;
    const int n_groups = (H + kCntSlots - 1) / kCntSlots;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_waves = wg_per_pair * (CNT_THREADS / 64);
    int B = 0;
    unsigned long long visits = 0;   // STATS: (hypothesis, block) evaluations this wavefront executed
    for (int g = blockIdx.x * (CNT_THREADS / 64) + wave; g < n_groups; g += n_waves) {
        const int h0 = g * kCntSlots;
        // the pair's bound as other workgroups see it: load now, use after this group
        const int gb = __hip_atomic_load(gbound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        B = __builtin_amdgcn_readfirstlane(max(B, *(volatile int *)&s_bound));
        // constant address space: wave-uniform scalar loads (s_load_dwordx16 through the scalar cache) instead of 18
        // same-address vector loads per group, which kept the texture-address unit busier than the VALU.  The records
        // were written by the solve launch; nothing writes them while this kernel runs.
        double F0[9], F1[9], F2[9], F3[9];
        const CDouble *f = (const CDouble *)(uintptr_t)(Fp + (size_t)h0 * kHypRec);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            F0[k] = f[k];
            F1[k] = f[kHypRec + k];
            F2[k] = f[2 * kHypRec + k];
            F3[k] = f[3 * kHypRec + k];
        }
        // a v_fma_f64 takes one SGPR operand: keep the addend of the inner FMA (F[6..8]) in VGPRs for the whole group,
        // otherwise it is copied there again for every block
#pragma unroll
        for (int k = 6; k < 9; ++k) {
            asm volatile("" : "+v"(F0[k]));
            asm volatile("" : "+v"(F1[k]));
            asm volatile("" : "+v"(F2[k]));
            asm volatile("" : "+v"(F3[k]));
        }
        const uint32_t ok4 = __builtin_amdgcn_readfirstlane(okp[g]);
        unsigned alive = 0;
#pragma unroll
        for (int k = 0; k < kCntSlots; ++k)
            alive |= (((ok4 >> (8 * k)) & 0xffu) != 0 && h0 + k < H) ? (1u << k) : 0u;
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        // two register sets for the block's points, used alternately: the next block's points are requested before this
        // block's arithmetic (the uniform branches keep the compiler from hoisting the loads) and the LDS round trip hides
        // under the four slots; with one set plus a "next" set the loop carried eight v_mov_b64 per block, a fifth of
        // its vector instructions once two of the four slots have died
        double2 pa0[PPL], pb0[PPL], pa1[PPL], pb1[PPL];
        auto load = [&](double2 (&pa)[PPL], double2 (&pb)[PPL], int blk) {
            const int nb = min(blk, nblk - 1) * BW;
#pragma unroll
            for (int u = 0; u < PPL; ++u) {
                pa[u] = L1[nb + u * 64];
                pb[u] = L2[nb + u * 64];
            }
        };
        auto process = [&](const double2 (&pa)[PPL], const double2 (&pb)[PPL], int blk) {
            double4 p[PPL];
#pragma unroll
            for (int u = 0; u < PPL; ++u)
                p[u] = make_double4(pa[u].x, pa[u].y, pb[u].x, pb[u].y);
            const int need = B - max(M - (blk + 1) * BW, 0);   // a slot whose count stays below this cannot reach B
            if (STATS)
                visits += (unsigned)__builtin_popcount(alive);
            if (alive & 1u) {
#pragma unroll
                for (int u = 0; u < PPL; ++u)
                    c0 += count_block(F0, p[u], thr);
                if (c0 < need) alive &= ~1u;
            }
            if (alive & 2u) {
#pragma unroll
                for (int u = 0; u < PPL; ++u)
                    c1 += count_block(F1, p[u], thr);
                if (c1 < need) alive &= ~2u;
            }
            if (alive & 4u) {
#pragma unroll
                for (int u = 0; u < PPL; ++u)
                    c2 += count_block(F2, p[u], thr);
                if (c2 < need) alive &= ~4u;
            }
            if (alive & 8u) {
#pragma unroll
                for (int u = 0; u < PPL; ++u)
                    c3 += count_block(F3, p[u], thr);
                if (c3 < need) alive &= ~8u;
            }
        };
        load(pa0, pb0, 0);
        for (int blk = 0; blk < nblk && alive; blk += 2) {
            load(pa1, pb1, blk + 1);
            process(pa0, pb0, blk);
            if (!(blk + 1 < nblk && alive))
                break;
            load(pa0, pb0, blk + 2);
            process(pa1, pb1, blk + 1);
        }
        // a slot that is still alive has seen every point: its count is final
        const int v0 = (alive & 1u) ? c0 : -1, v1 = (alive & 2u) ? c1 : -1;
        const int v2 = (alive & 4u) ? c2 : -1, v3 = (alive & 8u) ? c3 : -1;
        if (lane < kCntSlots)
            cntp[h0 + lane] = lane == 0 ? v0 : lane == 1 ? v1 : lane == 2 ? v2 : v3;
        const int cm = __builtin_amdgcn_readfirstlane(max(max(v0, v1), max(v2, v3)));
        if (cm > B) {
            B = cm;
            if (lane == 0) {
                atomicMax(&s_bound, cm);
                __hip_atomic_fetch_max(gbound, cm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        B = max(B, __builtin_amdgcn_readfirstlane(gb));
    }
    if (STATS && lane == 0 && b.stats)
        atomicAdd(&b.stats[2], visits * (unsigned long long)BW);   // executed (hypothesis, point) evaluations incl. padding
}


// diagnostics: one 32 x 32 x 32 tile through the two MFMAs exactly as the counting kernels issue them.  A, B: [32][32] bf16
// bit patterns (row = point / hypothesis, column = K slot); out[point][hypothesis] (binary32).  Pins the K slot mapping, the
// accumulator layout and the accumulation error the bound assumes (tests/test_prescreen.py).
__global__ __launch_bounds__(64) void mfma_probe_kernel(const uint16_t *A, const uint16_t *B, float *out)
{
    const int lane = threadIdx.x, col = lane & 31, half = lane >> 5;
    v8bf a[2], bb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        uint32_t wa[4], wb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int s0 = j * 16 + half * 8 + 2 * e;
            wa[e] = (uint32_t)A[col * 32 + s0] | ((uint32_t)A[col * 32 + s0 + 1] << 16);
            wb[e] = (uint32_t)B[col * 32 + s0] | ((uint32_t)B[col * 32 + s0 + 1] << 16);
        }
        a[j] = __builtin_bit_cast(v8bf, make_uint4(wa[0], wa[1], wa[2], wa[3]));
        bb[j] = __builtin_bit_cast(v8bf, make_uint4(wb[0], wb[1], wb[2], wb[3]));
    }
    v16f acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bb[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bb[1], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;   // the point
        out[row * 32 + col] = acc[r];
    }
}
void launch_mfma_probe(const uint16_t *A, const uint16_t *B, float *out, hipStream_t stream)
{
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, stream, A, B, out);
}


// diagnostics: compare the unscaled sqrt / div sequences with the compiler's IEEE ones on caller-supplied operands.
// out[0] = sqrt mismatches among operands that pass sqrt_fast_ok, out[1] = div mismatches among operand pairs
// inside the guarded range, out[2] / out[3] = number of operands / pairs that were inside the guards.
__global__ __launch_bounds__(256) void fastmath_check_kernel(const double *x, const double *y, int n, unsigned long long *out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const double a = x[i], b = y[i];
    if (sqrt_fast_ok(a)) {
        const double f = sqrt_fast(a), g = dsqrt(a);
        atomicAdd(&out[2], 1ull);
        if (__double_as_longlong(f) != __double_as_longlong(g) && !(f != f && g != g))
            atomicAdd(&out[0], 1ull);
    }
    const double aa = dabs(a), ab = dabs(b);
    if (ab >= 0x1p-200 && ab <= 0x1p200 && ((aa >= 0x1p-200 && aa <= 0x1p200) || a == 0.0)) {
        const double f = div_fast(a, b), g = a / b;
        atomicAdd(&out[3], 1ull);
        if (__double_as_longlong(f) != __double_as_longlong(g))
            atomicAdd(&out[1], 1ull);
    }
}

