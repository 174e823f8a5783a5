// microbenchmark: does v_mfma_f64_16x16x4_f64 execute beside v_fma_f64 on gfx950, or do the two
// share the double-precision pipe?  And: is the MFMA's k-accumulation a chain of IEEE FMAs in a fixed order (which is
// what a bit-exact scoring GEMM would need)?
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coissue tools/mfma_coissue.hip && ./mfma_coissue
//
// Workgroup = 512 threads = 8 wavefronts = 2 per SIMD (LDS-limited to one workgroup per CU).  role[w] selects what
// wavefront w does: 0 = exit at once, 1 = v_fma_f64 stream (8 independent chains), 2 = v_mfma_f64_16x16x4_f64 stream
// (4 independent accumulators), 3 = both interleaved in one instruction stream.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef double double4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(double *out, int iters, int role_lo, int role_hi)
{
    extern __shared__ double sm[];
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? role_lo : role_hi;   // waves w and w + 4 share a SIMD
    double a0 = threadIdx.x, a1 = 1.0, a2 = 2.0, a3 = 3.0, a4 = 4.0, a5 = 5.0, a6 = 6.0, a7 = 7.0;
    const double c = 1.0000001, d = 1e-9;
    double4v acc0 = {0, 0, 0, 0}, acc1 = {1, 1, 1, 1}, acc2 = {2, 2, 2, 2}, acc3 = {3, 3, 3, 3};
    const double ma = 1e-3 * (threadIdx.x & 63), mb = 1.0 / (1 + (threadIdx.x & 15));
    if (role == 1) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        }
    } else if (role == 2) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {   // 8 MFMAs = 8 x 1024 FMAs = the FMA count of 128 v_fma_f64
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc3, 0, 0, 0);
            }
        }
    } else if (role == 3) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc0, 0, 0, 0);
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc1, 0, 0, 0);
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc2, 0, 0, 0);
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
                acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc3, 0, 0, 0);
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            }
        }
    }
    const double4v s = acc0 + acc1 + acc2 + acc3;
    out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s.x + s.y + s.z + s.w;
    if (iters < 0) sm[threadIdx.x] = a0;
}

// one MFMA on caller-supplied operands: layout and rounding probe
__global__ __launch_bounds__(64) void probe(const double *A, const double *B, const double *C, double *D)
{
    const int l = threadIdx.x;
    double4v c = {C[l * 4 + 0], C[l * 4 + 1], C[l * 4 + 2], C[l * 4 + 3]};
    const double4v d = __builtin_amdgcn_mfma_f64_16x16x4f64(A[l], B[l], c, 0, 0, 0);
    D[l * 4 + 0] = d.x; D[l * 4 + 1] = d.y; D[l * 4 + 2] = d.z; D[l * 4 + 3] = d.w;
}

static float run(int role_lo, int role_hi, int iters, double *out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 140000, 0, out, 10, role_lo, role_hi);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 140000, 0, out, iters, role_lo, role_hi);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main()
{
    double *out; hipMalloc(&out, 256 * 512 * 8);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
    const int iters = 10000;
    const double fma_per_wave = 64.0 * iters;          // v_fma_f64 wave-instructions
    const double mfma_per_wave = 8.0 * iters;          // v_mfma_f64_16x16x4 wave-instructions (1024 FMAs each)
    const float t_f = run(1, 0, iters, out), t_m = run(2, 0, iters, out), t_ff = run(1, 1, iters, out);
    const float t_mm = run(2, 2, iters, out), t_fm = run(1, 2, iters, out), t_mix = run(3, 0, iters, out);
    printf("one wave per SIMD:  v_fma_f64 stream %.3f ms (%.2f cyc@2.4GHz per instr)   v_mfma_f64_16x16x4 stream %.3f ms (%.1f cyc per MFMA)\n",
           t_f, t_f * 1e-3 * 2.4e9 / fma_per_wave, t_m, t_m * 1e-3 * 2.4e9 / mfma_per_wave);
    printf("two waves per SIMD: fma+fma %.3f ms   mfma+mfma %.3f ms   fma+mfma %.3f ms (sum of the solo times %.3f, max %.3f)\n", t_ff, t_mm,
           t_fm, t_f + t_m, t_f > t_m ? t_f : t_m);
    printf("one wave, interleaved 8 fma : 1 mfma: %.3f ms (solo sum %.3f, max %.3f)\n", t_mix, t_f + t_m, t_f > t_m ? t_f : t_m);
    printf("-> %s\n", t_fm < 0.75f * (t_f + t_m) ? "the MFMA executes beside the vector FMAs (separate pipe)"
                                                 : "the MFMA and the vector FMAs share the double-precision pipe");

    // layout + rounding probe
    double hA[64], hB[64], hC[256], hD[256];
    srand(12345);
    auto rnd = []() { return ldexp((double)rand() / RAND_MAX - 0.5, rand() % 40 - 20); };
    for (int i = 0; i < 64; ++i) { hA[i] = rnd(); hB[i] = rnd(); }
    for (int i = 0; i < 256; ++i) hC[i] = rnd();
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
    // assumed layout: A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j, D[4 (l / 16) + r][l % 16] in lane l, register r
    int n_asc = 0, n_desc = 0, n_unfused = 0, n_close = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * (l / 16) + r, j = l % 16;
            double asc = hC[l * 4 + r], desc = hC[l * 4 + r], unf = hC[l * 4 + r];
            for (int kk = 0; kk < 4; ++kk) asc = fma(hA[16 * kk + i], hB[16 * kk + j], asc);
            for (int kk = 3; kk >= 0; --kk) desc = fma(hA[16 * kk + i], hB[16 * kk + j], desc);
            for (int kk = 0; kk < 4; ++kk) { volatile double p = hA[16 * kk + i] * hB[16 * kk + j]; unf = unf + p; }
            const double got = hD[l * 4 + r];
            n_asc += memcmp(&got, &asc, 8) == 0;
            n_desc += memcmp(&got, &desc, 8) == 0;
            n_unfused += memcmp(&got, &unf, 8) == 0;
            n_close += fabs(got - asc) <= 1e-9 * (fabs(asc) + 1e-30);
        }
    printf("rounding probe over 256 outputs: layout ok (close) %d, == fma chain k ascending %d, == fma chain k descending %d, == unfused %d\n",
           n_close, n_asc, n_desc, n_unfused);
    return 0;
}
