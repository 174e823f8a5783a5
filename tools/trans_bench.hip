// microbenchmark: issue cost of the fp64 transcendental seeds (v_rcp_f64, v_rsq_f64) beside v_fma_f64, one wavefront per
// SIMD, dependency-free streams.   hipcc --offload-arch=gfx950 -O2 tools/trans_bench.hip -o /tmp/trans_bench
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>   // 0 = fma, 1 = rcp, 2 = rsq, 3 = mul
__global__ __launch_bounds__(256) void k(double *out, int iters)
{
    extern __shared__ double sm[];
    double a0 = 1.5 + threadIdx.x, a1 = 1.25, a2 = 2.5, a3 = 3.5, a4 = 4.5, a5 = 5.5, a6 = 6.5, a7 = 7.5;
    double b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0;
    const double c = 1.0000001, d = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (OP == 0)
                asm volatile("v_fma_f64 %0, %8, %16, %17\n v_fma_f64 %1, %9, %16, %17\n v_fma_f64 %2, %10, %16, %17\n v_fma_f64 %3, %11, %16, %17\n"
                             "v_fma_f64 %4, %12, %16, %17\n v_fma_f64 %5, %13, %16, %17\n v_fma_f64 %6, %14, %16, %17\n v_fma_f64 %7, %15, %16, %17\n"
                             : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3), "=v"(b4), "=v"(b5), "=v"(b6), "=v"(b7)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(c), "v"(d));
            else if (OP == 1)
                asm volatile("v_rcp_f64 %0, %8\n v_rcp_f64 %1, %9\n v_rcp_f64 %2, %10\n v_rcp_f64 %3, %11\n"
                             "v_rcp_f64 %4, %12\n v_rcp_f64 %5, %13\n v_rcp_f64 %6, %14\n v_rcp_f64 %7, %15\n"
                             : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3), "=v"(b4), "=v"(b5), "=v"(b6), "=v"(b7)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            else if (OP == 2)
                asm volatile("v_rsq_f64 %0, %8\n v_rsq_f64 %1, %9\n v_rsq_f64 %2, %10\n v_rsq_f64 %3, %11\n"
                             "v_rsq_f64 %4, %12\n v_rsq_f64 %5, %13\n v_rsq_f64 %6, %14\n v_rsq_f64 %7, %15\n"
                             : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3), "=v"(b4), "=v"(b5), "=v"(b6), "=v"(b7)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            else
                asm volatile("v_mul_f64 %0, %8, %16\n v_mul_f64 %1, %9, %16\n v_mul_f64 %2, %10, %16\n v_mul_f64 %3, %11, %16\n"
                             "v_mul_f64 %4, %12, %16\n v_mul_f64 %5, %13, %16\n v_mul_f64 %6, %14, %16\n v_mul_f64 %7, %15, %16\n"
                             : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3), "=v"(b4), "=v"(b5), "=v"(b6), "=v"(b7)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(c));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
    if (iters < 0) sm[threadIdx.x] = a0;
}
template <int OP>
float run(int iters, double *out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 140000, 0, out, 10);   // 140 KB of LDS: one workgroup per CU
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 140000, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    double *out; hipMalloc(&out, 256 * 256 * 8);
    const int iters = 20000;
    const char *name[4] = {"v_fma_f64", "v_rcp_f64", "v_rsq_f64", "v_mul_f64"};
    float t[4] = {run<0>(iters, out), run<1>(iters, out), run<2>(iters, out), run<3>(iters, out)};
    for (int o = 0; o < 4; ++o)
        printf("%s: %.3f ms for %d instructions per wavefront, 1 wavefront per SIMD = %.2f cycles @2.4 GHz each\n", name[o], t[o], 64 * iters,
               t[o] * 1e-3 * 2.4e9 / (64.0 * iters));
    return 0;
}
