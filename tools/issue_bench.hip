// microbenchmark: dependent-free stream of fp64 FMAs and 32-bit AGPR moves, 1 vs 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MOVS>
__global__ __launch_bounds__(256) void k(double *out, int iters, int ldsbytes_dummy)
{
    extern __shared__ double sm[];
    double a0 = threadIdx.x, a1 = 1.0, a2 = 2.0, a3 = 3.0, a4 = 4.0, a5 = 5.0, a6 = 6.0, a7 = 7.0;
    int m0 = threadIdx.x, m1 = 1, m2 = 2, m3 = 3;
    const double c = 1.0000001, d = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                         "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            if (MOVS) {
                asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_write_b32 a2, %2\n v_accvgpr_write_b32 a3, %3\n"
                             "v_accvgpr_read_b32 %0, a1\n v_accvgpr_read_b32 %1, a2\n v_accvgpr_read_b32 %2, a3\n v_accvgpr_read_b32 %3, a0\n"
                             : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) :: "a0", "a1", "a2", "a3");
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + m0 + m1 + m2 + m3;
    if (ldsbytes_dummy < 0) sm[threadIdx.x] = a0;
}
template <int MOVS>
float run(int blocks, int lds, int iters, double *out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MOVS>, dim3(blocks), dim3(256), lds, 0, out, 10, 0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MOVS>, dim3(blocks), dim3(256), lds, 0, out, iters, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    double *out; hipMalloc(&out, 4096 * 256 * 8);
    const int iters = 20000;
    hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
    hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
    // 1 block (4 waves) per CU via 140 KB LDS -> 1 wave/SIMD; 2 blocks per CU via 70 KB -> 2 waves/SIMD
    for (int mov = 0; mov < 2; ++mov) {
        float t1 = mov ? run<1>(256, 140000, iters, out) : run<0>(256, 140000, iters, out);
        float t2 = mov ? run<1>(512, 70000, iters, out) : run<0>(512, 70000, iters, out);
        const double fma = 64.0 * iters, movs = mov ? 64.0 * iters : 0;
        printf("movs=%d  1 wave/SIMD: %.3f ms (%.2f cyc/instr @2.4GHz)   2 waves/SIMD (2x work): %.3f ms (%.2f cyc/instr per SIMD)\n", mov,
               t1, t1 * 1e-3 * 2.4e9 / (fma + movs), t2, t2 * 1e-3 * 2.4e9 / (2 * (fma + movs)));
    }
    return 0;
}
