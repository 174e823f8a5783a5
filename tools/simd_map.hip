// Which SIMD does wavefront w of a 512-thread workgroup land on?  (HW_ID.SIMD_ID, bits 5:4 of hwreg 4 on gfx9-family.)
// The A / V wavefront pairs of ransac_solve_av_kernel assume that wavefronts w and w + 4 share a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned *out)
{
    extern __shared__ double sm[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    if ((threadIdx.x & 63) == 0)
        out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
    if (out[0] == 0xdeadbeef) sm[threadIdx.x] = 1.0;
}
int main()
{
    unsigned *d, h[64 * 8];
    hipMalloc(&d, sizeof(h));
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 60000);
    hipLaunchKernelGGL(k, dim3(64), dim3(512), 53000, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int same = 0;
    for (int b = 0; b < 64; ++b) {
        if (b < 6) {
            printf("workgroup %d: SIMD of wavefronts 0..7 =", b);
            for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3);
            printf("   (cu %u)\n", (h[b * 8] >> 8) & 15);
        }
        int ok = 1;
        for (int w = 0; w < 4; ++w) ok &= ((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 4] >> 4) & 3);
        same += ok;
    }
    printf("workgroups in which wavefront w and w + 4 share a SIMD for all w: %d of 64\n", same);
    return 0;
}
