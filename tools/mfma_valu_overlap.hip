// microbenchmark (gfx950): do the matrix pipe and the vector pipe of a SIMD overlap -- within one wavefront (independent
// instructions interleaved) and across wavefronts (some wavefronts issue only MFMAs, their SIMD neighbours only vector
// instructions)?  The dense counting loop (4 x v_mfma_f32_32x32x16_bf16 + ~38 plain vector instructions per tile pair) and the
// matcher (16 x v_mfma_i32_32x32x32_i8 + ~90) both run at about the SUM of the two issue times; this tool measures what the
// hardware allows.   Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/ovl tools/mfma_valu_overlap.hip && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

// per iteration: NM MFMAs (bf16 32x32x16 if !I8 else i8 32x32x32) and NV v_alignbit_b32 in four independent chains.
// ROLE 0: every wavefront issues both, the vector instructions on registers the MFMAs of this iteration do not write
//         (a software-pipelined loop's steady state);
// ROLE 1: wavefronts 0-3 of the workgroup (one per SIMD) issue only the MFMAs, wavefronts 4-7 only the vector instructions;
// ROLE 2: MFMAs only;  ROLE 3: vector instructions only
template <int ROLE, bool I8, int NM, int NV>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k(float *out, int iters)
{
    const int w = threadIdx.x >> 6;
    const bool do_m = ROLE == 0 || ROLE == 2 || (ROLE == 1 && w < 4);
    const bool do_v = ROLE == 0 || ROLE == 3 || (ROLE == 1 && w >= 4);
    v16f acc[4];
    v16i iacc[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) {
            acc[j][i] = 0.f;
            iacc[j][i] = 0;
        }
    uint4 ua = make_uint4(threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    v8bf A = __builtin_bit_cast(v8bf, ua), B = A;
    v4i Ai = {(int)threadIdx.x, 0x01010101, 0x01000100, 0x00010001}, Bi = Ai;
    unsigned c0 = threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3, x = 0x40000000u + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                if (I8)
                    iacc[j & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ai, Bi, iacc[j & 3], 0, 0, 0);
                else
                    acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[j & 3], 0, 0, 0);
            }
        }
        if (do_v) {
#pragma unroll
            for (int i = 0; i < NV; i += 4) {
                asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(c0) : "v"(x));
                asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(c1) : "v"(x));
                asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(c2) : "v"(x));
                asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(c3) : "v"(x));
            }
        }
        asm volatile("" : "+v"(A), "+v"(Ai));   // keep the MFMAs inside the loop
    }
    float r = (float)(c0 + c1 + c2 + c3);
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i)
            r += acc[j][i] + (float)iacc[j][i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int ROLE, bool I8, int NM, int NV>
float run(float *out)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<ROLE, I8, NM, NV>), dim3(512), dim3(512), 0, 0, out, 10);    // 2 workgroups per CU: 4 wavefronts per SIMD
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ROLE, I8, NM, NV>), dim3(512), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3f * 2.4e9f / iters;   // clocks (2.4 GHz-equivalent) per iteration of ONE wavefront slot; 4 slots per SIMD
}

template <bool I8, int NM, int NV>
void table(const char *name, float *out)
{
    // per SIMD and iteration: all four wavefronts do the work in ROLE 0 / 2 / 3; in ROLE 1 two do the MFMAs, two the vector work
    const float m = run<2, I8, NM, NV>(out), v = run<3, I8, NM, NV>(out), both = run<0, I8, NM, NV>(out), split = run<1, I8, NM, NV>(out);
    printf("%-34s  4 waves/SIMD, clocks per iteration:  MFMA only %7.1f   vector only %7.1f   same wavefront, both %7.1f "
           "(sum %7.1f, max %7.1f)   specialised wavefronts (2 MFMA + 2 vector per SIMD) %7.1f (half sum %7.1f, half max %7.1f)\n",
           name, m, v, both, m + v, m > v ? m : v, split, (m + v) / 2, (m > v ? m : v) / 2);
}

int main()
{
    float *out;
    hipMalloc(&out, 512 * 512 * sizeof(float));
    table<false, 4, 36>("bf16 32x32x16 x4 + 36 alignbit", out);   // the dense counting loop's tile pair
    table<false, 4, 16>("bf16 32x32x16 x4 + 16 alignbit", out);
    table<false, 4, 64>("bf16 32x32x16 x4 + 64 alignbit", out);
    table<true, 16, 88>("i8 32x32x32 x16 + 88 alignbit", out);    // the matcher's tile
    table<true, 16, 32>("i8 32x32x32 x16 + 32 alignbit", out);
    table<true, 16, 128>("i8 32x32x32 x16 + 128 alignbit", out);
    return 0;
}
