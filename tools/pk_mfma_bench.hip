// microbenchmark (gfx950): issue rate of packed binary32 vector instructions, of v_mfma_f32_32x32x16_bf16, and of the two
// interleaved the way the dense counting loop interleaves them.  Build: hipcc --offload-arch=gfx950 -O3 tools/pk_mfma_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

// MODE 0: 16 independent v_pk_fma_f32     1: 16 v_fma_f32      2: 8 x (pk_mul, pk_fma clamp, pk_add) as in dense_count
//      3: 4 independent MFMA bf16 32x32x16  4: 4 MFMA + 2 x 24 packed (the dense loop's tile pair)   5: 16 v_pk_mul_f32
//      6: 16 v_pk_add_f32
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k(float *out, int iters)
{
    f2 a[16];
    for (int i = 0; i < 16; ++i)
        a[i] = f2{(float)threadIdx.x + i, 1.f + i};
    float s[16];
    for (int i = 0; i < 16; ++i)
        s[i] = threadIdx.x * 0.5f + i;
    const f2 c = {1.0000001f, 0.9999999f}, d = {1e-9f, -1e-9f};
    v16f acc[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i)
            acc[j][i] = 0.f;
    uint4 ua = make_uint4(threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    v8bf A = __builtin_bit_cast(v8bf, ua), B = A;
    f2 cnt0 = {0.f, 0.f}, cnt1 = {0.f, 0.f};
    int icnt = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(c.x), "v"(d.x));
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(d));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f2 q;
                asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(q) : "v"(a[i]));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 clamp" : "+v"(q) : "v"(c), "v"(d));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(cnt0) : "v"(q));
            }
        } else if (MODE == 7) {
            // compare + add-with-carry counting: eight compares into eight SGPR pairs, then eight v_addc, twice
            unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
#pragma unroll
            for (int r = 0; r < 16; r += 8) {
                asm volatile("v_cmp_lt_f32_e64 %0, |%8|, %16\n v_cmp_lt_f32_e64 %1, |%9|, %16\n v_cmp_lt_f32_e64 %2, |%10|, %16\n"
                             "v_cmp_lt_f32_e64 %3, |%11|, %16\n v_cmp_lt_f32_e64 %4, |%12|, %16\n v_cmp_lt_f32_e64 %5, |%13|, %16\n"
                             "v_cmp_lt_f32_e64 %6, |%14|, %16\n v_cmp_lt_f32_e64 %7, |%15|, %16\n"
                             : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3), "=s"(m4), "=s"(m5), "=s"(m6), "=s"(m7)
                             : "v"(s[r]), "v"(s[r + 1]), "v"(s[r + 2]), "v"(s[r + 3]), "v"(s[r + 4]), "v"(s[r + 5]), "v"(s[r + 6]),
                               "v"(s[r + 7]), "v"(c.x));
                asm volatile("v_addc_co_u32_e64 %0, %1, %0, 0, %1\n v_addc_co_u32_e64 %0, %2, %0, 0, %2\n"
                             "v_addc_co_u32_e64 %0, %3, %0, 0, %3\n v_addc_co_u32_e64 %0, %4, %0, 0, %4\n"
                             "v_addc_co_u32_e64 %0, %5, %0, 0, %5\n v_addc_co_u32_e64 %0, %6, %0, 0, %6\n"
                             "v_addc_co_u32_e64 %0, %7, %0, 0, %7\n v_addc_co_u32_e64 %0, %8, %0, 0, %8\n"
                             : "+v"(icnt), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3), "+s"(m4), "+s"(m5), "+s"(m6), "+s"(m7));
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[j], 0, 0, 0);
        } else if (MODE == 4) {
            v16f z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            v16f a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, z, 0, 0, 0);
            v16f a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, z, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B, A, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B, A, a1, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f2 q = {a0[r], a0[r + 1]}, p = {a1[r], a1[r + 1]};
                asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(q));
                asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 clamp" : "+v"(q) : "v"(c), "v"(d));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2 clamp" : "+v"(p) : "v"(c), "v"(d));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(cnt0) : "v"(q));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(cnt1) : "v"(p));
            }
            asm volatile("" : "+v"(A));   // keep the MFMAs inside the loop
        }
    }
    float r = cnt0.x + cnt0.y + cnt1.x + cnt1.y + icnt;
    for (int i = 0; i < 16; ++i)
        r += a[i].x + a[i].y + s[i];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i)
            r += acc[j][i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name, double instr_per_iter, int wg_per_cu, float *out)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wg_per_cu waves, each iters * instr_per_iter instructions
    printf("%-44s %d waves/SIMD: %8.3f ms  -> %6.2f clk@2.4GHz per instruction per SIMD\n", name, wg_per_cu, ms,
           ms * 1e-3 * 2.4e9 / (iters * instr_per_iter * wg_per_cu));
}

int main()
{
    float *out;
    hipMalloc(&out, 8 * 256 * 256 * sizeof(float));
    for (int w : {1, 2, 4}) {
        if (w == 1) {
            run<0>("v_pk_fma_f32 x16", 16, 1, out); run<1>("v_fma_f32 x16", 16, 1, out); run<5>("v_pk_mul_f32 x16", 16, 1, out);
            run<6>("v_pk_add_f32 x16", 16, 1, out); run<2>("count triple x8 (24 packed)", 24, 1, out);
            run<3>("mfma_f32_32x32x16_bf16 x4", 4, 1, out); run<4>("tile pair: 4 mfma + 48 packed (per 52)", 52, 1, out);
        } else if (w == 2) {
            run<0>("v_pk_fma_f32 x16", 16, 2, out); run<1>("v_fma_f32 x16", 16, 2, out); run<2>("count triple x8 (24 packed)", 24, 2, out);
            run<3>("mfma_f32_32x32x16_bf16 x4", 4, 2, out); run<4>("tile pair: 4 mfma + 48 packed (per 52)", 52, 2, out);
        } else {
            run<0>("v_pk_fma_f32 x16", 16, 4, out); run<1>("v_fma_f32 x16", 16, 4, out); run<2>("count triple x8 (24 packed)", 24, 4, out);
            run<3>("mfma_f32_32x32x16_bf16 x4", 4, 4, out); run<4>("tile pair: 4 mfma + 48 packed (per 52)", 52, 4, out);
            run<7>("16 x (v_cmp_lt_f32 |a| + v_addc) (per 32)", 32, 4, out);
        }
    }
    return 0;
}
