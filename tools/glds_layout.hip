// Where does an LDS-DMA (global_load_lds_dwordx3 / x4) put each lane's bytes?  Every lane loads 16 (12) bytes holding
// its lane number and the dword index; the LDS image is dumped.   hipcc --offload-arch=gfx950 -O2 tools/glds_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ unsigned lds_offset(const void *p)
{
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
template <int W>
__global__ void k(const uint32_t *src, uint32_t *out, int high)
{
    __shared__ uint32_t pad[20000];   // pushes `img` beyond 64 KB when high != 0
    __shared__ uint32_t img[512];
    for (int i = threadIdx.x; i < 512; i += 64)
        img[i] = 0xdeadbeefu;
    if (high)
        pad[threadIdx.x] = 1;
    __syncthreads();
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_offset(img));
    const uint32_t *g = src + threadIdx.x * W;
    unsigned keep;
    if (W == 4)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64)
        out[i] = img[i];
    if (threadIdx.x == 0)
        out[512] = dst + (high ? pad[5] : 0);
}

int main()
{
    std::vector<uint32_t> h(64 * 4);
    for (int l = 0; l < 64; ++l)
        for (int d = 0; d < 4; ++d)
            h[l * 4 + d] = (l << 8) | d;
    uint32_t *src, *out;
    hipMalloc(&src, 1024);
    hipMalloc(&out, 513 * 4);
    std::vector<uint32_t> o(513);
    for (int w = 3; w <= 4; ++w) {
        for (int l = 0; l < 64; ++l)
            for (int d = 0; d < w; ++d)
                h[l * w + d] = (l << 8) | d;
        hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice);
        if (w == 4)
            hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, src, out, 0);
        else
            hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, src, out, 0);
        hipMemcpy(o.data(), out, 513 * 4, hipMemcpyDeviceToHost);
        printf("dwordx%d  lds base %u: first 24 dwords of the image (lane<<8 | dword):\n ", w, o[512]);
        for (int i = 0; i < 24; ++i)
            printf(" %04x", o[i]);
        int last = -1;
        for (int i = 0; i < 512; ++i)
            if (o[i] != 0xdeadbeefu)
                last = i;
        printf("\n  last written dword index %d\n", last);
    }
    return 0;
}
