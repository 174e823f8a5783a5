// microbenchmark (gfx950): what does a kernel boundary cost in a stream of dependent launches, and what does the grid-wide
// barrier cost that a cooperative ("persistent") kernel pays instead?  The single-pair path of the pre-screened stage is fifteen
// dependent launches of one pair's worth of work; this measures both sides of the trade.
//   (a) N empty kernels (one workgroup / 256 workgroups) queued back to back on one stream: time per launch;
//   (b) one kernel of 256 workgroups x 256 threads (one per CU) that passes K grid-wide barriers -- a monotone counter in
//       device memory, every workgroup adds one and spins until the counter reaches its round's target (sense-free: the target
//       grows by the grid size per round) -- time per barrier; with a little dependent work (a store + a load) around it.
// Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/lvb tools/launch_vs_barrier.hip && /tmp/lvb
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void empty_kernel(int *p)
{
    if (p && threadIdx.x == 1024)
        *p = 1;
}

__global__ __launch_bounds__(256) void barrier_kernel(unsigned *counter, unsigned *scratch, int rounds)
{
    const unsigned grid = gridDim.x;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) {
            scratch[blockIdx.x] = (unsigned)r;                      // a result the next phase of another workgroup would read
            __threadfence();
            atomicAdd(counter, 1u);
            const unsigned target = grid * (unsigned)(r + 1);
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target)
                ;
        }
        __syncthreads();
        // the neighbour's store of this round is visible (it may already have stored the next round's value: >= r is fine)
        if (threadIdx.x == 0 &&
            (int)__hip_atomic_load(&scratch[(blockIdx.x + 1) % grid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r)
            scratch[grid] = 0xdeadu;
    }
}

int main()
{
    unsigned *counter, *scratch;
    (void)hipMalloc(&counter, sizeof(unsigned));
    (void)hipMalloc(&scratch, 1024 * sizeof(unsigned));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms;
    for (int wgs : {1, 256}) {
        const int N = 2000;
        for (int i = 0; i < 50; ++i)
            hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(256), 0, 0, (int *)nullptr);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int i = 0; i < N; ++i)
            hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(256), 0, 0, (int *)nullptr);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel, %3d workgroups x 256 threads, %d back-to-back launches on one stream: %.2f us per launch\n", wgs, N,
               ms * 1e3f / N);
    }
    for (int rounds : {1, 101, 1001}) {
        (void)hipMemset(counter, 0, sizeof(unsigned));
        (void)hipMemset(scratch, 0xff, 1024 * sizeof(unsigned));
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(barrier_kernel, dim3(256), dim3(256), 0, 0, counter, scratch, rounds);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned bad = 0;
        (void)hipMemcpy(&bad, scratch + 256, sizeof(unsigned), hipMemcpyDeviceToHost);
        printf("persistent kernel, 256 workgroups (one per CU), %4d grid-wide barriers: %.1f us total%s\n", rounds, ms * 1e3f,
               bad == 0xdeadu ? "  (VISIBILITY FAILURE)" : "");
    }
    return 0;
}
