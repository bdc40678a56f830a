// atomic_calib.hip -- service rates of gfx950's memory-side atomics for the request shapes
// k_trace produces (timed with HIP events; prints requests/s).  Build: hipcc --offload-arch=gfx950.
//   scatter1    one lane per 64-B line, lines scattered           (k_trace VAR 1 far range)
//   line16      16 lanes per line (4 lines per wave instruction), lines scattered   (lc_flush)
//   hot1        every wave adds ONE lane to the same address       (sensor voxel, VAR 1)
//   hot16       every wave adds 16 lanes to the same line          (sensor patch, lc_flush)
//   hotset16    every wave adds 16 lanes to one of K lines (K = 8, 64, 512)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define LINES (1u << 24)     // 1 GiB of 64-B lines

__global__ void scatter1(uint32_t *p, uint32_t iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (uint32_t k = 0; k < iters; ++k) {
        const uint32_t i = k * nt + t;
        atomicAdd(&p[(size_t)((i * 2654435761u) & (LINES - 1)) * 16], 1u);
    }
}
__global__ void line16(uint32_t *p, uint32_t iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (uint32_t k = 0; k < iters; ++k) {
        const uint32_t i = (k * nt + t) >> 4;             // one line per 16 lanes
        atomicAdd(&p[(size_t)((i * 2654435761u) & (LINES - 1)) * 16 + (t & 15)], 1u);
    }
}
__global__ void hot1(uint32_t *p, uint32_t iters) {
    if ((threadIdx.x & 63) != 0) return;
    for (uint32_t k = 0; k < iters; ++k) atomicAdd(&p[0], 1u);
}
__global__ void hot16(uint32_t *p, uint32_t iters) {
    if ((threadIdx.x & 63) >= 16) return;
    for (uint32_t k = 0; k < iters; ++k) atomicAdd(&p[threadIdx.x & 15], 1u);
}
__global__ void hotset16(uint32_t *p, uint32_t iters, uint32_t K) {
    if ((threadIdx.x & 63) >= 16) return;
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (uint32_t k = 0; k < iters; ++k) {
        const uint32_t line = ((w + k) * 2654435761u >> 8) % K;
        atomicAdd(&p[(size_t)line * 16 * 4099 % ((size_t)LINES * 16) / 16 * 16 + (threadIdx.x & 15)], 1u);
    }
}
#define TIME(name, reqs, ...)                                                                   \
    do {                                                                                        \
        float best = 1e30f;                                                                     \
        for (int r = 0; r < 3; ++r) {                                                           \
            hipEventRecord(e0, 0); hipLaunchKernelGGL(__VA_ARGS__); hipEventRecord(e1, 0);      \
            hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);                \
            if (ms < best) best = ms;                                                           \
        }                                                                                       \
        printf("%-12s %10.0f requests in %8.1f us = %7.2f G req/s  (%6.2f ns/req)\n", name,   \
               (double)(reqs), best * 1e3, (reqs) / (best * 1e6), best * 1e6 / (reqs));        \
    } while (0)
int main() {
    uint32_t *a; hipEvent_t e0, e1;
    if (hipMalloc(&a, (size_t)LINES * 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 0, (size_t)LINES * 64); hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t G = 2048, B = 256, NT = G * B, NW = NT / 64;
    TIME("scatter1", (double)NT * 4, scatter1, dim3(G), dim3(B), 0, 0, a, 4u);
    TIME("line16", (double)NT * 16 / 16, line16, dim3(G), dim3(B), 0, 0, a, 16u);
    TIME("hot1", (double)NW * 4, hot1, dim3(G), dim3(B), 0, 0, a, 4u);
    TIME("hot16", (double)NW * 4, hot16, dim3(G), dim3(B), 0, 0, a, 4u);
    TIME("hotset16/8", (double)NW * 16, hotset16, dim3(G), dim3(B), 0, 0, a, 16u, 8u);
    TIME("hotset16/64", (double)NW * 16, hotset16, dim3(G), dim3(B), 0, 0, a, 16u, 64u);
    TIME("hotset16/512", (double)NW * 16, hotset16, dim3(G), dim3(B), 0, 0, a, 16u, 512u);
    TIME("hotset16/4096", (double)NW * 16, hotset16, dim3(G), dim3(B), 0, 0, a, 16u, 4096u);
    return 0;
}
