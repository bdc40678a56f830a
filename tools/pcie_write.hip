// pcie_write.hip -- GPU-initiated writes into pinned host memory: time to store 1.31 MB (the four
// returned maps of a 256 x 256 combine) as runs of 128 / 256 / 512 / 1024 contiguous bytes per
// wave (4- and 8-byte elements: k_map2d's stores; 16-byte elements, up to 1024-byte runs: the widest store there is),
// rows `pitch` bytes apart (what k_map2d's tile shape decides).  hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <stdint.h>
#include <stdio.h>
template <typename V>
__global__ void k_write(V *out, int run_elems, int rows_per_wave, long pitch_elems, long n_elems, int sys)
{
    // wave w writes rows_per_wave runs of run_elems elements; 64 lanes cover 64 / run_elems... elements
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lanes_per_run = run_elems;                 // one element per lane
    const int runs_per_instr = 64 / lanes_per_run;
    for (int r = 0; r < rows_per_wave; r += runs_per_instr) {
        const long row = wave * rows_per_wave + r + lane / lanes_per_run;
        const long idx = row * pitch_elems + (lane % lanes_per_run);
        if (idx < n_elems) {
            if constexpr (sizeof(V) == 16) { out[idx] = V{(unsigned)row, 1u, 2u, 3u}; }
            else if (sys) __hip_atomic_store(&out[idx], (V)row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else out[idx] = (V)row;
        }
    }
}
__global__ void k_empty() {}
int main()
{
    const size_t bytes = 1310720;
    void *host; hipHostMalloc(&host, 4 << 20, hipHostMallocMapped);
    void *dev; hipHostGetDevicePointer(&dev, host, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    {   float best = 1e9f;
        for (int it = 0; it < 20; ++it) { hipEventRecord(e0, 0); hipLaunchKernelGGL(k_empty, dim3(1280), dim3(256), 0, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
                                          float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("an empty kernel of 1280 blocks between the same events: %.1f us\n", best * 1e3); }
    for (int sys = 0; sys < 2; ++sys)
    for (int esz = 4; esz <= 16; esz *= 2)
    for (int run_bytes = 128; run_bytes <= 1024; run_bytes *= 2) {
        if (esz == 16 && sys) continue;
        const int run_elems = run_bytes / esz;
        if (run_elems > 64) continue;
        const long n_elems = bytes / esz, rows = n_elems / run_elems;
        const int rows_per_wave = 64 / run_elems * 2;     // two store instructions per wave
        const long waves = rows / rows_per_wave;
        const int blocks = (int)((waves * 64 + 255) / 256);
        float best = 1e9f;
        for (int it = 0; it < 20; ++it) {
            hipEventRecord(e0, 0);
            if (esz == 4) hipLaunchKernelGGL(k_write<int32_t>, dim3(blocks), dim3(256), 0, 0, (int32_t *)dev, run_elems, rows_per_wave, (long)run_elems, n_elems, sys);
            else if (esz == 16) hipLaunchKernelGGL(k_write<uint4>, dim3(blocks), dim3(256), 0, 0, (uint4 *)dev, run_elems, rows_per_wave, (long)run_elems, n_elems, sys);
            else hipLaunchKernelGGL(k_write<double>, dim3(blocks), dim3(256), 0, 0, (double *)dev, run_elems, rows_per_wave, (long)run_elems, n_elems, sys);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%s-scope %d-byte elements, %3d-byte runs: %.1f us = %.1f GB/s\n", sys ? "system" : "agent ", esz, run_bytes, best * 1e3, bytes / (best * 1e-3) / 1e9);
    }
    return 0;
}
