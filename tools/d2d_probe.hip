// Is hipMemcpy(device -> device) complete when it returns, and is it ordered against kernels on a
// hipStreamNonBlocking stream?  (round 2's ThreadComm moved the ranks' data with it.)
//   1. times hipMemcpy D2D of 2 GiB against the hipDeviceSynchronize that follows it: a call that
//      returns long before the copy can have run (2 GiB at ~2.5 TB/s = ~1.7 ms) is asynchronous;
//   2. fills a buffer with a slow kernel on a non-blocking stream, copies it with hipMemcpy WITHOUT
//      synchronising that stream, and counts stale words in the copy (the null stream does not wait
//      for non-blocking streams);
//   3. the reverse: hipMemcpy D2D, then at once a kernel on a non-blocking stream that reads the
//      destination -- stale words = the kernel overtook the copy.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/d2d_probe tools/d2d_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <time.h>
#define CK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("%s: %s\n", #c, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
__global__ void fill(uint32_t *p, size_t n, uint32_t v, int spin)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = v;
        for (int k = 0; k < spin; ++k) x = x * 1664525u + 1013904223u;
        p[i] = spin ? (x & 0u) | v : v;
    }
}
__global__ void count_ne(const uint32_t *p, size_t n, uint32_t v, unsigned long long *out)
{
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] != v;
    if (c) atomicAdd(out, c);
}
int main()
{
    const size_t n = (size_t)512 << 20;                  // 2 GiB of uint32
    uint32_t *a, *b; unsigned long long *cnt, h = 0;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&cnt, 8));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        const double t0 = now(); CK(hipMemcpy(b, a, n * 4, hipMemcpyDeviceToDevice));
        const double t1 = now(); CK(hipDeviceSynchronize()); const double t2 = now();
        printf("1. hipMemcpy D2D 2 GiB: call %.3f ms, hipDeviceSynchronize after it %.3f ms\n", t1 - t0, t2 - t1);
    }
    for (int rep = 0; rep < 3; ++rep) {
        const uint32_t v = 100u + rep;
        hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, s, a, n, v, 64);      // slow producer on the non-blocking stream
        CK(hipMemcpy(b, a, n * 4, hipMemcpyDeviceToDevice));                    // null stream: does not wait for `s`
        CK(hipDeviceSynchronize());
        CK(hipMemset(cnt, 0, 8));
        hipLaunchKernelGGL(count_ne, dim3(1024), dim3(256), 0, 0, b, n, v, cnt);
        CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
        printf("2. producer on a non-blocking stream, hipMemcpy without syncing it: %llu of %zu words stale\n", h, n);
    }
    for (int rep = 0; rep < 5; ++rep) {
        const uint32_t v = 200u + rep;
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, s, a, n, v, 0);
        CK(hipStreamSynchronize(s));
        CK(hipMemset(cnt, 0, 8)); CK(hipDeviceSynchronize());
        CK(hipMemcpy(b, a, n * 4, hipMemcpyDeviceToDevice));
        hipLaunchKernelGGL(count_ne, dim3(1024), dim3(256), 0, s, b, n, v, cnt);  // consumer on the non-blocking stream, at once
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
        printf("3. hipMemcpy D2D, then a consumer on a non-blocking stream at once: %llu of %zu words stale\n", h, n);
    }
    return 0;
}
