// pmc_calib.hip -- known-byte-count kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on
// gfx950 for the access widths the G-VOM kernels use (MI355X_MICROARCH.md "HBM": FETCH_SIZE
// reads 1/2 of a wide coalesced stream; other widths must be calibrated).  Each kernel moves
// exactly BYTES bytes (1 GiB, past the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define BYTES (1ull << 30)

__global__ void calib_read4(const uint32_t *p, size_t n, uint32_t *out) {
    uint32_t a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i];
    if (a == 0xdeadbeef) out[0] = a;
}
__global__ void calib_read16(const uint4 *p, size_t n, uint32_t *out) {
    uint32_t a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 0xdeadbeef) out[0] = a;
}
__global__ void calib_write4(uint32_t *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
__global__ void calib_write16(uint4 *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(i, 1, 2, 3);
}
// one scattered 4-byte atomic per lane (each lane its own 64-B line), n adds in total
__global__ void calib_atomic_scatter(uint32_t *p, size_t lines, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        atomicAdd(&p[((i * 2654435761ull) % lines) * 16], 1u);
}
int main() {
    void *a, *o;
    if (hipMalloc(&a, BYTES) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 0, BYTES);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_read4, dim3(4096), dim3(256), 0, 0, (const uint32_t *)a, BYTES / 4, (uint32_t *)o);
        hipLaunchKernelGGL(calib_read16, dim3(4096), dim3(256), 0, 0, (const uint4 *)a, BYTES / 16, (uint32_t *)o);
        hipLaunchKernelGGL(calib_write4, dim3(4096), dim3(256), 0, 0, (uint32_t *)a, BYTES / 4);
        hipLaunchKernelGGL(calib_write16, dim3(4096), dim3(256), 0, 0, (uint4 *)a, BYTES / 16);
        hipLaunchKernelGGL(calib_atomic_scatter, dim3(4096), dim3(256), 0, 0, (uint32_t *)a, BYTES / 64, (size_t)(16u << 20));
    }
    hipDeviceSynchronize();
    printf("calib done: each read/write kernel moved %llu bytes; atomic kernel issued %u adds\n", (unsigned long long)BYTES, 16u << 20);
    return 0;
}
