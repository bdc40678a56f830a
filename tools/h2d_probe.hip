// What does handing a HOST cloud to the GPU cost, and can chunking hide it behind the trace?  (the unchanged ROS
// node hands over 131,072 x 3 float64 = 3.1 MB of pageable memory per scan.)  Times, per size: one hipMemcpyAsync + sync from pageable
// memory, from pinned memory, the same bytes in 2 / 4 / 8 pageable chunks, and a CPU memcpy into a pinned ring followed by the DMA.
// Build on the GPU box: hipcc -O3 --offload-arch=gfx950 tools/h2d_probe.hip -o /tmp/h2d_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
static double now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t maxb = 16u << 20;
    char *dev, *pin, *page = (char *)malloc(maxb);
    CK(hipMalloc(&dev, maxb)); CK(hipHostMalloc(&pin, maxb, hipHostMallocDefault));
    memset(page, 1, maxb); memset(pin, 2, maxb);
    const size_t sizes[] = {393216, 786432, 1572864, 3145728, 12582912};
    for (size_t b : sizes) {
        double best[6] = {1e30, 1e30, 1e30, 1e30, 1e30, 1e30};
        for (int rep = 0; rep < 30; ++rep) {
            double t0 = now_us();
            CK(hipMemcpyAsync(dev, page, b, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
            double t = now_us() - t0; if (t < best[0]) best[0] = t;
            t0 = now_us();
            CK(hipMemcpyAsync(dev, pin, b, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
            t = now_us() - t0; if (t < best[1]) best[1] = t;
            for (int ci = 0; ci < 3; ++ci) {
                const int nc = 2 << ci;
                t0 = now_us();
                for (int c = 0; c < nc; ++c) CK(hipMemcpyAsync(dev + b / nc * c, page + b / nc * c, b / nc, hipMemcpyHostToDevice, st));
                CK(hipStreamSynchronize(st));
                t = now_us() - t0; if (t < best[2 + ci]) best[2 + ci] = t;
            }
            t0 = now_us();
            for (int c = 0; c < 4; ++c) { memcpy(pin + b / 4 * c, page + b / 4 * c, b / 4); CK(hipMemcpyAsync(dev + b / 4 * c, pin + b / 4 * c, b / 4, hipMemcpyHostToDevice, st)); }
            CK(hipStreamSynchronize(st));
            t = now_us() - t0; if (t < best[5]) best[5] = t;
        }
        printf("%8.2f MB: pageable %7.1f us (%5.1f GB/s)  pinned %7.1f us (%5.1f GB/s)  pageable in 2 / 4 / 8 chunks %7.1f / %7.1f / %7.1f us  "
               "memcpy to a pinned ring + DMA, 4 chunks %7.1f us\n", b / 1e6, best[0], b / best[0] * 1e-3, best[1], b / best[1] * 1e-3, best[2], best[3], best[4], best[5]);
    }
    return 0;
}
