/*
 * gvom_oracle.c -- CPU restatement of G-VOM's process_pointcloud -> combine_maps path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity referee for the HIP path and the
 * timed host-CPU baseline ("port") of bench.py.  Nothing under g-vom_amd/ may import,
 * link or call it.  It is pinned against golden vectors captured from the reference
 * itself (imported unmodified under Numba's CUDA simulator, tests/golden/make_golden.py;
 * fixtures in tests/golden/ *.npz; checked by tests/test_oracle_golden.py).
 *
 * Every function restates one kernel of /root/reference/scripts/gvom.py ("gvom.py:NNN"
 * below) with the SAME memory layout (voxel index = x + y*xy + z*xy*xy, 2-D maps [x][y]
 * C-order), the same dtypes and the same floating-point operation order as the
 * reference executes under the simulator.  Build with
 * -ffp-contract=off and without fast-math: one fused multiply-add changes a floor().
 *
 * The GPU kernels' atomics are order-independent for every quantity restated here
 * (int32 sums, f32 min), so a sequential loop is one valid serialisation.  Compact row
 * numbering (atomic counter order in the reference, gvom.py:964,993,1158) is assigned
 * in voxel order here; it is value-neutral and tests compare dense equivalents only.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define ORC_API __attribute__((visibility("default")))

/* ALL-CORE BUILD (libgvom_oracle_omp.so: -fopenmp -DORC_OMP; SURVEY 8d "B-cpu-N").  The same
 * source; loops over points / voxels / map cells run on all host threads.  int32 sums and the f32
 * min are order-independent, so the results equal the one-thread build's; compact rows of the fused
 * map are numbered in completion order there (value-neutral, as on the GPU: gvom.py:964,993).
 * tests/test_oracle_golden.py holds the all-core build to the same golden vectors. */
#ifdef ORC_OMP
#include <omp.h>
#define ORC_PRAGMA(x) _Pragma(#x)
#define ORC_PFOR ORC_PRAGMA(omp parallel for schedule(static))
#define ORC_PFOR_DYN ORC_PRAGMA(omp parallel for schedule(dynamic, 512))
#define ORC_ATOMIC ORC_PRAGMA(omp atomic)
#define ORC_NF 32
#define ORC_PARALLEL_UPDATES ORC_PRAGMA(omp parallel reduction(+ : updates))
#define ORC_FOR_DYN ORC_PRAGMA(omp for schedule(dynamic, 1))
/* WHO traces which ray (all-core build): the rays are grouped into azimuth sectors around the sensor (4 per thread) and a
 * sector is traced by one thread, so that beyond the voxels next to the sensor -- which every ray crosses and which each
 * thread accumulates privately, ORC_NF -- the threads add into disjoint wedges of the grid instead of taking turns at the
 * same cache lines (with rays dealt out in cloud order, 128 threads were slower than one: profiles/r3_cpu_thread_sweep.txt).
 * The adds stay atomic (sectors meet along their edges); sums are order-free, so the result is the one-thread build's. */
#define ORC_SECTORS_DECL                                                                        \
    int64_t nsec = 4 * (int64_t)omp_get_max_threads();                                          \
    if (nsec > 2048) nsec = 2048;                                                               \
    int32_t *sec_of = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * 4);                           \
    int64_t *sec_start = (int64_t *)calloc((size_t)nsec + 2, 8);                                \
    int64_t *order = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * 8);                            \
    if (sec_of && sec_start && order) {                                                         \
        ORC_PFOR                                                                                \
        for (int64_t i = 0; i < n; ++i) {                                                       \
            const double ax = (double)pts[i * stride] - ego[0], ay = (double)pts[i * stride + 1] - ego[1]; \
            double a = (atan2(ay, ax) + 3.14159265358979323846) * (0.5 / 3.14159265358979323846) * (double)nsec; \
            if (!(a >= 0.0)) a = 0.0;                                                           \
            int64_t b = (int64_t)a;                                                             \
            sec_of[i] = (int32_t)(b >= nsec ? nsec - 1 : b);                                    \
        }                                                                                       \
        for (int64_t i = 0; i < n; ++i) sec_start[sec_of[i] + 2] += 1;                          \
        for (int64_t b = 0; b < nsec; ++b) sec_start[b + 2] += sec_start[b + 1];                \
        for (int64_t i = 0; i < n; ++i) order[sec_start[sec_of[i] + 1]++] = i;                  \
    } else { nsec = 1; sec_start = NULL; }
#define ORC_SEC_BEGIN(B) (sec_start ? sec_start[B] : 0)
#define ORC_SEC_END(B) (sec_start ? sec_start[(B) + 1] : n)
#define ORC_RAY_OF(K) (sec_start ? order[K] : (K))
#define ORC_SECTORS_FREE free(sec_of); free(sec_start); free(order);
/* the threads' private copies of the ORC_NF^3 box around the sensor: one block per thread, summed into `total` by a
 * parallel loop over the box's cells once every thread has finished tracing.  (Round 3 let every thread flush its own copy
 * with atomics, all of them walking the box in the same order: T threads taking turns at each cache line made the flush
 * cost grow with T^2 -- 26 of a 16-thread step's 38 ms, 2.6 s per step at 256 threads.) */
#define ORC_NF_SHARED int32_t *nf_all = orc_nf_blocks((size_t)omp_get_max_threads());
#define ORC_NF_DECL int32_t *nf = nf_all ? nf_all + (size_t)omp_get_thread_num() * ORC_NF * ORC_NF * ORC_NF : NULL; \
    if (nf) memset(nf, 0, (size_t)ORC_NF * ORC_NF * ORC_NF * 4);
#define ORC_NF_ADD(X, Y, Z)                                                                     \
    {                                                                                           \
        const uint64_t bx = (uint64_t)((X) - nfx), by = (uint64_t)((Y) - nfy), bz = (uint64_t)((Z) - nfz); \
        if (nf && bx < ORC_NF && by < ORC_NF && bz < ORC_NF) nf[bx + ORC_NF * (by + ORC_NF * bz)] += 1;    \
        else { ORC_ATOMIC total[(X) + (Y) * xy + (Z) * xy * xy] += 1; }                         \
    }
#define ORC_NF_FLUSH                                                                            \
    if (nf_all) {                                                                               \
        const int64_t nthr = omp_get_num_threads();                                             \
        ORC_PRAGMA(omp barrier)                                                                 \
        ORC_PRAGMA(omp for schedule(static))                                                    \
        for (int64_t c = 0; c < (int64_t)ORC_NF * ORC_NF * ORC_NF; ++c) {                       \
            int32_t v = 0;                                                                      \
            for (int64_t t = 0; t < nthr; ++t) v += nf_all[t * ORC_NF * ORC_NF * ORC_NF + c];   \
            if (v) {                                                                            \
                const int64_t bx = c % ORC_NF, by = (c / ORC_NF) % ORC_NF, bz = c / (ORC_NF * ORC_NF); \
                /* (cells outside the grid were never counted privately: ORC_NF_ADD is only reached for in-grid voxels) */ \
                ORC_ATOMIC total[(bx + nfx) + (by + nfy) * xy + (bz + nfz) * xy * xy] += v;     \
            }                                                                                   \
        }                                                                                       \
    }
#define ORC_NF_FREE
#define ORC_PRAGMA_REDUCE_NIN ORC_PRAGMA(omp parallel for schedule(static) reduction(+ : n_in))
#define ORC_PRAGMA_CAPTURE ORC_PRAGMA(omp atomic capture)
#else
#define ORC_PRAGMA_CAPTURE
#define ORC_PARALLEL_UPDATES
#define ORC_FOR_DYN
#define ORC_SECTORS_DECL const int64_t nsec = 1;
#define ORC_SEC_BEGIN(B) 0
#define ORC_SEC_END(B) n
#define ORC_RAY_OF(K) (K)
#define ORC_SECTORS_FREE
#define ORC_NF 32
#define ORC_NF_SHARED
#define ORC_NF_DECL
#define ORC_NF_ADD(X, Y, Z) total[(X) + (Y) * xy + (Z) * xy * xy] += 1;
#define ORC_NF_FLUSH
#define ORC_NF_FREE
#define ORC_PRAGMA_REDUCE_NIN
#define ORC_PFOR
#define ORC_PFOR_DYN
#define ORC_ATOMIC
#endif

/* ------------------------------------------------------------------------------------
 * helpers
 * ---------------------------------------------------------------------------------- */
#ifdef ORC_OMP
/* the threads' near-field blocks (ORC_NF_SHARED): kept between calls (calls of one process do not overlap: the Python host
 * logic is one thread), each thread zeroes its own */
static int32_t *orc_nf_blocks(size_t threads)
{
    static int32_t *blocks = NULL;
    static size_t have = 0;
    if (threads > have) {
        free(blocks);
        blocks = (int32_t *)malloc(threads * ORC_NF * ORC_NF * ORC_NF * 4);
        have = blocks ? threads : 0;
    }
    return blocks;
}
#endif

/* Python's max(a, b): returns a unless b > a (matters only for NaN, where no DDA step
 * is taken anyway).  gvom.py:1116 */
static inline float py_maxf(float a, float b) { return (b > a) ? b : a; }
static inline double py_maxd(double a, double b) { return (b > a) ? b : a; }
static inline double py_mind(double a, double b) { return (b < a) ? b : a; }

/* floor() then range test done in double so that huge coordinates never hit an
 * undefined double->int conversion.  Returns 1 if 0 <= floor(v) < size. */
static inline int floor_in_range(double v, int64_t size, int64_t *out)
{
    double f = floor(v);
    if (!(f >= 0.0) || !(f < (double)size)) return 0;
    *out = (int64_t)f;
    return 1;
}

/* 0 (default): ray_length = sqrt((double)ss), what Numba's simulator executes and the fixtures pin;
 * 1: sqrtf(ss) -- gvom.py:1109 as typed for a real CUDA device.  Set through orc_set_cuda_f32_sqrt. */
static int orc_cuda_f32_sqrt = 0;

/* f32 min into shared memory (the reference's cuda.atomic.min, gvom.py:1329) */
static inline void orc_min_f32(float *dst, float v)
{
#ifdef ORC_OMP
    uint32_t expect = __atomic_load_n((uint32_t *)dst, __ATOMIC_RELAXED), want;
    float cur;
    memcpy(&cur, &expect, 4); memcpy(&want, &v, 4);
    while (v < cur) {
        if (__atomic_compare_exchange_n((uint32_t *)dst, &expect, want, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
        memcpy(&cur, &expect, 4);
    }
#else
    if (v < *dst) *dst = v;
#endif
}

/* ------------------------------------------------------------------------------------
 * Per-point kernels, generated for float32 and float64 clouds.  `stride` is the row
 * stride in ELEMENTS (>= 3): the reference indexes points[i, 0..2] of an (N, >=3) array.
 * ---------------------------------------------------------------------------------- */

#define GEN_POINT_KERNELS(T, SUF)                                                               \
                                                                                                \
/* gvom.py:1040-1056  __transform_pointcloud: f64 accumulate, left to right, rows 0..2 of     \
 * the 4x4 (row-major double[16]); written back in the cloud's dtype. */                       \
ORC_API void orc_transform_pointcloud_##SUF(T *pts, int64_t n, int64_t stride, const double *tf)\
{                                                                                               \
    ORC_PFOR                                                                                    \
    for (int64_t i = 0; i < n; ++i) {                                                           \
        T *p = pts + i * stride;                                                                \
        double x = (double)p[0], y = (double)p[1], z = (double)p[2];                            \
        double o0 = ((x * tf[0] + y * tf[1]) + z * tf[2]) + tf[3];                              \
        double o1 = ((x * tf[4] + y * tf[5]) + z * tf[6]) + tf[7];                              \
        double o2 = ((x * tf[8] + y * tf[9]) + z * tf[10]) + tf[11];                            \
        p[0] = (T)o0; p[1] = (T)o1; p[2] = (T)o2;                                               \
    }                                                                                           \
}                                                                                               \
                                                                                                \
/* gvom.py:1060-1150  __point_2_map: min-distance reject (from the WORLD origin),             \
 * endpoint hit, dominant-axis DDA from ego marking `total`.  Returns the number of            \
 * accumulator updates performed (sum of all +1s), used for roofline accounting. */            \
ORC_API int64_t orc_point_2_map_##SUF(double xy_res, double z_res, int64_t xy, int64_t zs,      \
        double min_distance, const T *pts, int64_t n, int64_t stride,                           \
        int32_t *hit, int32_t *total, const double *ego, const double *origin)                  \
{                                                                                               \
    int64_t updates = 0;                                                                        \
    const double md2 = min_distance * min_distance;                                             \
    /* all-core build: every ray crosses the voxels around the sensor, so their `total` words  \
     * would be hammered by all threads; a thread-private copy of the ORC_NF^3 box around the    \
     * sensor takes those adds and is summed into `total` at the end (same sums) */             \
    const int64_t nfx = (int64_t)floor(ego[0] / xy_res - origin[0]) - ORC_NF / 2;              \
    const int64_t nfy = (int64_t)floor(ego[1] / xy_res - origin[1]) - ORC_NF / 2;              \
    const int64_t nfz = (int64_t)floor(ego[2] / z_res - origin[2]) - ORC_NF / 2;               \
    (void)nfx; (void)nfy; (void)nfz;                                                            \
    ORC_SECTORS_DECL                                                                            \
    ORC_NF_SHARED                                                                               \
    ORC_PARALLEL_UPDATES                                                                        \
    {                                                                                           \
    ORC_NF_DECL                                                                                 \
    ORC_FOR_DYN                                                                                 \
    for (int64_t sec = 0; sec < nsec; ++sec)                                                    \
    for (int64_t kk = ORC_SEC_BEGIN(sec); kk < ORC_SEC_END(sec); ++kk) {                        \
        const int64_t i = ORC_RAY_OF(kk);                                                       \
        const T *p = pts + i * stride;                                                          \
        /* :1064 d2 in the cloud's dtype, (x*x + y*y) + z*z */                                  \
        T d2 = (T)((T)((T)(p[0] * p[0]) + (T)(p[1] * p[1])) + (T)(p[2] * p[2]));                \
        if ((double)d2 < md2) continue;                               /* :1067 */               \
        int64_t xi = 0, yi = 0, zi = 0;                                                         \
        int inx = floor_in_range((double)p[0] / xy_res - origin[0], xy, &xi);   /* :1072 */    \
        int iny = floor_in_range((double)p[1] / xy_res - origin[1], xy, &yi);   /* :1076 */    \
        int inz = floor_in_range((double)p[2] / z_res - origin[2], zs, &zi);    /* :1080 */    \
        if (inx && iny && inz) {                                                                \
            int64_t idx = xi + yi * xy + zi * xy * xy;                /* :1086 */               \
            ORC_ATOMIC                                                                          \
            hit[idx] += 1;                                            /* :1089 */               \
            ORC_ATOMIC                                                                          \
            total[idx] += 1;                                          /* :1090 */               \
            updates += 2;                                                                       \
        }                                                                                       \
        float pt[3], end[3], slope[3];                                /* :1093-1095 f32 */      \
        pt[0] = (float)(ego[0] / xy_res);                             /* :1097-1099 */          \
        pt[1] = (float)(ego[1] / xy_res);                                                       \
        pt[2] = (float)(ego[2] / z_res);                                                        \
        end[0] = (float)((double)p[0] / xy_res);                      /* :1101-1103 */          \
        end[1] = (float)((double)p[1] / xy_res);                                                \
        end[2] = (float)((double)p[2] / z_res);                                                 \
        slope[0] = end[0] - pt[0];                                    /* :1105-1107 f32 */      \
        slope[1] = end[1] - pt[1];                                                              \
        slope[2] = end[2] - pt[2];                                                              \
        /* :1109 f32 products and sums, then math.sqrt -> Python float (f64) */                 \
        float ss = (float)((float)((float)(slope[0] * slope[0]) + (float)(slope[1] * slope[1])) \
                           + (float)(slope[2] * slope[2]));                                     \
        /* orc_cuda_f32_sqrt: real Numba-CUDA types math.sqrt(float32) as float32 */ \
        double ray_length = orc_cuda_f32_sqrt ? (double)sqrtf(ss) : sqrt((double)ss);           \
        slope[0] = (float)((double)slope[0] / ray_length);            /* :1112-1114 */          \
        slope[1] = (float)((double)slope[1] / ray_length);                                      \
        slope[2] = (float)((double)slope[2] / ray_length);                                      \
        float a0 = fabsf(slope[0]), a1 = fabsf(slope[1]), a2 = fabsf(slope[2]);                 \
        float slope_max = py_maxf(a0, py_maxf(a1, a2));               /* :1116 */               \
        int si = 0;                                                                             \
        if (slope_max == a1) si = 1;                                  /* :1120 */               \
        if (slope_max == a2) si = 2;                                  /* :1122 */               \
        double length = 0.0;                                          /* :1125 */               \
        float adom = fabsf(slope[si]);                                                          \
        float direction = slope[si] / adom;                           /* :1126 f32 */           \
        int o1 = (si + 1) % 3, o2 = (si + 2) % 3;                                               \
        float inc1 = slope[o1] / adom;                                /* :1129-1132 f32 */      \
        float inc2 = slope[o2] / adom;                                                          \
        double step_len = fabs(1.0 / (double)slope[si]);              /* :1150 f64 */           \
        while (length < ray_length - 1.0) {                           /* :1127 */               \
            pt[si] = pt[si] + direction;                                                        \
            pt[o1] = pt[o1] + inc1;                                                             \
            pt[o2] = pt[o2] + inc2;                                                             \
            if (!floor_in_range((double)pt[0] - origin[0], xy, &xi)) break;   /* :1134-1136 */  \
            if (!floor_in_range((double)pt[1] - origin[1], xy, &yi)) break;   /* :1138-1140 */  \
            if (!floor_in_range((double)pt[2] - origin[2], zs, &zi)) break;   /* :1142-1144 */  \
            ORC_NF_ADD(xi, yi, zi)                                    /* :1146-1148 */          \
            updates += 1;                                                                       \
            length += step_len;                                       /* :1150 */               \
        }                                                                                       \
    }                                                                                           \
    ORC_NF_FLUSH                                                                                \
    }                                                                                           \
    ORC_SECTORS_FREE                                                                            \
    ORC_NF_FREE                                                                                 \
    return updates;                                                                             \
}                                                                                               \
                                                                                                \
/* gvom.py:1303-1329  __calculate_min_height: per in-grid point, f32 atomic-min of the        \
 * fractional z inside its voxel, keyed by compact row.  Returns #points that updated. */      \
ORC_API int64_t orc_calculate_min_height_##SUF(double xy_res, double z_res, int64_t xy,         \
        int64_t zs, double min_distance, const int32_t *index_map, const T *pts, int64_t n,     \
        int64_t stride, float *min_height, const double *origin)                                \
{                                                                                               \
    int64_t n_in = 0;                                                                           \
    const double md2 = min_distance * min_distance;                                             \
    ORC_PRAGMA_REDUCE_NIN                                                                       \
    for (int64_t i = 0; i < n; ++i) {                                                           \
        const T *p = pts + i * stride;                                                          \
        T d2 = (T)((T)((T)(p[0] * p[0]) + (T)(p[1] * p[1])) + (T)(p[2] * p[2]));                \
        if ((double)d2 < md2) continue;                               /* :1308 */               \
        int64_t xi = 0, yi = 0, zi = 0;                                                         \
        if (!floor_in_range((double)p[0] / xy_res - origin[0], xy, &xi)) continue;              \
        if (!floor_in_range((double)p[1] / xy_res - origin[1], xy, &yi)) continue;              \
        if (!floor_in_range((double)p[2] / z_res - origin[2], zs, &zi)) continue;               \
        double lz = ((double)p[2] / z_res - origin[2]) - (double)zi;  /* :1326 */               \
        int32_t row = index_map[xi + yi * xy + zi * xy * xy];         /* :1328 */               \
        float v = (float)lz;                                          /* :1329 f64 -> f32 */    \
        if (row >= 0) { orc_min_f32(&min_height[row], v); ++n_in; }                             \
    }                                                                                           \
    return n_in;                                                                                \
}                                                                                               \
                                                                                                \
/* gvom.py:1172-1220 __calculate_mean (pass = 0) and :1234-1285 __calculate_covariance        \
 * (pass = 1): every point adds to each OCCUPIED voxel of its (2*xy_e+1)^2*(2*z_e+1)           \
 * neighbourhood its coordinates relative to that voxel's corner (voxel units, f64).           \
 * metrics: double[C][10] = mean xyz, cov xx xy xz yy yz zz, count. */                         \
ORC_API void orc_calculate_stats_##SUF(int pass, double xy_res, double z_res, int64_t xy,       \
        int64_t zs, double min_distance, const int32_t *index_map, const T *pts, int64_t n,     \
        int64_t stride, double *metrics, const double *origin, int64_t xy_e, int64_t z_e)       \
{                                                                                               \
    const double md2 = min_distance * min_distance;                                             \
    for (int64_t i = 0; i < n; ++i) {                                                           \
        const T *p = pts + i * stride;                                                          \
        T d2 = (T)((T)((T)(p[0] * p[0]) + (T)(p[1] * p[1])) + (T)(p[2] * p[2]));                \
        if ((double)d2 < md2) continue;                                                         \
        const double ax = (double)p[0] / xy_res - origin[0];                                    \
        const double ay = (double)p[1] / xy_res - origin[1];                                    \
        const double az = (double)p[2] / z_res - origin[2];                                     \
        const double bx = floor(ax), by = floor(ay), bz = floor(az);                            \
        if (!(fabs(bx) < 1e15) || !(fabs(by) < 1e15) || !(fabs(bz) < 1e15)) continue;           \
        const int64_t xb = (int64_t)bx, yb = (int64_t)by, zb = (int64_t)bz;                     \
        for (int64_t xi = xb - xy_e; xi < xb + 1 + xy_e; ++xi) {                                \
            if (xi < 0 || xi >= xy) continue;                                                   \
            for (int64_t yi = yb - xy_e; yi < yb + 1 + xy_e; ++yi) {                            \
                if (yi < 0 || yi >= xy) continue;                                               \
                for (int64_t zi = zb - z_e; zi < zb + 1 + z_e; ++zi) {                          \
                    if (zi < 0 || zi >= zs) continue;                                           \
                    const int32_t row = index_map[xi + yi * xy + zi * xy * xy];                 \
                    if (row < 0) continue;                                                      \
                    const double lx = ax - (double)xi, ly = ay - (double)yi, lz = az - (double)zi; \
                    double *m = metrics + (int64_t)row * 10;                                    \
                    if (pass == 0) {                                                            \
                        m[0] += lx; m[1] += ly; m[2] += lz; m[9] += 1.0;                        \
                    } else {                                                                    \
                        m[3] += (lx - m[0]) * (lx - m[0]);                                      \
                        m[4] += (lx - m[0]) * (ly - m[1]);                                      \
                        m[5] += (lx - m[0]) * (lz - m[2]);                                      \
                        m[6] += (ly - m[1]) * (ly - m[1]);                                      \
                        m[7] += (ly - m[1]) * (lz - m[2]);                                      \
                        m[8] += (lz - m[2]) * (lz - m[2]);                                      \
                    }                                                                           \
                }                                                                               \
            }                                                                                   \
        }                                                                                       \
    }                                                                                           \
}

GEN_POINT_KERNELS(float, f32)
GEN_POINT_KERNELS(double, f64)

/* ------------------------------------------------------------------------------------
 * Sparse encoding of one scan
 * ---------------------------------------------------------------------------------- */

/* gvom.py:1154-1160 __assign_indices.  >=0 compact row (occupied), -1 never observed,
 * -m-1 free with m ray passes.  Returns cell_count. */
ORC_API int32_t orc_assign_indices(const int32_t *hit, const int32_t *total, int32_t *index_map,
                                   int64_t voxel_count)
{
    int32_t cell_count = 0;
#ifdef ORC_OMP
    /* the one-thread numbering (voxel order), in two passes: occupied voxels per chunk, then rows */
    enum { CH = 1 << 15 };
    const int64_t nch = (voxel_count + CH - 1) / CH;
    int32_t base[nch + 1];
    ORC_PFOR
    for (int64_t c = 0; c < nch; ++c) {
        int32_t k = 0;
        const int64_t e = (c + 1) * CH < voxel_count ? (c + 1) * CH : voxel_count;
        for (int64_t i = c * CH; i < e; ++i) k += hit[i] > 0;
        base[c + 1] = k;
    }
    base[0] = 0;
    for (int64_t c = 0; c < nch; ++c) base[c + 1] += base[c];
    cell_count = base[nch];
    ORC_PFOR
    for (int64_t c = 0; c < nch; ++c) {
        int32_t k = base[c];
        const int64_t e = (c + 1) * CH < voxel_count ? (c + 1) * CH : voxel_count;
        for (int64_t i = c * CH; i < e; ++i) {
            if (hit[i] > 0) index_map[i] = k++;
            else index_map[i] = -total[i] - 1;
        }
    }
#else
    for (int64_t i = 0; i < voxel_count; ++i) {
        if (hit[i] > 0) index_map[i] = cell_count++;
        else index_map[i] = -total[i] - 1;
    }
#endif
    return cell_count;
}

/* gvom.py:1164-1168 __move_data */
ORC_API void orc_move_data(const int32_t *old, int32_t *neu, const int32_t *index_map,
                           int64_t voxel_count)
{
    ORC_PFOR
    for (int64_t i = 0; i < voxel_count; ++i)
        if (index_map[i] >= 0) neu[index_map[i]] = old[i];
}

/* ------------------------------------------------------------------------------------
 * Temporal fusion
 * ---------------------------------------------------------------------------------- */

/* shared window test of gvom.py:950-956 / 979-985 / 829-834: d = combined - old origin
 * (f64, integer valued); returns 0 if the shifted voxel falls outside the old window. */
static inline int shifted_index(int64_t x, int64_t y, int64_t z, const double *d, int64_t xy,
                                int64_t zs, int64_t *index_old)
{
    double xs = (double)x + d[0], ys = (double)y + d[1], zz = (double)z + d[2];
    if (xs >= (double)xy || ys >= (double)xy || zz >= (double)zs || xs < 0 || ys < 0 || zz < 0)
        return 0;
    *index_old = (int64_t)(xs + ys * (double)xy + zz * (double)xy * (double)xy);
    return 1;
}

/* gvom.py:943-968 __combine_indices (one ring slot folded into the fused lookup table) */
ORC_API void orc_combine_indices(int64_t *combined_cell_count, int32_t *combined_index_map,
        const double *combined_origin, const int32_t *old_index_map, const double *old_origin,
        int64_t xy, int64_t zs)
{
    double d[3] = { combined_origin[0] - old_origin[0], combined_origin[1] - old_origin[1],
                    combined_origin[2] - old_origin[2] };
    ORC_PFOR
    for (int64_t z = 0; z < zs; ++z)
        for (int64_t y = 0; y < xy; ++y)
            for (int64_t x = 0; x < xy; ++x) {
                int64_t io;
                if (!shifted_index(x, y, z, d, xy, zs, &io)) continue;
                int64_t idx = x + y * xy + z * xy * xy;
                if (old_index_map[io] >= 0 && combined_index_map[idx] <= -1) {        /* :963 */
                    int64_t row;
                    ORC_PRAGMA_CAPTURE
                    row = (*combined_cell_count)++;
                    combined_index_map[idx] = (int32_t)row;
                }
                else if (old_index_map[io] < -1 && combined_index_map[idx] <= -1)     /* :967 */
                    combined_index_map[idx] += old_index_map[io] + 1;
            }
}

/* gvom.py:972-997 __combine_old_indices (previous fused map; decay window [-11,-1]) */
ORC_API void orc_combine_old_indices(int64_t *combined_cell_count, int32_t *combined_index_map,
        const double *combined_origin, const int32_t *old_index_map, const double *old_origin,
        int64_t xy, int64_t zs)
{
    double d[3] = { combined_origin[0] - old_origin[0], combined_origin[1] - old_origin[1],
                    combined_origin[2] - old_origin[2] };
    ORC_PFOR
    for (int64_t z = 0; z < zs; ++z)
        for (int64_t y = 0; y < xy; ++y)
            for (int64_t x = 0; x < xy; ++x) {
                int64_t io;
                if (!shifted_index(x, y, z, d, xy, zs, &io)) continue;
                int64_t idx = x + y * xy + z * xy * xy;
                int32_t c = combined_index_map[idx];
                if (old_index_map[io] >= 0 && c <= -1 && c >= -11) {                  /* :992 */
                    int64_t row;
                    ORC_PRAGMA_CAPTURE
                    row = (*combined_cell_count)++;
                    combined_index_map[idx] = (int32_t)row;
                }
                else if (old_index_map[io] < -1 && c <= -1)                           /* :996 */
                    combined_index_map[idx] += old_index_map[io] + 1;
            }
}

/* gvom.py:821-912 __combine_metrics, lines 910-912 only (hit/total sum, min-height min).
 * The pooled mean/covariance merge (:858-909) feeds no returned map. */
ORC_API void orc_combine_metrics(int32_t *combined_hit, int32_t *combined_total,
        float *combined_min_height, const int32_t *combined_index_map,
        const double *combined_origin, const int32_t *old_hit, const int32_t *old_total,
        const float *old_min_height, const int32_t *old_index_map, const double *old_origin,
        int64_t xy, int64_t zs)
{
    double d[3] = { combined_origin[0] - old_origin[0], combined_origin[1] - old_origin[1],
                    combined_origin[2] - old_origin[2] };
    ORC_PFOR
    for (int64_t z = 0; z < zs; ++z)
        for (int64_t y = 0; y < xy; ++y)
            for (int64_t x = 0; x < xy; ++x) {
                int64_t io;
                if (!shifted_index(x, y, z, d, xy, zs, &io)) continue;
                int32_t index = combined_index_map[x + y * xy + z * xy * xy];
                int32_t index_old = old_index_map[io];
                if (index < 0 || index_old < 0) continue;                             /* :841 */
                combined_hit[index] = combined_hit[index] + old_hit[index_old];       /* :910 */
                combined_total[index] = combined_total[index] + old_total[index_old]; /* :911 */
                float a = combined_min_height[index], b = old_min_height[index_old];
                combined_min_height[index] = (b < a) ? b : a;                         /* :912 */
            }
}

/* ------------------------------------------------------------------------------------
 * Column reductions and 2-D maps.  Maps are [x][y] C-order: m[x * xy + y].
 * ---------------------------------------------------------------------------------- */

/* gvom.py:525-540 __make_height_map */
ORC_API void orc_make_height_map(const double *combined_origin, const int32_t *combined_index_map,
        const float *min_height, int64_t xy, int64_t zs, double xy_res, double z_res,
        const double *ego, double radius, double ground_to_lidar_height, double *height_map)
{
    ORC_PFOR
    for (int64_t x = 0; x < xy; ++x)
        for (int64_t y = 0; y < xy; ++y) {
            double xp = ((combined_origin[0] + (double)x) * xy_res) - ego[0];         /* :531 */
            double yp = ((combined_origin[1] + (double)y) * xy_res) - ego[1];         /* :532 */
            if (xp * xp + yp * yp <= radius * radius)                                 /* :533 */
                height_map[x * xy + y] = ego[2] - ground_to_lidar_height;
            for (int64_t z = 0; z < zs; ++z) {
                int32_t index = combined_index_map[x + y * xy + z * xy * xy];
                if (index >= 0) {                                                     /* :538 */
                    height_map[x * xy + y] =
                        (((double)min_height[index] + (double)z) + combined_origin[2]) * z_res;
                    break;
                }
            }
        }
}

/* gvom.py:544-554 __make_inferred_height_map */
ORC_API void orc_make_inferred_height_map(const double *combined_origin,
        const int32_t *combined_index_map, int64_t xy, int64_t zs, double z_res,
        double *inferred_height_map)
{
    ORC_PFOR
    for (int64_t x = 0; x < xy; ++x)
        for (int64_t y = 0; y < xy; ++y)
            for (int64_t z = 0; z < zs; ++z) {
                int32_t index = combined_index_map[x + y * xy + z * xy * xy];
                if (index < -1) {                                                     /* :551 */
                    inferred_height_map[x * xy + y] = ((double)z + combined_origin[2]) * z_res;
                    break;
                }
            }
}

/* gvom.py:665-734 __calculate_slope: 3x3 least-squares plane, f64, source order. */
ORC_API void orc_calculate_slope(const double *height_map, int64_t xy, double xy_res,
        double *slope_x, double *slope_y, double *roughness)
{
    const int64_t radius = 1;
    ORC_PFOR
    for (int64_t x0 = 0; x0 < xy; ++x0)
        for (int64_t y0 = 0; y0 < xy; ++y0) {
            int64_t xlo = x0 - radius < 0 ? 0 : x0 - radius, xhi = x0 + radius + 1 > xy ? xy : x0 + radius + 1;
            int64_t ylo = y0 - radius < 0 ? 0 : y0 - radius, yhi = y0 + radius + 1 > xy ? xy : y0 + radius + 1;
            int n_good = 0;
            for (int64_t x = xlo; x < xhi; ++x)
                for (int64_t y = ylo; y < yhi; ++y)
                    if (height_map[x * xy + y] > -1000) ++n_good;                     /* :674 */
            if (n_good < 3) continue;                                                 /* :676 */
            double p0[9], p1[9], p2[9];
            int i = 0;
            double mean_x = 0.0, mean_y = 0.0, mean_z = 0.0;
            for (int64_t x = xlo; x < xhi; ++x)
                for (int64_t y = ylo; y < yhi; ++y)
                    if (height_map[x * xy + y] > -1000) {
                        p0[i] = (double)x * xy_res;                                   /* :687 */
                        p1[i] = (double)y * xy_res;
                        p2[i] = height_map[x * xy + y];
                        mean_x += p0[i]; mean_y += p1[i]; mean_z += p2[i];
                        ++i;
                    }
            mean_x /= (double)i; mean_y /= (double)i; mean_z /= (double)i;            /* :695 */
            double xx = 0.0, xyv = 0.0, xz = 0.0, yy = 0.0, yz = 0.0;
            for (int k = 0; k < n_good; ++k) {                                        /* :704 */
                xx += (p0[k] - mean_x) * (p0[k] - mean_x);
                xyv += (p0[k] - mean_x) * (p1[k] - mean_y);
                xz += (p0[k] - mean_x) * (p2[k] - mean_z);
                yy += (p1[k] - mean_y) * (p1[k] - mean_y);
                yz += (p1[k] - mean_y) * (p2[k] - mean_z);
            }
            double det = xx * yy - xyv * xyv;                                         /* :711 */
            if (det == 0.0) continue;
            double a0 = (yy * xz - xyv * yz) / det;                                   /* :715 */
            double a1 = (xx * yz - xyv * xz) / det;
            double m = sqrt((a0 * a0 + a1 * a1) + 1.0);                               /* :717 */
            a0 /= m; a1 /= m;
            double error = 0.0;
            for (int k = 0; k < n_good; ++k) {                                        /* :722 */
                double e = (p2[k] - mean_z) - (a0 * (p0[k] - mean_x) + a1 * (p1[k] - mean_y));
                error += e * e;
            }
            error /= (double)n_good;
            if (error > 0) error = log(error);                                        /* :727 */
            roughness[x0 * xy + y0] = error;
            slope_x[x0 * xy + y0] = atan2(a0, 1.0 / m);                               /* :731 */
            slope_y[x0 * xy + y0] = atan2(a1, 1.0 / m);
        }
}

/* gvom.py:558-661 __guess_height, including the loop-guard typo (:581: x_n_done twice)
 * and the y_nh guard typo (:655 tests x_nh). */
ORC_API void orc_guess_height(const double *height_map, const double *inferred_height_map,
        int64_t xy, double *guessed_height_delta)
{
    ORC_PFOR
    for (int64_t x0 = 0; x0 < xy; ++x0)
        for (int64_t y0 = 0; y0 < xy; ++y0) {
            if (height_map[x0 * xy + y0] > -1000 || inferred_height_map[x0 * xy + y0] == -1000.0)
                continue;                                                             /* :563 */
            int x_p_done = 0, x_n_done = 0, y_p_done = 0, y_n_done = 0;
            int64_t x_p = x0, x_n = x0, y_p = y0, y_n = y0;
            double x_ph = -1000, x_nh = -1000, y_ph = -1000, y_nh = -1000;
            int64_t i = 0;
            while (i < 15 && !(x_n_done && x_n_done && y_p_done && y_n_done)) {       /* :581 */
                x_p += 1; x_n -= 1; y_p += 1; y_n -= 1; i += 1;
                if (!x_p_done) {
                    if (x_p < xy) {
                        for (int64_t dy = -i; dy < i; ++dy) {                         /* :590 */
                            if (y0 + dy >= xy || y0 + dy < 0) continue;
                            if (height_map[x_p * xy + y0 + dy] > -1000) {
                                x_ph = height_map[x_p * xy + y0 + dy]; x_p_done = 1; break;
                            }
                        }
                    } else x_p_done = 1;
                }
                if (!x_n_done) {
                    if (x_n >= 0) {
                        for (int64_t dy = -i + 1; dy < i + 1; ++dy) {                 /* :603 */
                            if (y0 + dy >= xy || y0 + dy < 0) continue;
                            if (height_map[x_n * xy + y0 + dy] > -1000) {
                                x_nh = height_map[x_n * xy + y0 + dy]; x_n_done = 1; break;
                            }
                        }
                    } else x_n_done = 1;
                }
                if (!y_p_done) {
                    if (y_p < xy) {
                        for (int64_t dx = -i + 1; dx < i + 1; ++dx) {                 /* :616 */
                            if (x0 + dx >= xy || x0 + dx < 0) continue;
                            if (height_map[(x0 + dx) * xy + y_p] > -1000) {
                                y_ph = height_map[(x0 + dx) * xy + y_p]; y_p_done = 1; break;
                            }
                        }
                    } else y_p_done = 1;
                }
                if (!y_n_done) {
                    if (y_n >= 0) {
                        for (int64_t dx = -i; dx < i; ++dx) {                         /* :629 */
                            if (x0 + dx >= xy || x0 + dx < 0) continue;
                            if (height_map[(x0 + dx) * xy + y_n] > -1000) {
                                y_nh = height_map[(x0 + dx) * xy + y_n]; y_n_done = 1; break;
                            }
                        }
                    } else y_n_done = 1;
                }
            }
            double min_h = 1000.0;                                                    /* :640 */
            double max_h = inferred_height_map[x0 * xy + y0];
            if (x_ph > -1000) { min_h = py_mind(x_ph, min_h); max_h = py_maxd(x_ph, max_h); }
            if (x_nh > -1000) { min_h = py_mind(x_nh, min_h); max_h = py_maxd(x_nh, max_h); }
            if (y_ph > -1000) { min_h = py_mind(y_ph, min_h); max_h = py_maxd(y_ph, max_h); }
            if (x_nh > -1000) { min_h = py_mind(y_nh, min_h); max_h = py_maxd(y_nh, max_h); } /* :655 */
            double dh = max_h - min_h;
            if (dh > 0) guessed_height_delta[x0 * xy + y0] = dh;                      /* :660 */
        }
}

/* gvom.py:489-521 __make_positive_obstacle_map */
ORC_API void orc_make_positive_obstacle_map(const int32_t *combined_index_map,
        const double *height_map, int64_t xy, int64_t zs, double z_res,
        double positive_obstacle_threshold, const int32_t *hit_count, const int32_t *total_count,
        double robot_height, const double *origin, const double *x_slope, const double *y_slope,
        double slope_threshold, int32_t *obstacle_map)
{
    ORC_PFOR
    for (int64_t x = 0; x < xy; ++x)
        for (int64_t y = 0; y < xy; ++y) {
            double sx = x_slope[x * xy + y], sy = y_slope[x * xy + y];
            if (sqrt(sx * sx + sy * sy) >= slope_threshold) {                         /* :498 */
                obstacle_map[x * xy + y] = 100;
                continue;
            }
            double min_obs_height = height_map[x * xy + y] + positive_obstacle_threshold;
            double fmin = floor((min_obs_height / z_res) - origin[2]) + 1.0;          /* :503 */
            double max_obs_height = height_map[x * xy + y] + robot_height;
            double fmax = floor((max_obs_height / z_res) - origin[2]);                /* :505 */
            if (!(fmin >= 0 && fmin < (double)zs)) continue;                          /* :506 */
            if (!(fmax >= 0 && fmax < (double)zs)) continue;                          /* :508 */
            int64_t zmin = (int64_t)fmin, zmax = (int64_t)fmax;
            double density = 0.0, n = 0.0;
            for (int64_t z = zmin; z < zmax + 1; ++z) {                               /* :513 */
                int32_t index = combined_index_map[x + y * xy + z * xy * xy];
                if (index >= 0 && hit_count[index] > 10) {                            /* :515 */
                    n += (double)total_count[index];
                    density += (double)hit_count[index];
                }
            }
            if (n > 0.0) density /= n;
            obstacle_map[x * xy + y] = (int32_t)(density * 100);                      /* :521 */
        }
}

/* gvom.py:479-485 __make_negative_obstacle_map */
ORC_API void orc_make_negative_obstacle_map(const double *guessed_height_delta,
        int32_t *negative_obstacle_map, double negative_obstacle_threshold, int64_t xy)
{
    for (int64_t i = 0; i < xy * xy; ++i)
        if (guessed_height_delta[i] > negative_obstacle_threshold) negative_obstacle_map[i] = 100;
}

/* gvom.py:414-422 __make_visibility_map */
ORC_API void orc_make_visibility_map(int32_t *visibility, const double *height_map, int64_t xy)
{
    for (int64_t i = 0; i < xy * xy; ++i) visibility[i] = height_map[i] > -1000 ? 1 : 0;
}

/* ------------------------------------------------------------------------------------
 * "Next" rows: debug/accessor gathers.
 * ---------------------------------------------------------------------------------- */

/* gvom.py:426-438 __make_height_map_pointcloud -> f32[xy*xy, 7] */
ORC_API void orc_make_height_map_pointcloud(const double *height_map, const double *roughness,
        const double *x_slope, const double *y_slope, const double *origin, float *out,
        int64_t xy, double xy_res, double z_res)
{
    ORC_PFOR
    for (int64_t x = 0; x < xy; ++x)
        for (int64_t y = 0; y < xy; ++y) {
            int64_t index = x + y * xy;
            double sx = x_slope[x * xy + y], sy = y_slope[x * xy + y];
            out[index * 7 + 0] = (float)(((double)x + origin[0]) * xy_res);
            out[index * 7 + 1] = (float)(((double)y + origin[1]) * xy_res);
            out[index * 7 + 2] = (float)(height_map[x * xy + y] - z_res);
            out[index * 7 + 3] = (float)roughness[x * xy + y];
            out[index * 7 + 4] = (float)sx;
            out[index * 7 + 5] = (float)sy;
            out[index * 7 + 6] = (float)sqrt(sx * sx + sy * sy);
        }
}

/* gvom.py:442-450 __make_infered_height_map_pointcloud -> f32[xy*xy, 3] (fed with
 * guessed_height_delta by gvom.py:407) */
ORC_API void orc_make_inferred_height_map_pointcloud(const double *map, const double *origin,
        float *out, int64_t xy, double xy_res, double z_res)
{
    ORC_PFOR
    for (int64_t x = 0; x < xy; ++x)
        for (int64_t y = 0; y < xy; ++y) {
            int64_t index = x + y * xy;
            out[index * 3 + 0] = (float)(((double)x + origin[0]) * xy_res);
            out[index * 3 + 1] = (float)(((double)y + origin[1]) * xy_res);
            out[index * 3 + 2] = (float)(map[x * xy + y] - z_res);
        }
}

/* ------------------------------------------------------------------------------------
 * "Next" rows: per-voxel statistics, their temporal merge, eigenvalues and
 * the debug voxel cloud.  Float accumulation order is unspecified on a GPU, so everything
 * below is compared to a tolerance, not bit for bit.
 * ---------------------------------------------------------------------------------- */

/* gvom.py:1224-1230 __normalize_mean and :1289-1299 __normalize_covariance */
ORC_API void orc_normalize_stats(int pass, double *metrics, int64_t cell_count)
{
    for (int64_t i = 0; i < cell_count; ++i) {
        double *m = metrics + i * 10;
        if (pass == 0) { m[0] /= m[9]; m[1] /= m[9]; m[2] /= m[9]; }
        else for (int j = 3; j < 9; ++j) m[j] = (m[9] <= 0) ? 0.0 : m[j] / m[9];
    }
}

/* gvom.py:858-909 pooled mean / covariance merge of one voxel.  The fused metrics are float32
 * (gvom.py:234); a ring slot's are float64 (gvom.py:1011), the previous fused map's float32.
 * Under the simulator's numpy scalar rules f32*f32 stays f32 and anything touching an f64
 * operand is f64 -- TO is the old map's type and selects exactly that arithmetic. */
#define GEN_MERGE(TO, SUF)                                                                      \
static void merge_metrics_##SUF(float *c, const TO *o)                                          \
{                                                                                               \
    typedef __typeof__((float)1 * (TO)1) W;          /* float or double */                       \
    const float c0 = c[0], c1 = c[1], c2 = c[2], c9 = c[9];                                     \
    const TO o0 = o[0], o1 = o[1], o2 = o[2], o9 = o[9];                                        \
    const W n = (W)c9 + (W)o9;                                                                  \
    const W mx = ((W)(c0 * c9) + (W)(o0 * o9)) / n;                                             \
    const W my = ((W)(c1 * c9) + (W)(o1 * o9)) / n;                                             \
    const W mz = ((W)(c2 * c9) + (W)(o2 * o9)) / n;                                             \
    const W cm[3] = {mx, my, mz};                                                               \
    const float cmean[3] = {c0, c1, c2};                                                        \
    const TO omean[3] = {o0, o1, o2};                                                           \
    static const int A[6] = {0, 0, 0, 1, 1, 2}, B[6] = {0, 1, 2, 1, 2, 2};                      \
    float out[6];                                                                               \
    for (int k = 0; k < 6; ++k) {                                                               \
        const int a = A[k], b = B[k];                                                           \
        W t = (W)(c9 * c[3 + k]) + (W)(o9 * o[3 + k]);                                          \
        t = t + ((W)c9 * ((W)cmean[a] - cm[a])) * ((W)cmean[b] - cm[b]);                        \
        t = t + ((W)o9 * ((W)omean[a] - cm[a])) * ((W)omean[b] - cm[b]);                        \
        out[k] = (float)(t / n);                                                                \
    }                                                                                           \
    for (int k = 0; k < 6; ++k) c[3 + k] = out[k];                                              \
    c[0] = (float)mx; c[1] = (float)my; c[2] = (float)mz;                                       \
    c[9] = (float)n;                                                                            \
}                                                                                               \
/* gvom.py:821-912 __combine_metrics, complete (covariance merge + lines 910-912) */           \
ORC_API void orc_combine_metrics_full_##SUF(float *combined_metrics, int32_t *combined_hit,     \
        int32_t *combined_total, float *combined_min_height, const int32_t *combined_index_map, \
        const double *combined_origin, const TO *old_metrics, const int32_t *old_hit,           \
        const int32_t *old_total, const float *old_min_height, const int32_t *old_index_map,    \
        const double *old_origin, int64_t xy, int64_t zs)                                       \
{                                                                                               \
    double d[3] = { combined_origin[0] - old_origin[0], combined_origin[1] - old_origin[1],     \
                    combined_origin[2] - old_origin[2] };                                       \
    for (int64_t z = 0; z < zs; ++z)                                                            \
        for (int64_t y = 0; y < xy; ++y)                                                        \
            for (int64_t x = 0; x < xy; ++x) {                                                  \
                int64_t io;                                                                     \
                if (!shifted_index(x, y, z, d, xy, zs, &io)) continue;                          \
                int32_t index = combined_index_map[x + y * xy + z * xy * xy];                   \
                int32_t index_old = old_index_map[io];                                          \
                if (index < 0 || index_old < 0) continue;                                       \
                merge_metrics_##SUF(combined_metrics + (int64_t)index * 10,                     \
                                    old_metrics + (int64_t)index_old * 10);                     \
                combined_hit[index] = combined_hit[index] + old_hit[index_old];                 \
                combined_total[index] = combined_total[index] + old_total[index_old];           \
                float a = combined_min_height[index], b = old_min_height[index_old];            \
                combined_min_height[index] = (b < a) ? b : a;                                   \
            }                                                                                   \
}
GEN_MERGE(double, f64)
GEN_MERGE(float, f32)

/* gvom.py:1333-1378 __calculate_eigenvalues: closed-form eigenvalues of the symmetric 3x3
 * covariance (trigonometric method); metrics float32[C][10] -> eigenvalues float32[C][3]. */
ORC_API void orc_calculate_eigenvalues(float *ev, const float *metrics, int64_t cell_count)
{
    const double PI = 3.141592653589793;
    for (int64_t i = 0; i < cell_count; ++i) {
        const float *m = metrics + i * 10;
        const float xx = m[3], xy = m[4], xz = m[5], yy = m[6], yz = m[7], zz = m[8];
        const float p1 = (float)((float)(xy * xy) + (float)(xz * xz)) + (float)(yz * yz);   /* f32 */
        const double q = (double)((float)((float)(xx + yy) + zz)) / 3.0;
        float *e = ev + i * 3;
        if (p1 == 0) {
            e[0] = py_maxf(xx, py_maxf(yy, zz));
            const float mn_yz = (zz < yy) ? zz : yy;          /* Python min(yy, zz) */
            e[2] = (mn_yz < xx) ? mn_yz : xx;                 /* Python min(xx, .) */
            e[1] = (float)((3.0 * q - (double)e[0]) - (double)e[2]);
        } else {
            const double p2 = ((((double)xx - q) * ((double)xx - q) + ((double)yy - q) * ((double)yy - q))
                               + ((double)zz - q) * ((double)zz - q)) + 2.0 * (double)p1;
            const double p = sqrt(p2 / 6.0);
            const double B0 = ((double)xx - q) / p, B1 = (double)xy / p, B2 = (double)xz / p;
            const double B3 = ((double)yy - q) / p, B4 = (double)yz / p, B5 = ((double)zz - q) / p;
            double r = (B0 * (B3 * B5 - B4 * B4) - B1 * (B1 * B5 - B4 * B2)) + B2 * (B1 * B4 - B3 * B2);
            r = r / 2;
            double phi;
            if (r <= -1) phi = PI / 3.0;
            else if (r >= 1) phi = 0.0;
            else phi = acos(r) / 3.0;
            e[0] = (float)(q + 2.0 * p * cos(phi));
            e[2] = (float)(q + 2.0 * p * cos(phi + (2.0 * PI / 3.0)));
            e[1] = (float)((3.0 * q - (double)e[0]) - (double)e[2]);
        }
    }
}

/* gvom.py:454-473 __make_voxel_pointcloud -> float32[Cc][8], one row per occupied fused voxel,
 * row = its compact index (order unspecified on a GPU; tests sort the rows). */
ORC_API void orc_make_voxel_pointcloud(const int32_t *combined_index_map, const int32_t *hit,
        const int32_t *total, const float *ev, const double *origin, float *out, int64_t xy,
        int64_t zs, double xy_res, double z_res)
{
    for (int64_t z = 0; z < zs; ++z)
        for (int64_t y = 0; y < xy; ++y)
            for (int64_t x = 0; x < xy; ++x) {
                const int32_t index = combined_index_map[x + y * xy + z * xy * xy];
                if (index < 0) continue;
                float *o = out + (int64_t)index * 8;
                o[0] = (float)(((double)x + origin[0]) * xy_res);
                o[1] = (float)(((double)y + origin[1]) * xy_res);
                o[2] = (float)(((double)z + origin[2]) * z_res);
                o[3] = (float)((double)hit[index] / (double)total[index]);
                o[4] = (float)hit[index];
                o[5] = ev[index * 3 + 0] - ev[index * 3 + 1];
                o[6] = ev[index * 3 + 1] - ev[index * 3 + 2];
                o[7] = ev[index * 3 + 2];
            }
}

ORC_API int orc_abi_version(void) { return 1; }
ORC_API void orc_set_cuda_f32_sqrt(int on) { orc_cuda_f32_sqrt = on ? 1 : 0; }

/* threads the all-core build runs on (1 in the one-thread build) */
/* the V-sized fills of gvom.py:114-121, 190-228 (__init_1D_array, gvom.py:1382-1394): value into n int32 words, on all threads
 * in the all-core build (numpy's np.full is one thread, and 67 MB of it per scan and per combine was most of a 128-thread step) */
ORC_API void orc_fill_i32(int32_t *dst, int64_t n, int32_t value)
{
    ORC_PFOR
    for (int64_t i = 0; i < n; ++i) dst[i] = value;
}

ORC_API int orc_threads(void)
{
#ifdef ORC_OMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
ORC_API void orc_set_threads(int n)
{
#ifdef ORC_OMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
