// gvom_internal.h -- shared between the kernels (gvom_kernels.hip) and the C-ABI host
// layer (gvom_capi.hip).  Not part of the public interface (include/gvom_hip.h is).
//
// STORAGE LAYOUT (DESIGN.md "Data layout in HBM")
//   Voxels are stored WORLD-ANCHORED (toroidal): the voxel with world index (xw, yw, zw)
//   lives at storage coordinates (sx, sy, sz) = (xw mod xy, yw mod xy, zw mod zs) and linear
//   index L = (sy * zs + sz) * xy + sx   -- x fastest (coalesced 64-lane rows), then z (a
//   whole column of one y-row is a contiguous xy*zs*4-byte block), y slowest (multi-GPU
//   slabs are contiguous).  Every per-voxel array of every scan and of the fused map uses
//   the same L for the same world voxel, so temporal fusion is element-wise and rows
//   never migrate between GPUs when the robot-centred window moves.
//   A map with integer origin o (window voxel (0,0,0) == world voxel o) converts window
//   coordinates with om = o mod size:  s = w + om (minus size if >= size).
//   2-D maps are stored in the same anchoring: [sy][sx] (x fastest); a rank's rows are one
//   contiguous block, so the height-map all-gather of a sharded run is a plain all_gather.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GVOM_MAX_SLOTS 64

// Tuning / diagnostic switches read on the hot path (every scan and combine).
// A dozen getenv() misses per step cost ~1.5 us of host time in front of the k_trace launch, so they
// are only looked up when the process environment held some GVOM_ switch the first time one was
// asked for (tools/ab_step.py, which changes them while running, sets GVOM_ENV_DYNAMIC=1 first).
#include <stdlib.h>
#include <string.h>
extern char **environ;
static inline const char *gvom_tune_env(const char *name)
{
    static int mode = 0;                                   // 1: nothing to look up; 2: getenv every time
    if (mode == 0) {
        int m = 1;
        for (char **e = environ; e && *e; ++e)
            if (!strncmp(*e, "GVOM_", 5) && strncmp(*e, "GVOM_HIP_LIBRARY=", 17) && strncmp(*e, "GVOM_BENCH_", 11) &&
                strncmp(*e, "GVOM_AMD_HOME=", 14)) { m = 2; break; }
        mode = m;
    }
    return mode == 2 ? getenv(name) : nullptr;
}

// TILES: 64 consecutive sx of one (sy, sz) row = 256 bytes of every per-voxel array.  Tile index
// T = (sy*zs + sz)*nseg + (sx >> 6), nseg = ceil(xy/64).  Every scan / fused map carries one
// uint32 tag per tile; a tile holds valid data iff tag == the map's epoch, otherwise all its
// voxels read as "never observed" (-1) and their stored bytes are stale and never read.  Epochs
// only grow, so tag arrays are never cleared.
// ACCUMULATOR LAYOUT (hit[], total[] only -- private to k_trace / k_encode): micro-tiled so that one
// 64-byte line holds a 4 (x) x 4 (y) patch of voxels at one z:
//   A(sx, sy, sz) = ((((sy >> 2) * zs + sz) * sxq + (sx >> 2)) << 4) + ((sy & 3) << 2) + (sx & 3)
// The hardware merges the lanes of one atomic instruction that fall into the same 64-B line into
// one memory-side request, and k_trace runs at that request rate; with rows of 16 consecutive x
// per line an x-dominant ray bundle (all lanes at the same x, spread over y) needed one request
// per distinct y.  Patches make x- and y-dominant bundles equally cheap (about half the requests).
// device counter block (uint32 words; 512 bytes).  The two counters k_trace adds to live on separate
// cache lines: same-line atomics are serialised by the memory system.
#define GVOM_CNT_ROWS 0        // compact rows claimed by the scan in flight
#define GVOM_CNT_INGRID 64     // in-grid returns of the scan in flight (any rank's rows)
#define GVOM_CNT_VOTE 128     // order vote (k_encode): long rays of the first / second (+16) cloud half, two pairs (+32) by scan parity
#define GVOM_CNT_WORDS 192

struct ScanParams {
    double xy_res, z_res;
    double min_d2;        // min_distance * min_distance (f64 product, gvom.py:1067)
    double origin[3];     // window origin in voxels, integer valued (gvom.py:123-126)
    double tf[12];        // rows 0..2 of the 4x4 (gvom.py:1044-1052)
    float  pt0[3];        // (float)(ego / res)  (gvom.py:1097-1099)
    float  rinv[2];       // (float)(1 / xy_res), (float)(1 / z_res): k_trace's conservative early-exit estimate only
    int    has_tf;
    int    xy, zs;
    int    om[3];         // origin mod size (storage offset)
    int    sy_lo, sy_hi;  // storage rows owned by this rank: [sy_lo, sy_hi)
    int    nseg;          // tiles per (sy, sz) row
    int    nsegs, seg_len; // DDA steps are split into nsegs segments of seg_len steps (last: open-ended)
    int    seg_start[10];  // VAR 5/6: segment s covers steps (seg_start[s], seg_start[s+1]]; uniform = s * seg_len
    int    blk_reverse, nblk; // workgroup b of a segment handles returns [256 * (blk_reverse ? nblk-1-b : b), +256)
    unsigned long long seg_order; // nibble k = the segment handled by the workgroups with blockIdx.y == k
    int    sxq;           // accumulator layout: 4x4 (x,y) patches per row of patches = ceil(xy/4)
    uint32_t epoch;       // this scan's tile epoch
    // slab-sharded runs: the slab's rows as (up to two) intervals of WINDOW y, for ray culling
    int    off[3];        // element offsets of x, y, z inside a point record (0, 1, 2 unless PointCloud2 ingest)
    int    in_f32;        // 1: the records hold float32 fields that are widened to the (float64) compute type,
                          //    as ros_numpy hands the reference a float64 array (stride and offsets in 4-byte units)
    int    lc_period;     // k_trace line cache: flush every lc_period committing steps (GVOM_TRACE_PERIOD)
    int    dbg;           // GVOM_TRACE_DEBUG bits (timing experiments only; results wrong when set)
    int    cull;          // 1: skip rays that cannot reach the slab, stop rays that have left it
    int    wlo[2], whi[2];
};

struct MapDesc {          // one source map of the fusion (ring slot or previous fused map)
    const int32_t  *state;
    const uint32_t *hit;
    const uint32_t *total;
    const uint32_t *minh;   // float bits
    int d[3];               // fused origin - this map's origin (window shift), clamped
    uint32_t epoch;         // tile (T) of this map is live iff tags[T] == epoch
    const uint32_t *tags;
    const void *metrics;    // optional per-row statistics: double[rows][10] (ring slot) or float[rows][10] (fused)
};

#define GVOM_KARG_DESCS 17   // ring slots + previous map passed by kernel argument when they fit
struct FuseDescs { MapDesc d[GVOM_KARG_DESCS]; };

struct FuseParams {
    int xy, zs;
    int om[3];              // fused origin mod size
    int nslots;             // number of non-empty ring slots (descs[0..nslots))
    int has_prev;           // descs[nslots] is the previous fused map
    int sy_lo, sy_hi;
    int nseg;               // tiles per (sy, sz) row
    int hs;                 // row stride (elements) of the height / inferred-height maps
    uint32_t epoch;         // epoch of the fused map being written
    int nz;                 // z-chunks per workgroup (block = 64 * nz threads)
    int zc;                 // window-z cells per chunk (<= 64)
    int cpw;                // chunks per wave (a wave walks them in ascending z)
    int debug;              // GVOM_FUSE_DEBUG bits (timing experiments only): 1 no code stores, 2 no emit, 4 all tiles dead
    double origin[3];       // fused origin (voxels)
    double ego[3];          // latest ego (gvom.py:294-295)
    double xy_res, z_res;
    double radius2;         // robot_radius^2 (gvom.py:533)
    double ground_to_lidar_height;
};

struct Map2dParams {
    int xy, zs;
    int om[3];
    int y_lo, y_hi;         // STORAGE rows [sy] computed by this rank
    int nseg;
    int hs;                 // row stride (elements) of the height / inferred-height maps
    int gathered_pos;       // 1: positive-obstacle densities come from the gathered height buffer (sharded)
    uint32_t epoch;         // epoch of the fused map (tile liveness of fstate)
    int dbg;                // GVOM_MAP2D_DEBUG bits (timing experiments only)
    int out_yx;             // 1: returned maps in [y][x] memory order (column-major [x, y]); 0: row-major [x][y]
    double origin_z;        // fused origin z (voxels)
    double xy_res, z_res;
    double pos_thr, neg_thr, slope_thr, robot_height;
    int occ;                // 1: write the five int8 occupancy grids of gvom_ros.py:141-165 instead of the four maps
    double occ_density_thr, occ_min_rough, occ_max_rough;
};

// ---- launchers (gvom_kernels.hip) --------------------------------------------------------
hipError_t gvom_launch_trace(hipStream_t s, const ScanParams &P, int dtype, const void *pts,
                             int64_t stride_elems, int64_t n, void *world, uint32_t *hit,
                             uint32_t *total, int32_t *state, uint32_t *tags, uint32_t *cminh,
                             uint32_t *counters, int variant, double *stat_sums, double *stat_base,
                             uint32_t *stat_rowvox);
hipError_t gvom_launch_encode(hipStream_t s, const ScanParams &P, int dtype, const void *world,
                              int64_t n, uint32_t *hit, uint32_t *total, int32_t *state,
                              uint32_t *chit, uint32_t *ctotal, uint32_t *cminh, const uint32_t *tags,
                              uint32_t *counters, unsigned long long *host_flag, uint32_t seq);
hipError_t gvom_launch_fuse(hipStream_t s, const FuseParams &P, const FuseDescs &KD,
                            const MapDesc *descs_dev, int32_t *fstate, uint32_t *fhit, uint32_t *ftotal, uint32_t *fminh,
                            uint32_t *ftags, uint32_t *blockcounts, double *height, double *inferred);
hipError_t gvom_launch_map2d(hipStream_t s, const Map2dParams &P, const int32_t *fstate, const uint32_t *ftags,
                             const uint32_t *fhit, const uint32_t *ftotal, const double *height,
                             const double *inferred, double *slope_x, double *slope_y,
                             double *rough, double *guessed, int32_t *out_pos, int32_t *out_neg,
                             double *out_rough, int32_t *out_vis, const uint32_t *blockcounts,
                             int nblocks, unsigned long long *host_counter);
// ---- optional per-voxel statistics (gvom.py:1172-1299, 858-909, 1333-1378, 454-473)
hipError_t gvom_launch_stats(hipStream_t s, const ScanParams &P, int dtype, const void *world, int64_t n,
                             const int32_t *state, const uint32_t *tags, int xy_e, int z_e, double *base,
                             double *sums, const uint32_t *rowvox, const uint32_t *row_count_dev, int64_t cap);
hipError_t gvom_launch_fuse_stats(hipStream_t s, const FuseParams &P, const FuseDescs &KD, const MapDesc *descs_dev,
                                  const int32_t *fstate, const uint32_t *ftags, float *fmetrics);
hipError_t gvom_launch_voxel_cloud(hipStream_t s, const Map2dParams &P, double o0, double o1, double o2,
                                   const int32_t *fstate, const uint32_t *ftags, const uint32_t *fhit,
                                   const uint32_t *ftotal, const float *fmetrics, float *out, int64_t max_rows,
                                   unsigned long long *row_counter);
// test hooks / debug accessors
hipError_t gvom_launch_read_dense(hipStream_t s, int xy, int zs, const int om[3], int sy_lo, int sy_hi,
                                  const uint32_t *tags, uint32_t epoch, const int32_t *state, const uint32_t *chit,
                                  const uint32_t *ctotal, const uint32_t *cminh, int32_t *o_state,
                                  int32_t *o_hit, int32_t *o_total, float *o_minh);
// storage order [sy][sx] -> reference order [x][y] (window coordinates)
hipError_t gvom_launch_unwrap_f64(hipStream_t s, int xy, int om0, int om1, const double *in, int in_stride, double *out_xy);
hipError_t gvom_launch_posdens(hipStream_t s, const Map2dParams &P, const int32_t *fstate,
                               const uint32_t *ftags, const uint32_t *fhit, const uint32_t *ftotal,
                               double *hmaps, const uint32_t *blockcounts, int nblocks,
                               unsigned long long *host_counter, unsigned long long *dev_counter);
hipError_t gvom_launch_debug_height(hipStream_t s, int xy, int om0, int om1, const double origin[3], double xy_res,
                                    double z_res, const double *height, int hs, const double *rough,
                                    const double *sx, const double *sy, float *out7,
                                    const double *guessed, float *out3);
