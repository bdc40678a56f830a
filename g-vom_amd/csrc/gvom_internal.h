// gvom_internal.h -- shared between the kernels (gvom_trace / gvom_fuse / gvom_map2d / gvom_stats .hip) and the C-ABI host
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

// Timing experiments (parts of a kernel switched off; results are WRONG when set) exist only in the
// diagnostic build (make diag -> lib/libgvom_hip_diag.so, -DGVOM_DIAG): there the GVOM_*_DEBUG
// environment variables are read per call.  The production library compiles every such test away and
// reads no environment variable that can change a result.
#include <stdlib.h>
#include <string.h>
// TEST HOOKS (include/gvom_hip_test.h: the "epoch_bias" and "churn" knobs, the GVOM_TEST_IPC_REFUSE fault injector) exist only
// in lib/libgvom_hip_test.so (make test-lib, -DGVOM_TEST_HOOKS: the production kernels + the hooks) and in the diagnostic
// build; the production library does not contain them.
#if defined(GVOM_DIAG) || defined(GVOM_TEST_HOOKS)
#define GVOM_HOOKS 1
#endif
#ifdef GVOM_DIAG
#define GVOM_DBG(P, bits) ((P).dbg & (bits))
static inline int gvom_diag_env(const char *name) { const char *v = getenv(name); return v ? atoi(v) : 0; }
#else
#define GVOM_DBG(P, bits) 0
static inline int gvom_diag_env(const char *) { return 0; }
#endif

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
#define GVOM_CNT_INGRID 64     // != 0: some return of the scan in flight landed in the grid
#define GVOM_CNT_MAPDONE 32    // k_map2d: workgroups that have finished (the last one stores the combine's completion flag)
#define GVOM_CNT_WORDS 192

struct ScanParams {
    double xy_res, z_res;
    double min_d2;        // min_distance * min_distance (f64 product, gvom.py:1067)
    double origin[3];     // window origin in voxels, integer valued (gvom.py:123-126)
    double tf[12];        // rows 0..2 of the 4x4 (gvom.py:1044-1052)
    float  pt0[3];        // (float)(ego / res)  (gvom.py:1097-1099)
    float  rinv[2];       // (float)(1 / xy_res), (float)(1 / z_res): k_trace's conservative early-exit estimate only
    double drcp[2];       // RN(1 / xy_res), RN(1 / z_res) for div_by_res(): the EXACT quotient of a float32 coordinate without a divide
    int    fastdiv;       // bit 0 / 1: the host has verified drcp[0] / drcp[1] over all 2^23 float32 significands (gvom_create)
    float  win_lo[3], win_hi[3];   // k_trace: the window shrunk by 2 voxels, in voxel coordinates (origin + 2 .. origin + size - 2): a run
                          //   of steps between two positions inside it needs no window test.  lo > hi (never true) when |origin| >= 2^18
    int    has_tf;
    int    xy, zs;
    int    om[3];         // origin mod size (storage offset)
    int    sy_lo, sy_hi;  // storage rows ENCODED by this handle: [sy_lo, sy_hi) (a rank's slab; all rows otherwise)
    int    nseg;          // tiles per (sy, sz) row
    int    lg_nseg, lg_zs;  // log2 of nseg and zs when BOTH are powers of two, else -1 (k_encode decomposes quad numbers by shifts then)
    int    nsegs;         // k_trace: step segments s = (seg_start[s], seg_start[s+1]], the last one open-ended
    int    seg_start[10];
    int    ep_row;        // k_trace: blockIdx.y of the endpoint blocks (the other rows are the segments)
    int    prio_div;      // k_trace: issue priority of a wave = min(3, steps it still has to walk / prio_div); 0: hardware default
    int    lc_period;     // k_trace: flush the wave's line cache every lc_period committing steps
    int    ilv_lg;        // k_trace: log2 K of the sub-cloud interleave (0: lane l of bundle b takes return 64 b + l).  A cloud that is
                          //   K equally long sub-clouds behind one another (K sensors, K sweeps) whose returns of equal position
                          //   are neighbours in space: position p = 64 b + l takes return (p mod K) * ilv_len + p / K, so the K
                          //   neighbours sit in neighbouring lanes of ONE wave (merged steps, shared accumulator lines).  A
                          //   permutation of who traces which return: it cannot change a result.
    long   ilv_len;       // returns per sub-cloud (n / K; n % K == 0 or ilv_lg = 0)
    const uint32_t *perm; // k_trace: nullptr, or the DIRECTIONAL ORDER of an unordered cloud (k_dirbin_*): position p takes return perm[p].
                          //   A wave's 64 rays then point the same way (one of 1536 direction bins) and share accumulator lines; as with
                          //   the interleave, only WHO traces which return changes
    int    f32_sqrt;      // GVOM_FLAG_CUDA_F32_SQRT: ray_length = sqrtf(f32 sum) (real Numba-CUDA typing, gvom.py:1109)
    int    sxq;           // accumulator layout: 4x4 (x,y) patches per row of patches = ceil(xy/4) + padding
    uint32_t epoch;       // this scan's tile epoch
    int    off[3];        // element offsets of x, y, z inside a point record (0, 1, 2 unless PointCloud2 ingest)
    int    in_f32;        // 1: the records hold float32 fields that are widened to the (float64) compute type,
                          //    as ros_numpy hands the reference a float64 array (stride and offsets in 4-byte units)
    int    stat_e;        // xy_eigen_dist (sharded statistics: which ranks a return's neighbourhood reaches)
    int    shard_world, shard_rank, shard_rows;   // ranks of a sharded map (1, 0, xy when unsharded): rank r owns storage rows [r*shard_rows, (r+1)*shard_rows)
    int    dbg;           // diagnostic build only
    int    prof_on;       // diagnostic build only (GVOM_TRACE_STEPPROF): sampled waves record s_memtime stamps of their first 32 steps
    long   tl_words;      // diagnostic build only: words of per-wave records in tl; 8 summary words follow
    unsigned long long *tl;   // diagnostic build only (GVOM_TRACE_TIMELINE): 4 words per wave {start, set-up done, end, hardware id}
};

// rank exchange, k_trace side: endpoints in another rank's rows are appended to that rank's send list
struct ShardExchange {
    uint2    *ep_send;    // [world][ep_cap] {voxel L, min-height sample}
    uint32_t *ep_cnt;     // [world * 16] (one counter per 64-B line)
    long      ep_cap;
    // per-voxel statistics on a sharded map: a return goes (x, y, z in the cloud's type) to every OTHER rank that owns a
    // storage row of its (2 xy_eigen_dist + 1)-row neighbourhood -- at most two ranks (slabs are at least that high)
    void     *sp_send;    // [world][ep_cap] x 3 values; nullptr: statistics off
    uint32_t *sp_cnt;     // [world * 16]
};

#define GVOM_PACK_CHUNK 64     // quads per k_pack workgroup
#define GVOM_BASE_PITCH 16     // doubles per row of a scan's own-voxel moments (10 used): one 128-byte block per row
// rank exchange, owner side: received quads of all source ranks are unpacked by one launch
struct ShardUnpack { uint32_t q_off[GVOM_MAX_SLOTS + 1]; };    // q_off[s] = first quad (wave) of source s, q_off[world] = total

struct MapDesc {          // one source map of the fusion (ring slot or previous fused map)
    const int32_t  *state;
    const uint4 *rows;      // compact rows, 16 bytes each: {hit, total, min-height (float bits), 0} -- one line access per row
    int d[3];               // fused origin - this map's origin (window shift), clamped
    uint32_t epoch;         // tile (T) of this map is live iff tags[T] == epoch
    const uint32_t *tags;
    const void *metrics;    // optional per-row statistics: double[rows][10] (ring slot) or float[rows][10] (fused)
    union {
        const uint16_t *code16; // ring slots of xy % 4 == 0 grids: 16-bit codes (see k_encode); nullptr for the previous map
        const int32_t *link;    // k_fuse_stats, the PREVIOUS map of an eager fusion: link[fused row] = that voxel's row in the previous
                                //   map (< 0: it contributes nothing), written by k_encfuse -- the merge then reads neither the
                                //   previous map's states nor its tile tags (which the NEXT scan's k_encfuse overwrites)
    };
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
    int one_slot;           // 1: k_fuse1 (one ring slot + the previous map; nz <= 8 waves of cpw <= 4 chunks)
    int dbg;                // diagnostic build only (GVOM_FUSE_DEBUG): 1 no code stores, 2 no emit, 4 all tiles dead
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
    int dbg;                // diagnostic build only (GVOM_MAP2D_DEBUG)
    int out_yx;             // 1: returned maps in [y][x] memory order (column-major [x, y]); 0: row-major [x][y]
    double origin_z;        // fused origin z (voxels)
    double xy_res, z_res;
    double pos_thr, neg_thr, slope_thr, robot_height;
    int occ;                // 1: write the five int8 occupancy grids of gvom_ros.py:141-165 instead of the four maps
    double occ_density_thr, occ_min_rough, occ_max_rough;
    // completion flag of the synchronous combine, stored by the LAST workgroup to finish (no kernel behind k_map2d): every
    // wave waits for its own stores to be acknowledged, one lane per workgroup then counts itself in (device-scope atomic);
    // whoever sees the count complete stores done_seq into host-mapped memory and re-arms the counter
    unsigned long long *done_flag;   // nullptr: no flag
    uint32_t *done_count;
    uint32_t done_seq;
};

// ---- launchers (gvom_trace / _fuse / _map2d / _stats .hip) --------------------------------------------------------
hipError_t gvom_launch_trace(hipStream_t s, const ScanParams &P, const ShardExchange &X, int dtype, bool big_origin, const void *pts,
                             int64_t stride_elems, int64_t n, void *world, uint32_t *hit,
                             uint32_t *total, uint32_t *mh, int32_t *state, uint32_t *tags,
                             uint32_t *counters, double *stat_sums, double *stat_base,
                             uint32_t *stat_rowvox);
hipError_t gvom_launch_pack(hipStream_t s, const ScanParams &P, uint32_t *total, const uint32_t *tags, uint32_t *send_ids,
                            void *send_pay, uint32_t *qcnt, uint32_t *ecnt, uint32_t *spcnt, uint32_t *counters,
                            unsigned long long *host_out, uint32_t seq);
hipError_t gvom_launch_unpack(hipStream_t s, const ScanParams &P, const ShardUnpack &X, const uint32_t *ids_all,
                              const void *pay_all, uint32_t my_quads, uint32_t ne, const void *eps, long row_base,
                              uint32_t *hit, uint32_t *total, uint32_t *mh, int32_t *state, uint32_t *tags,
                              double *stat_sums, double *stat_base, uint32_t *stat_rowvox);
hipError_t gvom_launch_encode(hipStream_t s, const ScanParams &P, uint32_t *hit, uint32_t *total, uint32_t *mh,
                              int32_t *state, uint16_t *code16, uint4 *crows, const uint32_t *tags,
                              uint32_t *counters, unsigned long long *host_flag, uint32_t seq, unsigned resident_blocks);
hipError_t gvom_launch_layout_probe(hipStream_t s, const ScanParams &P, int dtype, const void *pts, int64_t stride_elems, int64_t n,
                                    int max_lg, unsigned long long *host_word);
hipError_t gvom_launch_publish_seq(hipStream_t s, unsigned long long *host_flag, uint32_t seq);
// directional order of an unordered cloud: keys[n] (scratch), hist / cursor: GVOM_DIRBINS counters each (hist zero on entry, zeroed
// again for the next use on exit: the caller alternates two), perm[n] out
#define GVOM_DIRBINS 8192      // mode 1: 6 cube faces x 16 x 16 (1536 used); mode 2: 256 sin(elevation) rows x 32 azimuth sectors
hipError_t gvom_launch_dirbin(hipStream_t s, const ScanParams &P, int mode, int dtype, const void *pts, int64_t stride_elems, int64_t n,
                              uint16_t *keys, uint32_t *hist, uint32_t *hist_next, uint32_t *cursor, uint32_t *perm);
hipError_t gvom_launch_fuse(hipStream_t s, const FuseParams &P, const FuseDescs &KD,
                            const MapDesc *descs_dev, int32_t *fstate, uint4 *frows,
                            uint32_t *ftags, uint32_t *blockcounts, double *height, double *inferred);
void gvom_encfuse_shape(int xy, int zs, int nw_override, int *nw, int *nblocks, size_t *row_cap);
// flink (or nullptr): per fused row, the voxel's row in the previous map -- what the statistics merge of this speculative fusion needs
hipError_t gvom_launch_encfuse(hipStream_t s, const ScanParams &P, const FuseParams &F, const MapDesc &prev, int32_t *flink, uint32_t *hit,
                               uint32_t *total, uint32_t *mh, int32_t *state, uint4 *crows, const uint32_t *stags,
                               int32_t *fstate, uint4 *frows, uint32_t *ftags, uint32_t *blockcounts, double *height,
                               double *inferred, uint32_t *counters, unsigned long long *host_flag, uint32_t seq);
hipError_t gvom_launch_map2d(hipStream_t s, const Map2dParams &P, const int32_t *fstate, const uint32_t *ftags,
                             const uint4 *frows, const double *height,
                             const double *inferred, double *slope_x, double *slope_y,
                             double *rough, double *guessed, int32_t *out_pos, int32_t *out_neg,
                             double *out_rough, int32_t *out_vis, const uint32_t *blockcounts,
                             int nblocks, unsigned long long *host_counter);
// ---- optional per-voxel statistics (gvom.py:1172-1299, 858-909, 1333-1378, 454-473)
// nrows: candidate compact rows (the scan's returns, + received endpoints on a sharded map); extra / n_extra: returns
// received from other ranks (sharded statistics), accumulated like the rank's own
hipError_t gvom_launch_stats(hipStream_t s, const ScanParams &P, int dtype, const void *world, int64_t n,
                             const int32_t *state, const uint32_t *tags, int xy_e, int z_e, double *base,
                             double *sums, const uint32_t *rowvox, int64_t nrows, const void *extra, int64_t n_extra);
hipError_t gvom_launch_fuse_stats(hipStream_t s, const FuseParams &P, const FuseDescs &KD, const MapDesc *descs_dev,
                                  const int32_t *fstate, const uint32_t *ftags, float *fmetrics);
hipError_t gvom_launch_voxel_cloud(hipStream_t s, const Map2dParams &P, double o0, double o1, double o2,
                                   const int32_t *fstate, const uint32_t *ftags, const uint4 *frows,
                                   const float *fmetrics, float *out, float *eig, int64_t max_rows,
                                   unsigned long long *row_counter);
hipError_t gvom_launch_gather_rows10(hipStream_t s, int is_f64, const void *src, const int32_t *rows, int64_t n, void *out);
hipError_t gvom_launch_retag(hipStream_t s, uint32_t *tags, size_t n, uint32_t old_epoch, uint32_t new_epoch);
// test hooks / debug accessors
hipError_t gvom_launch_read_dense(hipStream_t s, int xy, int zs, const int om[3], int sy_lo, int sy_hi,
                                  const uint32_t *tags, uint32_t epoch, const int32_t *state, const uint4 *crows,
                                  int32_t *o_state,
                                  int32_t *o_hit, int32_t *o_total, float *o_minh, int32_t *o_row);
// storage order [sy][sx] -> reference order [x][y] (window coordinates)
hipError_t gvom_launch_unwrap_f64(hipStream_t s, int xy, int om0, int om1, const double *in, int in_stride, double *out_xy);
hipError_t gvom_launch_posdens(hipStream_t s, const Map2dParams &P, const int32_t *fstate,
                               const uint32_t *ftags, const uint4 *frows,
                               double *hmaps, const uint32_t *blockcounts, int nblocks,
                               unsigned long long *host_counter, unsigned long long *dev_counter);
hipError_t gvom_launch_debug_height(hipStream_t s, int xy, int om0, int om1, const double origin[3], double xy_res,
                                    double z_res, const double *height, int hs, const double *rough,
                                    const double *sx, const double *sy, float *out7,
                                    const double *guessed, float *out3);
