/*
 * gvom_hip.h -- C ABI of libgvom_hip.so: the MI355X (gfx950) implementation of G-VOM's
 * process_pointcloud -> combine_maps hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own (it is a Python class
 * that launches Numba-CUDA kernels); each entry point below replaces one *method* of the
 * reference class `Gvom` (/root/reference/scripts/gvom.py, "gvom.py:NNN") and is bound
 * from Python with ctypes by g-vom_amd/gvom.py (see INTEGRATION.md for the stub).
 * Plain pointers and sizes only; no PyTorch / numpy types.  All functions are
 * thread-safe per handle (an internal mutex replaces the reference's semaphores,
 * gvom.py:65-67,96); ctypes drops the GIL around every call.
 *
 * Array conventions at this boundary are the REFERENCE's:
 *   voxel arrays : index = x + y*xy_size + z*xy_size*xy_size        (gvom.py:1086,1146)
 *   2-D maps     : [x][y] C-order, i.e. m[x*xy_size + y]            (gvom.py:288-349)
 * (internally the library stores voxels world-anchored/toroidal as [y][z][x] and 2-D
 * maps as [y][x]; see DESIGN.md.)
 */
#ifndef GVOM_HIP_H
#define GVOM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GVOM_ABI_VERSION 8   /* 8: RCCL loopback transport (GVOM_TRANSPORT_LOOPBACK), gvom_comm_wire_stats, gvom_comm_abort;
                              * 7: eager fusion of one-slot rings ("eager" knob, gvom_get_tuning "eager_adopted" / "eager_dropped"); a sharded scan /
                              *    combine as ONE native call (gvom_comm_process_pointcloud, gvom_comm_combine_maps_into);
                              * 6: sub-cloud interleave of the trace ("interleave" knob, automatic by a layout probe), gvom_get_tuning;
                              *    peer transport absorbs refused exports / imports (gvom_shard_renew_region, gvom_comm_peer_renewed), gvom_comm_info;
                              * 5: second transport between ranks (peer copies: gvom_comm_create2, gvom_comm_transport),
                              *    gvom_alloc_generation;
                              * 4: per-voxel statistics on sharded maps (gvom_shard_stats_*, gvom_comm_exchange_stats);
                              * 3: rank-exchange (shard) and communicator entry points, gvom_set_tuning, flags;
                              *    2: *_into outputs column-major; occupancy and PointCloud2 entry points */

/* return codes (>= 0: the reference's documented outcomes; < 0: failures) */
#define GVOM_OK                0
#define GVOM_EMPTY_CLOUD       1   /* gvom.py:107-109 "[WARNING] Processing an empty pointcloud..." */
#define GVOM_NO_OVERLAP        2   /* gvom.py:148-150 "[WARNING] The pointcloud points don't overlap..." */
#define GVOM_EMPTY_BUFFER      3   /* gvom.py:179-181 "[WARNING] The map buffer is empty..." */
#define GVOM_NO_DATA           4   /* gvom.py:364-366,381-383,397-399 "No data" */
#define GVOM_ERR_INVALID      -1   /* bad argument */
#define GVOM_ERR_HIP          -2   /* a HIP runtime call failed; see gvom_last_error() */
#define GVOM_ERR_NO_DEVICE    -3   /* no usable gfx950 device / library built without GPU */
#define GVOM_ERR_CAPACITY     -4   /* grid too large for 32-bit voxel indices, or > 64 ring slots */

#define GVOM_DTYPE_F32 0
#define GVOM_DTYPE_F64 1

typedef struct gvom_handle gvom_t;

/* The 14 positional constructor arguments of Gvom.__init__ (gvom.py:29-31), same order. */
typedef struct gvom_params {
    double  xy_resolution;
    double  z_resolution;
    int32_t xy_size;
    int32_t z_size;
    int32_t buffer_size;
    int32_t reserved0;             /* flags: GVOM_FLAG_* */
    double  min_distance;
    double  positive_obstacle_threshold;
    double  negative_obstacle_threshold;
    double  slope_obstacle_threshold;
    double  robot_height;
    double  robot_radius;
    double  ground_to_lidar_height;
    int32_t xy_eigen_dist;
    int32_t z_eigen_dist;
} gvom_params;

/* gvom_params.reserved0 flags */
#define GVOM_FLAG_VOXEL_STATISTICS 1   /* also run the per-voxel mean/covariance path (gvom.py:1172-1299,
                                        * 858-909, 1333-1378) that feeds only make_debug_voxel_map; off by
                                        * default: it is not on the north-star path and costs scan time.
                                        * The environment variable GVOM_VOXEL_STATISTICS=0/1 overrides. */
#define GVOM_FLAG_STATISTICS_ON_DEMAND 4 /* (unsharded handles, ignored with GVOM_FLAG_VOXEL_STATISTICS) the per-voxel path runs
                                        * WHILE SOMEBODY READS IT: it starts on -- the reference computes it in every scan and
                                        * combine and its node reads it every tick (gvom_ros.py:171) -- goes off when three
                                        * combines in a row passed without a call of gvom_debug_voxel_map / _eigen /
                                        * gvom_gather_metrics (its buffers go back to the allocator then: 40 bytes per voxel and
                                        * fused map alone), and comes back with the next such call: that call returns
                                        * GVOM_NO_DATA, the scans that follow carry statistics again, and the fused map has them
                                        * once every ring slot does (they then restart from the ring: what the map had merged
                                        * before the pause is not in them).  g-vom_amd/gvom.py's default. */
#define GVOM_FLAG_NUMBA_CUDA_TYPING 2  /* the types Numba 0.54.1 infers for a REAL CUDA device where they differ from its simulator's
                                        * (profiles/numba_cuda_typing.txt: its type inference with the CUDA target's typing context
                                        * over gvom.py:1060-1150, 1303-1329): ray_length = math.sqrt(float32) is
                                        * float32 (gvom.py:1109), slope[k] / ray_length is float32 / float32 (:1112-1114) and the loop
                                        * bound is that float32 minus 1 in float64 (:1127) -- every other difference (math.floor
                                        * giving float64, the voxel index carried as float64) is value-neutral.  Default (flag
                                        * clear): the float64 square root of Numba's simulator, which is what the golden fixtures
                                        * were generated with.  What NO flag reproduces: NVVM's default contraction of a*b + c into
                                        * FMAs on a real device (code generation, not typing; unpinnable without one). */
#define GVOM_FLAG_CUDA_F32_SQRT    GVOM_FLAG_NUMBA_CUDA_TYPING   /* (its name before ABI 8) */

/* Ring-buffer bookkeeping visible on the reference object (gvom.py:56-58,172-175). */
typedef struct gvom_state {
    int32_t buffer_index;
    int32_t last_buffer_index;
    int32_t has_combined;          /* combine_maps has produced a fused map at least once */
    int32_t reserved0;
    int64_t combined_cell_count;   /* gvom.py:217 combined_cell_count_cpu (valid if has_combined) */
    double  combined_origin[3];    /* gvom.py:184 (voxel units, integer valued) */
    double  ego_position[3];       /* gvom.py:102-104 latest ego */
} gvom_state;

/* Exact integer accounting of the last accepted scan (used for roofline arithmetic). */
typedef struct gvom_scan_stats {
    int64_t points;                /* N */
    int64_t cells;                 /* C: occupied voxels of the scan (gvom.py:147 cell_count_cpu) */
    int64_t sum_hit;               /* sum of hit over the scan's voxels  */
    int64_t sum_total;             /* sum of total over the scan's voxels (endpoint adds included) */
} gvom_scan_stats;

/* --- lifetime: replaces Gvom.__init__ (gvom.py:29-97) ------------------------------------ */
int  gvom_create(const gvom_params *params, int device_id, gvom_t **out);
/* One rank of a map sharded over `world` GPUs (see "one map sharded over the GPUs of a node" below). */
int  gvom_create_sharded(const gvom_params *params, int device_id, int rank, int world,
                         gvom_t **out);
void gvom_destroy(gvom_t *h);

/* --- Gvom.process_pointcloud (gvom.py:99-175) ---------------------------------------------
 * xyz: N rows of >= 3 consecutive float32/float64 (row_stride_bytes apart), host memory for
 * gvom_process_pointcloud, device (HBM) memory for the *_device variant.  The cloud is never
 * modified.  transform: row-major 4x4 double or NULL (gvom.py:134-135).
 * Returns GVOM_OK / GVOM_EMPTY_CLOUD / GVOM_NO_OVERLAP (ring untouched, ego still updated). */
int gvom_process_pointcloud(gvom_t *h, const void *xyz, int64_t n, int64_t row_stride_bytes,
                            int dtype, const double ego[3], const double *transform_4x4);
int gvom_process_pointcloud_device(gvom_t *h, const void *xyz_dev, int64_t n,
                                   int64_t row_stride_bytes, int dtype, const double ego[3],
                                   const double *transform_4x4);

/* Ingest side of the ROS node (gvom_ros.py:93-109): scans the packed bytes of a
 * sensor_msgs/PointCloud2 directly -- `data` holds n_points (= width*height, no row padding)
 * records of point_step bytes with little-endian fields x, y, z of type `dtype` (GVOM_DTYPE_F32 =
 * PointField.FLOAT32, GVOM_DTYPE_F64 = FLOAT64) at byte offsets off_x/off_y/off_z (multiples of
 * the field size).  Replaces ros_numpy.point_cloud2.pointcloud2_to_xyz_array + process_pointcloud;
 * records with a non-finite coordinate (which ros_numpy removes) have no effect on the map.
 * Same return codes as gvom_process_pointcloud. */
int gvom_process_pointcloud2(gvom_t *h, const void *data, int64_t n_points, int64_t point_step,
                             int64_t off_x, int64_t off_y, int64_t off_z, int dtype,
                             const double ego[3], const double *transform_4x4);

/* --- Gvom.combine_maps (gvom.py:177-354) --------------------------------------------------
 * Caller-allocated xy_size*xy_size outputs (any of them may be NULL to skip its copy).
 * Returns GVOM_OK or GVOM_EMPTY_BUFFER. */
/* Note on very long runs: the fused map carries its predecessor's free (ray-pass) counts along (gvom.py:996); they
 * are held as -count - 1 in int32 states and stop at 2^30 here (the reference's wrap into the row-index range). */
int gvom_combine_maps(gvom_t *h, double origin_world[3], int32_t *positive, int32_t *negative,
                      double *roughness, int32_t *visibility);

/* Zero-copy variant of combine_maps: the four maps are written by the GPU straight into a
 * pinned, device-mapped host buffer obtained from gvom_output_buffer_alloc (20*xy*xy bytes:
 * [positive i32 | negative i32 | visibility i32 | roughness f64]).  Unlike gvom_combine_maps,
 * each map is stored COLUMN-MAJOR: cell (x, y) at m[y*xy_size + x] -- the Fortran-ordered form of
 * the reference's [x, y]-indexed arrays, which is what gvom_ros.py:141-162 reads
 * (np.reshape(map, -1, order='F')); the GPU writes it as contiguous runs without a transpose and
 * streams each map out as soon as it is known.  The caller owns the buffer (and may keep several alive) until gvom_output_buffer_free.
 * COHERENCE: the synchronous calls learn of completion from a flag k_map2d's last workgroup stores behind its maps (every wave waits
 * for its stores to be acknowledged first), not from a stream synchronisation.  That is sound for COHERENT (fine-grained) pinned
 * memory, which is what gvom_output_buffer_alloc returns (hipHostMallocMapped | hipHostMallocCoherent): system-scope stores are
 * written through.  A buffer of the caller's own must be allocated the same way; gvom_combine_maps_into /
 * gvom_combine_occupancy_into / gvom_combine_begin / gvom_combine_map2d_into (and gvom_comm_combine_maps_into through it)
 * refuse (GVOM_ERR_INVALID) a pinned buffer whose flags say otherwise. */
int gvom_output_buffer_alloc(gvom_t *h, void **host_ptr);
int gvom_output_buffer_free(gvom_t *h, void *host_ptr);
int gvom_combine_maps_into(gvom_t *h, double origin_world[3], void *pinned_out);

/* combine_maps fused with the ROS node's post-processing (gvom_ros.py:141-165):
 * advances the fusion exactly like gvom_combine_maps, but the GPU writes the five int8
 * nav_msgs/OccupancyGrid.data arrays the node publishes into the pinned buffer (from
 * gvom_output_buffer_alloc), as planes of xy*xy bytes, cell (x, y) at [y*xy_size + x]:
 *   0 hard obstacles   max(100*(positive > density_threshold), negative)          :141
 *   1 soft obstacles   100*(positive <= density_threshold)*(positive > 0)         :146
 *   2 ground certainty visibility*100                                             :151
 *   3 negative         negative                                                   :157
 *   4 roughness        ((clip(r, min, max) + min)/(max - min))*100, cast to int8 as numpy does  :162-163
 * (reproduced as written, including the "+ min" and the wrapping cast). */
int gvom_combine_occupancy_into(gvom_t *h, double origin_world[3], void *pinned_out,
                                double density_threshold, double min_roughness, double max_roughness);

/* Asynchronous combine (an extension: the reference's combine_maps, gvom.py:177-354, is synchronous).
 * gvom_combine_begin enqueues what gvom_combine_maps_into (occ == NULL) or gvom_combine_occupancy_into
 * (occ = {density_threshold, min_roughness, max_roughness}) computes and returns without waiting;
 * gvom_combine_end waits for the maps and completes the call (the fused cell count, origin_world).
 * Between the two the caller may hand the NEXT scan to gvom_process_pointcloud*: its kernels run while
 * the maps of this combine are stored to host memory (k_map2d runs on a second stream; the next fusion
 * waits for it on the device).  `pinned_out` must not be read before gvom_combine_end has returned.
 * One combine may be pending per handle; the synchronous combine entry points return GVOM_ERR_INVALID
 * while one is (and gvom_combine_begin does while another thread waits inside a synchronous combine;
 * synchronous combines of several threads simply queue up).  Results are those of the synchronous calls. */
int gvom_combine_begin(gvom_t *h, void *pinned_out, const double *occ);
int gvom_combine_end(gvom_t *h, double origin_world[3]);

/* --- one map sharded over the GPUs of a node (one rank = one process = one GPU) -------------------
 * No counterpart in the reference (it has no multi-GPU path); semantics = SURVEY 8(e):
 * the rays are data-parallel, the per-voxel accumulators (hit / total: int32 sum, min-height: f32 min)
 * are reduced onto the rank that owns the voxel's storage row, everything after is per voxel / per
 * column on the owner.  Rank r owns storage rows [r*xy/world, (r+1)*xy/world) of the world-anchored y
 * axis (xy must be a multiple of 4*world).  The result is bit-identical to one GPU fed with the
 * concatenated cloud.
 *
 * Per scan:  gvom_shard_scan_local   trace this rank's share (it may be empty) over the whole window;
 *                                    returns, per owner rank d, how many dirty quads (4 rows x 64 sx at
 *                                    one sz = 1 KiB of ray-pass counts + a 4-byte id) and endpoints
 *                                    ({voxel, min-height sample}, 8 bytes) are packed for d
 *            -- the transport exchanges the counts, then moves SEND regions to the owners' RECV
 *               regions (gvom_shard_buffer; RCCL: gvom_comm_exchange_scan) --
 *            gvom_shard_recv_reserve  size the endpoint receive regions from the counts
 *            gvom_shard_scan_merge    add the received contributions, encode this rank's rows, commit
 *                                     iff `accept` (any rank saw an in-grid return: gvom.py:147-150)
 * Per combine: gvom_combine_fuse (fusion + column reductions + positive-obstacle densities of this
 *            rank's rows) -> all-gather of GVOM_BUF_HEIGHT_MAPS rows -> gvom_combine_map2d_into (all rows
 *            of slope / roughness / guess / positive / negative / visibility on every rank).
 * gvom_process_pointcloud* / gvom_combine_maps* return GVOM_ERR_INVALID on a sharded handle. */
int gvom_shard_scan_local(gvom_t *h, const void *xyz, int on_device, int64_t n, int64_t row_stride_bytes,
                          int dtype, const double ego[3], const double *transform_4x4,
                          int64_t *send_quads, int64_t *send_eps, int *any_ingrid);
#define GVOM_XBUF_SEND_IDS   0   /* uint32 quad ids for rank `peer`                      */
#define GVOM_XBUF_SEND_QUADS 1   /* 1 KiB per quad, same order                          */
#define GVOM_XBUF_SEND_EPS   2   /* {uint32 voxel, uint32 min-height sample} per endpoint */
#define GVOM_XBUF_RECV_IDS   3
#define GVOM_XBUF_RECV_QUADS 4
#define GVOM_XBUF_RECV_EPS   5   /* valid after gvom_shard_recv_reserve                  */
#define GVOM_XBUF_SEND_RETURNS 6 /* statistics handles: returns {x, y, z} of the cloud's type for rank `peer` */
#define GVOM_XBUF_RECV_RETURNS 7 /* valid after gvom_shard_stats_reserve                 */
int gvom_shard_buffer(gvom_t *h, int which, int peer, void **ptr, int64_t *capacity_bytes);
int gvom_shard_recv_reserve(gvom_t *h, const int64_t *recv_eps);
int gvom_shard_scan_merge(gvom_t *h, const int64_t *recv_quads, const int64_t *recv_eps, int accept);
/* Per-voxel statistics on a sharded map (GVOM_FLAG_VOXEL_STATISTICS; 2*xy_eigen_dist + 1 <= rows per rank): a return adds
 * to every occupied voxel of its neighbourhood (gvom.py:1188-1220), so besides the endpoints every rank gets the returns
 * whose neighbourhood reaches into its rows: send_returns[d] after gvom_shard_scan_local, GVOM_XBUF_SEND_RETURNS ->
 * the peers' GVOM_XBUF_RECV_RETURNS (gvom_comm_exchange_stats) sized by gvom_shard_stats_reserve, before
 * gvom_shard_scan_merge.  All ranks must pass clouds of one type (float32 or float64). */
int gvom_shard_stats_counts(gvom_t *h, int64_t *send_returns);
int gvom_shard_stats_reserve(gvom_t *h, const int64_t *recv_returns, int dtype /* GVOM_DTYPE_*: the scan's cloud type */);
/* For the transport: region `which` (GVOM_XBUF_SEND_*, or -1 = GVOM_BUF_HEIGHT_MAPS) moves, contents included, into a FRESH
 * allocation of the same size (its gvom_region_generation changes); the old allocation is parked, never freed while the
 * process lives (another process may have it mapped). */
int gvom_shard_renew_region(gvom_t *h, int which);
int gvom_combine_fuse(gvom_t *h, int64_t *local_cells);
int gvom_set_combined_cell_count(gvom_t *h, int64_t global_cells);
#define GVOM_BUF_HEIGHT_MAPS  0   /* [sy][height row | inferred-height row | positive-density row], f64 */
#define GVOM_BUF_FUSED_CELLS  3   /* one int64: occupied voxels of this rank's rows of the fused map */
int gvom_sync(gvom_t *h);
int gvom_device_buffer(gvom_t *h, int which, void **ptr, int64_t *bytes, int64_t *row_stride_bytes);
int gvom_combine_map2d_into(gvom_t *h, double origin_world[3], void *pinned_out);

/* --- transport between the ranks of a sharded map: RCCL over xGMI, bound directly ------------------
 * name: the same string on every rank and unique to this communicator on the node (rank 0 creates
 * /dev/shm/<name> for the ncclUniqueId and the small host-side exchanges).  gvom_comm_exchange_host:
 * all[r*k + j] = rank r's mine[j] (k <= 208 = 3 * 64 ranks + 16).  gvom_comm_exchange_scan / gvom_comm_allgather_rows run
 * on the handle's stream and do not synchronise.  device < 0: host-only communicator (rendezvous +
 * gvom_comm_exchange_host / gvom_comm_barrier, no RCCL and no HIP call; the device collectives return
 * GVOM_ERR_INVALID) -- the CPU tests run the multi-process rendezvous with it. */
typedef struct gvom_comm gvom_comm_t;
int  gvom_comm_create(int rank, int world, int device, const char *name, gvom_comm_t **out);   /* = create2(..., GVOM_TRANSPORT_RCCL) */
/* Transports for the DEVICE data (the host-side vectors always travel through the shared-memory segment):
 * RCCL  grouped ncclSend / ncclRecv and an in-place ncclAllGather on the handle's stream, no host synchronisation;
 * PEER  peer copies: every rank exports its send regions (hipIpcGetMemHandle), the receiver maps them (lazy peer
 *       access) and pulls its bytes with hipMemcpyAsync on its own handle's stream, bracketed by two host barriers --
 *       xGMI between the GPUs of a node, plain device copies when several ranks share ONE GPU (which RCCL refuses:
 *       this is the transport a one-GPU box can run several rank processes with);
 * AUTO  RCCL; if librccl cannot be loaded, or ncclCommInitRank fails on any rank or does not return within
 *       GVOM_RCCL_INIT_TIMEOUT_S (default 90 s after the last rank has arrived), every rank uses PEER.
 * gvom_comm_transport: the transport in use (GVOM_TRANSPORT_RCCL, GVOM_TRANSPORT_PEER or GVOM_TRANSPORT_LOOPBACK).
 * Failure semantics: a rank whose device exchange failed marks the communicator (for every rank) as broken, and a rank
 * whose process has gone is noticed by whoever waits for it next: the others' next gvom_comm_exchange_host / _barrier
 * returns GVOM_ERR_HIP with a message naming the rank instead of waiting GVOM_COMM_TIMEOUT_S.  A broken communicator stays
 * broken: destroy it. */
#define GVOM_TRANSPORT_RCCL 0
#define GVOM_TRANSPORT_PEER 1
#define GVOM_TRANSPORT_AUTO 2
/* LOOPBACK  RCCL on a box with ONE GPU: the ranks are threads of one process that share the device (RCCL refuses two ranks of
 *       one communicator on one device), each with a 1-rank communicator of its own.  What RCCL moves with ncclSend on the
 *       sender and ncclRecv on the receiver, the RECEIVER moves with ncclSend(the peer's send region, itself) +
 *       ncclRecv(its receive region, itself) in one group on its handle's stream -- same group handling, capacity checks,
 *       ncclUint8 byte counts and position in front of the unpack kernels as GVOM_TRANSPORT_RCCL -- bracketed by two host
 *       barriers; the combine's rows the same way, followed by the in-place ncclAllGather of the 1-rank communicator.
 *       All ranks must live in one process (plain device addresses travel through the segment). */
#define GVOM_TRANSPORT_LOOPBACK 3
int  gvom_comm_create2(int rank, int world, int device, const char *name, int transport, gvom_comm_t **out);
int  gvom_comm_transport(gvom_comm_t *c);
/* Peer transport, asynchronous form (GVOM_PEER_ASYNC=1 on every rank, and every rank able to register the segment with HIP;
 * otherwise the host-synchronised form): an exchange enqueues its copies and returns; the GPUs write exchange numbers into the segment.
 * Call gvom_comm_before_scan before gvom_shard_scan_local and gvom_comm_before_combine before gvom_combine_fuse: they wait
 * (normally not at all) until every peer has pulled what this rank is about to overwrite.  No-ops on the other transports. */
int  gvom_comm_before_scan(gvom_comm_t *c);
int  gvom_comm_before_combine(gvom_comm_t *c);
int  gvom_comm_peer_async(gvom_comm_t *c);          /* 1: the peer transport runs in its asynchronous form */
/* peer transport bookkeeping: {bytes pulled, copies, exports made, refused hipIpc* calls that were repeated} */
int  gvom_comm_peer_stats(gvom_comm_t *c, int64_t out[4]);
/* Peer transport and the HSA runtime's inter-process memory.  Requires HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment of
 * every rank on hosts whose driver only supports dmabuf IPC (the communicator says so on stderr when it is unset); every
 * measurement in profiles/ was taken with it.  hipIpcGetMemHandle / hipIpcOpenMemHandle can REFUSE an allocation ("invalid
 * argument" / "invalid device pointer": seen once in several hundred exports under a test that exports a fresh allocation
 * every scan, never in steady state).  The library absorbs it: a refused export moves the region into a fresh allocation
 * (gvom_shard_renew_region) and exports that; a refused open is reported through the segment, the owner does the same, and
 * every rank tries again -- up to three fresh allocations, inside the exchange, before the call fails and the communicator
 * is marked broken.  gvom_comm_peer_renewed: how often that happened on this rank. */
int64_t gvom_comm_peer_renewed(gvom_comm_t *c);
/* What the communicator itself knows of the job: out = {ranks in RCCL's communicator (ncclCommCount; -1 without RCCL), this
 * rank's number there (ncclCommUserRank), HIP device, transport in use}; busid (optional): the device's PCI bus id. */
int  gvom_comm_info(gvom_comm_t *c, int64_t out[4], char *busid, size_t busid_len);
/* RCCL calls this rank has issued so far: {ncclSend + ncclRecv calls, their bytes, groups closed, ncclAllGather calls} */
int  gvom_comm_wire_stats(gvom_comm_t *c, int64_t out[4]);
/* A rank whose caller cannot go on marks the communicator broken for every rank (see "Failure semantics" above). */
int  gvom_comm_abort(gvom_comm_t *c);
void gvom_comm_destroy(gvom_comm_t *c);
int  gvom_comm_exchange_host(gvom_comm_t *c, const int64_t *mine, int k, int64_t *all);
int  gvom_comm_barrier(gvom_comm_t *c);
int  gvom_comm_exchange_scan(gvom_comm_t *c, gvom_t *h, const int64_t *send_quads, const int64_t *send_eps,
                             const int64_t *recv_quads, const int64_t *recv_eps);
int  gvom_comm_exchange_stats(gvom_comm_t *c, gvom_t *h, const int64_t *send_returns, const int64_t *recv_returns,
                              int bytes_per_return);
int  gvom_comm_allgather_rows(gvom_comm_t *c, gvom_t *h);
/* A whole sharded scan / combine in ONE call (handles without per-voxel statistics): the sequences documented above --
 * gvom_comm_before_scan, gvom_shard_scan_local, the host exchange of the counts, gvom_shard_recv_reserve,
 * gvom_comm_exchange_scan, gvom_shard_scan_merge; gvom_comm_before_combine, gvom_combine_fuse, gvom_comm_allgather_rows,
 * gvom_combine_map2d_into -- run natively, every rank calling with ITS share of the cloud (n may be 0).
 * out = {accepted (some rank saw a return in the grid, gvom.py:147-150), returns of all ranks, bytes this rank sent, received}.
 * gvom_comm_combine_maps_into returns GVOM_EMPTY_BUFFER while the ring is empty (gvom.py:179-181). */
int  gvom_comm_process_pointcloud(gvom_comm_t *c, gvom_t *h, const void *xyz, int on_device, int64_t n, int64_t row_stride_bytes,
                                  int dtype, const double ego[3], const double *transform_4x4, int64_t out[4]);
int  gvom_comm_combine_maps_into(gvom_comm_t *c, gvom_t *h, double origin_world[3], void *pinned_out);
int  gvom_comm_rank(gvom_comm_t *c);
int  gvom_comm_world(gvom_comm_t *c);
const char *gvom_comm_last_error(gvom_comm_t *c);

/* --- accessors of the reference object ------------------------------------------------- */
/* 1 if ring slot `slot` holds a scan (origin_buffer[slot] is not None, gvom.py:201). */
int gvom_slot_filled(gvom_t *h, int slot);
int gvom_get_state(gvom_t *h, gvom_state *out);
int gvom_get_scan_stats(gvom_t *h, gvom_scan_stats *out);
/* Gvom.get_map_as_occupancy_grid (gvom.py:356-361): uint8[xy][xy][z] C-order (== the
 * reference's order='F' reshape of the lookup table), 1 where occupied. */
int gvom_get_occupancy(gvom_t *h, uint8_t *out_xyz);
/* Gvom.make_debug_voxel_map (gvom.py:363-378; kernels :1333-1378 eigenvalues, :454-473): one row of
 * 8 float32 per occupied fused voxel {x, y, z, hit/total, hit, l0-l1, l1-l2, l2}; row order is
 * unspecified (as in the reference).  *rows = number of occupied voxels; at most max_rows are written.
 * GVOM_NO_DATA unless the handle computes the per-voxel statistics (GVOM_FLAG_VOXEL_STATISTICS / _ON_DEMAND), has combined and the
 * fused map carries them. */
int gvom_debug_voxel_map(gvom_t *h, float *out, int64_t max_rows, int64_t *rows);
/* The same with the three eigenvalues of every row (reference attribute voxels_eigenvalues,
 * gvom.py:1333-1378): eigen[row][3] = {l0, l1, l2}, row for row with `out`. */
int gvom_debug_voxel_eigen(gvom_t *h, float *out, float *eigen, int64_t max_rows, int64_t *rows);
/* Gvom.make_debug_height_map (gvom.py:380-394, kernel :426-438): float32[xy*xy][7]. */
int gvom_debug_height_map(gvom_t *h, float *out);
/* Gvom.make_debug_inferred_height_map (gvom.py:396-410, kernel :442-450): float32[xy*xy][3]. */
int gvom_debug_inferred_height_map(gvom_t *h, float *out);

/* --- test hooks: dense equivalents in the reference's voxel order ------------------------
 * which: 0..buffer_size-1 = ring slot (index_buffer/hit_count_buffer/... gvom.py:59-64),
 *        GVOM_WHICH_FUSED = current fused map (combined_* gvom.py:69-75).
 * state: 0 where occupied, -1 never observed, -m-1 free with m ray passes (gvom.py:1154-1160);
 * hit/total 0 and min_h 1.0f where not occupied.  origin: voxel units.  NULLs are skipped.
 * Returns GVOM_NO_DATA if the slot / fused map is empty. */
#define GVOM_WHICH_FUSED (-1)
int gvom_read_dense(gvom_t *h, int which, int32_t *state, int32_t *hit, int32_t *total,
                    float *min_h, double origin[3], int64_t *cell_count);
/* Reference attributes metrics_buffer[slot] / combined_metrics (gvom.py:54-83,234,281; handles with
 * GVOM_FLAG_VOXEL_STATISTICS): gvom_read_rows gives the compact row of every occupied voxel in the
 * reference's voxel order (-1 elsewhere), gvom_gather_metrics the 10 statistics {mean xyz, covariance
 * xx xy xz yy yz zz, count} of selected rows: float64 for a ring slot, float32 for the fused map. */
int gvom_read_rows(gvom_t *h, int which, int32_t *rows_dense);
int gvom_gather_metrics(gvom_t *h, int which, const int32_t *rows, int64_t n, void *out);
/* which2d: internal float64 maps of the last combine, [x][y] C-order like the reference's
 * attributes (gvom.py:85-88,310-313). */
#define GVOM_MAP_HEIGHT          0
#define GVOM_MAP_INFERRED_HEIGHT 1
#define GVOM_MAP_SLOPE_X         2
#define GVOM_MAP_SLOPE_Y         3
#define GVOM_MAP_ROUGHNESS       4
#define GVOM_MAP_GUESSED_DELTA   5
int gvom_read_map2d(gvom_t *h, int which2d, double *out);

/* --- measurement ------------------------------------------------------------------------
 * Device time (HIP events on the library's own stream) of the kernels of the last
 * process_pointcloud / combine_maps call, in milliseconds, by stage.  Stages:
 * 0 trace (transform+hit+DDA), 1 encode, 2 min-height, 3 fuse (+column reductions), 4 maps2d. */
#define GVOM_N_STAGES 5
int gvom_last_stage_ms(gvom_t *h, float ms[GVOM_N_STAGES]);
/* Enables/disables per-stage event timing (adds one host sync per call when on). */
int gvom_set_profiling(gvom_t *h, int on);
/* Host-side phase times (microseconds per call, averaged; enabled by GVOM_HOST_TIMING=1):
 * [0] scan launches [1] scan wait [2] combine launches [3] combine wait [4] output copies. */
int gvom_host_timing(gvom_t *h, double us[8]);
/* Performance knobs that never change a result.  name: "segs" (step segments per ray in the trace
 * kernel), "period" (committing steps between two flushes of a wave's LDS line cache), "ep_row"
 * (dispatch row of the endpoint blocks; -1: inside segment 0's waves), "prio" (steps of remaining walk per issue-priority
 * level of a trace wave, s_setprio; 0: the hardware's own arbitration); 0 / 0 / -2 / -1 = automatic.
 * "interleave": K = 2 .. 64 (a power of two that divides the number of returns) declares the cloud to be K equally long
 * sub-clouds behind one another -- K sensors at one place, K sweeps -- whose returns of equal position are neighbours in
 * space; the trace then puts those neighbours into neighbouring lanes of one wave (merged steps, shared accumulator lines:
 * 512^2 x 128, 4 x 262,144 returns: 11.5 M -> 4.1 M memory-side atomic requests, 549 -> 355 us).  0 (default): automatic -- a
 * one-wave probe kernel in front of the trace looks for that structure (64 sampled returns per candidate K <= 4) on the second
 * cloud of a length and every 32nd after it, and the following clouds of that length are traced accordingly; clouds whose
 * length changes from scan to scan are not looked at for sub-clouds (they are probed every 8th scan for the "dirsort" verdict only); 1: off.  Only WHO traces which return changes, never a result.
 * "eager": the EAGER FUSION of one-slot rings (buffer_size 1, unsharded, xy_size % 16 == 0; with per-voxel statistics their merge is
 * enqueued with the scan as well).  The scan launches
 * ONE kernel behind the trace that encodes the ring slot AND fuses it with the previous fused map (the work of the scan's
 * encode pass and of the next combine's fusion, in one pass over the scan's accumulators), into spare buffers -- speculating
 * that the next call is gvom_combine_maps*, the reference node's pattern (gvom_ros.py:82-115: one combine per scan).  That
 * call adopts the result iff no scan came in between; otherwise it is dropped and the combine fuses the encoded slot as
 * before.  -1 (default): automatic -- off after three dropped speculations in a row, on again once combines follow scans;
 * 1: always; 0: never.  gvom_get_tuning "eager_adopted" / "eager_dropped": how often either happened.
 * "dirsort": the DIRECTIONAL ORDER of clouds that are in no spatial order (BASELINE config c1's uniformly random points; any cloud
 * shuffled, merged or filtered out of its sensor order).  The trace's cost follows the accumulator lines a 64-ray bundle touches per
 * step, and 64 random returns touch 64; a counting sort by direction bin seen from the sensor (two small kernels in front of the
 * trace: 6 cube faces x 16 x 16 cells) gives every wave 64 rays that point the same way: c1's trace 65 -> 16 us + 15 us of sorting.
 * The same remedy serves organised clouds in azimuth-major ("firing") order, whose bundles are VERTICAL fans (the scan the
 * beam-major order traces in 37 us takes 280): they are sorted by (sin-elevation row, azimuth sector) -- 256 x 32 bins -- inside
 * which the returns keep the order they came in, and the beam-major fans come back (43 us + the sort).
 * 0 (default): automatic -- the layout probe also looks, in 64 samples, whether a return and its successor point more than ~6
 * degrees apart (mode 1: cube cells) or a return and the one 63 places behind it differ by more than ~3 degrees in elevation
 * (mode 2: elevation rows), and the following clouds of that length are traced accordingly; 1 / 2: always, in that mode; -1: never.
 * gvom_get_tuning("dirsort"): the mode the last scan ran in (0: the cloud's own order).  Only WHO traces which return changes.
 * "fastdiv": k_trace divides every float32 coordinate by xy_resolution / z_resolution (gvom.py:1072-1080, 1101-1103); with the
 * reciprocal r = RN(1 / d), q = x * r, e = fma(-q, d, x), fma(e, r, q) IS the IEEE quotient for every float32 x iff it is for the
 * 2^23 float32 significands, which gvom_create checks on the host for both resolutions (once per value and process); a
 * resolution that fails the check, float64 clouds and hosts without hardware fma keep the divide.  -1 (default): use it where
 * verified; 0: always divide.  gvom_get_tuning("fastdiv"): bit 0 / 1 = in use for xy_resolution / z_resolution.
 * "encfuse" (A/B of that kernel's shape: low 4 bits waves per column block, bit 4 no XCD pairing), "fuse1" (1: one-slot
 * fusions through the general kernel), "flag_kernel" (1: round 3's completion-flag kernel).
 * (Test hooks are not part of this library: include/gvom_hip_test.h, lib/libgvom_hip_test.so.) */
int gvom_set_tuning(gvom_t *h, const char *name, int value);
/* The value the LAST scan ran with ("segs", "period", "ep_row", "prio", "interleave": what automatic resolved to). */
int gvom_get_tuning(gvom_t *h, const char *name, int *value);
/* Raw HIP stream the library launches on (hipStream_t as void*), for external event timing. */
void *gvom_stream(gvom_t *h);
/* differs between any two handles of the process and changes whenever a SEND region of `h` (GVOM_XBUF_SEND_*: the only
 * grow-only buffers another rank reads) has been re-allocated: tells a cache of addresses derived from gvom_shard_buffer
 * when to look again */
uint64_t gvom_alloc_generation(gvom_t *h);
/* the same for one region (which = GVOM_XBUF_SEND_*, or -1 for GVOM_BUF_HEIGHT_MAPS): changes exactly when the allocation the
 * region lies in is replaced, and differs between handles */
uint64_t gvom_region_generation(gvom_t *h, int which);

const char *gvom_last_error(gvom_t *h);      /* never NULL */
int gvom_backend_info(char *buf, size_t len); /* "gfx950 ..." device + build string */
int gvom_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GVOM_HIP_H */
