// gvom_capi.hip -- host side of libgvom_hip.so: the C ABI declared in include/gvom_hip.h.
//
// Replaces the host bodies of the reference's Gvom.__init__ / process_pointcloud /
// combine_maps (/root/reference/scripts/gvom.py:29-354, "gvom.py:NNN").  Differences in
// mechanism, not in results:
//   * all device memory is a grow-only arena owned by the handle (the reference allocates
//     ~17 arrays per call);
//   * 3 launches per scan and 2 per combine on one private HIP stream (reference: ~20 and
//     ~35+3B), one 16-byte D2H per scan, one D2H of the four maps per combine;
//   * ring slots are swapped with a spare staging slot on commit, so a rejected scan
//     (gvom.py:148-150) leaves the ring untouched.
#include "gvom_internal.h"
#include "../../include/gvom_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <immintrin.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#define VIS __attribute__((visibility("default")))

namespace {

struct Buf {                                   // grow-only device buffer
    void *p = nullptr;
    size_t bytes = 0;
    uint64_t gen = 0;                          // changes with every (re-)allocation (process-wide unique: gvom_region_generation)
};

struct Slot {                                  // one scan in sparse form
    int32_t *state = nullptr;                  // [V] storage order
    uint16_t *code16 = nullptr;                // [V] 16-bit codes of the same voxels (xy % 4 == 0 grids), read by k_fuse4
    uint32_t *tags = nullptr;                  // [ntiles] tile epochs (live iff == epoch)
    uint32_t epoch = 0;
    Buf crows;                                 // compact rows, uint4 each: {hit, total, min-height bits, 0}
    Buf metrics, base, rowvox;                 // optional statistics: double[rows][10] x2 (metrics; own-voxel moments), row -> voxel
    int64_t origin[3] = {0, 0, 0};
    int64_t count = 0;
    bool filled = false;
    bool has_code16 = false;                   // the last encode wrote the 16-bit codes (k_fuse4 may read this slot)
    bool has_metrics = false;                  // the scan computed its per-voxel statistics (metrics / base / rowvox are this scan's)
    bool stats_valid = false;
    gvom_scan_stats stats = {0, 0, 0, 0};
};

struct Fused {
    int32_t *state = nullptr;
    uint32_t *tags = nullptr;
    uint32_t epoch = 0;
    Buf rows;                                  // compact rows, uint4 each (as a slot's)
    Buf metrics;                               // optional statistics: float[rows][10]
    int64_t origin[3] = {0, 0, 0};
    int64_t count = 0;                         // rows on THIS rank
    bool valid = false;
    bool has_metrics = false;                  // k_fuse_stats merged the statistics of this map (every source had its own)
};

}  // namespace

struct gvom_handle {
    gvom_params prm;
    int device = 0;
    int rank = 0, world = 1;
    bool sharded = false;                               // created by gvom_create_sharded: scans / combines go through the split entry points
    int sy_lo = 0, sy_hi = 0;
    size_t V = 0, slabV = 0, cells2d = 0, ntiles = 0;
    int nseg = 1;
    uint32_t epoch = 0;                                 // last tile epoch handed out
    hipStream_t stream = nullptr;
    std::mutex mu;                                      // handle state
    std::mutex scan_mu;                                 // one scan at a time (held across the wait for k_trace, during which `mu` is free)
    std::string err;

    uint32_t *hit = nullptr, *total = nullptr, *mh = nullptr;   // dense accumulators (hit, ray passes, min-height), zero between scans
    size_t acc_elems = 0;
    int tune_segs = 0, tune_ep_row = -2, tune_period = 0; // gvom_set_tuning (0 / -2: automatic)
    int tune_prio = -1;                                 // gvom_set_tuning "prio" (-1: automatic)
    int tune_ilv = 0;                                   // gvom_set_tuning "interleave": sub-clouds per cloud (0: automatic, 1: off)
    int last_knobs[5] = {0, 0, 0, 0, 1};                // gvom_get_tuning: segs, period, ep_row, prio, interleave of the last scan
    int64_t last_n = -1;                                // returns of the previous scan
    uint32_t probe_var_age = 0;                         // scans of changing length since the probe last ran
    int64_t probe_n = -1; uint32_t probe_age = 0;       // layout probe (k_layout_probe): the length it last looked at, scans since
    int tune_fuse1 = 0;                                 // gvom_set_tuning "fuse1": 1 = the one-slot fusion through k_fuse4 as well (A/B)
    int tune_flag_kernel = 0;                           // gvom_set_tuning "flag_kernel": 1 = the combine's completion flag from a kernel of its own (round 3's form)
    int tune_churn = 0;                                 // test hook: re-allocate the endpoint send region every scan
    uint64_t alloc_gen = 0;                             // changes whenever a send region of this handle is re-allocated
    uint64_t handle_gen = 0;                            // this handle's own number (its fixed allocations)
    bool exported = false;                              // a transport has exported this handle's send regions to other processes
    bool holds_pooled = false;                          // some region of this handle came out of the process-wide pool (a peer may still have it mapped)
    std::vector<Buf> retired;                           // outgrown / replaced exported regions (possibly still mapped by peers), with their sizes
    uint64_t fixed_gen[3] = {0, 0, 0};                  // generations of the fixed exported allocations: send ids, send quads, height-map rows
    // rank exchange of a sharded map (world > 1): send / receive regions, indexed by peer rank
    uint32_t *x_send_ids = nullptr, *x_recv_ids = nullptr;     // quad ids: [Q] by owner / [world][myQ] by source
    void *x_send_pay = nullptr, *x_recv_pay = nullptr;         // 1 KiB per quad, same indexing
    Buf x_send_eps, x_recv_eps;                                 // endpoints {L, min-height}: [world][ep_cap] / concatenated by source
    int64_t x_ep_cap = 0;
    std::vector<int64_t> x_recv_ep_off;                         // receive offsets (endpoints) by source, [world + 1]
    uint32_t *x_qcnt = nullptr, *x_ecnt = nullptr, *x_spcnt = nullptr;   // device counters, [world * 16] each
    unsigned long long *x_host = nullptr, *x_host_dev = nullptr;   // pinned, mapped: [3*world + 2]
    Buf x_send_sp, x_recv_sp;                                   // sharded statistics: returns (3 values each) for / from other ranks
    std::vector<int64_t> x_recv_sp_off;                         // receive offsets (returns) by source, [world + 1]
    int pending_dtype = 0;                                      // cloud type of the scan between scan_local and scan_merge
    size_t x_Q = 0, x_myQ = 0;
    ScanParams pending_P;                                       // scan parameters between scan_local and scan_merge
    unsigned resident_blocks = 2048;                    // 256-thread workgroups resident on the device (queried)
    bool f32_sqrt = false;                              // GVOM_FLAG_CUDA_F32_SQRT
    std::vector<Slot> slots;                            // buffer_size + 1 (one is staging)
    std::vector<int> ring;                              // ring position -> slots index
    int staging = 0;
    int buffer_index = 0, last_buffer_index = 0;
    Buf in_pts, world_pts[2];                           // world_pts: the returns as k_trace stored them for k_stats, alternating per scan
    uint32_t *counters = nullptr;                       // device: [0] scan rows, [2..3] fuse rows (u64)
    uint32_t *counters_host = nullptr;                  // pinned, device-mapped: kernels publish counts here
    uint32_t *counters_host_dev = nullptr;              // device view of counters_host

    // pending (uncommitted) scan
    bool pending = false;
    bool pending_any = false;
    int64_t pending_origin[3] = {0, 0, 0};
    int64_t pending_n = 0;

    Fused fused[2];
    int cur = 0;                                        // fused[cur] is the latest if valid
    bool has_combined = false;
    int64_t combined_cell_count = 0;                    // global count if set by the sharded layer
    MapDesc *descs_dev = nullptr, *descs_host = nullptr;
    uint32_t *blockcounts = nullptr;                    // per-workgroup occupied counts of k_fuse
    int fuse_blocks = 0;
    int cnt_blocks = 0;                                 // entries of blockcounts the last fusion wrote (k_map2d sums them)
    // EAGER FUSION (one-slot rings: buffer_size 1, unsharded, no statistics, xy % 16 == 0).  The scan launches k_encfuse
    // behind k_trace instead of k_encode: the slot is encoded AND fused with the previous map in one pass over the
    // accumulators, into the spare fused buffer, a spare height buffer and a spare count array -- speculating that the
    // next call is combine_maps (the reference node's pattern: one combine per scan).  fuse_impl adopts the result (swaps
    // the spares in) iff nothing has changed since; otherwise it is dropped and the combine runs k_fuse1 over the encoded
    // slot as before.  Same results either way (tests: eager on / off / mixed call orders).
    double *hmaps2 = nullptr;                           // spare [sy][3][sx] buffer (k_encfuse's column tails)
    uint32_t *blockcounts2 = nullptr;
    bool spec_valid = false;                            // a speculative fusion is waiting to be adopted
    int spec_nxt = 0, spec_slot = 0, spec_blocks = 0;
    uint32_t spec_epoch = 0;
    int64_t spec_origin[3] = {0, 0, 0};
    // ... with per-voxel statistics: the speculative fusion's statistics half (k_fuse_stats on the statistics stream, behind the
    // scan's own k_stats / k_stats_gather) is enqueued with the scan too; what it needs of eager_launch's frame is kept here
    bool spec_has_metrics = false;                      // the speculative fused map will carry merged statistics
    Buf flink[2];                                       // per fused buffer: link[fused row] = row in the previous map (k_encfuse -> k_fuse_stats)
    bool fs_reads[2] = {false, false};                  // the pending k_fuse_stats reads fused[i]'s states / tile tags
    FuseParams spec_FP;
    FuseDescs spec_KD;
    // DIRECTIONAL ORDER of unordered clouds (k_dirbin_*, ScanParams::perm): "dirsort" 1 always, -1 never, 0 automatic -- when the
    // layout probe found no spatial order in the previous cloud of this length (BASELINE c1's 50,000 random points: k_trace 65 -> 16 us)
    int tune_dirsort = 0;
    Buf dir_keys, dir_perm;                             // uint16 key / uint32 position -> return, per return
    uint32_t *dir_hist = nullptr;                       // [3][GVOM_DIRBINS]: two histograms (alternating, zero between uses) + the bins' cursors
    uint32_t dir_flip = 0;
    int last_dirsort = 0;                               // gvom_get_tuning "dirsort": the last scan ran in directional order
    int tune_encfuse = 0;                               // gvom_set_tuning "encfuse": A/B of k_encfuse's shape (low 4 bits: waves per block, bit 4: no XCD pairing)
    int tune_fastdiv = -1;                              // gvom_set_tuning "fastdiv": 0 = IEEE divides by the resolutions in k_trace, else the verified reciprocal form
    int fastdiv_ok = 0;                                 // bit 0 / 1: div_by_res() verified for xy_resolution / z_resolution (verify_fastdiv)
    int tune_eager = -1;                                // gvom_set_tuning "eager": 0 off, 1 always, -1 automatic (off after 3 wasted in a row)
    int eager_waste = 0;                                // speculations dropped in a row (saturates at 4)
    int eager_stat[2] = {0, 0};                         // adopted / dropped since creation (gvom_get_tuning "eager_adopted" / "eager_dropped")
    bool last_scan_spec = false;                        // the last accepted scan went through k_encfuse
    bool solo_encoded = false;                          // a sharded handle of ONE rank: gvom_shard_scan_local has already encoded the scan (nothing to wait for)
    bool fresh_scan = false;                            // a scan has been committed and no combine has looked at it yet

    double *hmaps = nullptr;                            // [sy][3][sx]: height | inferred height | positive density
    double *height = nullptr, *inferred = nullptr;      // = hmaps, hmaps + xy  (row stride hs = 3*xy)
    int hs = 0;
    double *slope_x = nullptr, *slope_y = nullptr, *rough = nullptr, *guessed = nullptr;   // [sy][sx]
    hipStream_t own_stream = nullptr;                   // created by the library
    // asynchronous combine (gvom_combine_begin / _end): k_map2d runs on a second stream, so the next
    // scan's k_trace / k_encode (instruction-bound) overlap its PCIe-bound stores
    hipStream_t stream_b = nullptr;
    // host clouds go up on a stream of their own when the main stream is busy (the ROS node's two threads: the cloud
    // callback hands scan k + 1 over while the timer thread's combine k is still running): the copy engine moves the
    // cloud while k_fuse / k_map2d run, and k_trace waits for it on the device
    hipStream_t stream_up = nullptr;
    hipEvent_t ev_up = nullptr;
    // per-voxel statistics (opt-in) run on a stream of their own: k_stats / k_stats_gather beside the combine's fusion,
    // k_fuse_stats beside k_map2d's PCIe-bound stores.  ev_enc_s / ev_fz_s: main (or fusion) stream -> statistics
    // stream; ev_sdone: everything enqueued on the statistics stream so far (the next k_trace rewrites what it reads);
    // ev_fsdone: the last k_fuse_stats (the next fusion rewrites the fused buffer it reads as "previous")
    hipStream_t stream_s = nullptr;
    hipEvent_t ev_enc_s = nullptr, ev_fz_s = nullptr, ev_sdone = nullptr, ev_fsdone = nullptr;
    // ev_before[k & 1]: the statistics stream's work enqueued BEFORE scan k's own -- all that can still read what scan
    // k + 1 rewrites (the slot it stages into left the ring at commit k; its buffer of stored returns was scan k - 1's):
    // scan k + 1 waits for that, not for scan k's statistics, which run beside it
    hipEvent_t ev_before[2] = {nullptr, nullptr};
    bool before_valid[2] = {false, false};
    uint32_t stats_scan = 0;                             // scans with statistics so far (parity selects the buffers above)
    bool stats_prev_committed = true;                    // a rejected scan leaves its slot as the staging slot: the next scan rewrites it
    bool s_pending = false, fs_pending = false;          // recorded and not known to have completed
    hipEvent_t ev_fused = nullptr, ev_mapped = nullptr, ev_done = nullptr;
    std::mutex combine_mu;                              // one combine call at a time (taken before `mu`)
    bool pending_combine = false;                       // begun, not ended
    uint32_t combine_seq = 0;                           // completion flag of the synchronous combine (counters_host + 4)
    double last_wait_ns[2] = {0.0, 0.0};                // how long the scan / the combine waited last time (wait_published)
    bool mapped_unjoined = false;                       // ev_mapped recorded; the main stream has not waited on it
    // a fusion enqueued on the second stream (asynchronous combine, rings of >= 3 filled slots) READS the ring slots
    // it was given; the main stream must not overwrite one of them (the second scan after the begin does: the
    // oldest slot becomes the staging slot) nor read the fused map it writes before it has finished
    hipEvent_t ev_fuse_b = nullptr;
    bool fuse_b_unjoined = false;                       // ev_fuse_b recorded; the main stream has not waited on it
    uint64_t fuse_b_slots = 0;                          // bit k: slots[k] is a source of that fusion
    bool scan_inflight = false;                         // a scan's kernels are enqueued and it is not committed yet (scan_mu held)
    std::vector<void *> out_bufs;                        // buffers handed out by gvom_output_buffer_alloc (coherent by construction)
    void *last_checked_out = nullptr;                    // a caller's own output buffer whose flags have been checked
    void *out_host = nullptr;                           // pinned, device-mapped staging for the 4 outputs
    char *out_host_dev = nullptr;                       // device view of out_host (zero-copy target)
    uint32_t scan_seq = 0;                              // sequence number of the {seq,count} flag
    bool ev_scan = false, ev_fuse = false, ev_map = false;   // which profiling events are recorded
    bool maps_valid = false;

    double ego[3] = {0, 0, 0};
    int in_off[3] = {0, 1, 2};                          // element offsets of x, y, z in the cloud being scanned
    bool in_f32 = false;                                // float32 records widened to a float64 computation (PointCloud2 ingest)

    Buf tl;                                             // diagnostic build: k_trace's timeline of the last scan (GVOM_TRACE_TIMELINE)
    int tl_grid[2] = {0, 0};
    double host_ns[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // host-side phase timing (GVOM_HOST_TIMING)
    long host_calls = 0;
    bool host_timing = false;
    bool stats = false;                                 // per-voxel statistics computed by the NEXT scan / merged by the next fusion
    // ON DEMAND (GVOM_FLAG_STATISTICS_ON_DEMAND): the statistics start ON -- the reference computes them in every scan and
    // combine (gvom.py:159, 276-284) and its node reads them every tick (gvom_ros.py:171) -- and go OFF when three combines
    // in a row went by without anybody reading them (gvom_debug_voxel_map*, gvom_read_rows, gvom_gather_metrics); a later read
    // finds no data and switches them ON again for the scans that follow
    bool stats_auto = false;
    int stats_idle = 0;                                 // combines since the statistics were last read
    bool stats_release = false;                         // they have just been switched off: their buffers go at the end of this combine
    int acc_pad = 7, sxq = 0;                           // accumulator row pitch (lines) = ceil(xy/4) + acc_pad
    bool profiling = false;
    hipEvent_t ev[8] = {nullptr};
    float stage_ms[GVOM_N_STAGES] = {0, 0, 0, 0, 0};
};

namespace {

inline double now_ns() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e9 + t.tv_nsec; }
#define HT(h, slot, t0) do { if ((h)->host_timing) { double n_ = now_ns(); (h)->host_ns[slot] += n_ - (t0); (t0) = n_; } } while (0)

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            char b_[512];                                                                       \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),      \
                     __FILE__, __LINE__);                                                       \
            (h)->err = b_;                                                                      \
            return GVOM_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

// (process-wide: a value names one state of one handle's allocations -- gvom_alloc_generation)
static std::atomic<uint64_t> g_alloc_generation{0};

// Allocations another process may map (the peer transport exports the send regions and the height-map rows) are whole
// multiples of 2 MiB: the HSA runtime carves smaller ones out of shared 2 MiB blocks, and a block cannot be exported
// twice -- a second small region landing in an exported block is what hipIpcGetMemHandle refused ("invalid argument").
inline size_t exportable_size(size_t bytes) { const size_t g = (size_t)2 << 20; return ((bytes ? bytes : 1) + g - 1) / g * g; }

// Regions another process has had mapped are never given back to the allocator while this process lives (a later allocation
// tends to get their address, and importers that open "it" have been seen reading the OLD memory: profiles/r3_peer_churn.txt).
// They wait in a process-wide POOL instead, with their real sizes and their generation (= the name the communicators know
// the allocation by: a region that becomes current again is neither exported nor opened a second time), and the next
// handle that needs an exportable region of that size takes one from there.  The pool is accounted exactly; past
// GVOM_POOL_WARN_BYTES it says so on stderr, once -- it never falls back to freeing.
#define GVOM_POOL_WARN_BYTES ((size_t)16 << 30)
struct ExportPool {
    std::mutex m;
    std::vector<Buf> v;
    size_t bytes = 0;
    bool warned = false;
};
static ExportPool g_pool;
void pool_put(void *p, size_t bytes, uint64_t gen)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool.m);
    Buf b; b.p = p; b.bytes = bytes; b.gen = gen;
    g_pool.v.push_back(b);
    g_pool.bytes += bytes;
    if (g_pool.bytes > GVOM_POOL_WARN_BYTES && !g_pool.warned) {
        g_pool.warned = true;
        fprintf(stderr, "libgvom_hip: %zu MiB of device memory that other processes may have mapped are parked (never freed while "
                        "the process lives; re-used by later sharded handles of the same sizes): create fewer exported handles per process\n",
                g_pool.bytes >> 20);
    }
}
// an exportable allocation of exactly `bytes` (a multiple of 2 MiB): a parked one, or a fresh one
// (*reused: the region came out of the pool, i.e. some peer may STILL have it mapped: its next owner must park it again
// whether or not it exports anything itself -- ADVICE r4)
hipError_t pool_get(size_t bytes, void **p, uint64_t *gen, bool *reused = nullptr)
{
    {
        std::lock_guard<std::mutex> lk(g_pool.m);
        for (size_t k = 0; k < g_pool.v.size(); ++k)
            if (g_pool.v[k].bytes == bytes) {
                *p = g_pool.v[k].p; *gen = g_pool.v[k].gen;
                if (reused) *reused = true;
                g_pool.bytes -= bytes;
                g_pool.v.erase(g_pool.v.begin() + (long)k);
                return hipSuccess;
            }
    }
    *gen = ++g_alloc_generation;
    return hipMalloc(p, bytes);
}

int ensure(gvom_handle *h, Buf &b, size_t bytes)
{
    if (b.bytes >= bytes) return GVOM_OK;
    size_t want = bytes + bytes / 2 + 256;
    const bool exported = &b == &h->x_send_eps || &b == &h->x_send_sp;
    if (exported) want = exportable_size(want);
    // (a region another process may have mapped is never freed while the handle lives: freed and re-allocated, the new region
    // tends to get the old one's address, and importers that re-open "it" were seen reading the OLD memory -- a silently
    // different map after ~50 scans of the churn test.  Growth is geometric: the retired regions add up to less than the last.)
    if (b.p && exported) { h->retired.push_back(b); b.p = nullptr; }
    if (b.p) HIPCHK(h, hipFree(b.p));
    b.p = nullptr; b.bytes = 0;
    if (exported) HIPCHK(h, pool_get(want, &b.p, &b.gen, &h->holds_pooled));
    else { HIPCHK(h, hipMalloc(&b.p, want)); b.gen = ++g_alloc_generation; }
    b.bytes = want;
    if (&b == &h->x_send_eps || &b == &h->x_send_sp) h->alloc_gen = b.gen;   // (see gvom_alloc_generation)
    return GVOM_OK;
}

// grows a device buffer KEEPING its contents (stream-ordered copy, then a wait: rare)
int ensure_keep(gvom_handle *h, Buf &b, size_t bytes)
{
    if (b.bytes >= bytes) return GVOM_OK;
    const size_t want = bytes + bytes / 2 + 256;
    void *np = nullptr;
    HIPCHK(h, hipMalloc(&np, want));
    if (b.p) {
        HIPCHK(h, hipMemcpyAsync(np, b.p, b.bytes, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipFree(b.p));
    }
    b.p = np; b.bytes = want;
    return GVOM_OK;
}

inline int64_t floor_mod(int64_t a, int64_t n) { int64_t r = a % n; return r < 0 ? r + n : r; }

inline int clamp_delta(int64_t d, int size)
{   // any |d| >= size puts the whole source window outside; keep ints small
    if (d > size) return size;
    if (d < -size) return -size;
    return (int)d;
}

void fill_scan_params(const gvom_handle *h, const int64_t origin[3], const double *tf, ScanParams &P, int64_t n_points = 0)
{
    const gvom_params &p = h->prm;
    P.xy_res = p.xy_resolution; P.z_res = p.z_resolution;
    P.min_d2 = p.min_distance * p.min_distance;
    for (int k = 0; k < 3; ++k) P.origin[k] = (double)origin[k];
    P.has_tf = tf ? 1 : 0;
    for (int k = 0; k < 12; ++k) P.tf[k] = tf ? tf[k] : 0.0;
    P.pt0[0] = (float)(h->ego[0] / p.xy_resolution);
    P.pt0[1] = (float)(h->ego[1] / p.xy_resolution);
    P.pt0[2] = (float)(h->ego[2] / p.z_resolution);
    P.rinv[0] = (float)(1.0 / p.xy_resolution); P.rinv[1] = (float)(1.0 / p.z_resolution);
    P.drcp[0] = 1.0 / p.xy_resolution; P.drcp[1] = 1.0 / p.z_resolution;
    P.fastdiv = h->tune_fastdiv == 0 ? 0 : h->fastdiv_ok;
    for (int k = 0; k < 3; ++k) {
        const int64_t size = k < 2 ? p.xy_size : p.z_size;
        const bool small = origin[k] > -(1ll << 18) && origin[k] < (1ll << 18) && size >= 5;
        P.win_lo[k] = small ? (float)(origin[k] + 2) : 1.0f;      // (integers below 2^24: exact)
        P.win_hi[k] = small ? (float)(origin[k] + size - 2) : 0.0f;
    }
    P.xy = p.xy_size; P.zs = p.z_size;
    P.om[0] = (int)floor_mod(origin[0], p.xy_size);
    P.om[1] = (int)floor_mod(origin[1], p.xy_size);
    P.om[2] = (int)floor_mod(origin[2], p.z_size);
    P.sy_lo = h->sy_lo; P.sy_hi = h->sy_hi;
    P.off[0] = h->in_off[0]; P.off[1] = h->in_off[1]; P.off[2] = h->in_off[2];
    P.in_f32 = h->in_f32 ? 1 : 0;
    P.nseg = h->nseg;
    {
        auto lg = [](int v) { int k = 0; while ((1 << k) < v) ++k; return (1 << k) == v ? k : -1; };
        P.lg_nseg = lg(h->nseg); P.lg_zs = lg(p.z_size);
        if (P.lg_nseg < 0 || P.lg_zs < 0) P.lg_nseg = P.lg_zs = -1;
    }
    P.sxq = h->sxq;
    P.shard_world = h->world; P.shard_rank = h->rank; P.shard_rows = h->world > 1 ? p.xy_size / h->world : p.xy_size;
    P.stat_e = p.xy_eigen_dist;
    // DDA step segments: the ego sits at the window centre, so a ray takes at most size/2 + 2 steps.
    // Segments exist to fill the chip with waves when a scan has few returns (a 131 k-point scan is 2
    // waves per SIMD); every segment wave repeats the ray set-up and replays the earlier steps, so
    // clouds with plenty of returns use fewer segments.  The last two segments are shortened: their
    // workgroups are dispatched last and only the longest rays reach them.  (Measured, 256^3 / 131 k
    // points: 5 segments 43.6 us, 6: 44.1, 7: 43.8, 4: 46.4; flush period 16: -0.4 us against 12.)
    {
        const int maxsteps = (p.xy_size > p.z_size ? p.xy_size : p.z_size) / 2 + 2;
        // "plenty of returns": more 64-ray bundles than three quarters of the device's resident wave slots
        // (MI355X: 256 CUs x 32 waves -> 393,216 returns)
        const bool many = n_points / 64 > 3 * (int64_t)h->resident_blocks;
        int nsegs = many ? 3 : 5;
        if (h->tune_segs > 0) nsegs = h->tune_segs;
        if (nsegs > 9) nsegs = 9;
        if (nsegs > maxsteps / 8) nsegs = maxsteps / 8 > 0 ? maxsteps / 8 : 1;
        P.nsegs = nsegs;
        // flush period (final step body; c4: 8 / 12 / 16 / 24 / 32 -> 552 / 537 / 545 / 608 / 638 us, c5 12 / 16: 2438 / 2460,
        // m256 and c3: 12 = 16)
        P.lc_period = h->tune_period > 0 ? h->tune_period : (many ? 12 : 16);
        if (P.lc_period > 32) P.lc_period = 32;          // the line cache is direct-mapped with 64 entries
        // endpoint blocks first: their atomics retire under the walk; clouds with plenty of returns (several dispatch rounds
        // anyway) do the endpoint work inside segment 0's waves instead (-1): one wave and one load of the returns less per
        // bundle (r5, ep_row 0 / -1: c4 615.6 / 605.6 us per step, c5 2297 / 2052; c3 145.9 / 148.0 and m256 98.2 / 98.1 keep the row)
        P.ep_row = h->tune_ep_row >= -1 ? h->tune_ep_row : (many ? -1 : 0);
        if (P.ep_row > P.nsegs) P.ep_row = P.nsegs;
        // issue priority by remaining work (k_trace, prio_by_remaining): steps per priority level.  Measured, off / 4 / 8 / 16:
        // m256 40.1 / 40.1 / 39.0 / 40.4 us, c2 39.2 / - / 36.7 / -, c3 70.6 / - / 68.2 / -, c4 unchanged
        P.prio_div = h->tune_prio >= 0 ? h->tune_prio : 8;
        P.f32_sqrt = h->f32_sqrt ? 1 : 0;
        P.ilv_lg = 0; P.ilv_len = n_points;
        {
            // automatic: what the layout probe found in the previous cloud of as many returns, at most 4 sub-clouds side by
            // side (c5, 16 sensors: interleave 4 / 8 / 16 -> k_trace 1450 / 1653 / 1745 us: a wave should keep >= 16 azimuths)
            int K = h->tune_ilv > 1 ? h->tune_ilv : 1;
            if (h->tune_ilv == 0 && h->counters_host) {
                // the layout probe's last answer {n << 8 | log2 K} (host-mapped; written by a kernel of an EARLIER scan)
                const unsigned long long w = *(volatile unsigned long long *)(h->counters_host + 8);
                if ((int64_t)(w >> 8) == n_points) K = 1 << (int)(w & 7ull);
            }
            if (K > 64 || (K & (K - 1)) != 0 || n_points <= 0 || n_points % K != 0) K = 1;
            while ((1 << P.ilv_lg) < K) ++P.ilv_lg;
            P.ilv_len = n_points / K;
            // interleaved sub-clouds keep a wave's 64 rays inside a few accumulator lines per step: the line cache holds a
            // longer run (c4, interleave 4: period 12 / 16 / 24 / 32 -> 390 / 379 / 370 / 372 us)
            if (K > 1 && h->tune_period <= 0) P.lc_period = 24;
        }
        P.dbg = gvom_diag_env("GVOM_TRACE_DEBUG");
        const double w_last = nsegs <= 3 ? 0.35 : 0.6;
        const double unit = nsegs >= 3 ? (double)maxsteps / ((nsegs - 2) + 0.85 + w_last) : (double)maxsteps / nsegs;
        double acc = 0.0;
        for (int k = 0; k < 10; ++k) {
            P.seg_start[k] = (int)(acc + 0.5);
            acc += (nsegs < 3 || k < nsegs - 2) ? unit : (k == nsegs - 2 ? 0.85 * unit : (k == nsegs - 1 ? w_last * unit : unit));
        }
    }
    P.epoch = 0;
}

// div_by_res() (gvom_trace.hip) replaces (double)x / d, x a float32, by two fused multiply-adds around r = RN(1 / d).  It is
// used for a divisor only after this check: every float32 significand (2^23 of them: the rounding of the quotient depends on
// nothing else, see there) through the same three operations, against the divide.  ~30 ms per divisor, once per process.
__attribute__((target("fma"))) static bool verify_fastdiv_all(double d, double r)
{
    for (uint32_t m = 0; m < (1u << 23); ++m) {
        const uint32_t bits = 0x3f800000u | m;
        float xf;
        memcpy(&xf, &bits, 4);
        const double x = (double)xf;
        const double q = x * r;
        const double e = __builtin_fma(-q, d, x);
        const double q2 = __builtin_fma(e, r, q);
        if (q2 != x / d) return false;
    }
    return true;
}
static bool verify_fastdiv(double d)
{
    static std::mutex mu;
    static std::vector<std::pair<double, bool>> known;
    if (!(d > 0x1p-64 && d < 0x1p64)) return false;        // (also NaN: no float32 coordinate over or underflows anything for such a d)
    if (!__builtin_cpu_supports("fma")) return false;      // (no hardware fma on this host to check with: the divide stays)
    std::lock_guard<std::mutex> lk(mu);
    for (const auto &k : known) if (k.first == d) return k.second;
    const bool ok = verify_fastdiv_all(d, 1.0 / d);
    known.emplace_back(d, ok);
    return ok;
}

int create_impl(const gvom_params *params, int device_id, int rank, int world, bool sharded, gvom_t **out)
{
    if (!params || !out) return GVOM_ERR_INVALID;
    *out = nullptr;
    if (params->xy_size <= 0 || params->z_size <= 0 || params->buffer_size <= 0 ||
        !(params->xy_resolution > 0) || !(params->z_resolution > 0) || world <= 0 || rank < 0 ||
        rank >= world)
        return GVOM_ERR_INVALID;
    if (params->buffer_size >= GVOM_MAX_SLOTS || params->z_size > 1024 || world > GVOM_MAX_SLOTS) return GVOM_ERR_CAPACITY;
    // a sharded map: every rank owns xy/world storage rows, a multiple of 4 (accumulator patches are
    // 4 rows high); per-voxel statistics need every return on the owner and are not exchanged
    if (sharded && params->xy_size % (4 * world) != 0) return GVOM_ERR_INVALID;
    const double Vd = (double)params->xy_size * params->xy_size * params->z_size;
    if (Vd >= 2147483648.0) return GVOM_ERR_CAPACITY;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev)
        return GVOM_ERR_NO_DEVICE;
    gvom_handle *h = new gvom_handle();
    h->alloc_gen = h->handle_gen = ++g_alloc_generation;
    h->prm = *params;
    h->device = device_id;
    h->rank = rank; h->world = world; h->sharded = sharded;
    h->stats = (params->reserved0 & GVOM_FLAG_VOXEL_STATISTICS) != 0;
    h->stats_auto = !h->stats && !sharded && (params->reserved0 & GVOM_FLAG_STATISTICS_ON_DEMAND) != 0;
    if (h->stats_auto) h->stats = true;
    h->f32_sqrt = (params->reserved0 & GVOM_FLAG_CUDA_F32_SQRT) != 0;
    h->fastdiv_ok = (verify_fastdiv(params->xy_resolution) ? 1 : 0) | (verify_fastdiv(params->z_resolution) ? 2 : 0);
    if (const char *v = getenv("GVOM_VOXEL_STATISTICS")) { h->stats = atoi(v) != 0; h->stats_auto = false; }
    // sharded statistics send a return to the ranks that own the first and the last row of its neighbourhood: the
    // neighbourhood (2 xy_eigen_dist + 1 rows) must not reach over a whole slab
    if (sharded && h->stats && world > 1 && 2 * params->xy_eigen_dist + 1 > params->xy_size / world) { delete h; return GVOM_ERR_INVALID; }
    if (const char *v = getenv("GVOM_HOST_TIMING")) h->host_timing = atoi(v) != 0;
    const int xy = params->xy_size, zs = params->z_size;
    h->sy_lo = (int)((int64_t)xy * rank / world);
    h->sy_hi = (int)((int64_t)xy * (rank + 1) / world);
    h->V = (size_t)xy * xy * zs;
    h->slabV = (size_t)(h->sy_hi - h->sy_lo) * xy * zs;
    h->cells2d = (size_t)xy * xy;
    h->nseg = (xy + 63) / 64;
    h->ntiles = (size_t)xy * zs * h->nseg;
#define CK(call)                                                                                \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "gvom_create: %s failed: %s\n", #call, hipGetErrorString(e_));      \
            gvom_destroy(h);                                                                    \
            return GVOM_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)
    CK(hipSetDevice(device_id));
    CK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = h->stream;
    {   // the second stream's kernels (fusion, k_map2d: latency- and PCIe-bound, few waves) go ahead of the
        // instruction-bound k_trace they overlap with in an asynchronous combine
        int lo = 0, hi = 0;
        CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CK(hipStreamCreateWithPriority(&h->stream_b, hipStreamNonBlocking, hi));
    }
    CK(hipEventCreateWithFlags(&h->ev_fused, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&h->ev_mapped, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&h->ev_fuse_b, hipEventDisableTiming));
    CK(hipStreamCreateWithFlags(&h->stream_up, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&h->stream_s, hipStreamNonBlocking));
    for (hipEvent_t *e : {&h->ev_enc_s, &h->ev_fz_s, &h->ev_sdone, &h->ev_fsdone, &h->ev_before[0], &h->ev_before[1]})
        CK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&h->ev_up, hipEventDisableTiming));
    // accumulators are micro-tiled in 4x4 (x,y) patches (gvom_internal.h "ACCUMULATOR LAYOUT")
    // row pitch of the patch rows, padded (GVOM_ACC_PAD lines of 64 B) so that the z levels of one
    // (x, y) patch -- xy*16 bytes apart, a multiple of 4 KiB for xy = 256 -- do not all map to the
    // same memory channel
    h->sxq = (xy + 3) / 4 + h->acc_pad;
    const size_t acc_elems = (size_t)h->sxq * ((xy + 3) / 4) * zs * 16 + 256;
    h->acc_elems = acc_elems;
    CK(hipMalloc((void **)&h->hit, acc_elems * 4));
    CK(hipMalloc((void **)&h->total, acc_elems * 4));
    CK(hipMalloc((void **)&h->mh, acc_elems * 4));
    CK(hipMemsetAsync(h->hit, 0, acc_elems * 4, h->stream));
    CK(hipMemsetAsync(h->total, 0, acc_elems * 4, h->stream));
    CK(hipMemsetAsync(h->mh, 0, acc_elems * 4, h->stream));
    {   // 256-thread workgroups the device keeps resident (8 per CU at <= 64 VGPRs / <= 80 SGPRs)
        hipDeviceProp_t pr;
        CK(hipGetDeviceProperties(&pr, device_id));
        h->resident_blocks = (unsigned)pr.multiProcessorCount * 8u;
    }
    h->slots.resize(params->buffer_size + 1);
    for (auto &s : h->slots) {
        CK(hipMalloc((void **)&s.state, h->V * 4));
        if ((params->xy_size & 3) == 0) CK(hipMalloc((void **)&s.code16, h->V * 2));
        CK(hipMalloc((void **)&s.tags, h->ntiles * 4));
        CK(hipMemsetAsync(s.tags, 0, h->ntiles * 4, h->stream));
    }
    h->ring.resize(params->buffer_size);
    for (int i = 0; i < params->buffer_size; ++i) h->ring[i] = i;
    h->staging = params->buffer_size;
    for (int k = 0; k < 2; ++k) {
        CK(hipMalloc((void **)&h->fused[k].state, h->V * 4));
        CK(hipMalloc((void **)&h->fused[k].tags, h->ntiles * 4));
        CK(hipMemsetAsync(h->fused[k].tags, 0, h->ntiles * 4, h->stream));
    }
    CK(hipMalloc((void **)&h->dir_hist, 3 * GVOM_DIRBINS * 4));
    CK(hipMemsetAsync(h->dir_hist, 0, 3 * GVOM_DIRBINS * 4, h->stream));
    CK(hipMalloc((void **)&h->counters, GVOM_CNT_WORDS * 4));
    CK(hipHostMalloc((void **)&h->counters_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&h->counters_host_dev, h->counters_host, 0));
    memset(h->counters_host, 0, 64);
    CK(hipMemsetAsync(h->counters, 0, GVOM_CNT_WORDS * 4, h->stream));
    CK(hipMalloc((void **)&h->descs_dev, sizeof(MapDesc) * (GVOM_MAX_SLOTS + 1)));
    CK(hipHostMalloc((void **)&h->descs_host, sizeof(MapDesc) * (GVOM_MAX_SLOTS + 1)));
    h->fuse_blocks = ((xy + 63) / 64) * (h->sy_hi - h->sy_lo);
    CK(hipMalloc((void **)&h->blockcounts, (size_t)(h->fuse_blocks > 0 ? h->fuse_blocks : 1) * 4));
    CK(hipMemsetAsync(h->blockcounts, 0, (size_t)(h->fuse_blocks > 0 ? h->fuse_blocks : 1) * 4, h->stream));
    h->cnt_blocks = h->fuse_blocks;
    if (!sharded && params->buffer_size == 1 && xy % 16 == 0 && zs >= 4) {       // eager fusion possible (with or without statistics)
        CK(hipMalloc((void **)&h->blockcounts2, (size_t)(h->fuse_blocks > 0 ? h->fuse_blocks : 1) * 4));
        CK(hipMemsetAsync(h->blockcounts2, 0, (size_t)(h->fuse_blocks > 0 ? h->fuse_blocks : 1) * 4, h->stream));
        CK(hipMalloc((void **)&h->hmaps2, h->cells2d * 24));
    }
    h->hs = 3 * xy;
    if (sharded) CK(pool_get(exportable_size(h->cells2d * 24), (void **)&h->hmaps, &h->fixed_gen[2], &h->holds_pooled));
    else CK(hipMalloc((void **)&h->hmaps, h->cells2d * 24));
    h->height = h->hmaps; h->inferred = h->hmaps + xy;
    double **maps[4] = {&h->slope_x, &h->slope_y, &h->rough, &h->guessed};
    for (auto m : maps) CK(hipMalloc((void **)m, h->cells2d * 8));

    // coherent: gvom_combine_maps learns of completion from a flag a later kernel stores (finish_combine), not from a
    // stream synchronisation; k_map2d's stores to non-coherent host memory would only be guaranteed visible after one
    CK(hipHostMalloc(&h->out_host, h->cells2d * 20, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&h->out_host_dev, h->out_host, 0));
    if (sharded) {
        // rank exchange regions (DESIGN.md "Multi-GPU"): a quad = 4 storage rows x 64 sx at one sz
        h->x_Q = (size_t)(xy / 4) * zs * h->nseg;
        h->x_myQ = h->x_Q / world;
        CK(pool_get(exportable_size(h->x_Q * 4), (void **)&h->x_send_ids, &h->fixed_gen[0], &h->holds_pooled));
        CK(pool_get(exportable_size(h->x_Q * 1024), &h->x_send_pay, &h->fixed_gen[1], &h->holds_pooled));
        CK(hipMalloc((void **)&h->x_recv_ids, h->x_Q * 4));
        CK(hipMalloc(&h->x_recv_pay, h->x_Q * 1024));
        CK(hipMalloc((void **)&h->x_qcnt, (size_t)world * 64));
        CK(hipMalloc((void **)&h->x_ecnt, (size_t)world * 64));
        CK(hipMalloc((void **)&h->x_spcnt, (size_t)world * 64));
        CK(hipMemsetAsync(h->x_qcnt, 0, (size_t)world * 64, h->stream));
        CK(hipMemsetAsync(h->x_ecnt, 0, (size_t)world * 64, h->stream));
        CK(hipMemsetAsync(h->x_spcnt, 0, (size_t)world * 64, h->stream));
        CK(hipHostMalloc((void **)&h->x_host, (size_t)(3 * world + 2) * 8, hipHostMallocMapped | hipHostMallocCoherent));
        CK(hipHostGetDevicePointer((void **)&h->x_host_dev, h->x_host, 0));
        memset(h->x_host, 0, (size_t)(3 * world + 2) * 8);
        h->x_recv_ep_off.assign(world + 1, 0);
        h->x_recv_sp_off.assign(world + 1, 0);
    }
    for (auto &e : h->ev) CK(hipEventCreate(&e));
    CK(hipStreamSynchronize(h->stream));
#undef CK
    *out = h;
    return GVOM_OK;
}

// Somebody reads the per-voxel statistics: a handle that computes them ON DEMAND keeps doing so, or starts again (the caller of
// this read finds no data; the scans that follow carry statistics, and the fused map does once every ring slot does).
static void stats_demand(gvom_handle *h)
{
    h->stats_idle = 0;
    if (h->stats_auto && !h->stats) { h->stats = true; h->stats_release = false; }
}

// Statistics on demand, switched off: their buffers (the slots' metrics / own-voxel moments / row tables, the fused maps' metrics,
// the kept clouds -- 40 bytes per voxel and fused map alone: 2.7 GB on the 512^2 x 128 grid) go back to the allocator; a later
// demand allocates them again.  Everything that can still read or write them runs on the statistics stream: waited for first
// (a scan of another thread whose trace is still in flight writes its slot's tables: the release then waits for a later combine).
static void release_statistics_buffers(gvom_handle *h)
{
    h->stats_release = false;
    (void)hipStreamSynchronize(h->stream_s);
    h->s_pending = h->fs_pending = false;
    h->before_valid[0] = h->before_valid[1] = false;
    auto fb = [](Buf &b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; };
    for (auto &sl : h->slots) { fb(sl.metrics); fb(sl.base); fb(sl.rowvox); sl.has_metrics = false; }
    for (int k = 0; k < 2; ++k) { fb(h->fused[k].metrics); h->fused[k].has_metrics = false; }
    fb(h->world_pts[0]); fb(h->world_pts[1]); fb(h->flink[0]); fb(h->flink[1]);
    h->spec_has_metrics = false;
}

// A failed scan must not leak into the next one: k_trace may already have added to the dense
// accumulators and raised the in-grid flag.  Clears them (stream-ordered, best effort).
void scan_abort(gvom_handle *h)
{
    (void)hipMemsetAsync(h->hit, 0, h->acc_elems * 4, h->stream);
    (void)hipMemsetAsync(h->total, 0, h->acc_elems * 4, h->stream);
    (void)hipMemsetAsync(h->mh, 0, h->acc_elems * 4, h->stream);
    (void)hipMemsetAsync(h->counters, 0, GVOM_CNT_WORDS * 4, h->stream);
    if (h->stream_s) (void)hipStreamSynchronize(h->stream_s);
    h->s_pending = h->fs_pending = false;
    h->before_valid[0] = h->before_valid[1] = false;
    (void)hipStreamSynchronize(h->stream);
    (void)hipGetLastError();
    h->pending = false;
    h->scan_inflight = false;
}

// Tile epochs are 32-bit and only grow (two per step).  Before the counter can wrap, every live map
// is re-tagged with a small epoch and every other tag is zeroed, so a stale tile can never collide
// with a future epoch.  (2^32 epochs = ~3 days at 7.7 kHz.)
int renumber_epochs(gvom_handle *h);

// host waits for everything the handle has enqueued (both streams)
hipError_t sync_streams(gvom_handle *h)
{
    if (h->mapped_unjoined || h->fuse_b_unjoined) {
        hipError_t e = hipStreamSynchronize(h->stream_b);
        if (e != hipSuccess) return e;
        h->mapped_unjoined = false;
        h->fuse_b_unjoined = false;
    }
    if (h->s_pending || h->fs_pending) {
        hipError_t e = hipStreamSynchronize(h->stream_s);
        if (e != hipSuccess) return e;
        h->s_pending = h->fs_pending = false;
        h->before_valid[0] = h->before_valid[1] = false;
    }
    return hipStreamSynchronize(h->stream);
}
// main stream waits (on the device) for a fusion still running on the second stream: before it overwrites a
// ring slot that fusion reads, or reads the fused map it writes
hipError_t join_fuse_stream(gvom_handle *h)
{
    if (!h->fuse_b_unjoined) return hipSuccess;
    h->fuse_b_unjoined = false;
    return hipStreamWaitEvent(h->stream, h->ev_fuse_b, 0);
}
// main stream waits (on the device) for a k_map2d still running on the second stream: it reads what
// the next fusion writes (fused double buffer, height maps, block counts)
hipError_t join_map_stream(gvom_handle *h)
{
    if (!h->mapped_unjoined) return hipSuccess;
    h->mapped_unjoined = false;
    h->fuse_b_unjoined = false;                            // (k_map2d runs behind that fusion on the same stream)
    return hipStreamWaitEvent(h->stream, h->ev_mapped, 0);
}

// the read hooks launch on the main stream and read what an asynchronous combine's kernels write on the second
hipError_t join_second_stream(gvom_handle *h)
{
    hipError_t e = join_map_stream(h);
    if (e == hipSuccess) e = join_fuse_stream(h);
    if (e == hipSuccess && h->s_pending) e = hipStreamWaitEvent(h->stream, h->ev_sdone, 0);   // the statistics stream's work
    return e;
}

// Optional per-voxel statistics of the scan just encoded, on the statistics stream beside whatever
// follows.  nrows: candidate compact rows; extra / n_extra: returns received from other ranks (sharded map).
int enqueue_scan_stats(gvom_handle *h, const ScanParams &P, int dtype, Slot &st, int64_t n, int64_t nrows, const void *extra, int64_t n_extra)
{
    const uint32_t par = h->stats_scan & 1u;
    HIPCHK(h, hipEventRecord(h->ev_before[par], h->stream_s));   // the statistics stream's work up to here
    h->before_valid[par] = true;
    ++h->stats_scan;
    HIPCHK(h, hipEventRecord(h->ev_enc_s, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->stream_s, h->ev_enc_s, 0));
    HIPCHK(h, gvom_launch_stats(h->stream_s, P, dtype, h->world_pts[par].p, n, st.state, st.tags,
                                h->prm.xy_eigen_dist, h->prm.z_eigen_dist, (double *)st.base.p, (double *)st.metrics.p,
                                (const uint32_t *)st.rowvox.p, nrows, extra, n_extra));
    HIPCHK(h, hipEventRecord(h->ev_sdone, h->stream_s));
    h->s_pending = true;
    return GVOM_OK;
}

// Scan kernels up to (not including) the commit.  `dev_pts` is device memory.
// Waits until the GPU has published sequence number `seq` in the 64-bit host-mapped word `flag` (high
// half, or the whole word).  `lk` (the handle mutex) is RELEASED while waiting, so combine_maps from
// another thread is not locked out for the length of a trace.  A wait that took long the last time (c5:
// milliseconds) first SLEEPS most of that time away (all but a fifth, at least 250 us) and spins over the rest: the core is free meanwhile
// and the wake-up still comes within microseconds (a sleep has a granularity of ~60 us; spinning with short
// sleeps in between overshot a 250 us wait by 30 us).  `last_ns`: this wait's duration the previous time.
bool wait_published(gvom_handle *h, std::unique_lock<std::mutex> &lk, volatile unsigned long long *flag, uint32_t seq,
                    bool high_half, double *last_ns)
{
    auto done = [&]() { return (uint32_t)(high_half ? (*flag >> 32) : *flag) == seq; };
    lk.unlock();
    const double start = now_ns();
    bool slept = false;
    if (last_ns && *last_ns > 6.0e5 && !done()) {
        // all but the last fifth, at least 250 us (a sleep may run 100 us over and more)
        const double margin = *last_ns * 0.2 > 2.5e5 ? *last_ns * 0.2 : 2.5e5;
        usleep((useconds_t)((*last_ns - margin) * 1e-3));
        slept = true;
    }
    // the estimate follows only what has been OBSERVED: completion seen while spinning gives the true duration;
    // a flag already set after the sleep means the sleep was too long -- halve it (an outlier, e.g. a first
    // call that allocated, cannot keep every later wait long); never more than 20 ms
    const bool overslept = slept && done();
    unsigned spins = 0;
    bool ok = true;
    while (!done()) {
        _mm_pause();
        if ((++spins & 0x3ff) == 0) {
            const double waited = now_ns() - start;
            if (waited > 2.0e9) { ok = false; break; }                // device trouble: the caller falls back
            if (waited > 2.0e7) usleep(50);                           // far beyond anything expected: stop burning the core
        }
    }
    if (last_ns) {
        const double est = overslept ? *last_ns * 0.5 : now_ns() - start;
        *last_ns = est < 2.0e7 ? est : 2.0e7;
    }
    lk.lock();
    return ok;
}

// what a fusion into the frame `origin` needs besides its sources (fuse_impl, eager_launch)
void fill_fuse_frame(const gvom_handle *h, const int64_t origin[3], FuseParams &P)
{
    const gvom_params &p = h->prm;
    memset(&P, 0, sizeof P);
    P.xy = p.xy_size; P.zs = p.z_size;
    P.om[0] = (int)floor_mod(origin[0], p.xy_size);
    P.om[1] = (int)floor_mod(origin[1], p.xy_size);
    P.om[2] = (int)floor_mod(origin[2], p.z_size);
    P.sy_lo = h->sy_lo; P.sy_hi = h->sy_hi;
    P.nseg = h->nseg;
    P.hs = h->hs;
    for (int k = 0; k < 3; ++k) { P.origin[k] = (double)origin[k]; P.ego[k] = h->ego[k]; }
    P.xy_res = p.xy_resolution; P.z_res = p.z_resolution;
    P.radius2 = p.robot_radius * p.robot_radius;
    P.ground_to_lidar_height = p.ground_to_lidar_height;
}

// k_encfuse behind k_trace (eager fusion, see gvom_handle::hmaps2): the staging slot encoded and fused with the previous
// map into the SPARE fused / height / count buffers.  Nothing the handle's readers look at changes before fuse_impl adopts it.
int eager_launch(gvom_handle *h, const ScanParams &P, Slot &st, const int64_t origin[3], uint32_t seq)
{
    const gvom_params &p = h->prm;
    const int nxt = h->has_combined ? 1 - h->cur : 0;
    Fused &F = h->fused[nxt];
    const Fused *prev = (h->has_combined && h->fused[h->cur].valid) ? &h->fused[h->cur] : nullptr;
    FuseParams FP;
    fill_fuse_frame(h, origin, FP);
    FP.nslots = 1; FP.has_prev = prev ? 1 : 0;
    FP.nz = h->tune_encfuse & 15; FP.cpw = (h->tune_encfuse >> 4) & 1;      // A/B knob (waves per column block, XCD pairing off)
    MapDesc pd;
    memset(&pd, 0, sizeof pd);
    if (prev) {
        pd.state = prev->state; pd.rows = (const uint4 *)prev->rows.p;
        pd.d[0] = clamp_delta(origin[0] - prev->origin[0], p.xy_size);
        pd.d[1] = clamp_delta(origin[1] - prev->origin[1], p.xy_size);
        pd.d[2] = clamp_delta(origin[2] - prev->origin[2], p.z_size);
        pd.epoch = prev->epoch; pd.tags = prev->tags;
    }
    int nw, nblocks; size_t row_cap;
    gvom_encfuse_shape(p.xy_size, p.z_size, FP.nz, &nw, &nblocks, &row_cap);
    if (row_cap >= 2147483648ull) { h->err = "fused row space exceeds 31 bits"; return GVOM_ERR_CAPACITY; }
    int rc;
    if ((rc = ensure(h, F.rows, row_cap * 16))) return rc;
    F.valid = false;                                       // (the spare buffer: nobody reads it as a map)
    FP.epoch = ++h->epoch;
    // a pending k_fuse_stats that reads THIS buffer's states / tile tags -- the merge of a fusion through k_fuse1 / k_fuse4 (its
    // "previous map"), or of a dropped speculation (its target) -- must be done before this kernel rewrites them (the wait sits
    // between k_trace and this kernel).  The merge of an ADOPTED speculation does not: it reads its previous map through a link table.
    if (h->fs_pending && h->fs_reads[nxt]) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_fsdone, 0));
    h->spec_has_metrics = false;
    int32_t *flink = nullptr;
    if (h->stats) {
        if ((rc = ensure(h, h->flink[nxt], row_cap * 4))) return rc;
        flink = (int32_t *)h->flink[nxt].p;
        // the statistics half of this speculative fusion (enqueued by scan_launch behind the scan's own statistics): sources = the
        // staging slot (same frame: no shift) and the previous map if it carries statistics of its own
        if ((rc = ensure(h, F.metrics, row_cap * 40))) return rc;
        h->spec_FP = FP;
        memset(&h->spec_KD, 0, sizeof h->spec_KD);
        MapDesc &sd = h->spec_KD.d[0];
        sd.state = st.state; sd.rows = (const uint4 *)st.crows.p; sd.epoch = st.epoch; sd.tags = st.tags; sd.metrics = st.metrics.p;
        if (prev) { h->spec_KD.d[1] = pd; h->spec_KD.d[1].metrics = prev->has_metrics ? prev->metrics.p : nullptr; h->spec_KD.d[1].link = flink; }
        h->spec_has_metrics = true;
    }
    HIPCHK(h, gvom_launch_encfuse(h->stream, P, FP, pd, flink, h->hit, h->total, h->mh, st.state, (uint4 *)st.crows.p, st.tags,
                                  F.state, (uint4 *)F.rows.p, F.tags, h->blockcounts2, h->hmaps2, h->hmaps2 + p.xy_size,
                                  h->counters, (unsigned long long *)h->counters_host_dev, seq));
    h->spec_valid = true; h->spec_nxt = nxt; h->spec_slot = h->staging; h->spec_blocks = nblocks; h->spec_epoch = FP.epoch;
    h->spec_origin[0] = origin[0]; h->spec_origin[1] = origin[1]; h->spec_origin[2] = origin[2];
    h->last_scan_spec = true;
    return GVOM_OK;
}

int scan_launch(gvom_handle *h, std::unique_lock<std::mutex> &lk, const void *dev_pts, int64_t n, int64_t stride_elems,
                int dtype, const double *tf, bool shard_local = false)
{
    const gvom_params &p = h->prm;
    int64_t origin[3];
    origin[0] = (int64_t)floor((h->ego[0] / p.xy_resolution) - p.xy_size / 2.0);     // gvom.py:124
    origin[1] = (int64_t)floor((h->ego[1] / p.xy_resolution) - p.xy_size / 2.0);
    origin[2] = (int64_t)floor((h->ego[2] / p.z_resolution) - p.z_size / 2.0);
    int rc;
    if (h->epoch >= 0xFFFFFF00u && (rc = renumber_epochs(h))) return rc;
    ScanParams P;
    fill_scan_params(h, origin, tf, P, n);
    // (a combine in flight -- combine_maps_async, or a second thread inside combine_maps -- has k_map2d's PCIe stores running
    // beside this trace: raised trace waves would take its issue slots (pipelined m256 91.7 us per step against 89.6, c3 122.0 / 115.5))
    if (h->pending_combine && h->tune_prio < 0) P.prio_div = 0;
    h->last_knobs[0] = P.nsegs; h->last_knobs[1] = P.lc_period; h->last_knobs[2] = P.ep_row; h->last_knobs[3] = P.prio_div;
    h->last_knobs[4] = 1 << P.ilv_lg;
    Slot &st = h->slots[h->staging];
    // the staging slot may still be a SOURCE of a fusion running on the second stream (asynchronous combine: the
    // second scan after gvom_combine_begin writes the slot the ring has just evicted)
    if (h->fuse_b_unjoined && ((h->fuse_b_slots >> h->staging) & 1ull)) HIPCHK(h, join_fuse_stream(h));
    st.epoch = ++h->epoch;                                 // tiles stamped by this scan
    P.epoch = st.epoch;
    st.has_metrics = h->stats;
    h->scan_inflight = true;
    const size_t esz = dtype == GVOM_DTYPE_F32 ? 4 : 8;
    // compact rows are indexed by return (the row of an occupied voxel = the index of one of its returns)
    const size_t cap = std::max<size_t>(1, (size_t)n);
    if ((rc = ensure(h, st.crows, cap * 16))) return rc;
    Buf &wpts = h->world_pts[h->stats_scan & 1u];
    // (a sharded map's received endpoints get rows behind the rank's own returns: room for as many again; gvom_shard_scan_merge
    // grows the arrays, keeping their contents, if more arrive)
    const size_t scap = h->sharded ? 2 * cap : cap;
    if (h->stats && ((rc = ensure(h, wpts, (size_t)(n > 0 ? n : 1) * 3 * esz)) || (rc = ensure(h, st.metrics, scap * 80)) ||
                     (rc = ensure(h, st.base, scap * 8 * GVOM_BASE_PITCH)) || (rc = ensure(h, st.rowvox, scap * 4)))) return rc;
    double t0 = now_ns();
    const uint32_t seq = ++h->scan_seq;
    if (h->profiling) HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
    // voxels are looked up with 32-bit integer arithmetic: exact while |origin| < 2^30 (2e8 m at
    // 0.2 m voxels); beyond that the kernels take the reference's literal f64 form
    bool big = false;
    for (int k = 0; k < 3; ++k)
        if (origin[k] >= (1ll << 30) || origin[k] <= -(1ll << 30)) big = true;
    if (h->stats) {
        // what the statistics stream may still be reading of the things this scan rewrites (see ev_before): the
        // previous scan's own statistics are NOT among them and keep running beside this scan's trace
        const uint32_t prev_par = (h->stats_scan + 1u) & 1u;
        if (!h->stats_prev_committed && h->s_pending) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_sdone, 0));
        else if (h->before_valid[prev_par]) {
            // (normally long complete: then no wait packet goes in front of the trace)
            if (hipEventQuery(h->ev_before[prev_par]) == hipSuccess) h->before_valid[prev_par] = false;
            else { (void)hipGetLastError(); HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_before[prev_par], 0)); }
        }
        HIPCHK(h, hipMemsetAsync(st.rowvox.p, 0xFF, st.rowvox.bytes, h->stream));   // no row claimed yet
    }
    ShardExchange X;
    X.ep_send = nullptr; X.ep_cnt = nullptr; X.ep_cap = 0; X.sp_send = nullptr; X.sp_cnt = nullptr;
    if (h->sharded) {
#ifdef GVOM_HOOKS
        if (h->tune_churn > 0 && h->x_send_eps.p) {            // test hook (gvom_set_tuning "churn"): a fresh allocation every scan
            h->retired.push_back(h->x_send_eps);                 // (as a region that grows: retired, not freed)
            h->x_send_eps.p = nullptr; h->x_send_eps.bytes = 0;
        }
#endif
        if ((rc = ensure(h, h->x_send_eps, (size_t)h->world * (size_t)(n > 0 ? n : 1) * 8))) return rc;
        h->x_ep_cap = n > 0 ? n : 1;
        X.ep_send = (uint2 *)h->x_send_eps.p; X.ep_cnt = h->x_ecnt; X.ep_cap = (long)h->x_ep_cap;
        if (h->stats && h->world > 1) {
            if ((rc = ensure(h, h->x_send_sp, (size_t)h->world * (size_t)h->x_ep_cap * 3 * esz))) return rc;
            X.sp_send = h->x_send_sp.p; X.sp_cnt = h->x_spcnt;
        }
    }
    P.tl = nullptr; P.tl_words = 0;
#ifdef GVOM_DIAG
    if (gvom_diag_env("GVOM_TRACE_TIMELINE") && n > 0) {
        h->tl_grid[0] = (int)((n + 511) / 512); h->tl_grid[1] = P.nsegs + (P.ep_row >= 0 ? 1 : 0);
        P.prof_on = gvom_diag_env("GVOM_TRACE_STEPPROF");
        // (+ step profiles: 128 words for every 64th wave)
        const size_t bytes = (size_t)h->tl_grid[0] * h->tl_grid[1] * 8 * 32 + 64 + ((size_t)h->tl_grid[0] * h->tl_grid[1] * 8 / 64 + 1) * 1024;
        P.tl_words = (long)h->tl_grid[0] * h->tl_grid[1] * 8 * 4;
        if ((rc = ensure(h, h->tl, bytes))) return rc;
        HIPCHK(h, hipMemsetAsync(h->tl.p, 0, bytes, h->stream));
        P.tl = (unsigned long long *)h->tl.p;
    }
#endif
    hipError_t le = hipSuccess;
    // the layout probe's LAST answer, read before this scan's probe is launched: every kernel of the earlier scans has completed
    // (their process_pointcloud returned behind k_trace), so which scan starts to sort is deterministic (read behind
    // the launch, scan k or k + 1 saw the new verdict depending on how fast the probe ran)
    const unsigned long long probe_word = h->counters_host ? *(volatile unsigned long long *)(h->counters_host + 8) : 0ull;
    if (h->tune_ilv == 0 && n >= 8192) {
        // layout probe: only for clouds whose length is STABLE (the same as the previous scan's: a node that drops invalid
        // returns hands over a different length every scan -- its sub-clouds do not start at multiples of n / K, nothing to find,
        // nothing to pay), on the second cloud of a length and every 32nd scan after it; in front of k_trace (the caller may
        // free the cloud as soon as this call has returned, i.e. once k_trace is done); its answer serves LATER scans
        // A stream whose length CHANGES from scan to scan (invalid returns dropped) is looked at every 8th scan all the same -- for the
        // "is this order any good for the trace" half of the answer only, which does not depend on the exact length
        const bool stable = n == h->last_n && (n != h->probe_n || ++h->probe_age >= 32);
        const bool varying = n != h->last_n && ++h->probe_var_age >= 8;
        if (stable || varying) {
            if (stable) { h->probe_n = n; h->probe_age = 0; }
            h->probe_var_age = 0;
            le = gvom_launch_layout_probe(h->stream, P, dtype, dev_pts, stride_elems, n, 2, (unsigned long long *)(h->counters_host_dev + 8));
            if (le != hipSuccess) { scan_abort(h); HIPCHK(h, le); }
        }
        h->last_n = n;
    }
    P.perm = nullptr;
    {   // directional order: forced, or the probe's verdict on the previous cloud of this length ("scattered", bit 3 of its answer)
        int sort = h->tune_dirsort == 1 || h->tune_dirsort == 2 ? h->tune_dirsort : 0;   // 1: cube cells, 2: elevation rows
        if (h->tune_dirsort == 0 && h->tune_ilv == 0 && h->counters_host) {
            const unsigned long long w = probe_word;
            const int64_t np_ = (int64_t)(w >> 8);
            // (the verdict on a cloud of about this length -- within a quarter -- holds for this one)
            if (np_ > 0 && (np_ == n || (n > np_ - np_ / 4 && n < np_ + np_ / 4))) sort = (int)((w >> 3) & 3ull);
        }
        h->last_dirsort = 0;
        if (sort && n >= 256 && n < (1ll << 31)) {
            if ((rc = ensure(h, h->dir_keys, (size_t)n * 2)) || (rc = ensure(h, h->dir_perm, (size_t)n * 4))) return rc;
            uint32_t *hist = h->dir_hist + (h->dir_flip & 1u) * GVOM_DIRBINS, *next = h->dir_hist + ((h->dir_flip + 1u) & 1u) * GVOM_DIRBINS;
            ++h->dir_flip;
            le = gvom_launch_dirbin(h->stream, P, sort == 2 ? 2 : 1, dtype, dev_pts, stride_elems, n, (uint16_t *)h->dir_keys.p, hist, next,
                                    h->dir_hist + 2 * GVOM_DIRBINS, (uint32_t *)h->dir_perm.p);
            if (le != hipSuccess) { scan_abort(h); HIPCHK(h, le); }
            P.perm = (const uint32_t *)h->dir_perm.p;
            P.ilv_lg = 0; P.ilv_len = n;
            h->last_dirsort = sort == 2 ? 2 : 1;
            h->last_knobs[4] = 1;
        }
    }
    le = gvom_launch_trace(h->stream, P, X, dtype, big, dev_pts, stride_elems, n,
                                      h->stats ? wpts.p : nullptr, h->hit, h->total, h->mh, st.state,
                                      st.tags, h->counters, h->stats ? (double *)st.metrics.p : nullptr,
                                      h->stats ? (double *)st.base.p : nullptr,
                                      h->stats ? (uint32_t *)st.rowvox.p : nullptr);
    if (le != hipSuccess) { scan_abort(h); HIPCHK(h, le); }
    if (h->profiling) HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
    // (a sharded map of ONE rank has no other ranks' rows: nothing to pack, no counts to publish and to wait for -- the scan
    // is encoded right behind the trace, as on an unsharded handle, and gvom_shard_scan_merge only commits it)
    const bool solo = shard_local && h->world == 1;
    h->solo_encoded = solo;
    if (shard_local && !solo) {
        // sharded map: the ray passes in other ranks' rows are packed for their owners; the counts go
        // to host-mapped memory (the caller sizes the exchange with them); k_encode follows in
        // scan_merge, once the other ranks' contributions have been added
        le = gvom_launch_pack(h->stream, P, h->total, st.tags, h->x_send_ids, h->x_send_pay, h->x_qcnt, h->x_ecnt, h->x_spcnt,
                              h->counters, h->x_host_dev, seq);
        if (le != hipSuccess) { scan_abort(h); HIPCHK(h, le); }
        volatile unsigned long long *flag = (volatile unsigned long long *)(h->x_host + 2 * h->world + 1);
        if (!wait_published(h, lk, flag, seq, false, &h->last_wait_ns[0])) {
            hipError_t se = hipStreamSynchronize(h->stream);
            if (se != hipSuccess || (uint32_t)*flag != seq) {
                scan_abort(h);
                h->err = "sharded scan: pack counts were never published";
                return GVOM_ERR_HIP;
            }
        }
        h->pending_any = h->x_host[2 * h->world] != 0;
        h->pending_P = P;
        h->pending_dtype = dtype;
        st.count = -1;
        st.origin[0] = origin[0]; st.origin[1] = origin[1]; st.origin[2] = origin[2];
        st.stats_valid = false;
        st.stats.points = n;
        h->pending = true;
        h->pending_n = n;
        return GVOM_OK;
    }
    // one-slot rings: encode + fuse in one pass, speculating that combine_maps comes next (see gvom_handle::hmaps2)
    const bool eager = h->hmaps2 && h->tune_eager != 0 && (h->tune_eager == 1 || h->eager_waste < 3) && !gvom_diag_env("GVOM_TRACE_DEBUG");
    h->last_scan_spec = false;
    if (eager) {
        if ((rc = eager_launch(h, P, st, origin, seq))) { scan_abort(h); return rc; }
        st.has_code16 = false;
    } else {
        // the 16-bit codes are read by k_fuse4 only: a one-slot ring fuses through k_fuse1 (unless the A/B knob says otherwise)
        const bool codes = st.code16 && (p.buffer_size > 1 || h->tune_fuse1 == 1 || h->sharded);
        le = gvom_launch_encode(h->stream, P, h->hit, h->total, h->mh, st.state, codes ? st.code16 : nullptr, (uint4 *)st.crows.p,
                                st.tags, h->counters,
                                (unsigned long long *)h->counters_host_dev, seq, h->resident_blocks);
        if (le != hipSuccess) { scan_abort(h); HIPCHK(h, le); }
        st.has_code16 = codes;
    }
    if (h->profiling) { HIPCHK(h, hipEventRecord(h->ev[2], h->stream)); h->ev_scan = true; }
    if (h->stats && (rc = enqueue_scan_stats(h, P, dtype, st, n, n, nullptr, 0))) return rc;
    if (eager && h->spec_has_metrics) {
        // behind k_stats / k_stats_gather on the statistics stream (which waited for k_encfuse): the merge of the speculative fusion
        Fused &F = h->fused[h->spec_nxt];
        HIPCHK(h, gvom_launch_fuse_stats(h->stream_s, h->spec_FP, h->spec_KD, nullptr, F.state, F.tags, (float *)F.metrics.p));
        HIPCHK(h, hipEventRecord(h->ev_fsdone, h->stream_s));
        HIPCHK(h, hipEventRecord(h->ev_sdone, h->stream_s));
        h->s_pending = h->fs_pending = true;
        h->fs_reads[h->spec_nxt] = true; h->fs_reads[1 - h->spec_nxt] = false;      // (its previous map through the link table)
    }
    HT(h, 0, t0);                                        // scan: launches
    // Wait only for k_trace: k_encode's first thread publishes {seq, any-in-grid} to host-mapped
    // memory.  The caller gets control back while k_encode still runs; everything it can do next
    // with this handle is stream-ordered behind it.
    {
        volatile unsigned long long *flag = (volatile unsigned long long *)h->counters_host;
        if (!wait_published(h, lk, flag, seq, true, &h->last_wait_ns[0])) {
            hipError_t se = hipStreamSynchronize(h->stream);
            if (se != hipSuccess || (uint32_t)(*flag >> 32) != seq) {
                scan_abort(h);
                h->err = se != hipSuccess ? std::string("scan failed: ") + hipGetErrorString(se)
                                          : "scan completion was never published";
                return GVOM_ERR_HIP;
            }
        }
    }
    HT(h, 1, t0);                                        // scan: wait
    if (h->host_timing) h->host_calls++;
    {
        const unsigned long long fl = *(volatile unsigned long long *)h->counters_host;
        h->pending_any = (fl & 0x80000000ull) != 0;        // some return landed in the grid
    }
    st.count = -1;                                         // occupied voxels: counted on demand (test hooks)
    st.origin[0] = origin[0]; st.origin[1] = origin[1]; st.origin[2] = origin[2];
    st.stats_valid = false;
    st.stats.points = n;
    h->pending = true;
    h->pending_n = n;
    return GVOM_OK;
}

void scan_commit(gvom_handle *h, bool accept)
{
    h->scan_inflight = false;
    if (!h->pending) return;
    h->pending = false;
    h->stats_prev_committed = accept;
    if (!accept) { h->spec_valid = false; h->last_scan_spec = false; return; }   // gvom.py:148-150: ring untouched (a speculative fusion of it is dropped)
    const int b = h->buffer_index;                         // gvom.py:163-175
    const int old = h->ring[b];
    h->ring[b] = h->staging;
    h->staging = old;
    h->slots[h->ring[b]].filled = true;
    h->fresh_scan = true;
    h->slots[old].filled = false;
    h->last_buffer_index = b;
    h->buffer_index = (b + 1 >= h->prm.buffer_size) ? 0 : b + 1;
}

int renumber_epochs(gvom_handle *h)
{
    HIPCHK(h, sync_streams(h));
    h->spec_valid = false;                                 // (its fused tiles carry an epoch of the old numbering)
    uint32_t next = 0;
    for (size_t k = 0; k < h->slots.size(); ++k) {
        Slot &sl = h->slots[k];
        // the staging slot's tiles are dead -- unless a scan is in flight in it (a combine thread can get here while
        // the scan's thread waits for k_trace with the handle mutex released, or between the two halves of a sharded
        // scan): its kernels have completed (sync above), its tiles keep a live epoch and the commit finds it intact
        const bool inflight = (int)k == h->staging && (h->scan_inflight || h->pending);
        const uint32_t fresh = (sl.filled || inflight) ? ++next : 0u;
        HIPCHK(h, gvom_launch_retag(h->stream, sl.tags, h->ntiles, sl.epoch, fresh));
        sl.epoch = fresh;
        if (inflight) h->pending_P.epoch = fresh;         // gvom_shard_scan_merge's kernels stamp / test this epoch
    }
    for (int k = 0; k < 2; ++k) {
        Fused &f = h->fused[k];
        const uint32_t fresh = f.valid ? ++next : 0u;
        HIPCHK(h, gvom_launch_retag(h->stream, f.tags, h->ntiles, f.epoch, fresh));
        f.epoch = fresh;
    }
    h->epoch = next;
    HIPCHK(h, sync_streams(h));
    return GVOM_OK;
}

int process_impl(gvom_handle *h, const void *xyz, bool on_device, int64_t n, int64_t row_stride_bytes,
                 int dtype, const double ego[3], const double *tf, bool defer, const int64_t *off_bytes = nullptr,
                 bool widen_f32 = false)
{
    if (!h || !ego || n < 0 || (dtype != GVOM_DTYPE_F32 && dtype != GVOM_DTYPE_F64))
        return GVOM_ERR_INVALID;
    // widen_f32: float32 records (esz = 4 for addressing) scanned by the float64 kernels
    const size_t esz = (dtype == GVOM_DTYPE_F32 || widen_f32) ? 4 : 8;
    if (widen_f32 && dtype != GVOM_DTYPE_F64) return GVOM_ERR_INVALID;
    if (n > 0 && (!xyz || row_stride_bytes < (int64_t)(3 * esz) || row_stride_bytes % esz != 0))
        return GVOM_ERR_INVALID;
    if (n >= 2147483647LL) return GVOM_ERR_CAPACITY;
    int64_t last_field = 2 * (int64_t)esz;                               // byte offset of the last field read
    if (off_bytes) {
        last_field = 0;
        for (int k = 0; k < 3; ++k) {
            if (off_bytes[k] < 0 || off_bytes[k] % (int64_t)esz != 0 || off_bytes[k] + (int64_t)esz > row_stride_bytes)
                return GVOM_ERR_INVALID;
            if (off_bytes[k] > last_field) last_field = off_bytes[k];
        }
    }
    std::lock_guard<std::mutex> scan_lk(h->scan_mu);      // scans of one handle are serialised (one staging slot, one set of accumulators)
    std::unique_lock<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    for (int k = 0; k < 3; ++k) h->in_off[k] = off_bytes ? (int)(off_bytes[k] / (int64_t)esz) : k;
    h->in_f32 = widen_f32;
    h->ego[0] = ego[0]; h->ego[1] = ego[1]; h->ego[2] = ego[2];       // gvom.py:102-104
    // a sharded scan whose second half never came (the exchange failed between gvom_shard_scan_local and
    // gvom_shard_scan_merge): k_trace's additions to this rank's rows were never encoded or zeroed
    if (h->pending && h->sharded) scan_abort(h);
    h->pending = false;
    if (h->spec_valid) {                                   // a scan behind a scan: the speculative fusion of the first is dropped
        h->spec_valid = false;
        if (h->eager_waste < 4) ++h->eager_waste;
        ++h->eager_stat[1];
    }
    if (h->sharded && !defer) { h->err = "a sharded handle scans through gvom_shard_scan_local / gvom_shard_scan_merge"; return GVOM_ERR_INVALID; }
    if (n == 0 && !defer) return GVOM_EMPTY_CLOUD;                     // gvom.py:107-109 (a rank's share of a sharded scan may be empty)
    const void *dev = xyz;
    if (!on_device && n > 0) {
        int rc = ensure(h, h->in_pts, (size_t)n * row_stride_bytes);
        if (rc) return rc;
        const size_t up_bytes = (size_t)(n - 1) * row_stride_bytes + (size_t)last_field + esz;
        // in_pts is free: the previous scan's k_trace (its only reader) had completed when that call returned
        if (hipStreamQuery(h->stream) == hipErrorNotReady) {
            // the main stream still has work queued (a combine from another thread, the previous k_encode): upload beside it
            lk.unlock();                                  // a pageable source is staged by the calling thread: not under the handle mutex
            const hipError_t ue = hipMemcpyAsync(h->in_pts.p, xyz, up_bytes, hipMemcpyHostToDevice, h->stream_up);
            lk.lock();
            HIPCHK(h, ue);
            HIPCHK(h, hipEventRecord(h->ev_up, h->stream_up));
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_up, 0));
        } else {
            (void)hipGetLastError();
            HIPCHK(h, hipMemcpyAsync(h->in_pts.p, xyz, up_bytes, hipMemcpyHostToDevice, h->stream));
        }
        dev = h->in_pts.p;
    }
    int rc = scan_launch(h, lk, dev, n, row_stride_bytes / (int64_t)esz, dtype, tf, defer);
    if (rc) return rc;
    if (defer) return GVOM_OK;
    const bool accept = h->pending_any;                   // == (global occupied-voxel count > 0), gvom.py:147-150
    scan_commit(h, accept);
    return accept ? GVOM_OK : GVOM_NO_OVERLAP;
}

// z decomposition of k_fuse: chunks of zc levels (16 whenever z_size <= 256), cpw chunks per wave,
// nz waves per workgroup.  Small workgroups (<= 4 waves when possible) keep several of them
// resident per CU, so one workgroup's end-of-column barrier never idles the CU.
int choose_nz(int zs, int *zc, int *cpw)
{
    int nchunks = (zs + 15) / 16;
    if (nchunks < 1) nchunks = 1;
    if (nchunks > 16) nchunks = 16;
    *zc = (zs + nchunks - 1) / nchunks;
    nchunks = (zs + *zc - 1) / *zc;
    int want_waves = 4;
    *cpw = (nchunks + want_waves - 1) / want_waves;
    if (*cpw < 1) *cpw = 1;
    if (*cpw > 4) *cpw = 4;                               // a wave's tiles (16 per chunk) fit one 64-bit mask
    return (nchunks + *cpw - 1) / *cpw;
}

// fusion + column reductions (k_fuse) into fused[1 - cur]; `on`: the stream (nullptr: the main one; the second
// stream for an asynchronous combine, where it is ordered behind the previous k_map2d by itself)
int fuse_impl(gvom_handle *h, hipStream_t on = nullptr)
{
    const hipStream_t fs = on ? on : h->stream;
    const gvom_params &p = h->prm;
    const Slot &last = h->slots[h->ring[h->last_buffer_index]];
    if (!last.filled) return GVOM_EMPTY_BUFFER;                        // gvom.py:179-181
    // before any map descriptor below copies an epoch
    if (h->epoch >= 0xFFFFFF00u) { int rc0 = renumber_epochs(h); if (rc0) return rc0; }
    // statistics on demand: three combines in a row that nobody read the statistics of -> the scans stop computing them
    if (h->stats_auto && h->stats && ++h->stats_idle > 3) { h->stats = false; h->stats_release = true; }
    if (h->spec_valid && !on && h->spec_slot == h->ring[h->last_buffer_index] && h->spec_nxt == (h->has_combined ? 1 - h->cur : 0)) {
        // eager fusion: k_encfuse has (or will have, in stream order) written exactly what this call would compute -- the one
        // slot, the previous map and the ego are what they were when the scan launched it.  Adopt: swap the spares in.
        Fused &S = h->fused[h->spec_nxt];
        S.origin[0] = h->spec_origin[0]; S.origin[1] = h->spec_origin[1]; S.origin[2] = h->spec_origin[2];
        S.epoch = h->spec_epoch;
        S.valid = true;
        S.has_metrics = h->spec_has_metrics;               // (its k_fuse_stats runs, or has run, on the statistics stream: readers join it)
        std::swap(h->hmaps, h->hmaps2);
        h->height = h->hmaps; h->inferred = h->hmaps + h->prm.xy_size;
        std::swap(h->blockcounts, h->blockcounts2);
        h->cnt_blocks = h->spec_blocks;
        h->cur = h->spec_nxt;
        h->has_combined = true;
        h->maps_valid = false;
        h->spec_valid = false;
        h->eager_waste = 0;
        ++h->eager_stat[0];
        h->last_scan_spec = false; h->fresh_scan = false;
        h->stage_ms[3] = 0.0f;                             // (the fusion's time is inside the scan's second kernel)
        if (h->stats_release && !h->scan_inflight) release_statistics_buffers(h);
        return GVOM_OK;
    }
    if (h->spec_valid) { h->spec_valid = false; ++h->eager_stat[1]; }
    else if (h->fresh_scan && !h->last_scan_spec && h->eager_waste > 0) --h->eager_waste;   // a combine right behind a plainly encoded scan: the pattern is coming back
    h->last_scan_spec = false;
    h->fresh_scan = false;
    h->cnt_blocks = h->fuse_blocks;
    if (!on) HIPCHK(h, join_map_stream(h));
    const int nxt = h->has_combined ? 1 - h->cur : 0;
    Fused &F = h->fused[nxt];
    const Fused *prev = (h->has_combined && h->fused[h->cur].valid) ? &h->fused[h->cur] : nullptr;
    F.origin[0] = last.origin[0]; F.origin[1] = last.origin[1]; F.origin[2] = last.origin[2];
    FuseParams P;
    fill_fuse_frame(h, F.origin, P);
    int ns = 0;
    bool all_codes = true;
    // this fusion merges the statistics iff every slot of the ring carries its own; a previous map WITHOUT them (the
    // statistics were switched on again after a pause) contributes none (k_fuse_stats skips a source without metrics): the
    // statistics restart from the ring
    bool fstats = h->stats;
    for (int i = 0; i < p.buffer_size && fstats; ++i) { const Slot &s = h->slots[h->ring[i]]; if (s.filled && !s.has_metrics) fstats = false; }
    for (int i = 0; i < p.buffer_size; ++i) {                          // slot order, gvom.py:198
        const Slot &s = h->slots[h->ring[i]];
        if (!s.filled) continue;
        MapDesc &d = h->descs_host[ns++];
        d.state = s.state; d.rows = (const uint4 *)s.crows.p;
        d.d[0] = clamp_delta(F.origin[0] - s.origin[0], p.xy_size);
        d.d[1] = clamp_delta(F.origin[1] - s.origin[1], p.xy_size);
        d.d[2] = clamp_delta(F.origin[2] - s.origin[2], p.z_size);
        d.epoch = s.epoch; d.tags = s.tags; d.metrics = fstats ? s.metrics.p : nullptr;
        d.code16 = s.code16;
        all_codes = all_codes && s.has_code16;
    }
    P.nslots = ns;
    P.has_prev = prev ? 1 : 0;
    if (prev) {
        MapDesc &d = h->descs_host[ns];
        d.state = prev->state; d.rows = (const uint4 *)prev->rows.p;
        d.d[0] = clamp_delta(F.origin[0] - prev->origin[0], p.xy_size);
        d.d[1] = clamp_delta(F.origin[1] - prev->origin[1], p.xy_size);
        d.d[2] = clamp_delta(F.origin[2] - prev->origin[2], p.z_size);
        d.epoch = prev->epoch; d.tags = prev->tags; d.metrics = (fstats && prev->has_metrics) ? prev->metrics.p : nullptr;
        d.code16 = nullptr;
    }
    P.nz = choose_nz(p.z_size, &P.zc, &P.cpw);
    P.dbg = gvom_diag_env("GVOM_FUSE_DEBUG");
    // one slot in the ring (buffer_size 1, or a ring that has only just begun): k_fuse1 -- up to 8 waves per column block, 2
    // chunks per wave where the grid is high enough (a shorter chain of dependent round trips per wave)
    if (ns == 1 && P.zc == 16 && p.xy_size % 4 == 0 && (h->tune_fuse1 != 1 || !all_codes) && !(P.dbg & 8)) {
        const int nchunks = (p.z_size + 15) / 16;
        int nz = nchunks < 8 ? nchunks : 8;
        int cpw = (nchunks + nz - 1) / nz;
        if (cpw <= 4) { P.one_slot = 1; P.nz = nz; P.cpw = cpw; }
    }
    F.epoch = ++h->epoch;
    P.epoch = F.epoch;
    // every wave of k_fuse owns a static range of 64*zc compact rows (no global reservation)
    const size_t row_cap = (size_t)h->fuse_blocks * P.nz * 64 * P.zc * P.cpw;
    if (row_cap >= 2147483648ull) { h->err = "fused row space exceeds 31 bits"; return GVOM_ERR_CAPACITY; }
    int rc;
    if ((rc = ensure(h, F.rows, row_cap * 16))) return rc;
    if (fstats && (rc = ensure(h, F.metrics, row_cap * 40))) return rc;
    F.has_metrics = fstats;
    const int nsrc = ns + (prev ? 1 : 0);
    // the previous k_fuse_stats reads (as its "previous map") the fused buffer this fusion writes, and the descriptor
    // table this call refills
    if (h->fs_pending) HIPCHK(h, hipStreamWaitEvent(fs, h->ev_fsdone, 0));
    FuseDescs KD;
    const MapDesc *descs_mem = nullptr;
    if (nsrc <= GVOM_KARG_DESCS) {
        memcpy(KD.d, h->descs_host, sizeof(MapDesc) * nsrc);
    } else {
        HIPCHK(h, hipMemcpyAsync(h->descs_dev, h->descs_host, sizeof(MapDesc) * nsrc,
                                 hipMemcpyHostToDevice, fs));
        descs_mem = h->descs_dev;
    }
    if (h->profiling) HIPCHK(h, hipEventRecord(h->ev[4], fs));
    HIPCHK(h, gvom_launch_fuse(fs, P, KD, descs_mem, F.state, (uint4 *)F.rows.p,
                               F.tags, h->blockcounts,
                               h->height, h->inferred));
    if (h->profiling) { HIPCHK(h, hipEventRecord(h->ev[5], fs)); h->ev_fuse = true; }
    if (fstats) {                                        // beside k_map2d, behind this fusion and the scans' statistics
        HIPCHK(h, hipEventRecord(h->ev_fz_s, fs));
        HIPCHK(h, hipStreamWaitEvent(h->stream_s, h->ev_fz_s, 0));
        HIPCHK(h, gvom_launch_fuse_stats(h->stream_s, P, KD, descs_mem, F.state, F.tags, (float *)F.metrics.p));
        h->fs_reads[0] = h->fs_reads[1] = true;          // (its target's states and, as "previous map", the other buffer's)
        HIPCHK(h, hipEventRecord(h->ev_fsdone, h->stream_s));
        HIPCHK(h, hipEventRecord(h->ev_sdone, h->stream_s));
        h->s_pending = h->fs_pending = true;
    }
    F.valid = true;
    h->cur = nxt;
    h->has_combined = true;
    h->maps_valid = false;
    if (h->stats_release && !h->scan_inflight) release_statistics_buffers(h);   // (this fusion merged no statistics: fstats was false)
    return GVOM_OK;
}

// 2-D maps (k_map2d) from height/inferred of the whole window (all rows must be present)
// 2-D maps (k_map2d) from height/inferred of the whole window (all rows must be present).
// gathered: sharded run -- every row of the interleaved height buffer (heights + owner-computed
// positive densities) has been all-gathered and this rank computes ALL rows of the outputs.
int map2d_impl(gvom_handle *h, bool gathered, bool publish, char *out_dev, bool yx, const double *occ = nullptr,
               hipStream_t on = nullptr, uint32_t done_seq = 0)
{
    const hipStream_t ms = on ? on : h->stream;
    const gvom_params &p = h->prm;
    const Fused &F = h->fused[h->cur];
    Map2dParams P;
    memset(&P, 0, sizeof P);
    P.dbg = gvom_diag_env("GVOM_MAP2D_DEBUG");
    P.xy = p.xy_size; P.zs = p.z_size;
    P.om[0] = (int)floor_mod(F.origin[0], p.xy_size);
    P.om[1] = (int)floor_mod(F.origin[1], p.xy_size);
    P.om[2] = (int)floor_mod(F.origin[2], p.z_size);
    P.y_lo = gathered ? 0 : h->sy_lo; P.y_hi = gathered ? p.xy_size : h->sy_hi;
    P.origin_z = (double)F.origin[2];
    P.xy_res = p.xy_resolution; P.z_res = p.z_resolution;
    P.pos_thr = p.positive_obstacle_threshold; P.neg_thr = p.negative_obstacle_threshold;
    P.slope_thr = p.slope_obstacle_threshold; P.robot_height = p.robot_height;
    P.out_yx = yx ? 1 : 0;
    if (occ) { P.occ = 1; P.occ_density_thr = occ[0]; P.occ_min_rough = occ[1]; P.occ_max_rough = occ[2]; }
    P.gathered_pos = gathered ? 1 : 0;
    P.nseg = h->nseg;
    P.hs = h->hs;
    P.epoch = F.epoch;
    if (done_seq && !h->tune_flag_kernel) {       // the synchronous combine's completion flag (finish_combine): stored by k_map2d's last workgroup
        P.done_flag = (unsigned long long *)(h->counters_host_dev + 4);
        P.done_count = h->counters + GVOM_CNT_MAPDONE;
        P.done_seq = done_seq;
    }
    const size_t n2 = h->cells2d;
    int32_t *o_pos = (int32_t *)out_dev, *o_neg = o_pos + n2, *o_vis = o_neg + n2;
    double *o_rgh = (double *)(o_vis + n2);
    if (h->profiling) HIPCHK(h, hipEventRecord(h->ev[6], ms));
    HIPCHK(h, gvom_launch_map2d(ms, P, F.state, F.tags, (const uint4 *)F.rows.p,
                                h->height, h->inferred, h->slope_x,
                                h->slope_y, h->rough, h->guessed, o_pos, o_neg, o_rgh, o_vis,
                                h->blockcounts, h->cnt_blocks,
                                publish ? (unsigned long long *)(h->counters_host_dev + 2) : nullptr));
    if (h->profiling) { HIPCHK(h, hipEventRecord(h->ev[7], ms)); h->ev_map = true; }
    h->maps_valid = true;
    return GVOM_OK;
}

// sharded runs: positive-obstacle densities of this rank's rows into the height buffer
int posdens_impl(gvom_handle *h)
{
    const gvom_params &p = h->prm;
    const Fused &F = h->fused[h->cur];
    Map2dParams P;
    memset(&P, 0, sizeof P);
    P.xy = p.xy_size; P.zs = p.z_size;
    P.om[2] = (int)floor_mod(F.origin[2], p.z_size);
    P.y_lo = h->sy_lo; P.y_hi = h->sy_hi;
    P.origin_z = (double)F.origin[2];
    P.z_res = p.z_resolution;
    P.pos_thr = p.positive_obstacle_threshold; P.robot_height = p.robot_height;
    P.nseg = h->nseg; P.hs = h->hs; P.epoch = F.epoch;
    // its first workgroup also publishes the fused cell count (k_fuse has completed by then)
    HIPCHK(h, gvom_launch_posdens(h->stream, P, F.state, F.tags, (const uint4 *)F.rows.p,
                                  h->hmaps, h->blockcounts, h->fuse_blocks,
                                  (unsigned long long *)(h->counters_host_dev + 2),
                                  (unsigned long long *)(h->counters + 10)));
    return GVOM_OK;
}

// stage times of every profiling event pair recorded since the last collection (stream is idle)
void collect_stage_ms(gvom_handle *h)
{
    if (h->ev_scan) {
        hipEventElapsedTime(&h->stage_ms[0], h->ev[0], h->ev[1]);
        hipEventElapsedTime(&h->stage_ms[1], h->ev[1], h->ev[2]);
        h->stage_ms[2] = 0.0f;             // min-height runs inside the k_encode launch
    }
    if (h->ev_fuse) hipEventElapsedTime(&h->stage_ms[3], h->ev[4], h->ev[5]);
    // (the combine's completion flag goes out before k_map2d's launch has formally ended: wait for its event)
    if (h->ev_map && hipEventSynchronize(h->ev[7]) == hipSuccess) hipEventElapsedTime(&h->stage_ms[4], h->ev[6], h->ev[7]);
    h->ev_scan = h->ev_fuse = h->ev_map = false;
    (void)hipGetLastError();               // never leave a sticky error behind for the launchers
}

// Waits for the combine's kernels with the handle mutex RELEASED (a second thread -- the ROS node's
// cloud callback -- can hand the next scan over meanwhile: its kernels queue up behind k_map2d and the
// GPU does not idle between the steps); other combine calls are held off by combine_mu / pending_combine.
int finish_combine(gvom_handle *h, std::unique_lock<std::mutex> &lk, uint32_t seq)
{
    // completion: k_map2d's last workgroup stores the sequence number into host-mapped memory (map2d_impl) and the host
    // spins on it (an event wait notices the end of the stream several microseconds later; round 3's one-thread kernel
    // behind k_map2d cost 4 us of every step)
    if (h->tune_flag_kernel) HIPCHK(h, gvom_launch_publish_seq(h->stream, (unsigned long long *)(h->counters_host_dev + 4), seq));
    HIPCHK(h, hipEventRecord(h->ev_done, h->stream));
    h->pending_combine = true;
    hipError_t e = hipSuccess;
    if (!wait_published(h, lk, (volatile unsigned long long *)(h->counters_host + 4), seq, false, &h->last_wait_ns[1])) {
        lk.unlock();
        e = hipEventSynchronize(h->ev_done);
        lk.lock();
    }
    h->pending_combine = false;
    HIPCHK(h, e);
    Fused &F = h->fused[h->cur];
    unsigned long long c;
    memcpy(&c, h->counters_host + 2, 8);
    F.count = (int64_t)c;
    h->combined_cell_count = F.count;
    collect_stage_ms(h);
    return GVOM_OK;
}

// the split (sharded) combine calls: plain wait under the handle mutex
int finish_combine(gvom_handle *h)
{
    HIPCHK(h, sync_streams(h));
    Fused &F = h->fused[h->cur];
    unsigned long long c;
    memcpy(&c, h->counters_host + 2, 8);
    F.count = (int64_t)c;
    h->combined_cell_count = F.count;
    collect_stage_ms(h);
    return GVOM_OK;
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

VIS int gvom_create(const gvom_params *params, int device_id, gvom_t **out)
{
    return create_impl(params, device_id, 0, 1, false, out);
}

VIS int gvom_create_sharded(const gvom_params *params, int device_id, int rank, int world, gvom_t **out)
{
    return create_impl(params, device_id, rank, world, true, out);
}

VIS void gvom_destroy(gvom_t *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream_b) hipStreamSynchronize(h->stream_b);
    if (h->stream) hipStreamSynchronize(h->stream);
    auto fb = [](Buf &b) { if (b.p) hipFree(b.p); b.p = nullptr; b.bytes = 0; };
    hipFree(h->hit); hipFree(h->total); hipFree(h->mh);
    // Regions another process has had mapped (the peer transport exported them: "exported" is set through gvom_set_tuning) go
    // to the process-wide pool (see pool_put), the others back to the allocator
    auto park = [&](void *ptr, size_t bytes, uint64_t gen) {
        if (!ptr) return;
        if (h->exported || h->holds_pooled) pool_put(ptr, bytes, gen); else hipFree(ptr);
    };
    for (Buf &r : h->retired) park(r.p, r.bytes, r.gen);
    if (h->sharded) {
        park(h->x_send_ids, exportable_size(h->x_Q * 4), h->fixed_gen[0]); park(h->x_send_pay, exportable_size(h->x_Q * 1024), h->fixed_gen[1]);
        park(h->x_send_eps.p, h->x_send_eps.bytes, h->x_send_eps.gen); park(h->x_send_sp.p, h->x_send_sp.bytes, h->x_send_sp.gen);
        park(h->hmaps, exportable_size(h->cells2d * 24), h->fixed_gen[2]); h->hmaps = nullptr;
    }
    h->x_send_eps.p = nullptr; h->x_send_sp.p = nullptr;
    hipFree(h->x_recv_ids); hipFree(h->x_recv_pay);
    hipFree(h->x_qcnt); hipFree(h->x_ecnt); hipFree(h->x_spcnt); fb(h->x_recv_eps); fb(h->x_recv_sp);
    if (h->x_host) hipHostFree(h->x_host);
    for (auto &s : h->slots) { hipFree(s.state); hipFree(s.code16); hipFree(s.tags); fb(s.crows); fb(s.metrics); fb(s.base); fb(s.rowvox); }
    for (auto &f : h->fused) { hipFree(f.state); hipFree(f.tags); fb(f.rows); fb(f.metrics); }
    fb(h->in_pts); fb(h->world_pts[0]); fb(h->world_pts[1]); fb(h->flink[0]); fb(h->flink[1]); fb(h->tl); fb(h->dir_keys); fb(h->dir_perm); hipFree(h->dir_hist);
    hipFree(h->counters); if (h->counters_host) hipHostFree(h->counters_host);
    hipFree(h->descs_dev); if (h->descs_host) hipHostFree(h->descs_host);
    hipFree(h->blockcounts); hipFree(h->blockcounts2); hipFree(h->hmaps2);
    hipFree(h->hmaps); hipFree(h->slope_x); hipFree(h->slope_y);
    hipFree(h->rough); hipFree(h->guessed);
    if (h->out_host) hipHostFree(h->out_host);
    for (auto &e : h->ev) if (e) hipEventDestroy(e);
    if (h->ev_fused) hipEventDestroy(h->ev_fused);
    if (h->ev_mapped) hipEventDestroy(h->ev_mapped);
    if (h->ev_done) hipEventDestroy(h->ev_done);
    if (h->ev_fuse_b) hipEventDestroy(h->ev_fuse_b);
    if (h->ev_up) hipEventDestroy(h->ev_up);
    if (h->stream_up) { hipStreamSynchronize(h->stream_up); hipStreamDestroy(h->stream_up); }
    if (h->stream_s) { hipStreamSynchronize(h->stream_s); hipStreamDestroy(h->stream_s); }
    for (hipEvent_t e : {h->ev_enc_s, h->ev_fz_s, h->ev_sdone, h->ev_fsdone, h->ev_before[0], h->ev_before[1]}) if (e) hipEventDestroy(e);
    if (h->stream_b) hipStreamDestroy(h->stream_b);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    delete h;
}

VIS int gvom_process_pointcloud(gvom_t *h, const void *xyz, int64_t n, int64_t row_stride_bytes,
                                int dtype, const double ego[3], const double *transform_4x4)
{
    return process_impl(h, xyz, false, n, row_stride_bytes, dtype, ego, transform_4x4, false);
}

VIS int gvom_process_pointcloud_device(gvom_t *h, const void *xyz_dev, int64_t n,
                                       int64_t row_stride_bytes, int dtype, const double ego[3],
                                       const double *transform_4x4)
{
    return process_impl(h, xyz_dev, true, n, row_stride_bytes, dtype, ego, transform_4x4, false);
}

// Ingest side of the ROS node (gvom_ros.py:93-109): the packed bytes of a
// sensor_msgs/PointCloud2 (`data`, width*height records of point_step bytes, little-endian FLOAT32
// or FLOAT64 fields x, y, z at the given byte offsets) are scanned directly; the host-side
// ros_numpy expansion to an xyz array disappears.  ros_numpy drops records with a non-finite
// coordinate; here such records take part and have no effect (no in-grid endpoint, no ray step).
VIS int gvom_process_pointcloud2(gvom_t *h, const void *data, int64_t n_points, int64_t point_step,
                                 int64_t off_x, int64_t off_y, int64_t off_z, int dtype,
                                 const double ego[3], const double *transform_4x4)
{
    const int64_t off[3] = {off_x, off_y, off_z};
    // ros_numpy returns a float64 array whatever the field type (get_xyz_points, dtype=np.float), so
    // the reference computes a FLOAT32 cloud in f64 as well: widen on load, run the f64 kernels
    return process_impl(h, data, false, n_points, point_step, GVOM_DTYPE_F64, ego, transform_4x4, false, off,
                        dtype == GVOM_DTYPE_F32);
}

// ---- scan of a sharded map (one rank per GPU; DESIGN.md "Multi-GPU") ---------------------------
// gvom_shard_scan_local: this rank's share of the scan (possibly empty): traces ITS rays over the whole
//   window, leaves what fell into its own rows in its accumulators and packs the rest for the owners.
//   send_quads[d] / send_eps[d] = dirty quads (1 KiB each + a 4-byte id) / endpoints (8 bytes each) for
//   rank d; *any_ingrid = some return of this rank landed in the grid.
// gvom_shard_buffer: the send / receive regions, by peer rank, for the transport (RCCL: gvom_comm_*).
// gvom_shard_recv_reserve: sizes the endpoint receive regions once the counts are known.
// gvom_shard_scan_merge: adds what the other ranks sent, encodes this rank's rows and commits the scan
//   iff `accept` (the reference's "no overlap" test, gvom.py:147-150, is on the whole scan).
VIS int gvom_shard_scan_local(gvom_t *h, const void *xyz, int on_device, int64_t n, int64_t row_stride_bytes,
                              int dtype, const double ego[3], const double *transform_4x4,
                              int64_t *send_quads, int64_t *send_eps, int *any_ingrid)
{
    if (!h || !h->sharded) return GVOM_ERR_INVALID;
    int rc = process_impl(h, xyz, on_device != 0, n, row_stride_bytes, dtype, ego, transform_4x4, true);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    for (int d = 0; d < h->world; ++d) {
        if (send_quads) send_quads[d] = h->solo_encoded ? 0 : (int64_t)h->x_host[d];
        if (send_eps) send_eps[d] = h->solo_encoded ? 0 : (int64_t)h->x_host[h->world + d];
    }
    if (any_ingrid) *any_ingrid = h->pending_any ? 1 : 0;
    return GVOM_OK;
}

VIS int gvom_shard_recv_reserve(gvom_t *h, const int64_t *recv_eps)
{
    if (!h || !h->sharded || !recv_eps) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    int64_t tot = 0;
    for (int sidx = 0; sidx < h->world; ++sidx) {
        h->x_recv_ep_off[sidx] = tot;
        if (recv_eps[sidx] < 0) return GVOM_ERR_INVALID;
        tot += sidx == h->rank ? 0 : recv_eps[sidx];
    }
    h->x_recv_ep_off[h->world] = tot;
    return ensure(h, h->x_recv_eps, (size_t)(tot > 0 ? tot : 1) * 8);
}

// Per-voxel statistics on a sharded map (handles created with GVOM_FLAG_VOXEL_STATISTICS): after gvom_shard_scan_local,
// send_returns[d] = returns (3 values of the cloud's type each) this rank has for rank d -- every return whose
// (2 xy_eigen_dist + 1)-row neighbourhood reaches into d's rows; the transport moves GVOM_XBUF_SEND_RETURNS to the peers'
// GVOM_XBUF_RECV_RETURNS (sized by gvom_shard_stats_reserve from the exchanged counts) before gvom_shard_scan_merge.
VIS int gvom_shard_stats_counts(gvom_t *h, int64_t *send_returns)
{
    if (!h || !h->sharded || !send_returns) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    for (int d = 0; d < h->world; ++d) send_returns[d] = (h->stats && h->pending && !h->solo_encoded) ? (int64_t)h->x_host[2 * h->world + 2 + d] : 0;
    return GVOM_OK;
}

VIS int gvom_shard_stats_reserve(gvom_t *h, const int64_t *recv_returns, int dtype)
{
    if (!h || !h->sharded || !recv_returns || (dtype != GVOM_DTYPE_F32 && dtype != GVOM_DTYPE_F64)) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    if (h->pending_n > 0 && dtype != h->pending_dtype) { h->err = "the ranks' clouds differ in type"; return GVOM_ERR_INVALID; }
    h->pending_dtype = dtype;                             // (a rank with an empty share takes the others' type)
    int64_t tot = 0;
    for (int sidx = 0; sidx < h->world; ++sidx) {
        h->x_recv_sp_off[sidx] = tot;
        if (recv_returns[sidx] < 0) return GVOM_ERR_INVALID;
        tot += sidx == h->rank ? 0 : recv_returns[sidx];
    }
    h->x_recv_sp_off[h->world] = tot;
    const size_t esz3 = (h->pending_dtype == GVOM_DTYPE_F32 ? 4 : 8) * 3;
    return ensure(h, h->x_recv_sp, (size_t)(tot > 0 ? tot : 1) * esz3);
}

VIS int gvom_shard_buffer(gvom_t *h, int which, int peer, void **ptr, int64_t *capacity_bytes)
{
    if (!h || !h->sharded || !ptr || peer < 0 || peer >= h->world) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    const size_t myQ = h->x_myQ;
    int64_t cap = 0;
    switch (which) {
    case GVOM_XBUF_SEND_IDS: *ptr = h->x_send_ids + (size_t)peer * myQ; cap = (int64_t)myQ * 4; break;
    case GVOM_XBUF_SEND_QUADS: *ptr = (char *)h->x_send_pay + (size_t)peer * myQ * 1024; cap = (int64_t)myQ * 1024; break;
    case GVOM_XBUF_SEND_EPS: *ptr = (char *)h->x_send_eps.p + (size_t)peer * (size_t)h->x_ep_cap * 8; cap = h->x_ep_cap * 8; break;
    case GVOM_XBUF_RECV_IDS: *ptr = h->x_recv_ids + (size_t)peer * myQ; cap = (int64_t)myQ * 4; break;
    case GVOM_XBUF_RECV_QUADS: *ptr = (char *)h->x_recv_pay + (size_t)peer * myQ * 1024; cap = (int64_t)myQ * 1024; break;
    case GVOM_XBUF_RECV_EPS:
        *ptr = (char *)h->x_recv_eps.p + (size_t)h->x_recv_ep_off[peer] * 8;
        cap = (h->x_recv_ep_off[peer + 1] - h->x_recv_ep_off[peer]) * 8; break;
    case GVOM_XBUF_SEND_RETURNS: {
        const size_t esz3 = (h->pending_dtype == GVOM_DTYPE_F32 ? 4 : 8) * 3;
        if (!h->x_send_sp.p) return GVOM_NO_DATA;
        *ptr = (char *)h->x_send_sp.p + (size_t)peer * (size_t)h->x_ep_cap * esz3; cap = h->x_ep_cap * (int64_t)esz3; break;
    }
    case GVOM_XBUF_RECV_RETURNS: {
        const size_t esz3 = (h->pending_dtype == GVOM_DTYPE_F32 ? 4 : 8) * 3;
        if (!h->x_recv_sp.p) return GVOM_NO_DATA;
        *ptr = (char *)h->x_recv_sp.p + (size_t)h->x_recv_sp_off[peer] * esz3;
        cap = (h->x_recv_sp_off[peer + 1] - h->x_recv_sp_off[peer]) * (int64_t)esz3; break;
    }
    default: return GVOM_ERR_INVALID;
    }
    if (capacity_bytes) *capacity_bytes = cap;
    return GVOM_OK;
}

VIS int gvom_shard_scan_merge(gvom_t *h, const int64_t *recv_quads, const int64_t *recv_eps, int accept)
{
    if (!h || !h->sharded || !recv_quads || !recv_eps) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->pending) return GVOM_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->solo_encoded) {                                 // one rank: gvom_shard_scan_local has encoded the scan already
        h->solo_encoded = false;
        scan_commit(h, accept != 0);
        return GVOM_OK;
    }
    Slot &st = h->slots[h->staging];
    const ScanParams &P = h->pending_P;
    int64_t tot_eps = 0;
    for (int sidx = 0; sidx < h->world; ++sidx) {
        if (sidx == h->rank) continue;
        if (recv_quads[sidx] < 0 || recv_quads[sidx] > (int64_t)h->x_myQ || recv_eps[sidx] < 0 ||
            recv_eps[sidx] != h->x_recv_ep_off[sidx + 1] - h->x_recv_ep_off[sidx]) {
            h->err = "gvom_shard_scan_merge: receive counts do not match gvom_shard_recv_reserve / the slab size";
            return GVOM_ERR_INVALID;
        }
        tot_eps += recv_eps[sidx];
    }
    // compact rows: this rank's returns first, the received endpoints behind them
    // (at least one row: k_fuse redirects the loads of unoccupied voxels to row 0 of every source)
    const size_t cap = std::max<size_t>(1, (size_t)h->pending_n + (size_t)tot_eps);
    int rc;
    if ((rc = ensure(h, st.crows, cap * 16))) return rc;
    if (h->stats) {                                       // rows of the received endpoints: behind the rank's own (contents kept)
        const size_t old_rows = st.rowvox.bytes / 4;
        if ((rc = ensure_keep(h, st.metrics, cap * 80)) || (rc = ensure_keep(h, st.base, cap * 8 * GVOM_BASE_PITCH)) ||
            (rc = ensure_keep(h, st.rowvox, cap * 4))) return rc;
        if (st.rowvox.bytes / 4 > old_rows)
            HIPCHK(h, hipMemsetAsync((uint32_t *)st.rowvox.p + old_rows, 0xFF, st.rowvox.bytes - old_rows * 4, h->stream));
    }
    double t0 = now_ns();
    {   // everything received, whatever the source, in one launch each (quads, endpoints)
        ShardUnpack X;
        uint32_t acc = 0;
        for (int sidx = 0; sidx <= GVOM_MAX_SLOTS; ++sidx) {
            X.q_off[sidx] = acc;
            if (sidx < h->world && sidx != h->rank) acc += (uint32_t)recv_quads[sidx];
        }
        hipError_t le = gvom_launch_unpack(h->stream, P, X, h->x_recv_ids, h->x_recv_pay, (uint32_t)h->x_myQ, (uint32_t)tot_eps,
                                           h->x_recv_eps.p, (long)h->pending_n, h->hit, h->total, h->mh, st.state, st.tags,
                                           h->stats ? (double *)st.metrics.p : nullptr, h->stats ? (double *)st.base.p : nullptr,
                                           h->stats ? (uint32_t *)st.rowvox.p : nullptr);
        if (le != hipSuccess) { scan_abort(h); HIPCHK(h, le); }
    }
    const uint32_t seq = ++h->scan_seq;
    hipError_t le = gvom_launch_encode(h->stream, P, h->hit, h->total, h->mh, st.state, st.code16, (uint4 *)st.crows.p,
                                       st.tags, h->counters,
                                       (unsigned long long *)h->counters_host_dev, seq, h->resident_blocks);
    if (le != hipSuccess) { scan_abort(h); HIPCHK(h, le); }
    st.has_code16 = st.code16 != nullptr;
    if (h->profiling) { HIPCHK(h, hipEventRecord(h->ev[2], h->stream)); h->ev_scan = true; }
    if (h->stats) {
        // this rank's own returns (all of them: a return outside its rows can still reach into them) and the ones the
        // other ranks sent, in the direct form over this rank's rows
        const int64_t tot_sp = h->world > 1 ? h->x_recv_sp_off[h->world] : 0;
        if ((rc = enqueue_scan_stats(h, P, h->pending_dtype, st, h->pending_n, (int64_t)cap, h->x_recv_sp.p, tot_sp))) return rc;
    }
    HT(h, 0, t0);
    // (no host wait: everything the caller can do next with this handle is stream-ordered behind k_encode)
    scan_commit(h, accept != 0);
    return GVOM_OK;
}

VIS int gvom_combine_maps(gvom_t *h, double origin_world[3], int32_t *positive, int32_t *negative,
                          double *roughness, int32_t *visibility)
{
    if (!h || h->sharded) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> ck(h->combine_mu);
    std::unique_lock<std::mutex> lk(h->mu);
    if (h->pending_combine) { h->err = "a combine begun with gvom_combine_begin has not been ended"; return GVOM_ERR_INVALID; }
    HIPCHK(h, hipSetDevice(h->device));
    double t0 = now_ns();
    int rc = fuse_impl(h);
    if (rc) return rc;
    const uint32_t done_seq = ++h->combine_seq;
    if ((rc = map2d_impl(h, false, true, h->out_host_dev, false, nullptr, nullptr, done_seq))) return rc;
    const size_t n2 = h->cells2d;
    char *stage = (char *)h->out_host;
    HT(h, 2, t0);                                        // combine: launches
    if ((rc = finish_combine(h, lk, done_seq))) return rc;
    HT(h, 3, t0);                                        // combine: wait
    if (positive) memcpy(positive, stage, n2 * 4);
    if (negative) memcpy(negative, stage + n2 * 4, n2 * 4);
    if (visibility) memcpy(visibility, stage + n2 * 8, n2 * 4);
    if (roughness) memcpy(roughness, stage + n2 * 12, n2 * 8);
    HT(h, 4, t0);                                        // combine: pinned -> caller copies
    if (origin_world) {                                                // gvom.py:185-188
        const Fused &F = h->fused[h->cur];
        origin_world[0] = (double)F.origin[0] * h->prm.xy_resolution;
        origin_world[1] = (double)F.origin[1] * h->prm.xy_resolution;
        origin_world[2] = (double)F.origin[2] * h->prm.z_resolution;
    }
    return GVOM_OK;
}

// A caller's own output buffer must be COHERENT pinned memory (see gvom_hip.h, gvom_combine_maps_into): the completion flag is only
// ordered behind the maps for write-through stores.  Buffers from gvom_output_buffer_alloc are; others are asked once.
static int check_out_buffer(gvom_handle *h, void *pinned_out)
{
    if (std::find(h->out_bufs.begin(), h->out_bufs.end(), pinned_out) != h->out_bufs.end() || pinned_out == h->last_checked_out) return GVOM_OK;
    unsigned int flags = 0;
    if (hipHostGetFlags(&flags, pinned_out) != hipSuccess) { (void)hipGetLastError(); h->err = "output buffer is not pinned host memory (hipHostMalloc)"; return GVOM_ERR_INVALID; }
    if (!(flags & hipHostMallocCoherent) || !(flags & hipHostMallocMapped)) {
        h->err = "output buffer must be coherent, device-mapped pinned memory (hipHostMallocMapped | hipHostMallocCoherent; gvom_output_buffer_alloc returns such)";
        return GVOM_ERR_INVALID;
    }
    h->last_checked_out = pinned_out;
    return GVOM_OK;
}

// ---- zero-copy outputs ---------------------------------------------------------------------
// gvom_output_buffer_alloc returns a pinned, device-mapped host buffer of 20*xy*xy bytes laid out
// [positive i32 | negative i32 | visibility i32 | roughness f64] (each xy*xy, COLUMN-major:
// cell (x, y) at m[y*xy + x]).
// gvom_combine_maps_into makes k_map2d write the four maps straight into such a buffer: no D2H
// copy command and no pinned->caller memcpy.  The caller owns the buffer until it frees it
// (g-vom_amd/gvom.py recycles them through a pool when the returned numpy arrays are collected).
VIS int gvom_output_buffer_alloc(gvom_t *h, void **host_ptr)
{
    if (!h || !host_ptr) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    // GVOM_OUT_COHERENT=1: fine-grained (coherent) pinned memory -- stores leave the GPU as they are
    // issued instead of being written back from L2 at the end of the kernel
    HIPCHK(h, hipHostMalloc(host_ptr, h->cells2d * 20, hipHostMallocMapped | hipHostMallocCoherent));
    h->out_bufs.push_back(*host_ptr);
    return GVOM_OK;
}

VIS int gvom_output_buffer_free(gvom_t *h, void *host_ptr)
{
    if (!h || !host_ptr) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, sync_streams(h));
    for (size_t k = 0; k < h->out_bufs.size(); ++k)
        if (h->out_bufs[k] == host_ptr) { h->out_bufs[k] = h->out_bufs.back(); h->out_bufs.pop_back(); break; }
    if (h->last_checked_out == host_ptr) h->last_checked_out = nullptr;
    HIPCHK(h, hipHostFree(host_ptr));
    return GVOM_OK;
}

VIS int gvom_combine_maps_into(gvom_t *h, double origin_world[3], void *pinned_out)
{
    if (!h || !pinned_out || h->sharded) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> ck(h->combine_mu);
    std::unique_lock<std::mutex> lk(h->mu);
    if (h->pending_combine) { h->err = "a combine begun with gvom_combine_begin has not been ended"; return GVOM_ERR_INVALID; }
    HIPCHK(h, hipSetDevice(h->device));
    { const int rc0 = check_out_buffer(h, pinned_out); if (rc0) return rc0; }
    double t0 = now_ns();
    int rc = fuse_impl(h);
    if (rc) return rc;
    char *dev = nullptr;
    HIPCHK(h, hipHostGetDevicePointer((void **)&dev, pinned_out, 0));
    const uint32_t done_seq = ++h->combine_seq;
    if ((rc = map2d_impl(h, false, true, dev, true, nullptr, nullptr, done_seq))) return rc;
    HT(h, 2, t0);
    if ((rc = finish_combine(h, lk, done_seq))) return rc;
    HT(h, 3, t0);
    if (origin_world) {
        const Fused &F = h->fused[h->cur];
        origin_world[0] = (double)F.origin[0] * h->prm.xy_resolution;
        origin_world[1] = (double)F.origin[1] * h->prm.xy_resolution;
        origin_world[2] = (double)F.origin[2] * h->prm.z_resolution;
    }
    return GVOM_OK;
}

// combine_maps + the ROS node's post-processing (gvom_ros.py:141-165) in one call: the fusion
// advances exactly as in gvom_combine_maps, but k_map2d writes the five int8
// nav_msgs/OccupancyGrid.data arrays [hard | soft | certainty | negative | roughness] (each xy*xy
// bytes, x fastest = the node's reshape(order='F')) into the pinned buffer: 5 bytes per cell cross
// PCIe instead of 20.
VIS int gvom_combine_occupancy_into(gvom_t *h, double origin_world[3], void *pinned_out,
                                    double density_threshold, double min_roughness, double max_roughness)
{
    if (!h || !pinned_out || h->sharded) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> ck(h->combine_mu);
    std::unique_lock<std::mutex> lk(h->mu);
    if (h->pending_combine) { h->err = "a combine begun with gvom_combine_begin has not been ended"; return GVOM_ERR_INVALID; }
    HIPCHK(h, hipSetDevice(h->device));
    { const int rc0 = check_out_buffer(h, pinned_out); if (rc0) return rc0; }
    double t0 = now_ns();
    int rc = fuse_impl(h);
    if (rc) return rc;
    char *dev = nullptr;
    HIPCHK(h, hipHostGetDevicePointer((void **)&dev, pinned_out, 0));
    const double occ[3] = {density_threshold, min_roughness, max_roughness};
    const uint32_t done_seq = ++h->combine_seq;
    if ((rc = map2d_impl(h, false, true, dev, true, occ, nullptr, done_seq))) return rc;
    HT(h, 2, t0);
    if ((rc = finish_combine(h, lk, done_seq))) return rc;
    HT(h, 3, t0);
    if (origin_world) {
        const Fused &F = h->fused[h->cur];
        origin_world[0] = (double)F.origin[0] * h->prm.xy_resolution;
        origin_world[1] = (double)F.origin[1] * h->prm.xy_resolution;
        origin_world[2] = (double)F.origin[2] * h->prm.z_resolution;
    }
    return GVOM_OK;
}

// ---- asynchronous combine --------------------------------------------------------------------
// gvom_combine_begin = gvom_combine_maps_into / gvom_combine_occupancy_into (occ != NULL: its three
// thresholds) without the wait: the fusion is enqueued on the handle's stream, k_map2d on a second
// stream behind it.  The caller may hand the next scan to gvom_process_pointcloud* right away: its
// k_trace / k_encode run WHILE k_map2d stores the maps over PCIe (the next fusion waits for it on the
// device).  gvom_combine_end waits for the maps (handle mutex released while it waits) and completes
// the call; `pinned_out` must not be read before it returns.  One combine may be pending at a time.
VIS int gvom_combine_begin(gvom_t *h, void *pinned_out, const double *occ)
{
    if (!h || !pinned_out || h->sharded) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->pending_combine) { h->err = "a combine begun with gvom_combine_begin has not been ended"; return GVOM_ERR_INVALID; }
    HIPCHK(h, hipSetDevice(h->device));
    { const int rc0 = check_out_buffer(h, pinned_out); if (rc0) return rc0; }
    double t0 = now_ns();
    // k_map2d goes to the second stream, and with a ring of three or more filled slots the fusion too (behind
    // the scan's k_encode on the main stream): the next scan's k_trace / k_encode overlap them -- they touch the
    // accumulators and the spare slot only.  (Measured, pipelined use, fusion on the main / the second stream:
    // m256 86.5 / 88.8 us per step, c4 680 / 692, but c3 137 / 115, m256b8 119 / 102: a long fusion is worth it.)
    int filled = 0;
    for (int i = 0; i < h->prm.buffer_size; ++i) filled += h->slots[h->ring[i]].filled ? 1 : 0;
    const bool fuse_on_b = filled >= 3;
    int rc;
    if (fuse_on_b) {
        HIPCHK(h, hipEventRecord(h->ev_fused, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream_b, h->ev_fused, 0));
        if ((rc = fuse_impl(h, h->stream_b))) return rc;
        HIPCHK(h, hipEventRecord(h->ev_fuse_b, h->stream_b));
        h->fuse_b_unjoined = true;
        h->fuse_b_slots = 0;
        for (int i = 0; i < h->prm.buffer_size; ++i)
            if (h->slots[h->ring[i]].filled) h->fuse_b_slots |= 1ull << h->ring[i];
    } else {
        if ((rc = fuse_impl(h))) return rc;
        HIPCHK(h, hipEventRecord(h->ev_fused, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream_b, h->ev_fused, 0));
    }
    char *dev = nullptr;
    HIPCHK(h, hipHostGetDevicePointer((void **)&dev, pinned_out, 0));
    if ((rc = map2d_impl(h, false, true, dev, true, occ, h->stream_b))) return rc;
    HIPCHK(h, hipEventRecord(h->ev_mapped, h->stream_b));
    h->mapped_unjoined = true;
    h->pending_combine = true;
    HT(h, 2, t0);
    return GVOM_OK;
}

VIS int gvom_combine_end(gvom_t *h, double origin_world[3])
{
    if (!h) return GVOM_ERR_INVALID;
    std::unique_lock<std::mutex> lk(h->mu);
    if (!h->pending_combine) { h->err = "gvom_combine_end without gvom_combine_begin"; return GVOM_ERR_INVALID; }
    HIPCHK(h, hipSetDevice(h->device));
    double t0 = now_ns();
    // (an event wait: in the pipelined use the maps are usually there already, and a completion-flag kernel on
    // the second stream would cost more than it saves -- measured 91.0 against 86.5 us per step)
    lk.unlock();                                           // process_pointcloud may run meanwhile
    const hipError_t e = hipEventSynchronize(h->ev_mapped);
    lk.lock();
    h->pending_combine = false;                            // (also on failure: the handle must not stay blocked)
    h->fuse_b_unjoined = false;                            // k_map2d has completed, and the fusion in front of it
    HIPCHK(h, e);
    Fused &F = h->fused[h->cur];
    unsigned long long c;
    memcpy(&c, h->counters_host + 2, 8);
    F.count = (int64_t)c;
    h->combined_cell_count = F.count;
    HT(h, 3, t0);
    if (origin_world) {
        origin_world[0] = (double)F.origin[0] * h->prm.xy_resolution;
        origin_world[1] = (double)F.origin[1] * h->prm.xy_resolution;
        origin_world[2] = (double)F.origin[2] * h->prm.z_resolution;
    }
    return GVOM_OK;
}

// ---- split combine for the sharded layer (g-vom_amd/gvom_sharded.py) ---------------------
// 1. gvom_combine_fuse: local slab fusion; height/inferred rows of this rank are valid.
// 2. gvom_rows_export / gvom_rows_import: device<->device copies of 2-D map rows in storage
//    order ([sy][sx], row range of a rank is contiguous) to/from caller-owned device buffers
//    (the collectives run on those, e.g. torch.distributed all_gather over RCCL).
// 3. gvom_combine_map2d: local rows of the four outputs, in storage order.
// 4. gvom_finalize_outputs: storage order -> the reference's [x][y] window order (rank 0).
VIS int gvom_combine_fuse(gvom_t *h, int64_t *local_cells)
{
    if (!h) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    int rc = fuse_impl(h);
    if (rc) return rc;
    // third row of the height buffer + the cell count -- unless this rank holds every row (a sharded map of ONE rank): nothing
    // is gathered then, and k_map2d computes the densities of its own cells and publishes the count, as on an unsharded handle
    if (!(h->sharded && h->world == 1) && (rc = posdens_impl(h))) return rc;
    if (h->sharded) {                                     // no host wait: the count stays on the device (GVOM_BUF_FUSED_CELLS)
        if (local_cells) *local_cells = -1;
        return GVOM_OK;
    }
    if ((rc = finish_combine(h))) return rc;
    if (local_cells) *local_cells = h->fused[h->cur].count;
    return GVOM_OK;
}

VIS int gvom_set_combined_cell_count(gvom_t *h, int64_t global_cells)
{
    if (!h) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    h->combined_cell_count = global_cells;
    return GVOM_OK;
}

static void *map_ptr(gvom_handle *h, int which, size_t *esz, int *stride)
{
    *esz = 8; *stride = h->prm.xy_size;
    switch (which) {
    case GVOM_MAP_HEIGHT: *stride = h->hs; return h->height;
    case GVOM_MAP_INFERRED_HEIGHT: *stride = h->hs; return h->inferred;
    case GVOM_MAP_SLOPE_X: return h->slope_x;
    case GVOM_MAP_SLOPE_Y: return h->slope_y;
    case GVOM_MAP_ROUGHNESS: return h->rough;
    case GVOM_MAP_GUESSED_DELTA: return h->guessed;
    default: return nullptr;
    }
}

// ---- plumbing for the sharded layer: the library's own device buffers take part in the collectives
// directly (a rank's rows are one contiguous block of each buffer), on the library's stream, without
// host synchronisation in between.
VIS int gvom_sync(gvom_t *h)
{
    if (!h) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, sync_streams(h));
    collect_stage_ms(h);
    return GVOM_OK;
}

VIS int gvom_device_buffer(gvom_t *h, int which, void **ptr, int64_t *bytes, int64_t *row_stride_bytes)
{
    if (!h || !ptr) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    int64_t b = 0, rs = 0;
    switch (which) {
    case GVOM_BUF_HEIGHT_MAPS: *ptr = h->hmaps; rs = (int64_t)h->hs * 8; b = rs * h->prm.xy_size; break;
    case GVOM_BUF_FUSED_CELLS: *ptr = h->counters + 10; b = 8; rs = 8; break;
    default: return GVOM_ERR_INVALID;
    }
    if (bytes) *bytes = b;
    if (row_stride_bytes) *row_stride_bytes = rs;
    return GVOM_OK;
}

// sharded runs, after the in-place all_gather of GVOM_BUF_HEIGHT_MAPS: all rows of the four
// outputs, written by the GPU straight into a pinned buffer from gvom_output_buffer_alloc
// (same layout as gvom_combine_maps_into).  Synchronises.
VIS int gvom_combine_map2d_into(gvom_t *h, double origin_world[3], void *pinned_out)
{
    if (!h || !pinned_out) return GVOM_ERR_INVALID;
    std::unique_lock<std::mutex> lk(h->mu);              // (ONE lock object: finish_combine releases it while the host spins)
    if (!h->has_combined) return GVOM_NO_DATA;
    HIPCHK(h, hipSetDevice(h->device));
    { const int rc0 = check_out_buffer(h, pinned_out); if (rc0) return rc0; }   // (the completion flag is only sound for coherent pinned memory)
    char *dev = nullptr;
    HIPCHK(h, hipHostGetDevicePointer((void **)&dev, pinned_out, 0));
    // completion as in the unsharded combine: k_map2d's last workgroup stores a flag the host spins on (a stream
    // synchronisation notices the end of the stream several microseconds later); the count was published by k_posdens
    const uint32_t done_seq = ++h->combine_seq;
    const bool solo = h->sharded && h->world == 1;       // (every row is this rank's: no gathered densities, see gvom_combine_fuse)
    int rc = map2d_impl(h, !solo, solo, dev, true, nullptr, nullptr, done_seq);
    if (rc == GVOM_OK) rc = finish_combine(h, lk, done_seq);
    if (rc) return rc;
    if (origin_world) {
        const Fused &F = h->fused[h->cur];
        origin_world[0] = (double)F.origin[0] * h->prm.xy_resolution;
        origin_world[1] = (double)F.origin[1] * h->prm.xy_resolution;
        origin_world[2] = (double)F.origin[2] * h->prm.z_resolution;
    }
    return GVOM_OK;
}

VIS int gvom_get_state(gvom_t *h, gvom_state *out)
{
    if (!h || !out) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    memset(out, 0, sizeof *out);
    out->buffer_index = h->buffer_index;
    out->last_buffer_index = h->last_buffer_index;
    out->has_combined = h->has_combined ? 1 : 0;
    out->combined_cell_count = h->combined_cell_count;
    if (h->has_combined)
        for (int k = 0; k < 3; ++k) out->combined_origin[k] = (double)h->fused[h->cur].origin[k];
    for (int k = 0; k < 3; ++k) out->ego_position[k] = h->ego[k];
    return GVOM_OK;
}

VIS int gvom_slot_filled(gvom_t *h, int slot)
{
    if (!h || slot < 0 || slot >= h->prm.buffer_size) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    return h->slots[h->ring[slot]].filled ? 1 : 0;
}

VIS int gvom_read_dense(gvom_t *h, int which, int32_t *state, int32_t *hit, int32_t *total,
                        float *min_h, double origin[3], int64_t *cell_count)
{
    if (!h) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    const int32_t *st; const uint4 *cr; const uint32_t *tg; uint32_t ep; const int64_t *org; int64_t cnt;
    if (which == GVOM_WHICH_FUSED) {
        if (!h->has_combined) return GVOM_NO_DATA;
        const Fused &F = h->fused[h->cur];
        st = F.state; cr = (const uint4 *)F.rows.p; org = F.origin; cnt = F.count; tg = F.tags; ep = F.epoch;
    } else {
        if (which < 0 || which >= h->prm.buffer_size) return GVOM_ERR_INVALID;
        const Slot &s = h->slots[h->ring[which]];
        if (!s.filled) return GVOM_NO_DATA;
        st = s.state; cr = (const uint4 *)s.crows.p; org = s.origin; cnt = s.count; tg = s.tags; ep = s.epoch;
    }
    const size_t V = h->V;
    int32_t *tmp = nullptr;
    HIPCHK(h, hipMalloc((void **)&tmp, V * 16));
    int om[3] = {(int)floor_mod(org[0], h->prm.xy_size), (int)floor_mod(org[1], h->prm.xy_size),
                 (int)floor_mod(org[2], h->prm.z_size)};
    hipError_t e = join_second_stream(h);
    if (e == hipSuccess) e = gvom_launch_read_dense(h->stream, h->prm.xy_size, h->prm.z_size, om, h->sy_lo, h->sy_hi,
                                          tg, ep, st, cr,
                                          tmp, tmp + V, tmp + 2 * V, (float *)(tmp + 3 * V), nullptr);
    if (e == hipSuccess) e = sync_streams(h);
    if (e == hipSuccess && state) e = hipMemcpy(state, tmp, V * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && hit) e = hipMemcpy(hit, tmp + V, V * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && total) e = hipMemcpy(total, tmp + 2 * V, V * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && min_h) e = hipMemcpy(min_h, tmp + 3 * V, V * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && cell_count && cnt < 0) {       // a scan's occupied voxels are counted on demand
        std::vector<int32_t> stv(V);
        e = hipMemcpy(stv.data(), tmp, V * 4, hipMemcpyDeviceToHost);
        cnt = 0;
        for (size_t i = 0; i < V; ++i) cnt += stv[i] >= 0;
    }
    hipFree(tmp);
    HIPCHK(h, e);
    if (origin) for (int k = 0; k < 3; ++k) origin[k] = (double)org[k];
    if (cell_count) *cell_count = cnt;
    return GVOM_OK;
}

// Test hook / reference attributes metrics_buffer, combined_metrics (gvom.py:54-83,234,281; statistics
// handles only): rows_dense[V] = compact row of every occupied voxel of slot / fused map `which` in the
// reference's voxel order, -1 elsewhere.
VIS int gvom_read_rows(gvom_t *h, int which, int32_t *rows_dense)
{
    if (!h || !rows_dense) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    const int32_t *st; const uint32_t *tg; uint32_t ep; const int64_t *org;
    if (which == GVOM_WHICH_FUSED) {
        if (!h->has_combined) return GVOM_NO_DATA;
        const Fused &F = h->fused[h->cur];
        st = F.state; org = F.origin; tg = F.tags; ep = F.epoch;
    } else {
        if (which < 0 || which >= h->prm.buffer_size) return GVOM_ERR_INVALID;
        const Slot &sl = h->slots[h->ring[which]];
        if (!sl.filled) return GVOM_NO_DATA;
        st = sl.state; org = sl.origin; tg = sl.tags; ep = sl.epoch;
    }
    const size_t V = h->V;
    int32_t *tmp = nullptr;
    HIPCHK(h, hipMalloc((void **)&tmp, V * 4));
    int om[3] = {(int)floor_mod(org[0], h->prm.xy_size), (int)floor_mod(org[1], h->prm.xy_size),
                 (int)floor_mod(org[2], h->prm.z_size)};
    hipError_t e = join_second_stream(h);
    if (e == hipSuccess) e = gvom_launch_read_dense(h->stream, h->prm.xy_size, h->prm.z_size, om, h->sy_lo, h->sy_hi, tg, ep, st,
                                          nullptr, nullptr, nullptr, nullptr, nullptr, tmp);
    if (e == hipSuccess) e = sync_streams(h);
    if (e == hipSuccess) e = hipMemcpy(rows_dense, tmp, V * 4, hipMemcpyDeviceToHost);
    hipFree(tmp);
    HIPCHK(h, e);
    return GVOM_OK;
}

// out[j][0..9] = the statistics of compact row rows[j] of slot `which` (float64: {mean xyz, covariance
// xx xy xz yy yz zz, count}) or of the fused map (float32).  GVOM_NO_DATA without GVOM_FLAG_VOXEL_STATISTICS.
VIS int gvom_gather_metrics(gvom_t *h, int which, const int32_t *rows, int64_t n, void *out)
{
    if (!h || !rows || !out || n < 0) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    stats_demand(h);
    HIPCHK(h, hipSetDevice(h->device));
    const void *src; int f64;
    if (which == GVOM_WHICH_FUSED) {
        if (!h->has_combined || !h->fused[h->cur].has_metrics) return GVOM_NO_DATA;
        src = h->fused[h->cur].metrics.p; f64 = 0;
    } else {
        if (which < 0 || which >= h->prm.buffer_size) return GVOM_ERR_INVALID;
        const Slot &sl = h->slots[h->ring[which]];
        if (!sl.filled || !sl.has_metrics) return GVOM_NO_DATA;
        src = sl.metrics.p; f64 = 1;
    }
    if (n == 0) return GVOM_OK;
    const size_t esz = f64 ? 8 : 4;
    char *tmp = nullptr;
    HIPCHK(h, hipMalloc((void **)&tmp, (size_t)n * 4 + (size_t)n * 10 * esz));
    hipError_t e = join_second_stream(h);
    if (e == hipSuccess) e = hipMemcpy(tmp + (size_t)n * 10 * esz, rows, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = gvom_launch_gather_rows10(h->stream, f64, src, (const int32_t *)(tmp + (size_t)n * 10 * esz), n, tmp);
    if (e == hipSuccess) e = sync_streams(h);
    if (e == hipSuccess) e = hipMemcpy(out, tmp, (size_t)n * 10 * esz, hipMemcpyDeviceToHost);
    hipFree(tmp);
    HIPCHK(h, e);
    return GVOM_OK;
}

VIS int gvom_read_map2d(gvom_t *h, int which2d, double *out)
{
    if (!h || !out) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->has_combined || !h->maps_valid) return GVOM_NO_DATA;
    HIPCHK(h, hipSetDevice(h->device));
    size_t esz; int stride; const double *src = (const double *)map_ptr(h, which2d, &esz, &stride);
    if (!src) return GVOM_ERR_INVALID;
    const Fused &F = h->fused[h->cur];
    double *tmp = nullptr;
    HIPCHK(h, hipMalloc((void **)&tmp, h->cells2d * 8));
    hipError_t e = join_second_stream(h);
    if (e == hipSuccess) e = gvom_launch_unwrap_f64(h->stream, h->prm.xy_size, (int)floor_mod(F.origin[0], h->prm.xy_size),
                                          (int)floor_mod(F.origin[1], h->prm.xy_size), src, stride, tmp);
    if (e == hipSuccess) e = sync_streams(h);
    if (e == hipSuccess) e = hipMemcpy(out, tmp, h->cells2d * 8, hipMemcpyDeviceToHost);
    hipFree(tmp);
    HIPCHK(h, e);
    return GVOM_OK;
}

VIS int gvom_get_occupancy(gvom_t *h, uint8_t *out_xyz)
{
    if (!h || !out_xyz) return GVOM_ERR_INVALID;
    std::vector<int32_t> st;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        if (!h->has_combined) return GVOM_NO_DATA;
        st.resize(h->V);
    }
    int rc = gvom_read_dense(h, GVOM_WHICH_FUSED, st.data(), nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    const int xy = h->prm.xy_size, zs = h->prm.z_size;
    // reference: lookup.reshape((xy, xy, z), order='F') >= 0  -> out[x][y][z]
    for (int x = 0; x < xy; ++x)
        for (int y = 0; y < xy; ++y)
            for (int z = 0; z < zs; ++z)
                out_xyz[((size_t)x * xy + y) * zs + z] = st[(size_t)x + (size_t)y * xy + (size_t)z * xy * xy] >= 0;
    return GVOM_OK;
}

static int debug_maps(gvom_t *h, float *out7, float *out3)
{
    if (!h) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->has_combined || !h->maps_valid) return GVOM_NO_DATA;      // gvom.py:381-383
    HIPCHK(h, hipSetDevice(h->device));
    const Fused &F = h->fused[h->cur];
    const size_t n2 = h->cells2d;
    float *tmp = nullptr;
    HIPCHK(h, hipMalloc((void **)&tmp, n2 * 7 * 4));
    double org[3] = {(double)F.origin[0], (double)F.origin[1], (double)F.origin[2]};
    hipError_t e = join_second_stream(h);
    if (e == hipSuccess) e = gvom_launch_debug_height(h->stream, h->prm.xy_size, (int)floor_mod(F.origin[0], h->prm.xy_size),
                                            (int)floor_mod(F.origin[1], h->prm.xy_size), org,
                                            h->prm.xy_resolution, h->prm.z_resolution, h->height, h->hs, h->rough,
                                            h->slope_x, h->slope_y, out7 ? tmp : nullptr, h->guessed,
                                            out3 ? tmp : nullptr);
    if (e == hipSuccess) e = sync_streams(h);
    if (e == hipSuccess) e = hipMemcpy(out7 ? out7 : out3, tmp, n2 * (out7 ? 7 : 3) * 4, hipMemcpyDeviceToHost);
    hipFree(tmp);
    HIPCHK(h, e);
    return GVOM_OK;
}

// Gvom.make_debug_voxel_map (gvom.py:363-378, kernels :1333-1378, :454-473)
VIS int gvom_debug_voxel_map(gvom_t *h, float *out, int64_t max_rows, int64_t *rows)
{
    return gvom_debug_voxel_eigen(h, out, nullptr, max_rows, rows);
}

// the same, also returning the three eigenvalues of every row (reference attribute voxels_eigenvalues,
// gvom.py:1333-1378), row for row with `out`
VIS int gvom_debug_voxel_eigen(gvom_t *h, float *out, float *eigen, int64_t max_rows, int64_t *rows)
{
    if (!h || !out || max_rows < 0) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    stats_demand(h);
    if (!h->has_combined || !h->fused[h->cur].has_metrics) return GVOM_NO_DATA;
    HIPCHK(h, hipSetDevice(h->device));
    const gvom_params &p = h->prm;
    const Fused &F = h->fused[h->cur];
    Map2dParams P;
    memset(&P, 0, sizeof P);
    P.xy = p.xy_size; P.zs = p.z_size;
    P.om[0] = (int)floor_mod(F.origin[0], p.xy_size);
    P.om[1] = (int)floor_mod(F.origin[1], p.xy_size);
    P.om[2] = (int)floor_mod(F.origin[2], p.z_size);
    P.y_lo = h->sy_lo; P.y_hi = h->sy_hi;
    P.xy_res = p.xy_resolution; P.z_res = p.z_resolution;
    P.nseg = h->nseg; P.epoch = F.epoch;
    float *tmp = nullptr;
    const size_t mr = (size_t)(max_rows > 0 ? max_rows : 1);
    HIPCHK(h, hipMalloc((void **)&tmp, mr * 44));         // 8 + 3 floats per row
    float *tmp_e = eigen ? tmp + mr * 8 : nullptr;
    hipError_t e = join_second_stream(h);
    if (e == hipSuccess) e = hipMemsetAsync(h->counters + 12, 0, 8, h->stream);
    if (e == hipSuccess)
        e = gvom_launch_voxel_cloud(h->stream, P, (double)F.origin[0], (double)F.origin[1], (double)F.origin[2],
                                    F.state, F.tags, (const uint4 *)F.rows.p,
                                    (const float *)F.metrics.p, tmp, tmp_e, max_rows,
                                    (unsigned long long *)(h->counters + 12));
    unsigned long long cnt = 0;
    if (e == hipSuccess) e = sync_streams(h);
    if (e == hipSuccess) e = hipMemcpy(&cnt, h->counters + 12, 8, hipMemcpyDeviceToHost);
    const int64_t nrows = (int64_t)cnt < max_rows ? (int64_t)cnt : max_rows;
    if (e == hipSuccess && nrows > 0) e = hipMemcpy(out, tmp, (size_t)nrows * 32, hipMemcpyDeviceToHost);
    if (e == hipSuccess && nrows > 0 && eigen) e = hipMemcpy(eigen, tmp_e, (size_t)nrows * 12, hipMemcpyDeviceToHost);
    hipFree(tmp);
    HIPCHK(h, e);
    if (rows) *rows = (int64_t)cnt;
    return GVOM_OK;
}

VIS int gvom_debug_height_map(gvom_t *h, float *out) { return out ? debug_maps(h, out, nullptr) : GVOM_ERR_INVALID; }
VIS int gvom_debug_inferred_height_map(gvom_t *h, float *out) { return out ? debug_maps(h, nullptr, out) : GVOM_ERR_INVALID; }

VIS int gvom_get_scan_stats(gvom_t *h, gvom_scan_stats *out)
{
    if (!h || !out) return GVOM_ERR_INVALID;
    int slot;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        slot = h->last_buffer_index;
        if (!h->slots[h->ring[slot]].filled) return GVOM_NO_DATA;
    }
    const size_t V = h->V;
    std::vector<int32_t> hit(V), total(V);
    int64_t cells = 0;
    int rc = gvom_read_dense(h, slot, nullptr, hit.data(), nullptr, nullptr, nullptr, &cells);
    if (rc) return rc;
    // total of free voxels lives in the state code; read it densely
    std::vector<int32_t> state(V);
    rc = gvom_read_dense(h, slot, state.data(), nullptr, total.data(), nullptr, nullptr, nullptr);
    if (rc) return rc;
    int64_t sh = 0, st = 0;
    for (size_t i = 0; i < V; ++i) {
        sh += hit[i];
        st += state[i] >= 0 ? (int64_t)total[i] : (int64_t)(-(int64_t)state[i] - 1);
    }
    std::lock_guard<std::mutex> lk(h->mu);
    out->points = h->slots[h->ring[slot]].stats.points;
    out->cells = cells; out->sum_hit = sh; out->sum_total = st;
    return GVOM_OK;
}

VIS int gvom_last_stage_ms(gvom_t *h, float ms[GVOM_N_STAGES])
{
    if (!h || !ms) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->ev_scan || h->ev_fuse || h->ev_map) {
        HIPCHK(h, hipSetDevice(h->device));
        HIPCHK(h, sync_streams(h));
        collect_stage_ms(h);
    }
    for (int k = 0; k < GVOM_N_STAGES; ++k) ms[k] = h->stage_ms[k];
    return GVOM_OK;
}

// host-side phase times in microseconds per call, averaged since creation (GVOM_HOST_TIMING=1):
// [0] scan launches [1] scan wait [2] combine launches [3] combine wait [4] output copies
VIS int gvom_host_timing(gvom_t *h, double us[8])
{
    if (!h || !us) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    for (int k = 0; k < 8; ++k) us[k] = h->host_calls ? h->host_ns[k] / h->host_calls / 1e3 : 0.0;
    return GVOM_OK;
}

// Performance knobs that never change a result: "segs" = step segments per ray in k_trace, "period" =
// committing steps between two flushes of a wave's line cache, "ep_row" = dispatch row of the endpoint
// blocks, "prio" = steps of remaining walk per issue-priority level of a trace wave (0 / 0 / -2 / -1: automatic).
VIS int gvom_set_tuning(gvom_t *h, const char *name, int value)
{
    if (!h || !name) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    if (!strcmp(name, "segs")) h->tune_segs = value;
    else if (!strcmp(name, "ep_row")) h->tune_ep_row = value;
    else if (!strcmp(name, "period")) h->tune_period = value;
    else if (!strcmp(name, "prio")) h->tune_prio = value;
    else if (!strcmp(name, "interleave")) h->tune_ilv = value;
    else if (!strcmp(name, "flag_kernel")) h->tune_flag_kernel = value;
    else if (!strcmp(name, "fuse1")) h->tune_fuse1 = value;
    else if (!strcmp(name, "encfuse")) h->tune_encfuse = value;
    else if (!strcmp(name, "dirsort")) h->tune_dirsort = value;
    else if (!strcmp(name, "fastdiv")) h->tune_fastdiv = value;
    else if (!strcmp(name, "eager")) { h->tune_eager = value; h->eager_waste = 0; }
    else if (!strcmp(name, "exported")) h->exported = value != 0;       // (set by the peer transport, gvom_comm.hip)
#ifdef GVOM_HOOKS
    // test hooks (include/gvom_hip_test.h; lib/libgvom_hip_test.so only)
    else if (!strcmp(name, "churn")) h->tune_churn = value;             // a fresh endpoint send region every scan
    else if (!strcmp(name, "epoch_bias")) h->epoch += (uint32_t)value;   // advances the tile-epoch counter (towards its wrap)
#endif
    else return GVOM_ERR_INVALID;
    return GVOM_OK;
}

VIS int gvom_get_tuning(gvom_t *h, const char *name, int *value)
{
    if (!h || !name || !value) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    static const char *const names[5] = {"segs", "period", "ep_row", "prio", "interleave"};
    for (int k = 0; k < 5; ++k)
        if (!strcmp(name, names[k])) { *value = h->last_knobs[k]; return GVOM_OK; }
    if (!strcmp(name, "dirsort")) { *value = h->last_dirsort; return GVOM_OK; }
    if (!strcmp(name, "eager_adopted")) { *value = h->eager_stat[0]; return GVOM_OK; }
    if (!strcmp(name, "eager_dropped")) { *value = h->eager_stat[1]; return GVOM_OK; }
    if (!strcmp(name, "fastdiv")) { *value = h->tune_fastdiv == 0 ? 0 : h->fastdiv_ok; return GVOM_OK; }   // bit 0 / 1: xy / z resolution divided by reciprocal
    return GVOM_ERR_INVALID;
}

VIS int gvom_set_profiling(gvom_t *h, int on)
{
    if (!h) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    h->profiling = on != 0;
    return GVOM_OK;
}

VIS void *gvom_stream(gvom_t *h) { return h ? (void *)h->stream : nullptr; }
// A value that differs between any two handles of the process and changes whenever one of the handle's SEND regions
// (gvom_shard_buffer GVOM_XBUF_SEND_*; the only grow-only buffers another rank ever reads) has been re-allocated:
// whoever caches something derived from their addresses (the peer transport's exported allocations) knows when to
// look again.  (GVOM_BUF_HEIGHT_MAPS and the quad regions are allocated once, with the handle.)
VIS uint64_t gvom_alloc_generation(gvom_t *h) { return h ? h->alloc_gen : 0; }
// The same for ONE region (which = GVOM_XBUF_SEND_* or -1 for GVOM_BUF_HEIGHT_MAPS): a value that names the allocation the
// region lies in -- it changes exactly when that allocation is replaced (and differs between handles).
// The region `which` (GVOM_XBUF_SEND_IDS / _QUADS / _EPS / _RETURNS, or -1: GVOM_BUF_HEIGHT_MAPS) moves into a FRESH allocation of the
// same size, contents included; the old one is parked (a peer may have it mapped).  For the transport: an allocation the HSA
// runtime refuses to export, or that a peer cannot open, is replaced by one that has no history.
VIS int gvom_shard_renew_region(gvom_t *h, int which)
{
    if (!h || !h->sharded) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    void **pp = nullptr;
    size_t bytes = 0;
    uint64_t *gen = nullptr;
    switch (which) {
    case GVOM_XBUF_SEND_IDS: pp = (void **)&h->x_send_ids; bytes = exportable_size(h->x_Q * 4); gen = &h->fixed_gen[0]; break;
    case GVOM_XBUF_SEND_QUADS: pp = &h->x_send_pay; bytes = exportable_size(h->x_Q * 1024); gen = &h->fixed_gen[1]; break;
    case GVOM_XBUF_SEND_EPS: pp = &h->x_send_eps.p; bytes = h->x_send_eps.bytes; gen = &h->x_send_eps.gen; break;
    case GVOM_XBUF_SEND_RETURNS: pp = &h->x_send_sp.p; bytes = h->x_send_sp.bytes; gen = &h->x_send_sp.gen; break;
    case -1: pp = (void **)&h->hmaps; bytes = exportable_size(h->cells2d * 24); gen = &h->fixed_gen[2]; break;
    default: return GVOM_ERR_INVALID;
    }
    if (!*pp || !bytes) return GVOM_NO_DATA;
    void *np = nullptr;
    HIPCHK(h, hipMalloc(&np, bytes));                      // (fresh: never from the pool)
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(np, *pp, bytes, hipMemcpyDeviceToDevice));
    HIPCHK(h, hipDeviceSynchronize());
    Buf old; old.p = *pp; old.bytes = bytes; old.gen = *gen;
    h->retired.push_back(old);
    *pp = np;
    *gen = ++g_alloc_generation;
    if (which == -1) { h->height = h->hmaps; h->inferred = h->hmaps + h->prm.xy_size; }
    if (which == GVOM_XBUF_SEND_EPS || which == GVOM_XBUF_SEND_RETURNS) h->alloc_gen = *gen;
    return GVOM_OK;
}

VIS uint64_t gvom_region_generation(gvom_t *h, int which)
{
    if (!h) return 0;
    if (which == GVOM_XBUF_SEND_EPS) return h->x_send_eps.gen ? h->x_send_eps.gen : h->handle_gen;
    if (which == GVOM_XBUF_SEND_RETURNS) return h->x_send_sp.gen ? h->x_send_sp.gen : h->handle_gen;
    if (which == GVOM_XBUF_SEND_IDS && h->fixed_gen[0]) return h->fixed_gen[0];
    if (which == GVOM_XBUF_SEND_QUADS && h->fixed_gen[1]) return h->fixed_gen[1];
    if (which == -1 && h->fixed_gen[2]) return h->fixed_gen[2];
    return h->handle_gen;
}

VIS const char *gvom_last_error(gvom_t *h) { return h ? h->err.c_str() : "null handle"; }

VIS int gvom_backend_info(char *buf, size_t len)
{
    if (!buf || len == 0) return GVOM_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        snprintf(buf, len, "libgvom_hip abi %d (gfx950 code object); no HIP device visible", GVOM_ABI_VERSION);
        return GVOM_ERR_NO_DEVICE;
    }
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) return GVOM_ERR_HIP;
    snprintf(buf, len, "libgvom_hip abi %d; device0=%s arch=%s CUs=%d; %d device(s)", GVOM_ABI_VERSION,
             pr.name, pr.gcnArchName, pr.multiProcessorCount, ndev);
    return GVOM_OK;
}

VIS int gvom_abi_version(void) { return GVOM_ABI_VERSION; }

#ifdef GVOM_DIAG
// diagnostic library only (not part of include/gvom_hip.h): k_trace's per-wave timeline of the last scan,
// 4 uint64 per wave {start, set-up done (0: the wave left before it walked), end, HW_ID | XCC_ID << 32} in
// dispatch order [row][workgroup][wave]; grid[0] workgroups per row, grid[1] rows (tools/trace_timeline.py)
VIS int gvom_diag_timeline(gvom_t *h, unsigned long long *out, int64_t max_words, int grid[2])
{
    if (!h || !out || !grid) return GVOM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, sync_streams(h));
    grid[0] = h->tl_grid[0]; grid[1] = h->tl_grid[1];
    const int64_t words = (int64_t)grid[0] * grid[1] * 8 * 4 + 8 +     // + 8 summary words (steps by lookup mode)
                          ((int64_t)grid[0] * grid[1] * 8 / 64 + 1) * 128;   // + the step profiles of every 64th wave
    if (!h->tl.p || words <= 8) return GVOM_NO_DATA;
    HIPCHK(h, hipMemcpy(out, h->tl.p, (size_t)(words < max_words ? words : max_words) * 8, hipMemcpyDeviceToHost));
    return GVOM_OK;
}
#endif

}  // extern "C"
