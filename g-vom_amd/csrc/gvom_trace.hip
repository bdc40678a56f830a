// gvom_trace.hip -- the SCAN kernels of libgvom_hip.so (gfx950, wave64), reference /root/reference/scripts/gvom.py = "gvom.py:NNN":
//
//   k_trace   gvom.py:1040-1056 (transform) + :1060-1150 (hit + dominant-axis DDA)
//             + the row-claim half of :1154-1160 (first hit of a voxel claims its compact row)
//             + :1303-1329 (min-height: a third accumulator, atomicMax of 1.0f's bits minus the sample's)
//   k_pack / k_shard_publish / k_unpack_quads / k_unpack_eps   the rank exchange of a sharded map around it
//   k_dirbin_hist / k_dirbin_scatter   directional order of clouds whose own order does not suit the trace
//   k_layout_probe                     sub-cloud / order verdicts for the following scans
//
// Numerics are the reference's as executed by the Numba simulator:
// compile with -ffp-contract=off, IEEE division/sqrt, no fast-math.  Integer results are
// bit-exact; only log()/atan2() may differ from glibc in the last ulp.
// No MFMA: there is no dense contraction on this path.
#include "gvom_device.h"

// ---- wave-private accumulator-line cache (k_trace) -----------------------------------------------
// One 64-entry direct-mapped table per wave in LDS: key = accumulator line (64 B = 4x4 (x,y)
// patch at one z), 16 counters per entry.  DDA steps add into the table with LDS atomics; at the
// end of the wave's item the wave flushes it cooperatively, 4 lines per instruction with 16 lanes
// per line, so one line costs ONE memory-side atomic request however many steps of however many
// lanes fell into it.
#define LC_EMPTY 0xFFFFFFFFu
#define LC_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT)
#define LC_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT)
// slot s keeps its key at keys[LC_KEYPOS(s)]: the flush handles entry 4*it + g in iteration it of
// lane group g, so group g finds its 16 keys in 16 consecutive words (4 x 16-byte LDS reads)
#define LC_KEYPOS(s) ((((s) & 3u) << 4) | ((s) >> 2))
__device__ __forceinline__ void lc_flush(const ScanParams &P, uint32_t *keys, uint32_t *cnt, uint32_t *total, int lane)
{
    // entry e = 4*it + (lane >> 4), counter c = lane & 15  <=>  cnt[it*64 + lane]: linear LDS reads.
    // Two batches of 8 entries: 16 registers in flight instead of 32 (the step loop's own state has
    // to stay in registers across an in-loop flush).
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint32_t v[8], k[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const v4i kk = *(const v4i *)&keys[g * 16 + half * 8 + q * 4];
            k[q * 4 + 0] = (uint32_t)kk.x; k[q * 4 + 1] = (uint32_t)kk.y; k[q * 4 + 2] = (uint32_t)kk.z; k[q * 4 + 3] = (uint32_t)kk.w;
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) v[it] = LC_LD(&cnt[(half * 8 + it) * 64 + lane]);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            if (v[it] != 0u) {
                if (!GVOM_DBG(P, 1)) atomicAdd(&total[(k[it] << 4) + (uint32_t)c], v[it]);
                LC_ST(&cnt[(half * 8 + it) * 64 + lane], 0u);
            }
        }
    }
    LC_ST(&keys[lane], LC_EMPTY);
}

// One lidar return: load (any record layout, gvom_ros.py:93-109) + optional rigid transform in f64,
// source order, rounded to the cloud's dtype (gvom.py:1040-1056).
template <typename T>
__device__ __forceinline__ void load_return(const ScanParams &P, const T *__restrict__ in, long stride, long i,
                                            T &x, T &y, T &z)
{
    if (sizeof(T) == 8 && P.in_f32) {                    // PointCloud2 FLOAT32 fields, computed in f64
        const float *p = reinterpret_cast<const float *>(in) + i * stride;
        x = (T)p[P.off[0]]; y = (T)p[P.off[1]]; z = (T)p[P.off[2]];
    } else {
        const T *p = in + i * stride;
        x = p[P.off[0]]; y = p[P.off[1]]; z = p[P.off[2]];
    }
    if (P.has_tf) {
        const double dx = (double)x, dy = (double)y, dz = (double)z;
        const double o0 = ((dx * P.tf[0] + dy * P.tf[1]) + dz * P.tf[2]) + P.tf[3];
        const double o1 = ((dx * P.tf[4] + dy * P.tf[5]) + dz * P.tf[6]) + P.tf[7];
        const double o2 = ((dx * P.tf[8] + dy * P.tf[9]) + dz * P.tf[10]) + P.tf[11];
        x = (T)o0; y = (T)o1; z = (T)o2;
    }
}

// Window voxel of a ray position (gvom.py:1121-1144: floor((f64)p - origin), inside test).
// LIT = false: the window origin is an integer (gvom.py:124-126 floors it), so
// floor((double)p - origin) == (int)floorf(p) - origin -- no f64 in the lookup.  The f64 subtraction
// rounds across an integer only when p lies within half an f64 ulp BELOW an integer, which an f32 p
// can only do just below 0 (|p| < 2^-23, given |origin| < 2^30, which the host checks before selecting
// the integer form); callers use LIT = true (the reference's literal f64 expression) wherever a
// coordinate may come that close to zero, and always when |origin| >= 2^30.
// Returns "inside the window"; wx/wy/wz are only meaningful then.
// (o0..o2, uxy, zpad: the integer form's wave-uniform constants -- (int)origin, xy, xy - zs -- handed in by walk_steps,
// which pins them in scalar registers across its loop; the literal form reads P)
template <bool LIT>
__device__ __forceinline__ bool window_voxel(const ScanParams &P, float px, float py, float pz,
                                             uint32_t &wx, uint32_t &wy, uint32_t &wz,
                                             uint32_t o0 = 0, uint32_t o1 = 0, uint32_t o2 = 0, uint32_t uxy_ = 0, uint32_t zpad = 0)
{
    if (LIT) {
        const double fx = floor((double)px - P.origin[0]);
        const double fy = floor((double)py - P.origin[1]);
        const double fz = floor((double)pz - P.origin[2]);
        const bool in = fx >= 0.0 && fx < (double)P.xy && fy >= 0.0 && fy < (double)P.xy && fz >= 0.0 && fz < (double)P.zs;
        wx = in ? (uint32_t)(int)fx : 0u; wy = in ? (uint32_t)(int)fy : 0u; wz = in ? (uint32_t)(int)fz : 0u;
        return in;
    }
    wx = (uint32_t)cvt_floor_i32(px) - o0;
    wy = (uint32_t)cvt_floor_i32(py) - o1;
    wz = (uint32_t)cvt_floor_i32(pz) - o2;
    // ONE compare for the three axes (its result is the lane mask the step body needs, no boolean to
    // re-materialise): z is padded up to the xy bound with a saturating add ("negative" differences are
    // huge unsigned values and stay huge).  Requires z_size <= xy_size: callers take the literal form for
    // grids taller than wide.
    return max(max(wx, wy), __builtin_elementwise_add_sat(wz, zpad)) < uxy_;
}
// Number of DDA steps the reference's length test lets a ray take (gvom.py:1127,1149):
//   length_0 = 0, length_j = fl(length_{j-1} + step_len) in f64; step j runs iff length_{j-1} < lim,
// i.e. n = the smallest j with length_j >= lim (0 if lim <= 0), capped at `cap` + 1 (callers only need
// to know "more than cap").  The accumulated sum differs from j * step_len by at most j^2 * step_len *
// 2^-53, so n = ceil(lim / step_len) unless lim lies within that band of a multiple of step_len; only
// then (probability ~1e-12 per ray) the sum is accumulated literally.
__device__ __forceinline__ uint32_t ray_steps(double lim, double step_len, double inv_step, uint32_t cap)
{
    if (!(0.0 < lim)) return 0u;
    const double q = lim * inv_step;                      // ~ lim / step_len (inv_step ~ 1 / step_len: any error is caught by the band test)
    if (!(q < (double)cap + 2.0)) return cap + 1u;                        // also inf / NaN quotients: literal path below never needed
    const double jc = ceil(q);
    const double e = (jc * jc) * step_len * 0x1p-51 + step_len * 0x1p-50;
    const double lo = (jc - 1.0) * step_len, hi = jc * step_len;
    if (lo + e < lim && hi - e >= lim) return (uint32_t)jc;
    uint32_t n = 0;
    double length = 0.0;
    while (length < lim && n <= cap) { length += step_len; ++n; }
    return n;
}

// (double)x / d for a FLOAT32 coordinate x and a wave-uniform divisor d (xy_res, z_res), bit for bit, without the divide
// (an IEEE f64 division is ~15 instructions on gfx950: v_div_scale x2, v_rcp_f64, four Newton v_fma_f64, v_div_fmas,
// v_div_fixup ...): with r = RN(1 / d) from the host, q = x * r is within an ulp of the quotient, e = fma(-q, d, x) is its
// EXACT residual and fma(e, r, q) the correctly rounded quotient (Markstein's correction step).  Whether that holds for a
// given d is not taken from a theorem but CHECKED: rounding depends on the significands only (scaling x by a power of two
// scales q, e and the result exactly; no float32 x brings any of them near the ends of the f64 range for 2^-64 < d < 2^64), and
// a float32 has 2^23 significands -- gvom_create tries them all against the divide (verify_fastdiv, once per divisor and
// process) and clears the bit in P.fastdiv if one differs.  Zeros and non-finite x keep x * r, which is the quotient there
// (signed zero, inf, NaN).  Explicit fma() calls are not subject to -ffp-contract=off.  T = double (clouds handed over in
// float64): the IEEE divide, always.
template <typename T>
__device__ __forceinline__ double div_by_res(T x, double d, double r, bool fast)
{
    if (sizeof(T) == 4 && fast) {
        const double xd = (double)x;
        const double q = xd * r;
        const double e = __builtin_fma(-q, d, xd);
        const double q2 = __builtin_fma(e, r, q);
        return (fabs(xd) < INFINITY && xd != 0.0) ? q2 : q;
    }
    return (double)x / d;
}

// Endpoint voxel of one return (gvom.py:1070-1086): storage index L, accumulator index A, storage row
// sy, and its min-height sample (gvom.py:1303-1329).
struct Endpoint { bool ingrid; uint32_t L, A, mbits; int sy; };
template <typename T>
__device__ __forceinline__ Endpoint endpoint_of(const ScanParams &P, bool pass, T x, T y, T z)
{
    Endpoint E;
    E.ingrid = false; E.L = 0; E.A = 0; E.mbits = 0; E.sy = 0;
    if (pass) {
        const double fx = floor(div_by_res<T>(x, P.xy_res, P.drcp[0], P.fastdiv & 1) - P.origin[0]);
        const double fy = floor(div_by_res<T>(y, P.xy_res, P.drcp[0], P.fastdiv & 1) - P.origin[1]);
        const double az = div_by_res<T>(z, P.z_res, P.drcp[1], P.fastdiv & 2) - P.origin[2];
        const double fz = floor(az);
        if (fx >= 0.0 && fx < (double)P.xy && fy >= 0.0 && fy < (double)P.xy && fz >= 0.0 && fz < (double)P.zs) {
            E.ingrid = true;
            const int sx = wrap_add((int)fx, P.om[0], P.xy);
            const int sy = wrap_add((int)fy, P.om[1], P.xy);
            const int sz = wrap_add((int)fz, P.om[2], P.zs);
            E.sy = sy;
            E.L = ((uint32_t)sy * P.zs + sz) * P.xy + sx;
            E.A = acc_idx(sx, sy, sz, P.zs, P.sxq);
            // local_point[2], f64 -> f32 (gvom.py:1326,1329): in [0, 1], so the float order equals the
            // order of its bit pattern; kept as 1.0f's bits MINUS the value's (0 = the 1.0f the reference
            // initialises with, gvom.py:1014-1015) so that the accumulator is zero between scans
            E.mbits = 0x3f800000u - __float_as_uint((float)(az - fz));
        }
    }
    return E;
}

// hit += 1, total += 1, min-height for the endpoints of a wave (gvom.py:1087-1090, 1329); the voxel's
// compact row is `row` of (one of) its returns -- no row counter, no barrier.  All atomics are
// fire-and-forget.
__device__ __forceinline__ void endpoint_commit(const ScanParams &P, int lane, long row, bool ingrid, uint32_t L, uint32_t A,
                                                uint32_t mbits, uint32_t *hit, uint32_t *total, uint32_t *mh, int32_t *state,
                                                uint32_t *tags, double *stat_sums, double *stat_base, uint32_t *stat_rowvox)
{
    // neighbouring returns of a beam end in the same voxel (33 consecutive azimuths at 2 m range):
    // the first lane of each run of equal voxels adds the whole run
    const uint32_t key = ingrid ? A : (0xFFFFFF00u | (uint32_t)lane);
    const uint32_t leftk = (uint32_t)__builtin_amdgcn_update_dpp((int)~key, (int)key, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    const bool ehead = ingrid && leftk != key;
    const unsigned long long followers = lanes(ingrid) & ~lanes(ehead);
    if (ingrid && mbits && !GVOM_DBG(P, 4)) atomicMax(&mh[A], mbits);
    if (ehead && !GVOM_DBG(P, 4)) {
        const uint32_t run = (uint32_t)__ffsll((long long)~((followers >> lane) >> 1));   // 1 + followers
        // ONE add per run: the endpoint's own `total += 1` (gvom.py:1090) is not accumulated -- it always equals its
        // `hit += 1` (gvom.py:1089), so k_encode adds hit to the ray passes instead (two memory-side requests per run
        // head and line instead of three)
        atomicAdd(&hit[A], run);
        tags[(L / P.xy) * P.nseg + ((L % P.xy) >> 6)] = P.epoch;   // stamp the tile (idempotent)
        // the voxel's compact row = this return's.  Several runs (other waves) may end in the same
        // voxel: the last store wins, every candidate is a valid, unique row.
        state[L] = (int32_t)row;
        if (stat_sums) {                            // optional statistics: zeroed metrics (gvom.py:1011-1012)
            for (int m = 0; m < 10; ++m) { stat_sums[(size_t)row * 10 + m] = 0.0; stat_base[(size_t)row * GVOM_BASE_PITCH + m] = 0.0; }
            stat_rowvox[row] = L;                   // row -> voxel, for the per-row neighbour gather
        }
    }
}

// Ray set-up of one return (gvom.py:1093-1118): per-step increments in natural (x, y, z) order, the
// f64 step length and the length limit of the reference's loop test.
struct RaySetup { float incx, incy, incz; double step_len, inv_step, lim; bool finite; };
template <typename T>
__device__ __forceinline__ RaySetup ray_setup(const ScanParams &P, T x, T y, T z)
{
    const float e0 = (float)div_by_res<T>(x, P.xy_res, P.drcp[0], P.fastdiv & 1);
    const float e1 = (float)div_by_res<T>(y, P.xy_res, P.drcp[0], P.fastdiv & 1);
    const float e2 = (float)div_by_res<T>(z, P.z_res, P.drcp[1], P.fastdiv & 2);
    float s0 = e0 - P.pt0[0], s1 = e1 - P.pt0[1], s2 = e2 - P.pt0[2];
    const float ss = (s0 * s0 + s1 * s1) + s2 * s2;
    // math.sqrt -> f64; GVOM_FLAG_CUDA_F32_SQRT: sqrt of the f32 sum in f32, as real
    // Numba-CUDA types it (gvom.py:1109-1114)
    const double ray_length = P.f32_sqrt ? (double)sqrtf(ss) : sqrt((double)ss);
    s0 = (float)((double)s0 / ray_length);
    s1 = (float)((double)s1 / ray_length);
    s2 = (float)((double)s2 / ray_length);
    const float a0 = fabsf(s0), a1 = fabsf(s1), a2 = fabsf(s2);
    const float smax = py_maxf(a0, py_maxf(a1, a2));
    int si = 0;
    if (smax == a1) si = 1;
    if (smax == a2) si = 2;                              // ties: z over y over x
    const float sd  = si == 0 ? s0 : (si == 1 ? s1 : s2);
    const float so1 = si == 0 ? s1 : (si == 1 ? s2 : s0);
    const float so2 = si == 0 ? s2 : (si == 1 ? s0 : s1);
    const float adom = fabsf(sd);
    const float dir = sd / adom;
    const float inc1 = so1 / adom;
    const float inc2 = so2 / adom;
    const double step_len = fabs(1.0 / (double)sd);
    const double lim = ray_length - 1.0;
    // natural (x, y, z) order: the same three f32 additions per step as the reference's
    // (dominant, other, other) triple, without the axis permutation
    const float incx = si == 0 ? dir : (si == 1 ? inc2 : inc1);
    const float incy = si == 0 ? inc1 : (si == 1 ? dir : inc2);
    const float incz = si == 0 ? inc2 : (si == 1 ? inc1 : dir);
    // non-finite increments (degenerate returns): the reference's first step lands on NaN/inf,
    // which is outside the grid, and the ray ends without an update
    const bool finite = fabsf(incx) < INFINITY && fabsf(incy) < INFINITY && fabsf(incz) < INFINITY;
    RaySetup R;
    R.incx = incx; R.incy = incy; R.incz = incz; R.step_len = step_len; R.inv_step = fabs((double)sd); R.lim = lim; R.finite = finite;
    return R;
}

// The step loop's wave-uniform constants, read from the kernel arguments ONCE per wave: twelve scalar registers that stay put.
struct WalkConsts { uint32_t uxy, uzs, usxq, unseg, om0, om1, om2, o0, o1, o2, zpad, epoch;
#ifdef GVOM_DIAG
    unsigned long long *prof;   // diagnostic build (GVOM_TRACE_STEPPROF): this wave's step profile block (nullptr: not sampled)
    uint32_t prof_n;            // steps recorded so far
#endif
};
__device__ __forceinline__ WalkConsts walk_consts(const ScanParams &P)
{
    WalkConsts C;
    C.uxy = (uint32_t)P.xy; C.uzs = (uint32_t)P.zs; C.usxq = (uint32_t)P.sxq; C.unseg = (uint32_t)P.nseg;
    C.om0 = (uint32_t)P.om[0]; C.om1 = (uint32_t)P.om[1]; C.om2 = (uint32_t)P.om[2];
    C.o0 = (uint32_t)(int)P.origin[0]; C.o1 = (uint32_t)(int)P.origin[1]; C.o2 = (uint32_t)(int)P.origin[2];
    C.zpad = C.uxy - C.uzs; C.epoch = P.epoch;
    return C;
}

// ------------------------------------------------------------------------------------------
// walk_steps: at most `steps` lock-step DDA steps of a 64-ray bundle, total += 1 per step
// (gvom.py:1119-1150).  The step body is straight-line: ray state in natural (x,y,z) order (x and y as
// one packed f32 add), the step counter and the mask of rays still running (`alive`) on the scalar unit,
// voxel lookup in 32-bit integers, left neighbour's key by a DPP wave shift; lanes stepping into the same
// voxel as their left neighbour are merged (run heads and run lengths by mask arithmetic on the scalar unit,
// the head mask goes straight into EXEC) and the merged adds go into the wave-private LDS line cache
// (lc_flush), flushed after the run with one memory-side request per line; tile tags stamped on cache misses only.
// LIT: the reference's literal f64 lookup instead of the integer one (window_voxel).
// NOWIN (power-of-two grids): the run stays inside the window (walk_item has checked its first and last positions
// with a margin): no window test, storage coordinates straight from the floor.
// ------------------------------------------------------------------------------------------
template <bool LIT, bool P2, bool NOWIN>
__device__ __forceinline__ void walk_steps(const ScanParams &P, int lane, uint32_t &j, uint32_t cnt, float &px_, float &py_, float &pz,
                                           float incx, float incy, float incz, bool &active, int steps,
                                           uint32_t *lck, uint32_t *lcc, uint32_t *total, uint32_t *tags, const WalkConsts &C)
{
    const uint32_t uxy = C.uxy, uzs = C.uzs, usxq = C.usxq, unseg = C.unseg, om0 = C.om0, om1 = C.om1, om2 = C.om2;
    const uint32_t o0 = C.o0, o1 = C.o1, o2 = C.o2, zpad = C.zpad, epoch = C.epoch;
    lds_u32 *const keys3 = (lds_u32 *)lck;
    lds_u32 *const cnt3 = (lds_u32 *)lcc;
    glb_u32 *const total1 = (glb_u32 *)total;
    uint32_t memo = LC_EMPTY;
    uint32_t ju = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);      // the step counter is the same in every lane
    // Who takes part in a step: the rays that were running after the previous one (`alive`, a lane mask on the scalar unit)
    // and are still inside the window; a ray runs on while it has steps left (gvom.py:1127).  The loop ends with the run's
    // last step (cnt_run) or when no ray of the bundle runs any more: ONE condition, alive != 0.
    const uint32_t cnt_run = min(cnt, ju + (uint32_t)steps);
    unsigned long long alive = lanes(active);
    v2f pxy = {px_, py_};
    const v2f incxy = {incx, incy};
    // left neighbour's key: lane 0 has none and keeps this value, which no accumulator index equals
    uint32_t leftk = 0xFFFFFFFFu;
    unsigned long long cmask;
#ifdef GVOM_DIAG
    // step profile (sampled waves): s_memtime at the top of a step, in front of its head region, behind it, and at the loop's end
    WalkConsts &CW = const_cast<WalkConsts &>(C);
#define PROF_STAMP(k) do { if (CW.prof && CW.prof_n < 32u) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) CW.prof[CW.prof_n * 4u + (k)] = t_; } } while (0)
#else
#define PROF_STAMP(k) do { } while (0)
#endif
    do {
        PROF_STAMP(0);
        ++ju;
        // every lane computes (a finished ray's lanes produce values nobody uses): no divergent
        // region around the arithmetic
        pxy += incxy; pz += incz;
        uint32_t sx, sy, sz;                                              // toroidal storage coordinates
        uint32_t wx, wy, wz;                                              // window voxel
        if (NOWIN && P2) {
            // every ray of the bundle that is still running stays inside the window for this whole run (walk_item has
            // checked the run's first and last position with a margin): no window test, and on power-of-two grids the
            // storage coordinate comes straight from the floor -- floor(p) - o + om, wrapped by the mask
            sx = ((uint32_t)cvt_floor_i32(pxy.x) + (om0 - o0)) & (uxy - 1u);
            sy = ((uint32_t)cvt_floor_i32(pxy.y) + (om1 - o1)) & (uxy - 1u);
            sz = ((uint32_t)cvt_floor_i32(pz) + (om2 - o2)) & (uzs - 1u);
            cmask = alive;
        } else {
            const bool inwin = window_voxel<LIT>(P, pxy.x, pxy.y, pz, wx, wy, wz, o0, o1, o2, uxy, zpad);
            if (P2) { sx = (wx + om0) & (uxy - 1u); sy = (wy + om1) & (uxy - 1u); sz = (wz + om2) & (uzs - 1u); }
            else { sx = min(wx + om0, wx + om0 - uxy); sy = min(wy + om1, wy + om1 - uxy); sz = min(wz + om2, wz + om2 - uzs); }
            cmask = NOWIN ? alive : (lanes(inwin) & alive);               // gvom.py:1135-1144 (left the grid)
        }
        const uint32_t line = mad24s(mad24s(sy >> 2, uzs, sz), usxq, sx >> 2);   // accumulator line (acc_idx24)
        const uint32_t low4 = ((sy & 3u) << 2) | (sx & 3u);
        const uint32_t Ls = (line << 4) | low4;
        // merge runs of equal voxel indices among neighbouring lanes.  Lanes that do not take part hold
        // arbitrary indices: a run also starts where the left neighbour does not take part, and at lane 32
        // (its length is then found in the low word of a shifted mask)
        leftk = (uint32_t)__builtin_amdgcn_update_dpp((int)leftk, (int)Ls, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        const unsigned long long hm = (lanes(leftk != Ls) | ~(cmask << 1) | (1ull << 32)) & cmask;   // heads of runs
        // a run ends in front of the next head or of the next lane without a step -- or with the half-wave
        const unsigned long long ends = ((hm | ~cmask) >> 1) | (1ull << 63) | (1ull << 31);
        PROF_STAMP(1);
        if (__builtin_amdgcn_inverse_ballot_w64(hm) && !GVOM_DBG(P, 16)) {
            // memo: the (line, row-in-line) this lane added to last; a miss looks the line up
            // (or inserts it) and stamps the voxel's tile tag
            const uint32_t lrow = Ls >> 2;
            const bool miss = lrow != memo;
            // direct-mapped: 4 x 4 patches x 4 z levels around wherever the bundle is
            const uint32_t hh = ((sz & 3u) << 4) | (((sy >> 2) & 3u) << 2) | ((sx >> 2) & 3u);
            uint32_t was = LC_EMPTY;
            if (miss) {
                __hip_atomic_compare_exchange_strong(&keys3[LC_KEYPOS(hh)], &was, line, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!GVOM_DBG(P, 2)) tags[mad24s(mad24s(sy, uzs, sz), unseg, sx >> 6)] = epoch;
            }
            // (independent of the look-up: issued while the LDS compare-and-swap is in flight)
            const uint32_t run = 1u + (uint32_t)__builtin_ctz((uint32_t)(ends >> lane));   // lanes in my run (<= 32: see `ends`; never 0)
            // the slot is a function of the voxel (a memo hit means: same line, still in its slot -- slots are
            // only released by the flush); a miss whose slot holds another line adds directly
            const bool ok = (was == LC_EMPTY) | (was == line);
            memo = ok ? lrow : memo;
            // one LDS add for every head: a lane whose slot is taken adds into the spare entry behind the table (never
            // read) and makes its global add as well
            if (!GVOM_DBG(P, 128))
            __hip_atomic_fetch_add(&cnt3[ok ? hh * 16u + low4 : 1024u], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!ok) __hip_atomic_fetch_add(&total1[Ls], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // table congested: direct add
        }
        PROF_STAMP(2);
        alive = cmask & lanes(ju < cnt_run);                              // gvom.py:1127 (length test)
        PROF_STAMP(3);
#ifdef GVOM_DIAG
        if (CW.prof) ++CW.prof_n;
#endif
    } while (alive != 0ull);
#undef PROF_STAMP
    j = ju;
    px_ = pxy.x; py_ = pxy.y;
    active = ((cmask >> lane) & 1ull) != 0ull && ju < cnt;
    lc_flush(P, lck, lcc, total, lane);                                   // (a run in which no ray took a step finds an empty table)
}

// Issue priority of the wave by the work it still has in front of it (s_setprio: the SIMD's arbiter takes the ready wave of
// the highest priority, the oldest among equals).  Every step is a ~1000-cycle dependent chain of which ~220 are VALU
// issue slots: 4 to 5 walking waves saturate a SIMD, and with equal priorities the YOUNGEST waves of a SIMD get what
// the older ones leave -- next to nothing -- until those have finished, and then walk on alone, latency-bound, as the
// kernel's tail.  Longest-remaining-work-first lets the waves of a SIMD finish together.
__device__ __forceinline__ void prio_by_remaining(uint32_t rem, uint32_t div)
{
    const uint32_t q = rem / div;                      // (wave-uniform: scalar unit)
    if (q >= 3u) __builtin_amdgcn_s_setprio(3);
    else if (q == 2u) __builtin_amdgcn_s_setprio(2);
    else if (q == 1u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

// A wave's steps, in runs of at most `period` steps (the line cache is flushed after each run): per run the
// step loop takes the literal f64 lookup iff a coordinate of some active ray may come within 2^-21 of
// zero during the run (or the origin is beyond 2^30) -- a ray that crosses a coordinate plane costs one
// run of the slow form, not its whole segment; power-of-two grids wrap by masking.
template <bool BIG>
__device__ __forceinline__ void walk_item(const ScanParams &P, int lane, uint32_t j, uint32_t cnt, float px, float py, float pz,
                                          float incx, float incy, float incz, bool active, int steps, int period,
                                          uint32_t *lck, uint32_t *lcc, uint32_t *total, uint32_t *tags, uint32_t jend, const WalkConsts &C)
{
    const bool p2 = ((P.xy & (P.xy - 1)) | (P.zs & (P.zs - 1))) == 0;
    while (steps > 0 && lanes(active) != 0ull) {
        const int run = min(steps, period);
        steps -= run;
        if (P.prio_div > 0) {
            const uint32_t ju = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
            prio_by_remaining(jend > ju ? jend - ju : 0u, (uint32_t)P.prio_div);
        }
        // (the positions visited run monotonically from p + inc -- the first one, an exact f32 add as in
        // the loop -- to about p + run * inc; the margin is far above the rounding of that estimate, which
        // is below run * ulp(run) wherever the hull is near zero; NaN estimates compare false: such a
        // lane is inactive or leaves the grid at once)
        bool lit = BIG || P.zs > P.xy;                       // (the integer window test assumes z_size <= xy_size)
        bool nowin = false;
        if (!lit) {
            const float fs = (float)min((uint32_t)run, cnt - j);              // steps this ray can still take here (active lanes: cnt > j)
            const float ax = px + incx, ay = py + incy, az = pz + incz;
            const float qx = px + fs * incx, qy = py + fs * incy, qz = pz + fs * incz;
            const float lx = fminf(ax, qx), hx = fmaxf(ax, qx), ly = fminf(ay, qy), hy = fmaxf(ay, qy), lz = fminf(az, qz), hz = fmaxf(az, qz);
            // (the integer lookup differs from the literal one only for a position within 2^-21 BELOW zero: an axis whose first
            // position is >= 0 and whose increment is >= 0 -- exact statements, no estimate involved -- only visits positions
            // >= 0, e.g. every ray of a sensor that sits ON a coordinate plane and looks along it or away from it)
            const bool nz = (lx <= 1e-4f && hx >= -1e-4f && !(incx >= 0.0f && ax >= 0.0f)) ||
                            (ly <= 1e-4f && hy >= -1e-4f && !(incy >= 0.0f && ay >= 0.0f)) ||
                            (lz <= 1e-4f && hz >= -1e-4f && !(incz >= 0.0f && az >= 0.0f));
            lit = lanes(active & nz) != 0ull;
            // The run's positions lie between its first and its last one (straight line; the f32 accumulation strays from it
            // by less than run * ulp(|p|) <= 32 * 2^-6 voxels while |p| < 2^18, which win_lo / win_hi being set guarantees): a
            // bundle whose running rays keep 2 voxels from every face of the window takes the step body without window test.
            const bool safe = lx >= P.win_lo[0] && hx <= P.win_hi[0] && ly >= P.win_lo[1] && hy <= P.win_hi[1] && lz >= P.win_lo[2] && hz <= P.win_hi[2];
            nowin = p2 && !lit && lanes(active & !safe) == 0ull;
        }
        if (lit) {
            if (p2) walk_steps<true, true, false>(P, lane, j, cnt, px, py, pz, incx, incy, incz, active, run, lck, lcc, total, tags, C);
            else walk_steps<true, false, false>(P, lane, j, cnt, px, py, pz, incx, incy, incz, active, run, lck, lcc, total, tags, C);
        } else if (nowin) {
            walk_steps<false, true, true>(P, lane, j, cnt, px, py, pz, incx, incy, incz, active, run, lck, lcc, total, tags, C);
        } else {
            if (p2) walk_steps<false, true, false>(P, lane, j, cnt, px, py, pz, incx, incy, incz, active, run, lck, lcc, total, tags, C);
            else walk_steps<false, false, false>(P, lane, j, cnt, px, py, pz, incx, incy, incz, active, run, lck, lcc, total, tags, C);
        }
    }
}

// Diagnostic build only (GVOM_TRACE_TIMELINE): every wave of k_trace leaves {start, set-up done, end} times (100 MHz
// s_memrealtime ticks) and where it ran (HW_ID: wave / SIMD / CU / SE; XCC_ID) -- tools/trace_timeline.py turns them
// into the kernel's timeline: when each dispatch row starts, how full the chip is, where the tail is.
#ifdef GVOM_DIAG
#define TL_MARK(P, widx, k) do { if ((P).tl && (threadIdx.x & 63) == 0) (P).tl[(size_t)(widx) * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TL_WHERE(P, widx, pos) do { if ((P).tl && (threadIdx.x & 63) == 0) (P).tl[(size_t)(widx) * 4 + 3] = \
    (unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffu) | ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) << 32) | \
    ((unsigned long long)(pos) << 40); } while (0)
#else
#define TL_MARK(P, widx, k) do { } while (0)
#define TL_WHERE(P, widx, pos) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------
// k_trace.  Grid (ceil(N/512), nsegs + 1), 8 waves per workgroup: row P.ep_row holds the endpoint
// blocks (endpoint_update), every other row one STEP SEGMENT of the rays: a wave sets its 64 rays up
// (ray_setup, ray_steps), replays the steps of the earlier segments -- three f32 additions per step,
// the reference's exact accumulation, no lookup, no memory traffic -- and runs the step body for its
// own steps (seg_start[s], seg_start[s+1]]; the last segment is open-ended.  Coordinates are monotone,
// so "the ray has already ended before step k" is decided by the state AT step k alone, given that
// step 1 lies inside the grid, which every wave checks.
// ------------------------------------------------------------------------------------------
// One (dispatch row, 64-ray bundle) item of the trace: the endpoint work of the bundle (row == P.ep_row) or one step
// segment of its rays.  lck / lcc: the wave's line cache, clean on entry and on exit.
template <typename T, bool BIG>
__device__ __forceinline__ void trace_item(const ScanParams &P, const ShardExchange &X, const T *__restrict__ in, long stride, long n,
                                           T *__restrict__ world, uint32_t *hit, uint32_t *total, uint32_t *mh, int32_t *state,
                                           uint32_t *tags, uint32_t *counters, double *stat_sums, double *stat_base,
                                           uint32_t *stat_rowvox, int row, long bundle, int lane, uint32_t *lck, uint32_t *lcc,
                                           size_t widx, const WalkConsts &C)
{
    const long pos = bundle * 64 + lane;
    const bool live = pos < n;
    // sub-cloud interleave (ScanParams::ilv_lg): which return this lane takes
    // ... or, for a cloud in no spatial order, the return the directional order puts here (ScanParams::perm)
    const long i = P.perm ? (live ? (long)P.perm[pos] : pos) : (P.ilv_lg ? (pos & ((1L << P.ilv_lg) - 1)) * P.ilv_len + (pos >> P.ilv_lg) : pos);
    T x = 0, y = 0, z = 0;
    if (live) load_return(P, in, stride, i, x, y, z);
    const T d2 = (x * x + y * y) + z * z;
    const bool pass = live && !((double)d2 < P.min_d2);
    // endpoint work: in the items of row P.ep_row, or (P.ep_row < 0) in the waves of segment 0
    const bool ep_here = P.ep_row >= 0 ? row == P.ep_row : row == 0;
    if (ep_here) {
        if (live && world) { world[3 * i + 0] = x; world[3 * i + 1] = y; world[3 * i + 2] = z; }   // statistics only
        const Endpoint E = endpoint_of<T>(P, pass, x, y, z);
        // some return landed in the grid: the scan will be committed (gvom.py:147-150)
        if (lanes(E.ingrid) != 0ull && lane == 0) counters[GVOM_CNT_INGRID] = 1u;
        bool mine = E.ingrid;
        if (P.shard_world > 1) {
            // ranks of a sharded map: an endpoint in another rank's rows travels to its owner as
            // {voxel, min-height sample} (8 bytes); one counter atomic per wave and destination
            const int d = E.sy / P.shard_rows;
            const bool foreign = E.ingrid && d != P.shard_rank;
            mine = E.ingrid && !foreign;
            unsigned long long fm = lanes(foreign);
            while (fm != 0ull) {                                         // wave-uniform: the destinations present
                const int first = __ffsll((long long)fm) - 1;
                const int dd = __builtin_amdgcn_readlane(d, first);
                const unsigned long long m = lanes(foreign && d == dd);
                uint32_t base = 0;
                if (lane == first) base = atomicAdd(&X.ep_cnt[dd * 16], (uint32_t)__popcll(m));
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
                if (foreign && d == dd) X.ep_send[(size_t)dd * X.ep_cap + base + (uint32_t)__popcll(m & lanemask_lt())] = make_uint2(E.L, E.mbits);
                fm &= ~m;
            }
            if (X.sp_send) {
                // statistics: this return adds to every occupied voxel of its neighbourhood (gvom.py:1188-1220): the ranks
                // that own the first and the last in-window row of it get the return itself (the rank's own returns stay in
                // `world`); rows wrap with the storage, slabs are >= 2 e + 1 rows, so those two ranks are all there are
                int t0 = -1, t1 = -1;
                const double ay = floor((double)y / P.xy_res - P.origin[1]);
                if (pass && fabs(ay) < 1e9) {
                    const int yb = (int)ay, lo = max(yb - P.stat_e, 0), hi = min(yb + P.stat_e, P.xy - 1);
                    if (lo <= hi) {
                        t0 = wrap_add(lo, P.om[1], P.xy) / P.shard_rows;
                        t1 = wrap_add(hi, P.om[1], P.xy) / P.shard_rows;
                        if (t1 == t0) t1 = -1;
                    }
                }
#pragma unroll 1
                for (int pass_k = 0; pass_k < 2; ++pass_k) {
                    const int d = pass_k == 0 ? t0 : t1;
                    const bool go = d >= 0 && d != P.shard_rank;
                    unsigned long long gm = lanes(go);
                    while (gm != 0ull) {                                     // wave-uniform: the destinations present
                        const int first = __ffsll((long long)gm) - 1;
                        const int dd = __builtin_amdgcn_readlane(d, first);
                        const unsigned long long m = lanes(go && d == dd);
                        uint32_t base = 0;
                        if (lane == first) base = atomicAdd(&X.sp_cnt[dd * 16], (uint32_t)__popcll(m));
                        base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
                        if (go && d == dd) {
                            T *dst = (T *)X.sp_send + ((size_t)dd * X.ep_cap + base + (uint32_t)__popcll(m & lanemask_lt())) * 3;
                            dst[0] = x; dst[1] = y; dst[2] = z;
                        }
                        gm &= ~m;
                    }
                }
            }
        }
        endpoint_commit(P, lane, i, mine, E.L, E.A, E.mbits, hit, total, mh, state, tags, stat_sums, stat_base, stat_rowvox);
        if (P.ep_row >= 0) { TL_MARK(P, widx, 2); return; }
    }
    int seg = P.ep_row >= 0 ? row - (row > P.ep_row ? 1 : 0) : row;
    uint32_t j0 = 0;
#ifdef GVOM_DIAG
    // diagnostic build, GVOM_TRACE_DEBUG bits 8..11 = k: k EXTRA dispatch rows whose waves all die at the early-exit test below
    // (what a (row, bundle) wave that cannot walk costs: the launch slot, the load of its returns, the test)
    if (seg >= P.nsegs) { seg = P.nsegs; j0 = 0x3ffffff0u; }
    else
#endif
    j0 = (uint32_t)P.seg_start[seg];
    if (P.prio_div > 0) __builtin_amdgcn_s_setprio(3);   // set-up and replay: everything is still in front of this wave
    // ---- later segments: leave before the f64 set-up when no ray of the wave can still be running ----
    // After j0 steps `length` is >= j0 * (1 - 2^-22) (every step adds |1 / sd| with |sd| <= 1 + 2^-23),
    // and a ray stops once length >= ray_length - 1 (gvom.py:1127): a ray with ray_length <= j0 + 0.9
    // takes no step in this segment.  Decided conservatively in f32 from the raw return, with a
    // margin far above the rounding of this estimate; NaN/inf compare false and take the full path.
    if (seg > 0) {
        const float ax = (float)x * P.rinv[0], ay = (float)y * P.rinv[0], az = (float)z * P.rinv[1];
        const float ux = ax - P.pt0[0], uy = ay - P.pt0[1], uz = az - P.pt0[2];
        const float r = sqrtf((ux * ux + uy * uy) + uz * uz);
        const float mag = ((fabsf(ax) + fabsf(ay)) + fabsf(az)) + ((fabsf(ux) + fabsf(uy)) + fabsf(uz));
        const bool dead = !pass || (r + (r * 1e-5f + mag * 4e-6f) <= (float)j0 + 0.9f);
        if (lanes(!dead) == 0ull) { TL_MARK(P, widx, 2); return; }       // wave-uniform
    }
    const RaySetup R = ray_setup<T>(P, x, y, z);
    float px = P.pt0[0], py = P.pt0[1], pz = P.pt0[2];
    bool run = pass && R.finite;
    if (run) {                                           // step 1 outside the grid: no step at all
        uint32_t wx, wy, wz;
        run = window_voxel<true>(P, px + R.incx, py + R.incy, pz + R.incz, wx, wy, wz);
    }
    const uint32_t cnt = (run && !GVOM_DBG(P, 8)) ? ray_steps(R.lim, R.step_len, R.inv_step, 0x7ffffff0u) : 0u;     // steps the length test allows
    const bool active = j0 < cnt;
    if (lanes(active) == 0ull) { TL_MARK(P, widx, 2); return; }   // wave-uniform: every ray of the bundle ends earlier
    {   // replay (the reference's exact f32 accumulation; x and y as one packed add)
        v2f pxy = {px, py};
        const v2f incxy = {R.incx, R.incy};
        for (uint32_t k = j0; k > 0; --k) { pxy += incxy; pz += R.incz; }
        px = pxy.x; py = pxy.y;
    }
    const int steps = seg == P.nsegs - 1 ? 0x3fffffff : P.seg_start[seg + 1] - (int)j0;
    uint32_t jend = 0;                                   // last step any ray of the wave takes in this segment (priority only)
    if (P.prio_div > 0) {
        uint32_t m = active ? min(cnt, j0 + (uint32_t)min(steps, 1 << 20)) : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
        jend = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
    }
    TL_MARK(P, widx, 1);
    walk_item<BIG>(P, lane, j0, cnt, px, py, pz, R.incx, R.incy, R.incz, active, steps, P.lc_period, lck, lcc, total, tags, jend, C);
    TL_MARK(P, widx, 2);
}

// k_trace: grid (ceil(N/512), nsegs + 1), 8 waves per workgroup, one (row, bundle) item per wave.
// (Round 3 measured three other ways of handing out the items, each with per-wave timelines -- profiles/r3_timeline_*:
// a work queue drawn from one atomic counter (same-address atomics retire at ~90 per us: 256 us), workgroups whose
// waves take bundles from all over the cloud (half of every workgroup's waves die at once and its LDS keeps the slots
// from being reused: +11 %, c4 +28 %), and a dispatch order planned from the previous scan so that every CU gets 3 or 4
// walking workgroups instead of 1 to 5 (no gain: all walking waves are resident from t = 0 either way and the kernel
// runs at the VALU issue rate).)
template <typename T, bool BIG, int WPB>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_trace(
    const ScanParams P, const ShardExchange X, const T *__restrict__ in, long stride, long n, T *__restrict__ world,
    uint32_t *hit, uint32_t *total, uint32_t *mh, int32_t *state, uint32_t *tags, uint32_t *counters, double *stat_sums,
    double *stat_base, uint32_t *stat_rowvox)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int row = (int)blockIdx.y;
    const long bundle = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
    const size_t widx = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * WPB + (threadIdx.x >> 6);   // (timeline only)
    TL_MARK(P, widx, 0); TL_WHERE(P, widx, blockIdx.y * gridDim.x + blockIdx.x);
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[WPB * 64];
    __shared__ uint32_t s_cnt[WPB * 1040];                 // 64 entries x 16 counters + a spare word per wave (walk_steps)
    uint32_t *lck = s_keys + (threadIdx.x >> 6) * 64;
    uint32_t *lcc = s_cnt + (threadIdx.x >> 6) * 1040;
    if (row != P.ep_row) {                               // (endpoint blocks never touch the line cache)
        LC_ST(&lck[lane], LC_EMPTY);
#pragma unroll
        for (int q = 0; q < 16; ++q) LC_ST(&lcc[q * 64 + lane], 0u);
    }
    WalkConsts WC = walk_consts(P);
#ifdef GVOM_DIAG
    // (every 64th wave of the walking rows records the first 32 steps it takes: 4 stamps each, behind the per-wave records)
    WC.prof = (P.tl && P.prof_on && (widx & 63) == 0 && row != P.ep_row) ? P.tl + P.tl_words + 8 + (widx >> 6) * 128 : nullptr;
    WC.prof_n = 0;
#endif
    trace_item<T, BIG>(P, X, in, stride, n, world, hit, total, mh, state, tags, counters, stat_sums, stat_base, stat_rowvox,
                       row, bundle, lane, lck, lcc, widx, WC);
}

// ------------------------------------------------------------------------------------------
// Rank exchange of a sharded map (DESIGN.md "Multi-GPU").  Every rank traces ITS OWN rays over the
// whole window into private accumulators; the ray passes that fell into another rank's rows travel
// to their owner as dirty QUADS (4 storage rows x 64 sx at one sz = 16 accumulator lines = 1 KiB of
// `total`), the endpoints as {voxel, min-height} pairs (k_trace).  Integer sums and minima commute,
// so the owner's accumulators end up exactly as if it had traced every ray itself.
//
// k_pack: grid (ceil(slab quads / 64), world - 1): block (c, p) looks at 64 consecutive quads of
// peer p's rows; the dirty ones (a tile tag == this scan's epoch) are numbered with ONE counter
// atomic per block, copied to the peer's send region (quad id + 1 KiB in the lane order k_encode
// reads: lane (p4, r) = 4 voxels sx = 64*seg + 4*p4.. of row 4q + r) and zeroed.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack(const ScanParams P, uint32_t *total, const uint32_t *__restrict__ tags,
                                              uint32_t *send_ids, uint4 *send_pay, uint32_t *qcnt)
{
    __shared__ uint32_t s_list[GVOM_PACK_CHUNK];
    __shared__ uint32_t s_count, s_base;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    int d = (int)blockIdx.y;
    if (d >= P.shard_rank) ++d;                          // peers in rank order, skipping this rank
    const uint32_t nseg = (uint32_t)P.nseg, zs = (uint32_t)P.zs;
    const uint32_t u_begin = ((uint32_t)(d * P.shard_rows) >> 2) * zs * nseg;
    const uint32_t u_end = ((uint32_t)((d + 1) * P.shard_rows) >> 2) * zs * nseg;
    if (wv == 0) {                                       // wave 0: which of the block's quads are dirty (4 tile tags each)
        const uint32_t u = u_begin + blockIdx.x * GVOM_PACK_CHUNK + (uint32_t)lane;
        bool dirty = false;
        if (u < u_end) {
            const uint32_t seg = u % nseg, sz = (u / nseg) % zs, q = u / (nseg * zs);
#pragma unroll
            for (int r = 0; r < 4; ++r) dirty = dirty || tags[((q * 4 + r) * zs + sz) * nseg + seg] == P.epoch;
        }
        const unsigned long long dm = lanes(dirty);
        if (dirty) s_list[__popcll(dm & lanemask_lt())] = u;
        if (lane == 0) {
            const uint32_t count = (uint32_t)__popcll(dm);
            s_count = count;
            s_base = count ? atomicAdd(&qcnt[d * 16], count) : 0u;    // ONE counter atomic per block
        }
    }
    __syncthreads();
    const uint32_t count = s_count;
    if (count == 0) return;
    // the peer's regions start at its first quad: at most (u_end - u_begin) quads can be dirty
    uint32_t *ids = send_ids + u_begin;
    uint4 *pay = send_pay + (size_t)u_begin * 64;
    const int p4 = lane >> 2, r = lane & 3;
    for (uint32_t k = (uint32_t)wv; k < count; k += 4) {
        const uint32_t uq = s_list[k];
        const uint32_t seg = uq % nseg, sz = (uq / nseg) % zs, q = uq / (nseg * zs);
        const uint32_t sx0 = seg * 64 + p4 * 4, sy = q * 4 + r;
        const bool ok = sx0 < (uint32_t)P.xy;
        uint4 tv = make_uint4(0, 0, 0, 0);
        const uint32_t A0 = ok ? acc_idx((int)sx0, (int)sy, (int)sz, P.zs, P.sxq) : 0u;
        if (ok) tv = *reinterpret_cast<const uint4 *>(total + A0);
        pay[(size_t)(s_base + k) * 64 + lane] = tv;      // = the quad's 16 accumulator lines in memory order
        if (ok && (tv.x | tv.y | tv.z | tv.w)) *reinterpret_cast<uint4 *>(total + A0) = make_uint4(0, 0, 0, 0);
        if (lane == 0) ids[s_base + k] = uq;
    }
}

// counts of k_pack / k_trace's endpoint lists -> host-mapped memory (the host sizes the exchange with
// them) and re-armed: out[d] = quads for rank d, out[world + d] = endpoints, out[2*world] = some return
// of THIS rank landed in the grid, then the sequence number
__global__ void k_shard_publish(int world, uint32_t *qcnt, uint32_t *ecnt, uint32_t *spcnt, uint32_t *counters,
                                unsigned long long *host_out, uint32_t seq)
{
    const int d = threadIdx.x;
    if (d < world) {
        __hip_atomic_store(&host_out[d], (unsigned long long)qcnt[d * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_out[world + d], (unsigned long long)ecnt[d * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_out[2 * world + 2 + d], (unsigned long long)spcnt[d * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        qcnt[d * 16] = 0; ecnt[d * 16] = 0; spcnt[d * 16] = 0;
    }
    __syncthreads();
    if (d == 0) {
        __hip_atomic_store(&host_out[2 * world], (unsigned long long)(counters[GVOM_CNT_INGRID] ? 1u : 0u), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host_out[2 * world + 1], (unsigned long long)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// k_unpack_quads: one wave per received quad, all source ranks in ONE launch (X.q_off[s] = first wave of
// source s): total += the sender's 1 KiB.  A quad's 16 accumulator lines are contiguous, so lane l
// adds words l, l + 64, l + 128, l + 192: every instruction covers 4 whole lines (16 lanes per 64-B
// line = one memory-side request per line); tile tags of the rows that carry something.
__global__ __launch_bounds__(256) void k_unpack_quads(const ScanParams P, const ShardUnpack X, const uint32_t *__restrict__ ids_all,
                                                      const uint32_t *__restrict__ pay_all, uint32_t my_quads, uint32_t *total,
                                                      uint32_t *tags)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= X.q_off[P.shard_world]) return;
    int src = 0;
    while (w >= X.q_off[src + 1]) ++src;                 // wave-uniform, <= world steps
    const uint32_t k = w - X.q_off[src];
    const uint32_t nseg = (uint32_t)P.nseg, zs = (uint32_t)P.zs;
    const uint32_t uq = ids_all[(size_t)src * my_quads + k];
    const uint32_t *pay = pay_all + ((size_t)src * my_quads + k) * 256;
    const uint32_t seg = uq % nseg, sz = (uq / nseg) % zs, q = uq / (nseg * zs);
    if (seg * 64 >= (uint32_t)P.xy || q * 4 + 3 >= (uint32_t)P.xy) return;    // (a malformed id: never from k_pack)
    const uint32_t base = acc_idx((int)(seg * 64), (int)(q * 4), (int)sz, P.zs, P.sxq);
    uint32_t rows = 0;                                   // bit r: row 4q + r carries something
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t v = pay[i * 64 + lane];
        // word i*64 + lane = line 4i + (lane >> 4) (columns beyond xy hold zeros), row (lane >> 2) & 3
        if (v) atomicAdd(&total[base + (uint32_t)(i * 64 + lane)], v);
        const unsigned long long m = lanes(v != 0u);
#pragma unroll
        for (int r = 0; r < 4; ++r) if (m & (0x000F000F000F000Full << (4 * r))) rows |= 1u << r;
    }
    if (lane < 4 && ((rows >> lane) & 1u)) tags[((q * 4 + (uint32_t)lane) * zs + sz) * nseg + seg] = P.epoch;
}

// k_unpack_eps: one lane per received endpoint {voxel, min-height sample} (the receive regions are
// concatenated by source rank): the owner's share of k_trace's endpoint work; rows continue behind
// this rank's own returns
__global__ __launch_bounds__(256) void k_unpack_eps(const ScanParams P, uint32_t ne, const uint2 *__restrict__ eps, long row_base,
                                                    uint32_t *hit, uint32_t *total, uint32_t *mh, int32_t *state, uint32_t *tags,
                                                    double *stat_sums, double *stat_base, uint32_t *stat_rowvox)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < ne;
    uint32_t L = 0, A = 0, mbits = 0;
    if (live) {
        const uint2 e = eps[i];
        L = e.x; mbits = e.y;
        const uint32_t sx = L % (uint32_t)P.xy, rz = L / (uint32_t)P.xy;
        A = acc_idx((int)sx, (int)(rz / (uint32_t)P.zs), (int)(rz % (uint32_t)P.zs), P.zs, P.sxq);
    }
    endpoint_commit(P, lane, row_base + (long)i, live, L, A, mbits, hit, total, mh, state, tags, stat_sums, stat_base, stat_rowvox);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
hipError_t gvom_launch_trace(hipStream_t s, const ScanParams &P, const ShardExchange &X, int dtype, bool big_origin, const void *pts,
                             int64_t stride_elems, int64_t n, void *world, uint32_t *hit,
                             uint32_t *total, uint32_t *mh, int32_t *state, uint32_t *tags,
                             uint32_t *counters, double *stat_sums, double *stat_base,
                             uint32_t *stat_rowvox)
{
    if (n <= 0) return hipSuccess;
#define TRACE_LAUNCH(TT, BB, WW)                                                                             \
    hipLaunchKernelGGL((k_trace<TT, BB, WW>), dim3((unsigned)((n + 64 * WW - 1) / (64 * WW)), (unsigned)P.nsegs + (P.ep_row >= 0 ? 1u : 0u) + ((unsigned)GVOM_DBG(P, 0xF00) >> 8)), dim3(64 * WW), 0, s, P, X, (const TT *)pts, \
                       (long)stride_elems, (long)n, (TT *)world, hit, total, mh, state, tags, counters,     \
                       stat_sums, stat_base, stat_rowvox)
    // 8 waves per workgroup (measured on m256: 1 / 2 / 4 / 8 / 16 waves -> 47.4 / 44.6 / 41.7 / 40.5 / 42.8 us)
    if (dtype == 0) { if (big_origin) TRACE_LAUNCH(float, true, 8); else TRACE_LAUNCH(float, false, 8); }
    else { if (big_origin) TRACE_LAUNCH(double, true, 8); else TRACE_LAUNCH(double, false, 8); }
#undef TRACE_LAUNCH
    return hipGetLastError();
}

hipError_t gvom_launch_pack(hipStream_t s, const ScanParams &P, uint32_t *total, const uint32_t *tags, uint32_t *send_ids,
                            void *send_pay, uint32_t *qcnt, uint32_t *ecnt, uint32_t *spcnt, uint32_t *counters,
                            unsigned long long *host_out, uint32_t seq)
{
    const uint32_t slab_quads = ((uint32_t)P.shard_rows >> 2) * (uint32_t)P.zs * (uint32_t)P.nseg;
    if (P.shard_world > 1 && slab_quads > 0)
        hipLaunchKernelGGL(k_pack, dim3((slab_quads + GVOM_PACK_CHUNK - 1) / GVOM_PACK_CHUNK, (unsigned)P.shard_world - 1u),
                           dim3(256), 0, s, P, total, tags, send_ids, (uint4 *)send_pay, qcnt);
    hipLaunchKernelGGL(k_shard_publish, dim3(1), dim3(64), 0, s, P.shard_world, qcnt, ecnt, spcnt, counters, host_out, seq);
    return hipGetLastError();
}

hipError_t gvom_launch_unpack(hipStream_t s, const ScanParams &P, const ShardUnpack &X, const uint32_t *ids_all,
                              const void *pay_all, uint32_t my_quads, uint32_t ne, const void *eps, long row_base,
                              uint32_t *hit, uint32_t *total, uint32_t *mh, int32_t *state, uint32_t *tags,
                              double *stat_sums, double *stat_base, uint32_t *stat_rowvox)
{
    const uint32_t nq = X.q_off[P.shard_world];
    if (nq) hipLaunchKernelGGL(k_unpack_quads, dim3((nq + 3) / 4), dim3(256), 0, s, P, X, ids_all, (const uint32_t *)pay_all,
                               my_quads, total, tags);
    if (ne) hipLaunchKernelGGL(k_unpack_eps, dim3((ne + 255) / 256), dim3(256), 0, s, P, ne, (const uint2 *)eps, row_base, hit,
                               total, mh, state, tags, stat_sums, stat_base, stat_rowvox);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Directional order of a cloud whose OWN order does not suit the trace.  k_trace's cost follows the accumulator lines (4 x 4 voxels
// at one level) a bundle of 64 consecutive returns touches per step: a rotating lidar's beam-major rows (64 azimuths of one beam: a
// horizontal fan) are ideal; 64 random returns (BASELINE config c1; any shuffled, merged, filtered cloud) touch 64 lines -- c1:
// k_trace 65 us as given, 15.6 us with the same returns ordered by direction (tools/c1_sort_probe.py) -- and an azimuth-major
// ("firing order") organised cloud makes every bundle a VERTICAL fan: 280 us for the scan the beam-major order traces in 37
// (tools/bundle_shape_probe.py).  A counting sort by direction bin seen from the sensor, two launches in front of the trace:
//   mode 1 (no spatial order)   6 cube faces x 16 x 16 cells in Morton order: 1536 bins of ~5.6 degrees
//   mode 2 (vertical fans)      256 rows of sin(elevation) x 32 azimuth sectors, row-major: inside a cell the returns keep (roughly)
//                               the order they came in, which for a firing-order cloud is increasing azimuth -- the beam-major
//                               fans come back (tools/sphere_sort_probe.py: 37-43 us)
//   k_dirbin_hist     key of every return (float arithmetic: the order need not be exact) + per-block LDS histogram -> global
//   k_dirbin_scatter  every block scans the counts itself (no scan launch), reserves its share of each bin with one global atomic
//                     per non-empty bin and block, and writes perm[position] = return
// k_trace's results do not depend on who traces which return.
// (DIRBIN_ITEMS returns per thread, 256 threads per block: 1 for clouds up to 131 k returns -- both kernels are latency chains -- 2
// up to 524 k, 8 above)
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ uint32_t dirbin_key(const ScanParams &P, int mode, const T *__restrict__ in, long stride, long i)
{
    T x, y, z;
    load_return(P, in, stride, i, x, y, z);
    const float ux = (float)x * P.rinv[0] - P.pt0[0], uy = (float)y * P.rinv[0] - P.pt0[1], uz = (float)z * P.rinv[1] - P.pt0[2];
    if (mode == 2) {
        const float r = sqrtf((ux * ux + uy * uy) + uz * uz);
        if (!(r > 0.0f) || !(r < INFINITY)) return 0u;     // the sensor's own position, NaN, inf: anywhere
        const uint32_t row = (uint32_t)min(255, max(0, (int)((uz / r + 1.0f) * 128.0f)));
        const uint32_t sec = (uint32_t)min(31, max(0, (int)((atan2f(uy, ux) + 3.14159265f) * (32.0f / 6.2831853f))));
        return row * 32u + sec;
    }
    const float ax = fabsf(ux), ay = fabsf(uy), az = fabsf(uz);
    uint32_t face;
    float a, u, v;
    if (ax >= ay && ax >= az) { face = ux < 0.0f ? 1u : 0u; a = ax; u = uy; v = uz; }
    else if (ay >= az) { face = uy < 0.0f ? 3u : 2u; a = ay; u = uz; v = ux; }
    else { face = uz < 0.0f ? 5u : 4u; a = az; u = ux; v = uy; }
    if (!(a > 0.0f) || !(a < INFINITY)) return 0u;
    const uint32_t qu = (uint32_t)min(15, max(0, (int)((u / a + 1.0f) * 8.0f)));
    const uint32_t qv = (uint32_t)min(15, max(0, (int)((v / a + 1.0f) * 8.0f)));
    // Morton order of the face's 16 x 16 cells: bins that follow each other point in neighbouring directions
    auto part = [](uint32_t t) { t = (t | (t << 2)) & 0x33u; return (t | (t << 1)) & 0x55u; };
    return face * 256u + (part(qu) | (part(qv) << 1));
}
template <typename T, int DIRBIN_ITEMS, int NBINS>
__global__ __launch_bounds__(256) void k_dirbin_hist(const ScanParams P, int mode, const T *__restrict__ in, long stride, long n, uint16_t *keys,
                                                     uint32_t *hist, uint32_t *cursor)
{
    __shared__ uint32_t s_h[NBINS];
    for (int b = threadIdx.x; b < NBINS; b += 256) s_h[b] = 0u;
    if (blockIdx.x == 0) for (int b = threadIdx.x; b < NBINS; b += 256) cursor[b] = 0u;      // (nobody reads it before k_dirbin_scatter)
    __syncthreads();
    const long base = (long)blockIdx.x * (256 * DIRBIN_ITEMS);
#pragma unroll
    for (int k = 0; k < DIRBIN_ITEMS; ++k) {
        const long i = base + k * 256 + threadIdx.x;
        if (i < n) {
            const uint32_t key = dirbin_key(P, mode, in, stride, i);
            keys[i] = (uint16_t)key;
            atomicAdd(&s_h[key], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < NBINS; b += 256) { const uint32_t c = s_h[b]; if (c) atomicAdd(&hist[b], c); }
}
template <int DIRBIN_ITEMS, int NBINS>
__global__ __launch_bounds__(256) void k_dirbin_scatter(long n, const uint16_t *__restrict__ keys, const uint32_t *__restrict__ hist,
                                                        uint32_t *hist_next, uint32_t *cursor, uint32_t *perm)
{
    constexpr int BPT = NBINS / 256;               // consecutive bins per thread of the prefix
    __shared__ uint32_t s_start[NBINS];             // first position of every bin (exclusive prefix of the counts), then this block's share
    __shared__ uint32_t s_cnt[NBINS];               // this block's returns per bin
    __shared__ uint32_t s_part[256];
    uint32_t acc = 0;
    for (int k = 0; k < BPT; ++k) { const uint32_t c = hist[threadIdx.x * BPT + k]; s_start[threadIdx.x * BPT + k] = c; acc += c; s_cnt[threadIdx.x * BPT + k] = 0u; }
    // the other histogram, for the next cloud: ALL of it (the next cloud may be sorted in the other mode, with more bins)
    if (blockIdx.x == 0) for (int k = threadIdx.x; k < GVOM_DIRBINS; k += 256) hist_next[k] = 0u;
    // inclusive scan of the 256 partial sums: inside each wave by shuffles, the four waves' totals through LDS (one barrier)
    uint32_t inc = acc;
    const int ln = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)inc, o); if (ln >= o) inc += v; }
    if (ln == 63) s_part[threadIdx.x >> 6] = inc;
    __syncthreads();
    for (int wv = 0; wv < (int)(threadIdx.x >> 6); ++wv) inc += s_part[wv];
    uint32_t run = inc - acc;                               // exclusive
    for (int k = 0; k < BPT; ++k) { const uint32_t c = s_start[threadIdx.x * BPT + k]; s_start[threadIdx.x * BPT + k] = run; run += c; }
    __syncthreads();
    const long base = (long)blockIdx.x * (256 * DIRBIN_ITEMS);
    uint32_t key[DIRBIN_ITEMS], rank[DIRBIN_ITEMS];
#pragma unroll
    for (int k = 0; k < DIRBIN_ITEMS; ++k) {
        const long i = base + k * 256 + threadIdx.x;
        key[k] = i < n ? (uint32_t)keys[i] : 0xffffu;
        rank[k] = key[k] != 0xffffu ? atomicAdd(&s_cnt[key[k]], 1u) : 0u;                          // my place among the block's returns of the bin
    }
    __syncthreads();
    // one global atomic per non-empty bin of the block reserves its share of the bin
    for (int b = threadIdx.x; b < NBINS; b += 256) {
        const uint32_t c = s_cnt[b];
        if (c) s_start[b] += atomicAdd(&cursor[b], c);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DIRBIN_ITEMS; ++k) {
        const long i = base + k * 256 + threadIdx.x;
        if (key[k] != 0xffffu) perm[s_start[key[k]] + rank[k]] = (uint32_t)i;
    }
}
hipError_t gvom_launch_dirbin(hipStream_t s, const ScanParams &P, int mode, int dtype, const void *pts, int64_t stride_elems, int64_t n,
                              uint16_t *keys, uint32_t *hist, uint32_t *hist_next, uint32_t *cursor, uint32_t *perm)
{
    if (n <= 0) return hipSuccess;
#define DIRBIN_LAUNCH(IT, NB)                                                                                                  \
    do {                                                                                                                       \
        const unsigned blocks = (unsigned)((n + 256 * IT - 1) / (256 * IT));                                                   \
        if (dtype == 0) hipLaunchKernelGGL((k_dirbin_hist<float, IT, NB>), dim3(blocks), dim3(256), 0, s, P, mode, (const float *)pts, (long)stride_elems, (long)n, keys, hist, cursor); \
        else hipLaunchKernelGGL((k_dirbin_hist<double, IT, NB>), dim3(blocks), dim3(256), 0, s, P, mode, (const double *)pts, (long)stride_elems, (long)n, keys, hist, cursor);          \
        hipLaunchKernelGGL((k_dirbin_scatter<IT, NB>), dim3(blocks), dim3(256), 0, s, (long)n, keys, hist, hist_next, cursor, perm);  \
    } while (0)
    // (mode 2's 8192 bins cost every block a scan of their own, and the order inside a cell is kept per block: 131,072 returns in
    // firing order, 8 / 2 / 1 returns per thread: sort + trace 72.8 / 69.9 / 80.1 us)
    if (mode == 2) { if (n <= 524288) DIRBIN_LAUNCH(2, 8192); else DIRBIN_LAUNCH(8, 8192); }
    else if (n <= 131072) DIRBIN_LAUNCH(1, 1536); else if (n <= 524288) DIRBIN_LAUNCH(2, 1536); else DIRBIN_LAUNCH(8, 1536);
#undef DIRBIN_LAUNCH
    return hipGetLastError();
}

// one store of `seq` into host-mapped memory: launched behind the last kernel of a call, it tells the
// spinning host that everything before it on the stream has completed (lower latency than an event wait)
// Layout probe (ONE wave, a launch of its own in front of k_trace -- on the first cloud of a new length and every 32nd scan
// after it): is this cloud K equally long sub-clouds behind one another -- K sensors at one place, K sweeps -- whose returns of
// equal position point in neighbouring directions?  Then the next scans of as many returns are traced with their sub-clouds
// interleaved (ScanParams::ilv_lg).  64 samples per candidate K, spread over sub-cloud 0: the return at the same position of
// the next and of the last sub-cloud must lie closer in direction (seen from the sensor) than the return's own successor in
// the cloud; K passes with 56 of 64.  The answer {n, log2 K} goes to host-mapped memory as one 8-byte store.  A heuristic that
// decides WHO traces which return, never what is added where: any answer gives the same maps.
template <typename T>
__global__ __launch_bounds__(64) void k_layout_probe(const ScanParams P, const T *__restrict__ in, long stride, long n, int max_lg,
                                                     unsigned long long *host_word)
{
    const int lane = threadIdx.x;
    int best = 0;
    // Does the cloud's order suit the trace?  k_trace's cost follows the accumulator lines (4 x 4 voxels at ONE level) a bundle of 64
    // consecutive returns touches per step.  64 samples say "scattered" (bit 3 of the answer: the next clouds of this length are
    // traced in directional order, k_dirbin_*) when in most of them
    //   * a return and its successor point more than ~6 degrees apart (no spatial order at all: BASELINE c1's random points), or
    //   * a return and the one 63 places behind it differ by more than ~3 degrees in ELEVATION: a bundle is a vertical fan -- an
    //     azimuth-major ("firing order") organised cloud, every beam of one azimuth behind one another: 280 us for the cloud the
    //     beam-major order traces in 37 (tools/bundle_shape_probe.py); a rotating lidar's beam-major rows have neither property.
    int scattered = 0;
    if (n >= 8192) {
        const long q = (n / 64) * lane + n / 128;            // q + 63 < n
        float d0[3] = {0.0f, 0.0f, 0.0f}, e1 = 0.0f, ez = 0.0f;
        bool good = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            T x, y, z;
            load_return(P, in, stride, q + (k == 0 ? 0 : (k == 1 ? 1 : 63)), x, y, z);
            const float ux = (float)x * P.rinv[0] - P.pt0[0], uy = (float)y * P.rinv[0] - P.pt0[1], uz = (float)z * P.rinv[1] - P.pt0[2];
            const float r = sqrtf((ux * ux + uy * uy) + uz * uz);
            good = good && r > 0.0f && r < INFINITY;
            const float dx = ux / r, dy = uy / r, dz = uz / r;
            if (k == 0) { d0[0] = dx; d0[1] = dy; d0[2] = dz; }
            else if (k == 1) { const float a = dx - d0[0], b = dy - d0[1], c = dz - d0[2]; e1 = (a * a + b * b) + c * c; }
            else ez = fabsf(dz - d0[2]);
        }
        if (__popcll(lanes(good && e1 > 0.01f)) >= 48) scattered = 1;            // no spatial order: cube cells (bit 3)
        else if (__popcll(lanes(good && ez > 0.05f)) >= 48) scattered = 2;       // vertical fans: elevation rows (bit 4)
    }
    for (int lg = 1; lg <= max_lg; ++lg) {
        const long K = 1L << lg;
        if (n % K != 0 || n / K < 4096) break;
        const long M = n / K;
        const long q = (M / 64) * lane + M / 128;                 // q + 1 < M
        float e[3], d0[3];
        bool good = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long idx = k == 0 ? q : (k == 1 ? q + 1 : (k == 2 ? q + M : q + (K - 1) * M));
            T x, y, z;
            load_return(P, in, stride, idx, x, y, z);
            const float ux = (float)x * P.rinv[0] - P.pt0[0], uy = (float)y * P.rinv[0] - P.pt0[1], uz = (float)z * P.rinv[1] - P.pt0[2];
            const float r = sqrtf((ux * ux + uy * uy) + uz * uz);
            good = good && r > 0.0f && r < INFINITY;
            const float dx = ux / r, dy = uy / r, dz = uz / r;
            if (k == 0) { d0[0] = dx; d0[1] = dy; d0[2] = dz; }
            else { const float a = dx - d0[0], b = dy - d0[1], c = dz - d0[2]; e[k - 1] = (a * a + b * b) + c * c; }
        }
        const bool pass = good && e[1] <= e[0] && e[2] <= e[0];       // (NaN compares false)
        if (__popcll(lanes(pass)) >= 56) best = lg;
    }
    if (lane == 0)
        __hip_atomic_store(host_word, ((unsigned long long)n << 8) | (unsigned long long)(best | (scattered << 3)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t gvom_launch_layout_probe(hipStream_t s, const ScanParams &P, int dtype, const void *pts, int64_t stride_elems, int64_t n,
                                    int max_lg, unsigned long long *host_word)
{
    if (dtype == 0) hipLaunchKernelGGL(k_layout_probe<float>, dim3(1), dim3(64), 0, s, P, (const float *)pts, (long)stride_elems, (long)n, max_lg, host_word);
    else hipLaunchKernelGGL(k_layout_probe<double>, dim3(1), dim3(64), 0, s, P, (const double *)pts, (long)stride_elems, (long)n, max_lg, host_word);
    return hipGetLastError();
}
