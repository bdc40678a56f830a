// gvom_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the G-VOM hot path.
//
// Four kernels replace the reference's ~55 Numba-CUDA launches per scan+combine
// (reference: /root/reference/scripts/gvom.py, cited as "gvom.py:NNN"):
//
//   k_trace   gvom.py:1040-1056 (transform) + :1060-1150 (hit + dominant-axis DDA)
//             + the row-claim half of :1154-1160 (first hit of a voxel claims its compact row)
//             + :1303-1329 (min-height: a third accumulator, atomicMax of 1.0f's bits minus the sample's)
//   k_encode  gvom.py:1154-1168 (state code + dense->compact move) + the three V-sized clears
//             of :114-121 (accumulators are cleared as they are read; no separate fill), only on
//             the 64-voxel tiles the scan touched
//   k_fuse4   gvom.py:943-968 x slots, :972-997, :821-912 (count/min lines 910-912) x (slots+1),
//   (k_fuse1  :525-540 (height) and :544-554 (inferred height): ONE pass over the fused grid
//    k_fuse)  (k_fuse1: one ring slot; k_fuse: grids with xy % 4 != 0 or chunks other than 16 levels)
//   k_map2d   gvom.py:665-734 (slope/roughness), :558-661 (guess height), :489-521 (positive),
//             :479-485 (negative), :414-422 (visibility)
//
// Numerics are the reference's as executed by the Numba simulator:
// compile with -ffp-contract=off, IEEE division/sqrt, no fast-math.  Integer results are
// bit-exact; only log()/atan2() may differ from glibc in the last ulp.
//
// No MFMA: there is no dense contraction on this path.  The work is scattered 4-byte
// atomics (k_trace) and streaming passes over V = xy*xy*zs voxels (k_encode, k_fuse).
#include "gvom_internal.h"
#include <limits.h>

// Written for ONE target: 64-wide waves, 160 KB of LDS per workgroup (k_dirbin_scatter<., 8192> alone declares 65 KB of static
// LDS), v_cvt_flr_i32_f32, DPP wave shifts, the memory side's merging of same-line atomics.  Any other --offload-arch is a
// build error here, not a launch failure later.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libgvom_hip.so is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

#define WAVE 64

// Pointers read out of a descriptor table are generic ("flat") to the compiler; these casts tell
// it they point to global memory so it emits global_load (vmcnt only) instead of flat_load.
typedef const __attribute__((address_space(1))) int32_t *gptr_i32;
typedef const __attribute__((address_space(1))) uint32_t *gptr_u32;
typedef const __attribute__((address_space(1))) uint16_t *gptr_u16;
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) v2u *gptr_u2;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4u *gptr_v4u;
// The fusion kernels index their source descriptors with wave-uniform values.  Through a pointer that
// may be kernel-argument or global memory (one generic pointer) every field read was a FLAT vector
// load + readfirstlane and a round trip of its own ahead of the load it feeds; as constant-address-space
// reads they are scalar loads.  MEM (template) = the descriptors did not fit the kernel arguments.
typedef const __attribute__((address_space(4))) MapDesc *cptr_desc;
typedef int v4i __attribute__((ext_vector_type(4)));                       // 16-byte vector of 4 ints
typedef const __attribute__((address_space(1))) v4i *gptr_v4i;

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));                      // packed f32 pair (v_pk_add_f32)
__device__ __forceinline__ uint32_t pk_add_sat_u16(uint32_t a, uint32_t b) {     // v_pk_add_u16 ... clamp
    const us2 r = __builtin_elementwise_add_sat(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b));
    return __builtin_bit_cast(uint32_t, r);
}
// Free (ray-pass) counts live in the state as -count - 1 and the fused map carries its predecessor's along
// (gvom.py:996): in a voxel every ray passes -- the sensor's own -- the sum reaches 2^31 after ~1000 combines of a
// 262 k-point, 8-slot ring, and an int32 that wraps turns into a non-negative value, which every reader takes for
// a ROW INDEX (the reference's int32 wraps the same way and then indexes out of bounds).  Here the count stops at
// 2^30: it stays a free count for ever; results are the reference's wherever it has not overflowed itself.
#define GVOM_FREE_FLOOR (-(1 << 30))
__device__ __forceinline__ int add_free(int c, int st_plus_1) { return max(c + max(st_plus_1, GVOM_FREE_FLOOR), GVOM_FREE_FLOOR); }
__device__ __forceinline__ int wrap_add(int a, int b, int n) { int s = a + b; return s >= n ? s - n : s; }
__device__ __forceinline__ int wrap_sub(int a, int b, int n) { int s = a - b; return s < 0 ? s + n : s; }
// accumulator (hit/total) index of storage voxel (sx, sy, sz): 4x4 (x,y) patches per 64-B line
__device__ __forceinline__ uint32_t acc_idx(int sx, int sy, int sz, int zs, int sxq) {
    return (((((uint32_t)sy >> 2) * zs + sz) * sxq + ((uint32_t)sx >> 2)) << 4) + (((uint32_t)sy & 3u) << 2) + ((uint32_t)sx & 3u);
}
// Python's max(a, b): a unless b > a  (gvom.py:1116; differs from fmaxf only for NaN)
__device__ __forceinline__ float py_maxf(float a, float b) { return (b > a) ? b : a; }
__device__ __forceinline__ double py_maxd(double a, double b) { return (b > a) ? b : a; }
__device__ __forceinline__ double py_mind(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << (threadIdx.x & 63)) - 1ull;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

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

// floor(x) as int32 in ONE instruction (v_cvt_flr_i32_f32: round toward -inf, saturating, NaN -> 0):
// identical to (int)floorf(x) wherever that is defined
__device__ __forceinline__ int cvt_floor_i32(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
// a * b + c for a, b < 2^24 (full-rate v_mad_u32_u24; the 32-bit multiply is quarter rate)
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// the same with a wave-uniform multiplier (kept in an SGPR: no v_mov per use)
__device__ __forceinline__ uint32_t mad24s(uint32_t a, uint32_t sb, uint32_t c)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(sb), "v"(c));
    return r;
}
// lane mask of a predicate without the bool -> int -> compare round trip of __ballot / __any
__device__ __forceinline__ unsigned long long lanes(bool p) { return __builtin_amdgcn_ballot_w64(p); }
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(1))) uint32_t glb_u32;
// accumulator index as acc_idx(), with 24-bit multiplies (zs <= 1024, sxq < 2^12, sy >> 2 < 2^12)
__device__ __forceinline__ uint32_t acc_idx24(uint32_t sx, uint32_t sy, uint32_t sz, uint32_t zs, uint32_t sxq)
{
    const uint32_t t = mad24s(sy >> 2, zs, sz);
    return (mad24s(t, sxq, sx >> 2) << 4) | ((sy & 3u) << 2) | (sx & 3u);
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
// k_encode: visits only the tiles k_trace stamped with this scan's epoch.  Per voxel of a dirty tile:
//   occupied (hit > 0): its row is the index of one of its returns (k_trace's endpoint blocks left it
//                       in state[]) -> move hit / total / min-height to the compact arrays
//                                                                      gvom.py:1164-1168,1303-1329
//   else              : state = -total - 1                             gvom.py:1160
//   and the accumulators are zeroed for the next scan (replaces the fills of gvom.py:114-121).
// The same information goes out a second time as a 16-bit code per voxel (code16, xy % 4 == 0 grids):
// min(passes, 65535) of a free voxel (0: never observed), 65535 if occupied -- what k_fuse4 reads of
// a ring slot: the slots' codes are summed with saturating packed adds (2 voxels per instruction, no
// compare), and a sum of 65535 = "occupied in some slot, or more than 65534 passes" sends the voxel
// to the per-voxel path that reads the 32-bit states.
// Untouched tiles are neither read nor written: their tag != epoch makes every consumer treat
// them as "never observed" (-1), which is what the reference's -1 fill + __assign_indices yield.
// Min-height arrives as a third dense accumulator (1.0f's bits minus the value's bits, atomicMax:
// zero between scans like the other two), read only where a voxel is occupied.
// ------------------------------------------------------------------------------------------
template <int ENC_T>
__global__ __launch_bounds__(ENC_T) void k_encode(const ScanParams P, uint32_t t_begin, uint32_t t_end,
                                                uint32_t *hit, uint32_t *total, uint32_t *mh, int32_t *state,
                                                uint16_t *code16, uint4 *crows,
                                                const uint32_t *__restrict__ tags, uint32_t epoch,
                                                uint32_t *counters, unsigned long long *host_flag,
                                                uint32_t seq)
{
    const int xy = P.xy, nseg = P.nseg;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // k_trace has completed.  Publish {seq, any-in-grid} as ONE 8-byte system-scope store to
        // host-mapped memory (the host spins on it and returns to its caller while this kernel
        // still runs) and re-arm the flag.
        const uint32_t any = counters[GVOM_CNT_INGRID] ? 0x80000000u : 0u;     // some return landed in the grid
        counters[GVOM_CNT_INGRID] = 0;
        __hip_atomic_store(host_flag, ((unsigned long long)seq << 32) | any, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // one wave per QUAD = 4 storage rows (sy = 4q .. 4q+3) x 64 sx at one sz,
    // i.e. 4 tiles = 16 accumulator lines.  Lane (p = lane >> 2, r = lane & 3) owns the 4 voxels
    // sx = 64*seg + 4p .. +3 of row sy = 4q + r: one 16-byte load of hit and of total (its quarter of
    // a 4x4 patch line) and one 16-byte store of state.
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t wid = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t nw = (gridDim.x * blockDim.x) >> 6;
    const int p4 = lane >> 2, r = lane & 3;
    const bool vec_state = (xy & 3) == 0;                // 16-byte aligned state rows
    for (uint32_t u0 = t_begin + wid * 2; u0 < t_end; u0 += nw * 2) {
        // two quads per iteration; lane j (< 8) fetches the tag of row (j & 3) of quad (j >> 2)
        uint32_t dmask;
        {
            const uint32_t u = u0 + ((lane >> 2) & 1);
            uint32_t seg, sz, q;
            if (P.lg_nseg >= 0) { seg = u & (nseg - 1); sz = (u >> P.lg_nseg) & (P.zs - 1); q = u >> (P.lg_nseg + P.lg_zs); }   // (power-of-two grids: no division)
            else { seg = u % nseg; sz = (u / nseg) % P.zs; q = u / (nseg * P.zs); }
            const uint32_t syl = q * 4 + (lane & 3);
            const bool ok = lane < 8 && u < t_end && syl < (uint32_t)xy && (int)syl >= P.sy_lo && (int)syl < P.sy_hi;
            const uint32_t tagv = tags[ok ? (syl * P.zs + sz) * nseg + seg : 0];
            dmask = GVOM_DBG(P, 64) ? 0u : (uint32_t)__ballot(ok && tagv == epoch);
        }
        if (dmask == 0) continue;                                    // wave-uniform
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (((dmask >> (4 * j)) & 0xfu) == 0) continue;          // wave-uniform
            const uint32_t u = u0 + j;
            uint32_t seg, sz, q;
            if (P.lg_nseg >= 0) { seg = u & (nseg - 1); sz = (u >> P.lg_nseg) & (P.zs - 1); q = u >> (P.lg_nseg + P.lg_zs); }
            else { seg = u % nseg; sz = (u / nseg) % P.zs; q = u / (nseg * P.zs); }
            const uint32_t sy = q * 4 + r, sx0 = seg * 64 + p4 * 4;
            const bool dirty = ((dmask >> (4 * j + r)) & 1u) && sx0 < (uint32_t)xy;
            const uint32_t A0 = dirty ? acc_idx((int)sx0, (int)sy, (int)sz, P.zs, P.sxq) : (uint32_t)(lane * 4);
            const uint32_t L0 = dirty ? (sy * P.zs + sz) * xy + sx0 : 0u;
            const uint4 hv = *reinterpret_cast<const uint4 *>(hit + A0);
            const uint4 tv = *reinterpret_cast<const uint4 *>(total + A0);
            if (!dirty) continue;
            // total = ray passes (k_trace's steps) + the endpoints' own count, which k_trace leaves in `hit` alone (endpoint_commit)
            const uint32_t h[4] = {hv.x, hv.y, hv.z, hv.w}, t[4] = {tv.x + hv.x, tv.y + hv.y, tv.z + hv.z, tv.w + hv.w};
            int32_t st[4];
            const uint32_t any_h = h[0] | h[1] | h[2] | h[3], any_t = tv.x | tv.y | tv.z | tv.w;
#pragma unroll
            for (int i = 0; i < 4; ++i) st[i] = -(int32_t)t[i] - 1;
            if (any_h) {                                 // rare: an occupied voxel; its row was left in state[] by k_trace
                int32_t rows[4];                         // all four fetched before the first use (one round trip)
#pragma unroll
                for (int i = 0; i < 4; ++i) rows[i] = state[L0 + ((h[i] > 0 && sx0 + i < (uint32_t)xy) ? (uint32_t)i : 0u)];
                const uint4 mv = *reinterpret_cast<const uint4 *>(mh + A0);
                const uint32_t m[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (h[i] > 0 && sx0 + i < (uint32_t)xy) {
                        crows[rows[i]] = make_uint4(h[i], t[i], 0x3f800000u - m[i], 0u);   // min-height (gvom.py:1014-1015, 1329)
                        st[i] = rows[i];
                    }
                }
                *reinterpret_cast<uint4 *>(hit + A0) = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4 *>(mh + A0) = make_uint4(0, 0, 0, 0);
            }
            if (any_t) *reinterpret_cast<uint4 *>(total + A0) = make_uint4(0, 0, 0, 0);
            if (vec_state) {
                *reinterpret_cast<int4 *>(state + L0) = make_int4(st[0], st[1], st[2], st[3]);
                uint32_t cd[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) cd[i] = st[i] >= 0 ? 0xffffu : min(t[i], 0xffffu);
                if (code16) *reinterpret_cast<uint2 *>(code16 + L0) = make_uint2(cd[0] | (cd[1] << 16), cd[2] | (cd[3] << 16));   // (nullptr: no k_fuse4 will read this slot)
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (sx0 + i < (uint32_t)xy) state[L0 + i] = st[i];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_fuse: temporal fusion + per-column height reductions in ONE pass over the fused grid.
//
// Workgroup = 64 storage columns (consecutive sx of one storage row sy) x all z, as nz waves
// that each own a chunk of zc consecutive WINDOW z levels.  Storage is world-anchored, so
// every source map holds the voxel at the same linear index L and only a window test
// (is this world voxel inside the source's window?) replaces the reference's shifted gather.
//
//  * per voxel: fold the ring slots in slot order, then the previous fused map
//    (gvom.py:963-968, 992-997) into "occupied" or a free/unknown code.  For zc <= 16 the 16
//    state words of a source are loaded back to back (predicated, independent) before folding.
//  * compact rows: every WAVE owns a static row range [wave_id * 64*zc, ...) of the fused compact
//    arrays and numbers its occupied voxels inside it by ballot + prefix popcount.  Rows are
//    therefore not globally dense (the reference's are, gvom.py:964,993 -- value-neutral: every
//    consumer goes through the state map) and NO global atomic or row-reservation barrier
//    exists; the occupied-voxel count is the sum of per-workgroup counts (blockcounts[]).
//  * occupied voxel: hit/total sums and min-height min over every source where it is occupied
//    (gvom.py:910-912).
//  * column tail: lowest occupied z (+ its min-height) and lowest observed-free z per column
//    are combined across the nz waves through LDS (the kernel's only barrier)
//    -> height_map (gvom.py:525-540) / inferred_height_map (gvom.py:544-554).
// ------------------------------------------------------------------------------------------
template <bool ZC16, bool MEM>
__global__ __launch_bounds__(1024) void k_fuse(const FuseParams P, const FuseDescs KD,
                                               const MapDesc *__restrict__ descs_mem,
                                               int32_t *fstate, uint4 *frows,
                                               uint32_t *ftags, uint32_t *blockcounts,
                                               double *height, double *inferred)
{
    __shared__ uint32_t s_cnt[16];
    __shared__ unsigned long long s_live[16][GVOM_MAX_SLOTS + 1];   // [wave][source]: live-tile masks
    __shared__ int s_zocc[16][WAVE];
    __shared__ uint32_t s_hocc[16][WAVE];
    __shared__ int s_zfree[16][WAVE];

    // source descriptors: by kernel argument when they fit (no H2D copy per combine)
    const cptr_desc descs = MEM ? (cptr_desc)descs_mem : (cptr_desc)KD.d;
    const int lane = threadIdx.x & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform by construction
    const int sx = blockIdx.x * WAVE + lane;
    const int sy = P.sy_lo + blockIdx.y;
    const bool col_ok = sx < P.xy;
    const int x = wrap_sub(col_ok ? sx : 0, P.om[0], P.xy);
    const int y = wrap_sub(sy, P.om[1], P.xy);
    const int nsrc = P.nslots + P.has_prev;

    unsigned long long okmask = 0ull;                    // sources whose window contains (x, y)
    for (int s = 0; s < nsrc; ++s) {
        const int xs = x + descs[s].d[0], ys = y + descs[s].d[1];
        if (xs >= 0 && xs < P.xy && ys >= 0 && ys < P.xy) okmask |= 1ull << s;
    }
    if (!col_ok) okmask = 0ull;

    const uint32_t colbase = (uint32_t)sy * P.zs * P.xy + (col_ok ? sx : 0);
    const uint32_t tbase = (uint32_t)sy * P.zs * P.nseg + blockIdx.x;   // tile of (sy, sz=0, this segment)
    uint32_t running = 0;                                // rows used by this wave so far
    int zocc = INT_MAX, zfree = INT_MAX;
    uint32_t hocc = 0x3f800000u;
    // this wave owns window-z chunks [w*cpw, (w+1)*cpw) of zc levels each (ascending z) and the
    // static compact-row range starting at rbase
    const uint32_t rbase = ((blockIdx.y * gridDim.x + blockIdx.x) * (uint32_t)P.nz + w) *
                           (uint32_t)(WAVE * P.zc * P.cpw);
    if (ZC16) {
        // Phase 0: tile liveness of EVERY source for all (<= 64) tiles this wave will visit: lane
        // (16*cc + k) fetches the tag of tile k of chunk cc -- one vector load per source, four
        // sources in flight -- and a ballot turns each into a wave-uniform 64-bit mask kept in LDS.
        const int cc_l = lane >> 4, k_l = lane & 15;
        const int zl = (w * P.cpw + cc_l) * P.zc + k_l;
        const bool valid_l = cc_l < P.cpw && k_l < P.zc && zl < P.zs;
        const uint32_t tl = tbase + (uint32_t)wrap_add(valid_l ? zl : 0, P.om[2], P.zs) * P.nseg;
        for (int s0 = 0; s0 < nsrc; s0 += 4) {
            uint32_t tv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ss = min(s0 + j, nsrc - 1);
                tv[j] = ((gptr_u32)descs[ss].tags)[tl];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ss = min(s0 + j, nsrc - 1);
                const unsigned long long m = GVOM_DBG(P, 4) ? 0ull : __ballot(valid_l && tv[j] == descs[ss].epoch);
                if (lane == 0 && s0 + j < nsrc) s_live[w][s0 + j] = m;
            }
        }
    }
    for (int cc = 0; cc < P.cpw; ++cc) {
    const int z0 = (w * P.cpw + cc) * P.zc;
    if (z0 >= P.zs) break;
    const int z1 = min(z0 + P.zc, P.zs);
    if (ZC16) {
        // a chunk whose 16 tiles are dead in EVERY source is "never observed" throughout: nothing
        // to read, fold or write (wave-uniform early-out; most chunks above/below the lidar's
        // vertical field of view take it)
        uint32_t anylive = 0;
        for (int s = 0; s < nsrc; ++s) anylive |= (uint32_t)(s_live[w][s] >> (16 * cc)) & 0xffffu;
        if (__builtin_amdgcn_readfirstlane(anylive) == 0) continue;
    }

    // one occupied voxel: gather over the sources, store its compact row
    auto emit = [&](int z, uint32_t L, uint32_t row) {
        uint32_t h = 0, t = 0, m = 0x3f800000u;          // gvom.py:222-228 (0, 0, 1.0f)
        for (int s = 0; s < nsrc; ++s) {
            const int zz = z + descs[s].d[2];
            if (((okmask >> s) & 1ull) && zz >= 0 && zz < P.zs &&
                descs[s].tags[tbase + (L - colbase) / P.xy * P.nseg] == descs[s].epoch) {   // live tile
                const int st = descs[s].state[L];
                if (st >= 0) {                                            // gvom.py:841,910-912
                    const uint4 rv = descs[s].rows[st];
                    h += rv.x; t += rv.y; m = min(m, rv.z);
                }
            }
        }
        fstate[L] = (int32_t)row;
        frows[row] = make_uint4(h, t, m, 0u);
        if (z == zocc) hocc = m;
    };

    if (ZC16) {
        int c[16];
        uint32_t occbits = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = -1;
        // sources two at a time: 32 UNCONDITIONAL independent 256-byte row loads in flight (a dead
        // tile redirects its load to the always-hot first row of the array, costing no HBM
        // traffic; a branch-guarded load would make hipcc emit `s_waitcnt vmcnt(0)` in front of
        // every load).  The window test is applied to the loaded value afterwards; folding is
        // in source order (ring slots, then the previous fused map).
        auto fold = [&](const int (&st)[16], bool is_prev) {
            if (!is_prev) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (st[k] >= 0) occbits |= 1u << k;                                   // gvom.py:963
                    else if (st[k] < -1 && !((occbits >> k) & 1u)) c[k] = add_free(c[k], st[k] + 1);   // gvom.py:967
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (!((occbits >> k) & 1u)) {
                        if (st[k] >= 0 && c[k] >= -11) occbits |= 1u << k;                // gvom.py:992
                        else if (st[k] < -1) c[k] = add_free(c[k], st[k] + 1);             // gvom.py:996
                    }
                }
            }
        };
        for (int s0 = 0; s0 < nsrc; s0 += 2) {
            const int sA = s0, sB = min(s0 + 1, nsrc - 1);
            const bool hasB = s0 + 1 < nsrc;
            const gptr_i32 spA = (gptr_i32)descs[sA].state, spB = (gptr_i32)descs[sB].state;
            const uint32_t liveA = (uint32_t)(s_live[w][sA] >> (16 * cc)) & 0xffffu;
            const uint32_t liveB = hasB ? ((uint32_t)(s_live[w][sB] >> (16 * cc)) & 0xffffu) : 0u;
            int stA[16], stB[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int z = z0 + k;
                const int sz = wrap_add(z < P.zs ? z : 0, P.om[2], P.zs);
                const uint32_t real = colbase + (uint32_t)sz * P.xy;
                stA[k] = spA[((liveA >> k) & 1u) ? real : (uint32_t)lane];
                stB[k] = spB[((liveB >> k) & 1u) ? real : (uint32_t)lane];
            }
            const int dzA = descs[sA].d[2], dzB = descs[sB].d[2];
            const bool okA = (okmask >> sA) & 1ull, okB = hasB && ((okmask >> sB) & 1ull);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int z = z0 + k;
                const bool inA = okA && ((liveA >> k) & 1u) && z < z1 && z + dzA >= 0 && z + dzA < P.zs;
                const bool inB = okB && ((liveB >> k) & 1u) && z < z1 && z + dzB >= 0 && z + dzB < P.zs;
                stA[k] = inA ? stA[k] : -1;               // -1 == "never observed": no effect
                stB[k] = inB ? stB[k] : -1;
            }
            fold(stA, sA >= P.nslots);
            if (hasB) fold(stB, sB >= P.nslots);
        }
        if (!col_ok) occbits = 0;
        // free / unknown codes, first free z, first occupied z.  A tile in which every voxel is
        // still "never observed" is not written at all (its tag stays != epoch).
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int z = z0 + k;
            const bool occ = (occbits >> k) & 1u;
            const bool inside = col_ok && z < z1;
            if (!GVOM_DBG(P, 1) && __any(inside && (occ || c[k] != -1))) {
                const int sz = wrap_add(z < P.zs ? z : 0, P.om[2], P.zs);
                if (lane == 0) ftags[tbase + (uint32_t)sz * P.nseg] = P.epoch;
                if (inside) {
                    if (occ) {
                        if (zocc == INT_MAX) zocc = z;
                    } else {
                        fstate[colbase + (uint32_t)sz * P.xy] = c[k];
                        if (c[k] < -1 && zfree == INT_MAX) zfree = z;                     // gvom.py:551
                    }
                }
            }
        }
        // occupied voxels (sparse, skipped by most waves): hit/total sums and min-height min over
        // every source where the voxel is occupied (gvom.py:841,910-912).  Per source, the state
        // rows of all occupied z levels are fetched together, then the three compact arrays are
        // gathered in batches of 16 independent loads -- not one dependent chain per voxel.
        for (int kg = 0; kg < 16; kg += 4) {             // groups of 4 z levels keep the registers low
            const uint32_t gbits = GVOM_DBG(P, 2) ? 0u : (occbits >> kg) & 0xfu;
            if (!__any(gbits != 0)) continue;               // wave-uniform: most groups are empty
            uint32_t hh[4], tt[4], mm[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { hh[j] = 0; tt[j] = 0; mm[j] = 0x3f800000u; }    // gvom.py:222-228
            for (int s = 0; s < nsrc; ++s) {
                const gptr_i32 sp = (gptr_i32)descs[s].state;
                const gptr_v4u rp = (gptr_v4u)descs[s].rows;
                const uint32_t live = ((uint32_t)(s_live[w][s] >> (16 * cc)) >> kg) & 0xfu;
                const int dz = descs[s].d[2];
                const bool okS = (okmask >> s) & 1ull;
                int st[4];
                bool use[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int z = z0 + kg + j;
                    use[j] = ((gbits >> j) & 1u) && okS && ((live >> j) & 1u) && z + dz >= 0 && z + dz < P.zs;
                    const int sz = wrap_add(z < P.zs ? z : 0, P.om[2], P.zs);
                    st[j] = sp[use[j] ? colbase + (uint32_t)sz * P.xy : (uint32_t)lane];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) st[j] = (use[j] && st[j] >= 0) ? st[j] : -1;
                uint32_t gh[4], gt[4], gm[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t r = st[j] >= 0 ? (uint32_t)st[j] : 0u;
                    const v4u rv = rp[r];
                    gh[j] = rv.x; gt[j] = rv.y; gm[j] = rv.z;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (st[j] >= 0) { hh[j] += gh[j]; tt[j] += gt[j]; mm[j] = min(mm[j], gm[j]); }
            }
            // rows by ballot + prefix popcount inside the wave's static range
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool occ = (gbits >> j) & 1u;
                const unsigned long long b = __ballot(occ);
                if (occ) {
                    const int z = z0 + kg + j;
                    const int sz = wrap_add(z, P.om[2], P.zs);
                    const uint32_t row = rbase + running + (uint32_t)__popcll(b & lanemask_lt());
                    fstate[colbase + (uint32_t)sz * P.xy] = (int32_t)row;
                    frows[row] = make_uint4(hh[j], tt[j], mm[j], 0u);
                    if (z == zocc) hocc = mm[j];
                }
                running += (uint32_t)__popcll(b);
            }
        }
    } else {
        for (int z = z0; z < z1; ++z) {
            const int sz = wrap_add(z, P.om[2], P.zs);
            const uint32_t L = colbase + (uint32_t)sz * P.xy;
            int c = -1;
            bool occ = false;
            const uint32_t T = tbase + (uint32_t)sz * P.nseg;
            for (int s = 0; s < P.nslots; ++s) {
                const int zz = z + descs[s].d[2];
                if (((okmask >> s) & 1ull) && zz >= 0 && zz < P.zs && descs[s].tags[T] == descs[s].epoch) {
                    const int st = descs[s].state[L];
                    if (st >= 0) occ = true;                              // gvom.py:963
                    else if (st < -1 && !occ) c = add_free(c, st + 1);    // gvom.py:967
                }
            }
            if (P.has_prev) {
                const int s = P.nslots;
                const int zz = z + descs[s].d[2];
                if (((okmask >> s) & 1ull) && zz >= 0 && zz < P.zs && !occ &&
                    descs[s].tags[T] == descs[s].epoch) {
                    const int p = descs[s].state[L];
                    if (p >= 0 && c >= -11) occ = true;                   // gvom.py:992
                    else if (p < -1) c = add_free(c, p + 1);              // gvom.py:996
                }
            }
            occ = occ && col_ok;
            const bool nonempty = __any(col_ok && (occ || c != -1));
            if (nonempty && lane == 0) ftags[T] = P.epoch;
            if (nonempty && col_ok && !occ) {
                fstate[L] = c;
                if (c < -1 && zfree == INT_MAX) zfree = z;                // gvom.py:551
            }
            const unsigned long long b = __ballot(occ);
            if (b != 0ull) {
                if (occ) {
                    if (zocc == INT_MAX) zocc = z;
                    emit(z, L, rbase + running + (uint32_t)__popcll(b & lanemask_lt()));
                }
                running += (uint32_t)__popcll(b);
            }
        }
    }

    }   // chunks of this wave

    // ---- column tail: height (gvom.py:525-540) and inferred height (gvom.py:544-554) ------
    s_zocc[w][lane] = zocc; s_hocc[w][lane] = hocc; s_zfree[w][lane] = zfree;
    if (lane == 0) s_cnt[w] = running;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int k = 0; k < P.nz; ++k) tot += s_cnt[k];
        blockcounts[blockIdx.y * gridDim.x + blockIdx.x] = tot;
    }
    if (w == 0 && col_ok) {
        int zo = INT_MAX, zf = INT_MAX;
        uint32_t hb = 0x3f800000u;
        for (int k = 0; k < P.nz; ++k) {
            if (zo == INT_MAX && s_zocc[k][lane] != INT_MAX) { zo = s_zocc[k][lane]; hb = s_hocc[k][lane]; }
            if (zf == INT_MAX && s_zfree[k][lane] != INT_MAX) zf = s_zfree[k][lane];
        }
        double hval = -1000.0;
        const double xp = ((P.origin[0] + (double)x) * P.xy_res) - P.ego[0];
        const double yp = ((P.origin[1] + (double)y) * P.xy_res) - P.ego[1];
        if (xp * xp + yp * yp <= P.radius2) hval = P.ego[2] - P.ground_to_lidar_height;
        if (zo != INT_MAX)
            hval = (((double)__uint_as_float(hb) + (double)zo) + P.origin[2]) * P.z_res;
        height[(size_t)sy * P.hs + sx] = hval;
        inferred[(size_t)sy * P.hs + sx] =
            (zf != INT_MAX) ? ((double)zf + P.origin[2]) * P.z_res : -1000.0;
    }
}

// ------------------------------------------------------------------------------------------
// k_fuse4: the same fusion as k_fuse<true> with 16-byte accesses (xy % 4 == 0, 16-level chunks).
// Lane (g = lane & 15, q = lane >> 4) owns the 4 columns sx = 64*seg + 4g .. +3 and, in every
// chunk, the 4 levels z = z0 + 4j + q (j = 0..3): one int4 load per source and j covers 4 tiles
// (4 levels x 64 columns) of the wave -- 4 loads instead of 16 per source, at the 16-B/lane rate
// (6.5 TB/s vs 4.0 TB/s for 4-B/lane loads on this part) -- and the codes go out as int4 stores.
// Cell (j, i) of a lane is voxel (sx = 4g + i, z = z0 + 4j + q); its fold state lives in
// c[4j + i] / bit 4j + i of occbits.
// ------------------------------------------------------------------------------------------
// Ring slots are read through their 16-bit codes (k_encode), SPR slots per round trip, and summed with
// saturating packed adds; the previous fused map through its 32-bit states, loaded alongside and
// folded last.  A saturated sum marks a CANDIDATE: occupied in some slot, or more than 65534 passes
// (the sensor's own voxel); the per-voxel path below settles which from the 32-bit states.
// sources per round trip of the tag phase / of the per-voxel path for rings longer than 2 (measured on
// c3 / m256b8: 12 / 12 -> 36.6 / 39.9 us, 4 / 4 -> 39.7 / 43.8)
#define FUSE_NG 12
#define FUSE_TW 12
template <int SPR, bool MEM>
__global__ __launch_bounds__(1024) void k_fuse4(const FuseParams P, const FuseDescs KD,
                                                const MapDesc *__restrict__ descs_mem,
                                                int32_t *fstate, uint4 *frows,
                                                uint32_t *ftags, uint32_t *blockcounts,
                                                double *height, double *inferred)
{
    __shared__ uint32_t s_cnt[16];
    __shared__ unsigned long long s_live[16][GVOM_MAX_SLOTS + 1];
    __shared__ unsigned long long s_zh[16][WAVE];          // per wave and column: min of (z << 32 | min-height bits) over occupied voxels
    __shared__ uint32_t s_zf[16][WAVE];                    // per wave and column: lowest observed-free z
    __shared__ uint16_t s_list[16][8 * WAVE];              // per wave: the occupied voxels of a chunk, compacted (j << 8 | lane << 2 | i)

    const cptr_desc descs = MEM ? (cptr_desc)descs_mem : (cptr_desc)KD.d;
    const int lane = threadIdx.x & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane & 15, q = lane >> 4;
    const int sxb = blockIdx.x * WAVE + 4 * g;            // first of this lane's 4 columns
    const int sy = P.sy_lo + blockIdx.y;
    const bool col_ok = sxb < P.xy;                        // xy % 4 == 0: all four or none
    const int y = wrap_sub(sy, P.om[1], P.xy);
    const int nsrc = P.nslots + P.has_prev;

    unsigned long long okm[4] = {0ull, 0ull, 0ull, 0ull};   // per column: sources whose window contains (x, y)
    int xw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xw[i] = wrap_sub(col_ok ? sxb + i : 0, P.om[0], P.xy);
    for (int s = 0; s < nsrc; ++s) {
        const int ys = y + descs[s].d[1];
        const bool yok = ys >= 0 && ys < P.xy;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int xs = xw[i] + descs[s].d[0];
            if (col_ok && yok && xs >= 0 && xs < P.xy) okm[i] |= 1ull << s;
        }
    }

    const uint32_t colbase = (uint32_t)sy * P.zs * P.xy + (col_ok ? sxb : 0);
    const uint32_t tbase = (uint32_t)sy * P.zs * P.nseg + blockIdx.x;
    uint32_t running = 0;
    int zfree[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX};
    s_zh[w][lane] = ((unsigned long long)INT_MAX << 32) | 0x3f800000ull;      // wave-private until the tail
    s_zf[w][lane] = (uint32_t)INT_MAX;
    const uint32_t rbase = ((blockIdx.y * gridDim.x + blockIdx.x) * (uint32_t)P.nz + w) *
                           (uint32_t)(WAVE * P.zc * P.cpw);

    {   // phase 0: live-tile masks of every source for the (<= 64) tiles of this wave
        const int cc_l = lane >> 4, k_l = lane & 15;
        const int zl = (cc_l * P.nz + w) * P.zc + k_l;      // chunks are dealt round-robin to the waves (see below)
        const bool valid_l = cc_l < P.cpw && k_l < P.zc && zl < P.zs;
        const uint32_t tl = tbase + (uint32_t)wrap_add(valid_l ? zl : 0, P.om[2], P.zs) * P.nseg;
        constexpr int TW = SPR == 2 ? 4 : FUSE_TW;
        for (int s0 = 0; s0 < nsrc; s0 += TW) {
            uint32_t tv[TW];
#pragma unroll
            for (int j = 0; j < TW; ++j) tv[j] = ((gptr_u32)descs[min(s0 + j, nsrc - 1)].tags)[tl];
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                const unsigned long long m = GVOM_DBG(P, 4) ? 0ull : __ballot(valid_l && tv[j] == descs[min(s0 + j, nsrc - 1)].epoch);
                if (lane == 0 && s0 + j < nsrc) s_live[w][s0 + j] = m;
            }
        }
    }

    for (int cc = 0; cc < P.cpw; ++cc) {
        // chunk cc of wave w is chunk cc*nz + w of the column: the observed band (ground +- a few
        // metres) is a run of neighbouring chunks, and this spreads it over all waves of the workgroup
        const int z0 = (cc * P.nz + w) * P.zc;
        if (z0 >= P.zs) break;
        const int z1 = min(z0 + P.zc, P.zs);
        uint32_t anylive = 0;
        for (int s = 0; s < nsrc; ++s) anylive |= (uint32_t)(s_live[w][s] >> (16 * cc)) & 0xffffu;
        if (__builtin_amdgcn_readfirstlane(anylive) == 0) continue;     // chunk dead in every source

        int zq[4];                                        // this lane's 4 levels and their row offsets
        uint32_t roff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            zq[j] = z0 + 4 * j + q;
            roff[j] = colbase + (uint32_t)wrap_add(zq[j] < P.zs ? zq[j] : 0, P.om[2], P.zs) * P.xy;
        }
        int c[16];
        uint32_t occbits = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = -1;

        auto fold = [&](const v4i (&v)[4], uint32_t live, int dz, int s, bool is_prev) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool zin = ((live >> (4 * j + q)) & 1u) && zq[j] < z1 && zq[j] + dz >= 0 && zq[j] + dz < P.zs;
                const int vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int st = (zin && ((okm[i] >> s) & 1ull)) ? vv[i] : -1;     // -1: no effect
                    const int k = 4 * j + i;
                    if (!is_prev) {
                        if (st >= 0) occbits |= 1u << k;                                  // gvom.py:963
                        else if (st < -1 && !((occbits >> k) & 1u)) c[k] = add_free(c[k], st + 1);   // gvom.py:967
                    } else if (!((occbits >> k) & 1u)) {
                        if (st >= 0 && c[k] >= -11) occbits |= 1u << k;                   // gvom.py:992
                        else if (st < -1) c[k] = add_free(c[k], st + 1);                  // gvom.py:996
                    }
                }
            }
        };
        // codes of one ring slot: 4 x 16 bit = this lane's 4 columns at one level.  A slot's free counts
        // only matter while no slot has the voxel occupied, so the sum is order-free (gvom.py:963-967).
        uint32_t cpk[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};    // cell k = 4j + i: half (i & 1) of cpk[2j + (i >> 1)]
        auto fold16 = [&](const v2u (&b)[4], uint32_t live, int dz, int s) {
            const uint32_t cm0 = (((okm[0] >> s) & 1ull) ? 0xffffu : 0u) | (((okm[1] >> s) & 1ull) ? 0xffff0000u : 0u);
            const uint32_t cm1 = (((okm[2] >> s) & 1ull) ? 0xffffu : 0u) | (((okm[3] >> s) & 1ull) ? 0xffff0000u : 0u);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool zin = ((live >> (4 * j + q)) & 1u) && zq[j] < z1 && zq[j] + dz >= 0 && zq[j] + dz < P.zs;
                cpk[2 * j] = pk_add_sat_u16(cpk[2 * j], zin ? (b[j].x & cm0) : 0u);
                cpk[2 * j + 1] = pk_add_sat_u16(cpk[2 * j + 1], zin ? (b[j].y & cm1) : 0u);
            }
        };
        v4i vp[4];
        uint32_t livep = 0;
        if (P.has_prev) {                                 // in flight while the slots are folded
            livep = (uint32_t)(s_live[w][P.nslots] >> (16 * cc)) & 0xffffu;
            const gptr_i32 sp = (gptr_i32)descs[P.nslots].state;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                vp[j] = *(gptr_v4i)(sp + (((livep >> (4 * j + q)) & 1u) ? roff[j] : (uint32_t)(4 * lane)));
        }
        for (int s0 = 0; s0 < P.nslots; s0 += SPR) {
            v2u b[SPR][4];
            uint32_t live[SPR];
#pragma unroll
            for (int u = 0; u < SPR; ++u) {               // 4 * SPR unconditional 8-byte loads in flight
                const int sI = min(s0 + u, P.nslots - 1);
                const gptr_u16 cp = (gptr_u16)descs[sI].code16;
                live[u] = s0 + u < P.nslots ? ((uint32_t)(s_live[w][sI] >> (16 * cc)) & 0xffffu) : 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    b[u][j] = *(gptr_u2)(cp + (((live[u] >> (4 * j + q)) & 1u) ? roff[j] : (uint32_t)(4 * lane)));
            }
#pragma unroll
            for (int u = 0; u < SPR; ++u) {
                const int sI = min(s0 + u, P.nslots - 1);
                if (s0 + u < P.nslots) fold16(b[u], live[u], descs[sI].d[2], sI);
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t half = (cpk[k >> 1] >> (16 * (k & 1))) & 0xffffu;
            c[k] = -1 - (int)half;
            if (half == 0xffffu) occbits |= 1u << k;                       // candidate
        }
        if (P.has_prev) fold(vp, livep, descs[P.nslots].d[2], P.nslots, true);
        if (!col_ok) occbits = 0;

        // codes: one int4 store per live-or-new tile row segment; first free / occupied z per column
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool inside = col_ok && zq[j] < z1;
            bool nonempty = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) nonempty = nonempty || ((occbits >> (4 * j + i)) & 1u) || c[4 * j + i] != -1;
            const unsigned long long nb = __ballot(inside && nonempty);
            if (((nb >> (16 * q)) & 0xffffull) && !GVOM_DBG(P, 1)) {           // some lane of MY tile (same q) has content
                if (g == 0 && zq[j] < z1)
                    ftags[tbase + (uint32_t)wrap_add(zq[j], P.om[2], P.zs) * P.nseg] = P.epoch;
                if (inside) {
                    *reinterpret_cast<int4 *>(fstate + roff[j]) = make_int4(c[4 * j], c[4 * j + 1], c[4 * j + 2], c[4 * j + 3]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (!((occbits >> (4 * j + i)) & 1u) && c[4 * j + i] < -1 && zfree[i] == INT_MAX) zfree[i] = zq[j];   // gvom.py:551
                    }
                }
            }
        }

        // occupied voxels (sparse): the occupied voxels of the chunk are compacted across the wave and
        // handled ONE LANE PER VOXEL: every source's state at that voxel is fetched in one round trip
        // and the rows' counts in a second one, whatever the number of levels and sources (the
        // per-level, per-source form was a chain of up to 16 round trips; per 4-level group and 4
        // sources at a time it still was 2 x 4 x ceil(sources / 4)).
        uint32_t n = 0;
        auto emit = [&]() {
            for (uint32_t base = 0; base < n; base += WAVE) {
                const uint32_t t = base + (uint32_t)lane;
                const bool on = t < n;
                const uint32_t e = s_list[w][on ? t : 0u];
                const int ci = (int)(e & 3u), go = (int)((e >> 2) & 15u), qo = (int)((e >> 6) & 3u), jo = (int)(e >> 8);
                const int zv = z0 + 4 * jo + qo;                                 // window level of my voxel
                const int col = 4 * go + ci;                                     // column within the workgroup
                const int sxc = blockIdx.x * WAVE + col;
                const int xwc = wrap_sub(sxc, P.om[0], P.xy);
                const uint32_t off = (uint32_t)sy * P.zs * P.xy + (uint32_t)wrap_add(zv, P.om[2], P.zs) * P.xy + (uint32_t)sxc;
                const int lbit = 16 * cc + 4 * jo + qo;                           // my tile in s_live
                uint32_t hh = 0, tt = 0, mm = 0x3f800000u;
                bool slot_occ = false;                                        // occupied in some ring slot
                int cnt = -1, stp = -1;                                        // exact free count of the slots; the previous map's state
                constexpr int NG = SPR == 2 ? 4 : FUSE_NG;
                for (int s0 = 0; s0 < nsrc; s0 += NG) {
                    int st[NG];
                    bool ok[NG];
#pragma unroll
                    for (int u = 0; u < NG; ++u) {
                        const int sI = min(s0 + u, nsrc - 1);
                        const int xs = xwc + descs[sI].d[0], ys = y + descs[sI].d[1], zs2 = zv + descs[sI].d[2];
                        ok[u] = on && s0 + u < nsrc && ((s_live[w][sI] >> lbit) & 1ull) &&
                                xs >= 0 && xs < P.xy && ys >= 0 && ys < P.xy && zs2 >= 0 && zs2 < P.zs;
                        st[u] = ((gptr_i32)descs[sI].state)[ok[u] ? off : (uint32_t)lane];
                    }
                    uint32_t gh[NG], gt[NG], gm[NG];
#pragma unroll
                    for (int u = 0; u < NG; ++u) {
                        const int sI = min(s0 + u, nsrc - 1);
                        if (!ok[u]) st[u] = -1;
                        const uint32_t r = st[u] >= 0 ? (uint32_t)st[u] : 0u;
                        const v4u rv = ((gptr_v4u)descs[sI].rows)[r];           // one 16-byte row: hit, total, min-height
                        gh[u] = rv.x; gt[u] = rv.y; gm[u] = rv.z;
                    }
#pragma unroll
                    for (int u = 0; u < NG; ++u) {
                        if (st[u] >= 0) { hh += gh[u]; tt += gt[u]; mm = min(mm, gm[u]); }          // gvom.py:910-912
                        if (s0 + u < P.nslots) {
                            if (st[u] >= 0) slot_occ = true;                                      // gvom.py:963
                            else if (st[u] < -1) cnt = add_free(cnt, st[u] + 1);                  // gvom.py:967
                        } else if (s0 + u == P.nslots) stp = st[u];
                    }
                }
                // a candidate is occupied (gvom.py:963, 992) -- or free with more passes than a code holds
                const bool occupied = on && (slot_occ || (stp >= 0 && cnt >= -11));
                const unsigned long long ob = __ballot(occupied);
                if (occupied) {
                    const uint32_t row = rbase + running + (uint32_t)__popcll(ob & lanemask_lt());
                    fstate[off] = (int32_t)row;
                    frows[row] = make_uint4(hh, tt, mm, 0u);
                    atomicMin(&s_zh[w][col], ((unsigned long long)(uint32_t)zv << 32) | mm);   // lowest occupied level wins
                } else if (on) {
                    fstate[off] = stp < -1 ? add_free(cnt, stp + 1) : cnt;                        // gvom.py:996
                    atomicMin(&s_zf[w][col], (uint32_t)zv);                                       // gvom.py:551
                }
                running += (uint32_t)__popcll(ob);
            }
            n = 0;
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t gbits = GVOM_DBG(P, 2) ? 0u : (occbits >> (4 * j)) & 0xfu;
            if (!__any(gbits != 0)) continue;               // wave-uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool occ = (gbits >> i) & 1u;
                const unsigned long long b = __ballot(occ);
                if (occ) s_list[w][n + (uint32_t)__popcll(b & lanemask_lt())] = (uint16_t)(((uint32_t)j << 8) | ((uint32_t)lane << 2) | (uint32_t)i);
                n += (uint32_t)__popcll(b);
            }
            if (n > 4 * WAVE) emit();                       // the next group (<= 256 voxels) might not fit
        }
        if (n) emit();
    }   // chunks

    // ---- column tail: lowest occupied z (+ its min-height) / lowest free z per column: LDS minima
    // per wave (a column's levels are spread over 4 lanes and over the waves), merged below.
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (zfree[i] != INT_MAX) atomicMin(&s_zf[w][4 * g + i], (uint32_t)zfree[i]);
    if (lane == 0) s_cnt[w] = running;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int k = 0; k < P.nz; ++k) tot += s_cnt[k];
        blockcounts[blockIdx.y * gridDim.x + blockIdx.x] = tot;
    }
    const int sx = blockIdx.x * WAVE + lane;
    if (w == 0 && sx < P.xy && !GVOM_DBG(P, 16)) {
        const int x = wrap_sub(sx, P.om[0], P.xy);
        unsigned long long zh = ((unsigned long long)INT_MAX << 32) | 0x3f800000ull;
        uint32_t zfu = (uint32_t)INT_MAX;
        for (int k = 0; k < P.nz; ++k) { zh = min(zh, s_zh[k][lane]); zfu = min(zfu, s_zf[k][lane]); }
        const int zo = (int)(zh >> 32), zf = (int)zfu;
        const uint32_t hb = (uint32_t)zh;
        double hval = -1000.0;
        const double xp = ((P.origin[0] + (double)x) * P.xy_res) - P.ego[0];
        const double yp = ((P.origin[1] + (double)y) * P.xy_res) - P.ego[1];
        if (xp * xp + yp * yp <= P.radius2) hval = P.ego[2] - P.ground_to_lidar_height;
        if (zo != INT_MAX)
            hval = (((double)__uint_as_float(hb) + (double)zo) + P.origin[2]) * P.z_res;
        height[(size_t)sy * P.hs + sx] = hval;
        inferred[(size_t)sy * P.hs + sx] =
            (zf != INT_MAX) ? ((double)zf + P.origin[2]) * P.z_res : -1000.0;
    }
}

// ------------------------------------------------------------------------------------------
// k_fuse1: the fusion of ONE ring slot (+ the previous fused map) -- buffer_size = 1 rings (the headline 256^3 config, c2)
// and any ring that holds a single scan.  Same column decomposition, lane layout, outputs and row numbering as k_fuse4,
// built for latency: with one slot there is nothing to sum, so the slot is read through its 32-bit STATES (its free count
// and, where it is occupied, its compact row arrive with the one load), the occupied voxels of a chunk are settled in ONE
// round trip (both sources' rows are known), nothing is kept per cell between the two sources -- 64 VGPRs instead of
// k_fuse4's 111, twice the waves per SIMD -- and a wave takes 2 chunks instead of 4 (workgroups of up to 8 waves): a wave's
// chain of dependent round trips is 1 + 2 x 2 instead of 1 + 4 x 3.  (k_fuse4 at S = 1: 20 us for 50 MB on the 256^3 grid.)
// ------------------------------------------------------------------------------------------
#define FUSE1_LIST 256
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_fuse1(
    const FuseParams P, const FuseDescs KD, int32_t *fstate, uint4 *frows, uint32_t *ftags, uint32_t *blockcounts,
    double *height, double *inferred)
{
    __shared__ uint32_t s_cnt[8];
    __shared__ unsigned long long s_live[8][2];
    __shared__ unsigned long long s_zh[8][WAVE];           // per wave and column: min of (z << 32 | min-height bits) over occupied voxels
    __shared__ uint32_t s_zf[8][WAVE];                     // per wave and column: lowest observed-free z
    __shared__ uint16_t s_list[8][FUSE1_LIST];             // per wave: the occupied voxels of a chunk, compacted (j << 8 | lane << 2 | i)
    __shared__ int32_t s_sts[8][FUSE1_LIST], s_stp[8][FUSE1_LIST];   // ... and the two sources' states there

    const cptr_desc descs = (cptr_desc)KD.d;
    const int lane = threadIdx.x & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane & 15, q = lane >> 4;
    const int sxb = blockIdx.x * WAVE + 4 * g;            // first of this lane's 4 columns
    const int sy = P.sy_lo + blockIdx.y;
    const bool col_ok = sxb < P.xy;                        // xy % 4 == 0: all four or none
    const int y = wrap_sub(sy, P.om[1], P.xy);
    const bool has_prev = P.has_prev != 0;
    const int dsx = descs[0].d[0], dsy = descs[0].d[1], dsz = descs[0].d[2];
    const int dpx = has_prev ? descs[1].d[0] : 0, dpy = has_prev ? descs[1].d[1] : 0, dpz = has_prev ? descs[1].d[2] : 0;

    uint32_t oks = 0, okp = 0;                             // bit i: the source's window contains column i of this lane
    {
        const bool ys = y + dsy >= 0 && y + dsy < P.xy, yp = has_prev && y + dpy >= 0 && y + dpy < P.xy;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int xw = wrap_sub(col_ok ? sxb + i : 0, P.om[0], P.xy);
            if (col_ok && ys && xw + dsx >= 0 && xw + dsx < P.xy) oks |= 1u << i;
            if (col_ok && yp && xw + dpx >= 0 && xw + dpx < P.xy) okp |= 1u << i;
        }
    }
    const uint32_t colbase = (uint32_t)sy * P.zs * P.xy + (col_ok ? sxb : 0);
    const uint32_t tbase = (uint32_t)sy * P.zs * P.nseg + blockIdx.x;
    uint32_t running = 0;
    int zfree[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX};
    s_zh[w][lane] = ((unsigned long long)INT_MAX << 32) | 0x3f800000ull;      // wave-private until the tail
    s_zf[w][lane] = (uint32_t)INT_MAX;
    const uint32_t rbase = ((blockIdx.y * gridDim.x + blockIdx.x) * (uint32_t)P.nz + w) * (uint32_t)(WAVE * P.zc * P.cpw);

    {   // live-tile masks of the two sources for the (<= 64) tiles of this wave
        const int cc_l = lane >> 4, k_l = lane & 15;
        const int zl = (cc_l * P.nz + w) * P.zc + k_l;      // chunks are dealt round-robin to the waves
        const bool valid_l = cc_l < P.cpw && k_l < P.zc && zl < P.zs;
        const uint32_t tl = tbase + (uint32_t)wrap_add(valid_l ? zl : 0, P.om[2], P.zs) * P.nseg;
        const uint32_t ts = ((gptr_u32)descs[0].tags)[tl];
        const uint32_t tp = has_prev ? ((gptr_u32)descs[1].tags)[tl] : 0u;
        const unsigned long long ms = __ballot(valid_l && ts == descs[0].epoch);
        const unsigned long long mp = __ballot(has_prev && valid_l && tp == descs[1].epoch);
        if (lane == 0) { s_live[w][0] = ms; s_live[w][1] = mp; }
    }
    const gptr_i32 ss = (gptr_i32)descs[0].state;
    const gptr_i32 sp = (gptr_i32)descs[has_prev ? 1 : 0].state;

    for (int cc = 0; cc < P.cpw; ++cc) {
        const int z0 = (cc * P.nz + w) * P.zc;
        if (z0 >= P.zs) break;
        const int z1 = min(z0 + P.zc, P.zs);
        const uint32_t lives = (uint32_t)(s_live[w][0] >> (16 * cc)) & 0xffffu;
        const uint32_t livep = (uint32_t)(s_live[w][1] >> (16 * cc)) & 0xffffu;
        if (__builtin_amdgcn_readfirstlane(lives | livep) == 0) continue;     // chunk dead in both sources

        uint32_t roff[4];
        v4i vs[4], vp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                     // 8 unconditional 16-byte loads in flight
            const int zq = z0 + 4 * j + q;
            roff[j] = colbase + (uint32_t)wrap_add(zq < P.zs ? zq : 0, P.om[2], P.zs) * P.xy;
            vs[j] = *(gptr_v4i)(ss + (((lives >> (4 * j + q)) & 1u) ? roff[j] : (uint32_t)(4 * lane)));
            vp[j] = *(gptr_v4i)(sp + (((livep >> (4 * j + q)) & 1u) ? roff[j] : (uint32_t)(4 * lane)));
        }
        uint32_t n = 0;
        // the occupied voxels listed so far: one lane per voxel, both sources' rows in one round trip
        auto emit = [&]() {
            for (uint32_t base = 0; base < n; base += WAVE) {
                const uint32_t t = base + (uint32_t)lane;
                const bool on = t < n;
                const uint32_t e = s_list[w][on ? t : 0u];
                const int sts = on ? s_sts[w][t] : -1, stp = on ? s_stp[w][t] : -1;
                const int ci = (int)(e & 3u), go = (int)((e >> 2) & 15u), qo = (int)((e >> 6) & 3u), jo = (int)(e >> 8);
                const int zv = z0 + 4 * jo + qo;
                const int col = 4 * go + ci;
                const int sxc = blockIdx.x * WAVE + col;
                const uint32_t off = (uint32_t)sy * P.zs * P.xy + (uint32_t)wrap_add(zv, P.om[2], P.zs) * P.xy + (uint32_t)sxc;
                const v4u rs = ((gptr_v4u)descs[0].rows)[sts >= 0 ? (uint32_t)sts : 0u];
                const v4u rp = ((gptr_v4u)descs[has_prev ? 1 : 0].rows)[stp >= 0 ? (uint32_t)stp : 0u];
                uint32_t hh = 0, tt = 0, mm = 0x3f800000u;
                if (sts >= 0) { hh += rs.x; tt += rs.y; mm = min(mm, rs.z); }            // gvom.py:910-912
                if (stp >= 0) { hh += rp.x; tt += rp.y; mm = min(mm, rp.z); }
                const unsigned long long ob = __ballot(on);
                if (on) {
                    const uint32_t row = rbase + running + (uint32_t)__popcll(ob & lanemask_lt());
                    fstate[off] = (int32_t)row;
                    frows[row] = make_uint4(hh, tt, mm, 0u);
                    atomicMin(&s_zh[w][col], ((unsigned long long)(uint32_t)zv << 32) | mm);   // lowest occupied level wins
                }
                running += (uint32_t)__popcll(ob);
            }
            n = 0;
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int zq = z0 + 4 * j + q;
            const bool zs_in = ((lives >> (4 * j + q)) & 1u) && zq < z1 && zq + dsz >= 0 && zq + dsz < P.zs;
            const bool zp_in = ((livep >> (4 * j + q)) & 1u) && zq < z1 && zq + dpz >= 0 && zq + dpz < P.zs;
            const int a[4] = {vs[j].x, vs[j].y, vs[j].z, vs[j].w}, b[4] = {vp[j].x, vp[j].y, vp[j].z, vp[j].w};
            int c[4];
            uint32_t occ = 0;
            int sts_[4], stp_[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int st = (zs_in && ((oks >> i) & 1u)) ? a[i] : -1;                 // -1: no effect
                const int pv = (zp_in && ((okp >> i) & 1u)) ? b[i] : -1;
                sts_[i] = st; stp_[i] = pv;
                c[i] = -1;
                if (st >= 0) occ |= 1u << i;                                              // gvom.py:963
                else if (st < -1) c[i] = add_free(c[i], st + 1);                          // gvom.py:967
                if (!((occ >> i) & 1u)) {
                    if (pv >= 0 && c[i] >= -11) occ |= 1u << i;                           // gvom.py:992
                    else if (pv < -1) c[i] = add_free(c[i], pv + 1);                      // gvom.py:996
                }
            }
            if (!col_ok) occ = 0;
            const bool inside = col_ok && zq < z1;
            const bool nonempty = occ != 0u || c[0] != -1 || c[1] != -1 || c[2] != -1 || c[3] != -1;
            const unsigned long long nb = __ballot(inside && nonempty);
            if (((nb >> (16 * q)) & 0xffffull) && !GVOM_DBG(P, 1)) {          // some lane of MY tile (same q) has content
                if (g == 0 && zq < z1) ftags[tbase + (uint32_t)wrap_add(zq, P.om[2], P.zs) * P.nseg] = P.epoch;
                if (inside) {
                    *reinterpret_cast<int4 *>(fstate + roff[j]) = make_int4(c[0], c[1], c[2], c[3]);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (!((occ >> i) & 1u) && c[i] < -1 && zfree[i] == INT_MAX) zfree[i] = zq;     // gvom.py:551
                }
            }
            if (__any(occ != 0u)) {                          // wave-uniform
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool o = (occ >> i) & 1u;
                    const unsigned long long bm = __ballot(o);
                    if (o) {
                        const uint32_t at = n + (uint32_t)__popcll(bm & lanemask_lt());
                        s_list[w][at] = (uint16_t)(((uint32_t)j << 8) | ((uint32_t)lane << 2) | (uint32_t)i);
                        s_sts[w][at] = sts_[i]; s_stp[w][at] = stp_[i];
                    }
                    n += (uint32_t)__popcll(bm);
                    if (n > FUSE1_LIST - WAVE) emit();     // the next column of cells (<= 64 voxels) might not fit
                }
            }
        }
        if (n) emit();
    }   // chunks

    // ---- column tail (as k_fuse4): lowest occupied z (+ its min-height) / lowest free z per column
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (zfree[i] != INT_MAX) atomicMin(&s_zf[w][4 * g + i], (uint32_t)zfree[i]);
    if (lane == 0) s_cnt[w] = running;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int k = 0; k < P.nz; ++k) tot += s_cnt[k];
        blockcounts[blockIdx.y * gridDim.x + blockIdx.x] = tot;
    }
    const int sx = blockIdx.x * WAVE + lane;
    if (w == 0 && sx < P.xy && !GVOM_DBG(P, 16)) {
        const int x = wrap_sub(sx, P.om[0], P.xy);
        unsigned long long zh = ((unsigned long long)INT_MAX << 32) | 0x3f800000ull;
        uint32_t zfu = (uint32_t)INT_MAX;
        for (int k = 0; k < P.nz; ++k) { zh = min(zh, s_zh[k][lane]); zfu = min(zfu, s_zf[k][lane]); }
        const int zo = (int)(zh >> 32), zf = (int)zfu;
        const uint32_t hb = (uint32_t)zh;
        double hval = -1000.0;
        const double xp = ((P.origin[0] + (double)x) * P.xy_res) - P.ego[0];
        const double yp = ((P.origin[1] + (double)y) * P.xy_res) - P.ego[1];
        if (xp * xp + yp * yp <= P.radius2) hval = P.ego[2] - P.ground_to_lidar_height;
        if (zo != INT_MAX)
            hval = (((double)__uint_as_float(hb) + (double)zo) + P.origin[2]) * P.z_res;
        height[(size_t)sy * P.hs + sx] = hval;
        inferred[(size_t)sy * P.hs + sx] =
            (zf != INT_MAX) ? ((double)zf + P.origin[2]) * P.z_res : -1000.0;
    }
}

// ------------------------------------------------------------------------------------------
// k_encfuse: k_encode + k_fuse1 in ONE pass, for mappers with a ONE-slot ring (buffer_size = 1: the headline 256^3
// config, c1, c2), launched right behind k_trace (gvom_capi.hip, "eager fusion").  It reads the scan's accumulators
// once and writes
//   * the ring slot exactly as k_encode does (state + compact rows; gvom.py:1154-1168, 1303-1329) and zeroes the
//     accumulators -- a later combine without a new scan, the debug reads and the statistics still find the slot;
//   * the fusion of that slot with the previous fused map (gvom.py:943-968 for the one slot, :972-997, :910-912),
//     fused state / rows / tile tags, as k_fuse1 does;
//   * the column tails height / inferred height (gvom.py:525-554).
// The two-pass form reads the accumulators, writes the slot's states, reads them back with the previous map's and
// writes the fused ones (72.5 + 56.9 MB in two launches on the 256^3 grid); here the slot's states never come back.
// The scan's window IS the fused window (the fused frame is the newest slot's, gvom.py:184), so only the previous
// map needs the shifted-window test.
//
// Workgroup = a QUAD COLUMN BLOCK: 16 sx x 4 storage rows (sy = 4Q .. 4Q+3) x all z, NW waves.  Lane (zl = lane >> 4,
// p = (lane >> 2) & 3, r = lane & 3) owns the 4 voxels sx = 16 bx + 4 p .. + 3 of row 4Q + r at level 4 (it NW + w) + zl:
// its quarter of a 4x4 accumulator patch line (a wave instruction = 4 levels x 4 whole lines = 4 x 256 B), 16 bytes
// of every state array.  A lane owns its 4 COLUMNS for all the levels its wave visits: the column tails are
// lane-private minima, combined over zl and over the waves once, at the end.  Four workgroups share a 64-voxel tile,
// so a tile that is live in a source is written and stamped whole by each of them (never "only if it has content").
// The two workgroups that share the 128-byte lines of the state arrays get neighbouring dispatch slots of ONE XCD
// (block ids 8 apart; speed only).
// ------------------------------------------------------------------------------------------
#define ENCFUSE_MAXIT 8                                    // wave iterations whose tile tags are fetched together
#define ENCFUSE_LIST 128                                   // per wave: occupied voxels listed before they are settled
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_encfuse(
    const ScanParams P, const FuseParams F, const MapDesc prev, uint32_t *hit, uint32_t *total, uint32_t *mh,
    int32_t *state, uint4 *crows, const uint32_t *__restrict__ stags, int32_t *fstate, uint4 *frows, uint32_t *ftags,
    uint32_t *blockcounts, double *height, double *inferred, uint32_t *counters, unsigned long long *host_flag, uint32_t seq)
{
    __shared__ unsigned long long s_zh[8][WAVE];            // per wave and column: min of (window z << 32 | min-height bits) over occupied voxels
    __shared__ uint32_t s_zf[8][WAVE];                      // per wave and column: lowest observed-free window z
    __shared__ uint32_t s_cnt[8];
    // per wave: the occupied voxels of the levels in hand, one entry each -- storage voxel, accumulator index, the scan's hit and
    // total there, the previous map's state, {window z | column << 10 | occupied in the scan << 16}
    __shared__ uint32_t s_eL[8][ENCFUSE_LIST], s_eA[8][ENCFUSE_LIST], s_eh[8][ENCFUSE_LIST], s_et[8][ENCFUSE_LIST], s_em[8][ENCFUSE_LIST];
    __shared__ int32_t s_ep[8][ENCFUSE_LIST];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // k_trace has completed: {seq, any-in-grid} to the spinning host, as k_encode's first thread does
        const uint32_t any = counters[GVOM_CNT_INGRID] ? 0x80000000u : 0u;
        counters[GVOM_CNT_INGRID] = 0;
        __hip_atomic_store(host_flag, ((unsigned long long)seq << 32) | any, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const int lane = threadIdx.x & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = (int)(blockDim.x >> 6);
    const int xy = P.xy, zs = P.zs, nseg = P.nseg;
    const uint32_t nbx = (uint32_t)xy >> 4;
    // dispatch slot -> column block: ids b and b + 8 share an XCD (observed round-robin placement), so the two blocks
    // that split a 32-sx run (the 128-byte lines of state / fstate) are ids 8 apart
    uint32_t M = blockIdx.x;
    if ((gridDim.x & 15u) == 0u && !(F.cpw & 1)) { const uint32_t slot = M >> 3, xcd = M & 7u; M = ((slot >> 1) << 4) + (xcd << 1) + (slot & 1u); }
    const uint32_t bx = M % nbx, Q = M / nbx;
    const int zl = lane >> 4, p4 = (lane >> 2) & 3, r = lane & 3;
    const uint32_t sx0 = bx * 16u + (uint32_t)p4 * 4u, sy = Q * 4u + (uint32_t)r;
    const uint32_t seg = sx0 >> 6;
    const uint32_t col0 = (uint32_t)(r * 16 + p4 * 4);      // this lane's first column inside the block (row-major 4 x 16)
    const int y = wrap_sub((int)sy, F.om[1], xy);
    const bool has_prev = F.has_prev != 0;
    const int dpx = has_prev ? prev.d[0] : 0, dpy = has_prev ? prev.d[1] : 0, dpz = has_prev ? prev.d[2] : 0;
    uint32_t okp = 0;                                       // bit i: the previous map's window contains column i of this lane
    {
        const bool yp = has_prev && y + dpy >= 0 && y + dpy < xy;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int xw = wrap_sub((int)sx0 + i, F.om[0], xy);
            if (yp && xw + dpx >= 0 && xw + dpx < xy) okp |= 1u << i;
        }
    }
    // (no previous map: its pointers are null -- the unconditional dummy loads below then read the slot's arrays)
    const gptr_i32 sp = has_prev ? (gptr_i32)prev.state : (gptr_i32)state;
    const gptr_u32 ptags = has_prev ? (gptr_u32)prev.tags : (gptr_u32)stags;
    const gptr_v4u prows = has_prev ? (gptr_v4u)prev.rows : (gptr_v4u)crows;
    const int niter = (zs + 4 * nw - 1) / (4 * nw);         // wave iterations (4 levels each)
    // every wave numbers its occupied voxels inside a static range of the fused compact rows (as k_fuse1)
    const uint32_t rbase = (M * (uint32_t)nw + (uint32_t)w) * (uint32_t)(niter * 256);
    uint32_t running = 0, n = 0;
    uint32_t zf[4] = {(uint32_t)INT_MAX, (uint32_t)INT_MAX, (uint32_t)INT_MAX, (uint32_t)INT_MAX};
    s_zh[w][lane] = ((unsigned long long)INT_MAX << 32) | 0x3f800000ull;      // wave-private until the tail
    // the occupied voxels listed so far, one lane each: the slot's row (k_trace's endpoint blocks left it in the slot's state),
    // the min-height accumulator and the previous map's row in ONE round trip; then the slot's compact row (k_encode's move,
    // gvom.py:1164-1168, 1303-1329), the fused row (gvom.py:910-912) and the fused state
    auto emit = [&]() {
        // (the lists were written by other lanes of this wave: their LDS stores are made visible to the whole wave first)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t base = 0; base < n; base += WAVE) {
            const uint32_t e = base + (uint32_t)lane;
            const bool on = e < n;
            const uint32_t ei = on ? e : 0u;
            const uint32_t L = s_eL[w][ei], A = s_eA[w][ei], h_ = s_eh[w][ei], t_ = s_et[w][ei], meta = s_em[w][ei];
            const int stp = on ? s_ep[w][ei] : -1;
            const bool so = on && ((meta >> 16) & 1u);
            const int32_t row_s = state[so ? L : 0u];
            const uint32_t mraw = mh[so ? A : 0u];
            const v4u rp = prows[stp >= 0 ? (uint32_t)stp : 0u];
            uint32_t hh = 0, tt = 0, mn = 0x3f800000u;
            if (so) {
                const uint32_t mbits = 0x3f800000u - mraw;
                crows[row_s] = make_uint4(h_, t_, mbits, 0u);
                mh[A] = 0u;                                    // (hit and total are zeroed with the whole line quarter, below)
                hh += h_; tt += t_; mn = min(mn, mbits);       // gvom.py:910-912
            }
            if (stp >= 0) { hh += rp.x; tt += rp.y; mn = min(mn, rp.z); }
            const unsigned long long ob = __ballot(on);
            if (on) {
                const uint32_t row = rbase + running + (uint32_t)__popcll(ob & lanemask_lt());
                frows[row] = make_uint4(hh, tt, mn, 0u);
                fstate[L] = (int32_t)row;
                atomicMin(&s_zh[w][(meta >> 10) & 63u], ((unsigned long long)(meta & 1023u) << 32) | mn);   // lowest occupied level wins
            }
            running += (uint32_t)__popcll(ob);
        }
        n = 0;
        __builtin_amdgcn_wave_barrier();                       // (nobody refills the lists before every lane has read its entry)
    };

    for (int it0 = 0; it0 < niter; it0 += ENCFUSE_MAXIT) {
        // tile tags of up to ENCFUSE_MAXIT iterations: lane l < 16 the scan's tag of (level l >> 2, row l & 3), lanes
        // 16..31 the previous map's; all fetched before the first use
        uint32_t tv_[ENCFUSE_MAXIT];
#pragma unroll
        for (int k = 0; k < ENCFUSE_MAXIT; ++k) {
            const int szt = ((it0 + k) * nw + w) * 4 + ((lane >> 2) & 3);
            const bool okt = it0 + k < niter && lane < 32 && szt < zs && (lane < 16 || has_prev);
            const uint32_t ti = ((Q * 4u + (uint32_t)(lane & 3)) * (uint32_t)zs + (uint32_t)(okt ? szt : 0)) * (uint32_t)nseg + seg;
            tv_[k] = okt ? (lane < 16 ? stags[ti] : ptags[ti]) : 0u;
        }
        // iteration k's mask in lane k: bits 0..15 scan-live (level, row), bits 16..31 the previous map's
        uint32_t lmv = 0u;
#pragma unroll
        for (int k = 0; k < ENCFUSE_MAXIT; ++k) {
            const uint32_t b_ = (uint32_t)__ballot(it0 + k < niter && lane < 32 && tv_[k] != 0u && tv_[k] == (lane < 16 ? P.epoch : prev.epoch));
            if (lane == k) lmv = b_;
        }
        const int kend = min(ENCFUSE_MAXIT, niter - it0);
#pragma unroll 1
        for (int k = 0; k < kend; ++k) {
            const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)lmv, k);
            if (m == 0u) continue;                           // wave-uniform: the 4 levels are dead in both sources
            const int sz = ((it0 + k) * nw + w) * 4 + zl;
            const bool zok = sz < zs;
            const bool live_s = zok && ((m >> (zl * 4 + r)) & 1u);
            const bool live_p = zok && ((m >> (16 + zl * 4 + r)) & 1u);
            const uint32_t A0 = acc_idx((int)sx0, (int)sy, zok ? sz : 0, zs, P.sxq);
            const uint32_t L0 = (sy * (uint32_t)zs + (uint32_t)(zok ? sz : 0)) * (uint32_t)xy + sx0;
            // unconditional 16-byte loads (a dead tile reads a valid dummy address): all in flight together
            const v4u hv = *(gptr_v4u)(hit + (live_s ? A0 : (uint32_t)(4 * lane)));
            const v4u tv = *(gptr_v4u)(total + (live_s ? A0 : (uint32_t)(4 * lane)));
            const v4i pv = *(gptr_v4i)(sp + (live_p ? L0 : (uint32_t)(4 * lane)));
            const int zw = wrap_sub(zok ? sz : 0, F.om[2], zs);
            const bool zp_in = live_p && zw + dpz >= 0 && zw + dpz < zs;
            const uint32_t h[4] = {hv.x, hv.y, hv.z, hv.w};
            const uint32_t t[4] = {tv.x + hv.x, tv.y + hv.y, tv.z + hv.z, tv.w + hv.w};   // passes + the endpoints' own count (endpoint_commit)
            const int b[4] = {pv.x, pv.y, pv.z, pv.w};
            const uint32_t any_h = live_s ? (h[0] | h[1] | h[2] | h[3]) : 0u;
            const uint32_t any_t = live_s ? (tv.x | tv.y | tv.z | tv.w) : 0u;
            int st[4], c[4], stp[4];
            uint32_t occ = 0, socc = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                st[i] = live_s ? -(int32_t)t[i] - 1 : -1;    // gvom.py:1160 (occupied voxels keep the row k_trace left there)
                if (live_s && h[i] > 0u) socc |= 1u << i;
                stp[i] = (zp_in && ((okp >> i) & 1u)) ? b[i] : -1;
                c[i] = -1;
                if ((socc >> i) & 1u) occ |= 1u << i;                                       // gvom.py:963
                else if (st[i] < -1) c[i] = add_free(c[i], st[i] + 1);                      // gvom.py:967
                if (!((occ >> i) & 1u)) {
                    if (stp[i] >= 0 && c[i] >= -11) occ |= 1u << i;                         // gvom.py:992
                    else if (stp[i] < -1) c[i] = add_free(c[i], stp[i] + 1);                // gvom.py:996
                }
            }
            if (live_s) {
                if (socc == 0u) *reinterpret_cast<int4 *>(state + L0) = make_int4(st[0], st[1], st[2], st[3]);
                else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (!((socc >> i) & 1u)) state[L0 + i] = st[i];
                }
                if (any_t) *reinterpret_cast<uint4 *>(total + A0) = make_uint4(0, 0, 0, 0);
                if (any_h) *reinterpret_cast<uint4 *>(hit + A0) = make_uint4(0, 0, 0, 0);
            }
            if (live_s || live_p) {
                // (an occupied voxel's word belongs to emit(), where ANOTHER lane stores the voxel's row: the owner never writes
                // it, so no ordering between two lanes' stores to one address is relied on -- ADVICE r5)
                if (occ == 0u) *reinterpret_cast<int4 *>(fstate + L0) = make_int4(c[0], c[1], c[2], c[3]);
                else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (!((occ >> i) & 1u)) fstate[L0 + i] = c[i];
                }
                if ((sx0 & 63u) == 0u) ftags[(sy * (uint32_t)zs + (uint32_t)sz) * (uint32_t)nseg + seg] = F.epoch;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (!((occ >> i) & 1u) && c[i] < -1) zf[i] = min(zf[i], (uint32_t)zw);  // gvom.py:551
            }
            if (__builtin_amdgcn_readfirstlane((int)(__ballot(occ != 0u) != 0ull))) {       // rare: occupied voxels in these 4 levels
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool o = (occ >> i) & 1u;
                    const unsigned long long bm = __ballot(o);
                    if (o) {
                        const uint32_t at = n + (uint32_t)__popcll(bm & lanemask_lt());
                        s_eL[w][at] = L0 + (uint32_t)i; s_eA[w][at] = A0 + (uint32_t)i; s_eh[w][at] = h[i]; s_et[w][at] = t[i];
                        s_ep[w][at] = stp[i];
                        s_em[w][at] = (uint32_t)zw | ((col0 + (uint32_t)i) << 10) | (((socc >> i) & 1u) << 16);
                    }
                    n += (uint32_t)__popcll(bm);
                    if (n > ENCFUSE_LIST - WAVE) emit();       // the next batch (<= 64 voxels) might not fit
                }
            }
        }
    }
    if (n) emit();

    // ---- column tails: over the 4 level groups of the wave (lanes l, l ^ 16, l ^ 32, l ^ 48), then over the waves
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        zf[i] = min(zf[i], (uint32_t)__shfl_xor((int)zf[i], 16));
        zf[i] = min(zf[i], (uint32_t)__shfl_xor((int)zf[i], 32));
    }
    if (zl == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s_zf[w][col0 + i] = zf[i];
    }
    if (lane == 0) s_cnt[w] = running;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int k = 0; k < nw; ++k) tot += s_cnt[k];
        blockcounts[M] = tot;
    }
    if (w == 0) {                                            // lane = column: row 4Q + (lane >> 4), sx = 16 bx + (lane & 15)
        const int csx = (int)(bx * 16u) + (lane & 15), csy = (int)(Q * 4u) + (lane >> 4);
        unsigned long long zhm = ((unsigned long long)INT_MAX << 32) | 0x3f800000ull;
        uint32_t zfm = (uint32_t)INT_MAX;
        for (int k = 0; k < nw; ++k) { zhm = min(zhm, s_zh[k][lane]); zfm = min(zfm, s_zf[k][lane]); }
        const int x = wrap_sub(csx, F.om[0], xy), yy = wrap_sub(csy, F.om[1], xy);
        const int zo = (int)(zhm >> 32), zfi = (int)zfm;
        double hval = -1000.0;
        const double xp = ((F.origin[0] + (double)x) * F.xy_res) - F.ego[0];
        const double yp = ((F.origin[1] + (double)yy) * F.xy_res) - F.ego[1];
        if (xp * xp + yp * yp <= F.radius2) hval = F.ego[2] - F.ground_to_lidar_height;          // gvom.py:531-534
        if (zo != INT_MAX) hval = (((double)__uint_as_float((uint32_t)zhm) + (double)zo) + F.origin[2]) * F.z_res;   // gvom.py:536-540
        height[(size_t)csy * F.hs + csx] = hval;
        inferred[(size_t)csy * F.hs + csx] = (zfi != INT_MAX) ? ((double)zfi + F.origin[2]) * F.z_res : -1000.0;   // gvom.py:544-554
    }
}

// sum of the per-workgroup occupied-voxel counts of k_fuse -> host-mapped memory
__device__ __forceinline__ void publish_block_counts(const uint32_t *blockcounts, int nblocks,
                                                     volatile unsigned long long *host_counter,
                                                     unsigned long long *s_red, int tid, int nthreads)
{
    unsigned long long a = 0;
    for (int i = tid; i < nblocks; i += nthreads) a += blockcounts[i];
    s_red[tid] = a;
    __syncthreads();
    for (int o = nthreads >> 1; o > 0; o >>= 1) {
        if (tid < o) s_red[tid] += s_red[tid + o];
        __syncthreads();
    }
    // (system scope: k_map2d publishes its completion before the kernel ends, the count must have left the L2 by then)
    if (tid == 0) __hip_atomic_store((unsigned long long *)host_counter, s_red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_publish_count(const uint32_t *blockcounts, int nblocks, unsigned long long *host_counter,
                                unsigned long long *dev_counter)
{
    __shared__ unsigned long long s_red[256];
    publish_block_counts(blockcounts, nblocks, host_counter, s_red, threadIdx.x, 256);
    if (threadIdx.x == 0) *dev_counter = s_red[0];
}

// Store that leaves the GPU now (system scope: written through L2) instead of staying in the
// write-back L2 until the end-of-kernel release.  (Measured: the 1.3 MB of returned maps cost
// ~23 us of PCIe time either way -- the link, not the issue order, is the limit.)
template <typename V>
__device__ __forceinline__ void st_sys(V *p, V v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("" ::: "memory");                       // keep the compiler from sinking it to the kernel's end
}

// ------------------------------------------------------------------------------------------
// k_map2d: every 2-D output of combine_maps from height/inferred height, one lane per cell.
//
// Workgroup = an 8 (x) x 32 (y) tile of WINDOW cells.  The height map of the tile plus a 15-cell halo
// (the reach of __guess_height) is staged once in LDS together with two sets of validity
// bitmasks (one 64-bit word per tile row over x, one per tile column over y).  The reference's
// expanding-ring search (up to 15 rings x 4 directions x 30 cells of dependent global loads
// per cell) becomes at most 60 LDS word reads + count-trailing-zeros per cell.
// Internal maps are [sy][sx] storage order.  The four returned maps are transposed through LDS
// and written in the reference's [x][y] window order as runs of 32 consecutive y (128/256 B)
// -- straight into host-mapped memory on the single-GPU path (no D2H copy command) -- or left
// in storage order for sharded runs.
// ------------------------------------------------------------------------------------------
#define M2_HALO 15

// TX x TY window cells per workgroup (256 threads).  YX = false: 8 x 32 tile, the four returned
// maps in row-major [x][y] order (transposed through LDS, runs of 32 consecutive y).  YX = true:
// 32 x 8 tile, the maps in [y][x] memory order -- the SAME arrays seen as column-major (numpy:
// Fortran-ordered, indexed [x, y]); a lane's 32 neighbours in x form 128/256-byte runs without a
// transpose, and each map is stored as soon as it is known (all global loads come first: vmcnt is
// in-order, a load behind a host-memory store would wait for the store to drain over PCIe).
//
// TWO WAVE ROLES (512 threads = the tile's 256 cells twice).  The kernel is a latency chain (tile
// staging, then three dependent global round trips for the positive-obstacle density) followed by
// 1.31 MB of stores into host memory (23 us of PCIe at 57 GB/s): when every wave walks the whole
// chain, all stores are issued in the kernel's last microseconds and the link idles until then.
// Waves 0-3 therefore compute ONLY slope / roughness -- LDS data, no global load -- and store the
// f64 roughness map (40 % of the bytes) while waves 4-7 are still waiting for their density loads;
// those then store visibility / positive / negative.  The slope-obstacle flag crosses through LDS.
template <bool GATHERED_POS, bool YX>
__global__ __launch_bounds__(512) void k_map2d(const Map2dParams P, const int32_t *__restrict__ fstate,
                                               const uint32_t *__restrict__ ftags,
                                               const uint4 *__restrict__ frows,
                                               const double *__restrict__ height,
                                               const double *__restrict__ inferred,
                                               double *slope_x, double *slope_y, double *rough,
                                               double *guessed, int32_t *out_pos, int32_t *out_neg,
                                               double *out_rough, int32_t *out_vis,
                                               const uint32_t *blockcounts, int nblocks,
                                               unsigned long long *host_counter)
{
    constexpr int M2_TX = YX ? 32 : 8, M2_TY = YX ? 8 : 32;
    constexpr int M2_W = M2_TX + 2 * M2_HALO, M2_H = M2_TY + 2 * M2_HALO;      // 62 x 38 (YX) or 38 x 62
    __shared__ double ht[M2_H][M2_W];
    __shared__ unsigned long long rowm[M2_H];
    __shared__ unsigned long long colm[M2_W];
    __shared__ int o_pos[YX ? 1 : M2_TX][M2_TY + 1], o_neg[YX ? 1 : M2_TX][M2_TY + 1], o_vis[YX ? 1 : M2_TX][M2_TY + 1];
    __shared__ double o_rgh[YX ? 1 : M2_TX][M2_TY + 1];

    __shared__ unsigned char s_steep[256];                   // slope >= threshold (role A -> role B)
    const int xy = P.xy;
    const int tid = threadIdx.x;
    const int cell = tid & 255;                              // the tile's cell this thread works on
    const bool role_b = tid >= 256;                          // waves 4-7: density, guess height, i32 maps
    const int tx = cell & (M2_TX - 1), ty = cell / M2_TX;
    const int lane = tid & 63, wv = tid >> 6;
    const int X0 = blockIdx.x * M2_TX, Y0 = blockIdx.y * M2_TY;
    if (host_counter && blockIdx.x == 0 && blockIdx.y == 0 && !GVOM_DBG(P, 16)) {
        // k_fuse is complete: publish the fused occupied-voxel count (host-mapped memory)
        __shared__ unsigned long long s_red[512];
        publish_block_counts(blockcounts, nblocks, host_counter, s_red, tid, 512);
    }

    // ---- stage the tile (+halo) and its row masks ------------------------------------------
    {   // all of a wave's rows are fetched before the first use: independent, unconditional loads
        // (out-of-window cells read a valid dummy address and are replaced by -1000)
        constexpr int NR = (M2_H + 7) / 8;
        double v[NR];
        bool inw[NR];
        const int gx = X0 - M2_HALO + lane;
        const int sxh = wrap_add((gx >= 0 && gx < xy) ? gx : 0, P.om[0], xy);
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int r = wv + 8 * k, gy = Y0 - M2_HALO + r;
            inw[k] = r < M2_H && lane < M2_W && gy >= 0 && gy < xy && gx >= 0 && gx < xy;
            v[k] = height[inw[k] ? (size_t)wrap_add(gy, P.om[1], xy) * P.hs + sxh : (size_t)0];
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int r = wv + 8 * k;
            const double vv = inw[k] ? v[k] : -1000.0;
            const unsigned long long m = __ballot(vv > -1000);
            if (r < M2_H) {
                if (lane < M2_W) ht[r][lane] = vv;
                if (lane == 0) rowm[r] = m;
            }
        }
    }
    __syncthreads();
    if (tid < M2_W) {
        unsigned long long m = 0ull;
        for (int r = 0; r < M2_H; ++r) m |= ((rowm[r] >> tid) & 1ull) << r;
        colm[tid] = m;
    }
    __syncthreads();

    const int x0 = X0 + tx, y0 = Y0 + ty;                            // window cell
    bool mine = x0 < xy && y0 < xy;
    const int sx0 = wrap_add(mine ? x0 : 0, P.om[0], xy), sy0 = wrap_add(mine ? y0 : 0, P.om[1], xy);
    mine = mine && sy0 >= P.y_lo && sy0 < P.y_hi;                      // else: another rank's row
    const int lx = tx + M2_HALO, ly = ty + M2_HALO;
    const size_t c_out = (size_t)y0 * xy + x0;                         // YX: [y][x] (column-major [x, y])
    const bool wr = !GVOM_DBG(P, 1);
    double h00 = -1000.0, inf00 = 0.0, rv = -1.0;
    int dens_pos = 0, pos = 0, negv = 0, visv = 0;           // dens_pos: positive-obstacle density x100 (gvom.py:489-521)
    int8_t *const occ = reinterpret_cast<int8_t *>(out_pos);
    const size_t n2 = (size_t)xy * xy;
    const size_t c_yx = (size_t)sy0 * xy + sx0;
    if (mine) h00 = ht[ly][lx];
    if (!role_b) {
    if (mine) {
    // visibility needs only the staged height: it leaves with the first stores (gvom.py:414-422)
    // OCC: instead of the four maps, the five int8 nav_msgs/OccupancyGrid.data arrays the ROS node
    // derives from them (gvom_ros.py:141-165), planes [hard | soft | certainty | negative | roughness]
    visv = h00 > -1000 ? 1 : 0;
    if (YX && wr) { if (P.occ) st_sys(&occ[2 * n2 + c_out], (int8_t)(visv * 100)); else st_sys(&out_vis[c_out], visv); }
    if (!YX) o_vis[tx][ty] = visv;
    // ---- role A: slope / roughness: 3x3 least-squares plane (gvom.py:665-734) ---------------------
    // cells outside the window hold -1000 in the tile, i.e. are skipped exactly like the
    // reference's clipped ranges; iteration order is x outer / y inner as in the reference.
    double sxv = 0.0, syv = 0.0;
    {
        int n_good = 0;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
                if (ht[ly + dy][lx + dx] > -1000) ++n_good;
        if (n_good >= 3 && !GVOM_DBG(P, 4)) {
            double mean_x = 0.0, mean_y = 0.0, mean_z = 0.0;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy) {
                    const double hz = ht[ly + dy][lx + dx];
                    if (hz > -1000) {
                        mean_x += (double)(x0 + dx) * P.xy_res;
                        mean_y += (double)(y0 + dy) * P.xy_res;
                        mean_z += hz;
                    }
                }
            const double fi = (double)n_good;
            mean_x /= fi; mean_y /= fi; mean_z /= fi;
            double cxx = 0.0, cxy = 0.0, cxz = 0.0, cyy = 0.0, cyz = 0.0;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy) {
                    const double hz = ht[ly + dy][lx + dx];
                    if (hz > -1000) {
                        const double px = (double)(x0 + dx) * P.xy_res, py = (double)(y0 + dy) * P.xy_res;
                        cxx += (px - mean_x) * (px - mean_x);
                        cxy += (px - mean_x) * (py - mean_y);
                        cxz += (px - mean_x) * (hz - mean_z);
                        cyy += (py - mean_y) * (py - mean_y);
                        cyz += (py - mean_y) * (hz - mean_z);
                    }
                }
            const double det = cxx * cyy - cxy * cxy;
            if (det != 0.0) {
                double a0 = (cyy * cxz - cxy * cyz) / det;
                double a1 = (cxx * cyz - cxy * cxz) / det;
                const double m = sqrt((a0 * a0 + a1 * a1) + 1.0);
                a0 /= m; a1 /= m;
                double err = 0.0;
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
                    for (int dy = -1; dy <= 1; ++dy) {
                        const double hz = ht[ly + dy][lx + dx];
                        if (hz > -1000) {
                            const double px = (double)(x0 + dx) * P.xy_res, py = (double)(y0 + dy) * P.xy_res;
                            const double e = (hz - mean_z) - (a0 * (px - mean_x) + a1 * (py - mean_y));
                            err += e * e;
                        }
                    }
                err /= fi;
                if (err > 0) err = log(err);
                rv = err;
                sxv = atan2(a0, 1.0 / m);
                syv = atan2(a1, 1.0 / m);
            }
        }
    }
    slope_x[c_yx] = sxv; slope_y[c_yx] = syv; rough[c_yx] = rv;
    if (YX && wr) {
        if (P.occ) {
            // ((clip(r, min, max) + min) / (max - min)) * 100 in f64 as written (it ADDS min), then numpy's
            // float64 -> int8 cast: truncate to a 32-bit integer, keep the low byte (gvom_ros.py:162-163)
            const double rr = ((py_maxd(py_mind(rv, P.occ_max_rough), P.occ_min_rough) + P.occ_min_rough) / (P.occ_max_rough - P.occ_min_rough)) * 100.0;
            const int32_t ri = (fabs(rr) < 2147483648.0) ? (int32_t)rr : INT_MIN;       // x86 cvttsd2si: out of range / NaN -> INT_MIN
            st_sys(&occ[4 * n2 + c_out], (int8_t)(uint8_t)(uint32_t)ri);
        } else st_sys(&out_rough[c_out], rv);
    }
    s_steep[cell] = (sqrt(sxv * sxv + syv * syv) >= P.slope_thr) ? 1 : 0;   // gvom.py:489-521, used by role B
    if (!YX) o_rgh[tx][ty] = rv;
    }   // mine
    } else {
    if (mine) {
    // ---- role B, before the barrier: every global load of the cell.  Stores into host-mapped memory
    // are acknowledged slowly and vmcnt is in-order: a load issued after one would stall the wave
    // until the store has drained over PCIe, so this role stores nothing before its loads are back.
    inf00 = inferred[(size_t)sy0 * P.hs + sx0];
    if (GATHERED_POS) {
        // sharded runs: the slab owner computed the density (k_posdens), all-gathered with the heights
        dens_pos = (int)height[(size_t)sy0 * P.hs + 2 * (size_t)xy + sx0];
    } else {
        const double fmin = floor(((h00 + P.pos_thr) / P.z_res) - P.origin_z) + 1.0;
        const double fmax = floor(((h00 + P.robot_height) / P.z_res) - P.origin_z);
        if (fmin >= 0 && fmin < (double)P.zs && fmax >= 0 && fmax < (double)P.zs && !GVOM_DBG(P, 2)) {
            const int zmin = (int)fmin, zmax = (int)fmax;
            double density = 0.0, nn = 0.0;
            // 8 levels per round: tags, then states, then counts -- three dependent round trips
            // per round instead of three per level (unconditional loads, dummy index when dead)
            for (int zb = zmin; zb <= zmax; zb += 8) {
                uint32_t rz[8], tg[8], hc[8], tc[8];
                int32_t row[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int z = (zb + k <= zmax) ? zb + k : zmin;
                    rz[k] = (uint32_t)sy0 * P.zs + (uint32_t)wrap_add(z, P.om[2], P.zs);
                    tg[k] = ftags[rz[k] * P.nseg + (sx0 >> 6)];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool live = zb + k <= zmax && tg[k] == P.epoch;       // live tile
                    row[k] = fstate[live ? rz[k] * xy + sx0 : 0u];
                    if (!live) row[k] = -1;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t r = row[k] >= 0 ? (uint32_t)row[k] : 0u;
                    const uint2 ht = *reinterpret_cast<const uint2 *>(frows + r);
                    hc[k] = ht.x; tc[k] = ht.y;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (row[k] >= 0 && (int32_t)hc[k] > 10) { nn += (double)(int32_t)tc[k]; density += (double)(int32_t)hc[k]; }
            }
            if (nn > 0.0) density /= nn;
            dens_pos = (int)(density * 100);
        }
    }
    }   // mine
    }
    __syncthreads();
    if (role_b && mine) {
    visv = h00 > -1000 ? 1 : 0;                    // gvom.py:414-422 (stored by role A)
    pos = s_steep[cell] ? 100 : dens_pos;                    // gvom.py:489-521 (slope test done by role A)
    if (YX && wr) {
        if (P.occ) st_sys(&occ[1 * n2 + c_out], (int8_t)(((double)pos <= P.occ_density_thr && pos > 0) ? 100 : 0));   // soft, :146
        else st_sys(&out_pos[c_out], pos);
    }

    // ---- guess height (gvom.py:558-661), typos at :581 and :655 reproduced ---------------
    // ring i, direction +x: first valid cell of column x0+i for dy in [-i, i)   -> colm bit-scan
    //                   -x: column x0-i, dy in [-i+1, i];  +y: row y0+i, dx in [-i+1, i];
    //                   -y: row y0-i, dx in [-i, i)                               (gvom.py:588-638)
    double dh_out = 0.0;
    if (!(h00 > -1000 || inf00 == -1000.0) && !GVOM_DBG(P, 8)) {
        bool x_p_done = false, x_n_done = false, y_p_done = false, y_n_done = false;
        // position of each direction's first valid cell in the tile (row, col); -1: none.  The four
        // masks of a ring are read together and the heights only after the search: one LDS round
        // trip per ring instead of eight dependent ones.
        int rxp = -1, cxp = 0, rxn = -1, cxn = 0, ryp = -1, cyp = 0, ryn = -1, cyn = 0;
        int i = 0;
        while (i < 15 && !(x_n_done && x_n_done && y_p_done && y_n_done)) {
            i += 1;
            const unsigned long long span = (1ull << (2 * i)) - 1ull;
            const unsigned long long mxp = (colm[lx + i] >> (ly - i)) & span;
            const unsigned long long mxn = (colm[lx - i] >> (ly - i + 1)) & span;
            const unsigned long long myp = (rowm[ly + i] >> (lx - i + 1)) & span;
            const unsigned long long myn = (rowm[ly - i] >> (lx - i)) & span;
            if (!x_p_done) {
                if (x0 + i < xy) {
                    if (mxp) { rxp = ly - i + __ffsll((long long)mxp) - 1; cxp = lx + i; x_p_done = true; }
                } else x_p_done = true;
            }
            if (!x_n_done) {
                if (x0 - i >= 0) {
                    if (mxn) { rxn = ly - i + 1 + __ffsll((long long)mxn) - 1; cxn = lx - i; x_n_done = true; }
                } else x_n_done = true;
            }
            if (!y_p_done) {
                if (y0 + i < xy) {
                    if (myp) { ryp = ly + i; cyp = lx - i + 1 + __ffsll((long long)myp) - 1; y_p_done = true; }
                } else y_p_done = true;
            }
            if (!y_n_done) {
                if (y0 - i >= 0) {
                    if (myn) { ryn = ly - i; cyn = lx - i + __ffsll((long long)myn) - 1; y_n_done = true; }
                } else y_n_done = true;
            }
        }
        const double hxp = ht[max(rxp, 0)][cxp], hxn = ht[max(rxn, 0)][cxn], hyp = ht[max(ryp, 0)][cyp], hyn = ht[max(ryn, 0)][cyn];
        const double x_ph = rxp >= 0 ? hxp : -1000.0, x_nh = rxn >= 0 ? hxn : -1000.0;
        const double y_ph = ryp >= 0 ? hyp : -1000.0, y_nh = ryn >= 0 ? hyn : -1000.0;
        double min_h = 1000.0, max_h = inf00;
        if (x_ph > -1000) { min_h = py_mind(x_ph, min_h); max_h = py_maxd(x_ph, max_h); }
        if (x_nh > -1000) { min_h = py_mind(x_nh, min_h); max_h = py_maxd(x_nh, max_h); }
        if (y_ph > -1000) { min_h = py_mind(y_ph, min_h); max_h = py_maxd(y_ph, max_h); }
        if (x_nh > -1000) { min_h = py_mind(y_nh, min_h); max_h = py_maxd(y_nh, max_h); }
        const double dh = max_h - min_h;
        if (dh > 0) dh_out = dh;
    }
    guessed[c_yx] = dh_out;
    negv = dh_out > P.neg_thr ? 100 : 0;           // gvom.py:479-485
    if (YX && wr) {
        if (P.occ) {
            st_sys(&occ[3 * n2 + c_out], (int8_t)negv);                                                    // negative, :157
            st_sys(&occ[0 * n2 + c_out], (int8_t)max((double)pos > P.occ_density_thr ? 100 : 0, negv));   // hard, :141
        } else st_sys(&out_neg[c_out], negv);
    }

    if (!YX) { o_pos[tx][ty] = pos; o_neg[tx][ty] = negv; }
    }   // role B, mine
    if (!YX) {
        __syncthreads();
        const int ox = tid >> 5, oy = tid & 31;              // 32 consecutive lanes -> 32 consecutive y
        const int gx = X0 + ox, gy = Y0 + oy;
        if (tid < 256 && gx < xy && gy < xy && !GVOM_DBG(P, 1)) {
            const size_t c_xy = (size_t)gx * xy + gy;
            out_pos[c_xy] = o_pos[ox][oy]; out_neg[c_xy] = o_neg[ox][oy];
            out_vis[c_xy] = o_vis[ox][oy]; out_rough[c_xy] = o_rgh[ox][oy];
        }
    }
    if (P.done_flag) {
        // The maps lie in host memory once every wave's stores have been acknowledged (s_waitcnt vmcnt(0): system-scope
        // stores are written through, and what still sat in this XCD's L2 leaves with the workgroup's one release); the flag
        // store of the last workgroup travels the same ordered path behind them.
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
        __syncthreads();
        if (tid == 0) {
            if (!YX) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");       // (plain stores of the row-major form: write back, system scope)
            const uint32_t arrived = __hip_atomic_fetch_add(P.done_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (arrived + 1u == gridDim.x * gridDim.y) {
                __hip_atomic_store(P.done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(P.done_flag, (unsigned long long)P.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Optional per-voxel statistics.  Off the north-star path; enabled per handle.
// Float accumulation order is unspecified on a GPU (as in the reference's f64 atomics), so these
// results match the reference to a tolerance, not bit for bit.
// ------------------------------------------------------------------------------------------

// k_stats: gvom.py:1172-1220 + :1234-1285.  The reference adds every return to each OCCUPIED
// voxel of its (2*xy_e+1)^2 x (2*z_e+1) neighbourhood (up to 270 f64 atomics per point).  Here a
// return whose own voxel lies in the grid adds the raw moments of its in-voxel position l in
// [0,1)^3 to ITS OWN voxel only (10 atomics): base[row] = {Sx, Sy, Sz, Sxx, Sxy, Sxz, Syy, Syz,
// Szz, n}; k_stats_gather then gives every occupied voxel the moments of its neighbours, shifted
// by the voxel offset d (l' = l + d):  S l' = S l + n d,  S l'l'^T = S l l^T + d (S l)^T +
// (S l) d^T + n d d^T -- the same sums, 27x fewer atomics.  Returns whose own voxel is outside the
// grid (they can still touch border voxels) and slab-sharded handles (a neighbour's moments may
// live on another rank) use the direct form into `sums`.  Both buffers are zeroed at row claim.
// Runs after k_encode: state >= 0 in a live tile identifies an occupied voxel and its row.
template <typename T>
__global__ __launch_bounds__(256) void k_stats(const ScanParams P, const T *__restrict__ world, long n,
                                               const int32_t *__restrict__ state,
                                               const uint32_t *__restrict__ tags, int xy_e, int z_e,
                                               double *base, double *sums, int direct_only)
{
    // own-voxel moments of a wave's 64 returns, staged for the transposed adds below
    __shared__ double s_m[4][64][10];
    __shared__ int32_t s_row[4][64];
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    int32_t own = -1;
    double mom[10];
    if (i < n) {
        const T x = world[3 * i + 0], y = world[3 * i + 1], z = world[3 * i + 2];
        const T d2 = (x * x + y * y) + z * z;
        const double ax = (double)x / P.xy_res - P.origin[0];
        const double ay = (double)y / P.xy_res - P.origin[1];
        const double az = (double)z / P.z_res - P.origin[2];
        const double bx = floor(ax), by = floor(ay), bz = floor(az);
        if (!((double)d2 < P.min_d2) && fabs(bx) < 1e9 && fabs(by) < 1e9 && fabs(bz) < 1e9) {
            const int xb = (int)bx, yb = (int)by, zb = (int)bz;
            const bool base_in = xb >= 0 && xb < P.xy && yb >= 0 && yb < P.xy && zb >= 0 && zb < P.zs;
            if (base_in && !direct_only) {
                const int sx = wrap_add(xb, P.om[0], P.xy), sy = wrap_add(yb, P.om[1], P.xy), sz = wrap_add(zb, P.om[2], P.zs);
                own = state[((uint32_t)sy * P.zs + sz) * P.xy + sx];      // >= 0: this voxel has a hit
                const double lx = ax - bx, ly = ay - by, lz = az - bz;
                mom[0] = lx; mom[1] = ly; mom[2] = lz;
                mom[3] = lx * lx; mom[4] = lx * ly; mom[5] = lx * lz; mom[6] = ly * ly; mom[7] = ly * lz; mom[8] = lz * lz;
                mom[9] = 1.0;
            } else {
                for (int xi = xb - xy_e; xi <= xb + xy_e; ++xi) {
                    if (xi < 0 || xi >= P.xy) continue;
                    for (int yi = yb - xy_e; yi <= yb + xy_e; ++yi) {
                        if (yi < 0 || yi >= P.xy) continue;
                        const int sy = wrap_add(yi, P.om[1], P.xy);
                        if (sy < P.sy_lo || sy >= P.sy_hi) continue;
                        const int sx = wrap_add(xi, P.om[0], P.xy);
                        for (int zi = zb - z_e; zi <= zb + z_e; ++zi) {
                            if (zi < 0 || zi >= P.zs) continue;
                            const int sz = wrap_add(zi, P.om[2], P.zs);
                            const uint32_t rz = (uint32_t)sy * P.zs + sz;
                            if (tags[rz * P.nseg + (sx >> 6)] != P.epoch) continue;      // untouched tile
                            const int32_t row = state[rz * P.xy + sx];
                            if (row < 0) continue;
                            const double lx = ax - (double)xi, ly = ay - (double)yi, lz = az - (double)zi;
                            double *m = sums + (size_t)row * 10;
                            unsafeAtomicAdd(m + 0, lx); unsafeAtomicAdd(m + 1, ly); unsafeAtomicAdd(m + 2, lz);
                            unsafeAtomicAdd(m + 3, lx * lx); unsafeAtomicAdd(m + 4, lx * ly); unsafeAtomicAdd(m + 5, lx * lz);
                            unsafeAtomicAdd(m + 6, ly * ly); unsafeAtomicAdd(m + 7, ly * lz); unsafeAtomicAdd(m + 8, lz * lz);
                            unsafeAtomicAdd(m + 9, 1.0);
                        }
                    }
                }
            }
        }
    }
    // The adds are memory-side atomics and run at the REQUEST rate (one per 64-B line an instruction touches): ten
    // instructions with 64 lanes in 64 different rows are 640-1280 requests per wave.  Transposed -- lane = (return, moment),
    // six returns per instruction, a row's ten moments in ONE 128-byte block -- they are two requests per return.
    s_row[wv][lane] = own;
    if (own >= 0) {
#pragma unroll
        for (int k = 0; k < 10; ++k) s_m[wv][lane][k] = mom[k];
    }
    __syncthreads();
    if (lane < 60) {
        const int c = lane % 10, q = lane / 10;
#pragma unroll 1
        for (int it = 0; it < 11; ++it) {
            const int pnt = it * 6 + q;
            if (pnt < 64) {
                const int32_t r = s_row[wv][pnt];
                if (r >= 0) unsafeAtomicAdd(base + (size_t)r * GVOM_BASE_PITCH + c, s_m[wv][pnt][c]);
            }
        }
    }
}

// k_stats_gather: a QUARTER wave (16 lanes) per occupied voxel (= compact row; rowvox[row] is its voxel).  A wave looks at
// GATHER_CPW candidate rows at once (a row is the index of one of the voxel's returns: near the sensor most candidates are not in
// use, far away -- one return per voxel -- all of them are) and takes the used ones FOUR at a time: every used row is a chain of
// dependent round trips (neighbours' tags + states, their moments, the row's own sums), and the kernel's time is its longest
// wave -- 16 used candidates at two per turn were 8 turns of ~4 us while a thousand SIMDs idled (round 6: SQ_WAVE_CYCLES says 0.9
// resident waves per SIMD over its 46 us; 64 candidates per wave took 91 us, a wave per candidate 40).  The lanes of a quarter
// take the neighbourhood's voxels two at a time (lane <-> neighbour offset; both neighbours' loads in flight together), shift
// their own-voxel moments by the offset (see k_stats) and a butterfly reduction sums them; the quarter's first lane adds the
// directly accumulated part (fetched BEFORE the neighbourhood, it does not depend on it) and turns the raw moments into the
// reference's per-scan metrics layout: mean xyz, population covariance xx xy xz yy yz zz (gvom.py:1224-1230, 1289-1299), count.
#define GATHER_CPW 16
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_stats_gather(const ScanParams P, const int32_t *__restrict__ state,
                                                      const uint32_t *__restrict__ tags, int xy_e, int z_e,
                                                      const double *__restrict__ base, double *sums,
                                                      const uint32_t *__restrict__ rowvox,
                                                      uint32_t nrows, int direct_only)
{
    const int lane = threadIdx.x & (WAVE - 1), ql = lane & 15, quarter = lane >> 4;
    const uint32_t wid = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t nw = (gridDim.x * blockDim.x) >> 6;
    const int wx = 2 * xy_e + 1, wz = 2 * z_e + 1, nb = wx * wx * wz;
    // a compact row is the index of one of the voxel's returns (k_trace): row `row` is in use iff the
    // return claimed a voxel (rowvox != ~0, reset per scan) and that voxel's state still names it
    // candidates of a wave: rows wid + lane * waves of each pass (NOT 16 consecutive rows: consecutive returns of a beam far from the
    // sensor are one voxel each -- 16 used candidates -- and near it all one voxel; strided, every wave gets its share of both)
    for (uint32_t pass0 = 0; pass0 < nrows; pass0 += nw * GATHER_CPW) {
        const uint32_t cand = pass0 + (uint32_t)lane * nw + wid;
        const uint32_t Lc = (lane < GATHER_CPW && cand < nrows) ? rowvox[cand] : 0xFFFFFFFFu;
        const bool used = Lc != 0xFFFFFFFFu && state[Lc] == (int32_t)cand;
        unsigned long long todo = lanes(used);
        while (todo != 0ull) {                                         // wave-uniform: four used rows per turn
            int ks[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ks[q] = -1;
                if (todo != 0ull) { ks[q] = __ffsll((long long)todo) - 1; todo &= todo - 1ull; }
            }
            const int k = quarter == 0 ? ks[0] : (quarter == 1 ? ks[1] : (quarter == 2 ? ks[2] : ks[3]));
            const bool have = k >= 0;
            const uint32_t row = pass0 + (uint32_t)(have ? k : ks[0]) * nw + wid;
            const uint32_t Lr = (uint32_t)__shfl((int)Lc, have ? k : ks[0]);
            double *o = sums + (size_t)row * 10;
            double own[10];
            const bool fin = ql == 0 && have;                           // the lane that finishes the row
#pragma unroll
            for (int q = 0; q < 10; ++q) own[q] = fin ? o[q] : 0.0;     // directly accumulated part: in flight beside the neighbourhood
            double m[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) m[q] = 0.0;
            if (!direct_only && have) {
                const uint32_t L = Lr;
                const int sx = (int)(L % P.xy), sz = (int)((L / P.xy) % P.zs), sy = (int)(L / ((uint32_t)P.xy * P.zs));
                const int x = wrap_sub(sx, P.om[0], P.xy), y = wrap_sub(sy, P.om[1], P.xy), z = wrap_sub(sz, P.om[2], P.zs);
                for (int j0 = ql; j0 < nb; j0 += 32) {                  // two neighbours per pass: j0 and j0 + 16
                    int dxa[2], dya[2], dza[2];
                    bool in[2];
                    uint32_t tix[2], six[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int j = j0 + 16 * u;
                        const int dx = j / (wx * wz) - xy_e, dy = (j / wz) % wx - xy_e, dz = j % wz - z_e;
                        const int xn = x + dx, yn = y + dy, zn = z + dz;
                        in[u] = j < nb && xn >= 0 && xn < P.xy && yn >= 0 && yn < P.xy && zn >= 0 && zn < P.zs;
                        const int sxn = wrap_add(in[u] ? xn : x, P.om[0], P.xy), syn = wrap_add(in[u] ? yn : y, P.om[1], P.xy),
                                  szn = wrap_add(in[u] ? zn : z, P.om[2], P.zs);
                        const uint32_t rzn = (uint32_t)syn * P.zs + szn;
                        tix[u] = rzn * P.nseg + ((uint32_t)sxn >> 6); six[u] = rzn * P.xy + (uint32_t)sxn;
                        dxa[u] = dx; dya[u] = dy; dza[u] = dz;
                    }
                    // (all four loads issued together; a state is stale where its tile is dead, and then unused)
                    const uint32_t tg0 = tags[tix[0]], tg1 = tags[tix[1]];
                    const int32_t rn0 = state[six[0]], rn1 = state[six[1]];
                    const bool ok0 = in[0] && tg0 == P.epoch && rn0 >= 0, ok1 = in[1] && tg1 == P.epoch && rn1 >= 0;
                    const double *b0 = base + (size_t)(ok0 ? rn0 : 0) * GVOM_BASE_PITCH, *b1 = base + (size_t)(ok1 ? rn1 : 0) * GVOM_BASE_PITCH;
                    double v[2][10];
#pragma unroll
                    for (int q = 0; q < 10; ++q) { v[0][q] = ok0 ? b0[q] : 0.0; v[1][q] = ok1 ? b1[q] : 0.0; }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const double nn = v[u][9];
                        if (!(u == 0 ? ok0 : ok1) || !(nn > 0.0)) continue;
                        // a point of voxel (x+dx, ...) with in-voxel position l sits at l + d relative to THIS voxel
                        const double ddx = (double)dxa[u], ddy = (double)dya[u], ddz = (double)dza[u];
                        const double s0 = v[u][0], s1 = v[u][1], s2 = v[u][2];
                        m[0] += s0 + nn * ddx; m[1] += s1 + nn * ddy; m[2] += s2 + nn * ddz;
                        m[3] += v[u][3] + 2.0 * ddx * s0 + nn * ddx * ddx;
                        m[4] += v[u][4] + ddx * s1 + ddy * s0 + nn * ddx * ddy;
                        m[5] += v[u][5] + ddx * s2 + ddz * s0 + nn * ddx * ddz;
                        m[6] += v[u][6] + 2.0 * ddy * s1 + nn * ddy * ddy;
                        m[7] += v[u][7] + ddy * s2 + ddz * s1 + nn * ddy * ddz;
                        m[8] += v[u][8] + 2.0 * ddz * s2 + nn * ddz * ddz;
                        m[9] += nn;
                    }
                }
            }
            if (!direct_only) {
#pragma unroll
                for (int q = 0; q < 10; ++q)
#pragma unroll
                    for (int sh = 8; sh > 0; sh >>= 1) m[q] += __shfl_xor(m[q], sh);      // within the quarter (xor < 16)
            }
            if (!fin) continue;
#pragma unroll
            for (int q = 0; q < 10; ++q) m[q] += own[q];                     // directly accumulated part
            const double nn = m[9];
            if (!(nn > 0.0)) { for (int q = 0; q < 9; ++q) o[q] = 0.0; o[9] = nn; continue; }
            const double mx = m[0] / nn, my = m[1] / nn, mz = m[2] / nn;
            o[0] = mx; o[1] = my; o[2] = mz;
            o[3] = m[3] / nn - mx * mx; o[4] = m[4] / nn - mx * my; o[5] = m[5] / nn - mx * mz;
            o[6] = m[6] / nn - my * my; o[7] = m[7] / nn - my * mz; o[8] = m[8] / nn - mz * mz;
            o[9] = nn;
        }
    }
}

// gvom.py:858-909: pooled mean / covariance merge of one voxel; the fused metrics are float32, a ring
// slot's float64, the previous fused map's float32; TO selects the reference's arithmetic (f32*f32
// stays f32, anything touching an f64 operand is f64 -- numpy scalar rules of the simulator).
template <typename TO>
__device__ __forceinline__ void merge_metrics(float (&c)[10], const TO *o)
{
    typedef decltype((float)1 * (TO)1) W;
    const float c0 = c[0], c1 = c[1], c2 = c[2], c9 = c[9];
    const TO o0 = o[0], o1 = o[1], o2 = o[2], o9 = o[9];
    const W nn = (W)c9 + (W)o9;
    const W cm[3] = {((W)(c0 * c9) + (W)(o0 * o9)) / nn, ((W)(c1 * c9) + (W)(o1 * o9)) / nn,
                     ((W)(c2 * c9) + (W)(o2 * o9)) / nn};
    const float cmean[3] = {c0, c1, c2};
    const TO omean[3] = {o0, o1, o2};
    const int A[6] = {0, 0, 0, 1, 1, 2}, B[6] = {0, 1, 2, 1, 2, 2};
    float out[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int a = A[k], b = B[k];
        W t = (W)(c9 * c[3 + k]) + (W)(o9 * o[3 + k]);
        t = t + ((W)c9 * ((W)cmean[a] - cm[a])) * ((W)cmean[b] - cm[b]);
        t = t + ((W)o9 * ((W)omean[a] - cm[a])) * ((W)omean[b] - cm[b]);
        out[k] = (float)(t / nn);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) c[3 + k] = out[k];
    c[0] = (float)cm[0]; c[1] = (float)cm[1]; c[2] = (float)cm[2];
    c[9] = (float)nn;
}

// k_fuse_stats: the covariance half of gvom.py:821-912 for every occupied voxel of the fused map
// written by k_fuse: sources in the reference's order (ring slots, then the previous fused map).
// Occupied voxels are a few per 64-voxel tile (a surface) -- and a few HUNDRED in the tiles of a piece of ground plane -- and the
// merge is ~300 instructions and two round trips per source.  A WAVE owns 64 tiles, SCATTERED through the slab (below: the
// dense tiles of the ground plane, neighbours in tile order and all at one z level, spread over all waves): their tags in ONE instruction,
// the live ones' states four tiles per round trip, their occupied voxels {voxel, row} listed in wave-private LDS, merged one lane
// per voxel whenever 256 are listed and at the end -- no workgroup barrier.  Per voxel the sources' tags and states are fetched
// two sources at a time, then their metrics, then merged in order.  (Round 5: one-wave workgroups of 16 CONSECUTIVE tiles, 46 us
// at less than one resident wave per SIMD -- SQ_WAVE_CYCLES, profiles/r6_experiments.txt; four-wave workgroups over the same 16
// tiles behind a barrier: 83 us.)
#define FS_LIST 512
template <bool MEM>
__global__ __launch_bounds__(256) void k_fuse_stats(const FuseParams P, const FuseDescs KD,
                                                    const MapDesc *__restrict__ descs_mem,
                                                    const int32_t *__restrict__ fstate,
                                                    const uint32_t *__restrict__ ftags, float *fmetrics)
{
    __shared__ uint32_t s_list[4][FS_LIST];
    __shared__ int32_t s_rowl[4][FS_LIST];
    const cptr_desc descs = MEM ? (cptr_desc)descs_mem : (cptr_desc)KD.d;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid >> 6;
    const uint32_t t0 = (uint32_t)P.sy_lo * P.zs * P.nseg, nt = (uint32_t)(P.sy_hi - P.sy_lo) * P.zs * P.nseg;
    const uint32_t nwaves = gridDim.x * 4u, wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (uint32_t)w));
    const int nsrc = P.nslots + P.has_prev;
    uint32_t n = 0;                                                          // voxels listed and not yet merged (wave-uniform)
    auto merge_listed = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");               // (the list was written by other lanes of this wave)
        __builtin_amdgcn_wave_barrier();
        for (uint32_t e = (uint32_t)lane; e < n; e += 64u) {
            const uint32_t L = s_list[w][e];
            const int32_t row = s_rowl[w][e];
            const uint32_t rz = L / (uint32_t)P.xy;
            const int sx = (int)(L - rz * (uint32_t)P.xy), sy = (int)(rz / P.zs), sz = (int)(rz % P.zs);
            const uint32_t tl = rz * P.nseg + ((uint32_t)sx >> 6);
            const int x = wrap_sub(sx, P.om[0], P.xy), y = wrap_sub(sy, P.om[1], P.xy), z = wrap_sub(sz, P.om[2], P.zs);
            float c[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) c[k] = 0.0f;                        // gvom.py:234-236
            for (int s0 = 0; s0 < nsrc; s0 += 2) {                           // two sources per pass: their loads in flight together
                bool in[2];
                uint32_t tgv[2];
                int stv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int sI = s0 + u < nsrc ? s0 + u : s0;
                    const int xs = x + descs[sI].d[0], ys = y + descs[sI].d[1], zs_ = z + descs[sI].d[2];
                    in[u] = s0 + u < nsrc && descs[sI].metrics && !(xs < 0 || xs >= P.xy || ys < 0 || ys >= P.xy || zs_ < 0 || zs_ >= P.zs);
                    tgv[u] = descs[sI].tags[tl];
                    stv[u] = descs[sI].state[L];
                }
                float pm[10];                                                // the previous fused map's metrics (float32), if in this pass
                double sm[2][10];                                            // ring slots' (float64)
                bool ok[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int sI = s0 + u < nsrc ? s0 + u : s0;
                    ok[u] = in[u] && tgv[u] == descs[sI].epoch && stv[u] >= 0;
                    if (!ok[u]) continue;
                    if (sI < P.nslots) {
                        const double *q = (const double *)descs[sI].metrics + (size_t)stv[u] * 10;
#pragma unroll
                        for (int k = 0; k < 10; ++k) sm[u][k] = q[k];
                    } else {
                        const float *q = (const float *)descs[sI].metrics + (size_t)stv[u] * 10;
#pragma unroll
                        for (int k = 0; k < 10; ++k) pm[k] = q[k];
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (!ok[u]) continue;
                    if (s0 + u < P.nslots) merge_metrics<double>(c, sm[u]);
                    else merge_metrics<float>(c, pm);
                }
            }
#pragma unroll
            for (int k = 0; k < 10; ++k) fmetrics[(size_t)row * 10 + k] = c[k];
        }
        n = 0;
        __builtin_amdgcn_wave_barrier();                                     // (nobody refills the list before every lane has read its entries)
    };
    // owned tile k of wave wid = k * waves + (wid + 149 k) mod waves: a plain stride of `waves` tiles is a whole number of storage rows
    // on power-of-two grids and would hand a wave 64 tiles of ONE z level -- the ground plane's to a few waves, nothing to the rest
    for (uint32_t kb = 0; kb * nwaves < nt; kb += 64u) {                      // 64 owned tiles per pass
        const uint32_t kk = kb + (uint32_t)lane;
        const uint32_t ti = kk * nwaves + (wid + 149u * kk) % nwaves;        // (index inside the slab)
        const uint32_t tg = ti < nt ? ftags[t0 + ti] : ~P.epoch;
        unsigned long long live = lanes(tg == P.epoch);
        while (live != 0ull) {                                               // wave-uniform: four live tiles per round trip
            int tq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tq[q] = -1;
                if (live != 0ull) { tq[q] = __ffsll((long long)live) - 1; live &= live - 1ull; }
            }
            int32_t st[4];
            uint32_t Lb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t kq = kb + (uint32_t)(tq[q] >= 0 ? tq[q] : 0);
                const uint32_t tile = t0 + kq * nwaves + (wid + 149u * kq) % nwaves;
                const int sx = (int)(tile % P.nseg) * 64 + lane;
                Lb[q] = (tile / P.nseg) * P.xy + (uint32_t)sx;
                st[q] = (tq[q] >= 0 && sx < P.xy) ? fstate[Lb[q]] : -1;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned long long m = lanes(st[q] >= 0);
                if (st[q] >= 0) {
                    const uint32_t pos = n + (uint32_t)__popcll(m & lanemask_lt());
                    s_list[w][pos] = Lb[q];
                    s_rowl[w][pos] = st[q];
                }
                n += (uint32_t)__popcll(m);
            }
            if (n > FS_LIST - 256u) merge_listed();                          // the next batch (<= 256 voxels) might not fit
        }
    }
    if (n) merge_listed();
}

// gvom.py:1333-1378 (eigenvalues) + :454-473 (debug voxel cloud): one output row of 8 floats per
// occupied fused voxel; output position from an atomic counter (row order is unspecified).
__global__ __launch_bounds__(256) void k_voxel_cloud(const Map2dParams P, double o0, double o1, double o2,
                                                     const int32_t *__restrict__ fstate,
                                                     const uint32_t *__restrict__ ftags,
                                                     const uint4 *__restrict__ frows,
                                                     const float *__restrict__ fmetrics, float *out, float *eig,
                                                     long max_rows, unsigned long long *row_counter)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t wid = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t nw = (gridDim.x * blockDim.x) >> 6;
    const uint32_t t0 = (uint32_t)P.y_lo * P.zs * P.nseg, t1 = (uint32_t)P.y_hi * P.zs * P.nseg;
    const double PI = 3.141592653589793;
    for (uint32_t tile = t0 + wid; tile < t1; tile += nw) {
        if (ftags[tile] != P.epoch) continue;
        const uint32_t rz = tile / P.nseg;
        const int sx = (int)(tile % P.nseg) * 64 + lane, sy = (int)(rz / P.zs), sz = (int)(rz % P.zs);
        int32_t row = -1;
        if (sx < P.xy) row = fstate[rz * P.xy + sx];
        const unsigned long long b = __ballot(row >= 0);
        if (b == 0ull) continue;
        unsigned long long base = 0;
        const int leader = __ffsll((long long)b) - 1;
        if (lane == leader) base = atomicAdd(row_counter, (unsigned long long)__popcll(b));
        base = __shfl((long long)base, leader);
        if (row < 0) continue;
        const long pos = (long)base + __popcll(b & lanemask_lt());
        if (pos >= max_rows) continue;
        const float *m = fmetrics + (size_t)row * 10;
        const float xx = m[3], xy = m[4], xz = m[5], yy = m[6], yz = m[7], zz = m[8];
        const float p1 = (xy * xy + xz * xz) + yz * yz;
        const double q = (double)((xx + yy) + zz) / 3.0;
        float e0, e1, e2;
        if (p1 == 0) {
            e0 = py_maxf(xx, py_maxf(yy, zz));
            const float mn = (zz < yy) ? zz : yy;
            e2 = (mn < xx) ? mn : xx;
            e1 = (float)((3.0 * q - (double)e0) - (double)e2);
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
            e0 = (float)(q + 2.0 * p * cos(phi));
            e2 = (float)(q + 2.0 * p * cos(phi + (2.0 * PI / 3.0)));
            e1 = (float)((3.0 * q - (double)e0) - (double)e2);
        }
        const int x = wrap_sub(sx, P.om[0], P.xy), y = wrap_sub(sy, P.om[1], P.xy), z = wrap_sub(sz, P.om[2], P.zs);
        float *o = out + pos * 8;
        const uint32_t hc = frows[row].x, tc = frows[row].y;
        o[0] = (float)(((double)x + o0) * P.xy_res);
        o[1] = (float)(((double)y + o1) * P.xy_res);
        o[2] = (float)(((double)z + o2) * P.z_res);
        o[3] = (float)((double)(int32_t)hc / (double)(int32_t)tc);
        o[4] = (float)(int32_t)hc;
        o[5] = e0 - e1; o[6] = e1 - e2; o[7] = e2;
        if (eig) { eig[pos * 3 + 0] = e0; eig[pos * 3 + 1] = e1; eig[pos * 3 + 2] = e2; }   // voxels_eigenvalues (gvom.py:1374-1377)
    }
}

hipError_t gvom_launch_stats(hipStream_t s, const ScanParams &P, int dtype, const void *world, int64_t n,
                             const int32_t *state, const uint32_t *tags, int xy_e, int z_e, double *base,
                             double *sums, const uint32_t *rowvox, int64_t nrows, const void *extra, int64_t n_extra)
{
    // slab-sharded handles use the direct form only (a neighbour voxel's moments may live on another rank)
    const int direct_only = (P.sy_hi - P.sy_lo) < P.xy ? 1 : 0;
    for (int part = 0; part < 2; ++part) {
        const void *pts = part == 0 ? world : extra;
        const int64_t np = part == 0 ? n : n_extra;
        if (np <= 0 || !pts) continue;
        const unsigned blocks = (unsigned)((np + 255) / 256);
        if (dtype == 0)
            hipLaunchKernelGGL(k_stats<float>, dim3(blocks), dim3(256), 0, s, P, (const float *)pts, (long)np, state,
                               tags, xy_e, z_e, base, sums, direct_only);
        else
            hipLaunchKernelGGL(k_stats<double>, dim3(blocks), dim3(256), 0, s, P, (const double *)pts, (long)np, state,
                               tags, xy_e, z_e, base, sums, direct_only);
    }
    if (nrows > 0) {
        unsigned gb = (unsigned)((nrows + 4 * GATHER_CPW - 1) / (4 * GATHER_CPW));   // a wave per GATHER_CPW candidate rows
        if (gb > 16384) gb = 16384;
        hipLaunchKernelGGL(k_stats_gather, dim3(gb), dim3(256), 0, s, P, state, tags, xy_e, z_e, base, sums, rowvox,
                           (uint32_t)nrows, direct_only);
    }
    return hipGetLastError();
}

hipError_t gvom_launch_fuse_stats(hipStream_t s, const FuseParams &P, const FuseDescs &KD, const MapDesc *descs_dev,
                                  const int32_t *fstate, const uint32_t *ftags, float *fmetrics)
{
    const uint32_t ntiles = (uint32_t)(P.sy_hi - P.sy_lo) * P.zs * P.nseg;
    if (ntiles == 0) return hipSuccess;
    unsigned blocks = (ntiles + 255u) / 256u;               // a wave per 64 tiles, four waves per workgroup
    if (blocks > 8192) blocks = 8192;                       // (beyond 2 M tiles a wave takes further passes of 64)
    if (descs_dev) hipLaunchKernelGGL(k_fuse_stats<true>, dim3(blocks), dim3(256), 0, s, P, KD, descs_dev, fstate, ftags, fmetrics);
    else hipLaunchKernelGGL(k_fuse_stats<false>, dim3(blocks), dim3(256), 0, s, P, KD, descs_dev, fstate, ftags, fmetrics);
    return hipGetLastError();
}

hipError_t gvom_launch_voxel_cloud(hipStream_t s, const Map2dParams &P, double o0, double o1, double o2,
                                   const int32_t *fstate, const uint32_t *ftags, const uint4 *frows,
                                   const float *fmetrics, float *out, float *eig, int64_t max_rows,
                                   unsigned long long *row_counter)
{
    const uint32_t ntiles = (uint32_t)(P.y_hi - P.y_lo) * P.zs * P.nseg;
    if (ntiles == 0) return hipSuccess;
    unsigned blocks = (ntiles + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_voxel_cloud, dim3(blocks), dim3(256), 0, s, P, o0, o1, o2, fstate, ftags, frows,
                       fmetrics, out, eig, (long)max_rows, row_counter);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Test hooks / debug accessors (not on the hot path)
// ------------------------------------------------------------------------------------------
// storage order + compact rows -> dense arrays in the reference's x + y*xy + z*xy*xy order
__global__ void k_read_dense(int xy, int zs, int om0, int om1, int om2, int sy_lo, int sy_hi,
                             const uint32_t *__restrict__ tags, uint32_t epoch,
                             const int32_t *__restrict__ state, const uint4 *__restrict__ crows,
                             int32_t *o_state, int32_t *o_hit, int32_t *o_total, float *o_minh, int32_t *o_row)
{
    const size_t V = (size_t)xy * xy * zs;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < V;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % xy), y = (int)((idx / xy) % xy), z = (int)(idx / ((size_t)xy * xy));
        const int sx = wrap_add(x, om0, xy), sy = wrap_add(y, om1, xy), sz = wrap_add(z, om2, zs);
        const int nseg = (xy + 63) / 64;
        int32_t st = -1;
        if (sy >= sy_lo && sy < sy_hi && tags[((size_t)sy * zs + sz) * nseg + (sx >> 6)] == epoch)
            st = state[((size_t)sy * zs + sz) * xy + sx];
        if (o_row) { o_row[idx] = st >= 0 ? st : -1; continue; }     // compact row of every occupied voxel
        if (st >= 0) {
            const uint4 rv = crows[st];
            o_state[idx] = 0; o_hit[idx] = (int32_t)rv.x; o_total[idx] = (int32_t)rv.y;
            o_minh[idx] = __uint_as_float(rv.z);
        } else {
            o_state[idx] = st; o_hit[idx] = 0; o_total[idx] = 0; o_minh[idx] = 1.0f;
        }
    }
}

template <typename E>
__global__ void k_unwrap(int xy, int om0, int om1, const E *__restrict__ in, int in_stride, E *out_xy)
{
    const int sx = blockIdx.x * 64 + threadIdx.x, sy = blockIdx.y * 4 + threadIdx.y;
    if (sx < xy && sy < xy)
        out_xy[(size_t)wrap_sub(sx, om0, xy) * xy + wrap_sub(sy, om1, xy)] = in[(size_t)sy * in_stride + sx];
}

// Sharded runs: positive-obstacle density of the slab's own cells (the z-range gather of
// gvom.py:502-521, everything of __make_positive_obstacle_map except the slope override),
// stored as the third row of the interleaved height buffer so that it travels with the heights.
__global__ __launch_bounds__(256) void k_posdens(const Map2dParams P, const int32_t *__restrict__ fstate,
                                                 const uint32_t *__restrict__ ftags,
                                                 const uint4 *__restrict__ frows, double *hmaps,
                                                 const uint32_t *blockcounts, int nblocks,
                                                 unsigned long long *host_counter, unsigned long long *dev_counter)
{
    const int xy = P.xy;
    if (host_counter && blockIdx.x == 0 && blockIdx.y == 0) {
        // k_fuse is complete: publish this rank's fused occupied-voxel count (host-mapped memory and
        // the device word the sharded layer all-reduces on demand)
        __shared__ unsigned long long s_red[256];
        const int tid = threadIdx.y * 64 + threadIdx.x;
        publish_block_counts(blockcounts, nblocks, host_counter, s_red, tid, 256);
        if (tid == 0) *dev_counter = s_red[0];
    }
    const int sx0 = blockIdx.x * 64 + threadIdx.x, sy0 = P.y_lo + blockIdx.y * 4 + threadIdx.y;
    if (sx0 >= xy || sy0 >= P.y_hi) return;
    const double h00 = hmaps[(size_t)sy0 * P.hs + sx0];
    int pos = 0;
    const double fmin = floor(((h00 + P.pos_thr) / P.z_res) - P.origin_z) + 1.0;
    const double fmax = floor(((h00 + P.robot_height) / P.z_res) - P.origin_z);
    if (fmin >= 0 && fmin < (double)P.zs && fmax >= 0 && fmax < (double)P.zs) {
        const int zmin = (int)fmin, zmax = (int)fmax;
        double density = 0.0, nn = 0.0;
        // 8 levels per round: tags, then states, then counts -- three dependent round trips per round
        // instead of three per level (unconditional loads, dummy index when dead); same sums in the
        // same (ascending z) order as k_map2d's unsharded path
        for (int zb = zmin; zb <= zmax; zb += 8) {
            uint32_t rz[8], tg[8], hc[8], tc[8];
            int32_t row[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int z = (zb + k <= zmax) ? zb + k : zmin;
                rz[k] = (uint32_t)sy0 * P.zs + (uint32_t)wrap_add(z, P.om[2], P.zs);
                tg[k] = ftags[rz[k] * P.nseg + (sx0 >> 6)];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool live = zb + k <= zmax && tg[k] == P.epoch;       // live tile
                row[k] = fstate[live ? rz[k] * xy + sx0 : 0u];
                if (!live) row[k] = -1;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t r = row[k] >= 0 ? (uint32_t)row[k] : 0u;
                const uint2 ht = *reinterpret_cast<const uint2 *>(frows + r);
                hc[k] = ht.x; tc[k] = ht.y;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (row[k] >= 0 && (int32_t)hc[k] > 10) { nn += (double)(int32_t)tc[k]; density += (double)(int32_t)hc[k]; }
        }
        if (nn > 0.0) density /= nn;
        pos = (int)(density * 100);
    }
    hmaps[(size_t)sy0 * P.hs + 2 * (size_t)xy + sx0] = (double)pos;
}

// gvom.py:426-438 (7 columns) and :442-450 (3 columns, fed with guessed_height_delta :407)
__global__ void k_debug_height(int xy, int om0, int om1, double o0, double o1, double xy_res,
                               double z_res, const double *height, int hs, const double *rough,
                               const double *sx, const double *sy, float *out7,
                               const double *guessed, float *out3)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= xy || y >= xy) return;
    const size_t c = (size_t)y * xy + x;                 // == index = x + y*xy_size
    const size_t g = (size_t)wrap_add(y, om1, xy) * xy + wrap_add(x, om0, xy);   // storage cell
    const float wx = (float)(((double)x + o0) * xy_res), wy = (float)(((double)y + o1) * xy_res);
    if (out7) {
        const double a = sx[g], b = sy[g];
        out7[c * 7 + 0] = wx; out7[c * 7 + 1] = wy;
        out7[c * 7 + 2] = (float)(height[(size_t)wrap_add(y, om1, xy) * hs + wrap_add(x, om0, xy)] - z_res);
        out7[c * 7 + 3] = (float)rough[g];
        out7[c * 7 + 4] = (float)a; out7[c * 7 + 5] = (float)b;
        out7[c * 7 + 6] = (float)sqrt(a * a + b * b);
    }
    if (out3) {
        out3[c * 3 + 0] = wx; out3[c * 3 + 1] = wy;
        out3[c * 3 + 2] = (float)(guessed[g] - z_res);
    }
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

hipError_t gvom_launch_encode(hipStream_t s, const ScanParams &P, uint32_t *hit, uint32_t *total, uint32_t *mh,
                              int32_t *state, uint16_t *code16, uint4 *crows, const uint32_t *tags,
                              uint32_t *counters, unsigned long long *host_flag, uint32_t seq, unsigned resident_blocks)
{
    // units = quads (4 rows x 64 sx at one sz) that intersect the slab; one wave handles 2 quads per
    // iteration
    const uint32_t q_lo = (uint32_t)P.sy_lo >> 2, q_hi = ((uint32_t)P.sy_hi + 3) >> 2;
    const uint32_t t_begin = q_lo * P.zs * P.nseg, t_end = q_hi * P.zs * P.nseg;
    const uint32_t ntiles = t_end - t_begin;
    // workgroup size (every wave handles 2 quads per iteration either way).  Measured, 64 / 128 / 256 / 512 / 1024
    // threads: 256^3 (65 k quads) 18.8 / 18.8 / 21.2 / 21.0 / 30.1 us, c4 (131 k quads) 60 / 69 / 80 / 94 / 83, c5
    // (524 k quads) 127 / 126 / 100: one-wave workgroups up to 262 k quads, four-wave ones above
    const unsigned T = ntiles <= 262144u ? 64u : 256u;
    unsigned enc_blocks = (ntiles + (T / 32) - 1) / (T / 32);
    // at most four resident rounds (measured on 256^3 / 2048 resident blocks: 4096 -> 20.0 us, 8192 -> 18.8 us)
    const unsigned enc_cap = (resident_blocks > 0 ? 4u * resident_blocks : 8192u) * 256u / T;
    if (enc_blocks > enc_cap) enc_blocks = enc_cap;
    if (enc_blocks < 1) enc_blocks = 1;
    if (T == 64u)
        hipLaunchKernelGGL(k_encode<64>, dim3(enc_blocks), dim3(64), 0, s, P, t_begin, t_end, hit, total, mh, state,
                           code16, crows, tags, P.epoch, counters, host_flag, seq);
    else
        hipLaunchKernelGGL(k_encode<256>, dim3(enc_blocks), dim3(256), 0, s, P, t_begin, t_end, hit, total, mh, state,
                           code16, crows, tags, P.epoch, counters, host_flag, seq);
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

__global__ void k_publish_seq(unsigned long long *host_flag, uint32_t seq)
{
    __hip_atomic_store(host_flag, (unsigned long long)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t gvom_launch_publish_seq(hipStream_t s, unsigned long long *host_flag, uint32_t seq)
{
    hipLaunchKernelGGL(k_publish_seq, dim3(1), dim3(1), 0, s, host_flag, seq);
    return hipGetLastError();
}

// epoch renumbering (gvom_capi.hip renumber_epochs): live tiles get the map's new epoch, all others 0
__global__ void k_retag(uint32_t *tags, size_t n, uint32_t old_epoch, uint32_t new_epoch)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        tags[i] = (new_epoch != 0u && tags[i] == old_epoch) ? new_epoch : 0u;
}

hipError_t gvom_launch_retag(hipStream_t s, uint32_t *tags, size_t n, uint32_t old_epoch, uint32_t new_epoch)
{
    hipLaunchKernelGGL(k_retag, dim3(1024), dim3(256), 0, s, tags, n, old_epoch, new_epoch);
    return hipGetLastError();
}

hipError_t gvom_launch_fuse(hipStream_t s, const FuseParams &P, const FuseDescs &KD,
                            const MapDesc *descs_dev, int32_t *fstate, uint4 *frows,
                            uint32_t *ftags, uint32_t *blockcounts,
                            double *height, double *inferred)
{
    const dim3 grid((P.xy + 63) / 64, P.sy_hi - P.sy_lo);
    if (grid.y == 0) return hipSuccess;
#define FUSE_LAUNCH(...) hipLaunchKernelGGL((__VA_ARGS__), grid, dim3(64 * P.nz), 0, s, P, KD, descs_dev, fstate, frows, \
                                          ftags, blockcounts, height, inferred)
    const bool mem = descs_dev != nullptr;
    if (P.one_slot) {                                    // (the host has checked: one slot, 16-level chunks, xy % 4 == 0, descriptors by argument)
        hipLaunchKernelGGL(k_fuse1, grid, dim3(64 * P.nz), 0, s, P, KD, fstate, frows, ftags, blockcounts, height, inferred);
        return hipGetLastError();
    }
    if (P.zc == 16 && (P.xy & 3) == 0 && !GVOM_DBG(P, 8)) {
        if (P.nslots <= 2) { if (mem) FUSE_LAUNCH(k_fuse4<2, true>); else FUSE_LAUNCH(k_fuse4<2, false>); }
        else { if (mem) FUSE_LAUNCH(k_fuse4<4, true>); else FUSE_LAUNCH(k_fuse4<4, false>); }
    } else if (P.zc <= 16) {
        if (mem) FUSE_LAUNCH(k_fuse<true, true>); else FUSE_LAUNCH(k_fuse<true, false>);
    } else {
        if (mem) FUSE_LAUNCH(k_fuse<false, true>); else FUSE_LAUNCH(k_fuse<false, false>);
    }
#undef FUSE_LAUNCH
    return hipGetLastError();
}

// k_encfuse (one-slot rings; the host has checked xy % 16 == 0, z_size >= 4, the whole grid on this handle): grid = one
// workgroup per 16-sx x 4-row column block, up to 4 waves of 4 levels per iteration.  Returns the number of workgroups
// (= entries of blockcounts written) in *nblocks and the fused compact rows the launch may number in *row_cap.
// nw_override (A/B knob "encfuse", 0: none): fewer waves per column block; the row range follows the shape that is
// LAUNCHED -- a wave numbers rows from (block * nw + wave) * niter * 256, and nw' * ceil(zs / 4 nw') can exceed the default
// shape's product (z_size 16: 4 x 1 = 4 against 3 x 2 = 6).
void gvom_encfuse_shape(int xy, int zs, int nw_override, int *nw, int *nblocks, size_t *row_cap)
{
    // 4 waves per block (measured against 8 / 2: m256 98.9 / 100.2 / 102.2 us per step, c2 87.1 / 91.1 / 88.4): at 70 VGPRs a
    // SIMD holds 7 waves, i.e. 7 four-wave blocks per CU but only 3 eight-wave ones
    int w = (zs + 3) / 4;
    if (w > 4) w = 4;
    if (w < 1) w = 1;
    if (nw_override > 0 && nw_override <= w) w = nw_override;
    const int niter = (zs + 4 * w - 1) / (4 * w);
    *nw = w; *nblocks = (xy / 16) * (xy / 4);
    *row_cap = (size_t)*nblocks * (size_t)w * (size_t)niter * 256;
}
hipError_t gvom_launch_encfuse(hipStream_t s, const ScanParams &P, const FuseParams &F, const MapDesc &prev, uint32_t *hit,
                               uint32_t *total, uint32_t *mh, int32_t *state, uint4 *crows, const uint32_t *stags,
                               int32_t *fstate, uint4 *frows, uint32_t *ftags, uint32_t *blockcounts, double *height,
                               double *inferred, uint32_t *counters, unsigned long long *host_flag, uint32_t seq)
{
    int nw, nblocks; size_t cap;
    gvom_encfuse_shape(P.xy, P.zs, F.nz, &nw, &nblocks, &cap);   // (F.nz: A/B knob "encfuse"; the caller sized the fused rows with the same call)
    hipLaunchKernelGGL(k_encfuse, dim3((unsigned)nblocks), dim3(64u * (unsigned)nw), 0, s, P, F, prev, hit, total, mh, state, crows,
                       stags, fstate, frows, ftags, blockcounts, height, inferred, counters, host_flag, seq);
    return hipGetLastError();
}

hipError_t gvom_launch_map2d(hipStream_t s, const Map2dParams &P, const int32_t *fstate,
                             const uint32_t *ftags, const uint4 *frows, const double *height,
                             const double *inferred, double *slope_x, double *slope_y,
                             double *rough, double *guessed, int32_t *out_pos, int32_t *out_neg,
                             double *out_rough, int32_t *out_vis, const uint32_t *blockcounts,
                             int nblocks, unsigned long long *host_counter)
{
    if (P.y_hi <= P.y_lo) return hipSuccess;
    const int tx = P.out_yx ? 32 : 8, ty = P.out_yx ? 8 : 32;
    const dim3 grid((P.xy + tx - 1) / tx, (P.xy + ty - 1) / ty);
#define MAP2D_LAUNCH(G, Y)                                                                              \
    hipLaunchKernelGGL((k_map2d<G, Y>), grid, dim3(512), 0, s, P, fstate, ftags, frows, height, \
                       inferred, slope_x, slope_y, rough, guessed, out_pos, out_neg, out_rough, out_vis, \
                       blockcounts, nblocks, host_counter)
    if (P.gathered_pos) { if (P.out_yx) MAP2D_LAUNCH(true, true); else MAP2D_LAUNCH(true, false); }
    else { if (P.out_yx) MAP2D_LAUNCH(false, true); else MAP2D_LAUNCH(false, false); }
#undef MAP2D_LAUNCH
    return hipGetLastError();
}

hipError_t gvom_launch_read_dense(hipStream_t s, int xy, int zs, const int om[3], int sy_lo, int sy_hi,
                                  const uint32_t *tags, uint32_t epoch, const int32_t *state, const uint4 *crows,
                                  int32_t *o_state,
                                  int32_t *o_hit, int32_t *o_total, float *o_minh, int32_t *o_row)
{
    hipLaunchKernelGGL(k_read_dense, dim3(2048), dim3(256), 0, s, xy, zs, om[0], om[1], om[2], sy_lo, sy_hi,
                       tags, epoch, state,
                       crows, o_state, o_hit, o_total, o_minh, o_row);
    return hipGetLastError();
}

// rows[j] -> out[j][0..9]: the statistics of selected compact rows (reference attributes metrics_buffer /
// combined_metrics, gvom.py:54-83,234,281)
template <typename E>
__global__ void k_gather_rows10(const E *__restrict__ src, const int32_t *__restrict__ rows, long n, E *out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 10) return;
    out[i] = src[(size_t)rows[i / 10] * 10 + (i % 10)];
}

hipError_t gvom_launch_gather_rows10(hipStream_t s, int is_f64, const void *src, const int32_t *rows, int64_t n, void *out)
{
    if (n <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n * 10 + 255) / 256);
    if (is_f64) hipLaunchKernelGGL(k_gather_rows10<double>, dim3(blocks), dim3(256), 0, s, (const double *)src, rows, (long)n, (double *)out);
    else hipLaunchKernelGGL(k_gather_rows10<float>, dim3(blocks), dim3(256), 0, s, (const float *)src, rows, (long)n, (float *)out);
    return hipGetLastError();
}

hipError_t gvom_launch_unwrap_f64(hipStream_t s, int xy, int om0, int om1, const double *in, int in_stride, double *out_xy)
{
    hipLaunchKernelGGL(k_unwrap<double>, dim3((xy + 63) / 64, (xy + 3) / 4), dim3(64, 4), 0, s, xy,
                       om0, om1, in, in_stride, out_xy);
    return hipGetLastError();
}

hipError_t gvom_launch_posdens(hipStream_t s, const Map2dParams &P, const int32_t *fstate,
                               const uint32_t *ftags, const uint4 *frows,
                               double *hmaps, const uint32_t *blockcounts, int nblocks,
                               unsigned long long *host_counter, unsigned long long *dev_counter)
{
    if (P.y_hi <= P.y_lo) {                               // a rank without rows still publishes its (zero) count
        hipLaunchKernelGGL(k_publish_count, dim3(1), dim3(256), 0, s, blockcounts, nblocks, host_counter, dev_counter);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_posdens, dim3((P.xy + 63) / 64, (P.y_hi - P.y_lo + 3) / 4), dim3(64, 4), 0, s, P,
                       fstate, ftags, frows, hmaps, blockcounts, nblocks, host_counter, dev_counter);
    return hipGetLastError();
}

hipError_t gvom_launch_unwrap_i32(hipStream_t s, int xy, int om0, int om1, const int32_t *in, int32_t *out_xy)
{
    hipLaunchKernelGGL(k_unwrap<int32_t>, dim3((xy + 63) / 64, (xy + 3) / 4), dim3(64, 4), 0, s, xy,
                       om0, om1, in, xy, out_xy);
    return hipGetLastError();
}

hipError_t gvom_launch_debug_height(hipStream_t s, int xy, int om0, int om1, const double origin[3],
                                    double xy_res, double z_res, const double *height, int hs,
                                    const double *rough, const double *sx, const double *sy,
                                    float *out7, const double *guessed, float *out3)
{
    hipLaunchKernelGGL(k_debug_height, dim3((xy + 63) / 64, (xy + 3) / 4), dim3(64, 4), 0, s, xy, om0,
                       om1, origin[0], origin[1], xy_res, z_res, height, hs, rough, sx, sy, out7, guessed, out3);
    return hipGetLastError();
}
