// gvom_device.h -- what the kernel translation units of libgvom_hip.so share: vector / address-space typedefs and the small
// device helpers (storage wrap-around, the accumulator index, Python's max / min, one-instruction floor and 24-bit multiply-add,
// lane masks).  Kernels: gvom_trace.hip (scan), gvom_fuse.hip (encode + temporal fusion), gvom_map2d.hip (2-D stage, debug
// reads), gvom_stats.hip (per-voxel statistics).
#ifndef GVOM_DEVICE_H
#define GVOM_DEVICE_H
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

#endif  // GVOM_DEVICE_H
