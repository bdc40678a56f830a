// gvom_fuse.hip -- ENCODING and TEMPORAL FUSION kernels of libgvom_hip.so (gfx950, wave64), reference gvom.py:
//
//   k_encode  gvom.py:1154-1168 (state code + dense->compact move) + the three V-sized clears
//             of :114-121 (accumulators are cleared as they are read; no separate fill), only on
//             the 64-voxel tiles the scan touched
//   k_fuse4   gvom.py:943-968 x slots, :972-997, :821-912 (count/min lines 910-912) x (slots+1),
//   (k_fuse1  :525-540 (height) and :544-554 (inferred height): ONE pass over the fused grid
//    k_fuse)  (k_fuse1: one ring slot; k_fuse: grids with xy % 4 != 0 or chunks other than 16 levels)
//   k_encfuse one-slot rings: k_encode + k_fuse1 in one pass over the scan's accumulators (eager fusion)
//   k_publish_seq, k_retag   completion flag (A/B form), epoch renumbering
//
// Numerics are the reference's as executed by the Numba simulator:
// compile with -ffp-contract=off, IEEE division/sqrt, no fast-math.  Integer results are
// bit-exact; only log()/atan2() may differ from glibc in the last ulp.
// No MFMA: there is no dense contraction on this path.
#include "gvom_device.h"

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
    const ScanParams P, const FuseParams F, const MapDesc prev, int32_t *flink, uint32_t *hit, uint32_t *total, uint32_t *mh,
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
                if (flink) flink[row] = stp;                   // (statistics: which row of the previous map merges into this one, if any)
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
hipError_t gvom_launch_encfuse(hipStream_t s, const ScanParams &P, const FuseParams &F, const MapDesc &prev, int32_t *flink, uint32_t *hit,
                               uint32_t *total, uint32_t *mh, int32_t *state, uint4 *crows, const uint32_t *stags,
                               int32_t *fstate, uint4 *frows, uint32_t *ftags, uint32_t *blockcounts, double *height,
                               double *inferred, uint32_t *counters, unsigned long long *host_flag, uint32_t seq)
{
    int nw, nblocks; size_t cap;
    gvom_encfuse_shape(P.xy, P.zs, F.nz, &nw, &nblocks, &cap);   // (F.nz: A/B knob "encfuse"; the caller sized the fused rows with the same call)
    hipLaunchKernelGGL(k_encfuse, dim3((unsigned)nblocks), dim3(64u * (unsigned)nw), 0, s, P, F, prev, flink, hit, total, mh, state, crows,
                       stags, fstate, frows, ftags, blockcounts, height, inferred, counters, host_flag, seq);
    return hipGetLastError();
}
