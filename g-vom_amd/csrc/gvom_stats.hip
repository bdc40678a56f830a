// gvom_stats.hip -- the PER-VOXEL STATISTICS kernels of libgvom_hip.so (gfx950, wave64), reference gvom.py:
//
//   k_stats, k_stats_gather   gvom.py:1172-1299 (mean / covariance of every occupied voxel's neighbourhood)
//   k_fuse_stats              gvom.py:858-909 (their merge over the ring and the previous map)
//   k_voxel_cloud             gvom.py:1333-1378 (eigenvalues) + :454-473 (debug voxel cloud)
//   k_gather_rows10           reference attributes metrics_buffer / combined_metrics
// Float accumulation order is unspecified on a GPU (as in the reference's f64 atomics): tolerance-compared.
#include "gvom_device.h"

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
                    // the previous map of an eager fusion comes with a link table (k_encfuse: this fused row <- that row of the previous
                    // map): neither its states nor its tile tags are read -- the next scan's k_encfuse may be rewriting them by now
                    const int32_t *lk = sI >= P.nslots ? descs[sI].link : nullptr;
                    tgv[u] = lk ? descs[sI].epoch : descs[sI].tags[tl];
                    stv[u] = lk ? lk[row] : descs[sI].state[L];
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
    // a wave per 64 tiles on big grids, fewer tiles per wave on small ones (at least ~4096 waves wherever there are that many
    // tiles: c1's 2,048 tiles in 32 waves left the chip empty), four waves per workgroup
    unsigned per_wave = ntiles / 4096u;
    per_wave = per_wave < 1u ? 1u : (per_wave > 64u ? 64u : per_wave);
    unsigned blocks = ((ntiles + per_wave - 1u) / per_wave + 3u) / 4u;
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
