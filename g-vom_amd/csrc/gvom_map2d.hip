// gvom_map2d.hip -- the 2-D STAGE of libgvom_hip.so (gfx950, wave64) and the dense read-back / debug kernels, reference gvom.py:
//
//   k_map2d   gvom.py:665-734 (slope/roughness), :558-661 (guess height), :489-521 (positive),
//             :479-485 (negative), :414-422 (visibility); the four returned maps straight into pinned host memory
//   k_posdens sharded maps: the positive-obstacle densities of a rank's own cells (gvom.py:489-521)
//   k_read_dense, k_unwrap, k_debug_height   test hooks and the debug accessors (gvom.py:356-410)
//
// Numerics are the reference's as executed by the Numba simulator:
// compile with -ffp-contract=off, IEEE division/sqrt, no fast-math.  Integer results are
// bit-exact; only log()/atan2() may differ from glibc in the last ulp.
// No MFMA: there is no dense contraction on this path.
#include "gvom_device.h"

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
