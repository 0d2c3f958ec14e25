// LDS-tiled 27-cell sweeps for gfx950 (the hot kernels).
//
// One workgroup owns a BRICK of 8x4x4 grid cells.  It stages the positions of every listed
// particle (first min(count,32) slots of each cell = the reference's bucket rows) of the brick plus
// a halo into LDS once, together with a per-cell table (LDS offset, listed count, global slot),
// and then sweeps its own particles against LDS:
//   density  : halo 1 -- at density time a particle sits in the cell it was binned into;
//   VelPos   : halo 2 -- the grid is frozen for the 5 sub-steps of a frame (SphFluidSimulation.cs
//              :98-102) while particles move, so a target may have drifted one cell away from
//              its bin and still needs the bins around its CURRENT cell.
// Anything the tile cannot serve exactly is handed to the generic per-target kernels of
// sph_kernels_sweep.hip through a device queue: targets whose 27 cells are not all inside the
// halo box (moved more than one cell, aliased bins), bricks whose listed particles exceed the
// LDS tile, limbo particles, and launches flagged "wild".  Both paths evaluate the reference
// statements in the reference order, so the result does not depend on which path ran.
//
// VelPos is two-phase as in sph_kernels_sweep.hip (distance test -> per-lane LDS list ->
// pair arithmetic), with phase 1 reading LDS only; phase 2 gathers rho_j / vel_j of the ~15
// survivors per target from L2.
#include "sph_sweep_common.hpp"

namespace sph {

constexpr int BX = 8, BY = 4, BZ = 4;  // owned cells per brick
constexpr int kT = 256;                // threads per brick
constexpr int kRows = BY * BZ;

template <int H>
struct Halo {
    static constexpr int HX = BX + 2 * H, HY = BY + 2 * H, HZ = BZ + 2 * H;
    static constexpr int NH = HX * HY * HZ;
    static constexpr int PER = (NH + kT - 1) / kT;  // halo cells per thread during set-up
};

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, kWave);
        if (lane >= o) v += t;
    }
    return v;
}

// exclusive scan across the kT threads of a brick; *total = sum.  lds: kT/kWave + 1 words.
__device__ __forceinline__ uint32_t brick_excl_scan(uint32_t v, uint32_t *lds, uint32_t *total)
{
    constexpr int NW = kT / kWave;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t incl = wave_incl_scan_u32(v);
    if (lane == kWave - 1) lds[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int w = 0; w < NW; w++) {
            const uint32_t t = lds[w];
            lds[w] = run;
            run += t;
        }
        lds[NW] = run;
    }
    __syncthreads();
    *total = lds[NW];
    return lds[wave] + incl - v;
}

// Every in-grid neighbour coordinate i-1..i+1 must fall inside the halo box [lo_box, lo_box+n).
__device__ __forceinline__ bool axis_ok(int i, int lo_box, int n_box, int R)
{
    const int lo = i - 1 > 0 ? i - 1 : 0;
    const int hi = i + 1 < R - 1 ? i + 1 : R - 1;
    if (lo > hi) return true;  // no in-grid neighbour on this axis: nothing will be looked up
    return lo >= lo_box && hi <= lo_box + n_box - 1;
}

struct BrickGeom {
    int x0, y0, z0;
};

__device__ __forceinline__ BrickGeom brick_geom(int nbx, int nby)
{
    const int b = blockIdx.x;
    return BrickGeom{(b % nbx) * BX, ((b / nbx) % nby) * BY, (b / (nbx * nby)) * BZ};
}

__device__ __forceinline__ void enqueue(const Dev &d, uint32_t slot)
{
    const unsigned long long q = atomicAdd(&d.counters[kCntQueue], 1ull);
    d.queue[q] = slot;
}

// slot of the t-th owned particle of the brick (rows are contiguous slot runs)
__device__ __forceinline__ uint32_t target_slot(const uint32_t *row_start, const uint32_t *row_pref, uint32_t t)
{
    int r = 0;
    while (r < kRows - 1 && t >= row_pref[r + 1]) r++;
    return row_start[r] + (t - row_pref[r]);
}

// Common set-up: owned rows, halo cell table, staging of positions.  Returns the number of owned
// particles (0: nothing to do or everything was queued).
template <int H, int CAP>
__device__ __forceinline__ uint32_t brick_setup(const Dev &d, const BrickGeom g, bool force_queue, float *sx,
                                                float *sy, float *sz, uint32_t *s_info, uint32_t *s_g0,
                                                uint32_t *s_row_start, uint32_t *s_row_pref, uint32_t *s_scan)
{
    using HB = Halo<H>;
    const uint32_t tid = threadIdx.x;
    const int R = (int)d.R;

    if (blockIdx.x == 0)  // limbo particles belong to no brick
        for (uint32_t t = d.cell_start[d.num_cells] + tid; t < d.count; t += kT) enqueue(d, t);

    if (tid < kRows) {
        const int cy = g.y0 + (int)(tid % BY), cz = g.z0 + (int)(tid / BY);
        uint32_t st = 0, len = 0;
        if (cy < R && cz < R) {
            const int xb = g.x0 + BX - 1 < R - 1 ? g.x0 + BX - 1 : R - 1;
            const uint32_t row = (uint32_t)cy * d.R + (uint32_t)cz * d.R * d.R;
            st = d.cell_start[row + (uint32_t)g.x0];
            len = d.cell_start[row + (uint32_t)xb + 1u] - st;
        }
        s_row_start[tid] = st;
        s_row_pref[tid + 1] = len;
    }
    __syncthreads();
    if (tid == 0) {
        s_row_pref[0] = 0;
        for (int r = 0; r < kRows; r++) s_row_pref[r + 1] += s_row_pref[r];
    }
    __syncthreads();
    const uint32_t Tn = s_row_pref[kRows];
    if (Tn == 0) return 0;

    // halo cell table
    uint32_t cnt[HB::PER], g0[HB::PER], sum = 0;
#pragma unroll
    for (int i = 0; i < HB::PER; i++) {
        const int hc = (int)tid * HB::PER + i;
        cnt[i] = 0;
        g0[i] = 0;
        if (hc < HB::NH) {
            const int cx = g.x0 - H + hc % HB::HX, cy = g.y0 - H + (hc / HB::HX) % HB::HY,
                      cz = g.z0 - H + hc / (HB::HX * HB::HY);
            if (cx >= 0 && cy >= 0 && cz >= 0 && cx < R && cy < R && cz < R) {
                const uint32_t key = (uint32_t)cx + (uint32_t)cy * d.R + (uint32_t)cz * d.R * d.R;
                const uint32_t a = d.cell_start[key];
                const uint32_t n = d.cell_start[key + 1] - a;
                cnt[i] = n < kMaxPerVoxel ? n : kMaxPerVoxel;
                g0[i] = a;
            }
        }
        sum += cnt[i];
    }
    uint32_t total;
    uint32_t run = brick_excl_scan(sum, s_scan, &total);
    if (force_queue || total > (uint32_t)CAP) {
        for (uint32_t t = tid; t < Tn; t += kT) enqueue(d, target_slot(s_row_start, s_row_pref, t));
        return 0;
    }
#pragma unroll
    for (int i = 0; i < HB::PER; i++) {
        const int hc = (int)tid * HB::PER + i;
        if (hc < HB::NH) {
            s_info[hc] = run | (cnt[i] << 16);
            s_g0[hc] = g0[i];
            for (uint32_t k = 0; k < cnt[i]; k++) {
                sx[run + k] = d.cur.x[g0[i] + k];
                sy[run + k] = d.cur.y[g0[i] + k];
                sz[run + k] = d.cur.z[g0[i] + k];
            }
            run += cnt[i];
        }
    }
    __syncthreads();
    return Tn;
}

// ---- Density.compute:29-60 ---------------------------------------------------------------------

template <int H, int CAP>
__global__ void __launch_bounds__(kT) k_density_brick(Dev d, int nbx, int nby)
{
    using HB = Halo<H>;
    __shared__ float sx[CAP], sy[CAP], sz[CAP];
    __shared__ uint32_t s_info[HB::NH], s_g0[HB::NH];
    __shared__ uint32_t s_row_start[kRows], s_row_pref[kRows + 1], s_scan[kT / kWave + 1];

    const BrickGeom g = brick_geom(nbx, nby);
    const uint32_t Tn = brick_setup<H, CAP>(d, g, false, sx, sy, sz, s_info, s_g0, s_row_start, s_row_pref, s_scan);
    if (Tn == 0) return;

    const int R = (int)d.R;
    const int hx0 = g.x0 - H, hy0 = g.y0 - H, hz0 = g.z0 - H;
    unsigned long long visited = 0;
    for (uint32_t t = threadIdx.x; t < Tn; t += kT) {
        const uint32_t s = target_slot(s_row_start, s_row_pref, t);
        const float pix = d.cur.x[s], piy = d.cur.y[s], piz = d.cur.z[s];        // :38
        const int ix = cell_coord(pix, d.Rm1, R), iy = cell_coord(piy, d.Rm1, R), iz = cell_coord(piz, d.Rm1, R);
        if (!(axis_ok(ix, hx0, HB::HX, R) && axis_ok(iy, hy0, HB::HY, R) && axis_ok(iz, hz0, HB::HZ, R))) {
            enqueue(d, s);
            continue;
        }
        float rho = 0.0f;                                                        // :41
        for (int xo = -1; xo <= 1; xo++)                                         // :42-44
            for (int yo = -1; yo <= 1; yo++)
                for (int zo = -1; zo <= 1; zo++) {
                    const int cx = ix + xo, cy = iy + yo, cz = iz + zo;
                    if (cx < 0 || cy < 0 || cz < 0 || cx >= R || cy >= R || cz >= R) continue;  // :46
                    const uint32_t inf = s_info[(cx - hx0) + HB::HX * ((cy - hy0) + HB::HY * (cz - hz0))];
                    const uint32_t base = inf & 0xFFFFu, cnt = inf >> 16;
                    visited += cnt;
                    for (uint32_t k = 0; k < cnt; k++)                           // :50-55
                        rho += density_term(d, pix - sx[base + k], piy - sy[base + k], piz - sz[base + k]);
                }
        d.rho[s] = rho;                                                          // :59
        if (is_wild(d.gas_const * (rho - d.rest_density))) d.counters[kCntWildP] = 1ull;
    }
    visited = wave_sum_u64(visited);
    if ((threadIdx.x & (kWave - 1)) == 0 && visited) atomicAdd(&d.counters[kCntCandidates], visited);
}

// ---- VelPos.compute:46-158 ---------------------------------------------------------------------

template <int H, int CAP, int LCAP>
__global__ void __launch_bounds__(kT) k_velpos_brick(Dev d, int nbx, int nby)
{
    using HB = Halo<H>;
    __shared__ float sx[CAP], sy[CAP], sz[CAP];
    __shared__ uint32_t s_info[HB::NH], s_g0[HB::NH];
    __shared__ uint32_t s_row_start[kRows], s_row_pref[kRows + 1], s_scan[kT / kWave + 1];
    __shared__ uint32_t s_list32[(LCAP / 2) * kT];  // u16 entries, dword column = thread
    uint16_t *list = reinterpret_cast<uint16_t *>(s_list32);

    const BrickGeom g = brick_geom(nbx, nby);
    const uint32_t Tn = brick_setup<H, CAP>(d, g, launch_is_wild(d), sx, sy, sz, s_info, s_g0, s_row_start,
                                           s_row_pref, s_scan);
    if (Tn == 0) return;

    const uint32_t tid = threadIdx.x;
    const int R = (int)d.R;
    const int hx0 = g.x0 - H, hy0 = g.y0 - H, hz0 = g.z0 - H;

    for (uint32_t tb = 0; tb < Tn; tb += kT) {                                   // block-uniform
        const uint32_t t = tb + tid;
        bool live = t < Tn;
        const uint32_t s = target_slot(s_row_start, s_row_pref, live ? t : Tn - 1);
        const float pix = d.cur.x[s], piy = d.cur.y[s], piz = d.cur.z[s];        // :57
        const int ix = cell_coord(pix, d.Rm1, R), iy = cell_coord(piy, d.Rm1, R), iz = cell_coord(piz, d.Rm1, R);
        const bool inside = axis_ok(ix, hx0, HB::HX, R) && axis_ok(iy, hy0, HB::HY, R) && axis_ok(iz, hz0, HB::HZ, R);
        if (live && !inside) enqueue(d, s);
        live = live && inside;                 // dead lanes keep voting but look nothing up
        const float rho_i = d.rho[s];                                            // :60
        const float p_i = d.gas_const * (rho_i - d.rest_density);                // :61
        const float vix = d.cur.vx[s], viy = d.cur.vy[s], viz = d.cur.vz[s], viw = d.cur.vw[s]; // :62
        Force f{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};                             // :64-65

        // phase 2: pair arithmetic over the listed survivors, in list (= reference) order
        auto phase2 = [&](uint32_t n) {
            for (uint32_t e = 0; e < n; e++) {
                const uint32_t ent = list[(((e >> 1) * kT + tid) << 1) + (e & 1u)];
                const int c = (int)(ent >> 5);
                const uint32_t k = ent & 31u;
                const int cx = ix + c / 9 - 1, cy = iy + (c / 3) % 3 - 1, cz = iz + c % 3 - 1;
                const int hc = (cx - hx0) + HB::HX * ((cy - hy0) + HB::HY * (cz - hz0));
                const uint32_t idx = (s_info[hc] & 0xFFFFu) + k;
                const uint32_t j = s_g0[hc] + k;
                pair_force_vals(d, pix - sx[idx], piy - sy[idx], piz - sz[idx], d.rho[j], d.cur.vx[j],
                                d.cur.vy[j], d.cur.vz[j], vix, viy, viz, p_i, f);   // :86-95
            }
        };

        // phase 1: distance test against LDS, reference order (:67-81)
        uint32_t n = 0;
        for (int ci = 0; ci < 27; ci++) {                                        // wave-uniform
            const int cx = ix + ci / 9 - 1, cy = iy + (ci / 3) % 3 - 1, cz = iz + ci % 3 - 1;
            uint32_t base = 0, cnt = 0, self_k = 0xFFFFFFFFu;
            if (live && cx >= 0 && cy >= 0 && cz >= 0 && cx < R && cy < R && cz < R) {  // :73
                const int hc = (cx - hx0) + HB::HX * ((cy - hy0) + HB::HY * (cz - hz0));
                const uint32_t inf = s_info[hc];
                base = inf & 0xFFFFu;
                cnt = inf >> 16;
                self_k = s - s_g0[hc];         // :82: k == self_k is the target itself (if listed here)
            }
            uint32_t k = 0;
            while (true) {
                while (k < cnt && n < (uint32_t)LCAP) {
                    const float rx = pix - sx[base + k], ry = piy - sy[base + k], rz = piz - sz[base + k];
                    const float r2 = (rx * rx + ry * ry) + rz * rz;
                    if (k != self_k && r2 < d.h2_up) {   // superset of |r| < h (VelPos.compute:37,43)
                        list[(((n >> 1) * kT + tid) << 1) + (n & 1u)] = (uint16_t)((ci << 5) | k);
                        n++;
                    }
                    k++;
                }
                if (!__any(k < cnt)) break;    // every lane finished this cell
                phase2(n);                     // some lane's list is full: drain all lists
                n = 0;
            }
        }
        phase2(n);
        if (live) integrate_store(d, s, pix, piy, piz, vix, viy, viz, viw, rho_i, f);  // :100-157
    }
}

// ---- launchers -----------------------------------------------------------------------------------

void launch_density_all(const Dev &d, hipStream_t s);
void launch_velpos_all(const Dev &d, hipStream_t s);
void launch_density_queue(const Dev &d, hipStream_t s);
void launch_velpos_queue(const Dev &d, hipStream_t s);

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

void launch_density(const Dev &d, hipStream_t s)
{
    (void)hipMemsetAsync(d.counters + kCntCandidates, 0, sizeof(unsigned long long), s);
    (void)hipMemsetAsync(d.counters + kCntWildP, 0, sizeof(unsigned long long), s);
    if (d.use_bricks == 0u) {
        launch_density_all(d, s);
        return;
    }
    (void)hipMemsetAsync(d.counters + kCntQueue, 0, sizeof(unsigned long long), s);
    const uint32_t nbx = cdiv(d.R, BX), nby = cdiv(d.R, BY), nbz = cdiv(d.R, BZ);
    hipLaunchKernelGGL((k_density_brick<1, 3072>), dim3(nbx * nby * nbz), dim3(kT), 0, s, d, (int)nbx, (int)nby);
    launch_density_queue(d, s);
}

void launch_velpos(const Dev &d, hipStream_t s)
{
    // velocity wildness of this pass's output is collected in the other slot
    (void)hipMemsetAsync(d.counters + kCntWildV0 + (d.wild_v_cur ^ 1), 0, sizeof(unsigned long long), s);
    if (d.use_bricks == 0u) {
        launch_velpos_all(d, s);
        return;
    }
    (void)hipMemsetAsync(d.counters + kCntQueue, 0, sizeof(unsigned long long), s);
    const uint32_t nbx = cdiv(d.R, BX), nby = cdiv(d.R, BY), nbz = cdiv(d.R, BZ);
    hipLaunchKernelGGL((k_velpos_brick<2, 4096, 24>), dim3(nbx * nby * nbz), dim3(kT), 0, s, d, (int)nbx, (int)nby);
    launch_velpos_queue(d, s);
}

}  // namespace sph
