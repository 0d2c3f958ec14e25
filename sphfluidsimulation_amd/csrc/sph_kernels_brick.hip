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

#ifndef SPH_ABLATE
#define SPH_ABLATE 0
#endif

namespace sph {

// tuning knobs (diagnostic builds override them with -D, see scripts/)
#ifndef SPH_BRICK_BY
#define SPH_BRICK_BY 4
#endif
#ifndef SPH_VELPOS_T
#define SPH_VELPOS_T 512
#endif
#ifndef SPH_DENSITY_T
#define SPH_DENSITY_T 256
#endif
#ifndef SPH_VELPOS_CAP
#define SPH_VELPOS_CAP 3328
#endif
#ifndef SPH_DENSITY_CAP
#define SPH_DENSITY_CAP 2048
#endif
#ifndef SPH_LCAP
#define SPH_LCAP 32
#endif
#ifndef SPH_DENSITY_MINWAVES
#define SPH_DENSITY_MINWAVES 5
#endif
#ifndef SPH_SCAN_UNROLL
#define SPH_SCAN_UNROLL 4
#endif
constexpr int kU = SPH_SCAN_UNROLL;             // slots per trip of the neighbour scans
constexpr int BX = 8, BY = SPH_BRICK_BY, BZ = 4;  // owned cells per brick
constexpr int kRows = BY * BZ;

template <int H, int T = 1>
struct Halo {
    static constexpr int HX = BX + 2 * H, HY = BY + 2 * H, HZ = BZ + 2 * H;
    static constexpr int NH = HX * HY * HZ;
    static constexpr int PER = (NH + T - 1) / T;  // halo cells per thread during set-up
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

// exclusive scan across the T threads of a brick; *total = sum.  lds: T/kWave + 1 words.
template <int T>
__device__ __forceinline__ uint32_t brick_excl_scan(uint32_t v, uint32_t *lds, uint32_t *total)
{
    constexpr int NW = T / kWave;
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

// LDS index of halo cell (cx,cy,cz): z runs fastest, so the reference's innermost loop (z_offset,
// Density.compute:44 / VelPos.compute:69) walks ONE contiguous LDS run per (x_offset, y_offset).
template <int H>
__device__ __forceinline__ int hcz(const BrickGeom g, int cx, int cy, int cz)
{
    using HB = Halo<H>;
    return (cz - (g.z0 - H)) + HB::HZ * ((cy - (g.y0 - H)) + HB::HY * (cx - (g.x0 - H)));
}

// Common set-up for the owned cells [g.x0, g.x0+nx) x BY x BZ: owned rows, halo cell table, staging of
// positions.  Returns kEmpty (no owned particle), kReady (tile staged, *Tn owned particles) or
// kOverflow (the listed particles of the halo box exceed the tile; rows are valid, nothing staged).
enum { kEmpty = 0, kReady = 1, kOverflow = 2 };

// Slot runs of the owned rows (y,z) of the cells [g.x0, g.x0+nx): fills s_row_start / s_row_pref and returns
// the number of owned particles (every member of a cell is a target, also the ones beyond the 32-cap).
__device__ __forceinline__ uint32_t brick_rows(const Dev &d, const BrickGeom g, int nx, uint32_t *s_row_start,
                                               uint32_t *s_row_pref)
{
    const uint32_t tid = threadIdx.x;
    const int R = (int)d.R;
    if (tid < kRows) {
        const int cy = g.y0 + (int)(tid % BY), cz = g.z0 + (int)(tid / BY);
        uint32_t st = 0, len = 0;
        if (cy < R && cz < R && g.x0 < R && cz >= d.zo_lo && cz < d.zo_hi) {   // ghost planes hold no targets
            const int xb = g.x0 + nx - 1 < R - 1 ? g.x0 + nx - 1 : R - 1;
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
    return s_row_pref[kRows];
}

template <int H, int CAP, int T>
__device__ __forceinline__ int brick_setup(const Dev &d, const BrickGeom g, int nx, float *sx, float *sy, float *sz,
                                           uint32_t *s_info, uint32_t *s_g0, uint32_t *s_row_start,
                                           uint32_t *s_row_pref, uint32_t *s_scan, uint32_t *Tn_out)
{
    using HB = Halo<H, T>;
    const uint32_t tid = threadIdx.x;
    const int R = (int)d.R;
    const uint32_t Tn = brick_rows(d, g, nx, s_row_start, s_row_pref);
    *Tn_out = Tn;
    if (Tn == 0) return kEmpty;

    // A: bucket rows of the halo cells, threads x-fastest (coalesced cell_start reads)
    uint32_t cnt[HB::PER], g0[HB::PER];
    int lds_cell[HB::PER];
#pragma unroll
    for (int i = 0; i < HB::PER; i++) {
        const int a = (int)tid * HB::PER + i;
        cnt[i] = 0;
        g0[i] = 0;
        lds_cell[i] = -1;
        if (a < HB::NH) {
            const int cx = g.x0 - H + a % HB::HX, cy = g.y0 - H + (a / HB::HX) % HB::HY,
                      cz = g.z0 - H + a / (HB::HX * HB::HY);
            if (a % HB::HX < nx + 2 * H && cx >= 0 && cy >= 0 && cz >= 0 && cx < R && cy < R && cz < R) {
                const uint32_t key = (uint32_t)cx + (uint32_t)cy * d.R + (uint32_t)cz * d.R * d.R;
                const uint32_t st = d.cell_start[key];
                const uint32_t n = d.cell_start[key + 1] - st;
                cnt[i] = n < kMaxPerVoxel ? n : kMaxPerVoxel;
                g0[i] = st;
            }
            lds_cell[i] = hcz<H>(g, cx, cy, cz);
            s_info[lds_cell[i]] = cnt[i];
            s_g0[lds_cell[i]] = g0[i];
        }
    }
    __syncthreads();
    // B: LDS offsets in z-fastest cell order
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < HB::PER; i++) {
        const int b = (int)tid * HB::PER + i;
        if (b < HB::NH) sum += s_info[b];
    }
    uint32_t total;
    uint32_t run = brick_excl_scan<T>(sum, s_scan, &total);
    if (total > (uint32_t)CAP) return kOverflow;
#pragma unroll
    for (int i = 0; i < HB::PER; i++) {
        const int b = (int)tid * HB::PER + i;
        if (b < HB::NH) {
            const uint32_t c = s_info[b];
            s_info[b] = run | (c << 16);
            run += c;
        }
    }
    __syncthreads();
    // C: copy the listed particles, threads x-fastest again (coalesced global reads)
#pragma unroll
    for (int i = 0; i < HB::PER; i++) {
        if (lds_cell[i] >= 0) {
            const uint32_t base = s_info[lds_cell[i]] & 0xFFFFu;
            for (uint32_t k = 0; k < cnt[i]; k++) {
                sx[base + k] = d.cur.x[g0[i] + k];
                sy[base + k] = d.cur.y[g0[i] + k];
                sz[base + k] = d.cur.z[g0[i] + k];
            }
        }
    }
    __syncthreads();
    return kReady;
}

// Walks the brick's owned x-range: the whole brick if its halo fits the tile, otherwise halves
// (down to 2 cells) so that crowded regions stay on the LDS path; what still does not fit goes to
// the generic queue.  `body(sub_geom, nx, Tn)` sweeps one staged sub-brick.
template <int H, int CAP, int T, typename Body>
__device__ __forceinline__ void for_each_subbrick(const Dev &d, const BrickGeom g, bool force_queue, float *sx,
                                                  float *sy, float *sz, uint32_t *s_info, uint32_t *s_g0,
                                                  uint32_t *s_row_start, uint32_t *s_row_pref, uint32_t *s_scan,
                                                  Body body)
{
    if (blockIdx.x == 0)  // limbo particles belong to no brick
        for (uint32_t t = d.cell_start[d.num_cells] + threadIdx.x; t < d.count; t += T) enqueue(d, t);
    int xa = g.x0, w = BX;
    while (xa < g.x0 + BX) {                                   // block-uniform
        __syncthreads();                                       // the previous sweep is done with the tile
        const BrickGeom sg{xa, g.y0, g.z0};
        uint32_t Tn = 0;
        int rc = kOverflow;
        if (force_queue) {
            w = BX;                                            // rows only: the literal kernel takes the whole brick
            Tn = brick_rows(d, sg, w, s_row_start, s_row_pref);
            rc = Tn ? kOverflow : kEmpty;
        } else {
            rc = brick_setup<H, CAP, T>(d, sg, w, sx, sy, sz, s_info, s_g0, s_row_start, s_row_pref, s_scan, &Tn);
            if (rc == kOverflow && w > 2) {
                w >>= 1;                                       // retry the left half
                continue;
            }
        }
        if (rc == kOverflow)
            for (uint32_t t = threadIdx.x; t < Tn; t += T) enqueue(d, target_slot(s_row_start, s_row_pref, t));
        else if (rc == kReady)
            body(sg, w, Tn);
        xa += w;
        while (w < BX && ((xa - g.x0) % (2 * w)) == 0) w <<= 1;  // try the widest aligned range again
    }
}

// One (x_offset, y_offset) column of the 27-cell stencil: the in-grid cells (cx, cy, iz-1..iz+1)
// form one contiguous LDS run [base, base+cnt) in reference order (z_offset ascending, slots ascending).
struct Run {
    uint32_t base, cnt;
    int first;  // LDS cell index of the first in-grid cell of the run
};

template <int H>
__device__ __forceinline__ Run stencil_run(const BrickGeom g, const uint32_t *s_info, int cx, int cy, int iz, int R)
{
    Run r{0u, 0u, 0};
    const int zlo = iz - 1 > 0 ? iz - 1 : 0;
    const int zhi = iz + 1 < R - 1 ? iz + 1 : R - 1;
    if (cx >= 0 && cy >= 0 && cx < R && cy < R && zlo <= zhi) {   // Density.compute:46 / VelPos.compute:73
        r.first = hcz<H>(g, cx, cy, zlo);
        const uint32_t a = s_info[r.first], b = s_info[r.first + (zhi - zlo)];
        r.base = a & 0xFFFFu;
        r.cnt = (b & 0xFFFFu) + (b >> 16) - r.base;
    }
    return r;
}

// ---- Density.compute:29-60 ---------------------------------------------------------------------

template <int H, int CAP, int T>
__global__ void __launch_bounds__(T, SPH_DENSITY_MINWAVES) k_density_brick(Dev d, int nbx, int nby)
{
    using HB = Halo<H, T>;
    __shared__ float sx[CAP + kU], sy[CAP + kU], sz[CAP + kU];   // +kU: the unrolled scan may read past a run
    __shared__ uint32_t s_info[HB::NH], s_g0[HB::NH];
    __shared__ uint32_t s_row_start[kRows], s_row_pref[kRows + 1], s_scan[T / kWave + 1];

    const BrickGeom g0 = brick_geom(nbx, nby);
    const int R = (int)d.R;
    unsigned long long visited = 0;
    for_each_subbrick<H, CAP, T>(d, g0, false, sx, sy, sz, s_info, s_g0, s_row_start, s_row_pref, s_scan,
                                 [&](const BrickGeom g, int nx, uint32_t Tn) {
    const int hx0 = g.x0 - H, hy0 = g.y0 - H, hz0 = g.z0 - H;
    for (uint32_t t = threadIdx.x; t < Tn; t += T) {
        const uint32_t s = target_slot(s_row_start, s_row_pref, t);
        const float pix = d.cur.x[s], piy = d.cur.y[s], piz = d.cur.z[s];        // :38
        const int ix = cell_coord(pix, d.Rm1, R), iy = cell_coord(piy, d.Rm1, R), iz = cell_coord(piz, d.Rm1, R);
        if (d.ghost && d.ghost[s]) continue;   // ghosts are candidates only
        if (!(axis_ok(ix, hx0, nx + 2 * H, R) && axis_ok(iy, hy0, HB::HY, R) && axis_ok(iz, hz0, HB::HZ, R))) {
            enqueue(d, s);
            continue;
        }
        float rho = 0.0f;                                                        // :41
        for (int xo = -1; xo <= 1; xo++)                                         // :42
            for (int yo = -1; yo <= 1; yo++) {                                   // :43
                const Run r = stencil_run<H>(g, s_info, ix + xo, iy + yo, iz, R); // :44-49, z_offset run
                visited += r.cnt;
                for (uint32_t p = 0; p < r.cnt; p += kU) {                       // :50-55, kU slots per trip
                    float w[kU];
#pragma unroll
                    for (int u = 0; u < kU; u++) {
                        const uint32_t idx = r.base + p + u;
                        w[u] = density_term(d, pix - sx[idx], piy - sy[idx], piz - sz[idx]);
                    }
#pragma unroll
                    for (int u = 0; u < kU; u++) rho += (p + u < r.cnt) ? w[u] : 0.0f;  // + 0.0f is exact
                }
            }
        d.cur.rv[s].x = rho;                                                     // :59
        if (is_wild(d.gas_const * (rho - d.rest_density))) d.counters[kCntWildP] = 1ull;
    }
                                 });
    visited = wave_sum_u64(visited);
    if ((threadIdx.x & (kWave - 1)) == 0 && visited) atomicAdd(&d.counters[kCntCandidates], visited);
}

// ---- VelPos.compute:46-158 ---------------------------------------------------------------------

// two 512-thread bricks per CU (LDS-limited): 4 waves per SIMD, i.e. at most 128 VGPRs
template <int H, int CAP, int LCAP, int T>
__global__ void __launch_bounds__(T, 2 * T / 256) k_velpos_brick(Dev d, int nbx, int nby)
{
    constexpr int kT = T;
    using HB = Halo<H, T>;
    __shared__ float sx[CAP + kU], sy[CAP + kU], sz[CAP + kU];
    __shared__ uint32_t s_info[HB::NH], s_g0[HB::NH];
    __shared__ uint32_t s_row_start[kRows], s_row_pref[kRows + 1], s_scan[kT / kWave + 1];
    __shared__ uint32_t s_list32[(LCAP / 2) * kT];  // u16 entries, dword column = thread
    uint16_t *list = reinterpret_cast<uint16_t *>(s_list32);

    const BrickGeom g0 = brick_geom(nbx, nby);
    const uint32_t tid = threadIdx.x;
    const int R = (int)d.R;
    for_each_subbrick<H, CAP, T>(d, g0, launch_is_wild(d), sx, sy, sz, s_info, s_g0, s_row_start, s_row_pref, s_scan,
                                 [&](const BrickGeom g, int nx, uint32_t Tn) {
    const int hx0 = g.x0 - H, hy0 = g.y0 - H, hz0 = g.z0 - H;

    for (uint32_t tb = 0; tb < Tn; tb += kT) {                                   // block-uniform
        const uint32_t t = tb + tid;
        bool live = t < Tn;
        const uint32_t s = target_slot(s_row_start, s_row_pref, live ? t : Tn - 1);
        const float pix = d.cur.x[s], piy = d.cur.y[s], piz = d.cur.z[s];        // :57
        const int ix = cell_coord(pix, d.Rm1, R), iy = cell_coord(piy, d.Rm1, R), iz = cell_coord(piz, d.Rm1, R);
        const bool inside = axis_ok(ix, hx0, nx + 2 * H, R) && axis_ok(iy, hy0, HB::HY, R) && axis_ok(iz, hz0, HB::HZ, R);
        if (d.ghost && d.ghost[s]) live = false;                                  // candidates only
        if (live && !inside) enqueue(d, s);
        live = live && inside;                 // dead lanes keep voting but look nothing up
        if (live && (iz < d.zt_lo || iz > d.zt_hi)) atomicAdd(&d.counters[kCntHaloViolation], 1ull);
        const float4 rvi = d.cur.rv[s];
        const float rho_i = rvi.x;                                               // :60
        const float p_i = d.gas_const * (rho_i - d.rest_density);                // :61
        const float vix = rvi.y, viy = rvi.z, viz = rvi.w, viw = d.cur.vw[s];    // :62
        Force f{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};                             // :64-65

        // list entry = (column 0..8) << 7 | position in the column's LDS run
        struct Cand {
            float x, y, z, rho, vx, vy, vz;
            bool self;
        };
        auto fetch = [&](uint32_t e) {
            const uint32_t ent = list[(((e >> 1) * kT + tid) << 1) + (e & 1u)];
            const int c = (int)(ent >> 7);
            uint32_t k = ent & 127u;
            const int zlo = iz - 1 > 0 ? iz - 1 : 0;
            int hc = hcz<H>(g, ix + c / 3 - 1, iy + c % 3 - 1, zlo);
            uint32_t inf = s_info[hc];
            const uint32_t idx = (inf & 0xFFFFu) + k;
            if (k >= (inf >> 16)) {            // which of the (up to 3) cells of the run
                k -= inf >> 16;
                inf = s_info[++hc];
                if (k >= (inf >> 16)) {
                    k -= inf >> 16;
                    ++hc;
                }
            }
            const uint32_t j = s_g0[hc] + k;
            const float4 rv = d.cur.rv[j];     // one 16-byte gather: rho_j, vel_j (:86,:94)
            return Cand{sx[idx], sy[idx], sz[idx], rv.x, rv.y, rv.z, rv.w, j == s};
        };
        // phase 2: pair arithmetic over the listed survivors, in list (= reference) order
        auto phase2 = [&](uint32_t n) {
#if SPH_ABLATE == 2   /* diagnostic: no phase 2 at all */
            f.px += (float)n; return;
#endif
            if (n == 0) return;
            Cand a = fetch(0);
            for (uint32_t e = 0; e < n; e++) {
                Cand b = a;
                if (e + 1 < n) b = fetch(e + 1);       // next survivor's gathers fly during the arithmetic
#if SPH_ABLATE == 1   /* diagnostic: no pair arithmetic */
                f.px += a.x + a.rho; f.py += a.y + a.vx; f.pz += a.z + a.vy; f.vx += a.vz;
#else
                if (!a.self)                                                      // :82
                    pair_force_vals(d, pix - a.x, piy - a.y, piz - a.z, a.rho, a.vx, a.vy, a.vz, vix, viy, viz,
                                    p_i, f);                                      // :86-95
#endif
                a = b;
            }
        };

        // phase 1: distance test against LDS, reference order (:67-81), 4 slots per trip
        uint32_t n = 0;
#if SPH_ABLATE == 3   /* diagnostic: set-up + staging only */
        for (int c = 0; c < 0; c++) {
#else
        for (int c = 0; c < 9; c++) {                                            // wave-uniform
#endif
            Run r = stencil_run<H>(g, s_info, ix + c / 3 - 1, iy + c % 3 - 1, iz, R);
            if (!live) r.cnt = 0;
            uint32_t p = 0;
            while (true) {
                while (p < r.cnt && n + kU <= (uint32_t)LCAP) {
                    float r2[kU];
#pragma unroll
                    for (int u = 0; u < kU; u++) {
                        const uint32_t idx = r.base + p + u;
                        const float rx = pix - sx[idx], ry = piy - sy[idx], rz = piz - sz[idx];
                        r2[u] = (rx * rx + ry * ry) + rz * rz;
                    }
#pragma unroll
                    for (int u = 0; u < kU; u++)
                        if (p + u < r.cnt && r2[u] < d.h2_up) {   // superset of |r| < h (VelPos.compute:37,43)
                            list[(((n >> 1) * kT + tid) << 1) + (n & 1u)] = (uint16_t)((c << 7) | (p + u));
                            n++;
                        }
                    p += kU;
                }
                if (!__any(p < r.cnt)) break;  // every lane finished this column
                phase2(n);                     // some lane's list is nearly full: drain all lists
                n = 0;
            }
        }
        phase2(n);
        if (live) integrate_store(d, s, pix, piy, piz, vix, viy, viz, viw, rho_i, f);  // :100-157
    }
                                 });
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
    hipLaunchKernelGGL((k_density_brick<1, SPH_DENSITY_CAP, SPH_DENSITY_T>), dim3(nbx * nby * nbz), dim3(SPH_DENSITY_T), 0, s, d, (int)nbx, (int)nby);
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
    hipLaunchKernelGGL((k_velpos_brick<2, SPH_VELPOS_CAP, SPH_LCAP, SPH_VELPOS_T>), dim3(nbx * nby * nbz), dim3(SPH_VELPOS_T), 0, s, d, (int)nbx, (int)nby);
    launch_velpos_queue(d, s);
}

}  // namespace sph
