// 27-cell neighbour sweeps for gfx950: replace Density.compute::CSMain (:29-60) and
// VelPos.compute::CSMain (:46-158) on the cell-sorted SoA state built by sph_kernels_grid.hip.
//
// EXACT-ORDER kernels: one lane per target particle; every contribution is added in the
// reference's order -- x_offset, y_offset, z_offset outer to inner, slots ascending -- and every
// reference statement is evaluated one IEEE fp32 operation at a time (built with
// -ffp-contract=off, correctly rounded / and sqrt), so results are bit-identical to
// oracle/sph_oracle.c.  Targets are processed in cell-sorted order, so the 64 lanes of a wave sit
// in a handful of x-adjacent cells and their candidate reads fall into a few cache lines.
//
// k_velpos_exact is two-phase.  ~85 % of the listed candidates lie outside the kernel radius h and
// contribute exact zeros (grad_W_press = grad_W_vis = 0, VelPos.compute:37,43).  Phase 1 walks the
// 27 cells in reference order doing only the distance test and appends the survivors to a
// per-lane list in LDS; phase 2 runs the expensive pair arithmetic (1 sqrt + 7 IEEE divisions)
// over the lists with all lanes busy.  Dropping "+= coefficient * 0" is exact as long as the
// coefficient is finite; a device-side "wild" flag (any |v| or |p| >= 1e15, inf or NaN -- set by
// the kernels that produce those values) routes the whole launch through the literal one-phase
// loop instead, so NaN/inf propagation stays bit-identical too (tests: K7, weird inputs).
#include "sph_internal.hpp"

namespace sph {

constexpr int kBlock = 256;
constexpr int kListCap = 64;          // entries per lane; a cell adds at most 32
constexpr float kWildLimit = 1e15f;

struct CellRange {
    uint32_t start, cnt;
};

// Bounds test of Density.compute:46 / VelPos.compute:73 + the bucket row of that cell:
// the first min(count, 32) slots, ascending particle id.
__device__ __forceinline__ CellRange cell_range(const Dev &d, int jx, int jy, int jz)
{
    CellRange r{0u, 0u};
    const int R = (int)d.R;
    if (jx >= 0 && jy >= 0 && jz >= 0 && jx < R && jy < R && jz < R) {
        const uint32_t c = (uint32_t)jx + (uint32_t)jy * d.R + (uint32_t)jz * d.R * d.R;
        r.start = d.cell_start[c];
        const uint32_t n = d.cell_start[c + 1] - r.start;
        r.cnt = n < kMaxPerVoxel ? n : kMaxPerVoxel;
    }
    return r;
}

__device__ __forceinline__ bool is_wild(float v) { return !(fabsf(v) < kWildLimit); }

// ---- Density.compute:29-60 ---------------------------------------------------------------------

__global__ void __launch_bounds__(kBlock) k_density_exact(Dev d)
{
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= d.count) return;
    const float pix = d.cur.x[s], piy = d.cur.y[s], piz = d.cur.z[s];
    const int R = (int)d.R;
    const int ix = cell_coord(pix, d.Rm1, R), iy = cell_coord(piy, d.Rm1, R), iz = cell_coord(piz, d.Rm1, R);
    float rho = 0.0f;
    uint32_t visited = 0;
    for (int xo = -1; xo <= 1; xo++)
        for (int yo = -1; yo <= 1; yo++)
            for (int zo = -1; zo <= 1; zo++) {
                const CellRange cr = cell_range(d, ix + xo, iy + yo, iz + zo);
                visited += cr.cnt;
                for (uint32_t k = 0; k < cr.cnt; k++) {
                    const uint32_t j = cr.start + k;
                    const float rx = pix - d.cur.x[j], ry = piy - d.cur.y[j], rz = piz - d.cur.z[j];
                    const float abs_r2 = (rx * rx + ry * ry) + rz * rz;         // :24
                    const float diff_r2 = d.h2 - abs_r2;                         // :25
                    const float w = 0.0f < diff_r2                               // :26
                                        ? d.c_poly6 * ((diff_r2 * diff_r2) * diff_r2) / d.h9
                                        : 0.0f;
                    rho += d.mass * w;                                           // :54
                }
            }
    d.rho[s] = rho;                                                              // :59
    // pressure that a later VelPos pass derives from this density (VelPos.compute:61,87)
    if (is_wild(d.gas_const * (rho - d.rest_density))) d.counters[kCntWildP] = 1ull;
    // statistics: P-bar = candidates per particle (one atomic per wave)
    unsigned long long v = visited;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) atomicAdd(&d.counters[kCntCandidates], v);
}

// ---- VelPos.compute:86-95: one listed candidate j != i -------------------------------------------

struct Force {
    float px, py, pz;  // f_press
    float vx, vy, vz;  // f_vis
};

__device__ __forceinline__ void pair_force(const Dev &d, uint32_t j, float pix, float piy, float piz,
                                           float vix, float viy, float viz, float p_i, Force &f)
{
    const float epsilon = 1e-6f;                                                 // :5
    const float h = d.h;
    const float rho_j = d.rho[j];                                                // :86
    const float p_j = d.gas_const * (rho_j - d.rest_density);                    // :87
    const float rx = pix - d.cur.x[j], ry = piy - d.cur.y[j], rz = piz - d.cur.z[j]; // :88-89
    if (rho_j > epsilon) {                                                       // :91
        const float abs_r = sqrtf((rx * rx + ry * ry) + rz * rz);                // :35, :42
        const float diff_r = h - abs_r;                                          // :36
        float gpx = 0.0f, gpy = 0.0f, gpz = 0.0f;
        if ((epsilon < diff_r) && (epsilon < abs_r)) {                           // :37
            const float sc = d.c_over_h6 * ((diff_r * diff_r) * diff_r);
            gpx = (sc * rx) / abs_r;
            gpy = (sc * ry) / abs_r;
            gpz = (sc * rz) / abs_r;
        }
        const float cp = (p_i + p_j) / (2.0f * rho_j);                           // :93
        f.px += cp * gpx;
        f.py += cp * gpy;
        f.pz += cp * gpz;
        const float gv = abs_r < h ? d.c_over_h6 * (h - abs_r) : 0.0f;           // :43
        f.vx += ((d.cur.vx[j] - vix) / rho_j) * gv;                              // :94
        f.vy += ((d.cur.vy[j] - viy) / rho_j) * gv;
        f.vz += ((d.cur.vz[j] - viz) / rho_j) * gv;
    }
}

// ---- VelPos.compute:100-157: scaling, wall force, NaN guard, integrate, clamp, store --------------

__device__ __forceinline__ void integrate_store(const Dev &d, uint32_t s, float pix, float piy, float piz,
                                                float vix, float viy, float viz, float viw, float rho_i,
                                                Force f)
{
    const float epsilon = 1e-6f;
    const float h = d.h, m = d.mass;
    if (rho_i > epsilon) {                                                       // :101
        const float sp = m * m / rho_i;                                          // :103
        f.px *= sp; f.py *= sp; f.pz *= sp;
        const float sv = d.viscosity * m * m / rho_i;                            // :104
        f.vx *= sv; f.vy *= sv; f.vz *= sv;
    }
    float fwx = 0.0f, fwy = 0.0f, fwz = 0.0f;                                    // :107
    float wrx = 0.0f, wry = 0.0f, wrz = 0.0f;                                    // :108
    if (pix < h) wrx += h - pix;                                                 // :110-113
    else if (pix > 1.0f - h) wrx += 1.0f - pix - h;
    if (piy < h) wry += h - piy;                                                 // :115-118
    else if (piy > 1.0f - h) wry += 1.0f - piy - h;
    if (piz < h) wrz += h - piz;                                                 // :120-123
    else if (piz > 1.0f - h) wrz += 1.0f - piz - h;
    const float max_r = fmaxf(fabsf(wrx), fmaxf(fabsf(wry), fabsf(wrz)));        // :125-126
    const float dpx = (wrx != 0.0f) ? d.damping : 0.0f;                          // :128-131
    const float dpy = (wry != 0.0f) ? d.damping : 0.0f;
    const float dpz = (wrz != 0.0f) ? d.damping : 0.0f;
    if (max_r > 0.0f) {                                                          // :133
        const float dd = (dpx * vix + dpy * viy) + dpz * viz;                    // :135 scalar
        fwx = wrx * d.stiffness - dd;
        fwy = wry * d.stiffness - dd;
        fwz = wrz * d.stiffness - dd;
        fwx *= m; fwy *= m; fwz *= m;                                            // :136
    }
    const float ax = d.gx + ((f.px + f.vx) + fwx) / m;                           // :139
    const float ay = d.gy + ((f.py + f.vy) + fwy) / m;
    const float az = d.gz + ((f.pz + f.vz) + fwz) / m;
    float vox, voy, voz, vow;
    if (ax != ax || ay != ay || az != az) {                                      // :143
        vox = 0.0f; voy = 0.0f; voz = 0.0f; vow = 0.003f;                        // :145-146
        atomicAdd(&d.counters[kCntNaN], 1ull);
    } else {
        vox = vix + ax * d.dt;                                                   // :150
        voy = viy + ay * d.dt;
        voz = viz + az * d.dt;
        vow = viw + 1.0f;
    }
    d.nxt.x[s] = clamp01(pix + d.dt * vox);                                      // :153-154
    d.nxt.y[s] = clamp01(piy + d.dt * voy);
    d.nxt.z[s] = clamp01(piz + d.dt * voz);
    d.nxt.vx[s] = vox; d.nxt.vy[s] = voy; d.nxt.vz[s] = voz; d.nxt.vw[s] = vow;  // :156-157
    if (is_wild(vox) || is_wild(voy) || is_wild(voz)) d.counters[kCntWildV0 + (d.wild_v_cur ^ 1)] = 1ull;
}

// ---- VelPos.compute:46-158 ---------------------------------------------------------------------

__global__ void __launch_bounds__(kBlock) k_velpos_exact(Dev d)
{
    // per-lane candidate lists, u16 entries (cell 0..26) << 5 | slot 0..31, two per dword;
    // dword column = thread, so equal-entry accesses of a wave never share a bank
    __shared__ uint32_t lds_list[(kListCap / 2) * kBlock];
    uint16_t *list = reinterpret_cast<uint16_t *>(lds_list);

    const uint32_t tid = threadIdx.x;
    const uint32_t s_raw = blockIdx.x * kBlock + tid;
    const bool live = s_raw < d.count;
    const uint32_t s = live ? s_raw : d.count - 1;       // dead lanes shadow the last particle, store nothing
    const float pix = d.cur.x[s], piy = d.cur.y[s], piz = d.cur.z[s];            // :57
    const int R = (int)d.R;
    const int ix = cell_coord(pix, d.Rm1, R), iy = cell_coord(piy, d.Rm1, R), iz = cell_coord(piz, d.Rm1, R);
    const float rho_i = d.rho[s];                                                // :60
    const float p_i = d.gas_const * (rho_i - d.rest_density);                    // :61
    const float vix = d.cur.vx[s], viy = d.cur.vy[s], viz = d.cur.vz[s], viw = d.cur.vw[s]; // :62

    Force f{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};                                 // :64-65

    const bool wild = d.force_slow != 0u || d.counters[kCntWildP] != 0ull ||
                      d.counters[kCntWildV0 + d.wild_v_cur] != 0ull;             // launch-uniform
    if (wild) {
        // literal loop, :67-98
        for (int xo = -1; xo <= 1; xo++)
            for (int yo = -1; yo <= 1; yo++)
                for (int zo = -1; zo <= 1; zo++) {
                    const CellRange cr = cell_range(d, ix + xo, iy + yo, iz + zo);
                    for (uint32_t k = 0; k < cr.cnt; k++) {
                        const uint32_t j = cr.start + k;
                        if (j == s) continue;                                    // :82 (slot <-> id is 1:1)
                        pair_force(d, j, pix, piy, piz, vix, viy, viz, p_i, f);
                    }
                }
    } else {
        int ci = 0;                                                              // wave-uniform
        while (ci < 27) {
            // phase 1: distance test only, reference order
            uint32_t n = 0;
            for (; ci < 27; ci++) {
                const int xo = ci / 9 - 1, yo = (ci / 3) % 3 - 1, zo = ci % 3 - 1;
                const CellRange cr = cell_range(d, ix + xo, iy + yo, iz + zo);
                for (uint32_t k = 0; k < cr.cnt; k++) {
                    const uint32_t j = cr.start + k;
                    const float rx = pix - d.cur.x[j], ry = piy - d.cur.y[j], rz = piz - d.cur.z[j];
                    const float r2 = (rx * rx + ry * ry) + rz * rz;
                    if (j != s && r2 < d.h2_up) {        // superset of |r| < h (VelPos.compute:37,43)
                        list[(((n >> 1) * kBlock + tid) << 1) + (n & 1u)] = (uint16_t)((ci << 5) | k);
                        n++;
                    }
                }
                if (__any(n > (uint32_t)(kListCap - kMaxPerVoxel))) {            // next cell may not fit
                    ci++;
                    break;
                }
            }
            // phase 2: pair arithmetic over the survivors, same order
            for (uint32_t e = 0; e < n; e++) {
                const uint32_t ent = list[(((e >> 1) * kBlock + tid) << 1) + (e & 1u)];
                const int c = (int)(ent >> 5);
                const int xo = c / 9 - 1, yo = (c / 3) % 3 - 1, zo = c % 3 - 1;
                const uint32_t cell = (uint32_t)(ix + xo) + (uint32_t)(iy + yo) * d.R + (uint32_t)(iz + zo) * d.R * d.R;
                const uint32_t j = d.cell_start[cell] + (ent & 31u);
                pair_force(d, j, pix, piy, piz, vix, viy, viz, p_i, f);
            }
        }
    }
    if (live) integrate_store(d, s, pix, piy, piz, vix, viy, viz, viw, rho_i, f);
}

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

void launch_density(const Dev &d, hipStream_t s)
{
    (void)hipMemsetAsync(d.counters + kCntCandidates, 0, sizeof(unsigned long long), s);
    (void)hipMemsetAsync(d.counters + kCntWildP, 0, sizeof(unsigned long long), s);
    hipLaunchKernelGGL(k_density_exact, dim3(cdiv(d.count, kBlock)), dim3(kBlock), 0, s, d);
}

void launch_velpos(const Dev &d, hipStream_t s)
{
    // velocity wildness of this pass's output is collected in the other slot
    (void)hipMemsetAsync(d.counters + kCntWildV0 + (d.wild_v_cur ^ 1), 0, sizeof(unsigned long long), s);
    hipLaunchKernelGGL(k_velpos_exact, dim3(cdiv(d.count, kBlock)), dim3(kBlock), 0, s, d);
}

}  // namespace sph
