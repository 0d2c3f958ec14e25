// 27-cell neighbour sweeps for gfx950: replace Density.compute::CSMain (:29-60) and
// VelPos.compute::CSMain (:46-158) on the cell-sorted SoA state built by sph_kernels_grid.hip.
//
// EXACT-ORDER kernels (this file): one lane per target particle, candidates visited in the
// reference's order -- x_offset, y_offset, z_offset outer to inner, slots ascending -- with the
// reference's statements evaluated one IEEE fp32 operation at a time (built with
// -ffp-contract=off, correctly rounded / and sqrt), so results are bit-identical to
// oracle/sph_oracle.c.  Because targets are processed in cell-sorted order, the 64 lanes of a
// wave sit in a handful of x-adjacent cells and their candidate reads fall into a few cache lines.
#include "sph_internal.hpp"

namespace sph {

constexpr int kBlock = 256;

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
    // statistics: P-bar = candidates per particle (one atomic per wave)
    unsigned long long v = visited;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) atomicAdd(&d.counters[kCntCandidates], v);
}

// ---- VelPos.compute:46-158 ---------------------------------------------------------------------

__global__ void __launch_bounds__(kBlock) k_velpos_exact(Dev d)
{
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= d.count) return;
    const float epsilon = 1e-6f;                                                 // :5
    const float pix = d.cur.x[s], piy = d.cur.y[s], piz = d.cur.z[s];            // :57
    const int R = (int)d.R;
    const int ix = cell_coord(pix, d.Rm1, R), iy = cell_coord(piy, d.Rm1, R), iz = cell_coord(piz, d.Rm1, R);
    const float rho_i = d.rho[s];                                                // :60
    const float p_i = d.gas_const * (rho_i - d.rest_density);                    // :61
    const float vix = d.cur.vx[s], viy = d.cur.vy[s], viz = d.cur.vz[s], viw = d.cur.vw[s]; // :62
    const float h = d.h, m = d.mass;

    float fpx = 0.0f, fpy = 0.0f, fpz = 0.0f;                                    // :64
    float fvx = 0.0f, fvy = 0.0f, fvz = 0.0f;                                    // :65

    for (int xo = -1; xo <= 1; xo++)                                             // :67-69
        for (int yo = -1; yo <= 1; yo++)
            for (int zo = -1; zo <= 1; zo++) {
                const CellRange cr = cell_range(d, ix + xo, iy + yo, iz + zo);   // :71-76
                for (uint32_t k = 0; k < cr.cnt; k++) {                          // :77-81
                    const uint32_t j = cr.start + k;
                    if (j == s) continue;                                        // :82 (slot <-> id is 1:1)
                    const float rho_j = d.rho[j];                                // :86
                    const float p_j = d.gas_const * (rho_j - d.rest_density);    // :87
                    const float rx = pix - d.cur.x[j], ry = piy - d.cur.y[j], rz = piz - d.cur.z[j]; // :89
                    if (rho_j > epsilon) {                                       // :91
                        const float abs_r = sqrtf((rx * rx + ry * ry) + rz * rz); // :35, :42
                        const float diff_r = h - abs_r;                          // :36
                        float gpx = 0.0f, gpy = 0.0f, gpz = 0.0f;
                        if ((epsilon < diff_r) && (epsilon < abs_r)) {           // :37
                            const float sc = d.c_over_h6 * ((diff_r * diff_r) * diff_r);
                            gpx = (sc * rx) / abs_r;
                            gpy = (sc * ry) / abs_r;
                            gpz = (sc * rz) / abs_r;
                        }
                        const float cp = (p_i + p_j) / (2.0f * rho_j);           // :93
                        fpx += cp * gpx;
                        fpy += cp * gpy;
                        fpz += cp * gpz;
                        const float gv = abs_r < h ? d.c_over_h6 * (h - abs_r) : 0.0f; // :43
                        fvx += ((d.cur.vx[j] - vix) / rho_j) * gv;               // :94
                        fvy += ((d.cur.vy[j] - viy) / rho_j) * gv;
                        fvz += ((d.cur.vz[j] - viz) / rho_j) * gv;
                    }
                }
            }

    if (rho_i > epsilon) {                                                       // :101
        const float sp = m * m / rho_i;                                          // :103
        fpx *= sp; fpy *= sp; fpz *= sp;
        const float sv = d.viscosity * m * m / rho_i;                            // :104
        fvx *= sv; fvy *= sv; fvz *= sv;
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

    float ax = d.gx + ((fpx + fvx) + fwx) / m;                                   // :139
    float ay = d.gy + ((fpy + fvy) + fwy) / m;
    float az = d.gz + ((fpz + fvz) + fwz) / m;
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
}

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

void launch_density(const Dev &d, hipStream_t s)
{
    (void)hipMemsetAsync(d.counters + kCntCandidates, 0, sizeof(unsigned long long), s);
    hipLaunchKernelGGL(k_density_exact, dim3(cdiv(d.count, kBlock)), dim3(kBlock), 0, s, d);
}

void launch_velpos(const Dev &d, hipStream_t s)
{
    hipLaunchKernelGGL(k_velpos_exact, dim3(cdiv(d.count, kBlock)), dim3(kBlock), 0, s, d);
}

}  // namespace sph
