// Per-target 27-cell sweeps on global memory (gfx950): the GENERIC path behind the LDS-tiled brick
// kernels of sph_kernels_brick.hip.  They replace Density.compute::CSMain (:29-60) and
// VelPos.compute::CSMain (:46-158) for the targets the brick kernels hand over through a device
// queue: particles outside the grid ("limbo"), particles that moved more than one cell since the
// grid was built, bricks that do not fit the LDS tile, and whole launches flagged "wild".
// One lane per target, reference accumulation order, one IEEE fp32 operation per source
// operation -- bit-identical to oracle/sph_oracle.c.
//
// VelPos is two-phase.  ~85 % of the listed candidates lie outside the kernel radius h and
// contribute exact zeros (grad_W_press = grad_W_vis = 0, VelPos.compute:37,43).  Phase 1 walks the
// 27 cells in reference order doing only the distance test and appends the survivors to a
// per-lane list in LDS; phase 2 runs the expensive pair arithmetic (1 sqrt + 7 IEEE divisions)
// over the lists with all lanes busy.  Dropping "+= coefficient * 0" is exact as long as the
// coefficient is finite; a device-side "wild" flag (any |v| or |p| >= 1e15, inf or NaN -- set by
// the kernels that produce those values) routes the whole launch through the literal one-phase
// loop instead, so NaN/inf propagation stays bit-identical too (tests: K7, weird inputs).
#include "sph_sweep_common.hpp"

namespace sph {

#ifndef SPH_QUEUE_TRIP
#define SPH_QUEUE_TRIP 8   /* candidates per phase-1 trip (2 / 4 / 8: 45.2 / 44.0 / 43.6 ms per late frame) */
#endif
#ifndef SPH_DQUEUE_TRIP
#define SPH_DQUEUE_TRIP 4
#endif
constexpr int kBlock = 256;
constexpr int kListCap = 64;          // entries per lane; a cell adds at most 32
// The queue kernels are persistent grids of as many blocks as the device keeps resident (queue_grid below); the blocks
// claim the queue in chunks of kBlock entries from a device counter (queue_claim).  Targets differ a lot in cost -- a member
// of a floor cell late in a dam break walks 300+ candidates, a target that merely changed cell ~100 -- so a static deal of
// the chunks leaves blocks idle: 2048 blocks strided (until round 3; 1280 resident) 5.2 ms, 1280 blocks strided 6.6 ms.
__device__ __forceinline__ uint32_t queue_claim(const Dev &d, uint32_t *s_q0)
{
    __syncthreads();                                   // everybody is done with the previous chunk (and with s_q0)
    if (threadIdx.x == 0) *s_q0 = (uint32_t)atomicAdd(&d.counters[kCntQueueHead], (unsigned long long)kBlock);
    __syncthreads();
    const uint32_t q0 = *s_q0;
    __syncthreads();                                   // (the word may be part of a buffer the chunk's work reuses)
    return q0;
}

// ---- Density.compute:29-60 for target slot s -------------------------------------------------------

// RECORD (the density pass of a fresh grid with neighbour lists on, sph_kernels_list.hip): the candidates within h + skin
// also go to the target's list, exactly as k_density_brick<RECORD> writes them -- crowded regions (cells at the 32-cap,
// whose 27-cell neighbourhoods never fit an LDS tile) are where a list saves most: ~130 entries instead of 864 candidates
// in each of the frame's five sub-steps.
template <bool RECORD>
__device__ __forceinline__ uint32_t density_target(const Dev &d, uint32_t s)
{
    const Pos3 pi = load_pos(d, d.cur, s);
    const float pix = pi.x, piy = pi.y, piz = pi.z;
    const int R = (int)d.R;
    const int ix = cell_coord(pix, d.Rm1, R), iy = cell_coord(piy, d.Rm1, R), iz = cell_coord(piz, d.Rm1, R);
    float rho = 0.0f;
    uint32_t visited = 0, gn = 0;
    uint32_t *nl_row = RECORD ? nl_entries(d, s) : nullptr;
    for (int xo = -1; xo <= 1; xo++)
        for (int yo = -1; yo <= 1; yo++)
            for (int zo = -1; zo <= 1; zo++) {
                const CellRange cr = cell_range(d, ix + xo, iy + yo, iz + zo);
                visited += cr.cnt;
                const uint32_t code = (uint32_t)(((xo + 1) * 3 + (yo + 1)) * 3 + (zo + 1));
                // four candidates per trip, their loads issued together (one 16-byte load each where the packed copy exists): a
                // trip per candidate waited for one memory round trip per candidate
                constexpr uint32_t kTrip = SPH_DQUEUE_TRIP;
                for (uint32_t k0 = 0; k0 < cr.cnt; k0 += kTrip) {
                    Pos3 pj[kTrip];
#pragma unroll
                    for (uint32_t u = 0; u < kTrip; u++) {
                        const uint32_t j = cr.start + (k0 + u < cr.cnt ? k0 + u : cr.cnt - 1u);
                        pj[u] = d.cur.p4 ? load_p4(d, d.cur, j) : load_pos(d, d.cur, j);
                    }
#pragma unroll
                    for (uint32_t u = 0; u < kTrip; u++) {
                        if (k0 + u >= cr.cnt) break;
                        const uint32_t j = cr.start + k0 + u;
                        const float rx = pix - pj[u].x, ry = piy - pj[u].y, rz = piz - pj[u].z;
                        const float r2 = (rx * rx + ry * ry) + rz * rz;          // Density.compute:24
                        rho += d.fast ? density_term_r2<true>(d, r2) : density_term_r2<false>(d, r2);
                        if (RECORD && r2 < d.h2_list) {                          // superset of |r| < h + skin
                            if (gn < d.nl_cap) nl_row[gn * kWave] = (code << kNlSlotBits) | j;
                            gn++;
                        }
                    }
                }
            }
    if (RECORD) {
        d.counters[kCntGlobalLists] = 1ull;
        if (gn <= d.nl_cap) d.nl_cnt[s] = (uint16_t)gn;
        else {                                                                   // list too long: its brick keeps scanning
            const uint32_t ux = (uint32_t)(ix < 0 ? 0 : (ix > R - 1 ? R - 1 : ix)), uy = (uint32_t)(iy < 0 ? 0 : (iy > R - 1 ? R - 1 : iy)),
                           uz = (uint32_t)(iz < 0 ? 0 : (iz > R - 1 ? R - 1 : iz));
            d.brick_state[ux / kNlBX + nl_bricks_x(d.R) * (uy / kNlBY + nl_bricks_y(d.R) * (uz / kNlBZ))] = kBrickUnlisted;
            atomicAdd(&d.counters[kCntListTooLong], 1ull);
        }
    }
    d.cur.rv[s].x = rho;                                                         // :59
    // pressure that a later VelPos pass derives from this density (VelPos.compute:61,87)
    if (is_wild_density(d, rho)) d.counters[kCntWildP] = 1ull;
    return visited;
}

// ---- VelPos.compute:46-158 for target slot s (all lanes of the wave must call this) ---------------

__device__ __forceinline__ void velpos_target(const Dev &d, uint32_t s, bool live, uint16_t *list, uint32_t tid)
{
    const Pos3 pi = load_pos(d, d.cur, s);
    const float pix = pi.x, piy = pi.y, piz = pi.z;                              // :57
    const int R = (int)d.R;
    const int ix = cell_coord(pix, d.Rm1, R), iy = cell_coord(piy, d.Rm1, R), iz = cell_coord(piz, d.Rm1, R);
    const float4 rvi = d.cur.rv[s];
    const float rho_i = rvi.x;                                                   // :60
    const float p_i = d.gas_const * (rho_i - d.rest_density);                    // :61
    const float vix = rvi.y, viy = rvi.z, viz = rvi.w, viw = d.cur.vw[s];        // :62
    if (d.ghost && d.ghost[s]) live = false;             // ghosts are candidates only (their owner integrates them)
    if (live && (iz < d.zt_lo || iz > d.zt_hi)) atomicAdd(&d.counters[kCntHaloViolation], 1ull);

    Force f{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};                                 // :64-65

    if (launch_is_wild(d)) {
        // literal loop, :67-98
        for (int xo = -1; xo <= 1; xo++)
            for (int yo = -1; yo <= 1; yo++)
                for (int zo = -1; zo <= 1; zo++) {
                    const CellRange cr = cell_range(d, ix + xo, iy + yo, iz + zo);
                    for (uint32_t k = 0; k < cr.cnt; k++) {
                        const uint32_t j = cr.start + k;
                        if (j == s) continue;                                    // :82 (slot <-> id is 1:1)
                        pair_force(d, j, pix, piy, piz, vix, viy, viz, p_i, f, true);
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
                // four candidates per trip, their loads issued together: one candidate per trip exposed a memory round trip per
                // candidate, which five waves per SIMD do not hide (61 % of the VALU issue rate late in a dam break)
                constexpr uint32_t kTrip = SPH_QUEUE_TRIP;
                for (uint32_t k0 = 0; k0 < cr.cnt; k0 += kTrip) {
                    Pos3 pj[kTrip];
#pragma unroll
                    for (uint32_t u = 0; u < kTrip; u++) {
                        const uint32_t j = cr.start + (k0 + u < cr.cnt ? k0 + u : cr.cnt - 1u);
                        pj[u] = d.cur.p4 ? load_p4(d, d.cur, j) : load_pos(d, d.cur, j);   // (wave-uniform) the packed copy: one 16-byte load
                    }
#pragma unroll
                    for (uint32_t u = 0; u < kTrip; u++) {
                        const uint32_t k = k0 + u, j = cr.start + k;
                        const float rx = pix - pj[u].x, ry = piy - pj[u].y, rz = piz - pj[u].z;
                        const float r2 = (rx * rx + ry * ry) + rz * rz;
                        // branch-free append: always store at n, keep it only if the slot is a survivor (superset of |r| < h,
                        // VelPos.compute:37,43); n <= kListCap - 32 on entry to a cell leaves room
                        list[(((n >> 1) * kBlock + tid) << 1) + (n & 1u)] = (uint16_t)((ci << 5) | k);
                        n += (uint32_t)(k < cr.cnt) & (uint32_t)(j != s) & (uint32_t)(r2 < d.h2_up);   // (no short circuit: no branch)
                    }
                }
                if (__any(n > (uint32_t)(kListCap - kMaxPerVoxel))) {            // next cell may not fit
                    ci++;
                    break;
                }
            }
            // phase 2: pair arithmetic over the survivors, same order; the next survivor's fields are on their way meanwhile
            // (two survivors per trip, as in the brick kernel: no gain here, 43.4 vs 43.6 ms per late frame)
            struct Cand {
                Pos3 p;
                float4 rv;
            };
            auto fetch = [&](uint32_t e) {
                const uint32_t ent = list[(((e >> 1) * kBlock + tid) << 1) + (e & 1u)];
                const int c = (int)(ent >> 5);
                const int xo = c / 9 - 1, yo = (c / 3) % 3 - 1, zo = c % 3 - 1;
                const uint32_t cell = (uint32_t)(ix + xo) + (uint32_t)(iy + yo) * d.R + (uint32_t)(iz + zo) * d.R * d.R;
                const uint32_t j = d.cell_start[cell] + (ent & 31u);
                return Cand{d.cur.p4 ? load_p4(d, d.cur, j) : load_pos(d, d.cur, j), d.cur.rv[j]};   // :86-89
            };
            if (n) {
                Cand a = fetch(0);
                for (uint32_t e = 0; e < n; e++) {
                    const Cand b = fetch(e + 1 < n ? e + 1 : e);
                    pair_force_vals(d, pix - a.p.x, piy - a.p.y, piz - a.p.z, a.rv.x, a.rv.y, a.rv.z, a.rv.w, vix, viy, viz, p_i, f, true, false);
                    a = b;
                }
            }
        }
    }
    Pos3 pn{pix, piy, piz};
    if (live) {
        if (d.fast) pn = integrate_store<-1, true>(d, s, pix, piy, piz, vix, viy, viz, viw, rho_i, f);   // tolerance mode (launch-uniform)
        else pn = integrate_store(d, s, pix, piy, piz, vix, viy, viz, viw, rho_i, f);
    }
    if (d.box_mode && d.nl_mode == kNlUse) {             // (launch-uniform) per-brick displacement ranges: the brick of the target's BIN
        const Pos3 pb = load_binned(d, s);
        const uint32_t key = bucket_key(d, pb.x, pb.y, pb.z);
        const bool in_grid = live && key < d.num_cells;  // (limbo particles are nobody's candidates)
        box_update_wave(d, in_grid, in_grid ? nl_brick_of_key(d, key) : 0u, pn.x - pix, pn.y - piy, pn.z - piz);
    }
}

// ---- kernels over every slot (used when the brick path is disabled) ----------------------------------

__global__ void __launch_bounds__(kBlock) k_density_exact(Dev d)
{
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    unsigned long long v = 0;
    if (s < d.count && !(d.ghost && d.ghost[s])) v = density_target<false>(d, s);   // (no bricks: no lists)
    stat_add(d, kStatCandidates, v);
}

__global__ void __launch_bounds__(kBlock) k_velpos_exact(Dev d)
{
    // per-lane candidate lists, u16 entries (cell 0..26) << 5 | slot 0..31, two per dword;
    // dword column = thread, so equal-entry accesses of a wave never share a bank
    __shared__ uint32_t lds_list[(kListCap / 2) * kBlock];
    const uint32_t s_raw = blockIdx.x * kBlock + threadIdx.x;
    const bool live = s_raw < d.count;
    velpos_target(d, live ? s_raw : d.count - 1, live, reinterpret_cast<uint16_t *>(lds_list), threadIdx.x);
}

// ---- kernels over the device queue filled by the brick kernels ---------------------------------------

__global__ void __launch_bounds__(kBlock) k_density_queue(Dev d)
{
    __shared__ uint32_t s_q0;
    const uint32_t qlen = (uint32_t)d.counters[kCntQueue];
    if (qlen == 0u) return;                            // (usual case: nothing was handed over)
    unsigned long long v = 0;
    for (uint32_t q0 = queue_claim(d, &s_q0); q0 < qlen; q0 = queue_claim(d, &s_q0)) {   // block-uniform
        const uint32_t q = q0 + threadIdx.x;
        if (q < qlen && !(d.ghost && d.ghost[d.queue[q]]))
            v += d.nl_mode == kNlBuild ? density_target<true>(d, d.queue[q]) : density_target<false>(d, d.queue[q]);
    }
    stat_add(d, kStatCandidates, v);
}

__global__ void __launch_bounds__(kBlock) k_velpos_queue(Dev d)
{
    __shared__ uint32_t lds_list[(kListCap / 2) * kBlock];   // 32 KB: five blocks per CU (one word more would make it four)
    const uint32_t qlen = (uint32_t)d.counters[kCntQueue];
    if (qlen == 0u) return;                            // (block-uniform)
    for (uint32_t q0 = queue_claim(d, lds_list); q0 < qlen; q0 = queue_claim(d, lds_list)) {   // block-uniform
        const uint32_t q = q0 + threadIdx.x;
        const bool live = q < qlen;
        velpos_target(d, d.queue[live ? q : qlen - 1], live, reinterpret_cast<uint16_t *>(lds_list), threadIdx.x);
    }
}

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return a ? (a + b - 1) / b : 1u; }  // never an empty grid

// resident blocks of a queue kernel on the current device (asked once per kernel and device)
template <typename K>
static uint32_t queue_grid(K kernel, uint32_t &cached)
{
    if (cached == 0u) {
        int dev = 0, cus = 0, per_cu = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        cached = (uint32_t)(cus > 0 ? cus : 256) * (uint32_t)per_cu;
    }
    return cached;
}

void launch_density_all(const Dev &d, hipStream_t s)
{
    hipLaunchKernelGGL(k_density_exact, dim3(cdiv(d.count, kBlock)), dim3(kBlock), 0, s, d);
}

void launch_velpos_all(const Dev &d, hipStream_t s)
{
    hipLaunchKernelGGL(k_velpos_exact, dim3(cdiv(d.count, kBlock)), dim3(kBlock), 0, s, d);
}

void launch_density_queue(const Dev &d, hipStream_t s)
{
    static uint32_t resident = 0;
    const uint32_t g = cdiv(d.count, kBlock), cap = queue_grid(k_density_queue, resident);
    hipLaunchKernelGGL(k_density_queue, dim3(g < cap ? g : cap), dim3(kBlock), 0, s, d);
}

void launch_velpos_queue(const Dev &d, hipStream_t s)
{
    static uint32_t resident = 0;
    const uint32_t g = cdiv(d.count, kBlock), cap = queue_grid(k_velpos_queue, resident);
    hipLaunchKernelGGL(k_velpos_queue, dim3(g < cap ? g : cap), dim3(kBlock), 0, s, d);
}

}  // namespace sph
