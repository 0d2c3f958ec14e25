// Frame-coherent neighbour lists for VelPos (gfx950).
//
// The reference freezes the bucket for the 5 sub-steps of a frame (SphFluidSimulation.cs:98-102): every sub-step
// walks the SAME bins, only the positions inside them have moved.  The first sub-step (k_velpos_brick<BUILD>) therefore
// records, per target and in reference order, every listed candidate of its 27 bins with |r| < h + skin.  A later
// sub-step of the same frame can take a target's pair sums from that list instead of re-scanning ~100 candidates,
// and gets the reference's bits, as long as
//   (1) no unlisted candidate of those bins can have come within h: its distance was >= h + skin and the relative
//       displacement |delta_i - delta_j| of any pair around the target's brick is bounded by rho <= skin
//       (k_chunk_box: per-axis displacement range of every 64 consecutive slots; k_brick_activity: merged over the
//       slot runs that cover the brick's bins and their neighbours: one cell around the brick);
//   (2) the target's stencil is still known: its current cell is its bin or a face/edge/corner neighbour.  Bins that
//       left the stencil are masked per entry (the reference no longer visits them).  A bin that ENTERED it was not
//       scanned when the lists were recorded: k_velpos_list_changers runs the distance pre-test over the listed
//       particles of those <= 19 bins for every target that changed cell (~1 % per sub-step) and, if one is within h --
//       practically never: it was more than a cell away then -- queues the target for the generic scan, whose result
//       replaces the one k_velpos_list wrote (both read the same inputs).
// Everything else -- bricks that are not calm, targets failing (2), bricks with an unlistable target -- takes the
// scanning kernels exactly as without lists.  Out-of-range entries contribute exact zeros (VelPos.compute:37,43 and
// the "wild" guard of sph_kernels_sweep.hip), so which path served a target does not change a bit of the result.
#include "sph_list_common.hpp"

namespace sph {

constexpr int kBlock = 256;
#ifndef SPH_LIST_BLOCK
#define SPH_LIST_BLOCK 256
#endif
constexpr int kListBlock = SPH_LIST_BLOCK;
#ifndef SPH_LIST_MINWAVES
#define SPH_LIST_MINWAVES 4
#endif

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): give every XCD one contiguous range of the
// slot-ordered work, so that the rows and planes a block's gathers reach are in the L2 its neighbours filled.
__device__ __forceinline__ uint32_t xcd_contiguous_block()
{
    constexpr uint32_t kXcd = 8;
    const uint32_t nb = gridDim.x, q = nb / kXcd, r = nb % kXcd;
    const uint32_t xcd = blockIdx.x % kXcd, idx = blockIdx.x / kXcd;
    return xcd * q + (xcd < r ? xcd : r) + idx;
}

// One wave per 64 consecutive slots: per-axis range of the displacement since the grid build.
__global__ void __launch_bounds__(kBlock) k_chunk_box(Dev d)
{
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (s < d.count) {
        const Pos3 pc = load_pos(d, d.cur, s), pb = load_binned(d, s);
        const float dl[3] = {pc.x - pb.x, pc.y - pb.y, pc.z - pb.z};
#pragma unroll
        for (int a = 0; a < 3; a++) lo[a] = hi[a] = dl[a];
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int o = kChunkSlots / 2; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, kWave));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, kWave));
        }
    const uint32_t chunk = s / kChunkSlots;
    if ((threadIdx.x & (kChunkSlots - 1)) == 0 && chunk < d.n_chunks) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            d.chunk_box[(size_t)a * d.n_chunks + chunk] = lo[a];
            d.chunk_box[(size_t)(3 + a) * d.n_chunks + chunk] = hi[a];
        }
    }
}

// One wave per brick (4 bricks per block), one lane per (y,z) row of the brick's bins and their neighbours (6 x 6 rows; a
// row of bins is one run of slots): the per-axis ranges of the displacement over the 64-slot chunks those runs touch bound
// |delta_i - delta_j| for a target i of the brick and any j in its recorded stencil (condition 1).
// calm <=> every target of the brick has its list and the Euclidean bound rho <= skin_ok.
__global__ void __launch_bounds__(kBlock) k_brick_activity(Dev d, int nbx, int nby, int first, int n_bricks)
{
    __shared__ uint32_t s_act[2];
    const int b_raw = first + blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1), R = (int)d.R;
    const bool exists = b_raw < first + n_bricks;  // wave-uniform; the launch covers the brick layers that hold targets
    const int b = exists ? b_raw : first + n_bricks - 1;
    const bool blocked = nl_blocked(d);            // then the scanning kernel takes every brick
    if (threadIdx.x == 0) s_act[0] = 0;
    __syncthreads();
    const int x0 = (b % nbx) * kNlBX, y0 = ((b / nbx) % nby) * kNlBY, z0 = (b / (nbx * nby)) * kNlBZ;
    constexpr int kRY = kNlBY + 2, kRZ = kNlBZ + 2;
    static_assert(kRY * kRZ <= kWave, "one lane per row");
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool own_nonempty = false;
    if (d.nl_still) lo[0] = lo[1] = lo[2] = hi[0] = hi[1] = hi[2] = 0.0f;   // first sub-step: every displacement is zero, no chunk boxes yet
    if (lane < kRY * kRZ) {
        const int cy = y0 - 1 + lane % kRY, cz = z0 - 1 + lane / kRY;
        if (cy >= 0 && cz >= 0 && cy < R && cz < R) {
            const int xa = x0 - 1 > 0 ? x0 - 1 : 0, xb = x0 + kNlBX < R - 1 ? x0 + kNlBX : R - 1;   // bins xa..xb
            const uint32_t rk = (uint32_t)cy * d.R + (uint32_t)cz * d.R * d.R;
            const uint32_t s0 = d.cell_start[rk + (uint32_t)xa], s1 = d.cell_start[rk + (uint32_t)xb + 1u];
            if (s1 > s0) {
                own_nonempty = cy >= y0 && cy < y0 + kNlBY && cz >= z0 && cz < z0 + kNlBZ;   // (roughly: the row)
                if (!d.nl_still)
                for (uint32_t c = s0 / kChunkSlots; c <= (s1 - 1) / kChunkSlots; c++)
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        lo[a] = fminf(lo[a], d.chunk_box[(size_t)a * d.n_chunks + c]);
                        hi[a] = fmaxf(hi[a], d.chunk_box[(size_t)(3 + a) * d.n_chunks + c]);
                    }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, kWave));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, kWave));
        }
    const bool nonempty = __any(own_nonempty);
    uint32_t my_pos = 0xFFFFFFFFu;
    if (lane == 0 && exists) {
        float r2 = 0.0f, r[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            r[a] = (hi[a] >= lo[a] ? hi[a] - lo[a] : 0.0f) * 1.00001f;   // empty neighbourhood: nothing moved
            r2 += r[a] * r[a];
        }
        const float rho = sqrtf(r2) * 1.00001f;
        const uint8_t st = d.brick_state[b] & kBrickUnlisted;
        const bool active = blocked || st != 0 || !(rho <= d.skin_ok);   // NaN-safe
        d.brick_state[b] = st | (active ? kBrickActive : 0) | (nonempty ? 0 : kBrickEmpty);   // (Empty: k_velpos_tile need not look)
        if (active && nonempty && !blocked) atomicAdd(&d.counters[kCntActiveBricks], 1ull);
        if (active) my_pos = atomicAdd(&s_act[0], 1u);
    }
    // the block's active bricks go to the work list of the scanning kernel with one atomic
    __syncthreads();
    if (threadIdx.x == 0 && s_act[0]) s_act[1] = (uint32_t)atomicAdd(&d.counters[kCntActiveCount], (unsigned long long)s_act[0]);
    __syncthreads();
    if (my_pos != 0xFFFFFFFFu) d.active_list[s_act[1] + my_pos] = (uint32_t)b;
}

// k_brick_activity with PER-BRICK displacement ranges (Dev::box_mode): HALF a wave per brick (8 bricks per block), lane l < 27
// of the half looks at neighbour brick l.  A brick's range so far = the sum of its per-sub-step ranges (the sums up to the previous
// sub-step + the range the previous sub-step's kernels filled): a particle's displacement since the grid build is the sum of its
// sub-step displacements, so it lies in the sum of the ranges.  The half writes its OWN brick's new sums and resets its own
// per-sub-step range for the kernels of this sub-step (the other parity of Dev::boxrec: nobody reads that now).
// calm <=> every target of the brick has its list and the Euclidean norm of the ranges over the 27 bricks is <= skin_ok.
// z-slab contexts: the launch also covers the brick layers made of ghost planes only (t_first, t_count: the bricks with target
// planes) -- their ranges are summed and reset like the others (the ghost refresh adds the ghosts' displacements,
// k_slab_unpack_update), they have no state and never go to the work list.
__global__ void __launch_bounds__(kBlock) k_brick_activity_box(Dev d, int nbx, int nby, int first, int n_bricks, int t_first, int t_count)
{
    __shared__ uint32_t s_act[2];
    constexpr int kHalf = 32, kPerBlock = kBlock / kHalf;
    const int b_raw = first + blockIdx.x * kPerBlock + threadIdx.x / kHalf, lane = threadIdx.x & (kHalf - 1), R = (int)d.R;
    const bool exists = b_raw < first + n_bricks;  // uniform per half; the launch covers the brick layers that hold targets
    const int b = exists ? b_raw : first + n_bricks - 1;
    const bool blocked = nl_blocked(d);            // then the scanning kernel takes every brick
    if (threadIdx.x == 0) s_act[0] = 0;
    __syncthreads();
    const int bxi = b % nbx, byi = (b / nbx) % nby, bzi = b / (nbx * nby);
    const uint32_t nb = d.n_bricks, par = d.box_par, prev = par ^ 1u;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float mlo[3] = {0.0f, 0.0f, 0.0f}, mhi[3] = {0.0f, 0.0f, 0.0f};        // (own brick, kept by lane 13)
    bool own_nonempty = false;
    if (d.nl_still) {
        // first sub-step: every displacement is zero.  Which bricks hold targets: from the rows of its bins, once per frame
        // (later sub-steps keep the flag)
        lo[0] = lo[1] = lo[2] = hi[0] = hi[1] = hi[2] = 0.0f;
        if (lane < kNlBY * kNlBZ) {
            const int x0 = bxi * kNlBX, cy = byi * kNlBY + lane % kNlBY, cz = bzi * kNlBZ + lane / kNlBY;
            if (cy < R && cz < R && x0 < R) {
                const int xb = x0 + kNlBX - 1 < R - 1 ? x0 + kNlBX - 1 : R - 1;
                const uint32_t rk = (uint32_t)cy * d.R + (uint32_t)cz * d.R * d.R;
                own_nonempty = d.cell_start[rk + (uint32_t)xb + 1u] > d.cell_start[rk + (uint32_t)x0];
            }
        }
    } else {
        own_nonempty = !(d.brick_state[b] & kBrickEmpty);
        if (lane < 27) {
            const int bx = bxi + lane % 3 - 1, by = byi + (lane / 3) % 3 - 1, bz = bzi + lane / 9 - 1;
            if (bx >= 0 && by >= 0 && bz >= 0 && bx < nbx && by < nby && bz < (int)nl_bricks_z(d.R)) {
                const uint32_t n = (uint32_t)bx + (uint32_t)nbx * ((uint32_t)by + (uint32_t)nby * (uint32_t)bz);
                const uint4 *rp = reinterpret_cast<const uint4 *>(d.boxrec + ((size_t)prev * nb + n) * kBoxRecWords);
                const uint4 w0 = rp[0], w1 = rp[1], w2 = rp[2];      // keys lo xyz, hi x | hi yz, sums lo xy | sums lo z, hi xyz
                const uint32_t keys[6] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y};
                const float sums[6] = {__uint_as_float(w1.z), __uint_as_float(w1.w), __uint_as_float(w2.x), __uint_as_float(w2.y),
                                       __uint_as_float(w2.z), __uint_as_float(w2.w)};
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    mlo[a] = sums[a]; mhi[a] = sums[3 + a];
                    // (a brick nobody was integrated in holds no particle -- every particle is some kernel's target in every
                    // sub-step -- and stays out of the union)
                    if (keys[a] != kBoxKeyEmptyLo || keys[3 + a] != kBoxKeyEmptyHi) {
                        mlo[a] = lo[a] = sums[a] + key_float(keys[a]);
                        mhi[a] = hi[a] = sums[3 + a] + key_float(keys[3 + a]);
                    }
                }
            }
        }
    }
    if (exists && lane == 13) {                        // the centre: this half's own brick
        uint4 *wp = reinterpret_cast<uint4 *>(d.boxrec + ((size_t)par * nb + (uint32_t)b) * kBoxRecWords);
        wp[0] = make_uint4(kBoxKeyEmptyLo, kBoxKeyEmptyLo, kBoxKeyEmptyLo, kBoxKeyEmptyHi);
        wp[1] = make_uint4(kBoxKeyEmptyHi, kBoxKeyEmptyHi, __float_as_uint(mlo[0]), __float_as_uint(mlo[1]));
        wp[2] = make_uint4(__float_as_uint(mlo[2]), __float_as_uint(mhi[0]), __float_as_uint(mhi[1]), __float_as_uint(mhi[2]));
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int o = kHalf / 2; o > 0; o >>= 1) {      // (xor with < 32 stays inside the half)
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, kWave));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, kWave));
        }
    const unsigned long long ne = __ballot(own_nonempty);
    const bool nonempty = ((ne >> (threadIdx.x & kHalf)) & 0xFFFFFFFFull) != 0ull;   // (bit 5 of the lane index: which half)
    uint32_t my_pos = 0xFFFFFFFFu;
    if (lane == 0 && exists && b >= t_first && b < t_first + t_count) {
        float r2 = 0.0f, r[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            r[a] = (hi[a] >= lo[a] ? hi[a] - lo[a] : 0.0f) * 1.00001f;   // empty neighbourhood: nothing moved
            r2 += r[a] * r[a];
        }
        const float rho = sqrtf(r2) * 1.00001f;
        const uint8_t st = d.brick_state[b] & kBrickUnlisted;
        const bool active = blocked || st != 0 || !(rho <= d.skin_ok);   // NaN-safe
        d.brick_state[b] = st | (active ? kBrickActive : 0) | (nonempty ? 0 : kBrickEmpty);   // (Empty: k_velpos_tile need not look)
        if (active && nonempty && !blocked) atomicAdd(&d.counters[kCntActiveBricks], 1ull);
        if (active) my_pos = atomicAdd(&s_act[0], 1u);
    }
    // the block's active bricks go to the work list of the scanning kernel with one atomic
    __syncthreads();
    if (threadIdx.x == 0 && s_act[0]) s_act[1] = (uint32_t)atomicAdd(&d.counters[kCntActiveCount], (unsigned long long)s_act[0]);
    __syncthreads();
    if (my_pos != 0xFFFFFFFFu) d.active_list[s_act[1] + my_pos] = (uint32_t)b;
}

// every brick's record (both parities) to "nobody was integrated here, nothing has moved": once, when the buffers are allocated
// (k_brick_activity_box keeps the bricks it covers; the ones beyond a slab's ghost layers are only ever read)
__global__ void __launch_bounds__(kBlock) k_box_init(Dev d)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= 2u * d.n_bricks) return;
    uint4 *wp = reinterpret_cast<uint4 *>(d.boxrec + (size_t)i * kBoxRecWords);
    wp[0] = make_uint4(kBoxKeyEmptyLo, kBoxKeyEmptyLo, kBoxKeyEmptyLo, kBoxKeyEmptyHi);
    wp[1] = make_uint4(kBoxKeyEmptyHi, kBoxKeyEmptyHi, 0u, 0u);
    wp[2] = make_uint4(0u, 0u, 0u, 0u);
}
void launch_box_init(const Dev &d, hipStream_t s)
{
    if (d.boxrec && d.n_bricks) hipLaunchKernelGGL(k_box_init, dim3((2u * d.n_bricks + kBlock - 1) / kBlock), dim3(kBlock), 0, s, d);
}

// The pair sums of VelPos.compute:67-98 from the recorded list of slot s (n entries), then :100-157.
template <int PREC, bool FAST>
__device__ __forceinline__ Pos3 walk_list_and_integrate(const Dev &d, uint32_t s, uint32_t n, uint32_t mask, bool live,
                                                        float pix, float piy, float piz)
{
    const float4 rvi = d.cur.rv[s];
    const float viw = d.cur.vw[s];
    const float rho_i = rvi.x;                                                   // :60
    const float p_i = d.gas_const * (rho_i - d.rest_density);                    // :61
    const float vix = rvi.y, viy = rvi.z, viz = rvi.w;                           // :62
    Force f{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};                                 // :64-65
    struct Cand {
        float x, y, z, rho, vx, vy, vz;
        bool take;
    };
    const uint32_t slot_mask = (1u << kNlSlotBits) - 1u;
    const uint32_t *ent = nl_entries(d, s);
    auto word = [&](uint32_t e) { return ent[(e < n ? e : n - 1) * kWave]; };    // one coalesced 256 B row per wave
    auto gather = [&](uint32_t w) {
        const uint32_t j = w & slot_mask;
        const Pos3 pj = load_p4<PREC>(d, d.cur, j);
        const float4 rv = d.cur.rv[j];                                           // :86, :94
        return Cand{pj.x, pj.y, pj.z, rv.x, rv.y, rv.z, rv.w,
                    j != s && ((mask >> (w >> kNlSlotBits)) & 1u) != 0u};        // :82 and the stencil test :73
    };
#if defined(SPH_ABLATE) && SPH_ABLATE >= 5   /* diagnostic: no entry loop */
    if (false) {
#else
    if (n) {
#endif
        // two entries per trip; the entry words run two trips ahead of the arithmetic, their gathers one trip
        uint32_t w2 = word(2), w3 = word(3);
        Cand a0 = gather(word(0)), a1 = gather(word(1));
#ifdef SPH_LIST_UNROLL
#pragma unroll SPH_LIST_UNROLL
#endif
        for (uint32_t e = 0; e < n; e += 2) {
            const Cand b0 = gather(w2), b1 = gather(w3);
            w2 = word(e + 4);
            w3 = word(e + 5);
#if defined(SPH_ABLATE) && SPH_ABLATE == 7   /* diagnostic: the gathers without the pair arithmetic (0.40 of 2.0 ms) */
            f.px += a0.x + a0.rho; f.py += a0.y + a0.vx; f.pz += a1.z + a1.vy; f.vx += a1.vz + a1.x;
            a0 = b0; a1 = b1;
            continue;
#endif
            if (FAST) {                                                          // tolerance mode (compile time)
                pair_fast(d, pix - a0.x, piy - a0.y, piz - a0.z, a0.rho, a0.vx, a0.vy, a0.vz, vix, viy, viz, p_i, f, a0.take);
                pair_fast(d, pix - a1.x, piy - a1.y, piz - a1.z, a1.rho, a1.vx, a1.vy, a1.vz, vix, viy, viz, p_i, f,
                          e + 1 < n && a1.take);
                a0 = b0;
                a1 = b1;
                continue;
            }
            // (the list kernels never run in a wild launch: nl_blocked)
            const PairPre q0 = pair_pre<true>(d, pix - a0.x, piy - a0.y, piz - a0.z, a0.rho, a0.vx, a0.vy, a0.vz, vix, viy, viz,
                                              p_i, a0.take);                     // :86-95
            const PairPre q1 = pair_pre<true>(d, pix - a1.x, piy - a1.y, piz - a1.z, a1.rho, a1.vx, a1.vy, a1.vz, vix, viy, viz,
                                              p_i, e + 1 < n && a1.take);
            pair_force_two(d, q0, q1, f);
            a0 = b0;
            a1 = b1;
        }
    }
    if (live) return integrate_store<PREC, FAST>(d, s, pix, piy, piz, vix, viy, viz, viw, rho_i, f);  // :100-157
    return Pos3{pix, piy, piz};
}

// VelPos.compute:46-158 for the targets of CALM bricks, one lane per slot: pair sums from the list recorded by the
// frame's first sub-step.  Slot order keeps every per-target stream (9 loads, 6 stores, the 256 B entry rows) fully
// coalesced; the candidates of 64 consecutive slots (~17 bins of one row and their neighbours, ~20 KB of
// (x,y,z,-) + (rho,vx,vy,vz)) are gathered through L1/L2 -- there is no tile to stage.  The pair arithmetic
// (~125 VALU instructions per entry; 14 quotients per two entries from 4 refined reciprocals, sph_fastdiv.hpp) is what
// bounds the kernel: 1.6 of its 2.0 ms (SPH_ABLATE=7 leaves the gathers: 0.4 ms).  A wave runs as long as its longest list
// (mean 17.8 entries, mean of the per-wave maxima 24.7: scripts/list_balance.py); dealing the slots of a workgroup to its
// waves by list length cuts the trips by 19 % but ran 20-40 % SLOWER, stable or not (round 2): lanes that are no longer
// spatial neighbours gather 64 unrelated cache lines per instruction.  Consecutive slots per wave stay.
template <int PREC, bool FAST>
__global__ void __launch_bounds__(kListBlock, SPH_LIST_MINWAVES) k_velpos_list(Dev d)
{
    __shared__ uint32_t s_queue[2];
    if (nl_blocked(d)) return;                     // the scanning kernels take everything
    if (d.nl_tile && d.counters[kCntGlobalLists] == 0ull) return;   // every list of the frame holds tile indices: k_velpos_tile's
    uint32_t s_base = xcd_contiguous_block() * kListBlock, s_raw = s_base + threadIdx.x;
    bool live = s_raw < d.count;
    if (d.pass_mode == 1u) {
        // first pass of a two-pass sub-step: a SMALL grid over the two slot ranges of the boundary planes (their sizes are
        // host-known: zone_blocks; where they start is not: cell_start)
        const ZoneSlots z = zone_slots(d);
        const bool lower = blockIdx.x < d.zone_blocks[0];
        s_base = (lower ? z.r[0] + blockIdx.x * kListBlock : z.r[2] + (blockIdx.x - d.zone_blocks[0]) * kListBlock);
        s_raw = s_base + threadIdx.x;
        live = s_raw < (lower ? z.r[1] : z.r[3]) && s_raw < d.count;
        if (s_base >= (lower ? z.r[1] : z.r[3])) return;                         // block-uniform
    } else if (d.pass_mode == 2u) {                // second pass: everybody else
        const ZoneSlots z = zone_slots(d);
        const uint32_t s_end = s_base + kListBlock;
        if ((s_base >= z.r[0] && s_end <= z.r[1]) || (s_base >= z.r[2] && s_end <= z.r[3])) return;   // block-uniform
        live = live && !in_zone(z, s_raw);
    }
    const uint32_t s = live ? s_raw : 0u;
    ListTarget t = list_target<PREC>(d, s);
    uint32_t n = d.nl_cnt[s];
    if (n & kNlTileFormat) live = false;           // a list of tile indices: k_velpos_tile walks it
    n &= kNlCountMask;
    const uint32_t key = bucket_key(d, t.qx, t.qy, t.qz);                        // the bin, Bucket.compute:26-29
    live = live && key < d.num_cells;                                            // limbo: queued by the brick kernel
    if (d.ghost && d.ghost[s]) live = false;       // z-slab mode: ghosts are candidates only, their owner integrates them
    const uint32_t brick = live ? nl_brick_of_key(d, key) : 0u;
    live = live && !(d.brick_state[brick] & kBrickActive);                       // else the scanning kernel takes it
    bool hand_over = false;
    if (live && (t.shx | t.shy | t.shz) != 0) {
        // condition (2): the target changed cell.  One cell away, its recorded list minus the bins that left the stencil
        // is complete unless a bin that ENTERED the stencil holds something within h -- k_velpos_list_changers looks, and
        // in the (rare) positive case has the generic kernel redo the target.  Further away: generic kernel at once.
        hand_over = true;
        if (one_cell_away(t)) t.mask = shifted_mask(t);
        else live = false;
    }
    if (live && (t.c0z + t.shz < d.zt_lo || t.c0z + t.shz > d.zt_hi)) atomicAdd(&d.counters[kCntHaloViolation], 1ull);
    enqueue_block(d, hand_over, s, s_queue, d.queue2, kCntQueue2);
    if (!live) n = 0;
    const Pos3 pn = walk_list_and_integrate<PREC, FAST>(d, s, n, t.mask, live, t.px, t.py, t.pz);
    if (d.box_mode) box_update_wave(d, live, brick, pn.x - t.px, pn.y - t.py, pn.z - t.pz);   // (launch-uniform)
    stat_count(d, kStatListed, live);
    stat_add_u32(d, kStatListEntries, n);
}

// The targets of k_velpos_list that changed cell since the lists were recorded: a bin that ENTERED their stencil might
// hold a candidate within h.  HALF A WAVE per such target, one lane per bin of its current stencil (27 of 32 lanes; one lane
// per target walked its <= 19 new bins one dependent gather at a time: 0.135 ms for 1 % of the targets): distance pre-test
// on the listed particles of the bins that are not in the recorded stencil.  If any lane finds one, or if the target is more
// than one cell from its bin (k_velpos_list did not integrate it), the generic scanning kernel takes the target -- and
// overwrites what k_velpos_list stored for it.
__global__ void __launch_bounds__(kBlock) k_velpos_list_changers(Dev d)
{
    const uint32_t qlen = (uint32_t)d.counters[kCntQueue2];
    constexpr uint32_t kPerBlock = kBlock / 32;          // targets per workgroup per trip
    const uint32_t half = threadIdx.x / 32u, code = threadIdx.x & 31u;
    for (uint32_t q0 = blockIdx.x * kPerBlock; q0 < qlen; q0 += gridDim.x * kPerBlock) {   // block-uniform
        const uint32_t q = q0 + half;
        const bool live = q < qlen;
        const uint32_t s = d.queue2[live ? q : qlen - 1];
        const ListTarget t = list_target(d, s);
        const bool far = !one_cell_away(t);
        bool hit = false;
        if (live && !far && code < 27u) {
            // the bin of this lane; only those beyond c0 +- 1 on an axis that moved were not recorded
            const int cx = t.c0x + t.shx, cy = t.c0y + t.shy, cz = t.c0z + t.shz;
            const int bx = cx + (int)code / 9 - 1, by = cy + ((int)code / 3) % 3 - 1, bz = cz + (int)code % 3 - 1;
            const bool recorded = bx >= t.c0x - 1 && bx <= t.c0x + 1 && by >= t.c0y - 1 && by <= t.c0y + 1 &&
                                  bz >= t.c0z - 1 && bz <= t.c0z + 1;
            if (!recorded) {
                const CellRange cr = cell_range(d, bx, by, bz);                  // bounds test :73 + the bucket row
                constexpr uint32_t kTrip = 4;                                    // loads of a trip issued together
                for (uint32_t k0 = 0; k0 < cr.cnt; k0 += kTrip) {
                    Pos3 pj[kTrip];
#pragma unroll
                    for (uint32_t u = 0; u < kTrip; u++) pj[u] = load_p4(d, d.cur, cr.start + (k0 + u < cr.cnt ? k0 + u : cr.cnt - 1u));
#pragma unroll
                    for (uint32_t u = 0; u < kTrip; u++) {
                        const float rx = t.px - pj[u].x, ry = t.py - pj[u].y, rz = t.pz - pj[u].z;
                        if (__builtin_fmaf(rz, rz, __builtin_fmaf(ry, ry, rx * rx)) < d.h2_up) hit = true;   // superset of |r| < h (a repeated last member changes nothing)
                    }
                }
            }
        }
        // per half-wave: does the target go to the generic kernel?  lane 0 of the half speaks for it
        const unsigned long long hits = __ballot(hit);
        const unsigned long long mine = (hits >> (32u * (half & 1u))) & 0xFFFFFFFFull;
        const bool generic = live && code == 0u && (far || mine != 0ull);
        const unsigned long long m = __ballot(generic);                          // (wave-aggregated: this is rare)
        if (m) {
            const int lane = threadIdx.x & (kWave - 1), leader = __ffsll((long long)m) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(&d.counters[kCntQueue], (unsigned long long)__popcll(m));
            base = __shfl(base, leader, kWave);
            if (generic) d.queue[base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull))] = s;
        }
    }
}

// End of a frame: how many target sub-steps the lists served so far, into host-visible memory (no synchronisation: the
// engine's list policy reads it whenever it next issues a frame).  report[0] frames reported, [1] listed target
// sub-steps (cumulative since the last statistics reset), [2] frames of those that recorded lists, [3] targets of the
// last recording frame whose list exceeded nl_cap (the engine then re-allocates longer rows).
__global__ void __launch_bounds__(kStatStripes) k_frame_report(Dev d, unsigned long long *report, int lists_in_use)
{
    __shared__ unsigned long long s_part[kStatStripes / kWave];
    unsigned long long v = d.stripes[((size_t)kStatListed * kStatStripes + threadIdx.x) * kStatPad];
    v = wave_sum_u64(v);
    if ((threadIdx.x & (kWave - 1)) == 0) s_part[threadIdx.x / kWave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long total = 0;
        for (int w = 0; w < kStatStripes / kWave; w++) total += s_part[w];
        report[1] = total;
        report[2] += lists_in_use ? 1ull : 0ull;
        if (lists_in_use) report[3] = d.counters[kCntListTooLong];   // targets whose list did not fit nl_cap this frame
        {   // [4]: brick sub-steps the scanning kernel took, cumulative whatever sph_reset_stats does to the counter ([5]: its last value)
            const unsigned long long cur = d.counters[kCntActiveBricks], last = report[5];
            report[4] += cur >= last ? cur - last : cur;
            report[5] = cur;
        }
        __threadfence_system();
        report[0] += 1ull;                         // last: the host takes [1], [2] only as fresh as [0]
        __threadfence_system();
    }
}

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return a ? (a + b - 1) / b : 1u; }

void launch_frame_report(const Dev &d, unsigned long long *report, bool lists_in_use, hipStream_t s)
{
    hipLaunchKernelGGL(k_frame_report, dim3(1), dim3(kStatStripes), 0, s, d, report, lists_in_use ? 1 : 0);
}

void launch_list_prepare(const Dev &d, hipStream_t s)
{
    const uint32_t nbx = nl_bricks_x(d.R), nby = nl_bricks_y(d.R), nbz = nl_bricks_z(d.R);
    if (!d.nl_still && !d.box_mode) hipLaunchKernelGGL(k_chunk_box, dim3(cdiv(d.count, kBlock)), dim3(kBlock), 0, s, d);
    // the brick layers along z with planes that hold targets (z-slab mode: this rank's planes)
    const int zl = d.zo_lo < 0 ? 0 : d.zo_lo, zh = d.zo_hi > (int)d.R ? (int)d.R : d.zo_hi;
    const uint32_t bl = (uint32_t)zl / kNlBZ, bh = zh > zl ? ((uint32_t)zh + kNlBZ - 1) / kNlBZ : bl;
    const uint32_t count = nbx * nby * ((bh < nbz ? bh : nbz) - bl);
    if (count && d.box_mode) {
        // z-slab contexts: + the layers that hold ghost planes (halo_w = zo_lo - zt_lo + 1 planes per face that has a neighbour)
        const int gl = d.zt_lo == INT_MIN ? zl : (zl - (d.zo_lo - d.zt_lo + 1) > 0 ? zl - (d.zo_lo - d.zt_lo + 1) : 0);
        const int gh = d.zt_hi == INT_MAX ? zh : (zh + (d.zt_hi - d.zo_hi + 2) < (int)d.R ? zh + (d.zt_hi - d.zo_hi + 2) : (int)d.R);
        const uint32_t cl = (uint32_t)gl / kNlBZ, ch0 = ((uint32_t)gh + kNlBZ - 1) / kNlBZ, ch = ch0 < nbz ? ch0 : nbz;
        const uint32_t cover = nbx * nby * (ch - cl);
        hipLaunchKernelGGL(k_brick_activity_box, dim3(cdiv(cover, kBlock / 32)), dim3(kBlock), 0, s, d, (int)nbx, (int)nby,
                           (int)(nbx * nby * cl), (int)cover, (int)(nbx * nby * bl), (int)count);
    } else if (count)
        hipLaunchKernelGGL(k_brick_activity, dim3(cdiv(count, kBlock / kWave)), dim3(kBlock), 0, s, d, (int)nbx, (int)nby,
                           (int)(nbx * nby * bl), (int)count);
}

void launch_velpos_lists(const Dev &d, hipStream_t s, hipEvent_t list_begin, hipEvent_t list_end)
{
    if (list_begin) (void)hipEventRecord(list_begin, s);     // the events bracket k_velpos_list alone (bench.py roofline)
    const uint32_t blocks = d.pass_mode == 1u ? (d.zone_blocks[0] + d.zone_blocks[1] ? d.zone_blocks[0] + d.zone_blocks[1] : 1u)
                                              : cdiv(d.count, kListBlock);
    if (d.prec == 0u) {
        if (d.fast) hipLaunchKernelGGL((k_velpos_list<0, true>), dim3(blocks), dim3(kListBlock), 0, s, d);
        else hipLaunchKernelGGL((k_velpos_list<0, false>), dim3(blocks), dim3(kListBlock), 0, s, d);
    } else {
        if (d.fast) hipLaunchKernelGGL((k_velpos_list<1, true>), dim3(blocks), dim3(kListBlock), 0, s, d);
        else hipLaunchKernelGGL((k_velpos_list<1, false>), dim3(blocks), dim3(kListBlock), 0, s, d);
    }
    if (list_end) (void)hipEventRecord(list_end, s);
    if (d.nl_still) return;                        // first sub-step: nobody has changed cell, nothing was handed over
    const uint32_t g = cdiv(d.count, kBlock);
    hipLaunchKernelGGL(k_velpos_list_changers, dim3(g < 4096u ? g : 4096u), dim3(kBlock), 0, s, d);
}

}  // namespace sph
