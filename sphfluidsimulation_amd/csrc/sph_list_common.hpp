// What the two list-walking VelPos kernels share (k_velpos_list in sph_kernels_list.hip: lists of global slots, gathered
// through L1/L2; k_velpos_tile in sph_brick_impl.inc: lists of LDS tile indices): how a target stands against the stencil its
// list was recorded for, and the hand-over of the ones that need another kernel.
#pragma once

#include "sph_sweep_common.hpp"

namespace sph {

// Two entry formats share the rows of Dev::nl_ent (the high bit of Dev::nl_cnt says which one a target has); either way a
// target owns the dword [row][slot % 64] of every row of its 64-slot group, so targets of both kinds can sit side by side:
//   global  u32  (stencil cell 0..26) << kNlSlotBits | candidate slot, entry e in row e:
//                recorded by k_density_queue (sph_kernels_sweep.hip), walked by k_velpos_list;
//   tile    u16  (stencil cell 0..26) << kNlTileIdxBits | index in the LDS tile of the target's (sub-)brick with halo 1,
//                entries 2r and 2r + 1 in the low and high half of row r (one coalesced dword per two entries):
//                recorded by k_density_brick<RECORD>, walked by k_velpos_tile, which stages the same tile again -- the tile
//                is a function of cell_start alone, and the grid is frozen for the frame (SphFluidSimulation.cs:98-102).
// A list of global slots includes the target itself (the walk skips it, VelPos.compute:82); a list of tile indices does not.
// Which format the brick kernel records: Dev::nl_tile (tile indices unless SPH_FLAG_GLOBAL_LISTS) -- see DESIGN.md 4.
constexpr uint32_t kNlTileFormat = 0x8000u, kNlCountMask = 0x7FFFu;
constexpr uint32_t kNlTileIdxBits = 11;          // tiles hold <= 2048 listed particles
// An odd list's last row is completed with kNlTilePad: stencil cell 31, which no target's stencil mask has (27 cells), tile index 0
// (a staged particle: its fields are read and dropped) -- the walk takes both halves of every row without asking for the length.
constexpr uint32_t kNlTilePad = 31u << kNlTileIdxBits;

__device__ __forceinline__ uint16_t *nl_tile_entry(uint32_t *row0, uint32_t e)   // row0 = nl_entries(d, slot)
{
    return reinterpret_cast<uint16_t *>(row0 + (size_t)(e >> 1) * kWave) + (e & 1u);
}

// Hands the flagged targets of a block to the generic queue with ONE atomic on the queue counter (every thread of the
// block must call it).  ~1 % of the targets fail condition (2) in a sub-step, i.e. nearly every wave holds one: an
// atomic per lane, or even per wave, on that single address would serialise the launch (~3 ns each).
__device__ __forceinline__ void enqueue_block(const Dev &d, bool pred, uint32_t slot, uint32_t *s_cnt, uint32_t *queue,
                                              int counter)
{
    if (threadIdx.x == 0) s_cnt[0] = 0;
    __syncthreads();
    uint32_t k = 0;
    if (pred) k = atomicAdd(&s_cnt[0], 1u);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt[0]) s_cnt[1] = (uint32_t)atomicAdd(&d.counters[counter], (unsigned long long)s_cnt[0]);
    __syncthreads();
    if (pred) queue[s_cnt[1] + k] = slot;
}

// Where a target stands against the stencil its list was recorded for.
struct ListTarget {
    float qx, qy, qz;      // where it was binned
    float px, py, pz;      // where it is now, VelPos.compute:57
    int c0x, c0y, c0z;     // recorded stencil centre (its bin)
    int shx, shy, shz;     // current cell - c0
    uint32_t mask;         // recorded stencil cells (0..26) that are still in the stencil
};

template <int PREC = -1>
__device__ __forceinline__ ListTarget list_target(const Dev &d, uint32_t s)
{
    ListTarget t;
    const int R = (int)d.R;
    const Pos3 pb = load_binned<PREC>(d, s), pc = load_pos<PREC>(d, d.cur, s);
    t.qx = pb.x; t.qy = pb.y; t.qz = pb.z;
    t.px = pc.x; t.py = pc.y; t.pz = pc.z;
    t.c0x = cell_coord(t.qx, d.Rm1, R); t.c0y = cell_coord(t.qy, d.Rm1, R); t.c0z = cell_coord(t.qz, d.Rm1, R);
    t.shx = cell_coord(t.px, d.Rm1, R) - t.c0x;                                  // :58
    t.shy = cell_coord(t.py, d.Rm1, R) - t.c0y;
    t.shz = cell_coord(t.pz, d.Rm1, R) - t.c0z;
    t.mask = 0x7FFFFFFu;
    return t;
}

__device__ __forceinline__ bool one_cell_away(const ListTarget &t)
{
    return t.shx >= -1 && t.shx <= 1 && t.shy >= -1 && t.shy <= 1 && t.shz >= -1 && t.shz <= 1;
}

// offsets {-1,0,1} of the recorded stencil that lie within one cell of the shifted centre, per axis -> cell mask
__device__ __forceinline__ uint32_t shifted_mask(const ListTarget &t)
{
    const uint32_t ax = t.shx > 0 ? 6u : (t.shx < 0 ? 3u : 7u), ay = t.shy > 0 ? 6u : (t.shy < 0 ? 3u : 7u),
                   az = t.shz > 0 ? 6u : (t.shz < 0 ? 3u : 7u);
    // bit (9 i + 3 j + k) = bit i of ax & bit j of ay & bit k of az: a product of the three 3-bit sets spread to their strides
    // (no carries: the partial products do not overlap) -- a dozen instructions where the loop over the 27 cells took ~160
    const uint32_t sy = (ay & 1u) | ((ay & 2u) << 2) | ((ay & 4u) << 4), sx = (ax & 1u) | ((ax & 2u) << 8) | ((ax & 4u) << 16);
    return az * sy * sx;
}

}  // namespace sph
