// Internal declarations shared by the HIP translation units of libsph_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace sph {

constexpr uint32_t kMaxPerVoxel = 32;  // MaxParticlesPerVoxel, SphFluidSimulation.cs:9
constexpr int kWave = 64;              // CDNA4 wavefront

// A workgroup barrier that orders LDS only.  __syncthreads() fences every address space, which on gfx9 means
// s_waitcnt vmcnt(0): every global load, store and atomic in flight is waited for.  Use where the threads only hand each
// other LDS data and something global should stay in the air across the barrier.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// One set of cell-sorted SoA particle arrays (ping-pong partner of another set).
struct Soa {
    float *x, *y, *z;     // position, SoA: what the neighbour scans stream (pos.w is derived, see k_download); nullptr when
                          // SphParams.precision = 1, see h4
    uint2 *h4;            // precision 1 (config C4): position as an 8-byte record of three fp16 offsets (+ 3 x 5 sub-cell bits)
                          // relative to the slot's reference cell (Dev::cellkey) instead of 12 bytes; arithmetic stays fp32 (load_pos)
    float4 *rv;           // (rho, vx, vy, vz) packed: ONE 16-byte gather per in-range pair in VelPos
    float4 *p4;           // (x, y, z, -) packed copy for the neighbour-list kernel's gathers; nullptr when lists are off
    float *vw;            // vel.w: sub-step counter / 0.003 NaN marker (VelPos.compute:146,150)
};

// Everything a kernel needs, passed by value.
struct Dev {
    uint32_t count;        // live particles = res*res
    uint32_t sentinel_n;   // _NumParticles
    uint32_t R;            // _BucketResolution
    uint32_t num_cells;    // R^3 ; cell index num_cells is the "limbo" cell of dropped particles
    float Rm1;             // float(R-1)
    float Rf;              // float(R)
    float mass, h, h2, h9;
    float h2_up;           // h^2 * (1 + 1e-5): conservative pre-filter for |r| < h
    float c_poly6;         // 315/(64*pi)           Density.compute:23
    float w_scale;         // tolerance mode: mass * c_poly6 / h9 in one factor (Density.compute:26, :54)
    float c_over_h6;       // (45/pi)/h^6           VelPos.compute:37,43
    float viscosity, gas_const, rest_density, stiffness, damping;
    float gx, gy, gz;
    float dt;              // sub-step length, by value ... 
    const float *dt_ptr;   // ... or, when a captured frame graph is replayed with another delta_time, read from here
    uint32_t force_slow;   // 1: take the literal one-phase VelPos loop (set by sph_set_params)
    uint32_t wild_v_cur;   // which kCntWildV slot describes the velocities in `cur`
    uint32_t use_bricks;   // 1: LDS-tiled brick kernels + queue; 0: per-target kernels on global memory
    Soa cur, nxt;
    uint32_t prec;         // SphParams.precision: 0 = fp32 positions, 1 = fp16 cell-relative positions
    uint32_t fast;         // SphParams.exact_order == 0: tolerance mode (hardware reciprocal / square root, see pair_fast)
    uint32_t nl_tile;      // the brick kernels record lists of LDS tile indices and k_velpos_tile walks them (sph_list_common.hpp)
    uint32_t *cellkey;     // precision 1: reference cell of every slot, (cx+2) | (cy+2) << 10 | (cz+2) << 20; the cell the particle
                           //   was in when the slot was assigned (grid build / upload), fixed for a whole frame like `id`
    uint32_t *cellkey_nxt; // written by the reorder, swapped with cellkey after a grid build
    uint2 *p1h;            // precision 1: p1x/p1y/p1z as fp16 offsets (the binned positions of the neighbour lists)
    uint32_t *id;          // slot -> particle id (texel y*res + x); fixed for a whole frame
    uint32_t *id_nxt;      // written by the reorder, swapped with id after a grid build
    uint32_t *key;         // by slot (of cur, before the reorder)
    uint32_t *cell_count;  // [num_cells+1]
    uint32_t *cell_start;  // [num_cells+2] exclusive prefix of cell_count
    uint2 *order;          // [count] (particle id, source slot) in new slot order
    uint32_t *block_sums;  // scan scratch
    uint32_t *queue;       // [count] target slots handed from the brick kernels to the generic kernels
    // z-slab (multi-GPU) mode only; all nullptr / neutral otherwise
    uint8_t *ghost;        // by slot: 1 = copy of a neighbour rank's particle (candidate only, never a target)
    uint8_t *ghost_nxt;    // written by the reorder
    uint32_t *slot_of_src; // inverse of the reorder permutation: pre-build index -> slot
    uint32_t n_owned;      // pre-build indices [0, n_owned) are owned, [n_owned, count) are ghosts
    // z-slab mode, a sub-step in two passes (sph_slab_frame.hip: the halo refresh of the boundary targets travels while the
    // interior targets are computed): 0 = every target; 1 = only targets binned in the boundary planes [zone_p[0], zone_p[1])
    // and [zone_p[2], zone_p[3]) (what the neighbours hold as ghosts); 2 = all the others
    uint32_t pass_mode;
    int zone_p[4];
    uint32_t zone_blocks[2];   // 256-slot blocks that cover the owned particles of the lower / upper boundary planes (host-known:
                               // they are the particles this slab refreshes for its neighbours)
    int zt_lo, zt_hi;      // current-cell z range an owned target may reach without leaving the halo
    int zo_lo, zo_hi;      // grid planes whose particles are targets here (whole grid outside slab mode)
    unsigned long long *counters;  // device statistics, see Counter
    unsigned long long *stripes;   // [kStatCount][kStatStripes * kStatPad]: per-wave statistics, see stat_add
    // Frame-coherent neighbour lists (sph_kernels_list.hip): the grid is frozen for the S sub-steps of a frame, so the
    // survivors found by sub-step 1 (radius h + skin) are the only candidates that can come into range later while the
    // relative displacement around a brick stays below the skin.  All nullptr / 0 when the feature is off.
    uint32_t use_lists;    // 1: sub-step 1 records lists, later sub-steps of the frame use them where still valid
    uint32_t nl_mode;      // this launch: kNlOff, kNlBuild (the density pass that records the lists) or kNlUse (VelPos)
    uint32_t nl_still;     // kNlUse: nothing has moved since the lists were recorded (first sub-step of the frame)
    float *p1x, *p1y, *p1z;  // by slot: position at grid-build time (= the bin the particle sits in for the frame)
    uint32_t *nl_ent;      // [slot / 64][nl_cap][slot % 64]: (stencil cell 0..26) << kNlSlotBits | candidate slot, reference
                           //   order; the lists of 64 consecutive slots interleaved: a wave reads one 256 B row per entry
    uint16_t *nl_cnt;      // by slot: entries recorded
    uint32_t nl_cap;
    float *chunk_box;      // [6][chunks]: min x,y,z / max x,y,z of the displacement (cur - p1) over each kChunkSlots consecutive slots
    uint32_t n_chunks, n_bricks;
    uint32_t *queue2;      // [count]: targets that changed cell and may see a bin that entered their stencil
    uint32_t *active_list; // [bricks]: the bricks the scanning kernel must take this sub-step (kNlUse)
    uint8_t *brick_state;  // [bricks]: kBrickUnlisted (set while building), kBrickActive (set per sub-step)
    float h2_list;         // pre-filter radius^2 of the building sub-step: (h + skin)^2, rounded up
    float h2_stage;        // h2_list with a margin for fused multiply-adds: what phase 1 of the recording density pass stages
    float skin_ok;         // a brick is calm while the relative displacement bound stays <= this
    // Per-brick displacement boxes (Dev::box_mode; sph_kernels_list.hip): every kernel that integrates a target adds the target's
    // displacement of THIS sub-step to the range of its bin's brick (dbox, atomic min / max on order-preserving keys);
    // k_brick_activity sums the ranges over the sub-steps of the frame (cbox) -- a superset of the range of the displacements
    // since the grid build, without a pass over all positions (k_chunk_box) per sub-step.
    uint32_t box_mode;     // 1: per-brick boxes; 0: k_chunk_box + chunk boxes (SPH_FLAG_CHUNK_BOXES)
    uint32_t box_par;      // parity of the sub-step in flight within its frame: which half of boxrec it writes
    uint32_t *boxrec;      // [2][bricks][kBoxRecWords]: a 64-byte record per brick -- words 0..5 the keys of min x,y,z / max x,y,z of the
                           //   displacements of ONE sub-step (what the kernels add to), words 6..11 the ranges summed over the sub-steps
                           //   so far (floats, k_brick_activity's): a neighbour brick's twelve numbers in one cache line
    uint32_t tile_slots;   // which k_velpos_tile the launcher takes: 1 = whole-brick records, 2 = the records of halved bricks too
    uint32_t *brick_rec;   // [bricks][kBrickRecWords]: the set-up of every brick's LDS tile (cell table after the offset scan, owned
                           //   rows), written by the recording density pass and read back by k_velpos_tile in the S sub-steps of the
                           //   frame -- the tile is a function of cell_start alone, and the grid is frozen for the frame
};

enum : uint32_t { kNlOff = 0, kNlBuild = 1, kNlUse = 2 };
enum : uint8_t { kBrickUnlisted = 1, kBrickActive = 2, kBrickEmpty = 4 };   // (Empty: no slot in the rows of the brick's bins, set per sub-step)
#ifndef SPH_CHUNK_SLOTS
#define SPH_CHUNK_SLOTS 64
#endif
constexpr int kChunkSlots = SPH_CHUNK_SLOTS;      // consecutive slots per displacement box (a power of two <= 64)
constexpr uint32_t kNlSlotBits = 27;             // entry = cell code << 27 | slot
constexpr int kNlBX = 4, kNlBY = 4, kNlBZ = 4;   // brick shape of the VelPos kernels (sph_kernels_brick_velpos.hip)
// A brick record (Dev::brick_rec; 4x4x4-cell bricks with halo 1: 6^3 = 216 tile cells, 16 owned rows), in dwords:
//   [2c], [2c + 1]       tile cell c (z fastest): LDS offset | listed count << 16, first global slot   (one 8-byte load per thread)
//   [kBrickRecRows + r]  first slot of owned row r;   [kBrickRecPref + r]  targets in the rows before r (r = 16: all of them)
//   [kBrickRecStatus]    bit 0: the record is valid; bits 1..3: cells along x of its range (4 = the whole brick, 2 = a half);
//                        bits 8..17: listed particles of the tile; bits 18..: targets of the range (saturating)
// kBrickRecSlots records per brick: [0] the whole brick, or its left half when the whole does not fit one tile; [1] the right half.
//   [kBrickRecSrc + i]   global slot of the particle with tile index i (i < kBrickRecSrcCap)
//   [kBrickRecTgt + 2t], [.. + 1]   the t-th target of the range (t < kBrickRecTgtCap): its slot; its own tile index (0xFFFF: it is
//                        not one of the listed members of its cell) | owned row << 16 | cell along x within the range << 20
constexpr int kBrickRecCells = (kNlBX + 2) * (kNlBY + 2) * (kNlBZ + 2);
constexpr int kBrickRecRows = 2 * kBrickRecCells, kBrickRecPref = kBrickRecRows + kNlBY * kNlBZ,
              kBrickRecStatus = kBrickRecPref + kNlBY * kNlBZ + 1, kBrickRecSrc = 512, kBrickRecSrcCap = 1024,
              kBrickRecTgt = kBrickRecSrc + kBrickRecSrcCap, kBrickRecTgtCap = 256,
              kBrickRecWords = kBrickRecTgt + 2 * kBrickRecTgtCap, kBrickRecSlots = 2;
constexpr uint32_t kBrickRecTnMax = (1u << 14) - 1u;
static_assert(kBrickRecStatus < kBrickRecSrc, "a brick record fits its stride");

// order-preserving key of a float for integer atomic min / max (NaN sorts above +inf or below -inf: it reaches the decoder)
__device__ __forceinline__ uint32_t float_key(float f)
{
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }
constexpr uint32_t kBoxKeyEmptyLo = 0xFFFFFFFFu, kBoxKeyEmptyHi = 0u;   // a range nobody has added to
constexpr uint32_t kBoxRecWords = 16;

// k_velpos_tile asks for entry rows up to kNlRowsPad rows ahead without looking at a list's length: rows past its end are rows of the
// next 64 slots, or of the padding behind the last ones (Dev::nl_ent is allocated with kNlRowsPad more rows; never used: the walk ends first).
constexpr uint32_t kNlRowsPad = 4;
__device__ __forceinline__ uint32_t *nl_entries(const Dev &d, uint32_t slot)   // entry e of the slot: [e * kWave]
{
    return d.nl_ent + ((size_t)(slot / kWave) * d.nl_cap) * kWave + (slot % kWave);
}
__host__ __device__ __forceinline__ uint32_t nl_bricks_x(uint32_t R) { return (R + kNlBX - 1) / kNlBX; }
__host__ __device__ __forceinline__ uint32_t nl_bricks_y(uint32_t R) { return (R + kNlBY - 1) / kNlBY; }
__host__ __device__ __forceinline__ uint32_t nl_bricks_z(uint32_t R) { return (R + kNlBZ - 1) / kNlBZ; }

// Slot ranges of the boundary planes (slots are in bin order, z slowest): [r[0], r[1]) and [r[2], r[3]).
struct ZoneSlots {
    uint32_t r[4];
};
__device__ __forceinline__ ZoneSlots zone_slots(const Dev &d)
{
    ZoneSlots z;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int p = d.zone_p[k] < 0 ? 0 : (d.zone_p[k] > (int)d.R ? (int)d.R : d.zone_p[k]);
        z.r[k] = d.cell_start[(uint32_t)p * d.R * d.R];
    }
    if (z.r[2] < z.r[1]) z.r[2] = z.r[1];              // (a slab thinner than two halos: the ranges must not overlap)
    if (z.r[3] < z.r[2]) z.r[3] = z.r[2];
    return z;
}
__device__ __forceinline__ bool in_zone(const ZoneSlots &z, uint32_t s) { return (s >= z.r[0] && s < z.r[1]) || (s >= z.r[2] && s < z.r[3]); }
// does the pass in flight take a target of slot s / a brick of planes [z0, z0 + nz)?
__device__ __forceinline__ bool pass_takes_slot(const Dev &d, const ZoneSlots &z, uint32_t s)
{
    return d.pass_mode == 0u || (in_zone(z, s) == (d.pass_mode == 1u));
}
__device__ __forceinline__ bool pass_takes_brick(const Dev &d, int z0, int nz)
{
    if (d.pass_mode == 0u) return true;
    const bool touches = (z0 < d.zone_p[1] && z0 + nz > d.zone_p[0]) || (z0 < d.zone_p[3] && z0 + nz > d.zone_p[2]);
    return touches == (d.pass_mode == 1u);
}

// Statistics every wave adds to (candidates visited, list entries walked): one counter would take ~0.5 M atomics on a
// single address per launch, which alone costs milliseconds; they are spread over kStatStripes cache lines by block
// and summed by sph_get_stats.
enum Stat : int { kStatCandidates = 0, kStatListed, kStatListEntries, kStatCount };
constexpr int kStatStripes = 512, kStatPad = 16;   // 16 x 8 B = one 128 B line per stripe

enum Counter : int {
    kCntNaN = 0,        // NaN-guard hits
    kCntCandidates,     // (unused, see kStatCandidates)
    kCntMaxOcc,         // max cell occupancy (atomicMax)
    kCntOverflowCells,  // cells with > 32 members
    kCntDropped,        // particles whose linear cell index is outside the bucket
    kCntWildP,          // some pressure k*(rho-rho0) is >= 1e15, inf or NaN (set by the density pass)
    kCntWildV0,         // some velocity component is >= 1e15, inf or NaN (two slots, ping-pong)
    kCntWildV1,
    kCntQueue,          // length of `queue` for the pass in flight
    kCntHaloViolation,  // slab mode: owned targets whose 27 cells were not all covered by the halo
    kCntGlobalLists,    // tile-list mode: some target got a list of global slots this frame (else k_velpos_list has nothing to do)
    kCntActiveBricks,   // non-empty bricks that took the scanning kernel in a list-mode sub-step (since sph_reset_stats)
    kCntListEntries,    // (unused, see kStatListEntries)
    kCntActiveCount,    // length of active_list for the sub-step in flight
    kCntQueue2,         // length of queue2 (targets k_velpos_list hands to k_velpos_list_changers) for the sub-step in flight
    kCntIrregular,      // last grid build saw a position outside [0,1] (or NaN): bins say nothing about positions
    kCntListTooLong,    // targets of the last recording density pass whose list exceeded nl_cap (their bricks scan)
    kCntQueueHead,      // next unclaimed entry of `queue`: the queue kernels deal it out in chunks of one block
    kCntActiveHead,     // next unclaimed entry of active_list: the workgroups of k_velpos_brick claim their bricks one by one
    kCntCount
};

// ---- HLSL conversion semantics (identical to oracle/sph_oracle.c) -----------------------------
__device__ __forceinline__ int hlsl_int(float f)
{
    if (f != f) return 0;
    if (f <= -2147483648.0f) return INT32_MIN;
    if (f >= 2147483648.0f) return INT32_MAX;
    return (int)f;
}
__device__ __forceinline__ uint32_t hlsl_uint(float f)
{
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// Bucket row of a particle, Bucket.compute:26-29: float3 idx_3d = int3(pos * (R-1)); the linear index
// is evaluated in float (float3 component * uint) and converted to uint, then *= 32 in uint.
// Returns num_cells ("limbo") when the row lies outside the bucket buffer.
__device__ __forceinline__ uint32_t bucket_key(const Dev &d, float x, float y, float z)
{
    const float ix = (float)hlsl_int(x * d.Rm1), iy = (float)hlsl_int(y * d.Rm1), iz = (float)hlsl_int(z * d.Rm1);
    uint32_t index = hlsl_uint((ix + iy * d.Rf) + iz * d.Rf * d.Rf);
    index *= kMaxPerVoxel;
    const unsigned long long len = (unsigned long long)d.num_cells * kMaxPerVoxel;
    return (unsigned long long)index < len ? index / kMaxPerVoxel : d.num_cells;
}

// Cell coordinate of a position component: float3 idx_3d = int3(pos * (R-1))
// (Bucket.compute:27, Density.compute:39, VelPos.compute:58).  Returned clamped to [-2, R+1]:
// every coordinate outside that range behaves the same in the 27-cell sweep (all three
// neighbours fail the bounds test of Density.compute:46 / VelPos.compute:73).
__device__ __forceinline__ int cell_coord(float p, float Rm1, int R)
{
    int c = hlsl_int(p * Rm1);
    return c < -2 ? -2 : (c > R + 1 ? R + 1 : c);
}

// ---- positions: fp32 SoA (precision 0) or an 8-byte fp16 record per slot (precision 1) -----------------------------------
// Precision 1 stores a position relative to the slot's reference cell c (Dev::cellkey, fixed for the frame): with
// u = p * (R-1) - c (in [0,1) right after a grid build, a cell or two beyond while the particle drifts inside the frame),
//     sub = floor(8 u)  clamped to [-16, 15]          three 5-bit indices of an h/8 sub-cell, in the record's 4th half
//     o   = 8 u - sub - 0.5  as IEEE half, RN          three fp16 offsets from that sub-cell's centre, |o| <= 0.5
// i.e. the rounding of a stored position is <= 2^-13 of a sub-cell = 2^-16 of a cell (6e-8 of the box at R = 256, 3e-7 at
// R = 47): plain fp16 offsets from the cell centre (2^-13 of a CELL) left the velocities 4e-4 off the fp32 oracle after ONE
// sub-step -- the pressure term (h - r)^3 turns a relative position error of 2e-4 h into 2e-3 of the acceleration.
// Every kernel reads positions through load_pos(), i.e. sees the SAME fp32 value for a slot, so which kernel served a
// target still cannot change the result; only the stored state is rounded (integrate_store, the grid build, the upload).
struct Pos3 {
    float x, y, z;
};
__device__ __forceinline__ float half_bits_to_float(uint32_t h16) { return __half2float(__ushort_as_half((unsigned short)h16)); }
__device__ __forceinline__ uint32_t float_to_half_bits(float f) { return (uint32_t)__half_as_ushort(__float2half_rn(f)); }
__device__ __forceinline__ uint32_t pack_cellkey(int cx, int cy, int cz)   // coordinates in [-2, R+1] (cell_coord)
{
    return (uint32_t)(cx + 2) | (uint32_t)(cy + 2) << 10 | (uint32_t)(cz + 2) << 20;
}
constexpr float kSubCells = 8.0f;
__device__ __forceinline__ Pos3 decode_pos(const Dev &d, uint2 o, uint32_t ck)
{
    const float cx = (float)((int)(ck & 1023u) - 2), cy = (float)((int)((ck >> 10) & 1023u) - 2), cz = (float)((int)(ck >> 20) - 2);
    const uint32_t sub = o.y >> 16;
    const float sx = (float)((int)(sub & 31u) - 16) + 0.5f, sy = (float)((int)((sub >> 5) & 31u) - 16) + 0.5f,
                sz = (float)((int)((sub >> 10) & 31u) - 16) + 0.5f;
    return Pos3{(cx + (sx + half_bits_to_float(o.x & 0xFFFFu)) * (1.0f / kSubCells)) * d.h,
                (cy + (sy + half_bits_to_float(o.x >> 16)) * (1.0f / kSubCells)) * d.h,
                (cz + (sz + half_bits_to_float(o.y & 0xFFFFu)) * (1.0f / kSubCells)) * d.h};
}
__device__ __forceinline__ uint2 encode_pos(const Dev &d, Pos3 p, uint32_t ck)
{
    const float cx = (float)((int)(ck & 1023u) - 2), cy = (float)((int)((ck >> 10) & 1023u) - 2), cz = (float)((int)(ck >> 20) - 2);
    const float tx = (p.x * d.Rm1 - cx) * kSubCells, ty = (p.y * d.Rm1 - cy) * kSubCells, tz = (p.z * d.Rm1 - cz) * kSubCells;
    const float sx = fminf(fmaxf(floorf(tx), -16.0f), 15.0f), sy = fminf(fmaxf(floorf(ty), -16.0f), 15.0f),
                sz = fminf(fmaxf(floorf(tz), -16.0f), 15.0f);   // (NaN -> -16: the offset carries the NaN)
    const uint32_t sub = (uint32_t)((int)sx + 16) | (uint32_t)((int)sy + 16) << 5 | (uint32_t)((int)sz + 16) << 10;
    return make_uint2(float_to_half_bits(tx - sx - 0.5f) | float_to_half_bits(ty - sy - 0.5f) << 16,
                      float_to_half_bits(tz - sz - 0.5f) | sub << 16);
}
// PREC: 0 / 1 when the kernel is compiled per precision (k_velpos_list: a run-time test in its preamble and store cost 12 %),
// -1 to test Dev::prec at run time (everything else: positions are touched once per particle there).
template <int PREC = -1>
__device__ __forceinline__ bool is_fp32(const Dev &d) { return PREC == 0 || (PREC < 0 && d.prec == 0u); }

template <int PREC = -1>
__device__ __forceinline__ Pos3 load_pos(const Dev &d, const Soa &a, uint32_t j)
{
    if (is_fp32<PREC>(d)) return Pos3{a.x[j], a.y[j], a.z[j]};
    return decode_pos(d, a.h4[j], d.cellkey[j]);
}
template <int PREC = -1>
__device__ __forceinline__ void store_pos(const Dev &d, const Soa &a, uint32_t s, Pos3 p)
{
    if (is_fp32<PREC>(d)) {
        a.x[s] = p.x; a.y[s] = p.y; a.z[s] = p.z;
    } else {
        a.h4[s] = encode_pos(d, p, d.cellkey[s]);
    }
}
// the packed copy the neighbour-list kernels gather (Soa::p4): (x, y, z, -) fp32, or {record, reference cell} in precision 1
template <int PREC = -1>
__device__ __forceinline__ Pos3 load_p4(const Dev &d, const Soa &a, uint32_t j)
{
    if (is_fp32<PREC>(d)) {
        const float4 q = a.p4[j];
        return Pos3{q.x, q.y, q.z};
    }
    const uint4 q = reinterpret_cast<const uint4 *>(a.p4)[j];
    return decode_pos(d, make_uint2(q.x, q.y), q.z);
}
template <int PREC = -1>
__device__ __forceinline__ void store_p4(const Dev &d, const Soa &a, uint32_t s, Pos3 p)
{
    if (is_fp32<PREC>(d)) {
        a.p4[s] = make_float4(p.x, p.y, p.z, 0.0f);
    } else {
        const uint32_t ck = d.cellkey[s];
        const uint2 o = encode_pos(d, p, ck);
        reinterpret_cast<uint4 *>(a.p4)[s] = make_uint4(o.x, o.y, ck, 0u);
    }
}
// where the particle of slot s was when it was binned (neighbour lists)
template <int PREC = -1>
__device__ __forceinline__ Pos3 load_binned(const Dev &d, uint32_t s)
{
    if (is_fp32<PREC>(d)) return Pos3{d.p1x[s], d.p1y[s], d.p1z[s]};
    return decode_pos(d, d.p1h[s], d.cellkey[s]);
}

// launchers (sph_kernels_*.hip)
void launch_upload_split(const Dev &d, const float4 *pos4, const float4 *vel4, float *pw0, hipStream_t s);
void launch_download_merge(const Dev &d, const float *pw0, int substeps_done, float dt_last,
                           float4 *pos4, float4 *vel4, float *rho_by_id, hipStream_t s);
void launch_set_scalar(float *p, float v, hipStream_t s);
void launch_clear(const Dev &d, uint32_t counter_mask, uint32_t stat_mask, hipStream_t s);   // counters / striped statistics
void launch_grid_build(const Dev &d, hipStream_t s);  // keys, histogram, scan, scatter, id-sort, reorder: cur -> nxt
void launch_density(const Dev &d, hipStream_t s);     // sph_kernels_brick.hip (tiled) + queue
// cur -> nxt; the optional events bracket k_velpos_list (recorded only in a sub-step that uses neighbour lists)
// side / ev_fork / ev_join (optional): the scanning kernel over the few bricks that are not calm runs on `side`, beside the list
// kernels on `s` (forked after k_brick_activity, joined before the queue kernel)
void launch_velpos(const Dev &d, hipStream_t s, hipEvent_t list_begin = nullptr, hipEvent_t list_end = nullptr,
                   hipStream_t side = nullptr, hipEvent_t ev_fork = nullptr, hipEvent_t ev_join = nullptr);
void launch_box_init(const Dev &d, hipStream_t s);      // sph_kernels_list.hip: per-brick displacement ranges, once after allocation
void launch_list_prepare(const Dev &d, hipStream_t s);  // sph_kernels_list.hip: which bricks may use their lists
void launch_frame_report(const Dev &d, unsigned long long *report, bool lists_in_use, hipStream_t s);   // sph_kernels_list.hip
void launch_velpos_tile_lists(const Dev &d, hipStream_t s, hipEvent_t list_begin = nullptr, hipEvent_t list_end = nullptr);  // sph_kernels_brick_density.hip
void launch_velpos_lists(const Dev &d, hipStream_t s, hipEvent_t list_begin = nullptr, hipEvent_t list_end = nullptr);  // sph_kernels_list.hip: the targets of those bricks, cur -> nxt


// row f1: InitParticles.compute presets on the device (sph_kernels_init.hip)
void launch_init_particles(int preset, uint32_t res, float fill, uint32_t ppd, float4 *pos4, hipStream_t s);

// row f3: UpdateMeshProperties.compute (sph_kernels_mesh.hip); props = count x 5 float4, record index = particle id
void launch_mesh_properties(const Dev &d, float low, float high, const float *scale3, const float *sim_trs16,
                            float4 *props, hipStream_t s);

// sph_kernels_selftest.hip: out[4] = mismatches of the shared quotient, of the doubled-denominator quotient, of the unscaled sqrt, of the window test
void launch_selftest_arithmetic(uint64_t n, uint64_t seed, unsigned long long *out, hipStream_t s);

// z-slab mode (sph_kernels_slab.hip; protocol in sph_slab_frame.hip)
constexpr int kSlabRecordWords = 10;             // x y z rho vx vy vz vel.w id tag
constexpr int kSlabPosVelWords = 7;              // x y z rho vx vy vz
constexpr uint32_t kSlabMagic = 0x31485053u;     // "SPH1": sanity word of a message header
enum SlabCounter : int {
    kSlabKeep = 0,                 // particles that stay (compacted into nxt)
    kSlabMigL, kSlabMigR,          // migrants packed for the left / right neighbour
    kSlabKeptL, kSlabKeptR,        // ... of which stay here as ghosts
    kSlabGhostL, kSlabGhostR,      // keepers exported as ghosts to the left / right neighbour
    kSlabLost,                     // particles that jumped past a whole neighbour slab
    kSlabMisplaced,                // received migrants this slab cannot place (outside it, or in the far face's boundary planes)
    kSlabCounterCount = 16
};
struct SlabArgs {
    int z_lo, z_hi, halo_w;
    int has_left, has_right;
    int nb_lo, nb_hi;              // z_lo of the left neighbour's slab, z_hi of the right neighbour's
    float *send_mig[2], *send_ghost[2], *kept[2];   // [0] left neighbour, [1] right neighbour
    uint32_t *send_src[2];         // pre-build index of every particle this slab refreshes for that neighbour
    uint32_t *counters;            // SlabCounter
    uint32_t cap_records;          // capacity of each of the buffers above
};
void launch_slab_upload(const Dev &d, uint32_t n, const float4 *pos4, const float4 *vel4, const uint32_t *ids, hipStream_t s);
void launch_slab_classify(const Dev &d, const SlabArgs &a, uint32_t *hdr_send, unsigned long long *viol, hipStream_t s);
void launch_slab_append(const Dev &d, const float *recv, uint32_t n, uint32_t at, uint32_t *send_src, uint32_t own,
                        uint32_t cap_src, const SlabArgs &a, int from_side, hipStream_t s);
void launch_slab_pack_update(const Dev &d, const uint32_t *send_src, uint32_t n, int what, float *out, hipStream_t s);
void launch_slab_unpack_update(const Dev &d, uint32_t src0, uint32_t n, int what, const float *in, hipStream_t s);
void launch_slab_download(const Dev &d, int substeps_done, float dt_last, uint32_t *counter, uint32_t *ids,
                          float4 *pos4, float4 *vel4, float *rho, hipStream_t s);

}  // namespace sph
