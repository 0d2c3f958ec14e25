// z-slab domain decomposition (one slab per GPU): device side of the per-frame particle migration and of the
// ghost-plane exchange.  New work -- the reference is single-GPU.  A slab owns the grid planes
// cz in [z_lo, z_hi) by BUCKET ROW (Bucket.compute:26-29) and keeps copies ("ghosts") of the W planes on either side;
// ghosts are candidates only.  Membership is frozen for the frame like the reference's bucket
// (SphFluidSimulation.cs:98-102), so the ghost sets are fixed per frame and only their values are refreshed.
// The frame protocol (who sends what, when the host looks) is in sph_slab_frame.hip.
//
// Record layouts (fp32 words), in device buffers the transport moves (RCCL over xGMI, peer copies, or host staging):
//   particle record (migrants, ghost export): x y z rho vx vy vz vel.w id tag   -> 10 words
//        tag (migrants): index in the sender's list of migrants it KEEPS as ghosts, or 0xFFFFFFFF
//   ghost update, density pass               : rho                              -> 1 word
//   ghost update, sub-step                   : x y z rho vx vy vz               -> 7 words
//     (rho rides along: ghost slots are not integrated locally, so the ping-pong partner of a
//      ghost slot holds nothing until the update lands)
#include "sph_sweep_common.hpp"

namespace sph {

constexpr int kBlock = 256;
constexpr uint32_t kNoTag = 0xFFFFFFFFu;
constexpr int kClassifyBlock = 1024;   // one atomic on the keeper counter per workgroup: ~3 ns each on that single address

// The three position words of a record: x y z (fp32), or -- SphParams.precision = 1 -- the slot's fp16 record and its reference
// cell, bit for bit: a ghost must hold EXACTLY what its owner holds, and decode / encode is not the identity.
struct PosWords {
    float w0, w1, w2;
};
__device__ __forceinline__ PosWords get_pos_words(const Dev &d, const Soa &a, const uint32_t *cellkey, uint32_t p)
{
    if (d.prec == 0u) return PosWords{a.x[p], a.y[p], a.z[p]};
    const uint2 o = a.h4[p];
    return PosWords{__uint_as_float(o.x), __uint_as_float(o.y), __uint_as_float(cellkey[p])};
}
// cellkey == nullptr: the slot keeps its reference cell (a refresh inside the frame)
__device__ __forceinline__ void put_pos_words(const Dev &d, const Soa &a, uint32_t *cellkey, uint32_t p, PosWords w)
{
    if (d.prec == 0u) {
        a.x[p] = w.w0; a.y[p] = w.w1; a.z[p] = w.w2;
    } else {
        a.h4[p] = make_uint2(__float_as_uint(w.w0), __float_as_uint(w.w1));
        if (cellkey) cellkey[p] = __float_as_uint(w.w2);
    }
}
__device__ __forceinline__ Pos3 words_to_pos(const Dev &d, PosWords w)
{
    if (d.prec == 0u) return Pos3{w.w0, w.w1, w.w2};
    return decode_pos(d, make_uint2(__float_as_uint(w.w0), __float_as_uint(w.w1)), __float_as_uint(w.w2));
}

__device__ __forceinline__ void write_record(float *buf, uint32_t k, const Dev &d, uint32_t p, uint32_t tag)
{
    float *r = buf + (size_t)kSlabRecordWords * k;
    const float4 rv = d.cur.rv[p];
    const PosWords w = get_pos_words(d, d.cur, d.cellkey, p);
    r[0] = w.w0; r[1] = w.w1; r[2] = w.w2;
    r[3] = rv.x; r[4] = rv.y; r[5] = rv.z; r[6] = rv.w;
    r[7] = d.cur.vw[p];
    r[8] = __uint_as_float(d.id[p]);
    r[9] = __uint_as_float(tag);
}

// Stream compaction: every thread of the block calls it; the threads with `pred` get consecutive indices of the list
// behind `counter`.  ONE global atomic per block: an atomic per particle on a single counter costs ~3 ns each whatever
// else the kernel does (3 ms per frame at 16.7 M).
__device__ __forceinline__ uint32_t block_append(bool pred, uint32_t *counter, uint32_t *s_cnt /* [2] */)
{
    const int lane = threadIdx.x & (kWave - 1);
    if (threadIdx.x == 0) s_cnt[0] = 0;
    __syncthreads();
    const unsigned long long m = __ballot(pred);
    uint32_t wave_off = 0;
    if (m != 0ull) {                                   // wave-uniform
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) wave_off = atomicAdd(&s_cnt[0], (uint32_t)__popcll(m));
        wave_off = __shfl(wave_off, leader, kWave);
    }
    const uint32_t in_block = wave_off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt[0]) s_cnt[1] = atomicAdd(counter, s_cnt[0]);
    __syncthreads();
    return pred ? s_cnt[1] + in_block : 0u;
}

// Frame start, ONE pass over last frame's arrays (ghosts are dropped):
//   * particles whose bucket row is still in this slab are compacted into nxt ("keepers", index k); those binned in the
//     W boundary planes of a face are also packed for that neighbour (its ghosts) and remembered in send_src;
//   * particles whose row crossed into a neighbour's slab are packed as migrants; the ones that landed in the neighbour's
//     W boundary planes stay here as ghosts (`kept`), so that the neighbour need not send them back: the tag tells it
//     which entry of the later refresh messages they are.
// counters: see SlabCounter.
__global__ void __launch_bounds__(kClassifyBlock) k_slab_classify(Dev d, SlabArgs a)
{
    __shared__ uint32_t s_cnt[2];
    const uint32_t s = blockIdx.x * kClassifyBlock + threadIdx.x;
    int dest = -1;                                     // 0 keep, 1 to the left neighbour, 2 to the right one
    int cz = 0;
    bool in_grid = false;
    // every load of the particle is issued before the first barrier of the compaction (one latency, not two)
    PosWords pw{0.0f, 0.0f, 0.0f};
    float pvw = 0.0f;
    float4 prv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t pid = 0;
    if (s < d.count && !d.ghost[s]) {
        pw = get_pos_words(d, d.cur, d.cellkey, s);
        prv = d.cur.rv[s];
        pvw = d.cur.vw[s];
        pid = d.id[s];
        const Pos3 q = words_to_pos(d, pw);
        const uint32_t key = bucket_key(d, q.x, q.y, q.z);
        dest = 0;
        in_grid = key < d.num_cells;
        if (in_grid) {
            cz = (int)(key / (d.R * d.R));
            if (cz < a.z_lo && a.has_left) dest = 1;   // (a particle cannot lie beyond a missing neighbour)
            else if (cz >= a.z_hi && a.has_right) dest = 2;
        }
    }
    // jumped past a whole neighbour slab: nobody would hold it consistently (the host reports it)
    const bool lost = (dest == 1 && cz < a.nb_lo) || (dest == 2 && cz >= a.nb_hi);
    if (lost) atomicAdd(&a.counters[kSlabLost], 1u);
    const uint32_t k = block_append(dest == 0, &a.counters[kSlabKeep], s_cnt);
    if (dest == 0) {
        put_pos_words(d, d.nxt, d.cellkey_nxt, k, pw);
        d.nxt.rv[k] = prv;
        d.nxt.vw[k] = pvw;
        d.id_nxt[k] = pid;
    }
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const bool has = side == 0 ? a.has_left != 0 : a.has_right != 0;
        if (!has) continue;                            // block-uniform
        // keepers in this face's boundary planes -> the neighbour's ghosts
        const bool exported = dest == 0 && in_grid && (side == 0 ? cz < a.z_lo + a.halo_w : cz >= a.z_hi - a.halo_w);
        const uint32_t g = block_append(exported, &a.counters[kSlabGhostL + side], s_cnt);
        if (exported && g < a.cap_records) {
            write_record(a.send_ghost[side], g, d, s, kNoTag);
            a.send_src[side][g] = k;
        }
        // migrants to this neighbour; the ones within its W boundary planes stay here as ghosts
        const bool mig = dest == side + 1;
        const bool keep_copy = mig && (side == 0 ? cz >= a.z_lo - a.halo_w : cz < a.z_hi + a.halo_w);
        const uint32_t kc = block_append(keep_copy, &a.counters[kSlabKeptL + side], s_cnt);
        const uint32_t m = block_append(mig, &a.counters[kSlabMigL + side], s_cnt);
        if (mig && m < a.cap_records) write_record(a.send_mig[side], m, d, s, keep_copy ? kc : kNoTag);
        if (keep_copy && kc < a.cap_records) write_record(a.kept[side], kc, d, s, kNoTag);
    }
}

// Message headers for the two neighbours + this slab's cumulative halo-violation count, after k_slab_classify.
__global__ void k_slab_headers(Dev d, SlabArgs a, uint32_t *hdr_send, unsigned long long *viol)
{
    if (threadIdx.x < 2) {
        uint32_t *h = hdr_send + 4 * threadIdx.x;
        h[0] = a.counters[kSlabMigL + threadIdx.x];
        h[1] = a.counters[kSlabGhostL + threadIdx.x];
        h[2] = a.counters[kSlabKeptL + threadIdx.x];   // migrants this slab keeps as ghosts: the receiver refreshes them too
        h[3] = kSlabMagic ^ (uint32_t)a.halo_w;
    }
    if (threadIdx.x == 2) *viol = d.counters[kCntHaloViolation];
}

// Append received particle records behind the first `at` entries of cur.  Migrants (`send_src` given): a record with a
// tag is kept as a ghost by its sender, so it joins this slab's export list for that side at position own + tag.
__global__ void __launch_bounds__(kBlock) k_slab_append(Dev d, const float *__restrict__ recv, uint32_t n, uint32_t at,
                                                        uint32_t *send_src, uint32_t own, uint32_t cap_src,
                                                        SlabArgs a, int from_side)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    const float *r = recv + (size_t)kSlabRecordWords * k;
    const uint32_t p = at + k;
    const PosWords w{r[0], r[1], r[2]};
    put_pos_words(d, d.cur, d.cellkey, p, w);
    d.cur.rv[p] = make_float4(r[3], r[4], r[5], r[6]);
    d.cur.vw[p] = r[7];
    d.id[p] = __float_as_uint(r[8]);
    // values from another rank: the "wild" flags the local producers would have raised (sph_sweep_common.hpp)
    if (is_wild(r[4]) || is_wild(r[5]) || is_wild(r[6])) d.counters[kCntWildV0 + d.wild_v_cur] = 1ull;
    if (send_src) {
        const uint32_t tag = __float_as_uint(r[9]);
        if (tag != kNoTag) {
            if (own + tag < cap_src) send_src[own + tag] = p;
            else atomicAdd(&a.counters[kSlabMisplaced], 1u);
        }
        // a migrant must land in this slab, and not in the boundary planes of the FAR face (it would have to be
        // forwarded as a ghost: slabs are at least 2 W planes thick, so only a particle that outran the halo gets there)
        const Pos3 q = words_to_pos(d, w);
        const uint32_t key = bucket_key(d, q.x, q.y, q.z);
        bool bad = key >= d.num_cells;
        if (!bad) {
            const int cz = (int)(key / (d.R * d.R));
            bad = cz < a.z_lo || cz >= a.z_hi ||
                  (from_side == 0 ? (a.has_right && cz >= a.z_hi - a.halo_w) : (a.has_left && cz < a.z_lo + a.halo_w));
        }
        if (bad) atomicAdd(&a.counters[kSlabMisplaced], 1u);
    }
}

// Per-pass refresh of the exported ghosts: what = 0 -> rho (after the density pass),
// what = 1 -> position + velocity (after a sub-step).
__global__ void __launch_bounds__(kBlock) k_slab_pack_update(Dev d, const uint32_t *__restrict__ send_src,
                                                             uint32_t n, int what, float *__restrict__ out)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    const uint32_t s = d.slot_of_src[send_src[k]];
    const float4 rv = d.cur.rv[s];
    if (what == 0) {
        out[k] = rv.x;
    } else {
        float *r = out + 7ull * k;
        const PosWords w = get_pos_words(d, d.cur, d.cellkey, s);
        r[0] = w.w0; r[1] = w.w1; r[2] = w.w2;
        r[3] = rv.x; r[4] = rv.y; r[5] = rv.z; r[6] = rv.w;
    }
}

__global__ void __launch_bounds__(kBlock) k_slab_unpack_update(Dev d, uint32_t src0, uint32_t n, int what,
                                                               const float *__restrict__ in)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    const bool have = k < n;
    const uint32_t s = have ? d.slot_of_src[src0 + k] : 0u;
    if (!have) {
        // (no ghost for this lane; it still takes part in the wave's range reduction below)
    } else if (what == 0) {
        d.cur.rv[s].x = in[k];
        if (is_wild_density(d, in[k])) d.counters[kCntWildP] = 1ull;    // (as the density pass does for local particles)
    } else {
        const float *r = in + 7ull * k;
        put_pos_words(d, d.cur, nullptr, s, PosWords{r[0], r[1], r[2]});      // (the ghost slot keeps its reference cell: same as its owner's)
        if (d.cur.p4) {                                                       // the packed copy the list kernel gathers
            if (d.prec == 0u) d.cur.p4[s] = make_float4(r[0], r[1], r[2], 0.0f);
            else reinterpret_cast<uint4 *>(d.cur.p4)[s] = make_uint4(__float_as_uint(r[0]), __float_as_uint(r[1]), d.cellkey[s], 0u);
        }
        d.cur.rv[s] = make_float4(r[3], r[4], r[5], r[6]);
        if (is_wild(r[4]) || is_wild(r[5]) || is_wild(r[6])) d.counters[kCntWildV0 + d.wild_v_cur] = 1ull;
    }
    if (what != 0 && d.box_mode && d.nl_mode == kNlUse) {   // (launch-uniform)
        // per-brick displacement ranges (sph_kernels_list.hip): a ghost is nobody's target here -- its displacement of this
        // sub-step (d.cur = what the sub-step writes, d.nxt = what it read) goes to the range of its bin's brick with its refresh.
        // (Rounds per brick within the wave, 22 us per call at 8 slabs of 16.7 M; one set of atomics per ghost: 68 us -- the
        // ghosts of a brick meet on six addresses.  The owner's own per-brick ranges, sent along, would be cheaper still.)
        bool on = false;
        uint32_t brick = 0;
        float dx = 0.0f, dy = 0.0f, dz = 0.0f;
        if (have) {
            const Pos3 pb = load_binned(d, s), pn = load_pos(d, d.cur, s), po = load_pos(d, d.nxt, s);
            const uint32_t key = bucket_key(d, pb.x, pb.y, pb.z);
            on = key < d.num_cells;
            brick = on ? nl_brick_of_key(d, key) : 0u;
            dx = pn.x - po.x; dy = pn.y - po.y; dz = pn.z - po.z;
        }
        box_update_wave(d, on, brick, dx, dy, dz);
    }
}

// Back to the owned particles a frame started from (re-run with a wider halo): no ghosts, unsorted, wild flag re-raised.
__global__ void __launch_bounds__(kBlock) k_slab_restore(Dev d, uint32_t n, Soa snap, const uint32_t *__restrict__ snap_ck,
                                                         const float4 *__restrict__ rv, const float *__restrict__ vw,
                                                         const uint32_t *__restrict__ id)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const float4 v = rv[p];
    put_pos_words(d, d.cur, d.cellkey, p, get_pos_words(d, snap, snap_ck, p));
    d.cur.rv[p] = v;
    d.cur.vw[p] = vw[p];
    d.id[p] = id[p];
    if (is_wild(v.y) || is_wild(v.z) || is_wild(v.w)) d.counters[kCntWildV0 + d.wild_v_cur] = 1ull;
}

// Owned particles, compacted, in the reference's texture layout (+ ids).
__global__ void __launch_bounds__(kBlock) k_slab_download(Dev d, int substeps_done, float dt_last, uint32_t *counter,
                                                          uint32_t *ids, float4 *pos4, float4 *vel4, float *rho)
{
    __shared__ uint32_t s_cnt[2];
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    const bool mine = s < d.count && !d.ghost[s];
    const uint32_t k = block_append(mine, counter, s_cnt);
    if (!mine) return;
    const float4 rv = d.cur.rv[s];
    const float vw = d.cur.vw[s];
    const float pw = substeps_done > 0 ? clamp01(1.0f + dt_last * vw) : 1.0f;   // VelPos.compute:153-154
    ids[k] = d.id[s];
    Pos3 q = load_pos(d, d.cur, s);
    if (d.prec && substeps_done > 0) q = Pos3{clamp01(q.x), clamp01(q.y), clamp01(q.z)};   // (VelPos.compute:154 clamped before the rounding)
    pos4[k] = make_float4(q.x, q.y, q.z, pw);
    vel4[k] = make_float4(rv.y, rv.z, rv.w, vw);
    rho[k] = rv.x;
}

__global__ void __launch_bounds__(kBlock) k_slab_upload(Dev d, uint32_t n, const float4 *__restrict__ pos4,
                                                        const float4 *__restrict__ vel4, const uint32_t *__restrict__ ids)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const float4 q = pos4[p];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vel4) v = vel4[p];
    if (d.prec) {                                      // reference cell of an uploaded particle: the cell it is in
        const int R = (int)d.R;
        d.cellkey[p] = pack_cellkey(cell_coord(q.x, d.Rm1, R), cell_coord(q.y, d.Rm1, R), cell_coord(q.z, d.Rm1, R));
    }
    store_pos(d, d.cur, p, Pos3{q.x, q.y, q.z});
    d.cur.rv[p] = make_float4(0.0f, v.x, v.y, v.z);
    d.cur.vw[p] = v.w;
    d.id[p] = ids[p];
    d.ghost[p] = 0;
    if (!(fabsf(v.x) < 1e15f) || !(fabsf(v.y) < 1e15f) || !(fabsf(v.z) < 1e15f))
        d.counters[kCntWildV0 + d.wild_v_cur] = 1ull;
}

// ---- launchers ------------------------------------------------------------------------------------

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
static inline dim3 grid_for(uint32_t n) { return dim3(n ? cdiv(n, kBlock) : 1u); }

void launch_slab_classify(const Dev &d, const SlabArgs &a, uint32_t *hdr_send, unsigned long long *viol, hipStream_t s)
{
    hipLaunchKernelGGL(k_slab_classify, dim3(d.count ? cdiv(d.count, kClassifyBlock) : 1u), dim3(kClassifyBlock), 0, s, d, a);
    hipLaunchKernelGGL(k_slab_headers, dim3(1), dim3(64), 0, s, d, a, hdr_send, viol);
}

void launch_slab_append(const Dev &d, const float *recv, uint32_t n, uint32_t at, uint32_t *send_src, uint32_t own,
                        uint32_t cap_src, const SlabArgs &a, int from_side, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_slab_append, grid_for(n), dim3(kBlock), 0, s, d, recv, n, at, send_src, own, cap_src, a, from_side);
}

void launch_slab_restore(const Dev &d, uint32_t n, const Soa &snap, const uint32_t *snap_ck, const float4 *rv, const float *vw,
                         const uint32_t *id, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_slab_restore, grid_for(n), dim3(kBlock), 0, s, d, n, snap, snap_ck, rv, vw, id);
}

void launch_slab_pack_update(const Dev &d, const uint32_t *send_src, uint32_t n, int what, float *out, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_slab_pack_update, grid_for(n), dim3(kBlock), 0, s, d, send_src, n, what, out);
}

void launch_slab_unpack_update(const Dev &d, uint32_t src0, uint32_t n, int what, const float *in, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_slab_unpack_update, grid_for(n), dim3(kBlock), 0, s, d, src0, n, what, in);
}

void launch_slab_download(const Dev &d, int substeps_done, float dt_last, uint32_t *counter, uint32_t *ids,
                          float4 *pos4, float4 *vel4, float *rho, hipStream_t s)
{
    hipLaunchKernelGGL(k_slab_download, grid_for(d.count), dim3(kBlock), 0, s, d, substeps_done, dt_last, counter, ids,
                       pos4, vel4, rho);
}

void launch_slab_upload(const Dev &d, uint32_t n, const float4 *pos4, const float4 *vel4, const uint32_t *ids,
                        hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_slab_upload, grid_for(n), dim3(kBlock), 0, s, d, n, pos4, vel4, ids);
}

}  // namespace sph
