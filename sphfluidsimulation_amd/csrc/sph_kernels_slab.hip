// z-slab domain decomposition (one process per GPU): device side of the per-frame particle
// migration and of the ghost-plane exchange.  New work -- the reference is single-GPU
// (SURVEY.md 8e).  A rank owns the grid planes cz in [z_lo, z_hi) by BUCKET ROW (Bucket.compute:26-29)
// and keeps copies ("ghosts") of the W planes on either side; ghosts are candidates only.
// Membership is frozen for the frame like the reference's bucket (SphFluidSimulation.cs:98-102), so
// the ghost sets are fixed per frame and only their pos / vel values are refreshed per sub-step.
//
// Record layouts (fp32 words), written to caller-provided device buffers that the host side
// moves with torch.distributed (RCCL over xGMI, or gloo in tests):
//   particle record (migration, ghost export): x y z rho vx vy vz vel.w id     -> 9 words
//   ghost update, density pass               : rho                             -> 1 word
//   ghost update, sub-step                   : x y z rho vx vy vz              -> 7 words
//     (rho rides along: ghost slots are not integrated locally, so the ping-pong partner of a
//      ghost slot holds nothing until the update lands)
#include "sph_internal.hpp"

namespace sph {

constexpr int kBlock = 256;

struct SlabArgs {
    int z_lo, z_hi, halo_w;
    int has_left, has_right;
    float *send[2];         // [0] left neighbour, [1] right neighbour
    uint32_t *send_src[2];  // pre-build index of every exported ghost
    uint32_t *counters;     // [0] keep [1] to-left [2] to-right [3] lost (jumped past a neighbour / out of grid)
    uint32_t cap_records;
};

__device__ __forceinline__ void write_record(float *buf, uint32_t k, const Dev &d, uint32_t p)
{
    float *r = buf + 9ull * k;
    const float4 rv = d.cur.rv[p];
    r[0] = d.cur.x[p]; r[1] = d.cur.y[p]; r[2] = d.cur.z[p];
    r[3] = rv.x; r[4] = rv.y; r[5] = rv.z; r[6] = rv.w;
    r[7] = d.cur.vw[p];
    r[8] = __uint_as_float(d.id[p]);
}

// Stream compaction into up to three output lists at once: every thread of the block calls it with the list it
// appends to (0..2, or -1 for none) and gets its index in that list.  ONE global atomic per list and block: an atomic
// per particle on a single counter costs ~3 ns each whatever else the kernel does (3 ms per frame at 16.7 M).
__device__ __forceinline__ uint32_t block_append3(int which, uint32_t *counters, uint32_t *s_cnt /* [6] */)
{
    const int lane = threadIdx.x & (kWave - 1);
    if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t in_block = 0;
#pragma unroll
    for (int w = 0; w < 3; w++) {
        const unsigned long long m = __ballot(which == w);
        if (m == 0ull) continue;                   // wave-uniform
        uint32_t wave_off = 0;
        if (lane == __ffsll((long long)m) - 1) wave_off = atomicAdd(&s_cnt[w], (uint32_t)__popcll(m));
        wave_off = __shfl(wave_off, __ffsll((long long)m) - 1, kWave);
        if (which == w) in_block = wave_off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x < 3 && s_cnt[threadIdx.x]) s_cnt[3 + threadIdx.x] = atomicAdd(&counters[threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    return which >= 0 ? s_cnt[3 + which] + in_block : 0u;
}

// Frame start: drop last frame's ghosts, keep the particles whose bucket row is still in this slab
// (compacted into nxt), pack the ones that crossed into a neighbour's slab.
__global__ void __launch_bounds__(kBlock) k_slab_classify(Dev d, SlabArgs a)
{
    __shared__ uint32_t s_cnt[6];
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    int which = -1;                                // 0 keep, 1 to the left neighbour, 2 to the right one
    if (s < d.count && !d.ghost[s]) {
        const uint32_t key = bucket_key(d, d.cur.x[s], d.cur.y[s], d.cur.z[s]);
        which = 0;
        if (key < d.num_cells) {
            const int cz = (int)(key / (d.R * d.R));
            if (cz < a.z_lo && a.has_left) which = 1;          // (a particle cannot lie beyond a missing neighbour)
            else if (cz >= a.z_hi && a.has_right) which = 2;
        }
    }
    const uint32_t k = block_append3(which, a.counters, s_cnt);
    if (which == 0) {
        d.nxt.x[k] = d.cur.x[s]; d.nxt.y[k] = d.cur.y[s]; d.nxt.z[k] = d.cur.z[s];
        d.nxt.rv[k] = d.cur.rv[s];
        d.nxt.vw[k] = d.cur.vw[s];
        d.id_nxt[k] = d.id[s];
    } else if (which > 0) {
        if (k < a.cap_records) write_record(a.send[which - 1], k, d, s);
    }
}

// Append received particle records behind the first `at` entries of cur.
__global__ void __launch_bounds__(kBlock) k_slab_append(Dev d, const float *__restrict__ recv, uint32_t n, uint32_t at)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    const float *r = recv + 9ull * k;
    const uint32_t p = at + k;
    d.cur.x[p] = r[0]; d.cur.y[p] = r[1]; d.cur.z[p] = r[2];
    d.cur.rv[p] = make_float4(r[3], r[4], r[5], r[6]);
    d.cur.vw[p] = r[7];
    d.id[p] = __float_as_uint(r[8]);
}

// Owned particles binned into the W boundary planes of either face become the neighbour's ghosts.
__global__ void __launch_bounds__(kBlock) k_slab_export(Dev d, SlabArgs a)
{
    __shared__ uint32_t s_cnt[6];
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    bool to_left = false, to_right = false;
    if (p < d.n_owned) {
        const uint32_t key = bucket_key(d, d.cur.x[p], d.cur.y[p], d.cur.z[p]);
        if (key < d.num_cells) {
            const int cz = (int)(key / (d.R * d.R));
            to_left = a.has_left && cz < a.z_lo + a.halo_w;
            to_right = a.has_right && cz >= a.z_hi - a.halo_w;
        }
    }
    // (a slab thinner than 2 * halo_w exports a particle to both sides: two passes)
    const uint32_t kl = block_append3(to_left ? 1 : -1, a.counters, s_cnt);
    if (to_left && kl < a.cap_records) {
        write_record(a.send[0], kl, d, p);
        a.send_src[0][kl] = p;
    }
    const uint32_t kr = block_append3(to_right ? 2 : -1, a.counters, s_cnt);
    if (to_right && kr < a.cap_records) {
        write_record(a.send[1], kr, d, p);
        a.send_src[1][kr] = p;
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
        r[0] = d.cur.x[s]; r[1] = d.cur.y[s]; r[2] = d.cur.z[s];
        r[3] = rv.x; r[4] = rv.y; r[5] = rv.z; r[6] = rv.w;
    }
}

__global__ void __launch_bounds__(kBlock) k_slab_unpack_update(Dev d, uint32_t src0, uint32_t n, int what,
                                                               const float *__restrict__ in)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    const uint32_t s = d.slot_of_src[src0 + k];
    if (what == 0) {
        d.cur.rv[s].x = in[k];
    } else {
        const float *r = in + 7ull * k;
        d.cur.x[s] = r[0]; d.cur.y[s] = r[1]; d.cur.z[s] = r[2];
        if (d.cur.p4) d.cur.p4[s] = make_float4(r[0], r[1], r[2], 0.0f);   // the packed copy the list kernel gathers
        d.cur.rv[s] = make_float4(r[3], r[4], r[5], r[6]);
    }
}

// Owned particles, compacted, in the reference's texture layout (+ ids).
__global__ void __launch_bounds__(kBlock) k_slab_download(Dev d, int substeps_done, float dt_last, uint32_t *counter,
                                                          uint32_t *ids, float4 *pos4, float4 *vel4, float *rho)
{
    __shared__ uint32_t s_cnt[6];
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    const bool mine = s < d.count && !d.ghost[s];
    const uint32_t k = block_append3(mine ? 0 : -1, counter, s_cnt);
    if (!mine) return;
    const float4 rv = d.cur.rv[s];
    const float vw = d.cur.vw[s];
    const float pw = substeps_done > 0 ? clamp01(1.0f + dt_last * vw) : 1.0f;   // VelPos.compute:153-154
    ids[k] = d.id[s];
    pos4[k] = make_float4(d.cur.x[s], d.cur.y[s], d.cur.z[s], pw);
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
    d.cur.x[p] = q.x; d.cur.y[p] = q.y; d.cur.z[p] = q.z;
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

void launch_slab_classify(const Dev &d, int z_lo, int z_hi, int has_left, int has_right, float *send_l, float *send_r,
                          uint32_t cap_records, uint32_t *counters, hipStream_t s)
{
    SlabArgs a{z_lo, z_hi, 0, has_left, has_right, {send_l, send_r}, {nullptr, nullptr}, counters, cap_records};
    hipLaunchKernelGGL(k_slab_classify, grid_for(d.count), dim3(kBlock), 0, s, d, a);
}

void launch_slab_append(const Dev &d, const float *recv, uint32_t n, uint32_t at, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_slab_append, grid_for(n), dim3(kBlock), 0, s, d, recv, n, at);
}

void launch_slab_export(const Dev &d, int z_lo, int z_hi, int halo_w, int has_left, int has_right, float *send_l,
                        float *send_r, uint32_t *src_l, uint32_t *src_r, uint32_t cap_records, uint32_t *counters,
                        hipStream_t s)
{
    SlabArgs a{z_lo, z_hi, halo_w, has_left, has_right, {send_l, send_r}, {src_l, src_r}, counters, cap_records};
    hipLaunchKernelGGL(k_slab_export, grid_for(d.n_owned), dim3(kBlock), 0, s, d, a);
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
