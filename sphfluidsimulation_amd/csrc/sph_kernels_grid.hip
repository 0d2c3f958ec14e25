// Uniform-grid build for gfx950: replaces Bucket.compute::ClearBucket (:38-53) and ::CSMain (:15-36).
//
// The reference keeps a 32-slot table per cell (128 B/cell, R^3 cells) filled by a CAS race.  Here
// the particles themselves are kept in CELL-SORTED SoA order, rebuilt once per frame:
//   key/histogram -> exclusive scan (cell_start) -> scatter -> id-rank inside each cell -> gather.
// Inside a cell, slots are in ASCENDING PARTICLE ID (= a serial execution of Bucket.compute:30-35),
// so "the first min(count,32) slots of the cell" is exactly the reference's bucket row
// (SURVEY.md 9.2).  Particles whose linear cell index falls outside the bucket
// go to a "limbo" cell (index R^3) that no sweep ever reads.
#include "sph_internal.hpp"

namespace sph {

constexpr int kBlock = 256;
constexpr int kScanThreads = 1024;
constexpr int kScanItems = 4;                          // per thread
constexpr int kScanTile = kScanThreads * kScanItems;   // per block

// ---- AoS (reference texture layout) <-> SoA ---------------------------------------------------

__global__ void __launch_bounds__(kBlock) k_upload_split(Dev d, const float4 *__restrict__ pos4,
                                                         const float4 *__restrict__ vel4,
                                                         float *__restrict__ pw0)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= d.count) return;
    const float4 q = pos4[p];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);        // ClearFloat4, SphFluidSimulation.cs:189
    if (vel4) v = vel4[p];
    if (d.prec) {                                      // reference cell of an uploaded particle: the cell it is in
        const int R = (int)d.R;
        d.cellkey[p] = pack_cellkey(cell_coord(q.x, d.Rm1, R), cell_coord(q.y, d.Rm1, R), cell_coord(q.z, d.Rm1, R));
    }
    store_pos(d, d.cur, p, Pos3{q.x, q.y, q.z});
    d.cur.rv[p] = make_float4(0.0f, v.x, v.y, v.z);    // density texture starts cleared
    d.cur.vw[p] = v.w;
    d.id[p] = p;                                       // id = y*res + x, Bucket.compute:24
    pw0[p] = q.w;
    if (!(fabsf(v.x) < 1e15f) || !(fabsf(v.y) < 1e15f) || !(fabsf(v.z) < 1e15f))
        d.counters[kCntWildV0 + d.wild_v_cur] = 1ull;   // see sph_kernels_sweep.hip
}

__global__ void __launch_bounds__(kBlock) k_download_merge(Dev d, const float *__restrict__ pw0,
                                                           int substeps_done, float dt_last,
                                                           float4 *__restrict__ pos4,
                                                           float4 *__restrict__ vel4,
                                                           float *__restrict__ rho_by_id)
{
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= d.count) return;
    const uint32_t id = d.id[s];
    const float vw = d.cur.vw[s];
    // VelPos.compute:153-154: pos_out.w = clamp(1 + dt*vel_out.w, 0, 1); the uploaded w before that
    const float pw = substeps_done > 0 ? clamp01(1.0f + dt_last * vw) : pw0[id];
    if (pos4) {
        Pos3 q = load_pos(d, d.cur, s);
        if (d.prec && substeps_done > 0) q = Pos3{clamp01(q.x), clamp01(q.y), clamp01(q.z)};   // (VelPos.compute:154 clamped before the rounding)
        pos4[id] = make_float4(q.x, q.y, q.z, pw);
    }
    const float4 rv = d.cur.rv[s];
    if (vel4) vel4[id] = make_float4(rv.y, rv.z, rv.w, vw);
    if (rho_by_id) rho_by_id[id] = rv.x;
}

// ---- cell key + histogram: Bucket.compute:26-29 ------------------------------------------------

// The particle order is last frame's cell order, and few particles change cell in a frame: neighbouring lanes mostly
// hold the same key.  A run of equal keys in a wave is served by ONE atomic of its first lane (~3.7x fewer atomics).
struct KeyRun {
    int head;        // first lane of this lane's run
    uint32_t len;    // lanes in the run
};
__device__ __forceinline__ KeyRun key_run(uint32_t key)
{
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t prev = __shfl_up(key, 1, kWave);
    const unsigned long long heads = __ballot(lane == 0 || key != prev);
    const unsigned long long upto = heads & (lane == kWave - 1 ? ~0ull : ((2ull << lane) - 1ull));
    const int head = 63 - __clzll((long long)upto);
    const unsigned long long above = head == kWave - 1 ? 0ull : heads & ~((2ull << head) - 1ull);
    const int next = above ? __ffsll((long long)above) - 1 : kWave;
    return KeyRun{head, (uint32_t)(next - head)};
}

__global__ void __launch_bounds__(kBlock) k_cell_key_hist(Dev d)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = p < d.count;
    uint32_t key = 0xFFFFFFFFu;                                                // no cell: lanes past the end
    if (valid) {
        const Pos3 q = load_pos(d, d.cur, p);
        const float x = q.x, y = q.y, z = q.z;
        key = bucket_key(d, x, y, z);                                          // :26-29
        d.key[p] = key;
        if (!(x >= 0.0f && x <= 1.0f && y >= 0.0f && y <= 1.0f && z >= 0.0f && z <= 1.0f))
            d.counters[kCntIrregular] = 1ull;                                  // only straight from an upload: VelPos clamps
    }
    const KeyRun r = key_run(key);
    if (valid && (int)(threadIdx.x & (kWave - 1)) == r.head) atomicAdd(&d.cell_count[key], r.len);
}

// ---- exclusive scan of cell_count[0 .. n) -> cell_start[0 .. n], n = num_cells + 1 ------------

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, kWave);
        if (lane >= o) v += t;
    }
    return v;
}

// exclusive scan of one value per thread across a 1024-thread block; returns block total in *total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *lds_wave_tot, uint32_t *total)
{
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const uint32_t incl = wave_incl_scan(v);
    if (lane == kWave - 1) lds_wave_tot[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        uint32_t t = lane < (kScanThreads / kWave) ? lds_wave_tot[lane] : 0u;
        const uint32_t ti = wave_incl_scan(t);
        if (lane < (kScanThreads / kWave)) lds_wave_tot[lane] = ti - t;  // exclusive wave offsets
        if (lane == kWave - 1) lds_wave_tot[kScanThreads / kWave] = ti;   // block total
    }
    __syncthreads();
    *total = lds_wave_tot[kScanThreads / kWave];
    return lds_wave_tot[wave] + incl - v;
}

__global__ void __launch_bounds__(kScanThreads) k_scan_reduce(Dev d, uint32_t n)
{
    __shared__ uint32_t lds[kScanThreads / kWave + 1];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t sum = 0, mx = 0, over = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; i++) {
        const uint32_t idx = base + i;
        if (idx < n) {
            const uint32_t c = d.cell_count[idx];
            sum += c;
            if (idx < d.num_cells) {                    // statistics exclude the limbo cell
                mx = c > mx ? c : mx;
                over += c > kMaxPerVoxel ? 1u : 0u;
            }
        }
    }
    uint32_t total;
    block_excl_scan(sum, lds, &total);
    if (threadIdx.x == 0) d.block_sums[blockIdx.x] = total;
    // statistics: wave reduction, then ONE atomic per block (16 waves hammering one address cost 0.2 ms)
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const uint32_t m2 = __shfl_down(mx, o, kWave);
        mx = m2 > mx ? m2 : mx;
        over += __shfl_down(over, o, kWave);
    }
    __shared__ uint32_t s_mx[kScanThreads / kWave], s_over[kScanThreads / kWave];
    if ((threadIdx.x & (kWave - 1)) == 0) {
        s_mx[threadIdx.x / kWave] = mx;
        s_over[threadIdx.x / kWave] = over;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kScanThreads / kWave; w++) {
            mx = s_mx[w] > mx ? s_mx[w] : mx;
            over += s_over[w];
        }
        if (mx) atomicMax(&d.counters[kCntMaxOcc], (unsigned long long)mx);
        if (over) atomicAdd(&d.counters[kCntOverflowCells], (unsigned long long)over);
    }
}

__global__ void __launch_bounds__(kScanThreads) k_scan_blocksums(Dev d, uint32_t nb)
{
    __shared__ uint32_t lds[kScanThreads / kWave + 1];
    const uint32_t per = (nb + kScanThreads - 1) / kScanThreads;
    const uint32_t b0 = threadIdx.x * per;
    uint32_t sum = 0;
    for (uint32_t i = 0; i < per; i++)
        if (b0 + i < nb) sum += d.block_sums[b0 + i];
    uint32_t total;
    uint32_t run = block_excl_scan(sum, lds, &total);
    for (uint32_t i = 0; i < per; i++)
        if (b0 + i < nb) {
            const uint32_t v = d.block_sums[b0 + i];
            d.block_sums[b0 + i] = run;
            run += v;
        }
}

__global__ void __launch_bounds__(kScanThreads) k_scan_final(Dev d, uint32_t n)
{
    __shared__ uint32_t lds[kScanThreads / kWave + 1];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t c[kScanItems];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; i++) {
        c[i] = (base + i < n) ? d.cell_count[base + i] : 0u;
        sum += c[i];
    }
    uint32_t total;
    uint32_t run = block_excl_scan(sum, lds, &total) + d.block_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < kScanItems; i++) {
        if (base + i < n) d.cell_start[base + i] = run;
        run += c[i];
        if (base + i == n - 1) {
            d.cell_start[n] = run;                      // == count
            d.counters[kCntDropped] = c[i];             // members of the limbo cell
        }
    }
}

// ---- scatter: unordered slot inside the cell (atomicSub leaves cell_count all zero) -----------

__global__ void __launch_bounds__(kBlock) k_scatter(Dev d, uint2 *__restrict__ order_tmp)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = p < d.count;
    const uint32_t key = valid ? d.key[p] : 0xFFFFFFFFu;
    const KeyRun r = key_run(key);                                             // one atomic per run of equal keys
    const int lane = threadIdx.x & (kWave - 1);
    uint32_t top = 0;
    if (valid && lane == r.head) top = atomicSub(&d.cell_count[key], r.len);   // the run takes [top - len, top)
    top = __shfl(top, r.head, kWave);
    if (valid) order_tmp[d.cell_start[key] + top - 1u - (uint32_t)(lane - r.head)] = make_uint2(d.id[p], p);
}

// ---- ascending-id order inside each cell (the CAS race made deterministic) ---------------------
// One thread per element: rank = number of same-cell elements with a smaller particle id, counted
// up to 32.  rank < 32  -> slot = rank (the visible part of the cell, exact order);
// rank >= 32 -> invisible to every sweep (Bucket.compute:30-35 gave up on it): any unique slot
// behind the first 32 will do, taken from a per-cell ticket (cell_count is zero after k_scatter
// and is memset again at the start of the next build).

__global__ void __launch_bounds__(kBlock) k_cell_rank(Dev d, const uint2 *__restrict__ order_tmp)
{
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= d.count) return;
    const uint2 me = order_tmp[t];
    const uint32_t key = d.key[me.y];
    if (key == d.num_cells) {                           // limbo: never read by a sweep, keep as is
        d.order[t] = me;
        return;
    }
    const uint32_t start = d.cell_start[key];
    const uint32_t cnt = d.cell_start[key + 1] - start;
    uint32_t rank = 0;
    constexpr uint32_t kTrip = 8;                       // ids of a trip loaded together (crowded cells hold hundreds of members)
    for (uint32_t k0 = 0; k0 < cnt && rank < kMaxPerVoxel; k0 += kTrip) {
        uint32_t ids[kTrip];
#pragma unroll
        for (uint32_t u = 0; u < kTrip; u++) ids[u] = k0 + u < cnt ? order_tmp[start + k0 + u].x : 0xFFFFFFFFu;
#pragma unroll
        for (uint32_t u = 0; u < kTrip; u++) rank += ids[u] < me.x ? 1u : 0u;
    }
    uint32_t slot = rank;
    if (rank >= kMaxPerVoxel) slot = kMaxPerVoxel + atomicAdd(&d.cell_count[key], 1u);
    d.order[start + slot] = me;
}

// ---- gather the payload into the new order: cur -> nxt ----------------------------------------

__global__ void __launch_bounds__(kBlock) k_reorder(Dev d)
{
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= d.count) return;
    const uint2 o = d.order[s];
    const uint32_t src = o.y;
    const Pos3 q = load_pos(d, d.cur, src);
    const float x = q.x, y = q.y, z = q.z;
    if (d.prec == 0u) {
        d.nxt.x[s] = x; d.nxt.y[s] = y; d.nxt.z[s] = z;
        if (d.p1x) {                                   // neighbour lists: where the particle was when it was binned
            d.p1x[s] = x; d.p1y[s] = y; d.p1z[s] = z;
            d.nxt.p4[s] = make_float4(x, y, z, 0.0f);
        }
    } else {
        // the new slot's reference cell is the cell the particle is in NOW: offsets are back in [-0.5, 0.5]
        const int R = (int)d.R;
        const uint32_t ck = pack_cellkey(cell_coord(x, d.Rm1, R), cell_coord(y, d.Rm1, R), cell_coord(z, d.Rm1, R));
        const uint2 o = encode_pos(d, q, ck);
        d.cellkey_nxt[s] = ck;
        d.nxt.h4[s] = o;
        if (d.p1h) {
            d.p1h[s] = o;
            reinterpret_cast<uint4 *>(d.nxt.p4)[s] = make_uint4(o.x, o.y, ck, 0u);
        }
    }
    d.nxt.rv[s] = d.cur.rv[src];                       // the density texture persists by id
    d.nxt.vw[s] = d.cur.vw[src];
    d.id_nxt[s] = o.x;
    if (d.slot_of_src) {                               // z-slab mode
        d.slot_of_src[src] = s;
        d.ghost_nxt[s] = src >= d.n_owned ? 1 : 0;
    }
}

// Zeroes the device counters named by `counter_mask` (bit = Counter) and the striped statistics named by `stat_mask`
// (bit = Stat) in ONE launch: a pass used to start with up to four 8-byte memsets, ~5 us each on the stream -- a tenth of
// a frame of the reference's own 262 144-particle scene.
__global__ void __launch_bounds__(kStatStripes) k_clear(Dev d, uint32_t counter_mask, uint32_t stat_mask)
{
    const uint32_t t = threadIdx.x;
    if (t < (uint32_t)kCntCount && ((counter_mask >> t) & 1u)) d.counters[t] = 0ull;
#pragma unroll
    for (int k = 0; k < kStatCount; k++)
        if ((stat_mask >> k) & 1u) d.stripes[((size_t)k * kStatStripes + t) * kStatPad] = 0ull;
}

// one float into device memory, the value travelling in the launch packet (no host staging buffer to race with)
__global__ void k_set_scalar(float *p, float v) { *p = v; }

// ---- launchers ---------------------------------------------------------------------------------

void launch_set_scalar(float *p, float v, hipStream_t s) { hipLaunchKernelGGL(k_set_scalar, dim3(1), dim3(1), 0, s, p, v); }

void launch_clear(const Dev &d, uint32_t counter_mask, uint32_t stat_mask, hipStream_t s)
{
    static_assert(kCntCount <= 32 && kStatCount <= 32 && kCntCount <= kStatStripes, "one bit, one thread per counter");
    hipLaunchKernelGGL(k_clear, dim3(1), dim3(kStatStripes), 0, s, d, counter_mask, stat_mask);
}

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return a ? (a + b - 1) / b : 1u; }  // never an empty grid

void launch_upload_split(const Dev &d, const float4 *pos4, const float4 *vel4, float *pw0, hipStream_t s)
{
    hipLaunchKernelGGL(k_upload_split, dim3(cdiv(d.count, kBlock)), dim3(kBlock), 0, s, d, pos4, vel4, pw0);
}

void launch_download_merge(const Dev &d, const float *pw0, int substeps_done, float dt_last,
                           float4 *pos4, float4 *vel4, float *rho_by_id, hipStream_t s)
{
    hipLaunchKernelGGL(k_download_merge, dim3(cdiv(d.count, kBlock)), dim3(kBlock), 0, s, d, pw0,
                       substeps_done, dt_last, pos4, vel4, rho_by_id);
}

void launch_grid_build(const Dev &d, hipStream_t s)
{
    const uint32_t n = d.num_cells + 1;                // + limbo
    const uint32_t nb = cdiv(n, kScanTile);
    const uint32_t pb = cdiv(d.count, kBlock);
    // order_tmp lives behind order (allocated 2*count)
    uint2 *order_tmp = d.order + d.count;
    (void)hipMemsetAsync(d.cell_count, 0, sizeof(uint32_t) * (size_t)n, s);
    launch_clear(d, 1u << kCntMaxOcc | 1u << kCntOverflowCells | 1u << kCntDropped | 1u << kCntIrregular, 0u, s);
    hipLaunchKernelGGL(k_cell_key_hist, dim3(pb), dim3(kBlock), 0, s, d);
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(kScanThreads), 0, s, d, n);
    hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(kScanThreads), 0, s, d, nb);
    hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(kScanThreads), 0, s, d, n);
    hipLaunchKernelGGL(k_scatter, dim3(pb), dim3(kBlock), 0, s, d, order_tmp);
    hipLaunchKernelGGL(k_cell_rank, dim3(pb), dim3(kBlock), 0, s, d, (const uint2 *)order_tmp);
    hipLaunchKernelGGL(k_reorder, dim3(pb), dim3(kBlock), 0, s, d);
}

}  // namespace sph
