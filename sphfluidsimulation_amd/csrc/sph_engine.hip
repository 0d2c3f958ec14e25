// Host side of libsph_hip.so: context, device memory, the frame schedule of
// SphFluidSimulation.cs::Update (:96-102) and the C ABI declared in include/sph_hip.h.
#include <algorithm>
#include "sph_context.hpp"

#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <cstdlib>
#include <string>
#include <vector>

using namespace sph;

thread_local std::string g_last_error = "";

// ---- derived parameters: SphFluidSimulation.cs:84-85,159,176,236-237,255 -------

static uint32_t next_power_of_two(uint32_t v)
{
    uint32_t p = 1;
    while (p < v && p < 0x80000000u) p <<= 1;
    return p;
}

static void derive(const SphParams &p, SphDerived *d)
{
    const uint32_t N = next_power_of_two(p.num_particles);               // :84
    const uint32_t res = (uint32_t)(int32_t)(float)sqrt((double)(float)N); // :85 (int)Mathf.Sqrt
    d->num_particles = N;
    d->particle_resolution = res;
    d->particle_count = res * res;
    d->bucket_resolution = p.bucket_resolution;
    d->effective_radius = 1.0f / (float)((int32_t)p.bucket_resolution - 1); // :159
    d->particle_mass = p.dam_fill_rate / (float)(int32_t)N;              // :176
    d->effective_radius2 = d->effective_radius * d->effective_radius;    // :236
    d->effective_radius9 = (float)pow((double)d->effective_radius, 9.0); // :237 Mathf.Pow
    d->effective_radius6 = (float)pow((double)d->effective_radius, 6.0); // :255
}

void refresh_scalars(SphContext *c)
{
    const SphParams &p = c->params;
    const SphDerived &dv = c->derived;
    Dev &d = c->dev;
    const float PI = 3.14159265358979323846264338327950288f;  // Density.compute:5, VelPos.compute:6
    if (!c->slab) d.count = dv.particle_count;   // slab contexts hold a varying local subset
    d.sentinel_n = dv.num_particles;
    d.R = dv.bucket_resolution;
    d.num_cells = d.R * d.R * d.R;
    d.Rm1 = (float)(d.R - 1u);
    d.Rf = (float)d.R;
    d.mass = dv.particle_mass;
    d.h = dv.effective_radius;
    d.h2 = dv.effective_radius2;
    d.h9 = dv.effective_radius9;
    d.h2_up = nextafterf((float)((double)d.h * (double)d.h * (1.0 + 1e-5)), INFINITY);
    d.c_poly6 = 315.0f / (64.0f * PI);                        // Density.compute:23
    d.w_scale = (float)((double)dv.particle_mass * (double)d.c_poly6 / (double)dv.effective_radius9);
    d.c_over_h6 = (45.0f / PI) / dv.effective_radius6;        // VelPos.compute:34,37,41,43
    d.viscosity = p.viscosity;
    d.gas_const = p.gas_constant;
    d.rest_density = p.rest_density;
    d.stiffness = p.stiffness_coeff;
    d.damping = p.damping_coeff;
    d.gx = p.gravity[0];
    d.gy = p.gravity[1];
    d.gz = p.gravity[2];
    d.use_bricks = (p.flags & SPH_FLAG_NO_TILING) ? 0u : 1u;
    d.prec = p.precision;
    d.fast = p.exact_order == 0u ? 1u : 0u;
    d.nl_tile = (p.flags & SPH_FLAG_GLOBAL_LISTS) ? 0u : 1u;
    d.box_mode = (p.flags & SPH_FLAG_CHUNK_BOXES) ? 0u : 1u;
    {
        // k_velpos_tile with one record slot or two (sph_brick_impl.inc): a brick is halved when the listed particles of its
        // halo-1 box (216 cells) exceed the 928-particle tile -- expected where the occupied cells hold more than ~4 particles
        // (the dam fills `fill` x 0.9 of the box; 3.7 per cell at the scene's R, 5.5 under the R <= 256 cap of config C4)
        const double per_cell = (double)dv.particle_count / ((double)d.num_cells * 0.9 * (double)std::min(std::max(p.dam_fill_rate, 0.05f), 1.0f));
        d.tile_slots = per_cell * 216.0 > 0.95 * 928.0 ? 2u : 1u;
        if (p.flags & SPH_FLAG_TILE_SLOTS_ONE) d.tile_slots = 1u;
        if (p.flags & SPH_FLAG_TILE_SLOTS_TWO) d.tile_slots = 2u;
    }
    // neighbour lists: skin = nl_skin_frac * h; pre-filter radius (h + skin)^2 rounded up like h2_up; a brick is calm
    // while its relative-displacement bound stays below 90 % of the skin (the rest absorbs fp32 rounding of the bound)
    c->nl_enabled = d.use_bricks && !(p.flags & SPH_FLAG_NO_LISTS) && d.R <= 256u && !c->nl_unavailable;
    c->nl_skin_frac = p.list_skin > 0.0f ? p.list_skin : 0.04f;   // (profiles/: 0.04-0.06 is the optimum of the 16.7 M dam break)
    {
        const double hs = (double)d.h * (1.0 + (double)c->nl_skin_frac);
        d.h2_list = nextafterf((float)(hs * hs * (1.0 + 1e-5)), INFINITY);
        d.h2_stage = nextafterf((float)((double)d.h2_list * (1.0 + 1e-5)), INFINITY);
        d.skin_ok = (float)(0.9 * (double)c->nl_skin_frac * (double)d.h);
    }
    if (!c->slab) {
        d.zt_lo = INT_MIN;
        d.zt_hi = INT_MAX;
        d.zo_lo = 0;
        d.zo_hi = (int)d.R;
    }
}

static int validate(SphContext *ctx, const SphParams *p)
{
    if (!p) return fail(ctx, SPH_E_INVALID_ARG, "params is NULL");
    if (p->struct_size != sizeof(SphParams))
        return fail(ctx, SPH_E_INVALID_ARG, "SphParams.struct_size %u != %zu (ABI mismatch)",
                    p->struct_size, sizeof(SphParams));
    if (p->num_particles == 0 || p->num_particles > (1u << 28))
        return fail(ctx, SPH_E_INVALID_ARG, "num_particles %u out of range [1, 2^28]", p->num_particles);
    if (p->bucket_resolution < 2 || p->bucket_resolution > 256)
        return fail(ctx, SPH_E_INVALID_ARG, "bucket_resolution %u out of range [2, 256]", p->bucket_resolution);
    if (p->max_per_voxel != kMaxPerVoxel)
        return fail(ctx, SPH_E_UNSUPPORTED, "max_per_voxel must be 32 (MaxParticlesPerVoxel)");
    if (p->substeps == 0 || p->substeps > 1000) return fail(ctx, SPH_E_INVALID_ARG, "substeps out of range");
    if (!(p->dt_divisor > 0.0f)) return fail(ctx, SPH_E_INVALID_ARG, "dt_divisor must be > 0");
    if (p->precision > 1) return fail(ctx, SPH_E_UNSUPPORTED, "precision %u not supported (0 = fp32, 1 = fp16 positions)", p->precision);
    if (p->exact_order > 1) return fail(ctx, SPH_E_INVALID_ARG, "exact_order must be 0 (tolerance mode) or 1");
    if (p->num_devices > 8) return fail(ctx, SPH_E_INVALID_ARG, "num_devices %u > 8", p->num_devices);
    if (p->halo_planes > 128) return fail(ctx, SPH_E_INVALID_ARG, "halo_planes %u > 128", p->halo_planes);
    if (!(p->list_skin >= 0.0f && p->list_skin <= 1.0f)) return fail(ctx, SPH_E_INVALID_ARG, "list_skin must be in [0, 1]");
    return SPH_OK;
}

template <typename T>
static int dalloc(SphContext *c, T **out, size_t n)
{
    void *p = nullptr;
    HIP_TRY(c, hipMalloc(&p, n * sizeof(T)));
    c->allocs.push_back(p);
    *out = (T *)p;
    return SPH_OK;
}

static int alloc_soa(SphContext *c, Soa *s, size_t n)
{
    int rc;
    if (c->params.precision == 0) {
        float **f[] = {&s->x, &s->y, &s->z};
        for (auto pp : f)
            if ((rc = dalloc(c, pp, n)) != SPH_OK) return rc;
    } else if ((rc = dalloc(c, &s->h4, n)) != SPH_OK) {   // fp16 offsets from the slot's reference cell: 8 bytes instead of 12
        return rc;
    }
    if ((rc = dalloc(c, &s->vw, n)) != SPH_OK) return rc;
    return dalloc(c, &s->rv, n);
}

// ---- profiling events ---------------------------------------------------------------------------

static hipEvent_t get_event(SphContext *c)
{
    if (!c->free_events.empty()) {
        hipEvent_t e = c->free_events.back();
        c->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

static void fold_events(SphContext *c)
{
    for (auto &pe : c->pending) {
        float ms = 0.0f;
        (void)hipEventSynchronize(pe.b);
        if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
            c->ms[pe.kind] += ms;
            c->launches[pe.kind]++;
        }
        c->free_events.push_back(pe.a);
        c->free_events.push_back(pe.b);
    }
    c->pending.clear();
}

struct PassTimer {
    SphContext *c;
    PassEvents pe{};
    bool on;
    PassTimer(SphContext *ctx, int kind) : c(ctx), on(ctx->profile)
    {
        if (!on) return;
        pe.kind = kind;
        pe.a = get_event(c);
        pe.b = get_event(c);
        (void)hipEventRecord(pe.a, c->stream);
    }
    ~PassTimer()
    {
        if (!on) return;
        (void)hipEventRecord(pe.b, c->stream);
        c->pending.push_back(pe);
        if (c->pending.size() >= 4096) fold_events(c);
    }
};

// ---- the three driver passes ---------------------------------------------------------------------

// Buffers of the neighbour lists, on first use: 64 u32 entries + a count per particle, the binned positions, and the
// per-brick displacement boxes.  If the device cannot hold them the engine simply keeps scanning.
static void ensure_lists(SphContext *c)
{
    if (!c->nl_enabled || c->nl_allocated) return;
    Dev &d = c->dev;
    const size_t n = c->capacity;
    if (n == 0 || n >= (1u << kNlSlotBits)) {
        c->nl_enabled = false;
        c->nl_unavailable = true;
        return;
    }
    const size_t bricks = (size_t)nl_bricks_x(d.R) * nl_bricks_y(d.R) * nl_bricks_z(d.R);
    d.nl_cap = c->nl_cap_want;
    d.n_bricks = (uint32_t)bricks;
    d.n_chunks = (uint32_t)((n + kChunkSlots - 1) / kChunkSlots);
    const size_t before = c->allocs.size();
    const bool binned = d.prec ? dalloc(c, &d.p1h, n) == SPH_OK
                               : dalloc(c, &d.p1x, n) == SPH_OK && dalloc(c, &d.p1y, n) == SPH_OK && dalloc(c, &d.p1z, n) == SPH_OK;
    const bool ok = binned &&
                    dalloc(c, &d.nl_ent, (size_t)d.nl_cap * ((n + 63) & ~(size_t)63) + kNlRowsPad * kWave) == SPH_OK && dalloc(c, &d.nl_cnt, n) == SPH_OK &&
                    dalloc(c, &d.chunk_box, 6 * ((n + kChunkSlots - 1) / kChunkSlots)) == SPH_OK &&
                    dalloc(c, &d.active_list, bricks) == SPH_OK && dalloc(c, &d.queue2, n) == SPH_OK && dalloc(c, &d.brick_state, bricks + 4) == SPH_OK &&
                    dalloc(c, &d.cur.p4, n) == SPH_OK && dalloc(c, &d.nxt.p4, n) == SPH_OK &&
                    dalloc(c, &d.brick_rec, bricks * (size_t)(kBrickRecSlots * kBrickRecWords)) == SPH_OK &&
                    dalloc(c, &d.boxrec, 2 * kBoxRecWords * bricks) == SPH_OK;
    if (!ok) {
        while (c->allocs.size() > before) {
            (void)hipFree(c->allocs.back());
            c->allocs.pop_back();
        }
        (void)hipGetLastError();
        d.p1x = d.p1y = d.p1z = nullptr;
        d.p1h = nullptr;
        d.nl_ent = nullptr;
        d.nl_cnt = nullptr;
        d.chunk_box = nullptr;
        d.active_list = nullptr;
        d.queue2 = nullptr;
        d.brick_state = nullptr;
        d.brick_rec = nullptr;
        d.boxrec = nullptr;
        d.cur.p4 = d.nxt.p4 = nullptr;
        c->nl_enabled = false;
        c->nl_unavailable = true;     // (sticky: a later sph_set_params must not try the multi-GB allocation again)
        c->last_error.clear();
        return;
    }
    c->nl_allocated = true;
    launch_box_init(d, c->stream);
}

int do_bucket_generation(SphContext *c)
{
    if (!c->uploaded) return fail(c, SPH_E_BAD_STATE, "BucketGeneration before sph_upload");
    if (c->pol_on) ensure_lists(c);   // (several GB at 16.7 M: only once a frame is going to record)
    c->dev.use_lists = c->nl_enabled && c->nl_allocated && c->pol_on ? 1u : 0u;
    c->nl_fresh = c->dev.use_lists != 0u;
    c->nl_recorded = false;
    c->substep_in_frame = 0;
    {
        PassTimer t(c, 0);
        launch_grid_build(c->dev, c->stream);
    }
    HIP_TRY(c, hipGetLastError());
    std::swap(c->dev.cur, c->dev.nxt);
    std::swap(c->dev.id, c->dev.id_nxt);
    std::swap(c->dev.cellkey, c->dev.cellkey_nxt);
    std::swap(c->dev.ghost, c->dev.ghost_nxt);
    c->grid_valid = true;
    c->frames++;
    return SPH_OK;
}

int do_density(SphContext *c)
{
    if (!c->grid_valid) return fail(c, SPH_E_BAD_STATE, "DensityCalculation before BucketGeneration");
    // the density pass of a fresh grid sees every particle where it was binned: it records the neighbour lists
    c->dev.nl_mode = c->nl_fresh ? kNlBuild : kNlOff;
    {
        PassTimer t(c, 1);
        launch_density(c->dev, c->stream);
    }
    if (c->nl_fresh) c->nl_recorded = true;
    HIP_TRY(c, hipGetLastError());
    c->dev.force_slow = 0;  // the pressure flag now matches the current gas_const / rest_density
    return SPH_OK;
}

static void velpos_fields(SphContext *c, float dt)
{
    c->dev.dt = dt;
    c->dev.nl_mode = c->dev.use_lists && c->nl_recorded ? kNlUse : kNlOff;
    c->dev.nl_still = c->nl_fresh ? 1u : 0u;   // first sub-step since the grid build: the recorded positions still stand
    c->dev.box_par = (uint32_t)c->substep_in_frame & 1u;
    c->nl_fresh = false;              // positions move: a later density pass of this frame must not record
}

static void velpos_issue(SphContext *c, uint32_t pass_mode)
{
    Dev d = c->dev;
    d.pass_mode = pass_mode;
    PassTimer t(c, 2);
    if (c->profile && d.nl_mode == kNlUse) {   // the dominant kernel on its own, for the roofline figure
        PassEvents pe{};
        pe.kind = 3;
        pe.a = get_event(c);
        pe.b = get_event(c);
        launch_velpos(d, c->stream, pe.a, pe.b, c->fork_scan ? c->side_stream : nullptr, c->ev_fork, c->ev_join);
        c->pending.push_back(pe);
    } else {
        launch_velpos(d, c->stream, nullptr, nullptr, c->fork_scan ? c->side_stream : nullptr, c->ev_fork, c->ev_join);
    }
}

static void velpos_done(SphContext *c, float dt)
{
    std::swap(c->dev.cur, c->dev.nxt);  // Swap(), SphFluidSimulation.cs:267-268
    c->dev.wild_v_cur ^= 1u;
    c->substeps++;
    c->substeps_since_upload++;
    c->substep_in_frame++;
    c->dt_last = dt;
}

int do_velpos(SphContext *c, float dt)
{
    if (!c->grid_valid) return fail(c, SPH_E_BAD_STATE, "UpdateVelocityAndPosition before BucketGeneration");
    velpos_fields(c, dt);
    velpos_issue(c, 0u);
    HIP_TRY(c, hipGetLastError());
    velpos_done(c, dt);
    return SPH_OK;
}

// The same sub-step in two passes (z-slab mode): _begin issues the targets binned in the slab's boundary planes -- what the
// neighbours hold as ghosts -- so that the caller can send their new values while _end issues all the others.  Together
// they launch exactly what do_velpos launches, over disjoint target sets.
int do_velpos_begin(SphContext *c, float dt)
{
    if (!c->grid_valid) return fail(c, SPH_E_BAD_STATE, "UpdateVelocityAndPosition before BucketGeneration");
    velpos_fields(c, dt);
    velpos_issue(c, 1u);
    HIP_TRY(c, hipGetLastError());
    return SPH_OK;
}

int do_velpos_end(SphContext *c, float dt)
{
    velpos_issue(c, 2u);
    HIP_TRY(c, hipGetLastError());
    velpos_done(c, dt);
    return SPH_OK;
}

// Lists longer than the entry rows leave their whole brick scanning (crowded regions of a collapsed dam: ~130 candidates within
// h where the calm fluid has ~18).  When a frame reports many of them the rows are re-allocated twice as long (64 -> 128 -> 256
// entries; 4 B x nl_cap per particle, only the used part is ever touched).  Never inside a capture; frame graphs re-capture
// by themselves because the launch state (Dev) changes.
static void grow_list_rows(SphContext *c)
{
    if (!c->report_host || !c->nl_allocated || c->slab || c->nl_cap_want >= 256u) return;
    const unsigned long long too_long = c->report_host[3];
    const unsigned long long many = c->dev.count / 4096u > 64u ? c->dev.count / 4096u : 64u;
    if (too_long < many || c->report_host[0] <= c->nl_hold_until) return;   // (only frames issued with the current rows count)
    Dev &d = c->dev;
    const uint32_t cap = c->nl_cap_want * 2u;
    const size_t n = c->capacity;
    uint32_t *rows = nullptr;
    (void)hipStreamSynchronize(c->stream);
    if (hipMalloc((void **)&rows, ((size_t)cap * ((n + 63) & ~(size_t)63) + kNlRowsPad * kWave) * sizeof(uint32_t)) != hipSuccess) {
        (void)hipGetLastError();
        c->nl_cap_want = 256u;        // (no memory for longer rows: stop asking)
        return;
    }
    for (void *&p : c->allocs)
        if (p == (void *)d.nl_ent) {
            (void)hipFree(p);
            p = rows;
        }
    d.nl_ent = rows;
    d.nl_cap = cap;
    c->nl_cap_want = cap;
    c->nl_hold_until = c->pol_issued;
}

// ---- C ABI -----------------------------------------------------------------------------------------

extern "C" {

int sph_version(void) { return SPH_ABI_VERSION; }

int sph_params_default(SphParams *p)
{
    if (!p) return fail(nullptr, SPH_E_INVALID_ARG, "out is NULL");
    memset(p, 0, sizeof *p);
    p->struct_size = sizeof(SphParams);
    p->num_particles = 1024;      // SphFluidSimulation.cs:36
    p->bucket_resolution = 256;   // :37
    p->dam_fill_rate = 0.5f;      // :38
    p->viscosity = 0.01f;         // :41
    p->rest_density = 1.5f;       // :42
    p->gas_constant = 150.0f;     // :43
    p->stiffness_coeff = 5000.0f; // :44
    p->damping_coeff = 10.0f;     // :45
    p->max_per_voxel = 32;        // :9
    p->substeps = 5;              // :101
    p->dt_divisor = 25.0f;        // :102
    p->gravity[0] = 0.0f;         // VelPos.compute:7
    p->gravity[1] = -9.8f;
    p->gravity[2] = 0.0f;
    p->precision = 0;
    p->exact_order = 1;
    p->device_id = 0;
    p->flags = 0;
    return SPH_OK;
}

static int create_common(const SphParams *params, uint32_t capacity, bool slab, SphContext **out)
{
    if (!out) return fail(nullptr, SPH_E_INVALID_ARG, "out is NULL");
    *out = nullptr;
    int rc = validate(nullptr, params);
    if (rc != SPH_OK) return rc;

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, SPH_E_HIP, "no HIP device available (%s); libsph_hip has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (params->device_id < 0 || params->device_id >= ndev)
        return fail(nullptr, SPH_E_INVALID_ARG, "device_id %d not in [0,%d)", params->device_id, ndev);

    SphContext *c = new SphContext();
    c->params = *params;
    c->device = params->device_id;
    c->profile = (params->flags & SPH_FLAG_PROFILE) != 0;
    c->slab = slab;
    derive(c->params, &c->derived);
    refresh_scalars(c);
    c->capacity = slab ? capacity : c->dev.count;
    if (slab) c->dev.count = 0;

    auto bail = [&](int code) {
        std::string msg = c->last_error;
        sph_destroy(c);
        g_last_error = msg;
        return code;
    };
    if (hipSetDevice(c->device) != hipSuccess) {
        fail(c, SPH_E_HIP, "hipSetDevice(%d) failed", c->device);
        return bail(SPH_E_HIP);
    }
    // (plain contexts: a side stream for the scanning kernel of a list-using sub-step, sph_brick_impl.inc launch_velpos; without it
    // -- SPH_FLAG_NO_SIDE_STREAM, z-slab contexts -- everything is issued in stream order)
    if (!slab && !(params->flags & SPH_FLAG_NO_SIDE_STREAM) &&
        (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess ||
         hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
         hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)) {
        (void)hipGetLastError();
        c->side_stream = nullptr;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        fail(c, SPH_E_HIP, "hipStreamCreate failed");
        return bail(SPH_E_HIP);
    }
    c->own_stream = true;

    Dev &d = c->dev;
    const size_t n = c->capacity;
    const size_t cells = (size_t)d.num_cells + 2;
    const size_t scan_blocks = (cells + 4095) / 4096 + 1;
#define A(expr)                         \
    if ((rc = (expr)) != SPH_OK) return bail(rc)
    A(alloc_soa(c, &d.cur, n));
    A(alloc_soa(c, &d.nxt, n));
    A(dalloc(c, &d.id, n));
    A(dalloc(c, &d.id_nxt, n));
    if (d.prec) {
        A(dalloc(c, &d.cellkey, n));
        A(dalloc(c, &d.cellkey_nxt, n));
    }
    A(dalloc(c, &d.key, n));
    A(dalloc(c, &d.cell_count, cells));
    A(dalloc(c, &d.cell_start, cells));
    A(dalloc(c, &d.order, 2 * n));
    A(dalloc(c, &d.block_sums, scan_blocks));
    A(dalloc(c, &d.queue, n));
    A(dalloc(c, &d.counters, (size_t)kCntCount));
    A(dalloc(c, &d.stripes, (size_t)kStatCount * kStatStripes * kStatPad));
    A(dalloc(c, &c->dt_dev, (size_t)1));
    c->graphs_enabled = !slab && !c->profile && !(params->flags & SPH_FLAG_NO_GRAPHS);
    {
        void *h = nullptr, *dv = nullptr;
        if (hipHostMalloc(&h, 8 * sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer(&dv, h, 0) == hipSuccess) {
            memset(h, 0, 8 * sizeof(unsigned long long));
            c->report_host = (volatile unsigned long long *)h;
            c->report_dev = (unsigned long long *)dv;
        } else {
            (void)hipGetLastError();
            if (h) (void)hipHostFree(h);
        }
    }
    A(dalloc(c, &c->pos4_dev, n));
    A(dalloc(c, &c->vel4_dev, n));
    A(dalloc(c, &c->rho_id_dev, n));
    A(dalloc(c, &c->pw0, n));
    if (slab) {
        A(dalloc(c, &d.ghost, n));
        A(dalloc(c, &d.ghost_nxt, n));
        A(dalloc(c, &d.slot_of_src, n));
        A(dalloc(c, &c->download_counter, (size_t)4));
        A(dalloc(c, &c->ids_dev, n));
    }
#undef A
    if (hipMemsetAsync(d.counters, 0, sizeof(unsigned long long) * kCntCount, c->stream) != hipSuccess ||
        hipMemsetAsync(d.stripes, 0, sizeof(unsigned long long) * kStatCount * kStatStripes * kStatPad, c->stream) != hipSuccess ||
        hipMemsetAsync(d.cell_count, 0, sizeof(uint32_t) * cells, c->stream) != hipSuccess ||
        hipMemsetAsync(d.cell_start, 0, sizeof(uint32_t) * cells, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        fail(c, SPH_E_HIP, "device memset failed");
        return bail(SPH_E_HIP);
    }
    if (slab && (hipMemset(d.ghost, 0, n) != hipSuccess || hipMemset(d.ghost_nxt, 0, n) != hipSuccess)) {
        fail(c, SPH_E_HIP, "device memset failed");
        return bail(SPH_E_HIP);
    }
    *out = c;
    return SPH_OK;
}

int sph_create(const SphParams *params, SphContext **out)
{
    if (params && params->struct_size == sizeof(SphParams) && params->num_devices >= 2) {
        if (!out) return fail(nullptr, SPH_E_INVALID_ARG, "out is NULL");
        *out = nullptr;
        int rc = validate(nullptr, params);
        if (rc != SPH_OK) return rc;
        return group_create(params, out);               // sph_slab_frame.hip: z-slabs on several devices of this process
    }
    return create_common(params, 0, false, out);
}

extern "C++" {
int slab_context_create(const SphParams *params, uint32_t capacity, SphContext **out) { return create_common(params, capacity, true, out); }
void derive_params(const SphParams &p, SphDerived *d) { derive(p, d); }
int validate_params(SphContext *ctx, const SphParams *p) { return validate(ctx, p); }
}

int sph_destroy(SphContext *c)
{
    if (!c) return SPH_OK;
    if (c->is_group) return group_destroy(c);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    fold_events(c);
    for (hipEvent_t e : c->free_events) (void)hipEventDestroy(e);
    for (void *p : c->allocs) (void)hipFree(p);
    for (auto &g : c->graph)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (c->report_host) (void)hipHostFree((void *)c->report_host);
    if (c->mesh_va) {
        (void)hipMemUnmap(c->mesh_va, c->mesh_va_size);
        (void)hipMemAddressFree(c->mesh_va, c->mesh_va_size);
        (void)hipMemRelease(c->mesh_handle);
    }
    for (auto &e : c->state_export)
        if (e.va) {
            (void)hipMemUnmap(e.va, e.size);
            (void)hipMemAddressFree(e.va, e.size);
            (void)hipMemRelease(e.handle);
        }
    if (c->ext_sem) (void)hipDestroyExternalSemaphore(c->ext_sem);
    if (c->owns_group && c->group) slab_group_destroy(c->group);
    slab_rank_destroy(c);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SPH_OK;
}

int sph_upload(SphContext *c, const float *pos4, const float *vel4)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (c->is_group) return group_upload(c, pos4, vel4);
    if (c->slab) return fail(c, SPH_E_BAD_STATE, "slab context: use sph_slab_upload");
    if (!pos4) return fail(c, SPH_E_INVALID_ARG, "pos4 is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)c->dev.count * sizeof(float4);
    HIP_TRY(c, hipMemcpyAsync(c->pos4_dev, pos4, bytes, hipMemcpyHostToDevice, c->stream));
    if (vel4) HIP_TRY(c, hipMemcpyAsync(c->vel4_dev, vel4, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->dev.counters + kCntWildP, 0, sizeof(unsigned long long) * 3, c->stream));
    c->dev.wild_v_cur = 0;
    c->dev.force_slow = 0;
    launch_upload_split(c->dev, c->pos4_dev, vel4 ? c->vel4_dev : nullptr, c->pw0, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // the caller's arrays may be freed on return
    c->uploaded = true;
    c->grid_valid = false;
    c->mesh_valid = false;
    c->substeps_since_upload = 0;
    c->frames_since_upload = 0;
    reset_list_policy(c);
    return SPH_OK;
}

int sph_init_particles(SphContext *c, int preset)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (c->is_group) return fail(c, SPH_E_UNSUPPORTED, "sph_init_particles: not available on a multi-device context");
    if (c->slab) return fail(c, SPH_E_BAD_STATE, "slab context: initialise on the host and use sph_slab_upload");
    if (preset < 0 || preset > 2) return fail(c, SPH_E_INVALID_ARG, "preset %d not in [0,2] (SphFluidSimulation.cs:35)", preset);
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t res = c->derived.particle_resolution;
    const float fill = c->params.dam_fill_rate;
    // InitParticles.compute:20 / :53,:86: uint(ceil(pow(count / fill[^2], 1/3))), fp32
    const float denom = preset == 0 ? fill : fill * fill;
    const uint32_t ppd = (uint32_t)ceilf(powf((float)(res * res) / denom, 1.0f / 3.0f));
    if (ppd == 0) return fail(c, SPH_E_INVALID_ARG, "dam_fill_rate %g gives an empty lattice", (double)fill);
    launch_init_particles(preset, res, fill, ppd, c->pos4_dev, c->stream);
    HIP_TRY(c, hipMemsetAsync(c->dev.counters + kCntWildP, 0, sizeof(unsigned long long) * 3, c->stream));
    c->dev.wild_v_cur = 0;
    c->dev.force_slow = 0;
    launch_upload_split(c->dev, c->pos4_dev, nullptr, c->pw0, c->stream);   // velocities cleared, SphFluidSimulation.cs:189
    HIP_TRY(c, hipGetLastError());
    c->uploaded = true;
    c->grid_valid = false;
    c->mesh_valid = false;
    c->substeps_since_upload = 0;
    c->frames_since_upload = 0;
    reset_list_policy(c);
    return SPH_OK;
}

int sph_download(SphContext *c, float *pos4, float *vel4, float *rho)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (c->is_group) return group_download(c, pos4, vel4, rho);
    if (c->slab) return fail(c, SPH_E_BAD_STATE, "slab context: use sph_slab_download");
    if (!c->uploaded) return fail(c, SPH_E_BAD_STATE, "sph_download before sph_upload");
    HIP_TRY(c, hipSetDevice(c->device));
    launch_download_merge(c->dev, c->pw0, c->substeps_since_upload, c->dt_last, pos4 ? c->pos4_dev : nullptr,
                          vel4 ? c->vel4_dev : nullptr, rho ? c->rho_id_dev : nullptr, c->stream);
    HIP_TRY(c, hipGetLastError());
    const size_t n = c->dev.count;
    if (pos4) HIP_TRY(c, hipMemcpyAsync(pos4, c->pos4_dev, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (vel4) HIP_TRY(c, hipMemcpyAsync(vel4, c->vel4_dev, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (rho) HIP_TRY(c, hipMemcpyAsync(rho, c->rho_id_dev, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_update_mesh_properties(SphContext *c, const SphMeshParams *m)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (c->is_group) return fail(c, SPH_E_UNSUPPORTED, "sph_update_mesh_properties: not available on a multi-device context");
    if (!m || m->struct_size != sizeof(SphMeshParams))
        return fail(c, SPH_E_INVALID_ARG, "SphMeshParams NULL or struct_size != %zu", sizeof(SphMeshParams));
    if (c->slab) return fail(c, SPH_E_BAD_STATE, "slab context: render feed is built from sph_slab_download output");
    if (!c->uploaded) return fail(c, SPH_E_BAD_STATE, "sph_update_mesh_properties before sph_upload");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->mesh_props) {
        int rc = dalloc(c, &c->mesh_props, (size_t)c->dev.count * 5);
        if (rc != SPH_OK) return rc;
    }
    launch_mesh_properties(c->dev, m->low_speed, m->high_speed, m->particle_scale, m->sim_trs, c->mesh_props, c->stream);
    HIP_TRY(c, hipGetLastError());
    c->mesh_valid = true;
    return SPH_OK;
}

int sph_download_mesh_properties(SphContext *c, float *props20)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (!props20) return fail(c, SPH_E_INVALID_ARG, "props20 is NULL");
    if (!c->mesh_valid) return fail(c, SPH_E_BAD_STATE, "sph_download_mesh_properties before sph_update_mesh_properties");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(props20, c->mesh_props, (size_t)c->dev.count * 80, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_mesh_properties_device_ptr(SphContext *c, void **dptr, size_t *bytes)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (!dptr) return fail(c, SPH_E_INVALID_ARG, "dptr is NULL");
    if (!c->mesh_valid) return fail(c, SPH_E_BAD_STATE, "no mesh properties yet: call sph_update_mesh_properties first");
    *dptr = c->mesh_props;
    if (bytes) *bytes = (size_t)c->dev.count * 80;
    return SPH_OK;
}

int sph_export_mesh_properties_fd(SphContext *c, int *fd, size_t *bytes)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (c->is_group) return fail(c, SPH_E_UNSUPPORTED, "sph_export_mesh_properties_fd: not available on a multi-device context");
    if (!fd) return fail(c, SPH_E_INVALID_ARG, "fd is NULL");
    if (c->slab) return fail(c, SPH_E_BAD_STATE, "slab context: no render feed");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t used = (size_t)c->dev.count * 80;
    if (!c->mesh_va) {
        // move the render-feed buffer into a shareable physical allocation mapped at a reserved address
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = c->device;
        prop.requestedHandleTypes = hipMemHandleTypePosixFileDescriptor;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) {
            (void)hipGetLastError();
            return fail(c, SPH_E_UNSUPPORTED, "this device / driver has no exportable (virtual-memory) allocations");
        }
        const size_t size = (used + gran - 1) / gran * gran;
        hipMemGenericAllocationHandle_t h{};
        void *va = nullptr;
        if (hipMemCreate(&h, size, &prop, 0) != hipSuccess) {
            (void)hipGetLastError();
            return fail(c, SPH_E_UNSUPPORTED, "hipMemCreate of an exportable allocation failed");
        }
        hipMemAccessDesc acc{};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = c->device;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemAddressReserve(&va, size, 0, nullptr, 0) != hipSuccess || hipMemMap(va, size, 0, h, 0) != hipSuccess ||
            hipMemSetAccess(va, size, &acc, 1) != hipSuccess) {
            (void)hipGetLastError();
            if (va) (void)hipMemAddressFree(va, size);
            (void)hipMemRelease(h);
            return fail(c, SPH_E_HIP, "mapping the exportable allocation failed");
        }
        if (c->mesh_valid)   // keep what sph_update_mesh_properties already produced
            HIP_TRY(c, hipMemcpyAsync(va, c->mesh_props, used, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->mesh_handle = h;
        c->mesh_va = va;
        c->mesh_va_size = size;
        c->mesh_props = (float4 *)va;   // (a former hipMalloc'ed buffer stays in c->allocs until sph_destroy)
    }
    int out = -1;
    if (hipMemExportToShareableHandle(&out, c->mesh_handle, hipMemHandleTypePosixFileDescriptor, 0) != hipSuccess || out < 0) {
        (void)hipGetLastError();
        return fail(c, SPH_E_UNSUPPORTED, "hipMemExportToShareableHandle failed: no dma-buf export on this system");
    }
    *fd = out;
    if (bytes) *bytes = c->mesh_va_size;
    return SPH_OK;
}

// An exportable (dma-buf) device allocation of at least `bytes`, mapped read-write on the context's device.
static int make_exportable(SphContext *c, size_t bytes, hipMemGenericAllocationHandle_t *handle, void **va_out, size_t *size_out)
{
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    prop.requestedHandleTypes = hipMemHandleTypePosixFileDescriptor;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) {
        (void)hipGetLastError();
        return fail(c, SPH_E_UNSUPPORTED, "this device / driver has no exportable (virtual-memory) allocations");
    }
    const size_t size = (bytes + gran - 1) / gran * gran;
    hipMemGenericAllocationHandle_t h{};
    void *va = nullptr;
    if (hipMemCreate(&h, size, &prop, 0) != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, SPH_E_UNSUPPORTED, "hipMemCreate of an exportable allocation failed");
    }
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = c->device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemAddressReserve(&va, size, 0, nullptr, 0) != hipSuccess || hipMemMap(va, size, 0, h, 0) != hipSuccess ||
        hipMemSetAccess(va, size, &acc, 1) != hipSuccess) {
        (void)hipGetLastError();
        if (va) (void)hipMemAddressFree(va, size);
        (void)hipMemRelease(h);
        return fail(c, SPH_E_HIP, "mapping the exportable allocation failed");
    }
    *handle = h;
    *va_out = va;
    *size_out = size;
    return SPH_OK;
}

int sph_update_state_textures(SphContext *c)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (c->is_group || c->slab) return fail(c, SPH_E_UNSUPPORTED, "sph_update_state_textures: single-device contexts only");
    if (!c->uploaded) return fail(c, SPH_E_BAD_STATE, "sph_update_state_textures before sph_upload");
    HIP_TRY(c, hipSetDevice(c->device));
    launch_download_merge(c->dev, c->pw0, c->substeps_since_upload, c->dt_last, c->pos4_dev, c->vel4_dev, c->rho_id_dev, c->stream);
    HIP_TRY(c, hipGetLastError());
    return SPH_OK;
}

int sph_export_state_fd(SphContext *c, int which, int *fd, size_t *bytes)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (!fd || which < 0 || which > 2) return fail(c, SPH_E_INVALID_ARG, "fd is NULL or texture %d not in [0,2]", which);
    if (c->is_group || c->slab) return fail(c, SPH_E_UNSUPPORTED, "sph_export_state_fd: single-device contexts only");
    HIP_TRY(c, hipSetDevice(c->device));
    SphContext::ExportBuf &e = c->state_export[which];
    const size_t used = (size_t)c->dev.count * (which == SPH_STATE_DENSITY ? sizeof(float) : sizeof(float4));
    if (!e.va) {
        int rc = make_exportable(c, used, &e.handle, &e.va, &e.size);
        if (rc != SPH_OK) return rc;
        // the staging texture moves there (the hipMalloc'ed one stays in c->allocs until sph_destroy)
        void *old = which == SPH_STATE_POSITION ? (void *)c->pos4_dev : which == SPH_STATE_VELOCITY ? (void *)c->vel4_dev : (void *)c->rho_id_dev;
        HIP_TRY(c, hipMemcpyAsync(e.va, old, used, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (which == SPH_STATE_POSITION) c->pos4_dev = (float4 *)e.va;
        else if (which == SPH_STATE_VELOCITY) c->vel4_dev = (float4 *)e.va;
        else c->rho_id_dev = (float *)e.va;
    }
    int out = -1;
    if (hipMemExportToShareableHandle(&out, e.handle, hipMemHandleTypePosixFileDescriptor, 0) != hipSuccess || out < 0) {
        (void)hipGetLastError();
        return fail(c, SPH_E_UNSUPPORTED, "hipMemExportToShareableHandle failed: no dma-buf export on this system");
    }
    *fd = out;
    if (bytes) *bytes = e.size;
    return SPH_OK;
}

int sph_import_semaphore_fd(SphContext *c, int fd, int timeline)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (fd < 0) return fail(c, SPH_E_INVALID_ARG, "fd %d", fd);
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->ext_sem) {
        (void)hipDestroyExternalSemaphore(c->ext_sem);
        c->ext_sem = nullptr;
    }
    hipExternalSemaphoreHandleDesc desc{};
    desc.type = timeline ? hipExternalSemaphoreHandleTypeTimelineSemaphoreFd : hipExternalSemaphoreHandleTypeOpaqueFd;
    desc.handle.fd = fd;
    hipExternalSemaphore_t sem = nullptr;
    if (hipImportExternalSemaphore(&sem, &desc) != hipSuccess || !sem) {
        (void)hipGetLastError();
        return fail(c, SPH_E_UNSUPPORTED, "hipImportExternalSemaphore failed: not a %s semaphore fd of this GPU", timeline ? "timeline" : "binary");
    }
    c->ext_sem = sem;
    c->ext_sem_timeline = timeline != 0;
    return SPH_OK;
}

int sph_signal_semaphore(SphContext *c, uint64_t value)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (!c->ext_sem) return fail(c, SPH_E_BAD_STATE, "sph_signal_semaphore before sph_import_semaphore_fd");
    HIP_TRY(c, hipSetDevice(c->device));
    hipExternalSemaphoreSignalParams p{};
    p.params.fence.value = value;
    HIP_TRY(c, hipSignalExternalSemaphoresAsync(&c->ext_sem, &p, 1, c->stream));
    return SPH_OK;
}

int sph_bucket_generation(SphContext *c)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (c->is_group) return fail(c, SPH_E_UNSUPPORTED, "sph_bucket_generation: not available on a multi-device context");
    HIP_TRY(c, hipSetDevice(c->device));
    return do_bucket_generation(c);
}

int sph_density_calculation(SphContext *c)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (c->is_group) return fail(c, SPH_E_UNSUPPORTED, "sph_density_calculation: not available on a multi-device context");
    HIP_TRY(c, hipSetDevice(c->device));
    return do_density(c);
}

int sph_update_velocity_and_position(SphContext *c, float dt)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (c->is_group) return fail(c, SPH_E_UNSUPPORTED, "sph_update_velocity_and_position: not available on a multi-device context");
    HIP_TRY(c, hipSetDevice(c->device));
    c->dev.dt_ptr = nullptr;          // dt travels by value here (a replayed frame graph reads it from device memory)
    return do_velpos(c, dt);
}

// The launches of one Update() (SphFluidSimulation.cs:96-102), issued on the context's stream.
static int issue_frame(SphContext *c, float dt)
{
    int rc;
    const bool stamped = c->report_dev && c->nl_enabled && !c->slab;
    if ((rc = do_bucket_generation(c)) != SPH_OK) return rc;            // :98
    if ((rc = do_density(c)) != SPH_OK) return rc;                      // :99
    for (uint32_t i = 0; i < c->params.substeps; i++)                   // :101
        if ((rc = do_velpos(c, dt)) != SPH_OK) return rc;
    if (stamped) launch_frame_report(c->dev, c->report_dev, c->nl_recorded, c->stream);
    return SPH_OK;
}

static bool frame_reports(const SphContext *c) { return c->report_dev && c->nl_enabled && !c->slab; }

// Where the scanning kernel of a list-using sub-step runs (launch_velpos): beside the list kernels on the side stream while the
// last reported frame had many bricks that were not calm (late in a dam break: thousands per sub-step, 2.3 ms of a 42 ms frame
// that overlap with the list kernels -> 40.4 ms), in stream order while it had a handful (the calm frames of the bench: no gain,
// and a kernel profile in which the two kernels' durations mean what they say).
static void update_scan_policy(SphContext *c)
{
    if (!c->report_host || !c->side_stream) return;
    const unsigned long long frames = c->report_host[0], bricks = c->report_host[4];
    if (frames == c->scan_seen[0]) return;
    if (frames > c->scan_seen[0] && bricks >= c->scan_seen[1]) {
        const double per_substep = (double)(bricks - c->scan_seen[1]) / ((double)(frames - c->scan_seen[0]) * (double)c->params.substeps);
#ifndef SPH_SCAN_FORK_MIN
#define SPH_SCAN_FORK_MIN 1000.0   /* bricks per sub-step that are not calm, above which the scanning kernel gets the side stream */
#endif
        c->fork_scan = per_substep > SPH_SCAN_FORK_MIN;
    }
    c->scan_seen[0] = frames;
    c->scan_seen[1] = bricks;
}

// Decides, before a frame is issued, whether it records neighbour lists (see SphContext::pol_*).
static void update_list_policy(SphContext *c)
{
    if (!c->report_host || !c->nl_enabled || c->slab || (c->params.flags & SPH_FLAG_LISTS_ALWAYS)) return;
    if (c->pol_on && !c->pol_confirmed && c->frames_since_upload > 3 && c->pol_issued > c->report_host[0] + 2)
        (void)hipStreamSynchronize(c->stream);   // only while a (re)start of the recording is being judged
    const unsigned long long frames = c->report_host[0];
    if (frames != c->pol_seen[0]) {
        const unsigned long long listed = c->report_host[1], list_frames = c->report_host[2];
        const unsigned long long dlf = list_frames - c->pol_seen[2];
        if (dlf > 0 && listed >= c->pol_seen[1] && c->frames_since_upload > 3) {
            const double frac = (double)(listed - c->pol_seen[1]) / ((double)dlf * (double)c->dev.count * (double)c->params.substeps);
            if (c->pol_on) {
                // Break-even: recording + keeping the lists valid costs ~2 ms per frame at 16.7 M and a served target sub-step
                // saves ~0.11 ns against the scanning brick kernel -> ~0.25 of the target sub-steps at 16.7 M, more in small
                // boxes whose kernels are launch-bound (R < 100: the old 0.9).  Late in a dam break the crowded wall cells are
                // scanned anyway while the calm bulk (60-90 %) is still served three times faster (profiles/r02_long_run_16M.txt).
                if (frac < c->pol_min_frac) {
                    c->pol_high_streak = 0;
                    if (++c->pol_low_streak >= 2) {
                        c->pol_on = false;
                        c->pol_low_streak = 0;
                        c->pol_resume_at = c->frames + c->pol_backoff;
                        c->pol_backoff = c->pol_backoff < 128 ? c->pol_backoff * 2 : 128;
                    }
                } else {
                    c->pol_low_streak = 0;
                    if (++c->pol_high_streak >= 2) {
                        c->pol_confirmed = true;
                        c->pol_backoff = 8;
                    }
                }
            }
        }
        c->pol_seen[0] = frames;
        c->pol_seen[1] = listed;
        c->pol_seen[2] = list_frames;
    }
    if (!c->pol_on && c->frames >= c->pol_resume_at) {   // probe: two low reports switch it off again
        c->pol_on = true;
        c->pol_confirmed = false;
        c->pol_low_streak = c->pol_high_streak = 0;
    }
    if (!c->pol_on) c->pol_frames_off++;
}

// Host-side state a replayed frame leaves behind (what issue_frame's passes would have set).
static void account_replayed_frame(SphContext *c, const SphContext::FrameGraph &g, float dt)
{
    const uint32_t S = c->params.substeps;
    std::swap(c->dev.id, c->dev.id_nxt);                                // the grid build's swaps; cur/nxt: 1 + S swaps
    std::swap(c->dev.cellkey, c->dev.cellkey_nxt);
    if ((1u + S) & 1u) std::swap(c->dev.cur, c->dev.nxt);
    if (S & 1u) c->dev.wild_v_cur ^= 1u;
    c->dev.force_slow = 0;
    c->grid_valid = true;
    c->nl_fresh = false;
    c->nl_recorded = g.recorded_lists;
    c->frames++;
    c->substeps += S;
    c->substeps_since_upload += (int)S;
    c->substep_in_frame = (int)S;
    c->dt_last = dt;
}

int sph_step(SphContext *c, float delta_time)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (c->is_group) return group_step(c, delta_time);
    if (c->slab) return fail(c, SPH_E_BAD_STATE, "slab context: the host drives the passes and the halo exchange");
    HIP_TRY(c, hipSetDevice(c->device));
    const float dt = delta_time / c->params.dt_divisor;                 // :102 Time.deltaTime / 25
    update_list_policy(c);
    update_scan_policy(c);
    if (c->pol_on) ensure_lists(c);   // (never inside a stream capture)
    grow_list_rows(c);
    if (frame_reports(c)) c->pol_issued++;
    // the first two frames after an upload run ungraphed: they allocate (neighbour lists) and settle the pointer parity
    // ... and a stream the caller handed over (sph_set_stream) is captured like the context's own unless it is the legacy default
    // stream (cannot be captured) or the caller is capturing it already (the launches then simply join the caller's graph)
    bool stream_ok = c->own_stream;
    if (!stream_ok && c->stream != nullptr) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        stream_ok = hipStreamIsCapturing(c->stream, &st) == hipSuccess && st == hipStreamCaptureStatusNone;
    }
    const bool graphed = c->graphs_enabled && stream_ok && c->uploaded && c->frames_since_upload >= 2 && c->params.substeps > 0;
    c->frames_since_upload++;
    if (!graphed) {
        c->dev.dt_ptr = nullptr;
        return issue_frame(c, dt);
    }
    SphContext::FrameGraph &g = c->graph[c->frames_since_upload & 1u];
    launch_set_scalar(c->dt_dev, dt, c->stream);
    c->dev.dt_ptr = c->dt_dev;
    c->dev.dt = 0.0f;
    Dev start;                        // (memcpy: the comparison below is bytewise, padding included)
    memcpy(&start, &c->dev, sizeof(Dev));
    start.nl_mode = 0;                // per-launch scratch
    start.nl_still = 0;
    start.box_par = 0;
    const bool lists_on = c->nl_enabled && c->nl_allocated && c->pol_on;
    if (!g.exec || g.substeps != c->params.substeps || g.lists_on != lists_on || g.fork_scan != c->fork_scan ||
        memcmp(&g.start, &start, sizeof(Dev)) != 0) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        g.exec = nullptr;
        hipGraph_t graph = nullptr;
        Dev dev_before;
        memcpy(&dev_before, &c->dev, sizeof(Dev));
        const bool grid_valid = c->grid_valid, fresh = c->nl_fresh, recorded = c->nl_recorded;
        const uint64_t frames = c->frames, substeps = c->substeps;
        const int since = c->substeps_since_upload, in_frame = c->substep_in_frame;
        bool ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        int rc = ok ? issue_frame(c, dt) : SPH_OK;
        if (ok) ok = hipStreamEndCapture(c->stream, &graph) == hipSuccess && graph && rc == SPH_OK;
        if (ok) ok = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) (void)hipGraphDestroy(graph);
        g.recorded_lists = c->nl_recorded;
        // nothing has run yet: roll the host-side bookkeeping back to the frame start, the replay below redoes it
        memcpy(&c->dev, &dev_before, sizeof(Dev));
        c->grid_valid = grid_valid;
        c->nl_fresh = fresh;
        c->nl_recorded = recorded;
        c->frames = frames;
        c->substeps = substeps;
        c->substeps_since_upload = since;
        c->substep_in_frame = in_frame;
        if (!ok) {                    // no graphs on this system / for this launch sequence: keep issuing the launches
            (void)hipGetLastError();
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            g.exec = nullptr;
            c->graphs_enabled = false;
            c->dev.dt_ptr = nullptr;
            return issue_frame(c, dt);
        }
        memcpy(&g.start, &start, sizeof(Dev));
        g.substeps = c->params.substeps;
        g.lists_on = lists_on;
        g.fork_scan = c->fork_scan;
    }
    HIP_TRY(c, hipGraphLaunch(g.exec, c->stream));
    c->graph_replays++;
    account_replayed_frame(c, g, dt);
    return SPH_OK;
}

int sph_step_n(SphContext *c, float delta_time, int frames)
{
    if (frames < 0) return fail(c, SPH_E_INVALID_ARG, "frames < 0");
    for (int f = 0; f < frames; f++) {
        int rc = sph_step(c, delta_time);
        if (rc != SPH_OK) return rc;
    }
    return SPH_OK;
}

int sph_set_params(SphContext *c, const SphParams *p)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    int rc = validate(c, p);
    if (rc != SPH_OK) return rc;
    if (next_power_of_two(p->num_particles) != c->derived.num_particles || p->bucket_resolution != c->params.bucket_resolution ||
        p->dam_fill_rate != c->params.dam_fill_rate || p->device_id != c->params.device_id || p->precision != c->params.precision)
        return fail(c, SPH_E_INVALID_ARG,
                    "num_particles / bucket_resolution / dam_fill_rate / device_id / precision are fixed at sph_create");
    if (p->gas_constant != c->params.gas_constant || p->rest_density != c->params.rest_density)
        c->dev.force_slow = 1;  // pressures change without a density pass: flag is stale until the next one
    if (p->num_devices != c->params.num_devices || p->halo_planes != c->params.halo_planes)
        return fail(c, SPH_E_INVALID_ARG, "num_devices / halo_planes are fixed at sph_create");
    if (c->is_group) return group_set_params(c, p);
    // lists recorded by the density pass of the frame in flight (staged calls) were recorded FOR a format, a kernel set and a
    // skin: a change of any of them retires them (the rest of the frame scans; the next density pass records again)
    // (the instantiation of k_velpos_tile goes with the decomposition the recording pass chose; the displacement ranges of the
    // sub-steps so far are per brick or per chunk)
    constexpr uint32_t kListFlags = SPH_FLAG_GLOBAL_LISTS | SPH_FLAG_NO_TILING | SPH_FLAG_NO_LISTS | SPH_FLAG_TILE_SLOTS_ONE |
                                    SPH_FLAG_TILE_SLOTS_TWO | SPH_FLAG_CHUNK_BOXES;
    if (((p->flags ^ c->params.flags) & kListFlags) != 0u || p->list_skin != c->params.list_skin) c->nl_recorded = false;
    c->params = *p;
    c->profile = (p->flags & SPH_FLAG_PROFILE) != 0;
    // per-pass events cannot be recorded inside a captured frame: a context that profiles issues its launches directly
    c->graphs_enabled = !c->slab && !c->profile && !(p->flags & SPH_FLAG_NO_GRAPHS);
    refresh_scalars(c);
    return SPH_OK;
}

int sph_get_derived(SphContext *c, SphDerived *out)
{
    if (!c || !out) return fail(c, SPH_E_INVALID_ARG, "NULL argument");
    *out = c->derived;
    return SPH_OK;
}

int sph_synchronize(SphContext *c)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (c->is_group) return group_synchronize(c);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_get_stats(SphContext *c, SphStats *out)
{
    if (!c || !out) return fail(c, SPH_E_INVALID_ARG, "NULL argument");
    if (c->is_group) return group_get_stats(c, out);
    if (c->slab && c->group) {
        int rc = slab_group_finish(c->group);            // (collective in z-slab mode, see include/sph_hip.h)
        if (rc != SPH_OK) return rc;
    }
    return engine_get_stats(c, out);
}
}  // extern "C"

int engine_get_stats(SphContext *c, SphStats *out)
{
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    fold_events(c);
    unsigned long long cnt[kCntCount];
    HIP_TRY(c, hipMemcpy(cnt, c->dev.counters, sizeof cnt, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> stripes((size_t)kStatCount * kStatStripes * kStatPad);
    HIP_TRY(c, hipMemcpy(stripes.data(), c->dev.stripes, stripes.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long stat[kStatCount] = {};
    for (int k = 0; k < kStatCount; k++)
        for (int i = 0; i < kStatStripes; i++) stat[k] += stripes[((size_t)k * kStatStripes + i) * kStatPad];
    memset(out, 0, sizeof *out);
    out->frames = c->frames;
    out->substeps = c->substeps;
    out->ms_grid = c->ms[0];
    out->ms_density = c->ms[1];
    out->ms_velpos = c->ms[2];
    out->launches_grid = c->launches[0];
    out->launches_density = c->launches[1];
    out->launches_velpos = c->launches[2];
    out->graph_replays = c->graph_replays;
    out->frames_without_lists = c->pol_frames_off;
    out->ms_velpos_list = c->ms[3];
    out->launches_velpos_list = c->launches[3];
    out->max_cell_occupancy = (uint32_t)cnt[kCntMaxOcc];
    out->overflowed_cells = (uint32_t)cnt[kCntOverflowCells];
    out->dropped_particles = (uint32_t)cnt[kCntDropped];
    out->queued_targets = (uint32_t)cnt[kCntQueue];
    out->halo_violations = cnt[kCntHaloViolation];
    out->nan_resets = cnt[kCntNaN];
    out->candidates_density = stat[kStatCandidates];
    out->listed_target_substeps = stat[kStatListed];
    out->scanned_brick_substeps = cnt[kCntActiveBricks];
    out->list_entries = stat[kStatListEntries];
    out->list_row_entries = c->nl_allocated ? c->dev.nl_cap : 0;
    out->lists_too_long = cnt[kCntListTooLong];
    out->host_syncs = c->host_syncs;
    out->frames_rerun = c->group ? slab_group_reruns(c->group) : 0;
    out->halo_planes = c->slab ? (uint64_t)c->halo_w : 0;
    return SPH_OK;
}

extern "C" {

int sph_reset_stats(SphContext *c)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (c->is_group) return group_reset_stats(c);
    c->host_syncs = 0;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    fold_events(c);
    for (int i = 0; i < 4; i++) {
        c->ms[i] = 0;
        c->launches[i] = 0;
    }
    c->pol_seen[1] = 0;               // (the cumulative count the policy differences restarts)
    HIP_TRY(c, hipMemsetAsync(c->dev.counters + kCntActiveBricks, 0, sizeof(unsigned long long), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->dev.stripes + (size_t)kStatListed * kStatStripes * kStatPad, 0,
                              2 * sizeof(unsigned long long) * kStatStripes * kStatPad, c->stream));
    return SPH_OK;
}

int sph_set_stream(SphContext *c, void *hip_stream)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (c->is_group) return fail(c, SPH_E_UNSUPPORTED, "sph_set_stream: not available on a multi-device context");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    fold_events(c);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    return SPH_OK;
}

const char *sph_last_error(SphContext *c)
{
    if (c) return c->last_error.c_str();
    return g_last_error.c_str();
}


#ifdef SPH_STAMPS   /* diagnostic builds only: raw read of the sort scratch (holds the phase stamps) */
int sph_debug_read_order(SphContext *c, void *dst, size_t bytes)
{
    if (!c || !dst) return SPH_E_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(dst, c->dev.order, bytes, hipMemcpyDeviceToHost));
    return SPH_OK;
}
#endif

#ifdef SPH_DEBUG_API   /* diagnostic builds only (scripts/list_balance.py): raw read of the neighbour-list counts */
int sph_debug_read_list_counts(SphContext *c, void *dst, size_t bytes)
{
    if (!c || !dst || !c->dev.nl_cnt) return SPH_E_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(dst, c->dev.nl_cnt, bytes, hipMemcpyDeviceToHost));
    return SPH_OK;
}
#endif

}  // extern "C"
