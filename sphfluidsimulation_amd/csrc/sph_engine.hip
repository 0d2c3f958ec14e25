// Host side of libsph_hip.so: context, device memory, the frame schedule of
// SphFluidSimulation.cs::Update (:96-102) and the C ABI declared in include/sph_hip.h.
#include "../../include/sph_hip.h"
#include "sph_internal.hpp"

#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <cstdlib>
#include <string>
#include <vector>

using namespace sph;

static thread_local std::string g_last_error = "";

struct PassEvents {
    hipEvent_t a, b;
    int kind;  // 0 grid, 1 density, 2 velpos (whole pass), 3 k_velpos_list alone
};

struct SphContext {
    SphParams params{};
    SphDerived derived{};
    Dev dev{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::vector<void *> allocs;
    // staging in the reference's texture layout
    float4 *pos4_dev = nullptr, *vel4_dev = nullptr;
    float *rho_id_dev = nullptr, *pw0 = nullptr;
    // One frame (grid build, density pass, S sub-steps: ~60 launches) captured as a HIP graph and replayed: for scenes of the
    // reference's own size (262 144 particles, ~2 ms per frame) the launches cost as much as a third of the frame.
    // The device pointers a frame starts from alternate with period 2 (id / id_nxt and the wild-velocity slot flip once per
    // frame), so there are two graphs; each remembers the launch state (Dev) it was captured for and is re-captured when
    // that no longer matches (sph_set_params, another sub-step count).  delta_time is read from device memory.
    struct FrameGraph {
        hipGraphExec_t exec = nullptr;
        Dev start{};                  // launch state at frame start
        uint32_t substeps = 0;
        bool recorded_lists = false;  // the captured density pass recorded neighbour lists
        bool lists_on = false;        // neighbour lists were enabled and allocated at capture time
    } graph[2];
    bool graphs_enabled = false;
    float *dt_dev = nullptr;          // delta_time / dt_divisor of the frame being replayed (device scalar)
    uint64_t frames_since_upload = 0, graph_replays = 0;
    // frame-coherent neighbour lists (sph_kernels_list.hip)
    bool nl_enabled = false, nl_allocated = false;
    bool nl_fresh = false;            // the grid was just built: the next density pass runs on the binned positions
    bool nl_recorded = false;         // ... and recorded the lists the VelPos sub-steps of this frame may use
    // List policy.  Recording lists and keeping them valid costs ~25 % of a scanning frame at 16.7 M -- where 99 % of the
    // target sub-steps are then served three times faster -- but makes a frame of the reference's own 262 144-particle
    // scene 1.5x slower (a box that small is all wall layer, 28 % served; its kernels are latency-bound, every extra
    // launch counts).  k_frame_report publishes, frame by frame, how many target sub-steps the lists served; below 90 %
    // (two reports in a row) the engine stops recording and probes again after a back-off that doubles with every
    // failed probe.
    volatile unsigned long long *report_host = nullptr;   // mapped pinned memory, written by the device
    unsigned long long *report_dev = nullptr;
    unsigned long long pol_seen[3] = {0, 0, 0};
    bool pol_on = true;               // this frame records lists
    bool pol_confirmed = false;       // two good reports since recording was (re)started: until then the host does not run
                                      // more than two frames ahead of the reports (a caller queueing 20 frames would
                                      // otherwise pay for 20 frames of a probe that fails)
    unsigned long long pol_issued = 0;   // frames issued with a report kernel
    int pol_low_streak = 0, pol_high_streak = 0;
    uint64_t pol_backoff = 8, pol_resume_at = 0, pol_frames_off = 0;
    float nl_skin_frac = 0.04f;       // skin / h (profiles/: 0.04-0.06 is the optimum of the 16.7 M dam break)
    float4 *mesh_props = nullptr;     // _particleMeshPropertiesBuffer (:174), allocated on first use
    bool mesh_valid = false;
    // row f4: the same buffer in exportable device memory (HIP virtual-memory allocation), once a caller asked for its fd
    hipMemGenericAllocationHandle_t mesh_handle{};
    void *mesh_va = nullptr;
    size_t mesh_va_size = 0;
    bool uploaded = false, grid_valid = false;
    int substeps_since_upload = 0;
    float dt_last = 0.0f;
    uint64_t frames = 0, substeps = 0;
    // profiling
    bool profile = false;
    std::vector<PassEvents> pending;
    std::vector<hipEvent_t> free_events;
    double ms[4] = {0, 0, 0, 0};
    uint64_t launches[4] = {0, 0, 0, 0};
    std::string last_error;
    // z-slab (multi-GPU) mode
    bool slab = false;
    uint32_t capacity = 0;            // particles the device arrays can hold
    int z_lo = 0, z_hi = 0, halo_w = 2, has_left = 0, has_right = 0;
    uint32_t n_ghost[2] = {0, 0};     // ghosts appended this frame (from left, from right)
    uint32_t n_export[2] = {0, 0};    // owned particles exported as ghosts this frame (to left, to right)
    uint32_t *send_src[2] = {nullptr, nullptr};
    uint32_t *slab_counters = nullptr;  // 4 words, see k_slab_classify
    uint32_t *ids_dev = nullptr;
};

// A new particle set: the list policy starts over.  Prior: in a box of fewer than ~100 cells per axis most bricks lie
// within reach of the wall layer, whose particles the wall force scatters by more than any useful skin -- such scenes
// start without lists and probe late; large boxes start with lists.
static void reset_list_policy(SphContext *c)
{
    const bool big = c->dev.R >= 100u;
    c->pol_on = big || (c->params.flags & SPH_FLAG_LISTS_ALWAYS);
    c->pol_confirmed = false;
    c->pol_low_streak = c->pol_high_streak = 0;
    c->pol_backoff = big ? 8 : 32;
    c->pol_resume_at = c->frames + (c->pol_on ? 0 : c->pol_backoff);
}

static int fail(SphContext *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->last_error = buf;
    g_last_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return fail(ctx, _e == hipErrorOutOfMemory ? SPH_E_OOM : SPH_E_HIP, "%s: %s", #expr, \
                        hipGetErrorString(_e));                                                 \
    } while (0)

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

static void refresh_scalars(SphContext *c)
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
    // neighbour lists: skin = nl_skin_frac * h; pre-filter radius (h + skin)^2 rounded up like h2_up; a brick is calm
    // while its relative-displacement bound stays below 90 % of the skin (the rest absorbs fp32 rounding of the bound)
    c->nl_enabled = d.use_bricks && !(p.flags & SPH_FLAG_NO_LISTS) && d.R <= 256u;
    if (const char *e = getenv("SPH_NL_SKIN")) {   // tuning aid: skin / h
        const float v = (float)atof(e);
        if (v > 0.0f && v <= 1.0f) c->nl_skin_frac = v;
    }
    {
        const double hs = (double)d.h * (1.0 + (double)c->nl_skin_frac);
        d.h2_list = nextafterf((float)(hs * hs * (1.0 + 1e-5)), INFINITY);
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
    if (p->precision != 0) return fail(ctx, SPH_E_UNSUPPORTED, "precision %u not supported (fp32 only)", p->precision);
    if (p->exact_order != 1) return fail(ctx, SPH_E_UNSUPPORTED, "exact_order=0 (fast mode) is not built in this version");
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
    float **f[] = {&s->x, &s->y, &s->z, &s->vw};
    for (auto pp : f)
        if ((rc = dalloc(c, pp, n)) != SPH_OK) return rc;
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
        return;
    }
    const size_t bricks = (size_t)nl_bricks_x(d.R) * nl_bricks_y(d.R) * nl_bricks_z(d.R);
    d.nl_cap = 64;
    d.n_bricks = (uint32_t)bricks;
    d.n_chunks = (uint32_t)((n + kChunkSlots - 1) / kChunkSlots);
    const size_t before = c->allocs.size();
    const bool ok = dalloc(c, &d.p1x, n) == SPH_OK && dalloc(c, &d.p1y, n) == SPH_OK && dalloc(c, &d.p1z, n) == SPH_OK &&
                    dalloc(c, &d.nl_ent, (size_t)d.nl_cap * ((n + 63) & ~(size_t)63)) == SPH_OK && dalloc(c, &d.nl_cnt, n) == SPH_OK &&
                    dalloc(c, &d.chunk_box, 6 * ((n + kChunkSlots - 1) / kChunkSlots)) == SPH_OK &&
                    dalloc(c, &d.active_list, bricks) == SPH_OK && dalloc(c, &d.queue2, n) == SPH_OK && dalloc(c, &d.brick_state, bricks) == SPH_OK &&
                    dalloc(c, &d.cur.p4, n) == SPH_OK && dalloc(c, &d.nxt.p4, n) == SPH_OK;
    if (!ok) {
        while (c->allocs.size() > before) {
            (void)hipFree(c->allocs.back());
            c->allocs.pop_back();
        }
        (void)hipGetLastError();
        d.p1x = d.p1y = d.p1z = nullptr;
        d.nl_ent = nullptr;
        d.nl_cnt = nullptr;
        d.chunk_box = nullptr;
        d.active_list = nullptr;
        d.queue2 = nullptr;
        d.brick_state = nullptr;
        d.cur.p4 = d.nxt.p4 = nullptr;
        c->nl_enabled = false;
        c->last_error.clear();
        return;
    }
    c->nl_allocated = true;
}

static int do_bucket_generation(SphContext *c)
{
    if (!c->uploaded) return fail(c, SPH_E_BAD_STATE, "BucketGeneration before sph_upload");
    if (c->pol_on) ensure_lists(c);   // (several GB at 16.7 M: only once a frame is going to record)
    c->dev.use_lists = c->nl_enabled && c->nl_allocated && c->pol_on ? 1u : 0u;
    c->nl_fresh = c->dev.use_lists != 0u;
    c->nl_recorded = false;
    {
        PassTimer t(c, 0);
        launch_grid_build(c->dev, c->stream);
    }
    HIP_TRY(c, hipGetLastError());
    std::swap(c->dev.cur, c->dev.nxt);
    std::swap(c->dev.id, c->dev.id_nxt);
    std::swap(c->dev.ghost, c->dev.ghost_nxt);
    c->grid_valid = true;
    c->frames++;
    return SPH_OK;
}

static int do_density(SphContext *c)
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

static int do_velpos(SphContext *c, float dt)
{
    if (!c->grid_valid) return fail(c, SPH_E_BAD_STATE, "UpdateVelocityAndPosition before BucketGeneration");
    c->dev.dt = dt;
    c->dev.nl_mode = c->dev.use_lists && c->nl_recorded ? kNlUse : kNlOff;
    c->dev.nl_still = c->nl_fresh ? 1u : 0u;   // first sub-step since the grid build: the recorded positions still stand
    c->nl_fresh = false;              // positions move: a later density pass of this frame must not record
    {
        PassTimer t(c, 2);
        if (c->profile && c->dev.nl_mode == kNlUse) {   // the dominant kernel on its own, for the roofline figure
            PassEvents pe{};
            pe.kind = 3;
            pe.a = get_event(c);
            pe.b = get_event(c);
            launch_velpos(c->dev, c->stream, pe.a, pe.b);
            c->pending.push_back(pe);
        } else {
            launch_velpos(c->dev, c->stream);
        }
    }
    HIP_TRY(c, hipGetLastError());
    std::swap(c->dev.cur, c->dev.nxt);  // Swap(), SphFluidSimulation.cs:267-268
    c->dev.wild_v_cur ^= 1u;
    c->substeps++;
    c->substeps_since_upload++;
    c->dt_last = dt;
    return SPH_OK;
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
        if (hipHostMalloc(&h, 4 * sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer(&dv, h, 0) == hipSuccess) {
            memset(h, 0, 4 * sizeof(unsigned long long));
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
        A(dalloc(c, &c->send_src[0], n));
        A(dalloc(c, &c->send_src[1], n));
        A(dalloc(c, &c->slab_counters, (size_t)4));
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

int sph_create(const SphParams *params, SphContext **out) { return create_common(params, 0, false, out); }

int sph_slab_create(const SphParams *params, uint32_t local_capacity, SphContext **out)
{
    if (local_capacity == 0) return fail(nullptr, SPH_E_INVALID_ARG, "local_capacity is 0");
    return create_common(params, local_capacity, true, out);
}

int sph_destroy(SphContext *c)
{
    if (!c) return SPH_OK;
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
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SPH_OK;
}

int sph_upload(SphContext *c, const float *pos4, const float *vel4)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
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

int sph_bucket_generation(SphContext *c)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    return do_bucket_generation(c);
}

int sph_density_calculation(SphContext *c)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    return do_density(c);
}

int sph_update_velocity_and_position(SphContext *c, float dt)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
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
                // (the first sub-step of a frame is always served: 0.9 of all means 0.875 of the sub-steps that can fail)
                if (frac < 0.9) {
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
    if ((1u + S) & 1u) std::swap(c->dev.cur, c->dev.nxt);
    if (S & 1u) c->dev.wild_v_cur ^= 1u;
    c->dev.force_slow = 0;
    c->grid_valid = true;
    c->nl_fresh = false;
    c->nl_recorded = g.recorded_lists;
    c->frames++;
    c->substeps += S;
    c->substeps_since_upload += (int)S;
    c->dt_last = dt;
}

int sph_step(SphContext *c, float delta_time)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
    if (c->slab) return fail(c, SPH_E_BAD_STATE, "slab context: the host drives the passes and the halo exchange");
    HIP_TRY(c, hipSetDevice(c->device));
    const float dt = delta_time / c->params.dt_divisor;                 // :102 Time.deltaTime / 25
    update_list_policy(c);
    if (c->pol_on) ensure_lists(c);   // (never inside a stream capture)
    if (frame_reports(c)) c->pol_issued++;
    // the first two frames after an upload run ungraphed: they allocate (neighbour lists) and settle the pointer parity
    const bool graphed = c->graphs_enabled && c->own_stream && c->uploaded && c->frames_since_upload >= 2 && c->params.substeps > 0;
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
    const bool lists_on = c->nl_enabled && c->nl_allocated && c->pol_on;
    if (!g.exec || g.substeps != c->params.substeps || g.lists_on != lists_on || memcmp(&g.start, &start, sizeof(Dev)) != 0) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        g.exec = nullptr;
        hipGraph_t graph = nullptr;
        Dev dev_before;
        memcpy(&dev_before, &c->dev, sizeof(Dev));
        const bool grid_valid = c->grid_valid, fresh = c->nl_fresh, recorded = c->nl_recorded;
        const uint64_t frames = c->frames, substeps = c->substeps;
        const int since = c->substeps_since_upload;
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
        p->dam_fill_rate != c->params.dam_fill_rate || p->device_id != c->params.device_id)
        return fail(c, SPH_E_INVALID_ARG,
                    "num_particles / bucket_resolution / dam_fill_rate / device_id are fixed at sph_create");
    if (p->gas_constant != c->params.gas_constant || p->rest_density != c->params.rest_density)
        c->dev.force_slow = 1;  // pressures change without a density pass: flag is stale until the next one
    c->params = *p;
    c->profile = (p->flags & SPH_FLAG_PROFILE) != 0;
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
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_get_stats(SphContext *c, SphStats *out)
{
    if (!c || !out) return fail(c, SPH_E_INVALID_ARG, "NULL argument");
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
    return SPH_OK;
}

int sph_reset_stats(SphContext *c)
{
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");
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


// ---- z-slab (multi-GPU) entry points: see include/sph_hip.h ----------------------------------------

#define SLAB_CHECK(c)                                                                   \
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");                     \
    if (!c->slab) return fail(c, SPH_E_BAD_STATE, "not a slab context (sph_slab_create)"); \
    HIP_TRY(c, hipSetDevice(c->device))

int sph_slab_configure(SphContext *c, int z_lo, int z_hi, int halo_w, int has_left, int has_right)
{
    SLAB_CHECK(c);
    const int R = (int)c->dev.R;
    if (z_lo < 0 || z_hi > R || z_lo >= z_hi || halo_w < 2)
        return fail(c, SPH_E_INVALID_ARG, "bad slab range [%d,%d) of %d planes or halo %d < 2", z_lo, z_hi, R, halo_w);
    c->z_lo = z_lo;
    c->z_hi = z_hi;
    c->halo_w = halo_w;
    c->has_left = has_left != 0;
    c->has_right = has_right != 0;
    // an owned target may sit halo_w-1 planes outside the slab and still find its 27 cells in the halo
    c->dev.zt_lo = c->has_left ? z_lo - (halo_w - 1) : INT_MIN;
    c->dev.zt_hi = c->has_right ? z_hi - 1 + (halo_w - 1) : INT_MAX;
    c->dev.zo_lo = z_lo;   // bricks made only of ghost planes have no targets and return at once
    c->dev.zo_hi = z_hi;
    return SPH_OK;
}

int sph_slab_upload(SphContext *c, uint32_t n, const float *pos4, const float *vel4, const uint32_t *ids)
{
    SLAB_CHECK(c);
    if (n > c->capacity) return fail(c, SPH_E_INVALID_ARG, "%u particles exceed the local capacity %u", n, c->capacity);
    if (n && (!pos4 || !ids)) return fail(c, SPH_E_INVALID_ARG, "pos4 / ids is NULL");
    HIP_TRY(c, hipMemcpyAsync(c->pos4_dev, pos4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    if (vel4) HIP_TRY(c, hipMemcpyAsync(c->vel4_dev, vel4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->ids_dev, ids, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->dev.counters + kCntWildP, 0, sizeof(unsigned long long) * 3, c->stream));
    c->dev.wild_v_cur = 0;
    c->dev.force_slow = 0;
    c->dev.count = n;
    c->dev.n_owned = n;
    launch_slab_upload(c->dev, n, c->pos4_dev, vel4 ? c->vel4_dev : nullptr, c->ids_dev, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->uploaded = true;
    c->grid_valid = false;
    c->mesh_valid = false;
    c->substeps_since_upload = 0;
    c->frames_since_upload = 0;
    reset_list_policy(c);
    c->n_ghost[0] = c->n_ghost[1] = c->n_export[0] = c->n_export[1] = 0;
    return SPH_OK;
}

int sph_slab_begin_frame(SphContext *c, void *send_left, void *send_right, uint32_t cap_records, uint32_t counts[3])
{
    SLAB_CHECK(c);
    if (!c->uploaded) return fail(c, SPH_E_BAD_STATE, "sph_slab_begin_frame before sph_slab_upload");
    if (!counts) return fail(c, SPH_E_INVALID_ARG, "counts is NULL");
    HIP_TRY(c, hipMemsetAsync(c->slab_counters, 0, 4 * sizeof(uint32_t), c->stream));
    launch_slab_classify(c->dev, c->z_lo, c->z_hi, c->has_left, c->has_right, (float *)send_left, (float *)send_right,
                         cap_records, c->slab_counters, c->stream);
    HIP_TRY(c, hipGetLastError());
    uint32_t h[4];
    HIP_TRY(c, hipMemcpyAsync(h, c->slab_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (h[1] > cap_records || h[2] > cap_records)
        return fail(c, SPH_E_INVALID_ARG, "migration buffers too small: %u / %u records, capacity %u", h[1], h[2], cap_records);
    std::swap(c->dev.cur, c->dev.nxt);
    std::swap(c->dev.id, c->dev.id_nxt);
    c->dev.count = h[0];
    c->dev.n_owned = h[0];
    c->grid_valid = false;
    c->n_ghost[0] = c->n_ghost[1] = c->n_export[0] = c->n_export[1] = 0;
    counts[0] = h[0];
    counts[1] = h[1];
    counts[2] = h[2];
    return SPH_OK;
}

int sph_slab_append(SphContext *c, const void *recv_left, uint32_t n_left, const void *recv_right, uint32_t n_right,
                    int as_ghost)
{
    SLAB_CHECK(c);
    if ((uint64_t)c->dev.count + n_left + n_right > c->capacity)
        return fail(c, SPH_E_OOM, "slab overflow: %u + %u + %u particles exceed the local capacity %u", c->dev.count,
                    n_left, n_right, c->capacity);
    if (!as_ghost && c->dev.count != c->dev.n_owned)
        return fail(c, SPH_E_BAD_STATE, "owned particles must be appended before ghosts");
    launch_slab_append(c->dev, (const float *)recv_left, n_left, c->dev.count, c->stream);
    launch_slab_append(c->dev, (const float *)recv_right, n_right, c->dev.count + n_left, c->stream);
    HIP_TRY(c, hipGetLastError());
    c->dev.count += n_left + n_right;
    if (as_ghost) {
        c->n_ghost[0] = n_left;
        c->n_ghost[1] = n_right;
    } else {
        c->dev.n_owned = c->dev.count;
    }
    return SPH_OK;
}

int sph_slab_export_ghosts(SphContext *c, void *send_left, void *send_right, uint32_t cap_records, uint32_t counts[2])
{
    SLAB_CHECK(c);
    if (!counts) return fail(c, SPH_E_INVALID_ARG, "counts is NULL");
    if (c->dev.count != c->dev.n_owned) return fail(c, SPH_E_BAD_STATE, "export ghosts before appending any");
    HIP_TRY(c, hipMemsetAsync(c->slab_counters, 0, 4 * sizeof(uint32_t), c->stream));
    const uint32_t cap = cap_records < c->capacity ? cap_records : c->capacity;
    launch_slab_export(c->dev, c->z_lo, c->z_hi, c->halo_w, c->has_left, c->has_right, (float *)send_left,
                       (float *)send_right, c->send_src[0], c->send_src[1], cap, c->slab_counters, c->stream);
    HIP_TRY(c, hipGetLastError());
    uint32_t h[4];
    HIP_TRY(c, hipMemcpyAsync(h, c->slab_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (h[1] > cap || h[2] > cap)
        return fail(c, SPH_E_INVALID_ARG, "ghost buffers too small: %u / %u records, capacity %u", h[1], h[2], cap);
    c->n_export[0] = counts[0] = h[1];
    c->n_export[1] = counts[1] = h[2];
    return SPH_OK;
}

int sph_slab_pack_update(SphContext *c, int what, void *send_left, void *send_right)
{
    SLAB_CHECK(c);
    if (!c->grid_valid) return fail(c, SPH_E_BAD_STATE, "ghost update before BucketGeneration");
    if (what != SPH_SLAB_UPDATE_RHO && what != SPH_SLAB_UPDATE_POSVEL) return fail(c, SPH_E_INVALID_ARG, "bad update kind");
    launch_slab_pack_update(c->dev, c->send_src[0], c->n_export[0], what, (float *)send_left, c->stream);
    launch_slab_pack_update(c->dev, c->send_src[1], c->n_export[1], what, (float *)send_right, c->stream);
    HIP_TRY(c, hipGetLastError());
    return SPH_OK;
}

int sph_slab_unpack_update(SphContext *c, int what, const void *recv_left, const void *recv_right)
{
    SLAB_CHECK(c);
    if (!c->grid_valid) return fail(c, SPH_E_BAD_STATE, "ghost update before BucketGeneration");
    if (what != SPH_SLAB_UPDATE_RHO && what != SPH_SLAB_UPDATE_POSVEL) return fail(c, SPH_E_INVALID_ARG, "bad update kind");
    launch_slab_unpack_update(c->dev, c->dev.n_owned, c->n_ghost[0], what, (const float *)recv_left, c->stream);
    launch_slab_unpack_update(c->dev, c->dev.n_owned + c->n_ghost[0], c->n_ghost[1], what, (const float *)recv_right,
                              c->stream);
    HIP_TRY(c, hipGetLastError());
    return SPH_OK;
}

int sph_slab_counts(SphContext *c, uint32_t out[5])
{
    SLAB_CHECK(c);
    if (!out) return fail(c, SPH_E_INVALID_ARG, "out is NULL");
    out[0] = c->dev.n_owned;
    out[1] = c->n_ghost[0];
    out[2] = c->n_ghost[1];
    out[3] = c->n_export[0];
    out[4] = c->n_export[1];
    return SPH_OK;
}

int sph_slab_download(SphContext *c, uint32_t cap, uint32_t *ids, float *pos4, float *vel4, float *rho, uint32_t *n_out)
{
    SLAB_CHECK(c);
    if (!ids || !pos4 || !vel4 || !rho || !n_out) return fail(c, SPH_E_INVALID_ARG, "NULL argument");
    if (!c->grid_valid) {   // before the first build: owned particles are the first n_owned entries, no ghosts
        HIP_TRY(c, hipMemsetAsync(c->dev.ghost, 0, c->capacity, c->stream));
    }
    HIP_TRY(c, hipMemsetAsync(c->slab_counters, 0, 4 * sizeof(uint32_t), c->stream));
    launch_slab_download(c->dev, c->substeps_since_upload, c->dt_last, c->slab_counters, c->ids_dev, c->pos4_dev,
                         c->vel4_dev, c->rho_id_dev, c->stream);
    HIP_TRY(c, hipGetLastError());
    uint32_t n = 0;
    HIP_TRY(c, hipMemcpyAsync(&n, c->slab_counters, sizeof n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (n > cap) return fail(c, SPH_E_INVALID_ARG, "download arrays hold %u particles, %u owned", cap, n);
    HIP_TRY(c, hipMemcpyAsync(ids, c->ids_dev, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(pos4, c->pos4_dev, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(vel4, c->vel4_dev, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(rho, c->rho_id_dev, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_out = n;
    return SPH_OK;
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

}  // extern "C"
