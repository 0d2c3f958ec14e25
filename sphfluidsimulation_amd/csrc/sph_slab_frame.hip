// z-slab frames: one Update() (SphFluidSimulation.cs:96-102) of a particle set that is split along z over several GPUs,
// with the halo exchange between the slabs done from INSIDE the library.  New work: the reference is
// single-GPU.  The same frame driver serves
//   * one process per GPU (sph_slab_create + sph_slab_configure + sph_slab_comm_init, then sph_slab_step): RCCL
//     ncclSend/ncclRecv between z-neighbours over xGMI, one ncclAllReduce of a scalar per frame;
//   * several GPUs in ONE process behind the ordinary calls (SphParams.num_devices / device_ids: sph_create, sph_upload,
//     sph_step, sph_download): RCCL (ncclCommInitAll) or, when the devices repeat (rehearsal on one GPU), peer copies;
//   * any host-side transport (sph_slab_set_host_transport: the tests run several ranks on one GPU over gloo).
//
// Frame protocol of a slab (W = halo planes; every message goes to / comes from the two z-neighbours only):
//   A  k_slab_classify   keepers -> nxt; migrants (bucket row left the slab) and ghost exports (keepers binned in the W
//                        boundary planes of a face) packed per neighbour; a migrant that lands in the neighbour's W
//                        boundary planes STAYS here as a ghost, so nothing has to be sent back: one round instead of two.
//   B  16-byte headers (counts) to the neighbours + all-reduce of the halo-violation counter, then the ONE host
//      synchronisation of the frame: the host learns every message size, and whether the PREVIOUS frame was valid.
//   C  exact-size payloads (migrants, ghosts), appended behind the keepers; snapshot of the owned particles.
//   D  BucketGeneration, DensityCalculation, ghost rho refresh, S x { UpdateVelocityAndPosition, ghost pos/vel refresh }:
//      sizes known, nothing waits for the host.
// Halo violations.  Membership is frozen for the frame while particles move (SphFluidSimulation.cs:98-102): an owned
// target may drift W-1 planes out of its slab and still find its 27 cells among the ghosts; one that drifts further is
// counted (kCntHaloViolation).  The sum over all slabs is looked at in step B of the NEXT frame (or by sph_slab_finish):
// if it moved, the frame is re-run from its snapshot with twice the halo, so the result stays bit-identical to the
// single-GPU run.  Slabs must be at least 2 W planes thick (a migrant then never lands in the far face's boundary).
#include "sph_context.hpp"

#include <dlfcn.h>
#include <limits.h>
#include <string.h>

#include <rccl/rccl.h>   // types; the functions are resolved at run time (no hard dependency for single-GPU users)

#include <algorithm>
#include <memory>

namespace sph {

// ---- per-slab exchange state ------------------------------------------------------------------------------------------

struct SlabHost {                 // pinned: what the host reads after the synchronisation of step B
    uint32_t counters[kSlabCounterCount];
    uint32_t hdr_recv[8];
    unsigned long long viol;
};

struct SlabRank {
    int rank = 0, world = 1;
    int nb_lo = 0, nb_hi = 0;
    uint32_t cap_side = 0;        // records per exchange buffer
    float *send_mig[2] = {nullptr, nullptr}, *send_ghost[2] = {nullptr, nullptr};
    float *recv_mig[2] = {nullptr, nullptr}, *recv_ghost[2] = {nullptr, nullptr}, *kept[2] = {nullptr, nullptr};
    uint32_t *send_src[2] = {nullptr, nullptr};
    uint32_t *counters = nullptr, *hdr_send = nullptr, *hdr_recv = nullptr;
    unsigned long long *viol_dev = nullptr;
    SlabHost *host = nullptr;
    // owned particles at the start of the frame in flight (after migration): what a re-run starts from
    float *snap_x = nullptr, *snap_y = nullptr, *snap_z = nullptr, *snap_vw = nullptr;
    uint2 *snap_h4 = nullptr;     // precision 1: the fp16 records and their reference cells instead of x, y, z
    uint32_t *snap_ck = nullptr;
    float4 *snap_rv = nullptr;
    uint32_t *snap_id = nullptr;
    uint32_t snap_n = 0;
    int snap_substeps = 0;        // SphContext::substeps_since_upload at that point
    float snap_dt_last = 0.0f;
    bool snap_valid = false;
    // counts of the frame in flight
    uint32_t n_keep = 0, mig_out[2] = {0, 0}, kept_n[2] = {0, 0}, gh_out[2] = {0, 0}, mig_in[2] = {0, 0}, gh_in[2] = {0, 0};
    uint32_t n_export[2] = {0, 0};   // particles refreshed for each neighbour: own exports + its kept migrants
    uint32_t n_ghost[2] = {0, 0};    // ghosts held from each neighbour: received + own kept migrants
    hipEvent_t ev_packed = nullptr, ev_pulled = nullptr;   // peer-copy transport
    hipEvent_t ev_zone_done = nullptr, ev_refresh_done = nullptr;   // two-pass sub-step: compute stream <-> communication stream
    ncclComm_t comm = nullptr;       // RCCL transport
    // host-staged transport
    void *stage[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t stage_bytes[4] = {0, 0, 0, 0};
};

struct Xfer {                     // one exchange of one slab: [0] left neighbour, [1] right neighbour; device pointers
    const void *send[2] = {nullptr, nullptr};
    size_t send_bytes[2] = {0, 0};
    void *recv[2] = {nullptr, nullptr};
    size_t recv_bytes[2] = {0, 0};
};

struct SlabGroup;

struct Transport {
    virtual ~Transport() {}
    virtual const char *name() const = 0;
    // enqueue on every local slab's stream (comm = its communication stream: the refresh of a two-pass sub-step); ordered
    // like any other work of that stream
    virtual int exchange(SlabGroup &g, std::vector<Xfer> &x, bool comm = false) = 0;
    // sum over ALL slabs of SlabRank::viol_dev: begin() before the synchronisation (a transport that reduces on the device
    // does it here, in place), end() after it, from the pinned copies
    virtual int allreduce_begin(SlabGroup &) { return SPH_OK; }
    virtual int allreduce_end(SlabGroup &g, unsigned long long *total) = 0;
};

static inline hipStream_t stream_of(const SphContext *m, bool comm) { return comm ? m->comm_stream : m->stream; }

struct SlabGroup {
    std::vector<SphContext *> members;   // local slabs, ascending rank
    std::unique_ptr<Transport> transport;
    int world = 1;
    int halo_w = 2, max_halo = 2;
    bool pending = false;                // a frame was issued and not validated yet
    float delta_prev = 0.0f;
    unsigned long long viol_seen = 0;    // global violation count at the last look
    uint64_t reruns = 0;
    bool single_pass = false;            // SPH_FLAG_NO_OVERLAP: refresh after the whole sub-step (diagnostic)
    bool force_two_pass = false;         // SPH_FLAG_FORCE_TWO_PASS: also with one slab (diagnostic: what the second pass costs)
    SphContext *owner = nullptr;         // error messages go here
};

static int gfail(SlabGroup &g, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    for (SphContext *m : g.members) m->last_error = buf;
    if (g.owner) g.owner->last_error = buf;
    g_last_error = buf;
    return code;
}

#define G_HIP(g, expr)                                                                                        \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess)                                                                                 \
            return gfail(g, _e == hipErrorOutOfMemory ? SPH_E_OOM : SPH_E_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ---- transports ---------------------------------------------------------------------------------------------------------

// no neighbours anywhere (one slab): nothing to move
struct NullTransport : Transport {
    const char *name() const override { return "none"; }
    int exchange(SlabGroup &, std::vector<Xfer> &, bool) override { return SPH_OK; }
    int allreduce_end(SlabGroup &g, unsigned long long *total) override
    {
        *total = 0;
        for (SphContext *m : g.members) *total += m->rank->host->viol;
        return SPH_OK;
    }
};

// All slabs live in this process: a slab PULLS what its neighbours packed (hipMemcpyPeerAsync; a plain device copy when
// both are on one GPU -- the one-GPU rehearsal).  Events order the neighbours' streams: nobody reads a buffer before it
// was packed, nobody re-packs a buffer before the neighbour has pulled it.
struct PeerCopyTransport : Transport {
    const char *name() const override { return "peer-copy"; }
    int exchange(SlabGroup &g, std::vector<Xfer> &x, bool comm) override
    {
        const size_t L = g.members.size();
        for (size_t i = 0; i < L; i++) {
            SphContext *m = g.members[i];
            G_HIP(g, hipSetDevice(m->device));
            G_HIP(g, hipEventRecord(m->rank->ev_packed, stream_of(m, comm)));
        }
        for (size_t i = 0; i < L; i++) {
            SphContext *m = g.members[i];
            G_HIP(g, hipSetDevice(m->device));
            for (int side = 0; side < 2; side++) {
                if (!x[i].recv_bytes[side]) continue;
                const size_t j = side == 0 ? i - 1 : i + 1;   // the neighbour; it sends to me through ITS other side
                if (j >= L) return gfail(g, SPH_E_COMM, "peer-copy transport: slab %zu has no local neighbour on side %d", i, side);
                if (x[j].send_bytes[1 - side] != x[i].recv_bytes[side])
                    return gfail(g, SPH_E_COMM, "peer-copy transport: slab %zu expects %zu bytes from slab %zu, which sends %zu", i,
                                 x[i].recv_bytes[side], j, x[j].send_bytes[1 - side]);
                G_HIP(g, hipStreamWaitEvent(stream_of(m, comm), g.members[j]->rank->ev_packed, 0));
                G_HIP(g, hipMemcpyPeerAsync(x[i].recv[side], m->device, x[j].send[1 - side], g.members[j]->device,
                                            x[i].recv_bytes[side], stream_of(m, comm)));
            }
            G_HIP(g, hipEventRecord(m->rank->ev_pulled, stream_of(m, comm)));
        }
        for (size_t i = 0; i < L; i++) {
            SphContext *m = g.members[i];
            G_HIP(g, hipSetDevice(m->device));
            if (i > 0) G_HIP(g, hipStreamWaitEvent(stream_of(m, comm), g.members[i - 1]->rank->ev_pulled, 0));
            if (i + 1 < L) G_HIP(g, hipStreamWaitEvent(stream_of(m, comm), g.members[i + 1]->rank->ev_pulled, 0));
        }
        return SPH_OK;
    }
    int allreduce_end(SlabGroup &g, unsigned long long *total) override
    {
        *total = 0;
        for (SphContext *m : g.members) *total += m->rank->host->viol;
        return SPH_OK;
    }
};

// RCCL (librccl.so.1, resolved at first use: the image torch loaded if there is one, the system's otherwise).
struct RcclApi {
    void *lib = nullptr;
    std::string path;
    int version = 0;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
};

static RcclApi *rccl_api(std::string *why)
{
    static RcclApi api;
    static bool tried = false;
    static std::string error;
    if (!tried) {
        tried = true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) {
            error = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : "");
        } else {
#define RSYM(field, sym)                                                          \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, sym));       \
    if (!api.field) error += std::string(" missing ") + sym;
            RSYM(GetUniqueId, "ncclGetUniqueId")
            RSYM(CommInitRank, "ncclCommInitRank")
            RSYM(CommInitAll, "ncclCommInitAll")
            RSYM(CommDestroy, "ncclCommDestroy")
            RSYM(GroupStart, "ncclGroupStart")
            RSYM(GroupEnd, "ncclGroupEnd")
            RSYM(Send, "ncclSend")
            RSYM(Recv, "ncclRecv")
            RSYM(AllReduce, "ncclAllReduce")
            RSYM(GetErrorString, "ncclGetErrorString")
            RSYM(GetVersion, "ncclGetVersion")
#undef RSYM
            if (error.empty()) {
                Dl_info info;
                if (dladdr(reinterpret_cast<void *>(api.Send), &info) && info.dli_fname) api.path = info.dli_fname;
                (void)api.GetVersion(&api.version);
            }
        }
    }
    if (!error.empty()) {
        if (why) *why = error;
        return nullptr;
    }
    return &api;
}

#define G_NCCL(g, api, expr)                                                                                 \
    do {                                                                                                     \
        ncclResult_t _r = (expr);                                                                            \
        if (_r != ncclSuccess) return gfail(g, SPH_E_COMM, "%s: %s", #expr, (api)->GetErrorString(_r));      \
    } while (0)

struct RcclTransport : Transport {
    RcclApi *api;
    explicit RcclTransport(RcclApi *a) : api(a) {}
    const char *name() const override { return "rccl"; }
    int exchange(SlabGroup &g, std::vector<Xfer> &x, bool comm) override
    {
        bool any = false;
        for (const Xfer &t : x) any = any || t.send_bytes[0] || t.send_bytes[1] || t.recv_bytes[0] || t.recv_bytes[1];
        if (!any) return SPH_OK;
        // one group over every local slab: each talks to its two z-neighbours only (2 of the 7 xGMI links)
        G_NCCL(g, api, api->GroupStart());
        for (size_t i = 0; i < g.members.size(); i++) {
            SphContext *m = g.members[i];
            const SlabRank *r = m->rank;
            for (int side = 0; side < 2; side++) {
                const int peer = side == 0 ? r->rank - 1 : r->rank + 1;
                if (x[i].send_bytes[side])
                    G_NCCL(g, api, api->Send(x[i].send[side], x[i].send_bytes[side], ncclChar, peer, r->comm, stream_of(m, comm)));
                if (x[i].recv_bytes[side])
                    G_NCCL(g, api, api->Recv(x[i].recv[side], x[i].recv_bytes[side], ncclChar, peer, r->comm, stream_of(m, comm)));
            }
        }
        G_NCCL(g, api, api->GroupEnd());
        return SPH_OK;
    }
    int allreduce_begin(SlabGroup &g) override
    {
        if (g.world == 1) return SPH_OK;
        G_NCCL(g, api, api->GroupStart());
        for (SphContext *m : g.members)
            G_NCCL(g, api, api->AllReduce(m->rank->viol_dev, m->rank->viol_dev, 1, ncclUint64, ncclSum, m->rank->comm, m->stream));
        G_NCCL(g, api, api->GroupEnd());
        return SPH_OK;
    }
    int allreduce_end(SlabGroup &g, unsigned long long *total) override
    {
        *total = g.members[0]->rank->host->viol;   // already the global sum on every slab
        if (g.world == 1) {
            *total = 0;
            for (SphContext *m : g.members) *total += m->rank->host->viol;
        }
        return SPH_OK;
    }
};

// The host moves the bytes (tests: gloo between processes that share one GPU).  Staged through pinned memory.
struct HostTransport : Transport {
    SphHostExchangeFn exchange_fn;
    SphHostAllreduceFn allreduce_fn;
    void *user;
    HostTransport(SphHostExchangeFn e, SphHostAllreduceFn a, void *u) : exchange_fn(e), allreduce_fn(a), user(u) {}
    const char *name() const override { return "host-staged"; }
    int exchange(SlabGroup &g, std::vector<Xfer> &x, bool comm) override
    {
        if (g.members.size() != 1) return gfail(g, SPH_E_COMM, "host transport: one slab per process");
        SphContext *m = g.members[0];
        SlabRank *r = m->rank;
        G_HIP(g, hipSetDevice(m->device));
        const size_t need[4] = {x[0].send_bytes[0], x[0].send_bytes[1], x[0].recv_bytes[0], x[0].recv_bytes[1]};
        for (int k = 0; k < 4; k++)
            if (need[k] > r->stage_bytes[k]) {
                if (r->stage[k]) (void)hipHostFree(r->stage[k]);
                r->stage[k] = nullptr;
                r->stage_bytes[k] = 0;
                const size_t bytes = need[k] + need[k] / 2 + 4096;
                G_HIP(g, hipHostMalloc(&r->stage[k], bytes, hipHostMallocDefault));
                r->stage_bytes[k] = bytes;
            }
        for (int side = 0; side < 2; side++)
            if (need[side]) G_HIP(g, hipMemcpyAsync(r->stage[side], x[0].send[side], need[side], hipMemcpyDeviceToHost, stream_of(m, comm)));
        G_HIP(g, hipStreamSynchronize(stream_of(m, comm)));
        m->host_syncs++;
        const int rc = exchange_fn(user, r->stage[0], need[0], r->stage[1], need[1], r->stage[2], need[2], r->stage[3], need[3]);
        if (rc != 0) return gfail(g, SPH_E_COMM, "host transport: the exchange callback returned %d", rc);
        for (int side = 0; side < 2; side++)
            if (need[2 + side])
                G_HIP(g, hipMemcpyAsync(x[0].recv[side], r->stage[2 + side], need[2 + side], hipMemcpyHostToDevice, stream_of(m, comm)));
        return SPH_OK;
    }
    int allreduce_end(SlabGroup &g, unsigned long long *total) override
    {
        *total = g.members[0]->rank->host->viol;
        if (g.world > 1) {
            const int rc = allreduce_fn(user, total);
            if (rc != 0) return gfail(g, SPH_E_COMM, "host transport: the all-reduce callback returned %d", rc);
        }
        return SPH_OK;
    }
};

// ---- slab set-up ------------------------------------------------------------------------------------------------------------

void slab_planes(int R, int world, int rank, int *z_lo, int *z_hi)
{
    *z_lo = (int)(((long long)rank * R) / world);
    *z_hi = (int)(((long long)(rank + 1) * R) / world);
}

static int group_max_halo(int R, int world)
{
    int mn = INT_MAX;
    for (int r = 0; r < world; r++) {
        int lo, hi;
        slab_planes(R, world, r, &lo, &hi);
        mn = std::min(mn, hi - lo);
    }
    return world == 1 ? std::max(2, R) : mn / 2;
}

static void apply_halo(SphContext *c, int halo_w)
{
    c->halo_w = halo_w;
    // an owned target may sit halo_w-1 planes outside the slab and still find its 27 cells in the halo
    c->dev.zt_lo = c->has_left ? c->z_lo - (halo_w - 1) : INT_MIN;
    c->dev.zt_hi = c->has_right ? c->z_hi - 1 + (halo_w - 1) : INT_MAX;
    c->dev.zo_lo = c->z_lo;   // bricks made only of ghost planes have no targets and return at once
    c->dev.zo_hi = c->z_hi;
    // the boundary planes of a two-pass sub-step: exactly the planes whose owned particles the neighbours hold as ghosts
    c->dev.zone_p[0] = c->has_left ? c->z_lo : 0;
    c->dev.zone_p[1] = c->has_left ? std::min(c->z_lo + halo_w, c->z_hi) : 0;
    c->dev.zone_p[2] = c->has_right ? std::max(c->z_hi - halo_w, c->z_lo) : 0;
    c->dev.zone_p[3] = c->has_right ? c->z_hi : 0;
}

template <typename T>
static int ralloc(SphContext *c, T **out, size_t n)
{
    void *p = nullptr;
    HIP_TRY(c, hipMalloc(&p, (n ? n : 1) * sizeof(T)));
    c->allocs.push_back(p);
    *out = (T *)p;
    return SPH_OK;
}

void slab_rank_destroy(SphContext *c)
{
    SlabRank *r = c->rank;
    if (!r) return;
    if (r->comm) {
        if (RcclApi *api = rccl_api(nullptr)) (void)api->CommDestroy(r->comm);
    }
    if (r->host) (void)hipHostFree(r->host);
    for (void *p : r->stage)
        if (p) (void)hipHostFree(p);
    if (r->ev_packed) (void)hipEventDestroy(r->ev_packed);
    if (r->ev_pulled) (void)hipEventDestroy(r->ev_pulled);
    if (r->ev_zone_done) (void)hipEventDestroy(r->ev_zone_done);
    if (r->ev_refresh_done) (void)hipEventDestroy(r->ev_refresh_done);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    c->comm_stream = nullptr;
    delete r;
    c->rank = nullptr;
}

void slab_group_destroy(SlabGroup *g) { delete g; }

// Places slab `rank` of `world` on this context: plane range, neighbours, halo, exchange buffers.
int slab_configure(SphContext *c, int rank, int world)
{
    const int R = (int)c->dev.R;
    if (world < 1 || rank < 0 || rank >= world) return fail(c, SPH_E_INVALID_ARG, "slab %d of %d", rank, world);
    if (world > R) return fail(c, SPH_E_INVALID_ARG, "%d slabs for %d grid planes", world, R);
    if (c->rank) return fail(c, SPH_E_BAD_STATE, "slab already configured");
    int halo = c->params.halo_planes ? (int)c->params.halo_planes : 2;
    const int max_halo = group_max_halo(R, world);
    if (halo < 2) halo = 2;
    if (world > 1 && halo > max_halo)
        return fail(c, SPH_E_INVALID_ARG, "%d halo planes need slabs of >= %d planes; %d slabs of a %d-plane grid are thinner", halo,
                    2 * halo, world, R);
    SlabRank *r = new SlabRank();
    c->rank = r;
    r->rank = rank;
    r->world = world;
    slab_planes(R, world, rank, &c->z_lo, &c->z_hi);
    c->has_left = rank > 0;
    c->has_right = rank + 1 < world;
    int lo, hi;
    slab_planes(R, world, rank > 0 ? rank - 1 : 0, &lo, &hi);
    r->nb_lo = lo;
    slab_planes(R, world, rank + 1 < world ? rank + 1 : rank, &lo, &hi);
    r->nb_hi = hi;
    apply_halo(c, halo);
    HIP_TRY(c, hipSetDevice(c->device));
    int rc;
    const size_t n = c->capacity;
    r->cap_side = (uint32_t)std::max<size_t>(n / 2, 1024);
    const size_t words = (size_t)r->cap_side * kSlabRecordWords;
    for (int side = 0; side < 2; side++) {
        if (!(side == 0 ? c->has_left : c->has_right)) continue;
        if ((rc = ralloc(c, &r->send_mig[side], words)) != SPH_OK || (rc = ralloc(c, &r->send_ghost[side], words)) != SPH_OK ||
            (rc = ralloc(c, &r->recv_mig[side], words)) != SPH_OK || (rc = ralloc(c, &r->recv_ghost[side], words)) != SPH_OK ||
            (rc = ralloc(c, &r->kept[side], words)) != SPH_OK || (rc = ralloc(c, &r->send_src[side], (size_t)r->cap_side)) != SPH_OK)
            return rc;
    }
    if ((rc = ralloc(c, &r->counters, (size_t)kSlabCounterCount)) != SPH_OK || (rc = ralloc(c, &r->hdr_send, (size_t)8)) != SPH_OK ||
        (rc = ralloc(c, &r->hdr_recv, (size_t)8)) != SPH_OK || (rc = ralloc(c, &r->viol_dev, (size_t)1)) != SPH_OK)
        return rc;
    if (c->dev.prec == 0u) {
        if ((rc = ralloc(c, &r->snap_x, n)) != SPH_OK || (rc = ralloc(c, &r->snap_y, n)) != SPH_OK || (rc = ralloc(c, &r->snap_z, n)) != SPH_OK)
            return rc;
    } else if ((rc = ralloc(c, &r->snap_h4, n)) != SPH_OK || (rc = ralloc(c, &r->snap_ck, n)) != SPH_OK) {
        return rc;
    }
    if ((rc = ralloc(c, &r->snap_vw, n)) != SPH_OK || (rc = ralloc(c, &r->snap_rv, n)) != SPH_OK || (rc = ralloc(c, &r->snap_id, n)) != SPH_OK)
        return rc;
    HIP_TRY(c, hipMemset(r->counters, 0, kSlabCounterCount * sizeof(uint32_t)));   // (kSlabMisplaced accumulates from here on)
    HIP_TRY(c, hipMemset(r->hdr_send, 0, 8 * sizeof(uint32_t)));
    HIP_TRY(c, hipMemset(r->hdr_recv, 0, 8 * sizeof(uint32_t)));
    HIP_TRY(c, hipMemset(r->viol_dev, 0, sizeof(unsigned long long)));
    HIP_TRY(c, hipHostMalloc((void **)&r->host, sizeof(SlabHost), hipHostMallocDefault));
    memset(r->host, 0, sizeof(SlabHost));
    HIP_TRY(c, hipEventCreateWithFlags(&r->ev_packed, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&r->ev_pulled, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&r->ev_zone_done, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&r->ev_refresh_done, hipEventDisableTiming));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    return SPH_OK;
}

SlabGroup *slab_group_new(SphContext *owner, std::vector<SphContext *> members, int world)
{
    SlabGroup *g = new SlabGroup();
    g->members = std::move(members);
    g->world = world;
    g->owner = owner;
    g->halo_w = g->members[0]->halo_w;
    g->max_halo = group_max_halo((int)g->members[0]->dev.R, world);
    g->single_pass = (g->members[0]->params.flags & SPH_FLAG_NO_OVERLAP) != 0;
    g->force_two_pass = (g->members[0]->params.flags & SPH_FLAG_FORCE_TWO_PASS) != 0;
    if (world == 1) g->transport.reset(new NullTransport());
    for (SphContext *m : g->members) m->group = g;
    return g;
}

int slab_group_use_peer_copies(SlabGroup *g)
{
    for (size_t i = 0; i + 1 < g->members.size(); i++) {   // direct xGMI copies between neighbours where the devices differ
        const int a = g->members[i]->device, b = g->members[i + 1]->device;
        if (a == b) continue;
        int ok = 0;
        if (hipDeviceCanAccessPeer(&ok, a, b) == hipSuccess && ok) {
            (void)hipSetDevice(a);
            (void)hipDeviceEnablePeerAccess(b, 0);
            (void)hipSetDevice(b);
            (void)hipDeviceEnablePeerAccess(a, 0);
        }
        (void)hipGetLastError();
    }
    g->transport.reset(new PeerCopyTransport());
    return SPH_OK;
}

int slab_group_use_host(SlabGroup *g, SphHostExchangeFn e, SphHostAllreduceFn a, void *user)
{
    g->transport.reset(new HostTransport(e, a, user));
    return SPH_OK;
}

int slab_rccl_unique_id(void *id128)
{
    std::string why;
    RcclApi *api = rccl_api(&why);
    if (!api) return fail(nullptr, SPH_E_COMM, "RCCL unavailable: %s", why.c_str());
    static_assert(sizeof(ncclUniqueId) == 128, "SPH_COMM_ID_BYTES");
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, SPH_E_COMM, "ncclGetUniqueId: %s", api->GetErrorString(r));
    memcpy(id128, &id, sizeof id);
    return SPH_OK;
}

// one process per GPU: every rank calls this with the id rank 0 made
int slab_group_use_rccl_rank(SlabGroup *g, const void *id128)
{
    std::string why;
    RcclApi *api = rccl_api(&why);
    if (!api) return gfail(*g, SPH_E_COMM, "RCCL unavailable: %s", why.c_str());
    SphContext *m = g->members[0];
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    G_HIP(*g, hipSetDevice(m->device));
    G_NCCL(*g, api, api->CommInitRank(&m->rank->comm, g->world, id, m->rank->rank));
    g->transport.reset(new RcclTransport(api));
    return SPH_OK;
}

// several GPUs in this process
int slab_group_use_rccl_all(SlabGroup *g)
{
    std::string why;
    RcclApi *api = rccl_api(&why);
    if (!api) return gfail(*g, SPH_E_COMM, "RCCL unavailable: %s", why.c_str());
    std::vector<int> devs;
    for (SphContext *m : g->members) devs.push_back(m->device);
    std::vector<ncclComm_t> comms(devs.size());
    G_NCCL(*g, api, api->CommInitAll(comms.data(), (int)devs.size(), devs.data()));
    for (size_t i = 0; i < devs.size(); i++) g->members[i]->rank->comm = comms[i];
    g->transport.reset(new RcclTransport(api));
    return SPH_OK;
}

const char *slab_group_transport_name(SlabGroup *g) { return g->transport ? g->transport->name() : "unset"; }
uint64_t slab_group_reruns(SlabGroup *g) { return g->reruns; }
int slab_group_halo(SlabGroup *g) { return g->halo_w; }

// Which RCCL image this process has mapped, WITHOUT loading one: a second RCCL image next to another component's (e.g. the
// one bundled with PyTorch) corrupts the heap at exit, so the library only ever loads RCCL when a transport asks for it.
int slab_rccl_info(char *path, size_t path_len, int *version)
{
    void *lib = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        if (lib) break;
    }
    if (!lib) return SPH_E_COMM;
    auto get_version = reinterpret_cast<ncclResult_t (*)(int *)>(dlsym(lib, "ncclGetVersion"));
    if (get_version) {
        Dl_info info;
        if (path && path_len && dladdr(reinterpret_cast<void *>(get_version), &info) && info.dli_fname) snprintf(path, path_len, "%s", info.dli_fname);
        if (version) (void)get_version(version);
    }
    dlclose(lib);
    return SPH_OK;
}

// ---- the frame ----------------------------------------------------------------------------------------------------------------

static SlabArgs slab_args(const SphContext *c)
{
    const SlabRank *r = c->rank;
    SlabArgs a{};
    a.z_lo = c->z_lo;
    a.z_hi = c->z_hi;
    a.halo_w = c->halo_w;
    a.has_left = c->has_left;
    a.has_right = c->has_right;
    a.nb_lo = r->nb_lo;
    a.nb_hi = r->nb_hi;
    for (int s = 0; s < 2; s++) {
        a.send_mig[s] = r->send_mig[s];
        a.send_ghost[s] = r->send_ghost[s];
        a.kept[s] = r->kept[s];
        a.send_src[s] = r->send_src[s];
    }
    a.counters = r->counters;
    a.cap_records = r->cap_side;
    return a;
}

#define EACH(g, m) for (SphContext * m : (g).members)

// A: classify + pack, on the state the previous frame (or the upload) left
static int phase_classify(SlabGroup &g)
{
    EACH(g, m)
    {
        SlabRank *r = m->rank;
        G_HIP(g, hipSetDevice(m->device));
        if (!m->uploaded) return gfail(g, SPH_E_BAD_STATE, "slab frame before the particles were uploaded");
        G_HIP(g, hipMemsetAsync(r->counters, 0, kSlabMisplaced * sizeof(uint32_t), m->stream));   // (kSlabMisplaced accumulates)
        launch_slab_classify(m->dev, slab_args(m), r->hdr_send, r->viol_dev, m->stream);
        G_HIP(g, hipGetLastError());
    }
    return SPH_OK;
}

// B: headers + violation sum, then the host looks.  *viol_total = halo violations of all slabs since their creation.
static int phase_sync(SlabGroup &g, bool with_headers, unsigned long long *viol_total)
{
    if (!g.transport) return gfail(g, SPH_E_BAD_STATE, "no transport: sph_slab_comm_init / sph_slab_set_host_transport first");
    int rc;
    if (with_headers) {
        std::vector<Xfer> x(g.members.size());
        for (size_t i = 0; i < g.members.size(); i++) {
            SphContext *m = g.members[i];
            for (int side = 0; side < 2; side++) {
                if (!(side == 0 ? m->has_left : m->has_right)) continue;
                x[i].send[side] = m->rank->hdr_send + 4 * side;
                x[i].recv[side] = m->rank->hdr_recv + 4 * side;
                x[i].send_bytes[side] = x[i].recv_bytes[side] = 4 * sizeof(uint32_t);
            }
        }
        if ((rc = g.transport->exchange(g, x)) != SPH_OK) return rc;
    }
    if ((rc = g.transport->allreduce_begin(g)) != SPH_OK) return rc;
    EACH(g, m)
    {
        SlabRank *r = m->rank;
        G_HIP(g, hipSetDevice(m->device));
        G_HIP(g, hipMemcpyAsync(r->host->counters, r->counters, sizeof r->host->counters, hipMemcpyDeviceToHost, m->stream));
        G_HIP(g, hipMemcpyAsync(r->host->hdr_recv, r->hdr_recv, sizeof r->host->hdr_recv, hipMemcpyDeviceToHost, m->stream));
        G_HIP(g, hipMemcpyAsync(&r->host->viol, r->viol_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, m->stream));
    }
    EACH(g, m)
    {
        G_HIP(g, hipSetDevice(m->device));
        G_HIP(g, hipStreamSynchronize(m->stream));
        m->host_syncs++;
    }
    if (g.owner && g.owner != g.members[0]) g.owner->host_syncs++;
    return g.transport->allreduce_end(g, viol_total);
}

// what step B told every slab
static int read_counts(SlabGroup &g)
{
    EACH(g, m)
    {
        SlabRank *r = m->rank;
        const uint32_t *cn = r->host->counters;
        if (cn[kSlabLost])
            return gfail(g, SPH_E_COMM, "slab %d: %u particles moved past a whole neighbour slab in one frame (time step far beyond the "
                         "halo); use fewer devices", r->rank, cn[kSlabLost]);
        if (cn[kSlabMisplaced])
            return gfail(g, SPH_E_COMM, "slab %d: %u migrants could not be placed (slab thinner than twice the halo?)", r->rank,
                         cn[kSlabMisplaced]);
        r->n_keep = cn[kSlabKeep];
        for (int s = 0; s < 2; s++) {
            r->mig_out[s] = cn[kSlabMigL + s];
            r->kept_n[s] = cn[kSlabKeptL + s];
            r->gh_out[s] = cn[kSlabGhostL + s];
            const bool has = s == 0 ? m->has_left : m->has_right;
            const uint32_t *h = r->host->hdr_recv + 4 * s;
            if (has && (h[3] != (kSlabMagic ^ (uint32_t)m->halo_w)))
                return gfail(g, SPH_E_COMM, "slab %d: bad header from its %s neighbour (different halo or library version)", r->rank,
                             s == 0 ? "left" : "right");
            r->mig_in[s] = has ? h[0] : 0u;
            r->gh_in[s] = has ? h[1] : 0u;
            const uint32_t kept_by_nb = has ? h[2] : 0u;
            r->n_export[s] = r->gh_out[s] + kept_by_nb;
            r->n_ghost[s] = r->gh_in[s] + r->kept_n[s];
            const uint32_t big = std::max(std::max(r->mig_out[s], r->gh_out[s]), std::max(std::max(r->mig_in[s], r->gh_in[s]),
                                                                                        std::max(r->n_export[s], r->n_ghost[s])));
            if (big > r->cap_side)
                return gfail(g, SPH_E_OOM, "slab %d: %u records for one neighbour exceed the exchange buffers (%u); create the slab with a "
                             "larger local capacity", r->rank, big, r->cap_side);
        }
        const uint64_t total = (uint64_t)r->n_keep + r->mig_in[0] + r->mig_in[1] + r->n_ghost[0] + r->n_ghost[1];
        if (total > m->capacity)
            return gfail(g, SPH_E_OOM, "slab %d overflow: %llu particles (owned + ghosts) exceed the local capacity %u", r->rank,
                         (unsigned long long)total, m->capacity);
    }
    return SPH_OK;
}

static int exchange_records(SlabGroup &g, bool ghosts)
{
    std::vector<Xfer> x(g.members.size());
    for (size_t i = 0; i < g.members.size(); i++) {
        SlabRank *r = g.members[i]->rank;
        for (int s = 0; s < 2; s++) {
            x[i].send[s] = ghosts ? r->send_ghost[s] : r->send_mig[s];
            x[i].recv[s] = ghosts ? r->recv_ghost[s] : r->recv_mig[s];
            x[i].send_bytes[s] = (size_t)(ghosts ? r->gh_out[s] : r->mig_out[s]) * kSlabRecordWords * sizeof(float);
            x[i].recv_bytes[s] = (size_t)(ghosts ? r->gh_in[s] : r->mig_in[s]) * kSlabRecordWords * sizeof(float);
        }
    }
    return g.transport->exchange(g, x);
}

// C: payloads, appends, snapshot
static int phase_payload(SlabGroup &g)
{
    int rc;
    EACH(g, m)
    {
        std::swap(m->dev.cur, m->dev.nxt);              // the keepers were compacted into nxt
        std::swap(m->dev.id, m->dev.id_nxt);
        std::swap(m->dev.cellkey, m->dev.cellkey_nxt);
        m->dev.count = m->dev.n_owned = m->rank->n_keep;
        m->grid_valid = false;
    }
    if ((rc = exchange_records(g, false)) != SPH_OK) return rc;
    if ((rc = exchange_records(g, true)) != SPH_OK) return rc;
    EACH(g, m)
    {
        SlabRank *r = m->rank;
        G_HIP(g, hipSetDevice(m->device));
        const SlabArgs a = slab_args(m);
        uint32_t at = r->n_keep;
        for (int s = 0; s < 2; s++) {                   // migrants become owned; the tagged ones join the export list of their side
            launch_slab_append(m->dev, r->recv_mig[s], r->mig_in[s], at, r->send_src[s], r->gh_out[s], r->cap_side, a, s, m->stream);
            at += r->mig_in[s];
        }
        m->dev.n_owned = at;
        for (int s = 0; s < 2; s++) {                   // ghost block of a side: what the neighbour exported, then what I kept
            launch_slab_append(m->dev, r->recv_ghost[s], r->gh_in[s], at, nullptr, 0, 0, a, s, m->stream);
            at += r->gh_in[s];
            launch_slab_append(m->dev, r->kept[s], r->kept_n[s], at, nullptr, 0, 0, a, s, m->stream);
            at += r->kept_n[s];
        }
        m->dev.count = at;
        // the owned particles of the boundary planes are the ones refreshed for the neighbours: their number bounds the grid
        // of a two-pass sub-step's first pass (+1 block: the ranges start at any slot)
        for (int s = 0; s < 2; s++) m->dev.zone_blocks[s] = r->n_export[s] ? (r->n_export[s] + 255u) / 256u + 1u : 0u;
        G_HIP(g, hipGetLastError());
        // (the owned particles as the frame finds them -- what a re-run with a wider halo starts from -- are parked by
        // phase_compute once the grid build has read them: no copy)
        const size_t n = m->dev.n_owned;
        r->snap_n = (uint32_t)n;
        r->snap_substeps = m->substeps_since_upload;
        r->snap_dt_last = m->dt_last;
        r->snap_valid = true;
    }
    return SPH_OK;
}

// ghost values follow their owners: what = SPH_SLAB_UPDATE_RHO after the density pass, POSVEL after a sub-step.
// overlapped: the sub-step is being issued in two passes (do_velpos_begin done, do_velpos_end not yet): the refresh runs
// on the communication streams behind the first pass and reads / writes the arrays the sub-step is WRITING (nxt) -- the
// boundary targets' new values out, the ghosts' new values in; the second pass touches neither.
static int refresh_ghosts(SlabGroup &g, int what, bool overlapped)
{
    const size_t words = what == 0 ? 1 : kSlabPosVelWords;
    std::vector<Xfer> x(g.members.size());
    std::vector<Dev> view(g.members.size());
    for (size_t i = 0; i < g.members.size(); i++) {
        SphContext *m = g.members[i];
        SlabRank *r = m->rank;
        G_HIP(g, hipSetDevice(m->device));
        view[i] = m->dev;
        if (overlapped) {
            std::swap(view[i].cur, view[i].nxt);         // what Swap() will make current
            view[i].wild_v_cur ^= 1u;
            G_HIP(g, hipEventRecord(r->ev_zone_done, m->stream));
            G_HIP(g, hipStreamWaitEvent(m->comm_stream, r->ev_zone_done, 0));
        }
        const hipStream_t st = stream_of(m, overlapped);
        for (int s = 0; s < 2; s++) {
            launch_slab_pack_update(view[i], r->send_src[s], r->n_export[s], what, r->send_ghost[s], st);
            x[i].send[s] = r->send_ghost[s];
            x[i].recv[s] = r->recv_ghost[s];
            x[i].send_bytes[s] = (size_t)r->n_export[s] * words * sizeof(float);
            x[i].recv_bytes[s] = (size_t)r->n_ghost[s] * words * sizeof(float);
        }
        G_HIP(g, hipGetLastError());
    }
    int rc = g.transport->exchange(g, x, overlapped);
    if (rc != SPH_OK) return rc;
    for (size_t i = 0; i < g.members.size(); i++) {
        SphContext *m = g.members[i];
        SlabRank *r = m->rank;
        G_HIP(g, hipSetDevice(m->device));
        const hipStream_t st = stream_of(m, overlapped);
        launch_slab_unpack_update(view[i], m->dev.n_owned, r->n_ghost[0], what, r->recv_ghost[0], st);
        launch_slab_unpack_update(view[i], m->dev.n_owned + r->n_ghost[0], r->n_ghost[1], what, r->recv_ghost[1], st);
        G_HIP(g, hipGetLastError());
        if (overlapped) G_HIP(g, hipEventRecord(r->ev_refresh_done, m->comm_stream));
    }
    return SPH_OK;
}

// D: the reference's passes (SphFluidSimulation.cs:98-102) with the refreshes in between
static int phase_compute(SlabGroup &g, float delta_time)
{
    int rc;
    EACH(g, m)
    {
        G_HIP(g, hipSetDevice(m->device));
        m->dev.dt_ptr = nullptr;
        if ((rc = do_bucket_generation(m)) != SPH_OK) return rc;   // :98
        // The build gathered the particles into sorted order; the arrays it read (owned particles first, as the frame found
        // them) become the snapshot a re-run starts from, and the previous snapshot's arrays the new scratch set: three
        // sets rotate, nothing is copied.
        SlabRank *r = m->rank;
        std::swap(m->dev.nxt.x, r->snap_x);
        std::swap(m->dev.nxt.y, r->snap_y);
        std::swap(m->dev.nxt.z, r->snap_z);
        std::swap(m->dev.nxt.h4, r->snap_h4);
        std::swap(m->dev.cellkey_nxt, r->snap_ck);
        std::swap(m->dev.nxt.vw, r->snap_vw);
        std::swap(m->dev.nxt.rv, r->snap_rv);
        std::swap(m->dev.id_nxt, r->snap_id);
        if ((rc = do_density(m)) != SPH_OK) return rc;             // :99
    }
    if ((rc = refresh_ghosts(g, SPH_SLAB_UPDATE_RHO, false)) != SPH_OK) return rc;
    const uint32_t S = g.members[0]->params.substeps;
    // A sub-step whose results the neighbours need goes in TWO passes: the targets binned in the boundary planes first, then
    // -- while their new positions and velocities travel on the communication stream -- everybody else.
    const bool two_pass = (g.world > 1 || g.force_two_pass) && g.members[0]->dev.use_bricks != 0u && !g.single_pass;
    for (uint32_t i = 0; i < S; i++) {                             // :101
        const bool refresh = i + 1 < S;
        if (!(refresh && two_pass)) {
            EACH(g, m)
            {
                G_HIP(g, hipSetDevice(m->device));
                if ((rc = do_velpos(m, delta_time / m->params.dt_divisor)) != SPH_OK) return rc;   // :102
            }
            if (refresh && (rc = refresh_ghosts(g, SPH_SLAB_UPDATE_POSVEL, false)) != SPH_OK) return rc;
            continue;
        }
        EACH(g, m)
        {
            G_HIP(g, hipSetDevice(m->device));
            if ((rc = do_velpos_begin(m, delta_time / m->params.dt_divisor)) != SPH_OK) return rc;
        }
        if ((rc = refresh_ghosts(g, SPH_SLAB_UPDATE_POSVEL, true)) != SPH_OK) return rc;
        EACH(g, m)
        {
            G_HIP(g, hipSetDevice(m->device));
            if ((rc = do_velpos_end(m, delta_time / m->params.dt_divisor)) != SPH_OK) return rc;
            G_HIP(g, hipStreamWaitEvent(m->stream, m->rank->ev_refresh_done, 0));   // the next sub-step reads the ghosts
        }
    }
    return SPH_OK;
}

void launch_slab_restore(const Dev &d, uint32_t n, const Soa &snap, const uint32_t *snap_ck, const float4 *rv, const float *vw,
                         const uint32_t *id, hipStream_t s);   // sph_kernels_slab.hip

// The frame issued last broke its halo somewhere: every slab goes back to its snapshot and runs it again with a wider one.
static int rerun_previous(SlabGroup &g)
{
    if (g.halo_w >= g.max_halo)
        return gfail(g, SPH_E_COMM, "halo violation with %d ghost planes and slabs too thin for more (max %d): the result would differ "
                     "from the single-GPU run; use fewer devices or a smaller time step", g.halo_w, g.max_halo);
    g.halo_w = std::min(2 * g.halo_w, g.max_halo);
    g.reruns++;
    EACH(g, m)
    {
        SlabRank *r = m->rank;
        if (!r->snap_valid) return gfail(g, SPH_E_BAD_STATE, "no snapshot to re-run from");
        G_HIP(g, hipSetDevice(m->device));
        apply_halo(m, g.halo_w);
        m->dev.count = m->dev.n_owned = r->snap_n;
        G_HIP(g, hipMemsetAsync(m->dev.ghost, 0, m->capacity, m->stream));
        G_HIP(g, hipMemsetAsync(m->dev.counters + kCntWildV0 + m->dev.wild_v_cur, 0, sizeof(unsigned long long), m->stream));
        Soa snap{};
        snap.x = r->snap_x;
        snap.y = r->snap_y;
        snap.z = r->snap_z;
        snap.h4 = r->snap_h4;
        launch_slab_restore(m->dev, r->snap_n, snap, r->snap_ck, r->snap_rv, r->snap_vw, r->snap_id, m->stream);
        G_HIP(g, hipGetLastError());
        m->grid_valid = false;
        m->frames--;                                    // (the re-run counts it again)
        m->substeps -= m->params.substeps;
        m->substeps_since_upload = r->snap_substeps;
        m->dt_last = r->snap_dt_last;
    }
    int rc;
    unsigned long long total = 0;
    if ((rc = phase_classify(g)) != SPH_OK) return rc;   // everything keeps (the migration is in the snapshot); new ghost planes
    if ((rc = phase_sync(g, true, &total)) != SPH_OK) return rc;
    g.viol_seen = total;                                  // (the violations of the discarded run)
    if ((rc = read_counts(g)) != SPH_OK) return rc;
    if ((rc = phase_payload(g)) != SPH_OK) return rc;
    return phase_compute(g, g.delta_prev);
}

static int check_misplaced(SlabGroup &g)
{
    EACH(g, m)
    {
        if (m->rank->host->counters[kSlabMisplaced])
            return gfail(g, SPH_E_COMM, "slab %d: %u migrants could not be placed (slab thinner than twice the halo?)", m->rank->rank,
                         m->rank->host->counters[kSlabMisplaced]);
    }
    return SPH_OK;
}

// One Update() of every slab of the group.  Returns with the frame ENQUEUED (validated by the next frame or slab_group_finish).
int slab_group_step(SlabGroup *gp, float delta_time)
{
    SlabGroup &g = *gp;
    int rc;
    for (;;) {
        unsigned long long total = 0;
        if ((rc = phase_classify(g)) != SPH_OK) return rc;
        if ((rc = phase_sync(g, true, &total)) != SPH_OK) return rc;
        if (g.pending && total != g.viol_seen) {
            if ((rc = rerun_previous(g)) != SPH_OK) return rc;
            continue;                                   // classify again, on the repaired state
        }
        g.viol_seen = total;
        g.pending = false;
        break;
    }
    if ((rc = read_counts(g)) != SPH_OK) return rc;
    if ((rc = phase_payload(g)) != SPH_OK) return rc;
    if ((rc = phase_compute(g, delta_time)) != SPH_OK) return rc;
    g.pending = true;
    g.delta_prev = delta_time;
    return SPH_OK;
}

// Validates the frame in flight (collective: every slab's process calls it): after it the state is final.
int slab_group_finish(SlabGroup *gp)
{
    SlabGroup &g = *gp;
    int rc;
    while (g.pending) {
        unsigned long long total = 0;
        EACH(g, m)
        {
            G_HIP(g, hipSetDevice(m->device));
            G_HIP(g, hipMemcpyAsync(m->rank->viol_dev, m->dev.counters + kCntHaloViolation, sizeof(unsigned long long),
                                    hipMemcpyDeviceToDevice, m->stream));
        }
        if ((rc = phase_sync(g, false, &total)) != SPH_OK) return rc;
        if (total != g.viol_seen) {
            if ((rc = rerun_previous(g)) != SPH_OK) return rc;
            continue;
        }
        g.pending = false;
    }
    if ((rc = check_misplaced(g)) != SPH_OK) return rc;
    return SPH_OK;
}

// a new particle set: nothing in flight, nothing to re-run
void slab_group_reset(SlabGroup *g)
{
    g->pending = false;
    for (SphContext *m : g->members)
        if (m->rank) m->rank->snap_valid = false;
}

}  // namespace sph

// ---- several slabs in ONE process behind the ordinary calls (SphParams.num_devices / device_ids) -------------------------

using namespace sph;

// host-side twin of bucket_key() (sph_internal.hpp): the plane of a particle's BUCKET ROW, Bucket.compute:26-29; -1 outside
static inline int host_hlsl_int(float f)
{
    if (f != f) return 0;
    if (f <= -2147483648.0f) return INT_MIN;
    if (f >= 2147483648.0f) return INT_MAX;
    return (int)f;
}
static int host_bucket_plane(float x, float y, float z, uint32_t R)
{
    const float Rm1 = (float)(R - 1u), Rf = (float)R;
    const float ix = (float)host_hlsl_int(x * Rm1), iy = (float)host_hlsl_int(y * Rm1), iz = (float)host_hlsl_int(z * Rm1);
    const float fi = (ix + iy * Rf) + iz * Rf * Rf;
    uint32_t index = !(fi > 0.0f) ? 0u : (fi >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)fi);
    index *= kMaxPerVoxel;
    const unsigned long long len = (unsigned long long)R * R * R * kMaxPerVoxel;
    if ((unsigned long long)index >= len) return -1;
    return (int)((index / kMaxPerVoxel) / (R * R));
}

int group_create(const SphParams *params, SphContext **out)
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, SPH_E_HIP, "no HIP device available (%s); libsph_hip has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    const int world = (int)params->num_devices;
    for (int i = 0; i < world; i++)
        if (params->device_ids[i] < 0 || params->device_ids[i] >= ndev)
            return fail(nullptr, SPH_E_INVALID_ARG, "device_ids[%d] = %d not in [0,%d)", i, params->device_ids[i], ndev);
    if ((uint32_t)world > params->bucket_resolution)
        return fail(nullptr, SPH_E_INVALID_ARG, "%d slabs for %u grid planes", world, params->bucket_resolution);
    SphContext *c = new SphContext();
    c->params = *params;
    c->is_group = true;
    c->slab = false;
    derive_params(c->params, &c->derived);
    for (int i = 0; i < world; i++) c->member_devices.push_back(params->device_ids[i]);
    *out = c;
    return SPH_OK;
}

static void group_drop_members(SphContext *c)
{
    if (c->group) slab_group_destroy(c->group);
    c->group = nullptr;
    for (SphContext *m : c->members) {
        m->group = nullptr;
        sph_destroy(m);
    }
    c->members.clear();
}

int group_destroy(SphContext *c)
{
    group_drop_members(c);
    delete c;
    return SPH_OK;
}

// (Re)creates the member slabs once the particle distribution is known: capacities follow the owned counts.
static int group_build(SphContext *c, const std::vector<uint32_t> &owned, const std::vector<uint32_t> &plane_hist)
{
    const int world = (int)c->member_devices.size();
    const int R = (int)c->params.bucket_resolution;
    const uint32_t n_global = c->derived.particle_count;
    std::vector<uint32_t> cap(world);
    bool fits = (int)c->members.size() == world;
    for (int r = 0; r < world; r++) {
        int lo, hi;
        slab_planes(R, world, r, &lo, &hi);
        // ghosts: up to 8 planes either side (the halo may grow), from the measured plane populations
        uint64_t ghosts = 0;
        for (int z = std::max(0, lo - 8); z < std::min(R, hi + 8); z++)
            if (z < lo || z >= hi) ghosts += plane_hist[z];
        const uint64_t want = (uint64_t)((double)(owned[r] + ghosts) * 1.5) + 4096;
        cap[r] = (uint32_t)std::min<uint64_t>(want, (uint64_t)n_global + 64);
        if (fits && c->members[r]->capacity < owned[r] + ghosts) fits = false;
    }
    if (fits) return SPH_OK;
    group_drop_members(c);
    std::vector<SphContext *> ms;
    for (int r = 0; r < world; r++) {
        SphParams p = c->params;
        p.num_devices = 0;
        p.device_id = c->member_devices[r];
        SphContext *m = nullptr;
        int rc = slab_context_create(&p, cap[r], &m);
        if (rc == SPH_OK) rc = slab_configure(m, r, world);
        if (rc != SPH_OK) {
            c->last_error = m ? m->last_error : g_last_error;
            if (m) sph_destroy(m);
            for (SphContext *q : ms) sph_destroy(q);
            return rc;
        }
        ms.push_back(m);
    }
    c->members = ms;
    c->group = slab_group_new(c, ms, world);
    c->owns_group = true;
    bool distinct = true;
    for (int a = 0; a < world; a++)
        for (int b = a + 1; b < world; b++) distinct = distinct && c->member_devices[a] != c->member_devices[b];
    // RCCL needs one device per rank; repeated devices (one-GPU rehearsal) and RCCL-less systems copy between the slabs' buffers
    int rc = distinct && !(c->params.flags & SPH_FLAG_PEER_COPIES) ? slab_group_use_rccl_all(c->group) : SPH_E_COMM;
    if (rc != SPH_OK) {
        c->last_error.clear();
        rc = slab_group_use_peer_copies(c->group);
    }
    return rc;
}

int group_upload(SphContext *c, const float *pos4, const float *vel4)
{
    if (!pos4) return fail(c, SPH_E_INVALID_ARG, "pos4 is NULL");
    const int world = (int)c->member_devices.size();
    const uint32_t R = c->params.bucket_resolution, n = c->derived.particle_count;
    std::vector<int> lo(world), hi(world);
    for (int r = 0; r < world; r++) slab_planes((int)R, world, r, &lo[r], &hi[r]);
    std::vector<uint8_t> owner(n);
    std::vector<uint32_t> owned(world, 0), hist(R, 0);
    for (uint32_t i = 0; i < n; i++) {
        const int cz = host_bucket_plane(pos4[4 * (size_t)i], pos4[4 * (size_t)i + 1], pos4[4 * (size_t)i + 2], R);
        int r = 0;                                      // rows outside the bucket stay with slab 0
        if (cz >= 0) {
            hist[cz]++;
            r = (int)(std::upper_bound(hi.begin(), hi.end(), cz) - hi.begin());
        }
        owner[i] = (uint8_t)r;
        owned[r]++;
    }
    int rc = group_build(c, owned, hist);
    if (rc != SPH_OK) return rc;
    slab_group_reset(c->group);
    for (int r = 0; r < world; r++) {
        std::vector<float> p((size_t)owned[r] * 4), v(vel4 ? (size_t)owned[r] * 4 : 0);
        std::vector<uint32_t> ids(owned[r]);
        size_t k = 0;
        for (uint32_t i = 0; i < n; i++) {
            if (owner[i] != r) continue;
            memcpy(&p[4 * k], &pos4[4 * (size_t)i], 16);
            if (vel4) memcpy(&v[4 * k], &vel4[4 * (size_t)i], 16);
            ids[k++] = i;
        }
        rc = sph_slab_upload(c->members[r], owned[r], p.data(), vel4 ? v.data() : nullptr, ids.data());
        if (rc != SPH_OK) {
            c->last_error = c->members[r]->last_error;
            return rc;
        }
    }
    c->uploaded = true;
    return SPH_OK;
}

int group_step(SphContext *c, float delta_time)
{
    if (!c->uploaded) return fail(c, SPH_E_BAD_STATE, "sph_step before sph_upload");
    int rc = slab_group_step(c->group, delta_time);
    if (rc == SPH_OK) {
        c->frames++;
        c->substeps += c->params.substeps;
    }
    return rc;
}

int group_synchronize(SphContext *c)
{
    if (!c->group) return SPH_OK;
    int rc = slab_group_finish(c->group);
    if (rc != SPH_OK) return rc;
    for (SphContext *m : c->members)
        if ((rc = sph_synchronize(m)) != SPH_OK) return rc;
    return SPH_OK;
}

int group_download(SphContext *c, float *pos4, float *vel4, float *rho)
{
    if (!c->uploaded) return fail(c, SPH_E_BAD_STATE, "sph_download before sph_upload");
    int rc = slab_group_finish(c->group);
    if (rc != SPH_OK) return rc;
    const uint32_t n = c->derived.particle_count;
    uint64_t seen = 0;
    for (SphContext *m : c->members) {
        const uint32_t cap = m->capacity;
        std::vector<uint32_t> ids(cap);
        std::vector<float> p((size_t)cap * 4), v((size_t)cap * 4), r(cap);
        uint32_t k = 0;
        rc = sph_slab_download(m, cap, ids.data(), p.data(), v.data(), r.data(), &k);
        if (rc != SPH_OK) {
            c->last_error = m->last_error;
            return rc;
        }
        for (uint32_t j = 0; j < k; j++) {
            const uint32_t id = ids[j];
            if (id >= n) return fail(c, SPH_E_COMM, "particle id %u out of range after the exchange", id);
            if (pos4) memcpy(&pos4[4 * (size_t)id], &p[4 * (size_t)j], 16);
            if (vel4) memcpy(&vel4[4 * (size_t)id], &v[4 * (size_t)j], 16);
            if (rho) rho[id] = r[j];
        }
        seen += k;
    }
    if (seen != n) return fail(c, SPH_E_COMM, "particle ownership broken: the slabs hold %llu of %u particles", (unsigned long long)seen, n);
    return SPH_OK;
}

int group_get_stats(SphContext *c, SphStats *out)
{
    memset(out, 0, sizeof *out);
    out->frames = c->frames;
    out->substeps = c->substeps;
    if (!c->group) return SPH_OK;
    int rc = slab_group_finish(c->group);
    if (rc != SPH_OK) return rc;
    for (SphContext *m : c->members) {
        SphStats s;
        if ((rc = engine_get_stats(m, &s)) != SPH_OK) {
            c->last_error = m->last_error;
            return rc;
        }
        // times: the slowest slab (they run side by side); counts: summed
        out->ms_grid = std::max(out->ms_grid, s.ms_grid);
        out->ms_density = std::max(out->ms_density, s.ms_density);
        out->ms_velpos = std::max(out->ms_velpos, s.ms_velpos);
        out->ms_velpos_list = std::max(out->ms_velpos_list, s.ms_velpos_list);
        out->launches_grid = std::max(out->launches_grid, s.launches_grid);
        out->launches_density = std::max(out->launches_density, s.launches_density);
        out->launches_velpos = std::max(out->launches_velpos, s.launches_velpos);
        out->launches_velpos_list = std::max(out->launches_velpos_list, s.launches_velpos_list);
        out->max_cell_occupancy = std::max(out->max_cell_occupancy, s.max_cell_occupancy);
        out->overflowed_cells += s.overflowed_cells;
        out->dropped_particles += s.dropped_particles;
        out->queued_targets += s.queued_targets;
        out->nan_resets += s.nan_resets;
        out->candidates_density += s.candidates_density;
        out->halo_violations += s.halo_violations;
        out->listed_target_substeps += s.listed_target_substeps;
        out->scanned_brick_substeps += s.scanned_brick_substeps;
        out->list_entries += s.list_entries;
        out->frames_without_lists = std::max(out->frames_without_lists, s.frames_without_lists);
    }
    out->host_syncs = c->host_syncs;
    out->frames_rerun = slab_group_reruns(c->group);
    out->halo_planes = (uint64_t)slab_group_halo(c->group);
    return SPH_OK;
}

int group_reset_stats(SphContext *c)
{
    c->host_syncs = 0;
    for (SphContext *m : c->members) {
        int rc = sph_reset_stats(m);
        if (rc != SPH_OK) return rc;
    }
    return SPH_OK;
}

int group_set_params(SphContext *c, const SphParams *p)
{
    c->params = *p;
    for (size_t r = 0; r < c->members.size(); r++) {
        SphParams q = *p;
        q.num_devices = 0;
        q.device_id = c->member_devices[r];
        int rc = sph_set_params(c->members[r], &q);
        if (rc != SPH_OK) {
            c->last_error = c->members[r]->last_error;
            return rc;
        }
    }
    return SPH_OK;
}

// ---- C ABI of the one-process-per-GPU mode (include/sph_hip.h) ---------------------------------------------------------------

#define SLAB_CHECK(c)                                                                     \
    if (!c) return fail(nullptr, SPH_E_INVALID_ARG, "ctx is NULL");                       \
    if (!c->slab) return fail(c, SPH_E_BAD_STATE, "not a slab context (sph_slab_create)"); \
    HIP_TRY(c, hipSetDevice(c->device))

extern "C" {

int sph_slab_create(const SphParams *params, uint32_t local_capacity, SphContext **out)
{
    if (local_capacity == 0) return fail(nullptr, SPH_E_INVALID_ARG, "local_capacity is 0");
    if (params && params->struct_size == sizeof(SphParams) && params->num_devices > 1)
        return fail(nullptr, SPH_E_INVALID_ARG, "a slab context is one device: num_devices must be 0 or 1");
    return slab_context_create(params, local_capacity, out);
}

int sph_slab_planes(uint32_t bucket_resolution, int world, int rank, int *z_lo, int *z_hi)
{
    if (!z_lo || !z_hi || world < 1 || rank < 0 || rank >= world || bucket_resolution < 2)
        return fail(nullptr, SPH_E_INVALID_ARG, "sph_slab_planes: bad argument");
    slab_planes((int)bucket_resolution, world, rank, z_lo, z_hi);
    return SPH_OK;
}

int sph_slab_configure(SphContext *c, int rank, int world)
{
    SLAB_CHECK(c);
    int rc = slab_configure(c, rank, world);
    if (rc != SPH_OK) return rc;
    c->group = slab_group_new(c, std::vector<SphContext *>{c}, world);
    c->owns_group = true;
    return SPH_OK;
}

int sph_comm_unique_id(void *id_bytes)
{
    if (!id_bytes) return fail(nullptr, SPH_E_INVALID_ARG, "id_bytes is NULL");
    return slab_rccl_unique_id(id_bytes);
}

int sph_slab_comm_init(SphContext *c, const void *id_bytes)
{
    SLAB_CHECK(c);
    if (!c->group) return fail(c, SPH_E_BAD_STATE, "sph_slab_comm_init before sph_slab_configure");
    if (!id_bytes) return fail(c, SPH_E_INVALID_ARG, "id_bytes is NULL");
    return slab_group_use_rccl_rank(c->group, id_bytes);
}

int sph_slab_set_host_transport(SphContext *c, SphHostExchangeFn exchange, SphHostAllreduceFn allreduce, void *user)
{
    SLAB_CHECK(c);
    if (!c->group) return fail(c, SPH_E_BAD_STATE, "sph_slab_set_host_transport before sph_slab_configure");
    if (!exchange || !allreduce) return fail(c, SPH_E_INVALID_ARG, "callback is NULL");
    return slab_group_use_host(c->group, exchange, allreduce, user);
}

int sph_slab_upload(SphContext *c, uint32_t n, const float *pos4, const float *vel4, const uint32_t *ids)
{
    SLAB_CHECK(c);
    if (!c->rank) return fail(c, SPH_E_BAD_STATE, "sph_slab_upload before sph_slab_configure");
    if (n > c->capacity) return fail(c, SPH_E_INVALID_ARG, "%u particles exceed the local capacity %u", n, c->capacity);
    if (n && (!pos4 || !ids)) return fail(c, SPH_E_INVALID_ARG, "pos4 / ids is NULL");
    HIP_TRY(c, hipMemcpyAsync(c->pos4_dev, pos4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    if (vel4) HIP_TRY(c, hipMemcpyAsync(c->vel4_dev, vel4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->ids_dev, ids, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->dev.counters + kCntWildP, 0, sizeof(unsigned long long) * 3, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->dev.ghost, 0, c->capacity, c->stream));
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
    if (c->group && c->owns_group) slab_group_reset(c->group);
    SlabRank *r = c->rank;
    r->n_keep = n;
    for (int s = 0; s < 2; s++) r->n_export[s] = r->n_ghost[s] = 0;
    return SPH_OK;
}

int sph_slab_step(SphContext *c, float delta_time)
{
    SLAB_CHECK(c);
    if (!c->group || !c->owns_group) return fail(c, SPH_E_BAD_STATE, "sph_slab_step: configure the slab first (members of a multi-device context step with it)");
    return slab_group_step(c->group, delta_time);
}

int sph_slab_finish(SphContext *c)
{
    SLAB_CHECK(c);
    if (!c->group) return SPH_OK;
    return slab_group_finish(c->group);
}

int sph_slab_counts(SphContext *c, uint32_t out[5])
{
    SLAB_CHECK(c);
    if (!out) return fail(c, SPH_E_INVALID_ARG, "out is NULL");
    if (!c->rank) return fail(c, SPH_E_BAD_STATE, "sph_slab_counts before sph_slab_configure");
    out[0] = c->dev.n_owned;
    out[1] = c->rank->n_ghost[0];
    out[2] = c->rank->n_ghost[1];
    out[3] = c->rank->n_export[0];
    out[4] = c->rank->n_export[1];
    return SPH_OK;
}

int sph_slab_download(SphContext *c, uint32_t cap, uint32_t *ids, float *pos4, float *vel4, float *rho, uint32_t *n_out)
{
    SLAB_CHECK(c);
    if (!ids || !pos4 || !vel4 || !rho || !n_out) return fail(c, SPH_E_INVALID_ARG, "NULL argument");
    if (c->group) {
        int rc = slab_group_finish(c->group);
        if (rc != SPH_OK) return rc;
    }
    HIP_TRY(c, hipMemsetAsync(c->download_counter, 0, 4 * sizeof(uint32_t), c->stream));
    launch_slab_download(c->dev, c->substeps_since_upload, c->dt_last, c->download_counter, c->ids_dev, c->pos4_dev,
                         c->vel4_dev, c->rho_id_dev, c->stream);
    HIP_TRY(c, hipGetLastError());
    uint32_t n = 0;
    HIP_TRY(c, hipMemcpyAsync(&n, c->download_counter, sizeof n, hipMemcpyDeviceToHost, c->stream));
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

int sph_runtime_info(SphRuntimeInfo *out)
{
    if (!out || out->struct_size != sizeof(SphRuntimeInfo))
        return fail(nullptr, SPH_E_INVALID_ARG, "SphRuntimeInfo NULL or struct_size != %zu", sizeof(SphRuntimeInfo));
    memset(out, 0, sizeof *out);
    out->struct_size = sizeof(SphRuntimeInfo);
    (void)hipRuntimeGetVersion(&out->hip_runtime_version);
    (void)hipDriverGetVersion(&out->hip_driver_version);
    if (hipGetDeviceCount(&out->device_count) != hipSuccess) out->device_count = 0;
    (void)hipGetLastError();
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(static_cast<hipError_t (*)(void **, size_t)>(&hipMalloc)), &info) && info.dli_fname)
        snprintf(out->hip_library_path, sizeof out->hip_library_path, "%s", info.dli_fname);
    (void)slab_rccl_info(out->rccl_library_path, sizeof out->rccl_library_path, &out->rccl_version);
    return SPH_OK;
}

int sph_selftest_arithmetic(int device_id, uint64_t n, uint64_t seed, uint64_t mismatches[4])
{
    if (!mismatches) return fail(nullptr, SPH_E_INVALID_ARG, "mismatches is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, SPH_E_HIP, "no HIP device available");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, SPH_E_INVALID_ARG, "device_id %d not in [0,%d)", device_id, ndev);
    HIP_TRY((SphContext *)nullptr, hipSetDevice(device_id));
    unsigned long long *d = nullptr;
    HIP_TRY((SphContext *)nullptr, hipMalloc((void **)&d, 4 * sizeof(unsigned long long)));
    hipError_t e = hipMemset(d, 0, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) {
        launch_selftest_arithmetic(n, seed, d, nullptr);
        e = hipGetLastError();
    }
    unsigned long long h[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(nullptr, SPH_E_HIP, "sph_selftest_arithmetic: %s", hipGetErrorString(e));
    for (int i = 0; i < 4; i++) mismatches[i] = h[i];
    return SPH_OK;
}

}  // extern "C"
