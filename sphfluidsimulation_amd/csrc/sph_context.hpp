// Host-side context of libsph_hip.so, shared by sph_engine.hip (single-GPU schedule, C ABI) and sph_slab_frame.hip
// (z-slab frames: one slab per device, halo exchange).
#pragma once

#include "../../include/sph_hip.h"
#include "sph_internal.hpp"

#include <stdarg.h>
#include <stdio.h>

#include <string>
#include <vector>

namespace sph {
struct SlabGroup;   // sph_slab_frame.hip: the slabs that step together + their transport
struct SlabRank;    // sph_slab_frame.hip: exchange buffers and frame bookkeeping of one slab
}
using namespace sph;

extern thread_local std::string g_last_error;

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
        bool fork_scan = false;       // the scanning kernel was captured on the side stream
    } graph[2];
    bool graphs_enabled = false;
    float *dt_dev = nullptr;          // delta_time / dt_divisor of the frame being replayed (device scalar)
    uint64_t frames_since_upload = 0, graph_replays = 0;
    // frame-coherent neighbour lists (sph_kernels_list.hip)
    bool nl_enabled = false, nl_allocated = false;
    uint32_t nl_cap_want = 64;        // entries per list row (grown when frames report longer lists, grow_list_rows)
    unsigned long long nl_hold_until = 0;   // frames issued when the rows were last re-allocated
    bool nl_unavailable = false;      // the list buffers could not be allocated once: never tried again
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
    double pol_min_frac = 0.9;        // lists stay on while they serve at least this share of the target sub-steps
    uint64_t pol_backoff = 8, pol_resume_at = 0, pol_frames_off = 0;
    float nl_skin_frac = 0.04f;       // skin / h (profiles/: 0.04-0.06 is the optimum of the 16.7 M dam break)
    float4 *mesh_props = nullptr;     // _particleMeshPropertiesBuffer (:174), allocated on first use
    bool mesh_valid = false;
    // row f4: the same buffer in exportable device memory (HIP virtual-memory allocation), once a caller asked for its fd
    hipMemGenericAllocationHandle_t mesh_handle{};
    void *mesh_va = nullptr;
    size_t mesh_va_size = 0;
    // ... and the staging textures (position, velocity, density in the reference's texture layout), once asked for
    struct ExportBuf {
        hipMemGenericAllocationHandle_t handle{};
        void *va = nullptr;
        size_t size = 0;
    } state_export[3];
    hipExternalSemaphore_t ext_sem = nullptr;   // a renderer's semaphore, signalled on the stream (sph_import_semaphore_fd)
    bool ext_sem_timeline = false;
    bool uploaded = false, grid_valid = false;
    int substeps_since_upload = 0;
    int substep_in_frame = 0;         // sub-steps issued since the last grid build (parity of the per-brick displacement boxes)
    float dt_last = 0.0f;
    uint64_t frames = 0, substeps = 0;
    // profiling
    bool profile = false;
    std::vector<PassEvents> pending;
    std::vector<hipEvent_t> free_events;
    double ms[4] = {0, 0, 0, 0};
    uint64_t launches[4] = {0, 0, 0, 0};
    std::string last_error;
    // z-slab (multi-GPU) mode: this context is ONE slab (sph_slab_create, or a member of an in-process group) ...
    bool slab = false;
    uint32_t capacity = 0;            // particles the device arrays can hold
    int z_lo = 0, z_hi = 0, halo_w = 2, has_left = 0, has_right = 0;
    SlabRank *rank = nullptr;         // exchange buffers, counts, snapshot (allocated by sph_slab_configure)
    hipStream_t side_stream = nullptr;   // plain contexts: the scanning kernel of a list-using sub-step runs here, beside the list kernels
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool fork_scan = false;              // ... while the last reported frame had many bricks that were not calm (update_scan_policy)
    unsigned long long scan_seen[2] = {0, 0};
    hipStream_t comm_stream = nullptr;   // the halo refresh of a two-pass sub-step travels here while `stream` computes
    SlabGroup *group = nullptr;       // the group that steps this slab (owned by the group context, or by this context when it
                                      // is the only local member: one process per GPU)
    bool owns_group = false;
    uint32_t *ids_dev = nullptr;
    uint32_t *download_counter = nullptr;   // 4 words, k_slab_download
    // ... or the in-process GROUP itself (sph_create with num_devices >= 2): no device state of its own
    bool is_group = false;
    std::vector<SphContext *> members;
    std::vector<int> member_devices;
    uint64_t host_syncs = 0;          // host synchronisations issued by slab frames since the last sph_reset_stats
};

// A new particle set: the list policy starts over.  Prior: in a box of fewer than ~100 cells per axis most bricks lie
// within reach of the wall layer, whose particles the wall force scatters by more than any useful skin -- such scenes
// start without lists and probe late; large boxes start with lists.
inline void reset_list_policy(SphContext *c)
{
    const bool big = c->dev.R >= 100u;
    c->pol_min_frac = big ? 0.5 : 0.9;
    c->pol_on = big || (c->params.flags & SPH_FLAG_LISTS_ALWAYS);
    c->pol_confirmed = false;
    c->pol_low_streak = c->pol_high_streak = 0;
    c->pol_backoff = big ? 8 : 32;
    c->pol_resume_at = c->frames + (c->pol_on ? 0 : c->pol_backoff);
}

inline int fail(SphContext *ctx, int code, const char *fmt, ...)
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


// sph_engine.hip
int do_bucket_generation(SphContext *c);
int do_density(SphContext *c);
int do_velpos(SphContext *c, float dt);
int do_velpos_begin(SphContext *c, float dt);   // boundary-plane targets (two-pass sub-step of a z-slab)
int do_velpos_end(SphContext *c, float dt);     // all the others; Swap()
void refresh_scalars(SphContext *c);
int engine_get_stats(SphContext *c, SphStats *out);   // sph_get_stats of ONE device context
int slab_context_create(const SphParams *params, uint32_t capacity, SphContext **out);
void derive_params(const SphParams &p, SphDerived *d);
int validate_params(SphContext *ctx, const SphParams *p);

// sph_slab_frame.hip
namespace sph {
void slab_planes(int R, int world, int rank, int *z_lo, int *z_hi);
int slab_configure(SphContext *c, int rank, int world);
void slab_rank_destroy(SphContext *c);
SlabGroup *slab_group_new(SphContext *owner, std::vector<SphContext *> members, int world);
void slab_group_destroy(SlabGroup *g);
int slab_group_step(SlabGroup *g, float delta_time);
int slab_group_finish(SlabGroup *g);
void slab_group_reset(SlabGroup *g);
uint64_t slab_group_reruns(SlabGroup *g);
int slab_group_halo(SlabGroup *g);
}
int group_create(const SphParams *params, SphContext **out);
int group_destroy(SphContext *c);
int group_upload(SphContext *c, const float *pos4, const float *vel4);
int group_download(SphContext *c, float *pos4, float *vel4, float *rho);
int group_step(SphContext *c, float delta_time);
int group_synchronize(SphContext *c);
int group_get_stats(SphContext *c, SphStats *out);
int group_reset_stats(SphContext *c);
int group_set_params(SphContext *c, const SphParams *p);
