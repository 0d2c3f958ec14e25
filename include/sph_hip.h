/*
 * sph_hip.h -- C ABI of libsph_hip.so, the MI355X-native (gfx950) replacement for ONE path of
 * leandro-barcelos/SPHFluidSimulation: the per-frame particle update that
 * Assets/Scripts/SphFluidSimulation.cs::Update (:96-102) dispatches against
 * Assets/Resources/{Clear,Bucket,Density,VelPos}.compute.
 *
 * The reference has NO plugin/FFI boundary: the C# MonoBehaviour talks to the GPU through Unity's
 * ComputeShader API (Resources.Load :128-133, SetInt/SetFloat/SetVector/SetTexture/SetBuffer +
 * Dispatch :206-220, :229-240, :245-265).  This header is what a C# P/Invoke binding
 * ([DllImport("sph_hip")], CallingConvention.Cdecl -- see INTEGRATION.md) binds instead.  Each
 * entry point names the reference interface it replaces.
 *
 * Conventions
 *   - every call returns SPH_OK (0) or a negative SPH_E_* code; text via sph_last_error();
 *     nothing throws or aborts across the ABI.  NaN accelerations, overflowing cells and
 *     out-of-grid particles are COUNTED (SphStats), not errors -- reference semantics are kept.
 *   - host arrays are the reference's texture memory: float4 per particle, row-major
 *     res x res, particle id = y*res + x (Bucket.compute:24, VelPos.compute:55).
 *     pos.w == 1 and vel.w == sub-step counter / 0.003 NaN marker (VelPos.compute:146,150-154).
 *   - the engine owns all device memory (cell-sorted SoA, ping-pong); no host pointer is kept
 *     after a call returns.
 *   - a context is not thread-safe (the reference is driven from Unity's main thread only).
 *     sph_step, sph_step_n, sph_bucket_generation, ... enqueue on the context's HIP stream and return
 *     without a host sync; sph_download, sph_get_stats and sph_synchronize synchronise.
 *   - there is NO CPU fallback: without a gfx950 device sph_create fails with SPH_E_HIP.
 */
#ifndef SPH_HIP_H
#define SPH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPH_ABI_VERSION 2

#define SPH_OK 0
#define SPH_E_INVALID_ARG (-1)
#define SPH_E_BAD_STATE (-2)
#define SPH_E_OOM (-3)
#define SPH_E_HIP (-4)
#define SPH_E_UNSUPPORTED (-5)
#define SPH_E_COMM (-6)

/*
 * The reference has no SimulationParameters type; this blittable struct (4-byte members only,
 * LayoutKind.Sequential on the C# side) carries the inspector fields
 * SphFluidSimulation.cs:35-45 plus the constants the reference hard-codes.
 * Derived uniforms (h, h^2, h^6, h^9, m, res) are computed inside with the reference's own
 * recipe (SphFluidSimulation.cs:84-85,159,176,236-237,255) and are never passed in.
 */
typedef struct SphParams {
    uint32_t struct_size;       /* = sizeof(SphParams); guards against ABI drift                  */
    uint32_t num_particles;     /* particleNumber :36 (raw; NextPowerOfTwo + (int)sqrt applied :84-85) */
    uint32_t bucket_resolution; /* bucketResolution :37, 2..256                                     */
    float dam_fill_rate;        /* damFillRate :38 (only enters through m = fill / N, :176)         */
    float viscosity;            /* viscosity :41            -> _Viscosity                           */
    float rest_density;         /* restDensity :42          -> _RestDensity                         */
    float gas_constant;         /* gasConstant :43          -> _GasConst                            */
    float stiffness_coeff;      /* stiffnessCoefficient :44 -> _StiffnessCoeff                      */
    float damping_coeff;        /* dampingCoefficient :45   -> _DampingCoeff                        */
    uint32_t max_per_voxel;     /* MaxParticlesPerVoxel :9 ; must be 32                             */
    uint32_t substeps;          /* 5  (:101)                                                        */
    float dt_divisor;           /* 25 (:102)                                                        */
    float gravity[3];           /* (0,-9.8,0) VelPos.compute:7                                      */
    uint32_t precision;         /* 0 = fp32 state (bit-identical to the CPU oracle); 1 = positions stored as fp16 offsets
                                   from the centre of the particle's grid cell (8 B instead of 12), all arithmetic and the
                                   velocities fp32 -- BASELINE config 5; within 1e-4 of the fp32 result after a frame         */
    uint32_t exact_order;       /* 1 = the reference's arithmetic, IEEE operation by operation: bit-identical to the CPU oracle.
                                   0 = TOLERANCE MODE: same pairs, guards and summation order, but v_rcp_f32 / v_sqrt_f32
                                   (1 ulp) instead of IEEE division / sqrt and re-associated products; within BASELINE.json's
                                   1e-4 of the oracle after a frame (tests/test_gpu_tolerance_mode.py)                 */
    int32_t device_id;          /* HIP device ordinal (num_devices <= 1)                            */
    uint32_t flags;             /* SPH_FLAG_*                                                       */
    /* ---- ABI version 2 ---- */
    uint32_t num_devices;       /* 0 or 1: one GPU (device_id).  2..8: the particle set is split into that many z-slabs
                                   inside THIS process, slab i on device_ids[i], halo exchange over RCCL/xGMI from inside
                                   the library; sph_upload / sph_step / sph_download / sph_get_stats work unchanged      */
    int32_t device_ids[8];      /* HIP ordinals of the slabs' devices, ascending z.  Repeated ordinals are allowed (several
                                   slabs on one GPU: rehearsal; the exchange is then a device copy, not RCCL)              */
    uint32_t halo_planes;       /* ghost planes per slab face to start with (0 = default 2); doubled automatically, and the
                                   frame re-run, whenever a particle outruns them (SphStats.frames_rerun)                 */
    float list_skin;            /* neighbour-list radius = h * (1 + list_skin); 0 = default 0.04 (tuning; any value in
                                   (0, 1] gives the same results)                                                          */
} SphParams;

#define SPH_FLAG_PROFILE 1u     /* record HIP events around every pass (SphStats.ms_*)            */
#define SPH_FLAG_NO_TILING 2u   /* diagnostic: per-target kernels on global memory, no LDS bricks */
#define SPH_FLAG_NO_LISTS 4u    /* diagnostic: every sub-step re-scans its 27 bins (no frame-coherent neighbour lists) */
#define SPH_FLAG_LISTS_ALWAYS 16u /* diagnostic: record neighbour lists every frame (default: only while they pay, see
                                    SphStats.frames_without_lists) */
#define SPH_FLAG_NO_OVERLAP 64u  /* z-slab mode, diagnostic: refresh the ghosts after the whole sub-step instead of behind its boundary pass */
#define SPH_FLAG_FORCE_TWO_PASS 128u /* z-slab mode, diagnostic: two-pass sub-steps even with one slab (measures what the second pass costs) */
#define SPH_FLAG_PEER_COPIES 32u /* multi-device context: move the halo with peer-to-peer device copies instead of RCCL */
#define SPH_FLAG_GLOBAL_LISTS 256u /* diagnostic: the brick kernels record lists of global slots and k_velpos_list gathers the
                                      candidates through L1/L2 (rounds 1-2) instead of lists of LDS tile indices walked by
                                      k_velpos_tile; same bits, slower */
#define SPH_FLAG_TILE_SLOTS_ONE 512u  /* diagnostic: k_velpos_tile compiled for whole-brick records only (default: picked from the particles per cell) */
#define SPH_FLAG_TILE_SLOTS_TWO 1024u /* diagnostic: k_velpos_tile compiled for the records of halved bricks too (config C4's regime) */
#define SPH_FLAG_CHUNK_BOXES 2048u   /* diagnostic: displacement ranges per 64 slots from a pass over all positions per sub-step (k_chunk_box,
                                        rounds 1-2) instead of per-brick ranges accumulated where targets are integrated */
#define SPH_FLAG_NO_SIDE_STREAM 4096u /* diagnostic: the scanning kernel of a sub-step in stream order instead of beside the list kernels */
#define SPH_FLAG_NO_GRAPHS 8u   /* diagnostic: sph_step issues its ~60 launches every frame instead of replaying a HIP graph */

/* Values of the uniforms the reference pushes through ShaderIDs.cs:5-32. */
typedef struct SphDerived {
    uint32_t num_particles;       /* _NumParticles      (sentinel + mass divisor)   */
    uint32_t particle_resolution; /* _ParticleResolution (res)                       */
    uint32_t particle_count;      /* res*res = number of live particles              */
    uint32_t bucket_resolution;   /* _BucketResolution                               */
    float particle_mass;          /* _ParticleMass                                   */
    float effective_radius;       /* _EffectiveRadius                                */
    float effective_radius2;      /* _EffectiveRadius2                               */
    float effective_radius6;      /* _EffectiveRadius6                               */
    float effective_radius9;      /* _EffectiveRadius9                               */
} SphDerived;

typedef struct SphStats {
    uint64_t frames;             /* sph_step frames (or BucketGeneration calls) since create     */
    uint64_t substeps;           /* UpdateVelocityAndPosition passes since create                */
    /* GPU time per pass, accumulated since the last sph_reset_stats (0 unless SPH_FLAG_PROFILE) */
    double ms_grid;              /* cell keys + counting sort + reorder (replaces Bucket.compute) */
    double ms_density;           /* Density.compute::CSMain replacement                          */
    double ms_velpos;            /* VelPos.compute::CSMain replacement, all sub-steps            */
    uint64_t launches_grid;      /* number of timed grid builds                                  */
    uint64_t launches_density;
    uint64_t launches_velpos;
    /* state of the LAST grid build */
    uint32_t max_cell_occupancy; /* before the 32-cap                                            */
    uint32_t overflowed_cells;   /* cells holding more than max_per_voxel particles              */
    uint32_t dropped_particles;  /* linear cell index outside the bucket         */
    uint32_t queued_targets;     /* targets the LDS-tiled kernel handed to the generic kernel in
                                    the last sweep (moved > 1 cell, out of grid, tile overflow)  */
    uint64_t nan_resets;         /* NaN-guard hits (VelPos.compute:143-147) since create         */
    uint64_t candidates_density; /* listed candidates visited by the last density pass (P-bar*N) */
    uint64_t halo_violations;    /* slab mode: owned targets that drifted beyond the ghost planes  */
    /* frame-coherent neighbour lists, accumulated since the last sph_reset_stats */
    uint64_t listed_target_substeps;  /* target sub-steps whose pair sums came from the list of sub-step 1 */
    uint64_t scanned_brick_substeps;  /* non-empty 4x4x4 bricks that had to re-scan in a sub-step 2..S     */
    uint64_t list_entries;            /* list entries those target sub-steps walked                        */
    double ms_velpos_list;            /* GPU time of the list kernel alone -- k_velpos_tile, or k_velpos_list under
                                         SPH_FLAG_GLOBAL_LISTS -- (part of ms_velpos), SPH_FLAG_PROFILE                */
    uint64_t launches_velpos_list;
    uint64_t graph_replays;           /* sph_step frames replayed from a captured HIP graph since create         */
    uint64_t frames_without_lists;    /* sph_step frames the list policy ran without recording (lists would not pay) */
    /* z-slab mode (several GPUs), since the last sph_reset_stats */
    uint64_t host_syncs;              /* host <-> device synchronisations issued by slab frames (one per frame)  */
    uint64_t frames_rerun;            /* frames run again with a wider halo after a violation                     */
    uint64_t halo_planes;             /* ghost planes per face in use now                                         */
    uint64_t list_row_entries;        /* neighbour-list rows: entries per particle now (grown when lists do not fit)  */
    uint64_t lists_too_long;          /* targets of the last recording frame whose list did not fit (their bricks scan) */
} SphStats;

typedef struct SphContext SphContext;

int sph_version(void);

/* Inspector defaults, SphFluidSimulation.cs:35-45 (+ the hard-coded constants). */
int sph_params_default(SphParams *out);

/* Replaces Start(): InitShaders/CreateParticleTextures/InitializeBucketBuffer
 * (SphFluidSimulation.cs:82-94,126-155,192-197).  Particle initialisation is the caller's
 * (sph_upload), the reference does it with InitParticles.compute. */
int sph_create(const SphParams *params, SphContext **out);

/* Replaces OnDestroy() (SphFluidSimulation.cs:110-120). */
int sph_destroy(SphContext *ctx);

/* Writes _particlePositionTextures[Read] / _particleVelocityTextures[Read]
 * (what InitParticles.compute + ClearFloat4 write at SphFluidSimulation.cs:182-189).
 * pos4/vel4: res*res float4, id order.  vel4 may be NULL (= all zero, :189). */
int sph_upload(SphContext *ctx, const float *pos4, const float *vel4);

/* Row f1: fills the position texture like InitParticles.compute does (kernel index = preset,
 * SphFluidSimulation.cs:179-184: 0 Preset1 dam break :10-41, 1 Preset2 :43-74, 2 Preset3 :76-107, with the
 * diagonal simplex-noise jitter of noiseSimplex.cginc::snoise(float4)) and clears the velocities (:189),
 * on the device, without a host round trip.  Replaces sph_upload for callers that start like the reference. */
int sph_init_particles(SphContext *ctx, int preset);

/* Reads back the Read textures (+ _particleDensityTexture).  Any pointer may be NULL. */
int sph_download(SphContext *ctx, float *pos4, float *vel4, float *rho);

/* Row f3, the render feed after the path: UpdateMeshProperties() (SphFluidSimulation.cs:104, :271-284 driving
 * UpdateMeshProperties.compute:18-64).  Fills the context's _particleMeshPropertiesBuffer (:174) on the device from
 * the current Read state: record `id` = MeshProperties (:16-29) = float4x4 mat (column-major memory order, as HLSL
 * packs it and UnityEngine.Matrix4x4 stores it) = translate(pos * simScale - simScale/2) x scale(particle_scale),
 * then float4 color = lerp(blue, red, saturate((|vel| - low) / (high - low))): 20 floats = 80 bytes. */
typedef struct SphMeshParams {
    uint32_t struct_size;      /* = sizeof(SphMeshParams) */
    float low_speed;           /* lowSpeed  :52, :278 */
    float high_speed;          /* highSpeed :53, :277 */
    float particle_scale[3];   /* (particleRadius, particleRadius, particleRadius) :280 */
    float sim_trs[16];         /* transform.localToWorldMatrix :281, column-major (Matrix4x4 memory order) */
} SphMeshParams;
#define SPH_MESH_RECORD_FLOATS 20
int sph_update_mesh_properties(SphContext *ctx, const SphMeshParams *mesh);
/* Copies the buffer to the host (num_particles x 20 floats); synchronises. */
int sph_download_mesh_properties(SphContext *ctx, float *props20);
/* The device buffer itself (valid until sph_destroy), for a renderer that can import device memory (row f4). */
int sph_mesh_properties_device_ptr(SphContext *ctx, void **device_ptr, size_t *bytes);
/* Row f4, zero-copy hand-over: moves that buffer (once) into exportable device memory and returns a NEW POSIX file
 * descriptor for it (dma-buf) that another API on the same GPU can import without a copy -- Vulkan through
 * VK_KHR_external_memory_fd, which is what Unity's Linux renderer would use for _particleMeshPropertiesBuffer, or
 * another process through hipMemImportFromShareableHandle.  The caller owns (closes) the fd; the memory lives until
 * sph_destroy; `bytes` is the size of the allocation (num_particles * 80 rounded up to the device's granularity).
 * Later sph_update_mesh_properties calls write into the exported memory.  SPH_E_UNSUPPORTED if the driver cannot export.
 * Importing in the SAME process works (tests/test_mesh_properties.py::test_gpu_mesh_properties_same_process_import) as long as the
 * importer uses the HIP runtime image this library is bound to (sph_runtime_info): a second libamdhip64 image in the process
 * sees no device at all and faults -- the cause of round 1's crash (profiles/r02_fd_import_two_runtime_images.txt). */
int sph_export_mesh_properties_fd(SphContext *ctx, int *fd, size_t *bytes);

/* Row f4, second step: the Read textures themselves without a host round trip.  sph_update_state_textures fills the library's
 * position / velocity / density staging textures (float4 / float4 / float per texel, id = y*res + x: exactly what
 * SphFluidSimulation.cs:141-153 allocates as _particlePositionTextures[Read] ...) ON THE DEVICE from the current state;
 * sph_export_state_fd hands one of them out as a dma-buf fd (import once, VK_KHR_external_memory_fd; same rules as
 * sph_export_mesh_properties_fd).  Ordering across the APIs: either sph_synchronize before the renderer reads, or give the
 * library a semaphore the renderer created -- sph_import_semaphore_fd (VK_KHR_external_semaphore_fd: binary, or timeline = 1)
 * -- and sph_signal_semaphore(value) enqueues its signal on the context's stream behind everything issued so far.
 * (The semaphore path cannot be exercised in this image: nothing here creates an exportable semaphore.) */
#define SPH_STATE_POSITION 0
#define SPH_STATE_VELOCITY 1
#define SPH_STATE_DENSITY 2
int sph_update_state_textures(SphContext *ctx);
int sph_export_state_fd(SphContext *ctx, int which, int *fd, size_t *bytes);
int sph_import_semaphore_fd(SphContext *ctx, int fd, int timeline);
int sph_signal_semaphore(SphContext *ctx, uint64_t value);

/* One Update() (SphFluidSimulation.cs:96-102) with Time.deltaTime = delta_time:
 * BucketGeneration, DensityCalculation, substeps x UpdateVelocityAndPosition(delta_time/25). */
int sph_step(SphContext *ctx, float delta_time);
int sph_step_n(SphContext *ctx, float delta_time, int frames);

/* The three driver methods, for callers that keep the reference's dispatch granularity. */
int sph_bucket_generation(SphContext *ctx);                      /* :203-221 (ClearBucket + CSMain) */
int sph_density_calculation(SphContext *ctx);                    /* :223-241 (ClearFloat + CSMain)  */
int sph_update_velocity_and_position(SphContext *ctx, float dt); /* :243-269 (CSMain + 2x Swap)     */

/* Mid-run change of the inspector scalars (viscosity .. damping_coeff); sizes are fixed. */
int sph_set_params(SphContext *ctx, const SphParams *params);
int sph_get_derived(SphContext *ctx, SphDerived *out);
int sph_get_stats(SphContext *ctx, SphStats *out);
int sph_reset_stats(SphContext *ctx);
int sph_synchronize(SphContext *ctx);

/* Run on a caller-owned hipStream_t (e.g. torch's current stream) instead of the context's own.  Frames are still replayed from
 * captured HIP graphs on it, except on the legacy default stream (NULL: it cannot be captured -- launches are issued one by one) and
 * while the caller itself is capturing the stream (the launches join the caller's graph). */
int sph_set_stream(SphContext *ctx, void *hip_stream);

/* Which native libraries this process bound the engine to (diagnostics: a host that maps a second HIP runtime image --
 * e.g. one bundled with another plugin -- must hand file descriptors and pointers to THIS one). */
typedef struct SphRuntimeInfo {
    uint32_t struct_size;         /* = sizeof(SphRuntimeInfo) */
    int32_t hip_runtime_version;  /* hipRuntimeGetVersion */
    int32_t hip_driver_version;   /* hipDriverGetVersion  */
    int32_t device_count;
    int32_t rccl_version;         /* 0 while no RCCL image is mapped (the library loads it only when a transport needs it) */
    char hip_library_path[256];   /* image that implements this library's hip* calls (dladdr) */
    char rccl_library_path[256];
} SphRuntimeInfo;
int sph_runtime_info(SphRuntimeInfo *out);

/* Numerics self-test on the device: the shared-reciprocal quotients and the unscaled square root the pair kernels use
 * (csrc/sph_fastdiv.hpp) against the IEEE division / sqrtf instruction sequences, on n pseudo-random operand sets covering the
 * window in which the kernels use them.  mismatches[4] = {quotient, quotient by a doubled denominator, square root, window
 * test}; all zero on a correct build. */
int sph_selftest_arithmetic(int device_id, uint64_t n, uint64_t seed, uint64_t mismatches[4]);

/*
 * ---- z-slab domain decomposition, one process per GPU (new work; the reference is single-GPU) ----
 * (In ONE process, SphParams.num_devices does all of this behind sph_create / sph_step.)
 * A slab context holds the particles whose BUCKET ROW (Bucket.compute:26-29) lies in the grid planes cz in [z_lo, z_hi) of
 * slab `rank` of `world` (sph_slab_planes) plus "ghost" copies of halo planes on either side.  sph_slab_step runs one
 * Update() of the slab INCLUDING the exchange with its two z-neighbours (protocol: csrc/sph_slab_frame.hip):
 *
 *   sph_slab_create -> sph_slab_configure(rank, world) -> sph_slab_comm_init (RCCL; id from sph_comm_unique_id on rank 0,
 *   distributed by the host) or sph_slab_set_host_transport -> sph_slab_upload -> sph_slab_step ... -> sph_slab_download
 *
 * sph_slab_step / sph_slab_finish / sph_slab_download / sph_get_stats are COLLECTIVE: every rank calls them in the same
 * order.  A frame returns enqueued after ONE host synchronisation; whether its halo held is known at the next collective
 * call, which re-runs it with a wider halo if not (bit-identical to the single-GPU result either way).
 * SphParams.num_particles stays the GLOBAL particle count (mass and sentinel depend on it).
 */
#define SPH_SLAB_UPDATE_RHO 0
#define SPH_SLAB_UPDATE_POSVEL 1
#define SPH_COMM_ID_BYTES 128

int sph_slab_create(const SphParams *params, uint32_t local_capacity, SphContext **out);
/* [z_lo, z_hi) of slab `rank`: bucket_resolution planes split as evenly as possible (the rule every rank must share). */
int sph_slab_planes(uint32_t bucket_resolution, int world, int rank, int *z_lo, int *z_hi);
int sph_slab_configure(SphContext *ctx, int rank, int world);
/* RCCL transport: rank 0 makes the id, the host hands the 128 bytes to every rank (e.g. a torch.distributed broadcast). */
int sph_comm_unique_id(void *id_bytes);
int sph_slab_comm_init(SphContext *ctx, const void *id_bytes);
/* Host transport: the library stages the messages through pinned host memory and calls back (blocking) to move them:
 * exchange sends the two buffers to the left / right neighbour and fills the two receive buffers (sizes may be 0);
 * allreduce replaces *value by its sum over all ranks.  Both return 0 on success. */
typedef int (*SphHostExchangeFn)(void *user, const void *send_left, size_t send_left_bytes, const void *send_right,
                                 size_t send_right_bytes, void *recv_left, size_t recv_left_bytes, void *recv_right,
                                 size_t recv_right_bytes);
typedef int (*SphHostAllreduceFn)(void *user, unsigned long long *value);
int sph_slab_set_host_transport(SphContext *ctx, SphHostExchangeFn exchange, SphHostAllreduceFn allreduce, void *user);
/* n owned particles of this rank with their global ids (host arrays); vel4 may be NULL. */
int sph_slab_upload(SphContext *ctx, uint32_t n, const float *pos4, const float *vel4, const uint32_t *ids);
/* One Update() (SphFluidSimulation.cs:96-102) of this slab with Time.deltaTime = delta_time. */
int sph_slab_step(SphContext *ctx, float delta_time);
/* Validates the frame in flight (re-running it if its halo was violated): the state is final afterwards. */
int sph_slab_finish(SphContext *ctx);
/* out = {owned, ghosts_from_left, ghosts_from_right, refreshed_for_left, refreshed_for_right} of the last frame. */
int sph_slab_counts(SphContext *ctx, uint32_t out[5]);
/* Owned particles in arbitrary order with their ids; pos.w is 1 before the first sub-step.  Calls sph_slab_finish. */
int sph_slab_download(SphContext *ctx, uint32_t cap, uint32_t *ids, float *pos4, float *vel4, float *rho,
                      uint32_t *n_out);

/* Never NULL; ctx may be NULL for errors raised before a context exists. */
const char *sph_last_error(SphContext *ctx);

#ifdef __cplusplus
}
#endif
#endif /* SPH_HIP_H */
