// C++ host-side mirror of the reference driver Assets/Scripts/SphFluidSimulation.cs over the C ABI of
// include/sph_hip.h (the reference's own toolchain, Unity C#, is not available in the build image, so
// the host side above the ABI is C++ where the reference is compiled code; the C# binding a Unity
// maintainer would add is in INTEGRATION.md and integration/unity/).
// Same public field names (inspector fields :35-45) and method names (Start :82, Update :96,
// OnDestroy :110, BucketGeneration :203, DensityCalculation :223, UpdateVelocityAndPosition :243).
#pragma once

#include <stdexcept>
#include <string>
#include <vector>

#include "sph_hip.h"

class SphFluidSimulation {
public:
    // inspector fields, SphFluidSimulation.cs:35-45
    int preset = 1;
    int particleNumber = 1024;
    int bucketResolution = 256;
    float damFillRate = 0.5f;
    float viscosity = 0.01f;
    float restDensity = 1.5f;
    float gasConstant = 150.0f;
    float stiffnessCoefficient = 5000.0f;
    float dampingCoefficient = 10.0f;
    // rendering fields :49-53 (SampleScene.unity:372-376)
    float particleRadius = 0.01f, lowSpeed = 0.0f, highSpeed = 0.5f;
    int deviceId = 0;
    std::vector<int> deviceIds;   // 2..8 HIP ordinals: the particles are split into z-slabs over these GPUs (same calls)

    ~SphFluidSimulation() { OnDestroy(); }

    // :82-94; the reference fills the position texture with InitParticles.compute: pass positions == nullptr for
    // that (row f1, on the device), or hand the float4 arrays (id = y*res + x).  velocities may be nullptr
    // (ClearFloat4, :189).
    void Start(const float *positions, const float *velocities)
    {
        SphParams p = params();
        check(sph_create(&p, &ctx_));
        SphDerived d;
        check(sph_get_derived(ctx_, &d));
        particleNumber = (int)d.num_particles;            // :84
        _particleTextureResolution = (int)d.particle_resolution;  // :85
        _effectiveRadius = d.effective_radius;            // :159
        _particleMass = d.particle_mass;                  // :176
        count_ = d.particle_count;
        if (positions)
            check(sph_upload(ctx_, positions, velocities));
        else
            check(sph_init_particles(ctx_, preset));      // :179-189, InitParticles.compute kernel #preset
    }

    // Start() exactly as the reference does it: positions from InitParticles.compute, velocities cleared
    void Start() { Start(nullptr, nullptr); }

    // :96-102 with Time.deltaTime = deltaTime
    void Update(float deltaTime) { check(sph_step(ctx_, deltaTime)); }

    // :110-120
    void OnDestroy()
    {
        if (ctx_) sph_destroy(ctx_);
        ctx_ = nullptr;
    }

    // :271-284 (row f3): fills the device-side _particleMeshPropertiesBuffer; simTRS = transform.localToWorldMatrix,
    // column-major like UnityEngine.Matrix4x4 (nullptr = identity)
    void UpdateMeshProperties(const float *simTRS = nullptr)
    {
        SphMeshParams m{};
        m.struct_size = sizeof(SphMeshParams);
        m.low_speed = lowSpeed;
        m.high_speed = highSpeed;
        for (int k = 0; k < 3; k++) m.particle_scale[k] = particleRadius;   // :280
        for (int k = 0; k < 16; k++) m.sim_trs[k] = simTRS ? simTRS[k] : (k % 5 == 0 ? 1.0f : 0.0f);
        check(sph_update_mesh_properties(ctx_, &m));
    }
    // 20 floats per particle (MeshProperties, :16-29), record index = texel id
    void DownloadMeshProperties(float *props20) { check(sph_download_mesh_properties(ctx_, props20)); }
    // row f4: dma-buf fd of that buffer for a renderer on the same GPU (VK_KHR_external_memory_fd); the caller closes it
    int ExportMeshPropertiesFd(size_t *bytes = nullptr)
    {
        int fd = -1;
        check(sph_export_mesh_properties_fd(ctx_, &fd, bytes));
        return fd;
    }

    void BucketGeneration() { check(sph_bucket_generation(ctx_)); }                          // :203-221
    void DensityCalculation() { check(sph_density_calculation(ctx_)); }                      // :223-241
    void UpdateVelocityAndPosition(float dt) { check(sph_update_velocity_and_position(ctx_, dt)); }  // :243-269

    // read-back of the Read textures (the reference never reads back; a drop-in must)
    void Download(std::vector<float> &pos4, std::vector<float> &vel4, std::vector<float> &rho)
    {
        pos4.resize(4 * (size_t)count_);
        vel4.resize(4 * (size_t)count_);
        rho.resize(count_);
        check(sph_download(ctx_, pos4.data(), vel4.data(), rho.data()));
    }

    // push changed inspector scalars mid-run
    void ApplyParameters()
    {
        SphParams p = params();
        check(sph_set_params(ctx_, &p));
    }

    SphStats Stats()
    {
        SphStats s;
        check(sph_get_stats(ctx_, &s));
        return s;
    }

    unsigned count() const { return count_; }
    int _particleTextureResolution = 0;
    float _effectiveRadius = 0.0f, _particleMass = 0.0f;

private:
    SphContext *ctx_ = nullptr;
    unsigned count_ = 0;

    SphParams params() const
    {
        SphParams p;
        sph_params_default(&p);
        p.num_particles = (uint32_t)particleNumber;
        p.bucket_resolution = (uint32_t)bucketResolution;
        p.dam_fill_rate = damFillRate;
        p.viscosity = viscosity;
        p.rest_density = restDensity;
        p.gas_constant = gasConstant;
        p.stiffness_coeff = stiffnessCoefficient;
        p.damping_coeff = dampingCoefficient;
        p.device_id = deviceId;
        if (deviceIds.size() >= 2 && deviceIds.size() <= 8) {
            p.num_devices = (uint32_t)deviceIds.size();
            for (size_t k = 0; k < deviceIds.size(); k++) p.device_ids[k] = deviceIds[k];
        }
        return p;
    }

    void check(int rc)
    {
        if (rc != SPH_OK) throw std::runtime_error(std::string("libsph_hip: ") + sph_last_error(ctx_));
    }
};
