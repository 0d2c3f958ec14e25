// Drop-in for the compute half of Assets/Scripts/SphFluidSimulation.cs: same inspector fields
// (:35-45), Start/Update/OnDestroy with the same meaning, but BucketGeneration / DensityCalculation /
// 5 x UpdateVelocityAndPosition (:98-102) run in libsph_hip.so on an MI355X instead of Unity compute
// shaders.  Rendering (UpdateMeshProperties + DrawMeshInstancedIndirect, :104-107) is out of scope of
// the engine: this shim hands positions / velocities back through Texture2D.SetPixelData so the
// reference's own UpdateMeshProperties.compute can keep consuming them.
// NOT compiled in the build image (no Unity there); written against include/sph_hip.h.
using System;
using UnityEngine;

public class SphHipSimulation : MonoBehaviour
{
    [Header("Initialization")]
    [Range(0, 2)] public int preset = 1;                 // InitParticles.compute kernel index, SphFluidSimulation.cs:35
    [Range(1024, 4194304)] public int particleNumber = 1024;
    [Range(1, 256)] public int bucketResolution = 256;
    [Range(0.01f, 1f)] public float damFillRate = 0.5f;

    [Header("Parameters")]
    [Range(0f, 0.1f)] public float viscosity = 0.01f;
    [Range(0f, 5f)] public float restDensity = 1.5f;
    [Range(1f, 5000f)] public float gasConstant = 150.0f;
    [Range(1000f, 10000f)] public float stiffnessCoefficient = 5000.0f;
    [Range(1f, 50f)] public float dampingCoefficient = 10.0f;

    public Texture2D positionTexture, velocityTexture;   // RGBAFloat, res x res: what the render pass samples
    public bool readBackEveryFrame = true;

    [Header("Rendering (SphFluidSimulation.cs:46-53)")]
    [Range(0.001f, 1f)] public float particleRadius = 0.01f;
    [Range(0f, 1000f)] public float lowSpeed = 0f, highSpeed = 0.5f;
    /// Filled by the library's own UpdateMeshProperties (row f3) instead of positionTexture / velocityTexture:
    /// 80-byte MeshProperties records, bind it as the material's _Properties (SphFluidSimulation.cs:186).
    public ComputeBuffer meshPropertiesBuffer;
    float[] _props;

    IntPtr _ctx = IntPtr.Zero;
    SphNative.SphParams _p;
    float[] _pos, _vel;
    int _res;

    SphNative.SphParams Params()
    {
        SphNative.Check(SphNative.sph_params_default(out var p), IntPtr.Zero);
        p.num_particles = (uint)particleNumber;
        p.bucket_resolution = (uint)bucketResolution;
        p.dam_fill_rate = damFillRate;
        p.viscosity = viscosity;
        p.rest_density = restDensity;
        p.gas_constant = gasConstant;
        p.stiffness_coeff = stiffnessCoefficient;
        p.damping_coeff = dampingCoefficient;
        return p;
    }

    /// initialPositions: res*res float4 in texel order (id = y*res + x); null = let the library run the reference's
    /// InitParticles.compute preset itself (sph_init_particles, kernel index = preset as in SphFluidSimulation.cs:184).
    public void Begin(float[] initialPositions = null)
    {
        _p = Params();
        SphNative.Check(SphNative.sph_create(ref _p, out _ctx), IntPtr.Zero);
        SphNative.Check(SphNative.sph_get_derived(_ctx, out var d), _ctx);
        particleNumber = (int)d.num_particles;          // SphFluidSimulation.cs:84
        _res = (int)d.particle_resolution;              // :85
        _pos = new float[4 * d.particle_count];
        _vel = new float[4 * d.particle_count];
        if (initialPositions != null)
            SphNative.Check(SphNative.sph_upload(_ctx, initialPositions, null), _ctx);   // velocities start at zero, :189
        else
            SphNative.Check(SphNative.sph_init_particles(_ctx, preset), _ctx);           // :179-189
    }

    void Update()
    {
        if (_ctx == IntPtr.Zero) return;
        var p = Params();                                // inspector tweaks mid-run
        if (p.viscosity != _p.viscosity || p.gas_constant != _p.gas_constant || p.rest_density != _p.rest_density ||
            p.stiffness_coeff != _p.stiffness_coeff || p.damping_coeff != _p.damping_coeff)
        {
            SphNative.Check(SphNative.sph_set_params(_ctx, ref p), _ctx);
            _p = p;
        }
        SphNative.Check(SphNative.sph_step(_ctx, Time.deltaTime), _ctx);             // :98-102
        if (meshPropertiesBuffer != null)                                             // :104, :271-284 on the device
        {
            var m = new SphNative.SphMeshParams {
                struct_size = (uint)System.Runtime.InteropServices.Marshal.SizeOf<SphNative.SphMeshParams>(),
                low_speed = lowSpeed, high_speed = highSpeed,
                scale_x = particleRadius, scale_y = particleRadius, scale_z = particleRadius,
                sim_trs = transform.localToWorldMatrix };
            SphNative.Check(SphNative.sph_update_mesh_properties(_ctx, ref m), _ctx);
            if (_props == null) _props = new float[20 * _pos.Length / 4];
            SphNative.Check(SphNative.sph_download_mesh_properties(_ctx, _props), _ctx);
            meshPropertiesBuffer.SetData(_props);    // (zero-copy instead: sph_export_mesh_properties_fd + VK_KHR_external_memory_fd)
        }
        if (readBackEveryFrame && positionTexture != null)
        {
            SphNative.Check(SphNative.sph_download(_ctx, _pos, _vel, null), _ctx);
            positionTexture.SetPixelData(_pos, 0);
            positionTexture.Apply(false);
            if (velocityTexture != null) { velocityTexture.SetPixelData(_vel, 0); velocityTexture.Apply(false); }
        }
    }

    void OnDestroy()
    {
        if (_ctx != IntPtr.Zero) SphNative.sph_destroy(_ctx);                        // :110-120
        _ctx = IntPtr.Zero;
    }
}
