// P/Invoke binding of include/sph_hip.h for Unity (Linux player / editor, libsph_hip.so on the
// plugin path, e.g. Assets/Plugins/x86_64/).  Written against the header; it could NOT be compiled
// in the build image (no Unity / dotnet / mono there) -- the same ABI is exercised from C++
// (host/sph_host_demo.cpp) and Python ctypes (sphfluidsimulation_amd/_native.py).
using System;
using System.Runtime.InteropServices;
using UnityEngine;   // Matrix4x4 in SphMeshParams

// (Not bound here: sph_set_stream and the sph_slab_* / sph_comm_* entry points of the one-process-per-GPU mode.  A single
// Unity process reaches several GPUs through SphParams.num_devices / device_id0..7 and the SAME calls as below.)
public static class SphNative
{
    const string Lib = "sph_hip";

    public const int SPH_OK = 0;
    public const uint SPH_FLAG_PROFILE = 1u;

    [StructLayout(LayoutKind.Sequential)]
    public struct SphParams
    {
        public uint struct_size;        // = Marshal.SizeOf<SphParams>() (120)
        public uint num_particles;      // particleNumber,        SphFluidSimulation.cs:36
        public uint bucket_resolution;  // bucketResolution,      :37
        public float dam_fill_rate;     // damFillRate,           :38
        public float viscosity;         // viscosity,             :41
        public float rest_density;      // restDensity,           :42
        public float gas_constant;      // gasConstant,           :43
        public float stiffness_coeff;   // stiffnessCoefficient,  :44
        public float damping_coeff;     // dampingCoefficient,    :45
        public uint max_per_voxel;      // MaxParticlesPerVoxel,  :9
        public uint substeps;           // 5,                     :101
        public float dt_divisor;        // 25,                    :102
        public float gravity_x, gravity_y, gravity_z;  // VelPos.compute:7
        public uint precision;          // 0 fp32 positions; 1 fp16 position records (within 1e-4)
        public uint exact_order;        // 1 the reference's arithmetic bit for bit; 0 tolerance mode (hardware rcp / sqrt, within 1e-4, ~25 % faster)
        public int device_id;
        public uint flags;
        // ABI version 2: 2..8 = split the particles into that many z-slabs, slab i on HIP device device_id<i> (RCCL halo
        // exchange inside the library); Update() / GetData calls are unchanged
        public uint num_devices;
        public int device_id0, device_id1, device_id2, device_id3, device_id4, device_id5, device_id6, device_id7;  // int32_t device_ids[8]
        public uint halo_planes;        // 0 = default (2); grows automatically when particles outrun it
        public float list_skin;         // 0 = default (0.04)
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SphDerived
    {
        public uint num_particles, particle_resolution, particle_count, bucket_resolution;
        public float particle_mass, effective_radius, effective_radius2, effective_radius6, effective_radius9;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SphStats
    {
        public ulong frames, substeps;
        public double ms_grid, ms_density, ms_velpos;
        public ulong launches_grid, launches_density, launches_velpos;
        public uint max_cell_occupancy, overflowed_cells, dropped_particles, queued_targets;
        public ulong nan_resets, candidates_density, halo_violations;
        public ulong listed_target_substeps, scanned_brick_substeps, list_entries;
        public double ms_velpos_list;
        public ulong launches_velpos_list, graph_replays, frames_without_lists;
        public ulong host_syncs, frames_rerun, halo_planes, list_row_entries, lists_too_long;
    }

    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_version();
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_params_default(out SphParams p);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_create(ref SphParams p, out IntPtr ctx);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_destroy(IntPtr ctx);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_upload(IntPtr ctx, float[] pos4, float[] vel4);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_init_particles(IntPtr ctx, int preset);

    // row f3: UpdateMeshProperties() :271-284.  sim_trs = transform.localToWorldMatrix, column-major = Matrix4x4's own field order
    [StructLayout(LayoutKind.Sequential)]
    public struct SphMeshParams
    {
        public uint struct_size;
        public float low_speed, high_speed;
        public float scale_x, scale_y, scale_z;
        public Matrix4x4 sim_trs;
    }
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_update_mesh_properties(IntPtr ctx, ref SphMeshParams mesh);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_download_mesh_properties(IntPtr ctx, [Out] float[] props20);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_mesh_properties_device_ptr(IntPtr ctx, out IntPtr devicePtr, out UIntPtr bytes);
    // row f4: dma-buf fd of the render-feed buffer for VK_KHR_external_memory_fd (the caller closes the fd)
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_export_mesh_properties_fd(IntPtr ctx, out int fd, out UIntPtr bytes);
    // row f4, second step: the Read textures on the device (no sph_download + SetPixelData): which = 0 position, 1 velocity, 2 density
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_update_state_textures(IntPtr ctx);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_export_state_fd(IntPtr ctx, int which, out int fd, out UIntPtr bytes);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_import_semaphore_fd(IntPtr ctx, int fd, int timeline);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_signal_semaphore(IntPtr ctx, ulong value);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_download(IntPtr ctx, float[] pos4, float[] vel4, float[] rho);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_step(IntPtr ctx, float deltaTime);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_step_n(IntPtr ctx, float deltaTime, int frames);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_bucket_generation(IntPtr ctx);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_density_calculation(IntPtr ctx);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_update_velocity_and_position(IntPtr ctx, float dt);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_set_params(IntPtr ctx, ref SphParams p);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_get_derived(IntPtr ctx, out SphDerived d);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_get_stats(IntPtr ctx, out SphStats s);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_reset_stats(IntPtr ctx);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int sph_synchronize(IntPtr ctx);
    [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern IntPtr sph_last_error(IntPtr ctx);

    public static void Check(int rc, IntPtr ctx)
    {
        if (rc != SPH_OK)
            throw new InvalidOperationException("libsph_hip (" + rc + "): " + Marshal.PtrToStringAnsi(sph_last_error(ctx)));
    }
}
