// Row f3: the render feed that follows the path.
//   Assets/Resources/UpdateMeshProperties.compute:18-64, driven by SphFluidSimulation.cs:104, :271-284.
// Per particle: 28 B in (x, y, z, rv) + 4 B id, 80 B out (MeshProperties: column-major float4x4 + float4 colour,
// SphFluidSimulation.cs:16-29) -> HBM-bound, 112 B per particle.  Reads the cell-sorted SoA in slot order
// (coalesced) and writes record `id` (the reference's texel order), five 16 B stores that L2 merges.
#include "sph_internal.hpp"

namespace sph {

constexpr int kBlock = 256;

struct MeshArgs {
    float low, high;
    float scale[3];
    float sim_scale[3];   // (_m00, _m11, _m22) of _SimTRS, UpdateMeshProperties.compute:35
};

__global__ void __launch_bounds__(kBlock) k_mesh_properties(Dev d, MeshArgs a, float4 *__restrict__ props)
{
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= d.count) return;
    const uint32_t id = d.id[s];                                                      // :27
    const float4 rv = d.cur.rv[s];
    const float speed = sqrtf((rv.y * rv.y + rv.z * rv.z) + rv.w * rv.w);             // :32 length(vel.xyz)
    const Pos3 q = load_pos(d, d.cur, s);
    const float w[3] = {q.x * a.sim_scale[0] - (a.sim_scale[0] * 0.5f),        // :41
                        q.y * a.sim_scale[1] - (a.sim_scale[1] * 0.5f),
                        q.z * a.sim_scale[2] - (a.sim_scale[2] * 0.5f)};
    // :44-60 mul(translation, scale) as the literal 4-term sums (0 * inf = NaN poisons a row like on any IEEE GPU)
    const float S[4][4] = {{a.scale[0], 0.f, 0.f, 0.f}, {0.f, a.scale[1], 0.f, 0.f}, {0.f, 0.f, a.scale[2], 0.f},
                           {0.f, 0.f, 0.f, 1.f}};
    const float T[4][4] = {{1.f, 0.f, 0.f, w[0]}, {0.f, 1.f, 0.f, w[1]}, {0.f, 0.f, 1.f, w[2]}, {0.f, 0.f, 0.f, 1.f}};
    float m[4][4];   // m[c][r]: column-major memory order
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++)
            m[c][r] = ((T[r][0] * S[0][c] + T[r][1] * S[1][c]) + T[r][2] * S[2][c]) + T[r][3] * S[3][c];
    const float t = clamp01((speed - a.low) / (a.high - a.low));                      // :63 saturate
    float4 *o = props + (size_t)id * 5;
#pragma unroll
    for (int c = 0; c < 4; c++) o[c] = make_float4(m[c][0], m[c][1], m[c][2], m[c][3]);
    // :64 lerp(float4(0,0,1,1), float4(1,0,0,1), t) = x + t*(y - x)
    o[4] = make_float4(0.0f + t * (1.0f - 0.0f), 0.0f + t * (0.0f - 0.0f), 1.0f + t * (0.0f - 1.0f), 1.0f + t * (1.0f - 1.0f));
}

void launch_mesh_properties(const Dev &d, float low, float high, const float *scale3, const float *sim_trs16,
                            float4 *props, hipStream_t s)
{
    MeshArgs a;
    a.low = low;
    a.high = high;
    for (int k = 0; k < 3; k++) {
        a.scale[k] = scale3[k];
        a.sim_scale[k] = sim_trs16[5 * k];   // column-major: m00 = [0], m11 = [5], m22 = [10]
    }
    hipLaunchKernelGGL(k_mesh_properties, dim3((d.count + kBlock - 1) / kBlock + (d.count == 0)), dim3(kBlock), 0, s, d, a, props);
}

}  // namespace sph
