// Row f1: the reference's initial-condition generator on the device.
//   Assets/Resources/InitParticles.compute:10-107  Preset1 / Preset2 / Preset3
//   Assets/Resources/noiseSimplex.cginc:62-126, 296-425  mod289, permute, taylorInvSqrt, grad4, snoise(float4)
// dispatched at SphFluidSimulation.cs:179-184 (kernel index = preset), once at Start().
// One IEEE fp32 operation per source operation (-ffp-contract=off), so the positions are bit-identical
// to oracle/sph_oracle.c::oracle_init_particles.  The one transcendental, pow(count/fill, 1/3) at
// InitParticles.compute:20,53,86, is evaluated on the host (powf) and arrives as particle_per_dim.
//
// The 4-D simplex noise restated below (mod289, permute, taylorInvSqrt, grad4, snoise) is the algorithm of
// "webgl-noise" by Ashima Arts / Stefan Gustavson (https://github.com/ashima/webgl-noise), in the CgFx/HLSL port by
// Lex-DRL that the reference vendors as noiseSimplex.cginc.  Upstream notice, as that file carries it (MIT License):
//   Copyright (C) 2011 by Ashima Arts (Simplex noise)
//   Copyright (C) 2011 by Stefan Gustavson (Classic noise)
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated
//   documentation files (the "Software"), to deal in the Software without restriction, including without limitation the
//   rights to use, copy, modify, merge, publish, distribute, sublicense, and/or sell copies of the Software, and to permit
//   persons to whom the Software is furnished to do so, subject to the following conditions: The above copyright notice and
//   this permission notice shall be included in all copies or substantial portions of the Software.
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE
//   WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT.  IN NO EVENT SHALL THE AUTHORS OR
//   COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR
//   OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE.
#include "sph_internal.hpp"

namespace sph {

struct f4 {
    float x, y, z, w;
};

__device__ __forceinline__ float fracf_(float x) { return x - floorf(x); }
__device__ __forceinline__ float step_(float y, float x) { return x >= y ? 1.0f : 0.0f; }   // HLSL step(y, x)
__device__ __forceinline__ float saturate_(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ float dot4(f4 a, f4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
__device__ __forceinline__ float mod289f(float x) { return x - floorf(x * 0.00346020761245674740484429065744f) * 289.0f; }
__device__ __forceinline__ float permutef(float x) { return mod289f(x * x * 34.0f + x); }
__device__ __forceinline__ float taylor_inv_sqrt(float r) { return 1.79284291400159f - 0.85373472095314f * r; }

__device__ f4 grad4(float j, f4 ip)                                                    // noiseSimplex.cginc:115-130
{
    f4 p, s;
    p.x = floorf(fracf_(j * ip.x) * 7.0f) * ip.z - 1.0f;
    p.y = floorf(fracf_(j * ip.y) * 7.0f) * ip.z - 1.0f;
    p.z = floorf(fracf_(j * ip.z) * 7.0f) * ip.z - 1.0f;
    p.w = 1.5f - ((fabsf(p.x) * 1.0f + fabsf(p.y) * 1.0f) + fabsf(p.z) * 1.0f);
    s.x = 1.0f - step_(0.0f, p.x);
    s.y = 1.0f - step_(0.0f, p.y);
    s.z = 1.0f - step_(0.0f, p.z);
    s.w = 1.0f - step_(0.0f, p.w);
    p.x = p.x + (s.x * 2.0f - 1.0f) * s.w;
    p.y = p.y + (s.y * 2.0f - 1.0f) * s.w;
    p.z = p.z + (s.z * 2.0f - 1.0f) * s.w;
    return p;
}

__device__ float snoise4(f4 v)                                                         // noiseSimplex.cginc:296-425
{
    const float Cx = 0.138196601125011f, Cy = 0.276393202250021f, Cz = 0.414589803375032f, Cw = -0.447213595499958f;
    const float F4 = 0.309016994374947451f;
    const float sk = ((v.x * F4 + v.y * F4) + v.z * F4) + v.w * F4;
    f4 i = {floorf(v.x + sk), floorf(v.y + sk), floorf(v.z + sk), floorf(v.w + sk)};
    const float un = ((i.x * Cx + i.y * Cx) + i.z * Cx) + i.w * Cx;
    const f4 x0 = {v.x - i.x + un, v.y - i.y + un, v.z - i.z + un, v.w - i.w + un};
    const float isXx = step_(x0.y, x0.x), isXy = step_(x0.z, x0.x), isXz = step_(x0.w, x0.x);
    const float isYZx = step_(x0.z, x0.y), isYZy = step_(x0.w, x0.y), isYZz = step_(x0.w, x0.z);
    f4 i0;
    i0.x = isXx + isXy + isXz;
    i0.y = 1.0f - isXx;
    i0.z = 1.0f - isXy;
    i0.w = 1.0f - isXz;
    i0.y += isYZx + isYZy;
    i0.z += 1.0f - isYZx;
    i0.w += 1.0f - isYZy;
    i0.z += isYZz;
    i0.w += 1.0f - isYZz;
    const f4 i3 = {saturate_(i0.x), saturate_(i0.y), saturate_(i0.z), saturate_(i0.w)};
    const f4 i2 = {saturate_(i0.x - 1.0f), saturate_(i0.y - 1.0f), saturate_(i0.z - 1.0f), saturate_(i0.w - 1.0f)};
    const f4 i1 = {saturate_(i0.x - 2.0f), saturate_(i0.y - 2.0f), saturate_(i0.z - 2.0f), saturate_(i0.w - 2.0f)};
    const f4 x1 = {x0.x - i1.x + Cx, x0.y - i1.y + Cx, x0.z - i1.z + Cx, x0.w - i1.w + Cx};
    const f4 x2 = {x0.x - i2.x + Cy, x0.y - i2.y + Cy, x0.z - i2.z + Cy, x0.w - i2.w + Cy};
    const f4 x3 = {x0.x - i3.x + Cz, x0.y - i3.y + Cz, x0.z - i3.z + Cz, x0.w - i3.w + Cz};
    const f4 x4 = {x0.x + Cw, x0.y + Cw, x0.z + Cw, x0.w + Cw};
    i.x = mod289f(i.x); i.y = mod289f(i.y); i.z = mod289f(i.z); i.w = mod289f(i.w);
    const float j0 = permutef(permutef(permutef(permutef(i.w) + i.z) + i.y) + i.x);
    const float aw[4] = {i1.w, i2.w, i3.w, 1.0f}, az[4] = {i1.z, i2.z, i3.z, 1.0f};
    const float ay[4] = {i1.y, i2.y, i3.y, 1.0f}, ax[4] = {i1.x, i2.x, i3.x, 1.0f};
    float j1[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        j1[k] = permutef(permutef(permutef(permutef(i.w + aw[k]) + i.z + az[k]) + i.y + ay[k]) + i.x + ax[k]);
    const f4 ip = {0.003401360544217687075f, 0.020408163265306122449f, 0.142857142857142857143f, 0.0f};
    f4 p0 = grad4(j0, ip), p1 = grad4(j1[0], ip), p2 = grad4(j1[1], ip), p3 = grad4(j1[2], ip), p4 = grad4(j1[3], ip);
    const float n0 = taylor_inv_sqrt(dot4(p0, p0)), n1 = taylor_inv_sqrt(dot4(p1, p1));
    const float n2 = taylor_inv_sqrt(dot4(p2, p2)), n3 = taylor_inv_sqrt(dot4(p3, p3));
    p0.x *= n0; p0.y *= n0; p0.z *= n0; p0.w *= n0;
    p1.x *= n1; p1.y *= n1; p1.z *= n1; p1.w *= n1;
    p2.x *= n2; p2.y *= n2; p2.z *= n2; p2.w *= n2;
    p3.x *= n3; p3.y *= n3; p3.z *= n3; p3.w *= n3;
    const float n4 = taylor_inv_sqrt(dot4(p4, p4));
    p4.x *= n4; p4.y *= n4; p4.z *= n4; p4.w *= n4;
    float m00 = fmaxf(0.6f - dot4(x0, x0), 0.0f), m01 = fmaxf(0.6f - dot4(x1, x1), 0.0f);
    float m02 = fmaxf(0.6f - dot4(x2, x2), 0.0f);
    float m10 = fmaxf(0.6f - dot4(x3, x3), 0.0f), m11 = fmaxf(0.6f - dot4(x4, x4), 0.0f);
    m00 = m00 * m00; m01 = m01 * m01; m02 = m02 * m02;
    m10 = m10 * m10; m11 = m11 * m11;
    const float d0 = dot4(p0, x0), d1 = dot4(p1, x1), d2 = dot4(p2, x2), d3 = dot4(p3, x3), d4 = dot4(p4, x4);
    const float a = ((m00 * m00) * d0 + (m01 * m01) * d1) + (m02 * m02) * d2;
    const float b = (m10 * m10) * d3 + (m11 * m11) * d4;
    return 49.0f * (a + b);
}

// InitParticles.compute: Preset1 :10-41 (preset 0), Preset2 :43-74 (1), Preset3 :76-107 (2); one thread per texel
__global__ void __launch_bounds__(256) k_init_particles(int preset, uint32_t res, float fill, uint32_t ppd, float4 *pos4)
{
    const uint32_t ui = blockIdx.x * 256 + threadIdx.x;
    if (ui >= res * res) return;                                                       // :16
    const uint32_t xSize = (uint32_t)ceilf((float)ppd * fill), ySize = ppd;
    const uint32_t zSize = preset == 0 ? ppd : (uint32_t)ceilf((float)ppd * fill);
    const float cs = 1.0f / (float)ppd;                                                // :26
    const int i = (int)ui;                                                             // :28
    const float qx = (float)(ui / (zSize * ySize)), qy = (float)((ui / zSize) % ySize), qz = (float)(ui % zSize);
    f4 pos;
    if (preset == 0) {                                                                 // :30-35
        pos.x = cs / 2.0f + qx * fill / (float)xSize;
        pos.y = cs / 2.0f + qy * 0.9f / (float)ySize;
        pos.z = cs / 2.0f + qz * 1.0f / (float)zSize;
    } else if (preset == 1) {                                                          // :63-68
        pos.x = cs / 2.0f + fill / 2.0f + qx * fill / (float)xSize;
        pos.y = cs / 2.0f + qy * 0.9f / (float)ySize;
        pos.z = cs / 2.0f + fill / 2.0f + qz * fill / (float)zSize;
    } else {                                                                           // :96-101
        pos.x = cs / 2.0f + qx * fill / (float)xSize;
        pos.y = cs / 2.0f + qy * 0.9f / (float)ySize;
        pos.z = cs / 2.0f + qz * fill / (float)zSize;
    }
    pos.w = 1.0f;
    const float fi = (float)i;
    const float nz = snoise4(f4{pos.x + fi, pos.y + fi, pos.z + fi, pos.w + fi}) * cs;  // :37
    pos4[ui] = make_float4(pos.x + nz, pos.y + nz, pos.z + nz, 1.0f);                  // :38-40
}

void launch_init_particles(int preset, uint32_t res, float fill, uint32_t ppd, float4 *pos4, hipStream_t s)
{
    const uint32_t n = res * res;
    hipLaunchKernelGGL(k_init_particles, dim3((n + 255) / 256), dim3(256), 0, s, preset, res, fill, ppd, pos4);
}

}  // namespace sph
