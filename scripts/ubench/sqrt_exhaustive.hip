// Exhaustive check of sph::sqrt_unscaled (sph_fastdiv.hpp) against sqrtf (hipcc -fhip-fp32-correctly-rounded-divide-sqrt) on EVERY
// float of [2^-96, 2^101) -- 197 binades x 2^23 mantissas -- plus zero and the arguments below 2^-96 (result must stay <= 2^-47).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt [-DSPH_SQRT_RESIDUAL=0|1] \
//         -I sphfluidsimulation_amd/csrc -o scripts/ubench/bin/sqrt_exhaustive scripts/ubench/sqrt_exhaustive.hip
#include "sph_fastdiv.hpp"
#include <stdio.h>

__global__ void __launch_bounds__(256) k_check(uint32_t bits_lo, uint32_t bits_hi, int exact, unsigned long long *out)
{
    unsigned long long bad = 0;
    const uint64_t n = (uint64_t)bits_hi - bits_lo;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        const float x = __uint_as_float(bits_lo + (uint32_t)i), s = sph::sqrt_unscaled(x);
        bad += exact ? (__float_as_uint(s) != __float_as_uint(sqrtf(x))) : !(s >= 0.0f && s <= 0x1p-47f);
    }
    if (bad) atomicAdd(out, bad);
}

int main()
{
    unsigned long long *out, h[2];
    if (hipMalloc(&out, sizeof(h)) != hipSuccess || hipMemset(out, 0, sizeof(h)) != hipSuccess) return 2;
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, (127u - 96u) << 23, (127u + 101u) << 23, 1, out);
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, 0u, (127u - 96u) << 23, 0, out + 1);   // zero, denormals, tiny normals
    if (hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    printf("SPH_SQRT_RESIDUAL=%d  sqrt_unscaled != sqrtf on [2^-96, 2^101): %llu of %llu;  results outside [0, 2^-47] for arguments in [0, 2^-96): %llu\n",
           SPH_SQRT_RESIDUAL, h[0], (unsigned long long)197 << 23, h[1]);
    return (h[0] || h[1]) ? 1 : 0;
}
