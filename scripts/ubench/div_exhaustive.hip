// Exhaustive check over ALL 2^23 x 2^23 significand pairs: which shared-reciprocal quotient sequences equal IEEE a / b?
// (Inside the window of sph_fastdiv.hpp nothing is scaled and every step commutes with powers of two, so the significands decide.)
//   ONE correction step  q0 = a r; e0 = fma(-b, q0, a); q1 = fma(e0, r, q0)          (3 instructions per numerator; Markstein's theorem
//   asks for r = RN(1 / b)), against the compiler's two (5 instructions, div_shared_nofix), with
//     V1: r = v_rcp + one Newton step (recip_refined: what the kernels hold now)
//     V2: r = RN(1 / b), the IEEE reciprocal
//     V3: r = v_rcp + two Newton steps
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I sphfluidsimulation_amd/csrc \
//         -o scripts/ubench/bin/div_exhaustive scripts/ubench/div_exhaustive.hip ;  div_exhaustive [log2 of the a's per b, default 23]
#include "sph_fastdiv.hpp"
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ float one_step(float a, float b, float r)
{
    const float q0 = a * r;
    return __builtin_fmaf(__builtin_fmaf(-b, q0, a), r, q0);
}

// thread = one b; [a_lo, a_hi) significands of a in this launch.  bad[v]: mismatching pairs; flag[b] bit v: b had a mismatch
__global__ void __launch_bounds__(256) k_pairs(uint32_t a_lo, uint32_t a_hi, unsigned long long *bad, unsigned char *flag)
{
    const uint32_t mb = blockIdx.x * 256u + threadIdx.x;
    const float b = __uint_as_float(0x3F800000u + mb);
    const sph::Recip r1 = sph::recip_refined(b);
    const float r2 = 1.0f / b;
    const float r3 = __builtin_fmaf(__builtin_fmaf(-b, r1.r, 1.0f), r1.r, r1.r);
    uint32_t n1 = 0, n2 = 0, n3 = 0, n5 = 0, n0 = 0;
    for (uint32_t ma = a_lo; ma < a_hi; ma++) {
        const float a = __uint_as_float(0x3F800000u + ma);
        const uint32_t ref = __float_as_uint(a / b);
        n1 += __float_as_uint(one_step(a, b, r1.r)) != ref;
        n2 += __float_as_uint(one_step(a, b, r2)) != ref;
        n3 += __float_as_uint(one_step(a, b, r3)) != ref;
        n5 += __float_as_uint(sph::div_shared(a, r1)) != ref;
        n0 += __float_as_uint(a * r2) != ref;                                      // sanity: no correction step at all
    }
    if (n1) atomicAdd(&bad[0], (unsigned long long)n1);
    if (n2) atomicAdd(&bad[1], (unsigned long long)n2);
    if (n3) atomicAdd(&bad[2], (unsigned long long)n3);
    if (n5) atomicAdd(&bad[3], (unsigned long long)n5);
    if (n0) atomicAdd(&bad[4], (unsigned long long)n0);
    const unsigned char f = (n1 ? 1 : 0) | (n2 ? 2 : 0) | (n3 ? 4 : 0) | (n5 ? 8 : 0) | (__float_as_uint(r1.r) != __float_as_uint(r2) ? 16 : 0) |
                            (__float_as_uint(r3) != __float_as_uint(r2) ? 32 : 0);
    if (f) flag[mb] |= f;
}

int main(int argc, char **argv)
{
    const int la = argc > 1 ? atoi(argv[1]) : 23;
    const uint32_t n_a = 1u << la, n_b = 1u << 23, per = 1u << 15;
    unsigned long long *bad, h[5];
    unsigned char *flag, *hf = (unsigned char *)malloc(n_b);
    if (hipMalloc(&bad, sizeof(h)) != hipSuccess || hipMalloc(&flag, n_b) != hipSuccess) return 2;
    hipMemset(bad, 0, sizeof(h));
    hipMemset(flag, 0, n_b);
    // a's spread over the whole significand range when fewer than 2^23 are asked for: stride 2^(23-la) from a rotating offset
    for (uint32_t lo = 0; lo < n_a; lo += per) {
        const uint32_t base = la == 23 ? lo : (uint32_t)(((unsigned long long)lo << (23 - la)) + (lo / per) % (1u << (23 - la)));
        hipLaunchKernelGGL(k_pairs, dim3(n_b / 256u), dim3(256), 0, 0, base, base + per, bad, flag);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
        if ((lo / per) % 16 == 15) {
            hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
            printf("a < %u of %u: mismatches V1 %llu V2 %llu V3 %llu two-step %llu\n", lo + per, n_a, h[0], h[1], h[2], h[3]);
            fflush(stdout);
        }
    }
    hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(hf, flag, n_b, hipMemcpyDeviceToHost);
    unsigned long long nb[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t i = 0; i < n_b; i++)
        for (int k = 0; k < 6; k++) nb[k] += (hf[i] >> k) & 1;
    printf("pairs per b: %u (of 2^23).  mismatching pairs against a / b:  V1 (rcp + 1 Newton, 1 step) %llu in %llu b's;  V2 (RN(1/b), 1 step) %llu in %llu b's;  "
           "V3 (rcp + 2 Newton, 1 step) %llu in %llu b's;  div_shared (2 steps) %llu in %llu b's\n", n_a, h[0], nb[0], h[1], nb[1], h[2], nb[2], h[3], nb[3]);
    printf("sanity: a * RN(1/b) without a correction step differs from a / b in %llu pairs (%.1f %%)\n", h[4], 100.0 * (double)h[4] / ((double)n_a * n_b));
    printf("b's whose refined reciprocal differs from RN(1/b): one Newton step %llu, two %llu\n", nb[4], nb[5]);
    // the b's V2 fails for (expected: few, special significands)
    int shown = 0;
    for (uint32_t i = 0; i < n_b && shown < 16; i++)
        if (hf[i] & 2) { printf("  V2 fails for b significand 0x%06x\n", i); shown++; }
    return 0;
}
