// Device self-test of the arithmetic short cuts the pair kernels take (sph_fastdiv.hpp): shared-reciprocal quotients and the
// unscaled square root are compared, bit for bit, with the compiler's IEEE `/` and sqrtf on pseudo-random operands that
// cover the whole window in which the kernels use them (and its edges).
#include "sph_sweep_common.hpp"

namespace sph {

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// float with a uniformly drawn biased exponent in [e_lo, e_hi], random mantissa and sign
__device__ __forceinline__ float rand_float(uint64_t r, uint32_t e_lo, uint32_t e_hi, bool with_sign)
{
    const uint32_t e = e_lo + (uint32_t)((r >> 40) % (e_hi - e_lo + 1u));
    const uint32_t bits = (e << 23) | ((uint32_t)r & 0x7FFFFFu) | (with_sign ? ((uint32_t)(r >> 32) & 1u) << 31 : 0u);
    return __uint_as_float(bits);
}

__global__ void __launch_bounds__(256) k_selftest_arithmetic(uint64_t n, uint64_t seed, unsigned long long *out)
{
    unsigned long long bad_div = 0, bad_div2 = 0, bad_sqrt = 0, bad_key = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        const uint64_t r0 = mix64(seed + 3u * i), r1 = mix64(seed + 3u * i + 1u), r2 = mix64(seed + 3u * i + 2u);
        // numerator: 0 (1 in 16), else 2^-90 <= |a| < 2^52; denominator: 2^-20 <= b < 2^31 (edges included: mantissa 0 / all ones)
        float a = rand_float(r0, 127u - 90u, 127u + 51u, true);
        if ((r0 >> 60) == 0u) a = (r0 >> 59) & 1u ? 0.0f : -0.0f;
        float b = rand_float(r1, 127u - 20u, 127u + 30u, false);
        if (((r1 >> 56) & 0xFFu) == 1u) b = __uint_as_float(__float_as_uint(b) & 0xFF800000u);   // exact powers of two
        if (((r1 >> 56) & 0xFFu) == 2u) b = __uint_as_float(__float_as_uint(b) | 0x007FFFFFu);
        const Recip rb = recip_refined(b);
        const float q = div_shared(a, rb), q_ref = a / b;
        if (__float_as_uint(q) != __float_as_uint(q_ref)) bad_div++;
        const float q2 = div_shared(a, recip_twice(rb)), q2_ref = a / (2.0f * b);
        if (__float_as_uint(q2) != __float_as_uint(q2_ref)) bad_div2++;
        // without v_div_fixup (pair_finish_folded): the same bits except for the sign of a zero quotient
        const float qn = div_shared_nofix(a, rb);
        if (__float_as_uint(qn) != __float_as_uint(q_ref) && !(qn == 0.0f && q_ref == 0.0f)) bad_div++;
        // the window test itself: in-window numerators must pass it, 0 < |t| < 2^-90 must fail it
        if (mag_key(a) < kMagKeyMin) bad_key++;
        const float t = rand_float(r2, 1u, 127u - 91u, true);
        if (!(mag_key(t) < kMagKeyMin)) bad_key++;
        // sqrt: 0 (1 in 16), else 2^-96 <= x < 2^101
        float x = rand_float(r2, 127u - 96u, 127u + 100u, false);
        if ((r2 >> 60) == 0u) x = 0.0f;
        const float s = sqrt_unscaled(x), s_ref = sqrtf(x);
        if (__float_as_uint(s) != __float_as_uint(s_ref)) bad_sqrt++;
        // below the window the unscaled root only has to stay in [0, 2^-47]
        const float y = rand_float(r0 ^ r2, 1u, 127u - 97u, false), sy = sqrt_unscaled(y);
        if (!(sy >= 0.0f && sy <= 7.2e-15f)) bad_sqrt++;
    }
    if (bad_div) atomicAdd(&out[0], bad_div);
    if (bad_div2) atomicAdd(&out[1], bad_div2);
    if (bad_sqrt) atomicAdd(&out[2], bad_sqrt);
    if (bad_key) atomicAdd(&out[3], bad_key);
}

void launch_selftest_arithmetic(uint64_t n, uint64_t seed, unsigned long long *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_arithmetic, dim3(256 * 8), dim3(256), 0, s, n, seed, out);
}

}  // namespace sph
