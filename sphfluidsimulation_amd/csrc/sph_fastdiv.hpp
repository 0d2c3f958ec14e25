// Correctly rounded fp32 quotients and square roots WITHOUT the range scaling of the compiler's expansion (gfx950).
//
// The parity contract (oracle/sph_oracle.c) is IEEE fp32 `/` and `sqrt`.  hipcc expands a/b into
//     bs = v_div_scale(b,b,a); as = v_div_scale(a,b,a); r0 = v_rcp(bs); e = fma(-bs,r0,1); r = fma(e,r0,r0);
//     q0 = as*r; e0 = fma(-bs,q0,as); q1 = fma(e0,r,q0); e1 = fma(-bs,q1,as); q = v_div_fmas(e1,r,q1); v_div_fixup(q,b,a)
// (11 instructions, one of them transcendental).  v_div_scale only changes its operand -- and v_div_fmas only differs
// from a plain fma -- when an operand or the quotient is denormal / near overflow (CDNA ISA, V_DIV_SCALE_F32):
//     a == 0 or b == 0 | exp(a)-exp(b) >= 96 | b denormal | 1/b denormal | a/b denormal | exp(a) <= 23.
// Outside those cases the sequence is plain arithmetic on the operands themselves, and VelPos.compute divides SEVEN
// numerators by two denominators per pair (3 x |r|, 4 x rho_j or 2 rho_j: :37, :93, :94): the refined reciprocal `r`
// (v_rcp + two fma) is computed once per denominator, the six remaining instructions per numerator are the compiler's
// own, operation for operation, so the quotient is bit-identical to `/`.  v_div_fixup stays: it gives a zero numerator
// its signed zero exactly as in the full expansion.
//
// Window in which the callers may use it (everything else takes `/`; the test is ONE min-tree over the numerators'
// bit patterns per pair, see pair_unsafe_bits):
//     numerator   a == 0  or  2^-90 <= |a| <= 2^51      (velocities, pressures < 1e15: the "wild" launch flag)
//     denominator 2^-20 < b < 2^30                       (|r| in (1e-6, h), rho_j in (1e-6, 1e9): wild otherwise)
// -> |exp(a)-exp(b)| < 96, a/b >= 2^-120 is normal, nothing is scaled.  tests: sph_selftest_arithmetic (1e8 operand
// pairs against `/` and sqrtf on the device) and every bit-identity test of tests/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sph {

struct Recip {
    float b;   // the denominator
    float r;   // its reciprocal after the compiler's two refinement steps
};

__device__ __forceinline__ Recip recip_refined(float b)
{
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r0, 1.0f);
    return Recip{b, __builtin_fmaf(e, r0, r0)};
}

// reciprocal of 2b from the one of b: every step of the refinement scales exactly by 1/2
__device__ __forceinline__ Recip recip_twice(const Recip &d) { return Recip{2.0f * d.b, 0.5f * d.r}; }

__device__ __forceinline__ float div_shared(float a, const Recip &d)
{
    const float q0 = a * d.r;
    const float e0 = __builtin_fmaf(-d.b, q0, a);
    const float q1 = __builtin_fmaf(e0, d.r, q0);
    const float e1 = __builtin_fmaf(-d.b, q1, a);
    const float q2 = __builtin_fmaf(e1, d.r, q1);          // v_div_fmas with VCC = 0
    return __builtin_amdgcn_div_fixupf(q2, d.b, a);
}

// sqrtf(x) for x == 0 or x >= 2^-96 finite: v_sqrt (1 ulp) + the compiler's own one-ulp correction, without its
// 2^32 pre-scaling of tiny arguments and without the zero / inf pass-through (zero passes through the correction
// unharmed).  For 0 < x < 2^-96 the result is some value in [0, 2^-47]: the callers only compare it with 1e-6 and
// subtract it from h >= 2^-8, which every such value leaves unchanged.
__device__ __forceinline__ float sqrt_unscaled(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
    s = (0.0f >= rd) ? sd : s;
    s = (0.0f < ru) ? su : s;
    return s;
}

// Bit pattern of |a| shifted so that unsigned order == magnitude order and zero sorts LAST: (bits << 1) - 1.
__device__ __forceinline__ uint32_t mag_key(float a) { return (__float_as_uint(a) << 1) - 1u; }
constexpr uint32_t kMagKeyMin = ((127u - 90u) << 24) - 1u;   // mag_key(2^-90)

__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t m = a < b ? a : b;
    return m < c ? m : c;
}

}  // namespace sph
