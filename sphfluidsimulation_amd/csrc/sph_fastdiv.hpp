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
// (v_rcp + two fma) is computed once per denominator; the remaining instructions per numerator are the compiler's own
// (SPH_DIV_ONE_STEP 0) or one correction step fewer (1, checked exhaustively: below), so the quotient is bit-identical to
// `/`.  v_div_fixup stays where signed zeros matter: it gives a zero numerator its signed zero exactly as in the full expansion.
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

// SPH_DIV_ONE_STEP 1: ONE correction step per numerator, q1 = fma(fma(-b, q0, a), r, q0) with q0 = a r (3 instructions instead of the
// compiler's 5).  Markstein's theorem gives the correctly rounded quotient from one step when r = RN(1 / b); on gfx950 v_rcp_f32 plus
// the one Newton step of recip_refined IS RN(1 / b) for every significand, and q1 == a / b for ALL 2^23 x 2^23 significand pairs:
// scripts/ubench/div_exhaustive.hip (7.0e13 pairs on the device, 47 s, 0 mismatches; profiles/r03b_div_exhaustive.txt).  Inside the
// window below nothing is scaled and every step commutes with powers of two, so the significands decide; the exponents are covered
// by sph_selftest_arithmetic's random operands.  0: the compiler's two steps, operation for operation.
#ifndef SPH_DIV_ONE_STEP
#define SPH_DIV_ONE_STEP 1
#endif
__device__ __forceinline__ float div_shared_nofix(float a, const Recip &d)
{
    const float q0 = a * d.r;
    const float e0 = __builtin_fmaf(-d.b, q0, a);
    const float q1 = __builtin_fmaf(e0, d.r, q0);
#if SPH_DIV_ONE_STEP
    return q1;
#else
    const float e1 = __builtin_fmaf(-d.b, q1, a);
    return __builtin_fmaf(e1, d.r, q1);                    // v_div_fmas with VCC = 0
#endif
}

// ... with v_div_fixup: the sign of a zero quotient (and inf / NaN operands) exactly as in the compiler's expansion
__device__ __forceinline__ float div_shared(float a, const Recip &d) { return __builtin_amdgcn_div_fixupf(div_shared_nofix(a, d), d.b, a); }

// sqrtf(x) for x == 0 or x >= 2^-96 finite, without the compiler's 2^32 pre-scaling of tiny arguments and without its zero / inf
// pass-through.  For 0 < x < 2^-96 the result is some value in [0, 2^-47]: the callers only compare it with 1e-6 and subtract it
// from h >= 2^-8, which every such value leaves unchanged.
//   SPH_SQRT_RESIDUAL 1: s = v_sqrt(x) (1 ulp), one residual step s + (x - s^2) * (0.5 * v_rsq(x)) with the residual exact in an fma
//     (Markstein's final step; 2 transcendental + 4 plain instructions).  0.5 / sqrt(x) is capped at 2^47, its value at x = 2^-96:
//     for x = 0 the step then adds 0 * 2^47 instead of 0 * inf.  Equal to sqrtf on EVERY float of [2^-96, 2^101):
//     scripts/ubench/sqrt_exhaustive.hip (197 x 2^23 arguments on the device, 0 mismatches).
//   SPH_SQRT_RESIDUAL 0: v_sqrt + the compiler's own one-ulp correction (2 integer adds, 2 fma, 2 compares, 2 selects).
//   SPH_SQRT_RESIDUAL 2: from v_rsq alone, sqrt and half its reciprocal refined together (1 transcendental + 8 plain; exhaustively equal too).
//   k_velpos_tile with 0 / 1 / 2: 1.125 / 1.118 / 1.130 ms -- the walk pays per instruction, nearly whatever the instruction.
#ifndef SPH_SQRT_RESIDUAL
#define SPH_SQRT_RESIDUAL 1
#endif
__device__ __forceinline__ float sqrt_unscaled(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
#if SPH_SQRT_RESIDUAL == 2
    // from v_rsq alone (1 transcendental + 8 plain): g ~ sqrt(x) and h ~ 1 / (2 sqrt(x)) refined together once, then the residual step
    const float y = __builtin_fminf(__builtin_amdgcn_rsqf(x), 0x1p48f);
    float g = x * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    (void)s;
    return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
#elif SPH_SQRT_RESIDUAL
    const float hr = __builtin_fminf(0.5f * __builtin_amdgcn_rsqf(x), 0x1p47f);
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), hr, s);
#else
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
    s = (0.0f >= rd) ? sd : s;
    s = (0.0f < ru) ? su : s;
    return s;
#endif
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
