// Device code shared by the sweep kernels (sph_kernels_sweep.hip: per-target kernels on global
// memory; sph_kernels_brick.hip: LDS-tiled brick kernels).  Every function evaluates the
// reference's statements one IEEE fp32 operation at a time, in source order
// (Density.compute:22-27, VelPos.compute:33-44,86-157) -- see oracle/sph_oracle.c for the contract.
#pragma once

#include "sph_fastdiv.hpp"
#include "sph_internal.hpp"

namespace sph {

constexpr float kWildLimit = 1e15f;

__device__ __forceinline__ bool is_wild(float v) { return !(fabsf(v) < kWildLimit); }
// a density whose pressure k * (rho - rho0) (VelPos.compute:61,87) is wild, or that is itself beyond the denominator window of
// sph_fastdiv.hpp (the physical bound is ~27 * 32 * m * c / h^3 < 3e7 * dam_fill_rate)
__device__ __forceinline__ bool is_wild_density(const Dev &d, float rho)
{
    return is_wild(d.gas_const * (rho - d.rest_density)) || !(rho < 1e9f);
}

struct CellRange {
    uint32_t start, cnt;
};

// Bounds test of Density.compute:46 / VelPos.compute:73 + the bucket row of that cell:
// the first min(count, 32) slots, ascending particle id.
__device__ __forceinline__ CellRange cell_range(const Dev &d, int jx, int jy, int jz)
{
    CellRange r{0u, 0u};
    const int R = (int)d.R;
    if (jx >= 0 && jy >= 0 && jz >= 0 && jx < R && jy < R && jz < R) {
        const uint32_t c = (uint32_t)jx + (uint32_t)jy * d.R + (uint32_t)jz * d.R * d.R;
        r.start = d.cell_start[c];
        const uint32_t n = d.cell_start[c + 1] - r.start;
        r.cnt = n < kMaxPerVoxel ? n : kMaxPerVoxel;
    }
    return r;
}

// Density.compute:22-27 W() times _ParticleMass (:54), from abs_r2 = (rx*rx + ry*ry) + rz*rz (:24)
// FAST (tolerance mode, SphParams.exact_order = 0): mass * c / h^9 is one factor, no division.
template <bool FAST = false>
__device__ __forceinline__ float density_term_r2(const Dev &d, float abs_r2)
{
    const float diff_r2 = d.h2 - abs_r2;                                         // :25
    if (FAST) return 0.0f < diff_r2 ? d.w_scale * ((diff_r2 * diff_r2) * diff_r2) : 0.0f;
    const float w = 0.0f < diff_r2 ? d.c_poly6 * ((diff_r2 * diff_r2) * diff_r2) / d.h9 : 0.0f; // :26
    return d.mass * w;                                                           // :54
}
__device__ __forceinline__ float density_term(const Dev &d, float rx, float ry, float rz)
{
    return density_term_r2(d, (rx * rx + ry * ry) + rz * rz);                    // :24
}

struct Force {
    float px, py, pz;  // f_press
    float vx, vy, vz;  // f_vis
};

// VelPos.compute:86-95 for one listed candidate j != i whose fields are already in registers, in three steps so that
// callers can run TWO candidates through each step together (their sqrt / division chains interleave: a dependent fp32
// instruction issues only every ~8 cycles on gfx950):
//   pair_pre     everything up to the seven divisions (the numerators and the two denominators |r| and rho_j);
//   pair_min_key smallest magnitude among the nonzero numerators, as a bit key: below kMagKeyMin the shared-reciprocal
//                quotients of sph_fastdiv.hpp are not guaranteed to equal `/` and the WAVE takes the plain divisions;
//   pair_finish  the divisions (SHARED: one refined reciprocal per denominator; else the compiler's `/`) and the sums.
// `take` = false turns the candidate into a no-op (self / padding).  Written WITHOUT branches inside a step; the
// reference's guards become selects of an exact 0.0f, and adding +0.0f to an accumulator that is never -0 is the same
// as not adding (see sph_kernels_sweep.hpp).  FAST = false (wild launches: some |v|, |p| >= 1e15, rho >= 1e9, inf or
// NaN) evaluates sqrtf and `/` literally.
struct PairPre {
    float abs_r, rho_j, gv;
    float ngx, ngy, ngz;   // sc * r_ij          :37
    float ncp;             // p_i + p_j          :93
    float nvx, nvy, nvz;   // v_j - v_i          :94
    bool in_p, ok;
    bool sc_small;         // sc < 2^-27: a nonzero sc * r component may lie below the numerator window (pair_window_by_operands)
};

// RHO_GUARD false: `ok` is `take` alone -- for pair_finish_masked with a staged reciprocal that is 0 where rho_j <= epsilon (every
// quotient by rho_j is then an exact 0 and the pair adds +-0, like the select on the guard).
template <bool FAST, bool RHO_GUARD = true>
__device__ __forceinline__ PairPre pair_pre(const Dev &d, float rx, float ry, float rz, float rho_j, float vjx, float vjy,
                                            float vjz, float vix, float viy, float viz, float p_i, bool take)
{
    const float epsilon = 1e-6f;                                                 // :5
    const float h = d.h;
    PairPre q;
    const float p_j = d.gas_const * (rho_j - d.rest_density);                    // :87
    q.ok = RHO_GUARD ? take && (rho_j > epsilon) : take;                         // :91
    const float r2 = (rx * rx + ry * ry) + rz * rz;
    q.abs_r = FAST ? sqrt_unscaled(r2) : sqrtf(r2);                              // :35, :42
    const float diff_r = h - q.abs_r;                                            // :36
    q.in_p = (epsilon < diff_r) && (epsilon < q.abs_r);                          // :37
    const float sc = d.c_over_h6 * ((diff_r * diff_r) * diff_r);
    q.sc_small = sc < 0x1p-27f;
    q.ngx = sc * rx; q.ngy = sc * ry; q.ngz = sc * rz;
    q.ncp = p_i + p_j;                                                           // :93
    q.gv = q.abs_r < h ? d.c_over_h6 * (h - q.abs_r) : 0.0f;                     // :43
    q.nvx = vjx - vix; q.nvy = vjy - viy; q.nvz = vjz - viz;                     // :94
    q.rho_j = rho_j;
    return q;
}

__device__ __forceinline__ uint32_t pair_min_key(const PairPre &q)
{
    return umin3(umin3(mag_key(q.ngx), mag_key(q.ngy), mag_key(q.ngz)), umin3(mag_key(q.nvx), mag_key(q.nvy), mag_key(q.nvz)),
                 mag_key(q.ncp));
}

// The numerator window of sph_fastdiv.hpp from the OPERANDS instead of the seven numerators (pair_min_key: 7 shifts, 3 min3 and
// a compare per pair -- integer instructions cost 4.3 cycles per wave on gfx950 against 2.9 for a multiply-add, ~13 % of a pair).
// If no position component of the two particles is nonzero below 2^-40 and no velocity component or pressure nonzero below 2^-60
// (operand_tiny: tested once per PARTICLE by a kernel that stages its candidates), then
//   v_j - v_i, p_i + p_j   are 0 or >= 2^-60 * 2^-24 = 2^-84 in magnitude (a difference of two floats is a multiple of the smaller ulp);
//   r = pos_i - pos_j      has components that are 0 or >= 2^-40 * 2^-23 = 2^-63, so sc * r is 0 or >= 2^-90 when sc >= 2^-27
// and only `in_p && sc < 2^-27` is left to test per pair (with in_p false every quotient by |r| is dropped by its select).  The
// upper end of the window is the wild flags' as before.
__device__ __forceinline__ bool operand_tiny(const Dev &d, float x, float y, float z, float rho, float vx, float vy, float vz)
{
    constexpr uint32_t kKeyPos = ((127u - 40u) << 24) - 1u, kKeyVel = ((127u - 60u) << 24) - 1u;   // mag_key(2^-40), mag_key(2^-60)
    const float p = d.gas_const * (rho - d.rest_density);                        // :61 / :87, the same operations
    return umin3(mag_key(x), mag_key(y), mag_key(z)) < kKeyPos ||
           umin3(umin3(mag_key(vx), mag_key(vy), mag_key(vz)), mag_key(p), 0xFFFFFFFFu) < kKeyVel;
}

template <bool SHARED>
__device__ __forceinline__ void pair_finish(const PairPre &q, Force &f)
{
    float gpx, gpy, gpz, cp, qx, qy, qz;
    if (SHARED) {
        const Recip ra = recip_refined(q.abs_r), rr = recip_refined(q.rho_j), r2 = recip_twice(rr);
        gpx = div_shared(q.ngx, ra); gpy = div_shared(q.ngy, ra); gpz = div_shared(q.ngz, ra);
        cp = div_shared(q.ncp, r2);
        qx = div_shared(q.nvx, rr); qy = div_shared(q.nvy, rr); qz = div_shared(q.nvz, rr);
    } else {
        gpx = q.ngx / q.abs_r; gpy = q.ngy / q.abs_r; gpz = q.ngz / q.abs_r;     // :37
        cp = q.ncp / (2.0f * q.rho_j);                                           // :93
        qx = q.nvx / q.rho_j; qy = q.nvy / q.rho_j; qz = q.nvz / q.rho_j;        // :94
    }
    gpx = q.in_p ? gpx : 0.0f; gpy = q.in_p ? gpy : 0.0f; gpz = q.in_p ? gpz : 0.0f;
    const float tx = cp * gpx, ty = cp * gpy, tz = cp * gpz;
    const float ux = qx * q.gv, uy = qy * q.gv, uz = qz * q.gv;
    f.px += q.ok ? tx : 0.0f;
    f.py += q.ok ? ty : 0.0f;
    f.pz += q.ok ? tz : 0.0f;
    f.vx += q.ok ? ux : 0.0f;
    f.vy += q.ok ? uy : 0.0f;
    f.vz += q.ok ? uz : 0.0f;
}

// pair_finish<true> with the guards folded, for a launch that is not wild.  Every quantity that reaches a select is finite
// here or is thrown away by it:
//   * one select per sum on (ok && in_p) instead of three on in_p and three on ok: with in_p false the reference adds
//     cp * 0 = +-0 (cp finite: ok), which leaves a sum that is never -0 unchanged, as adding +0 does; a quotient by
//     |r| <= 1e-6 (inf, NaN) only ever reaches a select that drops it;
//   * the guard of grad_W_vis (:43) folded into the select of its sum the same way (q_xyz finite: |dv| < 2e15, rho_j > 1e-6);
//   * no v_div_fixup (div_shared_nofix, sph_fastdiv.hpp): inside the window of sph_fastdiv.hpp it only restores the sign of a zero quotient, and a zero
//     quotient of either sign contributes +-0.
// rr_r: the refined reciprocal of rho_j, recip_refined(q.rho_j).r -- a caller that holds it (k_velpos_tile stages it with the
// candidate) passes it in, everybody else computes it here.
__device__ __forceinline__ void pair_finish_folded(const Dev &d, const PairPre &q, Force &f, float rr_r)
{
    const Recip ra = recip_refined(q.abs_r), rr{q.rho_j, rr_r}, r2 = recip_twice(rr);
    const float gpx = div_shared_nofix(q.ngx, ra), gpy = div_shared_nofix(q.ngy, ra), gpz = div_shared_nofix(q.ngz, ra);
    const float cp = div_shared_nofix(q.ncp, r2);
    const float qx = div_shared_nofix(q.nvx, rr), qy = div_shared_nofix(q.nvy, rr), qz = div_shared_nofix(q.nvz, rr);
    const float gv = d.c_over_h6 * (d.h - q.abs_r);                              // :43 without its guard
    const bool sel_p = q.ok && q.in_p, sel_v = q.ok && (q.abs_r < d.h);
    const float tx = cp * gpx, ty = cp * gpy, tz = cp * gpz;
    const float ux = qx * gv, uy = qy * gv, uz = qz * gv;
    f.px += sel_p ? tx : 0.0f;
    f.py += sel_p ? ty : 0.0f;
    f.pz += sel_p ? tz : 0.0f;
    f.vx += sel_v ? ux : 0.0f;
    f.vy += sel_v ? uy : 0.0f;
    f.vz += sel_v ? uz : 0.0f;
}
// The same with the guards on OPERANDS (k_velpos_tile): |r| replaced by 1 where the pressure guard fails, the staged reciprocal of
// rho_j being 0 where rho_j <= epsilon (tile_put), every quotient is finite, and a sum needs one select on its scalar factor
// (cp, grad_W_vis) instead of three on its terms -- a failing pair adds cp_sel * gp = +-0.  Three selects per pair instead of six
// (a select costs 4.3 cycles per wave on gfx950, a multiply-add 2.9).
__device__ __forceinline__ void pair_finish_masked(const Dev &d, const PairPre &q, Force &f, float rr_r)
{
    const float abs_r_s = q.in_p ? q.abs_r : 1.0f;
    const Recip ra = recip_refined(abs_r_s), rr{q.rho_j, rr_r};
    const float gpx = div_shared_nofix(q.ngx, ra), gpy = div_shared_nofix(q.ngy, ra), gpz = div_shared_nofix(q.ngz, ra);
    // :93 divides by 2 rho_j: half the quotient by rho_j (both scalings by two are exact: rho_j < 2^30, the quotient >= 2^-120)
    const float cp = 0.5f * div_shared_nofix(q.ncp, rr);
    const float qx = div_shared_nofix(q.nvx, rr), qy = div_shared_nofix(q.nvy, rr), qz = div_shared_nofix(q.nvz, rr);
    // :43, c (h - |r|) where |r| < h and 0 elsewhere, is max(c (h - |r|), 0): h - |r| is exact in sign (positive iff |r| < h) and c > 0
    const float gv = __builtin_fmaxf(d.c_over_h6 * (d.h - q.abs_r), 0.0f);
    const float cps = (q.ok && q.in_p) ? cp : 0.0f, gvs = q.ok ? gv : 0.0f;
    f.px += cps * gpx;
    f.py += cps * gpy;
    f.pz += cps * gpz;
    f.vx += qx * gvs;
    f.vy += qy * gvs;
    f.vz += qz * gvs;
}
__device__ __forceinline__ void pair_finish_folded(const Dev &d, const PairPre &q, Force &f)
{
    pair_finish_folded(d, q, f, recip_refined(q.rho_j).r);
}

// Two candidates (a then b: the reference's order) of a launch that is not "wild".  FOLDED: pair_finish_folded.
template <bool FOLDED = false>
__device__ __forceinline__ void pair_force_two(const Dev &d, const PairPre &a, const PairPre &b, Force &f)
{
    const uint32_t ka = pair_min_key(a), kb = pair_min_key(b);
    if (__builtin_expect(__any((ka < kb ? ka : kb) < kMagKeyMin), 0)) {          // wave-uniform; practically never
        pair_finish<false>(a, f);
        pair_finish<false>(b, f);
    } else if (FOLDED) {
        pair_finish_folded(d, a, f);
        pair_finish_folded(d, b, f);
    } else {
        pair_finish<true>(a, f);
        pair_finish<true>(b, f);
    }
}

// TOLERANCE MODE (SphParams.exact_order = 0): the same pair, the same guards and the same summation order, with the seven
// IEEE divisions replaced by two hardware reciprocals (v_rcp_f32, 1 ulp) and sqrt by v_sqrt_f32 (1 ulp), products
// re-associated so that each sum takes one select and three multiply-adds.  ~45 VALU instructions per pair instead of ~135;
// relative error of a pair term ~1e-6, the parity bar of this mode is BASELINE.json's 1e-4 (tests/test_gpu_tolerance_mode.py).
// No "wild" detour: extreme operands give inf / NaN here as in IEEE arithmetic, not necessarily in the same lanes.
__device__ __forceinline__ void pair_fast(const Dev &d, float rx, float ry, float rz, float rho_j, float vjx, float vjy,
                                          float vjz, float vix, float viy, float viz, float p_i, Force &f, bool take, float rrho)
{
    // rrho = v_rcp_f32(rho_j): passed in by a caller that holds it (k_velpos_tile stages it with the candidate)
    const float epsilon = 1e-6f;                                                 // :5
    const float h = d.h;
    const float r2 = __builtin_fmaf(rz, rz, __builtin_fmaf(ry, ry, rx * rx));
    const float abs_r = __builtin_amdgcn_sqrtf(r2);                              // :35, :42
    const float rinv = __builtin_amdgcn_rcpf(abs_r);
    const float diff_r = h - abs_r;                                              // :36
    const bool ok = take && (rho_j > epsilon);                                   // :91
    const bool sel_p = ok && (epsilon < diff_r) && (epsilon < abs_r);            // :37
    const bool sel_v = ok && (abs_r < h);                                        // :43
    const float p_j = d.gas_const * (rho_j - d.rest_density);                    // :87
    const float cd = d.c_over_h6 * diff_r;                                       // shared by :37 (x diff_r^2 / |r|) and :43
    const float kp = ((p_i + p_j) * (0.5f * rrho)) * ((cd * (diff_r * diff_r)) * rinv);   // :93 x :37
    const float kv = cd * rrho;                                                  // :94 x :43
    const float sp = sel_p ? kp : 0.0f, sv = sel_v ? kv : 0.0f;
    f.px = __builtin_fmaf(sp, rx, f.px);
    f.py = __builtin_fmaf(sp, ry, f.py);
    f.pz = __builtin_fmaf(sp, rz, f.pz);
    f.vx = __builtin_fmaf(sv, vjx - vix, f.vx);
    f.vy = __builtin_fmaf(sv, vjy - viy, f.vy);
    f.vz = __builtin_fmaf(sv, vjz - viz, f.vz);
}

__device__ __forceinline__ void pair_fast(const Dev &d, float rx, float ry, float rz, float rho_j, float vjx, float vjy,
                                          float vjz, float vix, float viy, float viz, float p_i, Force &f, bool take)
{
    pair_fast(d, rx, ry, rz, rho_j, vjx, vjy, vjz, vix, viy, viz, p_i, f, take, __builtin_amdgcn_rcpf(rho_j));
}

// One candidate; `literal` = the launch is wild (wave-uniform).
__device__ __forceinline__ void pair_force_vals(const Dev &d, float rx, float ry, float rz, float rho_j,
                                                float vjx, float vjy, float vjz, float vix, float viy,
                                                float viz, float p_i, Force &f, bool take, bool literal)
{
    if (d.fast) {                                  // tolerance mode (launch-uniform)
        pair_fast(d, rx, ry, rz, rho_j, vjx, vjy, vjz, vix, viy, viz, p_i, f, take);
        return;
    }
    if (literal) {
        pair_finish<false>(pair_pre<false>(d, rx, ry, rz, rho_j, vjx, vjy, vjz, vix, viy, viz, p_i, take), f);
        return;
    }
    const PairPre q = pair_pre<true>(d, rx, ry, rz, rho_j, vjx, vjy, vjz, vix, viy, viz, p_i, take);
    if (__builtin_expect(__any(pair_min_key(q) < kMagKeyMin), 0)) pair_finish<false>(q, f);
    else pair_finish_folded(d, q, f);                  // (not wild here: every value that reaches a select is finite or dropped)
}

__device__ __forceinline__ void pair_force(const Dev &d, uint32_t j, float pix, float piy, float piz,
                                           float vix, float viy, float viz, float p_i, Force &f, bool literal)
{
    const float4 rv = d.cur.rv[j];
    const Pos3 pj = load_pos(d, d.cur, j);
    pair_force_vals(d, pix - pj.x, piy - pj.y, piz - pj.z, rv.x, rv.y, rv.z, rv.w, vix, viy,
                    viz, p_i, f, true, literal);                                 // :86-89
}

// VelPos.compute:100-157: scaling, wall force, NaN guard, integrate, clamp, store.  Returns the new position.
// FAST (tolerance mode, launch-uniform): the five divisions by rho_i and m become two reciprocals.
template <int PREC = -1, bool FAST = false>
__device__ __forceinline__ Pos3 integrate_store(const Dev &d, uint32_t s, float pix, float piy, float piz,
                                                float vix, float viy, float viz, float viw, float rho_i,
                                                Force f)
{
    const float epsilon = 1e-6f;
    const float h = d.h, m = d.mass;
    const float dt = d.dt_ptr ? *d.dt_ptr : d.dt;
    if (rho_i > epsilon) {                                                       // :101
        const float sp = FAST ? (m * m) * __builtin_amdgcn_rcpf(rho_i) : m * m / rho_i;                         // :103
        f.px *= sp; f.py *= sp; f.pz *= sp;
        const float sv = FAST ? (d.viscosity * m * m) * __builtin_amdgcn_rcpf(rho_i) : d.viscosity * m * m / rho_i;   // :104
        f.vx *= sv; f.vy *= sv; f.vz *= sv;
    }
    float fwx = 0.0f, fwy = 0.0f, fwz = 0.0f;                                    // :107
    float wrx = 0.0f, wry = 0.0f, wrz = 0.0f;                                    // :108
    if (pix < h) wrx += h - pix;                                                 // :110-113
    else if (pix > 1.0f - h) wrx += 1.0f - pix - h;
    if (piy < h) wry += h - piy;                                                 // :115-118
    else if (piy > 1.0f - h) wry += 1.0f - piy - h;
    if (piz < h) wrz += h - piz;                                                 // :120-123
    else if (piz > 1.0f - h) wrz += 1.0f - piz - h;
    const float max_r = fmaxf(fabsf(wrx), fmaxf(fabsf(wry), fabsf(wrz)));        // :125-126
    const float dpx = (wrx != 0.0f) ? d.damping : 0.0f;                          // :128-131
    const float dpy = (wry != 0.0f) ? d.damping : 0.0f;
    const float dpz = (wrz != 0.0f) ? d.damping : 0.0f;
    if (max_r > 0.0f) {                                                          // :133
        const float dd = (dpx * vix + dpy * viy) + dpz * viz;                    // :135 scalar
        fwx = wrx * d.stiffness - dd;
        fwy = wry * d.stiffness - dd;
        fwz = wrz * d.stiffness - dd;
        fwx *= m; fwy *= m; fwz *= m;                                            // :136
    }
    const float rm = FAST ? __builtin_amdgcn_rcpf(m) : 0.0f;
    const float ax = d.gx + (FAST ? ((f.px + f.vx) + fwx) * rm : ((f.px + f.vx) + fwx) / m);   // :139
    const float ay = d.gy + (FAST ? ((f.py + f.vy) + fwy) * rm : ((f.py + f.vy) + fwy) / m);
    const float az = d.gz + (FAST ? ((f.pz + f.vz) + fwz) * rm : ((f.pz + f.vz) + fwz) / m);
    float vox, voy, voz, vow;
    if (ax != ax || ay != ay || az != az) {                                      // :143
        vox = 0.0f; voy = 0.0f; voz = 0.0f; vow = 0.003f;                        // :145-146
        atomicAdd(&d.counters[kCntNaN], 1ull);
    } else {
        vox = vix + ax * dt;                                                   // :150
        voy = viy + ay * dt;
        voz = viz + az * dt;
        vow = viw + 1.0f;
    }
    const Pos3 pn{clamp01(pix + dt * vox), clamp01(piy + dt * voy), clamp01(piz + dt * voz)};  // :153-154
    store_pos<PREC>(d, d.nxt, s, pn);  // (precision 1: rounded to the fp16 record of the slot's reference cell HERE, nowhere else)
    if (d.nxt.p4) store_p4<PREC>(d, d.nxt, s, pn);
    d.nxt.rv[s] = make_float4(rho_i, vox, voy, voz);                             // :156 (+ rho rides along)
    d.nxt.vw[s] = vow;
    if (is_wild(vox) || is_wild(voy) || is_wild(voz)) d.counters[kCntWildV0 + (d.wild_v_cur ^ 1)] = 1ull;
    return pn;
}

// brick of a bin (row-major bricks of kNlBX x kNlBY x kNlBZ bins, as the VelPos brick kernels number them)
__device__ __forceinline__ uint32_t nl_brick_of_key(const Dev &d, uint32_t key)
{
    const uint32_t bx = key % d.R, by = (key / d.R) % d.R, bz = key / (d.R * d.R);
    return bx / kNlBX + nl_bricks_x(d.R) * (by / kNlBY + nl_bricks_y(d.R) * (bz / kNlBZ));
}

__device__ __forceinline__ bool launch_is_wild(const Dev &d)
{
    return d.force_slow != 0u || d.counters[kCntWildP] != 0ull || d.counters[kCntWildV0 + d.wild_v_cur] != 0ull;
}

// Neighbour lists may not be used for this launch: wild values need the literal loop, and after a grid build that saw
// positions outside the unit box a bin says nothing about where its members are (sph_kernels_list.hip, condition 2).
__device__ __forceinline__ bool nl_blocked(const Dev &d)
{
    return launch_is_wild(d) || d.counters[kCntIrregular] != 0ull;
}

// min / max over the 16 lanes of a DPP row in four VALU steps (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror:
// every step pairs each lane with one that holds the other half), no LDS traffic
template <bool MAX>
__device__ __forceinline__ float row16_reduce(float v)
{
#define SPH_DPP_STEP(ctrl)                                                                                               \
    {                                                                                                                    \
        const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, 0xF, 0xF, false)); \
        v = MAX ? fmaxf(v, o) : fminf(v, o);                                                                             \
    }
    SPH_DPP_STEP(0xB1) SPH_DPP_STEP(0x4E) SPH_DPP_STEP(0x141) SPH_DPP_STEP(0x140)
#undef SPH_DPP_STEP
    return v;
}

// ---- per-brick displacement ranges (Dev::box_mode; sph_kernels_list.hip) ----------------------------------------------------------
// Fields of Dev read from the kernel-argument segment at the point of use (Dev is every kernel's first argument) behind an opaque
// pointer: the compiler cannot keep them in registers across the caller's hot loop.  k_velpos_tile runs at its register limit,
// and five more live scalars spilled a vector register there.
typedef const __attribute__((address_space(4))) Dev *KernargDev;
__device__ __forceinline__ KernargDev late_kernarg_dev()
{
    KernargDev p = (KernargDev)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// A wave whose `on` lanes all belong to ONE brick (k_velpos_brick: the workgroup's) adds their displacements to the brick's range
// of this sub-step: reduction in DPP steps within the rows of 16 lanes and two cross-row exchanges, six atomics from lane 0.
__device__ __forceinline__ void box_update_uniform(const Dev &d, bool on, uint32_t brick, float dx, float dy, float dz)
{
    const float eps = d.prec ? d.h * 0x1p-14f : 0.0f;
    float lo[3] = {on ? dx - eps : INFINITY, on ? dy - eps : INFINITY, on ? dz - eps : INFINITY};
    float hi[3] = {on ? dx + eps : -INFINITY, on ? dy + eps : -INFINITY, on ? dz + eps : -INFINITY};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        lo[a] = row16_reduce<false>(lo[a]);
        hi[a] = row16_reduce<true>(hi[a]);
        lo[a] = fminf(lo[a], __shfl_xor(lo[a], 16, kWave));
        hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], 16, kWave));
        lo[a] = fminf(lo[a], __shfl_xor(lo[a], 32, kWave));
        hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], 32, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0 && lo[0] <= hi[0]) {                    // (somebody was on)
        uint32_t *box = d.boxrec + ((size_t)d.box_par * d.n_bricks + brick) * kBoxRecWords;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            atomicMin(box + a, float_key(lo[a]));
            atomicMax(box + 3 + a, float_key(hi[a]));
        }
    }
}

// Adds the displacements (dx, dy, dz) of the lanes that are `on` to the per-sub-step range of their bricks (Dev::dbox).  Every
// lane of the wave must call it.  One round per distinct brick among the lanes -- one for the brick kernels, a few for the
// kernels that walk slots; the range of a round is reduced across the wave and its first lane issues the six atomics.
// eps: what the stored position may differ by from the one the displacement was taken from (precision 1: the fp16 record).
__device__ __forceinline__ void box_update_wave(const Dev &d, bool on, uint32_t brick, float dx, float dy, float dz)
{
    const float eps = d.prec ? d.h * 0x1p-14f : 0.0f;
    unsigned long long todo = __ballot(on);
    const int lane = threadIdx.x & (kWave - 1);
    while (todo) {                                                               // wave-uniform
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t b = __shfl(brick, leader, kWave);
        const bool mine = on && brick == b;
        float lo[3] = {mine ? dx - eps : INFINITY, mine ? dy - eps : INFINITY, mine ? dz - eps : INFINITY};
        float hi[3] = {mine ? dx + eps : -INFINITY, mine ? dy + eps : -INFINITY, mine ? dz + eps : -INFINITY};
        // (displacements are differences of clamped positions: finite)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            lo[a] = row16_reduce<false>(lo[a]);
            hi[a] = row16_reduce<true>(hi[a]);
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], 16, kWave));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], 16, kWave));
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], 32, kWave));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], 32, kWave));
        }
        if (lane == leader) {
            uint32_t *box = d.boxrec + ((size_t)d.box_par * d.n_bricks + b) * kBoxRecWords;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                atomicMin(box + a, float_key(lo[a]));
                atomicMax(box + 3 + a, float_key(hi[a]));
            }
        }
        todo &= ~__ballot(mine);
    }
}

// k_velpos_tile's: the range of a wave in six DPP steps per value -- v_min_f32 / v_max_f32 with the DPP operand, written out
// because the compiler emits v_mov_dpp + canonicalising v_max + v_min per step, and this code runs at the END of every wave's
// life, where nothing hides it (what the kernel pays for its ranges, in any variant tried: ~4 % = 0.05 ms per launch, against
// the 0.1 ms pass over all positions and the launch it replaces: profiles/r03b_experiments.txt).  Within the rows of 16
// lanes (quad_perm x 2, row_half_mirror, row_mirror), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3:
// lane 63 holds the wave's range and writes its six keys to the wave's LDS slot; the workgroup combines the slots after its
// barrier and STORES -- its brick's targets are its own, the kernels that integrate the few others (handed over, lists of
// global slots) run later in the stream and add theirs with atomics.
__device__ __forceinline__ void box_wave_range_to_lds(uint32_t *s_wave6, bool on, float dx, float dy, float dz, float eps)
{
    float lo[3] = {on ? dx - eps : INFINITY, on ? dy - eps : INFINITY, on ? dz - eps : INFINITY};
    float hi[3] = {on ? dx + eps : -INFINITY, on ? dy + eps : -INFINITY, on ? dz + eps : -INFINITY};
    // (s_nop 1: two wait states between a VALU write of a register and a DPP read of it)
#define SPH_DPP6(op, v)                                                                           \
    asm volatile("s_nop 1\n\t" op " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
                 "s_nop 1\n\t" op " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
                 "s_nop 1\n\t" op " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"     \
                 "s_nop 1\n\t" op " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"          \
                 "s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"        \
                 "s_nop 1\n\t" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"              \
                 : "+v"(v))
#pragma unroll
    for (int a = 0; a < 3; a++) {
        SPH_DPP6("v_min_f32_dpp", lo[a]);
        SPH_DPP6("v_max_f32_dpp", hi[a]);
    }
#undef SPH_DPP6
    if ((threadIdx.x & (kWave - 1)) == kWave - 1) {
        const bool any = lo[0] <= hi[0];                 // (nobody on: the empty range's keys, which lose every min / max)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            s_wave6[a] = any ? float_key(lo[a]) : kBoxKeyEmptyLo;
            s_wave6[3 + a] = any ? float_key(hi[a]) : kBoxKeyEmptyHi;
        }
    }
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    return v;
}

// the same for per-lane values whose wave total fits 32 bits (half the instructions: the kernels that call it per wave are
// VALU-bound)
__device__ __forceinline__ void stat_add_u32(const Dev &d, int stat, uint32_t v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && v)
        atomicAdd(&d.stripes[((size_t)stat * kStatStripes + (blockIdx.x % kStatStripes)) * kStatPad], (unsigned long long)v);
}
// the wave's sum of a u32 in six DPP steps (VALU only; lane 63 holds it), for kernels whose waves end on it: the exchanges of
// stat_add_u32 go through the LDS pipe, ~12 dependent round trips at the end of a wave's life
__device__ __forceinline__ uint32_t wave_sum_u32_dpp(uint32_t v)
{
    // row_shr:1,2,3 leave the sum of each group of 4 in its last lane; bank-masked row_shr:4 and :8 the sum of a row of 16 in
    // lane 15; row_bcast:15 / :31 carry it across the rows (the wave reduction of the AMDGPU atomic optimiser)
    asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "+v"(v));
    return v;
}
__device__ __forceinline__ void stat_add_u32_dpp(const Dev &d, int stat, uint32_t v)
{
    v = wave_sum_u32_dpp(v);
    if ((threadIdx.x & (kWave - 1)) == kWave - 1 && v)
        atomicAdd(&d.stripes[((size_t)stat * kStatStripes + (blockIdx.x % kStatStripes)) * kStatPad], (unsigned long long)v);
}
// ... and for a per-lane flag: one ballot
__device__ __forceinline__ void stat_count(const Dev &d, int stat, bool flag)
{
    const unsigned long long m = __ballot(flag);
    if ((threadIdx.x & (kWave - 1)) == 0 && m)
        atomicAdd(&d.stripes[((size_t)stat * kStatStripes + (blockIdx.x % kStatStripes)) * kStatPad], (unsigned long long)__popcll(m));
}

// wave total of `v` into the block's stripe of statistic `stat` (every lane of the wave must call it)
__device__ __forceinline__ void stat_add(const Dev &d, int stat, unsigned long long v)
{
    v = wave_sum_u64(v);
    if ((threadIdx.x & (kWave - 1)) == 0 && v)
        atomicAdd(&d.stripes[((size_t)stat * kStatStripes + (blockIdx.x % kStatStripes)) * kStatPad], v);
}

}  // namespace sph
