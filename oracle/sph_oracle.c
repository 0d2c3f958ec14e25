/*
 * sph_oracle.c -- CPU restatement of the reference's per-frame particle update.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped engine.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED BY THE REFERENCE: leandro-barcelos/SPHFluidSimulation ships no
 * tests, golden vectors or fixtures, and its hot path (HLSL compute driven by Unity
 * C#) cannot be compiled or run in this image (no Unity / dxc / dotnet / mono).
 * The pins are therefore (1) the analytic known-answer tests derived from the
 * shader text (tests/test_oracle_kat.py) and (2) an independent pure-Python
 * restatement (oracle/sph_oracle_py.py) that must agree bit-for-bit at small N.
 *
 * What is restated, statement by statement (paths relative to /root/reference):
 *   Assets/Resources/Bucket.compute:15-36   CSMain       -> oracle_bucket()
 *   Assets/Resources/Bucket.compute:38-53   ClearBucket  -> oracle_clear_bucket()
 *   Assets/Resources/Clear.compute:6-12     ClearFloat   -> oracle_clear_float()
 *   Assets/Resources/Density.compute:22-60  W, CSMain    -> oracle_density()
 *   Assets/Resources/VelPos.compute:33-158  grad_W_*, CSMain -> oracle_velpos()
 *   Assets/Scripts/SphFluidSimulation.cs:84-85,159,176,236-237,255 derived scalars
 *                                                        -> oracle_derive()
 *   Assets/Scripts/SphFluidSimulation.cs:96-102,203-269,290-293 frame schedule
 *                                                        -> oracle_sim_update()
 *
 * Floating-point contract (the DEFINED oracle; HLSL compilers are free to fuse and
 * approximate, so the real GPU reference is not bit-reproducible anyway):
 *   strict IEEE-754 binary32, every statement evaluated left to right in source
 *   order, no FMA contraction, correctly rounded / and sqrt, dot(a,b) =
 *   (a.x*b.x + a.y*b.y) + a.z*b.z, length(r) = sqrt(dot(r,r)),
 *   clamp(x,0,1) = fmin(fmax(x,0),1) (NaN -> 0 like D3D max/min),
 *   int(x) = truncation toward zero, saturating, NaN -> 0 (v_cvt_i32_f32),
 *   uint(x) = truncation, negative/NaN -> 0, saturating (v_cvt_u32_f32).
 * Build with: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 *
 * Slot order (Bucket.compute:30-35 is a CAS race on the GPU): the canonical order
 * is ascending particle id, i.e. what a serial execution of CSMain produces; a
 * cell with more than 32 particles keeps the 32 lowest ids.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NUM_THREADS 32            /* Bucket.compute:1 (dispatch tiling only) */
#define MAX_PARTICLES_PER_VOXEL 32u /* Bucket.compute:2, Density.compute:2, VelPos.compute:2 */

typedef struct OracleParams {
    uint32_t num_particles;     /* _NumParticles: NextPowerOfTwo(particleNumber) SphFluidSimulation.cs:84 */
    uint32_t res;               /* _ParticleResolution: (int)sqrt(N)             SphFluidSimulation.cs:85 */
    uint32_t bucket_resolution; /* _BucketResolution                             SphFluidSimulation.cs:37 */
    float particle_mass;        /* _ParticleMass = damFillRate / particleNumber  :176 */
    float h;                    /* _EffectiveRadius  = 1f/(R-1)                  :159 */
    float h2;                   /* _EffectiveRadius2 = h*h                       :236 */
    float h6;                   /* _EffectiveRadius6 = Mathf.Pow(h,6)            :255 */
    float h9;                   /* _EffectiveRadius9 = Mathf.Pow(h,9)            :237 */
    float viscosity;            /* _Viscosity      :258 */
    float gas_const;            /* _GasConst       :259 */
    float rest_density;         /* _RestDensity    :260 */
    float stiffness;            /* _StiffnessCoeff :261 */
    float damping;              /* _DampingCoeff   :262 */
    uint32_t substeps;          /* 5   SphFluidSimulation.cs:101 */
    float dt_divisor;           /* 25  SphFluidSimulation.cs:102 */
    int32_t num_threads;        /* OpenMP threads for the sweeps (<=0: runtime default) */
} OracleParams;

/* ---- HLSL intrinsic semantics --------------------------------------------------- */

static inline int32_t hlsl_int(float f)
{
    if (f != f) return 0;
    if (f <= -2147483648.0f) return INT32_MIN;
    if (f >= 2147483648.0f) return INT32_MAX;
    return (int32_t)f; /* C conversion truncates toward zero */
}

static inline uint32_t hlsl_uint(float f)
{
    if (!(f > 0.0f)) return 0u; /* negative, -0, NaN */
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

static inline float hlsl_clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

static inline float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return (ax * bx + ay * by) + az * bz;
}

/* Mathf.NextPowerOfTwo: smallest power of two >= v. */
static uint32_t next_power_of_two(uint32_t v)
{
    uint32_t p = 1;
    if (v == 0) return 0;
    while (p < v && p < 0x80000000u) p <<= 1;
    return p;
}

/* ---- derived parameters: SphFluidSimulation.cs:84-85,159,176,236-237,255 -------- */

int oracle_derive(uint32_t particle_number, uint32_t bucket_resolution, float dam_fill_rate,
                  float viscosity, float rest_density, float gas_constant,
                  float stiffness, float damping, OracleParams *out)
{
    if (!out || bucket_resolution < 2 || bucket_resolution > 256 || particle_number == 0) return -1;
    memset(out, 0, sizeof *out);
    uint32_t N = next_power_of_two(particle_number);          /* :84 */
    /* :85  (int)Mathf.Sqrt(N); Mathf.Sqrt is (float)Math.Sqrt((double)f) */
    uint32_t res = (uint32_t)(int32_t)(float)sqrt((double)(float)N);
    out->num_particles = N;
    out->res = res;
    out->bucket_resolution = bucket_resolution;
    out->h = 1.0f / (float)((int32_t)bucket_resolution - 1);  /* :159 */
    out->particle_mass = dam_fill_rate / (float)(int32_t)N;   /* :176 float / int */
    out->h2 = out->h * out->h;                                /* :236 */
    out->h9 = (float)pow((double)out->h, 9.0);                /* :237 Mathf.Pow */
    out->h6 = (float)pow((double)out->h, 6.0);                /* :255 */
    out->viscosity = viscosity;
    out->rest_density = rest_density;
    out->gas_const = gas_constant;
    out->stiffness = stiffness;
    out->damping = damping;
    out->substeps = 5;                                        /* :101 */
    out->dt_divisor = 25.0f;                                  /* :102 */
    out->num_threads = 0;
    return 0;
}

uint64_t oracle_bucket_len(const OracleParams *p)
{
    uint64_t R = p->bucket_resolution;
    return R * R * R * MAX_PARTICLES_PER_VOXEL;               /* SphFluidSimulation.cs:195 */
}

static int threads_of(const OracleParams *p)
{
#ifdef _OPENMP
    return p->num_threads > 0 ? p->num_threads : omp_get_max_threads();
#else
    (void)p;
    return 1;
#endif
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- Bucket.compute:38-53 ClearBucket ------------------------------------------- */

void oracle_clear_bucket(const OracleParams *p, uint32_t *bucket)
{
    const uint64_t n = oracle_bucket_len(p);
    const uint32_t N = p->num_particles;
    int nt = threads_of(p);
    (void)nt;
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int64_t i = 0; i < (int64_t)n; i++) bucket[i] = N;   /* :51 */
}

/* ---- Bucket.compute:15-36 CSMain (serial => ascending-id slot order) ------------ */

void oracle_bucket(const OracleParams *p, const float *pos4, uint32_t *bucket)
{
    const uint32_t res = p->res, N = p->num_particles, R = p->bucket_resolution;
    const uint64_t len = oracle_bucket_len(p);
    const float Rm1 = (float)(R - 1u);     /* :27 float3 * uint: (R-1) in uint, then to float */
    const float Rf = (float)R;
    for (uint32_t y = 0; y < res; y++)
        for (uint32_t x = 0; x < res; x++) {               /* :21 bounds, :24 id_1d */
            const uint32_t id_1d = y * res + x;
            const float *pos = pos4 + 4ull * id_1d;         /* :26 */
            const float ix = (float)hlsl_int(pos[0] * Rm1); /* :27 float3 idx_3d = int3(..) */
            const float iy = (float)hlsl_int(pos[1] * Rm1);
            const float iz = (float)hlsl_int(pos[2] * Rm1);
            /* :28 float arithmetic (float3 component * uint -> float), then -> uint */
            uint32_t index = hlsl_uint((ix + iy * Rf) + iz * Rf * Rf);
            index *= MAX_PARTICLES_PER_VOXEL;               /* :29 (uint wrap) */
            for (uint32_t j = 0; j < MAX_PARTICLES_PER_VOXEL; j++) { /* :30 */
                const uint64_t a = (uint64_t)index + j;
                if (a >= len) continue; /* out-of-bounds UAV atomic: no write, returns 0 != N */
                if (bucket[a] == N) {   /* :33 InterlockedCompareExchange */
                    bucket[a] = id_1d;
                    break;              /* :34 */
                }
            }
        }
}

/* ---- Clear.compute:6-12 ClearFloat ---------------------------------------------- */

void oracle_clear_float(const OracleParams *p, float *tex)
{
    memset(tex, 0, sizeof(float) * (size_t)p->res * p->res);
}

/* ---- Density.compute:22-27 W ----------------------------------------------------- */

static inline float W_poly6(const OracleParams *p, float rx, float ry, float rz)
{
    const float PI = 3.14159265358979323846264338327950288f;  /* :5 */
    const float c = 315.0f / (64.0f * PI);                    /* :23 */
    const float abs_r2 = dot3(rx, ry, rz, rx, ry, rz);        /* :24 */
    const float diff_r2 = p->h2 - abs_r2;                     /* :25 */
    /* :26  0<diff ? c*POW3(diff)/h9 : 0 ;  POW3(x) = ((x)*(x)*(x)) */
    return 0.0f < diff_r2 ? c * ((diff_r2 * diff_r2) * diff_r2) / p->h9 : 0.0f;
}

/* ---- Density.compute:29-60 CSMain ------------------------------------------------ */

void oracle_density(const OracleParams *p, const float *pos4, const uint32_t *bucket, float *rho_tex)
{
    const uint32_t res = p->res, N = p->num_particles, R = p->bucket_resolution;
    const float Rm1 = (float)(R - 1u);
    const int64_t count = (int64_t)res * res;
    int nt = threads_of(p);
    (void)nt;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nt)
    for (int64_t id = 0; id < count; id++) {
        const float *pi = pos4 + 4 * id;                       /* :38 */
        const float pix = pi[0], piy = pi[1], piz = pi[2];
        const float ix = (float)hlsl_int(pix * Rm1);           /* :39 */
        const float iy = (float)hlsl_int(piy * Rm1);
        const float iz = (float)hlsl_int(piz * Rm1);
        float rho = 0.0f;                                      /* :41 */
        for (int xo = -1; xo <= 1; xo++)                       /* :42-44 */
            for (int yo = -1; yo <= 1; yo++)
                for (int zo = -1; zo <= 1; zo++) {
                    const int32_t jx = hlsl_int(ix + (float)xo); /* :45 float3 + int3 -> int3 */
                    const int32_t jy = hlsl_int(iy + (float)yo);
                    const int32_t jz = hlsl_int(iz + (float)zo);
                    if (jx >= 0 && jy >= 0 && jz >= 0 &&
                        (uint32_t)jx < R && (uint32_t)jy < R && (uint32_t)jz < R) { /* :46 */
                        uint32_t bkt = (uint32_t)jx + (uint32_t)jy * R + (uint32_t)jz * R * R; /* :48 */
                        bkt *= MAX_PARTICLES_PER_VOXEL;        /* :49 */
                        for (uint32_t j2 = 0; j2 < MAX_PARTICLES_PER_VOXEL; j2++) { /* :50 */
                            const uint32_t j = bucket[bkt + j2]; /* :51 */
                            if (j == N) break;                 /* :52 */
                            /* :53 j_2D = (j % res, j / res): texel (x,y) <-> id y*res+x */
                            const float *pj = pos4 + 4ull * ((j / res) * res + (j % res));
                            rho += p->particle_mass *
                                   W_poly6(p, pix - pj[0], piy - pj[1], piz - pj[2]); /* :54 */
                        }
                    }
                }
        rho_tex[id] = rho;                                     /* :59 */
    }
}

/* ---- VelPos.compute:33-158 -------------------------------------------------------- */

void oracle_velpos(const OracleParams *p, const float *pos4_r, const float *vel4_r,
                   const float *rho_tex, const uint32_t *bucket,
                   float *pos4_w, float *vel4_w, float time_step)
{
    const uint32_t res = p->res, N = p->num_particles, R = p->bucket_resolution;
    const float Rm1 = (float)(R - 1u);
    const float epsilon = 1e-6f;                               /* :5 */
    const float PI = 3.14159265358979323846264338327950288f;   /* :6 */
    const float gx = 0.0f, gy = -9.8f, gz = 0.0f;              /* :7 */
    const float c = 45.0f / PI;                                /* :34, :41 */
    const float h = p->h, h6 = p->h6, m = p->particle_mass;
    const int64_t count = (int64_t)res * res;
    int nt = threads_of(p);
    (void)nt;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nt)
    for (int64_t id = 0; id < count; id++) {
        const uint32_t id_1d = (uint32_t)id;                   /* :55 */
        const float *pi = pos4_r + 4 * id;                     /* :57 */
        const float pix = pi[0], piy = pi[1], piz = pi[2];
        const float ix = (float)hlsl_int(pix * Rm1);           /* :58 */
        const float iy = (float)hlsl_int(piy * Rm1);
        const float iz = (float)hlsl_int(piz * Rm1);
        const float rho_i = rho_tex[id];                       /* :60 */
        const float p_i = p->gas_const * (rho_i - p->rest_density); /* :61 */
        const float *vi = vel4_r + 4 * id;                     /* :62 */
        const float vix = vi[0], viy = vi[1], viz = vi[2], viw = vi[3];

        float fpx = 0.0f, fpy = 0.0f, fpz = 0.0f;              /* :64 */
        float fvx = 0.0f, fvy = 0.0f, fvz = 0.0f;              /* :65 */

        for (int xo = -1; xo <= 1; xo++)                       /* :67-69 */
            for (int yo = -1; yo <= 1; yo++)
                for (int zo = -1; zo <= 1; zo++) {
                    const int32_t jx = hlsl_int(ix + (float)xo); /* :71 */
                    const int32_t jy = hlsl_int(iy + (float)yo);
                    const int32_t jz = hlsl_int(iz + (float)zo);
                    if (!(jx >= 0 && jy >= 0 && jz >= 0 &&
                          (uint32_t)jx < R && (uint32_t)jy < R && (uint32_t)jz < R)) /* :73 */
                        continue;
                    uint32_t bkt = (uint32_t)jx + (uint32_t)jy * R + (uint32_t)jz * R * R; /* :75 */
                    bkt *= MAX_PARTICLES_PER_VOXEL;            /* :76 */
                    for (uint32_t j2 = 0; j2 < MAX_PARTICLES_PER_VOXEL; j2++) { /* :77 */
                        const uint32_t j = bucket[bkt + j2];   /* :79 */
                        if (j == N) break;                     /* :81 */
                        if (j == id_1d) continue;              /* :82 */
                        const uint64_t jt = (uint64_t)(j / res) * res + (j % res); /* :84 */
                        const float rho_j = rho_tex[jt];       /* :86 */
                        const float p_j = p->gas_const * (rho_j - p->rest_density); /* :87 */
                        const float *pj = pos4_r + 4 * jt;     /* :88 */
                        const float rx = pix - pj[0], ry = piy - pj[1], rz = piz - pj[2]; /* :89 */
                        if (rho_j > epsilon) {                 /* :91 */
                            /* grad_W_press :33-38 */
                            float gpx = 0.0f, gpy = 0.0f, gpz = 0.0f;
                            {
                                const float abs_r = sqrtf(dot3(rx, ry, rz, rx, ry, rz)); /* :35 */
                                const float diff_r = h - abs_r;                         /* :36 */
                                if ((epsilon < diff_r) && (epsilon < abs_r)) {          /* :37 */
                                    const float s = (c / h6) * ((diff_r * diff_r) * diff_r);
                                    gpx = (s * rx) / abs_r;
                                    gpy = (s * ry) / abs_r;
                                    gpz = (s * rz) / abs_r;
                                }
                            }
                            /* :93 f_press += ((p_i + p_j) / (2.0 * rho_j)) * grad_W_press */
                            const float cp = (p_i + p_j) / (2.0f * rho_j);
                            fpx += cp * gpx;
                            fpy += cp * gpy;
                            fpz += cp * gpz;
                            /* grad_W_vis :40-44 */
                            float gv = 0.0f;
                            {
                                const float abs_r = sqrtf(dot3(rx, ry, rz, rx, ry, rz)); /* :42 */
                                if (abs_r < h) gv = (c / h6) * (h - abs_r);             /* :43 */
                            }
                            /* :94 f_vis += ((vel_j - vel_i) / rho_j) * grad_W_vis */
                            const float *vj = vel4_r + 4 * jt;
                            fvx += ((vj[0] - vix) / rho_j) * gv;
                            fvy += ((vj[1] - viy) / rho_j) * gv;
                            fvz += ((vj[2] - viz) / rho_j) * gv;
                        }
                    }
                }

        if (rho_i > epsilon) {                                 /* :101 */
            const float sp = m * m / rho_i;                    /* :103 */
            fpx *= sp; fpy *= sp; fpz *= sp;
            const float sv = p->viscosity * m * m / rho_i;     /* :104 */
            fvx *= sv; fvy *= sv; fvz *= sv;
        }

        float fwx = 0.0f, fwy = 0.0f, fwz = 0.0f;              /* :107 */
        float rx = 0.0f, ry = 0.0f, rz = 0.0f;                 /* :108 */
        if (pix < h) rx += h - pix;                            /* :110-113 */
        else if (pix > 1.0f - h) rx += 1.0f - pix - h;
        if (piy < h) ry += h - piy;                            /* :115-118 */
        else if (piy > 1.0f - h) ry += 1.0f - piy - h;
        if (piz < h) rz += h - piz;                            /* :120-123 */
        else if (piz > 1.0f - h) rz += 1.0f - piz - h;

        const float max_r = fmaxf(fabsf(rx), fmaxf(fabsf(ry), fabsf(rz))); /* :125-126 */
        const float dx = (rx != 0.0f) ? p->damping : 0.0f;     /* :128-131 */
        const float dy = (ry != 0.0f) ? p->damping : 0.0f;
        const float dz = (rz != 0.0f) ? p->damping : 0.0f;
        if (max_r > 0.0f) {                                    /* :133 */
            const float dd = dot3(dx, dy, dz, vix, viy, viz);  /* :135 scalar, all 3 lanes */
            fwx = rx * p->stiffness - dd;
            fwy = ry * p->stiffness - dd;
            fwz = rz * p->stiffness - dd;
            fwx *= m; fwy *= m; fwz *= m;                      /* :136 */
        }

        /* :139 a_i = a_gravity + (float3(f_press + f_vis) + f_wall) / m */
        float ax = gx + ((fpx + fvx) + fwx) / m;
        float ay = gy + ((fpy + fvy) + fwy) / m;
        float az = gz + ((fpz + fvz) + fwz) / m;
        float vox, voy, voz, vow;
        if (ax != ax || ay != ay || az != az) {                /* :143 */
            ax = 0.0f; ay = 0.0f; az = 0.0f;                   /* :145 */
            vox = 0.0f; voy = 0.0f; voz = 0.0f; vow = 0.003f;  /* :146 */
        } else {
            vox = vix + ax * time_step;                        /* :150 */
            voy = viy + ay * time_step;
            voz = viz + az * time_step;
            vow = viw + 1.0f;
        }
        /* :153-154 pos_out = clamp(float4(pos_i,1) + dt*vel_out, 0, 1) */
        float *po = pos4_w + 4 * id, *vo = vel4_w + 4 * id;
        po[0] = hlsl_clamp01(pix + time_step * vox);
        po[1] = hlsl_clamp01(piy + time_step * voy);
        po[2] = hlsl_clamp01(piz + time_step * voz);
        po[3] = hlsl_clamp01(1.0f + time_step * vow);
        vo[0] = vox; vo[1] = voy; vo[2] = voz; vo[3] = vow;    /* :156-157 */
    }
}

/* ---- the frame: SphFluidSimulation.cs:96-102 ------------------------------------- */

typedef struct OracleSim {
    OracleParams p;
    float *pos[2], *vel[2];   /* ping-pong textures, [Read]=0 [Write]=1 after each Swap (:290-293) */
    float *rho;
    uint32_t *bucket;
} OracleSim;

OracleSim *oracle_sim_create(const OracleParams *p)
{
    OracleSim *s = (OracleSim *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->p = *p;
    const size_t cnt = (size_t)p->res * p->res;
    for (int i = 0; i < 2; i++) {
        s->pos[i] = (float *)calloc(cnt * 4, sizeof(float));
        s->vel[i] = (float *)calloc(cnt * 4, sizeof(float));
    }
    s->rho = (float *)calloc(cnt, sizeof(float));
    s->bucket = (uint32_t *)malloc(oracle_bucket_len(p) * sizeof(uint32_t));
    if (!s->pos[0] || !s->pos[1] || !s->vel[0] || !s->vel[1] || !s->rho || !s->bucket) {
        free(s->pos[0]); free(s->pos[1]); free(s->vel[0]); free(s->vel[1]);
        free(s->rho); free(s->bucket); free(s);
        return NULL;
    }
    return s;
}

void oracle_sim_destroy(OracleSim *s)
{
    if (!s) return;
    free(s->pos[0]); free(s->pos[1]); free(s->vel[0]); free(s->vel[1]);
    free(s->rho); free(s->bucket); free(s);
}

void oracle_sim_set_params(OracleSim *s, const OracleParams *p)
{
    /* only the inspector-tweakable scalars + thread count; sizes are fixed at create */
    s->p.viscosity = p->viscosity;
    s->p.gas_const = p->gas_const;
    s->p.rest_density = p->rest_density;
    s->p.stiffness = p->stiffness;
    s->p.damping = p->damping;
    s->p.num_threads = p->num_threads;
}

void oracle_sim_upload(OracleSim *s, const float *pos4, const float *vel4)
{
    const size_t cnt = (size_t)s->p.res * s->p.res;
    memcpy(s->pos[0], pos4, cnt * 4 * sizeof(float));
    memcpy(s->vel[0], vel4, cnt * 4 * sizeof(float));
}

void oracle_sim_download(const OracleSim *s, float *pos4, float *vel4, float *rho)
{
    const size_t cnt = (size_t)s->p.res * s->p.res;
    if (pos4) memcpy(pos4, s->pos[0], cnt * 4 * sizeof(float));
    if (vel4) memcpy(vel4, s->vel[0], cnt * 4 * sizeof(float));
    if (rho) memcpy(rho, s->rho, cnt * sizeof(float));
}

/* BucketGeneration :203-221 */
void oracle_sim_bucket_generation(OracleSim *s)
{
    oracle_clear_bucket(&s->p, s->bucket);
    oracle_bucket(&s->p, s->pos[0], s->bucket);
}

/* DensityCalculation :223-241 */
void oracle_sim_density_calculation(OracleSim *s)
{
    oracle_clear_float(&s->p, s->rho);
    oracle_density(&s->p, s->pos[0], s->bucket, s->rho);
}

/* UpdateVelocityAndPosition :243-269 */
void oracle_sim_update_velocity_and_position(OracleSim *s, float dt)
{
    oracle_velpos(&s->p, s->pos[0], s->vel[0], s->rho, s->bucket, s->pos[1], s->vel[1], dt);
    float *t;
    t = s->pos[0]; s->pos[0] = s->pos[1]; s->pos[1] = t;     /* Swap :267 */
    t = s->vel[0]; s->vel[0] = s->vel[1]; s->vel[1] = t;     /* Swap :268 */
}

/* Update :96-102 with Time.deltaTime = delta_time */
void oracle_sim_update(OracleSim *s, float delta_time)
{
    oracle_sim_bucket_generation(s);                          /* :98 */
    oracle_sim_density_calculation(s);                        /* :99 */
    for (uint32_t i = 0; i < s->p.substeps; i++)              /* :101 */
        oracle_sim_update_velocity_and_position(s, delta_time / s->p.dt_divisor); /* :102 */
}

const uint32_t *oracle_sim_bucket(const OracleSim *s) { return s->bucket; }

/* ======================================================================================
 * Row f1: initial-condition generator.
 *   Assets/Resources/InitParticles.compute:10-107   Preset1/2/3   -> oracle_init_particles()
 *   Assets/Resources/noiseSimplex.cginc:62-126,296-425  mod289, permute, taylorInvSqrt, grad4,
 *                                                        snoise(float4) -> oracle_snoise4()
 * Same floating-point contract as above.  The one transcendental, pow(count/fill, 1/3) at
 * InitParticles.compute:20,53,86, is evaluated by the CALLER (float powf) and passed in as
 * particle_per_dim, so oracle and engine cannot disagree on it.
 * The noise is the algorithm of webgl-noise (https://github.com/ashima/webgl-noise), Copyright (C) 2011 by Ashima Arts
 * (Simplex noise) and Stefan Gustavson (Classic noise), MIT License, in Lex-DRL's CgFx/HLSL port; the full upstream
 * notice is reproduced in sphfluidsimulation_amd/csrc/sph_kernels_init.hip.
 * ====================================================================================== */

typedef struct { float x, y, z, w; } f4;

static inline float fracf_(float x) { return x - floorf(x); }
static inline float step_(float y, float x) { return x >= y ? 1.0f : 0.0f; }          /* HLSL step(y, x) */
static inline float saturate_(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
static inline float dot4(f4 a, f4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
static inline float mod289f(float x) { return x - floorf(x * 0.00346020761245674740484429065744f) * 289.0f; } /* :65-67 */
static inline float permutef(float x) { return mod289f(x * x * 34.0f + x); }          /* :85-89 */
static inline float taylor_inv_sqrt(float r) { return 1.79284291400159f - 0.85373472095314f * r; } /* :105-107 */

static f4 grad4(float j, f4 ip)                                                       /* :115-130 */
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

float oracle_snoise4(float vx, float vy, float vz, float vw)                           /* :296-425 */
{
    const float Cx = 0.138196601125011f, Cy = 0.276393202250021f, Cz = 0.414589803375032f, Cw = -0.447213595499958f;
    const float F4 = 0.309016994374947451f;
    const f4 v = {vx, vy, vz, vw};
    const float sk = ((v.x * F4 + v.y * F4) + v.z * F4) + v.w * F4;                   /* dot(v, F4) :309-312 */
    f4 i = {floorf(v.x + sk), floorf(v.y + sk), floorf(v.z + sk), floorf(v.w + sk)};
    const float un = ((i.x * Cx + i.y * Cx) + i.z * Cx) + i.w * Cx;                   /* dot(i, C.xxxx) :315 */
    const f4 x0 = {v.x - i.x + un, v.y - i.y + un, v.z - i.z + un, v.w - i.w + un};
    /* rank sorting :320-329 */
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
    const f4 x1 = {x0.x - i1.x + Cx, x0.y - i1.y + Cx, x0.z - i1.z + Cx, x0.w - i1.w + Cx};   /* :341-344 */
    const f4 x2 = {x0.x - i2.x + Cy, x0.y - i2.y + Cy, x0.z - i2.z + Cy, x0.w - i2.w + Cy};
    const f4 x3 = {x0.x - i3.x + Cz, x0.y - i3.y + Cz, x0.z - i3.z + Cz, x0.w - i3.w + Cz};
    const f4 x4 = {x0.x + Cw, x0.y + Cw, x0.z + Cw, x0.w + Cw};
    /* permutations :347-364 */
    i.x = mod289f(i.x); i.y = mod289f(i.y); i.z = mod289f(i.z); i.w = mod289f(i.w);
    const float j0 = permutef(permutef(permutef(permutef(i.w) + i.z) + i.y) + i.x);
    float j1[4];
    const float aw[4] = {i1.w, i2.w, i3.w, 1.0f}, az[4] = {i1.z, i2.z, i3.z, 1.0f};
    const float ay[4] = {i1.y, i2.y, i3.y, 1.0f}, ax[4] = {i1.x, i2.x, i3.x, 1.0f};
    for (int k = 0; k < 4; k++)
        j1[k] = permutef(permutef(permutef(permutef(i.w + aw[k]) + i.z + az[k]) + i.y + ay[k]) + i.x + ax[k]);
    const f4 ip = {0.003401360544217687075f, 0.020408163265306122449f, 0.142857142857142857143f, 0.0f}; /* :368-373 */
    f4 p0 = grad4(j0, ip), p1 = grad4(j1[0], ip), p2 = grad4(j1[1], ip), p3 = grad4(j1[2], ip), p4 = grad4(j1[3], ip);
    const float n0 = taylor_inv_sqrt(dot4(p0, p0)), n1 = taylor_inv_sqrt(dot4(p1, p1));     /* :382-393 */
    const float n2 = taylor_inv_sqrt(dot4(p2, p2)), n3 = taylor_inv_sqrt(dot4(p3, p3));
    p0.x *= n0; p0.y *= n0; p0.z *= n0; p0.w *= n0;
    p1.x *= n1; p1.y *= n1; p1.z *= n1; p1.w *= n1;
    p2.x *= n2; p2.y *= n2; p2.z *= n2; p2.w *= n2;
    p3.x *= n3; p3.y *= n3; p3.z *= n3; p3.w *= n3;
    const float n4 = taylor_inv_sqrt(dot4(p4, p4));
    p4.x *= n4; p4.y *= n4; p4.z *= n4; p4.w *= n4;
    float m00 = fmaxf(0.6f - dot4(x0, x0), 0.0f), m01 = fmaxf(0.6f - dot4(x1, x1), 0.0f);   /* :396-412 */
    float m02 = fmaxf(0.6f - dot4(x2, x2), 0.0f);
    float m10 = fmaxf(0.6f - dot4(x3, x3), 0.0f), m11 = fmaxf(0.6f - dot4(x4, x4), 0.0f);
    m00 = m00 * m00; m01 = m01 * m01; m02 = m02 * m02;
    m10 = m10 * m10; m11 = m11 * m11;
    const float d0 = dot4(p0, x0), d1 = dot4(p1, x1), d2 = dot4(p2, x2), d3 = dot4(p3, x3), d4 = dot4(p4, x4);
    const float a = ((m00 * m00) * d0 + (m01 * m01) * d1) + (m02 * m02) * d2;               /* dot(float3) */
    const float b = (m10 * m10) * d3 + (m11 * m11) * d4;                                    /* dot(float2) */
    return 49.0f * (a + b);                                                                 /* :414-424 */
}

/* InitParticles.compute: Preset1 :10-41 (preset 0), Preset2 :43-74 (1), Preset3 :76-107 (2).
 * noise_on = 0 skips :37/:70/:103 (test aid). */
int oracle_init_particles(int preset, uint32_t res, float fill, uint32_t particle_per_dim, int noise_on, float *pos4)
{
    if (preset < 0 || preset > 2 || particle_per_dim == 0 || !pos4) return -1;
    const uint32_t ppd = particle_per_dim;
    const uint32_t xSize = (uint32_t)ceilf((float)ppd * fill);                          /* :22,:55,:88 */
    const uint32_t ySize = ppd;
    const uint32_t zSize = preset == 0 ? ppd : (uint32_t)ceilf((float)ppd * fill);      /* :24,:57,:90 */
    const float cs = 1.0f / (float)ppd;                                                 /* :26 */
    for (uint32_t y = 0; y < res; y++)
        for (uint32_t x = 0; x < res; x++) {
            const int32_t i = (int32_t)(x + y * res);                                   /* :28 */
            const uint32_t ui = (uint32_t)i;
            f4 pos;
            const float qx = (float)(ui / (zSize * ySize)), qy = (float)((ui / zSize) % ySize), qz = (float)(ui % zSize);
            if (preset == 0) {                                                          /* :30-35 */
                pos.x = cs / 2.0f + qx * fill / (float)xSize;
                pos.y = cs / 2.0f + qy * 0.9f / (float)ySize;
                pos.z = cs / 2.0f + qz * 1.0f / (float)zSize;
            } else if (preset == 1) {                                                   /* :63-68 */
                pos.x = cs / 2.0f + fill / 2.0f + qx * fill / (float)xSize;
                pos.y = cs / 2.0f + qy * 0.9f / (float)ySize;
                pos.z = cs / 2.0f + fill / 2.0f + qz * fill / (float)zSize;
            } else {                                                                    /* :96-101 */
                pos.x = cs / 2.0f + qx * fill / (float)xSize;
                pos.y = cs / 2.0f + qy * 0.9f / (float)ySize;
                pos.z = cs / 2.0f + qz * fill / (float)zSize;
            }
            pos.w = 1.0f;
            if (noise_on) {                                                             /* :37 pos += snoise(pos + i) * cs */
                const float fi = (float)i;
                const float nz = oracle_snoise4(pos.x + fi, pos.y + fi, pos.z + fi, pos.w + fi) * cs;
                pos.x += nz; pos.y += nz; pos.z += nz;
            }
            float *o = pos4 + 4ull * (size_t)ui;
            o[0] = pos.x; o[1] = pos.y; o[2] = pos.z; o[3] = 1.0f;                      /* :38 */
        }
    return 0;
}

/* InitParticles.compute:20 / :53,:86: uint(ceil(pow(particleCount / fill[^2], 1.0/3.0))) in fp32 (libm powf). */
uint32_t oracle_particle_per_dim(int preset, uint32_t res, float fill)
{
    const float count = (float)(res * res);
    const float denom = preset == 0 ? fill : fill * fill;
    return (uint32_t)ceilf(powf(count / denom, 1.0f / 3.0f));
}

/* ============================== row f3: UpdateMeshProperties.compute =====================================
 * The render feed that follows the path (SphFluidSimulation.cs:104, :271-284): per texel, a translate x scale
 * 4x4 and a blue->red speed colour (UpdateMeshProperties.compute:20-64).  Record = MeshProperties
 * (SphFluidSimulation.cs:16-29): float4x4 mat in column-major memory order (HLSL's default packing, and what
 * UnityEngine.Matrix4x4 holds: m00,m10,m20,m30, m01,...), then float4 color: 20 floats = 80 bytes.
 * sim_trs = transform.localToWorldMatrix in that same column-major order (:281).
 * mul(translation, scale) is evaluated as the literal 4-term sums, so a non-finite position poisons its row
 * exactly as IEEE says (0 * inf = NaN); the products with 0 and 1 are otherwise exact. */

void oracle_update_mesh_properties(uint32_t count, const float *pos4, const float *vel4, float low_speed,
                                   float high_speed, const float *particle_scale3, const float *sim_trs16,
                                   float *props20)
{
    /* :35 simScale = (_m00,_m11,_m22); :38 simPosition is read but never used */
    const float ss[3] = {sim_trs16[0], sim_trs16[5], sim_trs16[10]};
    const float S[4][4] = {{particle_scale3[0], 0, 0, 0}, {0, particle_scale3[1], 0, 0},
                           {0, 0, particle_scale3[2], 0}, {0, 0, 0, 1}};                       /* :44-49 */
#pragma omp parallel for schedule(static)
    for (int64_t id = 0; id < (int64_t)count; id++) {
        const float *p = pos4 + 4 * id, *v = vel4 + 4 * id;
        const float speed = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);                  /* :32 */
        float w[3];
        for (int k = 0; k < 3; k++) w[k] = p[k] * ss[k] - (ss[k] * 0.5f);                      /* :41 */
        const float T[4][4] = {{1, 0, 0, w[0]}, {0, 1, 0, w[1]}, {0, 0, 1, w[2]}, {0, 0, 0, 1}}; /* :52-57 */
        float *o = props20 + 20 * id;
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++)                                                         /* :60 */
                o[c * 4 + r] = ((T[r][0] * S[0][c] + T[r][1] * S[1][c]) + T[r][2] * S[2][c]) + T[r][3] * S[3][c];
        const float t = hlsl_clamp01((speed - low_speed) / (high_speed - low_speed));          /* :63 */
        const float a[4] = {0, 0, 1, 1}, b[4] = {1, 0, 0, 1};
        for (int k = 0; k < 4; k++) o[16 + k] = a[k] + t * (b[k] - a[k]);                       /* :64 lerp(x,y,s)=x+s(y-x) */
    }
}
