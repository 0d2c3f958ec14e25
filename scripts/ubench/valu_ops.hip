// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU operation classes k_velpos_list is made of.
// 8 independent chains per lane, 4 waves per SIMD: the dependent-issue latency is hidden, what is left is the pipe's rate.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_ops scripts/ubench/valu_ops.hip && /tmp/valu_ops
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

enum Op { FMA, MUL, ADD, CNDMASK, CMP_CNDMASK, MIN_U32, LSHL_ADD, AND_OR, DIV_FIXUP, RCP, SQRT, MAX_F32, CVT, MOV_DPP, FMA_PK, NUM_OPS };
static const char *kNames[NUM_OPS] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_cndmask_b32 (vcc fixed)", "v_cmp + v_cndmask (2 instrs)", "v_min_u32",
                                      "v_lshl_add_u32", "v_and_or_b32", "v_div_fixup_f32", "v_rcp_f32", "v_sqrt_f32", "v_max_f32", "v_cvt_f32_u32",
                                      "v_mov_b32 dpp row_shr:1", "v_pk_fma_f32 (2 results)"};

template <int OP>
__device__ __forceinline__ void step(float &x, float a, float b)
{
    uint32_t u = __float_as_uint(x), ua = __float_as_uint(a), ub = __float_as_uint(b);
    if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    if (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
    if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc");
    if (OP == CMP_CNDMASK) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(a), "v"(b) : "vcc");
    if (OP == MIN_U32) { asm volatile("v_min_u32 %0, %0, %1" : "+v"(u) : "v"(ua)); x = __uint_as_float(u); }
    if (OP == LSHL_ADD) { asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u) : "v"(ua)); x = __uint_as_float(u); }
    if (OP == AND_OR) { asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u) : "v"(ua), "v"(ub)); x = __uint_as_float(u); }
    if (OP == DIV_FIXUP) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
    if (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
    if (OP == MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    if (OP == CVT) { asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(u)); x = __uint_as_float(u); }
    if (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
}

template <int OP>
__global__ void __launch_bounds__(1024) k_ops(float *out, int iters, float a, float b)
{
    float x[8];
#pragma unroll
    for (int c = 0; c < 8; c++) x[c] = (float)threadIdx.x + c;
    if (OP == FMA_PK) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 y[8], a2 = {a, a}, b2 = {b, b};
#pragma unroll
        for (int c = 0; c < 8; c++) y[c] = f2{x[c], x[c] + 1.0f};
        for (int i = 0; i < iters; i++)
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int c = 0; c < 8; c++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[c]) : "v"(a2), "v"(b2));
#pragma unroll
        for (int c = 0; c < 8; c++) x[c] = y[c].x + y[c].y;
    } else {
        for (int i = 0; i < iters; i++)
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int c = 0; c < 8; c++) step<OP>(x[c], a, b);
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(float *out, int waves_per_simd)
{
    const int iters = 2048;
    int threads = 64 * 4 * waves_per_simd;            // one block per CU (two of 1024 threads for 8 waves per SIMD)
    int blocks = 256;
    if (threads > 1024) {
        blocks *= 2;
        threads /= 2;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_ops<OP>, dim3(blocks), dim3(threads), 0, 0, out, 16, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_ops<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instrs = (double)iters * 64 * (OP == CMP_CNDMASK ? 2 : 1);
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-32s waves/SIMD %d: %7.3f ms  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", kNames[OP], waves_per_simd, ms,
           cycles / (instrs * waves_per_simd));
}

template <int OP>
void sweep(float *out)
{
    run<OP>(out, 1);
    run<OP>(out, 4);
    if (OP == FMA || OP == MAX_F32 || OP == FMA_PK)
        for (int w : {2, 3, 5, 6, 8}) run<OP>(out, w);
    if (OP + 1 < NUM_OPS) sweep<(OP + 1 < NUM_OPS ? OP + 1 : OP)>(out);
}

int main()
{
    float *out;
    if (hipMalloc(&out, 64 << 20) != hipSuccess) return 1;
    sweep<0>(out);
    hipFree(out);
    return 0;
}
