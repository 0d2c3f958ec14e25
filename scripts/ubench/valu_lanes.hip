// Microbenchmark: does a wave64 VALU instruction cost less when only some of its lanes are active (EXEC)?
// 8 independent fma chains per lane, 4 waves per SIMD, the loop body executed under `if (lane < K)`.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_lanes scripts/ubench/valu_lanes.hip && /tmp/valu_lanes
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int OP>
__global__ void __launch_bounds__(1024) k_lanes(float *out, int iters, float a, float b, int active, int stride)
{
    float x[8];
#pragma unroll
    for (int c = 0; c < 8; c++) x[c] = (float)threadIdx.x + c;
    const int lane = threadIdx.x & 63;
    if ((lane % stride) == 0 && lane / stride < active) {          // EXEC mask fixed for the whole loop
        for (int i = 0; i < iters; i++)
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                    if (OP == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(a) : "vcc");
                    if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[c]));
                }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static void run(const char *name, float *out)
{
    const int iters = 2000, blocks = 256 * 2;                       // 2 blocks of 1024 threads per CU = 8 waves per SIMD... (4 resident per SIMD at a time x 2)
    const int cfg[][2] = {{64, 1}, {48, 1}, {32, 1}, {16, 1}, {8, 1}, {1, 1}, {16, 4}, {32, 2}};   // (active lanes, stride between them)
    for (auto &c : cfg) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_lanes<OP>, dim3(blocks), dim3(1024), 0, 0, out, 10, 1.0001f, 0.5f, c[0], c[1]);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_lanes<OP>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1.0001f, 0.5f, c[0], c[1]);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_simd = (double)blocks * 16 /*waves per block*/ * iters * 64 / (256.0 * 4);
        printf("%-14s active %2d stride %d: %8.3f ms  -> %.2f cycles per wave instruction and SIMD at 2.4 GHz\n", name, c[0], c[1], ms,
               ms * 1e-3 * 2.4e9 / insts_per_simd);
    }
}

int main()
{
    float *out;
    hipMalloc(&out, 512 * 1024 * sizeof(float));
    run<0>("v_fma_f32", out);
    run<1>("v_cndmask_b32", out);
    run<2>("v_rcp_f32", out);
    return 0;
}
