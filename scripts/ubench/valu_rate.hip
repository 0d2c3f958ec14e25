// Microbenchmark: VALU issue rate on gfx950 vs waves per SIMD and dependency structure.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int CHAINS>
__global__ void k_fma(float *out, int iters, float a, float b)
{
    float x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = (float)threadIdx.x + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) x[c] = __builtin_fmaf(x[c], a, b);
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
void run(int waves_per_simd, int ncu)
{
    float *out; hipMalloc(&out, 64 << 20);
    const int iters = 4096;
    const int threads = 64 * waves_per_simd * 4;       // one block per CU, waves_per_simd waves on each of 4 SIMDs
    dim3 grid(ncu), block(threads > 1024 ? 1024 : threads);
    int blocks_per_cu = threads > 1024 ? threads / 1024 : 1;
    grid.x = ncu * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_fma<CHAINS>, grid, block, 0, 0, out, 16, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_fma<CHAINS>, grid, block, 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_wave = (double)iters * 16 * CHAINS;
    double cycles = ms * 1e-3 * 2.4e9;
    printf("chains %d waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instr per wave, %.2f cycles per instr per SIMD\n", CHAINS,
           waves_per_simd, ms, cycles / instr_per_wave, cycles / (instr_per_wave * waves_per_simd));
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 3, 4, 8}) run<1>(w, 256);
    for (int w : {1, 2, 3, 4, 8}) run<4>(w, 256);
    for (int w : {1, 2, 3, 4}) run<8>(w, 256);
    return 0;
}
