// Microbenchmark: does a wave64 VALU instruction whose EXEC mask has an all-zero 32-lane half issue in one cycle instead of two
// on gfx950 (SIMD-32)? Times a dependent-free FMA stream with (a) all lanes, (b) the lower 32 lanes, (c) every other lane.
//   hipcc --offload-arch=gfx950 -O3 -o half_exec tools/ubench/half_exec.hip && ./half_exec
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, int iters)
{
    const unsigned lane = threadIdx.x;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    const bool on = MODE == 0 ? true : MODE == 1 ? lane < 32 : MODE == 2 ? (lane & 1) == 0 : lane < 16;
    if (on) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a0 = fmaf(a0, 1.0001f, 0.5f); a1 = fmaf(a1, 1.0001f, 0.5f); a2 = fmaf(a2, 1.0001f, 0.5f); a3 = fmaf(a3, 1.0001f, 0.5f);
                a4 = fmaf(a4, 1.0001f, 0.5f); a5 = fmaf(a5, 1.0001f, 0.5f); a6 = fmaf(a6, 1.0001f, 0.5f); a7 = fmaf(a7, 1.0001f, 0.5f);
            }
        }
    }
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
float run(float* d, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 32;       // 8 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float* d;
    hipMalloc(&d, 256 * 32 * 64 * sizeof(float));
    const int iters = 4000;
    const double insts = 256.0 * 32 * iters * 64;      // wave-instructions
    const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters);
    printf("all 64 lanes      %.3f ms  (%.2f cycles per wave-instruction per SIMD at 2.4 GHz)\n", t0, t0 * 1e-3 * 2.4e9 * 1024 / insts);
    printf("lower 32 lanes    %.3f ms  (%.2f)\n", t1, t1 * 1e-3 * 2.4e9 * 1024 / insts);
    printf("every other lane  %.3f ms  (%.2f)\n", t2, t2 * 1e-3 * 2.4e9 * 1024 / insts);
    printf("lower 16 lanes    %.3f ms  (%.2f)\n", t3, t3 * 1e-3 * 2.4e9 * 1024 / insts);
    return 0;
}
