// Microbenchmark: shader clocks per VALU instruction of ONE wave on a SIMD -- a chain of dependent FMAs, 2 / 4 / 8 independent chains,
// and the same with 2 / 4 waves resident per SIMD (blocks of 64 threads, grid = SIMDs x waves).  hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int CH>
__global__ void __launch_bounds__(64) chains(float* out, unsigned long long* clk, int iters, float a, float b)
{
    float x[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) x[c] = threadIdx.x * 0.001f + c;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64 / CH; k++)
#pragma unroll
            for (int c = 0; c < CH; c++) x[c] = __builtin_fmaf(x[c], a, b);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) s += x[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(64) trans(float* out, unsigned long long* clk, int iters, float a)
{
    float x = threadIdx.x * 0.001f + 1.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) x = __builtin_amdgcn_rcpf(x) + a;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = x;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(64) ldschain(float* out, unsigned long long* clk, int iters)
{
    __shared__ int idx[64];
    idx[threadIdx.x] = (threadIdx.x * 17 + 5) & 63;
    __syncthreads();
    int j = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) j = idx[j];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = (float)j;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main()
{
    const int SIMDS = 1024, iters = 2000;
    float* out; unsigned long long* clk;
    hipMalloc(&out, sizeof(float) * 64 * SIMDS * 8);
    hipMalloc(&clk, sizeof(unsigned long long) * SIMDS * 8);
    std::vector<unsigned long long> h(SIMDS * 8);
    auto report = [&](const char* name, int blocks, double instr) {
        hipDeviceSynchronize();
        hipMemcpy(h.data(), clk, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        double s = 0; unsigned long long mx = 0;
        for (int i = 0; i < blocks; i++) { s += (double)h[i]; if (h[i] > mx) mx = h[i]; }
        printf("%-44s %5d waves: %.2f clocks per instruction (mean wave), %.2f (slowest)\n", name, blocks, s / blocks / instr, (double)mx / instr);
    };
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = SIMDS * wps;
        hipLaunchKernelGGL(chains<1>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 0.999f, 0.001f); report("v_fma chain, 1 dependent chain", blocks, 64.0 * iters);
        hipLaunchKernelGGL(chains<2>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 0.999f, 0.001f); report("v_fma, 2 independent chains", blocks, 64.0 * iters);
        hipLaunchKernelGGL(chains<4>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 0.999f, 0.001f); report("v_fma, 4 independent chains", blocks, 64.0 * iters);
        hipLaunchKernelGGL(chains<8>, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 0.999f, 0.001f); report("v_fma, 8 independent chains", blocks, 64.0 * iters);
        hipLaunchKernelGGL(trans, dim3(blocks), dim3(64), 0, 0, out, clk, iters, 0.5f); report("v_rcp + v_add chain (2 instructions)", blocks, 128.0 * iters);
        hipLaunchKernelGGL(ldschain, dim3(blocks), dim3(64), 0, 0, out, clk, iters); report("dependent ds_read_b32 chain (+ address shift)", blocks, 16.0 * iters);
    }
    return 0;
}
