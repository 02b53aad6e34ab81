// FETCH_SIZE calibration for the compositing kernel's record gathers: N wave-instructions that gather 64-byte
// records at pseudo-random indices from an array of R records, each record as four 16-byte loads (the kernel's global_load_lds
// pattern, here into registers), against the same number of bytes streamed. The byte counts are known; run under
//   rocprofv3 --pmc FETCH_SIZE -- tools/micro/gather64        (and --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum, TCC_HIT_sum TCC_MISS_sum)
// and compare each kernel's FETCH_SIZE with the "bytes" line it prints.   build: hipcc --offload-arch=gfx950 -O3 gather64.hip -o gather64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MODE 0: random 64-byte records (4 x 16 B per lane)   1: streaming, 16 B per lane, the same total bytes   2: random 128-byte aligned pairs of records
// 3: random records, but only the first 16 bytes of each (one 16-byte load per record)
// 4: random records, four ADJACENT lanes share a record (lane & 3 = its 16-byte chunk): one load instruction covers 16 records per wave
// 5: as 0 with only the lower 32 lanes of every wave active (the compositing kernel's staging: lanes e < 32 gather 4 chunks each)
template <int MODE>
__global__ void __launch_bounds__(256) gather_kernel(const float4* __restrict__ rec, unsigned R, unsigned iters, float* __restrict__ sink)
{
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    float acc = 0.0f;
    for (unsigned it = 0; it < iters; it++) {
        if (MODE == 1) {
            const size_t i = ((size_t)it * nthreads + gid) * 4 % ((size_t)R * 4);
#pragma unroll
            for (int c = 0; c < 4; c++) { const float4 v = rec[i + c]; acc += v.x + v.w; }
        } else {
            if (MODE == 5 && (threadIdx.x & 32u)) continue;
            unsigned id = hash32((MODE == 4 ? gid >> 2 : gid) * 2654435761u + it * 40503u + 12345u) % R;
            if (MODE == 2) id &= ~1u;
            const float4* p = rec + (size_t)id * 4;
            if (MODE == 4) {
#pragma unroll
                for (int c = 0; c < 4; c++) {      // four instructions, 16 records each: the same 64 records per wave and iteration as mode 0
                    const unsigned id4 = hash32(((gid & ~63u) + 16u * c + ((gid & 63u) >> 2)) * 2654435761u + it * 40503u + 12345u) % R;
                    const float4 v = rec[(size_t)id4 * 4 + (gid & 3u)]; acc += v.x + v.w;
                }
                continue;
            }
            const int nc = MODE == 3 ? 1 : MODE == 2 ? 8 : 4;
#pragma unroll
            for (int c = 0; c < nc; c++) { const float4 v = p[c]; acc += v.x + v.w; }
        }
    }
    if (acc == 1.2345f) sink[gid] = acc;
}

int main(int argc, char** argv)
{
    const unsigned R = argc > 1 ? (unsigned)atoi(argv[1]) : 589824u;          // 37.7 MB: one view's records of the real merged set
    const unsigned blocks = 256 * 32, threads = 256, iters = 64;
    float4* rec; float* sink;
    CHECK(hipMalloc(&rec, (size_t)R * 64 + 128));
    CHECK(hipMalloc(&sink, (size_t)blocks * threads * 4));
    CHECK(hipMemset(rec, 0, (size_t)R * 64 + 128));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const double n = (double)blocks * threads * iters;
    const char* names[6] = {"random 64-byte records, 4 x 16 B", "streaming 16 B per lane x 4", "random 128-byte aligned pairs, 8 x 16 B", "random records, first 16 B only", "random records, 4 adjacent lanes per record", "random 64-byte records, lanes 0..31 only"};
    const double bytes[6] = {n * 64, n * 64, n * 128, n * 16, n * 64, n * 32};
    for (int rep = 0; rep < 2; rep++)
        for (int m = 0; m < 6; m++) {
            CHECK(hipEventRecord(a));
            if (m == 0) gather_kernel<0><<<blocks, threads>>>(rec, R, iters, sink);
            if (m == 1) gather_kernel<1><<<blocks, threads>>>(rec, R, iters, sink);
            if (m == 2) gather_kernel<2><<<blocks, threads>>>(rec, R, iters, sink);
            if (m == 3) gather_kernel<3><<<blocks, threads>>>(rec, R, iters, sink);
            if (m == 4) gather_kernel<4><<<blocks, threads>>>(rec, R, iters, sink);
            if (m == 5) gather_kernel<5><<<blocks, threads>>>(rec, R, iters, sink);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep) printf("gather_kernel<%d> (%s): records %u (%.1f MB), requested bytes %.4e, %.3f ms, %.1f GB/s requested\n", m, names[m], R, R * 64e-6, bytes[m], ms, bytes[m] / ms * 1e-6);
        }
    return 0;
}
