// f3dg_groupnorm.hip -- fused GroupNorm (+ optional SiLU) for the SongUNet backbone of the predictor.
//
// The backbone calls GroupNorm 78 times per pass, each followed by SiLU in the residual blocks
// (reference src/gaussian_predictor.py:250-262 GroupNorm, :318-323 `silu(norm(x))`). On PyTorch-ROCm that is three
// bandwidth-bound kernels per call (row moments, scale/shift, silu: five passes over the tensor); here it is ONE kernel:
// a workgroup owns one (sample, group) slab of Cg x H x W contiguous floats, accumulates sum and sum of squares in
// float64, and re-reads the slab (<= 1 MiB: L2-resident) to write silu(weight * (x - mean) * rstd + bias). Two HBM
// passes (one read, one write) instead of five.
#include "f3dg_common.h"

namespace {

constexpr int GN_THREADS = 1024;

__global__ void __launch_bounds__(GN_THREADS)
group_norm_silu_kernel(int C, int HW, int groups, const float* __restrict__ x, const float* __restrict__ weight,
                       const float* __restrict__ bias, float eps, int apply_silu, float* __restrict__ y)
{
    const int g = blockIdx.x % groups;
    const int n = blockIdx.x / groups;
    const int Cg = C / groups;
    const size_t slab = (size_t)Cg * HW;
    const float* xs = x + ((size_t)n * C + (size_t)g * Cg) * HW;
    float* ys = y + ((size_t)n * C + (size_t)g * Cg) * HW;

    // ---- moments: float partial sums per thread over short runs, folded into float64
    double s1 = 0.0, s2 = 0.0;
    const size_t n4 = slab / 4;                                   // HW is a multiple of 4 for every layer of the backbone
    const float4* x4 = reinterpret_cast<const float4*>(xs);
    for (size_t i = threadIdx.x; i < n4; i += GN_THREADS) {
        const float4 v = x4[i];
        s1 += (double)((v.x + v.y) + (v.z + v.w));
        s2 += (double)((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
    }
    for (size_t i = n4 * 4 + threadIdx.x; i < slab; i += GN_THREADS) {
        const float v = xs[i];
        s1 += v; s2 += (double)(v * v);
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        s1 += __shfl_xor(s1, m, 64);
        s2 += __shfl_xor(s2, m, 64);
    }
    __shared__ double w1[GN_THREADS / 64], w2[GN_THREADS / 64];
    __shared__ float s_mean, s_rstd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { w1[wave] = s1; w2[wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < GN_THREADS / 64; w++) { a += w1[w]; b += w2[w]; }
        const double mean = a / (double)slab;
        double var = b / (double)slab - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean = (float)mean;
        s_rstd = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const float mean = s_mean, rstd = s_rstd;

    // ---- normalise, affine, (silu): the slab is re-read from L2
    const int hw4 = HW / 4;
    if (HW % 4 == 0) {
        float4* y4 = reinterpret_cast<float4*>(ys);
        for (size_t i = threadIdx.x; i < n4; i += GN_THREADS) {
            const int c = g * Cg + (int)(i / hw4);
            const float sc = weight[c] * rstd;
            const float sh = bias[c] - mean * sc;
            float4 v = x4[i];
            v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
            if (apply_silu) {
                v.x = v.x / (1.0f + expf(-v.x)); v.y = v.y / (1.0f + expf(-v.y));
                v.z = v.z / (1.0f + expf(-v.z)); v.w = v.w / (1.0f + expf(-v.w));
            }
            y4[i] = v;
        }
    } else {
        for (size_t i = threadIdx.x; i < slab; i += GN_THREADS) {
            const int c = g * Cg + (int)(i / HW);
            const float sc = weight[c] * rstd;
            float v = xs[i] * sc + (bias[c] - mean * sc);
            if (apply_silu) v = v / (1.0f + expf(-v));
            ys[i] = v;
        }
    }
}

} // namespace

extern "C" int f3dg_group_norm_silu(void* stream, int N, int C, int HW, int groups, const float* x, const float* weight,
                                    const float* bias, float eps, int apply_silu, float* y)
{
    if (N < 0 || C <= 0 || HW <= 0 || groups <= 0 || C % groups != 0 || !x || !weight || !bias || !y) return F3DG_ERR_BAD_ARG;
    if (N == 0) return F3DG_OK;
    if (((uintptr_t)x | (uintptr_t)y) & 15u) return F3DG_ERR_BAD_ARG;       // 16-byte loads / stores
    hipLaunchKernelGGL(group_norm_silu_kernel, dim3((unsigned)(N * groups)), dim3(GN_THREADS), 0, (hipStream_t)stream, C, HW,
                       groups, x, weight, bias, eps, apply_silu, y);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}
