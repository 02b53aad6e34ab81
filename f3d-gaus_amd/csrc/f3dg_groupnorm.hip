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

// bf16 <-> f32 (round to nearest even; NaN kept quiet)
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f)
{
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// 16-byte packets: 4 floats or 8 bf16
template <typename T> struct Packet;
template <> struct Packet<float> {
    static constexpr int N = 4;
    float v[4];
    __device__ __forceinline__ void load(const float* p) { const float4 q = *reinterpret_cast<const float4*>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Packet<unsigned short> {
    static constexpr int N = 8;
    float v[8];
    __device__ __forceinline__ void load(const unsigned short* p)
    {
        const uint4 q = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int i = 0; i < 4; i++) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u); }
    }
    __device__ __forceinline__ void store(unsigned short* p) const
    {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; i++) w[i] = (unsigned)f32_to_bf16(v[2 * i]) | ((unsigned)f32_to_bf16(v[2 * i + 1]) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(unsigned short x) { return bf16_to_f32(x); }
__device__ __forceinline__ void from_f32(float& d, float x) { d = x; }
__device__ __forceinline__ void from_f32(unsigned short& d, float x) { d = f32_to_bf16(x); }

// T = float, or unsigned short holding bfloat16 (the bf16 option of the backbone: statistics and arithmetic stay float32 / float64)
template <typename T>
__global__ void __launch_bounds__(GN_THREADS)
group_norm_silu_kernel(int C, int HW, int groups, const T* __restrict__ x, const float* __restrict__ weight,
                       const float* __restrict__ bias, float eps, int apply_silu, T* __restrict__ y)
{
    constexpr int PN = Packet<T>::N;
    const int g = blockIdx.x % groups;
    const int n = blockIdx.x / groups;
    const int Cg = C / groups;
    const size_t slab = (size_t)Cg * HW;
    const T* xs = x + ((size_t)n * C + (size_t)g * Cg) * HW;
    T* ys = y + ((size_t)n * C + (size_t)g * Cg) * HW;

    // ---- moments: float partial sums per thread over short runs, folded into float64
    double s1 = 0.0, s2 = 0.0;
    const bool vec = HW % PN == 0;                                // true for every layer of the backbone at 256^2 / 32^2
    const size_t np = vec ? slab / PN : 0;
    for (size_t i = threadIdx.x; i < np; i += GN_THREADS) {
        Packet<T> p;
        p.load(xs + i * PN);
        float a = 0.0f, b = 0.0f;
#pragma unroll
        for (int k = 0; k < PN; k++) { a += p.v[k]; b += p.v[k] * p.v[k]; }
        s1 += (double)a;
        s2 += (double)b;
    }
    for (size_t i = np * PN + threadIdx.x; i < slab; i += GN_THREADS) {
        const float v = to_f32(xs[i]);
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
    if (vec) {
        const int hwp = HW / PN;
        for (size_t i = threadIdx.x; i < np; i += GN_THREADS) {
            const int c = g * Cg + (int)(i / hwp);
            const float sc = weight[c] * rstd;
            const float sh = bias[c] - mean * sc;
            Packet<T> p;
            p.load(xs + i * PN);
#pragma unroll
            for (int k = 0; k < PN; k++) {
                float v = p.v[k] * sc + sh;
                if (apply_silu) v = v / (1.0f + expf(-v));
                p.v[k] = v;
            }
            p.store(ys + i * PN);
        }
    } else {
        for (size_t i = threadIdx.x; i < slab; i += GN_THREADS) {
            const int c = g * Cg + (int)(i / HW);
            const float sc = weight[c] * rstd;
            float v = to_f32(xs[i]) * sc + (bias[c] - mean * sc);
            if (apply_silu) v = v / (1.0f + expf(-v));
            from_f32(ys[i], v);
        }
    }
}

template <typename T>
int launch_gn(void* stream, int N, int C, int HW, int groups, const T* x, const float* weight, const float* bias, float eps,
              int apply_silu, T* y)
{
    if (N < 0 || C <= 0 || HW <= 0 || groups <= 0 || C % groups != 0 || !x || !weight || !bias || !y) return F3DG_ERR_BAD_ARG;
    if (N == 0) return F3DG_OK;
    if (((uintptr_t)x | (uintptr_t)y) & 15u) return F3DG_ERR_BAD_ARG;       // 16-byte loads / stores
    F3DG_KLAUNCH(group_norm_silu_kernel<T>, dim3((unsigned)(N * groups)), dim3(GN_THREADS), 0, (hipStream_t)stream, C, HW,
                       groups, x, weight, bias, eps, apply_silu, y);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

} // namespace

extern "C" int f3dg_group_norm_silu(void* stream, int N, int C, int HW, int groups, const float* x, const float* weight,
                                    const float* bias, float eps, int apply_silu, float* y)
{
    return launch_gn<float>(stream, N, C, HW, groups, x, weight, bias, eps, apply_silu, y);
}

extern "C" int f3dg_group_norm_silu_bf16(void* stream, int N, int C, int HW, int groups, const uint16_t* x, const float* weight,
                                         const float* bias, float eps, int apply_silu, uint16_t* y)
{
    return launch_gn<unsigned short>(stream, N, C, HW, groups, x, weight, bias, eps, apply_silu, y);
}
