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

// ---- channels-last (NHWC) variant ---------------------------------------------------------------------------------------------------
// MIOpen's fastest convolutions on gfx950 are its NHWC kernels; handed NCHW tensors it wraps them in batched_transpose launches
// (15 % of a bf16 pass of the backbone, profiles/r04_final/unet.md), and torch's GroupNorm converts a channels_last tensor back, so the
// layout only pays if GroupNorm keeps it. x is [N][HW][C]: a group's Cg channels are 8..64 bytes of every pixel's C-vector, so a
// workgroup takes a run of pixels with ALL channels (whole lines), and the moments of a (sample, group) are summed across workgroups:
//   gn_nhwc_moments_kernel: thread (row, col) owns the 16-byte packet `col` of the pixels row, row + rows, ...: float sums per channel
//       over at most GN_NHWC_PIX / rows pixels, per-channel totals over the rows in LDS, per-group totals added in float64 (atomics) to
//       moments[n][g] = (sum, sum of squares);
//   gn_nhwc_apply_kernel: the same mapping; scale / shift of the thread's PN channels once, then one pass: silu(x * sc + sh).
// Two reads (the second from L2 / MALL) and one write, as the NCHW kernel.
constexpr int GN_NHWC_PIX = 256;      // pixels per workgroup
constexpr int GN_NHWC_MAXC = 1024;

template <typename T>
__global__ void __launch_bounds__(256)
gn_nhwc_moments_kernel(int C, int HW, int groups, int rows, const T* __restrict__ x, double* __restrict__ moments)
{
    constexpr int PN = Packet<T>::N;
    const int ppp = C / PN;                                   // packets per pixel
    const int col = threadIdx.x % ppp, row = threadIdx.x / ppp;
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * GN_NHWC_PIX;
    const int p1 = min(p0 + GN_NHWC_PIX, HW);
    float a[PN], b[PN];
#pragma unroll
    for (int k = 0; k < PN; k++) { a[k] = 0.0f; b[k] = 0.0f; }
    if (row < rows) {
        const T* xs = x + ((size_t)n * HW) * C + (size_t)col * PN;
        for (int p = p0 + row; p < p1; p += rows) {
            Packet<T> q;
            q.load(xs + (size_t)p * C);
#pragma unroll
            for (int k = 0; k < PN; k++) { a[k] += q.v[k]; b[k] += q.v[k] * q.v[k]; }
        }
    }
    __shared__ float sa[GN_NHWC_MAXC], sb[GN_NHWC_MAXC];
    for (int c = threadIdx.x; c < C; c += blockDim.x) { sa[c] = 0.0f; sb[c] = 0.0f; }
    __syncthreads();
    if (row < rows) {
#pragma unroll
        for (int k = 0; k < PN; k++) { atomicAdd(&sa[col * PN + k], a[k]); atomicAdd(&sb[col * PN + k], b[k]); }
    }
    __syncthreads();
    const int Cg = C / groups;
    for (int g = threadIdx.x; g < groups; g += blockDim.x) {
        double s1 = 0.0, s2 = 0.0;
        for (int c = g * Cg; c < (g + 1) * Cg; c++) { s1 += (double)sa[c]; s2 += (double)sb[c]; }
        unsafeAtomicAdd(&moments[2 * ((size_t)n * groups + g)], s1);
        unsafeAtomicAdd(&moments[2 * ((size_t)n * groups + g) + 1], s2);
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
gn_nhwc_apply_kernel(int C, int HW, int groups, int rows, const T* __restrict__ x, const double* __restrict__ moments,
                     const float* __restrict__ weight, const float* __restrict__ bias, float eps, int apply_silu, T* __restrict__ y)
{
    constexpr int PN = Packet<T>::N;
    const int ppp = C / PN;
    const int col = threadIdx.x % ppp, row = threadIdx.x / ppp;
    if (row >= rows) return;
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * GN_NHWC_PIX;
    const int p1 = min(p0 + GN_NHWC_PIX, HW);
    const int Cg = C / groups;
    const double cnt = (double)Cg * (double)HW;
    float sc[PN], sh[PN];
    {
        int g_prev = -1;
        float mean = 0.0f, rstd = 0.0f;
#pragma unroll
        for (int k = 0; k < PN; k++) {
            const int c = col * PN + k, g = c / Cg;
            if (g != g_prev) {
                const double m = moments[2 * ((size_t)n * groups + g)] / cnt;
                double var = moments[2 * ((size_t)n * groups + g) + 1] / cnt - m * m;
                if (var < 0.0) var = 0.0;
                mean = (float)m;
                rstd = (float)(1.0 / sqrt(var + (double)eps));
                g_prev = g;
            }
            sc[k] = weight[c] * rstd;
            sh[k] = bias[c] - mean * sc[k];
        }
    }
    const size_t base = ((size_t)n * HW) * C + (size_t)col * PN;
    for (int p = p0 + row; p < p1; p += rows) {
        Packet<T> q;
        q.load(x + base + (size_t)p * C);
#pragma unroll
        for (int k = 0; k < PN; k++) {
            float v = q.v[k] * sc[k] + sh[k];
            if (apply_silu) v = v / (1.0f + expf(-v));
            q.v[k] = v;
        }
        q.store(y + base + (size_t)p * C);
    }
}

template <typename T>
int launch_gn_nhwc(void* stream, int N, int C, int HW, int groups, const T* x, const float* weight, const float* bias, float eps,
                   int apply_silu, T* y, double* moments)
{
    constexpr int PN = Packet<T>::N;
    if (N < 0 || C <= 0 || HW <= 0 || groups <= 0 || C % groups != 0 || !x || !weight || !bias || !y || !moments) return F3DG_ERR_BAD_ARG;
    if (C % PN != 0 || C / PN > 256 || C > GN_NHWC_MAXC) return F3DG_ERR_BAD_ARG;   // whole 16-byte packets per pixel, one packet column per thread
    if (N == 0) return F3DG_OK;
    if (((uintptr_t)x | (uintptr_t)y) & 15u) return F3DG_ERR_BAD_ARG;
    const int ppp = C / PN, rows = 256 / ppp;
    hipStream_t s = (hipStream_t)stream;
    F3DG_HIP_CHECK(hipMemsetAsync(moments, 0, sizeof(double) * 2 * (size_t)N * groups, s));
    const dim3 grid((unsigned)((HW + GN_NHWC_PIX - 1) / GN_NHWC_PIX), (unsigned)N);
    F3DG_KLAUNCH(gn_nhwc_moments_kernel<T>, grid, dim3(256), 0, s, C, HW, groups, rows, x, moments);
    F3DG_KLAUNCH(gn_nhwc_apply_kernel<T>, grid, dim3(256), 0, s, C, HW, groups, rows, x, moments, weight, bias, eps, apply_silu, y);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

} // namespace

// GroupNorm (+ SiLU) of a channels-last tensor, x and y [N][HW][C]; `moments` is scratch of 2 * N * groups doubles (zeroed here)
extern "C" int f3dg_group_norm_silu_nhwc(void* stream, int N, int C, int HW, int groups, const float* x, const float* weight,
                                         const float* bias, float eps, int apply_silu, float* y, double* moments)
{
    return launch_gn_nhwc<float>(stream, N, C, HW, groups, x, weight, bias, eps, apply_silu, y, moments);
}

extern "C" int f3dg_group_norm_silu_nhwc_bf16(void* stream, int N, int C, int HW, int groups, const uint16_t* x, const float* weight,
                                              const float* bias, float eps, int apply_silu, uint16_t* y, double* moments)
{
    return launch_gn_nhwc<unsigned short>(stream, N, C, HW, groups, x, weight, bias, eps, apply_silu, y, moments);
}

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
