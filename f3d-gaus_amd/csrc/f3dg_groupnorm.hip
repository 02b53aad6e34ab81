// f3dg_groupnorm.hip -- what the SongUNet backbone of the predictor needs between MIOpen's convolutions:
//   * fused GroupNorm (+ SiLU), NCHW and channels-last, float32 and bfloat16 activations, optionally with the bias of the convolution
//     that produced the input folded in (f3dg_group_norm_silu*, reference src/gaussian_predictor.py:250-262 GroupNorm, :318-323
//     `silu(norm(x))`);
//   * the residual join of a block (f3dg_residual_join, :325-327): second convolution's bias + skip path (+ its bias) + skip_scale.
//
// The backbone calls GroupNorm 78 times per pass, each followed by SiLU in the residual blocks. On PyTorch-ROCm that is three
// bandwidth-bound kernels per call (row moments, scale/shift, silu: five passes over the tensor); the NCHW kernel is ONE:
// a workgroup owns one (sample, group) slab of Cg x H x W contiguous floats, accumulates sum and sum of squares in
// float64, and re-reads the slab (<= 1 MiB: L2-resident) to write silu(weight * (x - mean) * rstd + bias). Two HBM
// passes (one read, one write) instead of five. The channels-last pair of kernels is described where it is defined.
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
// 8 float16 (the fp16 option of the backbone: 10 mantissa bits at the bf16 MFMA rate; T = _Float16)
template <> struct Packet<_Float16> {
    static constexpr int N = 8;
    float v[8];
    __device__ __forceinline__ void load(const _Float16* p)
    {
        const uint4 q = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int i = 0; i < 4; i++) {
            v[2 * i] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[i] & 0xFFFFu));
            v[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[i] >> 16));
        }
    }
    __device__ __forceinline__ void store(_Float16* p) const
    {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
            w[i] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v[2 * i]) | ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)v[2 * i + 1]) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
// SiLU. float32 activations: expf and an IEEE division (the backbone's float32 parity bar is 2e-5 of the reference fixture).
// bfloat16 activations: the result is rounded to 8 bits, so hardware exp2 / reciprocal (~1 ulp of float32 each) are invisible and the
// ~30 instructions per element of the accurate form -- which made the kernel VALU-bound -- become 5.
template <typename T> __device__ __forceinline__ float silu_of(float v)
{
    if constexpr (sizeof(T) == 2)
        return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
    else
        return v / (1.0f + expf(-v));
}
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(unsigned short x) { return bf16_to_f32(x); }
__device__ __forceinline__ void from_f32(float& d, float x) { d = x; }
__device__ __forceinline__ void from_f32(unsigned short& d, float x) { d = f32_to_bf16(x); }
__device__ __forceinline__ float to_f32(_Float16 x) { return (float)x; }
__device__ __forceinline__ void from_f32(_Float16& d, float x) { d = (_Float16)x; }

// T = float, unsigned short holding bfloat16, or _Float16 (the bf16 / fp16 options of the backbone: statistics and arithmetic stay float32 / float64).
// pre_bias (nullable, [C] float32): added to x on the way in -- the bias of the convolution that produced x, which PyTorch-ROCm would
// otherwise apply as a separate read + write pass behind MIOpen's kernel (108 such passes per backbone pass, profiles/r04_final/unet.md).
template <typename T>
__global__ void __launch_bounds__(GN_THREADS)
group_norm_silu_kernel(int C, int HW, int groups, const T* __restrict__ x, const float* __restrict__ pre_bias, const float* __restrict__ weight,
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
    const int hwp = vec ? HW / PN : 1;
    for (size_t i = threadIdx.x; i < np; i += GN_THREADS) {
        Packet<T> p;
        p.load(xs + i * PN);
        const float pb = pre_bias ? pre_bias[g * Cg + (int)(i / hwp)] : 0.0f;
        float a = 0.0f, b = 0.0f;
#pragma unroll
        for (int k = 0; k < PN; k++) { const float v = p.v[k] + pb; a += v; b += v * v; }
        s1 += (double)a;
        s2 += (double)b;
    }
    for (size_t i = np * PN + threadIdx.x; i < slab; i += GN_THREADS) {
        const float v = to_f32(xs[i]) + (pre_bias ? pre_bias[g * Cg + (int)(i / HW)] : 0.0f);
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
        for (size_t i = threadIdx.x; i < np; i += GN_THREADS) {
            const int c = g * Cg + (int)(i / hwp);
            const float pb = pre_bias ? pre_bias[c] : 0.0f;
            const float sc = weight[c] * rstd;
            const float sh = bias[c] - mean * sc;
            Packet<T> p;
            p.load(xs + i * PN);
#pragma unroll
            for (int k = 0; k < PN; k++) {
                float v = (p.v[k] + pb) * sc + sh;
                if (apply_silu) v = silu_of<T>(v);
                p.v[k] = v;
            }
            p.store(ys + i * PN);
        }
    } else {
        for (size_t i = threadIdx.x; i < slab; i += GN_THREADS) {
            const int c = g * Cg + (int)(i / HW);
            const float pb = pre_bias ? pre_bias[c] : 0.0f;
            const float sc = weight[c] * rstd;
            float v = (to_f32(xs[i]) + pb) * sc + (bias[c] - mean * sc);
            if (apply_silu) v = silu_of<T>(v);
            from_f32(ys[i], v);
        }
    }
}

template <typename T>
int launch_gn(void* stream, int N, int C, int HW, int groups, const T* x, const float* pre_bias, const float* weight, const float* bias, float eps,
              int apply_silu, T* y)
{
    if (N < 0 || C <= 0 || HW <= 0 || groups <= 0 || C % groups != 0 || !x || !weight || !bias || !y) return F3DG_ERR_BAD_ARG;
    if (N == 0) return F3DG_OK;
    if (((uintptr_t)x | (uintptr_t)y) & 15u) return F3DG_ERR_BAD_ARG;       // 16-byte loads / stores
    F3DG_KLAUNCH(group_norm_silu_kernel<T>, dim3((unsigned)(N * groups)), dim3(GN_THREADS), 0, (hipStream_t)stream, C, HW,
                       groups, x, pre_bias, weight, bias, eps, apply_silu, y);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

// ---- channels-last (NHWC) variant ---------------------------------------------------------------------------------------------------
// MIOpen's fastest convolutions on gfx950 are its NHWC kernels; handed NCHW tensors it wraps them in batched_transpose launches
// (15 % of a bf16 pass of the backbone, profiles/r04_final/unet.md), and torch's GroupNorm converts a channels_last tensor back, so the
// layout only pays if GroupNorm keeps it. x is [N][HW][C]: a group's Cg channels are 8..64 bytes of every pixel's C-vector, so a
// workgroup takes a run of pixels with ALL channels (whole lines), and the moments of a (sample, group) are summed across workgroups:
//   gn_nhwc_moments_kernel: thread (row, col) owns the 16-byte packet `col` of the pixels row, row + rows, ...: float sums per channel
//       over at most GN_NHWC_PIX / rows pixels, per-channel totals over the rows in LDS, per-group totals in float64 written to
//       partial[n][g][block] = (sum, sum of squares); gn_nhwc_finish_kernel adds the blocks in order -> (mean, rstd) per (sample, group);
//   gn_nhwc_apply_kernel: the same mapping; scale / shift of the thread's PN channels once, then one pass: silu((x + pre_bias) * sc + sh).
// Two reads (the second from L2 / MALL) and one write, as the NCHW kernel.
constexpr int GN_NHWC_PIX = 512;      // pixels per workgroup
constexpr int GN_NHWC_MAXC = 1024;
constexpr int GN_NHWC_UNROLL = 4;     // independent 16-byte loads in flight per thread

template <typename T>
__global__ void __launch_bounds__(256)
gn_nhwc_moments_kernel(int C, int HW, int groups, int rows, const T* __restrict__ x, const float* __restrict__ pre_bias, double* __restrict__ moments)
{
    constexpr int PN = Packet<T>::N;
    const int ppp = C / PN;                                   // packets per pixel
    const int col = threadIdx.x % ppp, row = threadIdx.x / ppp;
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * GN_NHWC_PIX;
    const int p1 = min(p0 + GN_NHWC_PIX, HW);
    float a[PN], b[PN], pb[PN];
#pragma unroll
    for (int k = 0; k < PN; k++) { a[k] = 0.0f; b[k] = 0.0f; pb[k] = (pre_bias && row < rows) ? pre_bias[col * PN + k] : 0.0f; }
    if (row < rows) {
        const T* xs = x + ((size_t)n * HW) * C + (size_t)col * PN;
        for (int p = p0 + row; p < p1; p += rows * GN_NHWC_UNROLL) {
            Packet<T> q[GN_NHWC_UNROLL];
#pragma unroll
            for (int u = 0; u < GN_NHWC_UNROLL; u++)
                if (p + u * rows < p1) q[u].load(xs + (size_t)(p + u * rows) * C);
#pragma unroll
            for (int u = 0; u < GN_NHWC_UNROLL; u++)
                if (p + u * rows < p1) {
#pragma unroll
                    for (int k = 0; k < PN; k++) { const float v = q[u].v[k] + pb[k]; a[k] += v; b[k] += v * v; }
                }
        }
    }
    // per-channel totals over the rows: every (row, channel) has its own LDS word (rows * C <= 256 * PN), channel c is summed by thread c
    __shared__ float sa[256 * PN], sb[256 * PN];
    if (row < rows) {
#pragma unroll
        for (int k = 0; k < PN; k++) { sa[row * C + col * PN + k] = a[k]; sb[row * C + col * PN + k] = b[k]; }
    }
    __syncthreads();
    __shared__ float ca[GN_NHWC_MAXC], cb[GN_NHWC_MAXC];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float t1 = 0.0f, t2 = 0.0f;
        for (int r = 0; r < rows; r++) { t1 += sa[r * C + c]; t2 += sb[r * C + c]; }
        ca[c] = t1; cb[c] = t2;
    }
    __syncthreads();
    // per-group totals in float64, one (sum, sum of squares) pair per WORKGROUP: partial[n][g][block]. No atomics -- the second stage
    // (gn_nhwc_finish_kernel) adds a sample's blocks up in block order, so the statistics, and with them the whole backbone pass, are
    // bit-reproducible from run to run (until round 5 the blocks added into 8 atomic slots in arrival order)
    const int Cg = C / groups;
    for (int g = threadIdx.x; g < groups; g += blockDim.x) {
        double s1 = 0.0, s2 = 0.0;
        for (int c = g * Cg; c < (g + 1) * Cg; c++) { s1 += (double)ca[c]; s2 += (double)cb[c]; }
        double* part = moments + 2 * (((size_t)n * groups + g) * gridDim.x + blockIdx.x);       // [n][g][block]: the second stage reads a (sample, group)'s blocks contiguously
        part[0] = s1;
        part[1] = s2;
    }
}

// second stage: ONE WAVE per (sample, group) adds its nb partial pairs -- lane l takes blocks l, l + 64, ... in order, then a fixed
// butterfly over the lanes: the same association every run -- and leaves mean and 1 / sqrt(var + eps) as two floats behind the partials
// (stats[n][g]): the apply kernel's threads read them instead of each recomputing them in float64. (One THREAD per pair, the first
// version, walked 128 strided pairs serially: 12 us per launch, 78 launches per pass.)
__global__ void __launch_bounds__(256)
gn_nhwc_finish_kernel(int N, int groups, int nb, double cnt, float eps, const double* __restrict__ partial, float2* __restrict__ stats)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N * groups) return;
    const int lane = threadIdx.x & 63;
    const int n = i / groups, g = i % groups;
    double m1 = 0.0, m2 = 0.0;
    for (int b = lane; b < nb; b += 64) {
        const double* p = partial + 2 * (((size_t)n * groups + g) * nb + b);
        m1 += p[0]; m2 += p[1];
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        m1 += __shfl_xor(m1, m, 64);
        m2 += __shfl_xor(m2, m, 64);
    }
    if (lane == 0) {
        const double m = m1 / cnt;
        double var = m2 / cnt - m * m;
        if (var < 0.0) var = 0.0;
        stats[i] = make_float2((float)m, (float)(1.0 / sqrt(var + (double)eps)));
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
gn_nhwc_apply_kernel(int C, int HW, int groups, int rows, const T* __restrict__ x, const float* __restrict__ pre_bias, const float2* __restrict__ stats,
                     const float* __restrict__ weight, const float* __restrict__ bias, int apply_silu, T* __restrict__ y)
{
    constexpr int PN = Packet<T>::N;
    const int ppp = C / PN;
    const int col = threadIdx.x % ppp, row = threadIdx.x / ppp;
    if (row >= rows) return;
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * GN_NHWC_PIX;
    const int p1 = min(p0 + GN_NHWC_PIX, HW);
    const int Cg = C / groups;
    float sc[PN], sh[PN], pb[PN];
#pragma unroll
    for (int k = 0; k < PN; k++) {
        const int c = col * PN + k;
        const float2 st = stats[(size_t)n * groups + c / Cg];        // (mean, rstd)
        sc[k] = weight[c] * st.y;
        sh[k] = bias[c] - st.x * sc[k];
        pb[k] = pre_bias ? pre_bias[c] : 0.0f;
    }
    const size_t base = ((size_t)n * HW) * C + (size_t)col * PN;
    for (int p = p0 + row; p < p1; p += rows * GN_NHWC_UNROLL) {
        Packet<T> q[GN_NHWC_UNROLL];
#pragma unroll
        for (int u = 0; u < GN_NHWC_UNROLL; u++)
            if (p + u * rows < p1) q[u].load(x + base + (size_t)(p + u * rows) * C);
#pragma unroll
        for (int u = 0; u < GN_NHWC_UNROLL; u++)
            if (p + u * rows < p1) {
#pragma unroll
                for (int k = 0; k < PN; k++) {
                    float v = (q[u].v[k] + pb[k]) * sc[k] + sh[k];
                    if (apply_silu) v = silu_of<T>(v);
                    q[u].v[k] = v;
                }
                q[u].store(y + base + (size_t)(p + u * rows) * C);
            }
    }
}

extern "C" size_t f3dg_group_norm_nhwc_scratch_bytes(int N, int HW, int groups);

template <typename T>
int launch_gn_nhwc(void* stream, int N, int C, int HW, int groups, const T* x, const float* pre_bias, const float* weight, const float* bias, float eps,
                   int apply_silu, T* y, double* moments, size_t moments_bytes)
{
    constexpr int PN = Packet<T>::N;
    if (N < 0 || C <= 0 || HW <= 0 || groups <= 0 || C % groups != 0 || !x || !weight || !bias || !y || !moments) return F3DG_ERR_BAD_ARG;
    // the scratch grew when the statistics became atomics-free (round 5: a partial pair per workgroup): the caller says how much it
    // handed over, an allocation sized by an older header is refused instead of being written past its end
    if (N > 0 && moments_bytes < f3dg_group_norm_nhwc_scratch_bytes(N, HW, groups)) return F3DG_ERR_WORKSPACE;
    if (C % PN != 0 || C / PN > 256 || C > GN_NHWC_MAXC) return F3DG_ERR_BAD_ARG;   // whole 16-byte packets per pixel, one packet column per thread
    if (N == 0) return F3DG_OK;
    if (((uintptr_t)x | (uintptr_t)y) & 15u) return F3DG_ERR_BAD_ARG;
    const int ppp = C / PN, rows = 256 / ppp;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (HW + GN_NHWC_PIX - 1) / GN_NHWC_PIX;
    float2* stats = reinterpret_cast<float2*>(moments + 2 * (size_t)N * nb * groups);
    const dim3 grid((unsigned)nb, (unsigned)N);
    F3DG_KLAUNCH(gn_nhwc_moments_kernel<T>, grid, dim3(256), 0, s, C, HW, groups, rows, x, pre_bias, moments);
    F3DG_KLAUNCH(gn_nhwc_finish_kernel, dim3((unsigned)((N * groups + 3) / 4)), dim3(256), 0, s, N, groups, nb, (double)(C / groups) * (double)HW, eps,
                 moments, stats);
    F3DG_KLAUNCH(gn_nhwc_apply_kernel<T>, grid, dim3(256), 0, s, C, HW, groups, rows, x, pre_bias, stats, weight, bias, apply_silu, y);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

// ---- the residual join of a block -------------------------------------------------------------------------------------------------
// y = ((a + bias_a[c]) + (b + bias_b[c])) * scale, the three elementwise passes behind a residual block's second convolution
// (reference src/gaussian_predictor.py:325-327: `x = conv1(...)` [+ bias], `x = x + skip(orig)` [+ bias], `x = x * skip_scale`) in one:
// same float32 operations in the same order, rounded to the activation type once. nhwc = 0: [N][C][HW], channel = (i / HW) % C;
// nhwc = 1: [N][HW][C], channel = i % C. Either bias may be null. y may be a or b.
template <typename T>
__global__ void __launch_bounds__(256)
residual_join_kernel(size_t n_packets, int C, int HW, int nhwc, int packet_in_channel, const T* __restrict__ a, const float* __restrict__ bias_a,
                     const T* __restrict__ b, const float* __restrict__ bias_b, float scale, T* __restrict__ y)
{
    constexpr int PN = Packet<T>::N;
    const size_t first = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    if (nhwc) {
        // channels-last: C is a multiple of PN and the launch makes the stride a multiple of the packets per pixel, so a thread meets the
        // same PN channels in every iteration: its 2 PN bias values are loaded once (as 2 PN loads per packet they were what bounded
        // the first version of this kernel: 19 memory instructions per packet of which 3 moved data)
        const int c0 = (int)((first * PN) % (size_t)C);
        float ba[PN], bb[PN];
#pragma unroll
        for (int k = 0; k < PN; k++) { ba[k] = bias_a ? bias_a[c0 + k] : 0.0f; bb[k] = bias_b ? bias_b[c0 + k] : 0.0f; }
        for (size_t i = first; i < n_packets; i += stride) {
            Packet<T> pa, pb;
            pa.load(a + i * PN);
            pb.load(b + i * PN);
#pragma unroll
            for (int k = 0; k < PN; k++) {
                const float va = bias_a ? pa.v[k] + ba[k] : pa.v[k];
                const float vb = bias_b ? pb.v[k] + bb[k] : pb.v[k];
                pa.v[k] = (va + vb) * scale;
            }
            pa.store(y + i * PN);
        }
        return;
    }
    for (size_t i = first; i < n_packets; i += stride) {
        Packet<T> pa, pb;
        pa.load(a + i * PN);
        pb.load(b + i * PN);
        const size_t e0 = i * PN;
        // NCHW with HW a multiple of PN: a packet stays inside a channel plane; otherwise element by element
        const int c0 = (int)((e0 / (size_t)HW) % (size_t)C);
        const float ba0 = bias_a ? bias_a[c0] : 0.0f, bb0 = bias_b ? bias_b[c0] : 0.0f;
#pragma unroll
        for (int k = 0; k < PN; k++) {
            float ba = ba0, bb = bb0;
            if (!packet_in_channel) {
                const int c = (int)(((e0 + k) / (size_t)HW) % (size_t)C);
                ba = bias_a ? bias_a[c] : 0.0f; bb = bias_b ? bias_b[c] : 0.0f;
            }
            const float va = bias_a ? pa.v[k] + ba : pa.v[k];
            const float vb = bias_b ? pb.v[k] + bb : pb.v[k];
            pa.v[k] = (va + vb) * scale;
        }
        pa.store(y + i * PN);
    }
}

template <typename T>
int launch_join(void* stream, int N, int C, int HW, int nhwc, const T* a, const float* bias_a, const T* b, const float* bias_b, float scale, T* y)
{
    constexpr int PN = Packet<T>::N;
    if (N < 0 || C <= 0 || HW <= 0 || !a || !b || !y) return F3DG_ERR_BAD_ARG;
    const size_t total = (size_t)N * C * HW;
    if (total == 0) return F3DG_OK;
    if (total % PN != 0 || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)y) & 15u)) return F3DG_ERR_BAD_ARG;
    if (nhwc && C % PN != 0) return F3DG_ERR_BAD_ARG;
    const size_t np = total / PN;
    // channels-last: blocks * 256 must be a multiple of the packets per pixel (C / PN), which the kernel relies on
    unsigned blocks = (unsigned)((np + 255) / 256 < 16383 ? (np + 255) / 256 : 16383);
    if (nhwc) {
        const unsigned ppp = (unsigned)(C / PN);
        unsigned m = ppp;                       // smallest block count whose 256-fold is a multiple of ppp: ppp / gcd(ppp, 256)
        for (unsigned t = 256; t > 1 && (m % 2u) == 0; t >>= 1) m >>= 1;
        blocks = ((blocks + m - 1) / m) * m;
    }
    F3DG_KLAUNCH(residual_join_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, np, C, HW, nhwc, HW % PN == 0 ? 1 : 0, a, bias_a, b, bias_b, scale, y);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

} // namespace

extern "C" int f3dg_residual_join(void* stream, int N, int C, int HW, int nhwc, const float* a, const float* bias_a, const float* b,
                                  const float* bias_b, float scale, float* y)
{
    return launch_join<float>(stream, N, C, HW, nhwc, a, bias_a, b, bias_b, scale, y);
}

extern "C" int f3dg_residual_join_bf16(void* stream, int N, int C, int HW, int nhwc, const uint16_t* a, const float* bias_a, const uint16_t* b,
                                       const float* bias_b, float scale, uint16_t* y)
{
    return launch_join<unsigned short>(stream, N, C, HW, nhwc, a, bias_a, b, bias_b, scale, y);
}

extern "C" int f3dg_residual_join_f16(void* stream, int N, int C, int HW, int nhwc, const uint16_t* a, const float* bias_a, const uint16_t* b,
                                      const float* bias_b, float scale, uint16_t* y)
{
    return launch_join<_Float16>(stream, N, C, HW, nhwc, reinterpret_cast<const _Float16*>(a), bias_a, reinterpret_cast<const _Float16*>(b), bias_b, scale,
                                 reinterpret_cast<_Float16*>(y));
}

// bytes of the `moments` scratch of the channels-last GroupNorm: the workgroups' partial sums + the (mean, rstd) pairs
extern "C" size_t f3dg_group_norm_nhwc_scratch_bytes(int N, int HW, int groups)
{
    if (N <= 0 || HW <= 0 || groups <= 0) return 0;
    const size_t nb = (size_t)(HW + GN_NHWC_PIX - 1) / GN_NHWC_PIX;
    return sizeof(double) * 2 * (size_t)N * nb * groups + sizeof(float2) * (size_t)N * groups;
}

// GroupNorm (+ SiLU) with the producing convolution's bias folded in (pre_bias, nullable): NCHW ...
extern "C" int f3dg_group_norm_silu_pb(void* stream, int N, int C, int HW, int groups, const float* x, const float* pre_bias, const float* weight,
                                       const float* bias, float eps, int apply_silu, float* y)
{
    return launch_gn<float>(stream, N, C, HW, groups, x, pre_bias, weight, bias, eps, apply_silu, y);
}

extern "C" int f3dg_group_norm_silu_pb_bf16(void* stream, int N, int C, int HW, int groups, const uint16_t* x, const float* pre_bias, const float* weight,
                                            const float* bias, float eps, int apply_silu, uint16_t* y)
{
    return launch_gn<unsigned short>(stream, N, C, HW, groups, x, pre_bias, weight, bias, eps, apply_silu, y);
}

extern "C" int f3dg_group_norm_silu_pb_f16(void* stream, int N, int C, int HW, int groups, const uint16_t* x, const float* pre_bias, const float* weight,
                                           const float* bias, float eps, int apply_silu, uint16_t* y)
{
    return launch_gn<_Float16>(stream, N, C, HW, groups, reinterpret_cast<const _Float16*>(x), pre_bias, weight, bias, eps, apply_silu, reinterpret_cast<_Float16*>(y));
}

// ... and channels-last
extern "C" int f3dg_group_norm_silu_nhwc_pb_f16(void* stream, int N, int C, int HW, int groups, const uint16_t* x, const float* pre_bias,
                                                const float* weight, const float* bias, float eps, int apply_silu, uint16_t* y, double* moments, size_t moments_bytes)
{
    return launch_gn_nhwc<_Float16>(stream, N, C, HW, groups, reinterpret_cast<const _Float16*>(x), pre_bias, weight, bias, eps, apply_silu,
                                    reinterpret_cast<_Float16*>(y), moments, moments_bytes);
}

extern "C" int f3dg_group_norm_silu_nhwc_pb(void* stream, int N, int C, int HW, int groups, const float* x, const float* pre_bias, const float* weight,
                                            const float* bias, float eps, int apply_silu, float* y, double* moments, size_t moments_bytes)
{
    return launch_gn_nhwc<float>(stream, N, C, HW, groups, x, pre_bias, weight, bias, eps, apply_silu, y, moments, moments_bytes);
}

extern "C" int f3dg_group_norm_silu_nhwc_pb_bf16(void* stream, int N, int C, int HW, int groups, const uint16_t* x, const float* pre_bias,
                                                 const float* weight, const float* bias, float eps, int apply_silu, uint16_t* y, double* moments, size_t moments_bytes)
{
    return launch_gn_nhwc<unsigned short>(stream, N, C, HW, groups, x, pre_bias, weight, bias, eps, apply_silu, y, moments, moments_bytes);
}

// GroupNorm (+ SiLU) of a channels-last tensor, x and y [N][HW][C]; `moments` is scratch of f3dg_group_norm_nhwc_scratch_bytes(N, HW, groups) bytes
extern "C" int f3dg_group_norm_silu_nhwc(void* stream, int N, int C, int HW, int groups, const float* x, const float* weight,
                                         const float* bias, float eps, int apply_silu, float* y, double* moments, size_t moments_bytes)
{
    return launch_gn_nhwc<float>(stream, N, C, HW, groups, x, nullptr, weight, bias, eps, apply_silu, y, moments, moments_bytes);
}

extern "C" int f3dg_group_norm_silu_nhwc_bf16(void* stream, int N, int C, int HW, int groups, const uint16_t* x, const float* weight,
                                              const float* bias, float eps, int apply_silu, uint16_t* y, double* moments, size_t moments_bytes)
{
    return launch_gn_nhwc<unsigned short>(stream, N, C, HW, groups, x, nullptr, weight, bias, eps, apply_silu, y, moments, moments_bytes);
}

extern "C" int f3dg_group_norm_silu(void* stream, int N, int C, int HW, int groups, const float* x, const float* weight,
                                    const float* bias, float eps, int apply_silu, float* y)
{
    return launch_gn<float>(stream, N, C, HW, groups, x, nullptr, weight, bias, eps, apply_silu, y);
}

extern "C" int f3dg_group_norm_silu_bf16(void* stream, int N, int C, int HW, int groups, const uint16_t* x, const float* weight,
                                         const float* bias, float eps, int apply_silu, uint16_t* y)
{
    return launch_gn<unsigned short>(stream, N, C, HW, groups, x, nullptr, weight, bias, eps, apply_silu, y);
}
