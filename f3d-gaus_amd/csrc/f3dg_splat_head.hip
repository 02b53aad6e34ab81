// f3dg_splat_head.hip -- cycle-aggregative projection ("splat head") and the render epilogue.
//
// splat_head_kernel fuses everything GaussianSplatPredictor_gtunet.forward does after the U-Net
// (reference src/gaussian_predictor.py): get_pos_from_network_output :857-881 (pos = ray_dirs * depth + offset),
// the [pos,1] @ view_to_world bmm and perspective divide :961-970, sigmoid / exp / F.normalize activations
// :976-979, transform_rotations = quaternion_raw_multiply(cam_quat, q) :839-855 / :45-64, transform_SHs
// :821-837 with the constant sh<->v permutation matrices :649-655, flatten_vector :788-791 (NCHW -> N x C), and
// writes straight into the aggregated per-image Gaussian buffers at `n_offset`, which replaces the torch.cat
// chain of the cycle loop (visualize.py:336-340) -- ~15 small torch kernels + permutes + 9 reallocations become
// one bandwidth-bound pass: 96 B read + 96 B written per Gaussian.
//
// epilogue_kernel fuses the post-processing of render_predicted_more_v2_gof
// (src/gaussian_renderer/__init__.py:1043-1053 world normals, :881-909 depth_to_normal).
#include "f3dg_common.h"

namespace {

__global__ void __launch_bounds__(F3DG_BLOCK)
splat_head_kernel(int HW, const float* __restrict__ net_out, const float* __restrict__ depth,
                  const float* __restrict__ ray_dirs, const float* __restrict__ view_to_world,
                  const float* __restrict__ cam_quat, float squre_clip, long long n_total, long long n_offset,
                  float* __restrict__ xyz, float* __restrict__ opacity, float* __restrict__ scaling,
                  float* __restrict__ rotation, float* __restrict__ features_dc, float* __restrict__ features_rest,
                  float* __restrict__ unet_depth)
{
    const int n = blockIdx.x * F3DG_BLOCK + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= HW) return;
    const float* net = net_out + (size_t)b * 23 * HW + n;       // channel c at net[c * HW]
    const float* M = view_to_world + 16 * b;                     // row-major 4x4, row-vector convention (uniform)
    const float* qc = cam_quat + 4 * b;

    const float d = depth[(size_t)b * HW + n];
    // pos = ray_dirs * depth + offset  (two roundings, as torch evaluates it)
    const float p0 = ray_dirs[n] * d + net[0 * HW];
    const float p1 = ray_dirs[HW + n] * d + net[1 * HW];
    const float p2 = ray_dirs[2 * HW + n] * d + net[2 * HW];
    // [p,1] @ M
    const float w0 = p0 * M[0] + p1 * M[4] + p2 * M[8] + M[12];
    const float w1 = p0 * M[1] + p1 * M[5] + p2 * M[9] + M[13];
    const float w2 = p0 * M[2] + p1 * M[6] + p2 * M[10] + M[14];
    const float w3 = p0 * M[3] + p1 * M[7] + p2 * M[11] + M[15];
    const float den = w3 + 1e-10f;
    float X = w0 / den, Y = w1 / den;
    const float Z = w2 / den;
    if (squre_clip < 10.0f) {
        X = fminf(fmaxf(X, -squre_clip), squre_clip);
        Y = fminf(fmaxf(Y, -squre_clip), squre_clip);
    }

    const size_t o = (size_t)b * (size_t)n_total + (size_t)n_offset + n;
    xyz[3 * o + 0] = X; xyz[3 * o + 1] = Y; xyz[3 * o + 2] = Z;

    opacity[o] = 1.0f / (1.0f + expf(-net[3 * HW]));
    scaling[3 * o + 0] = expf(net[4 * HW]);
    scaling[3 * o + 1] = expf(net[5 * HW]);
    scaling[3 * o + 2] = expf(net[6 * HW]);

    // F.normalize(dim=channel, eps=1e-12) then Hamilton product cam_quat (x) q, real part first
    float bw = net[7 * HW], bx = net[8 * HW], by = net[9 * HW], bz = net[10 * HW];
    const float nrm = fmaxf(sqrtf(bw * bw + bx * bx + by * by + bz * bz), 1e-12f);
    bw /= nrm; bx /= nrm; by /= nrm; bz /= nrm;
    const float aw = qc[0], ax = qc[1], ay = qc[2], az = qc[3];
    rotation[4 * o + 0] = aw * bw - ax * bx - ay * by - az * bz;
    rotation[4 * o + 1] = aw * bx + ax * bw + ay * bz - az * by;
    rotation[4 * o + 2] = aw * by - ax * bz + ay * bw + az * bx;
    rotation[4 * o + 3] = aw * bz + ax * by - ay * bx + az * bw;

    features_dc[3 * o + 0] = net[11 * HW];
    features_dc[3 * o + 1] = net[12 * HW];
    features_dc[3 * o + 2] = net[13 * HW];

    // T = sh_to_v @ R @ v_to_sh with R = M[:3,:3]; the two constant matrices are signed permutations:
    //   X = sh_to_v @ R : rows (-R[1], R[2], -R[0]);   T[i] = (-X[i][1], X[i][2], -X[i][0])
    float Tm[3][3];
    {
        const float Xr[3][3] = { { -M[4], -M[5], -M[6] }, { M[8], M[9], M[10] }, { -M[0], -M[1], -M[2] } };
#pragma unroll
        for (int i = 0; i < 3; i++) { Tm[i][0] = -Xr[i][1]; Tm[i][1] = Xr[i][2]; Tm[i][2] = -Xr[i][0]; }
    }
    // rest'[t][c] = sum_s rest[s][c] * T[s][t], rest channel index = 14 + 3*s + c
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float s0 = net[(14 + c) * HW], s1 = net[(17 + c) * HW], s2 = net[(20 + c) * HW];
#pragma unroll
        for (int t = 0; t < 3; t++)
            features_rest[9 * o + 3 * t + c] = s0 * Tm[0][t] + s1 * Tm[1][t] + s2 * Tm[2][t];
    }
    unet_depth[o] = d;
}

// inverse of a 4x4 (row-major) by cofactors in float64, rounded to float32 (the camera-to-world matrix of the epilogue: one thread)
__device__ void invert4x4(const float* m_, float* out)
{
    double m[16], inv[16];
    for (int i = 0; i < 16; i++) m[i] = m_[i];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    const double r = 1.0 / det;
    for (int i = 0; i < 16; i++) out[i] = (float)(inv[i] * r);
}

// from_view: `c2w` holds the world_view matrices (row-vector convention, as the renderer receives them) and the kernel inverts their
// transposes itself -- the caller saves a torch.linalg.inv (a solver call and several launches per one-view call)
template <bool FROM_VIEW>
__global__ void __launch_bounds__(F3DG_BLOCK)
epilogue_kernel(int H, int W, const float* __restrict__ raster, const float* __restrict__ c2w, float fx, float fy,
                float* __restrict__ normal_world, float* __restrict__ depth_normal)
{
    const int HW = H * W;
    const int n = blockIdx.x * F3DG_BLOCK + threadIdx.x;
    const int v = blockIdx.y;
    __shared__ float s_c2w[16];
    if (FROM_VIEW) {
        if (threadIdx.x == 0) {
            float t[16];
            for (int r = 0; r < 4; r++)
                for (int c = 0; c < 4; c++) t[4 * r + c] = c2w[16 * v + 4 * c + r];      // world_view^T
            invert4x4(t, s_c2w);
        }
        __syncthreads();
    }
    if (n >= HW) return;
    const int y = n / W, x = n % W;
    const float* ras = raster + (size_t)v * F3DG_OUT_CHANNELS * HW;
    const float* Cw = FROM_VIEW ? s_c2w : c2w + 16 * v;     // row-major 4x4

    if (normal_world) {
        const float a = ras[3 * HW + n], b = ras[4 * HW + n], c = ras[5 * HW + n];
        const float nrm = fmaxf(sqrtf(a * a + b * b + c * c), 1e-12f);
        const float na = a / nrm, nb = b / nrm, nc = c / nrm;
        float* o = normal_world + (size_t)v * 3 * HW;
        o[n] = Cw[0] * na + Cw[1] * nb + Cw[2] * nc;
        o[HW + n] = Cw[4] * na + Cw[5] * nb + Cw[6] * nc;
        o[2 * HW + n] = Cw[8] * na + Cw[9] * nb + Cw[10] * nc;
    }
    if (depth_normal) {
        float* o = depth_normal + (size_t)v * 3 * HW;
        float r0 = 0, r1 = 0, r2 = 0;
        if (x >= 1 && x < W - 1 && y >= 1 && y < H - 1) {
            const float* dep = ras + 6 * HW;
            const float ax = 1.0f / fx, bx = -(W / 2.0f) / fx, ay = 1.0f / fy, by = -(H / 2.0f) / fy;
            // point(yy, xx) = depth * ([xx,yy,1] @ Kinv^T @ R^T) + origin
            auto point = [&](int yy, int xx, float& px, float& py, float& pz) {
                const float cx = xx * ax + bx, cy = yy * ay + by;
                const float dx = cx * Cw[0] + cy * Cw[1] + Cw[2];
                const float dy = cx * Cw[4] + cy * Cw[5] + Cw[6];
                const float dz = cx * Cw[8] + cy * Cw[9] + Cw[10];
                const float dd = dep[yy * W + xx];
                px = dd * dx + Cw[3]; py = dd * dy + Cw[7]; pz = dd * dz + Cw[11];
            };
            float ux, uy, uz, lx, ly, lz, rx, ry, rz, dx_, dy_, dz_;
            point(y + 1, x, ux, uy, uz);
            point(y - 1, x, lx, ly, lz);
            point(y, x + 1, rx, ry, rz);
            point(y, x - 1, dx_, dy_, dz_);
            const float ex = ux - lx, ey = uy - ly, ez = uz - lz;          // "dx": along rows
            const float gx = rx - dx_, gy = ry - dy_, gz = rz - dz_;       // "dy": along columns
            const float cx_ = ey * gz - ez * gy, cy_ = ez * gx - ex * gz, cz_ = ex * gy - ey * gx;
            const float nrm = fmaxf(sqrtf(cx_ * cx_ + cy_ * cy_ + cz_ * cz_), 1e-12f);
            r0 = cx_ / nrm; r1 = cy_ / nrm; r2 = cz_ / nrm;
        }
        o[n] = r0; o[HW + n] = r1; o[2 * HW + n] = r2;
    }
}


// 8-bit RGB frames for the video writer / the multi-GPU gather: dst[n][y][x][c] = (uint8)(255 * clamp(src[n][c][y][x], 0, 1))
// exactly as visualize.py:407,416 does on the host ((255 * np.clip(x, 0, 1)).astype(np.uint8): float32 product, truncation).
// src is planar with `src_channels` planes per frame (9 for the rasterizer output) of which the first three are read.
__global__ void __launch_bounds__(F3DG_BLOCK)
pack_frames_kernel(size_t HW, int src_channels, const float* __restrict__ src, unsigned char* __restrict__ dst)
{
    const size_t q = (size_t)blockIdx.x * F3DG_BLOCK + threadIdx.x;      // group of 4 pixels
    const size_t n = blockIdx.y;
    const float* s = src + n * (size_t)src_channels * HW;
    unsigned char* d = dst + n * 3 * HW;
    auto cv = [](float v) -> unsigned { return (unsigned)(255.0f * fminf(fmaxf(v, 0.0f), 1.0f)); };
    if (q * 4 + 4 <= HW && (HW & 3) == 0) {
        const float4 r = reinterpret_cast<const float4*>(s)[q];
        const float4 g = reinterpret_cast<const float4*>(s + HW)[q];
        const float4 b = reinterpret_cast<const float4*>(s + 2 * HW)[q];
        const unsigned w0 = cv(r.x) | (cv(g.x) << 8) | (cv(b.x) << 16) | (cv(r.y) << 24);
        const unsigned w1 = cv(g.y) | (cv(b.y) << 8) | (cv(r.z) << 16) | (cv(g.z) << 24);
        const unsigned w2 = cv(b.z) | (cv(r.w) << 8) | (cv(g.w) << 16) | (cv(b.w) << 24);
        unsigned* o = reinterpret_cast<unsigned*>(d + q * 12);
        o[0] = w0; o[1] = w1; o[2] = w2;
    } else {
        for (size_t p = q * 4; p < q * 4 + 4 && p < HW; p++) {
            d[3 * p] = (unsigned char)cv(s[p]);
            d[3 * p + 1] = (unsigned char)cv(s[HW + p]);
            d[3 * p + 2] = (unsigned char)cv(s[2 * HW + p]);
        }
    }
}

// The same frames written STRAIGHT into pinned host memory (device-accessible: hipHostMalloc / torch pin_memory): no staging copy in
// HBM and no copy command. HIP runs a device -> pinned-host hipMemcpyAsync as a shader copy over the whole chip
// (__amd_rocclr_copyBuffer, 0.43 ms per 23.6 MB on this box: tools/micro/d2h_engine.py) which takes wave slots from the next step's
// rendering; this kernel is PCIe-bound all the same, so a FEW workgroups (max_workgroups) move it in the same time and leave the
// rest of the chip alone. A lane produces one 16-byte chunk of the byte stream -- 5 1/3 pixels: byte B of a frame is channel B % 3 of
// pixel B / 3 -- so that a store instruction of a wave writes 1 KB of contiguous host memory (the fabric forwards partial lines as
// small PCIe writes); the 16 float reads of a lane are shared with its neighbours through L1 / L2.
__global__ void __launch_bounds__(F3DG_BLOCK)
pack_frames_host_kernel(size_t n_bytes, size_t HW, int src_channels, const float* __restrict__ src, unsigned char* __restrict__ dst)
{
    auto cv = [](float v) -> unsigned { return (unsigned)(255.0f * fminf(fmaxf(v, 0.0f), 1.0f)); };
    const size_t frame_bytes = 3 * HW, n_chunks = (n_bytes + 15) / 16;
    for (size_t c = (size_t)blockIdx.x * F3DG_BLOCK + threadIdx.x; c < n_chunks; c += (size_t)gridDim.x * F3DG_BLOCK) {
        const size_t b0 = 16 * c;
        size_t n = b0 / frame_bytes;
        unsigned rem = (unsigned)(b0 - n * frame_bytes);           // (a frame is < 4 GB)
        const float* s = src + n * (size_t)src_channels * HW;
        unsigned p = rem / 3u, ch = rem - 3u * p;
        unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (b0 + j < n_bytes)
                w[j >> 2] |= cv(s[(size_t)ch * HW + p]) << (8 * (j & 3));
            if (++ch == 3u) {
                ch = 0u;
                if (++p == (unsigned)HW) { p = 0u; s += (size_t)src_channels * HW; }       // the chunk runs into the next frame
            }
        }
        if (b0 + 16 <= n_bytes) {
            *reinterpret_cast<uint4*>(dst + b0) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int j = 0; b0 + j < n_bytes; j++) dst[b0 + j] = (unsigned char)(w[j >> 2] >> (8 * (j & 3)));
        }
    }
}

} // namespace

extern "C" int f3dg_splat_head(void* stream, int B, int H, int W, const float* net_out, const float* depth,
                               const float* ray_dirs, const float* view_to_world, const float* cam_quat,
                               float squre_clip, long long n_total, long long n_offset,
                               float* xyz, float* opacity, float* scaling, float* rotation,
                               float* features_dc, float* features_rest, float* unet_depth)
{
    if (B <= 0 || H <= 0 || W <= 0 || !net_out || !depth || !ray_dirs || !view_to_world || !cam_quat || !xyz ||
        !opacity || !scaling || !rotation || !features_dc || !features_rest || !unet_depth)
        return F3DG_ERR_BAD_ARG;
    const long long HW = (long long)H * W;
    if (n_offset < 0 || n_total < n_offset + HW) return F3DG_ERR_BAD_ARG;
    dim3 grid((unsigned)((HW + F3DG_BLOCK - 1) / F3DG_BLOCK), (unsigned)B);
    F3DG_KLAUNCH(splat_head_kernel, grid, dim3(F3DG_BLOCK), 0, (hipStream_t)stream, (int)HW, net_out, depth,
                       ray_dirs, view_to_world, cam_quat, squre_clip, n_total, n_offset, xyz, opacity, scaling,
                       rotation, features_dc, features_rest, unet_depth);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

extern "C" int f3dg_render_epilogue(void* stream, int n_views, int H, int W, const float* raster,
                                    const float* c2w, float fx, float fy,
                                    float* normal_world, float* depth_normal)
{
    if (n_views <= 0 || H <= 0 || W <= 0 || !raster || !c2w) return F3DG_ERR_BAD_ARG;
    if (!normal_world && !depth_normal) return F3DG_OK;
    dim3 grid((unsigned)(((long long)H * W + F3DG_BLOCK - 1) / F3DG_BLOCK), (unsigned)n_views);
    F3DG_KLAUNCH(epilogue_kernel<false>, grid, dim3(F3DG_BLOCK), 0, (hipStream_t)stream, H, W, raster, c2w, fx, fy,
                       normal_world, depth_normal);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

extern "C" int f3dg_render_epilogue_view(void* stream, int n_views, int H, int W, const float* raster,
                                         const float* world_view, float fx, float fy,
                                         float* normal_world, float* depth_normal)
{
    if (n_views <= 0 || H <= 0 || W <= 0 || !raster || !world_view) return F3DG_ERR_BAD_ARG;
    if (!normal_world && !depth_normal) return F3DG_OK;
    dim3 grid((unsigned)(((long long)H * W + F3DG_BLOCK - 1) / F3DG_BLOCK), (unsigned)n_views);
    F3DG_KLAUNCH(epilogue_kernel<true>, grid, dim3(F3DG_BLOCK), 0, (hipStream_t)stream, H, W, raster, world_view, fx, fy,
                       normal_world, depth_normal);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

extern "C" int f3dg_pack_frames(void* stream, int n_frames, int H, int W, int src_channels, const float* src,
                                unsigned char* dst)
{
    if (n_frames < 0 || H <= 0 || W <= 0 || src_channels < 3 || !src || !dst) return F3DG_ERR_BAD_ARG;
    if (n_frames == 0) return F3DG_OK;
    const size_t HW = (size_t)H * W;
    if (((uintptr_t)src & 15u) || ((uintptr_t)dst & 3u)) return F3DG_ERR_BAD_ARG;
    F3DG_KLAUNCH(pack_frames_kernel, dim3((unsigned)((HW / 4 + F3DG_BLOCK) / F3DG_BLOCK), (unsigned)n_frames), dim3(F3DG_BLOCK), 0,
                       (hipStream_t)stream, HW, src_channels, src, dst);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

extern "C" int f3dg_pack_frames_host(void* stream, int n_frames, int H, int W, int src_channels, const float* src,
                                     unsigned char* dst_host, int max_workgroups)
{
    if (n_frames < 0 || H <= 0 || W <= 0 || src_channels < 3 || !src || !dst_host) return F3DG_ERR_BAD_ARG;
    if (n_frames == 0) return F3DG_OK;
    if ((uintptr_t)dst_host & 15u) return F3DG_ERR_BAD_ARG;
    const size_t HW = (size_t)H * W, n_bytes = 3 * HW * (size_t)n_frames;
    if (3 * HW >= ((size_t)1 << 32)) return F3DG_ERR_BAD_ARG;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, dst_host) != hipSuccess || at.type != hipMemoryTypeHost) {      // pinned, device-accessible host memory only
        (void)hipGetLastError();
        return F3DG_ERR_BAD_ARG;
    }
    const size_t need = ((n_bytes + 15) / 16 + F3DG_BLOCK - 1) / F3DG_BLOCK;
    size_t grid = max_workgroups > 0 ? (size_t)max_workgroups : 64;
    if (grid > need) grid = need;
    F3DG_KLAUNCH(pack_frames_host_kernel, dim3((unsigned)grid), dim3(F3DG_BLOCK), 0, (hipStream_t)stream, n_bytes, HW, src_channels, src,
                 dst_host);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

// ------------------------------------------------------------------------------------------------ cycle inputs
// Hand-off of the cycle aggregation (reference visualize.py:311, 331-333): every rendered view becomes the next predictor
// input  cat(clamp(rgb, 0, 1), alpha)  [4, H, W]  with its median depth [1, H, W] as unet_depth. One pass over the 9-channel
// raster instead of clamp + two slices + cat (and the .cpu() round trips of the reference). Frame n = b * V + v of the raster is
// written to slot v * B + b, so that the B inputs of one novel view are one contiguous predictor batch.
namespace {
__global__ void __launch_bounds__(F3DG_BLOCK)
cycle_inputs_kernel(size_t HW, int B, int V, const float* __restrict__ raster, float* __restrict__ xin, float* __restrict__ depth)
{
    const size_t i = ((size_t)blockIdx.x * F3DG_BLOCK + threadIdx.x) * 4;
    if (i >= HW) return;
    const unsigned n = blockIdx.y, b = n / (unsigned)V, v = n % (unsigned)V;
    const float* src = raster + (size_t)n * F3DG_OUT_CHANNELS * HW;
    const size_t slot = (size_t)v * B + b;
    float* x = xin + slot * 4 * HW;
    auto clamp4 = [](float4 q) {      // torch.clamp(x, 0, 1): NaN stays NaN
        q.x = q.x < 0.0f ? 0.0f : (q.x > 1.0f ? 1.0f : q.x); q.y = q.y < 0.0f ? 0.0f : (q.y > 1.0f ? 1.0f : q.y);
        q.z = q.z < 0.0f ? 0.0f : (q.z > 1.0f ? 1.0f : q.z); q.w = q.w < 0.0f ? 0.0f : (q.w > 1.0f ? 1.0f : q.w);
        return q;
    };
    if (i + 4 <= HW) {
        *reinterpret_cast<float4*>(x + 0 * HW + i) = clamp4(*reinterpret_cast<const float4*>(src + 0 * HW + i));
        *reinterpret_cast<float4*>(x + 1 * HW + i) = clamp4(*reinterpret_cast<const float4*>(src + 1 * HW + i));
        *reinterpret_cast<float4*>(x + 2 * HW + i) = clamp4(*reinterpret_cast<const float4*>(src + 2 * HW + i));
        *reinterpret_cast<float4*>(x + 3 * HW + i) = *reinterpret_cast<const float4*>(src + 7 * HW + i);
        *reinterpret_cast<float4*>(depth + slot * HW + i) = *reinterpret_cast<const float4*>(src + 6 * HW + i);
    } else {
        for (size_t k = i; k < HW; k++) {
            for (int c = 0; c < 3; c++) { const float q = src[c * HW + k]; x[c * HW + k] = q < 0.0f ? 0.0f : (q > 1.0f ? 1.0f : q); }
            x[3 * HW + k] = src[7 * HW + k];
            depth[slot * HW + k] = src[6 * HW + k];
        }
    }
}
} // namespace

extern "C" int f3dg_cycle_inputs(void* stream, int B, int V, int H, int W, const float* raster, float* xin, float* depth)
{
    if (B < 0 || V <= 0 || H <= 0 || W <= 0 || !raster || !xin || !depth) return F3DG_ERR_BAD_ARG;
    if (B == 0) return F3DG_OK;
    const size_t HW = (size_t)H * W;
    if ((HW & 3u) || ((uintptr_t)raster & 15u) || ((uintptr_t)xin & 15u) || ((uintptr_t)depth & 15u)) return F3DG_ERR_BAD_ARG;
    F3DG_KLAUNCH(cycle_inputs_kernel, dim3((unsigned)((HW / 4 + F3DG_BLOCK - 1) / F3DG_BLOCK), (unsigned)(B * V)), dim3(F3DG_BLOCK), 0,
                       (hipStream_t)stream, HW, B, V, raster, xin, depth);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}
