// f3dg_render.hip -- per-tile front-to-back GOF compositing (the roofline kernel of the path).
//
// Replaces renderCUDA<3> (reference RAST/cuda_rasterizer/forward.cu:409-612): for every pixel of a 16x16 tile,
// walk the tile's depth-sorted Gaussian list, intersect the pixel ray with each Gaussian's quadric
// (view2gaussian: Sigma', B, C), turn the minimum of the quadric along the ray into an alpha, and blend RGB,
// view-space normal, median depth, alpha and the 2DGS-style distortion term front to back.
//
// MI355X shape: one 256-thread workgroup (4 wave64, each a 16x4 pixel strip) per (view, tile); ALL views of a
// call are one launch. The tile's list is staged through LDS 256 Gaussians per round as whole 64-byte records
// (one float4 x4 coalesced-by-record gather per thread), and every lane then reads the SAME record per step
// (LDS broadcast, 4 ds_read_b128 per Gaussian per wave). The blend itself is a strict per-pixel recurrence in
// float32/float64 whose operation order is the reference's (built with -ffp-contract=off): the exponent
// -(1/2)(C - B^2/4A) cancels 1e5..1e6 x and any re-association moves isolated pixels by 1e-2.
//
// Only `done`-voting and staging are cooperative; there is no inter-pixel arithmetic, hence no MFMA.
#include "f3dg_common.h"

namespace {

template <bool SAVE_AUX>
__global__ void __launch_bounds__(F3DG_BLOCK)
render_fwd_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
                  const F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                  const unsigned* __restrict__ point_list, const F3dgRec* __restrict__ rec,
                  const float* __restrict__ background, int bg_per_view, float* __restrict__ out_color,
                  float* __restrict__ final_T, unsigned* __restrict__ n_contrib)
{
    // XCD-aware placement: consecutive workgroup ids land on different XCDs (id % 8), so give every XCD its
    // own views: all tiles of a view then share one XCD's L2 for the record gather.
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned slot = blockIdx.x >> 3;
    const unsigned view = (slot / (unsigned)T) * 8u + xcd;
    const unsigned tile = slot % (unsigned)T;
    if (view >= (unsigned)V)
        return;

    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    const unsigned lx = threadIdx.x & 15u, ly = threadIdx.x >> 4;
    const unsigned pix_x = tile_x * F3DG_TILE + lx, pix_y = tile_y * F3DG_TILE + ly;
    const bool inside = pix_x < (unsigned)W && pix_y < (unsigned)H;
    const size_t HW = (size_t)H * W;
    const size_t pix_id = (size_t)W * pix_y + pix_x;
    const float pixf_x = (float)pix_x + 0.5f, pixf_y = (float)pix_y + 0.5f;
    const float ray_x = (float)((pixf_x - W / 2.) / focal_x);
    const float ray_y = (float)((pixf_y - H / 2.) / focal_y);

    uint2 range = ranges[(size_t)view * T + tile];
    if (hdr->overflow) range = make_uint2(0, 0);
    const int rounds = (int)((range.y - range.x + F3DG_BLOCK - 1) / F3DG_BLOCK);
    int toDo = (int)(range.y - range.x);

    __shared__ float4 staged[F3DG_BLOCK * 4];     // 256 records x 64 B = 16 KiB

    const F3dgRec* vrec = rec + (size_t)view * P;

    bool done = !inside;
    float Tr = 1.0f;
    unsigned contributor = 0, last_contributor = 0, max_contributor = (unsigned)-1;
    float C0 = 0, C1 = 0, C2 = 0, C3 = 0, C4 = 0, C5 = 0, C6 = 0, C7 = 0;
    float dist1 = 0, dist2 = 0, distortion = 0;

    for (int i = 0; i < rounds; i++, toDo -= F3DG_BLOCK) {
        const int num_done = __syncthreads_count(done);
        if (num_done == F3DG_BLOCK)
            break;

        const unsigned progress = (unsigned)i * F3DG_BLOCK + threadIdx.x;
        if (range.x + progress < range.y) {
            const unsigned id = point_list[range.x + progress];
            const float4* src = reinterpret_cast<const float4*>(vrec + id);
            const float4 a = src[0], b = src[1], c = src[2], d = src[3];
            staged[threadIdx.x * 4 + 0] = a;
            staged[threadIdx.x * 4 + 1] = b;
            staged[threadIdx.x * 4 + 2] = c;
            staged[threadIdx.x * 4 + 3] = d;
        }
        __syncthreads();

        const int n = min(F3DG_BLOCK, toDo);
        for (int j = 0; !done && j < n; j++) {
            contributor++;
            const float4 q0 = staged[j * 4 + 0];      // v0 v1 v2 v3
            const float4 q1 = staged[j * 4 + 1];      // v4 v5 v6 v7
            const float4 q2 = staged[j * 4 + 2];      // v8 v9 opac r

            const float n0 = q0.x * ray_x + q0.y * ray_y + q0.z;
            const float n1 = q0.y * ray_x + q0.w * ray_y + q1.x;
            const float n2 = q0.z * ray_x + q1.x * ray_y + q1.y;

            const double AA = ray_x * n0 + ray_y * n1 + n2;
            const double BB = 2 * (q1.z * ray_x + q1.w * ray_y + q2.x);
            const float CC = q2.y;

            const float t = (float)(-BB / (2 * AA));
            if (t <= F3DG_NEAR_PLANE)
                continue;

            const double min_value = -(BB / AA) * (BB / 4.) + CC;
            float power = (float)(-0.5f * min_value);
            if (power > 0.0f)
                power = 0.0f;

            const float alpha = fminf(0.99f, q2.z * expf(power));
            if (alpha < 1.0f / 255.0f)
                continue;
            const float test_T = Tr * (1 - alpha);
            if (test_T < 0.0001f) {
                done = true;
                continue;
            }

            const float4 q3 = staged[j * 4 + 3];      // g b depth -
            const float mapped_max_t = (float)((F3DG_FAR_PLANE * t - F3DG_FAR_PLANE * F3DG_NEAR_PLANE) / ((F3DG_FAR_PLANE - F3DG_NEAR_PLANE) * t));

            const float length = (float)sqrt(n0 * n0 + n1 * n1 + n2 * n2 + 1e-7);
            const float nn0 = -n0 / length, nn1 = -n1 / length, nn2 = -n2 / length;

            const float A = 1 - Tr;
            const float error = mapped_max_t * mapped_max_t * A + dist2 - 2 * mapped_max_t * dist1;
            distortion += error * alpha * Tr;
            dist1 += mapped_max_t * alpha * Tr;
            dist2 += mapped_max_t * mapped_max_t * alpha * Tr;

            C0 += q2.w * alpha * Tr;
            C1 += q3.x * alpha * Tr;
            C2 += q3.y * alpha * Tr;
            C3 += nn0 * alpha * Tr;
            C4 += nn1 * alpha * Tr;
            C5 += nn2 * alpha * Tr;
            if (Tr > 0.5) {
                C6 = t;
                max_contributor = contributor;
            }
            C7 += alpha * Tr;

            Tr = test_T;
            last_contributor = contributor;
        }
    }

    if (inside) {
        const float* bg = background + (bg_per_view ? 3 * view : 0);
        const float distortion_before_normalized = distortion;
        distortion = (float)(distortion / ((1 - Tr) * (1 - Tr) + 1e-7));

        if (SAVE_AUX) {
            float* fT = final_T + (size_t)view * 4 * HW;
            fT[pix_id] = Tr;
            fT[pix_id + HW] = dist1;
            fT[pix_id + 2 * HW] = dist2;
            fT[pix_id + 3 * HW] = distortion_before_normalized;
            unsigned* nc = n_contrib + (size_t)view * 2 * HW;
            nc[pix_id] = last_contributor;
            nc[pix_id + HW] = max_contributor;
        }
        float* out = out_color + (size_t)view * F3DG_OUT_CHANNELS * HW;
        out[0 * HW + pix_id] = C0 + Tr * bg[0];
        out[1 * HW + pix_id] = C1 + Tr * bg[1];
        out[2 * HW + pix_id] = C2 + Tr * bg[2];
        out[3 * HW + pix_id] = C3;
        out[4 * HW + pix_id] = C4;
        out[5 * HW + pix_id] = C5;
        out[6 * HW + pix_id] = C6;
        out[7 * HW + pix_id] = C7;
        out[8 * HW + pix_id] = distortion;
    }
}

} // namespace

int f3dg_launch_render(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y,
                       const F3dgHeader* hdr, const uint2* ranges, const unsigned* point_list, const F3dgRec* rec,
                       const float* background, int bg_per_view, float* out_color, float* final_T,
                       unsigned* n_contrib, int save_aux)
{
    const int tiles_x = (W + F3DG_TILE - 1) / F3DG_TILE, tiles_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = tiles_x * tiles_y;
    const unsigned groups = (unsigned)((V + 7) / 8);
    dim3 grid(groups * 8u * (unsigned)T);
    if (save_aux)
        hipLaunchKernelGGL(render_fwd_kernel<true>, grid, dim3(F3DG_BLOCK), 0, s, V, P, W, H, tiles_x, T, focal_x,
                           focal_y, hdr, ranges, point_list, rec, background, bg_per_view, out_color, final_T, n_contrib);
    else
        hipLaunchKernelGGL(render_fwd_kernel<false>, grid, dim3(F3DG_BLOCK), 0, s, V, P, W, H, tiles_x, T, focal_x,
                           focal_y, hdr, ranges, point_list, rec, background, bg_per_view, out_color, final_T, n_contrib);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}
