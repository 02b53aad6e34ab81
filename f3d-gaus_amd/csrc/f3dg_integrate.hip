// f3dg_integrate.hip -- Gaussians -> points integration (GaussianRasterizer_GOF.integrate) for gfx950.
//
// Replaces FORWARD::preprocess_points / createWithKeys / FORWARD::integrate of the reference
// (RAST/cuda_rasterizer/forward.cu:722-766, 801-1197; rasterizer_impl.cu:113-144, 530-792). The reference runs ONE
// kernel per tile whose threads keep 7 KB of private arrays each (1024 contributor ids, 256 projected points with
// their alphas / transmittances) and sweep the tile's depth-sorted point list in chunks of 256 points per pixel. The
// results only depend on per-pixel and per-point quantities, so the work is reorganised MI355X-first without changing
// a single value:
//
//   pass 1  (one workgroup per tile, one pixel per lane)  the first loop of integrateCUDA: five rays per pixel
//           (centre + 4 corners), colour / alpha / maximal depth of the pixel, and the pixel's list of contributing
//           Gaussians -- written to a global [pixel][1024] u16 table instead of a private array. The culled per-16-lane
//           entry lists and the conservative K pre-test of the compositing kernel are reused (boxes widened by the half
//           pixel of the corner rays); both only remove (ray, Gaussian) pairs the reference `continue`s on.
//   points  (one LANE PER POINT, visited in pixel order)  preprocessPointsCUDA fused with the second loop of integrateCUDA:
//           the point finds its pixel, walks that pixel's contributor list and accumulates its alpha. Loop interchange
//           (points outside, Gaussians inside) is exact because every point's recurrence is independent. The tile's
//           depth-sorted point list of the reference is only needed for one thing, see below.
//   epilogue  the reference stores `total_projected` in the distortion channel. It equals the number of points in the
//           pixel, EXCEPT when some pixel of the tile holds more than 256 points: the block then sweeps again, and
//           every thread that had already finished re-collects the LAST point of the tile's sorted list if it lies in
//           its pixel (point_counter_last = point_counter - 1, forward.cu:1099). That is reproduced arithmetically:
//           total = n + (S - s) * [last tile point in this pixel], s = max(1, ceil(n / 256)), S = max over the tile;
//           the last point of the stable (tile, depth) sort is the maximum of (depth bits, index), kept by atomicMax.
//
// Arithmetic: the reference's float32 / float64 operation order, no contraction (-ffp-contract=off); only expf differs
// (device libm vs glibc), as in the compositing kernel.
#include "f3dg_common.h"
#include "f3dg_ellipse.h"

namespace {

#define F3DG_ROUND (F3DG_BLOCK - 1)            // staged entries per round; LDS slot F3DG_ROUND is the sentinel
#define F3DG_MAX_CONTRIB 1024                  // MAX_NUM_CONTRIBUTORS * 4, auxiliary.h:26 / forward.cu:862
#define F3DG_MAX_PROJECTED 256                 // auxiliary.h:34

struct Pass1State {
    float Ts[5];
    float C0, C1, C2, C6, C7;
    unsigned last_contributor, nloc;
};

// One (ray, Gaussian) evaluation of forward.cu:903-962. Returns true when the ray "used" the Gaussian.
template <bool FILTER, int K>
__device__ __forceinline__ bool ray_entry(Pass1State& st, float rx, float ry, const float4& q0, const float4& q1,
                                          const float4& q2, const float4& q3)
{
    const float n0 = q0.x * rx + q0.y * ry + q0.z;
    const float n1 = q0.y * rx + q0.w * ry + q1.x;
    const float n2 = q0.z * rx + q1.x * ry + q1.y;
    const float AA = rx * n0 + ry * n1 + n2;
    const float bhalf = q1.z * rx + q1.w * ry + q2.x;
    if (FILTER) {
        // Conservative pre-test (f3dg_preprocess.hip pretest_constant): certainly alpha < 1/255. This loop divides
        // BB / AA in float32 (the compositing loop in float64); the extra relative error 2^-24 of b^2/a is inside the
        // 5e-7 margin of K (3.2e-7 used by the product roundings and K's own narrowing).
        if (bhalf * bhalf < q2.w * AA)
            return false;
    }
    const float BB = 2 * bhalf;
    const float CC = q2.y;
    const float q = BB / AA;                            // one float32 division: -BB / (2 * AA) == -0.5f * (BB / AA) exactly
    const float t = -0.5f * q;
    if (t <= F3DG_NEAR_PLANE)
        return false;
    const double min_value = -q * (BB / 4.) + CC;
    float power = (float)(-0.5f * min_value);
    if (power > 0.0f)
        power = 0.0f;
    const float alpha = fminf(0.99f, q2.z * expf(power));
    if (alpha < 1.0f / 255.0f)
        return false;
    const float test_T = st.Ts[K] * (1 - alpha);
    if (test_T < 0.0001f)
        return false;                                   // `continue`, NOT done (forward.cu:934-938)
    if (K == 0) {
        st.C0 += q3.x * alpha * st.Ts[0];
        st.C1 += q3.y * alpha * st.Ts[0];
        st.C2 += q3.z * alpha * st.Ts[0];
    }
    if (t > st.C6)
        st.C6 = t;
    if (K == 0)
        st.C7 += alpha * st.Ts[0];
    st.Ts[K] = test_T;
    return true;
}

// ray_entry split into its stateless part (everything up to alpha: branch-free, so that the compiler can interleave the evaluations of
// TWO entries -- the dependent chain of one is a float32 division, a float64 island and an exponential) and the recurrence. The same
// values and the same decisions: `hit` is the conjunction of the three tests ray_entry leaves on, evaluated on the same operands.
struct RayEval { float alpha, t; bool hit; };
__device__ __forceinline__ RayEval ray_eval(float rx, float ry, const float4& q0, const float4& q1, const float4& q2)
{
    const float n0 = q0.x * rx + q0.y * ry + q0.z;
    const float n1 = q0.y * rx + q0.w * ry + q1.x;
    const float n2 = q0.z * rx + q1.x * ry + q1.y;
    const float AA = rx * n0 + ry * n1 + n2;
    const float bhalf = q1.z * rx + q1.w * ry + q2.x;
    const bool pre = bhalf * bhalf < q2.w * AA;          // certainly alpha < 1/255 (see ray_entry)
    const float BB = 2 * bhalf;
    const float CC = q2.y;
    const float q = BB / AA;
    RayEval r;
    r.t = -0.5f * q;
    const double min_value = -q * (BB / 4.) + CC;
    float power = (float)(-0.5f * min_value);
    if (power > 0.0f)
        power = 0.0f;
    r.alpha = fminf(0.99f, q2.z * expf(power));
    r.hit = !pre && !(r.t <= F3DG_NEAR_PLANE) && !(r.alpha < 1.0f / 255.0f);
    return r;
}
template <int K>
__device__ __forceinline__ bool ray_apply(Pass1State& st, const RayEval& e, const float4& q3)
{
    const float test_T = st.Ts[K] * (1 - e.alpha);
    if (!e.hit || test_T < 0.0001f)
        return false;                                   // `continue`, NOT done (forward.cu:934-938)
    if (K == 0) {
        st.C0 += q3.x * e.alpha * st.Ts[0];
        st.C1 += q3.y * e.alpha * st.Ts[0];
        st.C2 += q3.z * e.alpha * st.Ts[0];
    }
    if (e.t > st.C6)
        st.C6 = e.t;
    if (K == 0)
        st.C7 += e.alpha * st.Ts[0];
    st.Ts[K] = test_T;
    return true;
}

#ifdef F3DG_LAB      // ---- round 1's per-pixel pass and its plain variant, lab builds only (the baseline of the bit-identity tests)
template <bool FILTER>
__global__ void __launch_bounds__(F3DG_BLOCK)
integrate_pass1_kernel(int W, int H, int tiles_x, float focal_x, float focal_y, const F3dgHeader* __restrict__ hdr,
                       const uint2* __restrict__ ranges, const unsigned* __restrict__ point_list,
                       const F3dgRec* __restrict__ rec, const float4* __restrict__ bbox,
                       const float* __restrict__ background, float* __restrict__ out_color,
                       float* __restrict__ final_T, unsigned* __restrict__ n_contrib,
                       unsigned short* __restrict__ contrib_ids, unsigned* __restrict__ contrib_n)
{
    const unsigned tile = blockIdx.x;
    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    // lane -> pixel as in the compositing kernel: wave = 8x8 quadrant, 16-lane group = 4x4 block
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned grp = lane >> 4, gi = lane & 15u;
    const unsigned blk_x = (wave & 1u) * 2u + (grp & 1u), blk_y = (wave >> 1) * 2u + (grp >> 1);
    const unsigned lx = blk_x * 4u + (gi & 3u), ly = blk_y * 4u + (gi >> 2);
    const unsigned pix_x = tile_x * F3DG_TILE + lx, pix_y = tile_y * F3DG_TILE + ly;
    const bool inside = pix_x < (unsigned)W && pix_y < (unsigned)H;
    const size_t HW = (size_t)H * W;
    const size_t pix_id = (size_t)W * pix_y + pix_x;
    const float pixf_x = (float)pix_x + 0.5f, pixf_y = (float)pix_y + 0.5f;
    // the five rays of forward.cu:864-866, 905: offsets (0,0) (-.5,-.5) (.5,-.5) (-.5,.5) (.5,.5)
    const float rx0 = (float)((pixf_x + 0.0f - W / 2.) / focal_x), ry0 = (float)((pixf_y + 0.0f - H / 2.) / focal_y);
    const float rxm = (float)((pixf_x + -0.5f - W / 2.) / focal_x), rym = (float)((pixf_y + -0.5f - H / 2.) / focal_y);
    const float rxp = (float)((pixf_x + 0.5f - W / 2.) / focal_x), ryp = (float)((pixf_y + 0.5f - H / 2.) / focal_y);

    uint2 range = ranges[tile];
    if (hdr->overflow) range = make_uint2(0, 0);
    const int rounds = (int)((range.y - range.x + F3DG_ROUND - 1) / F3DG_ROUND);
    int toDo = (int)(range.y - range.x);

    __shared__ float4 sq0[F3DG_BLOCK];            // v0 v1 v2 v3
    __shared__ float4 sq1[F3DG_BLOCK];            // v4 v5 v6 v7
    __shared__ float4 sq2[F3DG_BLOCK];            // v8 v9 opac K
    __shared__ float4 sq3[F3DG_BLOCK];            // r g b, and (FILTER) the 16-bit block mask in place of the depth
    __shared__ __align__(16) unsigned char grp_list[FILTER ? F3DG_BLOCK / 64 : 1][FILTER ? 4 : 1][FILTER ? F3DG_BLOCK : 1];
    if (threadIdx.x == F3DG_ROUND) {
        // sentinel (see f3dg_render.hip): fails the pre-test, pads the culled lists; .x/.y of sq3 hold the vote counters
        sq0[F3DG_ROUND] = make_float4(1.0f, 0.0f, 0.0f, 1.0f);
        sq1[F3DG_ROUND] = make_float4(0.0f, 1.0f, 0.0f, 0.0f);
        sq2[F3DG_ROUND] = make_float4(0.0f, 0.0f, 0.0f, __builtin_inff());
        sq3[F3DG_ROUND] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    int* done_cnt = reinterpret_cast<int*>(&sq3[F3DG_ROUND]);
    __syncthreads();

    const float tile_px0 = (float)(tile_x * F3DG_TILE), tile_py0 = (float)(tile_y * F3DG_TILE);

    bool done = !inside;
    Pass1State st;
#pragma unroll
    for (int k = 0; k < 5; k++) st.Ts[k] = 1.0f;
    st.C0 = st.C1 = st.C2 = st.C6 = st.C7 = 0;
    st.last_contributor = 0; st.nloc = 0;
    unsigned short* my_ids = contrib_ids + pix_id * F3DG_MAX_CONTRIB;

    for (int i = 0; i < rounds; i++, toDo -= F3DG_ROUND) {
        {
            const unsigned long long dl = __ballot(done);
            if (lane == 0)
                atomicAdd(&done_cnt[i & 1], __popcll(dl));
        }
        __syncthreads();
        const int num_done = done_cnt[i & 1];
        if (threadIdx.x == 0)
            done_cnt[(i + 1) & 1] = 0;
        if (num_done == F3DG_BLOCK)
            break;

        const unsigned progress = (unsigned)i * F3DG_ROUND + threadIdx.x;
        if (threadIdx.x < F3DG_ROUND && range.x + progress < range.y) {
            const unsigned id = point_list[range.x + progress] & F3DG_ID_MASK;
            const float4* src = reinterpret_cast<const float4*>(rec + id);
            const float4 a = src[0], b = src[1], c = src[2];
            float4 d = src[3];
            sq0[threadIdx.x] = a;
            sq1[threadIdx.x] = b;
            sq2[threadIdx.x] = c;
            if (FILTER) {
                const float4 bx = bbox[id];                       // pixel-index coordinates of the alpha >= 1/255 region
                unsigned mx = 0, my = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {                     // corner rays reach half a pixel beyond the block
                    if (bx.x <= tile_px0 + (float)(4 * q + 3) + 0.5f && bx.y >= tile_px0 + (float)(4 * q) - 0.5f) mx |= 1u << q;
                    if (bx.z <= tile_py0 + (float)(4 * q + 3) + 0.5f && bx.w >= tile_py0 + (float)(4 * q) - 0.5f) my |= 1u << q;
                }
                const unsigned m = ((my & 1u) ? mx : 0u) | ((my & 2u) ? mx << 4 : 0u) | ((my & 4u) ? mx << 8 : 0u) |
                                   ((my & 8u) ? mx << 12 : 0u);
                d.w = __uint_as_float(m);
            }
            sq3[threadIdx.x] = d;
        } else if (FILTER && threadIdx.x < F3DG_ROUND) {
            sq3[threadIdx.x].w = 0.0f;
        }
        __syncthreads();

        const int n = min(F3DG_ROUND, toDo);
        int count = n;
        if (FILTER) {
            {
                uint4* fill = reinterpret_cast<uint4*>(&grp_list[wave][0][0]);
                const unsigned ss = 0x01010101u * F3DG_ROUND;
                fill[lane] = make_uint4(ss, ss, ss, ss);
            }
            int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            const unsigned qx2 = (wave & 1u) * 2u, qy2 = (wave >> 1) * 2u;
            const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
            for (int c = 0; c < F3DG_BLOCK / 64; c++) {
                const unsigned e = c * 64 + lane;
                const unsigned m = __float_as_uint(sq3[e].w);
                const bool b0 = (m >> ((qy2 + 0u) * 4u + qx2 + 0u)) & 1u, b1 = (m >> ((qy2 + 0u) * 4u + qx2 + 1u)) & 1u;
                const bool b2 = (m >> ((qy2 + 1u) * 4u + qx2 + 0u)) & 1u, b3 = (m >> ((qy2 + 1u) * 4u + qx2 + 1u)) & 1u;
                const unsigned long long l0 = __ballot(b0), l1 = __ballot(b1), l2 = __ballot(b2), l3 = __ballot(b3);
                if (b0) grp_list[wave][0][c0 + __popcll(l0 & lt)] = (unsigned char)e;
                if (b1) grp_list[wave][1][c1 + __popcll(l1 & lt)] = (unsigned char)e;
                if (b2) grp_list[wave][2][c2 + __popcll(l2 & lt)] = (unsigned char)e;
                if (b3) grp_list[wave][3][c3 + __popcll(l3 & lt)] = (unsigned char)e;
                c0 += __popcll(l0); c1 += __popcll(l1); c2 += __popcll(l2); c3 += __popcll(l3);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            count = max(max(c0, c1), max(c2, c3));
        }
        const unsigned char* my_list = FILTER ? grp_list[wave][grp] : nullptr;
        const unsigned round_base = (unsigned)i * F3DG_ROUND;

        for (int kk = 0; !done && kk < count; kk++) {
            const int j = FILTER ? (int)my_list[kk] : kk;
            const float4 q0 = sq0[j], q1 = sq1[j], q2 = sq2[j], q3 = sq3[j];
            bool used = false;
            used |= ray_entry<FILTER, 0>(st, rx0, ry0, q0, q1, q2, q3);
            used |= ray_entry<FILTER, 1>(st, rxm, rym, q0, q1, q2, q3);
            used |= ray_entry<FILTER, 2>(st, rxp, rym, q0, q1, q2, q3);
            used |= ray_entry<FILTER, 3>(st, rxm, ryp, q0, q1, q2, q3);
            used |= ray_entry<FILTER, 4>(st, rxp, ryp, q0, q1, q2, q3);
            if (used) {
                const unsigned contributor = round_base + (unsigned)j + 1u;
                st.last_contributor = contributor;
                my_ids[st.nloc] = (unsigned short)contributor;     // (u_int16_t) cast of forward.cu:969
                st.nloc += 1;
                if (st.nloc >= F3DG_MAX_CONTRIB)
                    done = true;                                    // "Maximal contributors are met", forward.cu:972-976
            }
        }
    }

    if (inside) {                                                  // forward.cu:984-996
        final_T[pix_id] = st.Ts[0];
        n_contrib[pix_id] = st.last_contributor;
        contrib_n[pix_id] = st.nloc;
        out_color[0 * HW + pix_id] = st.C0 + st.Ts[0] * background[0];
        out_color[1 * HW + pix_id] = st.C1 + st.Ts[0] * background[1];
        out_color[2 * HW + pix_id] = st.C2 + st.Ts[0] * background[2];
        out_color[3 * HW + pix_id] = 0.0f;                         // the caller's zero fill, rasterize_points.cu:273
        out_color[4 * HW + pix_id] = 0.0f;
        out_color[5 * HW + pix_id] = 0.0f;
        out_color[6 * HW + pix_id] = st.C6;
        out_color[7 * HW + pix_id] = st.C7;
    }
}

#endif // F3DG_LAB (integrate_pass1_kernel)

// The same pass with the culling machinery of the compositing forward (f3dg_render.hip: render2): the per-ray K pre-test above costs
// ~25 instructions per (ray, Gaussian) = 125 per (pixel, list entry), and no pixel ever leaves the loop early here (a saturated ray
// `continue`s), so the pass was 60-90 % of an integrate call. A list entry is instead tested ONCE per pixel against the record's
// conservative alpha >= 1/255 ellipse, Gaussians across the lanes (two FMAs per pixel, one comparison = one wave ballot), and only
// the passing (pixel, entry) pairs go through the five rays. The corner rays leave the pixel centre by (+-0.5, +-0.5) px: the ellipse
// is scaled uniformly about its centre by 1 + 0.7072 px / semi-minor axis, which contains its Minkowski sum with that square
// (a >= b: (a + r, b + r) fits inside (a, b) (b + r) / b). As everywhere, this only removes pairs the reference `continue`s on.
__global__ void __launch_bounds__(F3DG_BLOCK)
integrate_pass1_cull_kernel(int V, int P, int T, int W, int H, int tiles_x, float focal_x, float focal_y, const F3dgHeader* __restrict__ hdr,
                            const uint2* __restrict__ ranges, const unsigned* __restrict__ point_list,
                            const F3dgRec* __restrict__ rec, const float4* __restrict__ cull,
                            const float* __restrict__ background, float* __restrict__ out_color,
                            float* __restrict__ final_T, unsigned* __restrict__ n_contrib,
                            unsigned short* __restrict__ contrib_ids, unsigned* __restrict__ contrib_n,
                            const unsigned* __restrict__ only /* null, or [V*T] flags: the tiles to compute */)
{
    constexpr int ROUND = F3DG_BLOCK;             // staged entries per round (byte indices 0..255)
    if (only != nullptr) {                        // the tiles integrate_pass1_rays_kernel handed back (a pixel reached 1,024 contributors)
        unsigned view_, tile_;
        f3dg_xcd_map(blockIdx.x, (unsigned)V, (unsigned)T, view_, tile_);
        if (only[(size_t)view_ * T + tile_] == 0u) return;
    }
    // several cameras of the same Gaussians in one launch (f3dg_integrate_prepare_batched): one 256^2 camera is 256 workgroups,
    // a wave per SIMD and nothing to overlap its latencies with; every per-camera array is the camera's slice of a [V, ...] array
    unsigned view, tile;
    f3dg_xcd_map(blockIdx.x, (unsigned)V, (unsigned)T, view, tile);
    {
        const size_t HWv = (size_t)H * W;
        ranges += (size_t)view * T; rec += (size_t)view * P; cull += (size_t)view * P;
        out_color += (size_t)view * F3DG_OUT_CHANNELS * HWv; final_T += (size_t)view * 4 * HWv; n_contrib += (size_t)view * 2 * HWv;
        contrib_ids += (size_t)view * HWv * F3DG_MAX_CONTRIB; contrib_n += (size_t)view * HWv;
    }
    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned grp = lane >> 4, gi = lane & 15u;
    const unsigned blk_x = (wave & 1u) * 2u + (grp & 1u), blk_y = (wave >> 1) * 2u + (grp >> 1);
    const unsigned lx = blk_x * 4u + (gi & 3u), ly = blk_y * 4u + (gi >> 2);
    const unsigned pix_x = tile_x * F3DG_TILE + lx, pix_y = tile_y * F3DG_TILE + ly;
    const bool inside = pix_x < (unsigned)W && pix_y < (unsigned)H;
    const size_t HW = (size_t)H * W;
    const size_t pix_id = (size_t)W * pix_y + pix_x;
    const float pixf_x = (float)pix_x + 0.5f, pixf_y = (float)pix_y + 0.5f;
    const float rx0 = (float)((pixf_x + 0.0f - W / 2.) / focal_x), ry0 = (float)((pixf_y + 0.0f - H / 2.) / focal_y);
    const float rxm = (float)((pixf_x + -0.5f - W / 2.) / focal_x), rym = (float)((pixf_y + -0.5f - H / 2.) / focal_y);
    const float rxp = (float)((pixf_x + 0.5f - W / 2.) / focal_x), ryp = (float)((pixf_y + 0.5f - H / 2.) / focal_y);
    const float blk_px0 = (float)(tile_x * F3DG_TILE + blk_x * 4u), blk_py0 = (float)(tile_y * F3DG_TILE + blk_y * 4u);
    const float tile_px0 = (float)(tile_x * F3DG_TILE), tile_py0 = (float)(tile_y * F3DG_TILE);

    uint2 range = ranges[tile];
    if (hdr->overflow) range = make_uint2(0, 0);
    const int rounds = (int)((range.y - range.x + ROUND - 1) / ROUND);

    __shared__ float4 sq0[ROUND], sq1[ROUND], sq2[ROUND], sq3[ROUND];      // v0..v3 | v4..v7 | v8 v9 opac K | r g b -
    __shared__ float4 sE[ROUND];                  // inflated ellipse: cx cy a b
    __shared__ float sF[ROUND];                   //                    c
    __shared__ unsigned short sM[ROUND];          // which of the tile's 16 4x4 blocks its box touches
    __shared__ __align__(16) unsigned char lists[F3DG_BLOCK / 64][4][ROUND];
    __shared__ int done_cnt[2];
    if (threadIdx.x < 2) done_cnt[threadIdx.x] = 0;
    __syncthreads();

    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned pull = (gi + 16u * (grp >> 1)) * 4u;       // ds_bpermute source of this pixel's ballot half
    const unsigned pull_shift = 16u * (grp & 1u);
    const unsigned char* my_list = lists[wave][grp];

    bool done = !inside;
    Pass1State st;
#pragma unroll
    for (int k = 0; k < 5; k++) st.Ts[k] = 1.0f;
    st.C0 = st.C1 = st.C2 = st.C6 = st.C7 = 0;
    st.last_contributor = 0; st.nloc = 0;
    unsigned short* my_ids = contrib_ids + pix_id * F3DG_MAX_CONTRIB;

    for (int i = 0; i < rounds; i++) {
        const unsigned long long alive = __ballot(!done);
        if (lane == 0)
            atomicAdd(&done_cnt[i & 1], 64 - __popcll(alive));
        __syncthreads();
        const int num_done = done_cnt[i & 1];
        if (threadIdx.x == 0)
            done_cnt[(i + 1) & 1] = 0;
        if (num_done == F3DG_BLOCK)
            break;

        const unsigned progress = (unsigned)i * ROUND + threadIdx.x;
        unsigned short m16 = 0;
        if (range.x + progress < range.y) {
            const unsigned id = point_list[range.x + progress] & F3DG_ID_MASK;
            const float4* src = reinterpret_cast<const float4*>(rec + id);
            const float4 a = src[0], b = src[1], c = src[2], d = src[3];
            float4 e = cull[id];
            float ec = d.w;
            // scale about the centre by s = 1 + 0.7072 sqrt(lmax) (lmax = 1 / semi-minor axis^2), rounded outwards; "everything"
            // records (a = b = c = 0) stay what they are, an overflowing lmax ends as a = b = c = 0 = "everything"
            const float hd = 0.5f * (e.z - ec);
            const float lmax = 0.5f * (e.z + ec) + sqrtf(hd * hd + 0.25f * e.w * e.w);
            const float sc = 1.0f + 0.70715f * sqrtf(lmax) * 1.0001f;
            const float inv = (1.0f / (sc * sc)) * 0.99999f;
            e.z *= inv; e.w *= inv; ec *= inv;
            if (!(e.z == e.z) || !(e.w == e.w) || !(ec == ec)) { e.z = 0.0f; e.w = 0.0f; ec = 0.0f; }
            sq0[threadIdx.x] = a; sq1[threadIdx.x] = b; sq2[threadIdx.x] = c; sq3[threadIdx.x] = d;
            sE[threadIdx.x] = e;
            sF[threadIdx.x] = ec;
            m16 = (unsigned short)ellipse_block_mask(e, ec, tile_px0, tile_py0);
        }
        sM[threadIdx.x] = m16;
        __syncthreads();

        // four compacted lists per wave, one per 16-lane group (4x4 block), in list order
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        {
            const unsigned qx2 = (wave & 1u) * 2u, qy2 = (wave >> 1) * 2u;
            const bool g0 = (alive & 0xFFFFull) != 0, g1 = (alive & 0xFFFF0000ull) != 0, g2 = (alive & 0xFFFF00000000ull) != 0,
                       g3 = (alive >> 48) != 0;
#pragma unroll
            for (int c = 0; c < ROUND / 64; c++) {
                const unsigned e = c * 64 + lane;
                const unsigned m = sM[e];
                const bool b0 = g0 && ((m >> ((qy2 + 0u) * 4u + qx2 + 0u)) & 1u), b1 = g1 && ((m >> ((qy2 + 0u) * 4u + qx2 + 1u)) & 1u);
                const bool b2 = g2 && ((m >> ((qy2 + 1u) * 4u + qx2 + 0u)) & 1u), b3 = g3 && ((m >> ((qy2 + 1u) * 4u + qx2 + 1u)) & 1u);
                const unsigned long long l0 = __ballot(b0), l1 = __ballot(b1), l2 = __ballot(b2), l3 = __ballot(b3);
                if (b0) lists[wave][0][c0 + __popcll(l0 & lt)] = (unsigned char)e;
                if (b1) lists[wave][1][c1 + __popcll(l1 & lt)] = (unsigned char)e;
                if (b2) lists[wave][2][c2 + __popcll(l2 & lt)] = (unsigned char)e;
                if (b3) lists[wave][3][c3 + __popcll(l3 & lt)] = (unsigned char)e;
                c0 += __popcll(l0); c1 += __popcll(l1); c2 += __popcll(l2); c3 += __popcll(l3);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const int count = max(max(c0, c1), max(c2, c3));
        const int my_len = grp == 0 ? c0 : grp == 1 ? c1 : grp == 2 ? c2 : c3;
        const unsigned round_base = (unsigned)i * ROUND;

        for (int w0 = 0; w0 < count; w0 += 64) {
            // ---- phase 1: lane (g, e) tests entry w0 + 16 sub + e of group g's list against the 16 pixels of g's block
            unsigned pass_lo = 0, pass_hi = 0;
#pragma unroll 1
            for (int sub = 0; sub < 4; sub++) {
                const int base = w0 + 16 * sub;
                if (base >= count)
                    break;
                const int pos = base + (int)gi;
                const int j = (int)my_list[pos];
                const float4 e = sE[j];
                const float cc = sF[j];
                const float u0 = pos < my_len ? blk_px0 - e.x : __builtin_nanf("");     // NaN: every comparison below is false
                const float v0 = blk_py0 - e.y;
                float dxx[4], adx[4], dyy[4], cdy[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    dxx[q] = u0 + (float)q;
                    adx[q] = e.z * dxx[q];
                    dyy[q] = v0 + (float)q;
                    cdy[q] = cc * dyy[q] * dyy[q];
                }
                int stage = 0;
                ellipse_ballots<0>(stage, fmaf(dxx[0], fmaf(e.w, dyy[0], adx[0]), cdy[0]), dxx, adx, dyy, cdy, e.w);
                const unsigned piece = ((unsigned)__builtin_amdgcn_ds_bpermute((int)pull, stage) >> pull_shift) & 0xFFFFu;
                if (sub & 2) pass_hi |= piece << (16 * (sub & 1));
                else pass_lo |= piece << (16 * (sub & 1));
            }
            unsigned long long pass = done ? 0ull : ((unsigned long long)pass_hi << 32) | pass_lo;

            // ---- phase 2: the five rays of this pixel through its own passing entries, in list order
            while (pass != 0 && !done) {
                const int kk = __builtin_ctzll(pass);
                pass &= pass - 1;
                const int j = (int)my_list[w0 + kk];
                const float4 q0 = sq0[j], q1 = sq1[j], q2 = sq2[j], q3 = sq3[j];
                bool used = false;
                used |= ray_entry<true, 0>(st, rx0, ry0, q0, q1, q2, q3);
                used |= ray_entry<true, 1>(st, rxm, rym, q0, q1, q2, q3);
                used |= ray_entry<true, 2>(st, rxp, rym, q0, q1, q2, q3);
                used |= ray_entry<true, 3>(st, rxm, ryp, q0, q1, q2, q3);
                used |= ray_entry<true, 4>(st, rxp, ryp, q0, q1, q2, q3);
                if (used) {
                    const unsigned contributor = round_base + (unsigned)j + 1u;
                    st.last_contributor = contributor;
                    my_ids[st.nloc] = (unsigned short)contributor;     // (u_int16_t) cast of forward.cu:969
                    st.nloc += 1;
                    if (st.nloc >= F3DG_MAX_CONTRIB)
                        done = true;                                    // "Maximal contributors are met", forward.cu:972-976
                }
            }
        }
    }

    if (inside) {                                                  // forward.cu:984-996
        final_T[pix_id] = st.Ts[0];
        n_contrib[pix_id] = st.last_contributor;
        contrib_n[pix_id] = st.nloc;
        out_color[0 * HW + pix_id] = st.C0 + st.Ts[0] * background[0];
        out_color[1 * HW + pix_id] = st.C1 + st.Ts[0] * background[1];
        out_color[2 * HW + pix_id] = st.C2 + st.Ts[0] * background[2];
        out_color[3 * HW + pix_id] = 0.0f;                         // the caller's zero fill, rasterize_points.cu:273
        out_color[4 * HW + pix_id] = 0.0f;
        out_color[5 * HW + pix_id] = 0.0f;
        out_color[6 * HW + pix_id] = st.C6;
        out_color[7 * HW + pix_id] = st.C7;
    }
}

// ---- pass 1 by RAYS ----------------------------------------------------------------------------------------------------------------
// The five rays of a pixel are its centre and its four corners (forward.cu:880-901), and a corner belongs to four pixels: the ray
// through (x - 0.5, y - 0.5) is ray 1 of pixel (x, y), ray 2 of (x - 1, y), ray 3 of (x, y - 1) and ray 4 of (x - 1, y - 1) -- the same
// float32 direction to the bit ((x + 0.5f) - 0.5f and (x - 1 + 0.5f) + 0.5f are both exactly x) walking the same tile list, hence the
// same alphas, the same transmittance recurrence T[K] and the same "used" decisions in each of the four. The reference evaluates it
// four times; here a 16x16 tile is 256 centre rays + 17x17 corner rays = 545 rays instead of 1,280, in the reference's arithmetic.
// What is per PIXEL is cheap and derived afterwards: the contributor list = the entries any of its five rays used (in list order),
// the last contributor, max depth = the maximum over its rays; colour, alpha and final T come from the centre ray alone.
//
// One 512-thread workgroup per (camera, tile): waves 0-3 hold the centre rays of the four 8x8 quadrants (lane = pixel), waves 4-7 the
// corners (cx, cy) in 0..15 x 0..15 of the 17x17 grid; the 33 corners of its last column and row are a second, short pass of ONE of
// the eight waves (which one rotates with the tile, so that no SIMD of a CU collects them all). A round stages 256 list entries
// (waves 4-7, while waves 0-3 still merge the previous round) with a 5-bit mask per entry: which quadrants' ray positions / the edge
// the box of its conservative ellipse reaches. Every wave compacts the round to the entries of its region and takes them in 64-entry
// windows: phase 1 with the Gaussians across the lanes (the record's ellipse at the wave's 64 ray positions -- corners are half-
// integer positions of the same grid, no inflation as in integrate_pass1_cull_kernel --, quad_ballots), phase 2 with the rays
// across the lanes, each through its own passing entries; a used entry sets its bit in the ray's 256-bit mask of the round (LDS),
// and after the round's second barrier the pixel lanes OR their five rays' masks and append the contributors.
// A ray is FINISHED once fl(T fl(1 - 1/255)) < 0.0001: every later entry either has alpha < 1/255 or fails `test_T < 0.0001`
// (forward.cu:934-938; rounding is monotone and 1 - alpha <= 1 - 1/255), both a bare `continue` before anything is written -- the
// reference walks on through the rest of the list for nothing. Finished rays leave phase 2, a tile whose rays are all finished
// stops staging: the saturation exit of the compositing kernel, which integrate_pass1_cull_kernel does not have.
// The one thing a shared ray cannot reproduce is a pixel that stops at 1,024 contributors (forward.cu:972-976: its rays end there
// while its neighbours' go on): such a tile raises its `redo` flag and integrate_pass1_cull_kernel, launched behind this kernel on
// the flagged tiles only, computes it pixel by pixel.
#define F3DG_RAYS_THREADS 512
__global__ void __launch_bounds__(F3DG_RAYS_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
integrate_pass1_rays_kernel(int V, int P, int T, int W, int H, int tiles_x, float focal_x, float focal_y, const F3dgHeader* __restrict__ hdr,
                            const uint2* __restrict__ ranges, const unsigned* __restrict__ point_list,
                            const F3dgRec* __restrict__ rec, const float4* __restrict__ cull,
                            const float* __restrict__ background, float* __restrict__ out_color,
                            float* __restrict__ final_T, unsigned* __restrict__ n_contrib,
                            unsigned short* __restrict__ contrib_ids, unsigned* __restrict__ contrib_n, unsigned* __restrict__ redo)
{
    constexpr int ROUND = 256;
    unsigned view, tile;
    f3dg_xcd_map(blockIdx.x, (unsigned)V, (unsigned)T, view, tile);
    const size_t HW = (size_t)H * W;
    ranges += (size_t)view * T; rec += (size_t)view * P; cull += (size_t)view * P;
    out_color += (size_t)view * F3DG_OUT_CHANNELS * HW; final_T += (size_t)view * 4 * HW; n_contrib += (size_t)view * 2 * HW;
    contrib_ids += (size_t)view * HW * F3DG_MAX_CONTRIB; contrib_n += (size_t)view * HW;
    redo += (size_t)view * T;

    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned px0 = tile_x * F3DG_TILE, py0 = tile_y * F3DG_TILE;
    const unsigned quad = wave & 3u;
    const bool edge_wave = wave == ((tile + view) & 7u);     // this wave also walks the 33 rays of the corner grid's last column and row

    // this lane's ray: grid position (gx, gy) inside the tile, centre (waves 0-3: 0..15) or corner (0..16, half a pixel up and left)
    const bool centre = wave < 4u;
    const unsigned gx = (quad & 1u) * 8u + (lane & 7u), gy = (quad >> 1) * 8u + (lane >> 3);
    // a corner is walked when the pixel up and left of it (or, on the tile's first column / row, the pixel it is the corner of) is in the image
    const unsigned ax = px0 + (centre || gx == 0u ? gx : gx - 1u), ay = py0 + (centre || gy == 0u ? gy : gy - 1u);
    const bool active = ax < (unsigned)W && ay < (unsigned)H;
    const float off = centre ? 0.0f : -0.5f;
    const float ray_x = (float)(((float)(px0 + gx) + 0.5f + off - W / 2.) / focal_x), ray_y = (float)(((float)(py0 + gy) + 0.5f + off - H / 2.) / focal_y);
    // origin of the wave's 8x8 grid in the ellipse records' pixel-index coordinates
    const float org_x = (float)(px0 + (quad & 1u) * 8u) + off, org_y = (float)(py0 + (quad >> 1) * 8u) + off;
    // the edge ray of lanes 0..32 of the edge wave: corners (16, 0..16), then (0..15, 16)
    const unsigned egx = lane < 17u ? 16u : lane - 17u, egy = lane < 17u ? lane : 16u;
    const bool eactive = edge_wave && lane < 33u && (px0 + (egx == 0u ? 0u : egx - 1u)) < (unsigned)W && (py0 + (egy == 0u ? 0u : egy - 1u)) < (unsigned)H;
    const float eray_x = (float)(((float)(px0 + egx) + 0.5f + -0.5f - W / 2.) / focal_x), eray_y = (float)(((float)(py0 + egy) + 0.5f + -0.5f - H / 2.) / focal_y);

    uint2 range = ranges[tile];
    if (hdr->overflow) range = make_uint2(0, 0);
    const unsigned n = range.y - range.x;
    const int rounds = (int)((n + ROUND - 1) / ROUND);

    __shared__ float4 sq0[ROUND], sq1[ROUND], sq2[ROUND], sq3[ROUND];      // v0..v3 | v4..v7 | v8 v9 opac K | r g b c
    __shared__ float4 sE[ROUND];                                           // conservative ellipse: cx cy a b (c = sq3.w)
    __shared__ unsigned char sM[ROUND];                                    // bit q: its box reaches quadrant q's ray positions; bit 4: the edge
    __shared__ __align__(16) unsigned char lists[9][ROUND];                // per wave (8: the edge pass): the round's entries of its region
    __shared__ unsigned long long sUsedC[4][256];                          // [64 entries of the round][centre ray]: the entries the ray used
    __shared__ unsigned long long sUsedK[4][289];                          // [..][corner ray]
    __shared__ float sMaxK[289];
    __shared__ int s_over, s_alive[2];
    if (threadIdx.x == 0) { s_over = 0; s_alive[0] = 0; s_alive[1] = 0; }
    __syncthreads();
    const float kFinished = 1.0f - 1.0f / 255.0f;       // fl(1 - alpha) of the smallest alpha that is not skipped
    bool dead = !active, edead = !eactive;

    Pass1State st, se;                                                     // se: the edge ray (Ts[1], C6 only)
#pragma unroll
    for (int k = 0; k < 5; k++) { st.Ts[k] = 1.0f; se.Ts[k] = 1.0f; }
    st.C0 = st.C1 = st.C2 = st.C6 = st.C7 = 0;
    se.C0 = se.C1 = se.C2 = se.C6 = se.C7 = 0;
    st.last_contributor = 0; st.nloc = 0;
    // (pixel lanes = the centre-ray lanes)
    const bool inside = centre && active;
    const size_t pix_id = (size_t)W * (py0 + gy) + (px0 + gx);
    unsigned short* my_ids = contrib_ids + pix_id * F3DG_MAX_CONTRIB;
    const unsigned kidx = gy * 17u + gx, cidx = gy * 16u + gx, eidx = egy * 17u + egx;
    const unsigned long long lt = (1ull << lane) - 1ull;

    auto merge = [&](int i) {          // pixel lanes: contributors of round i = the entries any of the five rays used, in list order
        const unsigned in_round = min((unsigned)ROUND, n - (unsigned)i * ROUND);
        for (unsigned w = 0; w * 64u < in_round; w++) {
            unsigned long long m = sUsedC[w][cidx] | sUsedK[w][kidx] | sUsedK[w][kidx + 1u] | sUsedK[w][kidx + 17u] | sUsedK[w][kidx + 18u];
            if (!inside || st.nloc >= F3DG_MAX_CONTRIB) m = 0ull;
            while (m != 0ull) {
                const unsigned j = (unsigned)__builtin_ctzll(m);
                m &= m - 1ull;
                const unsigned contributor = (unsigned)i * ROUND + w * 64u + j + 1u;
                st.last_contributor = contributor;
                my_ids[st.nloc] = (unsigned short)contributor;         // (u_int16_t) cast of forward.cu:969
                st.nloc += 1;
                if (st.nloc >= F3DG_MAX_CONTRIB) {                      // "Maximal contributors are met": this tile is redone per pixel
                    s_over = 1;
                    break;
                }
            }
        }
    };

    bool merged_all = false;
    for (int i = 0; i < rounds; i++) {
        {
            const int alive = __popcll(__ballot(!dead)) + __popcll(__ballot(!edead));
            if (lane == 0 && alive) atomicAdd(&s_alive[i & 1], alive);
        }
        if (wave >= 4u) {                                    // stage round i
            const unsigned t = threadIdx.x - 256u, progress = (unsigned)i * ROUND + t;
            unsigned m5 = 0;
            if (progress < n) {
                const unsigned id = point_list[range.x + progress] & F3DG_ID_MASK;
                const float4* src = reinterpret_cast<const float4*>(rec + id);
                const float4 d3 = src[3];
                sq0[t] = src[0]; sq1[t] = src[1]; sq2[t] = src[2]; sq3[t] = d3;
                const float4 e = cull[id];
                sE[t] = e;
                // box of the ellipse (as ellipse_block_mask) against the ray positions: quadrant q's are q8 - 0.5 .. q8 + 7, the edge's 15.5
                const float det = fmaf(e.z, d3.w, -0.25f * e.w * e.w);
                m5 = 31u;
                if (det > 0.0f) {
                    const float hx = sqrtf(d3.w / det) * 1.0005f + 2e-3f, hy = sqrtf(e.z / det) * 1.0005f + 2e-3f;
                    const float x0 = e.x - hx - (float)px0, x1 = e.x + hx - (float)px0, y0 = e.y - hy - (float)py0, y1 = e.y + hy - (float)py0;
                    const unsigned mx = ((x0 <= 7.0f && x1 >= -0.5f) ? 1u : 0u) | ((x0 <= 15.0f && x1 >= 7.5f) ? 2u : 0u);
                    const unsigned my = ((y0 <= 7.0f && y1 >= -0.5f) ? 1u : 0u) | ((y0 <= 15.0f && y1 >= 7.5f) ? 2u : 0u);
                    m5 = ((my & 1u) ? mx : 0u) | ((my & 2u) ? mx << 2 : 0u);
                    if ((x1 >= 15.5f && x0 <= 15.5f && y1 >= -0.5f && y0 <= 15.5f) || (y1 >= 15.5f && y0 <= 15.5f && x1 >= -0.5f && x0 <= 15.5f)) m5 |= 16u;
                }
            }
            sM[t] = (unsigned char)m5;
        } else if (i > 0) {
            merge(i - 1);
        }
        __syncthreads();
        if (s_over)
            break;
        if (s_alive[i & 1] == 0) {                           // every ray of the tile is finished (round i - 1 is merged)
            merged_all = true;
            break;
        }
        if (threadIdx.x == 0) s_alive[(i + 1) & 1] = 0;

        // this wave's entries of the round (and, for the edge wave, the edge's), in list order; the rays' masks of the round start empty
        unsigned cnt = 0, ecnt = 0;
#pragma unroll
        for (int c = 0; c < ROUND / 64; c++) {
            const unsigned e = c * 64 + lane;
            const unsigned m = sM[e];
            const bool b = (m >> quad) & 1u;
            const unsigned long long l = __ballot(b);
            if (b) lists[wave][cnt + (unsigned)__popcll(l & lt)] = (unsigned char)e;
            cnt += (unsigned)__popcll(l);
            if (edge_wave) {
                const bool be = (m >> 4) & 1u;
                const unsigned long long le = __ballot(be);
                if (be) lists[8][ecnt + (unsigned)__popcll(le & lt)] = (unsigned char)e;
                ecnt += (unsigned)__popcll(le);
            }
            if (centre) sUsedC[c][cidx] = 0ull; else sUsedK[c][kidx] = 0ull;
            if (edge_wave && lane < 33u) sUsedK[c][eidx] = 0ull;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        for (unsigned w0 = 0; w0 < cnt; w0 += 64u) {
            if (__ballot(!dead) == 0ull)
                break;
            // ---- phase 1: lane e holds entry w0 + e of the wave's list and tests it at the wave's ray positions
            const bool valid = w0 + lane < cnt;
            const unsigned j1 = lists[wave][(w0 + lane) & 255u];
            const float4 e = sE[j1];
            const float ec = sq3[j1].w;
            int pass_lo = 0, pass_hi = 0;
            {
                const float u0 = valid ? org_x - e.x : __builtin_nanf("");      // NaN: every comparison below is false
                const float v0 = org_y - e.y;
                float dxx[8], adx[8], dyy[8], cdy[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    dxx[q] = u0 + (float)q;
                    adx[q] = e.z * dxx[q];
                    dyy[q] = v0 + (float)q;
                    cdy[q] = ec * dyy[q] * dyy[q];
                }
                quad_ballots<0>(pass_lo, pass_hi, fmaf(dxx[0], fmaf(e.w, dyy[0], adx[0]), cdy[0]), dxx, adx, dyy, cdy, e.w);
            }
            // ---- phase 2: lane = ray, through its own passing entries in list order
            unsigned long long pass = dead ? 0ull : ((unsigned long long)(unsigned)pass_hi << 32) | (unsigned)pass_lo;
            // (two entries per trip: their evaluations are independent and interleave; the recurrence takes them in list order)
            while (pass != 0ull && !dead) {
                const int ja = __builtin_ctzll(pass);
                pass &= pass - 1ull;
                const bool two = pass != 0ull;
                const int jb = two ? __builtin_ctzll(pass) : ja;
                pass &= pass - 1ull;                             // (0 stays 0)
                const unsigned a = lists[wave][w0 + (unsigned)ja], b = lists[wave][w0 + (unsigned)jb];
                const RayEval ea = ray_eval(ray_x, ray_y, sq0[a], sq1[a], sq2[a]);
                const RayEval eb = ray_eval(ray_x, ray_y, sq0[b], sq1[b], sq2[b]);
                if (centre) {
                    if (ray_apply<0>(st, ea, sq3[a])) { sUsedC[a >> 6][cidx] |= 1ull << (a & 63u); dead = st.Ts[0] * kFinished < 0.0001f; }
                    if (two && !dead && ray_apply<0>(st, eb, sq3[b])) { sUsedC[b >> 6][cidx] |= 1ull << (b & 63u); dead = st.Ts[0] * kFinished < 0.0001f; }
                } else {
                    if (ray_apply<1>(st, ea, sq3[a])) { sUsedK[a >> 6][kidx] |= 1ull << (a & 63u); dead = st.Ts[1] * kFinished < 0.0001f; }
                    if (two && !dead && ray_apply<1>(st, eb, sq3[b])) { sUsedK[b >> 6][kidx] |= 1ull << (b & 63u); dead = st.Ts[1] * kFinished < 0.0001f; }
                }
            }
        }
        if (edge_wave) {
            for (unsigned w0 = 0; w0 < ecnt; w0 += 64u) {
                if (__ballot(!edead) == 0ull)
                    break;
                const bool valid = w0 + lane < ecnt;
                const unsigned j1 = lists[8][(w0 + lane) & 255u];
                const float4 e = sE[j1];
                const float ec = sq3[j1].w;
                const float ex = valid ? e.x : __builtin_nanf("");
                int pass_lo = 0, pass_hi = 0;
#pragma unroll
                for (int k = 0; k < 33; k++) {
                    const float lx = (float)px0 + (float)(k < 17 ? 16 : k - 17) - 0.5f, ly = (float)py0 + (float)(k < 17 ? k : 16) - 0.5f;
                    const float dx = lx - ex, dy = ly - e.y;
                    const float E = fmaf(dx, fmaf(e.w, dy, e.z * dx), ec * dy * dy);
                    const unsigned long long b = __ballot(1.0f >= E);
                    // (s_nop: the wait states between the comparison's SGPR write and a VALU read of it, which the compiler cannot
                    // see inside inline assembly)
                    asm volatile("s_nop 1\n\tv_writelane_b32 %[lo], %[bl], %[l]\n\tv_writelane_b32 %[hi], %[bh], %[l]\n\ts_nop 0"
                                 : [lo] "+v"(pass_lo), [hi] "+v"(pass_hi)
                                 : [bl] "s"((unsigned)b), [bh] "s"((unsigned)(b >> 32)), [l] "n"(k));
                }
                unsigned long long pass = edead ? 0ull : ((unsigned long long)(unsigned)pass_hi << 32) | (unsigned)pass_lo;
                while (pass != 0ull && !edead) {
                    const int ja = __builtin_ctzll(pass);
                    pass &= pass - 1ull;
                    const bool two = pass != 0ull;
                    const int jb = two ? __builtin_ctzll(pass) : ja;
                    pass &= pass - 1ull;
                    const unsigned a = lists[8][w0 + (unsigned)ja], b = lists[8][w0 + (unsigned)jb];
                    const RayEval ea = ray_eval(eray_x, eray_y, sq0[a], sq1[a], sq2[a]);
                    const RayEval eb = ray_eval(eray_x, eray_y, sq0[b], sq1[b], sq2[b]);
                    if (ray_apply<1>(se, ea, sq3[a])) { sUsedK[a >> 6][eidx] |= 1ull << (a & 63u); edead = se.Ts[1] * kFinished < 0.0001f; }
                    if (two && !edead && ray_apply<1>(se, eb, sq3[b])) { sUsedK[b >> 6][eidx] |= 1ull << (b & 63u); edead = se.Ts[1] * kFinished < 0.0001f; }
                }
            }
        }
        __syncthreads();
    }
    if (centre && rounds > 0 && !s_over && !merged_all)
        merge(rounds - 1);
    if (!centre)
        sMaxK[kidx] = st.C6;
    if (edge_wave && lane < 33u)
        sMaxK[eidx] = se.C6;
    __syncthreads();
    if (s_over) {
        if (threadIdx.x == 0) redo[tile] = 1u;
        return;
    }
    if (threadIdx.x == 0) redo[tile] = 0u;

    if (inside) {                                                  // forward.cu:984-996
        float c6 = st.C6;
        const float k0 = sMaxK[kidx], k1 = sMaxK[kidx + 1u], k2 = sMaxK[kidx + 17u], k3 = sMaxK[kidx + 18u];
        if (k0 > c6) c6 = k0;
        if (k1 > c6) c6 = k1;
        if (k2 > c6) c6 = k2;
        if (k3 > c6) c6 = k3;
        final_T[pix_id] = st.Ts[0];
        n_contrib[pix_id] = st.last_contributor;
        contrib_n[pix_id] = st.nloc;
        out_color[0 * HW + pix_id] = st.C0 + st.Ts[0] * background[0];
        out_color[1 * HW + pix_id] = st.C1 + st.Ts[0] * background[1];
        out_color[2 * HW + pix_id] = st.C2 + st.Ts[0] * background[2];
        out_color[3 * HW + pix_id] = 0.0f;                         // the caller's zero fill, rasterize_points.cu:273
        out_color[4 * HW + pix_id] = 0.0f;
        out_color[5 * HW + pix_id] = 0.0f;
        out_color[6 * HW + pix_id] = c6;
        out_color[7 * HW + pix_id] = st.C7;
    }
}

// preprocessPointsCUDA (forward.cu:722-766): view-space depth and image position of one point; false = not integrated
__device__ __forceinline__ bool project_point(const float* __restrict__ points3D, unsigned idx, const float* view, int W, int H,
                                              float focal_x, float focal_y, float& ix, float& iy, float& depth)
{
    const float px = points3D[3 * (size_t)idx], py = points3D[3 * (size_t)idx + 1], pz = points3D[3 * (size_t)idx + 2];
    const float vx = view[0] * px + view[4] * py + view[8] * pz + view[12];
    const float vy = view[1] * px + view[5] * py + view[9] * pz + view[13];
    const float vz = view[2] * px + view[6] * py + view[10] * pz + view[14];
    if (vz <= 0.2f)                                                // in_frustum, auxiliary.h:193
        return false;
    ix = (float)(focal_x * vx / (vz + 0.0000001f) + W / 2.);
    iy = (float)(focal_y * vy / (vz + 0.0000001f) + H / 2.);
    if (ix < 0 || ix >= W || iy < 0 || iy >= H)
        return false;
    depth = vz;
    return true;
}

// Points are integrated in PIXEL order, not in the caller's order: the lanes of a wave then share a pixel (or a few), i.e.
// the same contributor list, the same records and the same trip count, instead of 64 unrelated lists (1 M randomly
// ordered points: 2.8 ms of divergent L2 reads). The order inside a pixel is irrelevant (every point's recurrence is
// independent), so this is a counting sort by pixel with atomic ranks:
//   bin:  project the point, rank = atomicAdd(points of its pixel), keep (pixel, rank); also the per-tile maximum of
//         (depth bits, index) the epilogue needs
//   scan: exclusive scan of the per-pixel counts
//   perm: perm[start[pixel] + rank] = point index
__global__ void __launch_bounds__(F3DG_BLOCK)
integrate_points_bin_kernel(int PN, const float* __restrict__ points3D, const float* __restrict__ viewmatrix, int W, int H,
                            int tiles_x, float focal_x, float focal_y, const F3dgHeader* __restrict__ hdr,
                            float* __restrict__ out_alpha_integrated, float* __restrict__ out_color_integrated,
                            unsigned* __restrict__ pix_points, unsigned long long* __restrict__ tile_last,
                            unsigned* __restrict__ pt_pix, unsigned* __restrict__ pt_rank)
{
    const unsigned idx = blockIdx.x * F3DG_BLOCK + threadIdx.x;
    if (idx >= (unsigned)PN)
        return;
    float ix, iy, depth;
    const bool ok = !hdr->overflow && project_point(points3D, idx, viewmatrix, W, H, focal_x, focal_y, ix, iy, depth);
    if (!ok) {                                                      // the caller's fills, rasterize_points.cu:275-276
        if (out_alpha_integrated)
            out_alpha_integrated[idx] = 1.0f;
        if (out_color_integrated) {
            out_color_integrated[3 * (size_t)idx] = 0.0f;
            out_color_integrated[3 * (size_t)idx + 1] = 0.0f;
            out_color_integrated[3 * (size_t)idx + 2] = 0.0f;
        }
        pt_pix[idx] = 0xFFFFFFFFu;
        return;
    }
    // the pixel whose thread collects this point: x in [pix, pix + 1) (forward.cu:1063-1064, evaluated in double on
    // exactly representable bounds) = truncation; its tile is the createWithKeys tile (rasterizer_impl.cu:135-136)
    const unsigned pix_x = (unsigned)(int)ix, pix_y = (unsigned)(int)iy;
    const unsigned pix_id = (unsigned)W * pix_y + pix_x;
    const unsigned tile = (pix_y / F3DG_TILE) * (unsigned)tiles_x + pix_x / F3DG_TILE;
    pt_pix[idx] = pix_id;
    pt_rank[idx] = atomicAdd(&pix_points[pix_id], 1u);
    // (1 M points share 256 tiles: thousands of atomics per address serialise at the L2. The running maximum only grows, so a point
    // below the value it can already see -- a relaxed load that bypasses the L1 -- need not take part; a stale, lower value only
    // costs an atomic that changes nothing)
    const unsigned long long cand = ((unsigned long long)__float_as_uint(depth) << 32) | idx;
    if (cand > __atomic_load_n(&tile_last[tile], __ATOMIC_RELAXED))
        atomicMax(&tile_last[tile], cand);
}

__global__ void __launch_bounds__(F3DG_BLOCK)
integrate_points_perm_kernel(int PN, const unsigned* __restrict__ pt_pix, const unsigned* __restrict__ pt_rank,
                             const unsigned* __restrict__ pix_start, unsigned* __restrict__ perm)
{
    const unsigned idx = blockIdx.x * F3DG_BLOCK + threadIdx.x;
    if (idx >= (unsigned)PN)
        return;
    const unsigned pix = pt_pix[idx];
    if (pix != 0xFFFFFFFFu)
        perm[pix_start[pix] + pt_rank[idx]] = idx;
}

__global__ void __launch_bounds__(F3DG_BLOCK)
integrate_points_kernel(int PN, const float* __restrict__ points3D, const float* __restrict__ viewmatrix, int W, int H,
                        int tiles_x, float focal_x, float focal_y, const F3dgHeader* __restrict__ hdr,
                        const uint2* __restrict__ ranges, const unsigned* __restrict__ point_list,
                        const F3dgRec* __restrict__ rec, const unsigned short* __restrict__ contrib_ids,
                        const unsigned* __restrict__ contrib_n, const unsigned* __restrict__ n_contrib,
                        const float* __restrict__ out_color, float* __restrict__ out_alpha_integrated,
                        float* __restrict__ out_color_integrated, const unsigned* __restrict__ pix_points,
                        const unsigned* __restrict__ pix_start, const unsigned* __restrict__ perm,
                        float* __restrict__ alpha_min)
{
    const unsigned i = blockIdx.x * F3DG_BLOCK + threadIdx.x;
    const size_t HW = (size_t)H * W;
    if (hdr->overflow || i >= pix_start[HW - 1] + pix_points[HW - 1])       // number of points inside the image
        return;
    const unsigned idx = perm[i];
    float ix, iy, depth;
    project_point(points3D, idx, viewmatrix, W, H, focal_x, focal_y, ix, iy, depth);     // true, by construction of perm
    const unsigned pix_x = (unsigned)(int)ix, pix_y = (unsigned)(int)iy;
    const size_t pix_id = (size_t)W * pix_y + pix_x;
    const unsigned tile = (pix_y / F3DG_TILE) * (unsigned)tiles_x + pix_x / F3DG_TILE;

    const float rx = (float)((ix - W / 2.) / focal_x);
    const float ry = (float)((iy - H / 2.) / focal_y);
    const unsigned range_x = ranges[tile].x;
    const unsigned nloc = contrib_n[pix_id];
    const unsigned last_contributor = n_contrib[pix_id];
    const unsigned short* ids = contrib_ids + pix_id * F3DG_MAX_CONTRIB;

    float point_alpha = 0.f, point_T = 1.f;
    // second loop of integrateCUDA (forward.cu:1111-1176) for this point. num_iterated only ever matches the next
    // stored id when that id is larger (the u16 ids wrap beyond 65535 entries exactly as in the reference).
    // (a contributor is three dependent loads -- its list position, the Gaussian id at that position, the record -- and the wave's time
    // was that chain: the loop is software-pipelined, positions three contributors ahead, ids two, records one, so that the three
    // loads of an iteration are independent of each other and of its arithmetic. Positions past the list or past the last contributor
    // read entry 0 instead; they end the loop when their turn comes, exactly as before.)
    unsigned num_iterated = 0;
    auto position = [&](unsigned p) -> unsigned { return p < nloc ? (unsigned)ids[p] : 0u; };
    auto gaussian = [&](unsigned t) -> unsigned { return (t >= 1u && t <= last_contributor) ? point_list[range_x + t - 1u] & F3DG_ID_MASK : 0u; };
    unsigned tA = position(0), tB = position(1), tC = position(2);
    unsigned gB = gaussian(tB);
    float4 q0, q1, q2;
    {
        const float4* src = reinterpret_cast<const float4*>(rec + gaussian(tA));
        q0 = src[0]; q1 = src[1]; q2 = src[2];
    }
    for (unsigned ptr = 0; ptr < nloc; ptr++) {
        const unsigned target = tA;
        if (target <= num_iterated || target > last_contributor)
            break;
        num_iterated = target;
        const float4* nsrc = reinterpret_cast<const float4*>(rec + gB);
        const float4 n0q = nsrc[0], n1q = nsrc[1], n2q = nsrc[2];          // the next contributor's record
        const unsigned gC = gaussian(tC);
        const unsigned tD = position(ptr + 3u);
        const float n0 = q0.x * rx + q0.y * ry + q0.z;
        const float n1 = q0.y * rx + q0.w * ry + q1.x;
        const float n2 = q0.z * rx + q1.x * ry + q1.y;
        const float AA = rx * n0 + ry * n1 + n2;
        const float BB = 2 * (q1.z * rx + q1.w * ry + q2.x);
        const float CC = q2.y;
        float t = -BB / (2 * AA);
        if (t > depth)
            t = depth;
        const float power = -0.5f * (AA * t * t + BB * t + CC);
        const float alpha = fminf(0.99f, q2.z * expf(power));
        if (!(alpha < 1.0f / 255.0f)) {
            const float test_T = point_T * (1 - alpha);
            point_alpha += alpha * point_T;
            point_T = test_T;
        }
        q0 = n0q; q1 = n1q; q2 = n2q;
        tA = tB; tB = tC; tC = tD; gB = gC;
    }
    if (out_alpha_integrated)
        out_alpha_integrated[idx] = point_alpha;
    if (out_color_integrated) {
        out_color_integrated[3 * (size_t)idx] = out_color[0 * HW + pix_id];     // C + T * bg of the pixel (forward.cu:1186)
        out_color_integrated[3 * (size_t)idx + 1] = out_color[1 * HW + pix_id];
        out_color_integrated[3 * (size_t)idx + 2] = out_color[2 * HW + pix_id];
    }
    if (alpha_min) {                 // torch.min(final_alpha, alpha_integrated) of the mesh-extraction sweep (visualize.py:463)
        const float cur = alpha_min[idx];
        alpha_min[idx] = (cur != cur || point_alpha != point_alpha) ? __builtin_nanf("") : fminf(cur, point_alpha);
    }
}

__global__ void __launch_bounds__(F3DG_BLOCK)
integrate_epilogue_kernel(int W, int H, int tiles_x, float focal_x, float focal_y, const float* __restrict__ points3D,
                          const float* __restrict__ viewmatrix, const unsigned* __restrict__ pix_points,
                          const unsigned long long* __restrict__ tile_last, float* __restrict__ out_color)
{
    const unsigned tile = blockIdx.x;
    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    const unsigned pix_x = tile_x * F3DG_TILE + (threadIdx.x & 15u), pix_y = tile_y * F3DG_TILE + (threadIdx.x >> 4);
    const bool inside = pix_x < (unsigned)W && pix_y < (unsigned)H;
    const size_t pix_id = (size_t)W * pix_y + pix_x;
    const unsigned n = inside ? pix_points[pix_id] : 0u;
    const unsigned sweeps = n == 0 ? 1u : (n + F3DG_MAX_PROJECTED - 1) / F3DG_MAX_PROJECTED;

    __shared__ unsigned s_max;
    if (threadIdx.x == 0) s_max = 1u;
    __syncthreads();
    atomicMax(&s_max, sweeps);
    __syncthreads();
    const unsigned S = s_max;

    unsigned extra = 0;
    const unsigned long long last = tile_last[tile];
    if (inside && S > sweeps && last != 0ull) {
        float ix, iy, depth;
        if (project_point(points3D, (unsigned)(last & 0xffffffffull), viewmatrix, W, H, focal_x, focal_y, ix, iy, depth) &&
            (unsigned)(int)ix == pix_x && (unsigned)(int)iy == pix_y)
            extra = S - sweeps;
    }
    if (inside)
        out_color[(size_t)F3DG_DISTORTION_OFFSET * H * W + pix_id] = (float)(int)(n + extra);
}

__global__ void integrate_fill_kernel(size_t HW, size_t PN, float* __restrict__ out_color,
                                      float* __restrict__ out_alpha_integrated, float* __restrict__ out_color_integrated)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 9 * HW; i += stride) out_color[i] = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < PN; i += stride) out_alpha_integrated[i] = 1.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * PN; i += stride) out_color_integrated[i] = 0.0f;
}

} // namespace

F3dgIntegLayout f3dg_integ_layout(int P, int PN, int W, int H, long long cap, int V)
{
    F3dgIntegLayout I;
    const F3dgLayout L = f3dg_layout(P, W, H, V, cap);
    const size_t HW = (size_t)W * H;
    const size_t T = (size_t)((W + F3DG_TILE - 1) / F3DG_TILE) * ((H + F3DG_TILE - 1) / F3DG_TILE);
    size_t off = (L.total + 255) & ~(size_t)255;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    I.pix_points = take(HW * sizeof(unsigned));                       // pix_points and tile_last are cleared together
    I.tile_last = take(T * sizeof(unsigned long long));
    I.clear_bytes = off - I.pix_points;
    I.contrib_n = take((size_t)V * HW * sizeof(unsigned));                            // per camera
    I.contrib_ids = take((size_t)V * HW * F3DG_MAX_CONTRIB * sizeof(unsigned short));   // per camera
    const size_t PNn = (size_t)(PN > 0 ? PN : 1);
    I.pix_start = take(HW * sizeof(unsigned));
    I.scan_tmp_elems = (unsigned)((HW + F3DG_SCAN_CHUNK - 1) / F3DG_SCAN_CHUNK + 1);
    I.scan_tmp = take((size_t)I.scan_tmp_elems * sizeof(unsigned));
    I.pt_pix = take(PNn * sizeof(unsigned));
    I.pt_rank = take(PNn * sizeof(unsigned));
    I.perm = take(PNn * sizeof(unsigned));
    I.redo = take((size_t)V * T * sizeof(unsigned));
    I.total = off;
    return I;
}

int f3dg_launch_integrate_fill(hipStream_t s, int W, int H, int PN, float* out_color, float* out_alpha_integrated,
                               float* out_color_integrated)
{
    F3DG_KLAUNCH(integrate_fill_kernel, dim3(1024), dim3(256), 0, s, (size_t)W * H, (size_t)PN, out_color,
                       out_alpha_integrated, out_color_integrated);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

// diagnostic (tools): resident workgroups per CU of the two pass-1 kernels as the runtime computes it
extern "C" int f3dg_debug_pass1_occupancy(int* rays_blocks, int* cull_blocks)
{
    F3DG_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(rays_blocks, integrate_pass1_rays_kernel, F3DG_RAYS_THREADS, 0));
    F3DG_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(cull_blocks, integrate_pass1_cull_kernel, F3DG_BLOCK, 0));
    return F3DG_OK;
}

// pass 1: depends on the Gaussians and the camera only (not on the points), so its result -- colours, last contributors and
// the per-pixel contributor table inside the workspace -- can be kept and reused for any number of point sets
int f3dg_launch_integrate_pass1(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y, const F3dgLayout& L,
                                const F3dgIntegLayout& I, char* ws, const float* background, float* out_color)
{
    const int tiles_x = (W + F3DG_TILE - 1) / F3DG_TILE, tiles_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = tiles_x * tiles_y;
    const size_t HW = (size_t)W * H;
    const F3dgHeader* hdr = reinterpret_cast<const F3dgHeader*>(ws + L.header);
    const uint2* ranges = reinterpret_cast<const uint2*>(ws + L.ranges);
    const unsigned* point_list = reinterpret_cast<const unsigned*>(ws + L.vals[0]);
    const F3dgRec* rec = reinterpret_cast<const F3dgRec*>(ws + L.rec);
    float* final_T = reinterpret_cast<float*>(ws + L.final_T);
    unsigned* n_contrib = reinterpret_cast<unsigned*>(ws + L.n_contrib);
    unsigned short* contrib_ids = reinterpret_cast<unsigned short*>(ws + I.contrib_ids);
    unsigned* contrib_n = reinterpret_cast<unsigned*>(ws + I.contrib_n);
#ifdef F3DG_LAB
    const bool rays = g_f3dg_render_pretest && g_f3dg_render_cull && g_f3dg_render_kernel >= 3;
#else
    const bool rays = true;
#endif
    if (rays) {
        // 545 shared rays per tile, then the per-pixel kernel on the tiles that hit the contributor limit (normally none)
        unsigned* redo = reinterpret_cast<unsigned*>(ws + I.redo);
        F3DG_KLAUNCH(integrate_pass1_rays_kernel, dim3((unsigned)V * (unsigned)T), dim3(F3DG_RAYS_THREADS), 0, s, V, P, T, W, H, tiles_x, focal_x, focal_y,
                           hdr, ranges, point_list, rec, reinterpret_cast<const float4*>(ws + L.cull), background,
                           out_color, final_T, n_contrib, contrib_ids, contrib_n, redo);
        F3DG_KLAUNCH(integrate_pass1_cull_kernel, dim3((unsigned)V * (unsigned)T), dim3(F3DG_BLOCK), 0, s, V, P, T, W, H, tiles_x, focal_x, focal_y,
                           hdr, ranges, point_list, rec, reinterpret_cast<const float4*>(ws + L.cull), background,
                           out_color, final_T, n_contrib, contrib_ids, contrib_n, redo);
    }
#ifdef F3DG_LAB
    else if (g_f3dg_render_pretest && g_f3dg_render_cull && g_f3dg_render_kernel >= 2)
        F3DG_KLAUNCH(integrate_pass1_cull_kernel, dim3((unsigned)V * (unsigned)T), dim3(F3DG_BLOCK), 0, s, V, P, T, W, H, tiles_x, focal_x, focal_y,
                           hdr, ranges, point_list, rec, reinterpret_cast<const float4*>(ws + L.cull), background,
                           out_color, final_T, n_contrib, contrib_ids, contrib_n, (const unsigned*)nullptr);
    else
        for (int v = 0; v < V; v++) {       // the A/B variants of the tests: one camera per launch
            const float4* bbox = reinterpret_cast<const float4*>(ws + L.bbox) + (size_t)v * P;
            if (g_f3dg_render_pretest && g_f3dg_render_cull)          // round 1's version: per-ray pre-test + block masks from the boxes
                F3DG_KLAUNCH((integrate_pass1_kernel<true>), dim3(T), dim3(F3DG_BLOCK), 0, s, W, H, tiles_x, focal_x, focal_y,
                                   hdr, ranges + (size_t)v * T, point_list, rec + (size_t)v * P, bbox, background,
                                   out_color + (size_t)v * F3DG_OUT_CHANNELS * HW, final_T + (size_t)v * 4 * HW, n_contrib + (size_t)v * 2 * HW,
                                   contrib_ids + (size_t)v * HW * F3DG_MAX_CONTRIB, contrib_n + (size_t)v * HW);
            else
                F3DG_KLAUNCH((integrate_pass1_kernel<false>), dim3(T), dim3(F3DG_BLOCK), 0, s, W, H, tiles_x, focal_x, focal_y,
                                   hdr, ranges + (size_t)v * T, point_list, rec + (size_t)v * P, bbox, background,
                                   out_color + (size_t)v * F3DG_OUT_CHANNELS * HW, final_T + (size_t)v * 4 * HW, n_contrib + (size_t)v * 2 * HW,
                                   contrib_ids + (size_t)v * HW * F3DG_MAX_CONTRIB, contrib_n + (size_t)v * HW);
        }
#else
    (void)HW;
#endif
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

// the point stage against a prepared workspace: counting sort by pixel (bin, scan, perm), integration in pixel order,
// points-per-pixel epilogue. Any of the three outputs may be null.
int f3dg_launch_integrate_points(hipStream_t s, int view, int P, int PN, int W, int H, float focal_x, float focal_y, const F3dgLayout& L,
                                 const F3dgIntegLayout& I, char* ws, const float* points3D, const float* viewmatrix,
                                 float* out_color, float* out_alpha_integrated, float* out_color_integrated, float* alpha_min)
{
    const int tiles_x = (W + F3DG_TILE - 1) / F3DG_TILE, tiles_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = tiles_x * tiles_y;
    const size_t HWv = (size_t)W * H;
    const F3dgHeader* hdr = reinterpret_cast<const F3dgHeader*>(ws + L.header);
    // camera `view` of a batched preparation: its slice of every per-camera array (the instance list is shared, the ranges index it)
    const uint2* ranges = reinterpret_cast<const uint2*>(ws + L.ranges) + (size_t)view * T;
    const unsigned* point_list = reinterpret_cast<const unsigned*>(ws + L.vals[0]);
    const F3dgRec* rec = reinterpret_cast<const F3dgRec*>(ws + L.rec) + (size_t)view * P;
    const unsigned* n_contrib = reinterpret_cast<const unsigned*>(ws + L.n_contrib) + (size_t)view * 2 * HWv;
    const unsigned short* contrib_ids = reinterpret_cast<const unsigned short*>(ws + I.contrib_ids) + (size_t)view * HWv * F3DG_MAX_CONTRIB;
    const unsigned* contrib_n = reinterpret_cast<const unsigned*>(ws + I.contrib_n) + (size_t)view * HWv;
    unsigned* pix_points = reinterpret_cast<unsigned*>(ws + I.pix_points);
    unsigned long long* tile_last = reinterpret_cast<unsigned long long*>(ws + I.tile_last);
    unsigned* pix_start = reinterpret_cast<unsigned*>(ws + I.pix_start);
    unsigned* pt_pix = reinterpret_cast<unsigned*>(ws + I.pt_pix);
    unsigned* pt_rank = reinterpret_cast<unsigned*>(ws + I.pt_rank);
    unsigned* perm = reinterpret_cast<unsigned*>(ws + I.perm);

    F3DG_HIP_CHECK(hipMemsetAsync(ws + I.pix_points, 0, I.clear_bytes, s));
    const dim3 pgrid((PN + F3DG_BLOCK - 1) / F3DG_BLOCK);
    F3DG_KLAUNCH(integrate_points_bin_kernel, pgrid, dim3(F3DG_BLOCK), 0, s, PN, points3D, viewmatrix, W, H, tiles_x,
                       focal_x, focal_y, hdr, out_alpha_integrated, out_color_integrated, pix_points, tile_last, pt_pix, pt_rank);
    int rc = f3dg_launch_scan_inclusive(s, pix_points, pix_start, (unsigned long long)W * H,
                                        reinterpret_cast<unsigned*>(ws + I.scan_tmp), I.scan_tmp_elems, 1, nullptr);
    if (rc != F3DG_OK) return rc;
    F3DG_KLAUNCH(integrate_points_perm_kernel, pgrid, dim3(F3DG_BLOCK), 0, s, PN, pt_pix, pt_rank, pix_start, perm);
    F3DG_KLAUNCH(integrate_points_kernel, pgrid, dim3(F3DG_BLOCK), 0, s, PN, points3D, viewmatrix, W, H, tiles_x,
                       focal_x, focal_y, hdr, ranges, point_list, rec, contrib_ids, contrib_n, n_contrib, out_color,
                       out_alpha_integrated, out_color_integrated, pix_points, pix_start, perm, alpha_min);
    F3DG_HIP_CHECK(hipGetLastError());
    F3DG_KLAUNCH(integrate_epilogue_kernel, dim3(T), dim3(F3DG_BLOCK), 0, s, W, H, tiles_x, focal_x, focal_y,
                       points3D, viewmatrix, pix_points, tile_last, out_color);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}
