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
//
// ALU diet that does not change results (the kernel is VALU-bound, ~200 pixel tests per 72-byte instance):
//  * conservative pre-test. Only ~5 % of (pixel, Gaussian) tests end in a blend; the rest leave through
//    `alpha < 1/255` (or `t <= 0.2`), both of which are a bare `continue`. A float32 estimate of the exponent,
//    With b = BB/2 and a = AA (the reference's own float32 values, computed in its order) the exponent is
//    p = -(C - b^2/a)/2, and alpha < 1/255 is certain when p < thr = log(1/(255*opacity)) - 1e-4, i.e. when
//    b^2 < K0*a with K0 = C + 2 thr. The record carries K = K0*(1 - 5e-7) (>= 0), which absorbs the two float32
//    product roundings of the test  fl(b*b) < fl(K*a)  -- three VALU instructions -- so a true test PROVES
//    alpha < 1/255 and the pair is skipped before any float64 instruction, expf or divide. NaN falls through to the
//    exact path. The tests run
//    every scene with the pre-test on and off and require bit-identical outputs.
//  * per-wave culling. A tile's list holds every Gaussian whose 3-sigma SQUARE touches the 16x16 tile, but a wave
//    owns an 8x8 quadrant and only ~1/3 of the (quadrant, Gaussian) pairs contain a pixel with alpha >= 1/255. The staging
//    thread therefore also fetches the Gaussian's conservative alpha >= 1/255 box (f3dg_preprocess.hip) and publishes
//    a 4-bit strip mask; each wave compacts the 256 staged entries to its own index list with ballots and walks only
//    those. Skipped entries would have been a bare `continue` for all 64 lanes, and `contributor` is set from the
//    entry's position, so every output and auxiliary plane is bit-identical (asserted with the option on and off).
//  * per-lane work queues. After the two filters above the expensive exact path still ran with ~1/4 of the lanes
//    active, because a wave executes it whenever ANY of its 64 pixels passes. The loop is therefore split in two
//    phases per window of 64 (compacted) entries: phase 1 runs only the cheap pre-test for all 64 entries with all
//    lanes busy and leaves a 64-bit pass mask per pixel; phase 2 lets every pixel walk ITS OWN set bits in ascending
//    order (per-lane LDS addresses, hence the SoA staging arrays), so the wave executes max-over-lanes(#passes)
//    exact iterations instead of #(entries with any pass) -- about half as many, at twice the lane utilisation.
//    Per pixel the sequence of blended Gaussians and every arithmetic operation on them is unchanged.
//  * t = -BB/(2*AA) is a double quotient of float-valued operands rounded to float: identical to ONE IEEE float32
//    divide (double rounding is innocuous for p = 24, q = 53 >= 2p + 2), so the float64 divide is not needed.
#include "f3dg_common.h"

namespace {

struct PixelState {
    float Tr;
    unsigned last_contributor, max_contributor;
    float C0, C1, C2, C3, C4, C5, C6, C7;
    float dist1, dist2, distortion;
};

// The reference's per-(pixel, Gaussian) arithmetic after the geometric terms (forward.cu:511-579), in its operation
// order. Returns true when the pixel saturates (`done = true`); `contributor` is the 1-based position in the tile list.
__device__ __forceinline__ bool blend_entry(PixelState& st, unsigned contributor, float n0, float n1, float n2, float aaf,
                                            float bhalf, float CC, float opac, float cr, float cg, float cb)
{
    const double AA = aaf;
    const float bbf = 2 * bhalf;
    const double BB = bbf;

    const float t = -bbf / (2.0f * aaf);                  // == (float)(-BB / (2 * AA)), see header
    if (t <= F3DG_NEAR_PLANE)
        return false;

    const double min_value = -(BB / AA) * (BB / 4.) + CC;
    float power = (float)(-0.5f * min_value);
    if (power > 0.0f)
        power = 0.0f;

    const float alpha = fminf(0.99f, opac * expf(power));
    if (alpha < 1.0f / 255.0f)
        return false;
    const float Tr = st.Tr;
    const float test_T = Tr * (1 - alpha);
    if (test_T < 0.0001f)
        return true;

    const float mapped_max_t = (float)((F3DG_FAR_PLANE * t - F3DG_FAR_PLANE * F3DG_NEAR_PLANE) / ((F3DG_FAR_PLANE - F3DG_NEAR_PLANE) * t));

    const float length = (float)sqrt(n0 * n0 + n1 * n1 + n2 * n2 + 1e-7);
    const float nn0 = -n0 / length, nn1 = -n1 / length, nn2 = -n2 / length;

    const float A = 1 - Tr;
    const float error = mapped_max_t * mapped_max_t * A + st.dist2 - 2 * mapped_max_t * st.dist1;
    st.distortion += error * alpha * Tr;
    st.dist1 += mapped_max_t * alpha * Tr;
    st.dist2 += mapped_max_t * mapped_max_t * alpha * Tr;

    st.C0 += cr * alpha * Tr;
    st.C1 += cg * alpha * Tr;
    st.C2 += cb * alpha * Tr;
    st.C3 += nn0 * alpha * Tr;
    st.C4 += nn1 * alpha * Tr;
    st.C5 += nn2 * alpha * Tr;
    if (Tr > 0.5) {
        st.C6 = t;
        st.max_contributor = contributor;
    }
    st.C7 += alpha * Tr;

    st.Tr = test_T;
    st.last_contributor = contributor;
    return false;
}

template <bool SAVE_AUX, bool PRETEST, bool CULL, bool QUEUE>
__global__ void __launch_bounds__(F3DG_BLOCK, 8)
render_fwd_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
                  const F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                  const unsigned* __restrict__ point_list, const F3dgRec* __restrict__ rec,
                  const float4* __restrict__ bbox, const float* __restrict__ background, int bg_per_view,
                  float* __restrict__ out_color, float* __restrict__ final_T, unsigned* __restrict__ n_contrib)
{
    // XCD-aware placement: consecutive workgroup ids land on different XCDs (id % 8), so give every XCD its
    // own views: all tiles of a view then share one XCD's L2 for the record gather.
#ifdef F3DG_PLAIN_MAP
    const unsigned view = blockIdx.x / (unsigned)T;
    const unsigned tile = blockIdx.x % (unsigned)T;
#else
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned slot = blockIdx.x >> 3;
    const unsigned view = (slot / (unsigned)T) * 8u + xcd;
    const unsigned tile = slot % (unsigned)T;
#endif
    if (view >= (unsigned)V)
        return;

    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    // each wave owns an 8x8 pixel quadrant of the tile (not a 16x4 strip): the more compact footprint is touched by
    // ~12 % fewer Gaussians, which is what the per-wave culling below removes
    const unsigned lane_ = threadIdx.x & 63u, wave_ = threadIdx.x >> 6;
    const unsigned lx = (wave_ & 1u) * 8u + (lane_ & 7u), ly = (wave_ >> 1) * 8u + (lane_ >> 3);
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

    // 256 staged records, structure-of-arrays by float4 so that per-lane (divergent) reads spread over the banks
    __shared__ float4 sq0[F3DG_BLOCK];            // v0 v1 v2 v3
    __shared__ float4 sq1[F3DG_BLOCK];            // v4 v5 v6 v7
    __shared__ float4 sq2[F3DG_BLOCK];            // v8 v9 opac thr
    __shared__ float4 sq3[F3DG_BLOCK];            // r g b depth
    __shared__ unsigned char strip_mask[CULL ? F3DG_BLOCK : 1];
    __shared__ unsigned short wave_list[CULL ? F3DG_BLOCK / 64 : 1][CULL ? F3DG_BLOCK : 1];

    const F3dgRec* vrec = rec + (size_t)view * P;
    const float4* vbox = bbox + (size_t)view * P;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const float tile_px0 = (float)(tile_x * F3DG_TILE), tile_py0 = (float)(tile_y * F3DG_TILE);

    bool done = !inside;
    PixelState st;
    st.Tr = 1.0f;
    st.last_contributor = 0; st.max_contributor = (unsigned)-1;
    st.C0 = st.C1 = st.C2 = st.C3 = st.C4 = st.C5 = st.C6 = st.C7 = 0;
    st.dist1 = st.dist2 = st.distortion = 0;

    for (int i = 0; i < rounds; i++, toDo -= F3DG_BLOCK) {
        const int num_done = __syncthreads_count(done);
        if (num_done == F3DG_BLOCK)
            break;

        const unsigned progress = (unsigned)i * F3DG_BLOCK + threadIdx.x;
        if (range.x + progress < range.y) {
            const unsigned id = point_list[range.x + progress];
            const float4* src = reinterpret_cast<const float4*>(vrec + id);
            const float4 a = src[0], b = src[1], c = src[2], d = src[3];
            sq0[threadIdx.x] = a;
            sq1[threadIdx.x] = b;
            sq2[threadIdx.x] = c;
            sq3[threadIdx.x] = d;
            if (CULL) {
                const float4 bx = vbox[id];                       // (x0, x1, y0, y1) in pixel coordinates
                // bit w = the box touches wave w's quadrant: x half (w & 1), y half (w >> 1)
                const unsigned mx = (bx.x <= tile_px0 + 7.0f && bx.y >= tile_px0 ? 1u : 0u) |
                                    (bx.x <= tile_px0 + 15.0f && bx.y >= tile_px0 + 8.0f ? 2u : 0u);
                const unsigned my = (bx.z <= tile_py0 + 7.0f && bx.w >= tile_py0 ? 1u : 0u) |
                                    (bx.z <= tile_py0 + 15.0f && bx.w >= tile_py0 + 8.0f ? 2u : 0u);
                const unsigned m = ((mx & 1u) && (my & 1u) ? 1u : 0u) | ((mx & 2u) && (my & 1u) ? 2u : 0u) |
                                   ((mx & 1u) && (my & 2u) ? 4u : 0u) | ((mx & 2u) && (my & 2u) ? 8u : 0u);
                strip_mask[threadIdx.x] = (unsigned char)m;
            }
        } else if (CULL) {
            strip_mask[threadIdx.x] = 0;
        }
        __syncthreads();

        const int n = min(F3DG_BLOCK, toDo);
        int count = n;
        if (CULL) {
            // this wave's compacted list of staged entries whose box touches its strip (order preserved)
            count = 0;
#pragma unroll
            for (int c = 0; c < F3DG_BLOCK / 64; c++) {
                const unsigned e = c * 64 + lane;
                const bool bit = (strip_mask[e] >> wave) & 1u;
                const unsigned long long bal = __ballot(bit);
                if (bit) wave_list[wave][count + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)e;
                count += __popcll(bal);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const unsigned round_base = (unsigned)i * F3DG_BLOCK;

        if (!QUEUE) {
            // ---- reference-shaped loop: every lane visits every (remaining) entry
            for (int kk = 0; !done && kk < count; kk++) {
                const int j = CULL ? (int)wave_list[wave][kk] : kk;
                const float4 q0 = sq0[j], q1 = sq1[j], q2 = sq2[j];
                const float n0 = q0.x * ray_x + q0.y * ray_y + q0.z;
                const float n1 = q0.y * ray_x + q0.w * ray_y + q1.x;
                const float n2 = q0.z * ray_x + q1.x * ray_y + q1.y;
                const float aaf = ray_x * n0 + ray_y * n1 + n2;
                const float bhalf = q1.z * ray_x + q1.w * ray_y + q2.x;
                if (PRETEST) {
                    if (bhalf * bhalf < q2.w * aaf)                   // certainly alpha < 1/255 (false for NaN)
                        continue;
                }
                const float4 q3 = sq3[j];
                done = blend_entry(st, round_base + (unsigned)j + 1u, n0, n1, n2, aaf, bhalf, q2.y, q2.z, q3.x, q3.y, q3.z);
            }
        } else {
            // ---- two-phase loop over windows of 64 entries
            for (int w0 = 0; w0 < count; w0 += 64) {
                const int wn = min(64, count - w0);
                unsigned long long pass = 0;
                if (PRETEST) {
                    if (!done) {
                        // phase 1: cheap test, wave-uniform entries (LDS broadcast); two entries per trip so that the
                        // second entry's index + record loads are in flight while the first is evaluated
                        auto test = [&](const float4& q0, const float4& q1, const float4& q2) -> bool {
                            const float n0 = q0.x * ray_x + q0.y * ray_y + q0.z;
                            const float n1 = q0.y * ray_x + q0.w * ray_y + q1.x;
                            const float n2 = q0.z * ray_x + q1.x * ray_y + q1.y;
                            const float aaf = ray_x * n0 + ray_y * n1 + n2;
                            const float bhalf = q1.z * ray_x + q1.w * ray_y + q2.x;
                            return !(bhalf * bhalf < q2.w * aaf);
                        };
                        int kk = 0;
                        for (; kk + 1 < wn; kk += 2) {
                            const int j0 = CULL ? (int)wave_list[wave][w0 + kk] : (w0 + kk);
                            const int j1 = CULL ? (int)wave_list[wave][w0 + kk + 1] : (w0 + kk + 1);
                            const float4 a0 = sq0[j0], a1 = sq1[j0], a2 = sq2[j0];
                            const float4 b0 = sq0[j1], b1 = sq1[j1], b2 = sq2[j1];
                            if (test(a0, a1, a2)) pass |= 1ull << kk;
                            if (test(b0, b1, b2)) pass |= 2ull << kk;
                        }
                        if (kk < wn) {
                            const int j0 = CULL ? (int)wave_list[wave][w0 + kk] : (w0 + kk);
                            if (test(sq0[j0], sq1[j0], sq2[j0])) pass |= 1ull << kk;
                        }
                    }
                } else {
                    pass = done ? 0ull : (wn == 64 ? ~0ull : ((1ull << wn) - 1ull));
                }
                while (pass != 0 && !done) {                     // phase 2: this pixel's own passing entries, in order
                    const int kk = __builtin_ctzll(pass);
                    pass &= pass - 1;
                    const int j = CULL ? (int)wave_list[wave][w0 + kk] : (w0 + kk);
                    const float4 q0 = sq0[j], q1 = sq1[j], q2 = sq2[j], q3 = sq3[j];
                    const float n0 = q0.x * ray_x + q0.y * ray_y + q0.z;
                    const float n1 = q0.y * ray_x + q0.w * ray_y + q1.x;
                    const float n2 = q0.z * ray_x + q1.x * ray_y + q1.y;
                    const float aaf = ray_x * n0 + ray_y * n1 + n2;
                    const float bhalf = q1.z * ray_x + q1.w * ray_y + q2.x;
                    done = blend_entry(st, round_base + (unsigned)j + 1u, n0, n1, n2, aaf, bhalf, q2.y, q2.z, q3.x, q3.y, q3.z);
                }
            }
        }
    }

    if (inside) {
        const float* bg = background + (bg_per_view ? 3 * view : 0);
        const float Tr = st.Tr;
        const float distortion_before_normalized = st.distortion;
        const float distortion = (float)(st.distortion / ((1 - Tr) * (1 - Tr) + 1e-7));

        if (SAVE_AUX) {
            float* fT = final_T + (size_t)view * 4 * HW;
            fT[pix_id] = Tr;
            fT[pix_id + HW] = st.dist1;
            fT[pix_id + 2 * HW] = st.dist2;
            fT[pix_id + 3 * HW] = distortion_before_normalized;
            unsigned* nc = n_contrib + (size_t)view * 2 * HW;
            nc[pix_id] = st.last_contributor;
            nc[pix_id + HW] = st.max_contributor;
        }
        float* out = out_color + (size_t)view * F3DG_OUT_CHANNELS * HW;
        out[0 * HW + pix_id] = st.C0 + Tr * bg[0];
        out[1 * HW + pix_id] = st.C1 + Tr * bg[1];
        out[2 * HW + pix_id] = st.C2 + Tr * bg[2];
        out[3 * HW + pix_id] = st.C3;
        out[4 * HW + pix_id] = st.C4;
        out[5 * HW + pix_id] = st.C5;
        out[6 * HW + pix_id] = st.C6;
        out[7 * HW + pix_id] = st.C7;
        out[8 * HW + pix_id] = distortion;
    }
}

} // namespace

int f3dg_launch_render(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y,
                       const F3dgHeader* hdr, const uint2* ranges, const unsigned* point_list, const F3dgRec* rec,
                       const float4* bbox, const float* background, int bg_per_view, float* out_color, float* final_T,
                       unsigned* n_contrib, int save_aux)
{
    const int tiles_x = (W + F3DG_TILE - 1) / F3DG_TILE, tiles_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = tiles_x * tiles_y;
    const unsigned groups = (unsigned)((V + 7) / 8);
    dim3 grid(groups * 8u * (unsigned)T);
#define F3DG_LAUNCH(AUX, PRE, CUL, QUE) hipLaunchKernelGGL((render_fwd_kernel<AUX, PRE, CUL, QUE>), grid, dim3(F3DG_BLOCK), 0, s, V, P, \
                                                            W, H, tiles_x, T, focal_x, focal_y, hdr, ranges, point_list, rec,   \
                                                            bbox, background, bg_per_view, out_color, final_T, n_contrib)
#define F3DG_LAUNCH_Q(AUX, PRE, CUL) do { if (g_f3dg_render_queue) F3DG_LAUNCH(AUX, PRE, CUL, true); else F3DG_LAUNCH(AUX, PRE, CUL, false); } while (0)
    const int variant = (save_aux ? 4 : 0) | (g_f3dg_render_pretest ? 2 : 0) | (g_f3dg_render_cull ? 1 : 0);
    switch (variant) {
    case 0: F3DG_LAUNCH_Q(false, false, false); break;
    case 1: F3DG_LAUNCH_Q(false, false, true); break;
    case 2: F3DG_LAUNCH_Q(false, true, false); break;
    case 3: F3DG_LAUNCH_Q(false, true, true); break;
    case 4: F3DG_LAUNCH_Q(true, false, false); break;
    case 5: F3DG_LAUNCH_Q(true, false, true); break;
    case 6: F3DG_LAUNCH_Q(true, true, false); break;
    default: F3DG_LAUNCH_Q(true, true, true); break;
    }
#undef F3DG_LAUNCH_Q
#undef F3DG_LAUNCH
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}
