// f3dg_render.hip -- per-tile front-to-back GOF compositing (the roofline kernel of the path).
//
// Replaces renderCUDA<3> (reference RAST/cuda_rasterizer/forward.cu:409-612): for every pixel of a 16x16 tile, walk the tile's
// depth-sorted Gaussian list, intersect the pixel ray with each Gaussian's quadric (view2gaussian: Sigma', B, C), turn the minimum
// of the quadric along the ray into an alpha, and blend RGB, view-space normal, median depth, alpha and the 2DGS-style distortion
// term front to back. All views of a call are ONE launch; an XCD-aware id -> (view, tile) map keeps a view's records in one L2.
//
// The blend is a strict per-pixel recurrence whose float32 / float64 operation order is the reference's (blend_entry; the file is
// built with -ffp-contract=off): the exponent -(1/2)(C - B^2/4A) cancels 1e5..1e6 x and any re-association moves isolated pixels by
// 1e-2. blend_entry_fast is the inference-mode variant: the same float32 a, b in the reference's order, the float64
// island replaced by error-free float32 pairs. There is no inter-pixel arithmetic, hence no MFMA.
//
// Three generations of the kernel live here; all of them only ever REMOVE (pixel, Gaussian) pairs that are a bare `continue` in the
// reference (alpha < 1/255 proven by a conservative test), so their images are bit-identical within an arithmetic mode:
//   render3_fwd_kernel (default, option render_kernel = 3): one wave64 per 8x8 pixel quadrant, no workgroup barriers. The wave
//       scans the tile list for the entries whose quadrant bit is set (F3DG_ID_BITS), stages 64 records per window by
//       global_load_lds, tests entries against the quadrant's pixels with the Gaussians across the lanes (ballots delivered by
//       v_writelane) and blends with the pixels across the lanes. See the comment above the kernel.
//   render2_fwd_kernel (render_kernel = 2): one 256-thread workgroup per tile, four coupled waves (an 8x8 quadrant each), 192 or
//       256 entries staged per round with two barriers, per-4x4-block compacted lists, phase 1 across the lanes.
//   render_fwd_kernel (render_kernel = 1): the round-1 pixel-lane kernel. Its plain variant (no pre-test, no culling, no queues) is
//       the transcription-order baseline every other variant is compared with bit for bit
//       (tests/test_raster_forward_gpu.py::test_pretest_is_conservative_bit_identical_outputs). Its filters:
//  * conservative pre-test. Only ~5 % of (pixel, Gaussian) tests end in a blend; the rest leave through `alpha < 1/255` (or
//    `t <= 0.2`), both of which are a bare `continue`. With b = BB/2 and a = AA (the reference's own float32 values, computed in
//    its order) the exponent is p = -(C - b^2/a)/2, and alpha < 1/255 is certain when p < thr = log(1/(255*opacity)) - 1e-4, i.e.
//    when b^2 < K0*a with K0 = C + 2 thr. The record carries K = K0*(1 - 5e-7) (>= 0), which absorbs the two float32 product
//    roundings of the test  fl(b*b) < fl(K*a)  -- three VALU instructions -- so a true test PROVES alpha < 1/255 and the pair is
//    skipped before any float64 instruction, expf or divide. NaN falls through to the exact path.
//  * per-group culling. A tile's list holds every Gaussian whose 3-sigma SQUARE touches the 16x16 tile, but a 16-lane group owns a
//    4x4 pixel block and only ~1/5 of the (block, Gaussian) pairs contain a pixel with alpha >= 1/255. The staging thread fetches
//    the Gaussian's conservative alpha >= 1/255 box (f3dg_preprocess.hip) and publishes a 16-bit block mask; each wave compacts the
//    staged entries into FOUR index lists (one per 16-lane group) with ballots, and every group walks only its own. `contributor`
//    is set from the entry's position, so every output and auxiliary plane is unchanged.
//  * per-lane work queues. The loop is split in two phases per window of 64 (compacted) entries: phase 1 runs only the cheap
//    pre-test for all 64 entries with all lanes busy and leaves a 64-bit pass mask per pixel; phase 2 lets every pixel walk ITS
//    OWN set bits in ascending order, so the wave executes max-over-lanes(#passes) exact iterations instead of #(entries with any
//    pass). Per pixel the sequence of blended Gaussians and every arithmetic operation on them is unchanged.
//  * t = -BB/(2*AA) is a double quotient of float-valued operands rounded to float: identical to ONE IEEE float32 divide (double
//    rounding is innocuous for p = 24, q = 53 >= 2p + 2), so the float64 divide is not needed.
#include "f3dg_common.h"
#include "f3dg_ellipse.h"

#include <stdio.h>
#include <string.h>

extern thread_local const char* g_f3dg_last_render_kernel;

namespace {

struct PixelState {
    float Tr;
    unsigned last_contributor, max_contributor;
    float C0, C1, C2, C3, C4, C5, C6, C7;
    float dist1, dist2, distortion;
};

// The reference's per-(pixel, Gaussian) arithmetic after the geometric terms (forward.cu:511-579), in its operation
// order. Returns true when the pixel saturates (`done = true`); `contributor` is the 1-based position in the tile list.
// NORMAL / DIST = false (f3dg_forward_sets with F3DG_FLAG_SKIP_NORMAL / F3DG_FLAG_SKIP_DISTORTION; the one-wave kernel only): the normal
// channels 3..5 / the distortion channel 8 are neither accumulated nor written; every other channel is bit-identical.
template <bool NORMAL = true, bool DIST = true>
__device__ __forceinline__ bool blend_entry(PixelState& st, unsigned contributor, float n0, float n1, float n2, float aaf,
                                            float bhalf, float CC, float opac, float cr, float cg, float cb)
{
    const double AA = aaf;
    const float bbf = 2 * bhalf;
    const double BB = bbf;

    // ONE float64 division serves both uses: -BB / (2 * AA) is the correctly rounded quotient BB / AA scaled by -1/2
    // (exact), so t below has the bits of the reference's (float)(-BB / (2 * AA)).
    const double q = BB / AA;
    const float t = (float)(-0.5 * q);
    if (t <= F3DG_NEAR_PLANE)
        return false;

    const double min_value = -q * (BB / 4.) + CC;
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

    if (DIST) {
        const float mapped_max_t = (float)((F3DG_FAR_PLANE * t - F3DG_FAR_PLANE * F3DG_NEAR_PLANE) / ((F3DG_FAR_PLANE - F3DG_NEAR_PLANE) * t));
        const float A = 1 - Tr;
        const float error = mapped_max_t * mapped_max_t * A + st.dist2 - 2 * mapped_max_t * st.dist1;
        st.distortion += error * alpha * Tr;
        st.dist1 += mapped_max_t * alpha * Tr;
        st.dist2 += mapped_max_t * mapped_max_t * alpha * Tr;
    }

    st.C0 += cr * alpha * Tr;
    st.C1 += cg * alpha * Tr;
    st.C2 += cb * alpha * Tr;
    if (NORMAL) {
        const float length = (float)sqrt(n0 * n0 + n1 * n1 + n2 * n2 + 1e-7);
        const float nn0 = -n0 / length, nn1 = -n1 / length, nn2 = -n2 / length;
        st.C3 += nn0 * alpha * Tr;
        st.C4 += nn1 * alpha * Tr;
        st.C5 += nn2 * alpha * Tr;
    }
    if (Tr > 0.5) {
        st.C6 = t;
        st.max_contributor = contributor;
    }
    st.C7 += alpha * Tr;

    st.Tr = test_T;
    st.last_contributor = contributor;
    return false;
}


// Fast-mode counterpart of blend_entry (option "render_fast", the default): the same decisions and the same float32
// accumulations, but the reference's float64 island (forward.cu:511-522,545,548) is evaluated with error-free float32
// pairs instead of float64 divides / square roots:
//   * aaf, bhalf (and n0..n2) are the reference's own float32 values, computed in its operation order by the caller:
//     their rounding errors are amplified 1e5..1e6 x by the cancellation and must be reproduced, not improved on;
//   * b^2/a is formed as a double-single quotient (q1 + q2, relative error ~2^-45): b*b = p + e exactly (FMA), q1 = p*r,
//     q2 = ((p - q1*a) + e)*r with r ~ 1/a; C - q1 is exact (Sterbenz) wherever the exponent matters, so
//     min_value = (C - q1) - q2 carries one float32 rounding of a number of magnitude <~ 20: |d power| <~ 1e-6;
//   * t = -b/a from the same reciprocal with one Newton step (<= 1 ulp), exp() as v_exp_f32(power * log2 e),
//     the NDC depth as c0 - c1/t with a hardware reciprocal, the normal with v_rsq_f32.
// Every output stays within ~1e-6 relative of blend_entry's; the parity tests gate this mode at the same 1e-4 / 99.9 % /
// 80 dB bar as the exact one (tests/test_raster_forward_gpu.py) and report both against the oracle.
template <bool NORMAL = true, bool DIST = true>
__device__ __forceinline__ bool blend_entry_fast(PixelState& st, unsigned contributor, float n0, float n1, float n2, float aaf,
                                                 float bhalf, float CC, float opac, float cr, float cg, float cb)
{
    float t, G;
    f3dg_fast_t_G(aaf, bhalf, CC, t, G);
    // (double)t <= 0.2  <=>  t < 0.2f: 0.2f is the float just above 0.2 (false for NaN, as the reference's test). Tested together
    // with alpha below: a wave nearly always holds a lane that passes, so an early branch here only costs scalar instructions
    const bool behind = t < 0.2f;
    const float alpha = fminf(0.99f, opac * G);
    if (behind || alpha < 1.0f / 255.0f)
        return false;
    const float Tr = st.Tr;
    const float test_T = Tr * (1 - alpha);
    if (test_T < 0.0001f)
        return true;

    // (the accumulations below are contracted into FMAs: fewer roundings than the reference's separate products and sums, ~1e-8
    // absolute on the distortion channel, whose values are 1e-7..1e-2)
    const float w = alpha * Tr;
    if (DIST) {
        // (FAR*t - FAR*NEAR) / ((FAR - NEAR)*t) = FAR/(FAR-NEAR) - (FAR*NEAR/(FAR-NEAR)) / t
        const float mapped_max_t = fmaf(-0.20040080160320642f, __builtin_amdgcn_rcpf(t), 1.0020040080160322f);
        const float A = 1 - Tr;
        const float m2 = mapped_max_t * mapped_max_t;
        const float error = fmaf(-2.0f * mapped_max_t, st.dist1, fmaf(m2, A, st.dist2));
        st.distortion = fmaf(error, w, st.distortion);
        st.dist1 = fmaf(mapped_max_t, w, st.dist1);
        st.dist2 = fmaf(m2, w, st.dist2);
    }
    st.C0 = fmaf(cr, w, st.C0);
    st.C1 = fmaf(cg, w, st.C1);
    st.C2 = fmaf(cb, w, st.C2);
    if (NORMAL) {
        // (the unit normal is formed first and then weighted -- the order of f3dg_blend.h, where the packed schedule of
        // f3dg_render4.hip hands it from the lane that evaluated the pair to the lane that owns the pixel)
        const float ninv = -__builtin_amdgcn_rsqf(fmaf(n2, n2, fmaf(n1, n1, n0 * n0)) + 1e-7f);
        st.C3 = fmaf(n0 * ninv, w, st.C3);
        st.C4 = fmaf(n1 * ninv, w, st.C4);
        st.C5 = fmaf(n2 * ninv, w, st.C5);
    }
    if (Tr > 0.5f) {
        st.C6 = t;
        st.max_contributor = contributor;
    }
    st.C7 += w;

    st.Tr = test_T;
    st.last_contributor = contributor;
    return false;
}

#ifdef F3DG_LAB      // ---- generation 1, lab builds only (the plain-transcription baseline of the bit-identity tests)
#define F3DG_ROUND (F3DG_BLOCK - 1)     // list entries staged per round; LDS slot F3DG_ROUND is the sentinel

template <bool SAVE_AUX, bool PRETEST, bool CULL, bool QUEUE, bool FAST>
__global__ void __launch_bounds__(F3DG_BLOCK, 8)
render_fwd_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
                  const F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                  const unsigned* __restrict__ point_list, const F3dgRec* __restrict__ rec,
                  const float4* __restrict__ bbox, const float* __restrict__ background, int bg_per_view,
                  float* __restrict__ out_color, float* __restrict__ final_T, unsigned* __restrict__ n_contrib)
{
    unsigned view, tile;                      // all tiles of a view share one XCD's L2 for the record gather
    f3dg_xcd_map(blockIdx.x, (unsigned)V, (unsigned)T, view, tile);

    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    // each wave owns an 8x8 pixel quadrant of the tile and each of its four 16-lane groups a 4x4 block of it: the culled
    // entry lists below are kept PER 16-LANE GROUP (a 4x4 block is touched by ~40 % fewer Gaussians than an 8x8 quadrant)
    const unsigned lane_ = threadIdx.x & 63u, wave_ = threadIdx.x >> 6;
    const unsigned grp_ = lane_ >> 4, gi_ = lane_ & 15u;
    const unsigned blk_x = (wave_ & 1u) * 2u + (grp_ & 1u), blk_y = (wave_ >> 1) * 2u + (grp_ >> 1);    // 4x4 block in tile
    const unsigned lx = blk_x * 4u + (gi_ & 3u), ly = blk_y * 4u + (gi_ >> 2);
    const unsigned pix_x = tile_x * F3DG_TILE + lx, pix_y = tile_y * F3DG_TILE + ly;
    const bool inside = pix_x < (unsigned)W && pix_y < (unsigned)H;
    const size_t HW = (size_t)H * W;
    const size_t pix_id = (size_t)W * pix_y + pix_x;
    const float pixf_x = (float)pix_x + 0.5f, pixf_y = (float)pix_y + 0.5f;
    const float ray_x = (float)((pixf_x - W / 2.) / focal_x);
    const float ray_y = (float)((pixf_y - H / 2.) / focal_y);

    uint2 range = ranges[(size_t)view * T + tile];
    if (hdr->overflow) range = make_uint2(0, 0);
    const int rounds = (int)((range.y - range.x + F3DG_ROUND - 1) / F3DG_ROUND);
    int toDo = (int)(range.y - range.x);

    // 256 staged records, structure-of-arrays by float4 so that per-lane (divergent) reads spread over the banks
    // A round stages F3DG_ROUND = 255 list entries; slot 255 holds the sentinel record that pads the culled lists (so
    // that list entries fit a byte and the whole block fits 20 KB of LDS = 8 blocks per CU).
    __shared__ float4 sq0[F3DG_BLOCK];            // v0 v1 v2 v3
    __shared__ float4 sq1[F3DG_BLOCK];            // v4 v5 v6 v7
    __shared__ float4 sq2[F3DG_BLOCK];            // v8 v9 opac thr
    __shared__ float4 sq3[F3DG_BLOCK];            // r g b, and (culling) the 16-bit block mask in place of the depth
    if (threadIdx.x == F3DG_ROUND) {
        // sentinel: A = x^2 + y^2 + 1 > 0, B = 0, K = +inf  =>  fails the pre-test (0 < inf), and in blend_entry t = -0 is
        // behind the near plane; opacity 0. It can never contribute, whatever filters are enabled.
        sq0[F3DG_ROUND] = make_float4(1.0f, 0.0f, 0.0f, 1.0f);
        sq1[F3DG_ROUND] = make_float4(0.0f, 1.0f, 0.0f, 0.0f);
        sq2[F3DG_ROUND] = make_float4(0.0f, 0.0f, 0.0f, __builtin_inff());
        sq3[F3DG_ROUND] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);     // block mask 0: in no list; .x/.y: see done_cnt
    }
    // block-wide count of finished pixels, double buffered by round parity. It lives in the sentinel's (never used)
    // colour so that the block needs exactly 20480 B of LDS; __syncthreads_count would add a 256 B scratch array.
    int* done_cnt = reinterpret_cast<int*>(&sq3[F3DG_ROUND]);
    __syncthreads();
    __shared__ __align__(16) unsigned char grp_list[CULL ? F3DG_BLOCK / 64 : 1][CULL ? 4 : 1][CULL ? F3DG_BLOCK : 1];   // per wave, per 16-lane group

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

    for (int i = 0; i < rounds; i++, toDo -= F3DG_ROUND) {
        {
            const unsigned long long dl = __ballot(done);
            if ((threadIdx.x & 63u) == 0)
                atomicAdd(&done_cnt[i & 1], __popcll(dl));
        }
        __syncthreads();
        const int num_done = done_cnt[i & 1];
        if (threadIdx.x == 0)
            done_cnt[(i + 1) & 1] = 0;          // everyone has read it (round i - 1); next added to after the barrier below
        if (num_done == F3DG_BLOCK)
            break;

        const unsigned progress = (unsigned)i * F3DG_ROUND + threadIdx.x;
        if (threadIdx.x < F3DG_ROUND && range.x + progress < range.y) {
            const unsigned id = point_list[range.x + progress] & F3DG_ID_MASK;
            const float4* src = reinterpret_cast<const float4*>(vrec + id);
            const float4 a = src[0], b = src[1], c = src[2];
            float4 d = src[3];
            sq0[threadIdx.x] = a;
            sq1[threadIdx.x] = b;
            sq2[threadIdx.x] = c;
            if (CULL) {
                const float4 bx = vbox[id];                       // (x0, x1, y0, y1) in pixel coordinates
                unsigned mx = 0, my = 0;                          // which of the 4 block columns / rows the box touches
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (bx.x <= tile_px0 + (float)(4 * q + 3) && bx.y >= tile_px0 + (float)(4 * q)) mx |= 1u << q;
                    if (bx.z <= tile_py0 + (float)(4 * q + 3) && bx.w >= tile_py0 + (float)(4 * q)) my |= 1u << q;
                }
                const unsigned m = ((my & 1u) ? mx : 0u) | ((my & 2u) ? mx << 4 : 0u) | ((my & 4u) ? mx << 8 : 0u) |
                                   ((my & 8u) ? mx << 12 : 0u);
                d.w = __uint_as_float(m);
            }
            sq3[threadIdx.x] = d;
        } else if (CULL) {
            sq3[threadIdx.x].w = 0.0f;
        }
        __syncthreads();

        const int n = min(F3DG_ROUND, toDo);
        int count = n;          // wave-uniform trip count: the longest of the wave's four group lists when culling
        if (CULL) {
            // four compacted lists per wave, one per 16-lane group: the staged entries whose box touches the group's
            // 4x4 block, in list order. The lists are first filled with the index of the never-visible sentinel record, so
            // that the three shorter lists are padded to the longest one and the loops below need no per-lane bounds.
            {
                uint4* fill = reinterpret_cast<uint4*>(&grp_list[wave][0][0]);       // 4 lists x 256 x u8 = 64 x 16 B
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
        const unsigned char* my_list = CULL ? grp_list[wave][grp_] : nullptr;
        const unsigned round_base = (unsigned)i * F3DG_ROUND;

        if (!QUEUE) {
            // ---- reference-shaped loop: every lane visits every (remaining) entry
            for (int kk = 0; !done && kk < count; kk++) {
                const int j = CULL ? (int)my_list[kk] : kk;
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
                done = (FAST ? blend_entry_fast<> : blend_entry<>)(st, round_base + (unsigned)j + 1u, n0, n1, n2, aaf, bhalf, q2.y, q2.z, q3.x, q3.y, q3.z);
            }
        } else {
            // ---- two-phase loop over windows of 64 entries
            for (int w0 = 0; w0 < count; w0 += 64) {
                const int wn = min(64, count - w0);
                unsigned long long pass = 0;
                if (PRETEST) {
                    if (!done) {
                        // phase 1: cheap test; every 16-lane group walks ITS list (4 distinct LDS addresses per read, a
                        // broadcast inside each group); two entries per trip so that the second entry's index + record
                        // loads are in flight while the first is evaluated
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
                            const int j0 = CULL ? (int)my_list[w0 + kk] : (w0 + kk);
                            const int j1 = CULL ? (int)my_list[w0 + kk + 1] : (w0 + kk + 1);
                            const float4 a0 = sq0[j0], a1 = sq1[j0], a2 = sq2[j0];
                            const float4 b0 = sq0[j1], b1 = sq1[j1], b2 = sq2[j1];
                            if (test(a0, a1, a2)) pass |= 1ull << kk;
                            if (test(b0, b1, b2)) pass |= 2ull << kk;
                        }
                        if (kk < wn) {
                            const int j0 = CULL ? (int)my_list[w0 + kk] : (w0 + kk);
                            if (test(sq0[j0], sq1[j0], sq2[j0])) pass |= 1ull << kk;
                        }
                    }
                } else {
                    pass = done ? 0ull : (wn == 64 ? ~0ull : ((1ull << wn) - 1ull));
                }
                while (pass != 0 && !done) {                     // phase 2: this pixel's own passing entries, in order
                    const int kk = __builtin_ctzll(pass);
                    pass &= pass - 1;
                    const int j = CULL ? (int)my_list[w0 + kk] : (w0 + kk);
                    const float4 q0 = sq0[j], q1 = sq1[j], q2 = sq2[j], q3 = sq3[j];
                    const float n0 = q0.x * ray_x + q0.y * ray_y + q0.z;
                    const float n1 = q0.y * ray_x + q0.w * ray_y + q1.x;
                    const float n2 = q0.z * ray_x + q1.x * ray_y + q1.y;
                    const float aaf = ray_x * n0 + ray_y * n1 + n2;
                    const float bhalf = q1.z * ray_x + q1.w * ray_y + q2.x;
                    done = (FAST ? blend_entry_fast<> : blend_entry<>)(st, round_base + (unsigned)j + 1u, n0, n1, n2, aaf, bhalf, q2.y, q2.z, q3.x, q3.y, q3.z);
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

#endif // F3DG_LAB (render_fwd_kernel)

// =====================================================================================================================
// render2: the same compositing with phase 1 turned around -- GAUSSIANS across the lanes instead of pixels.
//
// In the kernel above a phase-1 trip tests ONE list entry per 16-lane group against the group's 16 pixels and costs ~31 VALU
// instructions (the reference's own a, b in its operation order + the K test), i.e. ~0.5 instruction per (pixel, entry) test;
// with the float64 island gone it is half of the kernel. Here a sub-step gives every LANE one entry of its group's list (4 blocks
// x 16 entries per wave) and the lane tests it against all 16 pixels of the group's 4x4 block with the conservative ellipse of
// f3dg_preprocess.hip (E(dx, dy) = a dx^2 + b dx dy + c dy^2 <= 1 in pixel offsets from the ellipse centre: two FMAs per pixel
// after per-row / per-column set-up, no cancellation, so plain float32 is safe). The 16 per-pixel comparisons ARE wave ballots
// (v_cmp writes a lane mask): ballot p holds, for each of the wave's four blocks, which of its 16 entries can touch pixel p of
// that block. Two v_writelane per ballot park them in lanes p and 16 + p of one register, one ds_bpermute hands every pixel lane
// the 16 bits of its block, and four sub-steps fill the 64-bit pass mask phase 2 walks exactly as before. ~115 VALU instructions
// per 1024 (pixel, entry) tests instead of ~500, independent of how many of the wave's pixels are still alive; blocks whose 16
// pixels are all finished get an empty list.
//
// The ellipse test is conservative (it passes whenever alpha >= 1/255 is possible, with the worst-case bound on the reference's
// own float32 rounding of a and b that the culling box already used), so phase 2 sees a superset of the pairs the K pre-test let
// through and re-derives every decision from the reference's arithmetic: outputs are bit-identical to the kernel above in either
// arithmetic mode (tests/test_raster_forward_gpu.py::test_render2_bit_identical).


// Work counters of the one-wave kernel (option render_count = 1; f3dg_debug_render_counts): [0] list entries staged (record gathers),
// [1] list entries scanned, [2] phase-2 trips (wave iterations), [3] slides, [4] lane-trips = (pixel, entry) pairs that entered phase 2
// ([4] / (64 [2]) = lane utilisation of phase 2), [5] waves, [6] / [7] the trips of slides that began with at most 8 / at most 24 of the
// quadrant's 64 pixels still unsaturated, [8] / [9] those slides. 64 rows against atomic contention; summed on the host.
__device__ unsigned long long g_f3dg_counts[64][16];
#ifdef F3DG_LAB
// lab build: slides every quadrant wave of the last counting launch performed (option render_replay = 1 with render_count = 1), replayed
// by render3s_stage_only_kernel (render_replay = 2): the launch's scan + staging + phase 1 without any phase 2
#define F3DG_SLIDE_LOG_N (1u << 18)
__device__ unsigned g_f3dg_slide_log[F3DG_SLIDE_LOG_N];
#endif

// ---- optional phase timing (build with -DF3DG_TIMING: tools/render_timing.py). Shader-clock cycles per wave, summed over all
// waves of all launches since the last reset: [0] barrier waits, [1] staging, [2] list build, [3] phase 1, [4] phase 2,
// [5] unused, [6] total, [7] waves.
__device__ unsigned long long g_f3dg_timing[8];
#ifdef F3DG_TIMING
#define F3DG_T_DECL unsigned long long t_acc[6] = {0, 0, 0, 0, 0, 0}; unsigned long long t_last = __builtin_amdgcn_s_memtime(); const unsigned long long t_begin = t_last;
#define F3DG_T_MARK(k) do { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_acc[k] += t_now - t_last; t_last = t_now; } while (0)
#define F3DG_T_FLUSH do { if ((threadIdx.x & 63u) == 0) { for (int k_ = 0; k_ < 6; k_++) atomicAdd(&g_f3dg_timing[k_], t_acc[k_]); \
                          atomicAdd(&g_f3dg_timing[6], __builtin_amdgcn_s_memtime() - t_begin); atomicAdd(&g_f3dg_timing[7], 1ull); } } while (0)
#else
#define F3DG_T_DECL
#define F3DG_T_MARK(k) do { } while (0)
#define F3DG_T_FLUSH do { } while (0)
#endif

#ifdef F3DG_LAB      // ---- generation 2, lab builds only
template <bool SAVE_AUX, bool FAST, int ROUND, int OCC>
__global__ void __launch_bounds__(F3DG_BLOCK, OCC)
render2_fwd_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
                   const F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                   const unsigned* __restrict__ point_list, const F3dgRec* __restrict__ rec,
                   const float4* __restrict__ cull, const float* __restrict__ background, int bg_per_view,
                   float* __restrict__ out_color, float* __restrict__ final_T, unsigned* __restrict__ n_contrib)
{
    unsigned view, tile;                      // all tiles of a view share one XCD's L2 for the record gather
    f3dg_xcd_map(blockIdx.x, (unsigned)V, (unsigned)T, view, tile);

    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned grp = lane >> 4, gi = lane & 15u;
    const unsigned blk_x = (wave & 1u) * 2u + (grp & 1u), blk_y = (wave >> 1) * 2u + (grp >> 1);    // 4x4 block in tile
    const unsigned lx = blk_x * 4u + (gi & 3u), ly = blk_y * 4u + (gi >> 2);
    const unsigned pix_x = tile_x * F3DG_TILE + lx, pix_y = tile_y * F3DG_TILE + ly;
    const bool inside = pix_x < (unsigned)W && pix_y < (unsigned)H;
    const size_t HW = (size_t)H * W;
    const size_t pix_id = (size_t)W * pix_y + pix_x;
    const float pixf_x = (float)pix_x + 0.5f, pixf_y = (float)pix_y + 0.5f;
    const float ray_x = (float)((pixf_x - W / 2.) / focal_x);
    const float ray_y = (float)((pixf_y - H / 2.) / focal_y);
    const float blk_px0 = (float)(tile_x * F3DG_TILE + blk_x * 4u), blk_py0 = (float)(tile_y * F3DG_TILE + blk_y * 4u);

    uint2 range = ranges[(size_t)view * T + tile];
    if (hdr->overflow) range = make_uint2(0, 0);
    const int rounds = (int)((range.y - range.x + ROUND - 1) / ROUND);      // ROUND list entries are staged per round

    __shared__ float4 sA[ROUND];            // v0 v1 v2 v3
    __shared__ float4 sB[ROUND];            // v4 v5 v6 v7
    __shared__ float4 sC[ROUND];            // v8 v9 opacity r
    __shared__ float2 sD[ROUND];            // g b
    __shared__ float4 sE[ROUND];            // ellipse: cx cy a b
    __shared__ float sF[ROUND];             //          c
    __shared__ unsigned short sM[ROUND];    // which of the tile's 16 4x4 blocks the ellipse's box touches
    __shared__ __align__(16) unsigned char lists[F3DG_BLOCK / 64][4][ROUND];     // per wave, per 16-lane group
    __shared__ int done_cnt[2];
    if (threadIdx.x < 2) done_cnt[threadIdx.x] = 0;
    __syncthreads();

    const F3dgRec* vrec = rec + (size_t)view * P;
    const float4* vcull = cull + (size_t)view * P;
    const float tile_px0 = (float)(tile_x * F3DG_TILE), tile_py0 = (float)(tile_y * F3DG_TILE);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned pull = (gi + 16u * (grp >> 1)) * 4u;       // ds_bpermute source of this pixel's ballot half
    const unsigned pull_shift = 16u * (grp & 1u);
    const unsigned char* my_list = lists[wave][grp];

    bool done = !inside;
    PixelState st;
    st.Tr = 1.0f;
    st.last_contributor = 0; st.max_contributor = (unsigned)-1;
    st.C0 = st.C1 = st.C2 = st.C3 = st.C4 = st.C5 = st.C6 = st.C7 = 0;
    st.dist1 = st.dist2 = st.distortion = 0;
    F3DG_T_DECL

    for (int i = 0; i < rounds; i++) {
        const unsigned long long alive = __ballot(!done);
        if (lane == 0)
            atomicAdd(&done_cnt[i & 1], 64 - __popcll(alive));
        __syncthreads();
        F3DG_T_MARK(0);
        const int num_done = done_cnt[i & 1];
        if (threadIdx.x == 0)
            done_cnt[(i + 1) & 1] = 0;          // everyone has read it (round i - 1); next added to after the barrier below
        if (num_done == F3DG_BLOCK)
            break;

        const unsigned progress = (unsigned)i * ROUND + threadIdx.x;
        unsigned short m16 = 0;
        if (threadIdx.x < ROUND && range.x + progress < range.y) {
            const unsigned id = point_list[range.x + progress] & F3DG_ID_MASK;
            const float4* src = reinterpret_cast<const float4*>(vrec + id);
            const float4 a = src[0], b = src[1], c = src[2], d = src[3];
            const float4 e0 = vcull[id];
            sA[threadIdx.x] = a;
            sB[threadIdx.x] = b;
            sC[threadIdx.x] = make_float4(c.x, c.y, c.z, d.x);
            sD[threadIdx.x] = make_float2(d.y, d.z);
            sE[threadIdx.x] = e0;
            sF[threadIdx.x] = d.w;
            m16 = (unsigned short)ellipse_block_mask(e0, d.w, tile_px0, tile_py0);
        }
        if (threadIdx.x < ROUND) sM[threadIdx.x] = m16;
        F3DG_T_MARK(1);
        __syncthreads();
        F3DG_T_MARK(0);

        // four compacted lists per wave, one per 16-lane group (4x4 block), in list order; finished blocks get none
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
        F3DG_T_MARK(2);
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
            F3DG_T_MARK(3);
            unsigned long long pass = done ? 0ull : ((unsigned long long)pass_hi << 32) | pass_lo;

            // ---- phase 2: this pixel's own passing entries, in list order, through the reference's arithmetic
            while (pass != 0 && !done) {
                const int kk = __builtin_ctzll(pass);
                pass &= pass - 1;
                const int j = (int)my_list[w0 + kk];
                const float4 q0 = sA[j], q1 = sB[j], q2 = sC[j];
                const float2 q3 = sD[j];
                const float n0 = q0.x * ray_x + q0.y * ray_y + q0.z;
                const float n1 = q0.y * ray_x + q0.w * ray_y + q1.x;
                const float n2 = q0.z * ray_x + q1.x * ray_y + q1.y;
                const float aaf = ray_x * n0 + ray_y * n1 + n2;
                const float bhalf = q1.z * ray_x + q1.w * ray_y + q2.x;
                done = (FAST ? blend_entry_fast<> : blend_entry<>)(st, round_base + (unsigned)j + 1u, n0, n1, n2, aaf, bhalf, q2.y, q2.z, q2.w, q3.x, q3.y);
            }
            F3DG_T_MARK(4);
        }
    }
    F3DG_T_FLUSH;

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

#endif // F3DG_LAB (render2_fwd_kernel)

// =====================================================================================================================
// render3: ONE wave64 per 8x8 pixel quadrant of a tile -- no workgroup barriers, nothing shared between waves.
//
// render2 couples the four waves of a tile through two __syncthreads per staging round: an instrumented build attributes 25-30 % of
// a wave's life to waiting at them (the quadrants of a tile have different amounts of work) and 20-40 % to the staged gathers, which
// all four waves sit out together; VALU issue reaches ~62 %. Here a workgroup is one wave that owns a quadrant from its first list
// entry to its last pixel's saturation and then retires; the SIMD's other waves (other quadrants, other tiles, up to 8 per SIMD) fill
// every wait, and a quadrant stops staging as soon as ITS 64 pixels are finished instead of the tile's 256.
//
//   scan     the tile's list is read 64 ids at a time (the next 64 are always in flight); an entry is kept when the quadrant's bit
//            of the mask that instance generation left above the id is set (F3DG_ID_BITS: the box of the conservative ellipse
//            reaches the quadrant). Kept (list position, id) pairs queue up in a 128-entry LDS ring until 64 are waiting.
//   stage    lane e takes queue entry e: its 64-byte record goes to LDS by four global_load_lds_dwordx4 (no VGPRs, no ds_write; the
//            LDS image [chunk][entry] is exactly the structure-of-arrays phase 2 wants), its ellipse stays in five registers.
//   phase 1  Gaussians across the lanes: lane e evaluates its entry's ellipse at the quadrant's 64 pixels; each comparison is a wave
//            ballot and lands, by two v_writelane, in the lane that owns the pixel (quad_ballots): 5 instructions per 64 tests, no
//            per-block lists, no ds_bpermute.
//   phase 2  pixels across the lanes: every pixel walks its own 64-bit pass mask in list order through the reference's recurrence
//            (blend_entry / blend_entry_fast, unchanged); the bit index IS the LDS slot, so the list-byte read of render2 is gone.
// The conservative filters only drop pairs that are a bare `continue` in the reference, so the images are bit-identical to the
// plain transcription within an arithmetic mode (tests/test_raster_forward_gpu.py). LDS: 4 KB of records + 1 KB of queue per wave.

// phase 2 reads the four 16-byte chunks of a record; of chunks 2 and 3 it uses three words (K and the ellipse's c are phase 1's), and
// hipcc narrows those loads to ds_read_b96 -- which takes 8 LDS cycles per wave where ds_read_b128 takes 4 (MI355X_MICROARCH.md, LDS
// table). An empty asm that "uses" the fourth word keeps the loads 16 bytes wide.
#ifndef F3DG_R3_B128
#define F3DG_R3_B128 1
#endif
#if F3DG_R3_B128
#define F3DG_FULL16(a, b) asm volatile("" :: "v"((a).w), "v"((b).w))
#else
#define F3DG_FULL16(a, b) do { } while (0)
#endif
#ifndef F3DG_R3S_PRIO
#define F3DG_R3S_PRIO 0
#endif
#ifndef F3DG_R3S_BREAK
#define F3DG_R3S_BREAK 0
#endif

#define F3DG_R3_WIN 64              // list entries per window = lanes
#define F3DG_R3_RING 128            // queue ring of (list position, id) pairs
#define F3DG_R3_FLAG 0x80000000u    // contributor values of the current window are slots (flag | slot) until the window ends

#ifdef F3DG_LAB      // ---- generation 3 with fixed windows, lab builds only
template <bool SAVE_AUX, bool FAST, bool DMA, int OCC>
__global__ void __launch_bounds__(64, OCC)
render3_fwd_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
                   const F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                   const unsigned* __restrict__ point_list, const F3dgRec* __restrict__ rec,
                   const float4* __restrict__ cull, const float* __restrict__ background, int bg_per_view,
                   float* __restrict__ out_color, float* __restrict__ final_T, unsigned* __restrict__ n_contrib)
{
    unsigned view, unit;                      // the four quadrants of a tile and all tiles of a view share one XCD's L2
    f3dg_xcd_map(blockIdx.x, (unsigned)V, 4u * (unsigned)T, view, unit);
    const unsigned tile = unit >> 2, quad = unit & 3u;
    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    const unsigned lane = threadIdx.x;
    const unsigned qx0 = tile_x * F3DG_TILE + (quad & 1u) * 8u, qy0 = tile_y * F3DG_TILE + (quad >> 1) * 8u;
    const unsigned pix_x = qx0 + (lane & 7u), pix_y = qy0 + (lane >> 3);
    const bool inside = pix_x < (unsigned)W && pix_y < (unsigned)H;
    const size_t HW = (size_t)H * W;
    const size_t pix_id = (size_t)W * pix_y + pix_x;
    const float pixf_x = (float)pix_x + 0.5f, pixf_y = (float)pix_y + 0.5f;
    const float ray_x = (float)((pixf_x - W / 2.) / focal_x);
    const float ray_y = (float)((pixf_y - H / 2.) / focal_y);

    uint2 range = ranges[(size_t)view * T + tile];
    if (hdr->overflow) range = make_uint2(0, 0);
    const unsigned n = range.y - range.x;

    __shared__ float4 sR[4][F3DG_R3_WIN];     // records of the window, [16-byte chunk][entry]: v0..v3 | v4..v7 | v8 v9 opacity K | r g b c
    __shared__ uint2 sQ[F3DG_R3_RING];        // (list position, Gaussian id) of the kept entries, ring

    const F3dgRec* vrec = rec + (size_t)view * P;
    const float4* vcull = cull + (size_t)view * P;
    const unsigned qbit = 1u << (F3DG_ID_BITS + quad);
    const unsigned long long lt = (1ull << lane) - 1ull;

    bool done = !inside;
    PixelState st;
    st.Tr = 1.0f;
    st.last_contributor = 0; st.max_contributor = (unsigned)-1;
    st.C0 = st.C1 = st.C2 = st.C3 = st.C4 = st.C5 = st.C6 = st.C7 = 0;
    st.dist1 = st.dist2 = st.distortion = 0;
    F3DG_T_DECL

    unsigned cursor = 0, qhead = 0, qcount = 0;                       // wave-uniform
    unsigned idn = lane < n ? point_list[range.x + lane] : 0u;       // the 64 list entries at `cursor`, always one chunk ahead
    if (__ballot(!done) != 0ull)
    for (;;) {
        // ---- scan: keep the entries whose box reaches this quadrant
        while (qcount < F3DG_R3_WIN && cursor < n) {
            const unsigned idm = idn, pos = cursor + lane;
            cursor += 64u;
            idn = cursor + lane < n ? point_list[range.x + cursor + lane] : 0u;
            const bool keep = pos < n && (idm & qbit) != 0u;
            const unsigned long long kb = __ballot(keep);
            if (keep) sQ[(qhead + qcount + (unsigned)__popcll(kb & lt)) & (F3DG_R3_RING - 1)] = make_uint2(pos, idm & F3DG_ID_MASK);
            qcount += (unsigned)__popcll(kb);
        }
        if (qcount == 0u)
            break;
        const unsigned m = qcount < F3DG_R3_WIN ? qcount : F3DG_R3_WIN;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        F3DG_T_MARK(2);

        // ---- stage: lane e <- queue entry e
        float4 e4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float ec = 0.0f;
        if (lane < m) {
            const unsigned id = sQ[(qhead + lane) & (F3DG_R3_RING - 1)].y;
            const float4* src = reinterpret_cast<const float4*>(vrec + id);
            if (DMA) {
#pragma unroll
                for (int c = 0; c < 4; c++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c),
                                                     (__attribute__((address_space(3))) void*)&sR[c][0], 16, 0, 0);
                e4 = vcull[id];
            } else {
                const float4 a = src[0], b = src[1], c = src[2], d = src[3];
                e4 = vcull[id];
                sR[0][lane] = a; sR[1][lane] = b; sR[2][lane] = c; sR[3][lane] = d;
                ec = d.w;
            }
        }
        if (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane < m) ec = sR[3][lane].w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        F3DG_T_MARK(1);

        // ---- phase 1: lane e tests entry e against the 64 pixels of the quadrant
        int pass_lo = 0, pass_hi = 0;
        {
            const float u0 = lane < m ? (float)qx0 - e4.x : __builtin_nanf("");     // NaN: every comparison below is false
            const float v0 = (float)qy0 - e4.y;
            float dxx[8], adx[8], dyy[8], cdy[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                dxx[q] = u0 + (float)q;
                adx[q] = e4.z * dxx[q];
                dyy[q] = v0 + (float)q;
                cdy[q] = ec * dyy[q] * dyy[q];
            }
            quad_ballots<0>(pass_lo, pass_hi, fmaf(dxx[0], fmaf(e4.w, dyy[0], adx[0]), cdy[0]), dxx, adx, dyy, cdy, e4.w);
        }
        F3DG_T_MARK(3);
        unsigned long long pass = done ? 0ull : ((unsigned long long)(unsigned)pass_hi << 32) | (unsigned)pass_lo;

        // ---- phase 2: this pixel's own passing entries, in list order, through the reference's arithmetic
        while (pass != 0 && !done) {
            const int j = __builtin_ctzll(pass);
            pass &= pass - 1;
            const float4 q0 = sR[0][j], q1 = sR[1][j], q2 = sR[2][j], q3 = sR[3][j];
            F3DG_FULL16(q2, q3);
            const float n0 = q0.x * ray_x + q0.y * ray_y + q0.z;
            const float n1 = q0.y * ray_x + q0.w * ray_y + q1.x;
            const float n2 = q0.z * ray_x + q1.x * ray_y + q1.y;
            const float aaf = ray_x * n0 + ray_y * n1 + n2;
            const float bhalf = q1.z * ray_x + q1.w * ray_y + q2.x;
            done = (FAST ? blend_entry_fast<> : blend_entry<>)(st, F3DG_R3_FLAG | (unsigned)j, n0, n1, n2, aaf, bhalf, q2.y, q2.z, q3.x, q3.y, q3.z);
        }
        if (SAVE_AUX) {             // slots -> 1-based list positions (the reference's `contributor`)
            if (st.last_contributor - F3DG_R3_FLAG < (unsigned)F3DG_R3_WIN)
                st.last_contributor = sQ[(qhead + (st.last_contributor - F3DG_R3_FLAG)) & (F3DG_R3_RING - 1)].x + 1u;
            if (st.max_contributor - F3DG_R3_FLAG < (unsigned)F3DG_R3_WIN)
                st.max_contributor = sQ[(qhead + (st.max_contributor - F3DG_R3_FLAG)) & (F3DG_R3_RING - 1)].x + 1u;
        }
        F3DG_T_MARK(4);
        qhead += m;
        qcount -= m;
        if (__ballot(!done) == 0ull)
            break;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the window's slots are rewritten by the next one
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    F3DG_T_FLUSH;

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

#endif // F3DG_LAB (render3_fwd_kernel)

// ---- render3 with a SLIDING window (the default) ----------------------------------------------------------------------------------
// With fixed 64-entry windows every lane waits at the end of a window for the lane with the most passing entries: the CPU model
// (tests/tools/wave1_model.py) puts the lane utilisation of phase 2 at 0.54. Here the 64 staged entries are two halves of 32; a slide
// retires the older half -- which every live pixel has finished -- stages 32 new entries in its place and tests them (lanes e and
// e + 32 share entry e and split the quadrant's rows: half_ballots), and phase 2 runs until the now-older half is finished by
// everybody, pixels that are through with it already working on the newer half. Same LDS (4 KB of records), same phase-1 cost per
// entry; the model gives 0.64 (15 % fewer phase-2 trips). Per pixel the sequence of blended entries is unchanged.
//
// TAIL (option render_tail = N > 0): once at most N of the quadrant's 64 pixels are still unsaturated the wave changes its schedule for
// the rest of the list. On pixel-aligned splats over a real depth map a quadrant's last few pixels (depth edges, thin coverage) never
// saturate and walk the whole tile list; the sliding window then pays its fixed cost per 32 entries -- a record gather for every entry
// and 32 two-pixel ballot steps for 64 pixels of which a handful are alive. The tail schedule takes 64 kept entries per step, one per
// lane, gathers only their 20 bytes of ellipse, runs the ellipse test for the LIVE pixels only (a scalar loop over the set bits of the
// live mask: the same two FMAs and comparison, the ballot written to the pixel's lane by v_writelane with the lane in M0), gathers the
// records of the entries some live pixel passes (the OR of the ballots) and lets the live pixels walk their masks through the same
// phase 2. Per pixel the sequence of blended entries and every operation on them is unchanged: bit-identical images.
template <bool SAVE_AUX, bool FAST, int OCC, int WPB, bool NORMAL = true, bool DIST = true, bool COUNT = false, bool TAIL = false>
__global__ void __launch_bounds__(64 * WPB, OCC)
render3s_fwd_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
                    const F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                    const unsigned* __restrict__ point_list, const F3dgRec* __restrict__ rec,
                    const float4* __restrict__ cull, const float* __restrict__ background, int bg_per_view,
                    float* __restrict__ out_color, float* __restrict__ final_T, unsigned* __restrict__ n_contrib, int tail_n)
{
    // WPB = 1: a workgroup is one quadrant's wave. WPB = 4 (option render_wpb): the four quadrant waves of a tile are one workgroup --
    // still no barrier and nothing shared, but they start together on one CU, so the records the second to fourth wave gather are
    // in that CU's L1 / the XCD's L2 already
    unsigned view, unit;
    f3dg_xcd_map(blockIdx.x, (unsigned)V, (4u / (unsigned)WPB) * (unsigned)T, view, unit);
    const unsigned wv = WPB == 1 ? 0u : (threadIdx.x >> 6);
    const unsigned tile = WPB == 4 ? unit : unit >> 2, quad = WPB == 4 ? wv : unit & 3u;
    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned qx0 = tile_x * F3DG_TILE + (quad & 1u) * 8u, qy0 = tile_y * F3DG_TILE + (quad >> 1) * 8u;
    const unsigned pix_x = qx0 + (lane & 7u), pix_y = qy0 + (lane >> 3);
    const bool inside = pix_x < (unsigned)W && pix_y < (unsigned)H;
    const size_t HW = (size_t)H * W;
    const size_t pix_id = (size_t)W * pix_y + pix_x;
    const float pixf_x = (float)pix_x + 0.5f, pixf_y = (float)pix_y + 0.5f;
    const float ray_x = (float)((pixf_x - W / 2.) / focal_x);
    const float ray_y = (float)((pixf_y - H / 2.) / focal_y);

    uint2 range = ranges[(size_t)view * T + tile];
    if (hdr->overflow) range = make_uint2(0, 0);
    const unsigned n = range.y - range.x;

    __shared__ float4 sR_[WPB][4][F3DG_R3_WIN];     // records, [16-byte chunk][slot]; slots 0..31 and 32..63 are the two halves of the window
    __shared__ uint2 sQ_[WPB][F3DG_R3_RING];        // kept (list position, Gaussian id) pairs not staged yet, ring
    __shared__ unsigned sP_[WPB][SAVE_AUX ? F3DG_R3_WIN : 1];   // list position of every staged slot (the reference's `contributor`)
    float4 (*sR)[F3DG_R3_WIN] = sR_[wv];
    uint2* sQ = sQ_[wv];
    unsigned* sP = sP_[wv];

    const F3dgRec* vrec = rec + (size_t)view * P;
    const float4* vcull = cull + (size_t)view * P;
    const unsigned qbit = 1u << (F3DG_ID_BITS + quad);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned hl = lane & 31u;               // entry of a half this lane tests in phase 1 ...
    const unsigned row4 = (lane >> 5) * 4u;       // ... against the pixels of rows row4 .. row4 + 3

    bool done = !inside;
    PixelState st;
    st.Tr = 1.0f;
    st.last_contributor = 0; st.max_contributor = (unsigned)-1;
    st.C0 = st.C1 = st.C2 = st.C3 = st.C4 = st.C5 = st.C6 = st.C7 = 0;
    st.dist1 = st.dist2 = st.distortion = 0;

    auto translate = [&](unsigned half_or_all) {      // slots -> 1-based list positions for the slots of one physical half (2: both)
        if (SAVE_AUX) {
            const unsigned a = st.last_contributor - F3DG_R3_FLAG, b = st.max_contributor - F3DG_R3_FLAG;
            if (a < (unsigned)F3DG_R3_WIN && (half_or_all == 2u || (a >> 5) == half_or_all)) st.last_contributor = sP[a] + 1u;
            if (b < (unsigned)F3DG_R3_WIN && (half_or_all == 2u || (b >> 5) == half_or_all)) st.max_contributor = sP[b] + 1u;
        }
    };

    unsigned n_tail_steps = 0, n_tail_trips = 0, n_tail_tests = 0, n_useful = 0;
    bool go_tail = false;
    unsigned n_half_sep = 0, n_half_pair = 0;
    unsigned n_staged = 0, n_trips = 0, n_wave_trips = 0, n_slides = 0, n_t8 = 0, n_t24 = 0, n_s8 = 0, n_s24 = 0;    // COUNT (option render_count): what this wave did, summed into g_f3dg_counts at its end
    unsigned cursor = 0, qhead = 0, qpend = 0;    // wave-uniform: scan position, ring index of the first pending entry, pending entries
    unsigned flip = 0;                            // physical half (slots 32 flip ..) that holds the OLDER half of the window
    unsigned long long pass = 0ull;               // per pixel: bits 0..31 older half, 32..63 newer half, in list order
    unsigned idn = lane < n ? point_list[range.x + lane] : 0u;
    if (__ballot(!done) != 0ull)
    for (;;) {
#if F3DG_R3S_PRIO
        __builtin_amdgcn_s_setprio(F3DG_R3S_PRIO);     // scan + staging are chains of memory latencies: their loads should leave first
#endif
        // ---- scan: keep the entries whose box reaches this quadrant until 32 are pending
        while (qpend < 32u && cursor < n) {
            const unsigned idm = idn, pos = cursor + lane;
            cursor += 64u;
            idn = cursor + lane < n ? point_list[range.x + cursor + lane] : 0u;
            const bool keep = pos < n && (idm & qbit) != 0u;
            const unsigned long long kb = __ballot(keep);
            if (keep) sQ[(qhead + qpend + (unsigned)__popcll(kb & lt)) & (F3DG_R3_RING - 1)] = make_uint2(pos, idm & F3DG_ID_MASK);
            qpend += (unsigned)__popcll(kb);
        }
        const unsigned m = qpend < 32u ? qpend : 32u;
        // every live pixel has finished the older half (bits 0..31 of `pass` are clear): retire it
        translate(flip);
        if (m == 0u && __ballot(pass != 0ull) == 0ull)
            break;                                // nothing left to stage, nothing left in the newer half
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- stage m entries into the retired half; lanes e and e + 32 both take entry e
        const unsigned base = flip * 32u;
        float4 e4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float ec = 0.0f;
        if (hl < m) {
            const uint2 q = sQ[(qhead + hl) & (F3DG_R3_RING - 1)];
            if (lane < 32u) {
                const float4* src = reinterpret_cast<const float4*>(vrec + q.y);
#pragma unroll
                for (int c = 0; c < 4; c++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c),
                                                     (__attribute__((address_space(3))) void*)&sR[c][base], 16, 0, 0);
                if (SAVE_AUX) sP[base + lane] = q.x;
            }
            e4 = vcull[q.y];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (hl < m) ec = sR[3][base + hl].w;
        qhead += m;
        qpend -= m;
        if (COUNT) { n_staged += m; n_slides++; }
#if F3DG_R3S_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif

        // ---- phase 1: the 32 new entries against the quadrant's 64 pixels
        int fresh = 0;
        if (m != 0u) {
            const float u0 = hl < m ? (float)qx0 - e4.x : __builtin_nanf("");     // NaN: every comparison below is false
            const float v0 = (float)(qy0 + row4) - e4.y;
            float dxx[8], adx[8], dyy[4], cdy[4];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                dxx[q] = u0 + (float)q;
                adx[q] = e4.z * dxx[q];
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                dyy[q] = v0 + (float)q;
                cdy[q] = ec * dyy[q] * dyy[q];
            }
            half_ballots<0>(fresh, fmaf(dxx[0], fmaf(e4.w, dyy[0], adx[0]), cdy[0]), dxx, adx, dyy, cdy, e4.w);
        }
        if (COUNT) {        // staged entries that reach at least one pixel that is still alive (the others were gathered for nothing)
            unsigned u = done ? 0u : (unsigned)fresh;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) u |= (unsigned)__shfl_xor((int)u, o, 64);
            n_useful += (unsigned)__popc(u);
        }
        // ---- slide: the newer half becomes the older one, the fresh bits the newer one
        pass = (pass >> 32) | (done ? 0ull : ((unsigned long long)(unsigned)fresh << 32));
        flip ^= 1u;
        const unsigned xr = flip << 5;            // logical slot j (0..31 older, 32..63 newer) lives in physical slot j ^ xr

        // ---- phase 2: until every live pixel has finished the older half; pixels that have go on with the newer one
        // (a divergent loop: a pixel leaves it when its mask is empty -- it has nothing left in either half -- and the ballot, taken
        // over the pixels still inside, ends it for everybody once no older-half bit is left)
        const unsigned trips_before = n_trips;
        const unsigned live_now = COUNT ? (unsigned)__popcll(__ballot(!done)) : 0u;
        while (pass != 0ull && __ballot((unsigned)pass != 0u) != 0ull) {
            const unsigned j = (unsigned)__builtin_ctzll(pass) ^ xr;
            pass &= pass - 1;
            if (COUNT) n_trips++;
            const float4 q0 = sR[0][j], q1 = sR[1][j], q2 = sR[2][j], q3 = sR[3][j];
            F3DG_FULL16(q2, q3);
            const float n0 = q0.x * ray_x + q0.y * ray_y + q0.z;
            const float n1 = q0.y * ray_x + q0.w * ray_y + q1.x;
            const float n2 = q0.z * ray_x + q1.x * ray_y + q1.y;
            const float aaf = ray_x * n0 + ray_y * n1 + n2;
            const float bhalf = q1.z * ray_x + q1.w * ray_y + q2.x;
            done = (FAST ? blend_entry_fast<NORMAL, DIST> : blend_entry<NORMAL, DIST>)(st, F3DG_R3_FLAG | j, n0, n1, n2, aaf, bhalf, q2.y, q2.z, q3.x, q3.y, q3.z);
#if F3DG_R3S_BREAK
            if (done) break;              // a saturated pixel leaves the loop (its mask is cleared once, below, not on every trip)
#else
            if (done) pass = 0ull;
#endif
        }
#if F3DG_R3S_BREAK
        if (done) pass = 0ull;
#endif
        if (COUNT) {                // the loop ran as often as its busiest lane needed (lanes leave it, none re-enters)
            unsigned t = n_trips - trips_before;
            {
                // what TWO pixels per lane would buy, emulated at half scale: the wave's two 8 x 4 halves as separate 32-lane walks
                // (each lasts as long as its busiest pixel) against one 32-lane walk whose lane i takes pixel i and then pixel i + 32
                unsigned th = t, tp = t + (unsigned)__shfl_xor((int)t, 32, 64);
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) {
                    th = max(th, (unsigned)__shfl_xor((int)th, o, 64));
                    tp = max(tp, (unsigned)__shfl_xor((int)tp, o, 64));
                }
                n_half_sep += th + (unsigned)__shfl_xor((int)th, 32, 64);
                n_half_pair += tp;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t = max(t, (unsigned)__shfl_xor((int)t, o, 64));
            n_wave_trips += t;
            if (live_now <= 8u) { n_t8 += t; n_s8++; }
            if (live_now <= 24u) { n_t24 += t; n_s24++; }
        }
        const unsigned long long live = __ballot(!done);
        if (live == 0ull)
            break;
        if (TAIL && __popcll(live) <= tail_n) {
            go_tail = true;
            break;
        }
    }
    if (TAIL && go_tail) {
        // `pass` still holds the pending bits of the newer half (logical slots 32..63, physical slot = logical ^ xr): the first trip
        // round of the loop below finishes them; from the second round on the 64 slots are one window, slot = bit
        unsigned xr = flip << 5;
        for (;;) {
            const unsigned trips_before = n_trips;
            while (pass != 0ull) {
                const unsigned j = (unsigned)__builtin_ctzll(pass) ^ xr;
                pass &= pass - 1;
                if (COUNT) n_trips++;
                const float4 q0 = sR[0][j], q1 = sR[1][j], q2 = sR[2][j], q3 = sR[3][j];
                F3DG_FULL16(q2, q3);
                const float n0 = q0.x * ray_x + q0.y * ray_y + q0.z;
                const float n1 = q0.y * ray_x + q0.w * ray_y + q1.x;
                const float n2 = q0.z * ray_x + q1.x * ray_y + q1.y;
                const float aaf = ray_x * n0 + ray_y * n1 + n2;
                const float bhalf = q1.z * ray_x + q1.w * ray_y + q2.x;
                done = (FAST ? blend_entry_fast<NORMAL, DIST> : blend_entry<NORMAL, DIST>)(st, F3DG_R3_FLAG | j, n0, n1, n2, aaf, bhalf, q2.y, q2.z, q3.x, q3.y, q3.z);
                if (done) pass = 0ull;
            }
            if (COUNT) {
                unsigned t = n_trips - trips_before;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) t = max(t, (unsigned)__shfl_xor((int)t, o, 64));
                n_wave_trips += t;
                n_tail_trips += t;
            }
            translate(2u);
            const unsigned long long live = __ballot(!done);
            if (live == 0ull)
                break;
            // ---- scan: as above, until 64 are pending
            while (qpend < 64u && cursor < n) {
                const unsigned idm = idn, pos = cursor + lane;
                cursor += 64u;
                idn = cursor + lane < n ? point_list[range.x + cursor + lane] : 0u;
                const bool keep = pos < n && (idm & qbit) != 0u;
                const unsigned long long kb = __ballot(keep);
                if (keep) sQ[(qhead + qpend + (unsigned)__popcll(kb & lt)) & (F3DG_R3_RING - 1)] = make_uint2(pos, idm & F3DG_ID_MASK);
                qpend += (unsigned)__popcll(kb);
            }
            const unsigned m = qpend < 64u ? qpend : 64u;
            if (m == 0u)
                break;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- lane e <- entry e: its ellipse only (16 bytes + the c of record slot 15)
            const bool have = lane < m;
            uint2 q = make_uint2(0u, 0u);
            float4 e4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            float ec = 0.0f;
            if (have) {
                q = sQ[(qhead + lane) & (F3DG_R3_RING - 1)];
                e4 = vcull[q.y];
                ec = reinterpret_cast<const float*>(vrec + q.y)[15];
            }
            qhead += m;
            qpend -= m;
            // ---- phase 1 for the live pixels only (the arithmetic of quad_ballots: dx = (qx0 - cx) + column, dy = (qy0 - cy) + row)
            const float u0 = have ? (float)qx0 - e4.x : __builtin_nanf("");
            const float v0 = (float)qy0 - e4.y;
            int lo = 0, hi = 0;
            unsigned long long any = 0ull, lv = live;
            while (lv != 0ull) {
                const int p = __builtin_ctzll(lv);
                lv &= lv - 1;
                const float dx = u0 + (float)(p & 7), dy = v0 + (float)(p >> 3);
                const float adx = e4.z * dx, cdy = ec * dy * dy;
                const float E = fmaf(dx, fmaf(e4.w, dy, adx), cdy);
                // (the comparison IS the ballot; M0 selects the lane; two wait states between the VALU write of VCC and its VALU read)
                asm volatile("v_cmp_ge_f32 vcc, 1.0, %[e]\n\t"
                             "s_mov_b32 m0, %[p]\n\t"
                             "s_nop 1\n\t"
                             "v_writelane_b32 %[lo], vcc_lo, m0\n\t"
                             "v_writelane_b32 %[hi], vcc_hi, m0\n\t"
                             "s_or_b64 %[any], %[any], vcc"
                             : [lo] "+v"(lo), [hi] "+v"(hi), [any] "+s"(any)
                             : [e] "v"(E), [p] "s"(p)
                             : "vcc", "scc", "m0");
            }
            // ---- records of the entries some live pixel passes, slot = lane
            if ((any >> lane) & 1ull) {
                const float4* src = reinterpret_cast<const float4*>(vrec + q.y);
#pragma unroll
                for (int c = 0; c < 4; c++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c),
                                                     (__attribute__((address_space(3))) void*)&sR[c][0], 16, 0, 0);
                if (SAVE_AUX) sP[lane] = q.x;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (COUNT) { n_staged += (unsigned)__popcll(any); n_tail_steps++; n_tail_tests += m; }
            pass = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
            xr = 0u;
        }
    }
    translate(2u);
#ifdef F3DG_LAB
    if (COUNT && lane == 0 && blockIdx.x < F3DG_SLIDE_LOG_N) g_f3dg_slide_log[blockIdx.x] = n_slides;
#endif
    if (COUNT && lane == 0) {
        unsigned long long* c = g_f3dg_counts[blockIdx.x & 63u];
        atomicAdd(&c[10], (unsigned long long)n_tail_steps);
        atomicAdd(&c[11], (unsigned long long)n_tail_trips);
        atomicAdd(&c[12], (unsigned long long)n_tail_tests);
        atomicAdd(&c[13], (unsigned long long)n_useful);
        atomicAdd(&c[14], (unsigned long long)n_half_sep);
        atomicAdd(&c[15], (unsigned long long)n_half_pair);
        atomicAdd(&c[0], (unsigned long long)n_staged);
        atomicAdd(&c[1], (unsigned long long)(cursor < n ? cursor : n));
        atomicAdd(&c[2], (unsigned long long)n_wave_trips);
        atomicAdd(&c[3], (unsigned long long)n_slides);
        atomicAdd(&c[5], 1ull);
        atomicAdd(&c[6], (unsigned long long)n_t8);
        atomicAdd(&c[7], (unsigned long long)n_t24);
        atomicAdd(&c[8], (unsigned long long)n_s8);
        atomicAdd(&c[9], (unsigned long long)n_s24);
    }
    if (COUNT) {
        unsigned t = n_trips;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += (unsigned)__shfl_xor((int)t, o, 64);
        if (lane == 0) atomicAdd(&g_f3dg_counts[blockIdx.x & 63u][4], (unsigned long long)t);
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
        if (NORMAL) {
            out[3 * HW + pix_id] = st.C3;
            out[4 * HW + pix_id] = st.C4;
            out[5 * HW + pix_id] = st.C5;
        }
        out[6 * HW + pix_id] = st.C6;
        out[7 * HW + pix_id] = st.C7;
        if (DIST) out[8 * HW + pix_id] = distortion;
    }
}

#ifdef F3DG_LAB
// ---- lab: the staging half of a render3s launch, replayed ----------------------------------------------------------------------------
// Every quadrant wave repeats the list scan, the record gathers (global_load_lds) and phase 1 of exactly the slides the logged launch
// performed (g_f3dg_slide_log) and never enters phase 2: what the launch costs as a stream of list reads, 64-byte gathers and ellipse
// ballots, with the same addresses in the same order. The frames it writes are garbage (the XOR of the pass masks keeps the work alive).
__global__ void __launch_bounds__(64, 8)
render3s_stage_only_kernel(int V, int P, int W, int H, int tiles_x, int T, const F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                           const unsigned* __restrict__ point_list, const F3dgRec* __restrict__ rec, const float4* __restrict__ cull,
                           float* __restrict__ out_color, int gather_records)
{
    unsigned view, unit;
    f3dg_xcd_map(blockIdx.x, (unsigned)V, 4u * (unsigned)T, view, unit);
    const unsigned tile = unit >> 2, quad = unit & 3u;
    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned qx0 = tile_x * F3DG_TILE + (quad & 1u) * 8u, qy0 = tile_y * F3DG_TILE + (quad >> 1) * 8u;
    uint2 range = ranges[(size_t)view * T + tile];
    if (hdr->overflow) range = make_uint2(0, 0);
    const unsigned n = range.y - range.x;
    __shared__ float4 sR[4][F3DG_R3_WIN];
    __shared__ uint2 sQ[F3DG_R3_RING];
    const F3dgRec* vrec = rec + (size_t)view * P;
    const float4* vcull = cull + (size_t)view * P;
    const unsigned qbit = 1u << (F3DG_ID_BITS + quad);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned hl = lane & 31u, row4 = (lane >> 5) * 4u;
    const unsigned slides = blockIdx.x < F3DG_SLIDE_LOG_N ? g_f3dg_slide_log[blockIdx.x] : 0u;
    unsigned cursor = 0, qhead = 0, qpend = 0, flip = 0;
    int acc = 0;
    unsigned idn = lane < n ? point_list[range.x + lane] : 0u;
    for (unsigned sl = 0; sl < slides; sl++) {
        while (qpend < 32u && cursor < n) {
            const unsigned idm = idn, pos = cursor + lane;
            cursor += 64u;
            idn = cursor + lane < n ? point_list[range.x + cursor + lane] : 0u;
            const bool keep = pos < n && (idm & qbit) != 0u;
            const unsigned long long kb = __ballot(keep);
            if (keep) sQ[(qhead + qpend + (unsigned)__popcll(kb & lt)) & (F3DG_R3_RING - 1)] = make_uint2(pos, idm & F3DG_ID_MASK);
            qpend += (unsigned)__popcll(kb);
        }
        const unsigned m = qpend < 32u ? qpend : 32u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned base = flip * 32u;
        float4 e4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float ec = 0.0f;
        if (hl < m) {
            const uint2 q = sQ[(qhead + hl) & (F3DG_R3_RING - 1)];
            if (lane < 32u && gather_records) {
                const float4* src = reinterpret_cast<const float4*>(vrec + q.y);
#pragma unroll
                for (int c = 0; c < 4; c++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c),
                                                     (__attribute__((address_space(3))) void*)&sR[c][base], 16, 0, 0);
            }
            e4 = vcull[q.y];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (hl < m) ec = sR[3][base + hl].w;
        qhead += m;
        qpend -= m;
        int fresh = 0;
        if (m != 0u) {
            const float u0 = hl < m ? (float)qx0 - e4.x : __builtin_nanf("");
            const float v0 = (float)(qy0 + row4) - e4.y;
            float dxx[8], adx[8], dyy[4], cdy[4];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                dxx[q] = u0 + (float)q;
                adx[q] = e4.z * dxx[q];
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                dyy[q] = v0 + (float)q;
                cdy[q] = ec * dyy[q] * dyy[q];
            }
            half_ballots<0>(fresh, fmaf(dxx[0], fmaf(e4.w, dyy[0], adx[0]), cdy[0]), dxx, adx, dyy, cdy, e4.w);
        }
        acc ^= fresh;
        flip ^= 1u;
    }
    const unsigned pix_x = qx0 + (lane & 7u), pix_y = qy0 + (lane >> 3);
    if (pix_x < (unsigned)W && pix_y < (unsigned)H)
        out_color[(size_t)view * F3DG_OUT_CHANNELS * H * W + (size_t)W * pix_y + pix_x] = __int_as_float(acc & 0x3fffff);
}
#endif

#ifdef F3DG_LAB      // ---- the one-wave kernel for small launches, lab builds only (superseded by render3p / render3q)
// ---- render3 for SMALL launches: the next window's gathers in flight behind phase 2 ------------------------------------------------
// A call of one or two 256^2 views is 1,024-2,048 waves on a chip that holds 8,192: every wave is alone on its SIMD and its time is a
// chain of latencies -- list ids, record gathers, the dependent instructions of a phase-2 trip -- that no other wave fills. At that
// occupancy LDS is free, so this variant keeps TWO 64-entry windows of records: window k + 1 is scanned for and its records are
// requested (global_load_lds) right after phase 1 of window k, and they land while phase 2 of window k runs; two 64-id chunks of the
// list are always in flight. Fixed windows (lane utilisation is irrelevant for a latency-bound wave). Same images to the bit.
#define F3DG_R3L_RING 256
template <bool SAVE_AUX, bool FAST>
__global__ void __launch_bounds__(64, 2)
render3l_fwd_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
                    const F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                    const unsigned* __restrict__ point_list, const F3dgRec* __restrict__ rec,
                    const float4* __restrict__ cull, const float* __restrict__ background, int bg_per_view,
                    float* __restrict__ out_color, float* __restrict__ final_T, unsigned* __restrict__ n_contrib)
{
    unsigned view, unit;
    f3dg_xcd_map(blockIdx.x, (unsigned)V, 4u * (unsigned)T, view, unit);
    const unsigned tile = unit >> 2, quad = unit & 3u;
    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    const unsigned lane = threadIdx.x;
    const unsigned qx0 = tile_x * F3DG_TILE + (quad & 1u) * 8u, qy0 = tile_y * F3DG_TILE + (quad >> 1) * 8u;
    const unsigned pix_x = qx0 + (lane & 7u), pix_y = qy0 + (lane >> 3);
    const bool inside = pix_x < (unsigned)W && pix_y < (unsigned)H;
    const size_t HW = (size_t)H * W;
    const size_t pix_id = (size_t)W * pix_y + pix_x;
    const float pixf_x = (float)pix_x + 0.5f, pixf_y = (float)pix_y + 0.5f;
    const float ray_x = (float)((pixf_x - W / 2.) / focal_x);
    const float ray_y = (float)((pixf_y - H / 2.) / focal_y);

    uint2 range = ranges[(size_t)view * T + tile];
    if (hdr->overflow) range = make_uint2(0, 0);
    const unsigned n = range.y - range.x;

    __shared__ float4 sR[2][4][F3DG_R3_WIN];  // two windows of records, [window][16-byte chunk][entry]
    __shared__ uint2 sQ[F3DG_R3L_RING];       // (list position, Gaussian id) of the kept entries: the current window, the next one, the backlog

    const F3dgRec* vrec = rec + (size_t)view * P;
    const float4* vcull = cull + (size_t)view * P;
    const unsigned qbit = 1u << (F3DG_ID_BITS + quad);
    const unsigned long long lt = (1ull << lane) - 1ull;

    bool done = !inside;
    PixelState st;
    st.Tr = 1.0f;
    st.last_contributor = 0; st.max_contributor = (unsigned)-1;
    st.C0 = st.C1 = st.C2 = st.C3 = st.C4 = st.C5 = st.C6 = st.C7 = 0;
    st.dist1 = st.dist2 = st.distortion = 0;

    unsigned cursor = 0, qhead = 0, qcount = 0;                       // ring: [qhead, qhead + qcount) = current window + everything behind it
    unsigned id0 = lane < n ? point_list[range.x + lane] : 0u;       // the 64 list entries at `cursor` ...
    unsigned id1 = 64u + lane < n ? point_list[range.x + 64u + lane] : 0u;   // ... and the 64 after them, always in flight
    auto scan_until = [&](unsigned want) {      // keep scanning until `want` entries are queued (or the list ends)
        while (qcount < want && cursor < n) {
            const unsigned idm = id0, pos = cursor + lane;
            cursor += 64u;
            id0 = id1;
            id1 = cursor + 64u + lane < n ? point_list[range.x + cursor + 64u + lane] : 0u;
            const bool keep = pos < n && (idm & qbit) != 0u;
            const unsigned long long kb = __ballot(keep);
            if (keep) sQ[(qhead + qcount + (unsigned)__popcll(kb & lt)) & (F3DG_R3L_RING - 1)] = make_uint2(pos, idm & F3DG_ID_MASK);
            qcount += (unsigned)__popcll(kb);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto request = [&](unsigned buf, unsigned first, unsigned m, float4& e4) {   // records of ring entries [first, first + m) -> window `buf`
        e4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (lane < m) {
            const unsigned id = sQ[(first + lane) & (F3DG_R3L_RING - 1)].y;
            const float4* src = reinterpret_cast<const float4*>(vrec + id);
#pragma unroll
            for (int c = 0; c < 4; c++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c),
                                                 (__attribute__((address_space(3))) void*)&sR[buf][c][0], 16, 0, 0);
            e4 = vcull[id];
        }
    };

    if (__ballot(!done) != 0ull) {
        scan_until(F3DG_R3_WIN);
        unsigned m = qcount < F3DG_R3_WIN ? qcount : F3DG_R3_WIN, buf = 0;
        float4 e4;
        request(0, qhead, m, e4);
        while (m != 0u) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const float ec = lane < m ? sR[buf][3][lane].w : 0.0f;
            // ---- phase 1 of the current window
            int pass_lo = 0, pass_hi = 0;
            {
                const float u0 = lane < m ? (float)qx0 - e4.x : __builtin_nanf("");
                const float v0 = (float)qy0 - e4.y;
                float dxx[8], adx[8], dyy[8], cdy[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    dxx[q] = u0 + (float)q;
                    adx[q] = e4.z * dxx[q];
                    dyy[q] = v0 + (float)q;
                    cdy[q] = ec * dyy[q] * dyy[q];
                }
                quad_ballots<0>(pass_lo, pass_hi, fmaf(dxx[0], fmaf(e4.w, dyy[0], adx[0]), cdy[0]), dxx, adx, dyy, cdy, e4.w);
            }
            // ---- the next window: scan for it and request its records; they land during phase 2
            scan_until(m + F3DG_R3_WIN);
            const unsigned m_next = qcount - m < F3DG_R3_WIN ? qcount - m : F3DG_R3_WIN;
            float4 e4n;
            request(buf ^ 1u, qhead + m, m_next, e4n);

            // ---- phase 2 of the current window
            unsigned long long pass = done ? 0ull : ((unsigned long long)(unsigned)pass_hi << 32) | (unsigned)pass_lo;
            while (pass != 0 && !done) {
                const int j = __builtin_ctzll(pass);
                pass &= pass - 1;
                const float4 q0 = sR[buf][0][j], q1 = sR[buf][1][j], q2 = sR[buf][2][j], q3 = sR[buf][3][j];
                F3DG_FULL16(q2, q3);
                const float n0 = q0.x * ray_x + q0.y * ray_y + q0.z;
                const float n1 = q0.y * ray_x + q0.w * ray_y + q1.x;
                const float n2 = q0.z * ray_x + q1.x * ray_y + q1.y;
                const float aaf = ray_x * n0 + ray_y * n1 + n2;
                const float bhalf = q1.z * ray_x + q1.w * ray_y + q2.x;
                done = (FAST ? blend_entry_fast<> : blend_entry<>)(st, F3DG_R3_FLAG | (unsigned)j, n0, n1, n2, aaf, bhalf, q2.y, q2.z, q3.x, q3.y, q3.z);
            }
            if (SAVE_AUX) {             // slots -> 1-based list positions (the reference's `contributor`)
                if (st.last_contributor - F3DG_R3_FLAG < (unsigned)F3DG_R3_WIN)
                    st.last_contributor = sQ[(qhead + (st.last_contributor - F3DG_R3_FLAG)) & (F3DG_R3L_RING - 1)].x + 1u;
                if (st.max_contributor - F3DG_R3_FLAG < (unsigned)F3DG_R3_WIN)
                    st.max_contributor = sQ[(qhead + (st.max_contributor - F3DG_R3_FLAG)) & (F3DG_R3L_RING - 1)].x + 1u;
            }
            qhead += m;
            qcount -= m;
            m = m_next;
            e4 = e4n;
            buf ^= 1u;
            if (__ballot(!done) == 0ull)
                break;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS write of this wave may still be in flight when it ends
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

#endif // F3DG_LAB (render3l_fwd_kernel)

} // namespace

namespace {
thread_local char g_kernel_name[160] = "";       // (per host thread: the library is called from several)
void note_kernel(const char* base, int save_aux, int fast, const char* extra)
{
    snprintf(g_kernel_name, sizeof g_kernel_name, "%s<SAVE_AUX=%s, FAST=%s%s>", base, save_aux ? "true" : "false", fast ? "true" : "false", extra);
    g_f3dg_last_render_kernel = g_kernel_name;
}
} // namespace

int f3dg_render_uses_fast(int save_aux) { return g_f3dg_render_fast == 2 || (g_f3dg_render_fast == 1 && !save_aux); }

namespace {

// what every compositing launch is handed
struct RenderArgs {
    hipStream_t s;
    int V, P, W, H, tiles_x, T;
    float focal_x, focal_y;
    const F3dgHeader* hdr;
    const uint2* ranges;
    const unsigned* point_list;
    const F3dgRec* rec;
    const float4* cull;
    const float* background;
    int bg_per_view;
    float* out_color;
    float* final_T;
    unsigned* n_contrib;
};

// render3s_fwd_kernel, one quadrant wave per workgroup, 8 waves per SIMD (every variant fits 64 VGPRs and 5 KB of LDS)
template <bool AUX, bool FAST, bool NORMAL, bool DIST, bool COUNT>
void launch3s(const RenderArgs& a)
{
    F3DG_KLAUNCH((render3s_fwd_kernel<AUX, FAST, 8, 1, NORMAL, DIST, COUNT, false>), dim3((unsigned)a.V * (unsigned)a.T * 4u), dim3(64), 0, a.s,
                 a.V, a.P, a.W, a.H, a.tiles_x, a.T, a.focal_x, a.focal_y, a.hdr, a.ranges, a.point_list, a.rec, a.cull, a.background, a.bg_per_view,
                 a.out_color, a.final_T, a.n_contrib, 0);
}

} // namespace

#ifdef F3DG_LAB
// lab builds: the launches only an option of the lab reaches (kernel generations 1-3, render3l, the tail schedule, four-wave workgroups,
// LDS padding, the staging replay). Returns false when the launch is the default dispatch's.
static bool f3dg_launch_render_lab(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y,
                                   const F3dgHeader* hdr, const uint2* ranges, const unsigned* point_list, const F3dgRec* rec,
                                   const float4* bbox, const float4* cull, const float* background, int bg_per_view, float* out_color,
                                   float* final_T, unsigned* n_contrib, int save_aux, unsigned skip_channels, int fast, int* rc_out);
#endif

// The compositing forward of a call. The arithmetic is the CALL's (f3dg_forward_sets resolves its flags against the process default and
// records the choice in the workspace header for the backward): fast arithmetic is for inference calls -- a SAVE_AUX forward feeds
// f3dg_backward, which rebuilds every pixel's transmittance back to front by dividing final_T by (1 - alpha) with ITS alphas: they must be
// the forward's to the bit, or the 1e-6 relative difference is amplified by 1 / (1 - alpha) per layer (measured at C5: compositing-stage
// gradients 2.5e-5 instead of 1.8e-6 off the oracle).
int f3dg_launch_render(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y,
                       const F3dgHeader* hdr, const uint2* ranges, const unsigned* point_list, const F3dgRec* rec,
                       const float4* bbox, const float4* cull, const float* background, int bg_per_view, float* out_color,
                       float* final_T, unsigned* n_contrib, int save_aux, unsigned skip_channels, int fast_arg, int scan)
{
    const int tiles_x = (W + F3DG_TILE - 1) / F3DG_TILE, tiles_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = tiles_x * tiles_y;
    const int fast = fast_arg < 0 ? f3dg_render_uses_fast(save_aux) : fast_arg;
    const long long waves = (long long)V * T * 4;           // one wave per 8 x 8 pixel quadrant
#ifdef F3DG_LAB
    {
        int rc = F3DG_OK;
        if (f3dg_launch_render_lab(s, V, P, W, H, focal_x, focal_y, hdr, ranges, point_list, rec, bbox, cull, background, bg_per_view, out_color,
                                   final_T, n_contrib, save_aux, skip_channels, fast, &rc))
            return rc;
    }
#else
    (void)bbox;
#endif
    // (1) small launches -- at most two waves per SIMD: one or two 256^2 views -- are latency chains nobody fills
    const bool small_launch = g_f3dg_render_lowocc && waves <= (g_f3dg_render_lowocc > 1 ? 1024ll * g_f3dg_render_lowocc : 2048ll);
    // (2) the split-pixel schedule of f3dg_render5.hip: fast inference launches that ask for it (F3DG_FLAG_SCAN, whatever their size) or,
    // with option render_scan 1, every such launch that is not small
    if (fast && !save_aux && g_f3dg_render_scan != 0 && (scan || g_f3dg_render_scan == 1)) {
        if (small_launch)       // one or two views: four lanes per pixel instead of helper-lane batches (render5p_fwd_kernel)
            return f3dg_launch_render5_small(s, V, P, W, H, focal_x, focal_y, hdr, ranges, point_list, rec, cull, background, bg_per_view, out_color);
        return f3dg_launch_render5(s, V, P, W, H, focal_x, focal_y, hdr, ranges, point_list, rec, cull, background, bg_per_view, out_color,
                                   skip_channels, g_f3dg_render_count);
    }
    if (small_launch) {
        // the multi-wave kernels of f3dg_render4.hip. Defaults by measurement at 65,536 pixel-ordered Gaussians (profiles/r05_final/
        // one_view.md): fast arithmetic -- producer + consumer waves, two entries per trip (render3p, 76.6 -> 57.9 us; two views 79.7 ->
        // 63.6 us per call); the reference's arithmetic -- consumer + three evaluator waves + producer for one view (render3q, 134 -> 100 us:
        // its LDS does not fit two quadrants per SIMD), render3p for two
        const bool one_view = waves <= 1024ll;
        const int split = g_f3dg_render_split >= 1 ? g_f3dg_render_split : fast ? 1 : one_view ? 3 : 1;
        const int unroll = g_f3dg_render_unroll >= 1 ? g_f3dg_render_unroll : fast ? 2 : 1;
        return f3dg_launch_render_small(s, V, P, W, H, focal_x, focal_y, hdr, ranges, point_list, rec, cull, background, bg_per_view, out_color,
                                        fast, save_aux, final_T, n_contrib, unroll, split, g_f3dg_render_count);
    }
    // (3) the rank-packed kernel of f3dg_render4.hip (option render_pack: 1 = every launch, -1 = the default: launches in the reference's
    // arithmetic, whose stateless part is 2.5 x as long -- measured -38 % on the real merged set, -6 % at C2; in fast arithmetic the packed
    // trips' hand-over costs what they save: 8.4-8.7 against 8.6 ms). A SAVE_AUX forward in the reference's
    // arithmetic takes it too (its auxiliary planes are bit-identical to render3s's).
    if (g_f3dg_render_pack == 1 || (g_f3dg_render_pack < 0 && !fast))
        return f3dg_launch_render4(s, V, P, W, H, focal_x, focal_y, hdr, ranges, point_list, rec, cull, background, bg_per_view, out_color,
                                   fast, skip_channels, g_f3dg_render_count, save_aux, final_T, n_contrib);
    // (4) render3s_fwd_kernel: one wave per quadrant, sliding half-windows
    const RenderArgs a = { s, V, P, W, H, tiles_x, T, focal_x, focal_y, hdr, ranges, point_list, rec, cull, background, bg_per_view, out_color, final_T, n_contrib };
    // the batched loops of the build that consume RGB, depth and alpha only (cycle aggregation, orbit frames) skip the normal and
    // distortion accumulators: the channels they do write are bit-identical
    const bool lean = !save_aux && (skip_channels & (F3DG_FLAG_SKIP_NORMAL | F3DG_FLAG_SKIP_DISTORTION)) == (F3DG_FLAG_SKIP_NORMAL | F3DG_FLAG_SKIP_DISTORTION);
    const bool count = g_f3dg_render_count && !save_aux;        // (diagnostic: the same kernel with its work counters on)
    const char* extra = ", OCC=8, WPB=1";
    if (count) {
        if (fast) launch3s<false, true, true, true, true>(a); else launch3s<false, false, true, true, true>(a);
        extra = ", OCC=8, WPB=1, COUNT=true";
    } else if (lean) {
        if (fast) launch3s<false, true, false, false, false>(a); else launch3s<false, false, false, false, false>(a);
        extra = ", OCC=8, WPB=1, NORMAL=false, DIST=false";
    } else if (save_aux) {
        if (fast) launch3s<true, true, true, true, false>(a); else launch3s<true, false, true, true, false>(a);
    } else {
        if (fast) launch3s<false, true, true, true, false>(a); else launch3s<false, false, true, true, false>(a);
    }
    note_kernel("render3s_fwd_kernel", lean || count ? 0 : save_aux, fast, extra);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

#ifdef F3DG_LAB
// ---- lab builds only: the dispatch of rounds 1-5 for the launches a lab option selects ------------------------------------------------
static int f3dg_launch_render_lab_impl(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y,
                       const F3dgHeader* hdr, const uint2* ranges, const unsigned* point_list, const F3dgRec* rec,
                       const float4* bbox, const float4* cull, const float* background, int bg_per_view, float* out_color,
                       float* final_T, unsigned* n_contrib, int save_aux, unsigned skip_channels, int fast, int scan)
{
    const int tiles_x = (W + F3DG_TILE - 1) / F3DG_TILE, tiles_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = tiles_x * tiles_y;
    dim3 grid((unsigned)V * (unsigned)T);
    // fast arithmetic is for inference calls. A SAVE_AUX forward feeds f3dg_backward, which rebuilds every pixel's transmittance
    // back to front by dividing final_T by (1 - alpha) with ITS alphas: they must be the forward's to the bit, or the 1e-6 relative
    // difference is amplified by 1 / (1 - alpha) per layer (measured at C5: compositing-stage gradients 2.5e-5 vs 1.8e-6 off the oracle).
    // (the arithmetic is the CALL's: f3dg_forward_sets resolves its flags against the process default and records the choice in the
    // workspace header for the backward)
    const int g_f3dg_render_fast = fast < 0 ? f3dg_render_uses_fast(save_aux) : fast;
    if (g_f3dg_render_kernel == 3) {
        const dim3 grid3((unsigned)V * (unsigned)T * 4u);
#define F3DG_LAUNCH3D(AUX, FST, DMA, OCC) F3DG_KLAUNCH((render3_fwd_kernel<AUX, FST, DMA, OCC>), grid3, dim3(64), (size_t)g_f3dg_render_lds_pad, s, V, P, W, H, tiles_x, T,  \
                                                  focal_x, focal_y, hdr, ranges, point_list, rec, cull, background, bg_per_view,       \
                                                  out_color, final_T, n_contrib)
#define F3DG_LAUNCH3(AUX, FST, OCC) do { if (g_f3dg_render_dma) F3DG_LAUNCH3D(AUX, FST, true, OCC); else F3DG_LAUNCH3D(AUX, FST, false, OCC); } while (0)
        // the split-pixel schedule of f3dg_render5.hip: fast inference launches that ask for it (F3DG_FLAG_SCAN, whatever their size) or,
        // with option render_scan 1, every such launch that is not a one- or two-view latency chain (those keep their multi-wave kernels)
        const bool small_launch = g_f3dg_render_lowocc && (long long)V * T * 4 <= (g_f3dg_render_lowocc > 1 ? 1024ll * g_f3dg_render_lowocc : 2048ll);
        if (g_f3dg_render_slide && g_f3dg_render_fast && !save_aux && g_f3dg_render_scan != 0 && (scan || (g_f3dg_render_scan == 1 && !small_launch)))
            return f3dg_launch_render5(s, V, P, W, H, focal_x, focal_y, hdr, ranges, point_list, rec, cull, background, bg_per_view, out_color,
                                       skip_channels, g_f3dg_render_count);
        // every variant fits 64 VGPRs without spills: 8 waves per SIMD, 32 x 5 KB = the CU's 160 KB of LDS
        // small launches (at most two waves per SIMD: one or two 256^2 views) are latency chains: the prefetching variant
        if (small_launch) {
            // one view (at most one quadrant per SIMD): the multi-wave kernels of f3dg_render4.hip. Defaults by measurement at 65,536 pixel-ordered
            // Gaussians (profiles/r05_final/one_view.md): fast arithmetic -- producer + consumer waves, two entries per trip (render3p, 76.6 ->
            // 57.9 us); the reference's arithmetic -- consumer + three evaluator waves + producer (render3q, 134 -> 100 us)
            // (two views, 2,048 quadrants: render3p as well in fast arithmetic -- 79.7 -> 63.6 us per call; render3q's LDS would not fit)
            const bool one_view = (long long)V * T * 4 <= 1024ll;
            const int split = g_f3dg_render_split >= 0 ? g_f3dg_render_split : g_f3dg_render_fast ? 1 : one_view ? 3 : 0;
            const int unroll = g_f3dg_render_unroll >= 1 ? g_f3dg_render_unroll : g_f3dg_render_fast ? 2 : 1;
            if (split)
                return f3dg_launch_render_small(s, V, P, W, H, focal_x, focal_y, hdr, ranges, point_list, rec, cull, background, bg_per_view, out_color,
                                            g_f3dg_render_fast, save_aux, final_T, n_contrib, unroll, split, g_f3dg_render_count);
#define F3DG_LAUNCH3L(AUX, FST) F3DG_KLAUNCH((render3l_fwd_kernel<AUX, FST>), grid3, dim3(64), 0, s, V, P, W, H, tiles_x, T,  \
                                                  focal_x, focal_y, hdr, ranges, point_list, rec, cull, background, bg_per_view,       \
                                                  out_color, final_T, n_contrib)
            if (save_aux) { if (g_f3dg_render_fast) F3DG_LAUNCH3L(true, true); else F3DG_LAUNCH3L(true, false); }
            else { if (g_f3dg_render_fast) F3DG_LAUNCH3L(false, true); else F3DG_LAUNCH3L(false, false); }
#undef F3DG_LAUNCH3L
            note_kernel("render3l_fwd_kernel", save_aux, g_f3dg_render_fast, "");
            F3DG_HIP_CHECK(hipGetLastError());
            return F3DG_OK;
        }
#ifdef F3DG_LAB
        if (g_f3dg_render_replay >= 2 && !save_aux) {      // lab: the staging half of the last counting launch (3: without the record gathers)
            F3DG_KLAUNCH(render3s_stage_only_kernel, grid3, dim3(64), 0, s, V, P, W, H, tiles_x, T, hdr, ranges, point_list, rec, cull, out_color,
                         g_f3dg_render_replay == 2 ? 1 : 0);
            note_kernel("render3s_stage_only_kernel", 0, 0, "");
            F3DG_HIP_CHECK(hipGetLastError());
            return F3DG_OK;
        }
#endif
        // the rank-packed kernel of f3dg_render4.hip (option render_pack: 1 = every inference launch, -1 = the default: inference launches
        // in the reference's arithmetic, whose stateless part is 2.5 x as long -- measured -38 % on the real merged set, -6 % at C2; in
        // fast arithmetic the packed trips' hand-over costs what they save: 8.4-8.7 against 8.6 ms)
        if (g_f3dg_render_slide && (g_f3dg_render_pack == 1 || (g_f3dg_render_pack < 0 && !g_f3dg_render_fast && g_f3dg_render_wpb == 1 && g_f3dg_render_tail == 0)))
            return f3dg_launch_render4(s, V, P, W, H, focal_x, focal_y, hdr, ranges, point_list, rec, cull, background, bg_per_view, out_color,
                                       g_f3dg_render_fast, skip_channels, g_f3dg_render_count, save_aux, final_T, n_contrib);
        if (g_f3dg_render_slide) {
#define F3DG_R3S_ARGS s, V, P, W, H, tiles_x, T, focal_x, focal_y, hdr, ranges, point_list, rec, cull, background, bg_per_view, out_color, final_T, n_contrib, tail_n
            // render_tail = N > 0 (one-wave workgroups only): the tail schedule once at most N pixels of a quadrant are unsaturated
            const int tail_n = g_f3dg_render_wpb == 1 ? g_f3dg_render_tail : 0;
#define F3DG_LAUNCH3S(AUX, FST, OCC) do { if (g_f3dg_render_wpb == 4)                                                                           \
            F3DG_KLAUNCH((render3s_fwd_kernel<AUX, FST, OCC, 4>), grid, dim3(256), (size_t)g_f3dg_render_lds_pad, F3DG_R3S_ARGS);               \
        else if (tail_n > 0)                                                                                                                    \
            F3DG_KLAUNCH((render3s_fwd_kernel<AUX, FST, OCC, 1, true, true, false, true>), grid3, dim3(64), (size_t)g_f3dg_render_lds_pad, F3DG_R3S_ARGS); \
        else F3DG_KLAUNCH((render3s_fwd_kernel<AUX, FST, OCC, 1>), grid3, dim3(64), (size_t)g_f3dg_render_lds_pad, F3DG_R3S_ARGS); } while (0)
            // the batched loops of the build that consume RGB, depth and alpha only (cycle aggregation, orbit frames) skip the normal
            // and distortion accumulators: the channels they do write are bit-identical
            const bool lean = !save_aux && (skip_channels & (F3DG_FLAG_SKIP_NORMAL | F3DG_FLAG_SKIP_DISTORTION)) == (F3DG_FLAG_SKIP_NORMAL | F3DG_FLAG_SKIP_DISTORTION) &&
                              g_f3dg_render_wpb == 1;
#define F3DG_LAUNCH3S_LEAN(FST) do { if (tail_n > 0)                                                                                            \
            F3DG_KLAUNCH((render3s_fwd_kernel<false, FST, 8, 1, false, false, false, true>), grid3, dim3(64), (size_t)g_f3dg_render_lds_pad, F3DG_R3S_ARGS); \
        else F3DG_KLAUNCH((render3s_fwd_kernel<false, FST, 8, 1, false, false>), grid3, dim3(64), (size_t)g_f3dg_render_lds_pad, F3DG_R3S_ARGS); } while (0)
#define F3DG_LAUNCH3S_COUNT(FST) do { if (tail_n > 0)                                                                                           \
            F3DG_KLAUNCH((render3s_fwd_kernel<false, FST, 8, 1, true, true, true, true>), grid3, dim3(64), 0, F3DG_R3S_ARGS);                    \
        else F3DG_KLAUNCH((render3s_fwd_kernel<false, FST, 8, 1, true, true, true>), grid3, dim3(64), 0, F3DG_R3S_ARGS); } while (0)
            const char* const tail_tag = tail_n > 0 ? ", TAIL=true" : "";
            char extra[96];
            if (g_f3dg_render_count && !save_aux && g_f3dg_render_wpb == 1) {      // (diagnostic: the same kernel with its work counters on)
                if (g_f3dg_render_fast) F3DG_LAUNCH3S_COUNT(true); else F3DG_LAUNCH3S_COUNT(false);
                snprintf(extra, sizeof extra, ", OCC=8, WPB=1, COUNT=true%s", tail_tag);
            } else if (lean) {
                if (g_f3dg_render_fast) F3DG_LAUNCH3S_LEAN(true); else F3DG_LAUNCH3S_LEAN(false);
                snprintf(extra, sizeof extra, ", OCC=8, WPB=1, NORMAL=false, DIST=false%s", tail_tag);
            } else {
                if (save_aux) { if (g_f3dg_render_fast) F3DG_LAUNCH3S(true, true, 8); else F3DG_LAUNCH3S(true, false, 8); }
                else { if (g_f3dg_render_fast) F3DG_LAUNCH3S(false, true, 8); else F3DG_LAUNCH3S(false, false, 8); }
                snprintf(extra, sizeof extra, ", OCC=8, WPB=%d%s", g_f3dg_render_wpb == 4 ? 4 : 1, tail_tag);
            }
#undef F3DG_LAUNCH3S_COUNT
#undef F3DG_LAUNCH3S_LEAN
#undef F3DG_LAUNCH3S
#undef F3DG_R3S_ARGS
            note_kernel("render3s_fwd_kernel", lean || (g_f3dg_render_count && !save_aux && g_f3dg_render_wpb == 1) ? 0 : save_aux, g_f3dg_render_fast, extra);
            F3DG_HIP_CHECK(hipGetLastError());
            return F3DG_OK;
        }
        if (save_aux) { if (g_f3dg_render_fast) F3DG_LAUNCH3(true, true, 8); else F3DG_LAUNCH3(true, false, 8); }
        else { if (g_f3dg_render_fast) F3DG_LAUNCH3(false, true, 8); else F3DG_LAUNCH3(false, false, 8); }
#undef F3DG_LAUNCH3
#undef F3DG_LAUNCH3D
        note_kernel("render3_fwd_kernel", save_aux, g_f3dg_render_fast, g_f3dg_render_dma ? ", DMA=true, OCC=8" : ", DMA=false, OCC=8");
        F3DG_HIP_CHECK(hipGetLastError());
        return F3DG_OK;
    }
    if (g_f3dg_render_kernel == 2) {
#define F3DG_LAUNCH2R(AUX, FST, RND, OCC) F3DG_KLAUNCH((render2_fwd_kernel<AUX, FST, RND, OCC>), grid, dim3(F3DG_BLOCK), 0, s, V, P, W, H, tiles_x, T,  \
                                                  focal_x, focal_y, hdr, ranges, point_list, rec, cull, background, bg_per_view,     \
                                                  out_color, final_T, n_contrib)
        // fast arithmetic fits 72 VGPRs: 192-entry rounds (18 KB of LDS) run 7 workgroups per CU, 2.53 vs 2.66 ms at C2; the exact
        // arithmetic needs 80: 256-entry rounds, 6 per CU. (8 per CU spill at 64 VGPRs and measure 2.55 ms; 128-entry rounds 2.64 ms.)
#define F3DG_LAUNCH2(AUX, FST) do { if (FST && g_f3dg_render_round == 192) F3DG_LAUNCH2R(AUX, FST, 192, 7); else F3DG_LAUNCH2R(AUX, FST, 256, 6); } while (0)
        if (save_aux) { if (g_f3dg_render_fast) F3DG_LAUNCH2(true, true); else F3DG_LAUNCH2(true, false); }
        else { if (g_f3dg_render_fast) F3DG_LAUNCH2(false, true); else F3DG_LAUNCH2(false, false); }
#undef F3DG_LAUNCH2R
#undef F3DG_LAUNCH2
        note_kernel("render2_fwd_kernel", save_aux, g_f3dg_render_fast, "");
        F3DG_HIP_CHECK(hipGetLastError());
        return F3DG_OK;
    }
#define F3DG_LAUNCH(AUX, PRE, CUL, QUE, FST) F3DG_KLAUNCH((render_fwd_kernel<AUX, PRE, CUL, QUE, FST>), grid, dim3(F3DG_BLOCK), 0, s, V, P, \
                                                            W, H, tiles_x, T, focal_x, focal_y, hdr, ranges, point_list, rec,   \
                                                            bbox, background, bg_per_view, out_color, final_T, n_contrib)
#define F3DG_LAUNCH_Q(AUX, PRE, CUL) do { if (g_f3dg_render_fast) { if (g_f3dg_render_queue) F3DG_LAUNCH(AUX, PRE, CUL, true, true); else F3DG_LAUNCH(AUX, PRE, CUL, false, true); } \
                                          else { if (g_f3dg_render_queue) F3DG_LAUNCH(AUX, PRE, CUL, true, false); else F3DG_LAUNCH(AUX, PRE, CUL, false, false); } } while (0)
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
    note_kernel("render_fwd_kernel", save_aux, g_f3dg_render_fast, "");
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}


static bool f3dg_launch_render_lab(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y,
                                   const F3dgHeader* hdr, const uint2* ranges, const unsigned* point_list, const F3dgRec* rec,
                                   const float4* bbox, const float4* cull, const float* background, int bg_per_view, float* out_color,
                                   float* final_T, unsigned* n_contrib, int save_aux, unsigned skip_channels, int fast, int* rc_out)
{
    const int tiles_x = (W + F3DG_TILE - 1) / F3DG_TILE, tiles_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const long long waves = (long long)V * tiles_x * tiles_y * 4;
    const bool small_launch = g_f3dg_render_lowocc && waves <= (g_f3dg_render_lowocc > 1 ? 1024ll * g_f3dg_render_lowocc : 2048ll);
    const bool lab = g_f3dg_render_kernel != 3 || !g_f3dg_render_slide || g_f3dg_render_tail > 0 || g_f3dg_render_wpb == 4 || g_f3dg_render_lds_pad != 0 ||
                     (g_f3dg_render_replay >= 2 && !save_aux) || (small_launch && g_f3dg_render_split == 0);
    if (!lab) return false;
    const int keep_scan = g_f3dg_render_scan;
    g_f3dg_render_scan = 0;                 // (a lab option overrides the split-pixel mode)
    *rc_out = f3dg_launch_render_lab_impl(s, V, P, W, H, focal_x, focal_y, hdr, ranges, point_list, rec, bbox, cull, background, bg_per_view, out_color,
                                          final_T, n_contrib, save_aux, skip_channels, fast, 0);
    g_f3dg_render_scan = keep_scan;
    return true;
}
#endif // F3DG_LAB

// debug: the work counters of the counting variant of the one-wave kernel (option render_count = 1), summed over all launches since
// the last reset: h_out8[16] = { staged, scanned, wave trips, slides, lane-trips, waves, trips with <= 8 / <= 24 live pixels, slides with <= 8 / <= 24, 0... }
extern "C" int f3dg_debug_render_counts(unsigned long long* h_out8, int reset)
{
    unsigned long long rows[64][16];
    F3DG_HIP_CHECK(hipMemcpyFromSymbol(rows, HIP_SYMBOL(g_f3dg_counts), sizeof rows));
    if (h_out8)
        for (int k = 0; k < 16; k++) {
            h_out8[k] = 0;
            for (int r = 0; r < 64; r++) h_out8[k] += rows[r][k];
        }
    if (reset) {
        memset(rows, 0, sizeof rows);
        F3DG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_f3dg_counts), rows, sizeof rows));
    }
    return F3DG_OK;
}

// the compositing forward kernel the last f3dg_launch_render of this process launched (what `roofline.kernel` of bench.py prints)
thread_local const char* g_f3dg_last_render_kernel = "";
extern "C" const char* f3dg_debug_last_render_kernel(void) { return g_f3dg_last_render_kernel; }

// debug: read (and optionally reset) the phase-timing counters of a -DF3DG_TIMING build (zeros otherwise)
extern "C" int f3dg_debug_timing(unsigned long long* h_out8, int reset)
{
    if (h_out8) F3DG_HIP_CHECK(hipMemcpyFromSymbol(h_out8, HIP_SYMBOL(g_f3dg_timing), 8 * sizeof(unsigned long long)));
    if (reset) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        F3DG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_f3dg_timing), z, sizeof z));
    }
    return F3DG_OK;
}
