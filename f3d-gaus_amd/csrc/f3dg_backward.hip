// f3dg_backward.hip -- backward of the compositing stage and of the per-Gaussian projection stage.
//
// Replaces (reference RAST/cuda_rasterizer/backward.cu):
//   renderCUDA<3> (bwd)              :634-955   -> render_bwd_kernel
//   preprocessCUDA<3> (bwd)          :593-631   -> preprocess_bwd_kernel, with
//   computeView2Gaussian_backward    :381-587
//   computeColorFromSH (bwd)         :20-139
// The reference's computeCov2DCUDA / computeCov3D backward are commented out there (:992-1007, :627-630), so
// dL_dconic and dL_dcov3D are identically zero; they are left untouched here as well.
//
// MI355X shape. The reference issues 17 float atomicAdds per contributing (pixel, Gaussian) pair. Here all 64
// lanes of a wave walk the tile list in lock-step, so the 17 partial derivatives of one Gaussian are first
// summed across the wave with cross-lane butterflies and only lane 0 touches memory: <= 17 atomics per
// (wave, Gaussian) instead of per (pixel, Gaussian), and nothing at all for the (wave, Gaussian) pairs no lane of
// the strip contributes to (a ballot). The 10 dL/dview2gaussian entries are accumulated in float64
// (global_atomic_add_f64): the per-Gaussian stage below amplifies a 5e-7 ordering perturbation of them into tens
// of percent on dL/dscale (SURVEY 0.9 -- the reference's own run-to-run spread), so float64 sums make the result
// reproducible to rounding where the reference is not.
#include "f3dg_common.h"
#include "f3dg_ellipse.h"

namespace {

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// float64 butterfly: the 64 per-pixel float terms are summed exactly (to double rounding), so the only float32
// rounding left in dL/dview2gaussian is the final narrowing -- see the header comment on why that matters
__device__ __forceinline__ double wave_sum_f64(float v)
{
    double d = (double)v;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) d += __shfl_xor(d, m, 64);
    return d;
}


// Wave64 sums on the VALU with DPP (no LDS traffic): inclusive row scans (row_shr 1, 2, 4, 8 with zero fill), then the
// row totals are folded across the four rows (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3). The total
// ends up in lane 63. The backward kernel is LDS-pipe-bound with ds_bpermute butterflies or LDS atomics (measured:
// SQ_ACTIVE_INST_LDS ~ the kernel time), while its VALU is mostly idle.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
#define F3DG_DPP_ROW_SHR(n) (0x110 + (n))
#define F3DG_DPP_ROW_BCAST15 0x142
#define F3DG_DPP_ROW_BCAST31 0x143
__device__ __forceinline__ float wave_total_lane63(float v)
{
    v += dpp_f32<F3DG_DPP_ROW_SHR(1), 0xF>(v);
    v += dpp_f32<F3DG_DPP_ROW_SHR(2), 0xF>(v);
    v += dpp_f32<F3DG_DPP_ROW_SHR(4), 0xF>(v);
    v += dpp_f32<F3DG_DPP_ROW_SHR(8), 0xF>(v);
    v += dpp_f32<F3DG_DPP_ROW_BCAST15, 0xA>(v);
    v += dpp_f32<F3DG_DPP_ROW_BCAST31, 0xC>(v);
    return v;
}
__device__ __forceinline__ double wave_total_lane63_f64(float f)
{
    double v = (double)f;
    v += dpp_f64<F3DG_DPP_ROW_SHR(1), 0xF>(v);
    v += dpp_f64<F3DG_DPP_ROW_SHR(2), 0xF>(v);
    v += dpp_f64<F3DG_DPP_ROW_SHR(4), 0xF>(v);
    v += dpp_f64<F3DG_DPP_ROW_SHR(8), 0xF>(v);
    v += dpp_f64<F3DG_DPP_ROW_BCAST15, 0xA>(v);
    v += dpp_f64<F3DG_DPP_ROW_BCAST31, 0xC>(v);
    return v;
}

// ---- transposed wave reduction of several values at once -------------------------------------------------------------
// v_permlane32_swap / v_permlane16_swap (gfx950) exchange half-waves / 16-lane rows between two registers, so ONE add
// halves the lane span of TWO values: after both, every 16-lane row holds the partial sums of a different value and only
// ceil(n/4) registers are left for the four row-local DPP steps. 17 values: 34 adds instead of 102.
//   pair32(X, Y): lanes 0..31 = X[i] + X[i+32], lanes 32..63 = Y[i-32] + Y[i]
//   pair16(P, Q): row 0 = P.row0 + P.row1, row 1 = Q.row0 + Q.row1, row 2 = P.row2 + P.row3, row 3 = Q.row2 + Q.row3
typedef unsigned __attribute__((ext_vector_type(2))) f3dg_u2;
__device__ __forceinline__ float pair32(float x, float y)
{
    const f3dg_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float pair16(float x, float y)
{
    const f3dg_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ double pair32(double x, double y)
{
    const unsigned long long a = (unsigned long long)__double_as_longlong(x), b = (unsigned long long)__double_as_longlong(y);
    const f3dg_u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    const f3dg_u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)(a >> 32), (unsigned)(b >> 32), false, false);
    return __longlong_as_double((long long)(((unsigned long long)hi.x << 32) | lo.x)) +
           __longlong_as_double((long long)(((unsigned long long)hi.y << 32) | lo.y));
}
__device__ __forceinline__ double pair16(double x, double y)
{
    const unsigned long long a = (unsigned long long)__double_as_longlong(x), b = (unsigned long long)__double_as_longlong(y);
    const f3dg_u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
    const f3dg_u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)(a >> 32), (unsigned)(b >> 32), false, false);
    return __longlong_as_double((long long)(((unsigned long long)hi.x << 32) | lo.x)) +
           __longlong_as_double((long long)(((unsigned long long)hi.y << 32) | lo.y));
}
// total of each 16-lane row in its lane 15
__device__ __forceinline__ float row_total(float v)
{
    v += dpp_f32<F3DG_DPP_ROW_SHR(1), 0xF>(v);
    v += dpp_f32<F3DG_DPP_ROW_SHR(2), 0xF>(v);
    v += dpp_f32<F3DG_DPP_ROW_SHR(4), 0xF>(v);
    v += dpp_f32<F3DG_DPP_ROW_SHR(8), 0xF>(v);
    return v;
}
__device__ __forceinline__ double row_total(double v)
{
    v += dpp_f64<F3DG_DPP_ROW_SHR(1), 0xF>(v);
    v += dpp_f64<F3DG_DPP_ROW_SHR(2), 0xF>(v);
    v += dpp_f64<F3DG_DPP_ROW_SHR(4), 0xF>(v);
    v += dpp_f64<F3DG_DPP_ROW_SHR(8), 0xF>(v);
    return v;
}

#ifdef F3DG_LAB      // ---- rounds 1-2: the lock-step and the four-wave compositing backward, lab builds only
__global__ void __launch_bounds__(F3DG_BLOCK)
render_bwd_lockstep_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
                  const F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                  const unsigned* __restrict__ point_list_general, const unsigned* __restrict__ small_list, const F3dgRec* __restrict__ rec,
                  const float2* __restrict__ means2D, const float4* __restrict__ conic,
                  const float* __restrict__ background, int bg_per_view,
                  const float* __restrict__ final_T, const unsigned* __restrict__ n_contrib,
                  const float* __restrict__ dL_dpixels,
                  float* __restrict__ dL_dmean2D, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors,
                  double* __restrict__ dL_dv2g_acc)
{
    unsigned view, tile;                      // all tiles of a view share one XCD's L2 for the record gather
    f3dg_xcd_map(blockIdx.x, (unsigned)V, (unsigned)T, view, tile);

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
    // no lists to walk after an overflow -- and none that belong to this call when the workspace's last forward kept no auxiliary
    // planes (an inference call): all gradients stay zero, the header says why
    // (a one-view forward with auxiliary planes may have taken the small-call path: its lists live in the per-tile slots)
    const unsigned* __restrict__ point_list = hdr->small_path != 0u ? small_list : point_list_general;
    if (hdr->overflow || hdr->save_aux == 0u) {
        range = make_uint2(0, 0);
        if (!hdr->overflow && blockIdx.x == 0 && threadIdx.x == 0) const_cast<F3dgHeader*>(hdr)->bwd_stale = 1u;
    }
    const int rounds = (int)((range.y - range.x + F3DG_BLOCK - 1) / F3DG_BLOCK);
    int toDo = (int)(range.y - range.x);

    __shared__ float4 staged[F3DG_BLOCK * 4];
    __shared__ float4 staged_conic[F3DG_BLOCK];
    __shared__ float2 staged_xy[F3DG_BLOCK];
    __shared__ unsigned staged_id[F3DG_BLOCK];

    const bool alpha_fast = hdr->alpha_fast != 0;     // the arithmetic the forward of this workspace used for alpha: repeated to the bit
    const size_t vP = (size_t)view * P;
    const float* fT = final_T + (size_t)view * 4 * HW;
    const unsigned* nc = n_contrib + (size_t)view * 2 * HW;
    const float* dpix = dL_dpixels + (size_t)view * F3DG_OUT_CHANNELS * HW;
    const float* bg = background + (bg_per_view ? 3 * view : 0);

    const float T_final = inside ? fT[pix_id] : 0;
    float Tr = T_final;
    const float final_D = inside ? fT[pix_id + HW] : 0;
    const float final_A = 1 - T_final;
    const float dL_dreg = inside ? dpix[8 * HW + pix_id] : 0;

    unsigned contributor = (unsigned)toDo;
    const int last_contributor = inside ? (int)nc[pix_id] : 0;
    const int max_contributor = inside ? (int)nc[pix_id + HW] : 0;
    float accum_rec0 = 0, accum_rec1 = 0, accum_rec2 = 0;
    float dpx0 = 0, dpx1 = 0, dpx2 = 0, dn0 = 0, dn1 = 0, dn2 = 0, dL_dmax_depth = 0;
    if (inside) {
        dpx0 = dpix[pix_id]; dpx1 = dpix[HW + pix_id]; dpx2 = dpix[2 * HW + pix_id];
        dn0 = dpix[3 * HW + pix_id]; dn1 = dpix[4 * HW + pix_id]; dn2 = dpix[5 * HW + pix_id];
        dL_dmax_depth = dpix[6 * HW + pix_id];
    }
    float last_alpha = 0;
    float last_c0 = 0, last_c1 = 0, last_c2 = 0;
    float last_n0 = 0, last_n1 = 0, last_n2 = 0;
    float acc_n0 = 0, acc_n1 = 0, acc_n2 = 0;
    const float ddelx_dx = (float)(0.5 * W);
    const float ddely_dy = (float)(0.5 * H);
    const float bg_dot_dpixel = bg[0] * dpx0 + bg[1] * dpx1 + bg[2] * dpx2;
    const bool lane0 = (threadIdx.x & 63u) == 0;

    for (int i = 0; i < rounds; i++, toDo -= F3DG_BLOCK) {
        __syncthreads();
        const unsigned progress = (unsigned)i * F3DG_BLOCK + threadIdx.x;
        if (range.x + progress < range.y) {
            const unsigned id = point_list[range.y - progress - 1] & F3DG_ID_MASK;       // back to front
            const float4* src = reinterpret_cast<const float4*>(rec + vP + id);
            staged[threadIdx.x * 4 + 0] = src[0];
            staged[threadIdx.x * 4 + 1] = src[1];
            staged[threadIdx.x * 4 + 2] = src[2];
            staged[threadIdx.x * 4 + 3] = src[3];
            staged_conic[threadIdx.x] = conic[vP + id];
            staged_xy[threadIdx.x] = means2D[vP + id];
            staged_id[threadIdx.x] = id;
        }
        __syncthreads();

        const int n = min(F3DG_BLOCK, toDo);
        for (int j = 0; j < n; j++) {
            contributor--;
            bool active = inside && !(contributor >= (unsigned)last_contributor);

            const float4 q0 = staged[j * 4 + 0];
            const float4 q1 = staged[j * 4 + 1];
            const float4 q2 = staged[j * 4 + 2];
            const float n0 = q0.x * ray_x + q0.y * ray_y + q0.z;
            const float n1 = q0.y * ray_x + q0.w * ray_y + q1.x;
            const float n2 = q0.z * ray_x + q1.x * ray_y + q1.y;
            const float aaf = ray_x * n0 + ray_y * n1 + n2;
            const float bhalf = q1.z * ray_x + q1.w * ray_y + q2.x;
            // the forward's conservative pre-test (f3dg_render.hip): a true test proves alpha < 1/255, i.e. `continue`
            if (bhalf * bhalf < q2.w * aaf) active = false;
            if (__ballot(active) == 0)      // no pixel of this strip can contribute: skip before any float64 / expf work
                continue;

            const double AA = aaf;
            const double BB = 2 * bhalf;
            const float CC = q2.y;
            float t = 0, G = 0, alpha = 0;
            if (active) {
                if (alpha_fast) {
                    // the forward of this workspace took the fast arithmetic: the same function, to the bit
                    f3dg_fast_t_G(aaf, bhalf, CC, t, G);
                    if (t < 0.2f) active = false;
                } else {
                    const double q = BB / AA;                          // one division: -BB / (2 * AA) == -0.5 * (BB / AA) exactly
                    t = (float)(-0.5 * q);
                    if (t <= F3DG_NEAR_PLANE) active = false;
                    const double min_value = -q * (BB / 4.) + CC;
                    float power = (float)(-0.5f * min_value);
                    if (power > 0.0f) power = 0.0f;
                    G = expf(power);
                }
                alpha = fminf(0.99f, q2.z * G);
                if (alpha < 1.0f / 255.0f) active = false;
            }
            if (__ballot(active) == 0)
                continue;

            float g_col0 = 0, g_col1 = 0, g_col2 = 0, g_mx = 0, g_my = 0, g_mz = 0, g_op = 0;
            float g_v0 = 0, g_v1 = 0, g_v2 = 0, g_v3 = 0, g_v4 = 0, g_v5 = 0, g_v6 = 0, g_v7 = 0, g_v8 = 0, g_v9 = 0;
            if (active) {
                const float4 q3 = staged[j * 4 + 3];
                const float4 con = staged_conic[j];
                const float2 xy = staged_xy[j];
                const float d_x = (float)(xy.x - (pixf_x - 0.5)), d_y = (float)(xy.y - (pixf_y - 0.5));

                const float mapped_max_t = (float)((F3DG_FAR_PLANE * t - F3DG_FAR_PLANE * F3DG_NEAR_PLANE) / ((F3DG_FAR_PLANE - F3DG_NEAR_PLANE) * t));
                const float dmax_t_dd = (float)((F3DG_FAR_PLANE * F3DG_NEAR_PLANE) / ((F3DG_FAR_PLANE - F3DG_NEAR_PLANE) * t * t));
                const float length = (float)sqrt(n0 * n0 + n1 * n1 + n2 * n2 + 1e-7);
                const float nn0 = -n0 / length, nn1 = -n1 / length, nn2 = -n2 / length;

                Tr = Tr / (1.f - alpha);
                const float dchannel_dcolor = alpha * Tr;

                float dL_dalpha = 0.0f;
                const float c0 = q3.x, c1 = q3.y, c2 = q3.z;
                accum_rec0 = last_alpha * last_c0 + (1.f - last_alpha) * accum_rec0; last_c0 = c0;
                dL_dalpha += (c0 - accum_rec0) * dpx0; g_col0 = dchannel_dcolor * dpx0;
                accum_rec1 = last_alpha * last_c1 + (1.f - last_alpha) * accum_rec1; last_c1 = c1;
                dL_dalpha += (c1 - accum_rec1) * dpx1; g_col1 = dchannel_dcolor * dpx1;
                accum_rec2 = last_alpha * last_c2 + (1.f - last_alpha) * accum_rec2; last_c2 = c2;
                dL_dalpha += (c2 - accum_rec2) * dpx2; g_col2 = dchannel_dcolor * dpx2;

                // distortion: only dL/dmax_t survives, the weight gradient is detached (backward.cu:850-852) and
                // last_dL_dT therefore stays 0
                float dL_dmax_t = 0.0f;
                dL_dmax_t += 2.0f * (Tr * alpha) * (mapped_max_t * final_A - final_D) * dL_dreg * dmax_t_dd;
                dL_dalpha += 0.f - 0.f;

                acc_n0 = last_alpha * last_n0 + (1.f - last_alpha) * acc_n0; last_n0 = nn0;
                dL_dalpha += (nn0 - acc_n0) * dn0;
                const float dnn0 = alpha * Tr * dn0;
                acc_n1 = last_alpha * last_n1 + (1.f - last_alpha) * acc_n1; last_n1 = nn1;
                dL_dalpha += (nn1 - acc_n1) * dn1;
                const float dnn1 = alpha * Tr * dn1;
                acc_n2 = last_alpha * last_n2 + (1.f - last_alpha) * acc_n2; last_n2 = nn2;
                dL_dalpha += (nn2 - acc_n2) * dn2;
                const float dnn2 = alpha * Tr * dn2;

                float dL_dlength = (dnn0 * n0 + dnn1 * n1 + dnn2 * n2);
                dL_dlength *= 1.f / (length * length);
                float dLn0 = (-dnn0 + dL_dlength * n0) / length;
                float dLn1 = (-dnn1 + dL_dlength * n1) / length;
                float dLn2 = (-dnn2 + dL_dlength * n2) / length;

                float dL_dt = dL_dmax_t;
                if ((int)contributor == max_contributor - 1)
                    dL_dt += dL_dmax_depth;

                dL_dalpha *= Tr;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

                const float dL_dG = con.w * dL_dalpha;
                const float gdx = G * d_x;
                const float gdy = G * d_y;
                const float dG_ddelx = -gdx * con.x - gdy * con.y;
                const float dG_ddely = -gdy * con.z - gdx * con.y;
                g_mx = dL_dG * dG_ddelx * ddelx_dx;
                g_my = dL_dG * dG_ddely * ddely_dy;
                g_mz = fabsf(dL_dG * dG_ddelx * ddelx_dx) + fabsf(dL_dG * dG_ddely * ddely_dy);
                g_op = G * dL_dalpha;

                const float dL_dpower = dL_dG * G;
                const float dL_dmin_value = dL_dpower * -0.5f;
                double dL_dA = dL_dmin_value * (BB / AA) * (BB / AA) / 4.f;
                double dL_dB = dL_dmin_value * -BB / (2 * AA);
                const double dL_dC = dL_dmin_value * 1.0f;
                dL_dA += dL_dt * BB / (2 * AA * AA);
                dL_dB += dL_dt * -1.f / (2 * AA);
                dLn0 += dL_dA * ray_x;
                dLn1 += dL_dA * ray_y;
                dLn2 += dL_dA;

                g_v0 = dLn0 * ray_x;
                g_v1 = dLn0 * ray_y + dLn1 * ray_x;
                g_v2 = dLn0 + dLn2 * ray_x;
                g_v3 = dLn1 * ray_y;
                g_v4 = dLn1 + dLn2 * ray_y;
                g_v5 = dLn2;
                g_v6 = (float)(dL_dB * 2 * ray_x);
                g_v7 = (float)(dL_dB * 2 * ray_y);
                g_v8 = (float)(dL_dB * 2);
                g_v9 = (float)dL_dC;
            }

            // sum the 17 partials over the wave's 64 pixels, then one lane updates memory
            g_col0 = wave_sum(g_col0); g_col1 = wave_sum(g_col1); g_col2 = wave_sum(g_col2);
            g_mx = wave_sum(g_mx); g_my = wave_sum(g_my); g_mz = wave_sum(g_mz); g_op = wave_sum(g_op);
            const double s_v0 = wave_sum_f64(g_v0), s_v1 = wave_sum_f64(g_v1), s_v2 = wave_sum_f64(g_v2),
                         s_v3 = wave_sum_f64(g_v3), s_v4 = wave_sum_f64(g_v4), s_v5 = wave_sum_f64(g_v5),
                         s_v6 = wave_sum_f64(g_v6), s_v7 = wave_sum_f64(g_v7), s_v8 = wave_sum_f64(g_v8),
                         s_v9 = wave_sum_f64(g_v9);
            if (lane0) {
                const unsigned id = staged_id[j];
                const size_t gi = vP + id;
                unsafeAtomicAdd(&dL_dcolors[gi * 3 + 0], g_col0);
                unsafeAtomicAdd(&dL_dcolors[gi * 3 + 1], g_col1);
                unsafeAtomicAdd(&dL_dcolors[gi * 3 + 2], g_col2);
                unsafeAtomicAdd(&dL_dmean2D[gi * 3 + 0], g_mx);
                unsafeAtomicAdd(&dL_dmean2D[gi * 3 + 1], g_my);
                unsafeAtomicAdd(&dL_dmean2D[gi * 3 + 2], g_mz);
                unsafeAtomicAdd(&dL_dopacity[id], g_op);
                double* a = dL_dv2g_acc + gi * 10;
                unsafeAtomicAdd(a + 0, s_v0); unsafeAtomicAdd(a + 1, s_v1);
                unsafeAtomicAdd(a + 2, s_v2); unsafeAtomicAdd(a + 3, s_v3);
                unsafeAtomicAdd(a + 4, s_v4); unsafeAtomicAdd(a + 5, s_v5);
                unsafeAtomicAdd(a + 6, s_v6); unsafeAtomicAdd(a + 7, s_v7);
                unsafeAtomicAdd(a + 8, s_v8); unsafeAtomicAdd(a + 9, s_v9);
            }
        }
    }
}


// ---- culled + DPP-reduced variant (the default) -------------------------------------------------------------------
// Same arithmetic per contributing (pixel, Gaussian) pair as render_bwd_lockstep_kernel above, but
//   * a wave owns an 8x8 pixel quadrant (not a 16x4 strip) and walks only the staged entries whose conservative
//     alpha >= 1/255 box (written by the forward's preprocess) touches its quadrant -- every skipped pair is one the
//     reference `continue`s on (backward.cu:773-777), so no per-pixel state changes;
//   * the 17 partials are not butterflied over all 64 lanes (102 cross-lane exchanges + 60 float64 adds per
//     (wave, Gaussian), although only a handful of lanes contribute): the contributing lanes add them into 17 per-wave
//     LDS accumulators (ds_add_f32 / ds_add_f64), then lanes 0..16 each swap one accumulator out and issue ONE global
//     atomic in parallel. float64 accumulation of dL/dview2gaussian is kept.
// The position of an entry from the front of the list is computed from its staged slot, so skipped entries need no counter.
#ifndef F3DG_BWD_OCC
#define F3DG_BWD_OCC 5          // waves per SIMD the register allocation aims at (96 VGPRs): measured at C5, see DESIGN.md
#endif
__global__ void __launch_bounds__(F3DG_BLOCK, F3DG_BWD_OCC)
render_bwd_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
                  F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                  const unsigned* __restrict__ point_list_general, const unsigned* __restrict__ small_list, const F3dgRec* __restrict__ rec,
                  const float4* __restrict__ bbox,
                  const float2* __restrict__ means2D, const float4* __restrict__ conic,
                  const float* __restrict__ background, int bg_per_view,
                  const float* __restrict__ final_T, const unsigned* __restrict__ n_contrib,
                  const float* __restrict__ dL_dpixels,
                  float* __restrict__ dL_dmean2D, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors,
                  double* __restrict__ dL_dv2g_acc)
{
    unsigned view, tile;                      // all tiles of a view share one XCD's L2 for the record gather
    f3dg_xcd_map(blockIdx.x, (unsigned)V, (unsigned)T, view, tile);

    const unsigned tile_x = tile % (unsigned)tiles_x, tile_y = tile / (unsigned)tiles_x;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned lx = (wave & 1u) * 8u + (lane & 7u), ly = (wave >> 1) * 8u + (lane >> 3);
    const unsigned pix_x = tile_x * F3DG_TILE + lx, pix_y = tile_y * F3DG_TILE + ly;
    const bool inside = pix_x < (unsigned)W && pix_y < (unsigned)H;
    const size_t HW = (size_t)H * W;
    const size_t pix_id = (size_t)W * pix_y + pix_x;
    const float pixf_x = (float)pix_x + 0.5f, pixf_y = (float)pix_y + 0.5f;
    const float ray_x = (float)((pixf_x - W / 2.) / focal_x);
    const float ray_y = (float)((pixf_y - H / 2.) / focal_y);

    uint2 range = ranges[(size_t)view * T + tile];
    // no lists to walk after an overflow -- and none that belong to this call when the workspace's last forward kept no auxiliary
    // planes (an inference call): all gradients stay zero, the header says why
    // (a one-view forward with auxiliary planes may have taken the small-call path: its lists live in the per-tile slots)
    const unsigned* __restrict__ point_list = hdr->small_path != 0u ? small_list : point_list_general;
    if (hdr->overflow || hdr->save_aux == 0u) {
        range = make_uint2(0, 0);
        if (!hdr->overflow && blockIdx.x == 0 && threadIdx.x == 0) const_cast<F3dgHeader*>(hdr)->bwd_stale = 1u;
    }

    __shared__ float4 sq0[F3DG_BLOCK], sq1[F3DG_BLOCK], sq2[F3DG_BLOCK], sq3[F3DG_BLOCK];
    __shared__ float4 staged_conic[F3DG_BLOCK];
    __shared__ float2 staged_xy[F3DG_BLOCK];
    __shared__ unsigned staged_id[F3DG_BLOCK];
    __shared__ unsigned char quad_mask[F3DG_BLOCK];
    __shared__ unsigned char wave_list[F3DG_BLOCK / 64][F3DG_BLOCK];
    __shared__ int block_last_s;
    if (threadIdx.x == 0) block_last_s = 0;

    const bool alpha_fast = hdr->alpha_fast != 0;
    const size_t vP = (size_t)view * P;
    const float4* vbox = bbox + vP;
    const float* fT = final_T + (size_t)view * 4 * HW;
    const unsigned* nc = n_contrib + (size_t)view * 2 * HW;
    const float* dpix = dL_dpixels + (size_t)view * F3DG_OUT_CHANNELS * HW;
    const float* bg = background + (bg_per_view ? 3 * view : 0);
    const float tile_px0 = (float)(tile_x * F3DG_TILE), tile_py0 = (float)(tile_y * F3DG_TILE);

    const float T_final = inside ? fT[pix_id] : 0;
    float Tr = T_final;
    const float final_D = inside ? fT[pix_id + HW] : 0;
    const float final_A = 1 - T_final;
    const float dL_dreg = inside ? dpix[8 * HW + pix_id] : 0;

    const int last_contributor = inside ? (int)nc[pix_id] : 0;
    const int max_contributor = inside ? (int)nc[pix_id + HW] : 0;
    float accum_rec0 = 0, accum_rec1 = 0, accum_rec2 = 0;
    float dpx0 = 0, dpx1 = 0, dpx2 = 0, dn0 = 0, dn1 = 0, dn2 = 0, dL_dmax_depth = 0;
    if (inside) {
        dpx0 = dpix[pix_id]; dpx1 = dpix[HW + pix_id]; dpx2 = dpix[2 * HW + pix_id];
        dn0 = dpix[3 * HW + pix_id]; dn1 = dpix[4 * HW + pix_id]; dn2 = dpix[5 * HW + pix_id];
        dL_dmax_depth = dpix[6 * HW + pix_id];
    }
    float last_alpha = 0;
    float last_c0 = 0, last_c1 = 0, last_c2 = 0;
    float last_n0 = 0, last_n1 = 0, last_n2 = 0;
    float acc_n0 = 0, acc_n1 = 0, acc_n2 = 0;
    const float ddelx_dx = (float)(0.5 * W);
    const float ddely_dy = (float)(0.5 * H);
    const float bg_dot_dpixel = bg[0] * dpx0 + bg[1] * dpx1 + bg[2] * dpx2;
    // Entries at or behind a pixel's last contributor are skipped by the reference one by one (backward.cu:745-746); the
    // tile starts at the deepest last contributor of its 256 pixels instead of staging the whole list from the back
    // (with opaque scenes the forward stops after a small part of an 8 k-entry list, and so does this).
    const int wave_last = (int)__builtin_amdgcn_readfirstlane((int)__reduce_max_sync(~0ull, last_contributor));
    __syncthreads();
    if (lane == 0) atomicMax(&block_last_s, wave_last);
    __syncthreads();
    const int block_last = min(block_last_s, (int)(range.y - range.x));      // entries [0, block_last) can contribute
    const int rounds = (block_last + F3DG_BLOCK - 1) / F3DG_BLOCK;
    int toDo = block_last;
    unsigned n_pairs = 0;                 // contributing (pixel, Gaussian) pairs of this wave

    for (int i = 0; i < rounds; i++, toDo -= F3DG_BLOCK) {
        __syncthreads();
        const int progress = i * F3DG_BLOCK + (int)threadIdx.x;
        if (progress < block_last) {
            const unsigned id = point_list[range.x + (unsigned)(block_last - 1 - progress)] & F3DG_ID_MASK;       // back to front
            const float4* src = reinterpret_cast<const float4*>(rec + vP + id);
            sq0[threadIdx.x] = src[0];
            sq1[threadIdx.x] = src[1];
            sq2[threadIdx.x] = src[2];
            sq3[threadIdx.x] = src[3];
            staged_conic[threadIdx.x] = conic[vP + id];
            staged_xy[threadIdx.x] = means2D[vP + id];
            staged_id[threadIdx.x] = id;
            const float4 bx = vbox[id];
            const unsigned mx = (bx.x <= tile_px0 + 7.0f && bx.y >= tile_px0 ? 1u : 0u) |
                                (bx.x <= tile_px0 + 15.0f && bx.y >= tile_px0 + 8.0f ? 2u : 0u);
            const unsigned my = (bx.z <= tile_py0 + 7.0f && bx.w >= tile_py0 ? 1u : 0u) |
                                (bx.z <= tile_py0 + 15.0f && bx.w >= tile_py0 + 8.0f ? 2u : 0u);
            quad_mask[threadIdx.x] = (unsigned char)(((mx & 1u) && (my & 1u) ? 1u : 0u) | ((mx & 2u) && (my & 1u) ? 2u : 0u) |
                                                     ((mx & 1u) && (my & 2u) ? 4u : 0u) | ((mx & 2u) && (my & 2u) ? 8u : 0u));
        } else {
            quad_mask[threadIdx.x] = 0;
        }
        __syncthreads();

        // entries of this round are front positions [first_front - n + 1, first_front]; nothing to do while the whole
        // round lies behind every pixel's last contributor
        const int first_front = block_last - 1 - i * F3DG_BLOCK;
        const int n = min(F3DG_BLOCK, toDo);
        if (first_front - n + 1 >= wave_last)
            continue;
        int count = 0;
#pragma unroll
        for (int c = 0; c < F3DG_BLOCK / 64; c++) {
            const unsigned e = c * 64 + lane;
            const bool bit = (quad_mask[e] >> wave) & 1u;
            const unsigned long long bal = __ballot(bit);
            if (bit) wave_list[wave][count + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned char)e;
            count += __popcll(bal);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        for (int kk = 0; kk < count; kk++) {
            const int j = (int)wave_list[wave][kk];
            const int contributor = first_front - j;                       // 0-based position from the front
            bool active = inside && contributor < last_contributor;

            const float4 q0 = sq0[j], q1 = sq1[j], q2 = sq2[j];
            const float n0 = q0.x * ray_x + q0.y * ray_y + q0.z;
            const float n1 = q0.y * ray_x + q0.w * ray_y + q1.x;
            const float n2 = q0.z * ray_x + q1.x * ray_y + q1.y;
            const float aaf = ray_x * n0 + ray_y * n1 + n2;
            const float bhalf = q1.z * ray_x + q1.w * ray_y + q2.x;
            // the forward's conservative pre-test (f3dg_render.hip): a true test proves alpha < 1/255, i.e. `continue`
            if (bhalf * bhalf < q2.w * aaf) active = false;
            if (__ballot(active) == 0)
                continue;

            const float CC = q2.y;
            float t = 0, G = 0, alpha = 0;
            if (active) {
                if (alpha_fast) {
                    // the forward of this workspace took the fast arithmetic: the same function, to the bit
                    f3dg_fast_t_G(aaf, bhalf, CC, t, G);
                    if (t < 0.2f) active = false;
                    alpha = fminf(0.99f, q2.z * G);
                } else {
                    const double AA = aaf;
                    const double BB = 2 * bhalf;
                    const double q = BB / AA;                          // one division: -BB / (2 * AA) == -0.5 * (BB / AA) exactly
                    t = (float)(-0.5 * q);
                    if (t <= F3DG_NEAR_PLANE) active = false;
                    const double min_value = -q * (BB / 4.) + CC;
                    float power = (float)(-0.5f * min_value);
                    if (power > 0.0f) power = 0.0f;
                    G = expf(power);
                    alpha = fminf(0.99f, q2.z * G);
                }
                if (alpha < 1.0f / 255.0f) active = false;
            }
            {
                const unsigned long long act = __ballot(active);
                if (act == 0)
                    continue;
                n_pairs += (unsigned)__popcll(act);
            }

            float g_col0 = 0, g_col1 = 0, g_col2 = 0, g_mx = 0, g_my = 0, g_mz = 0, g_op = 0;
            float g_v0 = 0, g_v1 = 0, g_v2 = 0, g_v3 = 0, g_v4 = 0, g_v5 = 0, g_v6 = 0, g_v7 = 0, g_v8 = 0, g_v9 = 0;
            if (active) {
                // gradient terms only from here (alpha, the one value that must repeat the forward's bits, is done): products and sums
                // may contract into FMAs (the file is built with -ffp-contract=off for the alpha path)
#pragma clang fp contract(fast)
                const float4 q3 = sq3[j];
                const float4 con = staged_conic[j];
                const float2 xy = staged_xy[j];
                const float d_x = (float)(xy.x - (pixf_x - 0.5)), d_y = (float)(xy.y - (pixf_y - 0.5));

                // Only alpha has to repeat the forward to the bit (T is rebuilt by dividing by 1 - alpha): its float64 quotient q and
                // min_value above are the forward's. Everything below is a gradient term that ends in a float32 sum, so the reference's
                // remaining float64 divisions / square root (backward.cu:783-793, 923-928) are evaluated in float32 with one reciprocal
                // each (<= 1 ulp; the tests hold the result to 1e-5 of the maximum against the oracle's float64 evaluation):
                //   mapped = far/(far-near) - (far near/(far-near)) / t,   d mapped/dt = (far near/(far-near)) / t^2
                const float inv_t = 1.0f / t;
                const float mapped_max_t = fmaf(-0.20040080160320642f, inv_t, 1.0020040080160322f);
                const float dmax_t_dd = 0.20040080160320642f * inv_t * inv_t;
                const float inv_len = 1.0f / sqrtf(n0 * n0 + n1 * n1 + n2 * n2 + 1e-7f);
                const float nn0 = -n0 * inv_len, nn1 = -n1 * inv_len, nn2 = -n2 * inv_len;

                Tr = Tr / (1.f - alpha);
                const float dchannel_dcolor = alpha * Tr;

                float dL_dalpha = 0.0f;
                const float c0 = q3.x, c1 = q3.y, c2 = q3.z;
                accum_rec0 = last_alpha * last_c0 + (1.f - last_alpha) * accum_rec0; last_c0 = c0;
                dL_dalpha += (c0 - accum_rec0) * dpx0;
                g_col0 = dchannel_dcolor * dpx0;
                accum_rec1 = last_alpha * last_c1 + (1.f - last_alpha) * accum_rec1; last_c1 = c1;
                dL_dalpha += (c1 - accum_rec1) * dpx1;
                g_col1 = dchannel_dcolor * dpx1;
                accum_rec2 = last_alpha * last_c2 + (1.f - last_alpha) * accum_rec2; last_c2 = c2;
                dL_dalpha += (c2 - accum_rec2) * dpx2;
                g_col2 = dchannel_dcolor * dpx2;

                float dL_dmax_t = 0.0f;
                dL_dmax_t += 2.0f * (Tr * alpha) * (mapped_max_t * final_A - final_D) * dL_dreg * dmax_t_dd;
                dL_dalpha += 0.f - 0.f;

                acc_n0 = last_alpha * last_n0 + (1.f - last_alpha) * acc_n0; last_n0 = nn0;
                dL_dalpha += (nn0 - acc_n0) * dn0;
                const float dnn0 = alpha * Tr * dn0;
                acc_n1 = last_alpha * last_n1 + (1.f - last_alpha) * acc_n1; last_n1 = nn1;
                dL_dalpha += (nn1 - acc_n1) * dn1;
                const float dnn1 = alpha * Tr * dn1;
                acc_n2 = last_alpha * last_n2 + (1.f - last_alpha) * acc_n2; last_n2 = nn2;
                dL_dalpha += (nn2 - acc_n2) * dn2;
                const float dnn2 = alpha * Tr * dn2;

                float dL_dlength = (dnn0 * n0 + dnn1 * n1 + dnn2 * n2);
                dL_dlength *= inv_len * inv_len;
                float dLn0 = (-dnn0 + dL_dlength * n0) * inv_len;
                float dLn1 = (-dnn1 + dL_dlength * n1) * inv_len;
                float dLn2 = (-dnn2 + dL_dlength * n2) * inv_len;

                float dL_dt = dL_dmax_t;
                if (contributor == max_contributor - 1)
                    dL_dt += dL_dmax_depth;

                dL_dalpha *= Tr;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

                const float dL_dG = con.w * dL_dalpha;
                const float gdx = G * d_x;
                const float gdy = G * d_y;
                const float dG_ddelx = -gdx * con.x - gdy * con.y;
                const float dG_ddely = -gdy * con.z - gdx * con.y;
                g_mx = dL_dG * dG_ddelx * ddelx_dx;
                g_my = dL_dG * dG_ddely * ddely_dy;
                g_mz = fabsf(dL_dG * dG_ddelx * ddelx_dx) + fabsf(dL_dG * dG_ddely * ddely_dy);
                g_op = G * dL_dalpha;

                const float dL_dpower = dL_dG * G;
                const float dL_dmin_value = dL_dpower * -0.5f;
                // dL/dA = dL/dmin (B/A)^2 / 4 + dL/dt B / (2 A^2),  dL/dB = -dL/dmin B / (2 A) - dL/dt / (2 A): with qf = B/A = -2 t
                const float qf = -2.0f * t, inv_a = 1.0f / aaf;
                float dL_dA = dL_dmin_value * qf * qf * 0.25f;
                float dL_dB = dL_dmin_value * (-0.5f * qf);
                const float dL_dC = dL_dmin_value;
                dL_dA += dL_dt * (0.5f * qf * inv_a);
                dL_dB += dL_dt * (-0.5f * inv_a);
                dLn0 += dL_dA * ray_x;
                dLn1 += dL_dA * ray_y;
                dLn2 += dL_dA;

                g_v0 = dLn0 * ray_x;
                g_v1 = dLn0 * ray_y + dLn1 * ray_x;
                g_v2 = dLn0 + dLn2 * ray_x;
                g_v3 = dLn1 * ray_y;
                g_v4 = dLn1 + dLn2 * ray_y;
                g_v5 = dLn2;
                g_v6 = dL_dB * 2 * ray_x;
                g_v7 = dL_dB * 2 * ray_y;
                g_v8 = dL_dB * 2;
                g_v9 = dL_dC;
            }

            // sum the 17 partials over the wave's 64 pixels on the VALU: transposed across the rows (see pair32 / pair16), then
            // row-local DPP scans; lane 15 of row r holds the totals listed in its column
            //                     row 0          row 1          row 2          row 3
            const float f0 = row_total(pair16(pair32(g_col0, g_col1), pair32(g_col2, g_mx)));     // col0   col2   col1   mean2D.x
            const float f1 = row_total(pair16(pair32(g_my, g_mz), pair32(g_op, 0.0f)));           // m2D.y  opacity m2D.z  -
            // (the 64 partials of a wave are summed in float32 -- a fixed order, 64 terms; across waves, tiles and views the sums are
            // accumulated in float64, which is what keeps the per-Gaussian stage reproducible where the reference's float32 atomics are not)
            const float d0 = row_total(pair16(pair32(g_v0, g_v1), pair32(g_v2, g_v3)));   // v0 v2 v1 v3
            const float d1 = row_total(pair16(pair32(g_v4, g_v5), pair32(g_v6, g_v7)));   // v4 v6 v5 v7
            const float d2 = row_total(pair16(pair32(g_v8, g_v9), 0.0f));                 // v8 -  v9 -
            if ((lane & 15u) == 15u) {
                const unsigned row = lane >> 4;
                const unsigned id = staged_id[j];
                const size_t gi = vP + id;
                float* c = dL_dcolors + gi * 3;
                float* m = dL_dmean2D + gi * 3;
                unsafeAtomicAdd(row == 0 ? c : row == 1 ? c + 2 : row == 2 ? c + 1 : m, f0);
                if (row < 3) unsafeAtomicAdd(row == 0 ? m + 1 : row == 1 ? dL_dopacity + id : m + 2, f1);
                double* acc = dL_dv2g_acc + gi * 10;
                const unsigned perm = row == 0 ? 0u : row == 1 ? 2u : row == 2 ? 1u : 3u;
                unsafeAtomicAdd(acc + perm, (double)d0);
                unsafeAtomicAdd(acc + 4 + perm, (double)d1);
                if ((row & 1u) == 0) unsafeAtomicAdd(acc + 8 + (row >> 1), (double)d2);
            }
        }
    }
    if (lane == 0 && n_pairs)
        atomicAdd(&hdr->bwd_pairs, (unsigned long long)n_pairs);
}

#endif // F3DG_LAB (render_bwd_lockstep_kernel, render_bwd_kernel)

// ---- render3's counterpart: ONE wave64 per 8x8 quadrant, no workgroup barriers (the default) ------------------------------------
// Same arithmetic per contributing (pixel, Gaussian) pair and the same transposed wave reduction as render_bwd_kernel above. What
// changes is everything around it, as in the forward (f3dg_render.hip, render3_fwd_kernel):
//   * a workgroup is one wave that owns a quadrant from its deepest last contributor back to the first list entry; nothing is
//     shared with the other quadrants of the tile and nothing waits at a barrier;
//   * the wave scans the tile's list BACKWARDS 64 ids at a time and keeps the entries whose quadrant bit is set (F3DG_ID_BITS);
//   * a window of 64 kept entries is staged (record by global_load_lds, 2D conic, centre) and tested with the Gaussians across the
//     lanes against the conservative ellipse (quad_ballots_any): every pixel gets its pass mask and the wave the mask of entries
//     that can reach ANY of its pixels -- the others are never looked at (the four-wave kernel computes the reference's a, b and
//     the K pre-test for all 64 pixels of every box-culled entry before it can skip one);
//   * the surviving entries are walked in lock-step, back to front, because the 17 partials of a Gaussian are reduced across the
//     wave before they touch memory.
template <int OCC>
__global__ void __launch_bounds__(64, OCC)
render3_bwd_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
                   F3dgHeader* __restrict__ hdr, const uint2* __restrict__ ranges,
                   const unsigned* __restrict__ point_list_general, const unsigned* __restrict__ small_list, const F3dgRec* __restrict__ rec,
                   const float4* __restrict__ cull,
                   const float2* __restrict__ means2D, const float4* __restrict__ conic,
                   const float* __restrict__ background, int bg_per_view,
                   const float* __restrict__ final_T, const unsigned* __restrict__ n_contrib,
                   const float* __restrict__ dL_dpixels,
                   float* __restrict__ dL_dmean2D, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors,
                   double* __restrict__ dL_dv2g_acc, int debug_no_atomics)
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
    // no lists to walk after an overflow -- and none that belong to this call when the workspace's last forward kept no auxiliary
    // planes (an inference call): all gradients stay zero, the header says why
    // (a one-view forward with auxiliary planes may have taken the small-call path: its lists live in the per-tile slots)
    const unsigned* __restrict__ point_list = hdr->small_path != 0u ? small_list : point_list_general;
    if (hdr->overflow || hdr->save_aux == 0u) {
        range = make_uint2(0, 0);
        if (!hdr->overflow && blockIdx.x == 0 && threadIdx.x == 0) const_cast<F3dgHeader*>(hdr)->bwd_stale = 1u;
    }

    __shared__ float4 sR[4][64];          // records of the window, [16-byte chunk][entry] (global_load_lds image)
    __shared__ float4 sC[64];             // 2D conic + opacity * coef
    __shared__ float2 sX[64];             // projected centre
    __shared__ uint2 sQ[128];             // (list position, Gaussian id) of the kept entries, ring

    const bool alpha_fast = hdr->alpha_fast != 0;
    const size_t vP = (size_t)view * P;
    const F3dgRec* vrec = rec + vP;
    const float4* vcull = cull + vP;
    const float* fT = final_T + (size_t)view * 4 * HW;
    const unsigned* nc = n_contrib + (size_t)view * 2 * HW;
    const float* dpix = dL_dpixels + (size_t)view * F3DG_OUT_CHANNELS * HW;
    const float* bg = background + (bg_per_view ? 3 * view : 0);

    const float T_final = inside ? fT[pix_id] : 0;
    float Tr = T_final;
    const float final_D = inside ? fT[pix_id + HW] : 0;
    const float final_A = 1 - T_final;
    const float dL_dreg = inside ? dpix[8 * HW + pix_id] : 0;

    const int last_contributor = inside ? (int)nc[pix_id] : 0;
    const int max_contributor = inside ? (int)nc[pix_id + HW] : 0;
    float accum_rec0 = 0, accum_rec1 = 0, accum_rec2 = 0;
    float dpx0 = 0, dpx1 = 0, dpx2 = 0, dn0 = 0, dn1 = 0, dn2 = 0, dL_dmax_depth = 0;
    if (inside) {
        dpx0 = dpix[pix_id]; dpx1 = dpix[HW + pix_id]; dpx2 = dpix[2 * HW + pix_id];
        dn0 = dpix[3 * HW + pix_id]; dn1 = dpix[4 * HW + pix_id]; dn2 = dpix[5 * HW + pix_id];
        dL_dmax_depth = dpix[6 * HW + pix_id];
    }
    float last_alpha = 0;
    float last_c0 = 0, last_c1 = 0, last_c2 = 0;
    float last_n0 = 0, last_n1 = 0, last_n2 = 0;
    float acc_n0 = 0, acc_n1 = 0, acc_n2 = 0;
    const float ddelx_dx = (float)(0.5 * W);
    const float ddely_dy = (float)(0.5 * H);
    const float bg_dot_dpixel = bg[0] * dpx0 + bg[1] * dpx1 + bg[2] * dpx2;

    // entries at or behind a pixel's last contributor are skipped by the reference one by one (backward.cu:745-746): the wave starts
    // at the deepest last contributor of ITS 64 pixels
    const int wave_last = min((int)__builtin_amdgcn_readfirstlane((int)__reduce_max_sync(~0ull, last_contributor)),
                              (int)(range.y - range.x));
    const unsigned qbit = 1u << (F3DG_ID_BITS + quad);
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned n_pairs = 0;                 // contributing (pixel, Gaussian) pairs of this wave

    // Where lane 15 of row r adds its totals (see the reduction below): everything but the Gaussian id is fixed per lane, so the
    // address of an atomic is one multiply-add -- base + id * stride -- instead of per-row selects among four arrays.
    const unsigned row = lane >> 4;
    char* const add0 = row < 3 ? reinterpret_cast<char*>(dL_dcolors + vP * 3 + row) : reinterpret_cast<char*>(dL_dmean2D + vP * 3);   // stride 12
    char* const add1 = row < 2 ? reinterpret_cast<char*>(dL_dmean2D + vP * 3 + 1 + row) : reinterpret_cast<char*>(dL_dopacity);
    const unsigned stride1 = row < 2 ? 12u : 4u;
    char* const addD = reinterpret_cast<char*>(dL_dv2g_acc + vP * 10 + row);                                                           // stride 80

    unsigned cursor = (unsigned)wave_last, qhead = 0, qcount = 0;     // list positions [0, cursor) are still to be scanned
    unsigned idn = lane < cursor ? point_list[range.x + cursor - 1u - lane] : 0u;       // back to front: lane l reads position cursor - 1 - l
    for (;;) {
        while (qcount < 64u && cursor != 0u) {
            const unsigned idm = idn;
            const bool valid = lane < cursor;
            const unsigned pos = cursor - 1u - lane;
            cursor = cursor > 64u ? cursor - 64u : 0u;
            idn = lane < cursor ? point_list[range.x + cursor - 1u - lane] : 0u;
            const bool keep = valid && (idm & qbit) != 0u;
            const unsigned long long kb = __ballot(keep);
            if (keep) sQ[(qhead + qcount + (unsigned)__popcll(kb & lt)) & 127u] = make_uint2(pos, idm & F3DG_ID_MASK);
            qcount += (unsigned)__popcll(kb);
        }
        if (qcount == 0u)
            break;
        const unsigned m = qcount < 64u ? qcount : 64u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        float4 e4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float ec = 0.0f;
        if (lane < m) {
            const unsigned id = sQ[(qhead + lane) & 127u].y;
            const float4* src = reinterpret_cast<const float4*>(vrec + id);
#pragma unroll
            for (int c = 0; c < 4; c++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c),
                                                 (__attribute__((address_space(3))) void*)&sR[c][0], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(conic + vP + id),
                                             (__attribute__((address_space(3))) void*)&sC[0], 16, 0, 0);
            e4 = vcull[id];
            sX[lane] = means2D[vP + id];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane < m) ec = sR[3][lane].w;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- phase 1: lane e tests entry e against the 64 pixels of the quadrant
        int pass_lo = 0, pass_hi = 0;
        unsigned long long any = 0ull;
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
            quad_ballots_any<0>(pass_lo, pass_hi, any, fmaf(dxx[0], fmaf(e4.w, dyy[0], adx[0]), cdy[0]), dxx, adx, dyy, cdy, e4.w);
        }
        const unsigned long long pass = ((unsigned long long)(unsigned)pass_hi << 32) | (unsigned)pass_lo;

        // ---- lock-step walk over the entries that reach at least one pixel, back to front (window slot order)
        unsigned long long todo = any;
        while (todo != 0ull) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint2 pe = sQ[(qhead + (unsigned)j) & 127u];
            const int contributor = (int)pe.x;                             // 0-based position from the front
            bool active = inside && ((pass >> j) & 1ull) != 0ull && contributor < last_contributor;
            if (__ballot(active) == 0)
                continue;

            const float4 q0 = sR[0][j], q1 = sR[1][j], q2 = sR[2][j];
            const float n0 = q0.x * ray_x + q0.y * ray_y + q0.z;
            const float n1 = q0.y * ray_x + q0.w * ray_y + q1.x;
            const float n2 = q0.z * ray_x + q1.x * ray_y + q1.y;
            const float aaf = ray_x * n0 + ray_y * n1 + n2;
            const float bhalf = q1.z * ray_x + q1.w * ray_y + q2.x;

            const float CC = q2.y;
            float t = 0, G = 0, alpha = 0;
            if (active) {
                if (alpha_fast) {
                    // the forward of this workspace took the fast arithmetic: the same function, to the bit
                    f3dg_fast_t_G(aaf, bhalf, CC, t, G);
                    if (t < 0.2f) active = false;
                    alpha = fminf(0.99f, q2.z * G);
                } else {
                    const double AA = aaf;
                    const double BB = 2 * bhalf;
                    const double q = BB / AA;                          // one division: -BB / (2 * AA) == -0.5 * (BB / AA) exactly
                    t = (float)(-0.5 * q);
                    if (t <= F3DG_NEAR_PLANE) active = false;
                    const double min_value = -q * (BB / 4.) + CC;
                    float power = (float)(-0.5f * min_value);
                    if (power > 0.0f) power = 0.0f;
                    G = expf(power);
                    alpha = fminf(0.99f, q2.z * G);
                }
                if (alpha < 1.0f / 255.0f) active = false;
            }
            {
                const unsigned long long act = __ballot(active);
                if (act == 0)
                    continue;
                n_pairs += (unsigned)__popcll(act);
            }

            float g_col0 = 0, g_col1 = 0, g_col2 = 0, g_mx = 0, g_my = 0, g_mz = 0, g_op = 0;
            float g_v0 = 0, g_v1 = 0, g_v2 = 0, g_v3 = 0, g_v4 = 0, g_v5 = 0, g_v6 = 0, g_v7 = 0, g_v8 = 0, g_v9 = 0;
            if (active) {
                // gradient terms only from here (see render_bwd_kernel): float32 with one reciprocal each, FMA contraction allowed
#pragma clang fp contract(fast)
                const float4 q3 = sR[3][j];
                const float4 con = sC[j];
                const float2 xy = sX[j];
                const float d_x = (float)(xy.x - (pixf_x - 0.5)), d_y = (float)(xy.y - (pixf_y - 0.5));

                // (hardware reciprocals / reciprocal square root, <= 1 ulp: these feed float32 gradient sums only)
                const float inv_t = __builtin_amdgcn_rcpf(t);
                const float mapped_max_t = fmaf(-0.20040080160320642f, inv_t, 1.0020040080160322f);
                const float dmax_t_dd = 0.20040080160320642f * inv_t * inv_t;
                const float inv_len = __builtin_amdgcn_rsqf(n0 * n0 + n1 * n1 + n2 * n2 + 1e-7f);
                const float nn0 = -n0 * inv_len, nn1 = -n1 * inv_len, nn2 = -n2 * inv_len;

                // T is rebuilt by the reference's own IEEE division (backward.cu:803): a reciprocal + Newton step (<= 1 ulp per layer)
                // keeps the compositing-stage gradients within 1e-5, but the per-Gaussian stage amplifies even that on ill-conditioned
                // anisotropic scenes (dL/dscale of tests/test_raster_backward_gpu.py B2 moved from 2x to 3x the oracle's own error)
                const float oma = 1.f - alpha;
                Tr = Tr / oma;
                const float inv_oma = __builtin_amdgcn_rcpf(oma);      // background term only
                const float dchannel_dcolor = alpha * Tr;

                float dL_dalpha = 0.0f;
                const float c0 = q3.x, c1 = q3.y, c2 = q3.z;
                accum_rec0 = last_alpha * last_c0 + (1.f - last_alpha) * accum_rec0; last_c0 = c0;
                dL_dalpha += (c0 - accum_rec0) * dpx0;
                g_col0 = dchannel_dcolor * dpx0;
                accum_rec1 = last_alpha * last_c1 + (1.f - last_alpha) * accum_rec1; last_c1 = c1;
                dL_dalpha += (c1 - accum_rec1) * dpx1;
                g_col1 = dchannel_dcolor * dpx1;
                accum_rec2 = last_alpha * last_c2 + (1.f - last_alpha) * accum_rec2; last_c2 = c2;
                dL_dalpha += (c2 - accum_rec2) * dpx2;
                g_col2 = dchannel_dcolor * dpx2;

                float dL_dmax_t = 0.0f;
                dL_dmax_t += 2.0f * (Tr * alpha) * (mapped_max_t * final_A - final_D) * dL_dreg * dmax_t_dd;
                dL_dalpha += 0.f - 0.f;

                acc_n0 = last_alpha * last_n0 + (1.f - last_alpha) * acc_n0; last_n0 = nn0;
                dL_dalpha += (nn0 - acc_n0) * dn0;
                const float dnn0 = alpha * Tr * dn0;
                acc_n1 = last_alpha * last_n1 + (1.f - last_alpha) * acc_n1; last_n1 = nn1;
                dL_dalpha += (nn1 - acc_n1) * dn1;
                const float dnn1 = alpha * Tr * dn1;
                acc_n2 = last_alpha * last_n2 + (1.f - last_alpha) * acc_n2; last_n2 = nn2;
                dL_dalpha += (nn2 - acc_n2) * dn2;
                const float dnn2 = alpha * Tr * dn2;

                float dL_dlength = (dnn0 * n0 + dnn1 * n1 + dnn2 * n2);
                dL_dlength *= inv_len * inv_len;
                float dLn0 = (-dnn0 + dL_dlength * n0) * inv_len;
                float dLn1 = (-dnn1 + dL_dlength * n1) * inv_len;
                float dLn2 = (-dnn2 + dL_dlength * n2) * inv_len;

                float dL_dt = dL_dmax_t;
                if (contributor == max_contributor - 1)
                    dL_dt += dL_dmax_depth;

                dL_dalpha *= Tr;
                last_alpha = alpha;
                dL_dalpha += (-T_final * inv_oma) * bg_dot_dpixel;

                const float dL_dG = con.w * dL_dalpha;
                const float gdx = G * d_x;
                const float gdy = G * d_y;
                const float dG_ddelx = -gdx * con.x - gdy * con.y;
                const float dG_ddely = -gdy * con.z - gdx * con.y;
                g_mx = dL_dG * dG_ddelx * ddelx_dx;
                g_my = dL_dG * dG_ddely * ddely_dy;
                g_mz = fabsf(dL_dG * dG_ddelx * ddelx_dx) + fabsf(dL_dG * dG_ddely * ddely_dy);
                g_op = G * dL_dalpha;

                const float dL_dpower = dL_dG * G;
                const float dL_dmin_value = dL_dpower * -0.5f;
                const float qf = -2.0f * t, inv_a = __builtin_amdgcn_rcpf(aaf);
                float dL_dA = dL_dmin_value * qf * qf * 0.25f;
                float dL_dB = dL_dmin_value * (-0.5f * qf);
                const float dL_dC = dL_dmin_value;
                dL_dA += dL_dt * (0.5f * qf * inv_a);
                dL_dB += dL_dt * (-0.5f * inv_a);
                dLn0 += dL_dA * ray_x;
                dLn1 += dL_dA * ray_y;
                dLn2 += dL_dA;

                g_v0 = dLn0 * ray_x;
                g_v1 = dLn0 * ray_y + dLn1 * ray_x;
                g_v2 = dLn0 + dLn2 * ray_x;
                g_v3 = dLn1 * ray_y;
                g_v4 = dLn1 + dLn2 * ray_y;
                g_v5 = dLn2;
                g_v6 = dL_dB * 2 * ray_x;
                g_v7 = dL_dB * 2 * ray_y;
                g_v8 = dL_dB * 2;
                g_v9 = dL_dC;
            }

            // the 17 partials summed over the wave's 64 pixels (pair32 / pair16, then row-local DPP); the values are paired so that
            // lane 15 of row r holds element r of each target array:
            //                     row 0          row 1          row 2          row 3
            const float f0 = row_total(pair16(pair32(g_col0, g_col2), pair32(g_col1, g_mx)));     // col0   col1   col2   mean2D.x
            const float f1 = row_total(pair16(pair32(g_my, g_op), pair32(g_mz, 0.0f)));           // m2D.y  m2D.z  opacity  -
            const float d0 = row_total(pair16(pair32(g_v0, g_v2), pair32(g_v1, g_v3)));           // v0 v1 v2 v3
            const float d1 = row_total(pair16(pair32(g_v4, g_v6), pair32(g_v5, g_v7)));           // v4 v5 v6 v7
            const float d2 = row_total(pair16(pair32(g_v8, 0.0f), pair32(g_v9, 0.0f)));           // v8 v9 -  -
            if ((lane & 15u) == 15u && !debug_no_atomics) {
                const size_t id = pe.y;
                unsafeAtomicAdd(reinterpret_cast<float*>(add0 + id * 12u), f0);
                if (row < 3) unsafeAtomicAdd(reinterpret_cast<float*>(add1 + id * stride1), f1);
                double* acc = reinterpret_cast<double*>(addD + id * 80u);
                unsafeAtomicAdd(acc, (double)d0);
                unsafeAtomicAdd(acc + 4, (double)d1);
                if (row < 2) unsafeAtomicAdd(acc + 8, (double)d2);
            }
        }
        qhead += m;
        qcount -= m;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the window's slots are rewritten by the next one
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (lane == 0 && n_pairs)
        atomicAdd(&hdr->bwd_pairs, (unsigned long long)n_pairs);
}

struct M3 { float m[3][3]; };
struct M4 { float m[4][4]; };
__device__ __forceinline__ M3 mul(const M3& a, const M3& b)
{
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int q = 0; q < 3; q++)
            r.m[c][q] = a.m[0][q] * b.m[c][0] + a.m[1][q] * b.m[c][1] + a.m[2][q] * b.m[c][2];
    return r;
}
__device__ __forceinline__ M3 transpose(const M3& a)
{
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int q = 0; q < 3; q++)
            r.m[c][q] = a.m[q][c];
    return r;
}
__device__ __forceinline__ M4 mul(const M4& a, const M4& b)
{
    M4 r;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int q = 0; q < 4; q++)
            r.m[c][q] = a.m[0][q] * b.m[c][0] + a.m[1][q] * b.m[c][1] + a.m[2][q] * b.m[c][2] + a.m[3][q] * b.m[c][3];
    return r;
}

__device__ __constant__ float B_SH_C0 = 0.28209479177387814f;
__device__ __constant__ float B_SH_C1 = 0.4886025119029199f;
__device__ __constant__ float B_SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                             -1.0925484305920792f, 0.5462742152960396f };
__device__ __constant__ float B_SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                             0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                             -0.5900435899266435f };

// One thread per GAUSSIAN, looping over the views of the call: view2gaussian backward + SH backward. The per-Gaussian
// parameter gradients (mean, scale, rotation, SH) are summed over the views in registers and added to the caller's
// zero-filled outputs once, by their single writer: no atomics (a (view, Gaussian) grid with 22+ float atomics per
// thread measured 4.4 ms at 1 M Gaussians x 8 views, all of it contention), and a deterministic view order.
__global__ void __launch_bounds__(F3DG_BLOCK)
preprocess_bwd_kernel(int P, int D, int M, const float* __restrict__ means3D, const int* __restrict__ radii,
                      const float* __restrict__ shs, const unsigned char* __restrict__ clamped,
                      const float* __restrict__ scales, const float* __restrict__ rotations,
                      const float* __restrict__ viewmatrices, const float* __restrict__ cam_positions,
                      const double* __restrict__ dL_dv2g_acc, float* __restrict__ dL_dv2g_out,
                      float* __restrict__ dL_dcolor, float* __restrict__ dL_dmeans, float* __restrict__ dL_dsh,
                      float* __restrict__ dL_dscale, float* __restrict__ dL_drot, int V,
                      int acc_stride /* doubles per (view, Gaussian) of dL_dv2g_acc: 10, or 16 = the packed records of the dense compositing backward */,
                      float* __restrict__ dL_dmean2D, float* __restrict__ dL_dopacity)
{
    const int g = blockIdx.x * F3DG_BLOCK + threadIdx.x;
    if (g >= P) return;
    float sum_mean[3] = { 0, 0, 0 }, sum_scale[3] = { 0, 0, 0 }, sum_rot[4] = { 0, 0, 0, 0 };
    float sum_sh[48];
#pragma unroll
    for (int i = 0; i < 48; i++) sum_sh[i] = 0.0f;
    bool any = false;
    float op_sum = 0.0f;

    for (int v = 0; v < V; v++) {
    const size_t idx = (size_t)v * P + g;

    float dv[10];
    const double* arec = dL_dv2g_acc + idx * (size_t)acc_stride;
    if (acc_stride == 16) {
        // the dense compositing backward (f3dg_backward5.hip) adds colour, mean2D and opacity into the same 128-byte record as the ten
        // float64 sums (one line per atomic event instead of four arrays): the record is read as seven 16-byte words, its float32 sums
        // leave for the caller's arrays here, the opacity summed over the views by its single writer
        const double2* a2 = reinterpret_cast<const double2*>(arec);
        const double2 d0 = a2[0], d1 = a2[1], d2 = a2[2], d3 = a2[3], d4 = a2[4];
        const float4 f0 = reinterpret_cast<const float4*>(arec)[5], f1 = reinterpret_cast<const float4*>(arec)[6];
        dv[0] = (float)d0.x; dv[1] = (float)d0.y; dv[2] = (float)d1.x; dv[3] = (float)d1.y; dv[4] = (float)d2.x;
        dv[5] = (float)d2.y; dv[6] = (float)d3.x; dv[7] = (float)d3.y; dv[8] = (float)d4.x; dv[9] = (float)d4.y;
        dL_dcolor[idx * 3] = f0.x; dL_dcolor[idx * 3 + 1] = f0.y; dL_dcolor[idx * 3 + 2] = f0.z;
        dL_dmean2D[idx * 3] = f0.w; dL_dmean2D[idx * 3 + 1] = f1.x; dL_dmean2D[idx * 3 + 2] = f1.y;
        op_sum += f1.z;
    } else {
#pragma unroll
        for (int i = 0; i < 10; i++) dv[i] = (float)arec[i];
    }
#pragma unroll
    for (int i = 0; i < 10; i++) dL_dv2g_out[idx * 10 + i] = dv[i];
    if (!(radii[idx] > 0)) continue;
    any = true;
    const float* view = viewmatrices + 16 * v;

    float dmean[3] = { 0, 0, 0 };
    if (scales && rotations) {
        const float sx = scales[3 * (size_t)g], sy = scales[3 * (size_t)g + 1], sz = scales[3 * (size_t)g + 2];
        const float4 rot = reinterpret_cast<const float4*>(rotations)[g];
        const float mx = means3D[3 * (size_t)g], my = means3D[3 * (size_t)g + 1], mz = means3D[3 * (size_t)g + 2];
        const float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
        M3 R;
        R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[0][1] = 2.f * (x * y - r * z);       R.m[0][2] = 2.f * (x * z + r * y);
        R.m[1][0] = 2.f * (x * y + r * z);       R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[1][2] = 2.f * (y * z - r * x);
        R.m[2][0] = 2.f * (x * z - r * y);       R.m[2][1] = 2.f * (y * z + r * x);       R.m[2][2] = 1.f - 2.f * (x * x + y * y);

        M4 G2W, W2V;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            G2W.m[c][0] = R.m[0][c]; G2W.m[c][1] = R.m[1][c]; G2W.m[c][2] = R.m[2][c]; G2W.m[c][3] = 0.0f;
        }
        G2W.m[3][0] = mx; G2W.m[3][1] = my; G2W.m[3][2] = mz; G2W.m[3][3] = 1.0f;
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                W2V.m[c][q] = view[4 * c + q];
        const M4 G2V = mul(W2V, G2W);

        M3 Rt;
#pragma unroll
        for (int c = 0; c < 3; c++) { Rt.m[c][0] = G2V.m[0][c]; Rt.m[c][1] = G2V.m[1][c]; Rt.m[c][2] = G2V.m[2][c]; }
        const float tx = G2V.m[3][0], ty = G2V.m[3][1], tz = G2V.m[3][2];
        float t2[3];
        t2[0] = (-Rt.m[0][0]) * tx + (-Rt.m[1][0]) * ty + (-Rt.m[2][0]) * tz;
        t2[1] = (-Rt.m[0][1]) * tx + (-Rt.m[1][1]) * ty + (-Rt.m[2][1]) * tz;
        t2[2] = (-Rt.m[0][2]) * tx + (-Rt.m[1][2]) * ty + (-Rt.m[2][2]) * tz;

        const double S[3] = { 1.0f / ((double)sx * sx + 1e-7), 1.0f / ((double)sy * sy + 1e-7), 1.0f / ((double)sz * sz + 1e-7) };
        M3 SR;
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int q = 0; q < 3; q++)
                SR.m[c][q] = (float)(S[q] * Rt.m[c][q]);

        M3 dSig;
        dSig.m[0][0] = dv[0];        dSig.m[0][1] = 0.5f * dv[1]; dSig.m[0][2] = 0.5f * dv[2];
        dSig.m[1][0] = 0.5f * dv[1]; dSig.m[1][1] = dv[3];        dSig.m[1][2] = 0.5f * dv[4];
        dSig.m[2][0] = 0.5f * dv[2]; dSig.m[2][1] = 0.5f * dv[4]; dSig.m[2][2] = dv[5];
        const float dB[3] = { dv[6], dv[7], dv[8] };
        const float dC = dv[9];

        M3 Dm = mul(Rt, dSig);
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                Dm.m[i][j] = Dm.m[i][j] + t2[j] * dB[i];
        M3 dRt = transpose(mul(dSig, transpose(SR)));
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int q = 0; q < 3; q++)
                dRt.m[c][q] = dRt.m[c][q] + (float)(S[q] * Dm.m[c][q]);

        float dS[3], dt2[3];
#pragma unroll
        for (int q = 0; q < 3; q++)
            dS[q] = Dm.m[0][q] * Rt.m[0][q] + Dm.m[1][q] * Rt.m[1][q] + Dm.m[2][q] * Rt.m[2][q];
#pragma unroll
        for (int q = 0; q < 3; q++)
            dt2[q] = (float)(2 * t2[q] * S[q] * dC + dB[0] * SR.m[0][q] + dB[1] * SR.m[1][q] + dB[2] * SR.m[2][q]);
#pragma unroll
        for (int q = 0; q < 3; q++)
            dS[q] += dC * t2[q] * t2[q];
        const float sc[3] = { sx, sy, sz };
#pragma unroll
        for (int q = 0; q < 3; q++)
            sum_scale[q] += (float)(-2 / sc[q] * S[q] * dS[q]);

        const M3 dV2G_R_t = transpose(dRt);
        M3 dG2V_R;
        const float tv[3] = { tx, ty, tz };
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int q = 0; q < 3; q++)
                dG2V_R.m[c][q] = dV2G_R_t.m[c][q] + (-dt2[c] * tv[q]);
        const float nd[3] = { -dt2[0], -dt2[1], -dt2[2] };
        float dG2V_t[3];
#pragma unroll
        for (int c = 0; c < 3; c++)
            dG2V_t[c] = Rt.m[c][0] * nd[0] + Rt.m[c][1] * nd[1] + Rt.m[c][2] * nd[2];

        M4 dG2V, W2Vt;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            dG2V.m[c][0] = dG2V_R.m[c][0]; dG2V.m[c][1] = dG2V_R.m[c][1]; dG2V.m[c][2] = dG2V_R.m[c][2]; dG2V.m[c][3] = 0.0f;
        }
        dG2V.m[3][0] = dG2V_t[0]; dG2V.m[3][1] = dG2V_t[1]; dG2V.m[3][2] = dG2V_t[2]; dG2V.m[3][3] = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                W2Vt.m[c][q] = W2V.m[q][c];
        const M4 dG2W = mul(W2Vt, dG2V);

        dmean[0] = dG2W.m[3][0]; dmean[1] = dG2W.m[3][1]; dmean[2] = dG2W.m[3][2];
        float Mt[3][3];
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int q = 0; q < 3; q++)
                Mt[c][q] = dG2W.m[c][q];
        const float q0 = 2 * z * (Mt[0][1] - Mt[1][0]) + 2 * y * (Mt[2][0] - Mt[0][2]) + 2 * x * (Mt[1][2] - Mt[2][1]);
        const float q1 = 2 * y * (Mt[1][0] + Mt[0][1]) + 2 * z * (Mt[2][0] + Mt[0][2]) + 2 * r * (Mt[1][2] - Mt[2][1]) - 4 * x * (Mt[2][2] + Mt[1][1]);
        const float q2 = 2 * x * (Mt[1][0] + Mt[0][1]) + 2 * r * (Mt[2][0] - Mt[0][2]) + 2 * z * (Mt[1][2] + Mt[2][1]) - 4 * y * (Mt[2][2] + Mt[0][0]);
        const float q3 = 2 * r * (Mt[0][1] - Mt[1][0]) + 2 * x * (Mt[2][0] + Mt[0][2]) + 2 * y * (Mt[1][2] + Mt[2][1]) - 4 * z * (Mt[1][1] + Mt[0][0]);
        sum_rot[0] += q0; sum_rot[1] += q1; sum_rot[2] += q2; sum_rot[3] += q3;
    }

    if (shs) {
        const float* campos = cam_positions + 3 * v;
        const float o0 = means3D[3 * (size_t)g] - campos[0], o1 = means3D[3 * (size_t)g + 1] - campos[1], o2 = means3D[3 * (size_t)g + 2] - campos[2];
        const float len = sqrtf(o0 * o0 + o1 * o1 + o2 * o2);
        const float x = o0 / len, y = o1 / len, z = o2 / len;
        const float* sh = shs + (size_t)g * M * 3;
        const unsigned char cl = clamped[idx];
        float dRGB[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
            dRGB[ch] = dL_dcolor[idx * 3 + ch] * ((cl >> ch) & 1 ? 0 : 1);
        float ddx[3] = { 0, 0, 0 }, ddy[3] = { 0, 0, 0 }, ddz[3] = { 0, 0, 0 };
#define F3DG_DSH(k, val) do { const float _w = (val); _Pragma("unroll") for (int ch = 0; ch < 3; ch++) sum_sh[(k) * 3 + ch] += _w * dRGB[ch]; } while (0)
#define F3DG_SH(k, ch) sh[(k) * 3 + (ch)]
        F3DG_DSH(0, B_SH_C0);
        if (D > 0) {
            F3DG_DSH(1, -B_SH_C1 * y);
            F3DG_DSH(2, B_SH_C1 * z);
            F3DG_DSH(3, -B_SH_C1 * x);
            for (int ch = 0; ch < 3; ch++) {
                ddx[ch] = -B_SH_C1 * F3DG_SH(3, ch);
                ddy[ch] = -B_SH_C1 * F3DG_SH(1, ch);
                ddz[ch] = B_SH_C1 * F3DG_SH(2, ch);
            }
            if (D > 1) {
                const float xx = x * x, yy = y * y, zz = z * z;
                const float xy = x * y, yz = y * z, xz = x * z;
                F3DG_DSH(4, B_SH_C2[0] * xy);
                F3DG_DSH(5, B_SH_C2[1] * yz);
                F3DG_DSH(6, B_SH_C2[2] * (2.f * zz - xx - yy));
                F3DG_DSH(7, B_SH_C2[3] * xz);
                F3DG_DSH(8, B_SH_C2[4] * (xx - yy));
                for (int ch = 0; ch < 3; ch++) {
                    ddx[ch] += B_SH_C2[0] * y * F3DG_SH(4, ch) + B_SH_C2[2] * 2.f * -x * F3DG_SH(6, ch) + B_SH_C2[3] * z * F3DG_SH(7, ch) + B_SH_C2[4] * 2.f * x * F3DG_SH(8, ch);
                    ddy[ch] += B_SH_C2[0] * x * F3DG_SH(4, ch) + B_SH_C2[1] * z * F3DG_SH(5, ch) + B_SH_C2[2] * 2.f * -y * F3DG_SH(6, ch) + B_SH_C2[4] * 2.f * -y * F3DG_SH(8, ch);
                    ddz[ch] += B_SH_C2[1] * y * F3DG_SH(5, ch) + B_SH_C2[2] * 2.f * 2.f * z * F3DG_SH(6, ch) + B_SH_C2[3] * x * F3DG_SH(7, ch);
                }
                if (D > 2) {
                    F3DG_DSH(9, B_SH_C3[0] * y * (3.f * xx - yy));
                    F3DG_DSH(10, B_SH_C3[1] * xy * z);
                    F3DG_DSH(11, B_SH_C3[2] * y * (4.f * zz - xx - yy));
                    F3DG_DSH(12, B_SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                    F3DG_DSH(13, B_SH_C3[4] * x * (4.f * zz - xx - yy));
                    F3DG_DSH(14, B_SH_C3[5] * z * (xx - yy));
                    F3DG_DSH(15, B_SH_C3[6] * x * (xx - 3.f * yy));
                    for (int ch = 0; ch < 3; ch++) {
                        ddx[ch] += (
                            B_SH_C3[0] * F3DG_SH(9, ch) * 3.f * 2.f * xy +
                            B_SH_C3[1] * F3DG_SH(10, ch) * yz +
                            B_SH_C3[2] * F3DG_SH(11, ch) * -2.f * xy +
                            B_SH_C3[3] * F3DG_SH(12, ch) * -3.f * 2.f * xz +
                            B_SH_C3[4] * F3DG_SH(13, ch) * (-3.f * xx + 4.f * zz - yy) +
                            B_SH_C3[5] * F3DG_SH(14, ch) * 2.f * xz +
                            B_SH_C3[6] * F3DG_SH(15, ch) * 3.f * (xx - yy));
                        ddy[ch] += (
                            B_SH_C3[0] * F3DG_SH(9, ch) * 3.f * (xx - yy) +
                            B_SH_C3[1] * F3DG_SH(10, ch) * xz +
                            B_SH_C3[2] * F3DG_SH(11, ch) * (-3.f * yy + 4.f * zz - xx) +
                            B_SH_C3[3] * F3DG_SH(12, ch) * -3.f * 2.f * yz +
                            B_SH_C3[4] * F3DG_SH(13, ch) * -2.f * xy +
                            B_SH_C3[5] * F3DG_SH(14, ch) * -2.f * yz +
                            B_SH_C3[6] * F3DG_SH(15, ch) * -3.f * 2.f * xy);
                        ddz[ch] += (
                            B_SH_C3[1] * F3DG_SH(10, ch) * xy +
                            B_SH_C3[2] * F3DG_SH(11, ch) * 4.f * 2.f * yz +
                            B_SH_C3[3] * F3DG_SH(12, ch) * 3.f * (2.f * zz - xx - yy) +
                            B_SH_C3[4] * F3DG_SH(13, ch) * 4.f * 2.f * xz +
                            B_SH_C3[5] * F3DG_SH(14, ch) * (xx - yy));
                    }
                }
            }
        }
#undef F3DG_DSH
#undef F3DG_SH
        const float dd0 = ddx[0] * dRGB[0] + ddx[1] * dRGB[1] + ddx[2] * dRGB[2];
        const float dd1 = ddy[0] * dRGB[0] + ddy[1] * dRGB[1] + ddy[2] * dRGB[2];
        const float dd2 = ddz[0] * dRGB[0] + ddz[1] * dRGB[1] + ddz[2] * dRGB[2];
        // dnormvdv (auxiliary.h:145-155)
        const float sum2 = o0 * o0 + o1 * o1 + o2 * o2;
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        dmean[0] += ((+sum2 - o0 * o0) * dd0 - o1 * o0 * dd1 - o2 * o0 * dd2) * invsum32;
        dmean[1] += (-o0 * o1 * dd0 + (sum2 - o1 * o1) * dd1 - o2 * o1 * dd2) * invsum32;
        dmean[2] += (-o0 * o2 * dd0 - o1 * o2 * dd1 + (sum2 - o2 * o2) * dd2) * invsum32;
    }
    sum_mean[0] += dmean[0]; sum_mean[1] += dmean[1]; sum_mean[2] += dmean[2];
    }   // views

    if (acc_stride == 16) dL_dopacity[g] += op_sum;
    if (!any) return;
#pragma unroll
    for (int q = 0; q < 3; q++) dL_dmeans[3 * (size_t)g + q] += sum_mean[q];
    if (scales && rotations) {
#pragma unroll
        for (int q = 0; q < 3; q++) dL_dscale[3 * (size_t)g + q] += sum_scale[q];
#pragma unroll
        for (int q = 0; q < 4; q++) dL_drot[4 * (size_t)g + q] += sum_rot[q];
    }
    if (shs) {
        float* dsh = dL_dsh + (size_t)g * M * 3;
        const int ncoef = (D + 1) * (D + 1);
#pragma unroll
        for (int i = 0; i < 48; i++)
            if (i < ncoef * 3) dsh[i] += sum_sh[i];
    }
}

} // namespace

extern "C" int f3dg_backward(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                             int n_views, int P, int D, int M, const float* background, int W, int H,
                             const float* means3D, const float* shs, const float* colors_precomp,
                             const float* scales, float scale_modifier, const float* rotations,
                             const float* cov3D_precomp, const float* view2gaussian_precomp,
                             const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                             float tan_fovx, float tan_fovy, float kernel_size,
                             const int* radii, const float* dL_dpix,
                             float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                             float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                             float* dL_dview2gaussian, unsigned flags)
{
    (void)scale_modifier; (void)projmatrix; (void)kernel_size; (void)dL_dconic; (void)dL_dcov3D;
    (void)cov3D_precomp; (void)colors_precomp; (void)view2gaussian_precomp;
    hipStream_t s = (hipStream_t)stream;
    if (n_views <= 0 || P < 0 || W <= 0 || H <= 0 || !workspace || !dL_dpix || !background) return F3DG_ERR_BAD_ARG;
    if (P == 0) return F3DG_OK;
    if (!means3D || !viewmatrix || !cam_pos || !dL_dmean2D || !dL_dopacity || !dL_dcolor || !dL_dmean3D ||
        !dL_dview2gaussian)
        return F3DG_ERR_BAD_ARG;
    if ((scales != nullptr) != (rotations != nullptr)) return F3DG_ERR_BAD_ARG;
    if (scales && (!dL_dscale || !dL_drot)) return F3DG_ERR_BAD_ARG;
    if (shs && !dL_dsh) return F3DG_ERR_BAD_ARG;
    const F3dgLayout L = f3dg_layout(P, W, H, n_views, max_rendered);
    if (workspace_bytes < L.total) return F3DG_ERR_WORKSPACE;
    char* ws = static_cast<char*>(workspace);
    F3dgHeader* hdr = reinterpret_cast<F3dgHeader*>(ws + L.header);
    const int tiles_x = (W + F3DG_TILE - 1) / F3DG_TILE, tiles_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    const int T = tiles_x * tiles_y;
    const float focal_y = H / (2.0f * tan_fovy);
    const float focal_x = W / (2.0f * tan_fovx);
    const int* radii_used = radii ? radii : reinterpret_cast<const int*>(ws + L.radii);

    // float64 accumulator of dL/dview2gaussian (its own region of the workspace)
    double* acc = reinterpret_cast<double*>(ws + L.bwd_acc);
#ifdef F3DG_LAB
    const bool dense = g_f3dg_bwd_dense && g_f3dg_render_cull && g_f3dg_render_kernel == 3;
#else
    const bool dense = g_f3dg_bwd_dense != 0;
#endif
    const int acc_stride = dense ? 16 : 10;           // the dense kernel's 128-byte records (ten float64 + seven float32 sums) or [V*P][10]
    F3DG_HIP_CHECK(hipMemsetAsync(acc, 0, sizeof(double) * (size_t)acc_stride * (size_t)n_views * P, s));
    F3DG_HIP_CHECK(hipMemsetAsync(&hdr->bwd_pairs, 0, sizeof(hdr->bwd_pairs), s));
    if (!dense) {
        // the per-view outputs need no zero-fill by the caller: the dense path writes every element of them from its records
        // (preprocess_bwd_kernel); the lock-step kernels add into these two
        F3DG_HIP_CHECK(hipMemsetAsync(dL_dmean2D, 0, sizeof(float) * 3 * (size_t)n_views * P, s));
        F3DG_HIP_CHECK(hipMemsetAsync(dL_dcolor, 0, sizeof(float) * 3 * (size_t)n_views * P, s));
    }
    const int prof = f3dg_prof_bwd_begin(s);

#ifdef F3DG_LAB
    const bool bwd3 = g_f3dg_render_cull && g_f3dg_render_kernel == 3;
#else
    const bool bwd3 = true;
#endif
    if (dense) {
        const int rc5 = f3dg_launch_render5_bwd(s, n_views, P, W, H, tiles_x, T, focal_x, focal_y, hdr, reinterpret_cast<const uint2*>(ws + L.ranges),
                                                reinterpret_cast<const unsigned*>(ws + L.vals[0]), reinterpret_cast<const unsigned*>(ws + L.small_list),
                                                reinterpret_cast<const F3dgRec*>(ws + L.rec), reinterpret_cast<const float4*>(ws + L.cull),
                                                reinterpret_cast<const float2*>(ws + L.means2D), reinterpret_cast<const float4*>(ws + L.conic), background,
                                                (flags & F3DG_FLAG_BG_PER_VIEW) ? 1 : 0, reinterpret_cast<const float*>(ws + L.final_T),
                                                reinterpret_cast<const unsigned*>(ws + L.n_contrib), dL_dpix, dL_dmean2D, dL_dopacity, dL_dcolor, acc, g_f3dg_small_debug == 9);
        if (rc5 != F3DG_OK) return rc5;
    } else if (bwd3) {
#define F3DG_LAUNCH_BWD3(OCC) F3DG_KLAUNCH((render3_bwd_kernel<OCC>), dim3((unsigned)n_views * (unsigned)T * 4u), dim3(64), 0, s, n_views, P, W, H,  \
                           tiles_x, T, focal_x, focal_y, hdr, reinterpret_cast<const uint2*>(ws + L.ranges),                                        \
                           reinterpret_cast<const unsigned*>(ws + L.vals[0]), reinterpret_cast<const unsigned*>(ws + L.small_list), reinterpret_cast<const F3dgRec*>(ws + L.rec),                        \
                           reinterpret_cast<const float4*>(ws + L.cull),                                                                            \
                           reinterpret_cast<const float2*>(ws + L.means2D), reinterpret_cast<const float4*>(ws + L.conic),                         \
                           background, (flags & F3DG_FLAG_BG_PER_VIEW) ? 1 : 0, reinterpret_cast<const float*>(ws + L.final_T),                    \
                           reinterpret_cast<const unsigned*>(ws + L.n_contrib), dL_dpix, dL_dmean2D, dL_dopacity, dL_dcolor, acc, g_f3dg_small_debug == 9)
        if (g_f3dg_bwd_occ == 4) F3DG_LAUNCH_BWD3(4); else if (g_f3dg_bwd_occ == 6) F3DG_LAUNCH_BWD3(6); else if (g_f3dg_bwd_occ == 3) F3DG_LAUNCH_BWD3(3);
        else if (g_f3dg_bwd_occ == 2) F3DG_LAUNCH_BWD3(2); else F3DG_LAUNCH_BWD3(5);
#undef F3DG_LAUNCH_BWD3
    }
#ifdef F3DG_LAB
    else if (g_f3dg_render_cull)
        F3DG_KLAUNCH(render_bwd_kernel, dim3((unsigned)n_views * (unsigned)T), dim3(F3DG_BLOCK), 0, s, n_views, P, W, H,
                           tiles_x, T, focal_x, focal_y, hdr, reinterpret_cast<const uint2*>(ws + L.ranges),
                           reinterpret_cast<const unsigned*>(ws + L.vals[0]), reinterpret_cast<const unsigned*>(ws + L.small_list), reinterpret_cast<const F3dgRec*>(ws + L.rec),
                           reinterpret_cast<const float4*>(ws + L.bbox),
                           reinterpret_cast<const float2*>(ws + L.means2D), reinterpret_cast<const float4*>(ws + L.conic),
                           background, (flags & F3DG_FLAG_BG_PER_VIEW) ? 1 : 0, reinterpret_cast<const float*>(ws + L.final_T),
                           reinterpret_cast<const unsigned*>(ws + L.n_contrib), dL_dpix, dL_dmean2D, dL_dopacity, dL_dcolor,
                           acc);
    else     // option render_cull = 0: the lock-step kernel (every lane visits every entry, wave butterflies), kept for A/B
        F3DG_KLAUNCH(render_bwd_lockstep_kernel, dim3((unsigned)n_views * (unsigned)T), dim3(F3DG_BLOCK), 0, s, n_views, P, W, H,
                           tiles_x, T, focal_x, focal_y, hdr, reinterpret_cast<const uint2*>(ws + L.ranges),
                           reinterpret_cast<const unsigned*>(ws + L.vals[0]), reinterpret_cast<const unsigned*>(ws + L.small_list), reinterpret_cast<const F3dgRec*>(ws + L.rec),
                           reinterpret_cast<const float2*>(ws + L.means2D), reinterpret_cast<const float4*>(ws + L.conic),
                           background, (flags & F3DG_FLAG_BG_PER_VIEW) ? 1 : 0, reinterpret_cast<const float*>(ws + L.final_T),
                           reinterpret_cast<const unsigned*>(ws + L.n_contrib), dL_dpix, dL_dmean2D, dL_dopacity, dL_dcolor,
                           acc);
#endif
    f3dg_prof_bwd_mark(prof, 0, s);
    F3DG_KLAUNCH(preprocess_bwd_kernel, dim3((P + F3DG_BLOCK - 1) / F3DG_BLOCK), dim3(F3DG_BLOCK), 0, s, P,
                       D, M, means3D, radii_used, shs, reinterpret_cast<const unsigned char*>(ws + L.clamped), scales,
                       rotations, viewmatrix, cam_pos, acc, dL_dview2gaussian, dL_dcolor, dL_dmean3D, dL_dsh, dL_dscale,
                       dL_drot, n_views, acc_stride, dL_dmean2D, dL_dopacity);
    f3dg_prof_bwd_mark(prof, 1, s);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}
