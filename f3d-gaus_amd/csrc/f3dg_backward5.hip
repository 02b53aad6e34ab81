// f3dg_backward5.hip -- compositing backward with DENSE batches of (pixel, entry) pairs (option bwd_dense; the counterpart of the
// split-pixel forward of f3dg_render5.hip).
//
// render3_bwd_kernel (f3dg_backward.hip) walks the entries that reach any pixel of a quadrant in lock-step: every step costs the wave the
// whole loop body of backward.cu:745-950 -- alpha again, the 17 partials, a transposed reduction over the 64 lanes -- for the pixels the
// entry contributes to: 24 of 64 at C5, 7.7 of 64 on pixel-aligned predicted sets. Here a window's contributing pairs are compacted
// ENTRY-major (a scalar loop over the entries: the ballot of the pixels an entry reaches, every such pixel lane writes its pair at ring
// position tail + mbcnt) and processed 64 to a batch, one pair per lane:
//   * the pair's pixel constants (ray, dL/dpixel, the pixel's totals) come by ds_bpermute from the lane that owns the pixel, its record from
//     the staged window; alpha is recomputed exactly as the forward did (f3dg_fast_t_G or the reference's float32 / float64 order);
//   * the per-pixel recurrence -- T rebuilt back to front (a correctly rounded reciprocal per pair, a multiply per layer), the colour /
//     normal blended behind the entry -- lives in LDS
//     (one float4 per pixel) and is advanced entry after entry: the lanes of one entry's run update the slots of their pixels together,
//     successive runs of the batch follow each other. The six recurrences accum_rec[c] / accum_normal[k] of backward.cu are ONE here:
//     they only ever meet dL/dpixel as a dot product and dL/dpixel is constant per pixel;
//   * the 17 partials are summed over the pixels of an entry by segmented scans along its run (f3dg_segscan.h) and added to memory by the
//     run's last lane.
// Same arithmetic per pair as render3_bwd_kernel except for the folded recurrence and the order of the per-entry sums (a tree over the
// run instead of a transposed butterfly over the wave): compositing-stage gradients within 1e-5 of the oracle as before
// (tests/test_raster_backward_gpu.py runs with either kernel).
#include "f3dg_common.h"
#include "f3dg_ellipse.h"
#include "f3dg_segscan.h"

int g_f3dg_bwd_dense = 1;             // option bwd_dense: 1 (default) = render5_bwd_kernel, 0 = render3_bwd_kernel (the lock-step walk)

namespace {

#ifndef F3DG_B5_STAGE
#define F3DG_B5_STAGE 6             // runs whose totals go through LDS together (12: 8.5 ms at C5, 6: 8.0, 3: 8.2 -- LDS decides the occupancy)
#endif
#ifndef F3DG_B5_OCC
#define F3DG_B5_OCC 5               // 96 VGPRs, 8 KB of LDS: five waves per SIMD
#endif

template <int OCC>
__global__ void __launch_bounds__(64, OCC)
render5_bwd_kernel(int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y,
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
    __shared__ float4 sC[64];             // 2D conic (x, y, z) and, over the opacity * coef the record carries as well, the projected centre's x
    __shared__ float sY[64];              // the projected centre's y
    __shared__ uint2 sQ[128];             // (list position, Gaussian id) of the kept entries, ring
    __shared__ unsigned short sK[128];    // (owning lane << 6) | window slot: the pairs of the batches, entry-major, ring
    __shared__ __attribute__((aligned(16))) float sOut[F3DG_B5_STAGE][20];     // the 17 totals + Gaussian id of up to twelve runs on their way to one-element-per-lane
    __shared__ float4 sS[64];             // per pixel: (T in front of its last blended entry, blended dot behind it, that entry's alpha, its dot)

    const bool alpha_fast = hdr->alpha_fast != 0;
    const size_t vP = (size_t)view * P;
    const F3dgRec* vrec = rec + vP;
    const float4* vcull = cull + vP;
    const float* fT = final_T + (size_t)view * 4 * HW;
    const unsigned* nc = n_contrib + (size_t)view * 2 * HW;
    const float* dpix = dL_dpixels + (size_t)view * F3DG_OUT_CHANNELS * HW;
    const float* bg = background + (bg_per_view ? 3 * view : 0);

    const float T_final = inside ? fT[pix_id] : 0;
    const float final_D = inside ? fT[pix_id + HW] : 0;
    const float final_A = 1 - T_final;
    const float dL_dreg = inside ? dpix[8 * HW + pix_id] : 0;

    const int last_contributor = inside ? (int)nc[pix_id] : 0;
    const int max_contributor = inside ? (int)nc[pix_id + HW] : 0;
    float dpx0 = 0, dpx1 = 0, dpx2 = 0, dn0 = 0, dn1 = 0, dn2 = 0, dL_dmax_depth = 0;
    if (inside) {
        dpx0 = dpix[pix_id]; dpx1 = dpix[HW + pix_id]; dpx2 = dpix[2 * HW + pix_id];
        dn0 = dpix[3 * HW + pix_id]; dn1 = dpix[4 * HW + pix_id]; dn2 = dpix[5 * HW + pix_id];
        dL_dmax_depth = dpix[6 * HW + pix_id];
    }
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

    // the pixel's running state of backward.cu:745-810, folded: every accumulated colour / normal only ever meets dL/dpixel as a dot product,
    // and dL/dpixel is constant per pixel, so the six recurrences accum_rec[c] / accum_normal[k] are ONE: A <- alpha' u' + (1 - alpha') A with
    // u = c . dL/dC + n . dL/dN of the entry blended last (alpha', u')
    sS[lane] = make_float4(T_final, 0.0f, 0.0f, 0.0f);
    const unsigned el_run = lane / 20u, el = lane - 20u * el_run;      // lane 20 r + c adds element c of the r-th run of a group of three
    const float TfBg = T_final * bg_dot_dpixel;
    // one 128-byte record per (view, Gaussian) takes all 17 sums of an entry: ten float64 (view2gaussian) at byte 0, seven float32 (colour,
    // mean2D, opacity) at byte 80 -- an atomic event touches ONE line (three 64-byte requests at most) instead of four arrays;
    // preprocess_bwd_kernel hands the float32 ones to the caller's arrays
    double* const gacc = dL_dv2g_acc + vP * 16;
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
        float2 m2 = make_float2(0.0f, 0.0f);
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
            m2 = means2D[vP + id];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane < m) {
            ec = sR[3][lane].w;
            sC[lane].w = m2.x;          // (conic.w = opacity * coef is record word 10: sR[2][j].z)
            sY[lane] = m2.y;
        }
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

        // ---- the window's contributing (pixel, entry) pairs, ENTRY-major (back to front: window slot order), 64 to a batch
        unsigned long long todo = any;
        unsigned qh = 0u, qt = 0u;                        // wave-uniform ring counters of this window
        const int slot_pos = (int)sQ[(qhead + lane) & 127u].x;       // lane j: list position of window slot j (read back with v_readlane: no LDS round trip per entry)
        do {
            while (todo != 0ull && qt - qh < 64u) {
                const int j = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const int contributor = __builtin_amdgcn_readlane(slot_pos, j);      // 0-based position from the front
                const bool mine = inside && ((pass >> j) & 1ull) != 0ull && contributor < last_contributor;
                const unsigned long long rowmask = __ballot(mine);
                if (rowmask == 0ull)
                    continue;
                const unsigned r = __builtin_amdgcn_mbcnt_hi((unsigned)(rowmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)rowmask, 0u));
                if (mine) sK[(qt + r) & 127u] = (unsigned short)((lane << 6) | (unsigned)j);
                qt += (unsigned)__popcll(rowmask);
            }
            const unsigned nb = qt - qh < 64u ? qt - qh : 64u;
            if (nb == 0u)
                break;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ---- one batch: lane q takes pair q of the ring
            const unsigned kk = (unsigned)sK[(qh + lane) & 127u];
            const bool valid = lane < nb;
            const unsigned k = valid ? (unsigned)kk : (unsigned)__builtin_amdgcn_readfirstlane((int)kk);      // (lanes beyond the batch repeat pair 0, inactive)
            const unsigned own = (k >> 6) & 63u;
            const int j = (int)(k & 63u);
            // the runs of the batch: consecutive pairs of one entry (a window slot appears in one run only). A lane's distance to the first
            // lane of its run is what the segmented scans below need
            const int jprev = __builtin_amdgcn_update_dpp(-1, j, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            const unsigned long long heads_all = __ballot(valid && j != jprev);
            const unsigned long long upto = heads_all & (~0ull >> (63u - lane));
            const unsigned rr = valid ? lane - (63u - (unsigned)__builtin_clzll(upto | 1ull)) : 0u;
            const int oaddr = (int)(own << 2);
#define F3DG_B5_PULL(v) __int_as_float(__builtin_amdgcn_ds_bpermute(oaddr, __float_as_int(v)))
            const float p_ray_x = F3DG_B5_PULL(ray_x), p_ray_y = F3DG_B5_PULL(ray_y);
            const float p_dpx0 = F3DG_B5_PULL(dpx0), p_dpx1 = F3DG_B5_PULL(dpx1), p_dpx2 = F3DG_B5_PULL(dpx2);
            const float p_dn0 = F3DG_B5_PULL(dn0), p_dn1 = F3DG_B5_PULL(dn1), p_dn2 = F3DG_B5_PULL(dn2);
            const float p_dmaxd = F3DG_B5_PULL(dL_dmax_depth), p_dreg = F3DG_B5_PULL(dL_dreg);
            const float p_final_A = F3DG_B5_PULL(final_A), p_final_D = F3DG_B5_PULL(final_D), p_TfBg = F3DG_B5_PULL(TfBg);
            const int p_maxc = __builtin_amdgcn_ds_bpermute(oaddr, max_contributor);
            const float p_pixx = (float)(qx0 + (own & 7u)), p_pixy = (float)(qy0 + (own >> 3));      // the owner's pixel
#undef F3DG_B5_PULL
            const uint2 pe = sQ[(qhead + (unsigned)j) & 127u];
            const int contributor = (int)pe.x;

            const float4 q0 = sR[0][j], q1 = sR[1][j], q2 = sR[2][j];
            const float n0 = q0.x * p_ray_x + q0.y * p_ray_y + q0.z;
            const float n1 = q0.y * p_ray_x + q0.w * p_ray_y + q1.x;
            const float n2 = q0.z * p_ray_x + q1.x * p_ray_y + q1.y;
            const float aaf = p_ray_x * n0 + p_ray_y * n1 + n2;
            const float bhalf = q1.z * p_ray_x + q1.w * p_ray_y + q2.x;
            const float CC = q2.y;
            float t = 1.0f, G = 0, alpha = 0;
            bool active = valid;
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
            n_pairs += (unsigned)__popcll(__ballot(active));

            // ---- gradient terms only from here: float32 with one reciprocal each, FMA contraction allowed (as render3_bwd_kernel)
            float g[18];
#pragma unroll
            for (int c = 0; c < 18; c++) g[c] = 0.0f;
            {
#pragma clang fp contract(fast)
                const float4 q3 = sR[3][j];
                float4 con = sC[j];
                const float2 xy = make_float2(con.w, sY[j]);
                con.w = q2.z;
                const float inv_len = __builtin_amdgcn_rsqf(n0 * n0 + n1 * n1 + n2 * n2 + 1e-7f);
                const float nn0 = -n0 * inv_len, nn1 = -n1 * inv_len, nn2 = -n2 * inv_len;
                const float c0 = q3.x, c1 = q3.y, c2 = q3.z;
                const float u = c0 * p_dpx0 + c1 * p_dpx1 + c2 * p_dpx2 + nn0 * p_dn0 + nn1 * p_dn1 + nn2 * p_dn2;
                const float oma = 1.f - alpha;

                // ---- the recurrence, entry after entry (a pixel has at most one pair per entry, so the lanes of a run never share a
                // state slot; successive runs of the batch do)
                float Tr = 0.0f, A = 0.0f;
                // T is rebuilt back to front as the reference does (backward.cu:803, T = T / (1 - alpha)), with the division taken off the
                // serial chain: one correctly rounded reciprocal per pair up front, a multiply per run (<= 1 ulp from the quotient per layer;
                // C5 8.1 -> 7.8 ms; the gradient tests, incl. the ill-conditioned B2, hold their bars)
                const float inv_oma_ieee = 1.0f / oma;
                unsigned long long heads = heads_all;
                while (heads != 0ull) {
                    const unsigned a0 = (unsigned)__builtin_ctzll(heads);
                    heads &= heads - 1ull;
                    const unsigned b0 = heads != 0ull ? (unsigned)__builtin_ctzll(heads) : nb;
                    if (active && lane >= a0 && lane < b0) {
                        const float4 st = sS[own];
                        Tr = st.x * inv_oma_ieee;
                        A = st.z * st.w + (1.f - st.z) * st.y;
                        sS[own] = make_float4(Tr, A, alpha, u);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }

                if (active) {
                    const float d_x = xy.x - p_pixx, d_y = xy.y - p_pixy;
                    const float inv_t = __builtin_amdgcn_rcpf(t);
                    const float mapped_max_t = fmaf(-0.20040080160320642f, inv_t, 1.0020040080160322f);
                    const float dmax_t_dd = 0.20040080160320642f * inv_t * inv_t;
                    const float inv_oma = __builtin_amdgcn_rcpf(oma);      // background term only
                    const float dchannel_dcolor = alpha * Tr;
                    g[0] = dchannel_dcolor * p_dpx0;
                    g[1] = dchannel_dcolor * p_dpx1;
                    g[2] = dchannel_dcolor * p_dpx2;
                    const float dL_dmax_t = 2.0f * (Tr * alpha) * (mapped_max_t * p_final_A - p_final_D) * p_dreg * dmax_t_dd;
                    const float dnn0 = alpha * Tr * p_dn0, dnn1 = alpha * Tr * p_dn1, dnn2 = alpha * Tr * p_dn2;
                    float dL_dlength = (dnn0 * n0 + dnn1 * n1 + dnn2 * n2);
                    dL_dlength *= inv_len * inv_len;
                    float dLn0 = (-dnn0 + dL_dlength * n0) * inv_len;
                    float dLn1 = (-dnn1 + dL_dlength * n1) * inv_len;
                    float dLn2 = (-dnn2 + dL_dlength * n2) * inv_len;
                    float dL_dt = dL_dmax_t;
                    if (contributor == p_maxc - 1)
                        dL_dt += p_dmaxd;
                    float dL_dalpha = (u - A) * Tr;
                    dL_dalpha += (-inv_oma) * p_TfBg;

                    const float dL_dG = con.w * dL_dalpha;
                    const float gdx = G * d_x;
                    const float gdy = G * d_y;
                    const float dG_ddelx = -gdx * con.x - gdy * con.y;
                    const float dG_ddely = -gdy * con.z - gdx * con.y;
                    g[3] = dL_dG * dG_ddelx * ddelx_dx;
                    g[4] = dL_dG * dG_ddely * ddely_dy;
                    g[5] = fabsf(dL_dG * dG_ddelx * ddelx_dx) + fabsf(dL_dG * dG_ddely * ddely_dy);
                    g[6] = G * dL_dalpha;

                    const float dL_dpower = dL_dG * G;
                    const float dL_dmin_value = dL_dpower * -0.5f;
                    const float qf = -2.0f * t, inv_a = __builtin_amdgcn_rcpf(aaf);
                    float dL_dA = dL_dmin_value * qf * qf * 0.25f;
                    float dL_dB = dL_dmin_value * (-0.5f * qf);
                    const float dL_dC = dL_dmin_value;
                    dL_dA += dL_dt * (0.5f * qf * inv_a);
                    dL_dB += dL_dt * (-0.5f * inv_a);
                    dLn0 += dL_dA * p_ray_x;
                    dLn1 += dL_dA * p_ray_y;
                    dLn2 += dL_dA;

                    g[7] = dLn0 * p_ray_x;
                    g[8] = dLn0 * p_ray_y + dLn1 * p_ray_x;
                    g[9] = dLn0 + dLn2 * p_ray_x;
                    g[10] = dLn1 * p_ray_y;
                    g[11] = dLn1 + dLn2 * p_ray_y;
                    g[12] = dLn2;
                    g[13] = dL_dB * 2 * p_ray_x;
                    g[14] = dL_dB * 2 * p_ray_y;
                    g[15] = dL_dB * 2;
                    g[16] = dL_dC;
                }

                // ---- the 17 partials of an entry summed over its pixels: segmented scans along the entry's run, totals in its last lane
                const SegFlags sf = seg_flags(rr, lane);
                {
                    float v8[8] = { g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7] };
                    seg_sums(v8, sf);
#pragma unroll
                    for (int c = 0; c < 8; c++) g[c] = v8[c];
                }
                {
                    float v8[8] = { g[8], g[9], g[10], g[11], g[12], g[13], g[14], g[15] };
                    seg_sums(v8, sf);
#pragma unroll
                    for (int c = 0; c < 8; c++) g[8 + c] = v8[c];
                }
                {
                    float v2[2] = { g[16], g[17] };
                    seg_sums(v2, sf);
                    g[16] = v2[0];
                }
                const bool last = valid && (lane + 1u == nb || ((heads_all >> ((lane + 1u) & 63u)) & 1ull) != 0ull);
                // ---- to memory. One lane adding its run's 17 totals is 17 atomic instructions with one address each (measured: three times
                // the write transactions of render3_bwd_kernel, whose four row-end lanes add neighbouring elements in one instruction, and
                // the kernel waits for the atomic units: 28 ms at C5). So the totals of up to twelve runs at a time go through LDS and come back, three runs per instruction, one
                // ELEMENT per lane -- lane 20 r + c holds element c of run r: colour 0..2, mean2D 3..5, opacity 6, view2gaussian 7..16 (the record's float32 and float64 halves) -- and
                // one float32 and one float64 atomic instruction add runs of neighbouring addresses.
                const unsigned nruns = (unsigned)__popcll(heads_all);
                const unsigned ord = __builtin_amdgcn_mbcnt_hi((unsigned)(heads_all >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)heads_all, 0u)) +
                                     ((valid && j != jprev) ? 1u : 0u) - 1u;       // which run of the batch this lane belongs to
                for (unsigned base = 0u; base < nruns; base += F3DG_B5_STAGE) {
                    if (last && ord - base < (unsigned)F3DG_B5_STAGE) {
                        float* o = sOut[ord - base];
                        *reinterpret_cast<float4*>(o) = make_float4(g[0], g[1], g[2], g[3]);
                        *reinterpret_cast<float4*>(o + 4) = make_float4(g[4], g[5], g[6], g[7]);
                        *reinterpret_cast<float4*>(o + 8) = make_float4(g[8], g[9], g[10], g[11]);
                        *reinterpret_cast<float4*>(o + 12) = make_float4(g[12], g[13], g[14], g[15]);
                        *reinterpret_cast<float2*>(o + 16) = make_float2(g[16], __int_as_float((int)pe.y));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const unsigned staged = nruns - base < (unsigned)F3DG_B5_STAGE ? nruns - base : (unsigned)F3DG_B5_STAGE;
                    for (unsigned r3 = 0u; r3 < staged; r3 += 3u) {       // three runs per pair of atomic instructions, no waiting in between
                        const unsigned rsel = r3 + el_run;
                        if (el < 17u && el_run < 3u && rsel < staged && !debug_no_atomics) {
                            const float v = sOut[rsel][el];
                            const size_t id = (size_t)(unsigned)__float_as_int(sOut[rsel][17]);
                            double* rec = gacc + id * 16;
                            if (el < 7u)
                                unsafeAtomicAdd(reinterpret_cast<float*>(rec + 10) + el, v);
                            else
                                unsafeAtomicAdd(rec + (el - 7u), (double)v);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the next round overwrites the slots
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
            qh += nb;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the next compaction overwrites ring slots this batch has read
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } while (todo != 0ull || qt != qh);
        qhead += m;
        qcount -= m;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the window's slots are rewritten by the next one
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (lane == 0 && n_pairs)
        atomicAdd(&hdr->bwd_pairs, (unsigned long long)n_pairs);
}


} // namespace

int f3dg_launch_render5_bwd(hipStream_t s, int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y, F3dgHeader* hdr,
                            const uint2* ranges, const unsigned* point_list, const unsigned* small_list, const F3dgRec* rec, const float4* cull,
                            const float2* means2D, const float4* conic, const float* background, int bg_per_view, const float* final_T,
                            const unsigned* n_contrib, const float* dL_dpixels, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolors,
                            double* dL_dv2g_acc, int debug_no_atomics)
{
    F3DG_KLAUNCH((render5_bwd_kernel<F3DG_B5_OCC>), dim3((unsigned)V * (unsigned)T * 4u), dim3(64), 0, s, V, P, W, H, tiles_x, T, focal_x, focal_y, hdr, ranges,
                 point_list, small_list, rec, cull, means2D, conic, background, bg_per_view, final_T, n_contrib, dL_dpixels, dL_dmean2D,
                 dL_dopacity, dL_dcolors, dL_dv2g_acc, debug_no_atomics);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}
