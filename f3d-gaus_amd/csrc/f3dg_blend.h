// f3dg_blend.h -- the per-(pixel, Gaussian) arithmetic of the compositing forward, split where the reference's recurrence allows it.
//
// renderCUDA's loop body (reference RAST/cuda_rasterizer/forward.cu:493-583) has two parts:
//   * a STATELESS part -- the ray's normal in the Gaussian's frame, a = r' Sigma' r, b, the minimum of the quadric along the ray, the
//     exponent, alpha, the intersection depth t, its NDC image, the unit normal (forward.cu:499-540, 559-561): a function of the pixel's
//     ray and the Gaussian's record only;
//   * the RECURRENCE -- test_T = T (1 - alpha), the saturation stop, the distortion sums, the colour / normal / alpha accumulators, the
//     median-depth switch (forward.cu:543-583): a function of the pixel's running state and six numbers of the stateless part.
// f3dg_pair_eval is the first, f3dg_pair_apply the second; applying the one to the other's result reproduces blend_entry /
// blend_entry_fast of f3dg_render.hip operation for operation (the values handed over are float32 in the reference too). The split lets
// the packed schedule of f3dg_render4.hip evaluate the stateless part for (pixel, Gaussian) pairs of DIFFERENT pixels in one wave trip,
// one pair per lane, and leave only the recurrence to the lane that owns the pixel.
//
// FAST = false: the reference's float32 / float64 operation order (the file is built with -ffp-contract=off).
// FAST = true:  the inference arithmetic of f3dg_common.h (f3dg_fast_t_G: error-free float32 pairs for the float64 island, hardware
//               exp / rcp / rsq, contracted accumulations).
#pragma once
#include "f3dg_common.h"

struct F3dgPixel {
    float Tr;
    unsigned last_contributor, max_contributor;
    float C0, C1, C2, C3, C4, C5, C6, C7;
    float dist1, dist2, distortion;
};

__device__ __forceinline__ void f3dg_pixel_init(F3dgPixel& st)
{
    st.Tr = 1.0f;
    st.last_contributor = 0; st.max_contributor = (unsigned)-1;
    st.C0 = st.C1 = st.C2 = st.C3 = st.C4 = st.C5 = st.C6 = st.C7 = 0;
    st.dist1 = st.dist2 = st.distortion = 0;
}

// What the recurrence needs of a pair. alpha == 0 marks a pair the reference leaves by a bare `continue` (t <= 0.2 or alpha < 1/255):
// a blended pair always has alpha >= 1/255.
struct F3dgPair {
    float alpha, t, m, nn0, nn1, nn2;     // m: the NDC image of t (mapped_max_t); nn: the unit normal, already negated
};

// q0 = (v0 v1 v2 v3), q1 = (v4 v5 v6 v7), q2 = (v8 v9 opacity K): the first three 16-byte chunks of the Gaussian's record
template <bool FAST, bool NORMAL, bool DIST, bool SANITIZE = false>
__device__ __forceinline__ F3dgPair f3dg_pair_eval(float ray_x, float ray_y, const float4& q0, const float4& q1, const float4& q2)
{
    F3dgPair pr;
    pr.alpha = 0.0f; pr.t = 0.0f; pr.m = 0.0f; pr.nn0 = pr.nn1 = pr.nn2 = 0.0f;
    // the reference's own float32 a and b / 2, in its operation order (forward.cu:499-509)
    const float n0 = q0.x * ray_x + q0.y * ray_y + q0.z;
    const float n1 = q0.y * ray_x + q0.w * ray_y + q1.x;
    const float n2 = q0.z * ray_x + q1.x * ray_y + q1.y;
    const float aaf = ray_x * n0 + ray_y * n1 + n2;
    const float bhalf = q1.z * ray_x + q1.w * ray_y + q2.x;
    const float CC = q2.y, opac = q2.z;
    if (FAST) {
        float t, G;
        f3dg_fast_t_G(aaf, bhalf, CC, t, G);
        // (double)t <= 0.2  <=>  t < 0.2f: 0.2f is the float just above 0.2 (false for NaN, as the reference's test)
        const bool behind = t < 0.2f;
        const float alpha = fminf(0.99f, opac * G);
        pr.alpha = (behind || alpha < 1.0f / 255.0f) ? 0.0f : alpha;
        // a rejected pair gets a harmless depth: t = -b / a may be infinite or NaN there, and the branch-free recurrence
        // (f3dg_pair_apply_flat) multiplies every handed-over number by a zero weight instead of skipping it
        pr.t = SANITIZE ? (pr.alpha != 0.0f ? t : 1.0f) : t;
        // (FAR*t - FAR*NEAR) / ((FAR - NEAR)*t) = FAR/(FAR-NEAR) - (FAR*NEAR/(FAR-NEAR)) / t
        if (DIST) pr.m = fmaf(-0.20040080160320642f, __builtin_amdgcn_rcpf(pr.t), 1.0020040080160322f);
        if (NORMAL) {
            const float ninv = -__builtin_amdgcn_rsqf(fmaf(n2, n2, fmaf(n1, n1, n0 * n0)) + 1e-7f);
            pr.nn0 = n0 * ninv; pr.nn1 = n1 * ninv; pr.nn2 = n2 * ninv;
            // (the branch-free recurrences give a rejected pair weight 0 instead of skipping it: its normal must be finite whatever the record holds)
            if (SANITIZE && pr.alpha == 0.0f) { pr.nn0 = 0.0f; pr.nn1 = 0.0f; pr.nn2 = 0.0f; }
        }
    } else {
        const double AA = aaf;
        const float bbf = 2 * bhalf;
        const double BB = bbf;
        // ONE float64 division serves both uses: -BB / (2 * AA) is the correctly rounded quotient BB / AA scaled by -1/2 (exact)
        const double q = BB / AA;
        const float t = (float)(-0.5 * q);
        if (!(t <= F3DG_NEAR_PLANE)) {
            const double min_value = -q * (BB / 4.) + CC;
            float power = (float)(-0.5f * min_value);
            if (power > 0.0f)
                power = 0.0f;
            const float alpha = fminf(0.99f, opac * expf(power));
            if (!(alpha < 1.0f / 255.0f)) {
                pr.alpha = alpha;
                pr.t = t;
                if (DIST) pr.m = (float)((F3DG_FAR_PLANE * t - F3DG_FAR_PLANE * F3DG_NEAR_PLANE) / ((F3DG_FAR_PLANE - F3DG_NEAR_PLANE) * t));
                if (NORMAL) {
                    const float length = (float)sqrt(n0 * n0 + n1 * n1 + n2 * n2 + 1e-7);
                    pr.nn0 = -n0 / length; pr.nn1 = -n1 / length; pr.nn2 = -n2 / length;
                }
            }
        }
    }
    return pr;
}

// The recurrence for a pair with pr.alpha != 0. Returns true when the pixel saturates (`done = true`: the pair is NOT blended).
// `contributor` is whatever the caller wants recorded in last_contributor / max_contributor (a 1-based list position or a slot).
template <bool FAST, bool NORMAL, bool DIST>
__device__ __forceinline__ bool f3dg_pair_apply(F3dgPixel& st, unsigned contributor, const F3dgPair& pr, float cr, float cg, float cb)
{
    const float alpha = pr.alpha;
    const float Tr = st.Tr;
    const float test_T = Tr * (1 - alpha);
    if (test_T < 0.0001f)
        return true;
    if (FAST) {
        // (the accumulations are contracted into FMAs: fewer roundings than the reference's separate products and sums)
        const float w = alpha * Tr;
        if (DIST) {
            const float m = pr.m;
            const float A = 1 - Tr;
            const float m2 = m * m;
            const float error = fmaf(-2.0f * m, st.dist1, fmaf(m2, A, st.dist2));
            st.distortion = fmaf(error, w, st.distortion);
            st.dist1 = fmaf(m, w, st.dist1);
            st.dist2 = fmaf(m2, w, st.dist2);
        }
        st.C0 = fmaf(cr, w, st.C0);
        st.C1 = fmaf(cg, w, st.C1);
        st.C2 = fmaf(cb, w, st.C2);
        if (NORMAL) {
            st.C3 = fmaf(pr.nn0, w, st.C3);
            st.C4 = fmaf(pr.nn1, w, st.C4);
            st.C5 = fmaf(pr.nn2, w, st.C5);
        }
        if (Tr > 0.5f) {
            st.C6 = pr.t;
            st.max_contributor = contributor;
        }
        st.C7 += w;
    } else {
        if (DIST) {
            const float mapped_max_t = pr.m;
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
            st.C3 += pr.nn0 * alpha * Tr;
            st.C4 += pr.nn1 * alpha * Tr;
            st.C5 += pr.nn2 * alpha * Tr;
        }
        if (Tr > 0.5) {
            st.C6 = pr.t;
            st.max_contributor = contributor;
        }
        st.C7 += alpha * Tr;
    }
    st.Tr = test_T;
    st.last_contributor = contributor;
    return false;
}

// Branch-free form of the fast recurrence for the blend trips of the packed schedule (FAST only): a pair that is not blended -- alpha
// == 0, or the pair that saturates the pixel -- runs the same instructions with weight 0 (its six numbers must then be finite: the
// caller zeroes them for alpha == 0). Every accumulator receives fma(x, 0, acc) = acc, so the results are those of f3dg_pair_apply to
// the bit; what goes is the exec-mask bookkeeping of two nested branches (~20 scalar instructions per trip of a divergent loop).
template <bool NORMAL, bool DIST>
__device__ __forceinline__ bool f3dg_pair_apply_flat(F3dgPixel& st, unsigned contributor, const F3dgPair& pr, float cr, float cg, float cb)
{
    const float alpha = pr.alpha;
    const float Tr = st.Tr;
    const float test_T = Tr * (1 - alpha);
    const bool live = alpha != 0.0f;
    const bool sat = live && test_T < 0.0001f;
    const bool go = live && !sat;
    const float w = go ? alpha * Tr : 0.0f;
    if (DIST) {
        const float m = pr.m;
        const float A = 1 - Tr;
        const float m2 = m * m;
        const float error = fmaf(-2.0f * m, st.dist1, fmaf(m2, A, st.dist2));
        st.distortion = fmaf(error, w, st.distortion);
        st.dist1 = fmaf(m, w, st.dist1);
        st.dist2 = fmaf(m2, w, st.dist2);
    }
    st.C0 = fmaf(cr, w, st.C0);
    st.C1 = fmaf(cg, w, st.C1);
    st.C2 = fmaf(cb, w, st.C2);
    if (NORMAL) {
        st.C3 = fmaf(pr.nn0, w, st.C3);
        st.C4 = fmaf(pr.nn1, w, st.C4);
        st.C5 = fmaf(pr.nn2, w, st.C5);
    }
    const bool front = go && Tr > 0.5f;
    st.C6 = front ? pr.t : st.C6;
    st.max_contributor = front ? contributor : st.max_contributor;
    st.C7 += w;
    st.Tr = go ? test_T : Tr;
    st.last_contributor = go ? contributor : st.last_contributor;
    return sat;
}
