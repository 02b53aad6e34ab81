// f3dg_preprocess.hip -- per-(view, Gaussian) projection stage.
//
// Replaces preprocessCUDA<3> and its helpers (reference RAST/cuda_rasterizer/forward.cu:283-404 with
// computeCov3D :129-163, computeCov2D :74-124, computeView2Gaussian :168-279, computeColorFromSH :20-71,
// and auxiliary.h in_frustum :177-202, getRect :64-74, ndc2Pix :59-62).
//
// MI355X shape: one launch covers all views of the call (blockIdx.y = view, so the view matrices are
// wave-uniform scalar loads), every thread owns one Gaussian, and everything the compositing kernel needs
// about a Gaussian is packed into ONE 64-byte record so the later gather touches a single half cache line.
// The arithmetic keeps the reference's float/double operation order (file is built with -ffp-contract=off).
#include "f3dg_common.h"

extern int g_f3dg_debug_skip_all;
extern int g_f3dg_pre_order;

namespace {

__device__ __constant__ float SH_C0 = 0.28209479177387814f;
__device__ __constant__ float SH_C1 = 0.4886025119029199f;
__device__ __constant__ float SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                           -1.0925484305920792f, 0.5462742152960396f };
__device__ __constant__ float SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                           0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                           -0.5900435899266435f };

// 3x3 column-major matrix, m[col][row]; products follow glm 0.9.9.9's evaluation order:
// R[c][r] = (A[0][r]*B[c][0] + A[1][r]*B[c][1]) + A[2][r]*B[c][2]
struct M3 { float m[3][3]; };

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

// Rotation from the (r,x,y,z) quaternion, NOT normalised (forward.cu:138-149); arguments are columns.
__device__ __forceinline__ M3 quat_to_R(float4 q)
{
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    M3 R;
    R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[0][1] = 2.f * (x * y - r * z);       R.m[0][2] = 2.f * (x * z + r * y);
    R.m[1][0] = 2.f * (x * y + r * z);       R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[1][2] = 2.f * (y * z - r * x);
    R.m[2][0] = 2.f * (x * z - r * y);       R.m[2][1] = 2.f * (y * z + r * x);       R.m[2][2] = 1.f - 2.f * (x * x + y * y);
    return R;
}

__device__ __forceinline__ float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// Pre-test constant K of the compositing kernel. alpha = min(.99, opac*exp(power)) < 1/255 whenever
// power = -(C - b^2/a)/2 < thr, i.e. b^2 < K0 * a with K0 = C + 2 thr (a, b, C: the reference's own float32 values).
// The kernel evaluates fl(b*b) < fl(K*a) in float32; K = K0 (1 - 5e-7) absorbs both product roundings and the
// narrowing of K itself. K is clamped at 0 ("never skip") so that a <= 0 or K0 <= 0 cannot produce a false skip.
__device__ __forceinline__ float pretest_constant(const float* vg, float thr)
{
    if (thr == __builtin_inff()) return __builtin_inff();      // opacity <= 0: alpha <= 0 always
    const double K0 = ((double)vg[9] + 2.0 * (double)thr) * (1.0 - 5e-7);
    return K0 > 0.0 ? (float)K0 : 0.0f;                         // NaN -> 0
}

// Conservative pixel-space box of the region where this Gaussian's alpha can reach 1/255 in the compositing kernel.
//   alpha >= 1/255  =>  p^ = -(C - b^^2/a^)/2 >= thr  =>  (C - k) a^ - b^^2 <= 0 with k = -2 thr (a^ > 0), where a^, b^ are the
// reference's own float32 values of a = r^T Sigma' r and b = B^T r, r = (x, y, 1). Their rounding errors matter (they are
// amplified by C ~ 1e5..1e6), so the region is widened by a worst-case bound on them, standard forward error analysis of the
// reference's operation order (forward.cu:504-509; u = 2^-24):
//   |a^ - a| <= 6u A(r),  A(r) = |r|^T |Sigma'| |r|   (two nested 3-term dot products: gamma_3 twice),
//   |b^ - b| <= 3u Bn(r), Bn(r) = |B|^T |r|,
//   (C - k) a - b^2  <=  [(C - k) a^ - b^^2] + (C - k) |a^ - a| + |b^ - b| (2 |b| + |b^ - b|)  <=  6u ((C - k) A + Bn^2) =: D.
// |x| <= tan_fovx and |y| <= tan_fovy for every pixel centre, so A, Bn are bounded once per Gaussian with r = (tan_fovx,
// tan_fovy, 1), and the widened region is the exact conic r^T M r <= 0, M = (C - k) Sigma' - B B^T - diag(0, 0, 1.1 D); 2e-3 is
// added to k for the float32 rounding of the exponent itself. (Round 1 widened k by 8u C tr(Sigma')/S_min instead, which is
// valid but 3x too wide in area on the C2 workload.) Its axis-aligned extent follows from the dual conic adj(M). Everything is
// done in float64, and the box is inflated by 0.1 % + 0.25 px. Anything degenerate (camera inside the level set, non-finite,
// ill-conditioned, precomputed view2gaussian without scales) returns the "everything" box, i.e. no culling.
struct CullConic { double m00, m01, m02, m11, m12, m22; bool ok; };
__device__ __forceinline__ CullConic cull_conic(const float* vg, float thr, float tan_fovx, float tan_fovy)
{
#pragma clang fp contract(fast)      // a bound of this build's own, not a reference value: FMAs are welcome
    CullConic q;
    q.ok = false;
    const double u = 5.9604644775390625e-08;
    const double C = vg[9];
    const double k = -2.0 * (double)thr + 2e-3;
    const double cK = C - k;
    q.m00 = q.m01 = q.m02 = q.m11 = q.m12 = q.m22 = 0.0;
    if (!(cK > 0.0)) return q;
    const double X = tan_fovx, Y = tan_fovy;
    const double A = fabs((double)vg[0]) * X * X + 2.0 * fabs((double)vg[1]) * X * Y + 2.0 * fabs((double)vg[2]) * X +
                     fabs((double)vg[3]) * Y * Y + 2.0 * fabs((double)vg[4]) * Y + fabs((double)vg[5]);
    const double Bn = fabs((double)vg[6]) * X + fabs((double)vg[7]) * Y + fabs((double)vg[8]);
    // (7u on the b term: integrate's loop divides BB / AA in float32, forward.cu:917, one more rounding of b^2 / a than the 6u of the
    // compositing loop's float64 division)
    const double D = 1.1 * u * (6.0 * cK * A + 7.0 * Bn * Bn);
    const double B0 = vg[6], B1 = vg[7], B2 = vg[8];
    q.m00 = cK * vg[0] - B0 * B0; q.m01 = cK * vg[1] - B0 * B1; q.m02 = cK * vg[2] - B0 * B2;
    q.m11 = cK * vg[3] - B1 * B1; q.m12 = cK * vg[4] - B1 * B2; q.m22 = cK * vg[5] - B2 * B2 - D;
    q.ok = true;
    return q;
}
__device__ __forceinline__ float4 conservative_box(const CullConic& q, float thr, bool have_scale, int W, int H,
                                                   float focal_x, float focal_y)
{
#pragma clang fp contract(fast)      // a bound of this build's own, not a reference value: FMAs are welcome
    const float4 all = make_float4(-3.0e38f, 3.0e38f, -3.0e38f, 3.0e38f);
    if (!have_scale || !(thr < 3.0e38f)) return all;        // thr = +inf (alpha always < 1/255) is handled by the pre-test
    if (!q.ok) return all;
    const double m00 = q.m00, m01 = q.m01, m02 = q.m02, m11 = q.m11, m12 = q.m12, m22 = q.m22;
    const double D00 = m11 * m22 - m12 * m12, D11 = m00 * m22 - m02 * m02, D22 = m00 * m11 - m01 * m01;
    const double D02 = m01 * m12 - m02 * m11, D12 = m01 * m02 - m00 * m12;
    if (!(m00 > 0.0) || !(D22 > 1e-9 * fabs(m00 * m11))) return all;
    const double dx = D02 * D02 - D00 * D22, dy = D12 * D12 - D11 * D22;
    if (!(dx >= 0.0) || !(dy >= 0.0)) return all;
    const double cx = D02 / D22, cy = D12 / D22;
    const double hx = sqrt(dx) / D22 * 1.001 + 0.25 / focal_x, hy = sqrt(dy) / D22 * 1.001 + 0.25 / focal_y;
    const double x0 = (cx - hx) * focal_x + W / 2. - 0.5, x1 = (cx + hx) * focal_x + W / 2. - 0.5;
    const double y0 = (cy - hy) * focal_y + H / 2. - 0.5, y1 = (cy + hy) * focal_y + H / 2. - 0.5;
    if (!(x0 == x0) || !(x1 == x1) || !(y0 == y0) || !(y1 == y1)) return all;
    // round outwards when narrowing to float
    return make_float4((float)x0 - 1e-3f * (1.0f + fabsf((float)x0)), (float)x1 + 1e-3f * (1.0f + fabsf((float)x1)),
                       (float)y0 - 1e-3f * (1.0f + fabsf((float)y0)), (float)y1 + 1e-3f * (1.0f + fabsf((float)y1)));
}


// Conservative ELLIPSE of the same region (the compositing kernel's phase 1 tests pixels against it; f3dg_render.hip).
// The level set of cull_conic is an exact conic r^T M r <= 0 in ray space; with the centre (cx, cy) of that conic and Qc < 0
// its value there it reads  E(dx, dy) = a dx^2 + b dx dy + c dy^2 <= 1  in pixel offsets from the centre. The record is that
// ellipse scaled UNIFORMLY about its centre by s = 1.001 + 0.05 px / semi-minor axis (a point inside an ellipse stays inside
// any uniformly larger one), which covers the float32 rounding of the centre (<= 2.5e-4 px for images up to 4096 px) and of
// the kernel's float32 evaluation of E (<= ~2e-4 relative for aspect ratios up to 31). Flatter ellipses are fattened to that
// aspect ratio along their minor axis (the larger eigenvalue of the form is lowered: a superset). Output: e = (cx, cy, a, b)
// in pixel-index coordinates and c; the staging threads derive the ellipse's axis-aligned box from (a, b, c).
//   "everything" (nothing can be proven): a = b = c = 0 (E = 0 passes everywhere, no box);
//   "never" (alpha < 1/255 everywhere: opacity <= 0, or the level set is empty): a tiny ellipse far outside any image.
__device__ __forceinline__ void conservative_ellipse(const CullConic& q, float thr, bool have_scale, int W, int H,
                                                     float focal_x, float focal_y, double ifx, double ify, float4& e, float& ec)
{
#pragma clang fp contract(fast)      // a bound of this build's own, not a reference value: FMAs are welcome
    // (the kernel is VALU-bound and a float64 division costs ~15 float64 instructions: reciprocals are formed once and multiplied;
    // every rounding this introduces is far inside the 0.1 % inflation below, and the two square roots, which only feed that
    // inflation and the aspect limit, are float32 and rounded UP)
    e = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    ec = 0.0f;
    const float4 never = make_float4(-1.0e9f, -1.0e9f, 1.0e30f, 0.0f);
    if (thr == __builtin_inff()) { e = never; ec = 1.0e30f; return; }       // opacity <= 0: alpha <= 0 always
    if (!have_scale || !(thr < 3.0e38f) || !q.ok) return;
    const double m00 = q.m00, m01 = q.m01, m02 = q.m02, m11 = q.m11, m12 = q.m12, m22 = q.m22;
    const double D22 = m00 * m11 - m01 * m01;
    if (!(m00 > 0.0) || !(m11 > 0.0) || !(D22 > 1e-9 * fabs(m00 * m11))) return;
    // (hardware reciprocals instead of IEEE float64 divisions, ~25 instructions each and three per (view, Gaussian): these are this
    // build's own bounds, inflated by 0.1 % below. v_rcp_f64 is good to ~1e-8; the centre feeds Qc, where terms of ~1e10 cancel to
    // ~1e5, so ITS reciprocal takes one Newton step (-> ~1e-16); the other two only scale the form as a whole)
    double inv_D22 = __builtin_amdgcn_rcp(D22);
    inv_D22 = fma(fma(-D22, inv_D22, 1.0), inv_D22, inv_D22);
    const double cx = (m01 * m12 - m02 * m11) * inv_D22, cy = (m01 * m02 - m00 * m12) * inv_D22;
    const double Qc = m22 + m02 * cx + m12 * cy;            // value of the conic at its centre; the quadratic part is positive definite
    if (!(Qc == Qc) || !(cx == cx) || !(cy == cy)) return;
    if (Qc > 0.0 && Qc < 1.0e300) { e = never; ec = 1.0e30f; return; }      // empty level set: never visible
    if (!(Qc < 0.0)) return;
    const double k = -__builtin_amdgcn_rcp(Qc);
    double a = m00 * k * (ifx * ifx), b = 2.0 * m01 * k * (ifx * ify), c = m11 * k * (ify * ify);
    const double det = a * c - 0.25 * b * b, tr = a + c;
    if (!(det > 0.0) || !(tr < 1.0e300)) return;
    // larger eigenvalue, rounded up (it sets the inflation and the aspect limit: a larger value only enlarges the ellipse)
    const double disc = (double)(sqrtf((float)fmax(0.25 * tr * tr - det, 0.0)) * 1.000001f);
    double lmax = 0.5 * tr + disc;
    if (!(lmax > 0.0) || !(lmax < 1.0e300)) return;
    if (lmax * lmax > 1000.0 * det) {                        // lmax > 1000 lmin with lmin = det / lmax
        // [[a, b/2], [b/2, c]] - (lmax - 1000 lmin) v v^T, v = unit eigenvector of lmax
        const double lmin = det / lmax;
        if (!(lmin > 0.0)) return;
        double vx = 0.5 * b, vy = lmax - a;
        if (fabs(vx) + fabs(vy) < 1e-300 * lmax || a > c) { vx = lmax - c; vy = 0.5 * b; }
        const double n2 = vx * vx + vy * vy;
        if (!(n2 > 0.0)) return;
        const double d = (lmax - 1000.0 * lmin) / n2;
        a -= d * vx * vx; b -= 2.0 * d * vx * vy; c -= d * vy * vy;
        lmax = 1000.0 * lmin;
        if (!(a > 0.0) || !(c > 0.0) || !(a * c - 0.25 * b * b > 0.0)) return;
    }
    const double s = 1.001 + 0.05 * (double)(sqrtf((float)lmax) * 1.000001f);   // 0.05 px / semi-minor axis (= 1/sqrt(lmax))
    const double px = cx * focal_x + W / 2. - 0.5, py = cy * focal_y + H / 2. - 0.5;      // pixel-index coordinates
    if (!(fabs(px) < 8192.0) || !(fabs(py) < 8192.0)) return;
    const double is2 = __builtin_amdgcn_rcp(s * s) * (1.0 - 1e-7);      // rounded down: the form only shrinks, the ellipse only grows
    const float fa = (float)(a * is2), fb = (float)(b * is2), fc = (float)(c * is2);
    if (!(fa > 1.0e-30f) || !(fc > 1.0e-30f) || !(fa < 1.0e30f) || !(fc < 1.0e30f)) return;
    e = make_float4((float)px, (float)py, fa, fb);
    ec = fc;
}

// View-independent part of the projection, once per Gaussian instead of once per (view, Gaussian) (option pre_hoist; VERDICT r04 item 6:
// "hoist through memory, not registers"): the 3D covariance (computeCov3D, forward.cu:129-163), the rotation matrix of the
// quaternion, and the float64 reciprocals of the squared scales that computeView2Gaussian forms (forward.cu:168-200) -- 24 floats = 96
// bytes per Gaussian, [0..5] Sigma, [6..14] R (column-major m[c][r]), [15] unused, [16..21] Sx, Sy, Sz as float64 bit pairs. The same
// operations in the same order as the per-view code, so every output of the projection stays bit-identical.
__global__ void __launch_bounds__(256)
preprocess_hoist_kernel(int n, const float* __restrict__ scales, float scale_modifier, const float* __restrict__ rotations, float4* __restrict__ out)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
    const float3 scale = make_float3(scales[3 * (size_t)g], scales[3 * (size_t)g + 1], scales[3 * (size_t)g + 2]);
    const float4 rot = reinterpret_cast<const float4*>(rotations)[g];
    M3 S = {};
    S.m[0][0] = scale_modifier * scale.x;
    S.m[1][1] = scale_modifier * scale.y;
    S.m[2][2] = scale_modifier * scale.z;
    const M3 R = quat_to_R(rot);
    const M3 Mm = mul(S, R);
    const M3 Sigma = mul(transpose(Mm), Mm);
    const double Sx = 1.0f / ((double)scale.x * scale.x + 1e-7);
    const double Sy = 1.0f / ((double)scale.y * scale.y + 1e-7);
    const double Sz = 1.0f / ((double)scale.z * scale.z + 1e-7);
    float4* o = out + 6 * (size_t)g;
    o[0] = make_float4(Sigma.m[0][0], Sigma.m[0][1], Sigma.m[0][2], Sigma.m[1][1]);
    o[1] = make_float4(Sigma.m[1][2], Sigma.m[2][2], R.m[0][0], R.m[0][1]);
    o[2] = make_float4(R.m[0][2], R.m[1][0], R.m[1][1], R.m[1][2]);
    o[3] = make_float4(R.m[2][0], R.m[2][1], R.m[2][2], 0.0f);
    o[4] = make_float4(__int_as_float(__double2loint(Sx)), __int_as_float(__double2hiint(Sx)), __int_as_float(__double2loint(Sy)), __int_as_float(__double2hiint(Sy)));
    o[5] = make_float4(__int_as_float(__double2loint(Sz)), __int_as_float(__double2hiint(Sz)), 0.0f, 0.0f);
}

// SAVE_AUX = false (inference calls): the planes only the backward and the debug export read -- and the arithmetic only they need
// (the 2D conic: one IEEE division) -- are not produced.
template <bool SAVE_AUX, bool HOIST>
__global__ void __launch_bounds__(F3DG_BLOCK)
preprocess_kernel(int P, int D, int M, int views_per_set, const float4* __restrict__ hoist,
                  const float* __restrict__ means3D, const float* __restrict__ scales, float scale_modifier,
                  const float* __restrict__ rotations, const float* __restrict__ opacities,
                  const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
                  const float* __restrict__ colors_precomp, const float* __restrict__ v2g_precomp,
                  const float* __restrict__ viewmatrices, const float* __restrict__ projmatrices,
                  const float* __restrict__ cam_positions, int W, int H, int grid_x, int grid_y,
                  float tan_fovx, float tan_fovy, float focal_x, float focal_y, float kernel_size,
                  F3dgRec* __restrict__ rec, float2* __restrict__ means2D, float* __restrict__ depths_out,
                  unsigned* __restrict__ sort_keys, uint2* __restrict__ rects,
                  float4* __restrict__ bbox_out, float4* __restrict__ cull_out,
                  float4* __restrict__ conic_out,
                  int* __restrict__ radii, unsigned* __restrict__ tiles_touched,
                  unsigned char* __restrict__ clamped, int debug_skip_all, int tile_cull,
                  double inv_focal_x, double inv_focal_y, F3dgHeaderInit init, int chunk_major, uint2* __restrict__ chunk_boxes)
{
    if (init.hdr != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {
        // the workspace header (read by every later kernel of the call, by nothing in this one): zeroed, then its constants
        reinterpret_cast<unsigned*>(init.hdr)[threadIdx.x] = 0;
        if (threadIdx.x == 0) {
            init.hdr->capacity = init.capacity; init.hdr->alpha_fast = init.alpha_fast; init.hdr->save_aux = init.save_aux;
            init.hdr->small_path = init.small_path;
            init.hdr->small_shape[0] = init.shape[0]; init.hdr->small_shape[1] = init.shape[1];
            init.hdr->small_shape[2] = init.shape[2]; init.hdr->small_shape[3] = init.shape[3];
        }
    }
    // chunk_major (option pre_order = 1): blockIdx.x is the VIEW, so the workgroups that follow each other in dispatch order are the
    // views of ONE chunk of 256 Gaussians and its 23 KB of inputs are fetched once per XCD instead of once per view
    const int g = ((chunk_major & 1) ? blockIdx.y : blockIdx.x) * F3DG_BLOCK + threadIdx.x;
    const int v = (chunk_major & 1) ? blockIdx.x : blockIdx.y;
    const size_t gs = (size_t)(v / views_per_set) * P + g;       // this view's Gaussian set: inputs are [n_sets, P, ...]
    if (g >= P)
        return;
    const size_t idx = (size_t)v * P + g;
    const float* view = viewmatrices + 16 * v;      // wave-uniform -> scalar loads
    const float* proj = projmatrices + 16 * v;

    int my_radii = 0;
    unsigned my_tiles = 0;
    float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;   // the 64-byte record
    float2 xy = make_float2(0, 0);
    float4 con = make_float4(0, 0, 0, 0);
    float4 box = make_float4(-3.0e38f, 3.0e38f, -3.0e38f, 3.0e38f);     // "everything" unless proven smaller
    float4 ce = make_float4(-1.0e9f, -1.0e9f, 1.0e30f, 0.0f);      // "never": culled Gaussians are in no list anyway
    float cec = 1.0e30f;
    unsigned char clamp_bits = 0;
    float depth = 0.0f;
    uint2 rect = make_uint2(0u, 0u);          // tile rectangle (rminx | rmaxx << 16, rminy | rmaxy << 16): empty unless visible

    const float px_ = means3D[3 * gs], py_ = means3D[3 * gs + 1], pz_ = means3D[3 * gs + 2];

    // in_frustum (auxiliary.h:177-202): near cull only, the x/y test is commented out in the reference
    const float pvx = view[0] * px_ + view[4] * py_ + view[8] * pz_ + view[12];
    const float pvy = view[1] * px_ + view[5] * py_ + view[9] * pz_ + view[13];
    const float pvz = view[2] * px_ + view[6] * py_ + view[10] * pz_ + view[14];

    if (!(pvz <= 0.2f)) {
        const float hx = proj[0] * px_ + proj[4] * py_ + proj[8] * pz_ + proj[12];
        const float hy = proj[1] * px_ + proj[5] * py_ + proj[9] * pz_ + proj[13];
        const float hw = proj[3] * px_ + proj[7] * py_ + proj[11] * pz_ + proj[15];
        const float p_w = 1.0f / (hw + 0.0000001f);
        const float projx = hx * p_w, projy = hy * p_w;

        float3 scale = make_float3(0, 0, 0);
        float4 rot = make_float4(1, 0, 0, 0);
        if (!HOIST) {
            if (scales) scale = make_float3(scales[3 * gs], scales[3 * gs + 1], scales[3 * gs + 2]);
            if (rotations) rot = reinterpret_cast<const float4*>(rotations)[gs];
        }

        // ---- computeCov3D (forward.cu:129-163) or the precomputed one
        float c3[6];
        M3 Rh = {};                     // HOIST: the Gaussian's rotation matrix and scale reciprocals from preprocess_hoist_kernel
        double Sxh = 0.0, Syh = 0.0, Szh = 0.0;
        if (HOIST) {
            const float4* h = hoist + 6 * gs;
            const float4 h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3], h4 = h[4], h5 = h[5];
            c3[0] = h0.x; c3[1] = h0.y; c3[2] = h0.z; c3[3] = h0.w; c3[4] = h1.x; c3[5] = h1.y;
            Rh.m[0][0] = h1.z; Rh.m[0][1] = h1.w; Rh.m[0][2] = h2.x; Rh.m[1][0] = h2.y; Rh.m[1][1] = h2.z; Rh.m[1][2] = h2.w;
            Rh.m[2][0] = h3.x; Rh.m[2][1] = h3.y; Rh.m[2][2] = h3.z;
            Sxh = __hiloint2double(__float_as_int(h4.y), __float_as_int(h4.x));
            Syh = __hiloint2double(__float_as_int(h4.w), __float_as_int(h4.z));
            Szh = __hiloint2double(__float_as_int(h5.y), __float_as_int(h5.x));
        } else if (cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * gs + i];
        } else {
            M3 S = {};
            S.m[0][0] = scale_modifier * scale.x;
            S.m[1][1] = scale_modifier * scale.y;
            S.m[2][2] = scale_modifier * scale.z;
            const M3 R = quat_to_R(rot);
            const M3 Mm = mul(S, R);
            const M3 Sigma = mul(transpose(Mm), Mm);
            c3[0] = Sigma.m[0][0]; c3[1] = Sigma.m[0][1]; c3[2] = Sigma.m[0][2];
            c3[3] = Sigma.m[1][1]; c3[4] = Sigma.m[1][2]; c3[5] = Sigma.m[2][2];
        }

        // ---- computeCov2D (forward.cu:74-124)
        float tx = pvx, ty = pvy;
        const float tz = pvz;
        const float limx = 1.3f * tan_fovx;
        const float limy = 1.3f * tan_fovy;
        const float txtz = tx / tz;
        const float tytz = ty / tz;
        tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
        ty = fminf(limy, fmaxf(-limy, tytz)) * tz;

        M3 J;
        J.m[0][0] = focal_x / tz; J.m[0][1] = 0.0f;         J.m[0][2] = -(focal_x * tx) / (tz * tz);
        J.m[1][0] = 0.0f;         J.m[1][1] = focal_y / tz; J.m[1][2] = -(focal_y * ty) / (tz * tz);
        J.m[2][0] = 0.0f;         J.m[2][1] = 0.0f;         J.m[2][2] = 0.0f;
        M3 Wm;
        Wm.m[0][0] = view[0]; Wm.m[0][1] = view[4]; Wm.m[0][2] = view[8];
        Wm.m[1][0] = view[1]; Wm.m[1][1] = view[5]; Wm.m[1][2] = view[9];
        Wm.m[2][0] = view[2]; Wm.m[2][1] = view[6]; Wm.m[2][2] = view[10];
        const M3 T = mul(Wm, J);
        M3 Vrk;
        Vrk.m[0][0] = c3[0]; Vrk.m[0][1] = c3[1]; Vrk.m[0][2] = c3[2];
        Vrk.m[1][0] = c3[1]; Vrk.m[1][1] = c3[3]; Vrk.m[1][2] = c3[4];
        Vrk.m[2][0] = c3[2]; Vrk.m[2][1] = c3[4]; Vrk.m[2][2] = c3[5];
        const M3 cov = mul(mul(transpose(T), transpose(Vrk)), T);

        const float det_0 = (float)fmax(1e-6, (double)(cov.m[0][0] * cov.m[1][1] - cov.m[0][1] * cov.m[0][1]));
        const float det_1 = (float)fmax(1e-6, (double)((cov.m[0][0] + kernel_size) * (cov.m[1][1] + kernel_size) - cov.m[0][1] * cov.m[0][1]));
        float coef = (float)sqrt(det_0 / (det_1 + 1e-6) + 1e-6);
        if (det_0 <= 1e-6 || det_1 <= 1e-6)
            coef = 0.0f;
        const float cx = cov.m[0][0] + kernel_size;
        const float cy = cov.m[0][1];
        const float cz = cov.m[1][1] + kernel_size;

        // ---- invert, extent, tile rectangle (forward.cu:350-374)
        const float det = (cx * cz - cy * cy);
        if (det != 0.0f) {
            float conic_x = 0.0f, conic_y = 0.0f, conic_z = 0.0f;
            if (SAVE_AUX) {
                const float det_inv = 1.f / det;
                conic_x = cz * det_inv; conic_y = -cy * det_inv; conic_z = cx * det_inv;
            }
            const float mid = 0.5f * (cx + cz);
            const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            const float pix_x = ndc2Pix(projx, W), pix_y = ndc2Pix(projy, H);
            const int max_radius = (int)my_radius;
            const int rminx = min(grid_x, max(0, (int)((pix_x - max_radius) / F3DG_TILE)));
            const int rminy = min(grid_y, max(0, (int)((pix_y - max_radius) / F3DG_TILE)));
            const int rmaxx = min(grid_x, max(0, (int)((pix_x + max_radius + F3DG_TILE - 1) / F3DG_TILE)));
            const int rmaxy = min(grid_y, max(0, (int)((pix_y + max_radius + F3DG_TILE - 1) / F3DG_TILE)));
            const int area = (rmaxx - rminx) * (rmaxy - rminy);
            if (area != 0) {
                // ---- colour (forward.cu:20-71) or precomputed
                float cr, cg, cb;
                if (colors_precomp) {
                    cr = colors_precomp[3 * gs]; cg = colors_precomp[3 * gs + 1]; cb = colors_precomp[3 * gs + 2];
                } else {
                    const float* campos = cam_positions + 3 * v;
                    float dx = px_ - campos[0], dy = py_ - campos[1], dz = pz_ - campos[2];
                    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
                    dx = dx / len; dy = dy / len; dz = dz / len;
                    const float* sh = shs + gs * M * 3;
                    float res[3];
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) res[ch] = SH_C0 * sh[ch];
                    if (D > 0) {
                        const float x = dx, y = dy, z = dz;
#pragma unroll
                        for (int ch = 0; ch < 3; ch++)
                            res[ch] = res[ch] - SH_C1 * y * sh[3 + ch] + SH_C1 * z * sh[6 + ch] - SH_C1 * x * sh[9 + ch];
                        if (D > 1) {
                            const float xx = x * x, yy = y * y, zz = z * z;
                            const float xy_ = x * y, yz = y * z, xz = x * z;
#pragma unroll
                            for (int ch = 0; ch < 3; ch++)
                                res[ch] = res[ch] +
                                    SH_C2[0] * xy_ * sh[12 + ch] +
                                    SH_C2[1] * yz * sh[15 + ch] +
                                    SH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + ch] +
                                    SH_C2[3] * xz * sh[21 + ch] +
                                    SH_C2[4] * (xx - yy) * sh[24 + ch];
                            if (D > 2) {
#pragma unroll
                                for (int ch = 0; ch < 3; ch++)
                                    res[ch] = res[ch] +
                                        SH_C3[0] * y * (3.0f * xx - yy) * sh[27 + ch] +
                                        SH_C3[1] * xy_ * z * sh[30 + ch] +
                                        SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + ch] +
                                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + ch] +
                                        SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + ch] +
                                        SH_C3[5] * z * (xx - yy) * sh[42 + ch] +
                                        SH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + ch];
                            }
                        }
                    }
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        res[ch] += 0.5f;
                        if (res[ch] < 0) clamp_bits |= (unsigned char)(1u << ch);
                        res[ch] = fmaxf(res[ch], 0.0f);
                    }
                    cr = res[0]; cg = res[1]; cb = res[2];
                }

                // ---- view2gaussian (forward.cu:168-279) or precomputed [V,P,10]
                float vg[10];
                if (v2g_precomp) {
#pragma unroll
                    for (int i = 0; i < 10; i++) vg[i] = v2g_precomp[idx * 10 + i];
                } else {
                    const M3 R = HOIST ? Rh : quat_to_R(rot);
                    // G2V = W2V * G2W (glm mat4 product, 4 terms left to right); G2W columns are the ROWS of R
                    // with the mean as 4th column; W2V[c][r] = view[4c + r].
                    float G2V[4][3];   // [col][row], rows 0..2 only (row 3 is never read)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float b0 = R.m[0][c], b1 = R.m[1][c], b2 = R.m[2][c];    // G2W[c] = (R[0][c], R[1][c], R[2][c], 0)
#pragma unroll
                        for (int q = 0; q < 3; q++)
                            G2V[c][q] = view[q] * b0 + view[4 + q] * b1 + view[8 + q] * b2 + view[12 + q] * 0.0f;
                    }
#pragma unroll
                    for (int q = 0; q < 3; q++)
                        G2V[3][q] = view[q] * px_ + view[4 + q] * py_ + view[8 + q] * pz_ + view[12 + q] * 1.0f;

                    M3 Rt;   // "R_transpose": columns (G2V[0][k], G2V[1][k], G2V[2][k])
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        Rt.m[c][0] = G2V[0][c]; Rt.m[c][1] = G2V[1][c]; Rt.m[c][2] = G2V[2][c];
                    }
                    const float t_x = G2V[3][0], t_y = G2V[3][1], t_z = G2V[3][2];
                    // t2 = (-R_transpose) * t
                    const float t2x = (-Rt.m[0][0]) * t_x + (-Rt.m[1][0]) * t_y + (-Rt.m[2][0]) * t_z;
                    const float t2y = (-Rt.m[0][1]) * t_x + (-Rt.m[1][1]) * t_y + (-Rt.m[2][1]) * t_z;
                    const float t2z = (-Rt.m[0][2]) * t_x + (-Rt.m[1][2]) * t_y + (-Rt.m[2][2]) * t_z;

                    const double Sx = HOIST ? Sxh : 1.0f / ((double)scale.x * scale.x + 1e-7);
                    const double Sy = HOIST ? Syh : 1.0f / ((double)scale.y * scale.y + 1e-7);
                    const double Sz = HOIST ? Szh : 1.0f / ((double)scale.z * scale.z + 1e-7);
                    const double Cc = t2x * t2x * Sx + t2y * t2y * Sy + t2z * t2z * Sz;

                    M3 SR;
                    SR.m[0][0] = (float)(Sx * Rt.m[0][0]); SR.m[0][1] = (float)(Sy * Rt.m[0][1]); SR.m[0][2] = (float)(Sz * Rt.m[0][2]);
                    SR.m[1][0] = (float)(Sx * Rt.m[1][0]); SR.m[1][1] = (float)(Sy * Rt.m[1][1]); SR.m[1][2] = (float)(Sz * Rt.m[1][2]);
                    SR.m[2][0] = (float)(Sx * Rt.m[2][0]); SR.m[2][1] = (float)(Sy * Rt.m[2][1]); SR.m[2][2] = (float)(Sz * Rt.m[2][2]);

                    // B = t2 * SR (row vector times matrix): B_c = (SR[c][0]*t2.x + SR[c][1]*t2.y) + SR[c][2]*t2.z
                    const float Bx = SR.m[0][0] * t2x + SR.m[0][1] * t2y + SR.m[0][2] * t2z;
                    const float By = SR.m[1][0] * t2x + SR.m[1][1] * t2y + SR.m[1][2] * t2z;
                    const float Bz = SR.m[2][0] * t2x + SR.m[2][1] * t2y + SR.m[2][2] * t2z;
                    const M3 Sig = mul(transpose(Rt), SR);
                    vg[0] = Sig.m[0][0]; vg[1] = Sig.m[0][1]; vg[2] = Sig.m[0][2];
                    vg[3] = Sig.m[1][1]; vg[4] = Sig.m[1][2]; vg[5] = Sig.m[2][2];
                    vg[6] = Bx; vg[7] = By; vg[8] = Bz; vg[9] = (float)Cc;
                }

                const float opac = opacities[gs] * coef;
                my_radii = max_radius;
                my_tiles = (unsigned)area;
                rect = make_uint2((unsigned)rminx | ((unsigned)rmaxx << 16), (unsigned)rminy | ((unsigned)rmaxy << 16));
                xy = make_float2(pix_x, pix_y);
                con = make_float4(conic_x, conic_y, conic_z, opac);
                r0 = make_float4(vg[0], vg[1], vg[2], vg[3]);
                r1 = make_float4(vg[4], vg[5], vg[6], vg[7]);
                // Pre-test constant of the compositing kernel (see pretest_constant). alpha = min(.99, opac*exp(power))
                // < 1/255 whenever power < thr = log(1/(255*opac)); 1e-4 of slack covers logf/expf ulps and the final float
                // rounding of power. opac <= 0 -> alpha <= 0 always (K = +inf); NaN opacity disables the test.
                const float thr = debug_skip_all ? __builtin_inff() : opac > 0.0f ? -0.6931471805599453f * __builtin_amdgcn_logf(255.0f * opac) - 1e-4f : (opac <= 0.0f ? __builtin_inff() : opac);
                const float Kpre = pretest_constant(vg, thr);
                {
                    const bool have_scale = scales != nullptr && v2g_precomp == nullptr;
                    CullConic cq;
                    cq.ok = false;
                    if (have_scale && thr < 3.0e38f)
                        cq = cull_conic(vg, thr, tan_fovx, tan_fovy);
                    if (bbox_out)
                        box = conservative_box(cq, thr, have_scale, W, H, focal_x, focal_y);
                    conservative_ellipse(cq, thr, have_scale, W, H, focal_x, focal_y, inv_focal_x, inv_focal_y, ce, cec);
                }
                {
                    // Tile culling (option "tile_cull"): the reference instantiates the Gaussian in every tile of the square around
                    // its 3-sigma circle (rect above); only tiles that the axis-aligned box of the conservative ellipse reaches can
                    // hold a pixel with alpha >= 1/255, the others would be a bare `continue` for each of their pixels.
                    const float edet = fmaf(ce.z, cec, -0.25f * ce.w * ce.w);
                    if (edet > 0.0f) {
                        // (v_rcp_f32 / v_sqrt_f32, 1 ulp each: far inside the 0.05 % + 2e-3 px margin of this box)
                        const float iedet = __builtin_amdgcn_rcpf(edet);
                        const float hx = __builtin_amdgcn_sqrtf(cec * iedet) * 1.0005f + 2e-3f, hy = __builtin_amdgcn_sqrtf(ce.z * iedet) * 1.0005f + 2e-3f;
                        int fminx = rminx, fmaxx = rmaxx, fminy = rminy, fmaxy = rmaxy;
                        if (tile_cull) {
                            // tile t holds the pixel centres 16 t .. 16 t + 15
                            const float inv_tile = 1.0f / (float)F3DG_TILE;
                            const float lx = ceilf((ce.x - hx - (float)(F3DG_TILE - 1)) * inv_tile), ux = floorf((ce.x + hx) * inv_tile) + 1.0f;
                            const float ly = ceilf((ce.y - hy - (float)(F3DG_TILE - 1)) * inv_tile), uy = floorf((ce.y + hy) * inv_tile) + 1.0f;
                            const int cminx = max(rminx, (int)fminf(fmaxf(lx, 0.0f), (float)grid_x));
                            const int cmaxx = min(rmaxx, (int)fminf(fmaxf(ux, 0.0f), (float)grid_x));
                            const int cminy = max(rminy, (int)fminf(fmaxf(ly, 0.0f), (float)grid_y));
                            const int cmaxy = min(rmaxy, (int)fminf(fmaxf(uy, 0.0f), (float)grid_y));
                            if (cmaxx > cminx && cmaxy > cminy) {
                                my_tiles = (unsigned)((cmaxx - cminx) * (cmaxy - cminy));
                                rect = make_uint2((unsigned)cminx | ((unsigned)cmaxx << 16), (unsigned)cminy | ((unsigned)cmaxy << 16));
                                fminx = cminx; fmaxx = cmaxx; fminy = cminy; fmaxy = cmaxy;
                            } else {
                                my_tiles = 0;
                                rect = make_uint2(0u, 0u);
                            }
                        }
                        // Quadrant hints for the one-wave compositing kernel (f3dg_common.h: F3DG_RECT_SKIP_*): tile t's first
                        // half holds the pixel centres 16 t .. 16 t + 7, its second half 16 t + 8 .. 16 t + 15. The same box.
                        if (my_tiles != 0) {
                            if (ce.x - hx > (float)(F3DG_TILE * fminx + 7)) rect.x |= F3DG_RECT_SKIP_LO;
                            if (ce.x + hx < (float)(F3DG_TILE * (fmaxx - 1) + 8)) rect.x |= F3DG_RECT_SKIP_HI;
                            if (ce.y - hy > (float)(F3DG_TILE * fminy + 7)) rect.y |= F3DG_RECT_SKIP_LO;
                            if (ce.y + hy < (float)(F3DG_TILE * (fmaxy - 1) + 8)) rect.y |= F3DG_RECT_SKIP_HI;
                        }
                    }
                }
                r2 = make_float4(vg[8], vg[9], opac, Kpre);
                r3 = make_float4(cr, cg, cb, 0.0f);
                depth = pvz;
            }
        }
    }

    radii[idx] = my_radii;
    rects[idx] = rect;
    if (chunk_boxes) {
        // small-call path (f3dg_small.hip): the union of the tile rectangles of this wave's 64 consecutive Gaussians, so that a tile's
        // workgroup tests 64 Gaussians with one comparison before it reads their rectangles. (The view's last, partial wave -- the
        // lanes beyond P have left -- publishes "everything".)
        uint2 bx = make_uint2(F3DG_RECT_COORD << 16, F3DG_RECT_COORD << 16);
        if (__ballot(true) == ~0ull) {
            const bool vis = ((rect.x >> 16) & F3DG_RECT_COORD) > (rect.x & F3DG_RECT_COORD) && ((rect.y >> 16) & F3DG_RECT_COORD) > (rect.y & F3DG_RECT_COORD);
            const unsigned mnx = __reduce_min_sync(~0ull, vis ? rect.x & F3DG_RECT_COORD : F3DG_RECT_COORD);
            const unsigned mxx = __reduce_max_sync(~0ull, vis ? (rect.x >> 16) & F3DG_RECT_COORD : 0u);
            const unsigned mny = __reduce_min_sync(~0ull, vis ? rect.y & F3DG_RECT_COORD : F3DG_RECT_COORD);
            const unsigned mxy = __reduce_max_sync(~0ull, vis ? (rect.y >> 16) & F3DG_RECT_COORD : 0u);
            bx = make_uint2(mnx | (mxx << 16), mny | (mxy << 16));
        }
        if ((threadIdx.x & 63u) == 0u) chunk_boxes[(size_t)v * ((P + 63) / 64) + (g >> 6)] = bx;
    }
    sort_keys[idx] = my_tiles ? __float_as_uint(depth) : 0xFFFFFFFFu;      // key of the per-view depth sort (f3dg_binning.hip)
    if (bbox_out) bbox_out[idx] = box;
    // The 64-byte record and the 16-byte ellipse are read through the tile lists only: a (view, Gaussian) pair that is in no list
    // (behind the camera, off screen, or -- with tile culling -- nowhere above alpha 1/255) does not write its 80 bytes. (SAVE_AUX
    // calls write them all: the debug export hands the arrays out whole.)
    if (SAVE_AUX || my_tiles != 0u) {
        r3.w = cec;                         // record slot 15: the ellipse's c (the depth lives in depths_out)
        float4* dst = reinterpret_cast<float4*>(rec + idx);
        if (chunk_major & 2) {
            // (option pre_order bit 1: the 80 bytes per pair that nothing reads before the compositing kernel leave as streaming
            // stores, so that they do not push the chunk's inputs -- and the keys / rectangles the binning stage reads next -- out of L2)
            typedef float f4 __attribute__((ext_vector_type(4)));
            auto nt = [](float4* p, const float4& v) { f4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<f4*>(p)); };
            nt(cull_out + idx, ce);
            nt(dst, r0); nt(dst + 1, r1); nt(dst + 2, r2); nt(dst + 3, r3);
        } else {
            cull_out[idx] = ce;
            dst[0] = r0; dst[1] = r1; dst[2] = r2; dst[3] = r3;
        }
    }
    if (SAVE_AUX) {                         // planes only the backward and the debug export read
        tiles_touched[idx] = my_tiles;
        means2D[idx] = xy;
        depths_out[idx] = depth;
        conic_out[idx] = con;
        clamped[idx] = clamp_bits;
    }
}

__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view,
                                    unsigned char* __restrict__ present)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P) return;
    const float x = means3D[3 * (size_t)g], y = means3D[3 * (size_t)g + 1], z = means3D[3 * (size_t)g + 2];
    const float pvz = view[2] * x + view[6] * y + view[10] * z + view[14];
    present[g] = !(pvz <= 0.2f);
}

} // namespace

int f3dg_launch_preprocess(hipStream_t s, int V, int views_per_set, int P, int D, int M, const float* means3D, const float* scales,
                           float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                           const float* cov3D_precomp, const float* colors_precomp, const float* v2g_precomp,
                           const float* viewmatrix, const float* projmatrix, const float* cam_pos, int W, int H,
                           float tan_fovx, float tan_fovy, float focal_x, float focal_y, float kernel_size,
                           F3dgRec* rec, float2* means2D, float* depths, unsigned* sort_keys, uint2* rects, float4* bbox, float4* cull, float4* conic, int* radii,
                           unsigned* tiles, unsigned char* clamped, int save_aux, int tile_cull, F3dgHeaderInit init, float4* hoist, int n_sets, uint2* chunk_boxes)
{
    const int grid_x = (W + F3DG_TILE - 1) / F3DG_TILE, grid_y = (H + F3DG_TILE - 1) / F3DG_TILE;
    // option pre_hoist: the view-independent part once per Gaussian (only the common input form: scales + rotations, nothing precomputed)
    if (hoist && !(scales && rotations && !cov3D_precomp && !v2g_precomp)) hoist = nullptr;
    if (hoist)
        F3DG_KLAUNCH(preprocess_hoist_kernel, dim3((unsigned)(((size_t)n_sets * P + 255) / 256)), dim3(256), 0, s, n_sets * P, scales, scale_modifier, rotations, hoist);
    const int chunks = (P + F3DG_BLOCK - 1) / F3DG_BLOCK;
    const int chunk_major = ((g_f3dg_pre_order & 1) && chunks <= 65535 ? 1 : 0) | (g_f3dg_pre_order & 2);
    dim3 grid((chunk_major & 1) ? V : chunks, (chunk_major & 1) ? chunks : V, 1);
#define F3DG_LAUNCH_PRE2(AUX, HST) F3DG_KLAUNCH((preprocess_kernel<AUX, HST>), grid, dim3(F3DG_BLOCK), 0, s, P, D, M, views_per_set > 0 ? views_per_set : V, hoist, means3D, scales,     \
                       scale_modifier, rotations, opacities, shs, cov3D_precomp, colors_precomp, v2g_precomp, viewmatrix, projmatrix,                          \
                       cam_pos, W, H, grid_x, grid_y, tan_fovx, tan_fovy, focal_x, focal_y, kernel_size, rec, means2D,                                          \
                       depths, sort_keys, rects, bbox, cull, conic, radii, tiles, clamped, g_f3dg_debug_skip_all, tile_cull,                                    \
                       1.0 / (double)focal_x, 1.0 / (double)focal_y, init, chunk_major, chunk_boxes)
#define F3DG_LAUNCH_PRE(AUX) do { if (hoist) F3DG_LAUNCH_PRE2(AUX, true); else F3DG_LAUNCH_PRE2(AUX, false); } while (0)
    if (save_aux) F3DG_LAUNCH_PRE(true); else F3DG_LAUNCH_PRE(false);
#undef F3DG_LAUNCH_PRE
#undef F3DG_LAUNCH_PRE2
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}

extern "C" int f3dg_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix,
                                 const float* projmatrix, uint8_t* present)
{
    (void)projmatrix;   // the reference computes p_proj but only tests view-space z (auxiliary.h:192)
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) return F3DG_ERR_BAD_ARG;
    if (P == 0) return F3DG_OK;
    F3DG_KLAUNCH(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, means3D,
                       viewmatrix, present);
    F3DG_HIP_CHECK(hipGetLastError());
    return F3DG_OK;
}
