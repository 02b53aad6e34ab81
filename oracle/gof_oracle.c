/*
 * gof_oracle.c -- CPU restatement of the GOF rasterizer hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the parity ORACLE for the HIP kernels in f3d-gaus_amd/csrc. It is plain C,
 * written from the reference's algorithm; it is never linked into, imported by, or called
 * from the product path. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.
 *
 * PARITY PIN STATUS (also stated in DESIGN.md):
 *   "parity unpinned" for the rasterizer arithmetic as a whole. The reference rasterizer is
 *   CUDA-only (needs nvcc, the CUDA runtime headers and CUB -- none exist in this image) and
 *   ships no tests or golden vectors for this path, so it can be neither built nor run here.
 *   What IS pinned: the glm evaluation orders used below are checked against the reference's
 *   vendored, header-only glm compiled by g++ (oracle/ref_glm_check.cpp -> oracle/_ref/), and
 *   the SH colour / covariance / camera / splat-head Python pieces are pinned by fixtures
 *   generated from the importable Python half of the reference (tests/golden/).
 *
 * Reference files followed (RAST = src/gaussian-splatting/submodules/diff-gof-rasterization):
 *   RAST/cuda_rasterizer/auxiliary.h       constants, ndc2Pix, getRect, transformPoint*, in_frustum
 *   RAST/cuda_rasterizer/forward.cu        computeColorFromSH, computeCov2D, computeCov3D,
 *                                          computeView2Gaussian, preprocessCUDA, renderCUDA
 *   RAST/cuda_rasterizer/rasterizer_impl.cu getHigherMsb, duplicateWithKeys, identifyTileRanges,
 *                                          Rasterizer::forward / backward orchestration
 *                                          preprocessPointsCUDA, integrateCUDA; createWithKeys, Rasterizer::integrate
 *   RAST/cuda_rasterizer/backward.cu       renderCUDA (bwd), computeView2Gaussian_backward,
 *                                          computeColorFromSH (bwd), preprocessCUDA (bwd)
 *
 * Arithmetic contract: IEEE-754 binary32/binary64, the reference's operation order, the
 * reference's float<->double promotion points, NO fused multiply-add contraction
 * (build with -ffp-contract=off). See SURVEY.md section 0.9 for why this matters.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLOCK_X 16            /* config.h:16 */
#define BLOCK_Y 16            /* config.h:17 */
#define NEAR_PLANE 0.2        /* auxiliary.h:27 (double literal) */
#define FAR_PLANE 100.0       /* auxiliary.h:28 (double literal) */
#define DEPTH_OFFSET 6        /* auxiliary.h:21 */
#define ALPHA_OFFSET 7        /* auxiliary.h:22 */
#define DISTORTION_OFFSET 8   /* auxiliary.h:23 */

/* auxiliary.h:40-57 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f };
static const float SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                -0.5900435899266435f };

/* ---- glm stand-ins: column-major, m.c[col][row]; evaluation order as glm 0.9.9.9 ---------- */
typedef struct { float c[3][3]; } mat3;
typedef struct { float c[4][4]; } mat4;
typedef struct { float x, y, z; } vec3;

/* glm/detail/type_mat3x3.inl operator*(mat3, mat3): Result[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2] */
static mat3 m3_mul(mat3 a, mat3 b)
{
    mat3 r;
    for (int c = 0; c < 3; c++)
        for (int q = 0; q < 3; q++)
            r.c[c][q] = a.c[0][q] * b.c[c][0] + a.c[1][q] * b.c[c][1] + a.c[2][q] * b.c[c][2];
    return r;
}
static mat3 m3_transpose(mat3 a)
{
    mat3 r;
    for (int c = 0; c < 3; c++)
        for (int q = 0; q < 3; q++)
            r.c[c][q] = a.c[q][c];
    return r;
}
/* glm mat3 * vec3: (m[0][r]*v.x + m[1][r]*v.y) + m[2][r]*v.z */
static vec3 m3_mul_v(mat3 m, vec3 v)
{
    vec3 r;
    r.x = m.c[0][0] * v.x + m.c[1][0] * v.y + m.c[2][0] * v.z;
    r.y = m.c[0][1] * v.x + m.c[1][1] * v.y + m.c[2][1] * v.z;
    r.z = m.c[0][2] * v.x + m.c[1][2] * v.y + m.c[2][2] * v.z;
    return r;
}
/* glm vec3 * mat3: (m[c][0]*v.x + m[c][1]*v.y) + m[c][2]*v.z */
static vec3 v_mul_m3(vec3 v, mat3 m)
{
    vec3 r;
    r.x = m.c[0][0] * v.x + m.c[0][1] * v.y + m.c[0][2] * v.z;
    r.y = m.c[1][0] * v.x + m.c[1][1] * v.y + m.c[1][2] * v.z;
    r.z = m.c[2][0] * v.x + m.c[2][1] * v.y + m.c[2][2] * v.z;
    return r;
}
/* glm mat4 * mat4: Result[c] = ((A[0]*B[c][0] + A[1]*B[c][1]) + A[2]*B[c][2]) + A[3]*B[c][3] */
static mat4 m4_mul(mat4 a, mat4 b)
{
    mat4 r;
    for (int c = 0; c < 4; c++)
        for (int q = 0; q < 4; q++)
            r.c[c][q] = a.c[0][q] * b.c[c][0] + a.c[1][q] * b.c[c][1] + a.c[2][q] * b.c[c][2] + a.c[3][q] * b.c[c][3];
    return r;
}

/* exported for oracle/ref_glm_check (compares these against the reference's vendored glm) */
void gof_oracle_m3_mul(const float* a, const float* b, float* out)
{
    mat3 A, B; memcpy(&A, a, sizeof A); memcpy(&B, b, sizeof B);
    mat3 R = m3_mul(A, B); memcpy(out, &R, sizeof R);
}
void gof_oracle_m4_mul(const float* a, const float* b, float* out)
{
    mat4 A, B; memcpy(&A, a, sizeof A); memcpy(&B, b, sizeof B);
    mat4 R = m4_mul(A, B); memcpy(out, &R, sizeof R);
}
void gof_oracle_m3_mul_v(const float* a, const float* v, float* out)
{
    mat3 A; vec3 V; memcpy(&A, a, sizeof A); memcpy(&V, v, sizeof V);
    vec3 R = m3_mul_v(A, V); memcpy(out, &R, sizeof R);
}
void gof_oracle_v_mul_m3(const float* v, const float* a, float* out)
{
    mat3 A; vec3 V; memcpy(&A, a, sizeof A); memcpy(&V, v, sizeof V);
    vec3 R = v_mul_m3(V, A); memcpy(out, &R, sizeof R);
}

/* auxiliary.h:59-62 -- literals are double, result narrowed to float */
static float ndc2Pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

/* auxiliary.h:86-94 */
static vec3 transformPoint4x3(vec3 p, const float* m)
{
    vec3 t;
    t.x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    t.y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    t.z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    return t;
}
/* auxiliary.h:106-115 */
static void transformPoint4x4(vec3 p, const float* m, float out[4])
{
    out[0] = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    out[1] = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    out[2] = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    out[3] = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* auxiliary.h:64-74; (int) truncates toward zero; radius is an int here as in the reference */
static void getRect(float px, float py, int max_radius, int gx, int gy, int rmin[2], int rmax[2])
{
    rmin[0] = imin(gx, imax(0, (int)((px - max_radius) / BLOCK_X)));
    rmin[1] = imin(gy, imax(0, (int)((py - max_radius) / BLOCK_Y)));
    rmax[0] = imin(gx, imax(0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    rmax[1] = imin(gy, imax(0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* forward.cu:138-149: rotation matrix from an UN-normalised quaternion (r,x,y,z); the nine
 * constructor arguments are glm columns. */
static mat3 quat_to_R(const float* rot)
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 R;
    R.c[0][0] = 1.f - 2.f * (y * y + z * z); R.c[0][1] = 2.f * (x * y - r * z);       R.c[0][2] = 2.f * (x * z + r * y);
    R.c[1][0] = 2.f * (x * y + r * z);       R.c[1][1] = 1.f - 2.f * (x * x + z * z); R.c[1][2] = 2.f * (y * z - r * x);
    R.c[2][0] = 2.f * (x * z - r * y);       R.c[2][1] = 2.f * (y * z + r * x);       R.c[2][2] = 1.f - 2.f * (x * x + y * y);
    return R;
}

/* forward.cu:129-163 */
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D)
{
    mat3 S; memset(&S, 0, sizeof S);
    S.c[0][0] = mod * scale[0];
    S.c[1][1] = mod * scale[1];
    S.c[2][2] = mod * scale[2];
    mat3 R = quat_to_R(rot);
    mat3 M = m3_mul(S, R);
    mat3 Sigma = m3_mul(m3_transpose(M), M);
    cov3D[0] = Sigma.c[0][0];
    cov3D[1] = Sigma.c[0][1];
    cov3D[2] = Sigma.c[0][2];
    cov3D[3] = Sigma.c[1][1];
    cov3D[4] = Sigma.c[1][2];
    cov3D[5] = Sigma.c[2][2];
}
void gof_oracle_cov3d(const float* scale, float mod, const float* rot, float* cov3D) { computeCov3D(scale, mod, rot, cov3D); }

/* forward.cu:74-124; out = (cov00 + k, cov01, cov11 + k, coef) */
static void computeCov2D(vec3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                         float kernel_size, const float* cov3D, const float* view, float out[4])
{
    vec3 t = transformPoint4x3(mean, view);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z;
    const float tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;

    mat3 J;
    J.c[0][0] = focal_x / t.z; J.c[0][1] = 0.0f;          J.c[0][2] = -(focal_x * t.x) / (t.z * t.z);
    J.c[1][0] = 0.0f;          J.c[1][1] = focal_y / t.z; J.c[1][2] = -(focal_y * t.y) / (t.z * t.z);
    J.c[2][0] = 0.0f;          J.c[2][1] = 0.0f;          J.c[2][2] = 0.0f;

    mat3 W;
    W.c[0][0] = view[0]; W.c[0][1] = view[4]; W.c[0][2] = view[8];
    W.c[1][0] = view[1]; W.c[1][1] = view[5]; W.c[1][2] = view[9];
    W.c[2][0] = view[2]; W.c[2][1] = view[6]; W.c[2][2] = view[10];

    mat3 T = m3_mul(W, J);

    mat3 Vrk;
    Vrk.c[0][0] = cov3D[0]; Vrk.c[0][1] = cov3D[1]; Vrk.c[0][2] = cov3D[2];
    Vrk.c[1][0] = cov3D[1]; Vrk.c[1][1] = cov3D[3]; Vrk.c[1][2] = cov3D[4];
    Vrk.c[2][0] = cov3D[2]; Vrk.c[2][1] = cov3D[4]; Vrk.c[2][2] = cov3D[5];

    mat3 cov = m3_mul(m3_mul(m3_transpose(T), m3_transpose(Vrk)), T);

    /* CUDA max(double, float) -> double; stored to float (forward.cu:112-113) */
    const float det_0 = (float)fmax(1e-6, (double)(cov.c[0][0] * cov.c[1][1] - cov.c[0][1] * cov.c[0][1]));
    const float det_1 = (float)fmax(1e-6, (double)((cov.c[0][0] + kernel_size) * (cov.c[1][1] + kernel_size) - cov.c[0][1] * cov.c[0][1]));
    float coef = (float)sqrt(det_0 / (det_1 + 1e-6) + 1e-6);
    if (det_0 <= 1e-6 || det_1 <= 1e-6)
        coef = 0.0f;

    cov.c[0][0] += kernel_size;
    cov.c[1][1] += kernel_size;
    out[0] = cov.c[0][0]; out[1] = cov.c[0][1]; out[2] = cov.c[1][1]; out[3] = coef;
}

/* forward.cu:168-279 */
static void computeView2Gaussian(const float* scale, vec3 mean, const float* rot, const float* view, float* v2g)
{
    mat3 R = quat_to_R(rot);

    mat4 G2W;
    G2W.c[0][0] = R.c[0][0]; G2W.c[0][1] = R.c[1][0]; G2W.c[0][2] = R.c[2][0]; G2W.c[0][3] = 0.0f;
    G2W.c[1][0] = R.c[0][1]; G2W.c[1][1] = R.c[1][1]; G2W.c[1][2] = R.c[2][1]; G2W.c[1][3] = 0.0f;
    G2W.c[2][0] = R.c[0][2]; G2W.c[2][1] = R.c[1][2]; G2W.c[2][2] = R.c[2][2]; G2W.c[2][3] = 0.0f;
    G2W.c[3][0] = mean.x;    G2W.c[3][1] = mean.y;    G2W.c[3][2] = mean.z;    G2W.c[3][3] = 1.0f;

    mat4 W2V;
    for (int c = 0; c < 4; c++)
        for (int q = 0; q < 4; q++)
            W2V.c[c][q] = view[4 * c + q];

    mat4 G2V = m4_mul(W2V, G2W);

    mat3 Rt;   /* R_transpose(args) = columns (G2V[0][0],G2V[1][0],G2V[2][0]), ... */
    Rt.c[0][0] = G2V.c[0][0]; Rt.c[0][1] = G2V.c[1][0]; Rt.c[0][2] = G2V.c[2][0];
    Rt.c[1][0] = G2V.c[0][1]; Rt.c[1][1] = G2V.c[1][1]; Rt.c[1][2] = G2V.c[2][1];
    Rt.c[2][0] = G2V.c[0][2]; Rt.c[2][1] = G2V.c[1][2]; Rt.c[2][2] = G2V.c[2][2];

    vec3 t = { G2V.c[3][0], G2V.c[3][1], G2V.c[3][2] };
    mat3 negRt;
    for (int c = 0; c < 3; c++)
        for (int q = 0; q < 3; q++)
            negRt.c[c][q] = -Rt.c[c][q];
    vec3 t2 = m3_mul_v(negRt, t);

    /* forward.cu:255 -- 1.0f / ((double)s*s + 1e-7), all in double */
    double Sx = 1.0f / ((double)scale[0] * scale[0] + 1e-7);
    double Sy = 1.0f / ((double)scale[1] * scale[1] + 1e-7);
    double Sz = 1.0f / ((double)scale[2] * scale[2] + 1e-7);
    /* forward.cu:256 -- t2.x*t2.x is a float product, promoted when multiplied by the double */
    double C = t2.x * t2.x * Sx + t2.y * t2.y * Sy + t2.z * t2.z * Sz;

    mat3 SR;   /* S_inv_square_R: double products narrowed to float by the glm constructor */
    SR.c[0][0] = (float)(Sx * Rt.c[0][0]); SR.c[0][1] = (float)(Sy * Rt.c[0][1]); SR.c[0][2] = (float)(Sz * Rt.c[0][2]);
    SR.c[1][0] = (float)(Sx * Rt.c[1][0]); SR.c[1][1] = (float)(Sy * Rt.c[1][1]); SR.c[1][2] = (float)(Sz * Rt.c[1][2]);
    SR.c[2][0] = (float)(Sx * Rt.c[2][0]); SR.c[2][1] = (float)(Sy * Rt.c[2][1]); SR.c[2][2] = (float)(Sz * Rt.c[2][2]);

    vec3 B = v_mul_m3(t2, SR);
    mat3 Sigma = m3_mul(m3_transpose(Rt), SR);

    v2g[0] = Sigma.c[0][0];
    v2g[1] = Sigma.c[0][1];
    v2g[2] = Sigma.c[0][2];
    v2g[3] = Sigma.c[1][1];
    v2g[4] = Sigma.c[1][2];
    v2g[5] = Sigma.c[2][2];
    v2g[6] = B.x;
    v2g[7] = B.y;
    v2g[8] = B.z;
    v2g[9] = (float)C;
}
void gof_oracle_view2gaussian(const float* scale, const float* mean, const float* rot, const float* view, float* v2g)
{
    vec3 m = { mean[0], mean[1], mean[2] };
    computeView2Gaussian(scale, m, rot, view, v2g);
}

/* forward.cu:20-71 */
static void computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                               const float* shs, uint8_t* clamped, float* rgb_out)
{
    vec3 pos = { means[3 * idx], means[3 * idx + 1], means[3 * idx + 2] };
    vec3 dir = { pos.x - campos[0], pos.y - campos[1], pos.z - campos[2] };
    float len = sqrtf(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
    dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;

    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    float result[3];
    for (int ch = 0; ch < 3; ch++)
        result[ch] = SH_C0 * sh[0 * 3 + ch];

    if (deg > 0) {
        float x = dir.x, y = dir.y, z = dir.z;
        for (int ch = 0; ch < 3; ch++)
            result[ch] = result[ch] - SH_C1 * y * sh[1 * 3 + ch] + SH_C1 * z * sh[2 * 3 + ch] - SH_C1 * x * sh[3 * 3 + ch];
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            for (int ch = 0; ch < 3; ch++)
                result[ch] = result[ch] +
                    SH_C2[0] * xy * sh[4 * 3 + ch] +
                    SH_C2[1] * yz * sh[5 * 3 + ch] +
                    SH_C2[2] * (2.0f * zz - xx - yy) * sh[6 * 3 + ch] +
                    SH_C2[3] * xz * sh[7 * 3 + ch] +
                    SH_C2[4] * (xx - yy) * sh[8 * 3 + ch];
            if (deg > 2) {
                for (int ch = 0; ch < 3; ch++)
                    result[ch] = result[ch] +
                        SH_C3[0] * y * (3.0f * xx - yy) * sh[9 * 3 + ch] +
                        SH_C3[1] * xy * z * sh[10 * 3 + ch] +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11 * 3 + ch] +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12 * 3 + ch] +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13 * 3 + ch] +
                        SH_C3[5] * z * (xx - yy) * sh[14 * 3 + ch] +
                        SH_C3[6] * x * (xx - 3.0f * yy) * sh[15 * 3 + ch];
            }
        }
    }
    for (int ch = 0; ch < 3; ch++) {
        result[ch] += 0.5f;
        clamped[3 * idx + ch] = (result[ch] < 0);
        rgb_out[ch] = fmaxf(result[ch], 0.0f);
    }
}

/* rasterizer_impl.cu:35-50 */
static uint32_t getHigherMsb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}
uint32_t gof_oracle_higher_msb(uint32_t n) { return getHigherMsb(n); }

/* ------------------------------------------------------------------------------------------ */
typedef struct gof_ctx {
    int P, W, H, D, M, gx, gy, R;
    float focal_x, focal_y;
    /* geometry state (rasterizer_impl.cu:188-204) */
    float* depths; uint8_t* clamped; int* radii; float* means2D; float* cov3D; float* v2g;
    float* conic_opacity; float* rgb; uint32_t* tiles_touched; uint32_t* point_offsets;
    /* binning state (rasterizer_impl.cu:230-243) */
    uint64_t* keys; uint32_t* point_list;
    /* image state (rasterizer_impl.cu:218-228) */
    float* final_T; uint32_t* n_contrib; uint32_t* ranges;
    const float* colors_used;     /* colors_precomp or rgb */
    const float* v2g_used;        /* view2gaussian_precomp or v2g */
    int owns_colors;
} gof_ctx;

gof_ctx* gof_oracle_create(void) { return (gof_ctx*)calloc(1, sizeof(gof_ctx)); }

static void ctx_free_buffers(gof_ctx* c)
{
    free(c->depths); free(c->clamped); free(c->radii); free(c->means2D); free(c->cov3D); free(c->v2g);
    free(c->conic_opacity); free(c->rgb); free(c->tiles_touched); free(c->point_offsets);
    free(c->keys); free(c->point_list); free(c->final_T); free(c->n_contrib); free(c->ranges);
    memset(c, 0, sizeof *c);
}
void gof_oracle_destroy(gof_ctx* c) { if (c) { ctx_free_buffers(c); free(c); } }

/* accessors for stage-wise parity tests */
int gof_oracle_num_rendered(const gof_ctx* c) { return c->R; }
const float* gof_oracle_depths(const gof_ctx* c) { return c->depths; }
const float* gof_oracle_means2D(const gof_ctx* c) { return c->means2D; }
const float* gof_oracle_cov3D(const gof_ctx* c) { return c->cov3D; }
const float* gof_oracle_v2g(const gof_ctx* c) { return c->v2g; }
const float* gof_oracle_conic_opacity(const gof_ctx* c) { return c->conic_opacity; }
const float* gof_oracle_rgb(const gof_ctx* c) { return c->rgb; }
const uint8_t* gof_oracle_clamped(const gof_ctx* c) { return c->clamped; }
const uint32_t* gof_oracle_tiles_touched(const gof_ctx* c) { return c->tiles_touched; }
const uint32_t* gof_oracle_point_offsets(const gof_ctx* c) { return c->point_offsets; }
const uint64_t* gof_oracle_keys_sorted(const gof_ctx* c) { return c->keys; }
const uint32_t* gof_oracle_point_list(const gof_ctx* c) { return c->point_list; }
const uint32_t* gof_oracle_ranges(const gof_ctx* c) { return c->ranges; }
const float* gof_oracle_final_T(const gof_ctx* c) { return c->final_T; }
const uint32_t* gof_oracle_n_contrib(const gof_ctx* c) { return c->n_contrib; }

/* forward.cu:283-404, one Gaussian */
static void preprocess_one(gof_ctx* c, int idx, int D, int M, const float* orig_points, const float* scales,
                           float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                           const float* cov3D_precomp, const float* colors_precomp, const float* v2g_precomp,
                           const float* view, const float* proj, const float* campos, int W, int H,
                           float tan_fovx, float tan_fovy, float focal_x, float focal_y, float kernel_size, int* radii)
{
    radii[idx] = 0;
    c->tiles_touched[idx] = 0;

    /* in_frustum, auxiliary.h:177-202 */
    vec3 p_orig = { orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2] };
    vec3 p_view = transformPoint4x3(p_orig, view);
    if (p_view.z <= 0.2f)
        return;

    float p_hom[4];
    transformPoint4x4(p_orig, proj, p_hom);
    float p_w = 1.0f / (p_hom[3] + 0.0000001f);
    float p_proj[3] = { p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w };

    const float* scale = scales ? scales + 3 * (size_t)idx : NULL;
    const float* rot = rotations ? rotations + 4 * (size_t)idx : NULL;

    const float* cov3D;
    if (cov3D_precomp != NULL) {
        cov3D = cov3D_precomp + (size_t)idx * 6;
    } else {
        computeCov3D(scale, scale_modifier, rot, c->cov3D + (size_t)idx * 6);
        cov3D = c->cov3D + (size_t)idx * 6;
    }

    float cov[4];
    computeCov2D(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, kernel_size, cov3D, view, cov);

    float det = (cov[0] * cov[2] - cov[1] * cov[1]);
    if (det == 0.0f)
        return;
    float det_inv = 1.f / det;
    float conic[3] = { cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv };

    float mid = 0.5f * (cov[0] + cov[2]);
    float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    float px = ndc2Pix(p_proj[0], W), py = ndc2Pix(p_proj[1], H);
    int rmin[2], rmax[2];
    getRect(px, py, (int)my_radius, c->gx, c->gy, rmin, rmax);
    if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0)
        return;

    if (colors_precomp == NULL)
        computeColorFromSH(idx, D, M, orig_points, campos, shs, c->clamped, c->rgb + 3 * (size_t)idx);

    c->depths[idx] = p_view.z;
    radii[idx] = (int)my_radius;
    c->means2D[2 * (size_t)idx] = px;
    c->means2D[2 * (size_t)idx + 1] = py;
    c->conic_opacity[4 * (size_t)idx + 0] = conic[0];
    c->conic_opacity[4 * (size_t)idx + 1] = conic[1];
    c->conic_opacity[4 * (size_t)idx + 2] = conic[2];
    c->conic_opacity[4 * (size_t)idx + 3] = opacities[idx] * cov[3];
    c->tiles_touched[idx] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));

    if (v2g_precomp == NULL)
        computeView2Gaussian(scale, p_orig, rot, view, c->v2g + (size_t)idx * 10);
}

typedef struct { uint64_t key; uint32_t val; uint32_t seq; } kv_t;
static uint64_t g_sort_mask;
static int kv_cmp(const void* a, const void* b)
{
    const kv_t* x = (const kv_t*)a; const kv_t* y = (const kv_t*)b;
    uint64_t kx = x->key & g_sort_mask, ky = y->key & g_sort_mask;
    if (kx != ky) return kx < ky ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);   /* stable: cub::DeviceRadixSort is stable */
}

/* forward.cu:409-612, one pixel; list = this tile's slice of the sorted point list */
static void render_pixel(const gof_ctx* c, uint32_t px, uint32_t py, const uint32_t* list, int count, const float* bg, float* out_color)
{
    const int W = c->W, H = c->H;
    const uint32_t pix_id = (uint32_t)W * py + px;
    const float pixf_x = (float)px + 0.5f, pixf_y = (float)py + 0.5f;
    const float ray_x = (float)((pixf_x - W / 2.) / c->focal_x);
    const float ray_y = (float)((pixf_y - H / 2.) / c->focal_y);

    float T = 1.0f;
    uint32_t contributor = 0, last_contributor = 0, max_contributor = (uint32_t)-1;
    float C[8] = { 0 };
    float dist1 = 0, dist2 = 0, distortion = 0;

    for (int j = 0; j < count; j++) {
        contributor++;
        const uint32_t id = list[j];
        const float con_o_w = c->conic_opacity[4 * (size_t)id + 3];
        const float* v = c->v2g_used + (size_t)id * 10;

        const float normal[3] = {
            v[0] * ray_x + v[1] * ray_y + v[2],
            v[1] * ray_x + v[3] * ray_y + v[4],
            v[2] * ray_x + v[4] * ray_y + v[5]
        };
        double AA = ray_x * normal[0] + ray_y * normal[1] + normal[2];
        double BB = 2 * (v[6] * ray_x + v[7] * ray_y + v[8]);
        float CC = v[9];

        float t = (float)(-BB / (2 * AA));
        if (t <= NEAR_PLANE)
            continue;

        double min_value = -(BB / AA) * (BB / 4.) + CC;
        float power = (float)(-0.5f * min_value);
        if (power > 0.0f)
            power = 0.0f;

        float alpha = fminf(0.99f, con_o_w * expf(power));
        if (alpha < 1.0f / 255.0f)
            continue;
        float test_T = T * (1 - alpha);
        if (test_T < 0.0001f)
            break;          /* done = true: this Gaussian is NOT blended, nothing later is visited */

        const float max_t = t;
        const float mapped_max_t = (float)((FAR_PLANE * max_t - FAR_PLANE * NEAR_PLANE) / ((FAR_PLANE - NEAR_PLANE) * max_t));

        float length = (float)sqrt(normal[0] * normal[0] + normal[1] * normal[1] + normal[2] * normal[2] + 1e-7);
        const float nn[3] = { -normal[0] / length, -normal[1] / length, -normal[2] / length };

        float A = 1 - T;
        float error = mapped_max_t * mapped_max_t * A + dist2 - 2 * mapped_max_t * dist1;
        distortion += error * alpha * T;
        dist1 += mapped_max_t * alpha * T;
        dist2 += mapped_max_t * mapped_max_t * alpha * T;

        for (int ch = 0; ch < 3; ch++)
            C[ch] += c->colors_used[(size_t)id * 3 + ch] * alpha * T;
        for (int ch = 0; ch < 3; ch++)
            C[3 + ch] += nn[ch] * alpha * T;
        if (T > 0.5) {
            C[6] = t;
            max_contributor = contributor;
        }
        C[7] += alpha * T;
        T = test_T;
        last_contributor = contributor;
    }

    const size_t HW = (size_t)H * W;
    const float distortion_before_normalized = distortion;
    distortion = (float)(distortion / ((1 - T) * (1 - T) + 1e-7));

    c->final_T[pix_id] = T;
    c->final_T[pix_id + HW] = dist1;
    c->final_T[pix_id + 2 * HW] = dist2;
    c->final_T[pix_id + 3 * HW] = distortion_before_normalized;
    c->n_contrib[pix_id] = last_contributor;
    c->n_contrib[pix_id + HW] = max_contributor;
    for (int ch = 0; ch < 3; ch++)
        out_color[ch * HW + pix_id] = C[ch] + T * bg[ch];
    for (int ch = 0; ch < 3; ch++)
        out_color[(3 + ch) * HW + pix_id] = C[3 + ch];
    out_color[DEPTH_OFFSET * HW + pix_id] = C[6];
    out_color[ALPHA_OFFSET * HW + pix_id] = C[7];
    out_color[DISTORTION_OFFSET * HW + pix_id] = distortion;
}

/* rasterizer_impl.cu:247-405. Pointers that the reference receives as empty tensors are NULL here.
 * view2gaussian_precomp, if given, is [P,10] and is consumed by the compositing stage exactly as the
 * reference's render does (forward.cu:402 reads it 16-strided into a dead variable; render reads it
 * 10-strided, rasterizer_impl.cu:378). Returns num_rendered. */
static int geometry_and_binning(gof_ctx* c, int P, int D, int M, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                       const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* view2gaussian_precomp, const float* viewmatrix, const float* projmatrix,
                       const float* cam_pos, float tan_fovx, float tan_fovy, float kernel_size, int* radii)
{
    ctx_free_buffers(c);
    c->P = P; c->W = width; c->H = height; c->D = D; c->M = M;
    c->focal_y = height / (2.0f * tan_fovy);
    c->focal_x = width / (2.0f * tan_fovx);
    c->gx = (width + BLOCK_X - 1) / BLOCK_X;
    c->gy = (height + BLOCK_Y - 1) / BLOCK_Y;
    const size_t HW = (size_t)width * height;
    const int ntiles = c->gx * c->gy;
    size_t Pn = P > 0 ? (size_t)P : 1;

    c->depths = calloc(Pn, 4); c->clamped = calloc(Pn * 3, 1); c->radii = calloc(Pn, 4);
    c->means2D = calloc(Pn * 2, 4); c->cov3D = calloc(Pn * 6, 4); c->v2g = calloc(Pn * 10, 4);
    c->conic_opacity = calloc(Pn * 4, 4); c->rgb = calloc(Pn * 3, 4);
    c->tiles_touched = calloc(Pn, 4); c->point_offsets = calloc(Pn, 4);
    c->final_T = calloc(HW * 4, 4); c->n_contrib = calloc(HW * 2, 4); c->ranges = calloc((size_t)ntiles * 2, 4);
    if (radii == NULL) radii = c->radii;

    #pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++)
        preprocess_one(c, idx, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, cov3D_precomp,
                       colors_precomp, view2gaussian_precomp, viewmatrix, projmatrix, cam_pos, width, height,
                       tan_fovx, tan_fovy, c->focal_x, c->focal_y, kernel_size, radii);
    if (radii != c->radii && P > 0) memcpy(c->radii, radii, (size_t)P * 4);

    /* InclusiveSum, rasterizer_impl.cu:332 */
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += c->tiles_touched[i]; c->point_offsets[i] = acc; }
    const int R = (int)acc;
    c->R = R;

    /* duplicateWithKeys, rasterizer_impl.cu:70-111 */
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (R > 0 ? (size_t)R : 1));
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : c->point_offsets[idx - 1];
            int rmin[2], rmax[2];
            getRect(c->means2D[2 * (size_t)idx], c->means2D[2 * (size_t)idx + 1], radii[idx], c->gx, c->gy, rmin, rmax);
            uint32_t dbits; memcpy(&dbits, &c->depths[idx], 4);
            for (int y = rmin[1]; y < rmax[1]; y++)
                for (int x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * c->gx + x);
                    key <<= 32;
                    key |= dbits;
                    kv[off].key = key; kv[off].val = (uint32_t)idx; kv[off].seq = off;
                    off++;
                }
        }
    }

    /* SortPairs on bits [0, 32+bit), rasterizer_impl.cu:355-363 */
    int bit = (int)getHigherMsb((uint32_t)ntiles);
    int end_bit = 32 + bit;
    g_sort_mask = end_bit >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << end_bit) - 1);
    if (R > 1) qsort(kv, (size_t)R, sizeof(kv_t), kv_cmp);
    c->keys = (uint64_t*)malloc(8 * (R > 0 ? (size_t)R : 1));
    c->point_list = (uint32_t*)malloc(4 * (R > 0 ? (size_t)R : 1));
    for (int i = 0; i < R; i++) { c->keys[i] = kv[i].key; c->point_list[i] = kv[i].val; }
    free(kv);

    /* identifyTileRanges, rasterizer_impl.cu:149-171 (ranges zeroed first, :365) */
    for (int idx = 0; idx < R; idx++) {
        uint32_t currtile = (uint32_t)(c->keys[idx] >> 32);
        if (idx == 0)
            c->ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(c->keys[idx - 1] >> 32);
            if (currtile != prevtile) {
                c->ranges[2 * prevtile + 1] = (uint32_t)idx;
                c->ranges[2 * currtile] = (uint32_t)idx;
            }
        }
        if (idx == R - 1)
            c->ranges[2 * currtile + 1] = (uint32_t)R;
    }

    c->colors_used = colors_precomp != NULL ? colors_precomp : c->rgb;
    c->v2g_used = view2gaussian_precomp != NULL ? view2gaussian_precomp : c->v2g;
    return R;
}

int gof_oracle_forward(gof_ctx* c, int P, int D, int M, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                       const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* view2gaussian_precomp, const float* viewmatrix, const float* projmatrix,
                       const float* cam_pos, float tan_fovx, float tan_fovy, float kernel_size,
                       float* out_color, int* radii)
{
    const int R = geometry_and_binning(c, P, D, M, width, height, means3D, shs, colors_precomp, opacities, scales,
                                       scale_modifier, rotations, cov3D_precomp, view2gaussian_precomp, viewmatrix,
                                       projmatrix, cam_pos, tan_fovx, tan_fovy, kernel_size, radii);
    const int ntiles = c->gx * c->gy;

    /* renderCUDA: tile-parallel on the host cores */
    #pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < ntiles; tile++) {
        const int tx = tile % c->gx, ty = tile / c->gx;
        const uint32_t r0 = c->ranges[2 * tile], r1 = c->ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                uint32_t px = (uint32_t)(tx * BLOCK_X + lx), py = (uint32_t)(ty * BLOCK_Y + ly);
                if (px < (uint32_t)width && py < (uint32_t)height)
                    render_pixel(c, px, py, c->point_list + r0, (int)(r1 - r0), background, out_color);
            }
    }
    return R;
}

/* ========================================================================================== */
/*                       INTEGRATE (Gaussians -> points), SURVEY.md 8f-1                      */
/* ========================================================================================== */
/* Literal restatement of Rasterizer::integrate (rasterizer_impl.cu:530-792): preprocessPointsCUDA
 * (forward.cu:722-766), createWithKeys (rasterizer_impl.cu:113-144) and integrateCUDA (forward.cu:801-1197),
 * INCLUDING the per-thread arrays, the 256-point sweeps and the block-wide votes of the CUDA kernel: the 256
 * threads of a tile are simulated one after the other between two votes. */
#define MAX_NUM_CONTRIBUTORS 256   /* auxiliary.h:26 */
#define MAX_NUM_PROJECTED 256      /* auxiliary.h:34 */
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)

typedef struct {
    int inside;
    uint32_t px, py;
    float corner_Ts[5];
    float C[8];
    uint32_t last_contributor;
    uint32_t n_contrib_local;
    uint16_t contributed_ids[MAX_NUM_CONTRIBUTORS * 4];
    /* point phase */
    uint32_t point_counter_last;
    int point_done;
    int total_projected;
} integ_thread;

/* first loop of integrateCUDA (forward.cu:871-980) for one thread */
static void integrate_pass1(const gof_ctx* c, integ_thread* th, const uint32_t* list, int count)
{
    const int W = c->W, H = c->H;
    const float pixf_x = (float)th->px + 0.5f, pixf_y = (float)th->py + 0.5f;
    static const float offset_xs[5] = { 0.0f, -0.5f, 0.5f, -0.5f, 0.5f };
    static const float offset_ys[5] = { 0.0f, -0.5f, -0.5f, 0.5f, 0.5f };
    uint32_t contributor = 0;
    int done = !th->inside;

    for (int j = 0; !done && j < count; j++) {
        contributor++;
        const uint32_t id = list[j];
        const float con_o_w = c->conic_opacity[4 * (size_t)id + 3];
        const float* v = c->v2g_used + (size_t)id * 10;

        int used = 0;
        for (int k = 0; k < 5; ++k) {
            const float rx = (float)((pixf_x + offset_xs[k] - W / 2.) / c->focal_x);
            const float ry = (float)((pixf_y + offset_ys[k] - H / 2.) / c->focal_y);
            const float normal[3] = {
                v[0] * rx + v[1] * ry + v[2],
                v[1] * rx + v[3] * ry + v[4],
                v[2] * rx + v[4] * ry + v[5]
            };
            float AA = rx * normal[0] + ry * normal[1] + normal[2];
            float BB = 2 * (v[6] * rx + v[7] * ry + v[8]);
            float CC = v[9];

            float t = -BB / (2 * AA);
            if (t <= NEAR_PLANE)
                continue;

            double min_value = -(BB / AA) * (BB / 4.) + CC;     /* float quotient, then double */
            float power = (float)(-0.5f * min_value);
            if (power > 0.0f)
                power = 0.0f;

            float alpha = fminf(0.99f, con_o_w * expf(power));
            if (alpha < 1.0f / 255.0f)
                continue;
            float test_T = th->corner_Ts[k] * (1 - alpha);
            if (test_T < 0.0001f)
                continue;               /* NOT done: forward.cu:934-938 */

            if (k == 0)
                for (int ch = 0; ch < 3; ch++)
                    th->C[ch] += c->colors_used[(size_t)id * 3 + ch] * alpha * th->corner_Ts[k];
            if (t > th->C[6])
                th->C[6] = t;           /* maximal depth */
            if (k == 0)
                th->C[7] += alpha * th->corner_Ts[k];
            th->corner_Ts[k] = test_T;
            used = 1;
        }

        if (used) {
            th->last_contributor = contributor;
            th->contributed_ids[th->n_contrib_local] = (uint16_t)contributor;
            th->n_contrib_local += 1;
            if (th->n_contrib_local >= MAX_NUM_CONTRIBUTORS * 4) {
                done = 1;               /* "Maximal contributors are met" */
                break;
            }
        }
    }
}

/* points state (rasterizer_impl.cu:47-57) */
typedef struct { float* depths; float* points2D; uint32_t* tiles_touched; uint32_t* point_list; uint32_t* ranges; int num_integrated; } point_state;

/* Returns num_rendered. out_color [9,H,W] must be zero-filled by the caller exactly as rasterize_points.cu:273 does
 * (channels 3..5 are never written); out_alpha_integrated [PN] pre-filled with 1, out_color_integrated [PN,3] with 0
 * (rasterize_points.cu:275-276). */
int gof_oracle_integrate(gof_ctx* c, int PN, int P, int D, int M, const float* background, int width, int height,
                         const float* points3D, const float* means3D, const float* shs, const float* colors_precomp,
                         const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                         const float* cov3D_precomp, const float* view2gaussian_precomp, const float* viewmatrix,
                         const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                         float kernel_size, float* out_color, int* radii, float* out_alpha_integrated,
                         float* out_color_integrated, int* num_integrated_out)
{
    const int R = geometry_and_binning(c, P, D, M, width, height, means3D, shs, colors_precomp, opacities, scales,
                                       scale_modifier, rotations, cov3D_precomp, view2gaussian_precomp, viewmatrix,
                                       projmatrix, cam_pos, tan_fovx, tan_fovy, kernel_size, radii);
    const int W = width, H = height;
    const int ntiles = c->gx * c->gy;
    const size_t HW = (size_t)H * W;

    /* preprocessPointsCUDA, forward.cu:722-766 */
    point_state ps;
    size_t PNn = PN > 0 ? (size_t)PN : 1;
    ps.depths = calloc(PNn, 4); ps.points2D = calloc(PNn * 2, 4); ps.tiles_touched = calloc(PNn, 4);
    for (int idx = 0; idx < PN; idx++) {
        ps.tiles_touched[idx] = 0;
        vec3 p_orig = { points3D[3 * idx], points3D[3 * idx + 1], points3D[3 * idx + 2] };
        vec3 p_view = transformPoint4x3(p_orig, viewmatrix);
        if (p_view.z <= 0.2f)
            continue;
        const float ix = (float)(c->focal_x * p_view.x / (p_view.z + 0.0000001f) + W / 2.);
        const float iy = (float)(c->focal_y * p_view.y / (p_view.z + 0.0000001f) + H / 2.);
        if (ix < 0 || ix >= W || iy < 0 || iy >= H)
            continue;
        ps.depths[idx] = p_view.z;
        ps.points2D[2 * (size_t)idx] = ix; ps.points2D[2 * (size_t)idx + 1] = iy;
        ps.tiles_touched[idx] = 1;
    }
    /* InclusiveSum + createWithKeys (rasterizer_impl.cu:113-144) + SortPairs + identifyTileRanges */
    int NI = 0;
    for (int idx = 0; idx < PN; idx++) NI += (int)ps.tiles_touched[idx];
    ps.num_integrated = NI;
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (NI > 0 ? (size_t)NI : 1));
    {
        uint32_t off = 0;
        for (int idx = 0; idx < PN; idx++)
            if (ps.tiles_touched[idx] > 0) {
                const float pxf = ps.points2D[2 * (size_t)idx], pyf = ps.points2D[2 * (size_t)idx + 1];
                int x = imin(c->gx - 1, imax(0, (int)(pxf / BLOCK_X)));
                int y = imin(c->gy - 1, imax(0, (int)(pyf / BLOCK_Y)));
                uint64_t key = (uint64_t)(y * c->gx + x);
                key <<= 32;
                uint32_t dbits; memcpy(&dbits, &ps.depths[idx], 4);
                key |= dbits;
                kv[off].key = key; kv[off].val = (uint32_t)idx; kv[off].seq = off;
                off++;
            }
    }
    int bit = (int)getHigherMsb((uint32_t)ntiles);
    int end_bit = 32 + bit;
    g_sort_mask = end_bit >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << end_bit) - 1);
    if (NI > 1) qsort(kv, (size_t)NI, sizeof(kv_t), kv_cmp);
    ps.point_list = (uint32_t*)malloc(4 * (NI > 0 ? (size_t)NI : 1));
    ps.ranges = (uint32_t*)calloc((size_t)ntiles * 2, 4);
    for (int i = 0; i < NI; i++) ps.point_list[i] = kv[i].val;
    for (int idx = 0; idx < NI; idx++) {
        uint32_t currtile = (uint32_t)(kv[idx].key >> 32);
        if (idx == 0)
            ps.ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(kv[idx - 1].key >> 32);
            if (currtile != prevtile) {
                ps.ranges[2 * prevtile + 1] = (uint32_t)idx;
                ps.ranges[2 * currtile] = (uint32_t)idx;
            }
        }
        if (idx == NI - 1)
            ps.ranges[2 * currtile + 1] = (uint32_t)NI;
    }
    free(kv);
    if (num_integrated_out) *num_integrated_out = NI;

    /* integrateCUDA, one tile = one block of 256 simulated threads */
    #pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < ntiles; tile++) {
        const int tx = tile % c->gx, ty = tile / c->gx;
        const uint32_t r0 = c->ranges[2 * tile], r1 = c->ranges[2 * tile + 1];
        const uint32_t* list = c->point_list + r0;
        const int count = (int)(r1 - r0);
        const uint32_t p0 = ps.ranges[2 * tile], p1 = ps.ranges[2 * tile + 1];
        const int pcount = (int)(p1 - p0);
        integ_thread* th = (integ_thread*)calloc(BLOCK_SIZE, sizeof(integ_thread));

        for (int t = 0; t < BLOCK_SIZE; t++) {
            integ_thread* h = &th[t];
            h->px = (uint32_t)(tx * BLOCK_X + t % BLOCK_X); h->py = (uint32_t)(ty * BLOCK_Y + t / BLOCK_X);
            h->inside = h->px < (uint32_t)W && h->py < (uint32_t)H;
            for (int k = 0; k < 5; k++) h->corner_Ts[k] = 1.0f;
            integrate_pass1(c, h, list, count);
            if (h->inside) {                                  /* forward.cu:984-996 */
                const uint32_t pix_id = (uint32_t)W * h->py + h->px;
                c->final_T[pix_id] = h->corner_Ts[0];
                c->n_contrib[pix_id] = h->last_contributor;
                for (int ch = 0; ch < 3; ch++)
                    out_color[ch * HW + pix_id] = h->C[ch] + h->corner_Ts[0] * background[ch];
                out_color[DEPTH_OFFSET * HW + pix_id] = h->C[6];
                out_color[ALPHA_OFFSET * HW + pix_id] = h->C[7];
            }
            h->point_counter_last = 0;
            h->point_done = !h->inside;
            h->total_projected = 0;
        }

        /* forward.cu:1011-1189 */
        while (1) {
            int num_done = 0;
            for (int t = 0; t < BLOCK_SIZE; t++) num_done += th[t].point_done;
            if (num_done == BLOCK_SIZE)
                break;

            for (int t = 0; t < BLOCK_SIZE; t++) {
                integ_thread* h = &th[t];
                const float pixf_x = (float)h->px + 0.5f, pixf_y = (float)h->py + 0.5f;
                int projected_ids[MAX_NUM_PROJECTED];
                float projected_xy[MAX_NUM_PROJECTED][2];
                float projected_depth[MAX_NUM_PROJECTED];
                int num_projected = 0;
                int excced_max_projected = 0;
                int done = 0;
                uint32_t point_counter = 0;

                /* the block-wide "all done" vote inside the rounds loop (forward.cu:1034-1037) can only stop the loop
                 * once every thread has stopped iterating, so it changes no thread's state */
                for (int j = 0; !done && j < pcount; j++) {
                    point_counter++;
                    if (point_counter <= h->point_counter_last)
                        continue;
                    const uint32_t pid = ps.point_list[p0 + j];
                    const float qx = ps.points2D[2 * (size_t)pid], qy = ps.points2D[2 * (size_t)pid + 1];
                    const float depth = ps.depths[pid];
                    if ((qx >= (pixf_x - 0.5)) && (qx < (pixf_x + 0.5)) && (qy >= (pixf_y - 0.5)) && (qy < (pixf_y + 0.5))) {
                        if (num_projected >= MAX_NUM_PROJECTED) {
                            done = 1;
                            excced_max_projected = 1;
                            break;
                        }
                        projected_ids[num_projected] = (int)pid;
                        projected_xy[num_projected][0] = qx; projected_xy[num_projected][1] = qy;
                        projected_depth[num_projected] = depth;
                        num_projected += 1;
                    }
                }
                h->point_counter_last = point_counter - 1;
                h->point_done = !excced_max_projected;
                h->total_projected += num_projected;

                float point_alphas[MAX_NUM_PROJECTED];
                float point_Ts[MAX_NUM_PROJECTED];
                for (int i = 0; i < MAX_NUM_PROJECTED; i++) { point_alphas[i] = 0.f; point_Ts[i] = 0.f; }
                for (int i = 0; i < num_projected; i++) point_Ts[i] = 1.f;

                uint32_t num_iterated = 0;
                int second_done = !h->inside;
                uint16_t num_contributed_second = 0;
                for (int j = 0; !second_done && j < count; j++) {
                    num_iterated++;
                    if (num_iterated > h->last_contributor) {
                        second_done = 1;
                        continue;
                    }
                    if (num_iterated != (uint32_t)h->contributed_ids[num_contributed_second])
                        continue;
                    else
                        num_contributed_second += 1;

                    const uint32_t id = list[j];
                    const float con_o_w = c->conic_opacity[4 * (size_t)id + 3];
                    const float* v = c->v2g_used + (size_t)id * 10;
                    for (int k = 0; k < num_projected; k++) {
                        const float rx = (float)((projected_xy[k][0] - W / 2.) / c->focal_x);
                        const float ry = (float)((projected_xy[k][1] - H / 2.) / c->focal_y);
                        const float ray_depth = projected_depth[k];
                        const float normal[3] = {
                            v[0] * rx + v[1] * ry + v[2],
                            v[1] * rx + v[3] * ry + v[4],
                            v[2] * rx + v[4] * ry + v[5]
                        };
                        float AA = rx * normal[0] + ry * normal[1] + normal[2];
                        float BB = 2 * (v[6] * rx + v[7] * ry + v[8]);
                        float CC = v[9];
                        float t = -BB / (2 * AA);
                        if (t > ray_depth)
                            t = ray_depth;
                        float power = -0.5f * (AA * t * t + BB * t + CC);
                        float alpha = fminf(0.99f, con_o_w * expf(power));
                        if (alpha < 1.0f / 255.0f)
                            continue;
                        float test_T = point_Ts[k] * (1 - alpha);
                        point_alphas[k] += alpha * point_Ts[k];
                        point_Ts[k] = test_T;
                    }
                }

                if (h->inside)
                    for (int k = 0; k < num_projected; k++) {
                        out_alpha_integrated[projected_ids[k]] = point_alphas[k];
                        for (int ch = 0; ch < 3; ch++)
                            out_color_integrated[3 * (size_t)projected_ids[k] + ch] = h->C[ch] + h->corner_Ts[0] * background[ch];
                    }
            }
        }

        for (int t = 0; t < BLOCK_SIZE; t++)
            if (th[t].inside)
                out_color[DISTORTION_OFFSET * HW + (size_t)W * th[t].py + th[t].px] = (float)th[t].total_projected;
        free(th);
    }
    free(ps.depths); free(ps.points2D); free(ps.tiles_touched); free(ps.point_list); free(ps.ranges);
    return R;
}

/* exported for tests/test_oracle_pins.py: SH colour of ONE Gaussian (compared against the reference's python
 * eval_sh, src/gaussian-splatting/utils/sh_utils.py:57-116) */
void gof_oracle_color_from_sh(int deg, int max_coeffs, const float* mean, const float* campos, const float* sh,
                              float* rgb_out, uint8_t* clamped_out)
{
    computeColorFromSH(0, deg, max_coeffs, mean, campos, sh, clamped_out, rgb_out);
}

/* ========================================================================================== */
/*                                        BACKWARD                                            */
/* ========================================================================================== */

/* auxiliary.h:145-155 */
static void dnormvdv3(const float v[3], const float dv[3], float out[3])
{
    float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    out[0] = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
    out[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
    out[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

/* backward.cu:634-955, one pixel. Per-Gaussian sums are accumulated in DOUBLE here (the reference uses float
 * atomicAdd in a nondeterministic order; a double sum is the order-free value every such execution rounds around). */
static void render_pixel_backward(const gof_ctx* c, uint32_t px, uint32_t py, const uint32_t* list, int count,
                                  const float* bg, const float* dL_dpixels,
                                  double* dL_dmean2D, double* dL_dopacity, double* dL_dcolors, double* dL_dv2g)
{
    const int W = c->W, H = c->H;
    const size_t HW = (size_t)H * W;
    const uint32_t pix_id = (uint32_t)W * py + px;
    const float pixf_x = (float)px + 0.5f, pixf_y = (float)py + 0.5f;
    const float ray_x = (float)((pixf_x - W / 2.) / c->focal_x);
    const float ray_y = (float)((pixf_y - H / 2.) / c->focal_y);

    const float T_final = c->final_T[pix_id];
    float T = T_final;
    const float final_D = c->final_T[pix_id + HW];
    const float final_A = 1 - T_final;
    const float dL_dreg = dL_dpixels[DISTORTION_OFFSET * HW + pix_id];

    float last_dL_dT = 0;
    uint32_t contributor = (uint32_t)count;
    const int last_contributor = (int)c->n_contrib[pix_id];
    const int max_contributor = (int)c->n_contrib[pix_id + HW];
    float accum_rec[3] = { 0 };
    float dL_dpixel[3], dL_dnormal2D[3];
    for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * HW + pix_id];
    for (int i = 0; i < 3; i++) dL_dnormal2D[i] = dL_dpixels[(3 + i) * HW + pix_id];
    const float dL_dmax_depth = dL_dpixels[DEPTH_OFFSET * HW + pix_id];

    float last_alpha = 0;
    float last_color[3] = { 0 };
    float last_normal[3] = { 0 };
    float accum_normal_rec[3] = { 0 };

    const float ddelx_dx = (float)(0.5 * W);
    const float ddely_dy = (float)(0.5 * H);

    for (int j = 0; j < count; j++) {
        contributor--;
        if (contributor >= (uint32_t)last_contributor)
            continue;
        const uint32_t id = list[count - 1 - j];

        const float xy_x = c->means2D[2 * (size_t)id], xy_y = c->means2D[2 * (size_t)id + 1];
        const float d_x = (float)(xy_x - (pixf_x - 0.5)), d_y = (float)(xy_y - (pixf_y - 0.5));
        const float* con_o = c->conic_opacity + 4 * (size_t)id;
        const float* v = c->v2g_used + (size_t)id * 10;

        const float normal[3] = {
            v[0] * ray_x + v[1] * ray_y + v[2],
            v[1] * ray_x + v[3] * ray_y + v[4],
            v[2] * ray_x + v[4] * ray_y + v[5]
        };
        double AA = ray_x * normal[0] + ray_y * normal[1] + normal[2];
        double BB = 2 * (v[6] * ray_x + v[7] * ray_y + v[8]);
        float CC = v[9];

        float t = (float)(-BB / (2 * AA));
        if (t <= NEAR_PLANE)
            continue;
        double min_value = -(BB / AA) * (BB / 4.) + CC;
        float power = (float)(-0.5f * min_value);
        if (power > 0.0f)
            power = 0.0f;

        const float G = expf(power);
        const float alpha = fminf(0.99f, con_o[3] * G);
        if (alpha < 1.0f / 255.0f)
            continue;

        const float max_t = t;
        const float mapped_max_t = (float)((FAR_PLANE * max_t - FAR_PLANE * NEAR_PLANE) / ((FAR_PLANE - NEAR_PLANE) * max_t));
        float dmax_t_dd = (float)((FAR_PLANE * NEAR_PLANE) / ((FAR_PLANE - NEAR_PLANE) * max_t * max_t));

        float length = (float)sqrt(normal[0] * normal[0] + normal[1] * normal[1] + normal[2] * normal[2] + 1e-7);
        const float nn[3] = { -normal[0] / length, -normal[1] / length, -normal[2] / length };

        T = T / (1.f - alpha);
        const float dchannel_dcolor = alpha * T;

        float dL_dalpha = 0.0f;
        for (int ch = 0; ch < 3; ch++) {
            const float col = c->colors_used[(size_t)id * 3 + ch];
            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
            last_color[ch] = col;
            const float dL_dchannel = dL_dpixel[ch];
            dL_dalpha += (col - accum_rec[ch]) * dL_dchannel;
            dL_dcolors[(size_t)id * 3 + ch] += (double)(dchannel_dcolor * dL_dchannel);
        }

        float dL_dt = 0.0f;
        float dL_dmax_t = 0.0f;
        float dL_dweight = 0.0f;
        dL_dmax_t += 2.0f * (T * alpha) * (mapped_max_t * final_A - final_D) * dL_dreg * dmax_t_dd;
        /* the weight gradient of the distortion term is explicitly detached: backward.cu:850-852 */
        dL_dweight = 0.f;
        dL_dalpha += dL_dweight - last_dL_dT;
        last_dL_dT = dL_dweight * alpha + (1 - alpha) * last_dL_dT;

        float dL_dnn[3] = { 0 };
        for (int ch = 0; ch < 3; ch++) {
            accum_normal_rec[ch] = last_alpha * last_normal[ch] + (1.f - last_alpha) * accum_normal_rec[ch];
            last_normal[ch] = nn[ch];
            dL_dalpha += (nn[ch] - accum_normal_rec[ch]) * dL_dnormal2D[ch];
            dL_dnn[ch] = alpha * T * dL_dnormal2D[ch];
        }
        float dL_dlength = (dL_dnn[0] * normal[0] + dL_dnn[1] * normal[1] + dL_dnn[2] * normal[2]);
        dL_dlength *= 1.f / (length * length);
        float dL_dnormal[3] = {
            (-dL_dnn[0] + dL_dlength * normal[0]) / length,
            (-dL_dnn[1] + dL_dlength * normal[1]) / length,
            (-dL_dnn[2] + dL_dlength * normal[2]) / length
        };

        dL_dt = dL_dmax_t;
        if ((int)contributor == max_contributor - 1)
            dL_dt += dL_dmax_depth;

        dL_dalpha *= T;
        last_alpha = alpha;

        float bg_dot_dpixel = 0;
        for (int i = 0; i < 3; i++)
            bg_dot_dpixel += bg[i] * dL_dpixel[i];
        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

        const float dL_dG = con_o[3] * dL_dalpha;
        const float gdx = G * d_x;
        const float gdy = G * d_y;
        const float dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
        const float dG_ddely = -gdy * con_o[2] - gdx * con_o[1];

        dL_dmean2D[(size_t)id * 3 + 0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
        dL_dmean2D[(size_t)id * 3 + 1] += (double)(dL_dG * dG_ddely * ddely_dy);
        const float abs_dL_dmean2D = fabsf(dL_dG * dG_ddelx * ddelx_dx) + fabsf(dL_dG * dG_ddely * ddely_dy);
        dL_dmean2D[(size_t)id * 3 + 2] += (double)abs_dL_dmean2D;

        dL_dopacity[id] += (double)(G * dL_dalpha);

        const float dG_dpower = G;
        const float dL_dpower = dL_dG * dG_dpower;
        const float dL_dmin_value = dL_dpower * -0.5f;
        double dL_dA = dL_dmin_value * (BB / AA) * (BB / AA) / 4.f;
        double dL_dB = dL_dmin_value * -BB / (2 * AA);
        double dL_dC = dL_dmin_value * 1.0f;

        dL_dA += dL_dt * BB / (2 * AA * AA);
        dL_dB += dL_dt * -1.f / (2 * AA);

        dL_dnormal[0] += dL_dA * ray_x;
        dL_dnormal[1] += dL_dA * ray_y;
        dL_dnormal[2] += dL_dA;

        double* g = dL_dv2g + (size_t)id * 10;
        g[0] += (double)(float)(dL_dnormal[0] * ray_x);
        g[1] += (double)(float)(dL_dnormal[0] * ray_y + dL_dnormal[1] * ray_x);
        g[2] += (double)(float)(dL_dnormal[0] + dL_dnormal[2] * ray_x);
        g[3] += (double)(float)(dL_dnormal[1] * ray_y);
        g[4] += (double)(float)(dL_dnormal[1] + dL_dnormal[2] * ray_y);
        g[5] += (double)(float)(dL_dnormal[2]);
        g[6] += (double)(float)(dL_dB * 2 * ray_x);
        g[7] += (double)(float)(dL_dB * 2 * ray_y);
        g[8] += (double)(float)(dL_dB * 2);
        g[9] += (double)(float)(dL_dC);
    }
}

/* backward.cu:381-587: gradients of the 10 view2gaussian entries w.r.t. mean, scale and (un-normalised) quaternion */
static void computeView2Gaussian_backward(const float* scale, const float* mean, const float* rot, const float* view,
                                          const float* dL_dv2g, float* dL_dmean, float* dL_dscale, float* dL_drot)
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 R = quat_to_R(rot);
    mat4 G2W;
    G2W.c[0][0] = R.c[0][0]; G2W.c[0][1] = R.c[1][0]; G2W.c[0][2] = R.c[2][0]; G2W.c[0][3] = 0.0f;
    G2W.c[1][0] = R.c[0][1]; G2W.c[1][1] = R.c[1][1]; G2W.c[1][2] = R.c[2][1]; G2W.c[1][3] = 0.0f;
    G2W.c[2][0] = R.c[0][2]; G2W.c[2][1] = R.c[1][2]; G2W.c[2][2] = R.c[2][2]; G2W.c[2][3] = 0.0f;
    G2W.c[3][0] = mean[0];   G2W.c[3][1] = mean[1];   G2W.c[3][2] = mean[2];   G2W.c[3][3] = 1.0f;
    mat4 W2V;
    for (int cc = 0; cc < 4; cc++)
        for (int q = 0; q < 4; q++)
            W2V.c[cc][q] = view[4 * cc + q];
    mat4 G2V = m4_mul(W2V, G2W);

    mat3 Rt;
    Rt.c[0][0] = G2V.c[0][0]; Rt.c[0][1] = G2V.c[1][0]; Rt.c[0][2] = G2V.c[2][0];
    Rt.c[1][0] = G2V.c[0][1]; Rt.c[1][1] = G2V.c[1][1]; Rt.c[1][2] = G2V.c[2][1];
    Rt.c[2][0] = G2V.c[0][2]; Rt.c[2][1] = G2V.c[1][2]; Rt.c[2][2] = G2V.c[2][2];
    vec3 t = { G2V.c[3][0], G2V.c[3][1], G2V.c[3][2] };
    mat3 negRt;
    for (int cc = 0; cc < 3; cc++)
        for (int q = 0; q < 3; q++)
            negRt.c[cc][q] = -Rt.c[cc][q];
    vec3 t2 = m3_mul_v(negRt, t);

    double S[3] = { 1.0f / ((double)scale[0] * scale[0] + 1e-7), 1.0f / ((double)scale[1] * scale[1] + 1e-7),
                    1.0f / ((double)scale[2] * scale[2] + 1e-7) };
    mat3 SR;
    for (int cc = 0; cc < 3; cc++)
        for (int q = 0; q < 3; q++)
            SR.c[cc][q] = (float)(S[q] * Rt.c[cc][q]);

    mat3 dL_dSigma;
    dL_dSigma.c[0][0] = dL_dv2g[0];        dL_dSigma.c[0][1] = 0.5f * dL_dv2g[1]; dL_dSigma.c[0][2] = 0.5f * dL_dv2g[2];
    dL_dSigma.c[1][0] = 0.5f * dL_dv2g[1]; dL_dSigma.c[1][1] = dL_dv2g[3];        dL_dSigma.c[1][2] = 0.5f * dL_dv2g[4];
    dL_dSigma.c[2][0] = 0.5f * dL_dv2g[2]; dL_dSigma.c[2][1] = 0.5f * dL_dv2g[4]; dL_dSigma.c[2][2] = dL_dv2g[5];
    const float dB[3] = { dL_dv2g[6], dL_dv2g[7], dL_dv2g[8] };
    const float dL_dC = dL_dv2g[9];
    const float t2v[3] = { t2.x, t2.y, t2.z };

    /* dL_dS_inv_square_R = R_transpose * dL_dSigma + outerProduct(t2, dL_dB); outerProduct(c, r)[i][j] = c[j] * r[i] */
    mat3 D = m3_mul(Rt, dL_dSigma);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            D.c[i][j] = D.c[i][j] + t2v[j] * dB[i];
    mat3 dL_dRt = m3_transpose(m3_mul(dL_dSigma, m3_transpose(SR)));
    for (int cc = 0; cc < 3; cc++)
        for (int q = 0; q < 3; q++)
            dL_dRt.c[cc][q] = dL_dRt.c[cc][q] + (float)(S[q] * D.c[cc][q]);

    float dL_dS[3];
    for (int q = 0; q < 3; q++)
        dL_dS[q] = D.c[0][q] * Rt.c[0][q] + D.c[1][q] * Rt.c[1][q] + D.c[2][q] * Rt.c[2][q];
    float dL_dt2[3];
    for (int q = 0; q < 3; q++)
        dL_dt2[q] = (float)(2 * t2v[q] * S[q] * dL_dC + dB[0] * SR.c[0][q] + dB[1] * SR.c[1][q] + dB[2] * SR.c[2][q]);
    for (int q = 0; q < 3; q++)
        dL_dS[q] += dL_dC * t2v[q] * t2v[q];
    for (int q = 0; q < 3; q++)
        dL_dscale[q] = (float)(-2 / scale[q] * S[q] * dL_dS[q]);

    /* G2V_R_t == Rt, G2V_t == t */
    mat3 dL_dV2G_R_t = m3_transpose(dL_dRt);
    mat3 from_t;
    for (int cc = 0; cc < 3; cc++) {
        from_t.c[cc][0] = -dL_dt2[cc] * t.x;
        from_t.c[cc][1] = -dL_dt2[cc] * t.y;
        from_t.c[cc][2] = -dL_dt2[cc] * t.z;
    }
    mat3 dL_dG2V_R;
    for (int cc = 0; cc < 3; cc++)
        for (int q = 0; q < 3; q++)
            dL_dG2V_R.c[cc][q] = dL_dV2G_R_t.c[cc][q] + from_t.c[cc][q];
    vec3 ndt = { -dL_dt2[0], -dL_dt2[1], -dL_dt2[2] };
    vec3 dL_dG2V_t = v_mul_m3(ndt, Rt);

    mat4 dL_dG2V;
    for (int cc = 0; cc < 3; cc++) {
        dL_dG2V.c[cc][0] = dL_dG2V_R.c[cc][0]; dL_dG2V.c[cc][1] = dL_dG2V_R.c[cc][1];
        dL_dG2V.c[cc][2] = dL_dG2V_R.c[cc][2]; dL_dG2V.c[cc][3] = 0.0f;
    }
    dL_dG2V.c[3][0] = dL_dG2V_t.x; dL_dG2V.c[3][1] = dL_dG2V_t.y; dL_dG2V.c[3][2] = dL_dG2V_t.z; dL_dG2V.c[3][3] = 0.0f;
    mat4 W2Vt;
    for (int cc = 0; cc < 4; cc++)
        for (int q = 0; q < 4; q++)
            W2Vt.c[cc][q] = W2V.c[q][cc];
    mat4 dL_dG2W = m4_mul(W2Vt, dL_dG2V);

    dL_dmean[0] = dL_dG2W.c[3][0];
    dL_dmean[1] = dL_dG2W.c[3][1];
    dL_dmean[2] = dL_dG2W.c[3][2];

    float Mt[3][3];
    for (int cc = 0; cc < 3; cc++)
        for (int q = 0; q < 3; q++)
            Mt[cc][q] = dL_dG2W.c[cc][q];
    dL_drot[0] = 2 * z * (Mt[0][1] - Mt[1][0]) + 2 * y * (Mt[2][0] - Mt[0][2]) + 2 * x * (Mt[1][2] - Mt[2][1]);
    dL_drot[1] = 2 * y * (Mt[1][0] + Mt[0][1]) + 2 * z * (Mt[2][0] + Mt[0][2]) + 2 * r * (Mt[1][2] - Mt[2][1]) - 4 * x * (Mt[2][2] + Mt[1][1]);
    dL_drot[2] = 2 * x * (Mt[1][0] + Mt[0][1]) + 2 * r * (Mt[2][0] - Mt[0][2]) + 2 * z * (Mt[1][2] + Mt[2][1]) - 4 * y * (Mt[2][2] + Mt[0][0]);
    dL_drot[3] = 2 * r * (Mt[0][1] - Mt[1][0]) + 2 * x * (Mt[2][0] + Mt[0][2]) + 2 * y * (Mt[1][2] + Mt[2][1]) - 4 * z * (Mt[1][1] + Mt[0][0]);
}

/* backward.cu:20-139 */
static void computeColorFromSH_backward(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                                        const float* shs, const uint8_t* clamped, const float* dL_dcolor,
                                        float* dL_dmeans, float* dL_dshs)
{
    const float dir_orig[3] = { means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2] };
    float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
    const float x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    float dL_dRGB[3];
    for (int ch = 0; ch < 3; ch++)
        dL_dRGB[ch] = dL_dcolor[3 * (size_t)idx + ch] * (clamped[3 * idx + ch] ? 0 : 1);
    float dRGBdx[3] = { 0, 0, 0 }, dRGBdy[3] = { 0, 0, 0 }, dRGBdz[3] = { 0, 0, 0 };
    float* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
#define SH(k, ch) sh[(k) * 3 + (ch)]
#define DSH(k, val) do { float _w = (val); for (int ch = 0; ch < 3; ch++) dL_dsh[(k) * 3 + ch] = _w * dL_dRGB[ch]; } while (0)
    DSH(0, SH_C0);
    if (deg > 0) {
        DSH(1, -SH_C1 * y);
        DSH(2, SH_C1 * z);
        DSH(3, -SH_C1 * x);
        for (int ch = 0; ch < 3; ch++) {
            dRGBdx[ch] = -SH_C1 * SH(3, ch);
            dRGBdy[ch] = -SH_C1 * SH(1, ch);
            dRGBdz[ch] = SH_C1 * SH(2, ch);
        }
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            DSH(4, SH_C2[0] * xy);
            DSH(5, SH_C2[1] * yz);
            DSH(6, SH_C2[2] * (2.f * zz - xx - yy));
            DSH(7, SH_C2[3] * xz);
            DSH(8, SH_C2[4] * (xx - yy));
            for (int ch = 0; ch < 3; ch++) {
                dRGBdx[ch] += SH_C2[0] * y * SH(4, ch) + SH_C2[2] * 2.f * -x * SH(6, ch) + SH_C2[3] * z * SH(7, ch) + SH_C2[4] * 2.f * x * SH(8, ch);
                dRGBdy[ch] += SH_C2[0] * x * SH(4, ch) + SH_C2[1] * z * SH(5, ch) + SH_C2[2] * 2.f * -y * SH(6, ch) + SH_C2[4] * 2.f * -y * SH(8, ch);
                dRGBdz[ch] += SH_C2[1] * y * SH(5, ch) + SH_C2[2] * 2.f * 2.f * z * SH(6, ch) + SH_C2[3] * x * SH(7, ch);
            }
            if (deg > 2) {
                DSH(9, SH_C3[0] * y * (3.f * xx - yy));
                DSH(10, SH_C3[1] * xy * z);
                DSH(11, SH_C3[2] * y * (4.f * zz - xx - yy));
                DSH(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                DSH(13, SH_C3[4] * x * (4.f * zz - xx - yy));
                DSH(14, SH_C3[5] * z * (xx - yy));
                DSH(15, SH_C3[6] * x * (xx - 3.f * yy));
                for (int ch = 0; ch < 3; ch++) {
                    dRGBdx[ch] += (
                        SH_C3[0] * SH(9, ch) * 3.f * 2.f * xy +
                        SH_C3[1] * SH(10, ch) * yz +
                        SH_C3[2] * SH(11, ch) * -2.f * xy +
                        SH_C3[3] * SH(12, ch) * -3.f * 2.f * xz +
                        SH_C3[4] * SH(13, ch) * (-3.f * xx + 4.f * zz - yy) +
                        SH_C3[5] * SH(14, ch) * 2.f * xz +
                        SH_C3[6] * SH(15, ch) * 3.f * (xx - yy));
                    dRGBdy[ch] += (
                        SH_C3[0] * SH(9, ch) * 3.f * (xx - yy) +
                        SH_C3[1] * SH(10, ch) * xz +
                        SH_C3[2] * SH(11, ch) * (-3.f * yy + 4.f * zz - xx) +
                        SH_C3[3] * SH(12, ch) * -3.f * 2.f * yz +
                        SH_C3[4] * SH(13, ch) * -2.f * xy +
                        SH_C3[5] * SH(14, ch) * -2.f * yz +
                        SH_C3[6] * SH(15, ch) * -3.f * 2.f * xy);
                    dRGBdz[ch] += (
                        SH_C3[1] * SH(10, ch) * xy +
                        SH_C3[2] * SH(11, ch) * 4.f * 2.f * yz +
                        SH_C3[3] * SH(12, ch) * 3.f * (2.f * zz - xx - yy) +
                        SH_C3[4] * SH(13, ch) * 4.f * 2.f * xz +
                        SH_C3[5] * SH(14, ch) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef DSH
    float dL_ddir[3] = {
        dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
        dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
        dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2]
    };
    float dm[3];
    dnormvdv3(dir_orig, dL_ddir, dm);
    dL_dmeans[3 * (size_t)idx + 0] += dm[0];
    dL_dmeans[3 * (size_t)idx + 1] += dm[1];
    dL_dmeans[3 * (size_t)idx + 2] += dm[2];
}

/* rasterizer_impl.cu:409-526 after a gof_oracle_forward on the same context. All outputs must be zero-filled by
 * the caller; dL_dcov3D [P,6] and dL_dconic [P,4] are never written (their kernels are dead code in the reference,
 * backward.cu:992-1007, :627-630). */
void gof_oracle_backward(gof_ctx* c, const float* background, const float* means3D, const float* shs,
                         const float* scales, const float* rotations, const float* viewmatrix, const float* cam_pos,
                         const int* radii, const float* dL_dpix,
                         float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dsh,
                         float* dL_dscale, float* dL_drot, float* dL_dview2gaussian)
{
    const int P = c->P, ntiles = c->gx * c->gy;
    size_t Pn = P > 0 ? (size_t)P : 1;
    double* a_m2 = calloc(Pn * 3, sizeof(double));
    double* a_op = calloc(Pn, sizeof(double));
    double* a_col = calloc(Pn * 3, sizeof(double));
    double* a_v2g = calloc(Pn * 10, sizeof(double));
    if (radii == NULL) radii = c->radii;

    for (int tile = 0; tile < ntiles; tile++) {
        const int tx = tile % c->gx, ty = tile / c->gx;
        const uint32_t r0 = c->ranges[2 * tile], r1 = c->ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                uint32_t px = (uint32_t)(tx * BLOCK_X + lx), py = (uint32_t)(ty * BLOCK_Y + ly);
                if (px < (uint32_t)c->W && py < (uint32_t)c->H)
                    render_pixel_backward(c, px, py, c->point_list + r0, (int)(r1 - r0), background, dL_dpix,
                                          a_m2, a_op, a_col, a_v2g);
            }
    }
    for (size_t i = 0; i < (size_t)P * 3; i++) { dL_dmean2D[i] = (float)a_m2[i]; dL_dcolor[i] = (float)a_col[i]; }
    for (size_t i = 0; i < (size_t)P; i++) dL_dopacity[i] = (float)a_op[i];
    for (size_t i = 0; i < (size_t)P * 10; i++) dL_dview2gaussian[i] = (float)a_v2g[i];
    free(a_m2); free(a_op); free(a_col); free(a_v2g);

    /* backward.cu:593-631 */
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0))
            continue;
        computeView2Gaussian_backward(scales + 3 * (size_t)idx, means3D + 3 * (size_t)idx, rotations + 4 * (size_t)idx,
                                      viewmatrix, dL_dview2gaussian + 10 * (size_t)idx, dL_dmean3D + 3 * (size_t)idx,
                                      dL_dscale + 3 * (size_t)idx, dL_drot + 4 * (size_t)idx);
        if (shs)
            computeColorFromSH_backward(idx, c->D, c->M, means3D, cam_pos, shs, c->clamped, dL_dcolor, dL_dmean3D, dL_dsh);
    }
}
