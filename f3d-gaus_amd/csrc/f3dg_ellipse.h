// f3dg_ellipse.h -- device helpers around the conservative alpha >= 1/255 ellipse of a (view, Gaussian) record
// (f3dg_preprocess.hip: conservative_ellipse), shared by the compositing forward (f3dg_render.hip) and the per-pixel pass of
// integrate (f3dg_integrate.hip).
#ifndef F3DG_ELLIPSE_H
#define F3DG_ELLIPSE_H
#include "f3dg_common.h"

namespace {

// 16-bit mask of the tile's 4x4 blocks (bit 4 * row + column) that the axis-aligned box of a Gaussian's conservative ellipse
// (e = (cx, cy, a, b), c; f3dg_preprocess.hip) touches. Half extents of a x^2 + b x y + c y^2 <= 1: sqrt(c / det), sqrt(a / det)
// with det = a c - b^2 / 4, evaluated in float32 (relative error <= ~3e-5 for the aspect ratios the records are limited to)
// and widened by 0.05 % + 2e-3 px. "everything" records (a = b = c = 0) give det = 0: every block.
__device__ __forceinline__ unsigned ellipse_block_mask(float4 e, float c, float tile_px0, float tile_py0)
{
    const float det = fmaf(e.z, c, -0.25f * e.w * e.w);
    if (!(det > 0.0f))
        return 0xFFFFu;
    const float hx = sqrtf(c / det) * 1.0005f + 2e-3f, hy = sqrtf(e.z / det) * 1.0005f + 2e-3f;
    const float x0 = e.x - hx, x1 = e.x + hx, y0 = e.y - hy, y1 = e.y + hy;
    unsigned mx = 0, my = 0;                          // which of the 4 block columns / rows the box touches
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (x0 <= tile_px0 + (float)(4 * q + 3) && x1 >= tile_px0 + (float)(4 * q)) mx |= 1u << q;
        if (y0 <= tile_py0 + (float)(4 * q + 3) && y1 >= tile_py0 + (float)(4 * q)) my |= 1u << q;
    }
    return ((my & 1u) ? mx : 0u) | ((my & 2u) ? mx << 4 : 0u) | ((my & 4u) ? mx << 8 : 0u) | ((my & 8u) ? mx << 12 : 0u);
}

// Phase 1 of render2 for pixel P of the lane's 4x4 block (and, recursively, the following ones). One comparison = one wave
// ballot (v_cmp writes a lane mask); two v_writelane_b32 (which ignore EXEC and take the SGPR halves as data) park it in lanes
// P and 16 + P of `stage`. gfx950 needs two wait states between a VALU write of an SGPR / VCC and a VALU read of it, which the
// compiler cannot see inside inline assembly: the two FMAs that evaluate the NEXT pixel's ellipse value are placed in that gap
// (s_nop for the last pixel), so the sequence costs no extra issue slots.
template <int P>
__device__ __forceinline__ void ellipse_ballots(int& stage, float Ep, const float (&dxx)[4], const float (&adx)[4],
                                                const float (&dyy)[4], const float (&cdy)[4], float eb)
{
    if constexpr (P < 15) {
        float En;
        asm("v_cmp_ge_f32 vcc, 1.0, %[ep]\n\t"
            "v_fma_f32 %[en], %[eb], %[dy], %[ax]\n\t"
            "v_fma_f32 %[en], %[dx], %[en], %[cy]\n\t"
            "v_writelane_b32 %[st], vcc_lo, %[l0]\n\t"
            "v_writelane_b32 %[st], vcc_hi, %[l1]"
            : [st] "+v"(stage), [en] "=&v"(En)
            : [ep] "v"(Ep), [eb] "v"(eb), [dy] "v"(dyy[(P + 1) >> 2]), [ax] "v"(adx[(P + 1) & 3]), [dx] "v"(dxx[(P + 1) & 3]),
              [cy] "v"(cdy[(P + 1) >> 2]), [l0] "n"(P), [l1] "n"(16 + P)
            : "vcc");
        ellipse_ballots<P + 1>(stage, En, dxx, adx, dyy, cdy, eb);
    } else {
        asm("v_cmp_ge_f32 vcc, 1.0, %[ep]\n\t"
            "s_nop 1\n\t"
            "v_writelane_b32 %[st], vcc_lo, %[l0]\n\t"
            "v_writelane_b32 %[st], vcc_hi, %[l1]\n\t"
            "s_nop 0"
            : [st] "+v"(stage)
            : [ep] "v"(Ep), [l0] "n"(P), [l1] "n"(16 + P)
            : "vcc");
    }
}

// Phase 1 of render3 (one wave64 per 8x8 pixel quadrant): lane e holds list entry e of the window and evaluates its ellipse at
// pixel P = 8 * row + column of the quadrant (and, recursively, at the following ones); the comparison IS the ballot "which entries
// can reach pixel P", and two v_writelane_b32 hand its halves to lane P -- the lane that blends pixel P -- so that after the 64
// steps every lane holds the 64-bit pass mask of its own pixel. Same instruction pattern and wait states as ellipse_ballots.
template <int P>
__device__ __forceinline__ void quad_ballots(int& lo, int& hi, float Ep, const float (&dxx)[8], const float (&adx)[8],
                                             const float (&dyy)[8], const float (&cdy)[8], float eb)
{
    if constexpr (P < 63) {
        float En;
        asm("v_cmp_ge_f32 vcc, 1.0, %[ep]\n\t"
            "v_fma_f32 %[en], %[eb], %[dy], %[ax]\n\t"
            "v_fma_f32 %[en], %[dx], %[en], %[cy]\n\t"
            "v_writelane_b32 %[lo], vcc_lo, %[l]\n\t"
            "v_writelane_b32 %[hi], vcc_hi, %[l]"
            : [lo] "+v"(lo), [hi] "+v"(hi), [en] "=&v"(En)
            : [ep] "v"(Ep), [eb] "v"(eb), [dy] "v"(dyy[(P + 1) >> 3]), [ax] "v"(adx[(P + 1) & 7]), [dx] "v"(dxx[(P + 1) & 7]),
              [cy] "v"(cdy[(P + 1) >> 3]), [l] "n"(P)
            : "vcc");
        quad_ballots<P + 1>(lo, hi, En, dxx, adx, dyy, cdy, eb);
    } else {
        asm("v_cmp_ge_f32 vcc, 1.0, %[ep]\n\t"
            "s_nop 1\n\t"
            "v_writelane_b32 %[lo], vcc_lo, %[l]\n\t"
            "v_writelane_b32 %[hi], vcc_hi, %[l]\n\t"
            "s_nop 0"
            : [lo] "+v"(lo), [hi] "+v"(hi)
            : [ep] "v"(Ep), [l] "n"(P)
            : "vcc");
    }
}

// Half-window variant (render3's sliding window): 32 entries, lanes e and e + 32 both hold entry e; lane e evaluates the pixels of
// rows 0..3 (pixel P = 8 * row + column), lane e + 32 those of rows 4..7 (pixel P + 32) with its own dyy / cdy, so one comparison
// carries both ballots: vcc_lo = entries reaching pixel P, vcc_hi = entries reaching pixel P + 32, each 32 bits wide.
template <int P>
__device__ __forceinline__ void half_ballots(int& m, float Ep, const float (&dxx)[8], const float (&adx)[8],
                                             const float (&dyy)[4], const float (&cdy)[4], float eb)
{
    if constexpr (P < 31) {
        float En;
        asm("v_cmp_ge_f32 vcc, 1.0, %[ep]\n\t"
            "v_fma_f32 %[en], %[eb], %[dy], %[ax]\n\t"
            "v_fma_f32 %[en], %[dx], %[en], %[cy]\n\t"
            "v_writelane_b32 %[m], vcc_lo, %[l0]\n\t"
            "v_writelane_b32 %[m], vcc_hi, %[l1]"
            : [m] "+v"(m), [en] "=&v"(En)
            : [ep] "v"(Ep), [eb] "v"(eb), [dy] "v"(dyy[(P + 1) >> 3]), [ax] "v"(adx[(P + 1) & 7]), [dx] "v"(dxx[(P + 1) & 7]),
              [cy] "v"(cdy[(P + 1) >> 3]), [l0] "n"(P), [l1] "n"(P + 32)
            : "vcc");
        half_ballots<P + 1>(m, En, dxx, adx, dyy, cdy, eb);
    } else {
        asm("v_cmp_ge_f32 vcc, 1.0, %[ep]\n\t"
            "s_nop 1\n\t"
            "v_writelane_b32 %[m], vcc_lo, %[l0]\n\t"
            "v_writelane_b32 %[m], vcc_hi, %[l1]\n\t"
            "s_nop 0"
            : [m] "+v"(m)
            : [ep] "v"(Ep), [l0] "n"(P), [l1] "n"(P + 32)
            : "vcc");
    }
}

// The same with a wave-level summary: `any` collects the OR of the 64 ballots, i.e. the mask of the window's entries that can reach
// at least one pixel of the quadrant (the compositing backward walks entries in lock-step and skips the others outright).
template <int P>
__device__ __forceinline__ void quad_ballots_any(int& lo, int& hi, unsigned long long& any, float Ep, const float (&dxx)[8],
                                                 const float (&adx)[8], const float (&dyy)[8], const float (&cdy)[8], float eb)
{
    if constexpr (P < 63) {
        float En;
        asm("v_cmp_ge_f32 vcc, 1.0, %[ep]\n\t"
            "v_fma_f32 %[en], %[eb], %[dy], %[ax]\n\t"
            "v_fma_f32 %[en], %[dx], %[en], %[cy]\n\t"
            "v_writelane_b32 %[lo], vcc_lo, %[l]\n\t"
            "v_writelane_b32 %[hi], vcc_hi, %[l]\n\t"
            "s_or_b64 %[any], %[any], vcc"
            : [lo] "+v"(lo), [hi] "+v"(hi), [any] "+s"(any), [en] "=&v"(En)
            : [ep] "v"(Ep), [eb] "v"(eb), [dy] "v"(dyy[(P + 1) >> 3]), [ax] "v"(adx[(P + 1) & 7]), [dx] "v"(dxx[(P + 1) & 7]),
              [cy] "v"(cdy[(P + 1) >> 3]), [l] "n"(P)
            : "vcc", "scc");
        quad_ballots_any<P + 1>(lo, hi, any, En, dxx, adx, dyy, cdy, eb);
    } else {
        asm("v_cmp_ge_f32 vcc, 1.0, %[ep]\n\t"
            "s_nop 1\n\t"
            "v_writelane_b32 %[lo], vcc_lo, %[l]\n\t"
            "v_writelane_b32 %[hi], vcc_hi, %[l]\n\t"
            "s_or_b64 %[any], %[any], vcc"
            : [lo] "+v"(lo), [hi] "+v"(hi), [any] "+s"(any)
            : [ep] "v"(Ep), [l] "n"(P)
            : "vcc", "scc");
    }
}

} // namespace

#endif
