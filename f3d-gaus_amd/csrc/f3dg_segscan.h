// f3dg_segscan.h -- segmented scans over the lanes of a wave64 (gfx950 DPP), shared by the split-pixel compositing forward
// (f3dg_render5.hip: runs = the pairs of one pixel) and the dense compositing backward (f3dg_backward5.hip: runs = the pairs of one list
// entry). A batch holds 64 items, one per lane; consecutive lanes with the same key are a RUN; every lane knows `rr`, the number of lanes
// between it and the first lane of its run inside the batch. Hillis-Steele over six DPP steps: row_shr 1 2 4 8 inside the 16-lane rows,
// row_bcast 15 and 31 across them; a step applies to a lane when it reaches back no further than the run's first lane.
#pragma once
#include <hip/hip_runtime.h>

namespace {

// v_mov_b32_dpp with `old` for the lanes whose source lies outside the row / the wave or whose row is masked off
#define F3DG_DPP(old, x, ctrl, rowmask) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), ctrl, rowmask, 0xf, false))
#define F3DG_DPP_ROW_SHR(n) (0x110 + (n))
#define F3DG_DPP_WAVE_SHR1 0x138
#define F3DG_DPP_BCAST15 0x142
#define F3DG_DPP_BCAST31 0x143

// what a lane needs to know about its place in its run (wave-uniform code, all 64 lanes active)
struct SegFlags {
    bool c1, c2, c4, c8, c16, c32;          // does step d reach back inside the lane's run?
    float f1, f2, f4, f8, f16, f32;         // the same as 1.0f / 0.0f
};

__device__ __forceinline__ SegFlags seg_flags(unsigned rr, unsigned lane)
{
    SegFlags s;
    s.c1 = rr >= 1u; s.c2 = rr >= 2u; s.c4 = rr >= 4u; s.c8 = rr >= 8u;
    s.c16 = rr > (lane & 15u);          // the run began in an earlier row: take the previous row's total (row_bcast:15, rows 1 and 3)
    s.c32 = rr > (lane & 31u);          // the run began below lane 32: take lane 31's total (row_bcast:31, rows 2 and 3)
    s.f1 = s.c1 ? 1.0f : 0.0f; s.f2 = s.c2 ? 1.0f : 0.0f; s.f4 = s.c4 ? 1.0f : 0.0f; s.f8 = s.c8 ? 1.0f : 0.0f;
    s.f16 = s.c16 ? 1.0f : 0.0f; s.f32 = s.c32 ? 1.0f : 0.0f;
    return s;
}

// segmented inclusive product over the lanes of a run
__device__ __forceinline__ float seg_product(float x, const SegFlags& s)
{
    float t;
    t = F3DG_DPP(1.0f, x, F3DG_DPP_ROW_SHR(1), 0xf); x *= s.c1 ? t : 1.0f;
    t = F3DG_DPP(1.0f, x, F3DG_DPP_ROW_SHR(2), 0xf); x *= s.c2 ? t : 1.0f;
    t = F3DG_DPP(1.0f, x, F3DG_DPP_ROW_SHR(4), 0xf); x *= s.c4 ? t : 1.0f;
    t = F3DG_DPP(1.0f, x, F3DG_DPP_ROW_SHR(8), 0xf); x *= s.c8 ? t : 1.0f;
    t = F3DG_DPP(1.0f, x, F3DG_DPP_BCAST15, 0xa); x *= s.c16 ? t : 1.0f;
    t = F3DG_DPP(1.0f, x, F3DG_DPP_BCAST31, 0xc); x *= s.c32 ? t : 1.0f;
    return x;
}

// segmented inclusive sums of K registers over the lanes of a run: per step and register ONE instruction, x += dpp(x) * flag. Lanes whose
// DPP source is outside their row (or whose row is masked off) are not written. gfx950 wants two wait states between a VALU write of a
// VGPR and a DPP read of it, which the compiler cannot see inside inline assembly -- and it may place its own register copies between
// two asm statements -- so ALL steps of a group are one statement: an s_nop opens it, within it the K - 1 other registers of a step lie
// between a register's write and its next read (K = 2: one more wait state per step).
#define F3DG_S5_LINE(r, f, ctl) "v_fmac_f32_dpp %" #r ", %" #r ", %[" #f "] " ctl "\n\t"
#define F3DG_S5_STEPS(STEP)                                                                                                   \
    STEP(f1, "row_shr:1 row_mask:0xf bank_mask:0xf") STEP(f2, "row_shr:2 row_mask:0xf bank_mask:0xf")                       \
    STEP(f4, "row_shr:4 row_mask:0xf bank_mask:0xf") STEP(f8, "row_shr:8 row_mask:0xf bank_mask:0xf")                       \
    STEP(f16, "row_bcast:15 row_mask:0xa bank_mask:0xf") STEP(f32, "row_bcast:31 row_mask:0xc bank_mask:0xf")
#define F3DG_S5_FLAGS(s) [f1] "v"((s).f1), [f2] "v"((s).f2), [f4] "v"((s).f4), [f8] "v"((s).f8), [f16] "v"((s).f16), [f32] "v"((s).f32)
#define F3DG_S5_STEP2(f, ctl) F3DG_S5_LINE(0, f, ctl) F3DG_S5_LINE(1, f, ctl) "s_nop 0\n\t"
#define F3DG_S5_STEP4(f, ctl) F3DG_S5_LINE(0, f, ctl) F3DG_S5_LINE(1, f, ctl) F3DG_S5_LINE(2, f, ctl) F3DG_S5_LINE(3, f, ctl)
#define F3DG_S5_STEP8(f, ctl) F3DG_S5_STEP4(f, ctl) F3DG_S5_LINE(4, f, ctl) F3DG_S5_LINE(5, f, ctl) F3DG_S5_LINE(6, f, ctl) F3DG_S5_LINE(7, f, ctl)

__device__ __forceinline__ void seg_sums(float (&v)[2], const SegFlags& s)
{
    asm volatile("s_nop 1\n\t" F3DG_S5_STEPS(F3DG_S5_STEP2) : "+v"(v[0]), "+v"(v[1]) : F3DG_S5_FLAGS(s));
}
__device__ __forceinline__ void seg_sums(float (&v)[4], const SegFlags& s)
{
    asm volatile("s_nop 1\n\t" F3DG_S5_STEPS(F3DG_S5_STEP4) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : F3DG_S5_FLAGS(s));
}
__device__ __forceinline__ void seg_sums(float (&v)[8], const SegFlags& s)
{
    asm volatile("s_nop 1\n\t" F3DG_S5_STEPS(F3DG_S5_STEP8)
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : F3DG_S5_FLAGS(s));
}

} // namespace
