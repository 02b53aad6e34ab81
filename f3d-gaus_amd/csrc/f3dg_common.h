// f3dg_common.h -- internal declarations shared by the HIP translation units of libf3dg_hip.so.
// gfx950 only. Parity-critical arithmetic is compiled with -ffp-contract=off (see build.py / DESIGN.md):
// the GOF exponent cancels ~1e5..1e6 x, so the float32/float64 operation order of the reference is kept
// literally.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/f3dg.h"

#define F3DG_BLOCK 256                 // threads per workgroup = one 16x16 tile = 4 wave64
#define F3DG_REC_FLOATS 16             // per-(view,Gaussian) compositing record, 64 B
#define F3DG_NEAR_PLANE 0.2            // auxiliary.h:27 (double)
#define F3DG_FAR_PLANE 100.0           // auxiliary.h:28 (double)
#define F3DG_SORT_ITEMS 16             // keys per thread per radix block
#define F3DG_SORT_CHUNK (F3DG_BLOCK * F3DG_SORT_ITEMS)
#define F3DG_SCAN_ITEMS 16
#define F3DG_SCAN_CHUNK (F3DG_BLOCK * F3DG_SCAN_ITEMS)

// A point-list entry is the Gaussian id in its low 28 bits and, above them, the mask of the tile's four 8x8 pixel quadrants
// (bit 28 + 2 * qy + qx) that the axis-aligned box of the Gaussian's conservative alpha >= 1/255 ellipse reaches: the one-wave
// compositing kernel (render3, f3dg_render.hip) stages an entry only in the quadrants whose bit is set. Every other consumer
// strips the mask. The tile rectangles that carry the information from the projection kernel to the instance generation keep
// tile coordinates in 15 bits; bit 15 / 31 of a word says that the box misses the first / second half of the first / last tile.
#define F3DG_ID_BITS 28
#define F3DG_ID_MASK 0x0FFFFFFFu
#define F3DG_RECT_COORD 0x7FFFu
#define F3DG_RECT_SKIP_LO 0x8000u
#define F3DG_RECT_SKIP_HI 0x80000000u

// XCD-aware (view, unit) of workgroup b of a V * U grid. Consecutive workgroup ids land on different XCDs (id % 8); the views are
// handed out in groups of 8, one view per XCD, so that all units (tiles, sort chunks) of a view share one XCD's L2. The last
// V % 8 views are spread over all XCDs unit by unit (a single-view call uses the whole chip).
__device__ __forceinline__ void f3dg_xcd_map(unsigned b, unsigned V, unsigned U, unsigned& view, unsigned& unit)
{
    const unsigned full = (V & ~7u) * U;
    if (b < full) {
        const unsigned slot = b >> 3;
        view = (slot / U) * 8u + (b & 7u);
        unit = slot % U;
    } else {
        const unsigned r = b - full;
        view = (V & ~7u) + r / U;
        unit = r % U;
    }
}

// Workspace header (device memory, first 256 bytes of the workspace)
struct F3dgHeader {
    unsigned int num_rendered;   // total (Gaussian, tile) instances of the call (all views)
    unsigned int overflow;       // 1 if num_rendered > capacity
    unsigned int capacity;       // instance capacity the workspace was carved for
    unsigned int bwd_stale;      // 1: f3dg_backward ran on a workspace whose last forward was not a SAVE_AUX call of the general path
    unsigned int reserved1[2];
    unsigned int alpha_fast;      // arithmetic the compositing forward of this call used for alpha (1: error-free float32 pairs): the
                                  // backward must repeat it to the bit
    unsigned int save_aux;        // 1 when the forward of this workspace ran with F3DG_FLAG_SAVE_AUX (the auxiliary planes are valid)
    unsigned long long bwd_pairs; // contributing (pixel, Gaussian) pairs of the last f3dg_backward on this workspace ("C" of SURVEY 8d)
    unsigned int small_path;      // 1 when the forward of this workspace took the small-call path (per-tile lists in `small_list`)
    unsigned int small_overflow;  // 1 when a tile of that path held more than F3DG_SMALL_CAP entries (the caller re-runs the general path)
    unsigned int small_shape[4];  // P, n_views, W, H of that call (f3dg_read_status disables the path for a shape that overflowed)
    unsigned int reserved[48];
};

// Per-(view, Gaussian) record consumed by the compositing kernel: one 64-byte line.
//   f[0..9]   view2gaussian (Sigma' upper triangle, B, C)     forward.cu:268-277
//   f[10]     opacity * coef (conic_opacity.w)                 forward.cu:392
//   f[11]     pre-test constant K: alpha < 1/255 is CERTAIN when b^2 < K*a (see f3dg_render.hip)
//   f[12..14] rgb                                              forward.cu:379-384
//   f[15]     c of the conservative ellipse (f3dg_preprocess.hip); the view-space depth (forward.cu:388) is in `depths`
// The first three float4 are everything the conservative pre-test needs; the 4th is only read by contributors.
struct __attribute__((aligned(16))) F3dgRec { float f[F3DG_REC_FLOATS]; };

// The "fast" arithmetic of a (pixel, Gaussian) pair up to G = exp(power) (option render_fast; f3dg_render.hip: blend_entry_fast). ONE
// definition, because the compositing backward must repeat the forward's alpha to the bit when the forward took this mode
// (F3dgHeader::alpha_fast). aaf, bhalf are the reference's own float32 a and b / 2 (forward.cu:499-509, in its operation order: their
// rounding errors are amplified 1e5..1e6 x by the cancellation below and must be reproduced, not improved on), CC is the quadric's
// constant. The reference's float64 island (forward.cu:511-522) becomes error-free float32 pairs:
//   b^2 = p + e exactly (FMA), q1 = p r, q2 = ((p - q1 a) + e) r with r ~ 1/a: b^2/a = q1 + q2 to ~2^-45; C - q1 is exact (Sterbenz)
//   wherever the exponent matters, so min_value = (C - q1) - q2 carries one rounding of a number of magnitude <~ 20;
//   t = -b/a = -b r (<= 1.5 ulp: v_rcp_f32 is good to 1 ulp; round 3 spent two more FMAs on a Newton step for it);
//   G = exp(min(-min_value / 2, 0)) as v_exp_f32(min(min_value * (-log2(e) / 2), 0)): one multiply and one v_min instead of the
//   reference's multiply, compare + select, and the log2(e) multiply (a NaN exponent gives G = 1 where the reference's gives NaN:
//   both are garbage, and no finite record produces one).
#ifndef F3DG_FAST_R03
#define F3DG_FAST_R03 0
#endif
__device__ __forceinline__ void f3dg_fast_t_G(float aaf, float bhalf, float CC, float& t, float& G)
{
    const float r = __builtin_amdgcn_rcpf(aaf);
#if F3DG_FAST_R03      // A/B switch: round 3's sequence (Newton step for t, compare + select clamp, separate log2(e) multiply): +5 VALU per pair
    const float t0 = -bhalf * r;
    t = fmaf(fmaf(-aaf, t0, -bhalf), r, t0);
#else
    t = -bhalf * r;
#endif
    const float p = bhalf * bhalf;
    const float e = fmaf(bhalf, bhalf, -p);
    const float q1 = p * r;
    const float q2 = (fmaf(-q1, aaf, p) + e) * r;
    const float min_value = (CC - q1) - q2;
#if F3DG_FAST_R03
    float power = -0.5f * min_value;
    if (power > 0.0f) power = 0.0f;
    G = __builtin_amdgcn_exp2f(power * 1.4426950408889634f);
#else
    G = __builtin_amdgcn_exp2f(fminf(min_value * -0.7213475204444817f, 0.0f));
#endif
}

// Host-side description of where each array lives inside the workspace (byte offsets).
struct F3dgLayout {
    size_t header;
    size_t rec;            // [V*P] F3dgRec
    size_t means2D;        // [V*P] float2
    size_t depths;         // [V*P] float: view-space depth again, compact, for key generation (4 B instead of a 64-B record line)
    size_t bbox;           // [V*P] float4: conservative pixel-space box (x0,x1,y0,y1) outside of which alpha < 1/255 is certain
    size_t cull;           // [V*P] float4: conservative ellipse of the same region (cx, cy, a, b); its c is record slot 15
    size_t conic;          // [V*P] float4 (conic.xyz, opacity*coef) -- backward only
    size_t radii;          // [V*P] int   (internal copy when the caller passes none)
    size_t tiles;          // [V*P] u32   tiles_touched
    size_t offsets;        // [V*P] u32   inclusive scan of tiles_touched
    size_t clamped;        // [V*P] u8    bit c set when SH colour channel c was clamped
    size_t rects;          // [V*P] uint2: tile rectangle of every (view, Gaussian) (rminx | rmaxx << 16, rminy | rmaxy << 16)
    size_t gsort;          // [7][V*P] u32: ping-pong (key, id) buffers of the per-view depth sort of the Gaussians ([0..3]; without
                           // sort_fused_rects reused for the sorted tile counts and rectangles), [4..6] tile counts -> prefix sum, rx, ry in sorted order
    size_t scan_tmp;       // u32 block sums for the scans
    size_t keys[2];        // group stream (u16 / u32 per instance) of the two ping-pong halves, [cap] 4 B; [0] is [cap] 8 B: it also
                           // holds the rebuilt 64-bit keys of the debug export
    size_t vals[2];        // [cap] u32 Gaussian ids of the halves; vals[0] ends as the compositing kernel's point list
    size_t segtab;         // per-view segments of the instance arrays (vstart[V+1], bglob[V+1], xprefix[8][V/8+2]; f3dg_binning.hip)
    size_t hist;           // [256 * max((view, chunk) blocks of the depth sort, of the tile pass)] u32
    size_t ranges;         // [V*T] uint2
    size_t final_T;        // [V][4][H*W] float
    size_t n_contrib;      // [V][2][H*W] u32
    size_t bwd_acc;        // [V*P][16] double-sized slots (128 B): float64 accumulator of dL/dview2gaussian [10]; with the dense backward also the seven
                           // float32 sums of colour, mean2D and opacity at byte 80 (lock-step backward: the first 80 V P bytes as [V*P][10])
    // small-call path (f3dg_small.hip; carved only for the shapes it serves, small_cap = 0 otherwise)
    size_t small_boxes;    // [V][ceil(P/64)] uint2: union of the tile rectangles of every 64 consecutive Gaussians (small path only)
    size_t small_cnt;      // [V*T] u32: length of every (view, tile) list
    size_t small_list;     // [V*T][small_cap] u32: the sorted lists (the compositing kernel's point list on this path)
    unsigned int small_cap;
    size_t total;
    unsigned int sort_blocks;
    unsigned int scan_tmp_elems;
    unsigned int segtab_minmax;   // element offset inside segtab of the per-view key range of the depth sort (minmax[2 V])
};

F3dgLayout f3dg_layout(int P, int W, int H, int V, long long cap);

// Extra arrays of f3dg_integrate, carved behind the one-view forward layout (byte offsets into the same workspace)
struct F3dgIntegLayout {
    size_t pix_points;     // [H*W] u32: points per pixel                      } cleared together,
    size_t tile_last;      // [T] u64: max (depth bits << 32 | point index)    } clear_bytes from pix_points
    size_t clear_bytes;
    size_t contrib_n;      // [H*W] u32: entries of the pixel's contributor list
    size_t contrib_ids;    // [H*W][1024] u16: 1-based list positions of the contributing Gaussians (forward.cu:862, 969)
    size_t pix_start;      // [H*W] u32: exclusive scan of pix_points
    size_t scan_tmp;       // block sums of that scan
    size_t pt_pix, pt_rank, perm;   // [PN] u32 each: pixel of a point (~0: not integrated), its rank in the pixel, pixel order
    size_t redo;           // [V*T] u32: tiles of pass 1 that integrate_pass1_rays_kernel hands to the per-pixel kernel (1,024 contributors reached)
    size_t total;
    unsigned scan_tmp_elems;
};
F3dgIntegLayout f3dg_integ_layout(int P, int PN, int W, int H, long long cap, int V = 1);   // V cameras prepared together

// every kernel launch of the library goes through this macro: f3dg_debug_launch_count reports how many a call sequence issued
#include <atomic>
extern std::atomic<unsigned long long> g_f3dg_kernel_launches;
// (host-side cost of every launch site: option "time_launches" + f3dg_debug_launch_times, a diagnostic of the small-call path)
extern int g_f3dg_time_launches;
void f3dg_note_launch_time(const char* file, int line, long long ns);
long long f3dg_now_ns();
#define F3DG_KLAUNCH(...) do { g_f3dg_kernel_launches.fetch_add(1ull, std::memory_order_relaxed);                                                                   \
        if (g_f3dg_time_launches) { const long long t0_ = f3dg_now_ns(); hipLaunchKernelGGL(__VA_ARGS__);                  \
                                    f3dg_note_launch_time(__FILE__, __LINE__, f3dg_now_ns() - t0_); }                      \
        else hipLaunchKernelGGL(__VA_ARGS__); } while (0)

int f3dg_set_hip_error(hipError_t e, const char* where);
#define F3DG_HIP_CHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return f3dg_set_hip_error(_e, #expr); } while (0)

struct F3dgViewConsts {           // passed by pointer: [V] of these are the caller's flat arrays
    const float* viewmatrix;      // [V,16]
    const float* projmatrix;      // [V,16]
    const float* cam_pos;         // [V,3]
};

// Workspace header initialisation folded into the projection kernel's first workgroup (one launch less per call)
struct F3dgHeaderInit {
    F3dgHeader* hdr;              // null: the header is initialised elsewhere
    unsigned capacity, alpha_fast, save_aux, small_path;
    unsigned shape[4];            // P, n_views, W, H (recorded for the small-call path)
};

// ---- launchers (each returns F3DG_OK or a negative error) -------------------------------------------
// views_per_set: the V views are n_sets = V / views_per_set groups, group i renders Gaussian set i of the [n_sets, P, ...] inputs
int f3dg_launch_preprocess(hipStream_t s, int V, int views_per_set, int P, int D, int M, const float* means3D, const float* scales,
                           float scale_modifier, const float* rotations, const float* opacities, const float* shs,
                           const float* cov3D_precomp, const float* colors_precomp, const float* v2g_precomp,
                           const float* viewmatrix, const float* projmatrix, const float* cam_pos, int W, int H,
                           float tan_fovx, float tan_fovy, float focal_x, float focal_y, float kernel_size,
                           F3dgRec* rec, float2* means2D, float* depths, unsigned* sort_keys, uint2* rects, float4* bbox /* may be null */, float4* cull, float4* conic,
                           int* radii, unsigned* tiles, unsigned char* clamped, int save_aux, int tile_cull, F3dgHeaderInit init,
                           float4* hoist = nullptr /* scratch of n_sets * P * 96 bytes: option pre_hoist */, int n_sets = 1,
                           uint2* chunk_boxes = nullptr /* small path: [V][ceil(P/64)] unions of 64 rectangles */);
extern int g_f3dg_pre_hoist;            // 1: the view-independent part of the projection is computed once per Gaussian (preprocess_hoist_kernel)

// Small-call path: entries per (view, tile) it can hold, and the shapes it serves (one or two views of at most 2^18 Gaussians on at
// most 1024 tiles: the reference's one-view-per-call loops, visualize.py:293-314, 387-416)
#define F3DG_SMALL_CAP 4096
inline bool f3dg_small_shape(int P, int W, int H, int V)
{
    const long long T = (long long)((W + F3DG_TILE - 1) / F3DG_TILE) * ((H + F3DG_TILE - 1) / F3DG_TILE);
    return V >= 1 && V <= 2 && P >= 1 && P <= (1 << 18) && T <= 1024;
}
int f3dg_launch_small_bin(hipStream_t s, int V, int P, int W, int H, const F3dgLayout& L, char* ws);
int f3dg_launch_small_export(hipStream_t s, int V, int W, int H, const F3dgLayout& L, const char* ws, unsigned* point_list, unsigned* ranges);

int f3dg_launch_scan_inclusive(hipStream_t s, const unsigned* in, unsigned* out, unsigned long long n,
                               unsigned* tmp, unsigned tmp_elems, int exclusive,
                               F3dgHeader* hdr_total /* if not null: write total + overflow */);

// export_offsets: also compute the reference's (view, Gaussian)-ordered prefix sum of tiles_touched (debug export only)
int f3dg_launch_binning(hipStream_t s, int V, int P, int W, int H, const F3dgLayout& L, char* ws, int export_offsets);
int f3dg_launch_export_keys(hipStream_t s, int V, int P, int W, int H, const F3dgLayout& L, char* ws);

extern int g_f3dg_render_pretest;      // 1 (default): conservative f32 pre-test enabled; 0: plain path (A/B, tests)
extern int g_f3dg_render_cull;         // 1 (default): per-strip culling of the staged list by the conservative box
extern int g_f3dg_sort_fused_rects;    // 1: a view's last depth pass also gathers its rectangles into sorted order (no gsort_gather_rects_kernel); 0 (default): measured equal
extern int g_f3dg_sort_wide_groups;    // 0 (default): u16 group stream when it fits; 1: always u32 (tests)
extern int g_f3dg_render_queue;        // 1 (default): two-phase loop with per-lane work queues
extern int g_f3dg_render_kernel;       // 3 (default): render3 (one wave64 per 8x8 quadrant, no barriers); 2: render2 (four waves per tile,
                                       // Gaussians across the lanes in phase 1); 1: the pixel-lane kernel with its filters
extern int g_f3dg_small_debug;         // timing experiments of small_bin_kernel: 1 return at once, 2 after the collection, 3 no sort passes
extern int g_f3dg_small_path;          // 1 (default): inference calls of a small shape (f3dg_small_shape) take the three-launch path
extern int g_f3dg_bwd_occ;             // waves per SIMD render3_bwd_kernel is compiled for: 5 (default: 10.0 ms at C5), 2..4 (10.2-10.4: the kernel is VALU-bound at any of them) or 6 (spills, 12.0)
extern int g_f3dg_render_lds_pad;      // experiment: extra dynamic LDS bytes per render3 workgroup (lowers the occupancy)
extern int g_f3dg_render_unroll;       // small launches (render_lowocc): entries per phase-2 trip of render3p (1 or 2); -1 = by launch size and arithmetic
int f3dg_launch_render_small(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y, const F3dgHeader* hdr, const uint2* ranges,
                         const unsigned* point_list, const F3dgRec* rec, const float4* cull, const float* background, int bg_per_view,
                         float* out_color, int fast, int save_aux, float* final_T, unsigned* n_contrib, int unroll, int split, int count);
extern int g_f3dg_render_split;        // small launches: 1 = two waves per quadrant (render3p_fwd_kernel: a producer wave scans, gathers and runs phase 1 for the window after the one the consumer wave composites)
extern int g_f3dg_render_lowocc;       // n >= 1 (default 1): launches of at most max(2, n) x 1024 quadrant waves take the multi-wave kernels of f3dg_render4.hip; 0: never
extern int g_f3dg_render_slide;        // 1 (default): render3 with the sliding half-window (render3s_fwd_kernel); 0: fixed 64-entry windows
extern int g_f3dg_render_tail;         // N > 0: render3s switches a quadrant to the tail schedule once at most N of its pixels are unsaturated (0: never)
extern int g_f3dg_render_count;        // 1: the one-wave kernel's counting variant (diagnostic; f3dg_debug_render_counts)
extern int g_f3dg_render_wpb;          // quadrant waves per render3s workgroup: 1 (default) or 4 (a tile's four waves start together on one CU)
extern int g_f3dg_render_replay;       // lab builds (-DF3DG_LAB) only: 2 / 3 = launch render3s_stage_only_kernel instead of the compositing kernel
extern int g_f3dg_render_dma;          // render3 stages the records with global_load_lds_dwordx4 (1, default) or through registers (0)
int f3dg_prof_bwd_begin(hipStream_t s);
void f3dg_prof_bwd_mark(int slot, int stage_done, hipStream_t s);
int f3dg_render_uses_fast(int save_aux);       // the arithmetic mode a compositing launch with / without SAVE_AUX takes
extern int g_f3dg_render_round;        // list entries render2 stages per round in fast arithmetic: 192 (default, 7 workgroups per CU) or 256 (6)
extern int g_f3dg_render_fast;         // 1 (default): float64 island of the blend replaced by error-free float32 pairs in inference
                                       // calls (no SAVE_AUX); 2: also with SAVE_AUX (tests); 0: the reference's float32/float64 order always

int f3dg_launch_render(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y,
                       const F3dgHeader* hdr, const uint2* ranges, const unsigned* point_list, const F3dgRec* rec,
                       const float4* bbox, const float4* cull, const float* background, int bg_per_view, float* out_color,
                       float* final_T, unsigned* n_contrib, int save_aux, unsigned skip_channels = 0u, int fast = -1 /* -1: the process default */,
                       int scan = 0 /* the call carries F3DG_FLAG_SCAN */);

// the rank-packed compositing forward (f3dg_render4.hip; option render_kernel = 4, inference launches)
extern int g_f3dg_render_pack;         // -1 (default): inference launches in the reference's arithmetic take render4; 1: all inference launches; 0: none
extern int g_f3dg_render_pack_th;      // trips of a slide with at most this many participating pixels are packed (default 32; 0: never)
int f3dg_launch_render4(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y, const F3dgHeader* hdr, const uint2* ranges,
                        const unsigned* point_list, const F3dgRec* rec, const float4* cull, const float* background, int bg_per_view,
                        float* out_color, int fast, unsigned skip_channels, int count, int save_aux = 0, float* final_T = nullptr,
                        unsigned* n_contrib = nullptr);

// the split-pixel compositing forward (f3dg_render5.hip; F3DG_FLAG_SCAN / option render_scan: fast inference launches of the general path)
extern int g_f3dg_render_scan;         // -1 (default): calls with F3DG_FLAG_SCAN; 1: every eligible launch; 0: never
extern int g_f3dg_render_scan_min;     // stragglers holding fewer older-half entries than this finish the slide in fused trips (default 4; 0: always compact)
extern int g_f3dg_tile_split;          // two tile passes of the binning split their bits evenly (default 1; lab option tile_split)
extern int g_f3dg_render_scan_lanes;   // lanes per pixel of the one- and two-view scan kernel (render5p): 4 (default) or 2 (lab option render_scan_lanes)
extern int g_f3dg_render_scan_th;      // fused trips while more than this many pixels take part (default 20)
int f3dg_launch_render5(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y, const F3dgHeader* hdr, const uint2* ranges,
                        const unsigned* point_list, const F3dgRec* rec, const float4* cull, const float* background, int bg_per_view,
                        float* out_color, unsigned skip_channels, int count);

// the dense compositing backward (f3dg_backward5.hip; option bwd_dense)
int f3dg_launch_render5_small(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y, const F3dgHeader* hdr, const uint2* ranges,
                              const unsigned* point_list, const F3dgRec* rec, const float4* cull, const float* background, int bg_per_view,
                              float* out_color);
extern int g_f3dg_bwd_dense;           // 1: render5_bwd_kernel (entry-major batches of (pixel, entry) pairs, segmented scans); 0: render3_bwd_kernel (lock-step walk)
int f3dg_launch_render5_bwd(hipStream_t s, int V, int P, int W, int H, int tiles_x, int T, float focal_x, float focal_y, F3dgHeader* hdr,
                            const uint2* ranges, const unsigned* point_list, const unsigned* small_list, const F3dgRec* rec, const float4* cull,
                            const float2* means2D, const float4* conic, const float* background, int bg_per_view, const float* final_T,
                            const unsigned* n_contrib, const float* dL_dpixels, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolors,
                            double* dL_dv2g_acc, int debug_no_atomics);

int f3dg_launch_integrate_fill(hipStream_t s, int W, int H, int PN, float* out_color, float* out_alpha_integrated,
                               float* out_color_integrated);
int f3dg_launch_integrate_pass1(hipStream_t s, int V, int P, int W, int H, float focal_x, float focal_y, const F3dgLayout& L,
                                const F3dgIntegLayout& I, char* ws, const float* background, float* out_color);
int f3dg_launch_integrate_points(hipStream_t s, int view, int P, int PN, int W, int H, float focal_x, float focal_y, const F3dgLayout& L,
                                 const F3dgIntegLayout& I, char* ws, const float* points3D, const float* viewmatrix,
                                 float* out_color, float* out_alpha_integrated, float* out_color_integrated, float* alpha_min);
