/*
 * f3dg.h -- C ABI of libf3dg_hip.so: the MI355X (gfx950) implementation of F3D-Gaus's GOF rasterization +
 * cycle-aggregative projection hot path.
 *
 * Plain pointers and sizes only (no torch / no C++ types). Every pointer is a DEVICE pointer unless the
 * parameter name starts with `h_`. Nothing in this library synchronises the host with the device except
 * the functions documented as blocking. `stream` is a hipStream_t passed as void* (NULL = default stream).
 *
 * What each entry point replaces in the reference (paths relative to /root/reference; RAST =
 * src/gaussian-splatting/submodules/diff-gof-rasterization):
 *
 *   f3dg_forward / f3dg_forward_batched
 *       CudaRasterizer::Rasterizer::forward      RAST/cuda_rasterizer/rasterizer.h:31-55,
 *                                                RAST/cuda_rasterizer/rasterizer_impl.cu:247-405
 *       reached from pybind `rasterize_gaussians` RAST/ext.cpp:16, RAST/rasterize_points.cu:36-122
 *   f3dg_backward
 *       CudaRasterizer::Rasterizer::backward     RAST/cuda_rasterizer/rasterizer.h:57-90,
 *                                                RAST/cuda_rasterizer/rasterizer_impl.cu:409-526
 *       reached from pybind `rasterize_gaussians_backward` RAST/ext.cpp:18, RAST/rasterize_points.cu:124-211
 *   f3dg_mark_visible
 *       CudaRasterizer::Rasterizer::markVisible  RAST/cuda_rasterizer/rasterizer.h:23-29,
 *                                                RAST/cuda_rasterizer/rasterizer_impl.cu:172-186 (pybind `mark_visible`)
 *   f3dg_workspace_bytes / f3dg_read_status
 *       the std::function<char*(size_t)> resize callbacks + the blocking 4-byte D2H of num_rendered
 *       RAST/rasterize_points.cu:28-34, RAST/cuda_rasterizer/rasterizer_impl.cu:277-279,290-292,336-340
 *   f3dg_splat_head
 *       the post-network half of GaussianSplatPredictor_gtunet.forward   src/gaussian_predictor.py:857-881, 961-1007
 *       (python-only in the reference; there is no native FFI for it -- this is the build's fused kernel)
 *   f3dg_render_epilogue
 *       the torch post-processing of render_predicted_more_v2_gof  src/gaussian_renderer/__init__.py:881-909, 1043-1053
 *
 * Memory ownership mirrors the reference: outputs and the workspace are allocated and owned by the caller
 * (torch tensors on the Python side). Instead of growing buffers through callbacks in the middle of the call
 * (which forces the reference's host sync, rasterizer_impl.cu:336), the caller passes ONE workspace sized by
 * f3dg_workspace_bytes() for a chosen instance capacity `max_rendered`; if the (Gaussian, tile) instance count
 * of a call exceeds that capacity the device sets an overflow flag, the compositing stage is skipped, and
 * f3dg_read_status()/f3dg_forward() report F3DG_ERR_OVERFLOW together with the capacity that would have
 * sufficed, so the caller can grow the workspace and retry -- the same contract as the resize callback, moved
 * outside the launch sequence. The workspace layout is the forward<->backward contract (as the three byte
 * buffers are in the reference, rasterizer_impl.cu:444-446) and is private to the library.
 */
#ifndef F3DG_H_INCLUDED
#define F3DG_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F3DG_OK              0
#define F3DG_ERR_BAD_ARG    -1   /* NULL where data is required, non-positive sizes, both/neither of sh & colours ... */
#define F3DG_ERR_WORKSPACE  -2   /* workspace_bytes smaller than f3dg_workspace_bytes(...) */
#define F3DG_ERR_OVERFLOW   -3   /* more (Gaussian, tile) instances than max_rendered; see f3dg_read_status */
#define F3DG_ERR_HIP        -4   /* a HIP runtime call failed; f3dg_last_error() has the text */
#define F3DG_ERR_UNSUPPORTED -5  /* NUM_CHANNELS != 3 etc. (rasterizer_impl.cu:294-297) */
#define F3DG_ERR_STATE      -6   /* f3dg_backward on a workspace whose last forward was not a F3DG_FLAG_SAVE_AUX call: nothing was
                                    walked, every gradient is zero (reported by f3dg_backward_pairs / f3dg_read_status) */

/* flags for f3dg_forward_batched */
#define F3DG_FLAG_SAVE_AUX   1u  /* keep final_T / n_contrib / conic / clamped for f3dg_backward (training mode).
                                    Without it the compositing kernel writes only the 9 output channels. */
#define F3DG_FLAG_BG_PER_VIEW 2u /* background is [n_views,3] instead of [3] */
/* Output-channel mask of inference calls (not part of the reference's API: visualize.py:304-306, 400-402 consume only `render`,
 * `rendered_depth` and `rendered_alpha`, and the build's own batched loops -- cycle aggregation, orbit frames -- ask for just those).
 * With BOTH flags the compositing kernel neither accumulates nor writes the view-space normal (channels 3..5) and the distortion
 * (channel 8): those planes of out_color are left untouched, every other channel is bit-identical to the 9-channel call. Ignored
 * with F3DG_FLAG_SAVE_AUX and by the kernels of one- or two-view launches (which then write all nine). */
#define F3DG_FLAG_SKIP_NORMAL 4u
#define F3DG_FLAG_SKIP_DISTORTION 8u
/* Per-call selection of what f3dg_set_option sets as process-wide DEFAULTS (round 5). The reference hands every knob of a call through
 * GaussianRasterizationSettings_GOF (RAST/diff_gof_rasterization/__init__.py:168-182); these bits do the same for the knobs this build
 * adds, so that two streams / threads can render with different settings at the same time:
 *   F3DG_FLAG_EXACT          the compositing of this call runs the reference's float32 / float64 operation order, whatever "render_fast" says
 *                            (a consumer of the distortion channel -- 3-17 % relative off in fast arithmetic -- asks for it per call);
 *   F3DG_FLAG_FAST           ... runs the fast arithmetic, also with F3DG_FLAG_SAVE_AUX (the backward repeats the forward's alpha from the
 *                            workspace header, so the pair stays consistent); EXACT wins when both are given;
 *   F3DG_FLAG_NO_TILE_CULL   the reference's tile lists (every tile of the 3-sigma square, forward.cu:364-374) instead of the culled ones;
 *   F3DG_FLAG_NO_SMALL_PATH  the general launch sequence also for the shapes the three-launch small-call path serves;
 *   F3DG_FLAG_SCAN           (inference calls in the fast arithmetic, general launch sequence) the split-pixel compositing schedule of
 *                            csrc/f3dg_render5.hip: the entries of the pixels that hold a quadrant back are blended by helper lanes and
 *                            combined by a segmented wave scan. The set of blended entries per pixel is unchanged, the sums are associated
 *                            differently: every channel within the 1e-4 of the fast arithmetic against the reference, NOT bit-identical
 *                            to a call without the flag. Ignored with F3DG_FLAG_EXACT / F3DG_FLAG_SAVE_AUX (option "render_scan" sets the
 *                            process default: -1 flag only, 1 every eligible call, 0 never). */
#define F3DG_FLAG_EXACT 16u
#define F3DG_FLAG_FAST 32u
#define F3DG_FLAG_NO_TILE_CULL 64u
#define F3DG_FLAG_NO_SMALL_PATH 128u
#define F3DG_FLAG_SCAN 256u

#define F3DG_TILE 16             /* BLOCK_X = BLOCK_Y = 16, RAST/cuda_rasterizer/config.h:16-17 */
#define F3DG_OUT_CHANNELS 9      /* RGB, normal xyz, median depth, alpha, distortion: auxiliary.h:21-24 */
#define F3DG_DEPTH_OFFSET 6      /* auxiliary.h:21 */
#define F3DG_ALPHA_OFFSET 7      /* auxiliary.h:22 */
#define F3DG_DISTORTION_OFFSET 8 /* auxiliary.h:23 */

/* Library / build identification: "f3dg-hip gfx950 <version>" */
const char* f3dg_version(void);
/* Text of the last HIP error seen by this thread's calls (empty string if none). */
const char* f3dg_last_error(void);

/* Bytes of workspace needed for n_views views of P Gaussians at W x H with room for max_rendered
 * (Gaussian, tile) instances summed over all views of the call. Pure host arithmetic. */
size_t f3dg_workspace_bytes(int P, int W, int H, int n_views, long long max_rendered);

/* Batched forward: renders n_views views of the SAME P Gaussians in one launch sequence, no host sync.
 *   viewmatrix, projmatrix : [n_views,16]  (row-vector convention, i.e. already transposed; auxiliary.h:86-115)
 *   cam_pos                : [n_views,3]
 *   background             : [3] or [n_views,3] with F3DG_FLAG_BG_PER_VIEW
 *   shs [P,M,3] xor colors_precomp [P,3]; (scales [P,3] and rotations [P,4]) xor cov3D_precomp [P,6];
 *   view2gaussian_precomp  : NULL or [n_views,P,10]
 *   out_color              : [n_views,9,H,W]   radii: [n_views,P] int32 or NULL
 * Returns F3DG_OK or a negative error for host-detectable problems. Instance overflow is reported by
 * f3dg_read_status(). P == 0 is legal: outputs are filled with background / zeros (rasterize_points.cu:85). */
int f3dg_forward_batched(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                         int n_views, int P, int D, int M,
                         const float* background, int W, int H,
                         const float* means3D, const float* shs, const float* colors_precomp,
                         const float* opacities, const float* scales, float scale_modifier,
                         const float* rotations, const float* cov3D_precomp,
                         const float* view2gaussian_precomp,
                         const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                         float tan_fovx, float tan_fovy, float kernel_size,
                         float* out_color, int* radii, unsigned flags);

/* The same for n_sets DIFFERENT Gaussian sets of equal size in one launch sequence -- the image-batched form of the cycle
 * aggregation loop (reference visualize.py:293-314 renders 8 views of each of B images with B x 8 separate rasterizer calls):
 * means3D ... rotations are [n_sets, P, ...], the cameras [n_sets * views_per_set, ...] (set-major), out_color
 * [n_sets * views_per_set, 9, H, W], radii [n_sets * views_per_set, P]; view i renders set i / views_per_set.
 * f3dg_forward_batched is the n_sets = 1 case. Workspace: f3dg_workspace_bytes(P, W, H, n_sets * views_per_set, max_rendered).
 * n_sets > 1 is an inference path: with F3DG_FLAG_SAVE_AUX or view2gaussian_precomp it returns F3DG_ERR_BAD_ARG (f3dg_backward
 * indexes the Gaussian inputs of ONE set). */
int f3dg_forward_sets(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                      int n_sets, int views_per_set, int P, int D, int M,
                      const float* background, int W, int H,
                      const float* means3D, const float* shs, const float* colors_precomp,
                      const float* opacities, const float* scales, float scale_modifier,
                      const float* rotations, const float* cov3D_precomp,
                      const float* view2gaussian_precomp,
                      const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                      float tan_fovx, float tan_fovy, float kernel_size,
                      float* out_color, int* radii, unsigned flags);

/* BLOCKING. Waits for `stream`, then reads the workspace header written by the last forward on it.
 * h_num_rendered: total instances the call needed; returns F3DG_OK, or F3DG_ERR_OVERFLOW if that exceeded
 * the capacity the workspace was sized for (outputs of that call are then undefined). */
int f3dg_read_status(void* stream, const void* workspace, long long* h_num_rendered);

/* The same status WITHOUT blocking the caller -- what is left of the reference's blocking copy (rasterizer_impl.cu:336) once a loop
 * renders one view per call and never looks at num_rendered (visualize.py:293-314, 387-416).
 * f3dg_status_post: enqueues on `stream`, behind the forward just issued on it, a copy of the workspace header into pinned host
 *   memory owned by the library, and an event. Returns a ticket >= 0 (or a negative error). The workspace may be reused by the next
 *   call on the same stream at once: the copy is ordered before it.
 * f3dg_status_poll: F3DG_PENDING while the copy has not landed (wait == 0); otherwise -- after waiting for it if wait != 0 --
 *   F3DG_OK / F3DG_ERR_OVERFLOW as f3dg_read_status, *h_num_rendered as there, and the ticket is released. A caller that defers the
 *   check this way must be able to re-issue the call it belongs to (outputs of an overflowed call are undefined). */
#define F3DG_PENDING 1
int f3dg_status_post(void* stream, const void* workspace);
int f3dg_status_poll(int ticket, int wait, long long* h_num_rendered);

/* Reference-shaped single-view forward (Rasterizer::forward): f3dg_forward_batched(n_views = 1) followed by
 * f3dg_read_status(). BLOCKING, like the reference (rasterizer_impl.cu:336). Returns num_rendered >= 0 or a
 * negative error; on F3DG_ERR_OVERFLOW *h_needed (if not NULL) receives the required capacity. */
long long f3dg_forward(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                       int P, int D, int M, const float* background, int W, int H,
                       const float* means3D, const float* shs, const float* colors_precomp,
                       const float* opacities, const float* scales, float scale_modifier,
                       const float* rotations, const float* cov3D_precomp, const float* view2gaussian_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                       float tan_fovx, float tan_fovy, float kernel_size, int prefiltered,
                       float* out_color, int* radii, unsigned flags, long long* h_needed);

/* Backward of a forward made with F3DG_FLAG_SAVE_AUX on the same workspace (same P, W, H, n_views,
 * max_rendered). Argument meaning follows Rasterizer::backward (rasterizer.h:57-90). The reference's binding zero-fills every
 * dL_* tensor (rasterize_points.cu:160-170); here the PER-GAUSSIAN sums (dL_dopacity, dL_dmean3D, dL_dsh, dL_dscale, dL_drot) are
 * added into and must be zero-filled (or hold running sums) by the caller, the PER-VIEW outputs (dL_dmean2D, dL_dcolor,
 * dL_dview2gaussian: 64 floats per (view, Gaussian), 2 GB at BASELINE C5) are written in full by the call and may be handed over
 * uninitialised; dL_dconic and dL_dcov3D are never touched (the reference leaves them zero). Sizes per view v:
 *   dL_dpix [n_views,9,H,W] in;  dL_dmean2D [n_views,P,3], dL_dconic [n_views,P,4] (stays 0), dL_dopacity [P],
 *   dL_dcolor [n_views,P,3], dL_dmean3D [P,3], dL_dcov3D [P,6] (stays 0), dL_dsh [P,M,3], dL_dscale [P,3],
 *   dL_drot [P,4], dL_dview2gaussian [n_views,P,10].
 * Per-Gaussian parameter gradients (opacity, mean3D, sh, scale, rot) are summed over the views of the call. */
int f3dg_backward(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                  int n_views, int P, int D, int M, const float* background, int W, int H,
                  const float* means3D, const float* shs, const float* colors_precomp,
                  const float* scales, float scale_modifier, const float* rotations,
                  const float* cov3D_precomp, const float* view2gaussian_precomp,
                  const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                  float tan_fovx, float tan_fovy, float kernel_size,
                  const int* radii, const float* dL_dpix,
                  float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                  float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                  float* dL_dview2gaussian, unsigned flags /* F3DG_FLAG_BG_PER_VIEW as in the forward */);

/* Gaussians -> points integration: Rasterizer::integrate (rasterizer.h:92-123, rasterizer_impl.cu:530-792) as bound by
 * IntegrateGaussiansToPointsCUDA (rasterize_points.cu:233-343, `_C.integrate_gaussians_to_points`). One view.
 *   points3D [PN,3]; the Gaussian arguments as in f3dg_forward; subpixel_offset [H,W,2] is accepted and ignored (the
 *   reference only loads it into an unused variable, forward.cu:838).
 *   out_color [9,H,W]: 0..2 colour + T * background of the centre ray, 3..5 zero, 6 maximal ray depth, 7 alpha,
 *       8 the number of points that fell into the pixel (forward.cu:984-996, 1194-1196)
 *   out_alpha_integrated [PN] (1 for points outside the frustum / image), out_color_integrated [PN,3] (0 for those)
 * The library writes every output element itself (the reference relies on the caller's torch.full fills). P == 0 or
 * PN == 0: out_color = 0, alpha = 1, colour = 0, returns 0 (rasterize_points.cu:300).
 * BLOCKING like f3dg_forward; returns num_rendered >= 0 or a negative error; on F3DG_ERR_OVERFLOW *h_needed (if not
 * NULL) receives the required instance capacity. The workspace must hold f3dg_integrate_workspace_bytes(). */
size_t f3dg_integrate_workspace_bytes(int P, int PN, int W, int H, long long max_rendered);
long long f3dg_integrate(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                         int PN, int P, int D, int M, const float* background, int W, int H,
                         const float* points3D, const float* means3D, const float* shs, const float* colors_precomp,
                         const float* opacities, const float* scales, float scale_modifier,
                         const float* rotations, const float* cov3D_precomp, const float* view2gaussian_precomp,
                         const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                         float tan_fovx, float tan_fovy, float kernel_size, const float* subpixel_offset,
                         int prefiltered, float* out_color, int* radii, float* out_alpha_integrated,
                         float* out_color_integrated, long long* h_needed);

/* The two halves of f3dg_integrate, for callers that integrate MANY point sets against the same Gaussians and camera
 * (the mesh-extraction sweep of visualize.py:449-505 runs 9 point sets through each of 129 cameras): everything that does
 * not depend on the points -- projection, binning and the five-ray per-pixel pass with its contributor table -- is done
 * once by f3dg_integrate_prepare and stays in the workspace (sized with PN_max = the largest point set that will follow);
 * f3dg_integrate_points then costs one lane per point. out_color is written by prepare (channels 0..7) and read by
 * points (which also writes channel 8). out_alpha_integrated / out_color_integrated may be NULL; alpha_min, if not NULL,
 * is updated as alpha_min[i] = min(alpha_min[i], alpha_integrated[i]) (torch.min semantics, visualize.py:463). */
long long f3dg_integrate_prepare(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                                 int PN_max, int P, int D, int M, const float* background, int W, int H,
                                 const float* means3D, const float* shs, const float* colors_precomp,
                                 const float* opacities, const float* scales, float scale_modifier,
                                 const float* rotations, const float* cov3D_precomp, const float* view2gaussian_precomp,
                                 const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                 float tan_fovx, float tan_fovy, float kernel_size, float* out_color, int* radii,
                                 long long* h_needed);
int f3dg_integrate_points(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                          int PN, int P, int W, int H, const float* points3D, const float* viewmatrix,
                          float tan_fovx, float tan_fovy, float* out_color, float* out_alpha_integrated,
                          float* out_color_integrated, float* alpha_min);

/* The mesh extraction integrates its point sets through 129 cameras of the SAME Gaussians (visualize.py:449-505): n_views cameras
 * prepared in ONE launch sequence (projection, binning and the per-pixel pass with grid = n_views x tiles: a single 256^2 camera is
 * 256 workgroups, one wave per SIMD of an MI355X). viewmatrix / projmatrix [n_views,16], cam_pos [n_views,3], out_color
 * [n_views,9,H,W], radii [n_views,P] or NULL; max_rendered is the capacity for the instances of all cameras together.
 * Workspace: f3dg_integrate_workspace_bytes_batched. Every per-camera result is bit-identical to f3dg_integrate_prepare of that
 * camera. Returns the total instance count or a negative error (BLOCKING, as f3dg_integrate_prepare). */
size_t f3dg_integrate_workspace_bytes_batched(int P, int PN, int W, int H, int n_views, long long max_rendered);
long long f3dg_integrate_prepare_batched(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                                         int n_views, int PN_max, int P, int D, int M, const float* background, int W, int H,
                                         const float* means3D, const float* shs, const float* colors_precomp,
                                         const float* opacities, const float* scales, float scale_modifier,
                                         const float* rotations, const float* cov3D_precomp,
                                         const float* view2gaussian_precomp, const float* viewmatrix,
                                         const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                                         float kernel_size, float* out_color, int* radii, long long* h_needed);
/* One point set against camera `view` of a batched preparation (viewmatrix [16] and out_color [9,H,W] of THAT camera). */
int f3dg_integrate_points_view(void* stream, void* workspace, size_t workspace_bytes, long long max_rendered,
                               int n_views, int view, int PN, int P, int W, int H, const float* points3D,
                               const float* viewmatrix, float tan_fovx, float tan_fovy, float* out_color,
                               float* out_alpha_integrated, float* out_color_integrated, float* alpha_min);

/* Diagnostic, BLOCKING: how many (camera, tile) pairs of the last preparation on this workspace reached the reference's limit of 1,024
 * contributors in some pixel (forward.cu:972-976) and were therefore computed by the per-pixel kernel behind the shared-ray kernel
 * (f3dg_integrate.hip: integrate_pass1_rays_kernel). Same shape arguments as the preparation. */
int f3dg_debug_integrate_redo(void* stream, const void* workspace, int P, int PN_max, int W, int H, int n_views, long long max_rendered,
                              int* h_tiles);

/* present[i] = (view-space z of means3D[i] > 0.2), auxiliary.h:177-202. present is uint8 [P]. */
int f3dg_mark_visible(void* stream, int P, const float* means3D, const float* viewmatrix,
                      const float* projmatrix, uint8_t* present);

/* Cycle-aggregative projection ("splat head", src/gaussian_predictor.py:857-881, 961-1007) for B images:
 *   net_out [B,23,H,W] planar: offset 0:3, opacity 3, scaling 4:7, rotation 7:11, features_dc 11:14, features_rest 14:23
 *   depth [B,1,H,W], ray_dirs [3,H,W] (predictor buffer), view_to_world [B,16] (row-vector convention), cam_quat [B,4]
 * Outputs are written for image b at Gaussian offset `n_offset` of destination tensors whose per-image length is
 * `n_total` (>= n_offset + H*W): this is what lets the cycle loop aggregate the 9 passes in place instead of the
 * reference's torch.cat chain (visualize.py:336-340):
 *   xyz [B,n_total,3], opacity [B,n_total,1], scaling [B,n_total,3], rotation [B,n_total,4],
 *   features_dc [B,n_total,1,3], features_rest [B,n_total,3,3], unet_depth [B,n_total,1]
 * squre_clip: clamp |x|,|y| only when < 10 (gaussian_predictor.py:972-974). */
int f3dg_splat_head(void* stream, int B, int H, int W, const float* net_out, const float* depth,
                    const float* ray_dirs, const float* view_to_world, const float* cam_quat, float squre_clip,
                    long long n_total, long long n_offset,
                    float* xyz, float* opacity, float* scaling, float* rotation,
                    float* features_dc, float* features_rest, float* unet_depth);

/* Fused epilogue of render_predicted_more_v2_gof (gaussian_renderer/__init__.py:881-909, 1043-1053) for n_views
 * rendered frames raster [n_views,9,H,W]:
 *   normal_world [n_views,3,H,W] = c2w[:3,:3] @ normalize(raster[3:6]);  c2w = inverse(world_view^T), given as
 *       c2w_rot [n_views,9] row-major 3x3 (host-computed, it is a 4x4 inverse)
 *   depth_normal [n_views,3,H,W]: central-difference normal of the back-projected median depth, border = 0;
 *       needs c2w [n_views,16] row-major 4x4 and fx, fy.
 * Either output pointer may be NULL to skip it. */
int f3dg_render_epilogue(void* stream, int n_views, int H, int W, const float* raster,
                         const float* c2w, float fx, float fy,
                         float* normal_world, float* depth_normal);
/* The same from the world_view matrices themselves ([n_views,16], row-vector convention, as render_predicted_more_v2_gof receives
 * them): every workgroup inverts world_view^T in float64 (cofactors), so the caller needs no matrix inverse per call. */
int f3dg_render_epilogue_view(void* stream, int n_views, int H, int W, const float* raster,
                              const float* world_view, float fx, float fy,
                              float* normal_world, float* depth_normal);

/* 8-bit RGB frames for the video writer and the multi-GPU gather: dst [n_frames,H,W,3] uint8 =
 * (uint8)(255 * clamp(src[:, 0:3], 0, 1)) with src [n_frames,src_channels,H,W] float32 planar (src_channels = 9 for the
 * rasterizer output), i.e. what visualize.py:407,416 computes on the host per frame. */
int f3dg_pack_frames(void* stream, int n_frames, int H, int W, int src_channels, const float* src, unsigned char* dst);
/* The same frames written directly into PINNED HOST memory (hipHostMalloc / torch pin_memory; 16-byte aligned; anything else is
 * F3DG_ERR_BAD_ARG): the frames of visualize.py:407,416 arrive in host memory with one kernel and no copy command. HIP executes a
 * device -> pinned-host hipMemcpyAsync as a whole-chip shader copy, which competes with whatever renders next; this kernel uses at most
 * max_workgroups workgroups (0: 64) -- the transfer is PCIe-bound either way. Visible to the host after the stream (or an event
 * recorded behind the call) is synchronised. */
int f3dg_pack_frames_host(void* stream, int n_frames, int H, int W, int src_channels, const float* src, unsigned char* dst_host,
                          int max_workgroups);

/* Hand-off of the cycle aggregation (reference visualize.py:311, 331-333): from the rendered rasters [B * V, 9, H, W] (frame
 * b * V + v) to the next predictor inputs, view-major: xin [V, B, 4, H, W] = cat(clamp(rgb, 0, 1), alpha) and
 * depth [V, B, 1, H, W] = the median depth. One kernel; H * W must be a multiple of 4 and the pointers 16-byte aligned. */
int f3dg_cycle_inputs(void* stream, int B, int V, int H, int W, const float* raster, float* xin, float* depth);

/* Fused GroupNorm (+ SiLU when apply_silu != 0) of the predictor's SongUNet backbone (src/gaussian_predictor.py:250-262 and the
 * `silu(norm(x))` of its residual blocks, :318-323): x, y [N,C,HW] float32 contiguous (NCHW), weight / bias [C],
 * statistics per (sample, group) over C/groups x HW values, biased variance, y = (x - mean) / sqrt(var + eps) * weight + bias.
 * x == y (in place) is allowed. SURVEY.md 8f-3. */
int f3dg_group_norm_silu(void* stream, int N, int C, int HW, int groups, const float* x, const float* weight,
                         const float* bias, float eps, int apply_silu, float* y);
/* The same with bfloat16 activations in and out (the bf16 option of the backbone); weight / bias and all arithmetic float32,
 * the moments float64. */
int f3dg_group_norm_silu_bf16(void* stream, int N, int C, int HW, int groups, const uint16_t* x, const float* weight,
                              const float* bias, float eps, int apply_silu, uint16_t* y);
/* The same for channels-last activations (the backbone's "nhwc" layout option: MIOpen's NHWC convolution kernels without the layout
 * transposes around them): x, y [N,HW,C] contiguous, C a multiple of 4 (float32) / 8 (bfloat16) and at most 1024;
 * `moments` is scratch of f3dg_group_norm_nhwc_scratch_bytes(N, HW, groups) bytes: one (sum, sum of squares) pair per workgroup, sample
 * and group, added up in workgroup order by a second-stage kernel -- no atomics, the statistics are bit-reproducible from run to run
 * (round 5; until then the workgroups added into 8 atomic slots in arrival order). Same statistics, same formula.
 * `moments_bytes` = the size of the buffer behind `moments` (round 6: the scratch has grown with the kernel once -- a buffer smaller than
 * f3dg_group_norm_nhwc_scratch_bytes() is refused with F3DG_ERR_WORKSPACE instead of being written past its end). */
size_t f3dg_group_norm_nhwc_scratch_bytes(int N, int HW, int groups);
int f3dg_group_norm_silu_nhwc(void* stream, int N, int C, int HW, int groups, const float* x, const float* weight,
                              const float* bias, float eps, int apply_silu, float* y, double* moments, size_t moments_bytes);
int f3dg_group_norm_silu_nhwc_bf16(void* stream, int N, int C, int HW, int groups, const uint16_t* x, const float* weight,
                                   const float* bias, float eps, int apply_silu, uint16_t* y, double* moments, size_t moments_bytes);
/* The four GroupNorm entry points with the bias of the convolution that produced x folded in: y = gn(x + pre_bias[c]) (pre_bias [C]
 * float32, may be null). PyTorch-ROCm applies a convolution's bias as a separate elementwise pass behind MIOpen's kernel
 * (src/gaussian_predictor.py:163-178 `x = conv2d(x, w) ; x = x.add_(b)`); the residual blocks hand it to the GroupNorm that follows. */
int f3dg_group_norm_silu_pb(void* stream, int N, int C, int HW, int groups, const float* x, const float* pre_bias, const float* weight,
                            const float* bias, float eps, int apply_silu, float* y);
int f3dg_group_norm_silu_pb_bf16(void* stream, int N, int C, int HW, int groups, const uint16_t* x, const float* pre_bias, const float* weight,
                                 const float* bias, float eps, int apply_silu, uint16_t* y);
int f3dg_group_norm_silu_nhwc_pb(void* stream, int N, int C, int HW, int groups, const float* x, const float* pre_bias, const float* weight,
                                 const float* bias, float eps, int apply_silu, float* y, double* moments, size_t moments_bytes);
int f3dg_group_norm_silu_nhwc_pb_bf16(void* stream, int N, int C, int HW, int groups, const uint16_t* x, const float* pre_bias,
                                      const float* weight, const float* bias, float eps, int apply_silu, uint16_t* y, double* moments, size_t moments_bytes);
/* ... and with IEEE float16 activations in and out (the fp16 option of the backbone, round 5: the bf16 MFMA rate with 10 mantissa bits
 * instead of 7; every activation of the backbone is GroupNorm-bounded). Weight / bias and all arithmetic float32, the moments float64. */
int f3dg_group_norm_silu_pb_f16(void* stream, int N, int C, int HW, int groups, const uint16_t* x, const float* pre_bias, const float* weight,
                                const float* bias, float eps, int apply_silu, uint16_t* y);
int f3dg_group_norm_silu_nhwc_pb_f16(void* stream, int N, int C, int HW, int groups, const uint16_t* x, const float* pre_bias,
                                     const float* weight, const float* bias, float eps, int apply_silu, uint16_t* y, double* moments, size_t moments_bytes);
/* The residual join of a backbone block (src/gaussian_predictor.py:325-327: the second convolution's bias, `x = x + skip(orig)` with the
 * skip convolution's bias, `x = x * skip_scale`) as ONE elementwise pass: y = ((a + bias_a[c]) + (b + bias_b[c])) * scale in float32,
 * the reference's order. a, b, y [N,C,HW] (nhwc = 0) or [N,HW,C] (nhwc = 1), 16-byte aligned, N*C*HW a multiple of 4 (float32) / 8
 * (bfloat16) -- and C as well when nhwc = 1; either bias may be null; y may alias a or b. */
int f3dg_residual_join(void* stream, int N, int C, int HW, int nhwc, const float* a, const float* bias_a, const float* b,
                       const float* bias_b, float scale, float* y);
int f3dg_residual_join_bf16(void* stream, int N, int C, int HW, int nhwc, const uint16_t* a, const float* bias_a, const uint16_t* b,
                            const float* bias_b, float scale, uint16_t* y);
int f3dg_residual_join_f16(void* stream, int N, int C, int HW, int nhwc, const uint16_t* a, const float* bias_a, const uint16_t* b,
                           const float* bias_b, float scale, uint16_t* y);

/* Runtime switches: process-wide DEFAULTS for what a call's own flags do not say (every arithmetic / list / path choice of a call has a
 * flag, above). The library knows sixteen names and one diagnostic pair; everything else returns F3DG_ERR_BAD_ARG.
 *   "render_fast"   (default 1) arithmetic of the compositing forward: 0 = the reference's float32 / float64 operation order, 1 = error-free
 *                   float32 pairs for the float64 island and FMA-contracted accumulations downstream of alpha in inference calls (no
 *                   auxiliary planes; within 3e-7 of mode 0), exact in calls a backward follows, 2 = fast in every call.
 *   "tile_cull"     (default 1) a Gaussian is instantiated only in the tiles that the box of its conservative alpha >= 1/255 ellipse reaches
 *                   instead of every tile of the reference's 3-sigma square (forward.cu:364-374). The dropped (Gaussian, tile) pairs are a
 *                   bare `continue` for every pixel of the tile: outputs and gradients are unchanged (asserted on every scene of the
 *                   tests); num_rendered and the exported lists are SHORTER than the reference's. 0 = the reference's lists, bit for bit.
 *                   f3dg_integrate always uses the reference's lists.
 *   "small_path"    (default 1) calls of one or two views of at most 2^18 Gaussians on at most 1,024 tiles take the three-launch path of
 *                   f3dg_small.hip (2 = on, and forget the shapes an earlier overflow disabled); "small_path_aux" (default 1): forwards
 *                   with F3DG_FLAG_SAVE_AUX too -- f3dg_backward then walks the per-tile slots.
 *   "render_lowocc" (default 1; n > 1: up to n x 1,024 quadrant waves) compositing launches of at most 2,048 quadrant waves (one or two
 *                   256^2 views: every wave alone on its SIMD, its time a chain of latencies) take the multi-wave kernels of
 *                   f3dg_render4.hip, bit-identical to the general kernel; 0 = the general kernel for every launch.
 *   "render_split"  (default -1: by arithmetic) those kernels: 1 = render3p_fwd_kernel, a producer wave scans the list, gathers the records
 *                   and runs phase 1 of the next window while the consumer wave composites (the default in fast arithmetic, and for two
 *                   views); 2 / 3 = render3q_fwd_kernel, consumer + 2 / 3 evaluator waves (the stateless part of every pair, parked in
 *                   LDS) + producer (3: the default for one view in the reference's arithmetic).
 *   "render_unroll" (default -1: 2 in fast arithmetic, else 1) entries per phase-2 trip of render3p (1 or 2).
 *   "render_pack"   (default -1) the rank-packed kernel (f3dg_render4.hip; bit-identical to render3s_fwd_kernel, also its auxiliary planes):
 *                   -1 = launches in the reference's arithmetic (-38 % on pixel-aligned predicted sets, -6 % at C2), 1 = every launch,
 *                   0 = none; "render_pack_th" (default 32; 0: none, 64: all): the trips of a slide in which at most that many pixels take
 *                   part are packed.
 *   "render_scan"   (default -1) the split-pixel schedule (f3dg_render5.hip): -1 = the calls that carry F3DG_FLAG_SCAN, 1 = every fast
 *                   inference launch of the general path, 0 = never; "render_scan_th" (default 12; 64: every pending entry goes through
 *                   dense batches): fused trips while more than that many pixels take part.
 *   "bwd_dense"     (default 1) the compositing backward: 1 = render5_bwd_kernel (f3dg_backward5.hip: entry-major batches of (pixel, entry) pairs,
 *                   segmented scans, one 128-byte accumulation record per (view, Gaussian)), 0 = render3_bwd_kernel (the lock-step walk of
 *                   rounds 2-5); "bwd_occ" (default 5): waves per SIMD the lock-step kernel is compiled for (2..6).
 *   "render_scan_min" (default 4) split-pixel mode: stragglers that hold fewer older-half entries than this finish the slide in fused trips.
 *   diagnostics: "render_count" (default 0) swaps in the counting variants of the compositing kernels (f3dg_debug_render_counts /
 *                   _render4_counts / _render5_counts); "time_launches" (default 0): f3dg_debug_launch_times.
 * A library built with -DF3DG_LAB (`F3DG_LAB=1 python f3d-gaus_amd/build.py --force`; f3dg_version() then ends in "lab") also compiles the
 * kernel generations and schedules that were measured and superseded -- the round-1 pixel-lane kernel and its filters ("render_kernel",
 * "render_pretest", "render_cull", "render_queue"), render2 ("render_round"), render3 with fixed windows ("render_slide", "render_dma"),
 * render3l ("render_split" 0), the tail schedule ("render_tail"), four-wave workgroups ("render_wpb"), "render_lds_pad",
 * "sort_wide_groups", "sort_fused_rects", "pre_hoist", "pre_order", "small_debug", "debug_skip_all", the staging replay ("render_replay")
 * -- for the bit-identity tests against the plain transcription and for A/B runs; NOTES.md has their measurements. */
int f3dg_set_option(const char* name, int value);

/* Diagnostic: number of kernels this library has launched from this process since the last reset (host-side counter, every
 * launch site counts; memsets / copies do not). bench.py reports launches per call from it. */
long long f3dg_debug_launch_count(int reset);
/* Diagnostic: with option "time_launches" = 1 every launch site measures the host time of its hipLaunchKernelGGL; this prints the
 * per-site averages to stderr (tools/prof_small.py). */
int f3dg_debug_launch_times(int reset);

/* Optional per-stage timing of the forward path with HIP events recorded on the caller's stream (this is what
 * bench.py uses for the live roofline figure). f3dg_profile_enable(1) makes every following
 * f3dg_forward_batched on this host thread record 4 events; f3dg_profile_collect() is BLOCKING, sums the
 * milliseconds of all calls recorded since the last collect into h_stage_ms[3] = {projection (preprocess),
 * binning (scan + key duplication + radix sort + tile ranges), compositing} and returns the call count; h_stage_ms holds FIVE
 * doubles: [3] and [4] are the compositing backward and the per-Gaussian backward of the f3dg_backward calls in between. */
int f3dg_profile_enable(int on);
int f3dg_profile_collect(double* h_stage_ms, int* h_calls);
/* The same, plus the three forward stage times of every recorded call in call order: h_per_call[max_calls][3] (may be NULL). */
int f3dg_profile_collect_calls(double* h_stage_ms, int* h_calls, double* h_per_call, int max_calls);
/* Diagnostics of the compositing forward: the kernel (with its template arguments) the last launch of this process used, and -- with
 * f3dg_set_option("render_count", 1), which swaps in a counting variant of the one-wave kernel -- its work counters summed over the
 * launches since the last reset: h_out8[SIXTEEN] = { list entries staged, list entries scanned, phase-2 trips, slides, lane-trips, waves,
 * trips of slides that began with <= 8 / <= 24 live pixels, those slides, 0 ... }. */
const char* f3dg_debug_last_render_kernel(void);
int f3dg_debug_render_counts(unsigned long long* h_out8, int reset);
/* The counters of the rank-packed kernel's counting variant (render_kernel = 4 with render_count = 1): h_out[SIXTEEN] = { list entries
 * staged, scanned, fused trips, slides, lane-trips of fused trips, waves, packed batches (= dense trips), blend trips of the batches,
 * pairs evaluated in dense trips, pairs that reached a blend trip, 0 ... }. */
int f3dg_debug_render4_counts(unsigned long long* h_out, int reset);
/* The same for render5_fwd_kernel (F3DG_FLAG_SCAN): h_out[16] = { staged, scanned, fused trips, slides, lane-trips of fused trips, waves,
 * dense batches, pairs in dense batches, pixels compacted, slides with a compaction, 0... }. */
int f3dg_debug_render5_counts(unsigned long long* h_out, int reset);
/* Debug: shader clocks of the roles of the small-launch pipeline kernel (render3q_fwd_kernel, options render_split = 2 / 3 and
 * render_count = 1), summed over the launches since the last reset of f3dg_debug_render4_counts. h_out[4][16]: rows { consumer,
 * evaluators, producer } x { total, waiting at the window barrier, waiting for a round counter, windows, rounds, waves }; row 3 =
 * the longest { consumer, evaluator, producer } wave of any workgroup. */
int f3dg_debug_render3q_clocks(unsigned long long* h_out);
/* Diagnostic (tools/pmc_pass1.sh): resident workgroups per CU of the two pass-1 kernels of f3dg_integrate as the runtime computes it. */
int f3dg_debug_pass1_occupancy(int* rays_blocks, int* cull_blocks);

/* BLOCKING: number of contributing (pixel, Gaussian) pairs the last f3dg_backward on this workspace blended back through --
 * "C" of the byte formula 80 R + 60 W H + 68 C of the compositing backward. */
int f3dg_backward_pairs(void* stream, const void* workspace, long long* h_pairs);

/* Test/inspection hook: device-to-device copies of the library's internal per-call state into caller buffers
 * (any pointer may be NULL). Used by the stage-wise parity tests to pin each kernel separately, the way the
 * oracle exposes GeometryState / BinningState / ImageState (rasterizer_impl.cu:188-243).
 *   rec [V*P*16] (view2gaussian[10], opacity*coef, pre-test threshold, rgb[3], culling-ellipse c; after an INFERENCE call only the rows of
 *   (view, Gaussian) pairs that are in a tile list are written -- the others are stale); depths [V*P], means2D [V*P*2], conic [V*P*4],
 *   tiles [V*P], offsets [V*P], clamped [V*P] (bit c = channel c clamped), keys_sorted [cap] u64 (all SAVE_AUX: an inference call does not write them),
 *   point_list [cap], ranges [V*T*2], final_T [V*4*H*W] and n_contrib [V*2*H*W] (SAVE_AUX).
 * Asking for a SAVE_AUX-only plane of a workspace whose last forward was an inference call returns F3DG_ERR_BAD_ARG (BLOCKING then:
 * the header is read back); rec, point_list (Gaussian ids, quadrant masks stripped) and ranges are always available. */
int f3dg_debug_export(void* stream, const void* workspace, int P, int W, int H, int n_views,
                      long long max_rendered, float* rec, float* means2D, float* conic, unsigned* tiles,
                      unsigned* offsets, unsigned char* clamped, unsigned long long* keys_sorted,
                      unsigned* point_list, unsigned* ranges, float* final_T, unsigned* n_contrib, float* depths);

/* Inspection hook of the compositing kernel: shader-clock cycles per wave, summed over all waves of all launches since the
 * last reset, h_out8 = {barrier waits, staging, list build, phase 1, phase 2, repack, total, waves}. Only a library built
 * with -DF3DG_TIMING (tools/render_timing.sh) counts; the product build returns zeros. BLOCKING (device-to-host copy). */
int f3dg_debug_timing(unsigned long long* h_out8, int reset);

#ifdef __cplusplus
}
#endif
#endif /* F3DG_H_INCLUDED */
