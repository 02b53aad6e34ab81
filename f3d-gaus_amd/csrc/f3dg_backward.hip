// f3dg_backward.hip -- backward of the compositing and projection stages (placeholder until the kernels land).
#include "f3dg_common.h"

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
                             float* dL_dview2gaussian)
{
    return F3DG_ERR_UNSUPPORTED;
}
