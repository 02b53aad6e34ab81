"""``Unet_GS_gtunet``: the operator wrapper of the predictor (reference src/unet_gs.py:34-101), same constructor
``(cfg, renderer)``, same ``forward`` signature and return convention ``(x|None, depth|None, gaussian_splat_batch)``
(or ``x`` when ``return_3d_features`` is False). The optional in-module render loop calls ``self.renderer`` with the
exact argument pattern of src/unet_gs.py:82-87, so the reference's ``render_predicted_more_v2_gof`` (or this
package's) plugs in unchanged."""
import torch
from torch import nn

from .gaussian_predictor import GaussianSplatPredictor_gtunet


class Unet_GS_gtunet(nn.Module):
    def __init__(self, cfg, renderer):
        super().__init__()
        self.gaussian_predictor = GaussianSplatPredictor_gtunet(cfg)
        self.renderer = renderer
        self.cfg = cfg

    def forward(self, x_input, background, view_to_world_transforms, source_cv2wT_quat, return_3d_features=True,
                render=False, return_depth=False, squre_clip=10000.0, world_view_transforms=None,
                full_proj_transforms=None, camera_centers=None, config=None, image_size=None, unet_depth=None,
                out=None, n_offset=0):
        extra = {} if out is None else dict(out=out, n_offset=n_offset)
        gaussian_splats = self.gaussian_predictor(x_input, view_to_world_transforms, source_cv2wT_quat,
                                                  focals_pixels=None, return_depth=return_depth,
                                                  squre_clip=squre_clip, unet_depth=unet_depth, **extra)
        # the reference makes a contiguous copy of every entry (src/unet_gs.py:75); the fused kernel already
        # writes contiguous tensors, so .contiguous() is a no-op here
        gaussian_splat_batch = {k: v.contiguous() for k, v in gaussian_splats.items()}

        if render:
            bs = background.shape[0]
            x_novel, depth_novel = [], []
            for b in range(bs):
                output_dic = self.renderer(gaussian_splat_batch, b, world_view_transforms[b:b + 1].contiguous(),
                                           full_proj_transforms[b:b + 1].contiguous(),
                                           camera_centers[b:b + 1].contiguous(), background[b:b + 1].contiguous(),
                                           config)
                x_novel += [output_dic["render"].reshape(-1, 3, image_size, image_size)]
                depth_novel += [output_dic["rendered_depth"].reshape(-1, 1, image_size, image_size)]
            x = torch.concat(x_novel, dim=0)
            depth = torch.concat(depth_novel, dim=0)
        else:
            x, depth = None, None

        if return_3d_features:
            return x, depth, gaussian_splat_batch
        return x
