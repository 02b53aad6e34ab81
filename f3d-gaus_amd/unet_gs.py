"""``Unet_GS_gtunet``: the operator wrapper around the predictor (reference src/unet_gs.py:34-101). The contract it keeps: constructor
``(cfg, renderer)``; ``forward`` with the reference's argument names and order; the return convention ``(frames | None, depths | None,
gaussians)`` -- or the frames alone when ``return_3d_features`` is False --; and, with ``render=True``, one ``self.renderer`` call per
image of the batch with the argument pattern of src/unet_gs.py:82-87 (image index, that image's camera rows as [1, ...] slices, its
background row, ``config``), so the reference's ``render_predicted_more_v2_gof`` or this package's plugs in unchanged. Extension:
``out`` / ``n_offset`` hand preallocated merged buffers to the predictor (cycle aggregation in place)."""
import torch
from torch import nn

from .gaussian_predictor import GaussianSplatPredictor_gtunet


def _row(t, b):
    """Row b of a per-image tensor as a contiguous [1, ...] slice (what the renderer wrappers index with [0])."""
    return t[b:b + 1].contiguous()


class Unet_GS_gtunet(nn.Module):
    def __init__(self, cfg, renderer):
        super().__init__()
        self.gaussian_predictor = GaussianSplatPredictor_gtunet(cfg)
        self.renderer = renderer
        self.cfg = cfg

    def _render_each_image(self, gaussians, n_images, world_view_transforms, full_proj_transforms, camera_centers, background, config, size):
        """The in-module render loop: every image's Gaussians through ``self.renderer`` from that image's camera; frames and median
        depths of all images stacked along the batch axis."""
        per_image = [self.renderer(gaussians, b, _row(world_view_transforms, b), _row(full_proj_transforms, b), _row(camera_centers, b),
                                   _row(background, b), config) for b in range(n_images)]
        frames = torch.cat([o["render"].reshape(-1, 3, size, size) for o in per_image], dim=0)
        depths = torch.cat([o["rendered_depth"].reshape(-1, 1, size, size) for o in per_image], dim=0)
        return frames, depths

    def forward(self, x_input, background, view_to_world_transforms, source_cv2wT_quat, return_3d_features=True,
                render=False, return_depth=False, squre_clip=10000.0, world_view_transforms=None,
                full_proj_transforms=None, camera_centers=None, config=None, image_size=None, unet_depth=None,
                out=None, n_offset=0):
        in_place = {} if out is None else {"out": out, "n_offset": n_offset}
        predicted = self.gaussian_predictor(x_input, view_to_world_transforms, source_cv2wT_quat, focals_pixels=None,
                                            return_depth=return_depth, squre_clip=squre_clip, unet_depth=unet_depth, **in_place)
        # (the reference copies every entry to a contiguous tensor, src/unet_gs.py:75; the splat-head kernel writes them contiguous, so
        # this is a no-op that keeps the guarantee for any predictor)
        gaussians = {name: value.contiguous() for name, value in predicted.items()}

        frames = depths = None
        if render:
            frames, depths = self._render_each_image(gaussians, background.shape[0], world_view_transforms, full_proj_transforms,
                                                     camera_centers, background, config, image_size)
        return (frames, depths, gaussians) if return_3d_features else frames
