"""Generates tests/golden/cycle_loop.npz: the cycle-aggregation loop of the reference, run by the reference's OWN Python.

Build container only (the reference does not travel). What runs here:
  * the reference's predictor module (src/unet_gs.py, src/gaussian_predictor.py) on CPU torch with formula-defined weights
    (tests/helpers_weights.py; the 185 MB checkpoint does not travel either), at training_resolution = 32;
  * the reference's renderer wrapper src/gaussian_renderer.render_predicted_more_v2_gof, whose `diff_gof_rasterization` import is
    satisfied by a stand-in module backed by the plain-C oracle (the CUDA extension cannot be built or loaded here);
  * the loop itself: lines 224-340 of /root/reference/visualize.py are read from where they lie and executed as they are
    (textwrap.dedent + exec) in a namespace that provides the names the script defines earlier (model, data, dataset, config ...).
    Nothing of it is copied into this repository.
The fixture holds the inputs (8 images + depths @32x32) and, for images 0 and 5, the merged Gaussian set (9 x 1024 Gaussians) and
the 8 intermediate renders (RGB, alpha, median depth)."""
import copy
import os
import sys
import textwrap
import types
from typing import NamedTuple

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402
from helpers_weights import formula_state_dict  # noqa: E402
from oracle import gof  # noqa: E402

npy = lambda t: t.detach().cpu().numpy()


class Settings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    kernel_size: float
    subpixel_offset: torch.Tensor
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class OracleRasterizer(torch.nn.Module):
    """`GaussianRasterizer_GOF` stand-in: the plain-C oracle behind the reference's call signature (rast_py:201-239)."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, view2gaussian_precomp=None):
        rs = self.raster_settings
        out, radii, _ = gof.Oracle().forward(
            means3D=npy(means3D), opacities=npy(opacities), viewmatrix=npy(rs.viewmatrix).reshape(4, 4),
            projmatrix=npy(rs.projmatrix).reshape(4, 4), campos=npy(rs.campos).reshape(3), tanfovx=rs.tanfovx, tanfovy=rs.tanfovy,
            W=rs.image_width, H=rs.image_height, bg=npy(rs.bg).reshape(-1)[:3], shs=None if shs is None else npy(shs),
            colors_precomp=None if colors_precomp is None else npy(colors_precomp), scales=npy(scales), rotations=npy(rotations),
            sh_degree=rs.sh_degree, scale_modifier=rs.scale_modifier, kernel_size=rs.kernel_size)
        return torch.from_numpy(out), torch.from_numpy(radii)


def main():
    ref_import.install(Settings, OracleRasterizer)
    cfg = yaml.safe_load(open(os.path.join(ref_import.REF, "config/imagenetgs_256x256_v1.yaml")))
    cfg = copy.deepcopy(cfg)
    res = 32
    cfg['model']['training_resolution'] = res
    cams = np.load(os.path.join(ROOT, "tests", "golden", "cameras.npz"))          # reference-generated (gen_golden.py)
    with ref_import.Cuda2Cpu(), torch.no_grad():
        import src.camera as cam
        import src.dataio_gs_test_256_demo as dio
        import src.gaussian_renderer as gr
        import src.unet_gs as ugs
        import src.utils as U

        torch.manual_seed(0)
        model = ugs.Unet_GS_gtunet(cfg=cfg, renderer=None).eval()
        sd = model.state_dict()
        keep = {k: v for k, v in sd.items() if k.split(".")[-1] in ("ray_dirs", "sh_to_v_transform", "v_to_sh_transform") or k.endswith("resample_filter")}
        model.load_state_dict(formula_state_dict({k: tuple(v.shape) for k, v in sd.items()}, keep=keep))

        B = 8
        g = torch.Generator().manual_seed(3)
        images = torch.rand(B, 3, res, res, generator=g)
        depth = torch.rand(B, 1, res, res, generator=g) * 2 + 6.667
        dataset = types.SimpleNamespace(
            projection_matrix=torch.from_numpy(cams["projection"]), inverse_first_camera=torch.from_numpy(cams["inv_first"]),
            view_to_world_transforms=torch.from_numpy(cams["c_v2w"]), source_cv2wT_quat=torch.from_numpy(cams["c_quat"]))
        ns = dict(torch=torch, np=np, model=model, data={"images": images, "depth": depth}, device=torch.device("cpu"), config=cfg,
                  args=types.SimpleNamespace(output_path="unused"), step=1, dataset=dataset, image_size=res,
                  TensorGroup=U.TensorGroup, sample_front_circle_gs=U.sample_front_circle_gs,
                  compute_cam2world_matrix=cam.compute_cam2world_matrix, update_camera_pose=dio.update_camera_pose,
                  matrix_to_quaternion=dio.matrix_to_quaternion, render_predicted_more_v2_gof=gr.render_predicted_more_v2_gof,
                  tqdm=lambda it: it)
        lines = open(os.path.join(ref_import.REF, "visualize.py")).read().splitlines()
        assert lines[223].strip().startswith("bs = data['images'].shape[0]") and lines[341].strip() == "# re-define rendering views"
        exec(compile(textwrap.dedent("\n".join(lines[223:341])), "/root/reference/visualize.py:224-341", "exec"), ns)
        merged, rendered_8, alpha_8, depth_8 = (ns[k] for k in ("gaussian_splat_batch_merge", "rendered_8", "alpha_8", "depth_8"))
        assert merged["xyz"].shape == (B, 9 * res * res, 3), merged["xyz"].shape
        sel = [0, 5]
        out = os.path.join(ROOT, "tests", "golden", "cycle_loop.npz")
        np.savez_compressed(out, images=npy(images), depth=npy(depth), sel=np.array(sel), rendered_8=npy(rendered_8[sel]),
                            alpha_8=npy(alpha_8[sel]), depth_8=npy(depth_8[sel]), **{"m_" + k: npy(v[sel]) for k, v in merged.items()})
        print(out, os.path.getsize(out), {k: tuple(v.shape) for k, v in merged.items()})
        print("scaling range", float(merged["scaling"].min()), float(merged["scaling"].max()), "opacity mean", float(merged["opacity"].mean()),
              "alpha mean of the 8 renders", float(alpha_8.mean()))


if __name__ == "__main__":
    main()
