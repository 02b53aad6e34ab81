"""Generates the committed golden fixtures under tests/golden/ by IMPORTING the Python half of the reference
(/root/reference) on CPU torch. Runs only in the build container (the reference does not travel); the fixtures are
data only: inputs and the reference's outputs.

  cameras.npz       canonical camera, 8-view orbit, 1+128-view orbit as visualize.py builds them
  splat_head.npz    GaussianSplatPredictor_gtunet.forward with the network replaced by a fixed 23-channel map
                    (B=2, 32x32, two oblique cameras) -> the 7 output tensors (a10)
  songunet.npz      the reference backbone with formula-defined weights on a 1x4x32x32 input -> 23-channel output,
                    plus the state_dict key list / shapes (appendix C, checkpoint compatibility)
  renderer_post.npz render_predicted_more_v2_gof's own post-processing (world normals, depth_to_normal) run on a fixed
                    9-channel raster through a stand-in rasterizer module that returns that raster (a9)
  sh_cov.npz        utils/sh_utils.eval_sh (deg 0..3) and the python covariance builder
                    (scene/gaussian_model.py:27-31, utils/general_utils.py:78-110) on random inputs
"""
import copy
import math
import os
import sys
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

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
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


FIXED = {}


class FixedRasterizer(torch.nn.Module):
    """Stand-in for the CUDA extension: returns a pre-computed raster so that the reference's renderer wrapper
    (settings construction, argument plumbing, post-processing) runs end to end on CPU."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings
        FIXED["settings"] = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, view2gaussian_precomp=None):
        FIXED["call"] = dict(means3D=means3D, shs=shs, opacities=opacities, scales=scales, rotations=rotations)
        return FIXED["raster"].clone(), FIXED["radii"].clone()


def main():
    ref_import.install(Settings, FixedRasterizer)
    cfg = yaml.safe_load(open(os.path.join(ref_import.REF, "config/imagenetgs_256x256_v1.yaml")))
    with ref_import.Cuda2Cpu(), torch.no_grad():
        import src.gaussian_predictor as gp
        import src.camera as cam
        import src.utils as U
        import src.gaussian_renderer as gr
        import src.dataio_gs_test_256_demo as dio

        # ------------------------------------------------------------------ cameras (visualize.py:236-279, 343-381)
        def build(num_frames, yaw_diff, pitch_diff):
            params = U.TensorGroup(angles=torch.zeros(1, 3), fov=torch.ones(1) * cfg['model']['fov'],
                                   radius=torch.ones(1) * cfg['model']['radius'], look_at=torch.zeros(1, 3))
            params.look_at[:, 2] = cfg['model']['look_at']
            samples = U.sample_front_circle_gs(params, num_frames, fov_diff=0.0, yaw_diff=yaw_diff, pitch_diff=pitch_diff)
            return torch.inverse(cam.compute_cam2world_matrix(samples))

        # canonical camera: the block of dataio_gs_test_256_demo.py:78-133, driven through its own helper functions
        cparams = U.TensorGroup(angles=torch.zeros(1, 3), radius=torch.ones(1, 1) * cfg['model']['radius'],
                                look_at=torch.zeros(1, 3))
        cparams.look_at[:, 2] = cfg['model']['look_at']
        cam2w = torch.inverse(cam.compute_cam2world_matrix(cparams))
        Rt = torch.inverse(cam2w)
        c_wv = Rt.transpose(1, 2)
        c_v2w = cam2w.transpose(1, 2)
        c_cc = c_wv.inverse()[:, 3, :3]
        fov = cfg['model']['fov']
        projection = dio.getProjectionMatrix(znear=cfg["dataset_params"]["z_near"], zfar=cfg["dataset_params"]["z_far"],
                                             fovX=fov * 2 * np.pi / 360, fovY=fov * 2 * np.pi / 360).transpose(0, 1)
        c_fp = c_wv.bmm(projection.unsqueeze(0))
        c_wv, c_v2w, c_fp, c_cc, inv_first = dio.update_camera_pose(c_wv, c_v2w, c_fp, c_cc, None, first=True)
        c_quat = dio.matrix_to_quaternion(c_v2w[0, :3, :3].transpose(0, 1)).unsqueeze(0)

        def assemble(cam2w, num_quat, quat_before):
            Rt = torch.inverse(cam2w).contiguous()
            wv = Rt.transpose(1, 2).unsqueeze(1).contiguous()
            v2w = cam2w.transpose(1, 2).unsqueeze(1).contiguous()
            cc = wv.inverse()[:, :, 3, :3].contiguous()
            pm = projection.expand([cam2w.shape[0], -1, -1]).unsqueeze(1).contiguous()
            fp = (wv[:, 0].bmm(pm[:, 0])).unsqueeze(1).contiguous()
            quat = torch.zeros_like(fp[:, 0:1, 0, :])
            if quat_before:
                for i in range(num_quat):
                    quat[i] = dio.matrix_to_quaternion(v2w[i, 0, :3, :3].transpose(0, 1).contiguous())
            wv, v2w, fp, cc, _ = dio.update_camera_pose(wv, v2w, fp, cc, inv_first, False)
            if not quat_before:
                for i in range(num_quat):
                    quat[i] = dio.matrix_to_quaternion(v2w[i, 0, :3, :3].transpose(0, 1).contiguous())
            return wv, v2w, fp, cc, quat

        o8 = assemble(build(8, 0.25, 0.15), 8, False)
        cam2w129 = torch.cat([build(1, 0.0, 0.0), build(128, 0.25, 0.15)], 0)
        o129 = assemble(cam2w129, 128, True)
        np.savez_compressed(
            os.path.join(OUT, "cameras.npz"), projection=npy(projection), inv_first=npy(inv_first),
            c_wv=npy(c_wv), c_v2w=npy(c_v2w), c_fp=npy(c_fp), c_cc=npy(c_cc), c_quat=npy(c_quat),
            **{f"o8_{n}": npy(t) for n, t in zip(("wv", "v2w", "fp", "cc", "quat"), o8)},
            **{f"o129_{n}": npy(t) for n, t in zip(("wv", "v2w", "fp", "cc", "quat"), o129)})

        # ------------------------------------------------------------------ splat head (gaussian_predictor.py:883-1007)
        cfg32 = copy.deepcopy(cfg)
        cfg32['model']['training_resolution'] = 32
        torch.manual_seed(0)
        pred = gp.GaussianSplatPredictor_gtunet(cfg32).eval()
        g = torch.Generator().manual_seed(1)
        B, res = 2, 32
        net_out = torch.randn(B, 23, res, res, generator=g) * 0.5
        net_out[:, 4:7] = net_out[:, 4:7] * 0.3 + math.log(0.01)
        depth = torch.rand(B, 1, res, res, generator=g) * 2.0 + 6.667
        v2w = torch.stack([o8[1][2, 0], o8[1][5, 0]])              # two oblique cameras
        quat = torch.stack([o8[4][2, 0], o8[4][5, 0]])
        class _Fixed(torch.nn.Module):       # replaces the U-Net by a fixed 23-channel map
            def forward(self, x, **kw):
                return net_out
        pred.network_with_offset = _Fixed()
        x_dummy = torch.zeros(B, 1, 4, res, res)
        out = pred(x_dummy, v2w.unsqueeze(1), quat.unsqueeze(1), unet_depth=depth)
        np.savez_compressed(os.path.join(OUT, "splat_head.npz"), net_out=npy(net_out), depth=npy(depth), v2w=npy(v2w),
                            quat=npy(quat), ray_dirs=npy(pred.ray_dirs), **{"out_" + k: npy(v) for k, v in out.items()})
        # a clamped variant (squre_clip < 10)
        out_c = pred(x_dummy, v2w.unsqueeze(1), quat.unsqueeze(1), unet_depth=depth, squre_clip=0.3)
        np.savez_compressed(os.path.join(OUT, "splat_head_clip.npz"), out_xyz=npy(out_c["xyz"]))

        # ------------------------------------------------------------------ SongUNet (appendix C)
        torch.manual_seed(0)
        pred_full = gp.GaussianSplatPredictor_gtunet(cfg).eval()
        sd = pred_full.state_dict()
        new_sd = formula_state_dict({k: tuple(v.shape) for k, v in sd.items()}, keep={k: v for k, v in sd.items() if k in ("ray_dirs", "sh_to_v_transform", "v_to_sh_transform") or k.endswith("resample_filter")})
        pred_full.load_state_dict(new_sd)
        gi = torch.Generator().manual_seed(3)
        x = torch.rand(1, 4, 32, 32, generator=gi)
        y = pred_full.network_with_offset(x, film_camera_emb=None, N_views_xa=1)
        keys = sorted(sd.keys())
        np.savez_compressed(os.path.join(OUT, "songunet.npz"), x=npy(x), y=npy(y), keys=np.array(keys),
                            shapes=np.array([",".join(map(str, sd[k].shape)) for k in keys]),
                            n_elements=np.array(sum(v.numel() for v in sd.values())))

        # ------------------------------------------------------------------ renderer wrapper post-processing (a9)
        from oracle import gof
        import f3dgaus_amd  # noqa: F401  (synthetic scene only; no HIP call)
        from f3dgaus_amd import synthetic
        cfg64 = copy.deepcopy(cfg)
        cfg64['model']['training_resolution'] = 64
        gs = synthetic.make_gaussians(3000, s0=0.05, seed=11)
        wv, fp, cc = o8[0][2:3], o8[2][2:3], o8[3][2:3]          # [1,1,4,4], [1,1,4,4], [1,1,3] as visualize.py passes
        tanfov = math.tan(cfg['model']['fov'] * np.pi / 360)
        shs = torch.cat([gs["features_dc"], gs["features_rest"]], 1)
        orc = gof.Oracle()
        raster, radii, R = orc.forward(means3D=npy(gs["xyz"]), opacities=npy(gs["opacity"]), viewmatrix=npy(wv), projmatrix=npy(fp),
                                       campos=npy(cc), tanfovx=tanfov, tanfovy=tanfov, W=64, H=64, bg=[0, 0, 0], shs=npy(shs),
                                       scales=npy(gs["scaling"]), rotations=npy(gs["rotation"]), sh_degree=1)
        FIXED["raster"], FIXED["radii"] = torch.from_numpy(raster), torch.from_numpy(radii)
        pc = {k: v.unsqueeze(0).repeat(2, *([1] * v.ndim)) for k, v in gs.items()}
        bg = torch.zeros(1, 3)
        res_d = gr.render_predicted_more_v2_gof(pc, 1, wv, fp, cc, bg, cfg64)
        st = FIXED["settings"]
        np.savez_compressed(
            os.path.join(OUT, "renderer_post.npz"), raster=raster, radii=radii, wv=npy(wv), fp=npy(fp), cc=npy(cc),
            tanfovx=np.float64(st.tanfovx), sh_degree=np.int64(st.sh_degree), image_height=np.int64(st.image_height),
            shs_passed=npy(FIXED["call"]["shs"]),
            **{"out_" + k: npy(v) for k, v in res_d.items() if isinstance(v, torch.Tensor)},
            **{"g_" + k: npy(v) for k, v in gs.items()})

        # ------------------------------------------------------------------ python SH / covariance utilities
        sys.path.insert(0, os.path.join(ref_import.REF, "src/gaussian-splatting"))
        from utils.sh_utils import eval_sh
        from utils.general_utils import build_scaling_rotation, strip_symmetric
        g5 = torch.Generator().manual_seed(5)
        n = 256
        means = torch.randn(n, 3, generator=g5) * 2
        campos = torch.randn(3, generator=g5)
        sh = torch.randn(n, 16, 3, generator=g5) * 0.5                         # [P, M, 3] (rasterizer layout)
        dirs = means - campos
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        sh_res = {f"sh_deg{d}": npy(eval_sh(d, sh.transpose(1, 2), dirs)) for d in range(4)}     # [..., C, coeffs]
        scales = torch.exp(torch.randn(n, 3, generator=g5) * 0.5 - 3)
        rot = torch.randn(n, 4, generator=g5)
        rot = rot / rot.norm(dim=1, keepdim=True)
        Lm = build_scaling_rotation(0.7 * scales, rot)
        cov = strip_symmetric(Lm @ Lm.transpose(1, 2))
        np.savez_compressed(os.path.join(OUT, "sh_cov.npz"), means=npy(means), campos=npy(campos), sh=npy(sh),
                            scales=npy(scales), rot=npy(rot), scale_modifier=np.float32(0.7), cov3D=npy(cov), **sh_res)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
