"""Generates tests/golden/real_F6.npz and tests/golden/real_image_256.npz: SURVEY 8c fixture F6, "a real pipeline slice".

Build container only (the reference does not travel). What runs here, all of it the reference's OWN Python on CPU torch:
  * the dataset class src/dataio_gs_test_256_demo.ImagenetGS_Dataset_test_256_demo on /root/reference/images/1 (sorted: item 0 is
    n01644373_4548.jpg with its LeReS depth PNG) at image_size = 64: image loading, depth scaling (:152-175) and the canonical camera
    (:78-133). torchvision is not installed in this image; the three transforms the class composes (Resize(size, LANCZOS) on a PIL
    image = PIL.Image.resize, numpy.array, ToTensor = HWC -> CHW with uint8 scaled by 1/255 and other dtypes kept) are provided as a
    small stand-in module around PIL -- torchvision's documented behaviour for exactly these three calls, stated here because it is
    the one part of the slice that is not the reference's (or a pinned dependency's) own code;
  * the predictor src/unet_gs.Unet_GS_gtunet (SongUNet + splat head) with formula-defined weights (tests/helpers_weights.py; the
    185 MB checkpoint does not travel) -- EXCEPT its last 1x1 convolution, which keeps the reference's own initialisation
    (gaussian_predictor.py:573-580: per-group xavier gains and the biases xyz 0, scale log(0.01), rotation / colour 0), because that
    layer sets the magnitude of every predicted quantity: formula weights there give splats up to 50 world units wide, the
    reference's initialisation gives the sigma ~ 0.01 (1.5 px at 256^2) its training starts from. One override, stated: the opacity
    bias is 0 (opacity ~ 0.5) instead of the untrained -3 (opacity 0.047, no pixel would ever saturate). The layer's 23 x 23 + 23
    numbers are stored in the fixture. Called exactly as visualize.py:282-283 calls it -> 4,096 pixel-aligned Gaussians on the
    real depth map;
  * the 8-view orbit of visualize.py:236-279 (from the reference-generated tests/golden/cameras.npz);
  * src/gaussian_renderer.render_predicted_more_v2_gof for orbit views 1 and 5, its `diff_gof_rasterization` import satisfied by
    the plain-C oracle (the CUDA extension cannot be built or loaded here).
real_F6.npz holds the inputs (image, depth), the predicted Gaussians and, per view, the oracle's raster [9,64,64], radii and
num_rendered plus the wrapper's post-processed maps. real_image_256.npz holds the SAME image and depth as the dataset class yields
them at image_size = 256 (uint8 RGB, uint16-range depth as float16-safe integers): the input of `bench.py --data real` on the GPU
box, where neither the reference nor its images exist."""
import copy
import os
import sys
import types

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402
from gen_cycle_golden import OracleRasterizer, Settings  # noqa: E402
from helpers_weights import formula_state_dict  # noqa: E402

npy = lambda t: t.detach().cpu().numpy()


def install_pil_transforms():
    """torchvision.transforms for the three calls of the dataset class (see the module docstring)."""
    from PIL import Image

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Resize:
        def __init__(self, size, interpolation=Image.BILINEAR):
            self.size, self.interpolation = size, interpolation

        def __call__(self, img):
            w, h = img.size
            if isinstance(self.size, int):      # shorter side -> size, aspect kept
                if w <= h:
                    nw, nh = self.size, int(self.size * h / w)
                else:
                    nw, nh = int(self.size * w / h), self.size
            else:
                nh, nw = self.size
            return img if (nw, nh) == (w, h) else img.resize((nw, nh), self.interpolation)

    class ToTensor:
        def __call__(self, pic):
            a = np.asarray(pic)
            if a.ndim == 2:
                a = a[:, :, None]
            t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
            return t.to(torch.float32).div(255) if a.dtype == np.uint8 else t

    class ToPILImage:
        def __call__(self, x):
            raise NotImplementedError("not on the path of this fixture")

    tr = sys.modules['torchvision.transforms']
    tr.Compose, tr.Resize, tr.ToTensor, tr.ToPILImage = Compose, Resize, ToTensor, ToPILImage


def main():
    ref_import.install(Settings, OracleRasterizer)
    install_pil_transforms()
    cfg = copy.deepcopy(yaml.safe_load(open(os.path.join(ref_import.REF, "config/imagenetgs_256x256_v1.yaml"))))
    cams = np.load(os.path.join(ROOT, "tests", "golden", "cameras.npz"))          # reference-generated (gen_golden.py)
    out_dir = os.path.join(ROOT, "tests", "golden")
    with ref_import.Cuda2Cpu(), torch.no_grad():
        import src.dataio_gs_test_256_demo as dio
        import src.gaussian_renderer as gr
        import src.unet_gs as ugs

        folder = os.path.join(ref_import.REF, "images", "1")
        # ---- the image as the dataset class yields it at 256^2 (the input of bench.py --data real)
        ds256 = dio.ImagenetGS_Dataset_test_256_demo(folder, image_size=256, config=cfg, random_flip=False)
        d256 = ds256[0]
        assert d256["name"] == "n01644373_4548.jpg", d256["name"]
        img_u8 = np.round(npy(d256["images"]) * 255).astype(np.uint8)                       # exact: ToTensor divided uint8 by 255
        assert np.array_equal(img_u8.astype(np.float32) / 255, npy(d256["images"]))
        real256 = dict(name=np.array(d256["name"]), images_u8=img_u8, depth=npy(d256["depth"]).astype(np.float32),
                       z_near=np.float32(cfg['dataset_params']['z_near']))

        # ---- the 64^2 slice
        res = 64
        cfg['model']['training_resolution'] = res
        cfg['model']['opacity_bias'] = 0.0          # (see the module docstring)
        ds = dio.ImagenetGS_Dataset_test_256_demo(folder, image_size=res, config=cfg, random_flip=False)
        data = ds[0]
        images, depth = data["images"].unsqueeze(0).float(), data["depth"].unsqueeze(0).float()      # [1,3,64,64], [1,1,64,64]
        torch.manual_seed(0)
        model = ugs.Unet_GS_gtunet(cfg=cfg, renderer=None).eval()
        sd = model.state_dict()
        keep = {k: v for k, v in sd.items() if k.split(".")[-1] in ("ray_dirs", "sh_to_v_transform", "v_to_sh_transform") or k.endswith("resample_filter")
                or k.endswith("network_with_offset.out.weight") or k.endswith("network_with_offset.out.bias")}
        out_w = [v for k, v in sd.items() if k.endswith("network_with_offset.out.weight")][0].clone()
        out_b = [v for k, v in sd.items() if k.endswith("network_with_offset.out.bias")][0].clone()
        model.load_state_dict(formula_state_dict({k: tuple(v.shape) for k, v in sd.items()}, keep=keep))
        background = torch.zeros(1, 3)
        # visualize.py:228, 281-283
        input_feat = images.unsqueeze(1)
        input_feat = torch.cat([input_feat, torch.ones_like(input_feat[:, :, 0:1, :, :])], 2)
        cano_v2w = ds.view_to_world_transforms.expand([1, -1, -1]).unsqueeze(1).contiguous()
        cano_quat = ds.source_cv2wT_quat.expand([1, -1, -1]).contiguous()
        _, _, gsb = model(input_feat, background, cano_v2w, cano_quat, return_3d_features=True, render=False,
                          squre_clip=cfg['opt']['squre_clip'], unet_depth=depth)
        P = gsb["xyz"].shape[1]
        assert P == res * res
        views = [1, 5]
        per_view = {}
        for v in views:
            wv = torch.from_numpy(cams["o8_wv"][v:v + 1])
            fp = torch.from_numpy(cams["o8_fp"][v:v + 1])
            cc = torch.from_numpy(cams["o8_cc"][v:v + 1])
            r = gr.render_predicted_more_v2_gof(gsb, 0, wv, fp, cc, background[0:1], cfg)
            raster = torch.cat([r["render"], r["rendered_normal"] * 0, r["rendered_depth"], r["rendered_alpha"], r["distortion_map"]], 0)
            # the oracle's full 9-channel raster again (channels 3..5 are the view-space normals before the wrapper rotates them)
            from oracle import gof
            o = gof.Oracle()
            shs = torch.cat([gsb["features_dc"][0], gsb["features_rest"][0]], 1).contiguous()
            full, radii, R = o.forward(means3D=npy(gsb["xyz"][0]), opacities=npy(gsb["opacity"][0]), viewmatrix=npy(wv).reshape(4, 4),
                                       projmatrix=npy(fp).reshape(4, 4), campos=npy(cc).reshape(3),
                                       tanfovx=np.tan(cfg['model']['fov'] * np.pi / 360), tanfovy=np.tan(cfg['model']['fov'] * np.pi / 360),
                                       W=res, H=res, bg=[0, 0, 0], shs=npy(shs), scales=npy(gsb["scaling"][0]),
                                       rotations=npy(gsb["rotation"][0]), sh_degree=cfg['model']['max_sh_degree'])
            assert np.array_equal(full[:3], npy(raster[:3])) and np.array_equal(full[6:], npy(raster[6:]))
            inter = o.intermediates()
            lens = (inter["ranges"][..., 1] - inter["ranges"][..., 0]).reshape(-1)
            print("view", v, "num_rendered", R, "R/P %.2f" % (R / P), "tile list mean %.0f max %d" % (lens.mean(), lens.max()),
                  "alpha mean %.3f" % full[7].mean())
            per_view.update({f"v{v}_wv": npy(wv), f"v{v}_fp": npy(fp), f"v{v}_cc": npy(cc), f"v{v}_raster": full, f"v{v}_radii": radii,
                             f"v{v}_num_rendered": np.int64(R), f"v{v}_rendered_normal": npy(r["rendered_normal"]),
                             f"v{v}_depth_normal": npy(r["depth_normal"])})
        out = os.path.join(out_dir, "real_F6.npz")
        np.savez_compressed(out, name=np.array(data["name"]), images=npy(images), depth=npy(depth), views=np.array(views),
                            fov=np.float64(cfg['model']['fov']), sh_degree=np.int64(cfg['model']['max_sh_degree']),
                            cano_v2w=npy(cano_v2w), cano_quat=npy(cano_quat), out_weight=npy(out_w), out_bias=npy(out_b),
                            **{"g_" + k: npy(t[0]) for k, t in gsb.items() if isinstance(t, torch.Tensor)}, **per_view)
        np.savez_compressed(os.path.join(out_dir, "real_image_256.npz"), out_weight=npy(out_w), out_bias=npy(out_b), **real256)
        for f in ("real_F6.npz", "real_image_256.npz"):
            print(f, os.path.getsize(os.path.join(out_dir, f)))
        print("scaling range", float(gsb["scaling"].min()), float(gsb["scaling"].max()), "opacity mean", float(gsb["opacity"].mean()),
              "depth range", float(depth.min()), float(depth.max()))


if __name__ == "__main__":
    main()
