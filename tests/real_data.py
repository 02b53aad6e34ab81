"""The real image of fixture F6 (images/1/n01644373_4548.jpg + its LeReS depth, as the reference's dataset class yields them at 256 x 256:
tests/golden/real_image_256.npz, written by tests/tools/gen_real_golden.py) through THIS build's predictor and cycle aggregation.
Used by `bench.py --data real`, tools/culled_fraction.py and the tests; data only travels (the reference does not)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "tests", "golden", "real_image_256.npz")


def load_real_image(device):
    g = np.load(PATH)
    images = torch.from_numpy(g["images_u8"].astype(np.float32) / 255.0).unsqueeze(0).to(device)      # [1,3,256,256] as ToTensor gives it
    depth = torch.from_numpy(g["depth"]).unsqueeze(0).to(device)                                       # [1,1,256,256], z_near .. z_near + 2
    return images, depth, g


def real_predictor(device, res=256, backbone="fp32"):
    """The build's Unet_GS_gtunet with the weights of the fixture generator: formula-defined everywhere except the last 1x1
    convolution, which keeps the reference's own initialisation (scale exp(log 0.01), opacity bias 0: see gen_real_golden.py)."""
    import f3dgaus_amd as f3d
    from f3dgaus_amd import cameras
    from helpers_weights import formula_state_dict
    g = np.load(PATH)
    cfg = cameras.default_cfg(res)
    cfg['model']['opacity_bias'] = 0.0
    cfg['model']['backbone_dtype'] = backbone
    model = f3d.Unet_GS_gtunet(cfg, renderer=None).eval()
    sd = model.state_dict()
    keep = {k: v for k, v in sd.items() if k.split(".")[-1] in ("ray_dirs", "sh_to_v_transform", "v_to_sh_transform") or k.endswith("resample_filter")}
    new = formula_state_dict({k: tuple(v.shape) for k, v in sd.items()}, keep=keep)
    for k in new:
        if k.endswith("network_with_offset.out.weight"):
            new[k] = torch.from_numpy(g["out_weight"])
        if k.endswith("network_with_offset.out.bias"):
            new[k] = torch.from_numpy(g["out_bias"])
    model.load_state_dict(new)
    return model.to(device), cfg


def real_merged_set(device, backbone="fp32"):
    """The 9 x 65,536 merged Gaussians of the real image (predict -> 8 novel views -> 8 re-predictions, cycle.cycle_aggregate):
    dict of [589824, ...] tensors with the keys of `synthetic.make_gaussians`."""
    import f3dgaus_amd as f3d
    model, cfg = real_predictor(device, 256, backbone)
    images, depth, _ = load_real_image(device)
    with torch.no_grad():
        merged = f3d.cycle.cycle_aggregate(model, images, depth, cfg)
    return {k: merged[k][0].contiguous() for k in ("xyz", "scaling", "rotation", "opacity", "features_dc", "features_rest")}
