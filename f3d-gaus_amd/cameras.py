"""Camera helpers that feed the hot path.

Host-side 4x4 bookkeeping in torch, restating
  src/utils.py:64-90            sample_front_circle_gs   (orbit: yaw = -yd*sin(2*pi*s), pitch = pd*cos(2*pi*s))
  src/camera.py:17-32, 65-91    spherical2cartesian, compute_cam2world_matrix
  src/dataio_gs_test_256_demo.py:237-260   getProjectionMatrix
  src/dataio_gs_test_256_demo.py:262-297   matrix_to_quaternion (dup of src/gaussian_predictor.py:68-103)
  src/dataio_gs_test_256_demo.py:300-374   update_camera_pose
  src/dataio_gs_test_256_demo.py:78-133    canonical camera set-up
  visualize.py:236-279, 343-381            how the loop assembles the per-view matrices
All matrices use the reference's ROW-vector convention: ``p_view = [p, 1] @ world_view``; the flat memory of
``world_view`` is therefore the column-major W2C the device code indexes (auxiliary.h:86-94).
The reference inverts ``compute_cam2world_matrix`` twice (visualize.py:249-255); that is reproduced
literally, so that the golden camera fixtures match to rounding.
"""
import math
from typing import NamedTuple, Optional

import numpy as np
import torch


class CameraSet(NamedTuple):
    """Per-view camera tensors, shaped as visualize.py passes them ([V,1,4,4] / [V,1,3] / [V,1,4])."""
    world_view_transforms: torch.Tensor
    view_to_world_transforms: torch.Tensor
    full_proj_transforms: torch.Tensor
    camera_centers: torch.Tensor
    source_cv2wT_quat: torch.Tensor


def _normalize(x):
    return x / torch.norm(x, dim=-1, keepdim=True)


def sample_front_circle_gs(num_frames, yaw_diff=0.25, pitch_diff=0.1, yaw0=0.0, pitch0=0.0):
    """Yaw/pitch of the 'front_circle' orbit (src/utils.py:64-90). Returns angles [num_frames, 3]."""
    steps = torch.linspace(0, 1, num_frames)
    yaw = yaw0 - yaw_diff * torch.sin(steps * 2 * np.pi)
    pitch = pitch0 + pitch_diff * torch.cos(steps * 2 * np.pi)
    return torch.stack([yaw, pitch, torch.zeros_like(yaw)], dim=1)


def spherical2cartesian(yaw, pitch, radius, look_at):
    """src/camera.py:17-32."""
    x = -radius * torch.sin(yaw) * torch.cos(pitch) + look_at[:, 0]
    y = -radius * torch.sin(pitch) + look_at[:, 1]
    z = -radius * torch.cos(pitch) * torch.cos(yaw) + look_at[:, 2]
    return torch.stack([x, y, z], dim=-1)


def compute_cam2world_matrix(angles, radius, look_at):
    """src/camera.py:65-91. angles [n,3] (yaw,pitch,roll), radius [n] or scalar tensor, look_at [n,3]."""
    origins = spherical2cartesian(angles[:, 0], angles[:, 1], radius, look_at)
    forward = _normalize(_normalize(look_at - origins))
    n = forward.shape[0]
    up = torch.tensor([0, 1, 0], dtype=torch.float).expand_as(forward)
    left = _normalize(torch.cross(up, forward, dim=-1))
    up = _normalize(torch.cross(forward, left, dim=-1))
    rot = torch.eye(4).unsqueeze(0).repeat(n, 1, 1)
    rot[:, :3, :3] = torch.stack((-left, up, -forward), dim=-1)
    trans = torch.eye(4).unsqueeze(0).repeat(n, 1, 1)
    trans[:, :3, 3] = origins
    return trans @ rot


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """dataio_gs_test_256_demo.py:237-260 (P[2,2] = (n+f)/(f-n), P[2,3] = -f*n/(f-n), P[3,2] = 1)."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = 1.0 * (znear + zfar) / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def matrix_to_quaternion(M):
    """3x3 rotation -> (r,x,y,z); branch structure of dataio_gs_test_256_demo.py:262-297."""
    tr = 1 + M[0, 0] + M[1, 1] + M[2, 2]
    if tr > 0:
        r = torch.sqrt(tr) / 2.0
        x = (M[2, 1] - M[1, 2]) / (4 * r)
        y = (M[0, 2] - M[2, 0]) / (4 * r)
        z = (M[1, 0] - M[0, 1]) / (4 * r)
    elif (M[0, 0] > M[1, 1]) and (M[0, 0] > M[2, 2]):
        S = torch.sqrt(1.0 + M[0, 0] - M[1, 1] - M[2, 2]) * 2
        r = (M[2, 1] - M[1, 2]) / S
        x = 0.25 * S
        y = (M[0, 1] + M[1, 0]) / S
        z = (M[0, 2] + M[2, 0]) / S
    elif M[1, 1] > M[2, 2]:
        S = torch.sqrt(1.0 + M[1, 1] - M[0, 0] - M[2, 2]) * 2
        r = (M[0, 2] - M[2, 0]) / S
        x = (M[0, 1] + M[1, 0]) / S
        y = 0.25 * S
        z = (M[1, 2] + M[2, 1]) / S
    else:
        S = torch.sqrt(1.0 + M[2, 2] - M[0, 0] - M[1, 1]) * 2
        r = (M[1, 0] - M[0, 1]) / S
        x = (M[0, 2] + M[2, 0]) / S
        y = (M[1, 2] + M[2, 1]) / S
        z = 0.25 * S
    return torch.stack([r, x, y, z], dim=-1)


def _assemble(cam2w, projection_T):
    """visualize.py:250-256. ``cam2w`` is ALREADY ``torch.inverse(compute_cam2world_matrix(..))``
    (visualize.py:249); it is inverted again here, exactly as the reference does."""
    Rt = torch.inverse(cam2w).contiguous()
    world_view = Rt.transpose(1, 2).unsqueeze(1).contiguous()
    view_to_world = cam2w.transpose(1, 2).unsqueeze(1).contiguous()
    centers = world_view.inverse()[:, :, 3, :3].contiguous()
    proj = projection_T.expand([cam2w.shape[0], -1, -1]).contiguous()
    full_proj = world_view[:, 0].bmm(proj).unsqueeze(1).contiguous()
    return world_view, view_to_world, full_proj, centers


def update_camera_pose(world_view, view_to_world, full_proj, inverse_first_camera):
    """Re-base every view on the canonical camera (dataio_gs_test_256_demo.py:300-352, [4,4] branch)."""
    inv_first = inverse_first_camera
    new_wv = torch.zeros_like(world_view)
    new_v2w = torch.zeros_like(view_to_world)
    new_fp = torch.zeros_like(full_proj)
    new_c = torch.zeros(world_view.shape[0], 1, 3, dtype=world_view.dtype)
    inv_inv = inv_first.inverse()
    for c in range(world_view.shape[0]):
        new_wv[c, 0] = torch.bmm(inv_first.unsqueeze(0), world_view[c, 0].unsqueeze(0)).squeeze(0)
        new_v2w[c, 0] = torch.bmm(view_to_world[c, 0].unsqueeze(0), inv_inv.unsqueeze(0)).squeeze(0)
        new_fp[c, 0] = torch.bmm(inv_first.unsqueeze(0), full_proj[c, 0].unsqueeze(0)).squeeze(0)
        new_c[c] = new_wv[c, 0].inverse()[3, :3]
    return new_wv, new_v2w, new_fp, new_c


class OrbitRig:
    """Everything visualize.py derives from the config before touching an image.

    ``canonical`` reproduces the dataset's camera block (dataio_gs_test_256_demo.py:78-133);
    ``orbit(n)`` the n-view novel cameras of visualize.py:236-279; ``orbit_with_frontal(n)`` the
    1 + n cameras of visualize.py:343-381.
    """

    def __init__(self, cfg):
        m = cfg["model"]
        self.cfg = cfg
        self.fov = float(m["fov"])
        self.radius = float(m["radius"])
        self.look_at = float(m["look_at"])
        self.update_pose = bool(cfg["opt"]["update_pose"])
        self.projection_matrix = getProjectionMatrix(
            znear=cfg["dataset_params"]["z_near"], zfar=cfg["dataset_params"]["z_far"],
            fovX=self.fov * 2 * np.pi / 360, fovY=self.fov * 2 * np.pi / 360).transpose(0, 1)

        cam2w = torch.inverse(self._cam2world(torch.zeros(1, 3), torch.ones(1) * self.radius))
        wv, v2w, fp, centers = _assemble(cam2w, self.projection_matrix)
        self.inverse_first_camera: Optional[torch.Tensor] = None
        if self.update_pose:
            self.inverse_first_camera = wv[0, 0].inverse().clone()
            wv, v2w, fp, centers = update_camera_pose(wv, v2w, fp, self.inverse_first_camera)
        quat = matrix_to_quaternion(v2w[0, 0, :3, :3].transpose(0, 1)).reshape(1, 1, 4)
        self.canonical = CameraSet(wv, v2w, fp, centers, quat)

    def _cam2world(self, angles, radius):
        look_at = torch.zeros(angles.shape[0], 3)
        look_at[:, 2] = self.look_at
        return compute_cam2world_matrix(angles, radius, look_at)

    def _finish(self, cam2w, quat_before_update, n_quat=None):
        wv, v2w, fp, centers = _assemble(cam2w, self.projection_matrix)
        n = wv.shape[0]

        def quats(mats):
            q = torch.zeros(n, 1, 4)
            for i in range(n if n_quat is None else n_quat):
                q[i] = matrix_to_quaternion(mats[i, 0, :3, :3].transpose(0, 1).contiguous())
            return q

        if quat_before_update:          # visualize.py:373-375 computes them BEFORE update_camera_pose
            q = quats(v2w)
        if self.update_pose:
            wv, v2w, fp, centers = update_camera_pose(wv, v2w, fp, self.inverse_first_camera)
        if not quat_before_update:      # visualize.py:276-278 computes them AFTER
            q = quats(v2w)
        return CameraSet(wv, v2w, fp, centers, q)

    def orbit(self, num_frames, yaw_diff=0.25, pitch_diff=0.15):
        angles = sample_front_circle_gs(num_frames, yaw_diff, pitch_diff)
        radius = torch.ones(num_frames) * self.radius
        return self._finish(torch.inverse(self._cam2world(angles, radius)), quat_before_update=False)

    def orbit_with_frontal(self, num_frames, yaw_diff=0.25, pitch_diff=0.15):
        angles = sample_front_circle_gs(num_frames, yaw_diff, pitch_diff)
        frontal = sample_front_circle_gs(1, 0.0, 0.0)
        cam2w = torch.cat([
            torch.inverse(self._cam2world(frontal, torch.ones(1) * self.radius)),
            torch.inverse(self._cam2world(angles, torch.ones(num_frames) * self.radius))], 0)
        # visualize.py:373-375 fills quaternions for the first ``num_frames`` of the 1+num_frames entries only
        return self._finish(cam2w, quat_before_update=True, n_quat=num_frames)


def default_cfg(resolution=256):
    """The values of config/imagenetgs_256x256_v1.yaml that the hot path reads."""
    return {
        "model": {"fov": 13.164, "radius": 7.667, "look_at": 7.667, "training_resolution": resolution,
                  "max_sh_degree": 1, "inverted_x": False, "inverted_y": True, "network_with_offset": True,
                  "network_without_offset": False, "network_with_uncertainty": False, "cross_view_attention": True,
                  "origin_distances": False, "isotropic": False, "base_dim": 128, "name": "SingleUNet",
                  "attention_resolutions": [16], "num_blocks": 3,
                  "opacity_scale": 0.001, "opacity_bias": -3.0, "scale_bias": 0.01, "scale_scale": 0.0005,
                  "xyz_scale": 0.000001, "xyz_bias": 0.0, "depth_scale": 1.0, "depth_bias": 0.0},
        "dataset_params": {"z_near": 6.667, "z_far": 8.667},
        "opt": {"squre_clip": 10000.0, "update_pose": True},
    }
