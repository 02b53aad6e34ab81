"""Predictor operator of the hot path: ``GaussianSplatPredictor_gtunet`` with the cycle-aggregative projection
("splat head") as ONE fused HIP kernel, plus the SongUNet backbone as a plain PyTorch-ROCm module.

Interface mirrored from reference src/gaussian_predictor.py:
  GaussianSplatPredictor_gtunet.__init__/forward        :598-648, :883-1007   (same constructor cfg, same forward
      signature, same returned dict keys/shapes: xyz [B,N,3], opacity [B,N,1], scaling [B,N,3], rotation [B,N,4],
      features_dc [B,N,1,3], features_rest [B,N,3,3], unet_depth [B,N,1] with N = N_views*H*W)
  init_ray_dirs :657-681, init_sh_transform_matrices :649-655, get_splits_and_inits :683-734
  SongUNet as instantiated by F3D-Gaus  :137-193, :282-351, :361-510, :546-586
      -- same module tree and parameter/buffer names, so the released checkpoint's state_dict loads unchanged
      (387 tensors: gaussian_predictor.network_with_offset.encoder.{enc,dec}.<level>_<block>.<param>, ...out.*).

What is different by design (MI355X-first):
  * everything after the network -- ray back-projection, view->world bmm, sigmoid/exp/normalize, quaternion
    composition, SH rotation, NCHW -> N x C flattening, multi-view union -- is libf3dg_hip's ``f3dg_splat_head``
    (one pass, 96 B read + 96 B written per Gaussian) instead of ~15 torch kernels and permute copies;
  * ``forward(..., out=, n_offset=)`` lets the cycle loop write each pass straight into the aggregated buffers;
  * the backbone's x2 up / down resampling uses nearest-upsample / 2x2 mean (identical maths to the reference's
    depthwise [1,1] filter convolutions) and the 1-head attention uses scaled_dot_product_attention.
"""
import ctypes as C
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .diff_gof_rasterization import _stream

SPLIT_WITH_OFFSET = [3, 1, 3, 4, 3, 9]      # offset, opacity, scaling, rotation, features_dc, features_rest


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def quaternion_raw_multiply(a, b):
    """Hamilton product, real part first (gaussian_predictor.py:45-64). Host-side helper (tests, tools)."""
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)


# ------------------------------------------------------------------------------------------------ backbone
def _xavier_uniform(shape, fan_in, fan_out, gain=1.0):
    return gain * math.sqrt(6.0 / (fan_in + fan_out)) * (torch.rand(*shape) * 2 - 1)


def _is_nhwc(t):
    """A 4-D tensor stored channels-last (and not also NCHW-contiguous, as 1-channel or 1x1 tensors are)."""
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()


_HALF_DTYPES = (torch.bfloat16, torch.float16)
_KERNEL_SUFFIX = {torch.float32: "", torch.bfloat16: "_bf16", torch.float16: "_f16"}      # of the GroupNorm / residual-join entry points


class Conv2d(nn.Module):
    """Convolution with optional x2 up / down resampling BEFORE the convolution (gaussian_predictor.py:137-178 with
    resample_filter [1,1], fused_resample False). kernel = 0 means "resample only"."""

    def __init__(self, in_channels, out_channels, kernel, up=False, down=False, init_weight=1.0):
        super().__init__()
        self.in_channels, self.out_channels, self.up, self.down = in_channels, out_channels, up, down
        if kernel:
            fi, fo = in_channels * kernel * kernel, out_channels * kernel * kernel
            self.weight = nn.Parameter(_xavier_uniform([out_channels, in_channels, kernel, kernel], fi, fo, init_weight))
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.weight = None
            self.bias = None
        # present in the reference state_dict for resampling convs (persistent buffer, value 0.25 everywhere)
        self.register_buffer('resample_filter', torch.full((1, 1, 2, 2), 0.25) if (up or down) else None)

    def _filter(self, dtype, nhwc):
        """The filter in the activations' dtype and layout. In inference the converted copy (bfloat16 under the bf16 option, channels-last
        under the nhwc layout) is kept until the parameter changes (its version counter or storage) instead of being rebuilt per call:
        108 cast / transpose kernels per pass otherwise."""
        w = self.weight
        nhwc = nhwc and w.shape[1] > 1 and (w.shape[2] > 1 or w.shape[3] > 1)       # (1x1 filters and single-channel ones are both layouts at once)
        if dtype == w.dtype and not nhwc:
            return w
        if torch.is_grad_enabled() and w.requires_grad:
            t = w.to(dtype)
            return t.contiguous(memory_format=torch.channels_last) if nhwc else t
        try:
            key = (w._version, w.data_ptr(), w.device, dtype, nhwc)
        except RuntimeError:            # (inference-mode tensors have no version counter: no cache)
            t = w.to(dtype)
            return t.contiguous(memory_format=torch.channels_last) if nhwc else t
        cached = self.__dict__.get("_filter_cache")
        if cached is None or cached[0] != key:
            t = w.detach().to(dtype)
            if nhwc:
                t = t.contiguous(memory_format=torch.channels_last)
            cached = (key, t)
            self.__dict__["_filter_cache"] = cached
        return cached[1]

    def invalidate_filter_cache(self):
        """Drops the converted filter copy. The cache key (version counter, storage pointer, device) notices optimizer steps, in-place
        tensor methods, ``load_state_dict`` and ``.to()`` -- but NOT writes through ``param.data`` (``param.data.copy_()``, ``.mul_()``: EMA
        swaps, manual weight surgery) or through a raw pointer, which leave the version counter alone: after such a write call this (or
        ``GaussianSplatPredictor_gtunet.invalidate_filter_cache()``), or the backbone keeps convolving with the old filters."""
        self.__dict__.pop("_filter_cache", None)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_filter_cache()
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_filter_cache()
        return super()._apply(fn, *args, **kwargs)

    def forward(self, x, N_views_xa=1, bias=True):
        """``bias=False``: the convolution without its bias -- the caller hands ``self.bias`` to the kernel that consumes the result
        (GroupNorm's ``pre_bias`` / the residual join), which saves the separate bias pass PyTorch-ROCm runs behind MIOpen's kernel."""
        if self.up:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        if self.down:
            x = F.avg_pool2d(x, 2)
        if self.weight is not None:
            x = F.conv2d(x, self._filter(x.dtype, _is_nhwc(x)), self.bias.to(x.dtype) if bias else None, padding=self.weight.shape[-1] // 2)
        return x


class GroupNorm(nn.Module):
    def __init__(self, num_channels, num_groups=32, min_channels_per_group=4, eps=1e-5):
        super().__init__()
        self.num_groups = min(num_groups, num_channels // min_channels_per_group)
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))

    def _fusable(self, x):
        """Inference on a HIP device in a dtype the kernels take: the fused HIP kernels run (no autograd through them)."""
        return x.is_cuda and x.dtype in _KERNEL_SUFFIX and x.dim() >= 3 and self.weight.dtype == torch.float32 and not (
            torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad))

    def forward(self, x, N_views_xa=1, silu=False, pre_bias=None):
        """``silu=True`` returns ``F.silu(group_norm(x))``: the pair the residual blocks always apply together. In
        inference on a HIP device the pair is ONE fused kernel (``f3dg_group_norm_silu``); with autograd
        or on the host (the CPU fixtures of the backbone) it is the two PyTorch ops. ``pre_bias`` ([C] float32): the result is
        that of ``x + pre_bias[None, :, None, None]`` -- the bias of the convolution that produced ``x`` (``Conv2d.forward(bias=False)``)."""
        if self._fusable(x) and (pre_bias is None or pre_bias.dtype == torch.float32):
            L = _lib.lib()
            if _is_nhwc(x) and x.shape[1] % (4 if x.dtype == torch.float32 else 8) == 0 and x.shape[1] <= 1024:
                # layout option "nhwc": the channels-last kernel, channels-last out
                y = torch.empty_like(x)           # preserves the strides
                N, Cc = x.shape[0], x.shape[1]
                HW = x.shape[2] * x.shape[3]
                mom = torch.empty(L.f3dg_group_norm_nhwc_scratch_bytes(N, HW, self.num_groups) // 8 + 1, dtype=torch.float64, device=x.device)
                mom_bytes = mom.numel() * 8
                fn = getattr(L, "f3dg_group_norm_silu_nhwc_pb" + _KERNEL_SUFFIX[x.dtype])
                rc = fn(_stream(), N, Cc, x.shape[2] * x.shape[3], self.num_groups, _lib.ptr(x), _lib.ptr(pre_bias), _lib.ptr(self.weight),
                        _lib.ptr(self.bias), float(self.eps), 1 if silu else 0, _lib.ptr(y), _lib.ptr(mom), mom_bytes)
                _lib.check(rc, "f3dg_group_norm_silu_nhwc_pb")
                return y
            xc = x.contiguous()
            y = torch.empty_like(xc)
            N, Cc = xc.shape[0], xc.shape[1]
            fn = getattr(L, "f3dg_group_norm_silu_pb" + _KERNEL_SUFFIX[x.dtype])
            rc = fn(_stream(), N, Cc, xc.numel() // max(N * Cc, 1), self.num_groups, _lib.ptr(xc), _lib.ptr(pre_bias), _lib.ptr(self.weight),
                    _lib.ptr(self.bias), float(self.eps), 1 if silu else 0, _lib.ptr(y))
            _lib.check(rc, "f3dg_group_norm_silu_pb")
            return y
        if pre_bias is not None:
            x = x + pre_bias.to(x.dtype).reshape(1, -1, *([1] * (x.dim() - 2)))
        y = F.group_norm(x, self.num_groups, self.weight.to(x.dtype), self.bias.to(x.dtype), self.eps)
        return F.silu(y) if silu else y


def residual_join(a, bias_a, b, bias_b, scale):
    """((a + bias_a) + (b + bias_b)) * scale with per-channel biases (either may be None): the tail of a residual block. One HIP
    kernel (``f3dg_residual_join``) for inference tensors on a HIP device that share a dense layout (NCHW or channels-last); the
    PyTorch ops otherwise. Writes into ``a``'s storage when it can."""
    def tb(t, bias):
        return t if bias is None else t + bias.to(t.dtype).reshape(1, -1, 1, 1)
    ok = (a.is_cuda and a.dim() == 4 and a.dtype in _KERNEL_SUFFIX and b.dtype == a.dtype and a.shape == b.shape
          and not (torch.is_grad_enabled() and (a.requires_grad or b.requires_grad))
          and all(t is None or (t.dtype == torch.float32 and t.is_contiguous()) for t in (bias_a, bias_b))
          and a.numel() % (4 if a.dtype == torch.float32 else 8) == 0)
    if ok:
        nhwc = _is_nhwc(a)
        same = (_is_nhwc(b) if nhwc else b.is_contiguous()) and (nhwc or a.is_contiguous()) and not (
            nhwc and a.shape[1] % (4 if a.dtype == torch.float32 else 8))
        if same and (a.data_ptr() | b.data_ptr()) % 16 == 0:
            N, Cc, H, W = a.shape
            fn = getattr(_lib.lib(), "f3dg_residual_join" + _KERNEL_SUFFIX[a.dtype])
            _lib.check(fn(_stream(), N, Cc, H * W, 1 if nhwc else 0, _lib.ptr(a), _lib.ptr(bias_a), _lib.ptr(b), _lib.ptr(bias_b), float(scale),
                          _lib.ptr(a)), "f3dg_residual_join")
            return a
    return (tb(a, bias_a) + tb(b, bias_b)) * scale


class UNetBlock(nn.Module):
    """Residual block without embedding path (gaussian_predictor.py:282-351 with emb=None, adaptive_scale False,
    num_heads 1, skip_scale sqrt(.5), eps 1e-6, resample_proj True)."""

    def __init__(self, in_channels, out_channels, up=False, down=False, attention=False, dropout=0.10):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_heads = 1 if attention else 0
        self.dropout = dropout
        self.skip_scale = math.sqrt(0.5)
        self.norm0 = GroupNorm(in_channels, eps=1e-6)
        self.conv0 = Conv2d(in_channels, out_channels, 3, up=up, down=down)
        self.norm1 = GroupNorm(out_channels, eps=1e-6)
        self.conv1 = Conv2d(out_channels, out_channels, 3, init_weight=1e-5)
        self.skip = None
        if out_channels != in_channels or up or down:
            self.skip = Conv2d(in_channels, out_channels, 1, up=up, down=down)
        if self.num_heads:
            self.norm2 = GroupNorm(out_channels, eps=1e-6)
            self.qkv = Conv2d(out_channels, out_channels * 3, 1, init_weight=math.sqrt(0.2))
            self.proj = Conv2d(out_channels, out_channels, 1, init_weight=1e-5)

    def forward(self, x, emb=None, N_views_xa=1):
        orig = x
        if self.norm0._fusable(x) and not self.training:
            # inference on a HIP device: the three convolutions run without their biases, which go to the kernels that read the results
            # (norm1 takes conv0's; the residual join takes conv1's and the skip convolution's, adds, and scales): same float32
            # operations in the same order as the lines below, five elementwise passes fewer
            x = self.conv0(self.norm0(x, silu=True), bias=False)
            x = self.norm1(x, silu=True, pre_bias=self.conv0.bias)
            x = self.conv1(x, bias=False)
            if self.skip is not None:
                x = residual_join(x, self.conv1.bias, self.skip(orig, bias=False), self.skip.bias, self.skip_scale)
            else:
                x = residual_join(x, self.conv1.bias, orig, None, self.skip_scale)
        else:
            x = self.conv0(self.norm0(x, silu=True))
            x = self.norm1(x, silu=True)
            x = self.conv1(F.dropout(x, p=self.dropout, training=self.training))
            x = x + (self.skip(orig) if self.skip is not None else orig)
            x = x * self.skip_scale
        if self.num_heads:
            if N_views_xa != 1:      # fold the views into the token axis (gaussian_predictor.py:333-338)
                B, Cc, H, W = x.shape
                x = x.reshape(B // N_views_xa, N_views_xa, Cc, H, W).permute(0, 2, 1, 3, 4).reshape(B // N_views_xa, Cc, N_views_xa * H, W)
            b, c = x.shape[0], x.shape[1]
            qkv = self.qkv(self.norm2(x)).reshape(b, c, 3, -1)          # channel index = c*3 + {q,k,v}
            q, k, v = (t.transpose(1, 2).unsqueeze(1).float() for t in qkv.unbind(2))      # [b,1,L,c]
            a = F.scaled_dot_product_attention(q, k, v)                  # softmax(q k^T / sqrt(c)) v, fp32
            a = a.squeeze(1).transpose(1, 2).to(x.dtype).reshape(*x.shape)
            if _is_nhwc(x):
                a = a.contiguous(memory_format=torch.channels_last)
            x = self.proj(a) + x
            x = x * self.skip_scale
            if N_views_xa != 1:
                x = x.reshape(B // N_views_xa, Cc, N_views_xa, H, W).permute(0, 2, 1, 3, 4).reshape(B, Cc, H, W)
        return x


class SongUNet(nn.Module):
    """DDPM++ U-Net in the one configuration F3D-Gaus instantiates (gaussian_predictor.py:361-510 via :561-568):
    model_channels 128, channel_mult [1,2,2,2], standard encoder/decoder, no embedding, attention at the level
    NAMED '16x16' (names derive from img_resolution = cfg.base_dim = 128, not from the real input size)."""

    def __init__(self, img_resolution, in_channels, out_channels, num_blocks=3, attn_resolutions=(16,),
                 model_channels=128, channel_mult=(1, 2, 2, 2), dropout=0.10):
        super().__init__()
        self.enc = nn.ModuleDict()
        cout = in_channels
        for level, mult in enumerate(channel_mult):
            res = img_resolution >> level
            if level == 0:
                cin, cout = cout, model_channels
                self.enc[f'{res}x{res}_conv'] = Conv2d(cin, cout, 3)
            else:
                self.enc[f'{res}x{res}_down'] = UNetBlock(cout, cout, down=True, dropout=dropout)
            for idx in range(num_blocks):
                cin, cout = cout, model_channels * mult
                self.enc[f'{res}x{res}_block{idx}'] = UNetBlock(cin, cout, attention=(res in attn_resolutions), dropout=dropout)
        skips = [blk.out_channels for blk in self.enc.values()]

        self.dec = nn.ModuleDict()
        for level, mult in reversed(list(enumerate(channel_mult))):
            res = img_resolution >> level
            if level == len(channel_mult) - 1:
                self.dec[f'{res}x{res}_in0'] = UNetBlock(cout, cout, attention=True, dropout=dropout)
                self.dec[f'{res}x{res}_in1'] = UNetBlock(cout, cout, dropout=dropout)
            else:
                self.dec[f'{res}x{res}_up'] = UNetBlock(cout, cout, up=True, dropout=dropout)
            for idx in range(num_blocks + 1):
                cin = cout + skips.pop()
                cout = model_channels * mult
                self.dec[f'{res}x{res}_block{idx}'] = UNetBlock(cin, cout, attention=(idx == num_blocks and res in attn_resolutions), dropout=dropout)
            if level == 0:
                self.dec[f'{res}x{res}_aux_norm'] = GroupNorm(cout, eps=1e-6)
                self.dec[f'{res}x{res}_aux_conv'] = Conv2d(cout, out_channels, 3, init_weight=0.2)

    def forward(self, x, film_camera_emb=None, N_views_xa=1):
        skips = []
        for blk in self.enc.values():
            x = blk(x, N_views_xa=N_views_xa)
            skips.append(x)
        out = None
        for name, blk in self.dec.items():
            if name.endswith('aux_norm'):
                out = blk(x, silu=True)            # the reference applies silu before aux_conv (:487-488)
            elif name.endswith('aux_conv'):
                out = blk(out)
            else:
                if x.shape[1] != blk.in_channels:
                    x = torch.cat([x, skips.pop()], dim=1)
                x = blk(x, N_views_xa=N_views_xa)
        return out


class SingleImageSongUNetPredictor(nn.Module):
    """SongUNet + 1x1 output conv whose per-split initial scale/bias shape the initial Gaussians (:546-586)."""

    def __init__(self, cfg, out_channels, bias, scale):
        super().__init__()
        self.out_channels = out_channels
        self.cfg = cfg
        self.encoder = SongUNet(cfg['model']['base_dim'], 4, sum(out_channels), num_blocks=cfg['model']['num_blocks'],
                                attn_resolutions=cfg['model']['attention_resolutions'])
        self.out = nn.Conv2d(sum(out_channels), sum(out_channels), kernel_size=1)
        start = 0
        with torch.no_grad():
            for n, b, s in zip(out_channels, bias, scale):
                nn.init.xavier_uniform_(self.out.weight[start:start + n], s)
                nn.init.constant_(self.out.bias[start:start + n], b)
                start += n

    def forward(self, x, film_camera_emb=None, N_views_xa=1):
        return self.out(self.encoder(x, N_views_xa=N_views_xa))


def networkCallBack(cfg, name, out_channels, **kwargs):
    if name == "SingleUNet":
        return SingleImageSongUNetPredictor(cfg, out_channels, **kwargs)
    raise NotImplementedError


# ------------------------------------------------------------------------------------------------ splat head
GAUSSIAN_KEYS = ("xyz", "opacity", "scaling", "rotation", "features_dc", "features_rest", "unet_depth")
_KEY_SHAPE = {"xyz": (3,), "opacity": (1,), "scaling": (3,), "rotation": (4,), "features_dc": (1, 3),
              "features_rest": (3, 3), "unet_depth": (1,)}


def allocate_gaussians(B, N, device):
    """Empty aggregated Gaussian buffers: dict key -> [B, N, ...] float32 (the layout every consumer expects)."""
    return {k: torch.empty((B, N) + _KEY_SHAPE[k], dtype=torch.float32, device=device) for k in GAUSSIAN_KEYS}


def splat_head(net_out, depth, ray_dirs, view_to_world, cam_quat, squre_clip=10000.0, out=None, n_offset=0):
    """Fused cycle-aggregative projection (f3dg_splat_head). net_out [B,23,H,W], depth [B,1,H,W], ray_dirs [1,3,H,W],
    view_to_world [B,4,4] (row-vector convention), cam_quat [B,4]. Writes image b's H*W Gaussians at
    ``out[key][b, n_offset : n_offset + H*W]`` (allocating [B,H*W,...] buffers when ``out`` is None)."""
    B, Cc, H, W = net_out.shape
    if Cc != 23:
        raise RuntimeError("splat head expects the 23-channel with-offset / SH-degree-1 layout [3,1,3,4,3,9]")
    device = net_out.device
    if device.type != "cuda":
        raise RuntimeError("f3dgaus_amd splat head needs tensors on a HIP device (no CPU fallback)")
    HW = H * W
    if out is None:
        out = allocate_gaussians(B, HW, device)
    n_total = out["xyz"].shape[1]
    f = lambda t: t.to(device=device, dtype=torch.float32).contiguous()
    net_out, depth, ray_dirs = f(net_out), f(depth), f(ray_dirs)
    v2w, quat = f(view_to_world).reshape(B, 16), f(cam_quat).reshape(B, 4)
    for k in GAUSSIAN_KEYS:
        t = out[k]
        if not (t.is_contiguous() and t.dtype == torch.float32 and t.shape[0] == B and t.shape[1] == n_total):
            raise RuntimeError(f"aggregated buffer '{k}' must be contiguous float32 [B, n_total, ...]")
    rc = _lib.lib().f3dg_splat_head(
        _stream(), B, H, W, _lib.ptr(net_out), _lib.ptr(depth), _lib.ptr(ray_dirs), _lib.ptr(v2w), _lib.ptr(quat),
        float(squre_clip), int(n_total), int(n_offset), *[_lib.ptr(out[k]) for k in GAUSSIAN_KEYS])
    _lib.check(rc, "f3dg_splat_head")
    return out


class GaussianSplatPredictor_gtunet(nn.Module):
    def invalidate_filter_cache(self):
        """Forget every convolution's converted filter copy (see ``Conv2d.invalidate_filter_cache``: needed after writes through
        ``param.data`` or raw pointers, which no version counter sees)."""
        for mod in self.modules():
            if isinstance(mod, Conv2d):
                mod.invalidate_filter_cache()

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        m = cfg['model']
        assert m['network_with_offset'] or m['network_without_offset'], "Need at least one network"
        if not m['network_with_offset'] or m.get('network_with_uncertainty') or m['max_sh_degree'] != 1 \
                or m.get('origin_distances') or m.get('isotropic'):
            raise NotImplementedError(
                "this build accelerates the configuration F3D-Gaus ships (config/imagenetgs_256x256_v1.yaml: "
                "network_with_offset, max_sh_degree 1, anisotropic, no origin_distances / uncertainty head)")
        split_dimensions, scale_inits, bias_inits = self.get_splits_and_inits(True, cfg)
        self.network_with_offset = networkCallBack(cfg, m['name'], split_dimensions, scale=scale_inits, bias=bias_inits)
        # extension: "bf16" / "fp16" run the backbone's convolutions under bfloat16 / float16 autocast with 16-bit activations
        # between the layers (GroupNorm statistics, attention and the splat head stay float32); "fp32" (default) is the reference's
        # precision. bf16 costs the colour of a frame (RGB 24 dB against the fp32 backbone on the real image); fp16 has three more
        # mantissa bits at the same MFMA rate, and every activation is GroupNorm-bounded (profiles/r05_final/fp16_frames.md)
        self.backbone_dtype = str(m.get('backbone_dtype', 'fp32'))
        if self.backbone_dtype not in ("fp32", "bf16", "fp16"):
            raise ValueError("backbone_dtype must be 'fp32', 'bf16' or 'fp16'")
        # extension: inference passes of more than `backbone_chunk` images run as chunks of that many (the images of a pass are
        # independent: cross-view attention couples only the Nv views of one image, and chunks are whole images). Why: MIOpen's find step
        # BENCHMARKS its candidate kernels at the real problem size the first time a process (a box without a find-db) meets a shape --
        # measured on a fresh MI355X box 43 s for the first 8-image fp32 pass, 85 s at 16, 6-7 minutes at 64 -- and its fast find mode
        # (MIOPEN_FIND_MODE=2: 2.7 s) picks kernels 2.3x (fp32) to 12x (bf16) slower. The fp32 backbone's cost per image is flat from 8
        # images on (10.7 / 10.2 / 10.2 ms at 8 / 16 / 64), so its default is 8: a 64-image pass is eight 8-image passes, first use
        # bounded at 43 s whatever the batch. 0 = never chunk: the default of the 16-bit options, whose MFMA kernels still gain from
        # larger batches (2.3 against 2.8 ms per image at 64 / 16 images) -- set it there to bound the first use as well.
        self.backbone_chunk = int(m.get('backbone_chunk', 8 if self.backbone_dtype == "fp32" else 0))
        # extension: "nhwc" keeps the backbone's activations and filters channels-last, the layout of MIOpen's fastest kernels on gfx950 --
        # GroupNorm+SiLU and the residual join are channels-last HIP kernels, nothing converts in between (fp32 pass 92 -> 82 ms, bf16
        # 30 -> 25 ms per 8 images; the reference fixture within 1.0e-5); "nchw" is torch's layout, which is faster for ONE image per pass
        # (14.7 against 16.9 ms: three launches per channels-last GroupNorm instead of one, and a pass of one image is ~500 dependent
        # dispatches of small grids); "auto" (default) = nhwc for passes of two images or more (four with the bf16 option), on a HIP device
        # without autograd
        self.backbone_layout = str(m.get('backbone_layout', 'auto'))
        if self.backbone_layout not in ("auto", "nchw", "nhwc"):
            raise ValueError("backbone_layout must be 'auto', 'nchw' or 'nhwc'")
        self.init_ray_dirs()
        self.init_sh_transform_matrices()

    def init_sh_transform_matrices(self):
        v_to_sh = torch.tensor([[0, 0, -1], [-1, 0, 0], [0, 1, 0]], dtype=torch.float32)
        self.register_buffer('sh_to_v_transform', v_to_sh.transpose(0, 1).unsqueeze(0))
        self.register_buffer('v_to_sh_transform', v_to_sh.unsqueeze(0))

    def init_ray_dirs(self):
        res = self.cfg['model']['training_resolution']
        x = torch.linspace(-res // 2 + 0.5, res // 2 - 0.5, res)
        y = torch.linspace(res // 2 - 0.5, -res // 2 + 0.5, res)
        if self.cfg['model']['inverted_x']:
            x = -x
        if self.cfg['model']['inverted_y']:
            y = -y
        grid_x, grid_y = torch.meshgrid(x, y, indexing='xy')
        ray_dirs = torch.stack([grid_x, grid_y, torch.ones_like(grid_x)]).unsqueeze(0)
        ray_dirs[:, :2, ...] /= fov2focal(self.cfg['model']['fov'] * np.pi / 180, res)
        self.register_buffer('ray_dirs', ray_dirs)

    def get_splits_and_inits(self, with_offset, cfg):
        m = cfg['model']
        assert with_offset
        split = [3, 1, 3, 4, 3]
        scale = [m['xyz_scale'], m['opacity_scale'], m['scale_scale'], 1.0, 5.0]
        bias = [m['xyz_bias'], m['opacity_bias'], np.log(m['scale_bias']), 0.0, 0.0]
        if m['max_sh_degree'] != 0:
            split.append(((m['max_sh_degree'] + 1) ** 2 - 1) * 3)
            scale.append(0.0)
            bias.append(0.0)
        self.split_dimensions_with_offset = split
        return split, scale, bias

    def forward(self, x, source_cameras_view_to_world, source_cv2wT_quat=None, focals_pixels=None,
                return_depth=False, squre_clip=10000.0, unet_depth=None, out=None, n_offset=0):
        """x [B,Nv,4,H,W]; view_to_world [B,Nv,4,4]; quat [B,Nv,4]; unet_depth [B*Nv,1,H,W] (bchw).
        ``out`` / ``n_offset`` (extension): write into preallocated aggregated buffers [B, n_total, ...];
        requires Nv == 1 so that image b's Gaussians stay contiguous."""
        assert focals_pixels is None, "Unexpected argument for srn dataset"
        assert source_cv2wT_quat is not None
        assert unet_depth is not None, "F3D-Gaus feeds the (monocular or rendered) depth map as unet_depth"
        B, Nv = x.shape[0], x.shape[1]
        N_views_xa = Nv if self.cfg['model']['cross_view_attention'] else 1
        x = x.reshape(B * Nv, *x.shape[2:])
        v2w = source_cameras_view_to_world.reshape(B * Nv, 4, 4)
        quat = source_cv2wT_quat.reshape(B * Nv, 4)
        inference = x.is_cuda and not torch.is_grad_enabled()
        half = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(self.backbone_dtype) if inference else None

        def backbone(xc):
            # ("auto": measured crossover -- fp32 25.9 against 26.4 ms at two images, bf16 11.5 against 10.8 at two and 24.2 against 25.4 at eight)
            if inference and (self.backbone_layout == "nhwc" or (self.backbone_layout == "auto" and xc.shape[0] >= (4 if half is not None else 2))):
                xc = xc.contiguous(memory_format=torch.channels_last)       # (the filters follow per convolution: Conv2d._filter)
            if half is not None:
                with torch.autocast("cuda", dtype=half):
                    return self.network_with_offset(xc, film_camera_emb=None, N_views_xa=N_views_xa).float()
            return self.network_with_offset(xc, film_camera_emb=None, N_views_xa=N_views_xa)

        step = self.backbone_chunk * N_views_xa if self.backbone_chunk > 0 else 0
        if inference and step and x.shape[0] > step:
            net_out = None
            for i in range(0, x.shape[0], step):
                part = backbone(x[i:i + step])
                if net_out is None:
                    net_out = torch.empty((x.shape[0],) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)
                net_out[i:i + step] = part
        else:
            net_out = backbone(x)
        net_out = net_out.contiguous()                  # (the splat head reads planar channels)
        H, W = net_out.shape[-2:]
        depth = unet_depth.reshape(B * Nv, 1, H, W)
        if out is not None:
            if Nv != 1:
                raise RuntimeError("in-place aggregation (out=) needs one view per call")
            return splat_head(net_out, depth, self.ray_dirs, v2w, quat, squre_clip, out=out, n_offset=n_offset)
        res = splat_head(net_out, depth, self.ray_dirs, v2w, quat, squre_clip)
        # multi_view_union (:796-800): [B*Nv, HW, ...] -> [B, Nv*HW, ...]; a pure view of the same memory
        return {k: t.reshape(B, Nv * t.shape[1], *t.shape[2:]) for k, t in res.items()}
