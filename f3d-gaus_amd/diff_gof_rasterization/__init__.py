"""Drop-in replacement of the reference's ``diff_gof_rasterization`` package on MI355X.

Same names, argument meaning, return shapes and error behaviour as
RAST/diff_gof_rasterization/__init__.py (``GaussianRasterizationSettings_GOF`` :168-182,
``GaussianRasterizer_GOF`` :185-307, ``_RasterizeGaussians`` :46-165), but backed by libf3dg_hip.so through the
C ABI of include/f3dg.h instead of the pybind ``_C`` module (RAST/ext.cpp:15-20). Put the directory that contains
this package on ``sys.path`` (or ``import f3dgaus_amd`` first, which registers it) and the reference's
``from diff_gof_rasterization import GaussianRasterizationSettings_GOF, GaussianRasterizer_GOF``
(src/gaussian_renderer/__init__.py:10) resolves here unchanged.

Beyond the reference API, ``rasterize_views`` renders many cameras of the same Gaussians in one launch sequence
with no host synchronisation (the MI355X-first entry the batched loops use).
"""
from typing import NamedTuple

import ctypes as C
import torch
import torch.nn as nn

from .. import _lib

__all__ = ["GaussianRasterizationSettings_GOF", "GaussianRasterizer_GOF", "rasterize_gaussians", "rasterize_views",
           "set_deferred_status", "deferred_status", "flush",
           "integrate_gaussians_to_points", "integrate_prepare", "integrate_points", "PreparedIntegration", "Workspace"]


class GaussianRasterizationSettings_GOF(NamedTuple):
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


def _raw_stream(device_index=None):
    """The current HIP stream of a device (default: the current device) as an integer handle. torch.cuda.current_stream() builds a Stream
    object through two device-index resolutions (7 us a call, five calls per one-view render: a fifth of the drop-in loop's host time);
    the raw getter is what it wraps."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice() if device_index is None else device_index)
    except AttributeError:          # (a torch without the private getters)
        return torch.cuda.current_stream(device_index).cuda_stream


def _stream():
    return C.c_void_p(_raw_stream())


def _dev_f32(t, device):
    """contiguous float32 on `device`; empty / None -> None (the reference maps empty tensors to nullptr)."""
    if t is None or t.numel() == 0:
        return None
    if t.device != device or t.dtype != torch.float32:
        t = t.to(device=device, dtype=torch.float32)
    return t.contiguous()


class Workspace:
    """Caller-owned scratch for one forward (and its backward): the counterpart of the three resizable byte
    tensors of the reference (RAST/rasterize_points.cu:72-82), sized once per capacity instead of grown mid-call.
    The library carves the buffer from (P, W, H, n_views, max_rendered) on every call, so a buffer can be reused for any
    call it is large enough for (``fits`` re-targets it); forward and backward of one call must use the same five numbers."""

    def __init__(self, P, W, H, n_views, max_rendered, device):
        self.P, self.W, self.H, self.n_views, self.max_rendered = int(P), int(W), int(H), int(n_views), int(max_rendered)
        nbytes = _lib.lib().f3dg_workspace_bytes(self.P, self.W, self.H, self.n_views, self.max_rendered)
        if nbytes == 0:
            raise _lib.F3dgError(_lib.ERR_BAD_ARG, "f3dg_workspace_bytes")
        self.buffer = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        self.num_rendered = None
        self.save_aux = False         # whether the last forward on this workspace kept the planes f3dg_backward reads

    @property
    def nbytes(self):
        return self.buffer.numel()

    def fits(self, P, W, H, n_views, max_rendered):
        """True (and the workspace now describes that call) if the buffer is large enough for it."""
        same = (self.P, self.W, self.H, self.n_views) == (P, W, H, n_views) and self.max_rendered >= max_rendered
        # (asked again even for the same five numbers: the carving also depends on process options, e.g. sort_fused_rects)
        need = _lib.lib().f3dg_workspace_bytes(int(P), int(W), int(H), int(n_views), int(self.max_rendered if same else max_rendered))
        if need == 0 or need > self.buffer.numel():
            return False
        if same:
            return True
        self.P, self.W, self.H, self.n_views, self.max_rendered = int(P), int(W), int(H), int(n_views), int(max_rendered)
        return True


class _StreamCache(dict):
    """Per-(device, stream) cache that keeps the most recently used entries only: transient streams do not pin their workspaces."""

    def __init__(self, limit=8):
        super().__init__()
        self.limit = limit

    def get(self, key, default=None):
        if key in self:
            self[key] = super().pop(key)      # move to the end: most recently used
            return super().get(key)
        return default

    def __setitem__(self, key, value):
        if key in self:
            super().pop(key)
        super().__setitem__(key, value)
        while len(self) > self.limit:
            super().pop(next(iter(self)))


_WS_CACHE = _StreamCache()      # (device index, stream) -> Workspace reused by no-grad single-view calls on that stream
_CAP_HINT = {}      # (P, W, H, n_views) -> instances/capacity that worked last time


def _initial_capacity(P, W, H, n_views):
    hint = _CAP_HINT.get((P, W, H, n_views))
    if hint:
        return hint
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    return int(n_views * max(4 * P, 4 * tiles, 1 << 14))


def _check_inputs(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, view2gaussian_precomp=None,
                  n_views=1):
    """Shapes the kernels rely on (they receive raw pointers): a mismatch must be a Python error, not an out-of-bounds access."""
    if means3D.ndim != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")     # AT_ERROR, rasterize_points.cu:61-63
    P = means3D.size(0)

    def want(name, t, numel, last=None):
        if t is None or t.numel() == 0:
            return
        if t.numel() != numel or (last is not None and t.ndim >= 2 and t.size(-1) != last):
            raise RuntimeError(f"{name} has shape {tuple(t.shape)}; expected {numel} elements" +
                               (f" with last dimension {last}" if last else "") + f" for {P} points")
    if opacities is None or opacities.numel() != P:
        raise RuntimeError(f"opacities must have {P} elements (num_points, 1), got "
                           f"{None if opacities is None else tuple(opacities.shape)}")
    want("scales", scales, 3 * P, 3)
    want("rotations", rotations, 4 * P, 4)
    want("colors_precomp", colors_precomp, 3 * P, 3)
    want("cov3D_precomp", cov3Ds_precomp, 6 * P, 6)
    want("view2gaussian_precomp", view2gaussian_precomp, 10 * P * n_views, 10)
    if sh is not None and sh.numel():
        if sh.ndim != 3 or sh.size(0) != P or sh.size(2) != 3:
            raise RuntimeError(f"shs must have dimensions (num_points, M, 3), got {tuple(sh.shape)}")


def rasterize_views(means3D, opacities, viewmatrices, projmatrices, camposs, bg, *, image_height, image_width,
                    tanfovx, tanfovy, sh=None, colors_precomp=None, scales=None, rotations=None, cov3Ds_precomp=None,
                    view2gaussian_precomp=None, sh_degree=0, scale_modifier=1.0, kernel_size=0.0, workspace=None,
                    max_rendered=None, save_aux=False, out=None, radii=None, check=True, n_sets=1, channels="all",
                    exact=None, tile_cull=None, small_path=None, scan=None):
    """Render ``n_views`` cameras of the same Gaussians in ONE launch sequence (f3dg_forward_batched).

    viewmatrices / projmatrices: [V,4,4] (any leading singleton dims), camposs [V,3], bg [3] or [V,3].
    Returns (color [V,9,H,W], radii [V,P] int32, workspace). With ``check=False`` nothing synchronises; the
    caller may later call ``read_status(workspace)``. With ``check=True`` an overflow grows the workspace and
    re-runs the call (the analogue of the reference's resize callback).

    ``n_sets`` > 1 (f3dg_forward_sets): the Gaussian tensors hold n_sets sets of equal size one after the other
    ([n_sets * P, ...]) and the V cameras are n_sets groups of V / n_sets (set-major): camera i renders set i // (V / n_sets).
    This is how the cycle aggregation renders the 8 novel views of every image of a batch in one launch sequence.

    ``channels="rgb_depth_alpha"`` (inference only; the build's own batched loops): the compositing kernel neither accumulates nor
    writes the normal (3..5) and distortion (8) planes of ``color`` -- they hold whatever the buffer held -- and the channels it
    does write (0..2, 6, 7) are bit-identical to the 9-channel call (F3DG_FLAG_SKIP_NORMAL | F3DG_FLAG_SKIP_DISTORTION).

    Per-call settings (``None`` = the process-wide default of ``f3dg_set_option``): ``exact=True`` composites this call in the
    reference's float32 / float64 operation order (F3DG_FLAG_EXACT: what a consumer of the distortion channel wants), ``exact=False``
    in the fast arithmetic (F3DG_FLAG_FAST); ``tile_cull=False`` builds the reference's tile lists (F3DG_FLAG_NO_TILE_CULL);
    ``small_path=False`` keeps one- and two-view calls on the general launch sequence (F3DG_FLAG_NO_SMALL_PATH);
    ``scan=True`` (fast inference calls of the general launch sequence) composites with the split-pixel schedule of
    csrc/f3dg_render5.hip (F3DG_FLAG_SCAN): same blended entries per pixel, sums associated as a segmented wave scan -- within 1e-4 of
    the reference like every fast call, not bit-identical to ``scan=False``.
    """
    L = _lib.lib()
    device = means3D.device
    if device.type != "cuda":
        raise RuntimeError("f3dgaus_amd rasterizer needs tensors on a HIP device (no CPU fallback)")
    n_sets = int(n_sets)
    H, W = int(image_height), int(image_width)
    vm = _dev_f32(viewmatrices, device).reshape(-1, 16)
    V = vm.size(0)
    if n_sets < 1 or V % n_sets or (means3D.ndim == 2 and means3D.size(0) % n_sets):
        raise RuntimeError("n_sets must divide the number of views and the number of Gaussians")
    if n_sets > 1 and (save_aux or view2gaussian_precomp is not None):
        raise RuntimeError("n_sets > 1 is an inference path: no SAVE_AUX / backward, no view2gaussian_precomp")
    _check_inputs(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, view2gaussian_precomp, V)
    P = means3D.size(0) // n_sets
    pm = _dev_f32(projmatrices, device).reshape(-1, 16)
    cp = _dev_f32(camposs, device).reshape(-1, 3)
    bgt = _dev_f32(bg, device).reshape(-1, 3)
    if pm.size(0) != V or cp.size(0) != V or bgt.size(0) not in (1, V):
        raise RuntimeError("viewmatrices, projmatrices, camposs (and bg if per view) must agree on the number of views")
    flags = (_lib.FLAG_SAVE_AUX if save_aux else 0) | (_lib.FLAG_BG_PER_VIEW if (bgt.size(0) == V and V > 1) else 0)
    if channels == "rgb_depth_alpha":
        if save_aux:
            raise RuntimeError('channels="rgb_depth_alpha" is an inference option (the backward reads all nine channels\' state)')
        flags |= _lib.FLAG_SKIP_NORMAL | _lib.FLAG_SKIP_DISTORTION
    elif channels != "all":
        raise RuntimeError('channels must be "all" or "rgb_depth_alpha"')
    if exact is not None:
        flags |= _lib.FLAG_EXACT if exact else _lib.FLAG_FAST
    if tile_cull is not None and not tile_cull:
        flags |= _lib.FLAG_NO_TILE_CULL
    if small_path is not None and not small_path:
        flags |= _lib.FLAG_NO_SMALL_PATH
    if scan:
        flags |= _lib.FLAG_SCAN

    means3D = _dev_f32(means3D, device)
    sh = _dev_f32(sh, device)
    colors_precomp = _dev_f32(colors_precomp, device)
    opacities = _dev_f32(opacities, device)
    scales = _dev_f32(scales, device)
    rotations = _dev_f32(rotations, device)
    cov3Ds_precomp = _dev_f32(cov3Ds_precomp, device)
    view2gaussian_precomp = _dev_f32(view2gaussian_precomp, device)
    M = 0 if sh is None else (sh.size(1) if sh.ndim == 3 else sh.numel() // (3 * max(P, 1)))

    if out is None:
        out = torch.empty((V, 9, H, W), dtype=torch.float32, device=device)
    elif out.dtype != torch.float32 or out.device != device or not out.is_contiguous() or out.numel() != V * 9 * H * W:
        raise RuntimeError(f"out must be a contiguous float32 tensor of shape ({V}, 9, {H}, {W}) on {device}")
    if radii is None:
        radii = torch.empty((V, P), dtype=torch.int32, device=device)
    elif radii.dtype != torch.int32 or radii.device != device or not radii.is_contiguous() or radii.numel() != V * P:
        raise RuntimeError(f"radii must be a contiguous int32 tensor of shape ({V}, {P}) on {device}")

    cap = int(max_rendered) if max_rendered is not None else (workspace.max_rendered if workspace is not None else _initial_capacity(P, W, H, V))
    while True:
        if workspace is None or not workspace.fits(P, W, H, V, cap):
            workspace = Workspace(P, W, H, V, cap, device)
        rc = L.f3dg_forward_sets(
            _stream(), C.c_void_p(workspace.buffer.data_ptr()), workspace.nbytes, workspace.max_rendered, n_sets, V // n_sets, P,
            int(sh_degree), int(M), _lib.ptr(bgt), W, H, _lib.ptr(means3D), _lib.ptr(sh), _lib.ptr(colors_precomp),
            _lib.ptr(opacities), _lib.ptr(scales), float(scale_modifier), _lib.ptr(rotations),
            _lib.ptr(cov3Ds_precomp), _lib.ptr(view2gaussian_precomp), _lib.ptr(vm), _lib.ptr(pm), _lib.ptr(cp),
            float(tanfovx), float(tanfovy), float(kernel_size), _lib.ptr(out), _lib.ptr(radii), flags)
        _lib.check(rc, "f3dg_forward_sets")
        workspace.save_aux = bool(save_aux)
        if not check:
            workspace.num_rendered = None
            return out, radii, workspace
        n = C.c_longlong(0)
        rc = L.f3dg_read_status(_stream(), C.c_void_p(workspace.buffer.data_ptr()), C.byref(n))
        if rc == _lib.ERR_OVERFLOW:
            cap = int(n.value * 1.25) + 1024          # grow and retry
            workspace = None
            continue
        _lib.check(rc, "f3dg_read_status")
        workspace.num_rendered = int(n.value)
        _CAP_HINT[(P, W, H, V)] = max(int(n.value * 1.5) + 1024, 1 << 14)
        return out, radii, workspace


# ---- deferred status (opt-in): the reference's loops render ONE view per call and never look at num_rendered; its contract of a
# status per call (rasterizer_impl.cu:336) costs a host synchronisation per view. With `set_deferred_status(True)` the no-grad calls of
# `GaussianRasterizer_GOF` / `render_predicted_more_v2_gof` post the header read-back behind the call (f3dg_status_post) and check it
# when the NEXT call arrives on the same stream (by then it has long landed) or at `flush()`. An overflow found that way re-issues the
# call it belongs to with a grown workspace INTO THE SAME output tensors (and re-runs whatever the wrapper derived from them), and
# warns: anything the caller enqueued on those outputs in between read an incomplete frame. The first call of a shape -- no capacity
# is known for it yet -- is always checked at once. Default: off (the reference's blocking contract).
_DEFERRED = {"on": False, "depth": 1}
_PENDING = {}       # (device index, stream) -> [oldest .. newest] of {"ticket": int, "ws": Workspace, "redo": [callables], "shape": key of _CAP_HINT}


def set_deferred_status(on=True, depth=1):
    """Opt in / out of the deferred status check of no-grad single-view calls (see the comment above). Turning it off flushes.

    ``depth``: how many calls may be unchecked on a stream. 1 (default): call k is checked when call k + 1 arrives -- the host then waits
    for call k's kernels before it issues the next call's, and the device idles for the ~25 us that takes. 2: call k is checked when
    call k + 2 arrives; the host runs a whole call ahead and the device never waits for it (one-view loop at 65,536 Gaussians: 7.4 k -> 8.9 k
    views/s, `profiles/r05_final/one_view.md`). The caveats below then hold for two calls instead of one.

    Caveats of the opt-in mode, both about what happens between a call and the moment its status is looked at (the next call on the
    stream, or `flush()`): (1) consumers of the call's OUTPUTS enqueued in between read an incomplete frame if the call overflowed
    (the repair re-renders into the same tensors and warns); (2) the repair re-renders from the caller's INPUT tensors as they are
    THEN: a caller that refills those buffers in place between the call and the check (the cycle loop does: `splat_head(out=merged)`)
    gets the repaired frame from the new Gaussians. Keep the inputs of a deferred call untouched until the next call / `flush()`, or
    size the workspace so that the overflow cannot happen (the first call of a shape is always checked at once and sets the hint)."""
    if not on:
        flush()
    _DEFERRED["on"] = bool(on)
    _DEFERRED["depth"] = max(1, int(depth))


def deferred_status():
    return _DEFERRED["on"]


def _resolve(key, keep=0):
    """Checks the oldest unchecked calls of the stream until at most ``keep`` are left."""
    q = _PENDING.get(key)
    while q and len(q) > keep:
        try:
            _resolve_one(q[0])      # (an exception while REPAIRING an overflow -- no memory for the grown workspace -- leaves the entry queued:
        except Exception:           # the next call or flush() tries the repair again instead of losing it; a failed poll has released its
            if "polled" not in q[0]:    # ticket and cannot be repeated)
                q.pop(0)
            raise
        q.pop(0)
    if q is not None and not q:
        del _PENDING[key]


def _resolve_one(p):
    if "polled" not in p:           # (the poll releases the ticket: a retry after a failed repair must not poll it again)
        n = C.c_longlong(0)
        rc = _lib.lib().f3dg_status_poll(p["ticket"], 1, C.byref(n))
        if rc != _lib.ERR_OVERFLOW:
            _lib.check(rc, "f3dg_status_poll")
        p["polled"] = (rc, n)
    rc, n = p["polled"]
    if rc == _lib.ERR_OVERFLOW:
        import warnings
        warnings.warn("f3dgaus_amd: a rasterizer call whose status check was deferred needed %d instances, more than its workspace "
                      "held; it has been re-issued into the same output tensors -- work enqueued on them in between saw an "
                      "incomplete frame" % n.value)
        _CAP_HINT[p["shape"]] = max(int(n.value * 1.5) + 1024, 1 << 14)
        for fn in p["redo"]:
            fn()
        return
    _lib.check(rc, "f3dg_status_poll")
    p["ws"].num_rendered = int(n.value)
    _CAP_HINT[p["shape"]] = max(_CAP_HINT.get(p["shape"], 0), int(n.value * 1.5) + 1024, 1 << 14)


def flush(device=None):
    """Checks every deferred status (waits for the calls they belong to). Call it before relying on the frames of a loop that ran
    with `set_deferred_status(True)`; `torch.cuda.synchronize()` alone does not look at the overflow flags."""
    for key in list(_PENDING):
        if device is None or key[0] == torch.device(device).index:
            _resolve(key)


def add_redo(fn):
    """(wrapper-internal) work derived from the outputs of the call just issued on the current stream, to be repeated if that call
    turns out to have overflowed."""
    dev = torch.cuda.current_device()
    q = _PENDING.get((dev, _raw_stream(dev)))
    if q:
        q[-1]["redo"].append(fn)


def read_status(workspace):
    """BLOCKING: number of (Gaussian, tile) instances of the last call on this workspace; raises on overflow."""
    n = C.c_longlong(0)
    rc = _lib.lib().f3dg_read_status(_stream(), C.c_void_p(workspace.buffer.data_ptr()), C.byref(n))
    _lib.check(rc, "f3dg_read_status")
    workspace.num_rendered = int(n.value)
    return workspace.num_rendered


def cpu_deep_copy_tuple(input_tuple):
    """rast_py:15-17: tensors cloned to the host, everything else as is."""
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        view2gaussian_precomp, raster_settings, exact=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, view2gaussian_precomp, raster_settings, exact)


def _forward_impl(rs, needs_grad, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, view2gaussian_precomp,
                  exact=None, scan=None):
    """One reference-shaped rasterizer call (one view): (color [1,9,H,W], radii [1,P], workspace). Shared by the autograd Function
    and by the wrapper's inference path (`rasterize_nograd`)."""
    device = means3D.device
    key = (device.index, _raw_stream(device.index) if device.type == "cuda" else 0)

    shape_key = (means3D.size(0), int(rs.image_width), int(rs.image_height), 1)
    deferred = _DEFERRED["on"] and not needs_grad and not rs.debug and device.type == "cuda"
    if deferred:
        _resolve(key, keep=_DEFERRED["depth"] - 1)      # the call `depth` calls back on this stream: its status has landed long ago
        deferred = shape_key in _CAP_HINT             # no capacity known yet: check this one at once
    ws = None if needs_grad else _WS_CACHE.get(key)     # (after the checks: a repair replaces the cached workspace by a larger one)

    def call(check=True, out=None, radii=None, workspace=ws, max_rendered=None):
        return rasterize_views(
            means3D, opacities, rs.viewmatrix, rs.projmatrix, rs.campos, rs.bg.reshape(-1)[:3],
            image_height=rs.image_height, image_width=rs.image_width, tanfovx=rs.tanfovx, tanfovy=rs.tanfovy, sh=sh,
            colors_precomp=colors_precomp, scales=scales, rotations=rotations, cov3Ds_precomp=cov3Ds_precomp,
            view2gaussian_precomp=view2gaussian_precomp, sh_degree=rs.sh_degree, scale_modifier=rs.scale_modifier,
            kernel_size=rs.kernel_size, workspace=workspace, save_aux=needs_grad, check=check, out=out, radii=radii,
            max_rendered=max_rendered, exact=exact, scan=scan if not needs_grad else None)

    if rs.debug:
        # rast_py:88-98: keep a host copy of the arguments (in the order of the reference's tuple, rast_py:61-84) and dump
        # it if the rasterizer fails
        cpu_args = cpu_deep_copy_tuple((rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier,
                                        cov3Ds_precomp, view2gaussian_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
                                        rs.tanfovy, rs.kernel_size, rs.subpixel_offset, rs.image_height, rs.image_width, sh,
                                        rs.sh_degree, rs.campos, rs.prefiltered, rs.debug))
        try:
            color, radii, ws = call()
            torch.cuda.synchronize(device)       # debug mode: surface asynchronous HIP errors here (auxiliary.h:204-211)
        except Exception as ex:
            torch.save(cpu_args, "snapshot_fw.dump")
            print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
            raise ex
    elif deferred:
        if ws is not None and ws.max_rendered < _CAP_HINT[shape_key]:
            ws = None                                  # (the hint grew after an overflow: a workspace of that capacity)
        color, radii, ws = call(check=False, workspace=ws, max_rendered=None if ws is not None else _CAP_HINT[shape_key])
        ticket = _lib.check(_lib.lib().f3dg_status_post(_stream(), C.c_void_p(ws.buffer.data_ptr())), "f3dg_status_post")

        def redo(color=color, radii=radii):
            _, _, ws2 = call(check=True, out=color, radii=radii, workspace=None)
            _WS_CACHE[key] = ws2
        _PENDING.setdefault(key, []).append({"ticket": ticket, "ws": ws, "redo": [redo], "shape": shape_key})
    else:
        color, radii, ws = call()
    if not needs_grad:
        _WS_CACHE[key] = ws
    return color, radii, ws


def rasterize_nograd(means3D, sh, colors_precomp, opacities, scales, rotations, raster_settings, exact=None, scan=None):
    """The inference call of `GaussianRasterizer_GOF.forward` without the nn.Module and autograd.Function around it (the wrapper's
    own no-grad path: ~30 us of host time per call less). Returns (color [9,H,W], radii [P]). ``exact``, ``scan``: see `rasterize_views`
    (a one-view call with ``scan=True`` composites with four lanes per pixel: render5p_fwd_kernel)."""
    color, radii, _ = _forward_impl(raster_settings, False, means3D, sh, colors_precomp, opacities, scales, rotations, None, None,
                                    exact=exact, scan=scan)
    return color[0], radii[0]


class _RasterizeGaussians(torch.autograd.Function):
    """Autograd wrapper; argument order and the (color, radii) result follow rast_py:46-104, the gradient order
    rast_py:152-165."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                view2gaussian_precomp, raster_settings, exact=None):
        rs = raster_settings
        needs_grad = any(ctx.needs_input_grad)      # (grad mode is always off inside Function.forward)
        color, radii, ws = _forward_impl(rs, needs_grad, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                         view2gaussian_precomp, exact=exact)
        ctx.raster_settings = rs
        ctx.num_rendered = ws.num_rendered
        ctx.workspace = ws if needs_grad else None
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, view2gaussian_precomp,
                              radii, sh)
        color = color[0]
        radii = radii[0]
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        from .backward import rasterize_backward
        rs = ctx.raster_settings
        if not rs.debug:
            return rasterize_backward(ctx, grad_out_color)
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, view2gaussian_precomp, radii, sh = ctx.saved_tensors
        # rast_py:138-150 (argument order of rast_py:113-136; the three buffers of the reference are this build's one workspace)
        cpu_args = cpu_deep_copy_tuple((rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier,
                                        cov3Ds_precomp, view2gaussian_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
                                        rs.tanfovy, rs.kernel_size, rs.subpixel_offset, grad_out_color, sh, rs.sh_degree,
                                        rs.campos, ctx.num_rendered, rs.debug))
        try:
            grads = rasterize_backward(ctx, grad_out_color)
            torch.cuda.synchronize(means3D.device)
        except Exception as ex:
            torch.save(cpu_args, "snapshot_bw.dump")
            print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
            raise ex
        return grads


_INTEG_WS = _StreamCache()      # (device index, stream) -> (key, capacity, buffer) reused by integrate calls on that stream


def integrate_gaussians_to_points(points3D, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                  view2gaussian_precomp, raster_settings):
    """``_C.integrate_gaussians_to_points`` (RAST/rasterize_points.cu:233-343) through ``f3dg_integrate``.
    Returns (color [9,H,W], alpha_integrated [PN], color_integrated [PN,3], radii [P], num_rendered)."""
    L = _lib.lib()
    device = means3D.device
    if device.type != "cuda":
        raise RuntimeError("f3dgaus_amd rasterizer needs tensors on a HIP device (no CPU fallback)")
    if means3D.ndim != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")     # rasterize_points.cu:260-262
    if points3D.ndim != 2 or points3D.size(1) != 3:
        raise RuntimeError("points3D must have dimensions (num_points, 3)")    # rasterize_points.cu:263-265
    rs = raster_settings
    with torch.no_grad():
        P, PN = means3D.size(0), points3D.size(0)
        H, W = int(rs.image_height), int(rs.image_width)
        points3D = points3D.to(device=device, dtype=torch.float32).contiguous()
        means3D_ = _dev_f32(means3D, device)
        sh = _dev_f32(sh, device)
        colors_precomp = _dev_f32(colors_precomp, device)
        opacities_ = _dev_f32(opacities, device)
        scales = _dev_f32(scales, device)
        rotations = _dev_f32(rotations, device)
        cov3Ds_precomp = _dev_f32(cov3Ds_precomp, device)
        view2gaussian_precomp = _dev_f32(view2gaussian_precomp, device)
        vm = _dev_f32(rs.viewmatrix, device)
        pm = _dev_f32(rs.projmatrix, device)
        cp = _dev_f32(rs.campos, device)
        bgt = _dev_f32(rs.bg, device)
        M = 0 if sh is None else (sh.size(1) if sh.ndim == 3 else sh.numel() // (3 * max(P, 1)))

        color = torch.empty((9, H, W), dtype=torch.float32, device=device)
        radii = torch.zeros((P,), dtype=torch.int32, device=device)
        alpha_integrated = torch.empty((PN,), dtype=torch.float32, device=device)
        color_integrated = torch.empty((PN, 3), dtype=torch.float32, device=device)

        cap = _initial_capacity(P, W, H, 1)
        key = (P, PN, W, H)
        while True:
            ckey = (device.index, _raw_stream(device.index))
            cached = _INTEG_WS.get(ckey)
            if cached is None or cached[0] != key or cached[1] < cap:
                nbytes = L.f3dg_integrate_workspace_bytes(P, PN, W, H, cap)
                if nbytes == 0:
                    raise _lib.F3dgError(_lib.ERR_BAD_ARG, "f3dg_integrate_workspace_bytes")
                cached = (key, cap, torch.empty(int(nbytes), dtype=torch.uint8, device=device))
                _INTEG_WS[ckey] = cached
            buf = cached[2]
            needed = C.c_longlong(0)
            rc = L.f3dg_integrate(
                _stream(), C.c_void_p(buf.data_ptr()), buf.numel(), cached[1], PN, P, int(rs.sh_degree), int(M),
                _lib.ptr(bgt), W, H, _lib.ptr(points3D) if PN else None, _lib.ptr(means3D_), _lib.ptr(sh),
                _lib.ptr(colors_precomp), _lib.ptr(opacities_), _lib.ptr(scales), float(rs.scale_modifier),
                _lib.ptr(rotations), _lib.ptr(cov3Ds_precomp), _lib.ptr(view2gaussian_precomp), _lib.ptr(vm),
                _lib.ptr(pm), _lib.ptr(cp), float(rs.tanfovx), float(rs.tanfovy), float(rs.kernel_size), None,
                int(bool(rs.prefiltered)), _lib.ptr(color), _lib.ptr(radii), _lib.ptr(alpha_integrated) if PN else None,
                _lib.ptr(color_integrated) if PN else None, C.byref(needed))
            if rc == _lib.ERR_OVERFLOW:
                cap = int(needed.value * 1.25) + 1024
                continue
            _lib.check(rc, "f3dg_integrate")
            _CAP_HINT[(P, W, H, 1)] = max(int(rc * 1.5) + 1024, 1 << 14)
            return color, alpha_integrated, color_integrated, radii, int(rc)


class PreparedIntegration:
    """Everything of ``integrate`` that does not depend on the points, kept on the device for one camera: the workspace with
    the projected Gaussians, the sorted tile lists and the per-pixel contributor table, and the [9,H,W] image of the
    per-pixel pass. Built by ``integrate_prepare``; ``integrate_points`` runs any number of point sets against it.
    (MI355X-first: ~0.25 GB per camera at 589,824 Gaussians / 256^2, so the 129 cameras of a mesh-extraction sweep stay
    resident in the 288 GB of HBM instead of being recomputed for each of its 9 point sets.)"""

    def __init__(self, buffer, capacity, max_points, P, W, H, tanfovx, tanfovy, viewmatrix, color, radii, num_rendered,
                 n_views=1, view=0):
        self.buffer, self.capacity, self.max_points, self.P, self.W, self.H = buffer, capacity, max_points, P, W, H
        self.tanfovx, self.tanfovy, self.viewmatrix = tanfovx, tanfovy, viewmatrix
        self.color, self.radii, self.num_rendered = color, radii, num_rendered
        self.n_views, self.view = n_views, view       # camera `view` of `n_views` prepared together in `buffer` (integrate_prepare_batched)


def integrate_prepare_batched(means3D, sh, colors_precomp, opacities, scales, rotations, viewmatrices, projmatrices, camposs, bg, *,
                              image_height, image_width, tanfovx, tanfovy, sh_degree, max_points, scale_modifier=1.0, kernel_size=0.0):
    """``f3dg_integrate_prepare_batched``: projection + binning + per-pixel pass of ``integrate`` for V cameras of the same Gaussians
    in ONE launch sequence (the per-pixel pass of one 256^2 camera is 256 workgroups: a wave per SIMD). Returns a list of V
    ``PreparedIntegration`` sharing one workspace; each is bit-identical to ``integrate_prepare`` of its camera."""
    L = _lib.lib()
    device = means3D.device
    if device.type != "cuda":
        raise RuntimeError("f3dgaus_amd rasterizer needs tensors on a HIP device (no CPU fallback)")
    if means3D.ndim != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    with torch.no_grad():
        P = means3D.size(0)
        if P == 0:
            raise RuntimeError("integrate_prepare needs at least one Gaussian")
        H, W = int(image_height), int(image_width)
        means3D_, sh, colors_precomp = _dev_f32(means3D, device), _dev_f32(sh, device), _dev_f32(colors_precomp, device)
        opacities_, scales, rotations = _dev_f32(opacities, device), _dev_f32(scales, device), _dev_f32(rotations, device)
        vm = _dev_f32(viewmatrices, device).reshape(-1, 16).clone()
        V = vm.size(0)
        pm = _dev_f32(projmatrices, device).reshape(V, 16)
        cp = _dev_f32(camposs, device).reshape(V, 3)
        bgt = _dev_f32(bg, device).reshape(-1)[:3].contiguous()
        M = 0 if sh is None else (sh.size(1) if sh.ndim == 3 else sh.numel() // (3 * max(P, 1)))
        color = torch.empty((V, 9, H, W), dtype=torch.float32, device=device)
        radii = torch.zeros((V, P), dtype=torch.int32, device=device)
        cap = V * _initial_capacity(P, W, H, 1)
        while True:
            nbytes = L.f3dg_integrate_workspace_bytes_batched(P, int(max_points), W, H, V, cap)
            if nbytes == 0:
                raise _lib.F3dgError(_lib.ERR_BAD_ARG, "f3dg_integrate_workspace_bytes_batched")
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            needed = C.c_longlong(0)
            rc = L.f3dg_integrate_prepare_batched(
                _stream(), C.c_void_p(buf.data_ptr()), buf.numel(), cap, V, int(max_points), P, int(sh_degree), int(M),
                _lib.ptr(bgt), W, H, _lib.ptr(means3D_), _lib.ptr(sh), _lib.ptr(colors_precomp), _lib.ptr(opacities_),
                _lib.ptr(scales), float(scale_modifier), _lib.ptr(rotations), None, None, _lib.ptr(vm), _lib.ptr(pm), _lib.ptr(cp),
                float(tanfovx), float(tanfovy), float(kernel_size), _lib.ptr(color), _lib.ptr(radii), C.byref(needed))
            if rc == _lib.ERR_OVERFLOW:
                cap = int(needed.value * 1.25) + 1024
                continue
            _lib.check(rc, "f3dg_integrate_prepare_batched")
            _CAP_HINT[(P, W, H, 1)] = max(int(rc / V * 1.5) + 1024, 1 << 14)
            return [PreparedIntegration(buf, cap, int(max_points), P, W, H, float(tanfovx), float(tanfovy), vm[v], color[v], radii[v],
                                        int(rc), n_views=V, view=v) for v in range(V)]


def integrate_prepare(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, view2gaussian_precomp,
                      raster_settings, max_points, buffer=None):
    """``f3dg_integrate_prepare``: projection + binning + per-pixel pass of ``integrate`` for one camera. ``buffer``: an uint8 device
    tensor to carve the workspace from (used when it is large enough for the capacity hint; a fresh allocation otherwise)."""
    L = _lib.lib()
    device = means3D.device
    if device.type != "cuda":
        raise RuntimeError("f3dgaus_amd rasterizer needs tensors on a HIP device (no CPU fallback)")
    if means3D.ndim != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    rs = raster_settings
    with torch.no_grad():
        P = means3D.size(0)
        if P == 0:
            raise RuntimeError("integrate_prepare needs at least one Gaussian")
        H, W = int(rs.image_height), int(rs.image_width)
        means3D_ = _dev_f32(means3D, device)
        sh = _dev_f32(sh, device)
        colors_precomp = _dev_f32(colors_precomp, device)
        opacities_ = _dev_f32(opacities, device)
        scales = _dev_f32(scales, device)
        rotations = _dev_f32(rotations, device)
        cov3Ds_precomp = _dev_f32(cov3Ds_precomp, device)
        view2gaussian_precomp = _dev_f32(view2gaussian_precomp, device)
        vm = _dev_f32(rs.viewmatrix, device).reshape(16).clone()
        pm = _dev_f32(rs.projmatrix, device)
        cp = _dev_f32(rs.campos, device)
        bgt = _dev_f32(rs.bg, device)
        M = 0 if sh is None else (sh.size(1) if sh.ndim == 3 else sh.numel() // (3 * max(P, 1)))
        color = torch.empty((9, H, W), dtype=torch.float32, device=device)
        radii = torch.zeros((P,), dtype=torch.int32, device=device)
        cap = _initial_capacity(P, W, H, 1)
        while True:
            nbytes = L.f3dg_integrate_workspace_bytes(P, int(max_points), W, H, cap)
            if nbytes == 0:
                raise _lib.F3dgError(_lib.ERR_BAD_ARG, "f3dg_integrate_workspace_bytes")
            if buffer is not None and buffer.numel() >= nbytes:
                buf, buffer = buffer[:int(nbytes)], None
            else:
                buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            needed = C.c_longlong(0)
            rc = L.f3dg_integrate_prepare(
                _stream(), C.c_void_p(buf.data_ptr()), buf.numel(), cap, int(max_points), P, int(rs.sh_degree), int(M),
                _lib.ptr(bgt), W, H, _lib.ptr(means3D_), _lib.ptr(sh), _lib.ptr(colors_precomp), _lib.ptr(opacities_),
                _lib.ptr(scales), float(rs.scale_modifier), _lib.ptr(rotations), _lib.ptr(cov3Ds_precomp),
                _lib.ptr(view2gaussian_precomp), _lib.ptr(vm), _lib.ptr(pm), _lib.ptr(cp), float(rs.tanfovx),
                float(rs.tanfovy), float(rs.kernel_size), _lib.ptr(color), _lib.ptr(radii), C.byref(needed))
            if rc == _lib.ERR_OVERFLOW:
                cap = int(needed.value * 1.25) + 1024
                continue
            _lib.check(rc, "f3dg_integrate_prepare")
            _CAP_HINT[(P, W, H, 1)] = max(int(rc * 1.5) + 1024, 1 << 14)
            return PreparedIntegration(buf, cap, int(max_points), P, W, H, float(rs.tanfovx), float(rs.tanfovy), vm, color,
                                       radii, int(rc))


def integrate_points(prepared, points3D, alpha_min=None, want_outputs=True):
    """``f3dg_integrate_points``: one point set against a ``PreparedIntegration``. Returns (alpha_integrated [PN],
    color_integrated [PN,3]) (None, None with ``want_outputs=False``); ``alpha_min`` [PN] float32, if given, is updated in
    place to ``torch.min(alpha_min, alpha_integrated)`` -- the accumulation of visualize.py:463 without a second kernel."""
    pr = prepared
    device = pr.buffer.device
    if points3D.ndim != 2 or points3D.size(1) != 3:
        raise RuntimeError("points3D must have dimensions (num_points, 3)")
    PN = points3D.size(0)
    if PN > pr.max_points:
        raise RuntimeError(f"the integration was prepared for at most {pr.max_points} points, got {PN}")
    with torch.no_grad():
        pts = points3D.to(device=device, dtype=torch.float32).contiguous()
        ai = torch.empty((PN,), dtype=torch.float32, device=device) if want_outputs else None
        ci = torch.empty((PN, 3), dtype=torch.float32, device=device) if want_outputs else None
        if alpha_min is not None and (alpha_min.dtype != torch.float32 or not alpha_min.is_contiguous() or alpha_min.numel() != PN
                                      or alpha_min.device != device):
            raise RuntimeError("alpha_min must be a contiguous float32 tensor of PN elements on the same device")
        rc = _lib.lib().f3dg_integrate_points_view(
            _stream(), C.c_void_p(pr.buffer.data_ptr()), pr.buffer.numel(), pr.capacity, pr.n_views, pr.view, PN, pr.P, pr.W, pr.H, _lib.ptr(pts),
            _lib.ptr(pr.viewmatrix), pr.tanfovx, pr.tanfovy, _lib.ptr(pr.color), _lib.ptr(ai), _lib.ptr(ci), _lib.ptr(alpha_min))
        _lib.check(rc, "f3dg_integrate_points")
    return ai, ci


class GaussianRasterizer_GOF(nn.Module):
    def __init__(self, raster_settings, exact=None):
        """``raster_settings``: the reference's 14-field tuple (rast_py:168-182). ``exact`` (this build's one addition, keyword only in
        spirit): True / False select the compositing arithmetic of THIS rasterizer's calls (see `rasterize_views`), None the process default."""
        super().__init__()
        self.raster_settings = raster_settings
        self.exact = exact

    def markVisible(self, positions):
        """bool[P]: view-space z > 0.2 (rast_py:190-199 -> rasterizer_impl.cu:172-186)."""
        with torch.no_grad():
            rs = self.raster_settings
            pos = _dev_f32(positions, positions.device)
            P = pos.size(0)
            present = torch.zeros((P,), dtype=torch.uint8, device=pos.device)
            vm = _dev_f32(rs.viewmatrix, pos.device)
            pm = _dev_f32(rs.projmatrix, pos.device)
            rc = _lib.lib().f3dg_mark_visible(_stream(), P, _lib.ptr(pos), _lib.ptr(vm), _lib.ptr(pm), _lib.ptr(present))
            _lib.check(rc, "f3dg_mark_visible")
        return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, view2gaussian_precomp=None):
        raster_settings = self.raster_settings

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')

        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        empty = torch.Tensor([])
        if shs is None:
            shs = empty
        if colors_precomp is None:
            colors_precomp = empty
        if scales is None:
            scales = empty
        if rotations is None:
            rotations = empty
        if cov3D_precomp is None:
            cov3D_precomp = empty
        if view2gaussian_precomp is None:
            view2gaussian_precomp = empty

        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, view2gaussian_precomp, raster_settings, self.exact)

    def integrate(self, points3D, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                  rotations=None, cov3D_precomp=None, view2gaussian_precomp=None):
        """Integrate the Gaussians' opacity along the rays to ``points3D`` (rast_py:241-307 ->
        ``_C.integrate_gaussians_to_points``, rasterize_points.cu:233-343). Not differentiable, as in the reference.
        Returns (color [9,H,W], alpha_integrated [PN], color_integrated [PN,3], radii [P])."""
        raster_settings = self.raster_settings

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')

        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        color, alpha_integrated, color_integrated, radii, _ = integrate_gaussians_to_points(
            points3D, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, view2gaussian_precomp,
            raster_settings)
        return color, alpha_integrated, color_integrated, radii
