"""ctypes front-end of ``libgof_oracle.so`` (the plain-C restatement in ``gof_oracle.c``).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``. Works on numpy float32 arrays.
Mirrors the argument order of ``CudaRasterizer::Rasterizer::forward``
(reference RAST/cuda_rasterizer/rasterizer.h:31-55, rasterizer_impl.cu:247-405).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libgof_oracle.so")
    src = os.path.join(_HERE, "gof_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libgof_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.gof_oracle_create.restype = C.c_void_p
        L.gof_oracle_destroy.argtypes = [C.c_void_p]
        fp, ip, vp = C.c_void_p, C.c_void_p, C.c_void_p
        L.gof_oracle_forward.restype = C.c_int
        L.gof_oracle_forward.argtypes = [vp, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, fp, fp, fp, fp,
                                         C.c_float, fp, fp, fp, fp, fp, fp, C.c_float, C.c_float, C.c_float, fp, ip]
        for name in ("depths", "means2D", "cov3D", "v2g", "conic_opacity", "rgb", "clamped", "tiles_touched",
                     "point_offsets", "keys_sorted", "point_list", "ranges", "final_T", "n_contrib"):
            f = getattr(L, "gof_oracle_" + name)
            f.restype = C.c_void_p
            f.argtypes = [vp]
        L.gof_oracle_num_rendered.restype = C.c_int
        L.gof_oracle_num_rendered.argtypes = [vp]
        L.gof_oracle_higher_msb.restype = C.c_uint32
        L.gof_oracle_higher_msb.argtypes = [C.c_uint32]
        L.gof_oracle_integrate.restype = C.c_int
        L.gof_oracle_integrate.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, fp, fp, fp,
                                           fp, fp, C.c_float, fp, fp, fp, fp, fp, fp, C.c_float, C.c_float, C.c_float,
                                           fp, ip, fp, fp, ip]
        L.gof_oracle_backward.restype = None
        L.gof_oracle_backward.argtypes = [vp] + [fp] * 17
        _LIB = L
    return _LIB


def _f32(a):
    if a is None:
        return None
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _view(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return np.zeros(shape, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


class Oracle:
    """One forward (+ optional backward) evaluation context; keeps every intermediate buffer."""

    def __init__(self):
        self._L = lib()
        self._ctx = C.c_void_p(self._L.gof_oracle_create())
        self.args = None

    def __del__(self):
        try:
            self._L.gof_oracle_destroy(self._ctx)
        except Exception:
            pass

    def forward(self, *, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, W, H, bg,
                shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                view2gaussian_precomp=None, sh_degree=0, scale_modifier=1.0, kernel_size=0.0):
        means3D = _f32(means3D).reshape(-1, 3)
        P = means3D.shape[0]
        shs = _f32(shs)
        M = 0 if shs is None else shs.reshape(P, -1, 3).shape[1]
        a = dict(means3D=means3D, shs=shs, colors_precomp=_f32(colors_precomp), opacities=_f32(opacities),
                 scales=_f32(scales), rotations=_f32(rotations), cov3D_precomp=_f32(cov3D_precomp),
                 view2gaussian_precomp=_f32(view2gaussian_precomp), viewmatrix=_f32(viewmatrix).reshape(16),
                 projmatrix=_f32(projmatrix).reshape(16), campos=_f32(campos).reshape(3), bg=_f32(bg).reshape(3))
        out = np.zeros((9, H, W), np.float32)
        radii = np.zeros((P,), np.int32)
        R = self._L.gof_oracle_forward(
            self._ctx, P, int(sh_degree), M, _ptr(a["bg"]), int(W), int(H), _ptr(a["means3D"]), _ptr(a["shs"]),
            _ptr(a["colors_precomp"]), _ptr(a["opacities"]), _ptr(a["scales"]), C.c_float(scale_modifier),
            _ptr(a["rotations"]), _ptr(a["cov3D_precomp"]), _ptr(a["view2gaussian_precomp"]), _ptr(a["viewmatrix"]),
            _ptr(a["projmatrix"]), _ptr(a["campos"]), C.c_float(tanfovx), C.c_float(tanfovy), C.c_float(kernel_size),
            _ptr(out), _ptr(radii))
        self.args = a  # keep inputs alive: the context points into colors/v2g precomp
        self.P, self.W, self.H, self.M, self.D, self.R = P, W, H, M, int(sh_degree), R
        self.scale_modifier, self.kernel_size, self.tanfovx, self.tanfovy = scale_modifier, kernel_size, tanfovx, tanfovy
        self.out_color, self.radii = out, radii
        return out, radii, R

    def integrate(self, *, points3D, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, W, H, bg,
                  shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                  view2gaussian_precomp=None, sh_degree=0, scale_modifier=1.0, kernel_size=0.0):
        """``Rasterizer::integrate`` (rasterizer_impl.cu:530-792) with the output initialisation of
        ``IntegrateGaussiansToPointsCUDA`` (rasterize_points.cu:273-276). Returns
        (out_color[9,H,W], alpha_integrated[PN], color_integrated[PN,3], radii[P], num_rendered, num_integrated)."""
        means3D = _f32(means3D).reshape(-1, 3)
        points3D = _f32(points3D).reshape(-1, 3)
        P, PN = means3D.shape[0], points3D.shape[0]
        shs = _f32(shs)
        M = 0 if shs is None else shs.reshape(P, -1, 3).shape[1]
        a = dict(means3D=means3D, shs=shs, colors_precomp=_f32(colors_precomp), opacities=_f32(opacities),
                 scales=_f32(scales), rotations=_f32(rotations), cov3D_precomp=_f32(cov3D_precomp),
                 view2gaussian_precomp=_f32(view2gaussian_precomp), viewmatrix=_f32(viewmatrix).reshape(16),
                 projmatrix=_f32(projmatrix).reshape(16), campos=_f32(campos).reshape(3), bg=_f32(bg).reshape(3),
                 points3D=points3D)
        out = np.zeros((9, H, W), np.float32)
        radii = np.zeros((P,), np.int32)
        alpha_i = np.ones((PN,), np.float32)
        color_i = np.zeros((PN, 3), np.float32)
        ni = C.c_int(0)
        R = 0
        if P != 0 and PN != 0:          # rasterize_points.cu:300
            R = self._L.gof_oracle_integrate(
                self._ctx, PN, P, int(sh_degree), M, _ptr(a["bg"]), int(W), int(H), _ptr(points3D), _ptr(a["means3D"]),
                _ptr(a["shs"]), _ptr(a["colors_precomp"]), _ptr(a["opacities"]), _ptr(a["scales"]),
                C.c_float(scale_modifier), _ptr(a["rotations"]), _ptr(a["cov3D_precomp"]),
                _ptr(a["view2gaussian_precomp"]), _ptr(a["viewmatrix"]), _ptr(a["projmatrix"]), _ptr(a["campos"]),
                C.c_float(tanfovx), C.c_float(tanfovy), C.c_float(kernel_size), _ptr(out), _ptr(radii), _ptr(alpha_i),
                _ptr(color_i), C.byref(ni))
        self.args = a
        self.P, self.W, self.H, self.M, self.D, self.R = P, W, H, M, int(sh_degree), R
        return out, alpha_i, color_i, radii, R, int(ni.value)

    def backward(self, dL_dpix):
        """Gradients of the last forward w.r.t. its inputs given dL/d(out_color) [9,H,W]. Returns a dict with the
        nine tensors Rasterizer::backward produces (dL_dconic / dL_dcov3D are identically zero in the reference)."""
        a, P, M = self.args, self.P, self.M
        dL_dpix = _f32(dL_dpix).reshape(9, self.H, self.W)
        g = dict(dL_dmean2D=np.zeros((P, 3), np.float32), dL_dopacity=np.zeros((P, 1), np.float32),
                 dL_dcolor=np.zeros((P, 3), np.float32), dL_dmean3D=np.zeros((P, 3), np.float32),
                 dL_dsh=np.zeros((P, max(M, 0), 3), np.float32), dL_dscale=np.zeros((P, 3), np.float32),
                 dL_drot=np.zeros((P, 4), np.float32), dL_dview2gaussian=np.zeros((P, 10), np.float32))
        self._L.gof_oracle_backward(
            self._ctx, _ptr(a["bg"]), _ptr(a["means3D"]), _ptr(a["shs"]), _ptr(a["scales"]), _ptr(a["rotations"]),
            _ptr(a["viewmatrix"]), _ptr(a["campos"]), _ptr(self.radii), _ptr(dL_dpix),
            _ptr(g["dL_dmean2D"]), _ptr(g["dL_dopacity"]), _ptr(g["dL_dcolor"]), _ptr(g["dL_dmean3D"]),
            _ptr(g["dL_dsh"]), _ptr(g["dL_dscale"]), _ptr(g["dL_drot"]), _ptr(g["dL_dview2gaussian"]))
        g["dL_dconic"] = np.zeros((P, 2, 2), np.float32)
        g["dL_dcov3D"] = np.zeros((P, 6), np.float32)
        return g

    def intermediates(self):
        L, c, P, R = self._L, self._ctx, self.P, self.R
        T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        HW = self.H * self.W
        return dict(
            depths=_view(L.gof_oracle_depths(c), (P,), np.float32),
            means2D=_view(L.gof_oracle_means2D(c), (P, 2), np.float32),
            cov3D=_view(L.gof_oracle_cov3D(c), (P, 6), np.float32),
            view2gaussian=_view(L.gof_oracle_v2g(c), (P, 10), np.float32),
            conic_opacity=_view(L.gof_oracle_conic_opacity(c), (P, 4), np.float32),
            rgb=_view(L.gof_oracle_rgb(c), (P, 3), np.float32),
            clamped=_view(L.gof_oracle_clamped(c), (P, 3), np.uint8),
            tiles_touched=_view(L.gof_oracle_tiles_touched(c), (P,), np.uint32),
            point_offsets=_view(L.gof_oracle_point_offsets(c), (P,), np.uint32),
            keys_sorted=_view(L.gof_oracle_keys_sorted(c), (R,), np.uint64),
            point_list=_view(L.gof_oracle_point_list(c), (R,), np.uint32),
            ranges=_view(L.gof_oracle_ranges(c), (T, 2), np.uint32),
            final_T=_view(L.gof_oracle_final_T(c), (4, self.H, self.W), np.float32),
            n_contrib=_view(L.gof_oracle_n_contrib(c), (2, self.H, self.W), np.uint32),
            num_rendered=R,
        )
