"""ctypes binding of libf3dg_hip.so (include/f3dg.h). The product path has NO fallback: if the HIP library is
missing or fails to load, importing a function from here raises; nothing under this package touches ``oracle/``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libf3dg_hip.so")

OK, ERR_BAD_ARG, ERR_WORKSPACE, ERR_OVERFLOW, ERR_HIP, ERR_UNSUPPORTED, ERR_STATE = 0, -1, -2, -3, -4, -5, -6
FLAG_SAVE_AUX, FLAG_BG_PER_VIEW, FLAG_SKIP_NORMAL, FLAG_SKIP_DISTORTION = 1, 2, 4, 8
FLAG_EXACT, FLAG_FAST, FLAG_NO_TILE_CULL, FLAG_NO_SMALL_PATH, FLAG_SCAN = 16, 32, 64, 128, 256
PENDING = 1

_ERR_TEXT = {
    ERR_BAD_ARG: "bad argument",
    ERR_WORKSPACE: "workspace smaller than f3dg_workspace_bytes()",
    ERR_OVERFLOW: "instance capacity (max_rendered) exceeded",
    ERR_HIP: "HIP runtime error",
    ERR_UNSUPPORTED: "unsupported configuration",
    ERR_STATE: "backward on a workspace whose last forward kept no auxiliary planes (not a save_aux call of the general path)",
}

_p, _i, _f, _ll, _sz, _u = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_size_t, C.c_uint

# name -> (restype, argtypes); must list every symbol include/f3dg.h declares (tests/test_abi.py checks both ways)
SIGNATURES = {
    "f3dg_version": (C.c_char_p, []),
    "f3dg_last_error": (C.c_char_p, []),
    "f3dg_workspace_bytes": (_sz, [_i, _i, _i, _i, _ll]),
    "f3dg_forward_batched": (_i, [_p, _p, _sz, _ll, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p,
                                  _p, _p, _p, _f, _f, _f, _p, _p, _u]),
    "f3dg_forward_sets": (_i, [_p, _p, _sz, _ll, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p,
                               _p, _p, _p, _f, _f, _f, _p, _p, _u]),
    "f3dg_read_status": (_i, [_p, _p, C.POINTER(_ll)]),
    "f3dg_status_post": (_i, [_p, _p]),
    "f3dg_status_poll": (_i, [_i, _i, C.POINTER(_ll)]),
    "f3dg_forward": (_ll, [_p, _p, _sz, _ll, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p,
                           _f, _f, _f, _i, _p, _p, _u, C.POINTER(_ll)]),
    "f3dg_backward": (_i, [_p, _p, _sz, _ll, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p,
                           _f, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _u]),
    "f3dg_integrate_workspace_bytes": (_sz, [_i, _i, _i, _i, _ll]),
    # stream, ws, ws_bytes, cap | PN P D M | bg W H | points3D means3D shs colors opac scales | mod | rot cov3D v2g view
    # proj campos | tanx tany ks | subpix | prefiltered | out_color radii alpha_i color_i | h_needed
    "f3dg_integrate": (_ll, [_p, _p, _sz, _ll, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p,
                             _p, _p, _f, _f, _f, _p, _i, _p, _p, _p, _p, C.POINTER(_ll)]),
    "f3dg_integrate_prepare": (_ll, [_p, _p, _sz, _ll, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p,
                                     _p, _f, _f, _f, _p, _p, C.POINTER(_ll)]),
    "f3dg_integrate_points": (_i, [_p, _p, _sz, _ll, _i, _i, _i, _i, _p, _p, _f, _f, _p, _p, _p, _p]),
    "f3dg_integrate_workspace_bytes_batched": (_sz, [_i, _i, _i, _i, _i, _ll]),
    "f3dg_integrate_prepare_batched": (_ll, [_p, _p, _sz, _ll, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p,
                                             _p, _f, _f, _f, _p, _p, C.POINTER(_ll)]),
    "f3dg_integrate_points_view": (_i, [_p, _p, _sz, _ll, _i, _i, _i, _i, _i, _i, _p, _p, _f, _f, _p, _p, _p, _p]),
    "f3dg_debug_integrate_redo": (_i, [_p, _p, _i, _i, _i, _i, _i, _ll, C.POINTER(_i)]),
    "f3dg_mark_visible": (_i, [_p, _i, _p, _p, _p, _p]),
    "f3dg_splat_head": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p, _f, _ll, _ll, _p, _p, _p, _p, _p, _p, _p]),
    "f3dg_render_epilogue": (_i, [_p, _i, _i, _i, _p, _p, _f, _f, _p, _p]),
    "f3dg_render_epilogue_view": (_i, [_p, _i, _i, _i, _p, _p, _f, _f, _p, _p]),
    "f3dg_cycle_inputs": (_i, [_p, _i, _i, _i, _i, _p, _p, _p]),
    "f3dg_pack_frames": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "f3dg_pack_frames_host": (_i, [_p, _i, _i, _i, _i, _p, _p, _i]),
    "f3dg_group_norm_silu": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _f, _i, _p]),
    "f3dg_group_norm_silu_bf16": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _f, _i, _p]),
    "f3dg_group_norm_silu_pb": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _f, _i, _p]),
    "f3dg_group_norm_silu_pb_bf16": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _f, _i, _p]),
    "f3dg_group_norm_silu_nhwc_pb": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _f, _i, _p, _p, _sz]),
    "f3dg_group_norm_silu_nhwc_pb_bf16": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _f, _i, _p, _p, _sz]),
    "f3dg_group_norm_silu_pb_f16": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _f, _i, _p]),
    "f3dg_group_norm_silu_nhwc_pb_f16": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _f, _i, _p, _p, _sz]),
    "f3dg_group_norm_nhwc_scratch_bytes": (C.c_size_t, [_i, _i, _i]),
    "f3dg_residual_join_f16": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _f, _p]),
    "f3dg_residual_join": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _f, _p]),
    "f3dg_residual_join_bf16": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _f, _p]),
    "f3dg_group_norm_silu_nhwc": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _f, _i, _p, _p, _sz]),
    "f3dg_group_norm_silu_nhwc_bf16": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _f, _i, _p, _p, _sz]),
    "f3dg_set_option": (_i, [C.c_char_p, _i]),
    "f3dg_profile_enable": (_i, [_i]),
    "f3dg_debug_launch_count": (C.c_longlong, [_i]),
    "f3dg_debug_launch_times": (_i, [_i]),
    "f3dg_profile_collect": (_i, [C.POINTER(C.c_double), C.POINTER(_i)]),
    "f3dg_profile_collect_calls": (_i, [C.POINTER(C.c_double), C.POINTER(_i), C.POINTER(C.c_double), _i]),
    "f3dg_debug_last_render_kernel": (C.c_char_p, []),
    "f3dg_debug_render_counts": (_i, [C.POINTER(C.c_ulonglong), _i]),
    "f3dg_debug_render4_counts": (_i, [C.POINTER(C.c_ulonglong), _i]),
    "f3dg_debug_render5_counts": (_i, [C.POINTER(C.c_ulonglong), _i]),
    "f3dg_debug_render3q_clocks": (_i, [C.POINTER(C.c_ulonglong)]),
    "f3dg_debug_pass1_occupancy": (_i, [C.POINTER(_i), C.POINTER(_i)]),
    "f3dg_backward_pairs": (_i, [_p, _p, C.POINTER(_ll)]),
    "f3dg_debug_timing": (_i, [C.POINTER(C.c_ulonglong), _i]),
    "f3dg_debug_export": (_i, [_p, _p, _i, _i, _i, _i, _ll, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
}

_LIB = None


class F3dgError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        detail = ""
        try:
            detail = lib().f3dg_last_error().decode()
        except Exception:
            pass
        super().__init__(f"{where}: {_ERR_TEXT.get(code, 'error')} (code {code})" + (f": {detail}" if detail and code == ERR_HIP else ""))


def lib():
    """Load (once) and return the ctypes handle. Raises if the HIP extension is not built -- loudly, by design."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
                "There is no CPU or PyTorch fallback for the rasterization path.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)       # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(code, where):
    if code < 0:
        raise F3dgError(int(code), where)
    return code


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL; empty tensor -> NULL, as the reference maps empty to nullptr)."""
    if t is None or t.numel() == 0:
        return None
    return C.c_void_p(t.data_ptr())
