"""ctypes binding of libniw_hip.so (the C ABI declared in include/niw.h).

There is no CPU fallback: if the shared library is missing the import of any op fails loudly
with instructions to build it (`python -c "import __graft_entry__ as g; g.build()"` or
`make -C neural_invertible_warp_amd/csrc`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NIW_LIB_PATH") or os.path.join(_HERE, "libniw_hip.so")   # override: diagnostic builds (tools/)

_vp, _i, _i64, _u64, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float, ctypes.c_double

_i32 = ctypes.c_int32


class RenderDesc(ctypes.Structure):
    """niw_render_desc of include/niw.h, field for field"""
    _fields_ = [("intr", _vp), ("pose", _vp), ("n_views", _i32), ("H", _i32), ("W", _i32), ("ndc", _i32),
                ("first_pixel", _i64), ("n_pixels", _i64), ("ndc_near", _f), ("depth_min", _d), ("depth_max", _d),
                ("inverse_depth", _i32), ("n_samples", _i32), ("n_fine", _i32), ("density_activ", _i32), ("precision", _i32), ("has_bg", _i32), ("bg", _f),
                ("u", _vp), ("unif", _vp), ("bins", _vp), ("packed", _vp), ("packed_fine", _vp),
                ("band_w3d", ctypes.POINTER(_f)), ("band_wview", ctypes.POINTER(_f)), ("band_dev", _vp),
                ("band_w3d_fine", ctypes.POINTER(_f)), ("band_wview_fine", ctypes.POINTER(_f)), ("band_dev_fine", _vp)]


# name -> (restype, argtypes); mirrors include/niw.h one to one
SIGNATURES = {
    "niw_version": (_i, []),
    "niw_last_error_string": (ctypes.c_char_p, []),
    "niw_mlp_padded_rows": (_i64, [_i64, _i]),
    "niw_mlp_packed_floats": (_i64, []),
    "niw_mlp_bwd_workspace_floats": (_i64, [_i64, _i]),
    "niw_mlp_pack_weights": (_i, [_vp, _vp, _vp]),
    "niw_mlp_pack_index": (_i, [_vp, _vp]),
    "niw_mlp_pack_weights_indexed": (_i, [_vp, _vp, _vp, _vp]),
    "niw_mlp_packed_bytes": (_i64, [_i]),
    "niw_mlp_pack_weights_prec": (_i, [_vp, _i, _vp, _vp]),
    "niw_mlp_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "niw_mlp_bwd": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "niw_mlp_bwd_dx": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "niw_mlp_bwd_dw": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    "niw_composite_fwd": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "niw_composite_bwd": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "niw_sample_stratified": (_i, [_vp, _i64, _i, _d, _d, _i, _vp, _vp]),
    "niw_sample_stratified_rng": (_i, [_u64, _u64, _vp, _i64, _i, _d, _d, _i, _vp, _vp, _vp]),
    "niw_sample_pdf_merge": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    "niw_raygen": (_i, [_vp, _vp, _vp, _i64, _i, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "niw_draw_ray_idx": (_i, [_i64, _i64, _u64, _u64, _vp, _i64, _i64, _vp, _vp]),
    "niw_convert_ndc": (_i, [_vp, _vp, _vp, _i, _i64, _f, _vp, _vp, _vp]),
    "niw_warp_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "niw_warp_bwd_workspace_floats": (_i64, [_i, _i64]),
    "niw_warp_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "niw_warp_prep_fwd_workspace_floats": (_i64, [_i]),
    "niw_warp_prep_bwd_workspace_floats": (_i64, [_i]),
    "niw_warp_prep_fwd": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "niw_warp_prep_bwd": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "niw_kabsch_rotation_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "niw_kabsch_rotation_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "niw_align_moments": (_i, [_vp, _vp, _i, _i64, _vp, _vp]),
    "niw_align_solve": (_i, [_vp, _i, _vp, _vp]),
    "niw_align_loss": (_i, [_vp, _vp, _vp, _i, _i64, _d, _vp, _vp, _vp]),
    "niw_mse_fwd_bwd": (_i, [_vp, _vp, _vp, _i, _i64, _i64, _i64, _i64, _d, _f, _vp, _vp, _vp]),
    "niw_render_fwd_workspace_floats": (_i64, [_i, _i64, _i, _i]),
    "niw_render_fwd": (_i, [ctypes.POINTER(RenderDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "niw_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _d, _d, _d, _d, _i, _vp, _vp]),
}

_lib = None


class NiwError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises NiwError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NiwError(
            f"{LIB_PATH} not found: the HIP library is not built. Run "
            "`make -C neural_invertible_warp_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU fallback for the render path.")
    # PyTorch-ROCm ships its own HIP runtime; it must be the one this process uses.  Loaded first, libniw_hip.so would pull in the
    # system copy instead, and kernels launched through it find "no ROCm-capable device" once torch has initialised the other one.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def call(name, *args):
    """Call an int-returning entry point and raise NiwError(niw_last_error_string()) on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise NiwError(f"{name} failed ({rc}): {lib.niw_last_error_string().decode()}")
    return rc
