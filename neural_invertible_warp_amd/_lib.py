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


class AdamGroup(ctypes.Structure):
    """niw_adam_group of include/niw.h"""
    _fields_ = [("param", _vp), ("grad", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("n", _i64), ("lr", _d), ("step", _i32), ("reserved", _i32)]


class TrainDesc(ctypes.Structure):
    """niw_train_desc of include/niw.h, field for field"""
    _fp = ctypes.POINTER(_f)
    _fields_ = [("image", _vp), ("intr", _vp), ("pose_init", _vp), ("n_views", _i32), ("H", _i32), ("W", _i32),
                ("view0", _i32), ("view1", _i32), ("own0", _i32), ("own1", _i32), ("stratified", _i32),
                ("rays_per_view", _i64), ("ray_lo", _i64), ("ray_hi", _i64),
                ("pixel_seed", _u64), ("depth_seed", _u64), ("draw", _u64), ("draw_dev", _vp),
                ("n_samples", _i32), ("n_fine", _i32), ("inverse_depth", _i32), ("density_activ", _i32),
                ("depth_min", _d), ("depth_max", _d), ("unif", _vp), ("bins", _vp),
                ("nerf_params", _vp), ("nerf_fine_params", _vp), ("pack_index", _vp), ("precision", _i32), ("use_index_window", _i32),
                ("band_w3d", _fp), ("band_wview", _fp), ("band_dev", _vp),
                ("warp_params", _vp), ("latent", _vp), ("chan_w", _fp), ("index_window", _fp), ("window_dev", _vp),
                ("w_render", _f), ("w_render_fine", _f), ("w_align", _f), ("always_register", _i32), ("mse_norm", _d),
                ("loss", _vp), ("d_nerf", _vp), ("d_nerf_fine", _vp), ("d_warp", _vp), ("d_latent", _vp), ("poses", _vp),
                ("rgb", _vp), ("rgb_fine", _vp), ("overlap", _i32), ("reserved", _i32), ("fine_grads_ready", _vp),
                ("density_noise", _f), ("ndc", _i32), ("noise_seed", _u64), ("ndc_near", _f), ("has_bg", _i32), ("bg", _f), ("reserved3", _i32)]


# train stages (enum niw_train_stage), in execution order
TRAIN_STAGES = ("front", "warp_fwd", "mlp_fwd", "composite_fwd", "resample", "mlp_fwd_fine", "composite_fwd_fine", "loss",
                "composite_bwd_fine", "mlp_bwd_dx_fine", "mlp_bwd_dw_fine", "composite_bwd", "mlp_bwd_dx", "mlp_bwd_dw", "warp_bwd")

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
    "niw_composite_mse_train": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _i, _i64, _i64, _i64, _d, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "niw_mse_from_residuals": (_i, [_vp, _i64, _d, _vp, _vp]),
    "niw_sample_stratified": (_i, [_vp, _i64, _i, _d, _d, _i, _vp, _vp]),
    "niw_sample_stratified_rng": (_i, [_u64, _u64, _vp, _i64, _i, _d, _d, _i, _vp, _vp, _vp]),
    "niw_normal_rng": (_i, [_u64, _u64, _vp, _i64, _f, _vp, _vp]),
    "niw_sample_pdf_merge": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    "niw_raygen": (_i, [_vp, _vp, _vp, _i64, _i, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "niw_draw_ray_idx": (_i, [_i64, _i64, _u64, _u64, _vp, _i64, _i64, _vp, _vp]),
    "niw_convert_ndc": (_i, [_vp, _vp, _vp, _i, _i64, _f, _vp, _vp, _vp]),
    "niw_convert_ndc_bwd": (_i, [_vp, _vp, _vp, _i, _i64, _f, _vp, _vp, _vp, _vp, _vp]),
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
    "niw_adam_step_multi": (_i, [ctypes.POINTER(AdamGroup), _i, _d, _d, _d, _vp, _vp]),
    "niw_train_step_prepare": (_i, []),
    "niw_train_step_workspace_floats": (_i64, [ctypes.POINTER(TrainDesc)]),
    "niw_train_step": (_i, [ctypes.POINTER(TrainDesc), _vp, _i, _i, _vp]),
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
