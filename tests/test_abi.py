"""CPU-side checks of the C-ABI boundary: the library loads without a GPU, exports every symbol
include/niw.h declares, the ctypes table matches the header, and host-side argument checks
return error codes (no compute is launched)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "niw.h")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(niw_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    syms = header_symbols()
    for s in ("niw_mlp_fwd", "niw_mlp_bwd", "niw_composite_fwd", "niw_composite_bwd", "niw_warp_fwd", "niw_warp_bwd",
              "niw_sample_stratified", "niw_sample_pdf_merge", "niw_raygen", "niw_mse_fwd_bwd", "niw_adam_step"):
        assert s in syms


def test_library_exports_every_header_symbol():
    from neural_invertible_warp_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} missing: run __graft_entry__.build()")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in header_symbols():
        assert hasattr(lib, s), f"{s} declared in niw.h but not exported"
    assert sorted(_lib.SIGNATURES) == header_symbols(), "ctypes table and niw.h disagree"


def test_argument_count_matches_header():
    from neural_invertible_warp_amd import _lib
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", src, flags=re.S)
        assert m, name
        body = m.group(1).strip()
        n = 0 if body in ("", "void") else body.count(",") + 1
        assert n == len(args), f"{name}: header has {n} parameters, ctypes table {len(args)}"


def test_host_side_argument_checks_return_codes():
    from neural_invertible_warp_amd import _lib
    lib = _lib.load()
    assert lib.niw_version() >= 100
    assert lib.niw_mlp_padded_rows(10, 13) == 256
    rc = lib.niw_composite_fwd(None, None, None, None, 4, 8, 0, 0.0, None, None, None, None, None)
    assert rc == -1 and b"null pointer" in lib.niw_last_error_string()
    with pytest.raises(_lib.NiwError):
        _lib.call("niw_sample_pdf_merge", None, None, None, None, 1, 8, 8, None, None, None)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from neural_invertible_warp_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.NiwError, match="no CPU fallback"):
        _lib.load()


def test_every_entry_point_rejects_bad_arguments_without_touching_the_gpu():
    """Error behaviour of the boundary: status code < 0, message in niw_last_error_string, no exception,
    no launch.  Dummy non-null pointers are never dereferenced because the host-side checks fail first."""
    import ctypes
    from neural_invertible_warp_amd import _lib
    lib = _lib.load()
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    err = lambda: lib.niw_last_error_string().decode()
    # null pointers
    cases = {
        "niw_mlp_pack_weights": (None, None, None),
        "niw_mlp_fwd": (None,) * 5 + (4, 8, None, None, None, 1, 0, None, None, None, None),
        "niw_mlp_bwd_dx": (None,) * 4 + (4, 8, 1, 0) + (None,) * 8,
        "niw_mlp_bwd_dw": (None, None, 4, 8, 0, None, None, None),
        "niw_mlp_pack_weights_prec": (None, 1, None, None),
        "niw_sample_stratified_rng": (1, 1, None, 4, 8, 0.0, 1.0, 0, None, None, None),
        "niw_composite_bwd": (None,) * 4 + (4, 8, 0, 0.0) + (None,) * 8,
        "niw_sample_stratified": (None, 4, 8, 0.0, 1.0, 0, None, None),
        "niw_raygen": (None, None, None, 0, 2, 4, 8, 8, 0, None, None, None),
        "niw_draw_ray_idx": (64, 8, 1, 1, None, 0, 1, None, None),
        "niw_convert_ndc": (None, None, None, 2, 4, 1.0, None, None, None),
        "niw_warp_fwd": (None,) * 4 + (2, 4, None, None, None, 0, None, None, 0, None, None, None),
        "niw_warp_prep_fwd": (None, None, 2, None, None, None, None, None),
        "niw_warp_prep_bwd": (None, None, 2) + (None,) * 7,
        "niw_mse_fwd_bwd": (None, None, None, 2, 4, 64, 0, 0, 1.0, 1.0, None, None, None),
        "niw_adam_step": (None,) * 4 + (8, 1e-3, 0.9, 0.999, 1e-8, 1, None, None),
        "niw_align_moments": (None, None, 2, 8, None, None),
        "niw_align_solve": (None, 2, None, None),
        "niw_align_loss": (None, None, None, 2, 8, 48.0, None, None, None),
    }
    for name, args in cases.items():
        assert len(args) == len(_lib.SIGNATURES[name][1]), name
        rc = getattr(lib, name)(*args)
        assert rc == -1 and err(), (name, rc, err())
    # sizes / enums (pointers non-null)
    assert lib.niw_mlp_fwd(p, p, p, p, None, 0, 8, None, None, None, 1, 0, p, p, None, None) == -1 and "positive" in err()
    assert lib.niw_mlp_fwd(p, p, p, p, None, 4, 8, None, None, None, 7, 0, p, p, None, None) == -1 and "activation" in err()
    assert lib.niw_mlp_fwd(p, p, p, p, None, 4, 8, None, None, None, 1, 5, p, p, None, None) == -1 and "precision" in err()
    assert lib.niw_mlp_fwd(p, p, p, p, None, 1 << 20, 64, None, None, None, 1, 0, p, p, None, None) == -1 and "too many samples" in err()
    assert lib.niw_mlp_pack_weights_prec(p, 9, p, None) == -1 and "precision" in err()
    # the images of the precision classes: fp32 = the packed floats; the split-bf16 image holds two bf16 planes of the forward and
    # of the transposed (dX) fragments, i.e. about as many bytes
    assert lib.niw_mlp_packed_bytes(0) == 4 * lib.niw_mlp_packed_floats()
    assert lib.niw_mlp_packed_bytes(1) == lib.niw_mlp_packed_bytes(2) > 4 * 527872
    assert lib.niw_mse_fwd_bwd(p, p, None, 2, 4, 64, 6, 4, 1.0, 1.0, p, p, None) == -1 and "leave" in err()      # rays [6, 10) of a 2 x 4 batch
    assert lib.niw_warp_prep_fwd(p, p, 65, p, p, p, p, None) == -1 and "views" in err()
    assert lib.niw_warp_prep_fwd(p, p, 0, p, p, p, p, None) == -1
    # the one-call render: null descriptor, incomplete descriptor, pixel range outside the image, fine pass without its tables
    assert lib.niw_render_fwd(None, p, p, p, p, None, None, None, None) == -1 and "null pointer" in err()
    d = _lib.RenderDesc(n_views=1, H=4, W=4, first_pixel=0, n_pixels=16, n_samples=8)
    assert lib.niw_render_fwd(ctypes.byref(d), p, p, p, p, None, None, None, None) == -1 and "cameras" in err()
    d.intr = d.pose = d.packed = p
    d.first_pixel = 8
    assert lib.niw_render_fwd(ctypes.byref(d), p, p, p, p, None, None, None, None) == -1 and "outside" in err()
    d.first_pixel, d.n_fine = 0, 8
    assert lib.niw_render_fwd(ctypes.byref(d), p, p, p, p, None, None, None, None) == -1 and "fine pass" in err()
    assert lib.niw_render_fwd_workspace_floats(2, 10, 8, 0) == 4 * 60 + 3 * 160 + 480
    assert lib.niw_render_fwd_workspace_floats(2, 10, 8, 4) == 4 * 60 + 3 * 160 + 480 + 2 * 240 + 720
    # workspace queries are pure host arithmetic
    assert lib.niw_warp_prep_fwd_workspace_floats(18) == 3 * 18 * 128
    assert lib.niw_mlp_bwd_workspace_floats(4, 8) == 511 * (256 * 256 + 256) + 512 * (256 * 64 + 256) + 256 * (128 * 320 + 256) + 256 * (512 + 640)
    assert lib.niw_mlp_packed_floats() > 2 * 527872
