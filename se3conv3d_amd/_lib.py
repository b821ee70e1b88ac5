"""ctypes binding of libse3conv_hip.so (C ABI declared in include/se3conv.h).

There is no CPU fallback: if the library is missing or a call fails the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", f"libse3conv_hip{os.environ.get('SE3_LIB_SUFFIX', '')}.so")  # suffix: variant builds, see build.py

SE3_OK = 0
ABI_VERSION = 5  # SE3_ABI_VERSION of include/se3conv.h these signatures were written against
PRECISIONS = {"fp32": 0, "bf16x3": 1, "bf16x3_t16": 2}
REL_ROT = {"6D": (0, 9), "matrix": (1, 12), "quaternion": (2, 7)}  # p_rel_rot -> (SE3_REL_ROT_*, descriptor dims)


class Se3Shape(C.Structure):
    """struct se3conv_shape (include/se3conv.h)."""

    _fields_ = [
        ("n_in", C.c_int64), ("n_out", C.c_int64), ("n_edges", C.c_int64),
        ("f_in", C.c_int32), ("f_out", C.c_int32), ("c_in", C.c_int32), ("c_out", C.c_int32),
        ("num_basis", C.c_int32), ("precision", C.c_int32),
    ]


class Se3Prepared(C.Structure):
    """struct se3conv_prepared (include/se3conv.h): operands kept by the caller across calls."""

    _fields_ = [("geom_in", C.c_void_p), ("geom_out", C.c_void_p), ("feat_words", C.c_void_p),
                ("geom_in_valid", C.c_int32), ("geom_out_valid", C.c_int32), ("feat_words_valid", C.c_int32)]


class Se3LibraryError(RuntimeError):
    pass


_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
_SZ = C.c_size_t
_F = C.c_float
_SHP = C.POINTER(Se3Shape)

# name -> (restype, argtypes); must list every symbol include/se3conv.h declares
SIGNATURES = {
    "se3_abi_version": (C.c_int, []),
    "se3_error_string": (C.c_char_p, [C.c_int]),
    "se3conv_intermediate_bytes_per_element": (C.c_int, [_SHP, C.c_int]),
    "se3conv_intermediate_row_bytes": (C.c_int64, [_SHP, C.c_int]),
    "se3_compute_keys": (C.c_int, [_P, _P, _P, _P, _P, _I64, _P, _P]),
    "se3_batch_aabb": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P]),
    "se3_grid_subsample_workspace_bytes": (_SZ, [_I64, _I32]),
    "se3_grid_subsample": (C.c_int, [_P, _P, _I64, _I32, _F, _P, _SZ, _P, _P, _P, _P, _P, _P, _P]),
    "se3_segment_pool": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P, _P, _P]),
    "se3_segment_unpool": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "se3_frame_pool": (C.c_int, [_P, _I64, _I32, _I32, _I32, _P, _P, _P]),
    "se3_frame_unpool": (C.c_int, [_P, _P, _I64, _I32, _I32, _I32, _P, _P]),
    "se3_ball_query_grid": (C.c_int, [_P, _P, _I64, _I32, _F, _P, _P, _P, _P]),
    "se3_ball_query_grid_from_box": (C.c_int, [_P, _P, _I32, _F, _P, _P, _P]),
    "se3_ball_query_needs_grid": (C.c_int, [_I64]),
    "se3_ball_query_workspace_bytes": (_SZ, [_I64, _I64]),
    "se3_ball_query_count": (C.c_int, [_P, _P, _P, _P, _P, _P, _F, _I64, _I64, _P, _SZ, _P, _P]),
    "se3_ball_query_store": (C.c_int, [_P, _P, _F, _I64, _I64, _P, _SZ, _P, _I64, _P, _P]),
    "se3_ball_query_bounded": (C.c_int, [_P, _P, _P, _P, _P, _P, _F, _I64, _I64, _I32, _P, _SZ, _I64, _P, _P, _P, _P, _P]),
    "se3_ball_query_grid_bytes": (_SZ, [_I64]),
    "se3_ball_query_bounded_shared": (C.c_int, [_P, _P, _P, _P, _P, _P, _F, _I64, _I64, _I32, _P, _SZ, _I32, _P, _SZ, _I64, _P,
                                                _P, _P, _P, _P]),
    "se3_csr_transpose_workspace_bytes": (_SZ, [_I64]),
    "se3_csr_transpose": (C.c_int, [_P, _I64, _I64, _P, _SZ, _P, _P, _P, _P]),
    "se3_csr_transpose_bounded": (C.c_int, [_P, _I64, _P, _I64, _P, _SZ, _P, _P, _P, _P]),
    "se3_rot_tensors": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _SHP, _P, _P, _P, _P]),
    "se3_rot_tensors_rel": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _SHP, _I32, _P, _P, _P, _P]),
    "se3_feat_basis_proj": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I32, _I32, _P, _P]),
    "se3_feat_basis_proj_grad": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _I32, _P, _P, _P]),
    "se3conv_fwd_workspace_bytes": (_SZ, [_SHP, C.c_int]),
    "se3conv_fwd": (C.c_int, [_P] * 12 + [_SHP, _P, _P, _P, _SZ, _P]),
    "se3conv_bwd_workspace_bytes": (_SZ, [_SHP, C.c_int, C.c_int, C.c_int]),
    "se3conv_bwd_needs_t": (C.c_int, [_SHP, C.c_int]),
    "se3conv_bwd": (C.c_int, [_P] * 17 + [_SHP, _P, _P, _P, _P, _P, _SZ, _P]),
    "se3conv_fwd_prepared": (C.c_int, [_P] * 12 + [_SHP, _P, _P, _P, _SZ, _P, C.POINTER(Se3Prepared)]),
    "se3conv_bwd_prepared": (C.c_int, [_P] * 17 + [_SHP, _P, _P, _P, _P, _P, _SZ, _P, C.POINTER(Se3Prepared)]),
    "se3_knn_query": (C.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "se3_knn_query_pair": (C.c_int, [_P, _P, _I64, _P, _P, _I64, _I32, _P, _P]),
    "se3_grid_pick": (C.c_int, [_P, _P, _P, _I64, _P, _P, _P]),
    "se3_rows_gather": (C.c_int, [_P, _P, _I64, _I64, _P, _P]),
    "se3_rows_scatter": (C.c_int, [_P, _P, _I64, _I64, _P, _P]),
    "se3_knn_query_grid_workspace_bytes": (C.c_size_t, [_I64]),
    "se3_knn_grid_params": (C.c_int, [_P, _I64, _P, _P, _I32, _I32, C.c_float, _P, _P, _P, _P]),
    "se3_knn_query_grid": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _P, _P, C.c_size_t, _P]),
    "se3_pca_frames": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _P]),
    "se3_shuffle_frames": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _P]),
    "se3_glue_workspace_bytes": (_SZ, [_I32]),
    "se3_bn_fwd": (C.c_int, [_P, _P, _P, _I64, _I32, _F, _F, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "se3_affine_act": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "se3_bn_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _P, _P, _P, _P, _SZ, _P]),
    "se3_skip_fwd": (C.c_int, [_P, _P, _P, _P, _F, _P, _I64, _I32, _P, _P]),
    "se3_skip_bwd": (C.c_int, [_P, _P, _P, _P, _F, _P, _I64, _I32, _P, _P, _P, _SZ, _P]),
    "se3_bias_gelu_bwd": (C.c_int, [_P, _P, _P, _I64, _I32, _P, _P, _P, _SZ, _P]),
    "se3_linear_wgrad_workspace_bytes": (_SZ, [_I64, _I32, _I32]),
    "se3_linear_wgrad": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _P, _SZ, _P]),
    "se3_side_stream_stats": (C.c_int, [_P]),
    "se3_set_overlap_rows": (C.c_int, [C.c_int64]),
    "se3_profile_enable": (C.c_int, [C.c_int]),
    "se3_profile_reset": (C.c_int, []),
    "se3_profile_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "se3_profile_tags": (C.c_int, [C.c_char_p, _SZ]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared library (once).  Raises Se3LibraryError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Se3LibraryError(
            f"{LIB_PATH} not found: build it with `python -m se3conv3d_amd.build` "
            "(there is deliberately no CPU fallback for the HIP path)")
    lib = C.CDLL(LIB_PATH)
    lib.se3_abi_version.restype = C.c_int
    if lib.se3_abi_version() != ABI_VERSION:  # (checked before the symbols: an older library lacks some of them)
        raise Se3LibraryError(f"{LIB_PATH} has ABI version {lib.se3_abi_version()}, this binding expects {ABI_VERSION}: "
                              "rebuild it (`python -m se3conv3d_amd.build`)")
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:  # same ABI number, older build of it (entry points are added within a version): say so
            raise Se3LibraryError(f"{LIB_PATH} does not export {name}: it was built from older sources -- rebuild it "
                                  "(`python -m se3conv3d_amd.build`)") from None
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != SE3_OK:
        msg = load().se3_error_string(code).decode()
        raise Se3LibraryError(f"{what} failed with code {code}: {msg}")
