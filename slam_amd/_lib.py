"""ctypes loader of libslamk.so (the HIP library behind include/slamk.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded the import
of any compute entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libslamk.so")

SLAMK_OK, SLAMK_EINVAL, SLAMK_EHIP, SLAMK_ENOMATCH, SLAMK_EREFL, SLAMK_ECAPACITY, SLAMK_ENOTSPD = 0, -1, -2, -3, -4, -5, -6


class Params(C.Structure):
    _fields_ = [("Q0", C.c_double), ("R0", C.c_double), ("qbias", C.c_double), ("p0", C.c_double),
                ("pinit", C.c_double), ("alt_override", C.c_double), ("use_alt_override", C.c_int32),
                ("gate", C.c_double), ("d_min", C.c_double), ("d_max", C.c_double), ("dinit", C.c_double),
                ("update_form", C.c_int32)]


class Match(C.Structure):
    _fields_ = [("id", C.c_int32), ("sx", C.c_double), ("sy", C.c_double), ("rx", C.c_double), ("ry", C.c_double)]


# name -> (restype, argtypes); every symbol include/slamk.h declares
_dp, _ip, _vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
_H = C.c_void_p
SIGNATURES = {
    "slamk_default_params": (None, [C.POINTER(Params)]),
    "slamk_create": (C.c_int, [C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.POINTER(_H)]),
    "slamk_destroy": (C.c_int, [_H]),
    "slamk_last_error": (C.c_char_p, [_H]),
    "slamk_set_state": (C.c_int, [_H, C.c_int, _vp, _vp, _vp]),
    "slamk_get_state": (C.c_int, [_H, C.c_int, _vp, _vp, _vp]),
    "slamk_set_cov": (C.c_int, [_H, C.c_int, _vp, C.c_int, C.c_int]),
    "slamk_get_cov": (C.c_int, [_H, C.c_int, _vp, C.c_int, C.c_int]),
    "slamk_set_features": (C.c_int, [_H, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "slamk_get_features": (C.c_int, [_H, C.c_int, C.POINTER(C.c_int), _vp, _vp, _vp, _vp, _vp, _vp]),
    "slamk_num_features": (C.c_int, [_H, C.c_int]),
    "slamk_num_rows": (C.c_int, [_H, C.c_int]),
    "slamk_frame": (C.c_int, [_H, C.c_int, C.POINTER(Match), C.c_int, _vp]),
    "slamk_frame_batch_async": (C.c_int, [_H, C.POINTER(Match), C.c_int, _vp, _vp, _vp]),
    "slamk_set_frame_values": (C.c_int, [_H, C.c_int, _vp, _vp]),
    "slamk_predict": (C.c_int, [_H, _vp, _vp, _vp, _vp]),
    "slamk_update": (C.c_int, [_H, _vp, _vp, _vp]),
    "slamk_frame_begin_dev": (C.c_int, [_H, _vp]),
    "slamk_predict_dev": (C.c_int, [_H, _vp, _vp, _vp, _vp, _vp]),
    "slamk_update_dev": (C.c_int, [_H, _vp, _vp, _vp, _vp, _vp, _vp]),
    "slamk_step_dev": (C.c_int, [C.c_void_p] * 10),
    "slamk_set_pipeline_groups": (C.c_int, [C.c_void_p, C.c_int]),
    "slamk_set_graph_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "slamk_sync": (C.c_int, [_H, _vp]),
    "slamk_read_status": (C.c_int, [_H, _vp]),
    "slamk_stage_motion_dev": (C.c_int, [_H, _vp, _vp, _vp, _vp, _vp]),
    "slamk_stage_measure_dev": (C.c_int, [_H, _vp, _vp, _vp, _vp, _vp]),
    "slamk_stage_right_dev": (C.c_int, [_H, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "slamk_stage_left_dev": (C.c_int, [_H, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp]),
    "slamk_stage_eye_minus_kh_dev": (C.c_int, [_H, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "slamk_stage_gain_dev": (C.c_int, [_H, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "slamk_stage_gain_static_dev": (C.c_int, [_H, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "slamk_stage_apply_dev": (C.c_int, [_H, _vp, C.c_int, _vp, _vp]),
    "slamk_buffer_dev": (C.c_int, [_H, C.c_int, C.POINTER(_vp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "slamk_cholesky_inverse_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "slamk_nccl_unique_id": (C.c_int, [_vp]),
    "slamk_sharded_create": (C.c_int, [_H, C.c_int, C.c_int, _vp, _vp, _vp, C.POINTER(_vp)]),
    "slamk_sharded_destroy": (C.c_int, [_vp]),
    "slamk_sharded_load_cov_dev": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "slamk_sharded_predict_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "slamk_sharded_update_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "slamk_sharded_gather_cov_dev": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "slamk_sharded_dims": (C.c_int, [_vp] + [C.POINTER(C.c_int)] * 6),
    "slamk_pre_euler2qbw_dev": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "slamk_pre_rotate_dev": (C.c_int, [C.POINTER(C.c_double), _vp, _vp, C.c_int, _vp]),
    "slamk_pre_img2body_dev": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "slamk_pre_fir_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _vp]),
    "slamk_set_debug": (C.c_int, [C.c_void_p, C.c_int]),
    "slamk_set_block_sizes": (C.c_int, [C.c_int, C.c_int]),
    "slamk_debug_get": (C.c_int, [_H, C.c_int, _vp, _vp, _vp, _vp]),
    "slamk_debug_get_F": (C.c_int, [_H, C.c_int, _vp]),
    "slamk_dgemm_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_int, C.c_longlong,
                                  _vp, C.c_int, C.c_longlong, C.c_double, _vp, C.c_int, C.c_longlong, C.c_int, _vp]),
    "slamk_mfma_f64_probe": (C.c_int, [C.c_int, _dp]),
    "slamk_profile_enable": (C.c_int, [_H, C.c_int]),
    "slamk_profile_read": (C.c_int, [_H, _dp, C.POINTER(C.c_int), C.c_int]),
    "slamk_flops_predict": (C.c_double, [C.c_int]),
    "slamk_flops_update": (C.c_double, [C.c_int, C.c_int, C.c_int]),
    "slamk_flops_update_stage": (C.c_double, [C.c_int, C.c_int, C.c_int, C.c_int]),
}

_lib = None


def lib():
    """Load libslamk.so once and attach the prototypes.  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libslamk.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % LIB_PATH)
        # torch ships its own copy of the HIP runtime: when a process uses both, torch's must be the one that is loaded
        # first (loaded after this library's, torch reports no GPU).  A process without torch skips this.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
