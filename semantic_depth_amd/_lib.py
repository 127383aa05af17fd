"""ctypes binding of libsemdepth.so (include/semdepth.h).  No compute happens in Python.

The library is built in-tree by ``semantic_depth_amd.build`` / ``__graft_entry__.build()``.  If it is
missing this module raises: there is no CPU or PyTorch fallback for the hot path.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsemdepth.so")

SD_OK = 0
SD_ENC_VGG, SD_ENC_RESNET50 = 0, 1
SD_NET_FCN8S, SD_NET_MONODEPTH = 0, 1
SD_PREC_F32, SD_PREC_BF16X2, SD_PREC_MIXED, SD_PREC_PLAN, SD_PREC_BF16X3, SD_PREC_F16X2 = 0, 1, 2, 3, 4, 5


class sd_camera(C.Structure):
    _fields_ = [("cx", C.c_double), ("cy", C.c_double), ("f", C.c_double), ("b", C.c_double), ("disp_mult", C.c_double)]


class sd_rw_params(C.Structure):
    _fields_ = [("depth", C.c_double), ("z_cut", C.c_double), ("mad_y", C.c_double), ("mad_x", C.c_double),
                ("plane_thr", C.c_double), ("sor_k", C.c_int32), ("sor_ratio", C.c_double), ("ror_n", C.c_int32),
                ("ror_r", C.c_double), ("window", C.c_double), ("depth_offset", C.c_double), ("use_o3d", C.c_int32)]


class sd_rw_result(C.Structure):
    _fields_ = [("width", C.c_double), ("x_left", C.c_float), ("x_right", C.c_float), ("left_pt", C.c_float * 3),
                ("right_pt", C.c_float * 3), ("found", C.c_int32), ("n_road", C.c_int32), ("n_zcut", C.c_int32),
                ("n_mad_y", C.c_int32), ("n_mad_x", C.c_int32), ("n_plane", C.c_int32), ("n_sor", C.c_int32),
                ("n_ror", C.c_int32), ("plane", C.c_double * 4)]


class sd_f2f_params(C.Structure):
    _fields_ = [("depth", C.c_double), ("mad_y", C.c_double), ("z_max", C.c_double), ("mad_left", C.c_double),
                ("mad_right", C.c_double), ("plane_thr", C.c_double)]


class sd_f2f_result(C.Structure):
    _fields_ = [("dist", C.c_double), ("left_pt", C.c_double * 3), ("right_pt", C.c_double * 3), ("plane_left", C.c_double * 4),
                ("plane_right", C.c_double * 4), ("counts", C.c_int32 * 7), ("ok", C.c_int32)]


class sd_profile_bucket(C.Structure):
    _fields_ = [("kernel", C.c_char * 64), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


# every symbol include/semdepth.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_H = C.c_void_p
SIGNATURES = {
    "sd_version": (C.c_char_p, []),
    "sd_status_string": (C.c_char_p, [C.c_int]),
    "sd_create": (C.c_int, [C.POINTER(_H), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sd_create_with_plan": (C.c_int, [C.POINTER(_H), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]),
    "sd_default_plan": (C.c_char_p, [C.c_int]),
    "sd_precision_plan": (C.c_int, [_H, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double)]),
    "sd_destroy": (C.c_int, [_H]),
    "sd_last_error": (C.c_char_p, [_H]),
    "sd_query_memory": (C.c_int, [_H, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "sd_bind_memory": (C.c_int, [_H, _P, _P, _P]),
    "sd_weight_count": (C.c_int, [_H, C.c_int]),
    "sd_weight_info": (C.c_int, [_H, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "sd_load_weight": (C.c_int, [_H, C.c_int, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int]),
    "sd_fcn8s_forward": (C.c_int, [_H, _P, C.c_int, _P, _P, _P, _P, _P]),
    "sd_monodepth_forward": (C.c_int, [_H, _P, C.c_int, _P, _P, _P]),
    "sd_png_unfilter_bgr": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "sd_png_decode_bgr": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sd_jpeg_decode_bgr": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sd_image_decode_bgr": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sd_decode_files_bgr": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, _P, C.c_size_t, C.c_int, C.POINTER(C.c_int)]),
    "sd_ply_format_rows": (C.c_int64, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int]),
    "sd_post_process": (C.c_int, [_H, _P, C.c_int, _P, _P]),
    "sd_resize_cubic_u8": (C.c_int, [_H, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P]),
    "sd_fuse_backproject": (C.c_int, [_H, _P, _P, _P, _P, C.POINTER(sd_camera), C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sd_postprocess_fuse_backproject": (C.c_int, [_H, _P, _P, _P, _P, _P, C.POINTER(sd_camera), C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "sd_road_width": (C.c_int, [_H, _P, _P, _P, C.c_int, C.c_int, C.POINTER(sd_rw_params), _P, _P, _P, _P, _P]),
    "sd_fence_to_fence": (C.c_int, [_H, _P, _P, _P, C.c_int, C.c_int, _P, C.POINTER(sd_f2f_params), _P, _P, _P, _P, _P, _P]),
    "sd_pcl_extract_pcls": (C.c_int, [_H, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sd_pcl_remove_from_to": (C.c_int, [_H, _P, _P, C.c_int, C.c_int, C.c_double, _P, _P, _P, _P]),
    "sd_pcl_remove_noise_by_mad": (C.c_int, [_H, _P, _P, C.c_int, C.c_int, C.c_double, _P, _P, _P, _P, _P]),
    "sd_pcl_remove_noise_by_fitting_plane": (C.c_int, [_H, _P, _P, C.c_int, C.c_int, C.c_double, _P, _P, _P, _P, _P]),
    "sd_pcl_threshold_complete": (C.c_int, [_H, _P, _P, C.c_int, C.c_int, C.c_double, _P, _P, _P, _P]),
    "sd_pcl_get_end_points_of_road": (C.c_int, [_H, _P, C.c_int, C.c_double, C.c_double, _P, _P]),
    "sd_o3d_statistical_outlier_removal": (C.c_int, [_H, _P, _P, C.c_int, C.c_int, C.c_double, _P, _P, _P, _P, _P]),
    "sd_o3d_radius_outlier_removal": (C.c_int, [_H, _P, _P, C.c_int, C.c_int, C.c_double, _P, _P, _P, _P]),
    "sd_net_tensor": (C.c_int, [_H, C.c_int, C.c_char_p, _P, C.c_size_t, C.POINTER(C.c_int64), _P]),
    "sd_profile": (C.c_int, [_H, C.c_int]),
    "sd_profile_read": (C.c_int, [_H, C.POINTER(sd_profile_bucket), C.c_int, C.POINTER(C.c_int)]),
    "sd_net_flops_per_image": (C.c_double, [_H, C.c_int]),
    "sd_pass_frames": (C.c_int, [_H]),
    "sd_saturation_count": (C.c_int, [_H, C.POINTER(C.c_uint64), C.c_int]),
    "sd_saturation_count_async": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "sd_set_reserved_cus": (C.c_int, [_H, C.c_int]),
}

_lib = None


def load():
    """dlopen libsemdepth.so (after torch, so that it binds to the HIP runtime torch already loaded)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP library has not been built (run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `python -m semantic_depth_amd.build`).  There is no CPU fallback for this path.")
    import torch  # noqa: F401  (loads torch/lib/libamdhip64.so, SONAME libamdhip64.so.7, which libsemdepth then shares)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if os.environ.get("SEMDEPTH_SKIP_HASH_CHECK") != "1":
        from .build import source_hash
        built, want = lib.sd_version().decode().rpartition("src=")[2], source_hash()
        if built != want:
            raise RuntimeError(f"{LIB_PATH} was built from other sources (library src={built}, tree src={want}): rebuild it with "
                               "`python -m semantic_depth_amd.build` — a stale binary is never used silently")
    _lib = lib
    return lib


class SdError(RuntimeError):
    pass


def check(lib, handle, status, what=""):
    if status != SD_OK:
        msg = lib.sd_last_error(handle).decode() if handle else ""
        raise SdError(f"{what}: {lib.sd_status_string(status).decode()} ({status}) {msg}")
