"""ctypes loader of liborbx.so (the HIP kernels + C ABI declared in include/orbx.h).

There is no CPU fallback: if the shared library is missing it is built with hipcc, and if that fails, or
if a compute entry point is called without a GPU, the error propagates.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")]
)
NUM_STAGES = 7
STAGE_NAMES = ("pyramid", "fast", "octree", "layout", "blur", "orient_describe", "total")

_lib = None

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes).  Every symbol include/orbx.h declares is listed (tests check the export table).
SIGNATURES = {
    "orbx_create": (_i, [C.POINTER(_vp), _i, _f, _i, _i, _i, _i]),
    "orbx_destroy": (None, [_vp]),
    "orbx_last_error": (C.c_char_p, []),
    "orbx_get_levels": (_i, [_vp]),
    "orbx_get_scale_tables": (None, [_vp, _vp, _vp, _vp, _vp]),
    "orbx_get_features_per_level": (None, [_vp, _vp]),
    "orbx_set_input_resize": (_i, [_vp, _i, _i]),
    "orbx_max_keypoints": (_i, [_vp, _i, _i]),
    "orbx_set_tap_contraction": (_i, [_vp, _i]),
    "orbx_set_graph_replay": (_i, [_vp, _i]),
    "orbx_extract": (_i, [_vp, _vp, _i, _i, _sz, _i, _i, _vp, _vp, _i, _ip]),
    "orbx_extract_batch_device": (_i, [_vp, _vp, _i, _i, _i, _sz, _sz, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "orbx_stream_create": (_vp, [_i]),
    "orbx_stream_destroy": (None, [_vp]),
    "orbx_comm_unique_id": (_i, [_vp]),
    "orbx_comm_create": (_i, [C.POINTER(_vp), _i, _i, _i, _vp]),
    "orbx_comm_destroy": (None, [_vp]),
    "orbx_comm_rank": (_i, [_vp]),
    "orbx_comm_size": (_i, [_vp]),
    "orbx_gather": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "orbx_host_alloc": (_vp, [_sz]),
    "orbx_host_free": (None, [_vp]),
    "orbx_set_async_depth": (_i, [_vp, _i]),
    "orbx_extract_async": (_i, [_vp, _vp, _i, _i, _i, _sz, _sz, _i, _i, _vp, _vp, _vp, _i, _ip]),
    "orbx_async_device_buffers": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "orbx_async_download": (_i, [_vp, _i, _vp, _vp, _sz]),
    "orbx_wait": (_i, [_vp, _i]),
    "orbx_undistort_keypoints": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp]),
    "orbx_undistort_keypoints_device": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "orbx_image_bounds": (_i, [_i, _i, _vp, _vp, _i, _vp]),
    "orbx_stereo_matches": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _f, _f, _vp, _vp]),
    "orbx_stereo_matches_batch_device": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp]),
    "orbx_pyramid_level": (_i, [_vp, _i, _i, _vp, _sz, _ip, _ip]),
    "orbx_set_pyramid_download": (_i, [_vp, _i]),
    "orbx_set_profiling": (None, [_vp, _i]),
    "orbx_get_stage_ms": (_i, [_vp, _vp]),
    "orbx_debug_blurred_level": (_i, [_vp, _i, _i, _vp, _sz, _ip, _ip]),
    "orbx_debug_level_candidates": (_i, [_vp, _i, _i, _vp, _i]),
    "orbx_debug_level_keypoints": (_i, [_vp, _i, _i, _vp, _i]),
    "orbx_debug_octree_times": (_i, [_vp, _vp, _i]),
    "orbx_debug_calib_read": (_i, [_vp, C.c_longlong, _i]),
    "orbx_debug_buffers": (_i, [_vp, _vp, _i]),
    "orbx_debug_fault": (_i, [_vp, _vp]),
    "orbx_debug_valu_calib": (_i, [_i, _i, _i, _vp]),
    "orbx_debug_valu_calib_name": (C.c_char_p, [_i]),
    "orbx_debug_lds_calib": (_i, [_i, _i, _i, _i, _vp]),
    "orbx_debug_octree_device": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _ip]),
    "orbx_debug_wave_sort": (_i, [_vp, _vp, _i, _i]),
    "orbx_debug_octree_sorts": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _ip, _vp, _i]),
    "orbm_create": (_i, [C.POINTER(_vp), _i]),
    "orbm_destroy": (None, [_vp]),
    "orbm_last_error": (C.c_char_p, []),
    "orbm_hamming": (_i, [_vp, _vp]),
    "orbm_three_maxima": (None, [_vp, _i, _ip, _ip, _ip]),
    "orbm_match_groups": (_i, [_vp, _vp, _i, _vp, _i, _i] + [_vp] * 8 + [_i, _i, _f, _i, _vp, _vp]),
    "orbm_match_groups_two_cameras": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _vp]),
    "orbm_match_bruteforce_device": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "orbm_match_pairs_device": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "orbm_match_pairs_bow_device": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp]),
    "orbm_debug_counters": (_i, [_vp, _i, _vp]),
    "orbm_match_window": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _f, _f, _f, _f, _i, _i, _f, _i, _vp]),
    "orbm_search_by_projection": (_i, [_vp] * 7 + [_i] + [_vp] * 4 + [_i, _vp, _i, _f, _i, _f, _f, _f, _f, _f, _i, _i, _vp, _vp]),
    "orbm_search_by_projection_map": (_i, [_vp] * 7 + [_i] + [_vp] * 4 + [_i, _vp, _i, _f, _f, _f, _f, _f, _i, _vp, _vp]),
    "orbm_search_by_projection_map_two_cameras": (_i, [_vp, _vp, _i] + [_vp] * 10 + [_i, _i] + [_vp] * 4 + [_i, _f, _f, _f, _f, _f, _i, _vp]),
    "orbm_search_in_windows": (_i, [_vp] * 6 + [_i] + [_vp] * 4 + [_i, _vp, _vp, _i, _f, _i, _f, _f, _f, _f, _i, _vp]),
    "orbm_search_for_triangulation": (_i, [_vp] * 5 + [_i] + [_vp] * 4 + [_i, _i] + [_vp] * 8 + [_i] * 5 + [_vp]),
    "orbm_assign_features_to_grid_device": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "orbm_search_by_projection_batch_device": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _f, _f, _f, _f, _f, _i, _i, _vp, _vp, _vp, _vp]),
    "orbv_create": (_i, [C.POINTER(_vp), _i, _i, _i, _vp, _vp, _vp, _vp, _i]),
    "orbv_load_text": (_i, [C.POINTER(_vp), C.c_char_p, _i]),
    "orbv_destroy": (None, [_vp]),
    "orbv_last_error": (C.c_char_p, []),
    "orbv_info": (_i, [_vp, _ip, _ip, _ip, _ip]),
    "orbv_transform": (_i, [_vp, _vp, _i, _i, _vp, _vp, _ip, _vp, _vp, _vp, _ip]),
    "orbv_transform_batch_device": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
}
# debug hooks that run host builds of the shared host/device cores (no GPU needed)
HOST_MODEL_SIGNATURES = {
    "orbx_debug_octree_host": (_i, [_vp, _i, _i, _i, _i, _vp, _ip]),
    "orbx_debug_std_sort": (None, [_vp, _i, _i]),
    "orbx_debug_sincos_deg": (None, [_vp, _i, _vp, _vp]),
}


def lib():
    """Load (building first if needed) liborbx.so."""
    global _lib
    if _lib is None:
        path = _build.build()
        L = C.CDLL(path)
        for table in (SIGNATURES, HOST_MODEL_SIGNATURES):
            for name, (res, args) in table.items():
                fn = getattr(L, name)  # AttributeError if the library does not export it
                fn.restype = res
                fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    return lib().orbx_last_error().decode()


class OrbxError(RuntimeError):
    pass


def check(rc, what):
    if rc <= -2:
        raise OrbxError("%s failed (%d): %s" % (what, rc, last_error()))
    return rc


def ptr(a):
    """void* of a numpy array, a torch tensor (data_ptr) or an int."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(a.data_ptr())
