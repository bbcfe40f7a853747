"""ctypes binding of include/odometry_hip.h (the in-tree libodometry_hip.so).

Fails loudly when the library is missing or cannot be loaded — there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ODOMETRY_HIP_LIB") or os.path.join(_HERE, "lib", "libodometry_hip.so")

MAX_LEVELS = 8
NACC = 29
PYR_IMAGE, PYR_DEPTH = 0, 1


class Intrinsics(C.Structure):
    _fields_ = [("f0", C.c_float), ("cx0", C.c_float), ("cy0", C.c_float)]


class LmTraceRow(C.Structure):
    _fields_ = [("level", C.c_int), ("iter", C.c_int), ("n_res", C.c_int), ("accepted", C.c_int), ("stop", C.c_int),
                ("err", C.c_float), ("lambda_after", C.c_float), ("delta", C.c_float * 6)]


class TrackerParams(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("levels", C.c_int),
                ("lm_lambda", C.c_float), ("lm_precision", C.c_float), ("lm_max_iters", C.c_int * MAX_LEVELS),
                ("lm_robust", C.c_int), ("lm_huber_delta", C.c_float),
                ("grad_th", C.c_float), ("ssd_th", C.c_float), ("photo_th", C.c_float),
                ("min_depth", C.c_float), ("max_depth", C.c_float),
                ("depth_lambda", C.c_float), ("depth_huber_delta", C.c_float), ("depth_precision", C.c_float),
                ("depth_max_iters", C.c_int), ("boundary", C.c_int), ("max_residuals", C.c_int),
                ("max_disparity", C.c_int), ("any_size", C.c_int),
                ("K", Intrinsics), ("baseline", C.c_float),
                ("keyframe_weight", C.c_float * 6), ("keyframe_motion_th", C.c_float),
                ("smooth_image", C.c_int), ("overlap_depth", C.c_int)]


_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/odometry_hip.h declares.
SIGNATURES = {
    "odo_last_error": (C.c_char_p, []),
    "odo_version": (C.c_int, []),
    "odo_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "odo_ctx_create_high_priority": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "odo_ctx_destroy": (C.c_int, [_vp]),
    "odo_ctx_synchronize": (C.c_int, [_vp]),
    "odo_ctx_timer_start": (C.c_int, [_vp]),
    "odo_ctx_timer_stop": (C.c_int, [_vp, _fp]),
    "odo_dev_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "odo_dev_free": (C.c_int, [_vp, _vp]),
    "odo_dev_upload": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "odo_dev_download": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "odo_host_alloc": (_vp, [C.c_size_t]),
    "odo_host_free": (None, [_vp]),
    "odo_dev_upload_async": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "odo_dev_upload_2d_async": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_size_t, C.c_int]),
    "odo_ctx_wait_mark": (C.c_int, [_vp, C.c_ulong]),
    "odo_ctx_mark_reached": (C.c_int, [_vp, C.c_ulong]),
    "odo_depth_compact_bytes": (C.c_size_t, []),
    "odo_depth_compact_outputs_async": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "odo_host_scatter_outputs": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_ulonglong)]),
    "odo_host_scatter_outputs_prezeroed": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_ulonglong)]),
    "odo_host_fingerprint": (C.c_ulonglong, [_vp, C.c_size_t, C.c_size_t, C.c_int]),
    "odo_host_copy_fingerprint": (C.c_ulonglong, [_vp, C.c_size_t, _vp, C.c_size_t, C.c_size_t, C.c_int]),
    "odo_dev_upload_fp_async": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_ulonglong)]),
    "odo_dev_download_async": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "odo_dev_alloc_async": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp), C.POINTER(C.c_int)]),
    "odo_dev_free_async": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int]),
    "odo_pyramid_create": (C.c_int, [_vp, _fp, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "odo_pyramid_create_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "odo_pyramid_rebuild_dev": (C.c_int, [_vp, _vp, C.c_int]),
    "odo_pyramid_levels": (C.c_int, [_vp]),
    "odo_pyramid_level_dims": (C.c_int, [_vp, C.c_int, _ip, _ip]),
    "odo_pyramid_download": (C.c_int, [_vp, C.c_int, _fp]),
    "odo_pyramid_level_dev": (_vp, [_vp, C.c_int]),
    "odo_pyramid_destroy": (C.c_int, [_vp]),
    "odo_lm_create": (C.c_int, [_vp, C.c_float, C.c_float, _ip, C.c_int, _fp, C.c_int, C.c_float,
                                C.POINTER(Intrinsics), C.POINTER(_vp)]),
    "odo_lm_solve": (C.c_int, [_vp, _vp, _vp, _vp, _fp]),
    "odo_lm_reset": (C.c_int, [_vp, _fp, C.c_float]),
    "odo_lm_report": (C.c_int, [_vp, _ip, _fp]),
    "odo_lm_destroy": (C.c_int, [_vp]),
    "odo_lm_accumulate": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _fp, _dp]),
    "odo_lm_trace": (C.c_int, [_vp, C.POINTER(LmTraceRow), C.c_int, _ip]),
    "odo_lm_set_record": (C.c_int, [_vp, C.c_int]),
    "odo_lm_time_eval": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _fp, C.c_int, _fp, _fp, _dp, _ip]),
    "odo_lm_time_eval_batch": (C.c_int, [C.c_int, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.c_int, _fp, C.c_int, _fp, _fp, _dp, _ip]),
    "odo_lm_event_timing": (C.c_int, [_vp, C.c_int]),
    "odo_lm_event_stats": (C.c_int, [_vp, _dp, C.POINTER(C.c_long), C.POINTER(C.c_long), _dp]),
    "odo_lm_event_stats2": (C.c_int, [_vp, _dp, C.POINTER(C.c_long)]),
    "odo_lm_event_stats_ex": (C.c_int, [_vp, _dp]),
    "odo_lm_solve_begin": (C.c_int, [_vp, _vp, _vp, _vp]),
    "odo_lm_set_idle_callback": (C.c_int, [_vp, _vp, _vp]),
    "odo_lm_candidate_begin": (C.c_int, [_vp, _vp, _vp, _vp, C.c_ulong]),
    "odo_lm_solve_batch": (C.c_int, [C.c_int, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _fp, C.POINTER(C.c_int)]),
    "odo_lm_set_sampling": (C.c_int, [_vp, C.c_int]),
    "odo_lm_set_mode": (C.c_int, [_vp, C.c_int]),
    "odo_lm_points": (C.c_int, [_vp, _ip, _ip]),
    "odo_lm_launch_stats": (C.c_int, [_vp, _ip, _ip, _dp]),
    "odo_lm_persistent_stats": (C.c_int, [_vp, _ip, _ip]),
    "odo_lm_persistent_backoff": (C.c_int, [_vp, _ip, _ip, _ip]),
    "odo_lm_tdist_stats": (C.c_int, [_vp, C.POINTER(C.c_long), _ip]),
    "odo_debug_update_stamps": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _fp, C.POINTER(C.c_ulonglong)]),
    "odo_debug_solve": (C.c_int, [_vp, _dp, C.c_float, _fp]),
    "odo_depth_create": (C.c_int, [_vp] + [C.c_float] * 8 + [C.c_int, C.c_int, C.POINTER(Intrinsics), C.c_float,
                                                          C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "odo_depth_compute": (C.c_int, [_vp, _fp, _fp, C.c_int, C.c_int, _u8p, _fp, _fp]),
    "odo_depth_compute_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "odo_depth_prepare_left_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_ulonglong]),
    "odo_depth_prepare_left_dev_marked": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_ulonglong, C.c_ulong]),
    "odo_depth_compute_dev_stamped": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_ulonglong]),
    "odo_depth_compute_begin_dev": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_ulonglong, C.c_ulonglong, C.c_ulong]),
    "odo_depth_compute_end_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_ulonglong, C.c_ulonglong]),
    "odo_depth_early_pending": (C.c_int, [_vp]),
    "odo_depth_disparity": (C.c_int, [_vp, _fp, _fp, C.c_int, C.c_int, _u8p, _fp, _fp]),
    "odo_depth_time_stages": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _fp, _dp, _ip]),
    "odo_depth_report": (C.c_int, [_vp, _ip, _fp, _ip, _ip, _ip]),
    "odo_depth_persistent_stats": (C.c_int, [_vp, _ip, _ip]),
    "odo_depth_destroy": (C.c_int, [_vp]),
    "odo_tracker_default_params": (C.c_int, [C.POINTER(TrackerParams)]),
    "odo_tracker_create": (C.c_int, [C.c_int, C.POINTER(TrackerParams), C.POINTER(_vp)]),
    "odo_tracker_init": (C.c_int, [_vp, _vp, _vp, _fp]),
    "odo_tracker_track": (C.c_int, [_vp, _vp, _vp, _fp, _fp, _ip, _fp, _ip]),
    "odo_tracker_hint_next": (C.c_int, [_vp, _vp]),
    "odo_tracker_hint_next_pair": (C.c_int, [_vp, _vp, _vp]),
    "odo_tracker_stats": (C.c_int, [_vp, _ip, _ip, _ip, _ip]),
    "odo_tracker_outputs": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "odo_tracker_time_residual": (C.c_int, [_vp, C.c_int, C.c_int, _fp, _fp, _dp, _ip]),
    "odo_tracker_timing": (C.c_int, [_vp, _dp]),
    "odo_tracker_lm": (_vp, [_vp]),
    "odo_tracker_chain_stats": (C.c_int, [_vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "odo_tracker_arm_stats": (C.c_int, [_vp, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "odo_tracker_depth": (C.c_void_p, [_vp]),
    "odo_tracker_batch_lm": (_vp, [_vp, C.c_int]),
    "odo_tracker_ctx": (_vp, [_vp]),
    "odo_tracker_destroy": (C.c_int, [_vp]),
    "odo_ctx_upload_ticket": (C.c_ulong, [_vp]),
    "odo_ctx_upload_wait": (C.c_int, [_vp, C.c_ulong]),
    "odo_ctx_stream_wait": (C.c_int, [_vp, _vp]),
    "odo_ctx_mark": (C.c_ulong, [_vp]),
    "odo_ctx_stream_wait_mark": (C.c_int, [_vp, _vp, C.c_ulong]),
    "odo_tracker_quiesce": (C.c_int, [_vp]),
    "odo_gather_unique_id": (C.c_int, [C.POINTER(C.c_ubyte)]),
    "odo_gather_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "odo_gather_push": (C.c_int, [_vp, C.c_int, C.c_int, _fp]),
    "odo_gather_flush": (C.c_int, [_vp]),
    "odo_gather_rows": (C.c_int, [_vp, C.c_int, C.POINTER(_fp), _ip]),
    "odo_gather_issued": (C.c_int, [_vp]),
    "odo_gather_ranks": (C.c_int, [_vp]),
    "odo_gather_destroy": (C.c_int, [_vp]),
    "odo_tracker_batch_create": (C.c_int, [C.c_int, C.POINTER(TrackerParams), C.c_int, C.POINTER(_vp)]),
    "odo_tracker_batch_destroy": (C.c_int, [_vp]),
    "odo_tracker_batch_depth_persistent_stats": (C.c_int, [_vp, _ip, _ip]),
    "odo_tracker_batch_quiesce": (C.c_int, [_vp]),
    "odo_tracker_batch_size": (C.c_int, [_vp]),
    "odo_tracker_batch_ctx": (_vp, [_vp]),
    "odo_tracker_batch_init": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), _fp]),
    "odo_tracker_batch_init_one": (C.c_int, [_vp, C.c_int, _vp, _vp, _fp]),
    "odo_tracker_batch_hint_next": (C.c_int, [_vp, C.POINTER(_vp)]),
    "odo_tracker_batch_hint_next_pair": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "odo_tracker_batch_track": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), _fp, _fp, _ip, _fp, _ip]),
    "odo_tracker_batch_timing": (C.c_int, [_vp, _dp]),
    "odo_tracker_batch_stats": (C.c_int, [_vp, _ip, _ip, _ip, _ip]),
    "odo_tracker_batch_outputs": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "odo_camera_create": (C.c_int, [_vp, C.c_int] + [C.c_double] * 11 + [C.c_int, C.c_int, C.POINTER(_vp)]),
    "odo_camera_configure": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_int]),
    "odo_camera_levels": (C.c_int, [_vp]),
    "odo_camera_intrinsics": (C.c_int, [_vp, C.c_int, _dp]),
    "odo_camera_raw": (C.c_int, [_vp, _dp, _dp, _dp, _ip]),
    "odo_camera_map_size": (C.c_int, [_vp, _ip, _ip]),
    "odo_camera_download_maps": (C.c_int, [_vp, _fp, _fp]),
    "odo_camera_undistort_rectify": (C.c_int, [_vp, _fp, C.c_int, C.c_int, _fp, C.c_float]),
    "odo_camera_undistort_rectify_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_float]),
    "odo_camera_destroy": (C.c_int, [_vp]),
}

_lib = None


def load():
    """Loads the in-tree HIP library; raises if it is missing (build it with `python -m odometry_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m odometry_amd.build` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().odo_last_error().decode("utf-8", "replace")


class OdoError(RuntimeError):
    pass


def check(status, what):
    if status != 0:
        raise OdoError(f"{what}: {last_error()}")
