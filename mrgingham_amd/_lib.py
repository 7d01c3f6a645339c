"""ctypes binding of libmrgingham_amd.so (the C-ABI in include/mrgingham_amd.h).

The library is HIP-only.  If it is missing or cannot be loaded this module
raises -- there is no CPU implementation to fall back to.
"""
import ctypes
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRGINGHAM_AMD_LIB") or os.path.join(_HERE, "libmrgingham_amd.so")  # override: A/B builds

ADD_POINTS_F64 = ctypes.CFUNCTYPE(ctypes.c_bool, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_void_p)
ADD_POINTS_INT = ctypes.CFUNCTYPE(ctypes.c_bool, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_double,
                                  ctypes.c_void_p)


class Frames(ctypes.Structure):
    """mrgingham_amd_frames"""
    _fields_ = [("frames", ctypes.c_void_p), ("frame_pitch", ctypes.c_int64), ("nframes", ctypes.c_int),
                ("width", ctypes.c_int), ("height", ctypes.c_int), ("stride", ctypes.c_int)]


# every symbol include/mrgingham_amd.h declares
EXPORTS = [
    "mrgingham_ChESS_response_5", "find_chessboard_corners_from_image_array_C",
    "refine_chessboard_corners_from_image_array_C", "find_chessboard_from_image_array_C",
    "mrgingham_amd_find_grid_from_points", "mrgingham_amd_find_grid_from_points_traced", "mrgingham_amd_find_grid_from_points_perturbed", "mrgingham_amd_find_boards_stats", "mrgingham_amd_grid_clock", "mrgingham_amd_packed_layout", "mrgingham_amd_gather_rccl", "mrgingham_amd_create", "mrgingham_amd_destroy",
    "mrgingham_amd_last_error", "mrgingham_amd_abi_version", "mrgingham_amd_device_count", "mrgingham_amd_level_dims",
    "mrgingham_amd_chess_response_batch", "mrgingham_amd_decimate_batch", "mrgingham_amd_box_blur_batch",
    "mrgingham_amd_preprocess_batch", "mrgingham_amd_process_image", "mrgingham_amd_process_image_ex", "mrgingham_amd_preprocess_image16", "mrgingham_amd_preprocess_image",
    "find_chessboard_corners_from_image_file_C", "find_chessboard_from_image_file_C",
    "mrgingham_amd_detect_batch", "mrgingham_amd_refine_batch", "mrgingham_amd_chain_batch",
    "mrgingham_amd_find_boards_batch", "mrgingham_amd_cc_on_response_batch", "mrgingham_amd_scratch_bytes", "mrgingham_amd_chain_info", "mrgingham_amd_debug_refine_clock", "mrgingham_amd_debug_paths", "mrgingham_amd_read_image",
    "mrgingham_amd_set_option", "mrgingham_amd_sync", "mrgingham_amd_stream_wait", "mrgingham_amd_after_stream", "mrgingham_amd_set_kernel_timing",
    "mrgingham_amd_chess_kernel_ms", "mrgingham_amd_sclk_mhz", "mrgingham_amd_sparse_fallbacks", "mrgingham_amd_find_boards_submit",
    "mrgingham_amd_find_boards_collect", "mrgingham_amd_device_for_thread", "mrgingham_amd_set_thread_device",
    "mrgingham_amd_thread_device", "mrgingham_amd_host_alloc", "mrgingham_amd_host_free", "mrgingham_amd_host_register",
    "mrgingham_amd_host_unregister", "mrgingham_amd_shard_range", "mrgingham_amd_chain_multi", "mrgingham_amd_sync_multi",
    "mrgingham_amd_stream_wait_multi", "mrgingham_amd_kernel_id", "mrgingham_amd_set_wait_policy",
]


def build(force=False):
    """hipcc --offload-arch=gfx950 build of the shared library, in-tree."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", src, "clean"], stdout=sys.stderr)
    subprocess.check_call(["make", "-s", "-C", src, "-j4", "all"], stdout=sys.stderr)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch first: it brings its own libamdhip64, and device pointers / streams are
    # shared between torch and this library, so both must sit on ONE HIP runtime.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C mrgingham_amd/csrc` "
                           "(hipcc, gfx950).  mrgingham_amd has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    c_int, c_vp, c_bool = ctypes.c_int, ctypes.c_void_p, ctypes.c_bool
    FP = ctypes.POINTER(Frames)
    L.mrgingham_ChESS_response_5.argtypes = [c_vp, c_vp, c_int, c_int, c_int]
    L.mrgingham_ChESS_response_5.restype = None
    L.find_chessboard_corners_from_image_array_C.argtypes = [c_int, c_int, c_int, c_vp, c_int, c_bool, c_bool,
                                                             ADD_POINTS_INT, c_vp]
    L.find_chessboard_corners_from_image_array_C.restype = c_bool
    L.refine_chessboard_corners_from_image_array_C.argtypes = [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_int,
                                                               c_bool]
    L.refine_chessboard_corners_from_image_array_C.restype = c_int
    L.find_chessboard_from_image_array_C.argtypes = [c_int, c_int, c_int, c_vp, c_int, c_int, c_bool, c_bool, c_int,
                                                     c_int, ADD_POINTS_F64, c_vp]
    L.find_chessboard_from_image_array_C.restype = c_bool
    L.mrgingham_amd_find_grid_from_points.argtypes = [c_vp, c_int, c_int, c_vp]
    L.mrgingham_amd_find_grid_from_points.restype = c_bool
    L.mrgingham_amd_find_grid_from_points_traced.argtypes = [c_vp, c_int, c_int, c_vp, c_int, c_int, c_int]
    L.mrgingham_amd_find_grid_from_points_traced.restype = c_bool
    L.mrgingham_amd_find_grid_from_points_perturbed.argtypes = [c_vp, c_int, c_int, c_vp, ctypes.c_uint, c_int]
    L.mrgingham_amd_find_grid_from_points_perturbed.restype = c_bool
    L.mrgingham_amd_find_boards_stats.argtypes = [c_vp, c_vp, c_int, c_int]
    L.mrgingham_amd_grid_clock.argtypes = [c_vp, c_int]
    L.mrgingham_amd_packed_layout.argtypes = [c_int, c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    L.mrgingham_amd_gather_rccl.argtypes = [c_vp, c_vp, c_int, c_vp, ctypes.c_size_t, c_vp, c_vp]
    L.mrgingham_amd_create.argtypes = [c_int]
    L.mrgingham_amd_create.restype = c_vp
    L.mrgingham_amd_destroy.argtypes = [c_vp]
    L.mrgingham_amd_destroy.restype = None
    L.mrgingham_amd_last_error.argtypes = [c_vp]
    L.mrgingham_amd_last_error.restype = ctypes.c_char_p
    L.mrgingham_amd_abi_version.restype = c_int
    L.mrgingham_amd_kernel_id.restype = ctypes.c_char_p
    L.mrgingham_amd_device_count.restype = c_int
    L.mrgingham_amd_level_dims.argtypes = [c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    L.mrgingham_amd_chess_response_batch.argtypes = [c_vp, FP, c_int, c_int, c_vp, c_vp]
    L.mrgingham_amd_decimate_batch.argtypes = [c_vp, FP, c_int, c_vp, c_vp]
    L.mrgingham_amd_box_blur_batch.argtypes = [c_vp, FP, c_int, c_vp, c_vp]
    L.mrgingham_amd_preprocess_batch.argtypes = [c_vp, FP, c_int, c_int, c_vp, c_vp]
    L.mrgingham_amd_detect_batch.argtypes = [c_vp, FP, c_int, c_vp, c_int, c_vp]
    L.mrgingham_amd_refine_batch.argtypes = [c_vp, FP, c_int, c_vp, c_vp, c_vp, c_int, c_vp]
    L.mrgingham_amd_chain_batch.argtypes = [c_vp, FP, c_int, c_vp, c_vp, c_vp, c_int]
    L.mrgingham_amd_find_boards_batch.argtypes = [c_vp, FP, c_int, c_int, c_vp, c_vp, c_int]
    L.mrgingham_amd_find_boards_submit.argtypes = [c_vp, FP, c_int, c_int, c_vp, c_vp, c_int]
    L.mrgingham_amd_find_boards_collect.argtypes = [c_vp, c_int]
    L.mrgingham_amd_device_for_thread.argtypes = [c_int, c_int, ctypes.c_char_p]
    L.mrgingham_amd_set_thread_device.argtypes = [c_int]
    L.mrgingham_amd_host_alloc.argtypes = [ctypes.c_size_t]
    L.mrgingham_amd_host_alloc.restype = c_vp
    L.mrgingham_amd_host_free.argtypes = [c_vp]
    L.mrgingham_amd_host_free.restype = None
    L.mrgingham_amd_host_register.argtypes = [c_vp, ctypes.c_size_t]
    L.mrgingham_amd_host_unregister.argtypes = [c_vp]
    L.mrgingham_amd_set_wait_policy.argtypes = [c_int]
    L.mrgingham_amd_shard_range.argtypes = [c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    L.mrgingham_amd_chain_multi.argtypes = [ctypes.POINTER(c_vp), c_int, FP, c_int, c_vp, c_vp, c_vp, c_int]
    L.mrgingham_amd_sync_multi.argtypes = [ctypes.POINTER(c_vp), c_int]
    L.mrgingham_amd_stream_wait_multi.argtypes = [ctypes.POINTER(c_vp), c_int, c_vp]
    L.mrgingham_amd_cc_on_response_batch.argtypes = [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp,
                                                     c_vp, c_vp, c_vp, c_int, c_vp]
    L.mrgingham_amd_read_image.argtypes = [ctypes.c_char_p, c_int, c_vp, ctypes.c_size_t, ctypes.POINTER(c_int),
                                           ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    L.mrgingham_amd_debug_paths.argtypes = [c_vp, c_int, c_int, c_vp]
    L.mrgingham_amd_chain_info.argtypes = [c_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.mrgingham_amd_debug_refine_clock.argtypes = [c_vp, c_vp]
    L.mrgingham_amd_scratch_bytes.argtypes = [c_vp]
    L.mrgingham_amd_scratch_bytes.restype = ctypes.c_longlong
    L.mrgingham_amd_set_option.argtypes = [c_vp, ctypes.c_char_p, c_int]
    L.mrgingham_amd_sync.argtypes = [c_vp]
    L.mrgingham_amd_sparse_fallbacks.argtypes = [c_vp]
    L.mrgingham_amd_stream_wait.argtypes = [c_vp, c_vp]
    L.mrgingham_amd_after_stream.argtypes = [c_vp, c_vp]
    L.mrgingham_amd_set_kernel_timing.argtypes = [c_vp, c_int]
    L.mrgingham_amd_set_kernel_timing.restype = None
    L.mrgingham_amd_chess_kernel_ms.argtypes = [c_vp, ctypes.POINTER(c_int)]
    L.mrgingham_amd_chess_kernel_ms.restype = ctypes.c_double
    L.mrgingham_amd_sclk_mhz.argtypes = [c_vp]
    L.mrgingham_amd_sclk_mhz.restype = ctypes.c_double
    _lib = L
    return L
