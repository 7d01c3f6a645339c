"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under mrgingham_amd/ does.  See mrgingham_oracle.c for the
reference citations and the parity-pin status of each function.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libchess_ref.so")

_u8p = ctypes.POINTER(ctypes.c_uint8)
_i16p = ctypes.POINTER(ctypes.c_int16)
_i32p = ctypes.POINTER(ctypes.c_int32)
_f64p = ctypes.POINTER(ctypes.c_double)
_i8p = ctypes.POINTER(ctypes.c_int8)


def build(force=False):
    """Compile the restatement (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(
            os.path.getmtime(os.path.join(_HERE, f)) for f in ("mrgingham_oracle.c", "blobs_oracle.c", "cpu_bench.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"], stdout=sys.stderr)
    elif not os.path.exists(_REF) and os.path.exists("/root/reference/ChESS.c"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"], stdout=sys.stderr)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB)
        L.oracle_chess_response_5.argtypes = [_i16p, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.oracle_chess_response_5.restype = None
        L.oracle_level_dims.argtypes = [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int)] * 2
        L.oracle_decimate.argtypes = [_u8p, _u8p] + [ctypes.c_int] * 4
        L.oracle_box_blur.argtypes = [_u8p, _u8p] + [ctypes.c_int] * 4
        L.oracle_box_blur.restype = None
        L.oracle_clamped_response.argtypes = [_i16p, _u8p, _u8p] + [ctypes.c_int] * 4
        L.oracle_find_corners.argtypes = [_i32p, ctypes.c_int, _u8p] + [ctypes.c_int] * 4
        L.oracle_refine_corners.argtypes = [_f64p, _i8p, ctypes.c_int, _u8p] + [ctypes.c_int] * 4
        L.oracle_cc_detect_on_response.argtypes = [_i32p, ctypes.c_int, _i16p, _u8p] + [ctypes.c_int] * 3
        L.oracle_cc_refine_on_response.argtypes = [_f64p, _i8p, ctypes.c_int, _i16p, _u8p] + [ctypes.c_int] * 3
        _lib = L
    return _lib


def have_reference_build():
    return os.path.exists(_REF)


def ref_lib():
    """The REAL upstream ChESS.c, compiled by oracle/Makefile into oracle/_ref/."""
    global _ref
    if _ref is None:
        R = ctypes.CDLL(_REF)
        R.mrgingham_ChESS_response_5.argtypes = [_i16p, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        R.mrgingham_ChESS_response_5.restype = None
        _ref = R
    return _ref


def _img2d(image):
    """Rows may be strided; the last axis must be dense (as the reference's
    Python wrapper demands, mrgingham_pywrap.c:63-68)."""
    image = np.asarray(image)
    assert image.dtype == np.uint8 and image.ndim == 2 and (image.shape[1] <= 1 or image.strides[1] == 1)
    return image, image.shape[0], image.shape[1], (image.strides[0] if image.shape[0] > 1 else image.shape[1])


def _chess(fn, image, fill):
    image, H, W, stride = _img2d(image)
    out = np.full((H, W), fill, dtype=np.int16)
    fn(out.ctypes.data_as(_i16p), image.ctypes.data_as(_u8p), W, H, stride)
    return out


def chess_response_5(image, fill=0):
    """Restatement of ChESS.c:56-106; untouched border pixels hold `fill`."""
    return _chess(lib().oracle_chess_response_5, image, fill)


def ref_chess_response_5(image, fill=0):
    """The upstream mrgingham_ChESS_response_5 itself (oracle/_ref)."""
    return _chess(ref_lib().mrgingham_ChESS_response_5, image, fill)


def level_dims(W, H, level):
    w, h = ctypes.c_int(), ctypes.c_int()
    if lib().oracle_level_dims(W, H, level, ctypes.byref(w), ctypes.byref(h)) != 0:
        raise ValueError("level out of range")
    return w.value, h.value


def decimate(image, level):
    image, H, W, stride = _img2d(image)
    w, h = level_dims(W, H, level)
    out = np.empty((h, w), dtype=np.uint8)
    if lib().oracle_decimate(out.ctypes.data_as(_u8p), image.ctypes.data_as(_u8p), W, H, stride, level) != 0:
        raise RuntimeError("oracle_decimate failed")
    return out


def box_blur(image, radius=1):
    image, H, W, stride = _img2d(image)
    out = np.empty((H, W), dtype=np.uint8)
    lib().oracle_box_blur(out.ctypes.data_as(_u8p), image.ctypes.data_as(_u8p), W, H, stride, radius)
    return out


def normalize_minmax(image):
    image, H, W, stride = _img2d(image)
    out = np.empty((H, W), dtype=np.uint8)
    lib().oracle_normalize_minmax(out.ctypes.data_as(_u8p), image.ctypes.data_as(_u8p), W, H, stride)
    return out


def clahe(image, clip_limit=8.0):
    image, H, W, stride = _img2d(image)
    out = np.empty((H, W), dtype=np.uint8)
    lib().oracle_clahe.argtypes = [_u8p, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    if lib().oracle_clahe(out.ctypes.data_as(_u8p), image.ctypes.data_as(_u8p), W, H, stride, float(clip_limit)) != 0:
        raise RuntimeError("oracle_clahe failed")
    return out


def preprocess(image, clahe=True, blur_radius=1):
    """mrgingham-from-image.cc:71-111 for an 8-bit frame."""
    image, H, W, stride = _img2d(image)
    out = np.empty((H, W), dtype=np.uint8)
    if lib().oracle_preprocess(out.ctypes.data_as(_u8p), image.ctypes.data_as(_u8p), W, H, stride,
                               int(bool(clahe)), int(blur_radius)) != 0:
        raise RuntimeError("oracle_preprocess failed")
    return out


def preprocess16(image16, clahe=True, blur_radius=1):
    """mrgingham-from-image.cc:85-111 for a 16-bit frame -> the uint8 image the detector sees."""
    image16 = np.ascontiguousarray(image16, dtype=np.uint16)
    H, W = image16.shape
    out = np.empty((H, W), dtype=np.uint8)
    L = lib()
    L.oracle_preprocess16.argtypes = [_u8p, ctypes.POINTER(ctypes.c_uint16)] + [ctypes.c_int] * 5
    if L.oracle_preprocess16(out.ctypes.data_as(_u8p), image16.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), W, H, W,
                             int(bool(clahe)), int(blur_radius)) != 0:
        raise RuntimeError("oracle_preprocess16 failed")
    return out


def clamped_response(image, level):
    image, H, W, stride = _img2d(image)
    w, h = level_dims(W, H, level)
    resp = np.empty((h, w), dtype=np.int16)
    img = np.empty((h, w), dtype=np.uint8)
    if lib().oracle_clamped_response(resp.ctypes.data_as(_i16p), img.ctypes.data_as(_u8p),
                                     image.ctypes.data_as(_u8p), H, W, stride, level) != 0:
        raise RuntimeError("oracle_clamped_response failed")
    return resp, img


def find_corners(image, level):
    """-> int32 (N,2) array of (x,y)*1000 in reference order, or None on the
    reference's error paths (bad level / non-continuous level-0 input)."""
    image, H, W, stride = _img2d(image)
    cap = 4096
    while True:
        out = np.empty((cap, 2), dtype=np.int32)
        n = lib().oracle_find_corners(out.ctypes.data_as(_i32p), cap, image.ctypes.data_as(_u8p), H, W, stride,
                                      level)
        if n < 0:
            return None
        if n <= cap:
            return out[:n].copy()
        cap = n


def refine_corners(points, levels, image, level):
    """points float64 (N,2), levels int8 (N,), both updated copies returned with
    the refined count."""
    image, H, W, stride = _img2d(image)
    pts = np.ascontiguousarray(points, dtype=np.float64).copy()
    lv = np.ascontiguousarray(levels, dtype=np.int8).copy()
    n = lib().oracle_refine_corners(pts.ctypes.data_as(_f64p), lv.ctypes.data_as(_i8p), len(lv),
                                    image.ctypes.data_as(_u8p), H, W, stride, level)
    return pts, lv, n


def cc_detect_on_response(resp, level_image, level=0):
    d = np.ascontiguousarray(resp, dtype=np.int16).copy()
    img = np.ascontiguousarray(level_image, dtype=np.uint8)
    h, w = d.shape
    cap = 4096
    while True:
        dd = d.copy()
        out = np.empty((cap, 2), dtype=np.int32)
        n = lib().oracle_cc_detect_on_response(out.ctypes.data_as(_i32p), cap, dd.ctypes.data_as(_i16p),
                                               img.ctypes.data_as(_u8p), w, h, level)
        if n <= cap:
            return out[:n].copy()
        cap = n


def cc_refine_on_response(points, levels, resp, level_image, level):
    d = np.ascontiguousarray(resp, dtype=np.int16).copy()
    img = np.ascontiguousarray(level_image, dtype=np.uint8)
    h, w = d.shape
    pts = np.ascontiguousarray(points, dtype=np.float64).copy()
    lv = np.ascontiguousarray(levels, dtype=np.int8).copy()
    n = lib().oracle_cc_refine_on_response(pts.ctypes.data_as(_f64p), lv.ctypes.data_as(_i8p), len(lv),
                                           d.ctypes.data_as(_i16p), img.ctypes.data_as(_u8p), w, h, level)
    return pts, lv, n


def find_blobs(image):
    """find_blobs_from_image_array (find_blobs.cc:14-46) -> int32 (N,2) of (x,y)*1000 in keypoint order."""
    image, H, W, stride = _img2d(image)
    L = lib()
    L.oracle_find_blobs.argtypes = [_i32p, ctypes.c_int, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    cap = 4096
    while True:
        out = np.empty((cap, 2), dtype=np.int32)
        n = L.oracle_find_blobs(out.ctypes.data_as(_i32p), cap, image.ctypes.data_as(_u8p), W, H, stride)
        if n <= cap:
            return out[:n].copy()
        cap = n


def chain(image, start_level=3):
    """The reference's found-frame schedule without the grid finder
    (mrgingham.cc:50, :81-99): detect at `start_level`, then refine every
    candidate through start_level-1 .. 0.  Returns (points f64 (N,2), levels i8)."""
    cand = find_corners(image, start_level)
    if cand is None:
        return None
    pts = cand.astype(np.float64) / 1000.0
    lv = np.full(len(pts), start_level, dtype=np.int8)
    for L in range(start_level - 1, -1, -1):
        pts, lv, n = refine_corners(pts, lv, image, L)
        if n <= 0:
            break
    return pts, lv


def bench_chain(frames, start_level=3, nthreads=1, min_seconds=5.0):
    """pthread harness (cpu_bench.c): `nthreads` workers, one frame per worker at a time, whole chain per frame,
    for at least `min_seconds`.  frames: uint8 [n, H, W] contiguous.  -> (passes, elapsed_s, candidates_seen)."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n, H, W = frames.shape
    L = lib()
    L.oracle_bench_chain.argtypes = [_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]
    L.oracle_bench_chain.restype = ctypes.c_long
    el, pts = ctypes.c_double(0), ctypes.c_long(0)
    passes = L.oracle_bench_chain(frames.ctypes.data_as(_u8p), n, H, W, start_level, nthreads, float(min_seconds),
                                  ctypes.byref(el), ctypes.byref(pts))
    return int(passes), float(el.value), int(pts.value)


def bench_ref_chess(frames, nthreads=1, min_seconds=5.0):
    """The REAL upstream ChESS.c (oracle/_ref), level-0 response only, in the same harness.
    -> (passes, elapsed_s)."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n, H, W = frames.shape
    L = lib()
    L.oracle_bench_fn.argtypes = [ctypes.c_void_p, _u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    L.oracle_bench_fn.restype = ctypes.c_long
    fn = ctypes.cast(ref_lib().mrgingham_ChESS_response_5, ctypes.c_void_p)
    el = ctypes.c_double(0)
    passes = L.oracle_bench_fn(fn, frames.ctypes.data_as(_u8p), n, H, W, nthreads, float(min_seconds), ctypes.byref(el))
    return int(passes), float(el.value)


def bench_heap_reuse(on):
    """Allocator policy for the bench harness: see oracle_bench_heap_reuse in cpu_bench.c."""
    lib().oracle_bench_heap_reuse(1 if on else 0)
